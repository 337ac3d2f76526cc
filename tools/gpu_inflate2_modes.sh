#!/bin/bash
# the inflate harness on adversarial texts: runs / periodic patterns, noise (token regions overflow: the single kernel takes over), a
# mixture; then at a token fraction that makes SOME bedMethyl blocks overflow
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/inflate2_modes
for spec in "8192 1024 1 0.625 1" "8192 1024 1 0.625 2" "8192 1024 1 0.625 3" "8192 1024 1 0.40 0" "8192 8192 1 0.625 3"; do
  timeout 600 ./tools/inflate2_proto $spec > gpurun_out/inflate2_modes/m.log 2>&1; echo "== $spec rc=$?"
  grep -E "slab:|IDENTICAL|WRONG|tokens per block|phase" gpurun_out/inflate2_modes/m.log | cut -c1-220
done

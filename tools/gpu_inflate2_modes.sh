#!/bin/bash
# the inflate harness on adversarial texts: runs / periodic patterns, noise (token regions overflow: the single kernel takes over), a
# mixture; then at a token fraction that makes SOME bedMethyl blocks overflow
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/inflate2_modes
rm -f gpurun_out/inflate2_modes/all.log
for spec in "8192 1024 1 0.625 1" "8192 1024 1 0.625 2" "8192 1024 1 0.625 3" "8192 1024 1 0.40 0" "8192 8192 1 0.625 3"; do
  timeout 600 ./tools/inflate2_proto $spec > gpurun_out/inflate2_modes/m.log 2>&1; echo "== $spec rc=$?"
  { echo "== $spec"; grep -E "slab:|IDENTICAL|WRONG|differ|tokens per block|phase" gpurun_out/inflate2_modes/m.log | cut -c1-220; } | tee -a gpurun_out/inflate2_modes/all.log
done
echo "summary: $(grep -c IDENTICAL gpurun_out/inflate2_modes/all.log) IDENTICAL, $(grep -c -E "WRONG|[1-9][0-9]* of [0-9]+ blocks differ" gpurun_out/inflate2_modes/all.log) wrong"

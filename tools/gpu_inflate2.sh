#!/bin/bash
# the stand-alone inflate harness (tools/inflate2_proto.hip) on one box: every kernel timed on a 3 GiB slab, text checked
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/inflate2
timeout 600 ./tools/inflate2_proto ${1:-49152} ${2:-1024} ${3:-3} ${4:-0.625} ${5:-0} > gpurun_out/inflate2/proto.log 2>&1
echo "rc=$?" >> gpurun_out/inflate2/proto.log
cat gpurun_out/inflate2/proto.log

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/inflate
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/inflate_proto tools/inflate_proto.hip -lz || exit 1
{ timeout 600 /tmp/inflate_proto 64 6; timeout 600 /tmp/inflate_proto 1024 6; timeout 600 /tmp/inflate_proto 1024 1; } 2>&1 | tee gpurun_out/inflate/out2.txt

#!/bin/bash
# round 4: the LDS-staged transposed tile prototype (tools/lds_proto.hip): correctness on a small input, then the 1 Gbp shape
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/lds_proto
{
for v in "-DHALO=16 -DBOTH=0 -DWAVES=5" "-DHALO=32 -DBOTH=0 -DWAVES=4"; do
echo "== $v"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-int-to-pointer-cast $v -o /tmp/lds_proto tools/lds_proto.hip || exit 1
timeout 300 /tmp/lds_proto 66 10 1 | head -1
for f in 0 3; do
for c in 4 10 16; do
timeout 300 /tmp/lds_proto 122072 $c 0 $f
done
done
done
} 2>&1 | tee gpurun_out/lds_proto/out4.txt

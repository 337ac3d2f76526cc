#!/bin/bash
# round 4: match copies of the device inflate in 8-byte words with batched loads: prototype (checked against zlib), tests, timing
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4i
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/inflate_proto tools/inflate_proto.hip -lz && { timeout 600 /tmp/inflate_proto 1024 6; timeout 600 /tmp/inflate_proto 256 1; } 2>&1 | grep -E "device|probe" | tee gpurun_out/r4i/inflate_probes.txt
timeout 900 python -m pytest tests/test_gpu_bed_device.py tests/test_gpu_cli.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_r4c.sh 2>&1 | grep -E "^\[bed\] slab|rep|plain" | head -12
bash tools/gpu_cli_gz.sh 2>&1 | grep -E "^device|^gz_device|gz_over"

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_gpu_bed_device.py -x -q -m gpu 2>&1 | tail -3
export TMPDIR=/tmp
rm -rf /tmp/pb
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/pb -- python3 tools/bed_probe.py 20000000 5 > gpurun_out/r3o/probe.log 2>&1
grep "device parse\|bed " gpurun_out/r3o/probe.log
for f in $(find /tmp/pb -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do head -5 $f | cut -c1-120; done
timeout 900 python tools/cli_probe.py 20000000 5 device 2>&1 | tail -7

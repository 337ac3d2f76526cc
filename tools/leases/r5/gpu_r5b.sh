#!/bin/bash
# round 5: the device FASTA parser against the host reader, the planes packed from it, the CLI on it; box facts for the 1 Gbp files
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5b
(df -h /dev/shm /tmp . ; free -g; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null) > gpurun_out/r5b/box.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_fasta_device.py tests/test_gpu_cli.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5b/tests.log
tail -25 gpurun_out/r5b/tests.log
cat gpurun_out/r5b/box.txt

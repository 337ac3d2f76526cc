#!/bin/bash
# round 5: bed_inflate_kernel with parts switched off (probe build of the library: nanomotif_amd/libnmscan_probe.so)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ac
NM_LIB=$PWD/nanomotif_amd/libnmscan_probe.so timeout 900 python tools/inflate_probe.py 20000000 5 > gpurun_out/r5ac/probe.log 2>&1; echo "rc=$?"
grep "MODE\|PROBE\|slab\|text " gpurun_out/r5ac/probe.log | cut -c1-220

#!/bin/bash
# the device bedMethyl parser: tests, then plain text + bgzip parse times (tools/bed_probe.py) and a kernel trace of the same
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-parse_probe}
mkdir -p gpurun_out/$tag
timeout 1200 python -m pytest tests/test_gpu_bed_device.py -x -q -m gpu > gpurun_out/$tag/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/$tag/tests.log
export TMPDIR=/tmp; R=$PWD; cd /tmp; rm -rf /tmp/pp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/tools/inflate_pipeline_probe.py 20000000 8 2 two > $R/gpurun_out/$tag/trace.log 2>&1; echo "trace rc=$?"
find /tmp/pp -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
cd $R; grep MODE gpurun_out/$tag/trace.log
python3 - gpurun_out/$tag/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']; short=n.split('(anonymous namespace)::')[-1].split('(')[0][:40] if 'anonymous' in n else n[:40]
    print(f"{short:42s} calls {r['Calls']:>4s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e6:8.3f} ms {r['Percentage']:>6s}%")
PY

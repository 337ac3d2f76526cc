#!/bin/bash
# what lies between two scoring kernels of the headline step: ms_per_step with the counter clear left out
# (NM_SCORE_PROBE=1 on a PROBE BUILD of the library, made here with NM_CXXFLAGS=-DNM_SCORE_PROBES and unmade at the end: the
# counts are garbage, the line's own parity leg fails on purpose)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/step_gap
NM_CXXFLAGS=-DNM_SCORE_PROBES python3 -c "from nanomotif_amd import build; build.build()" > gpurun_out/step_gap/build.log 2>&1
trap 'python3 -c "from nanomotif_amd import build; build.build()" >> gpurun_out/step_gap/build.log 2>&1' EXIT
for rep in 1 2; do
  for probe in 0 1; do
    NM_SCORE_PROBE=$probe timeout 600 python bench.py --extras none --cpu-bins 0 --steps 200 --warmup 20 > gpurun_out/step_gap/p${probe}_$rep.json 2> gpurun_out/step_gap/p${probe}_$rep.log
    python3 - gpurun_out/step_gap/p${probe}_$rep.json $probe <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
    print("probe", sys.argv[2], "ms_per_step %.4f kernel_ms %.4f gap_us %.1f host_call_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], (d["ms_per_step"] - d["roofline"]["kernel_ms"]) * 1e3, d["per_rank"][0]["host_call_ms_per_step"]))
except Exception as e:
    print("probe", sys.argv[2], "no line:", e)
PY
  done
done

#!/bin/bash
# round 4: constraints consumed in pairs (inline-assembly walk, v_bitop3): parity, then same-device A/B of the cfg 5 kernel
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4g
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_baseline_configs.py tests/test_gpu_per_contig.py tests/test_gpu_synth.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2 3; do
for v in paired unpaired; do
  if [ $v = unpaired ]; then export NM_LIB=$PWD/tools/_ab/libnmscan_unpaired.so; else unset NM_LIB; fi
  timeout 600 python bench.py --extras none --cpu-bins 0 --steps 50 --warmup 5 > gpurun_out/r4g/bench_${v}_$rep.json 2> gpurun_out/r4g/bench_${v}_$rep.log
  python3 - <<PY
import json
d = json.load(open('gpurun_out/r4g/bench_${v}_$rep.json'))
r = d['roofline']; g = d.get('roofline_hbm_bound_round') or {}
print('$v rep $rep: ms_per_step %.4f kernel_ms %.4f frac %.4f value %.4g checksum %s greedy kernel_ms %s' % (d['ms_per_step'], r['kernel_ms'], r['frac'], d['value'], d.get('counts_checksum'), g.get('kernel_ms')))
PY
done
done

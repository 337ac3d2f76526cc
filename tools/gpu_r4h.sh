#!/bin/bash
# round 4: (1) where a lane of the device inflate spends its time (tools/inflate_proto.hip probes), (2) host threads of the
# native search at 1 Gbp (the box shows 256 hardware threads and grants 16 CPUs), (3) the new DEFLATE-kind tests
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r4h
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/inflate_proto tools/inflate_proto.hip -lz && timeout 600 /tmp/inflate_proto 1024 6 2>&1 | tail -5 | tee gpurun_out/r4h/inflate_probes.txt
timeout 900 python -m pytest tests/test_gpu_bed_device.py -x -q -m gpu 2>&1 | tail -3
for th in 4 8 12 16; do
  NM_SEARCH_THREADS=$th NM_SEARCH_TIMING=1 timeout 900 python bench.py --workload e2e > gpurun_out/r4h/e2e_t$th.json 2> gpurun_out/r4h/e2e_t$th.log
  echo "== threads $th"; grep "nm_search\]" gpurun_out/r4h/e2e_t$th.log | tail -1
  python3 - <<PY
import json
d = json.load(open('gpurun_out/r4h/e2e_t$th.json')); e = d.get('e2e', d); t = e.get('timings_rank0', {})
print({k: round(v, 4) for k, v in t.items() if k in ('search_s', 'native_search_s', 'postprocess_s', 'background_s', 'plan_s', 'upload_filter_s')}, 'wall', round(e.get('wall_s', 0), 4))
PY
done

#!/bin/bash
# round-3 closing run: the whole GPU suite, smoke(), the default bench line, the ingest kernel split
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|ERROR" | tail -6
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/bench_default_r3.json 2> gpurun_out/bench_default_r3.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_r3.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")})
print("roofline", {k: d["roofline"].get(k) for k in ("bound", "frac", "hbm_frac", "traffic_stale", "kernel_ms")})
print("e2e", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.get("e2e", {}).items() if not isinstance(v, (dict, list, str))})
c = d.get("cli", {})
print("cli", {p: {k: (round(v, 3) if isinstance(v, float) else v) for k, v in c.get(p, {}).items()} for p in ("device", "host")}, c.get("outputs_byte_equal"), c.get("warm_read_s"))
PY
bash tools/gpu_ingest_prof.sh prof_r3_ingest 2>&1 | tail -12

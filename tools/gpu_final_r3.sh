#!/bin/bash
# round-3 closing run: the whole GPU suite, smoke(), and the default bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/bench_default_r3.json 2> gpurun_out/bench_default_r3.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_default_r3.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step", "n_gpus")})
print("roofline", {k: d["roofline"].get(k) for k in ("bound", "frac", "hbm_frac", "traffic_stale", "kernel_ms")})
print("e2e", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.get("e2e", {}).items() if not isinstance(v, (dict, list))})
print("cli", d.get("cli"))
print("cpu_baseline", d.get("cpu_baseline"))
PY

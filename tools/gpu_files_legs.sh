#!/bin/bash
# the default bench's from-files leg (cfg 3, 100 Mbp) with extra CLI legs ($1: NM_BENCH_CLI1G_LEGS syntax) and every leg's timing lines
cd ${GRAFT_REPO_ROOT:-.}
tag=${2:-files_legs}
mkdir -p gpurun_out/$tag
NM_BENCH_CLI1G_LEGS="$1" timeout 1200 python bench.py --extras files --steps 3 --warmup 1 > gpurun_out/$tag/line.json 2> gpurun_out/$tag/bench.err
echo "rc=$?"
python3 - gpurun_out/$tag/line.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
f = d.get("e2e_files", {})
def show(name, wall, p, log):
    print(f"{name:10s} wall {wall:.3f}", {k: round(v, 3) for k, v in (p or {}).items() if isinstance(v, float)})
    for ln in (log or [])[:40]:
        print("      ", ln[:420])
show("cold", f.get("wall_s", 0), f.get("phases"), None)
show("again", f.get("wall_s_second_process", 0), None, f.get("second_process_log"))
for k, v in (f.get("other_legs") or {}).items():
    show(k, v.get("wall_s") or 0, v.get("phases"), v.get("log"))
print(f.get("parity"), f.get("both_runs_byte_equal"), d.get("extra_errors"))
PY

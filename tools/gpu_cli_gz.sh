#!/bin/bash
# round 4: the file-to-bin-motifs.tsv legs of bench.py (plain text and bgzip + tabix, device and host parser) on cfg 3
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/cli_gz
nproc > gpurun_out/cli_gz/nproc.txt
timeout 1500 python bench.py --extras cli --steps 5 --warmup 1 --cpu-bins 0 --hbm-round-steps 0 > gpurun_out/cli_gz/bench.json 2> gpurun_out/cli_gz/bench.log
echo rc=$?
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/cli_gz/bench.json'))
c = d.get('cli', {})
for k in ('device', 'host', 'gz_device', 'gz_host'):
    print(k, {x: (round(v, 3) if isinstance(v, float) else v) for x, v in (c.get(k) or {}).items()})
print({k: c.get(k) for k in ('outputs_byte_equal', 'gz_outputs_byte_equal', 'gz_over_plain_wall', 'bed_bytes', 'gz_bytes')}, d.get('extra_errors'))
PY
tail -5 gpurun_out/cli_gz/bench.log

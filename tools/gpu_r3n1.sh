#!/bin/bash
# the multi-rank legs of bench.py on one GPU: the shard of rank 0 of 4, the RCCL step with a world of one, two gloo ranks on one device
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 600 python bench.py --as-rank-of 4 --steps 10 --warmup 3 --cpu-bins 0 --extras none > gpurun_out/n_asrank.json 2> gpurun_out/n_asrank.err; echo "as-rank-of rc $?"
timeout 600 python bench.py --force-allreduce --steps 10 --warmup 3 --cpu-bins 0 --extras none > gpurun_out/n_force.json 2> gpurun_out/n_force.err; echo "force-allreduce rc $?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --dist-backend gloo --force-device 0 --cpu-bins 0 --extras e2e > gpurun_out/n_gloo2.json 2> gpurun_out/n_gloo2.err; echo "gloo2 rc $?"
python - <<'PY'
import json
for f in ("n_asrank", "n_force", "n_gloo2"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        e = d.get("e2e", {})
        print(f, d["n_gpus"], round(d["ms_per_step"], 4), d.get("checksum_matches_n1"), d.get("rccl"), {k: e.get(k) for k in ("wall_s", "motif_rows")} if e else "")
    except Exception as ex:
        print(f, "FAILED", ex, open(f"gpurun_out/{f}.err").read()[-600:])
PY

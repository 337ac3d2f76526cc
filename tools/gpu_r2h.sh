#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r2h
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_synth.py tests/test_gpu_per_contig.py -x -q -m gpu 2>&1 | tail -2
for env in "X=1" "X=2"; do
for cfg in "--total-bp 1000000000 --contigs 10000 --bins 500 --candidates 10000" "--total-bp 500000000 --contigs 5000 --bins 250 --candidates 5000" "--total-bp 250000000 --contigs 2500 --bins 125 --candidates 2500" "--total-bp 125000000 --contigs 1250 --bins 63 --candidates 1260"; do
 for w in "cfg5" "greedy --per-group 2"; do
  env $env python bench.py $cfg --workload $w --steps 50 --warmup 5 --cpu-bins 0 --extras none --hbm-round-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(\"$env $w $cfg\", \"ms/step %.4f kernel %.4f value %.3e\" % (d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"], d[\"value\"]), d[\"counts_checksum\"])" | tee -a gpurun_out/r2h/fine.log
 done
done
done

#!/bin/bash
# round 5: bench --extras cli1g — rehearsal at 100 Mbp, then the headline size (1 Gbp: FASTA + .bed.gz + .tbi -> one cold CLI process)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5c
timeout 900 python bench.py --steps 3 --warmup 1 --extras cli1g --cli1g-bp 100000000 --cpu-bins 0 > gpurun_out/r5c/cli100m.log 2>&1
echo "rehearsal rc=$?"; tail -1 gpurun_out/r5c/cli100m.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print(json.dumps(d.get('cli1g', d.get('extra_errors')), indent=1)[:6000])"
if [ "${1:-full}" = "full" ]; then
timeout 2400 python bench.py --steps 3 --warmup 1 --extras cli1g --cpu-bins 0 > gpurun_out/r5c/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/r5c/cli1g.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print(json.dumps(d.get('cli1g', d.get('extra_errors')), indent=1)[:9000])"
grep "cli1g\|part " gpurun_out/r5c/cli1g.log | tail -30
fi

#!/bin/bash
# the CLI at 1 Gbp from files (bench.py --extras cli1g) with A/B legs given as NM_BENCH_CLI1G_LEGS syntax in $1 (default: the single
# inflate kernel of rounds 4 - 5 against the two-phase inflate, twice each), one box; log under gpurun_out/$2
cd ${GRAFT_REPO_ROOT:-.}
tag=${2:-cli1g_ab}
mkdir -p gpurun_out/$tag
legs=${1:-"v1a:NM_BED_INFLATE_V1=1;NM_BED_TIMING=1,two_a:NM_BED_TIMING=1,v1b:NM_BED_INFLATE_V1=1;NM_BED_TIMING=1,two_b:NM_BED_TIMING=1"}
NM_BENCH_CLI1G_LEGS="$legs" timeout 2700 python bench.py --extras cli1g --cpu-bins 0 --steps 3 --warmup 1 > gpurun_out/$tag/cli1g.log 2>&1
echo "cli1g rc=$?"; tail -1 gpurun_out/$tag/cli1g.log > gpurun_out/$tag/line.json; python3 - gpurun_out/$tag/line.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).readline()); c = d.get('cli1g', d.get('extra_errors'))
if isinstance(c, dict) and 'legs' in c:
    for leg, v in c['legs'].items():
        if 'error' in v:
            print(leg, 'ERROR', v['error'][-600:]); continue
        p = v['phases']
        print(f"{leg:8s} wall {v['wall_s']:.3f}  pileup {p['pileup_s']:.3f} (read {p['pileup_read_s']:.3f} inflate-wait {p['pileup_inflate_s']:.3f} parse {p['pileup_parse_s']:.3f})  fasta {p['fasta_s']:.3f} engine {p['engine_start_s']:.3f} filters {p['filters_s']:.3f} search {p['search_s']:.3f} imports {p['interpreter_and_imports_s']:.3f}  rows {v['motif_rows']}")
        for ln in v.get('parser_slab_log', [])[:60]:
            if 'slab' not in ln or 'slab 0:' in ln or 'slab 1:' in ln or 'slab 5:' in ln or 'slab 23' in ln or 'allocated' in ln: print('      ', ln[:300])
    print('parity', c.get('parity', {}).get('byte_equal_to_the_oracle_pipeline'), 'all legs byte-equal', c.get('both_runs_byte_equal'), 'write_s', round(c.get('write_s', 0), 1), c.get('size_note'))
else:
    print(json.dumps(c)[:3000])
PY

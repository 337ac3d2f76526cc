"""One line per ingest kernel from the PMC passes of tools/gpu_ingest_pmc.sh (profiles/<round>/ingest/pmc_*.csv):
traffic per launch (FETCH_SIZE doubled on gfx950 + WRITE_SIZE), instruction counts, where the wave cycles go.
usage: python3 tools/ingest_pmc_summary.py profiles/r4/ingest"""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*.csv"):
    for r in csv.reader(open(f)):                 # rows without the header: 8 = kernel name, 15 = counter, 16 = value
        if len(r) < 17:
            continue
        k = r[8]
        name = "count_kernel" if "ingest_count_kernel" in k else ("decide_kernel" if "ingest_decide_kernel<true" in k else None)
        if name:
            agg[name][r[15]].append(float(r[16]))
for name, c in sorted(agg.items()):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    wc = m["SQ_WAVE_CYCLES"]
    print(f"{name}: traffic {(m['FETCH_SIZE'] * 2 + m['WRITE_SIZE']) * 1024 / 1e9:.2f} GB per launch (FETCH x 2 + WRITE); VALU {m['SQ_INSTS_VALU'] / 1e6:.0f} M, "
          f"SALU {m['SQ_INSTS_SALU'] / 1e6:.0f} M, LDS {m['SQ_INSTS_LDS'] / 1e6:.1f} M wave-instr; wave time: ACTIVE {100 * m['SQ_ACTIVE_INST_ANY'] / wc:.0f} %, "
          f"WAIT_INST_ANY {100 * m['SQ_WAIT_INST_ANY'] / wc:.0f} %, WAIT_ANY {100 * m['SQ_WAIT_ANY'] / wc:.0f} %")

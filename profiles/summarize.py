#!/usr/bin/env python3
"""Turn the per-kernel rocprofv3 rows copied back by profiles/run_profile.sh into profiles/<round>/ (committed)
plus profiles/traffic.json (read by bench.py).   usage: python profiles/summarize.py gpurun_out/prof_<tag> <round>"""
import collections
import csv
import hashlib
import json
import os
import shutil
import sys

src, rnd = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else ""          # optional sub-directory / workload tag (e.g. "greedy2")
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), rnd, sub)
os.makedirs(dst, exist_ok=True)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in ("nmscan.hip", "nmscan_device.h", "nmscan_internal.h"):
    _h.update(open(os.path.join(ROOT, "nanomotif_amd", "csrc", _f), "rb").read())
KERNEL_SHA = _h.hexdigest()[:16]
KEEP = ("score_kernel", "compile_kernel", "pack_kernel", "state_kernel", "needs_v_kernel")


def pmc(name, kernel):
    """Mean counter values over the launches of ``kernel`` only (a bench run also launches the other instantiations
    of score_kernel, e.g. the fused greedy round behind roofline_hbm_bound_round)."""
    agg = collections.defaultdict(list)
    path = os.path.join(src, name)
    if not os.path.exists(path):
        return {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"].replace("(anonymous namespace)::", ""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


rows = [r for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))]
with open(os.path.join(dst, "kernel_stats.csv"), "w") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    for r in rows:
        if any(k in r["Name"] for k in KEEP):
            w.writerow(r)
# the instantiation the bench's main workload launches: the score_kernel row with the most calls (pre-warm + warm-up + steps)
score = max((r for r in rows if "score_kernel" in r["Name"]), key=lambda r: int(r["Calls"]))
kernel_name = "score_kernel" + score["Name"].replace("(anonymous namespace)::", "").split("score_kernel")[1].split("(")[0]
names = {"pmc_FETCH_SIZE.csv": "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv": "pmc_WRITE_SIZE.csv",
         "pmc_SQ_WAVES_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_.csv": "pmc_sq_insts.csv",
         "pmc_SQ_WAIT_INST_ANY_SQ_WAIT_ANY_SQ_ACTIVE_I.csv": "pmc_sq_wait.csv",
         "pmc_TCC_HIT_sum_TCC_MISS_sum_TCC_EA0_RDREQ_s.csv": "pmc_tcc.csv",
         "kernel_trace_score.csv": "kernel_trace_score.csv", "bench_under_trace.json": "bench_under_trace.json"}
for a, b in names.items():
    if not os.path.exists(os.path.join(src, a)):
        continue
    if a.endswith(".csv"):
        # the pre-warm steps of bench.py multiply the rows: the committed copy keeps the header and the last launches only
        lines = open(os.path.join(src, a)).read().splitlines(True)
        keep = 24 if "trace" in a else 24 * 16
        open(os.path.join(dst, b), "w").writelines(lines[:1] + lines[1:][-keep:])
    else:
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
c = {}
for n in names:
    if n.startswith("pmc_"):
        c.update(pmc(n, kernel_name))
bench = json.load(open(os.path.join(src, "bench_under_trace.json")))
avg_us = float(score["AverageNs"]) / 1e3
algo = bench["roofline"]["algorithmic_bytes_per_launch"]
traffic = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
clock_ghz = c["GRBM_GUI_ACTIVE"] / 8 / (avg_us * 1e3)
out = {
    "round": rnd, "kernel": kernel_name,
    "workload": bench["config"]["workload"].split(":")[0], "total_bp": bench["config"]["total_bp"],
    "candidates": bench["config"]["candidates"], "n_gpus": 1, "kernel_avg_us_under_trace": avg_us,
    "FETCH_SIZE_KB_per_launch": c["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch": c["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide coalesced streaming reads -> doubled "
                  "(MI355X_MICROARCH.md, HBM); WRITE_SIZE taken as is",
    "hbm_bytes_per_launch": traffic, "sq_insts_valu_per_launch": c["SQ_INSTS_VALU"], "sq_insts_salu_per_launch": c["SQ_INSTS_SALU"],
    # the scalar side as numbers: cycles the scalar ALU spent on instructions, and cycles the sequencers were busy
    "sq_inst_cycles_salu_per_launch": c.get("SQ_INST_CYCLES_SALU"), "sq_busy_cycles_per_launch": c.get("SQ_BUSY_CYCLES"),
    "sq_wave_cycles_per_launch": c.get("SQ_WAVE_CYCLES"), "sq_active_inst_valu_per_launch": c.get("SQ_ACTIVE_INST_VALU"),
    # ties the entry to the kernel source it was measured on (bench.py: roofline.traffic_stale)
    "kernel_source_sha16": KERNEL_SHA,
    "source": [f"profiles/{rnd}/{sub + '/' if sub else ''}pmc_FETCH_SIZE.csv", f"profiles/{rnd}/{sub + '/' if sub else ''}pmc_WRITE_SIZE.csv",
               f"profiles/{rnd}/{sub + '/' if sub else ''}pmc_sq_insts.csv"],
}
# profiles/traffic.json holds one entry per measured (workload, size, candidates): bench.py looks its configuration up
tj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json")
old = json.load(open(tj)) if os.path.exists(tj) else {"entries": []}
entries = old if isinstance(old, list) else old.get("entries", [old])
key = lambda e: (e.get("workload"), e.get("total_bp"), e.get("candidates"), e.get("n_gpus", 1))
entries = [e for e in entries if key(e) != key(out)] + [out]
json.dump({"entries": entries}, open(tj, "w"), indent=1)
json.dump(out, open(os.path.join(dst, "counters.json"), "w"), indent=1)
md = f"""# {rnd} {sub} profile — `python3 bench.py --cpu-bins 0 ...` ({bench['config']['workload'][:60]}..., 1 MI355X)

Collected with `profiles/run_profile.sh` (rocprofv3, ROCm 7.2; `--kernel-trace --stats` in its own run, every `--pmc`
group in its own run, no trace domains combined with counters).  Raw rocprofv3 output stayed on the GPU box; the CSVs
here hold the rows of this repository's kernels only.  Generated by `profiles/summarize.py`.

| quantity | value | source |
|---|---|---|
| `{out['kernel']}` launches in the traced run | {score['Calls']} | `kernel_stats.csv` |
| average kernel duration | **{avg_us:.1f} us** (min {float(score['MinNs'])/1e3:.1f}, max {float(score['MaxNs'])/1e3:.1f}) | `kernel_stats.csv`; bench.py's own HIP-event mean in the same run: {bench['roofline']['kernel_ms']*1e3:.1f} us (`bench_under_trace.json`) |
| algorithmic bytes per launch | {algo:.4e} B (0.5 B/bp per (bin, mod type) step + 16 B per candidate) | DESIGN.md §3 |
| achieved (algorithmic) | {algo/avg_us/1e3:.0f} GB/s = **{algo/avg_us/1e3/8000*100:.1f} % of 8 TB/s** | |
| FETCH_SIZE / WRITE_SIZE | {c['FETCH_SIZE']:.0f} KB (x2 on gfx950) / {c['WRITE_SIZE']:.0f} KB per launch | `pmc_FETCH_SIZE.csv`, `pmc_WRITE_SIZE.csv` |
| HBM traffic per launch | **{traffic:.4e} B = {traffic/algo:.2f} x algorithmic** (chunk padding 4.2 %, halo words, V plane of boundary chunks, programs) | `../traffic.json` |
| TCC_EA0_RDREQ_sum | {c.get('TCC_EA0_RDREQ_sum', 0):.3e} requests (x 128 B = {c.get('TCC_EA0_RDREQ_sum', 0)*128:.3e} B); TCC hit {c.get('TCC_HIT_sum', 0):.3e} / miss {c.get('TCC_MISS_sum', 0):.3e} | `pmc_tcc.csv` |
| SQ_INSTS_VALU / SALU / SMEM / LDS | {c['SQ_INSTS_VALU']/1e6:.1f} M / {c['SQ_INSTS_SALU']/1e6:.1f} M / {c['SQ_INSTS_SMEM']/1e6:.2f} M / {c['SQ_INSTS_LDS']/1e6:.2f} M wave-instructions per launch | `pmc_sq_insts.csv` |
| integer-VALU rate | {c['SQ_INSTS_VALU']/(avg_us*1e-6)/1e11:.2f}e11 wave-instr/s = **{c['SQ_INSTS_VALU']/(avg_us*1e-6)/1.2288e12*100:.0f} % of the chip's issue peak** (256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 op = 1.23e12) = {c['SQ_INSTS_VALU']/(avg_us*1e-6)/6.1e11*100:.0f} % of the 6.1e11 this instruction mix reaches in `tools/valu_peak.hip` (half its ops are the half-rate v_alignbit_b32) | see DESIGN.md §4 |
| scalar side | SQ_INST_CYCLES_SALU {c.get('SQ_INST_CYCLES_SALU', 0)/1e6:.1f} M, SQ_BUSY_CYCLES {c.get('SQ_BUSY_CYCLES', 0)/1e6:.1f} M, SALU / VALU instructions {c['SQ_INSTS_SALU']/c['SQ_INSTS_VALU']:.2f} | `pmc_sq_insts.csv`, `pmc_sq_wait.csv` |
| wave time split | ACTIVE_INST_ANY {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']*100:.0f} %, WAIT_INST_ANY {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']*100:.0f} %, WAIT_ANY {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']*100:.0f} % of SQ_WAVE_CYCLES | `pmc_sq_wait.csv` |
"""
open(os.path.join(dst, "SUMMARY.md"), "w").write(md)
print(md)

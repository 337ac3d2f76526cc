"""Print the few numbers of a bench.py JSON line that A/B comparisons look at (reads stdin)."""
import json, sys
d = json.loads(sys.stdin.read().strip().split("\n")[-1])
out = {"ms_per_step": round(d["ms_per_step"], 4), "kernel_ms": round(d["roofline"]["kernel_ms"], 4), "frac": round(d["roofline"]["frac"], 3)}
if "roofline_hbm_bound_round" in d:
    out["greedy2_kernel_ms"] = round(d["roofline_hbm_bound_round"]["kernel_ms"], 4)
    out["greedy2_frac"] = round(d["roofline_hbm_bound_round"]["frac"], 3)
out["host_call_ms"] = round(d["per_rank"][0]["host_call_ms_per_step"], 4) if d.get("per_rank") else None
for k in ("two_lanes", "strict_order"):
    if k in d:
        out[k + "_ms"] = round(d[k]["ms_per_step"], 4)
out["checksum"] = d.get("counts_checksum")
print(sys.argv[1] if len(sys.argv) > 1 else "", out)

"""Throughput of the native bedMethyl reader (nm_bed_open) on this host: a synthetic pileup is written once, replicated
to ~2 GB, then parsed with different thread counts (plain text and bgzip-less gzip is not parallel, so plain only)."""
import os, sys, time, json, shutil
sys.path.insert(0, ".")
from nanomotif_amd import synth, pileup
tmp = sys.argv[1] if len(sys.argv) > 1 else "/tmp/reader_probe"
os.makedirs(tmp, exist_ok=True)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=8, total_bp=2_000_000, n_bins=2, mod_types=("a", "m"), seed=3))
t0 = time.perf_counter(); mg.write_bed(tmp + "/one.bed"); w = time.perf_counter() - t0
one = os.path.getsize(tmp + "/one.bed")
reps = max(1, int(2e9 // one))
with open(tmp + "/big.bed", "wb") as out:
    blob = open(tmp + "/one.bed", "rb").read()
    for _ in range(reps):
        out.write(blob)
size = os.path.getsize(tmp + "/big.bed")
res = {"file_bytes": size, "rows": None, "host_cores": os.cpu_count(), "write_one_s": w, "runs": []}
for th in (1, 8, 32, 64, 128, 0):
    t0 = time.perf_counter(); t = pileup.load_pileup(tmp + "/big.bed", threads=th); dt = time.perf_counter() - t0
    res["rows"] = len(t)
    res["runs"].append({"threads": th if th else "auto", "seconds": round(dt, 3), "GB_per_s": round(size / dt / 1e9, 2), "Mrows_per_s": round(len(t) / dt / 1e6, 1)})
    del t
print(json.dumps(res))
shutil.rmtree(tmp)

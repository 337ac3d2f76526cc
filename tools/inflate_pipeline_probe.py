"""The device parse of a bgzip pileup (nm_bed_parse_device) with the two-phase inflate and with the single kernel of rounds 4 - 5
(NM_BED_INFLATE_V1=1), NM_BED_TIMING=1: per-slab waits of the pipeline.  argv: total_bp of one copy, copies, repetitions."""
import os, sys, time, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nanomotif_amd import synth, pileup as pp, e2e_synth
from nanomotif_amd.engine import ScanEngine
total_bp = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["two", "v1"]
tmp = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/inflate_pipeline_probe"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=max(8, total_bp // 100_000), total_bp=total_bp, n_bins=max(2, total_bp // 2_000_000), mod_types=("a", "m"), seed=3))
mg.write_bed(tmp + "/one.bed")
bed = open(tmp + "/one.bed", "rb").read()
with open(tmp + "/pileup.bed", "wb") as f:
    for k in range(copies):
        f.write(bed.replace(b"contig_", b"k%d_contig_" % k))
del bed
text = os.path.getsize(tmp + "/pileup.bed")
e2e_synth.bgzip_tabix(tmp + "/pileup.bed", tmp + "/pileup.bed.gz")
os.remove(tmp + "/pileup.bed")
print("text %.2f GB, bgzip %.2f GB" % (text / 1e9, os.path.getsize(tmp + "/pileup.bed.gz") / 1e9), flush=True)
eng = ScanEngine(0)
os.environ["NM_BED_TIMING"] = "1"
for rep in range(reps):
    for mode in modes:
        if mode == "v1":
            os.environ["NM_BED_INFLATE_V1"] = "1"
        else:
            os.environ.pop("NM_BED_INFLATE_V1", None)
        t0 = time.perf_counter()
        d = pp.DevicePileup(eng, tmp + "/pileup.bed.gz")
        dt = time.perf_counter() - t0
        print("MODE %s: %d rows in %.3f s (%.1f GB/s of text); copying %.3f, waiting for the inflate %.3f, parsing %.3f" %
              (mode, len(d), dt, text / 1e9 / dt, d.seconds_reading, d.seconds_inflating, d.seconds_parsing), flush=True)
        d.close()
eng.close()
shutil.rmtree(tmp)

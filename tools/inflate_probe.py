"""What each part of bed_inflate_kernel costs: the device parse of a bgzip pileup on a PROBE build of the library (NM_LIB, built with
NM_CXXFLAGS=-DNM_BED_PROBES) with NM_BED_INFLATE_PROBE = 0 (everything), 1 (no match copies), 3 (no stores at all), 4 (tables only)."""
import os, sys, time, shutil
sys.path.insert(0, ".")
from nanomotif_amd import synth, pileup as pp, e2e_synth
from nanomotif_amd.engine import ScanEngine
total_bp = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 5
tmp = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/inflate_probe"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=max(8, total_bp // 100_000), total_bp=total_bp, n_bins=max(2, total_bp // 2_000_000), mod_types=("a", "m"), seed=3))
mg.write_bed(tmp + "/one.bed")
bed = open(tmp + "/one.bed", "rb").read()
with open(tmp + "/pileup.bed", "wb") as f:
    for k in range(copies):
        f.write(bed.replace(b"contig_", b"k%d_contig_" % k))
del bed
e2e_synth.bgzip_tabix(tmp + "/pileup.bed", tmp + "/pileup.bed.gz")
print("text %.2f GB, bgzip %.2f GB" % (os.path.getsize(tmp + "/pileup.bed") / 1e9, os.path.getsize(tmp + "/pileup.bed.gz") / 1e9), flush=True)
eng = ScanEngine(0)
os.environ["NM_BED_TIMING"] = "1"
for mode in ("0", "0", "1", "3", "4", "7", "0"):
    os.environ["NM_BED_INFLATE_PROBE"] = mode
    t0 = time.perf_counter()
    try:
        d = pp.DevicePileup(eng, tmp + "/pileup.bed.gz")
        print("MODE %s: %d rows in %.3f s" % (mode, len(d), time.perf_counter() - t0), flush=True)
        d.close()
    except Exception as e:
        print("MODE %s: %.3f s, ended with: %s" % (mode, time.perf_counter() - t0, str(e)[:120]), flush=True)
eng.close()
shutil.rmtree(tmp)

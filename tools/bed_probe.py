"""nm_bed_parse_device on a bedMethyl text of the given size (argv: total_bp copies): seconds per call, twice."""
import os, sys, time, shutil
sys.path.insert(0, ".")
from nanomotif_amd import synth, pileup as pp
from nanomotif_amd.engine import ScanEngine
total_bp = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tmp = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/bed_probe"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=max(8, total_bp // 100_000), total_bp=total_bp, n_bins=max(2, total_bp // 2_000_000), mod_types=("a", "m"), seed=3))
mg.write_bed(tmp + "/one.bed")
bed = open(tmp + "/one.bed", "rb").read()
with open(tmp + "/pileup.bed", "wb") as f:
    for k in range(copies):
        f.write(bed.replace(b"contig_", b"k%d_contig_" % k))
del bed
print("bed %.2f GB" % (os.path.getsize(tmp + "/pileup.bed") / 1e9), flush=True)
eng = ScanEngine(0)
for rep in range(3):
    t0 = time.perf_counter()
    d = pp.DevicePileup(eng, tmp + "/pileup.bed")
    dt = time.perf_counter() - t0
    print("device parse: %d rows, %d contigs in %.3f s (%.1f GB/s); library: total %.3f, copying the file %.3f" %
          (len(d), len(d.contig_names), dt, os.path.getsize(tmp + "/pileup.bed") / 1e9 / dt, d.seconds, d.seconds_reading), flush=True)
    d.close()
# the same rows as bgzip + tabix (blocks inflated on the device)
from nanomotif_amd import e2e_synth
e2e_synth.bgzip_tabix(tmp + "/pileup.bed", tmp + "/pileup.bed.gz")
print("bgzip %.2f GB" % (os.path.getsize(tmp + "/pileup.bed.gz") / 1e9), flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    d = pp.DevicePileup(eng, tmp + "/pileup.bed.gz")
    dt = time.perf_counter() - t0
    print("device parse of the bgzip file: %d rows in %.3f s (%.1f GB/s of text); library: total %.3f, copying the compressed bytes %.3f" %
          (len(d), dt, os.path.getsize(tmp + "/pileup.bed") / 1e9 / dt, d.seconds, d.seconds_reading), flush=True)
    d.close()
eng.close()
shutil.rmtree(tmp)

"""Wall-clock profile of `python -m nanomotif_amd motif_discovery` on text files (FASTA + bedMethyl) of a given size."""
import os, sys, time, subprocess, shutil, json
sys.path.insert(0, ".")
from nanomotif_amd import synth
total_bp = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20_000_000
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # replicate the data set under renamed contigs / bins
tmp = ("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp") + "/cli_probe"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=max(8, total_bp // 100_000), total_bp=total_bp, n_bins=max(2, total_bp // 2_000_000), mod_types=("a", "m"), seed=3))
t0 = time.perf_counter()
mg.write_fasta(tmp + "/assembly.fasta"); mg.write_bed(tmp + "/pileup.bed"); mg.write_contig_bin(tmp + "/contig_bin.tsv")
if copies > 1:
    fa, bed, cb = (open(tmp + "/" + n, "rb").read() for n in ("assembly.fasta", "pileup.bed", "contig_bin.tsv"))
    with open(tmp + "/assembly.fasta", "wb") as f1, open(tmp + "/pileup.bed", "wb") as f2, open(tmp + "/contig_bin.tsv", "wb") as f3:
        for k in range(copies):
            tag = b"k%d_" % k
            f1.write(fa.replace(b">contig_", b">" + tag + b"contig_"))
            f2.write(bed.replace(b"contig_", tag + b"contig_"))
            f3.write(cb.replace(b"contig_", tag + b"contig_").replace(b"\tbin_", b"\t" + tag + b"bin_"))
    del fa, bed, cb
print(f"wrote inputs in {time.perf_counter() - t0:.1f}s: bed {os.path.getsize(tmp + '/pileup.bed') / 1e9:.2f} GB", flush=True)
env = dict(os.environ, PYTHONPATH=os.getcwd())
if len(sys.argv) > 3 and sys.argv[3] == "host":
    env["NANOMOTIF_HOST_PARSER"] = "1"
t0 = time.perf_counter()
r = subprocess.run([sys.executable, "-X", "importtime", "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out"],
                   cwd=tmp, env=env, capture_output=True, text=True)
wall = time.perf_counter() - t0
log = [l for l in r.stdout.splitlines() if " - INFO - " in l]
print("\n".join(log[:4] + log[-8:]))
print(json.dumps({"total_bp": total_bp * copies, "cli_wall_s": wall, "rc": r.returncode, "parser": "host" if "NANOMOTIF_HOST_PARSER" in env else "device"}))
if r.returncode:
    print(r.stderr[-2000:])
shutil.rmtree(tmp)

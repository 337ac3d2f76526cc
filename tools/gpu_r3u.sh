#!/bin/bash
# A/B of the device parser's file access (mmap + memcpy against pread) through the CLI, cold process each time
cd ${GRAFT_REPO_ROOT:-.}
python - <<'PY'
import os, sys, time, subprocess, shutil, json
sys.path.insert(0, ".")
from nanomotif_amd import synth
tmp = "/dev/shm/cli_ab"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=200, total_bp=20_000_000, n_bins=10, mod_types=("a", "m"), seed=3))
mg.write_fasta(tmp + "/assembly.fasta"); mg.write_bed(tmp + "/pileup.bed"); mg.write_contig_bin(tmp + "/contig_bin.tsv")
fa, bed, cb = (open(tmp + "/" + n, "rb").read() for n in ("assembly.fasta", "pileup.bed", "contig_bin.tsv"))
with open(tmp + "/assembly.fasta", "wb") as f1, open(tmp + "/pileup.bed", "wb") as f2, open(tmp + "/contig_bin.tsv", "wb") as f3:
    for k in range(5):
        tag = b"k%d_" % k
        f1.write(fa.replace(b">contig_", b">" + tag + b"contig_")); f2.write(bed.replace(b"contig_", tag + b"contig_"))
        f3.write(cb.replace(b"contig_", tag + b"contig_").replace(b"\tbin_", b"\t" + tag + b"bin_"))
print("bed GB", os.path.getsize(tmp + "/pileup.bed") / 1e9, flush=True)
for rep in range(3):
    for lib in ("libnmscan.so", "libnmscan_mmap.so"):
        env = dict(os.environ, PYTHONPATH=os.getcwd(), NM_LIB=os.getcwd() + "/nanomotif_amd/" + lib)
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out"],
                           cwd=tmp, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        t = json.load(open(tmp + "/out/logs/timings.motif_discovery.json"))
        print(lib, "wall %.3f" % wall, {k: round(v, 3) for k, v in t.items() if isinstance(v, float) and not k.startswith("search_")}, flush=True)
shutil.rmtree(tmp)
PY

"""The same CLI run several times on one set of files: what the first process after the files were written pays that the
later ones do not (and whether a parent that holds a GPU context matters)."""
import os, sys, time, subprocess, shutil, json
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth
tmp = "/dev/shm/cli_rep"
shutil.rmtree(tmp, ignore_errors=True); os.makedirs(tmp)
mg = synth.make_metagenome(synth.config("cfg3"))
hold = len(sys.argv) > 1 and sys.argv[1] == "hold"
dev = torch.device("cuda:0")
sizes = e2e_synth.write_text_inputs(mg, tmp, dev)
print("inputs", sizes, flush=True)
if not hold:
    torch.cuda.empty_cache()
if len(sys.argv) > 2 and sys.argv[2] == "warm":
    t0 = time.perf_counter()
    with open(tmp + "/pileup.bed", "rb", buffering=0) as f:
        buf = bytearray(64 << 20)
        while f.readinto(buf):
            pass
    print("warm read of the bed file: %.2f s" % (time.perf_counter() - t0), flush=True)
env = dict(os.environ, PYTHONPATH=os.getcwd())
for rep in range(3):
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out"],
                       cwd=tmp, env=env, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    t = json.load(open(tmp + "/out/logs/timings.motif_discovery.json"))
    lib = [l.split(" - INFO - ")[1] for l in r.stdout.splitlines() if "device parse" in l or "library" in l]
    print("run %d wall %.3f" % (rep, wall), {k: round(v, 3) for k, v in t.items() if isinstance(v, float) and not k.startswith("search_")}, lib[:2], flush=True)
shutil.rmtree(tmp)

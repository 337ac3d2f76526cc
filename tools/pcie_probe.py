"""PCIe-inclusive cost of the boundary: host buffers -> nm_upload_contigs + nm_ingest_pileup (raw rows) -> one scoring
batch.  cfg 3 sized (100 Mbp, ~1e8 raw pileup rows), generated on the host by synth.py."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
import torch
from nanomotif_amd import synth
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.motif import Motif, MOD_TYPE_TO_CANONICAL
from nanomotif_amd.pileup import MOD_TYPES

mg = synth.make_metagenome(synth.config(sys.argv[1] if len(sys.argv) > 1 else "cfg3"))
t0 = time.perf_counter()
seqs = [mg.contig_ascii(i) for i in range(len(mg.names))]
cols = {k: [] for k in ("contig", "position", "mod", "strand", "frac", "nvalid")}
for mt in mg.spec.mod_types:
    p = mg.pileup_columns(mt)                       # raw rows: no filter applied
    n = len(p["position"])
    cols["contig"].append(p["contig_id"]); cols["position"].append(p["position"].astype(np.uint32))
    cols["mod"].append(np.full(n, MOD_TYPES.index(mt), np.int8)); cols["strand"].append(p["strand"])
    cols["frac"].append(p["fraction_mod"]); cols["nvalid"].append(p["nvalid"])
cat = {k: np.concatenate(v) for k, v in cols.items()}
gen_s = time.perf_counter() - t0
eng = ScanEngine(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.upload_assembly(mg.names, seqs, mg.bin_names)
t_asm = time.perf_counter() - t0
t0 = time.perf_counter()
labels = {MOD_TYPES.index(mt): (mt, MOD_TYPE_TO_CANONICAL[mt]) for mt in mg.spec.mod_types}
res = eng.ingest_pileup(cat["contig"], cat["position"], cat["mod"], cat["strand"], cat["frac"], cat["nvalid"], labels, want_rows=False)
t_ing = time.perf_counter() - t0
bins = sorted(set(mg.bin_names))
raw = synth.random_candidates(20 * len(bins), seed=2, mod_types=mg.spec.mod_types)
cands = [(Motif(s, p), mt, bins[(k // 2) % len(bins)]) for k, (s, p, mt) in enumerate(raw)]
eng.score(cands)
t0 = time.perf_counter()
eng.score(cands)
t_step = time.perf_counter() - t0
bp = int(sum(mg.lengths))
sites = 2 * 20 * bp
bytes_in = bp + sum(v.nbytes for v in cat.values())
print(json.dumps({"total_bp": bp, "rows_raw": int(len(cat["position"])), "host_bytes_handed_over": int(bytes_in), "host_generation_s": gen_s,
                  "upload_assembly_s": t_asm, "ingest_pileup_s": t_ing, "GB_per_s_over_the_boundary": bytes_in / (t_asm + t_ing) / 1e9,
                  "score_step_s": t_step, "motif_sites_per_step": sites,
                  "sites_per_s_resident": sites / t_step, "sites_per_s_pcie_inclusive_single_step": sites / (t_asm + t_ing + t_step)}))

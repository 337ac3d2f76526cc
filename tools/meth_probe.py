"""Per-contig read-methylation table (nm_readstats_upload + nm_contig_methylation) at 1 Gbp / 1e9 records: seconds for the
upload of both mod codes and for N motifs x 10 000 contigs."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from nanomotif_amd import synth, e2e_synth, _lib
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.contig_methylation import read_methylation_table
from nanomotif_amd.pileup import MOD_TYPES
_lib.use_torch_allocator()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
dev = torch.device("cuda:0")
mg = synth.make_metagenome(synth.config(name))
mine, lengths, offsets, bins, ascii_all, cat = e2e_synth.generate_raw(mg, dev)
eng = ScanEngine(0)
eng.upload_assembly_device([mg.names[i] for i in mine], lengths, [mg.bin_names[i] for i in mine], ascii_all.data_ptr(), bin_names=bins)
nmod = torch.round(cat["nvalid"].to(torch.float64) * cat["frac"]).to(torch.int32)
vp = lambda x: C.c_void_p(x.data_ptr())
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for mt in ("a", "m"):
        sel = torch.nonzero(cat["mod"] == MOD_TYPES.index(mt)).squeeze(1)
        cols = [cat[k][sel].contiguous() for k in ("contig", "position", "strand", "nvalid")] + [nmod[sel].contiguous()]
        kept = C.c_uint64(0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _lib.check(eng.lib.nm_readstats_upload(eng.ctx, MOD_TYPES.index(mt), int(sel.numel()), vp(cols[0]), vp(cols[1]), vp(cols[2]), vp(cols[3]), vp(cols[4]),
                                               None, 3, 0.8, 1, C.byref(kept)))
        print(f"readstats upload {mt}: {int(sel.numel()):,} records -> {kept.value:,} kept in {time.perf_counter() - t1:.3f} s", flush=True)
zoo = [m for b in sorted(mg.bin_motifs)[:40] for m in mg.bin_motifs[b]]
motifs = list(dict.fromkeys(f"{m}_{mt}_{p}" for m, p, mt in zoo))
for n in (8, len(motifs)):
    t0 = time.perf_counter()
    rows = read_methylation_table(eng, motifs[:n], "median")
    dt = time.perf_counter() - t0
    obs = sum(r["n_motif_obs"] for r in rows)
    print(f"contig methylation: {n} motifs x {len(mg.names)} contigs -> {len(rows):,} rows, {obs:,} sites with records in {dt:.3f} s", flush=True)
eng.close()

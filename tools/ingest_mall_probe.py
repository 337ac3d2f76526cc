"""Do a contig group's rows survive in the 256 MB Infinity Cache between the counting pass and the decide pass?  (the
one-pass pre-filter.)  nm_ingest_pileup on PREFIXES of the 1e9-row synthetic pileup cut at contig boundaries — 110 MB ... 22 GB of raw
rows —, three calls each; run under `rocprofv3 --kernel-trace` and read the decide kernel's duration per launch: bytes per second of the
small prefixes (rows just streamed by ingest_count_kernel, still in the cache if it holds them) against the whole file's."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from nanomotif_amd import synth, e2e_synth, _lib
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.motif import MOD_TYPE_TO_CANONICAL
from nanomotif_amd.pileup import MOD_TYPES
_lib.use_torch_allocator()
dev = torch.device("cuda:0")
mg = synth.make_metagenome(synth.config("cfg4"))
mine, lengths, offsets, bins, ascii_all, cat = e2e_synth.generate_raw(mg, dev)
eng = ScanEngine(0)
eng.upload_assembly_device([mg.names[i] for i in mine], lengths, [mg.bin_names[i] for i in mine], ascii_all.data_ptr(), bin_names=bins)
slot_of = (C.c_int32 * 8)(*([-1] * 8)); canon = (C.c_uint8 * 8)(*([0] * 8))
for k, mt in enumerate(mg.spec.mod_types):
    slot_of[MOD_TYPES.index(mt)] = k; canon[MOD_TYPES.index(mt)] = ord(MOD_TYPE_TO_CANONICAL[mt])
vp = lambda x: C.c_void_p(x.data_ptr())
n_all = int(cat["position"].numel())
contig = cat["contig"]
# first row of every run of equal contig ids (the rows are grouped by contig)
change = torch.nonzero(contig[1:] != contig[:-1]).flatten() + 1
starts = torch.cat([torch.zeros(1, dtype=change.dtype, device=dev), change]).cpu().numpy()
print("runs", len(starts), "rows", n_all, flush=True)
for k in (25, 50, 100, 200, 400, 1000, 4000, len(starts)):
    n = int(starts[k]) if k < len(starts) else n_all
    for rep in range(3):
        nk, ncf = C.c_uint64(0), C.c_uint64(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(eng.lib.nm_ingest_pileup(eng.ctx, n, vp(cat["contig"]), vp(cat["position"]), vp(cat["mod"]), vp(cat["strand"]), vp(cat["frac"]),
                                            vp(cat["nvalid"]), slot_of, canon, 0.3, 0.7, 1, C.byref(nk), C.byref(ncf)))
        dt = time.perf_counter() - t0
    print("PREFIX contigs %d rows %d bytes %.1f MB: last call %.3f ms, kept %d" % (k, n, n * 22 / 1e6, dt * 1e3, nk.value), flush=True)
eng.close()

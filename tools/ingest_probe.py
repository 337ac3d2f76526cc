"""nm_ingest_pileup on the 1 Gbp / 1e9-row synthetic pileup, three times: wall per call; run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (count / scatter / decide)."""
import ctypes as C, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from nanomotif_amd import synth, e2e_synth, _lib
from nanomotif_amd.engine import ScanEngine
from nanomotif_amd.motif import MOD_TYPE_TO_CANONICAL
from nanomotif_amd.pileup import MOD_TYPES
_lib.use_torch_allocator()
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
dev = torch.device("cuda:0")
mg = synth.make_metagenome(synth.config(name))
mine, lengths, offsets, bins, ascii_all, cat = e2e_synth.generate_raw(mg, dev)
eng = ScanEngine(0)
eng.upload_assembly_device([mg.names[i] for i in mine], lengths, [mg.bin_names[i] for i in mine], ascii_all.data_ptr(), bin_names=bins)
slot_of = (C.c_int32 * 8)(*([-1] * 8)); canon = (C.c_uint8 * 8)(*([0] * 8))
for k, mt in enumerate(mg.spec.mod_types):
    slot_of[MOD_TYPES.index(mt)] = k; canon[MOD_TYPES.index(mt)] = ord(MOD_TYPE_TO_CANONICAL[mt])
vp = lambda x: C.c_void_p(x.data_ptr())
n = int(cat["position"].numel())
for rep in range(3):
    nk, ncf = C.c_uint64(0), C.c_uint64(0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    _lib.check(eng.lib.nm_ingest_pileup(eng.ctx, n, vp(cat["contig"]), vp(cat["position"]), vp(cat["mod"]), vp(cat["strand"]), vp(cat["frac"]),
                                        vp(cat["nvalid"]), slot_of, canon, 0.3, 0.7, 1, C.byref(nk), C.byref(ncf)))
    print("ingest %d rows: %.4f s, kept %d, confident %d" % (n, time.perf_counter() - t0, nk.value, ncf.value), flush=True)
eng.close()

#!/bin/bash
# round 4: where the bgzip parse spends its time (device inflate), on the cfg 3 files
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/gz_timing
python3 - <<'PY' 2>&1 | tee gpurun_out/gz_timing/out.txt
import os, sys, time, tempfile, shutil
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, pileup as pp
from nanomotif_amd.engine import ScanEngine
tmp = tempfile.mkdtemp(prefix="nm_gz_", dir="/dev/shm")
try:
    mg = synth.make_metagenome(synth.config("cfg3"))
    e2e_synth.write_text_inputs(mg, tmp, torch.device("cuda:0"))
    e2e_synth.bgzip_tabix(tmp + "/pileup.bed", tmp + "/pileup.bed.gz")
    eng = ScanEngine(0)
    for env in ({}, {"NM_BED_NO_CRC": "1"}, {"NM_BED_INFLATE_SLAB": str(1 << 30)}, {"NM_BED_INFLATE_SLAB": str(512 << 20)}, {"NM_BED_HOST_INFLATE": "1"}):
        os.environ.update(env, NM_BED_TIMING="1")
        for rep in range(2):
            t0 = time.perf_counter()
            t = pp.DevicePileup(eng, tmp + "/pileup.bed.gz")
            print(env, "rep", rep, "rows", len(t), "wall %.3f s" % (time.perf_counter() - t0), "library %.3f s" % t.seconds, flush=True)
            t.close()
        for k in env: del os.environ[k]
    t0 = time.perf_counter(); t = pp.DevicePileup(eng, tmp + "/pileup.bed"); print("plain text: wall %.3f s" % (time.perf_counter() - t0)); t.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
PY

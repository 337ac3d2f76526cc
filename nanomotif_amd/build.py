"""In-tree build of the HIP library (hipcc cross-compiles gfx950 without a GPU).

Every translation unit is compiled to its own object (in parallel, only when it or a header is newer) and the objects are
linked into ``libnmscan.so`` next to this file, so the library travels with the tree."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
SRC_HIP = [os.path.join(_CSRC, f) for f in ("nmscan.hip", "nmingest.hip", "nmwindows.hip", "nmmeth.hip", "nmbedgpu.hip", "nmfasta.hip")]
SRC_HOST = [os.path.join(_CSRC, f) for f in ("nmbed.cpp", "nmhost.cpp", "nmcomm.cpp", "nmsearch.cpp", "nmpost.cpp", "nmpool.cpp")]
OUT = os.path.join(_HERE, "libnmscan.so")
# synthetic-data tooling of bench.py and the tests (a bedMethyl text writer, a bgzip + tabix writer): its own small library,
# not part of the product's C ABI
SYNTH_SRC = os.path.join(_CSRC, "bench", "nmsynth.cpp")
SYNTH_OUT = os.path.join(_HERE, "libnmsynth.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "nmscan.h")
OBJ_DIR = os.path.join(_CSRC, "_build")


def _headers():
    return [HEADER] + [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith(".h")]


def sources():
    return [s for s in SRC_HIP + SRC_HOST if os.path.exists(s)]


def build_synth(force: bool = False) -> str:
    if force or not os.path.exists(SYNTH_OUT) or os.path.getmtime(SYNTH_OUT) < os.path.getmtime(SYNTH_SRC):
        subprocess.run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SYNTH_OUT, SYNTH_SRC, "-lz", "-lpthread"], check=True)
    return SYNTH_OUT


def build(force: bool = False, verbose: bool = False) -> str:
    build_synth(force)
    srcs = sources()
    hdr_time = max(os.path.getmtime(h) for h in _headers())
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    extra = os.environ.get("NM_CXXFLAGS", "").split()
    os.makedirs(OBJ_DIR, exist_ok=True)
    flags_file = os.path.join(OBJ_DIR, "flags.txt")
    flags = " ".join(extra)
    if not os.path.exists(flags_file) or open(flags_file).read() != flags:
        force = True
    common = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off"] + extra
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s) + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_time):
            jobs.append(common + ["-c", s, "-o", o])
    if not jobs and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(o) for o in objs):
        return OUT

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), (os.cpu_count() or 2)))) as pool:
        list(pool.map(run, jobs))
    # explicit RUNPATH: the torch-free CLI path loads the system HIP runtime through it (nanomotif_amd/_lib.py)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lz", "-lpthread", "-ldl", f"-Wl,-rpath,{rocm}/lib"])
    with open(flags_file, "w") as f:
        f.write(flags)
    return OUT

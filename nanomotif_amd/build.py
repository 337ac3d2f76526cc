"""In-tree build of the HIP library (hipcc cross-compiles gfx950 without a GPU)."""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC_HIP = [os.path.join(_HERE, "csrc", f) for f in ("nmscan.hip", "nmingest.hip", "nmwindows.hip", "nmmeth.hip", "nmbedgpu.hip")]
SRC_HOST = [os.path.join(_HERE, "csrc", f) for f in ("nmbed.cpp", "nmhost.cpp", "nmcomm.cpp", "nmsearch.cpp", "nmpost.cpp")]
INTERNAL = os.path.join(_HERE, "csrc", "nmscan_internal.h")
DEVICE_H = os.path.join(_HERE, "csrc", "nmscan_device.h")
OUT = os.path.join(_HERE, "libnmscan.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "nmscan.h")


def build(force: bool = False, verbose: bool = False) -> str:
    deps = SRC_HIP + [HEADER, INTERNAL, DEVICE_H, os.path.join(_HERE, "csrc", "nmbed_parse.h"), os.path.join(_HERE, "csrc", "nmsearch_internal.h")] + SRC_HOST
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    # explicit RUNPATH: the torch-free CLI path loads the system HIP runtime through it (nanomotif_amd/_lib.py)
    cmd = [hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + SRC_HIP + SRC_HOST + \
          ["-lz", "-lpthread", "-ldl", "-ffp-contract=off", f"-Wl,-rpath,{rocm}/lib"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return OUT

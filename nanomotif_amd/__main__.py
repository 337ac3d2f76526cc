"""``python -m nanomotif_amd ...`` — the reference's command line (main.py:349-369).

The HIP runtime takes 0.1–0.3 s to come up in a fresh process and the interpreter spends 0.2 s importing numpy and this package: a
single-GPU ``motif_discovery`` run lets the two happen at the same time — a thread loads libnmscan (ctypes only) and creates and drops
an engine context, which is what initialises the runtime, while the imports below proceed.  ``main.find_motifs_bin`` waits for that
thread before it installs its allocator and creates the real engine.  ``NANOMOTIF_NO_EARLY_INIT=1``: off.  (A multi-rank run brings
torch's runtime up first — one HIP runtime for both — and is left alone.)"""
import os
import sys
import threading


def _warm_up_the_gpu_runtime():
    argv = sys.argv[1:]
    if (not argv or argv[0] != "motif_discovery" or "-h" in argv or "--help" in argv or os.environ.get("WORLD_SIZE", "1") != "1"
            or os.environ.get("NANOMOTIF_WITH_TORCH") == "1" or os.environ.get("NANOMOTIF_NO_EARLY_INIT") == "1"):
        return None
    device = int(os.environ.get("LOCAL_RANK", "0"))
    for i, a in enumerate(argv[:-1]):
        if a == "--device" and argv[i + 1].isdigit():
            device = int(argv[i + 1])
    import ctypes as C
    from . import _lib                       # (ctypes and os only: imported here so that the thread imports nothing)

    def work():
        try:
            lib = _lib.load()
            ctx = C.c_void_p()
            if lib.nm_ctx_create(device, C.byref(ctx)) == 0:
                lib.nm_ctx_destroy(ctx)
        except Exception:                    # (no library, no GPU: the regular start says so)
            pass
    t = threading.Thread(target=work, name="nm-runtime-warm-up", daemon=True)
    t.start()
    _lib.warm_up_thread = t
    return t


_warm_up_the_gpu_runtime()

from .main import main  # noqa: E402

main()

"""``python -m nanomotif_amd ...`` — the reference's command line (main.py:349-369).

The HIP runtime takes 0.1 - 0.2 s to come up in a fresh process (and the engine context's three streams 0.06 s more) while the interpreter
spends 0.2 s importing numpy and this package: a single-rank ``motif_discovery`` run lets the two happen at the same time — a thread
loads libnmscan (ctypes only), installs the block cache and creates THE engine context, which ``main.find_motifs_bin`` adopts instead of
making its own.  ``NANOMOTIF_NO_EARLY_INIT=1``: off.  (A multi-rank run brings torch's runtime up first — one HIP runtime for both —
and is left alone.)"""
import os
import sys
import threading


def _start_the_engine_early():
    argv = sys.argv[1:]
    if (not argv or argv[0] != "motif_discovery" or "-h" in argv or "--help" in argv or int(os.environ.get("WORLD_SIZE", "1") or 1) > 1
            or os.environ.get("NANOMOTIF_WITH_TORCH") == "1" or os.environ.get("NANOMOTIF_NO_EARLY_INIT") == "1"):
        return
    device = int(os.environ.get("LOCAL_RANK", "0") or 0)
    for i, a in enumerate(argv[:-1]):
        if a == "--device" and argv[i + 1].isdigit():
            device = int(argv[i + 1])
    import ctypes as C
    from . import _lib                       # (ctypes and os only: imported here so that the thread imports nothing)

    def work():
        try:
            lib = _lib.load()
            # the file parsers' three pinned buffers (32 MB + 64 KB each, 16 ms to pin) on a thread of their own, beside the streams of the ctx
            # (without a ctx the buffers are pinned for the thread's current device, which is device 0: another --device pins its own later)
            if device == 0:
                pin = threading.Thread(target=lambda: lib.nm_warm_file_parsers(None, (32 << 20) + (1 << 16), 3), name="nm-pin-early", daemon=True)
                pin.start()
                _lib.early_pin_thread = pin
            cache_gb = float(os.environ.get("NANOMOTIF_BLOCK_CACHE_GB", "16"))
            if cache_gb > 0 and lib.nm_block_cache(1, int(cache_gb * (1 << 30)), None) != 0:
                return
            ctx = C.c_void_p()
            if lib.nm_ctx_create(device, C.byref(ctx)) == 0:
                _lib.early_engine = (device, ctx, cache_gb > 0)
        except Exception:                    # (no library, no GPU: the regular start says so)
            pass
    t = threading.Thread(target=work, name="nm-engine-early", daemon=True)
    t.start()
    _lib.early_engine_thread = t


_start_the_engine_early()

from .main import main  # noqa: E402

main()
# The command has written and closed everything it writes (bin-motifs.tsv, the per-task files, logs/timings): what is left is teardown —
# the interpreter's finalisation, the HIP runtime unmapping ~20 GB of device memory, worker threads freeing search graphs and unmapping the
# pileup — 0.08 s that change nothing on disk.  A single-rank command leaves through _exit and lets the kernel reclaim the process at once
# (NANOMOTIF_FAST_EXIT=0: the ordinary way out; several ranks always take it: their process group is torn down in order).
if os.environ.get("NANOMOTIF_FAST_EXIT", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1") or 1) <= 1:
    import logging
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)

import os
import sys

from .main import main

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

"""CPU oracle: a plain restatement of nanomotif's motif-discovery hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it; the
product package ``nanomotif_amd`` never does, and fails loudly when its HIP library is missing
instead of falling back to anything in here.

It restates, function by function, the algorithm of the reference (citations are
``/root/reference/nanomotif/<file>:<line>``) with the same third-party primitives the reference
uses on this path (``regex`` overlapped ``finditer``, ``numpy.isin``, ``scipy.special.psi``,
``scipy.stats.entropy``, Python's ``random.sample`` and ``heapq``).

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every module against vectors
recorded by running the real reference in the build container (``tests/golden/gen_golden.py``,
fixtures ``tests/golden/g*.json``) and against the literal known-answer values of the reference's
own tests (tests/test_motif_find.py:14-39, tests/test_fasta.py:95-109,
tests/test_dataload.py:37-69, tests/test_candidate.py, tests/test_postprocess.py).
Pieces of the reference that cannot run here (they need real polars: ``dataload`` filters,
``MotifSearchResult``, ``postprocess.remove_sub_motifs/join_motif_complements``,
``merge_motifs_in_df`` glue) are restated from source and pinned by those known-answer values
only; this is stated again at each such function.
"""

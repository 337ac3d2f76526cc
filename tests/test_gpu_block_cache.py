"""nm_block_cache (csrc/nmpool.cpp): the library's own block cache behind nm_set_device_allocator, installed by the command line.
Run in a child process: the allocator slot is process-wide."""
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent("""
    import numpy as np
    from nanomotif_amd import _lib, synth
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif

    def counts(cache):
        if cache:
            _lib.use_block_cache(8 << 30)
        spec = synth.SynthSpec(n_contigs=6, total_bp=120_000_000, n_bins=2, mod_types=("a",), seed=5, min_contig_bp=1_000_000)
        mg = synth.make_metagenome(spec)
        out = []
        for rep in range(2):                       # the second engine's planes (32 MiB and more) come out of the cache
            eng = ScanEngine(0)
            eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(len(mg.names))], mg.bin_names)
            for i in range(len(mg.names)):
                p = mg.contig_pileup(i, "a")
                keep = p["nvalid"] > 5
                eng.upload_pileup("a", np.full(int(keep.sum()), i, np.uint32), p["position"][keep], p["strand"][keep],
                                  synth.pct_to_fraction(p["pct_hundredths"][keep]), append=i > 0)
            out.append(eng.score([(Motif("GATC", 1), "a", b) for b in sorted(set(mg.bin_names))]).tolist())
            eng.close()
        return out

    plain = counts(False)
    cached = counts(True)
    st = _lib.block_cache_stats()
    assert plain == cached and plain[0] == plain[1] and sum(map(sum, plain[0])) > 0, (plain, cached)
    assert st["served_from_cache"] >= 2 and st["blocks_in_use"] == 0 and st["idle_bytes"] >= (32 << 20), st
    _lib.use_block_cache(0)
    assert _lib.block_cache_stats()["idle_bytes"] == 0
    print("ok", st)
""")


def test_block_cache_serves_freed_blocks_and_changes_no_count():
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok " in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]

"""The ctypes stub printed in INTEGRATION.md, executed verbatim (only the library path is made absolute and `polars` /
`nanomotif.constants` are stood in for): the binding a reference maintainer would add must really work."""
import os
import re
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_counts_like_the_oracle(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import refstub
    from nanomotif_amd import _lib, synth
    from nanomotif_amd.motif import MOD_TYPE_TO_CANONICAL, Motif
    from oracle.scan import ContigPileup, score_candidates
    _lib.load()                                              # HIP runtime of torch first, as the product does
    pl, _ = refstub._make_polars()
    monkeypatch.setitem(sys.modules, "polars", pl)
    pkg, const = types.ModuleType("nanomotif"), types.ModuleType("nanomotif.constants")
    const.MOD_TYPE_TO_CANONICAL = dict(MOD_TYPE_TO_CANONICAL)
    pkg.constants = const
    monkeypatch.setitem(sys.modules, "nanomotif", pkg)
    monkeypatch.setitem(sys.modules, "nanomotif.constants", const)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C.*?)```", text, re.S).group(1)
    assert 'C.CDLL("libnmscan.so")' in code
    code = code.replace('C.CDLL("libnmscan.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)

    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=3, total_bp=200_000, n_bins=1, mod_types=("a",), seed=12,
                                               fixed_motifs=(("GATC", 1, "a"), ("GAAGNNNNNTAC", 2, "a"))))
    contigs = {n: types.SimpleNamespace(sequence=mg.contig_str(i)) for i, n in enumerate(mg.names)}
    cols, pile = {"contig": [], "position": [], "strand": [], "fraction_mod": []}, {}
    for i, n in enumerate(mg.names):
        p = mg.contig_pileup(i, "a")
        frac = synth.pct_to_fraction(p["pct_hundredths"])
        pile[n] = ContigPileup(p["position"], p["strand"], frac)
        cols["contig"] += [n] * len(frac); cols["position"] += p["position"].tolist()
        cols["strand"] += [chr(c) for c in p["strand"]]; cols["fraction_mod"] += frac.tolist()
    frame = refstub.make_pileup(cols["contig"], cols["position"], cols["strand"], cols["fraction_mod"])
    gpu = ns["GpuBin"](contigs, frame, "a", 0.3, 0.7)
    motifs = [("GATC", 1), ("GAAG.....TAC", 2), ("A", 0), ("[AG]GATC[CT]", 2), ("." * 19 + "GATC" + "." * 18, 20)]
    got = gpu.counts([Motif(s, p) for s, p in motifs])
    exp = score_candidates(pile, {n: contigs[n].sequence for n in contigs}, motifs)
    assert np.array_equal(got, exp) and exp[0, 0] > 100


def test_integration_md_multi_gpu_stub_runs_with_one_rank(monkeypatch):
    """INTEGRATION.md §5 (count-table all-reduce through the C ABI) executed verbatim on top of the §2 stub with a world of
    one rank: the all-reduced table equals the local one."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import refstub
    from nanomotif_amd import _lib, synth
    from nanomotif_amd.motif import MOD_TYPE_TO_CANONICAL, Motif
    _lib.load()
    pl, _ = refstub._make_polars()
    monkeypatch.setitem(sys.modules, "polars", pl)
    pkg, const = types.ModuleType("nanomotif"), types.ModuleType("nanomotif.constants")
    const.MOD_TYPE_TO_CANONICAL = dict(MOD_TYPE_TO_CANONICAL)
    pkg.constants = const
    monkeypatch.setitem(sys.modules, "nanomotif", pkg)
    monkeypatch.setitem(sys.modules, "nanomotif.constants", const)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    base = next(b for b in blocks if b.startswith("import ctypes as C")).replace('C.CDLL("libnmscan.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    multi = next(b for b in blocks if "nm_comm_unique_id" in b)
    ns = {}
    exec(compile(base, "INTEGRATION.md#2", "exec"), ns)
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=2, total_bp=120_000, n_bins=1, mod_types=("a",), seed=13, fixed_motifs=(("GATC", 1, "a"),)))
    contigs = {n: types.SimpleNamespace(sequence=mg.contig_str(i)) for i, n in enumerate(mg.names)}
    cols = {"contig": [], "position": [], "strand": [], "fraction_mod": []}
    for i, n in enumerate(mg.names):
        p = mg.contig_pileup(i, "a")
        frac = synth.pct_to_fraction(p["pct_hundredths"])
        cols["contig"] += [n] * len(frac); cols["position"] += p["position"].tolist()
        cols["strand"] += [chr(c) for c in p["strand"]]; cols["fraction_mod"] += frac.tolist()
    frame = refstub.make_pileup(cols["contig"], cols["position"], cols["strand"], cols["fraction_mod"])
    ns.update(gpu_bin=ns["GpuBin"](contigs, frame, "a", 0.3, 0.7), rank=0, world=1, broadcast_bytes=lambda b, root=0: b)
    exec(compile(multi, "INTEGRATION.md#5", "exec"), ns)
    motifs = [Motif("GATC", 1), Motif("A", 0)]
    local = ns["gpu_bin"].counts(motifs)
    assert np.array_equal(ns["counts_all_ranks"](ns["gpu_bin"], motifs), local) and local[0, 0] > 50

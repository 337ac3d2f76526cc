"""GPU parity: the HIP engine (through the C ABI) against the reference-recorded vectors and the CPU oracle.
Integer counts and hit positions must be bit-exact."""
import hashlib

import numpy as np
import pytest

from helpers import load_golden, oracle_bin_inputs, sha1, spec_from_json
from nanomotif_amd import synth
from nanomotif_amd.motif import Motif

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine_cls():
    from nanomotif_amd.engine import ScanEngine
    return ScanEngine


def _upload_metagenome(eng, mg, mod_types, low=0.3, high=0.7, min_cov=5, contigs=None):
    idx = list(range(len(mg.names))) if contigs is None else list(contigs)
    eng.upload_assembly([mg.names[i] for i in idx], [mg.contig_ascii(i) for i in idx], [mg.bin_names[i] for i in idx])
    for mt in mod_types:
        first = True
        for local, i in enumerate(idx):
            p = mg.contig_pileup(i, mt)
            keep = p["nvalid"] > min_cov
            eng.upload_pileup(mt, np.full(int(keep.sum()), local, np.uint32), p["position"][keep], p["strand"][keep],
                              synth.pct_to_fraction(p["pct_hundredths"][keep]), low=low, high=high, append=not first)
            first = False


def test_reference_kat_tiny_sequence(engine_cls):
    # tests/test_motif_find.py:14-39 expressed through the engine: motif ACG@0 in TACGGACGCCACG
    eng = engine_cls()
    eng.upload_assembly(["c"], ["TACGGACGCCACG"], ["b"])
    eng.upload_pileup("a", [0, 0, 0], [1, 5, 10], np.frombuffer(b"+++", np.uint8), [0.9, 0.95, 0.1])
    out = eng.score([(Motif("ACG", 0), "a", "b")])
    assert out.tolist() == [[2, 1]]
    assert eng.hit_positions("c", "a", Motif("ACG", 0), 0).tolist() == [1, 5]
    assert eng.hit_positions("c", "a", Motif("ACG", 0), 1).tolist() == [10]
    # subseq_indices KAT (tests/test_fasta.py:95-109): AA.T in AATTAAATTAAGTAAAT -> starts [0,4,5,9,13]
    seq = "AATTAAATTAAGTAAAT"
    eng.upload_assembly(["c"], [seq], ["b"])
    eng.upload_pileup("a", np.zeros(len(seq), np.uint32), np.arange(len(seq)), np.full(len(seq), ord("+"), np.uint8),
                      np.ones(len(seq)))
    assert eng.hit_positions("c", "a", Motif("AA.T", 0), 0).tolist() == [0, 4, 5, 9, 13]
    assert eng.hit_positions("c", "a", Motif("AATT", 0), 0).tolist() == [0, 5]
    eng.close()


def test_g2_counts_and_hit_positions_match_reference(engine_cls):
    g = load_golden("g2_motif_model_contig.json")
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    assert hashlib.sha1(mg.contig_str(0).encode()).hexdigest() == g["seq_sha1"]
    for (low, high) in ((0.3, 0.7), (0.1, 0.9)):
        eng = engine_cls()
        _upload_metagenome(eng, mg, ("a", "m"), low=low, high=high, min_cov=-1)
        cases = [c for c in g["cases"] if (c["low"], c["high"]) == (low, high)]
        cands = [(Motif(c["motif"], c["pos"]), c["mod_type"], mg.bin_names[0]) for c in cases]
        out = eng.score(cands)                       # mixed canonical / non-canonical -> general planes
        assert out.tolist() == [[c["n_mod"], c["n_nomod"]] for c in cases]
        # canonical-only sub-batch takes the compact (strand-implied) path: must agree
        can = [k for k, c in enumerate(cases)
               if Motif(c["motif"], c["pos"]).split()[c["pos"]] == {"a": "A", "m": "C"}[c["mod_type"]]]
        out2 = eng.score([cands[k] for k in can])
        assert eng.stats()["last_compact"] == len(can) and len(can) > 10
        assert out2.tolist() == [[cases[k]["n_mod"], cases[k]["n_nomod"]] for k in can]
        keys = ("index_meth_fwd", "index_nonmeth_fwd", "index_meth_rev", "index_nonmeth_rev")
        for c in cases[::3]:
            for which, k in enumerate(keys):
                pos = eng.hit_positions(0, c["mod_type"], Motif(c["motif"], c["pos"]), which)
                assert len(pos) == c[k]["n"] and sha1(pos) == c[k]["sha1"], (c["motif"], k)
        eng.close()


def _oracle_counts(mg, mt, contigs, cands, low=0.3, high=0.7):
    from oracle.scan import score_candidates
    pile, seqs = oracle_bin_inputs(mg, mt, contigs=contigs)
    return score_candidates(pile, seqs, cands, low, high)


def test_multi_bin_metagenome_matches_oracle(engine_cls):
    spec = synth.SynthSpec(n_contigs=12, total_bp=900_000, n_bins=3, mod_types=("a", "m"), seed=77,
                           min_contig_bp=9_000, n_fraction=0.003)
    mg = synth.make_metagenome(spec)
    eng = engine_cls()
    _upload_metagenome(eng, mg, ("a", "m"))
    zoo = synth.random_candidates(120, seed=5)
    zoo += [("." * 19 + "GATC" + "." * 18, 20, "a"), ("." * 19 + "CC[AT]GG" + "." * 15, 20, "m"), ("A", 0, "a"), ("C", 0, "m")]
    cands, expect = [], []
    for b in sorted(set(mg.bin_names)):
        contigs = [i for i, x in enumerate(mg.bin_names) if x == b]
        for mt in ("a", "m"):
            these = [(s, p) for s, p, t in zoo if t == mt]
            expect.append(_oracle_counts(mg, mt, contigs, these))
            cands += [(Motif(s, p), mt, b) for s, p in these]
    out = eng.score(cands)
    assert np.array_equal(out, np.concatenate(expect))
    assert eng.stats()["last_compact"] == len(cands)
    # scoring the same batch in a shuffled order gives the same rows
    perm = np.random.default_rng(0).permutation(len(cands))
    assert np.array_equal(eng.score([cands[i] for i in perm]), out[perm])
    eng.close()


def test_edge_cases(engine_cls):
    eng = engine_cls()
    # contigs of awkward lengths around the 8192-bp chunk and 32-bit word edges, sites at both ends
    rng = np.random.default_rng(3)
    lens = [1, 5, 31, 32, 33, 64, 8191, 8192, 8193, 16384 - 64, 16384 - 63, 16385, 40_000]
    seqs = ["".join(rng.choice(list("ACGT"), size=n)) for n in lens]
    seqs[3] = "GATC" * 8
    seqs[5] = "N" * 10 + "GATC" + "R" * 10 + "GATCGATC" + "n" * 32
    names = [f"c{i}" for i in range(len(lens))]
    eng.upload_assembly(names, seqs, ["b"] * len(lens))
    from oracle.scan import ContigPileup, score_candidates
    pile, cid, pos, st, fr = {}, [], [], [], []
    for i, s in enumerate(seqs):
        p = np.array([k for k, ch in enumerate(s.upper()) if ch in "AT"], dtype=np.int64)
        strand = np.array([ord("+") if s.upper()[k] == "A" else ord("-") for k in p], dtype=np.uint8)
        f = rng.choice([0.0, 0.3, 0.30000000000000004, 0.5, 0.7, 0.6999999999999999, 1.0], size=len(p))
        pile[names[i]] = ContigPileup(p, strand, f)
        cid += [i] * len(p); pos += p.tolist(); st += strand.tolist(); fr += f.tolist()
    eng.upload_pileup("a", cid, pos, np.array(st, np.uint8), fr)
    motifs = [("GATC", 1), ("A", 0), ("A.", 0), (".A", 1), ("A" + "." * 30 + "T", 0), ("T" + "." * 39 + "A", 40),
              ("A" + "." * 62 + "C", 0), ("G" + "." * 62 + "A", 63), ("[ACG]A[CGT]", 1), ("AA", 0), ("AA", 1),
              ("TA", 1), ("G.TC....A", 8), ("....A....", 4)]
    out = eng.score([(Motif(s, p), "a", "b") for s, p in motifs])
    exp = score_candidates(pile, dict(zip(names, [s.upper() for s in seqs])), motifs)
    assert np.array_equal(out, exp), (out.tolist(), exp.tolist())
    # errors are loud
    from nanomotif_amd._lib import NmScanError
    with pytest.raises(NmScanError):
        eng.score([(Motif("....", 1), "a", "b")])                       # no specified position
    with pytest.raises(NmScanError):
        eng.upload_pileup("a", [0, 0], [0, 0], np.frombuffer(b"++", np.uint8), [1.0, 1.0])   # duplicate row
    with pytest.raises(NmScanError):
        eng.upload_pileup("a", [0], [5], np.frombuffer(b"+", np.uint8), [1.0])                # beyond contig 0 (len 1)
    assert eng.score([]).shape == (0, 2)
    eng.close()


def test_large_contig_properties(engine_cls):
    """cfg 2 size (5 Mbp): size-independent properties + oracle spot checks."""
    mg = synth.make_metagenome(synth.config("cfg2"))
    eng = engine_cls()
    _upload_metagenome(eng, mg, ("a",))
    M = lambda s, p: (Motif(s, p), "a", mg.bin_names[0])
    out = eng.score([M("GATC", 1), M("A", 0), M("AA", 1), M("CA", 1), M("GA", 1), M("TA", 1), M(".A", 1), M("GATC", 1)])
    # linearity: sites of 'A' = sum over the four possible left neighbours, except the (at most one per strand)
    # site that has no neighbour because it sits on a contig end
    a, parts = out[1], out[2] + out[3] + out[4] + out[5]
    assert np.all(a - parts >= 0) and np.all(a - parts <= 2)
    assert np.array_equal(out[0], out[7])            # idempotent
    assert np.array_equal(out[6], out[1])            # leading '.' is stripped
    assert a.sum() > 2_000_000                       # every A / T row with a confident call is counted
    exp = _oracle_counts(mg, "a", None, [("GATC", 1), ("GCAC......GTT", 2), ("A", 0), ("G[AG].GAAG[CT]", 5)])
    got = eng.score([M("GATC", 1), M("GCAC......GTT", 2), M("A", 0), M("G[AG].GAAG[CT]", 5)])
    assert np.array_equal(got, exp)
    eng.close()


def test_window_engine_matches_host_store():
    """nm_win_* (bit planes over windows) against the numpy window store, request for request."""
    from nanomotif_amd import search as ps
    from nanomotif_amd.engine import DeviceWindowStore, ScanEngine
    rng = np.random.default_rng(12)
    eng = ScanEngine(0)
    dev, host = DeviceWindowStore(eng), ps.HostWindowStore()
    keys = []
    for t, n in enumerate([1, 31, 32, 33, 1000, 70_001]):
        sets = rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), size=(n, 41), p=[0.35, 0.15, 0.2, 0.3])
        sets[rng.random((n, 41)) < 0.002] = 15                      # a few N positions
        sets[:, 20] = 1
        key = (f"bin{t}", "a")
        keys.append(key)
        dev.add_task(key, sets)
        host.add_task(key, sets.copy())
    motifs = [Motif("." * 20 + "A" + "." * 20, 20), Motif("." * 19 + "GATC" + "." * 18, 20), Motif("." * 18 + "[AG]CA.T" + "." * 18, 20),
              Motif("." * 10 + "T" + "." * 9 + "A" + "." * 19 + "C", 20), Motif("G" + "." * 19 + "A" + "." * 20, 20)]
    for rnd, kind in enumerate(["total", "pssm", "pssm", "remove", "pssm", "remove", "pssm", "pssm"]):
        m = motifs[rnd % len(motifs)]
        batch = [(k, ps.WinReq(kind, None if kind == "total" else m)) for k in keys]
        a, b = dev.execute(batch), host.execute(batch)
        for x, y in zip(a, b):
            if kind == "pssm":
                assert x[0] == y[0]
                assert (y[1] is None and x[0] == 0) or np.array_equal(x[1], y[1])
            else:
                assert x == y
    eng.close()


def test_fused_slots_with_uneven_groups(engine_cls):
    """Two mod types fused in one pass (light batch) where one (bin, mod type) group needs several LDS passes."""
    spec = synth.SynthSpec(n_contigs=24, total_bp=1_200_000, n_bins=12, mod_types=("a", "m"), seed=55, min_contig_bp=9_000)
    mg = synth.make_metagenome(spec)
    eng = engine_cls()
    _upload_metagenome(eng, mg, ("a", "m"))
    bins = sorted(set(mg.bin_names))
    zoo = synth.random_candidates(90, seed=21)
    cands, expect = [], []
    for bi, b in enumerate(bins):
        contigs = [i for i, x in enumerate(mg.bin_names) if x == b]
        for mt in ("a", "m"):
            these = [(s, p) for s, p, t in zoo if t == mt]
            these = these[:40] if (bi == 3 and mt == "a") else (these[:19] if (bi == 5 and mt == "m") else these[bi % 3:bi % 3 + 1 + bi % 2])
            if not these:
                continue
            expect.append(_oracle_counts(mg, mt, contigs, these))
            cands += [(Motif(s, p), mt, b) for s, p in these]
    out = eng.score(cands)
    st = eng.stats()
    assert st["last_workgroups"] < 2 * 16 * len(bins) * 2      # fused launch: one workgroup column, not one per slot
    assert np.array_equal(out, np.concatenate(expect))
    eng.close()


def test_failed_pileup_upload_leaves_no_half_written_slot():
    """A row error (duplicate position, bad strand, position outside the contig) is reported AND the slot goes back to
    'no pileup': scoring against it is refused with NM_ESTATE instead of counting half-written planes."""
    from nanomotif_amd._lib import NmScanError
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif
    eng = ScanEngine(0)
    eng.upload_assembly(["c0"], ["ACGATCGATCGGATCCA" * 40], ["b"])
    pos = np.array([3, 7, 12], np.uint32)
    ok = dict(contig_id=np.zeros(3, np.uint32), position=pos, strand=np.frombuffer(b"+++", np.uint8), fraction_mod=np.array([0.9, 0.1, 0.8]))
    eng.upload_pileup("a", **ok)
    good = eng.score([(Motif("GATC", 1), "a", "b")])
    for bad in (dict(ok, position=np.array([3, 7, 3], np.uint32)),                  # duplicate (contig, position, strand) among the methylated rows
                dict(ok, strand=np.frombuffer(b"+x+", np.uint8)),                    # strand label
                dict(ok, position=np.array([3, 7, 100000], np.uint32))):             # outside the contig
        for append in (False, True):
            eng.upload_pileup("a", **ok)
            with pytest.raises(NmScanError):
                eng.upload_pileup("a", append=append, **bad)
            with pytest.raises(NmScanError, match="no pileup uploaded"):
                eng.score([(Motif("GATC", 1), "a", "b")])
    eng.upload_pileup("a", **ok)
    assert np.array_equal(eng.score([(Motif("GATC", 1), "a", "b")]), good)
    eng.close()


_ALLOC = r"""
import sys
sys.path.insert(0, sys.argv[1])
from nanomotif_amd import _lib
from nanomotif_amd.engine import ScanEngine
eng = ScanEngine(0)
try:
    _lib.use_torch_allocator(True)
    print("SWITCHED-UNDER-LIVE-CTX")
except _lib.NmScanError as e:
    assert "nm_ctx alive" in str(e), e
eng.close()
_lib.use_torch_allocator(True)          # allowed again once the last ctx is gone
eng = ScanEngine(0)
eng.upload_assembly(["c0"], ["ACGT" * 100], ["b"])
try:
    _lib.use_torch_allocator(False)
    print("SWITCHED-UNDER-LIVE-CTX")
except _lib.NmScanError as e:
    assert "nm_ctx alive" in str(e), e
eng.close()
_lib.use_torch_allocator(False)
print("ALLOC-OK")
"""


def test_allocator_cannot_change_under_a_live_context(tmp_path):
    """nm_set_device_allocator is refused (NM_ESTATE) while any nm_ctx is alive: a block must go back to the allocator
    it came from.  Own process: the allocator pair is process-global."""
    import os
    import subprocess
    import sys
    script = tmp_path / "alloc.py"
    script.write_text(_ALLOC)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ALLOC-OK" in r.stdout and "SWITCHED" not in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_hit_positions_compaction_on_a_long_contig():
    """nm_hit_positions compacts on the device (popcount prefix + scatter): capacity smaller than the number of hits
    returns the first `capacity` positions and the full count; positions are ascending and equal to the oracle's."""
    import regex
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif
    rng = np.random.default_rng(5)
    seq = "".join(np.array(list("ACGT"))[rng.integers(0, 4, 700_000)])
    eng = ScanEngine(0)
    eng.upload_assembly(["c0"], [seq], ["b"])
    a_pos = np.array([m.start() for m in regex.finditer("A", seq)], np.uint32)
    frac = np.where(rng.random(len(a_pos)) < 0.5, 0.95, 0.05)
    eng.upload_pileup("a", np.zeros(len(a_pos), np.uint32), a_pos, np.full(len(a_pos), ord("+"), np.uint8), frac)
    starts = np.array([m.start() for m in regex.finditer("GA", seq, overlapped=True)], np.int64) + 1
    meth = set(a_pos[frac > 0.7].tolist())
    want = np.array([p for p in starts if p in meth], np.int64)
    got = eng.hit_positions("c0", "a", Motif("GA", 1), 0)             # engine grows the capacity from 65 536
    assert len(want) > 20_000 and np.array_equal(got, want)
    import ctypes as C
    b = eng.make_batch([(Motif("GA", 1), "a", 0)])
    n = C.c_uint64(0)
    out = np.full(100, -1, np.int64)
    from nanomotif_amd import _lib
    _lib.check(eng.lib.nm_hit_positions(eng.ctx, 0, eng.slot_of_mod["a"], int(b.lens[0]), int(b.modpos[0]), b.masks.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        0, out.ctypes.data_as(C.POINTER(C.c_int64)), 100, C.byref(n)))
    assert n.value == len(want) and np.array_equal(out, want[:100])
    eng.close()


def test_extra_wide_motifs_reach_95_positions_from_the_modified_base(engine_cls):
    """Search frames above 127 make children that reach up to 95 positions from the modified base: the extra-wide kernels
    (three halo words either side, offsets in [-96, 95]) against the oracle scan, compact and general state, plus the
    per-contig counters; one position further is refused."""
    from nanomotif_amd._lib import NmScanError
    from oracle.scan import score_candidates
    spec = synth.SynthSpec(n_contigs=5, total_bp=400_000, n_bins=2, mod_types=("a", "m"), seed=77, min_contig_bp=20_000)
    mg = synth.make_metagenome(spec)
    eng = engine_cls()
    _upload_metagenome(eng, mg, ("a", "m"))
    zoo = {"a": [("T" + "." * 70 + "A", 71), ("A" + "." * 90 + "G", 0), ("C" + "." * 94 + "A" + "." * 94 + "G", 95), ("G.." + "A" + "." * 64 + "[CT]", 3),
                 ("GATC", 1), ("[AG]" + "." * 80 + "T" + "." * 10 + "A", 92), ("T" + "." * 94 + "A", 0)],
           "m": [("C" + "." * 64 + "G", 0), ("A" + "." * 94 + "C", 95), ("G" + "." * 70 + "C" + "." * 70 + "[AT]", 71)]}
    for mt, motifs in zoo.items():
        for b in sorted(set(mg.bin_names)):
            idx = [i for i, x in enumerate(mg.bin_names) if x == b]
            pile, seqs = oracle_bin_inputs(mg, mt, contigs=idx)
            want = score_candidates(pile, seqs, motifs)
            got = eng.score([(Motif(s, p), mt, b) for s, p in motifs])
            assert np.array_equal(got, want), (mt, b, got.tolist(), want.tolist())
            per = eng.score_per_contig([(Motif(s, p), mt, b) for s, p in motifs])
            assert np.array_equal(np.array([t.sum(axis=0) for _, t in per]), want)
    assert eng.score([(Motif("T" + "." * 70 + "A", 71), "a", "bin_000")]).sum() > 0
    # nm_hit_positions over the same reach (save_motif_positions=True, find_motifs_bin.py:1322-1329): all three kernel widths
    from oracle.scan import motif_model_contig
    from oracle.model import BetaBernoulliModel
    from oracle.motif import Motif as OracleMotif
    keys = ("index_meth_fwd", "index_nonmeth_fwd", "index_meth_rev", "index_nonmeth_rev")
    for mt, motifs in zoo.items():
        pile, seqs = oracle_bin_inputs(mg, mt)
        for s, p in motifs[:4] + [("G" + "." * 40 + "A", 41) if mt == "a" else ("C" + "." * 40 + "T", 0)]:
            for ci in (0, 3):
                _, want = motif_model_contig(pile[mg.names[ci]], seqs[mg.names[ci]], BetaBernoulliModel(), OracleMotif(s, p), save_motif_positions=True)
                for which, k in enumerate(keys):
                    assert eng.hit_positions(mg.names[ci], mt, Motif(s, p), which).tolist() == want[k].tolist(), (s, p, ci, k)
    with pytest.raises(NmScanError, match="outside \\[-96, 95\\]"):
        eng.hit_positions(mg.names[0], "a", Motif("A" + "." * 95 + "G", 0), 0)
    with pytest.raises(NmScanError, match="the engine reaches 95"):           # the fast kernels; ScanEngine.score routes such a list to nm_score_batch_wide
        eng.score(eng.make_batch([(Motif("A" + "." * 95 + "G", 0), "a", "bin_000")]))
    eng.close()


def test_candidates_of_frames_above_191_are_scored_by_the_wide_entry(engine_cls):
    """nm_score_batch_wide: any reach from the modified base (a --search_frame_size above 191 makes such children,
    find_motifs_bin.py:110-130).  (1) on candidates the fast kernels take, the same counts as nm_score_batch; (2) far-reaching
    ones against the oracle scan, including motifs longer than the shortest contig's margin (a site whose motif would leave
    its contig is no site: a regex match never leaves the string) and contigs shorter than the motif."""
    import ctypes as C
    from nanomotif_amd import _lib
    from nanomotif_amd._lib import NmScanError
    from oracle.scan import score_candidates
    spec = synth.SynthSpec(n_contigs=7, total_bp=300_000, n_bins=2, mod_types=("a", "m"), seed=78, min_contig_bp=300)
    mg = synth.make_metagenome(spec)
    eng = engine_cls()
    _upload_metagenome(eng, mg, ("a", "m"))
    ptr = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    # (1) the zoo of the fast kernels through the wide entry
    rng = np.random.default_rng(5)
    cands = [(Motif(s, p), mt, b) for s, p, mt in synth.random_candidates(300, seed=21, mod_types=("a", "m")) for b in ("bin_000", "bin_001")]
    batch = eng.make_batch(cands)
    want = eng.score(batch)
    got = np.zeros_like(want)
    _lib.check(eng.lib.nm_score_batch_wide(eng.ctx, len(batch), ptr(batch.bins, C.c_uint32), ptr(batch.slots, C.c_uint8),
                                           ptr(batch.lens.astype(np.uint16), C.c_uint16), ptr(batch.modpos.astype(np.uint16), C.c_uint16),
                                           ptr(batch.offsets, C.c_uint32), ptr(batch.masks, C.c_uint8), ptr(got, C.c_int64)))
    assert want.sum() > 0 and np.array_equal(got, want)
    # (2) far reaches against the oracle
    zoo = {"a": [("A" + "." * 95 + "G", 0), ("T" + "." * 150 + "A", 151), ("G" + "." * 199 + "A" + "." * 199 + "[CT]", 200), ("GATC", 1),
                 ("A" + "." * 400, 0), ("." * 200 + "A" + "." * 200, 200), ("[AG]" + "." * 250 + "A" + "." * 30 + "T", 251),
                 ("A" + "." * 299 + "[ACG]", 0), ("C" + "." * 1000 + "A", 1001), ("T.A" + "." * 120 + "[GT]", 2)],
           "m": [("C" + "." * 120 + "G", 0), ("A" + "." * 300 + "C", 301), ("G" + "." * 170 + "C" + "." * 170 + "[AT]", 171), ("CC[AT]GG", 1)]}
    any_hits = 0
    for mt, motifs in zoo.items():
        for b in sorted(set(mg.bin_names)):
            idx = [i for i, x in enumerate(mg.bin_names) if x == b]
            pile, seqs = oracle_bin_inputs(mg, mt, contigs=idx)
            want = score_candidates(pile, seqs, motifs)
            got = eng.score([(Motif(s, p), mt, b) for s, p in motifs])
            assert np.array_equal(got, want), (mt, b, got.tolist(), want.tolist())
            any_hits += int(want[:3].sum())
    assert any_hits > 0
    with pytest.raises(NmScanError, match="outside 1..4095"):
        eng.score([(Motif("A" + "." * 4095 + "G", 0), "a", "bin_000")])
    eng.close()

"""nm_set_score_lanes / nm_sync (include/nmscan.h): asynchronous batches alternate between two scoring streams and walk a
ring of four staging pairs.  Whatever overlaps on the device, every table must hold exactly what the synchronous call
returns for its batch — also when the batches differ from call to call, when the ring wraps around several times, when
the all-reduce of the C ABI follows every launch, and when other entry points are called in between."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(n_contigs=12, total_bp=3_000_000, n_bins=4, mods=("a", "m"), seed=11):
    from nanomotif_amd import synth
    from nanomotif_amd.engine import ScanEngine
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=n_contigs, total_bp=total_bp, n_bins=n_bins, mod_types=mods, seed=seed))
    eng = ScanEngine(0)
    eng.upload_assembly(mg.names, [mg.contig_ascii(i) for i in range(n_contigs)], mg.bin_names)
    for mt in mods:
        for i in range(n_contigs):
            p = mg.contig_pileup(i, mt)
            eng.upload_pileup(mt, np.full(len(p["position"]), i, np.uint32), p["position"], p["strand"],
                              synth.pct_to_fraction(p["pct_hundredths"]), append=i > 0)
    return mg, eng


def _batches(mg, n_batches):
    """Different batches: heavy ones (random cfg 5 candidates, 8 per bin and mod type), light sibling rounds, one-candidate ones."""
    from nanomotif_amd import synth
    from nanomotif_amd.motif import Motif
    bins = sorted(set(mg.bin_names))
    out = []
    for k in range(n_batches):
        if k % 5 == 3:            # wide: constraints more than 31 positions from the modified base (the four-word-group kernels)
            out.append([(Motif("G" + "." * 35 + can + "T." + "C", 36), mt, bn) for bn in bins for mt, can in (("a", "A"), ("m", "C"))] +
                       [(Motif("GATC", 1), "a", bins[0])])
        elif k % 5 == 4:          # general state planes: the modified position is not the canonical base of its mod type
            out.append([(Motif("G[AT]TC", 1), "a", bn) for bn in bins] + [(Motif("CC.GG", 2), "m", bins[-1]), (Motif("TTAA", 0), "a", bins[0])])
        elif k % 3 == 0:
            raw = synth.random_candidates(16 * len(bins), seed=100 + k, mod_types=mg.spec.mod_types)
            out.append([(Motif(s, p), mt, bins[(j // 2) % len(bins)]) for j, (s, p, mt) in enumerate(raw)])
        elif k % 3 == 1:
            core = list("." * 21)
            out.append([(Motif("".join(core[:8] + [b] + ["."] + [can] + list("T.G") + core[14:]), 10), mt, bn)
                        for bn in bins for mt, can in (("a", "A"), ("m", "C")) for b in "ACGT"[: 2 + k % 3]])
        else:
            out.append([(Motif("GATC", 1), "a", bins[k % len(bins)])])
    return out


def test_two_lanes_tables_equal_synchronous_scores():
    import torch
    mg, eng = _engine()
    batches = _batches(mg, 11)
    want = [eng.score(b) for b in batches]
    made = [eng.make_batch(b) for b in batches]
    for lanes in (2, 1, 2):
        tables = [torch.full((len(b), 2), -7, dtype=torch.int64, device="cuda:0") for b in batches]
        eng.set_score_lanes(lanes)
        for rep in range(3):                              # 33 launches: the ring of four pairs wraps eight times
            for k, b in enumerate(made):
                eng.score_into_device(b, tables[k].data_ptr())
                if rep == 1 and k == 5:
                    assert np.array_equal(eng.score(batches[2]), want[2])     # a synchronous call in between joins the lanes
        eng.sync()
        for k in range(len(batches)):
            assert np.array_equal(tables[k].cpu().numpy(), want[k]), (lanes, k)
    eng.set_score_lanes(1)
    eng.close()


def test_two_lanes_with_the_allreduce_step_and_other_entry_points():
    import torch
    from nanomotif_amd import synth
    mg, eng = _engine(seed=12)
    eng.comm_init(0, 1, eng.comm_unique_id())
    batches = _batches(mg, 6)
    want = [eng.score(b) for b in batches]
    made = [eng.make_batch(b) for b in batches]
    tables = [torch.zeros((len(b), 2), dtype=torch.int64, device="cuda:0") for b in batches]
    eng.set_score_lanes(2)
    for rep in range(4):
        for k, b in enumerate(made):
            slot = k % 4
            eng.comm_wait(slot)
            eng.score_into_device(b, tables[k].data_ptr())
            eng.allreduce_counts_device(tables[k].data_ptr(), tables[k].numel(), slot)
    eng.comm_sync()
    eng.sync()
    for k in range(len(batches)):
        assert np.array_equal(tables[k].cpu().numpy(), want[k]), k
    # replacing a classification while launches are in flight: the upload waits for both lanes first
    for k, b in enumerate(made):
        eng.score_into_device(b, tables[k].data_ptr())
    p = mg.contig_pileup(0, "a")
    eng.upload_pileup("a", np.zeros(len(p["position"]), np.uint32), p["position"], p["strand"], synth.pct_to_fraction(p["pct_hundredths"]))
    torch.cuda.synchronize()
    for k in range(len(batches)):
        assert np.array_equal(tables[k].cpu().numpy(), want[k]), k      # the launches saw the OLD planes
    assert not np.array_equal(eng.score(batches[0]), want[0])             # the new classification holds contig 0 only
    eng.close()


def test_lanes_argument_is_checked():
    from nanomotif_amd._lib import NmScanError
    from nanomotif_amd.engine import ScanEngine
    eng = ScanEngine(0)
    with pytest.raises(NmScanError, match="lanes must be 1 or 2"):
        eng.set_score_lanes(3)
    eng.set_score_lanes(2)
    eng.set_score_lanes(1)
    eng.sync()
    eng.close()


def test_two_lanes_order_launches_that_share_one_count_table():
    """Launches that write the SAME d_out are ordered by the library whatever lane they land on (a synchronous call in
    between flips the lane parity): the table always holds the counts of the LAST batch submitted to it."""
    import torch
    mg, eng = _engine(seed=13)
    batches = _batches(mg, 6)
    n = max(len(b) for b in batches)
    padded = [b + [b[0]] * (n - len(b)) for b in batches]            # equal length: one table serves every batch
    want = [eng.score(b) for b in padded]
    made = [eng.make_batch(b) for b in padded]
    table = torch.zeros((n, 2), dtype=torch.int64, device="cuda:0")
    eng.set_score_lanes(2)
    for rep in range(5):
        for k, b in enumerate(made):
            eng.score_into_device(b, table.data_ptr())
            if (rep, k) in ((1, 2), (3, 0)):
                eng.sync()
                assert np.array_equal(table.cpu().numpy(), want[k]), (rep, k)
                assert np.array_equal(eng.score(padded[1]), want[1])        # host-output call: advances the ring by one
    eng.sync()
    assert np.array_equal(table.cpu().numpy(), want[len(made) - 1])
    eng.set_score_lanes(1)
    eng.close()


def test_begin_end_halves_equal_the_synchronous_calls():
    """nm_score_batch_begin / _end and nm_win_batch_w_begin / _end (what a round of nm_search_run keeps in flight together):
    same counts as the one-call forms, with other calls on the ctx in between; a second _begin before _end and an _end
    without _begin are NM_ESTATE; _end(NULL) drops the batch."""
    import ctypes as C
    from nanomotif_amd import _lib
    NM_ESTATE = -3
    mg, eng = _engine()
    lib = eng.lib
    batches = _batches(mg, 7)
    want = [eng.score(b) for b in batches]
    # a window task to run requests on while scoring batches are open
    rng = np.random.default_rng(3)
    sets = rng.choice(np.array([1, 2, 4, 8], dtype=np.uint8), size=(500, 41))
    tid = C.c_uint32(0)
    _lib.check(lib.nm_win_add_task(eng.ctx, 500, 41, sets.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(tid)))
    req_task = np.array([tid.value, tid.value], dtype=np.uint32)
    req_kind = np.zeros(2, dtype=np.uint8)
    req_sets = np.full((2, 64), 15, dtype=np.uint8)
    req_sets[1, 20] = 1
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    win_want = np.zeros((2, 2 + 4 * 64), dtype=np.int32)
    _lib.check(lib.nm_win_batch_w(eng.ctx, 2, p(req_task, C.c_uint32), p(req_kind, C.c_uint8), p(req_sets, C.c_uint8), 64, p(win_want, C.c_int32)))
    assert win_want[0, 0] == 500 and 0 < win_want[1, 0] < 500
    for k, b in enumerate(batches):
        made = eng.make_batch(b)
        args = eng._batch_args(made)
        _lib.check(lib.nm_win_batch_w_begin(eng.ctx, 2, p(req_task, C.c_uint32), p(req_kind, C.c_uint8), p(req_sets, C.c_uint8), 64))
        _lib.check(lib.nm_score_batch_begin(eng.ctx, *args))
        assert lib.nm_score_batch_begin(eng.ctx, *args) == NM_ESTATE
        assert lib.nm_win_batch_w_begin(eng.ctx, 2, p(req_task, C.c_uint32), p(req_kind, C.c_uint8), p(req_sets, C.c_uint8), 64) == NM_ESTATE
        win_got = np.zeros_like(win_want)
        _lib.check(lib.nm_win_batch_w_end(eng.ctx, p(win_got, C.c_int32)))
        assert np.array_equal(win_got, win_want)
        if k % 2:
            assert np.array_equal(eng.score(batches[0]), want[0])           # another scoring call while the batch is open
        got = np.full((len(b), 2), -1, dtype=np.int64)
        _lib.check(lib.nm_score_batch_end(eng.ctx, p(got, C.c_int64)))
        assert np.array_equal(got, want[k]), k
    assert lib.nm_score_batch_end(eng.ctx, None) == NM_ESTATE
    assert lib.nm_win_batch_w_end(eng.ctx, None) == NM_ESTATE
    _lib.check(lib.nm_score_batch_begin(eng.ctx, *eng._batch_args(eng.make_batch(batches[1]))))
    _lib.check(lib.nm_score_batch_end(eng.ctx, None))                       # dropped
    assert np.array_equal(eng.score(batches[1]), want[1])
    eng.close()

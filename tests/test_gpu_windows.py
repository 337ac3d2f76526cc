"""Window extraction on the device (nm_win_add_task_rows / nm_contig_base_counts / nm_bg_counts) against the host numpy
path of the same interface (search.extract_windows, itself pinned to the reference by golden g3/g4)."""
import random

import numpy as np
import pytest

from nanomotif_amd import search as ps
from nanomotif_amd.motif import Motif

pytestmark = pytest.mark.gpu


def _assembly(rng, lengths, n_frac=0.0005):
    seqs = {}
    for i, n in enumerate(lengths):
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.3])
        s[rng.random(n) < n_frac] = ord("N")
        if i % 3 == 0:
            s[:25] = ord("A")                         # valid centres inside the edge padding must not be counted
            s[-25:] = ord("C")
        seqs[f"contig_{i:03d}"] = s
    return seqs


def _rows(rng, seqs, base):
    plus, minus = {}, {}
    comp = {"A": "T", "C": "G"}[base]
    for name, s in seqs.items():
        p = np.flatnonzero(s == ord(base))
        m = np.flatnonzero(s == ord(comp))
        plus[name] = np.sort(rng.choice(p, size=min(len(p), max(1, len(s) // 200)), replace=False)).astype(np.int64)
        minus[name] = np.sort(rng.choice(m, size=min(len(m), max(1, len(s) // 300)), replace=False)).astype(np.int64)
        # rows inside the edge padding exist in real pileups: both paths must drop them
        plus[name] = np.unique(np.concatenate([plus[name], [0, 3, 20, len(s) - 21, len(s) - 20, len(s) - 1]]))
    return plus, minus


@pytest.mark.parametrize("padding", [20, 7])
def test_device_extraction_matches_host(padding):
    from nanomotif_amd.engine import DeviceWindowExtractor, DeviceWindowStore, ScanEngine
    rng = np.random.default_rng(5)
    lengths = [6_000, 8_191, 8_192, 8_193, 16_400, 70_000, 5_500, 300_000]
    seqs = _assembly(rng, lengths)
    names = list(seqs)
    bins = [f"bin{i % 3}" for i in range(len(names))]
    eng = ScanEngine(0)
    eng.upload_assembly(names, [seqs[n] for n in names], bins)
    assert eng.other_letters() == 0
    W = 2 * padding + 1
    # valid-start counts
    for base in "ACGT":
        got = eng.contig_base_counts(base, padding)
        want = [int((seqs[n][padding:len(seqs[n]) - padding] == ord(base)).sum()) for n in names]
        assert got.tolist() == want
    n_valid = {b: dict(zip(names, eng.contig_base_counts(b, padding).tolist())) for b in "AC"}
    lengths_of = {n: len(seqs[n]) for n in names}
    dev_store, host_store = DeviceWindowStore(eng), ps.HostWindowStore()
    ext = DeviceWindowExtractor(eng, dev_store, lengths_of, n_valid, padding)
    keys, host_pssm, task_rows = [], {}, {}
    for b in sorted(set(bins)):
        members = [n for n, x in zip(names, bins) if x == b]
        for mt, base in (("a", "A"), ("m", "C")):
            plus, minus = _rows(rng, {n: seqs[n] for n in members}, base)
            key = (b, mt)
            random.seed(11)
            host = ps.extract_windows(seqs, plus, minus, mt, padding)
            state_after_host = random.getstate()
            random.seed(11)
            assert ext.plan(key, plus, minus, mt)
            assert random.getstate() == state_after_host          # same RNG consumption
            host_store.add_task(key, host[0])
            host_pssm[key] = host[1]
            keys.append(key)
            task_rows[key] = (plus, minus, members)
    dev_pssm = ext.finish()
    for key in keys:
        assert np.array_equal(dev_pssm[key], host_pssm[key]), key          # counts / n: bitwise equal
    # the same tasks without row lists: the rows are the bits of the methylated-state planes (plan_contigs)
    for mt in ("a", "m"):
        cid, pos, strand, frac = [], [], [], []
        for key, (plus, minus, members) in task_rows.items():
            if key[1] != mt:
                continue
            for name in members:
                for arr, ch in ((plus[name], "+"), (minus[name], "-")):
                    cid.append(np.full(len(arr), eng.contig_index[name])); pos.append(arr)
                    strand.append(np.full(len(arr), ord(ch), np.uint8)); frac.append(np.full(len(arr), 0.9))
                # rows that are not confidently methylated must not become windows
                free = np.setdiff1d(np.arange(30, 300), np.concatenate([plus[name], minus[name]]))
                cid.append(np.full(len(free), eng.contig_index[name])); pos.append(free)
                strand.append(np.full(len(free), ord("+"), np.uint8)); frac.append(np.where(free % 2 == 0, 0.1, 0.5))
        eng.upload_pileup(mt, np.concatenate(cid), np.concatenate(pos), np.concatenate(strand), np.concatenate(frac))
    row_counts = {}
    for mt in ("a", "m"):
        got = eng.methylated_row_counts(mt, padding)
        row_counts[mt] = dict(zip(names, got.tolist()))
        for key, (plus, minus, members) in task_rows.items():
            if key[1] == mt:
                for name in members:
                    L = len(seqs[name])
                    want = [int(((a > padding) & (a < L - padding)).sum()) for a in (plus[name], minus[name])]
                    assert row_counts[mt][name] == want, (key, name)
    ext2 = DeviceWindowExtractor(eng, dev_store, lengths_of, n_valid, padding, row_counts=row_counts)
    for key in keys:
        random.seed(11)
        assert ext2.plan_contigs((key, "planes"), task_rows[key][2], key[1])
    pssm2 = ext2.finish()
    for key in keys:
        assert np.array_equal(pssm2[(key, "planes")], host_pssm[key]), key
    pad = padding
    motifs = [Motif("." * pad + "A" + "." * pad, pad), Motif("." * pad + "C" + "." * pad, pad),
              Motif("." * (pad - 1) + "GATC" + "." * (pad - 2), pad), Motif("." * (pad - 2) + "[AG]CA.T" + "." * (pad - 2), pad),
              Motif("T" + "." * (pad - 1) + "C" + "." * (pad - 1) + "G", pad)]
    for rnd, kind in enumerate(["total", "pssm", "pssm", "pssm", "remove", "pssm", "remove", "pssm", "pssm"]):
        m = motifs[rnd % len(motifs)]
        batch = [(k, ps.WinReq(kind, None if kind == "total" else m)) for k in keys]
        b = host_store.execute(batch)
        for rename in (lambda k: k, lambda k: (k, "planes")):
            a = dev_store.execute([(rename(k), r) for k, r in batch])
            for x, y in zip(a, b):
                if kind == "pssm":
                    assert x[0] == y[0]
                    assert (y[1] is None and x[0] == 0) or np.array_equal(x[1], y[1])
                else:
                    assert x == y
    eng.close()


def test_task_without_windows_and_bad_rows():
    from nanomotif_amd import _lib
    from nanomotif_amd.engine import DeviceWindowExtractor, DeviceWindowStore, ScanEngine
    rng = np.random.default_rng(9)
    seqs = _assembly(rng, [9_000, 12_000], n_frac=0)
    names = list(seqs)
    eng = ScanEngine(0)
    eng.upload_assembly(names, [seqs[n] for n in names], ["b", "b"])
    store = DeviceWindowStore(eng)
    n_valid = {b: dict(zip(names, eng.contig_base_counts(b, 20).tolist())) for b in "AC"}
    ext = DeviceWindowExtractor(eng, store, {n: len(seqs[n]) for n in names}, n_valid, 20)
    # second contig has rows, but all within the edge padding: the reference gives up on the whole task (:662-664)
    plus = {names[0]: np.array([100, 200], dtype=np.int64), names[1]: np.array([5, 11_995], dtype=np.int64)}
    random.seed(3)
    host = ps.extract_windows(seqs, plus, {}, "a", 20)
    st = random.getstate()
    random.seed(3)
    assert host is None and ext.plan(("b", "a"), plus, {}, "a") is False
    assert random.getstate() == st
    assert ext.finish() == {}
    # the C ABI refuses rows the edge filter should have removed, and ranks beyond the valid starts
    with pytest.raises(_lib.NmScanError, match="within 20 bp"):
        store.add_task_rows("x", [0], [20], [0], 20)
    import ctypes as C
    out = np.zeros((1, 4, 41), dtype=np.int64)
    sc, sr = np.zeros(1, np.uint32), np.array([n_valid["A"][names[0]]], dtype=np.uint32)
    begin = np.array([0, 1], dtype=np.uint64)
    rc = eng.lib.nm_bg_counts(eng.ctx, ord("A"), 20, 1, sc.ctypes.data_as(C.POINTER(C.c_uint32)), sr.ctypes.data_as(C.POINTER(C.c_uint32)),
                              1, begin.ctypes.data_as(C.POINTER(C.c_uint64)), out.ctypes.data_as(C.POINTER(C.c_int64)))
    assert rc != 0 and b"valid starts" in eng.lib.nm_last_error()
    eng.close()


def test_other_letters_are_counted():
    from nanomotif_amd.engine import ScanEngine
    eng = ScanEngine(0)
    eng.upload_assembly(["c1", "c2"], ["ACGTNNRYacgtn" * 700, "ACGT" * 3000 + "W"], ["b", "b"])
    assert eng.other_letters() == 2 * 700 + 1
    eng.close()


def test_background_counts_from_runs_equal_counts_from_samples():
    """nm_bg_counts_runs (samples as (contig, count) runs, contig column written on the device) against nm_bg_counts with
    the explicit per-sample contig column: three tasks with one, two and no runs; bad arguments come back as errors."""
    import ctypes as C
    from nanomotif_amd.engine import ScanEngine
    rng = np.random.default_rng(17)
    seqs = _assembly(rng, [30_000, 41_000, 25_000], n_frac=0.01)
    names = list(seqs)
    eng = ScanEngine(0)
    eng.upload_assembly(names, [seqs[n] for n in names], ["b", "b", "c"])
    u32, u64, i64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
    for base in "AC":
        nv = eng.contig_base_counts(base, 20)
        run_contig = np.array([0, 1, 2], dtype=np.uint32)               # task 0: contig 0; task 1: contigs 1, 2; task 2: nothing
        run_count = np.array([700, 3000, 450], dtype=np.uint32)
        task_run_begin = np.array([0, 1, 3, 3], dtype=np.uint32)
        ranks = np.concatenate([rng.choice(int(nv[c]), size=int(k), replace=False) for c, k in zip(run_contig, run_count)]).astype(np.uint32)
        got = np.zeros((3, 4, 41), dtype=np.int64)
        rc = eng.lib.nm_bg_counts_runs(eng.ctx, ord(base), 20, 3, run_contig.ctypes.data_as(u32), run_count.ctypes.data_as(u32),
                                       ranks.ctypes.data_as(u32), 3, task_run_begin.ctypes.data_as(u32), got.ctypes.data_as(i64))
        assert rc == 0, eng.lib.nm_last_error()
        want = np.zeros((3, 4, 41), dtype=np.int64)
        sc = np.repeat(run_contig, run_count).astype(np.uint32)
        begin = np.array([0, 700, 4150, 4150], dtype=np.uint64)
        rc = eng.lib.nm_bg_counts(eng.ctx, ord(base), 20, len(sc), sc.ctypes.data_as(u32), ranks.ctypes.data_as(u32), 3,
                                  begin.ctypes.data_as(u64), want.ctypes.data_as(i64))
        assert rc == 0, eng.lib.nm_last_error()
        assert np.array_equal(got, want) and got[0, :, 20].sum() == 700
        assert got[2].sum() == 0 and got[1, :, 20].sum() == 3450 and got[1][{"A": 0, "C": 3}[base], 20] == 3450
    bad = np.array([0, 2, 1, 3], dtype=np.uint32)                       # not monotone
    rc = eng.lib.nm_bg_counts_runs(eng.ctx, ord("A"), 20, 3, run_contig.ctypes.data_as(u32), run_count.ctypes.data_as(u32),
                                   ranks.ctypes.data_as(u32), 3, bad.ctypes.data_as(u32), got.ctypes.data_as(i64))
    assert rc != 0
    far = np.array([0, 1, 9], dtype=np.uint32)                          # contig 9 does not exist
    rc = eng.lib.nm_bg_counts_runs(eng.ctx, ord("A"), 20, 3, far.ctypes.data_as(u32), run_count.ctypes.data_as(u32),
                                   ranks.ctypes.data_as(u32), 3, task_run_begin.ctypes.data_as(u32), got.ctypes.data_as(i64))
    assert rc != 0 and b"contig 9" in eng.lib.nm_last_error()
    eng.close()


@pytest.mark.parametrize("freq,per_bin", [(0.01, False), (0.05, False), (0.01, True)])
def test_device_draws_equal_host_draws(freq, per_bin, monkeypatch):
    """nm_plan_windows consumes the generator streams of all tasks on the device (bg_draw_kernel: one wave per stream over
    ONE sequence of MT19937 outputs) when they start from the same state: background PSSMs, window counts and the state
    the interpreter's generator is left in must equal the host-thread replay (NM_HOST_DRAWS=1) exactly.  freq 0.01: the
    set branch of random.sample; freq 0.05: samples so large that n <= setsize for most contigs, the pool branch;
    per_bin: the bgzip task order, where the tasks of a bin share one stream."""
    import random
    from nanomotif_amd import synth
    from nanomotif_amd.e2e_synth import load_and_filter
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.main import device_window_pipeline
    from nanomotif_amd.pileup import MOD_TYPES
    import torch
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=400, total_bp=8_000_000, n_bins=40, mod_types=("a", "m"), seed=5))
    results = []
    for host in (True, False):
        if host:
            monkeypatch.setenv("NM_HOST_DRAWS", "1")
        else:
            monkeypatch.delenv("NM_HOST_DRAWS", raising=False)
        eng = ScanEngine(0)
        assembly, filtered, _ = load_and_filter(eng, mg, torch.device("cuda:0"), host_assembly=False)
        lengths = dict(zip(mg.names, (int(x) for x in mg.lengths)))
        store, extractor = device_window_pipeline(eng, lengths, list(mg.names), 20)
        extractor.freq = freq
        bins = {}
        for c, b in zip(mg.names, mg.bin_names):
            bins.setdefault(b, []).append(c)
        tasks = []
        for b in bins:
            for mt in ("a", "m"):
                names = filtered.present(bins[b], MOD_TYPES.index(mt))
                if names:
                    tasks.append(((b, mt), names, mt))
        assert len(tasks) >= 32
        random.seed(99)
        pssms = extractor.plan_all(tasks, seed=1, one_stream_per_bin=per_bin)
        results.append((pssms, random.getstate(), dict(store.totals)))
        eng.close()
    (p_host, s_host, t_host), (p_dev, s_dev, t_dev) = results
    assert s_host == s_dev and t_host == t_dev
    assert list(p_host) == list(p_dev) and len(p_host) >= 32
    for k in p_host:
        assert np.array_equal(p_host[k], p_dev[k]), k

"""Python host of the HIP scan engine: owns an ``nm_ctx`` and speaks in the reference's vocabulary
(contigs, bins, pileup rows, motifs).  Thin by design — packing, classification, scanning and counting all
happen on the GPU inside libnmscan.so; this file only marshals numpy buffers through ctypes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .motif import MOD_TYPE_TO_CANONICAL, Motif


MAX_MOTIF_LEN = 191        # include/nmscan.h NM_MAX_MOTIF_LEN: stripped length of a candidate on the fast scoring kernels ...
MAX_REACH = 95             # ... and its furthest position from the modified base; beyond: nm_score_batch_wide
MAX_DEVICE_WINDOW_WIDTH = 191      # NM_WIN_MAX_WIDTH 192 columns: wider search frames keep their windows on the host


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class CandidateBatch:
    """Flat SoA form of a list of candidates (include/nmscan.h: nm_score_batch arguments)."""

    def __init__(self, bins, slots, lens, modpos, offsets, masks):
        self.bins, self.slots, self.lens, self.modpos, self.offsets, self.masks = bins, slots, lens, modpos, offsets, masks

    def __len__(self):
        return len(self.bins)


class DeviceWindowStore:
    """Window requests of the lock-step search served by the engine's window kernels (nm_win_*): same interface as
    search.HostWindowStore.  Windows are bit planes over windows in HBM; a ``pssm`` request returns integer counts,
    a ``remove`` request clears the matching windows from the task's alive mask."""

    def __init__(self, engine, allreduce=None):
        """``allreduce``: callable summing an int32 numpy array over the ranks of a multi-GPU run, in which every rank
        holds only the windows of its own contigs (tasks added with ``add_task_rows``)."""
        self.engine = engine
        self.allreduce = allreduce
        self.task_id = {}
        self.totals = {}
        self.width = {}
        _lib.check(engine.lib.nm_win_clear(engine.ctx))

    def add_task_rows(self, key, contig, position, minus, pad, total=None):
        """Windows gathered on the device from the resident sequence planes (nm_win_add_task_rows): rows are
        (engine contig index, position, minus flag), already edge-filtered.  ``total``: number of windows of the task
        over all ranks (default: the rows given)."""
        contig = np.ascontiguousarray(contig, dtype=np.uint32)
        position = np.ascontiguousarray(position, dtype=np.uint32)
        minus = np.ascontiguousarray(minus, dtype=np.uint8)
        tid = C.c_uint32(0)
        _lib.check(self.engine.lib.nm_win_add_task_rows(self.engine.ctx, len(contig), _ptr(contig, C.c_uint32),
                                                        _ptr(position, C.c_uint32), _ptr(minus, C.c_uint8), int(pad), C.byref(tid)))
        self.task_id[key], self.width[key] = int(tid.value), 2 * int(pad) + 1
        self.totals[key] = int(len(contig) if total is None else total)

    def add_task(self, key, sets: np.ndarray):
        sets = np.ascontiguousarray(sets, dtype=np.uint8)
        tid = C.c_uint32(0)
        _lib.check(self.engine.lib.nm_win_add_task(self.engine.ctx, sets.shape[0], sets.shape[1], _ptr(sets, C.c_uint8), C.byref(tid)))
        self.task_id[key], self.totals[key], self.width[key] = int(tid.value), int(sets.shape[0]), int(sets.shape[1])

    def add_task_contigs(self, key, label, contigs, pad, total=None):
        """Windows around every confidently methylated row (pad < pos < len - pad) of the listed resident contigs,
        taken from the methylated-state planes of the classification ``label`` (nm_win_add_task_contigs)."""
        contigs = np.ascontiguousarray(contigs, dtype=np.uint32)
        tid, n = C.c_uint32(0), C.c_uint64(0)
        _lib.check(self.engine.lib.nm_win_add_task_contigs(self.engine.ctx, self.engine.slot_of_mod[label], len(contigs),
                                                           _ptr(contigs, C.c_uint32), int(pad), C.byref(tid), C.byref(n)))
        self.task_id[key], self.width[key] = int(tid.value), 2 * int(pad) + 1
        self.totals[key] = int(n.value if total is None else total)
        return int(n.value)

    def execute(self, batch):
        out = [None] * len(batch)
        dev = [(i, key, req) for i, (key, req) in enumerate(batch) if req.kind != "total"]
        for i, (key, req) in enumerate(batch):
            if req.kind == "total":
                out[i] = self.totals[key]
        if dev:
            n = len(dev)
            task = np.fromiter((self.task_id[k] for _, k, _ in dev), dtype=np.uint32, count=n)
            kind = np.fromiter((1 if r.kind == "remove" else 0 for _, _, r in dev), dtype=np.uint8, count=n)
            ws = (max(self.width[k] for _, k, _ in dev) + 63) // 64 * 64        # columns per row of sets / counts (nm_win_batch_w)
            sets = np.full((n, ws), 15, dtype=np.uint8)
            for j, (_, k, r) in enumerate(dev):
                sets[j, :self.width[k]] = r.motif.sets
            res = np.zeros((n, 2 + 4 * ws), dtype=np.int32)
            _lib.check(self.engine.lib.nm_win_batch_w(self.engine.ctx, n, _ptr(task, C.c_uint32), _ptr(kind, C.c_uint8),
                                                      _ptr(sets, C.c_uint8), ws, res.ctypes.data_as(C.POINTER(C.c_int32))))
            if self.allreduce is not None:
                res = self.allreduce(res)
            for j, (i, k, r) in enumerate(dev):
                if r.kind == "remove":
                    out[i] = (int(res[j, 0]), int(res[j, 1]))
                else:
                    w = self.width[k]
                    out[i] = (int(res[j, 0]), res[j, 2:].reshape(4, ws)[:, :w].astype(np.int64))
        return out


class DeviceWindowExtractor:
    """``extract_windows`` (find_motifs_bin.py:625-686) with the window gather and the background letter counts on the
    device: the host only draws the sample indices (``random.sample`` order and consumption are the reference's) and
    applies the edge filter to the row positions; sequences are read from the resident bit planes.

    ``lengths`` / ``n_valid[base]``: contig name -> length / number of valid sample starts (``contig_base_counts``),
    for EVERY contig of the data set; ``resident``: name -> engine contig index for the contigs this rank holds (all of
    them on one GPU).  Samples and rows of other contigs are left to their rank; ``allreduce_i64`` sums the counts."""

    MAX_SAMPLES_PER_CALL = 4 << 20

    def __init__(self, engine, store: DeviceWindowStore, lengths, n_valid, padding, resident=None, allreduce_i64=None,
                 background_sampling_frequency=0.01, row_counts=None):
        """``row_counts[mod type]``: contig name -> (plus, minus) confident rows inside the edge padding
        (``methylated_row_counts``), over all ranks — needed by ``plan_contigs`` only."""
        self.engine, self.store, self.lengths, self.n_valid, self.pad = engine, store, lengths, n_valid, int(padding)
        self.row_counts = row_counts
        self.resident = engine.contig_index if resident is None else resident
        self.allreduce_i64 = allreduce_i64
        self.freq = background_sampling_frequency
        self.tasks = []           # (key, base, n_bg, [sample contig arrays], [sample rank arrays])
        self._groups = []         # deferred draws: [(MT19937 state at the group's start, [(task index or None, nv, n_samples, keep)])]

    def _draw(self, rng, name, base, s_contig, s_rank) -> int:
        """The background sample of one contig (seq.py:202-225): draws its start ranks from ``rng`` (consumed on every
        rank alike), keeps them if the contig is resident here, returns the number of samples."""
        import math
        length = int(self.lengths[name])
        n_samples = int(max(math.ceil(length * self.freq), 50))
        if n_samples > length - (2 * self.pad + 1) + 1:
            raise ValueError("Too many samples requested for unique subsequences")
        nv = int(self.n_valid[base][name])
        if nv < n_samples:
            raise ValueError(f"Not enough subsequences with '{base}' in the middle (found {nv}, need {n_samples})")
        ranks = rng.sample(nv, n_samples)
        ci = self.resident.get(name)
        if ci is not None:
            s_contig.append(np.full(n_samples, ci, dtype=np.uint32))
            s_rank.append(ranks.astype(np.uint32))
        return n_samples

    def plan(self, key, plus_pos: dict, minus_pos: dict, mod_type: str) -> bool:
        """One task from explicit row positions (name -> ascending positions of the confident rows per strand);
        consumes the interpreter's RNG exactly like extract_windows.  False = no methylation windows."""
        from .search import NativeRandom
        base, pad = MOD_TYPE_TO_CANONICAL[mod_type], self.pad
        s_contig, s_rank, rows_c, rows_p, rows_m = [], [], [], [], []
        n_bg = total = 0
        empty = np.zeros(0, np.int64)
        with NativeRandom() as rng:
            for name in sorted(plus_pos.keys() | minus_pos.keys()):
                n_bg += self._draw(rng, name, base, s_contig, s_rank)
                length = int(self.lengths[name])
                p, m = plus_pos.get(name, empty), minus_pos.get(name, empty)
                p = p[(p > pad) & (p < length - pad)]
                m = m[(m > pad) & (m < length - pad)]
                if len(p) + len(m) == 0:
                    return False                         # find_motifs_bin.py:662-664
                total += len(p) + len(m)
                ci = self.resident.get(name)
                if ci is not None:
                    rows_c.append(np.full(len(p) + len(m), ci, dtype=np.uint32))
                    rows_p += [p, m]
                    rows_m.append(np.concatenate([np.zeros(len(p), np.uint8), np.ones(len(m), np.uint8)]))
        if total == 0 or n_bg == 0:
            return False
        cat = lambda xs, dt: np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dt)
        self.store.add_task_rows(key, cat(rows_c, np.uint32), cat(rows_p, np.uint32), cat(rows_m, np.uint8), pad, total=total)
        self.tasks.append((key, base, n_bg, cat(s_contig, np.uint32), cat(s_rank, np.uint32)))
        return True

    def begin_group(self, seed=None):
        """Called right after the caller (re)seeded ``random`` for the next task(s): the ``plan_contigs`` calls that follow
        form one generator stream whose draws are DEFERRED to ``finish`` — where the streams of all tasks are drawn on
        several host threads at once (a thousand tasks x 1 % of a Gbp is 1e7 sequential Mersenne-Twister draws).
        ``seed``: what ``random.seed`` was just called with — every task of a plain-pileup run starts from the same
        generator state (find_motifs_bin.py:152-171), read from the interpreter once."""
        import random
        cache = self.__dict__.setdefault("_seed_state", {})
        state = cache.get(seed) if seed is not None else None
        if state is None:
            state = np.array(random.getstate()[1], dtype=np.uint32)
            if seed is not None:
                cache[seed] = state
        self._groups.append((state, []))

    def plan_contigs(self, key, names, mod_type: str) -> bool:
        """``plan`` without row lists: the task's windows are all confident rows of the contigs ``names`` (those present
        in the filtered pileup of this mod type), read on the device from the methylated-state planes.  The background
        samples of all its contigs are drawn by ONE native call (same draws, same order as contig-by-contig)."""
        import math
        from .search import NativeRandom
        base, counts = MOD_TYPE_TO_CANONICAL[mod_type], self.row_counts[mod_type]
        names = sorted(names)
        W = 2 * self.pad + 1
        lengths = [int(self.lengths[n]) for n in names]
        n_samples = [int(max(math.ceil(L * self.freq), 50)) for L in lengths]
        nv = [int(self.n_valid[base][n]) for n in names]
        # the reference draws contig by contig and stops at the first contig without rows (find_motifs_bin.py:662-664):
        # only the contigs before it consume random numbers
        stop = next((i for i, n in enumerate(names) if int(counts[n][0]) + int(counts[n][1]) == 0), len(names))
        for i in range(min(stop + 1, len(names))):
            if n_samples[i] > lengths[i] - W + 1:
                raise ValueError("Too many samples requested for unique subsequences")
            if nv[i] < n_samples[i]:
                raise ValueError(f"Not enough subsequences with '{base}' in the middle (found {nv[i]}, need {n_samples[i]})")
        drawn = min(stop + 1, len(names))
        deferred = bool(self._groups)
        if deferred:
            ranks = None
        else:
            with NativeRandom() as rng:
                ranks = rng.sample_many(nv[:drawn], n_samples[:drawn])
        if stop < len(names):
            if deferred:
                self._groups[-1][1].append((None, nv[:drawn], n_samples[:drawn], None))      # consumes numbers, keeps none
            return False
        total = sum(int(counts[n][0]) + int(counts[n][1]) for n in names)
        n_bg = sum(n_samples)
        if total == 0 or n_bg == 0:
            return False
        res = [self.resident.get(n) for n in names]
        mine = [ci for ci in res if ci is not None]
        if deferred:
            keep = [i for i, ci in enumerate(res) if ci is not None]
            # the samples stay RUNS (contig, count) — nm_bg_counts_runs writes the per-sample contig column on the device
            runs = (np.asarray(mine, dtype=np.uint32), np.asarray([n_samples[i] for i in keep], dtype=np.uint32))
            self._groups[-1][1].append((len(self.tasks), nv, n_samples, keep))
            self.store.add_task_contigs(key, mod_type, mine, self.pad, total=total)
            self.tasks.append([key, base, n_bg, runs, None])
            return True
        if len(mine) == len(names):
            s_contig = np.repeat(np.asarray(mine, dtype=np.uint32), n_samples)
            s_rank = ranks
        else:
            off = np.concatenate([[0], np.cumsum(n_samples)])
            keep = [i for i, ci in enumerate(res) if ci is not None]
            s_contig = np.repeat(np.asarray([res[i] for i in keep], dtype=np.uint32), [n_samples[i] for i in keep]) if keep else np.zeros(0, np.uint32)
            s_rank = np.concatenate([ranks[off[i]:off[i + 1]] for i in keep]) if keep else np.zeros(0, np.uint32)
        self.store.add_task_contigs(key, mod_type, mine, self.pad, total=total)
        self.tasks.append((key, base, n_bg, s_contig, np.ascontiguousarray(s_rank, dtype=np.uint32)))
        return True

    def _draw_deferred(self):
        _draw_deferred_impl(self)

    def plan_all(self, tasks, seed, one_stream_per_bin=False) -> dict:
        """Every task of a single-GPU run in ONE native call (nm_plan_windows): ``tasks`` = [(key, contig names present in
        the filtered pileup of the task's mod type — or their engine indices, a uint32 array in the order of the sorted names —,
        mod_type)] in task order; ``seed``: what the reference seeds ``random``
        with before a task (plain pileup, find_motifs_bin.py:152-171) or before a bin's tasks (``one_stream_per_bin``: the
        bgzip path, :219-248 — consecutive tasks of one bin share a stream).  Windows are gathered, backgrounds drawn and
        counted on the device / on native threads; the interpreter's generator ends where the sequential run would leave it.
        Returns {key: background PSSM float64[4, W]} for the tasks that have methylation windows (the others are the
        reference's "No methylation sequences found")."""
        import random
        if not tasks:
            return {}
        n = len(tasks)
        slot = np.fromiter((self.engine.slot_of_mod[t[2]] for t in tasks), dtype=np.uint32, count=n)
        base = np.fromiter((ord(MOD_TYPE_TO_CANONICAL[t[2]]) for t in tasks), dtype=np.uint8, count=n)
        begin = np.zeros(n + 1, dtype=np.uint32)
        res = self.resident
        parts = [t[1] if isinstance(t[1], np.ndarray) else np.fromiter((res[x] for x in sorted(t[1])), dtype=np.uint32, count=len(t[1]))
                 for t in tasks]
        np.cumsum(np.fromiter((len(x) for x in parts), dtype=np.uint32, count=n), out=begin[1:])
        ids = np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32)
        if one_stream_per_bin:
            group = np.zeros(n, dtype=np.uint32)
            g, last_bin = -1, object()
            for k, (key, _, _) in enumerate(tasks):
                if key[0] != last_bin:
                    g, last_bin = g + 1, key[0]
                group[k] = g
            n_groups = g + 1
        else:
            group, n_groups = np.arange(n, dtype=np.uint32), n
        version, _, gauss = random.getstate()
        random.seed(seed)
        init = np.array(random.getstate()[1], dtype=np.uint32)
        W = 2 * self.pad + 1
        status = np.zeros(n, dtype=np.uint8)
        window = np.zeros(n, dtype=np.uint32)
        n_win, n_bg = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
        counts = np.zeros((n, 4, W), dtype=np.int64)
        final = np.zeros(625, dtype=np.uint32)
        rc = self.engine.lib.nm_plan_windows(self.engine.ctx, n, _ptr(slot, C.c_uint32), _ptr(base, C.c_uint8), _ptr(group, C.c_uint32),
                                             _ptr(begin, C.c_uint32), _ptr(ids, C.c_uint32), self.pad, float(self.freq), n_groups,
                                             _ptr(init, C.c_uint32), 1, _ptr(status, C.c_uint8), _ptr(window, C.c_uint32),
                                             _ptr(n_win, C.c_uint64), _ptr(n_bg, C.c_uint64), _ptr(counts, C.c_int64), _ptr(final, C.c_uint32))
        if rc:
            msg = self.engine.lib.nm_last_error().decode()
            if msg.startswith("Too many samples") or msg.startswith("Not enough subsequences"):
                raise ValueError(msg)                       # seq.py:208-219 raises these
            _lib.check(rc)
        random.setstate((version, tuple(final.tolist()), gauss))
        out = {}
        # (one division for all tasks: a thousand small numpy calls cost more than the arithmetic; true division of int64 by
        #  float64 like the per-task expression counts[k] / float(n_bg[k]))
        pssm = counts / np.where(n_bg == 0, 1, n_bg).astype(np.float64)[:, None, None]
        ok, win, nw = (status == 0).tolist(), window.tolist(), n_win.tolist()
        task_id, width, totals = self.store.task_id, self.store.width, self.store.totals
        for k, (key, _, _) in enumerate(tasks):
            if not ok[k]:
                continue
            task_id[key], width[key], totals[key] = win[k], W, nw[k]
            out[key] = pssm[k]
        return out

    def finish(self) -> dict:
        """Background PSSM (``background_sequences.pssm()``, float64[4, W]) of every planned task."""
        W = 2 * self.pad + 1
        if self._groups:
            self._draw_deferred()
        counts = np.zeros((len(self.tasks), 4, W), dtype=np.int64)
        n_of = [len(t[4]) for t in self.tasks]                 # samples this rank holds of every task
        for base in sorted({t[1] for t in self.tasks}):
            idx = [i for i, t in enumerate(self.tasks) if t[1] == base]
            lo = 0
            while lo < len(idx):
                hi, n = lo, 0
                while hi < len(idx) and (hi == lo or n + n_of[idx[hi]] <= self.MAX_SAMPLES_PER_CALL):
                    n += n_of[idx[hi]]
                    hi += 1
                part = idx[lo:hi]
                sr = np.concatenate([self.tasks[i][4] for i in part]) if n else np.zeros(0, np.uint32)
                out = np.zeros((len(part), 4, W), dtype=np.int64)
                if all(isinstance(self.tasks[i][3], tuple) for i in part):
                    run_begin = np.zeros(len(part) + 1, dtype=np.uint32)
                    np.cumsum([len(self.tasks[i][3][0]) for i in part], out=run_begin[1:])
                    rc = np.ascontiguousarray(np.concatenate([self.tasks[i][3][0] for i in part]), dtype=np.uint32)
                    rn = np.ascontiguousarray(np.concatenate([self.tasks[i][3][1] for i in part]), dtype=np.uint32)
                    _lib.check(self.engine.lib.nm_bg_counts_runs(self.engine.ctx, ord(base), self.pad, len(rc), _ptr(rc, C.c_uint32),
                                                                 _ptr(rn, C.c_uint32), _ptr(sr, C.c_uint32), len(part),
                                                                 _ptr(run_begin, C.c_uint32), _ptr(out, C.c_int64)))
                else:
                    begin = np.zeros(len(part) + 1, dtype=np.uint64)
                    np.cumsum([n_of[i] for i in part], out=begin[1:])
                    sc = np.concatenate([self.tasks[i][3] if not isinstance(self.tasks[i][3], tuple) else np.repeat(*self.tasks[i][3])
                                         for i in part]) if n else np.zeros(0, np.uint32)
                    _lib.check(self.engine.lib.nm_bg_counts(self.engine.ctx, ord(base), self.pad, n, _ptr(sc, C.c_uint32), _ptr(sr, C.c_uint32),
                                                            len(part), _ptr(begin, C.c_uint64), _ptr(out, C.c_int64)))
                counts[part] = out
                lo = hi
        if self.allreduce_i64 is not None:
            counts = self.allreduce_i64(counts)
        res = {t[0]: counts[i] / t[2] for i, t in enumerate(self.tasks)}
        self.tasks = []
        return res


def _draw_deferred_impl(extractor):
    import random
    groups = extractor._groups
    calls = [c for _, cs in groups for c in cs]
    group_off = np.zeros(len(groups) + 1, dtype=np.uint64)
    np.cumsum([sum(len(c[1]) for c in cs) for _, cs in groups], out=group_off[1:])
    ns = np.ascontiguousarray(np.concatenate([np.asarray(c[1], dtype=np.uint64) for c in calls]) if calls else np.zeros(0, np.uint64))
    ks = np.ascontiguousarray(np.concatenate([np.asarray(c[2], dtype=np.uint64) for c in calls]) if calls else np.zeros(0, np.uint64))
    init = np.ascontiguousarray(np.stack([st for st, _ in groups]))
    out = np.empty(max(int(ks.sum()), 1), dtype=np.uint32)
    final = np.zeros(625, dtype=np.uint32)
    _lib.check(extractor.engine.lib.nm_py_random_sample_groups(len(groups), _ptr(init, C.c_uint32), _ptr(group_off, C.c_uint64), _ptr(ns, C.c_uint64),
                                                               _ptr(ks, C.c_uint64), _ptr(out, C.c_uint32), _ptr(final, C.c_uint32)))
    off = np.concatenate([[0], np.cumsum(ks)]).astype(np.int64)
    at = 0
    for task, nv, n_samples, keep in calls:
        m = len(nv)
        if task is not None:
            if len(keep) == m:
                ranks = out[off[at]:off[at + m]]
            else:
                ranks = np.concatenate([out[off[at + i]:off[at + i + 1]] for i in keep]) if keep else np.zeros(0, np.uint32)
            extractor.tasks[task][4] = np.ascontiguousarray(ranks, dtype=np.uint32)
        at += m
    version, _, gauss = random.getstate()
    random.setstate((version, tuple(final.tolist()), gauss))       # where a sequential run would have left the interpreter
    extractor._groups = []


class ScanEngine:
    """One engine per GPU / process.  Mirrors what ``motif_model_bin`` needs (find_motifs_bin.py:1265-1283):
    the bin's contig sequences and the (bin, mod_type) pileup, but resident in HBM across calls."""

    def __init__(self, device: int = 0, ctx=None):
        """``ctx``: an ``nm_ctx *`` (ctypes.c_void_p) on ``device`` that somebody else created — the engine takes it over and destroys it
        in ``close`` (the command line makes it on a thread while the interpreter still imports, __main__.py)."""
        self.lib = _lib.load()
        if ctx is not None:
            self.ctx = ctx
        else:
            self.ctx = C.c_void_p()
            _lib.check(self.lib.nm_ctx_create(int(device), C.byref(self.ctx)))
        self.device = int(device)
        self.contig_index = {}
        self.contig_names = []
        self.contig_lengths = None
        self.contig_bin = None
        self.bin_index = {}
        self.bin_names = []
        self.slot_of_mod = {}
        self.comm_world = 0

    def close(self):
        if self.ctx:
            self.lib.nm_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def use_stream(self, hip_stream_handle: int | None):
        """Run engine work on another HIP stream (e.g. a ``torch.cuda.Stream().cuda_stream``); None restores the
        engine's own stream.  The legacy default stream (handle 0) cannot be expressed through nm_set_stream —
        create a side stream and make it torch's current stream instead (see bench.py)."""
        if hip_stream_handle == 0:
            raise ValueError("handle 0 is the legacy default stream: pass a side stream's handle, or None")
        _lib.check(self.lib.nm_set_stream(self.ctx, C.c_void_p(hip_stream_handle or 0)))

    # ------------------------------------------------------------------ assembly
    def upload_assembly(self, names, sequences, bin_of_contig, bin_names=None):
        """names: contig names; sequences: str / bytes / uint8 arrays (any case, IUPAC letters);
        bin_of_contig: bin name per contig.  Bin ids are assigned in sorted bin-name order, or follow
        ``bin_names`` when given (a multi-GPU shard may hold no contig of some bins but must number them alike)."""
        names = list(names)
        if not names and bin_names is None:
            raise ValueError("assembly is empty")
        bufs = []
        for s in sequences:
            if isinstance(s, str):
                s = s.encode("ascii")
            bufs.append(np.frombuffer(s, dtype=np.uint8) if not isinstance(s, np.ndarray) else s.astype(np.uint8, copy=False))
        lengths = np.array([len(b) for b in bufs], dtype=np.uint64)
        offsets = np.zeros(len(bufs) + 1, dtype=np.uint64)
        np.cumsum(lengths, out=offsets[1:])
        # no contig at all = the shard of a rank that received nothing (more GPUs than pieces): it keeps the bin
        # numbering, scores everything to zero and joins every collective (nm_upload_contigs accepts n_contigs = 0)
        ascii_all = np.concatenate(bufs) if len(bufs) > 1 else (np.ascontiguousarray(bufs[0]) if bufs else np.zeros(1, np.uint8))
        self.bin_names = sorted(set(bin_of_contig)) if bin_names is None else list(bin_names)
        self.bin_index = {b: i for i, b in enumerate(self.bin_names)}
        bin_ids = np.array([self.bin_index[b] for b in bin_of_contig], dtype=np.uint32) if names else np.zeros(1, np.uint32)
        _lib.check(self.lib.nm_upload_contigs(self.ctx, len(names), _ptr(offsets, C.c_uint64), _ptr(bin_ids, C.c_uint32),
                                              len(self.bin_names), _ptr(ascii_all, C.c_uint8)))
        self.contig_names = names
        self.contig_index = {n: i for i, n in enumerate(names)}
        self.contig_lengths = lengths.astype(np.int64)
        self.contig_bin = bin_ids
        self.slot_of_mod = {}

    def upload_assembly_device(self, names, lengths, bin_of_contig, device_ptr: int, bin_names=None):
        """``upload_assembly`` for sequences that are already in device memory: ``device_ptr`` addresses the contigs'
        ASCII bytes back to back in the order of ``names`` (nm_upload_contigs_device), e.g. a torch uint8 tensor's
        ``data_ptr()``."""
        names = list(names)
        lengths = np.asarray(lengths, dtype=np.uint64)
        offsets = np.zeros(len(names) + 1, dtype=np.uint64)
        np.cumsum(lengths, out=offsets[1:])
        self.bin_names = sorted(set(bin_of_contig)) if bin_names is None else list(bin_names)
        self.bin_index = {b: i for i, b in enumerate(self.bin_names)}
        bin_ids = np.array([self.bin_index[b] for b in bin_of_contig], dtype=np.uint32)
        _lib.check(self.lib.nm_upload_contigs_device(self.ctx, len(names), _ptr(offsets, C.c_uint64), _ptr(bin_ids, C.c_uint32),
                                                     len(self.bin_names), C.c_void_p(int(device_ptr))))
        self.contig_names = names
        self.contig_index = {n: i for i, n in enumerate(names)}
        self.contig_lengths = lengths.astype(np.int64)
        self.contig_bin = bin_ids
        self.slot_of_mod = {}

    def upload_assembly_fasta(self, assembly, names, bin_of_contig, bin_names=None):
        """``upload_assembly`` for a ``fasta.DeviceAssembly`` (the FASTA was parsed on this device): contig ``names[i]`` is the
        record of that name; the planes are packed straight from the parser's device buffer (nm_upload_contigs_fasta)."""
        names = list(names)
        records = np.array([assembly.record[n] for n in names], dtype=np.uint32)
        self.bin_names = sorted(set(bin_of_contig)) if bin_names is None else list(bin_names)
        self.bin_index = {b: i for i, b in enumerate(self.bin_names)}
        bin_ids = np.array([self.bin_index[b] for b in bin_of_contig], dtype=np.uint32)
        _lib.check(self.lib.nm_upload_contigs_fasta(self.ctx, assembly._h, len(names), _ptr(records, C.c_uint32), _ptr(bin_ids, C.c_uint32),
                                                    len(self.bin_names)))
        self.contig_names = names
        self.contig_index = {n: i for i, n in enumerate(names)}
        self.contig_lengths = np.array([assembly.length(n) for n in names], dtype=np.int64)
        self.contig_bin = bin_ids
        self.slot_of_mod = {}

    def contig_base_counts(self, base: str, padding: int) -> np.ndarray:
        """Per resident contig: positions p in [padding, len - padding) whose base is ``base`` — the number of valid
        starts ``sample_n_subsequences`` draws from (seq.py:202-225)."""
        out = np.zeros(len(self.contig_names), dtype=np.uint64)
        _lib.check(self.lib.nm_contig_base_counts(self.ctx, ord(base), int(padding), _ptr(out, C.c_uint64)))
        return out

    def other_letters(self) -> int:
        """Assembly letters that are none of A C G T N (the reference's window code raises KeyError on them)."""
        n = C.c_uint64(0)
        _lib.check(self.lib.nm_assembly_other_letters(self.ctx, C.byref(n)))
        return int(n.value)

    # ------------------------------------------------------------------ pileup
    def upload_pileup(self, mod_type, contig_id, position, strand, fraction_mod, low=0.3, high=0.7, append=False,
                      label=None):
        """Rows of one mod type after the pre-filters (SoA).  strand: uint8 ASCII '+'/'-'.  ``label`` names the
        resident classification (default: the mod type); a second label of the same mod type holds another
        threshold pair (find_motifs_bin.state_label)."""
        label = mod_type if label is None else label
        if label not in self.slot_of_mod:
            n_slots = len(set(self.slot_of_mod.values()))
            if n_slots >= 8:
                raise ValueError("at most 8 pileup classifications resident")
            self.slot_of_mod[label] = n_slots
        slot = self.slot_of_mod[label]
        cid = np.ascontiguousarray(contig_id, dtype=np.uint32)
        pos = np.ascontiguousarray(position, dtype=np.uint32)
        st = np.ascontiguousarray(strand, dtype=np.uint8)
        fr = np.ascontiguousarray(fraction_mod, dtype=np.float64)
        if not (len(cid) == len(pos) == len(st) == len(fr)):
            raise ValueError("pileup columns differ in length")
        _lib.check(self.lib.nm_upload_pileup(self.ctx, slot, ord(MOD_TYPE_TO_CANONICAL[mod_type]), float(low), float(high),
                                             len(cid), _ptr(cid, C.c_uint32), _ptr(pos, C.c_uint32), _ptr(st, C.c_uint8),
                                             _ptr(fr, C.c_double), 1 if append else 0))

    def ingest_pileup(self, contig_local, position, mod_code, strand, fraction_mod, nvalid_cov, labels, low=0.3, high=0.7,
                      want_rows=True, max_part_rows=None, extra_parts=()):
        """RAW pileup rows -> device-side pre-filters (dataload.py:191-247) -> state planes.
        contig_local: engine contig index per row, 0xFFFFFFFF for contigs this engine does not hold;
        mod_code: int8 ids as numbered by the reader (0 = m, 1 = a, 2 = 21839, 3.. = others, which only take part in
        the filters); labels: {mod code id: (label, canonical base)} for the codes to classify.
        Returns dict(n_kept, n_confident, confident=(contig_local, position, strand, mod_code) of the surviving rows
        with fraction_mod >= high (None unless ``want_rows``: the device keeps them as the methylated-state planes, see
        ``methylated_row_counts`` / ``DeviceWindowStore.add_task_contigs``; ``confident_rows()`` fetches them later),
        kept=uint32[n_contigs, 8] surviving rows per (contig, mod code)).
        ``max_part_rows``: ingest a pileup whose rows are grouped by contig (modkit output is) in parts of about that many
        rows, cut at contig boundaries (nm_ingest_pileup_part) — bounds the device memory of the raw rows and of the
        dense adjacency arrays; the result is the same.  ``extra_parts``: further parts as dicts of the six columns
        (contig, position, mod_type, strand, fraction_mod, nvalid_cov), each holding whole contigs that appear in no other
        part (the rows of a contig that is a member of a second bin, under that placement's contig index)."""
        cid = np.ascontiguousarray(contig_local, dtype=np.uint32)
        pos = np.ascontiguousarray(position, dtype=np.uint32)
        mod = np.ascontiguousarray(mod_code, dtype=np.int8)
        st = np.ascontiguousarray(strand, dtype=np.uint8)
        fr = np.ascontiguousarray(fraction_mod, dtype=np.float64)
        nv = nvalid_cov if getattr(nvalid_cov, "dtype", None) == np.int32 else np.clip(nvalid_cov, -2**31, 2**31 - 1)
        nv = np.ascontiguousarray(nv, dtype=np.int32)
        n = len(cid)
        if not (len(pos) == len(mod) == len(st) == len(fr) == len(nv) == n):
            raise ValueError("pileup columns differ in length")
        slot_of = (C.c_int32 * 8)(*([-1] * 8))
        canon = (C.c_uint8 * 8)(*([0] * 8))
        for code, (label, base) in labels.items():
            if label not in self.slot_of_mod:
                n_slots = len(set(self.slot_of_mod.values()))
                if n_slots >= 8:
                    raise ValueError("at most 8 pileup classifications resident")
                self.slot_of_mod[label] = n_slots
            slot_of[int(code)] = self.slot_of_mod[label]
            canon[int(code)] = ord(base)
        n_kept, n_conf = C.c_uint64(0), C.c_uint64(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        # always ingest in PARTS of whole contigs listing the contigs that have rows (nm_ingest_pileup_part): the dense
        # adjacency scratch (16 B per bp) then covers those contigs only, not the whole resident assembly — a multi-Gbp
        # assembly with a modest pileup would otherwise ask for tens of GB it does not need
        parts = self._pileup_parts(cid, max_part_rows if (max_part_rows and n > max_part_rows) else max(n, 1)) if n else []
        if parts is None:                 # some contig's rows are not contiguous: one part with every contig present
            parts = [(0, n)]
        if n == 0:
            _lib.check(self.lib.nm_ingest_pileup(self.ctx, 0, None, None, None, None, None, None, slot_of, canon,
                                                 float(low), float(high), 0, C.byref(n_kept), C.byref(n_conf)))
        for k, (a, b) in enumerate(parts):
            seg = cid[a:b]
            edge = np.concatenate([[0], np.flatnonzero(seg[1:] != seg[:-1]) + 1]) if b > a else np.zeros(0, np.int64)
            ids = np.unique(seg[edge])
            ids = np.ascontiguousarray(ids[ids != 0xFFFFFFFF], dtype=np.uint32)
            sl = lambda x: vp(x[a:b]) if b > a else None
            _lib.check(self.lib.nm_ingest_pileup_part(self.ctx, b - a, sl(cid), sl(pos), sl(mod), sl(st), sl(fr), sl(nv), slot_of, canon,
                                                      float(low), float(high), 0, 1 if k == 0 else 0, len(ids), _ptr(ids, C.c_uint32),
                                                      C.byref(n_kept), C.byref(n_conf)))
        for x in extra_parts:
            xc = np.ascontiguousarray(x["contig"], dtype=np.uint32)
            cols_x = [xc, np.ascontiguousarray(x["position"], dtype=np.uint32), np.ascontiguousarray(x["mod_type"], dtype=np.int8),
                      np.ascontiguousarray(x["strand"], dtype=np.uint8), np.ascontiguousarray(x["fraction_mod"], dtype=np.float64),
                      np.ascontiguousarray(x["nvalid_cov"], dtype=np.int32)]
            ids = np.ascontiguousarray(np.unique(xc[xc != 0xFFFFFFFF]), dtype=np.uint32)
            if len(xc) == 0:
                continue
            _lib.check(self.lib.nm_ingest_pileup_part(self.ctx, len(xc), *[vp(a) for a in cols_x], slot_of, canon, float(low), float(high), 0,
                                                      0, len(ids), _ptr(ids, C.c_uint32), C.byref(n_kept), C.byref(n_conf)))
        self._n_confident = int(n_conf.value)
        kept = np.zeros((len(self.contig_names), 8), dtype=np.uint32)
        _lib.check(self.lib.nm_ingest_results(self.ctx, None, None, None, None, 0, _ptr(kept, C.c_uint32)))
        return dict(n_kept=int(n_kept.value), n_confident=self._n_confident,
                    confident=self.confident_rows() if want_rows else None, kept=kept)

    def ingest_device_pileup(self, table, lut, labels, low=0.3, high=0.7, max_part_rows=None):
        """``ingest_pileup`` for a ``pileup.DevicePileup`` (columns already in device memory, parsed there): ``lut`` maps
        the file's contig ids to engine contig indices (0xFFFFFFFF: not held here).  Parts are cut at the runs of equal
        contig names the parser reports.  Returns dict(n_kept, n_confident, kept)."""
        lut = np.ascontiguousarray(lut, dtype=np.uint32)
        table.map_contigs(lut)
        ptr = table.device_pointers()
        esz = {"contig": 4, "position": 4, "mod_type": 1, "strand": 1, "fraction_mod": 8, "nvalid_cov": 4}
        slot_of = (C.c_int32 * 8)(*([-1] * 8))
        canon = (C.c_uint8 * 8)(*([0] * 8))
        for code, (label, base) in labels.items():
            if label not in self.slot_of_mod:
                n_slots = len(set(self.slot_of_mod.values()))
                if n_slots >= 8:
                    raise ValueError("at most 8 pileup classifications resident")
                self.slot_of_mod[label] = n_slots
            slot_of[int(code)] = self.slot_of_mod[label]
            canon[int(code)] = ord(base)
        run_eng = lut[table.run_contig]                                   # engine contig of every run
        n = len(table)
        bounds = [0]
        if max_part_rows and n > max_part_rows:
            # a part must hold whole contigs: a run boundary is a cut when every contig seen so far has had its LAST run
            # (a modkit file has one run per contig: every boundary is one; a file that comes back to a contig later is cut
            # behind that contig's last run, so its parts can be larger than asked for — said below, never silently whole)
            last_run = {}
            for r, c in enumerate(run_eng.tolist()):
                if c != 0xFFFFFFFF:
                    last_run[c] = r
            at, open_until = 0, -1
            for r, (c, e) in enumerate(zip(run_eng.tolist(), table.run_row[1:].tolist())):
                if c != 0xFFFFFFFF:
                    open_until = max(open_until, last_run[c])
                if open_until <= r and e - at >= max_part_rows:
                    bounds.append(int(e))
                    at = e
        if bounds[-1] != n:
            bounds.append(n)
        part_rows = [b - a for a, b in zip(bounds, bounds[1:])]          # empty for a table without rows
        if max_part_rows and part_rows and max(part_rows) > 2 * max_part_rows:
            import logging
            logging.warning(f"pileup rows are not grouped by contig: ingestion parts of up to {max(part_rows):,} rows "
                            f"instead of {max_part_rows:,} (device memory for the raw rows and the adjacency scratch grows with them)")
        n_kept, n_conf = C.c_uint64(0), C.c_uint64(0)
        for k in range(len(bounds) - 1):
            a, b = bounds[k], bounds[k + 1]
            in_part = (table.run_row[:-1] >= a) & (table.run_row[:-1] < b)
            ids = np.unique(run_eng[in_part])
            ids = np.ascontiguousarray(ids[ids != 0xFFFFFFFF], dtype=np.uint32)
            col = lambda name: C.c_void_p(ptr[name] + a * esz[name])
            _lib.check(self.lib.nm_ingest_pileup_part(self.ctx, b - a, col("contig"), col("position"), col("mod_type"), col("strand"),
                                                      col("fraction_mod"), col("nvalid_cov"), slot_of, canon, float(low), float(high), 1,
                                                      1 if k == 0 else 0, len(ids), _ptr(ids, C.c_uint32), C.byref(n_kept), C.byref(n_conf)))
        self._n_confident = int(n_conf.value)
        kept = np.zeros((len(self.contig_names), 8), dtype=np.uint32)
        _lib.check(self.lib.nm_ingest_results(self.ctx, None, None, None, None, 0, _ptr(kept, C.c_uint32)))
        return dict(n_kept=int(n_kept.value), n_confident=self._n_confident, confident=None, kept=kept)

    @staticmethod
    def _pileup_parts(cid: np.ndarray, max_rows: int):
        """[(begin, end)] row ranges of about ``max_rows`` rows cut where the contig id changes, or None when some contig's
        rows are not contiguous (then the pileup has to be ingested in one piece)."""
        change = np.flatnonzero(cid[1:] != cid[:-1]) + 1
        starts = np.concatenate([[0], change])
        ids = cid[starts]
        real = ids[ids != 0xFFFFFFFF]
        if len(np.unique(real)) != len(real):
            return None
        ends = np.concatenate([change, [len(cid)]])
        parts, a = [], 0
        for e in ends.tolist():
            if e - a >= max_rows:
                parts.append((a, e))
                a = e
        if a < len(cid):
            parts.append((a, len(cid)))
        return parts

    def confident_rows(self):
        """(contig_local, position, strand, mod_code) of the rows the last ``ingest_pileup`` kept with
        fraction_mod >= high, copied from the device."""
        k = getattr(self, "_n_confident", 0)
        cc, cp = np.empty(k, np.uint32), np.empty(k, np.uint32)
        cs, cm = np.empty(k, np.uint8), np.empty(k, np.int8)
        if k:
            _lib.check(self.lib.nm_ingest_results(self.ctx, _ptr(cc, C.c_uint32), _ptr(cp, C.c_uint32), _ptr(cs, C.c_uint8),
                                                  cm.ctypes.data_as(C.POINTER(C.c_int8)), k, None))
        return cc, cp, cs, cm

    def methylated_row_counts(self, label, padding: int) -> np.ndarray:
        """uint64[n_contigs, 2]: confidently methylated rows (plus, minus) of the classification ``label`` per resident
        contig with padding < position < len - padding — the rows window extraction uses."""
        out = np.zeros((len(self.contig_names), 2), dtype=np.uint64)
        _lib.check(self.lib.nm_methylated_row_counts(self.ctx, self.slot_of_mod[label], int(padding), _ptr(out, C.c_uint64)))
        return out

    def alias_label(self, label, existing):
        """Make ``label`` refer to the classification already resident as ``existing``."""
        self.slot_of_mod[label] = self.slot_of_mod[existing]

    # ------------------------------------------------------------------ scoring
    def make_batch(self, candidates, slot_of=None) -> CandidateBatch:
        """candidates: sequence of (Motif, mod_type, bin name or id).  Parsing / stripping is done natively
        (nm_parse_motifs); Python only joins the strings.  ``slot_of``: mod type -> slot number, default the resident
        classifications (``slot_of_mod``)."""
        n = len(candidates)
        strings = [c[0].string for c in candidates]
        text = "".join(strings).encode("ascii")
        offsets = np.zeros(n + 1, dtype=np.uint32)
        np.cumsum(np.fromiter((len(s) for s in strings), dtype=np.uint32, count=n), out=offsets[1:])
        modpos_in = np.fromiter((c[0].mod_position for c in candidates), dtype=np.int32, count=n)
        bi, so = self.bin_index, self.slot_of_mod
        bins = np.fromiter((bi[c[2]] if isinstance(c[2], str) else c[2] for c in candidates), dtype=np.uint32, count=n)
        slots = np.fromiter(((so[c[1]] if slot_of is None else slot_of(c[1])) for c in candidates), dtype=np.uint8, count=n)
        lens = np.empty(n, dtype=np.uint8)
        modpos = np.empty(n, dtype=np.uint8)
        moff = np.empty(n, dtype=np.uint32)
        masks = np.empty(max(len(text), 1), dtype=np.uint8)
        used = C.c_uint64(0)
        _lib.check(self.lib.nm_parse_motifs(n, text, _ptr(offsets, C.c_uint32), _ptr(modpos_in, C.c_int32),
                                            _ptr(lens, C.c_uint8), _ptr(modpos, C.c_uint8), _ptr(moff, C.c_uint32),
                                            _ptr(masks, C.c_uint8), len(masks), C.byref(used)))
        return CandidateBatch(bins, slots, lens, modpos, moff, masks[:used.value])

    def _batch_args(self, b: CandidateBatch):
        args = getattr(b, "_args", None)            # the ctypes views are reusable as long as the arrays live
        if args is None:
            args = (len(b), _ptr(b.bins, C.c_uint32), _ptr(b.slots, C.c_uint8), _ptr(b.lens, C.c_uint8),
                    _ptr(b.modpos, C.c_uint8), _ptr(b.offsets, C.c_uint32), _ptr(b.masks, C.c_uint8))
            b._args = args
        return args

    def score(self, candidates) -> np.ndarray:
        """int64[n, 2] = (n_mod, n_nomod) per candidate — what ``model.update`` receives (find_motifs_bin.py:1320)."""
        if not isinstance(candidates, CandidateBatch) and any(len(c[0].tokens) > MAX_REACH + 1 for c in candidates):
            return self._score_with_wide(candidates)
        b = candidates if isinstance(candidates, CandidateBatch) else self.make_batch(candidates)
        out = np.zeros((len(b), 2), dtype=np.int64)
        if len(b):
            _lib.check(self.lib.nm_score_batch(self.ctx, *self._batch_args(b), out.ctypes.data_as(C.c_void_p)))
        return out

    def _score_with_wide(self, candidates) -> np.ndarray:
        """A search frame above 191 (find_motifs_bin.py:110-130 takes any): candidates that stay within 95 positions of the
        modified base once stripped go the usual way, the others through nm_score_batch_wide (plain kernel, any reach)."""
        stripped = [c[0].stripped_sets() for c in candidates]
        wide = [k for k, (sets, mp) in enumerate(stripped) if max(mp, len(sets) - 1 - mp) > MAX_REACH or len(sets) > MAX_MOTIF_LEN]
        is_wide = set(wide)
        narrow = [k for k in range(len(candidates)) if k not in is_wide]
        out = np.zeros((len(candidates), 2), dtype=np.int64)
        if narrow:
            out[narrow] = self.score(self.make_batch([candidates[k] for k in narrow]))
        if wide:
            n = len(wide)
            bi, so = self.bin_index, self.slot_of_mod
            bins = np.fromiter((bi[candidates[k][2]] if isinstance(candidates[k][2], str) else candidates[k][2] for k in wide), dtype=np.uint32, count=n)
            slots = np.fromiter((so[candidates[k][1]] for k in wide), dtype=np.uint8, count=n)
            lens = np.fromiter((len(stripped[k][0]) for k in wide), dtype=np.uint16, count=n)
            modpos = np.fromiter((stripped[k][1] for k in wide), dtype=np.uint16, count=n)
            moff = np.zeros(n, dtype=np.uint32)
            np.cumsum(lens[:-1], out=moff[1:])
            masks = np.concatenate([stripped[k][0] for k in wide]).astype(np.uint8)
            res = np.zeros((n, 2), dtype=np.int64)
            _lib.check(self.lib.nm_score_batch_wide(self.ctx, n, _ptr(bins, C.c_uint32), _ptr(slots, C.c_uint8), _ptr(lens, C.c_uint16),
                                                    _ptr(modpos, C.c_uint16), _ptr(moff, C.c_uint32), _ptr(masks, C.c_uint8), _ptr(res, C.c_int64)))
            out[wide] = res
            self.wide_scored = getattr(self, "wide_scored", 0) + n
        return out

    def bin_contigs(self, bin_name) -> list:
        """Resident contig names of a bin in the row order of ``score_per_contig`` (nm_bin_contigs)."""
        b = self.bin_index[bin_name] if isinstance(bin_name, str) else int(bin_name)
        n = C.c_uint32(0)
        _lib.check(self.lib.nm_bin_contigs(self.ctx, b, None, 0, C.byref(n)))
        ids = np.zeros(max(n.value, 1), dtype=np.uint32)
        _lib.check(self.lib.nm_bin_contigs(self.ctx, b, _ptr(ids, C.c_uint32), n.value, C.byref(n)))
        return [self.contig_names[i] for i in ids[:n.value].tolist()]

    def score_per_contig(self, candidates):
        """Per-contig (n_mod, n_nomod): ``motif_model_contig`` (find_motifs_bin.py:1285-1331) for every resident contig of
        each candidate's bin in one launch (nm_score_batch_per_contig).  Returns a list, per candidate, of
        (contig names, int64[n_contigs, 2])."""
        b = candidates if isinstance(candidates, CandidateBatch) else self.make_batch(candidates)
        names = {}
        for bid in np.unique(b.bins).tolist():
            names[bid] = self.bin_contigs(int(bid))
        rows = np.zeros(len(b) + 1, dtype=np.uint64)
        np.cumsum([len(names[int(x)]) for x in b.bins], out=rows[1:])
        out = np.zeros((max(int(rows[-1]), 1), 2), dtype=np.int64)
        if len(b):
            _lib.check(self.lib.nm_score_batch_per_contig(self.ctx, *self._batch_args(b), _ptr(rows, C.c_uint64), _ptr(out, C.c_int64)))
        return [(names[int(b.bins[k])], out[int(rows[k]):int(rows[k + 1])]) for k in range(len(b))]

    def set_score_lanes(self, lanes: int):
        """2: consecutive ``score_into_device`` calls alternate between two streams, so that independent batches
        overlap on the device (include/nmscan.h: nm_set_score_lanes); 1: strict order (default)."""
        _lib.check(self.lib.nm_set_score_lanes(self.ctx, int(lanes)))

    def sync(self):
        """All scoring work queued on this engine has finished (both lanes)."""
        _lib.check(self.lib.nm_sync(self.ctx))

    def score_into_device(self, batch: CandidateBatch, device_ptr: int):
        """Asynchronous variant: counts land in device memory (e.g. a torch int64 tensor) on the engine stream."""
        _lib.check(self.lib.nm_score_batch_device(self.ctx, *self._batch_args(batch), C.c_void_p(device_ptr)))

    def hit_positions(self, contig, mod_type, motif: Motif, which: int) -> np.ndarray:
        """Ascending contig-local positions; which = 0 meth fwd, 1 nonmeth fwd, 2 meth rev, 3 nonmeth rev
        (motif_model_contig(save_motif_positions=True), find_motifs_bin.py:1322-1329)."""
        cid = self.contig_index[contig] if isinstance(contig, str) else int(contig)
        b = self.make_batch([(motif, mod_type, 0)])
        n = C.c_uint64(0)
        cap = 1 << 16
        while True:
            out = np.empty(cap, dtype=np.int64)
            _lib.check(self.lib.nm_hit_positions(self.ctx, cid, self.slot_of_mod[mod_type], int(b.lens[0]), int(b.modpos[0]),
                                                 _ptr(b.masks, C.c_uint8), which, _ptr(out, C.c_int64), cap, C.byref(n)))
            if n.value <= cap:
                return out[:n.value].copy()
            cap = int(n.value)

    # ------------------------------------------------------------------ multi-GPU exchange (nm_comm_*, RCCL)
    def comm_unique_id(self) -> bytes:
        """The 128-byte id rank 0 creates; the host carries it to the other ranks (nm_comm_unique_id)."""
        buf = (C.c_uint8 * 128)()
        _lib.check(self.lib.nm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """Collective: join the RCCL communicator of the run (nm_comm_init)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self.lib.nm_comm_init(self.ctx, int(rank), int(world), buf))
        self.comm_world = int(world)

    def allreduce_counts_device(self, device_ptr: int, n: int, slot: int = 0):
        """Start the in-place sum of a device int64[n] table over all ranks on the communication stream
        (nm_allreduce_counts_async); ``comm_wait(slot)`` before the engine stream touches that buffer again."""
        _lib.check(self.lib.nm_allreduce_counts_async(self.ctx, C.c_void_p(device_ptr), int(n), int(slot)))

    def comm_wait(self, slot: int = 0):
        _lib.check(self.lib.nm_comm_wait(self.ctx, int(slot)))

    def comm_sync(self):
        _lib.check(self.lib.nm_comm_sync(self.ctx))

    def comm_info(self) -> dict:
        """What the RCCL communicator reports about itself (nm_comm_info): world, rank, device, RCCL version code;
        world 0 = no communicator on this engine."""
        info = (C.c_int32 * 4)()
        _lib.check(self.lib.nm_comm_info(self.ctx, info))
        return {"world": int(info[0]), "rank": int(info[1]), "device": int(info[2]), "rccl_version": int(info[3])}

    def allreduce_host(self, counts: np.ndarray) -> np.ndarray:
        """Sum an integer numpy array over all ranks (nm_allreduce_counts_host); returns the same dtype and shape."""
        a = np.ascontiguousarray(counts, dtype=np.int64).copy()
        if a.size:
            _lib.check(self.lib.nm_allreduce_counts_host(self.ctx, _ptr(a, C.c_int64), a.size))
        return a.astype(counts.dtype, copy=False).reshape(counts.shape)

    # ------------------------------------------------------------------ measurement
    def stats(self) -> dict:
        w = (C.c_uint64 * 8)()
        _lib.check(self.lib.nm_stats(self.ctx, w))
        keys = ["total_bp", "padded_bp", "seq_plane_bytes", "state_plane_bytes", "launches", "last_workgroups",
                "last_compact", "last_general"]
        return dict(zip(keys, [int(x) for x in w]))

    def timing_reset(self, enable=True):
        """True / 1: collect the scoring launches; 2: every device phase of the library (pre-filters, window gathers, window
        batches, background counts); False / 0: stop."""
        _lib.check(self.lib.nm_timing_reset(self.ctx, int(enable)))

    def timing_total(self):
        """(sum of scoring-kernel durations in ms, number of launches) since ``timing_reset(True)``."""
        ms, n = C.c_double(0), C.c_uint64(0)
        _lib.check(self.lib.nm_timing_total_ms(self.ctx, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def timing_intervals(self, epoch=None) -> np.ndarray:
        """float64[n, 2]: begin / end of every phase since ``timing_reset`` in ms after the first phase of ``epoch`` (another engine on
        this device; default: this one) — nm_timing_intervals."""
        ref = self if epoch is None else epoch
        n = C.c_uint64(0)
        _lib.check(self.lib.nm_timing_intervals(self.ctx, ref.ctx, 0, None, None, C.byref(n)))
        out = np.zeros((2, max(int(n.value), 1)), dtype=np.float64)
        if n.value:
            _lib.check(self.lib.nm_timing_intervals(self.ctx, ref.ctx, int(n.value), _ptr(out[0], C.c_double), _ptr(out[1], C.c_double), C.byref(n)))
        return np.ascontiguousarray(out[:, :int(n.value)].T)

    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        _lib.check(self.lib.nm_last_kernel_ms(self.ctx, C.byref(ms)))
        return float(ms.value)

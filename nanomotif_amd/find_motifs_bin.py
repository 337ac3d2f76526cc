"""Motif discovery driver of the MI355X build — the host-side mirror of nanomotif/find_motifs_bin.py.

Same public names and argument meaning as the reference for the pieces on the hot path
(``ProcessorConfig``, ``motif_model_bin``, ``get_parent_scores``, ``find_best_candidates``,
``process_subpileup``, ``predictive_evaluation_score``), but the structure is GPU-first:

    reference: Pool(T) workers, one (bin, mod type) task each; every candidate = one regex scan of the bin
    here:      ONE process per GPU holds every bin in HBM; all (bin, mod type) searches advance in lock-step
               and each round of open candidates is one nm_score_batch launch (+ one all-reduce with N GPUs)
"""
from __future__ import annotations

import logging as log
import os
import random
from dataclasses import dataclass

import numpy as np

from . import pileup as pileup_mod
from . import postprocess
from .model import BetaBernoulliModel, predictive_evaluation_score  # noqa: F401  (re-exported like the reference)
from .motif import MOD_TYPE_TO_CANONICAL, Motif
from .pileup import MOD_TYPES, PileupTable
from .search import HostWindowStore, extract_windows, find_best_candidates_co, get_parent_scores_co, run_lockstep

IUPAC_LETTERS = set("ATGCRYSWKMBDHVN")
MAX_WINDOW_WIDTH = 4095        # include/nmscan.h NM_MAX_WIDE_MOTIF_LEN: a candidate is at most one window long


# ------------------------------------------------------------------------------------------------ config
@dataclass
class ProcessorConfig:
    """find_motifs_bin.py:59-130 — same fields, same validation and messages."""
    assembly: dict
    pileup_path: str
    bin_contig: dict            # contig -> bin
    threads: int
    search_frame_size: int
    methylation_threshold_low: float
    methylation_threshold_high: float
    minimum_kl_divergence: float
    score_threshold: float
    log_dir: str
    seed: int
    output_dir: str
    verbose: bool = False

    def __post_init__(self):
        if self.assembly is None:
            raise ValueError("assembly cannot be None")
        if self.pileup_path is None:
            raise ValueError("pileup_path cannot be None")
        if not self.bin_contig:
            raise ValueError("bin_contig cannot be None or empty")
        missing = sorted(set(self.bin_contig) - set(self.assembly))
        if missing:
            log.warning("Removing %d contig-bin assignments because the contigs are absent from the assembly: %s%s",
                        len(missing), ", ".join(missing[:5]), "..." if len(missing) > 5 else "")
            self.bin_contig = {c: b for c, b in self.bin_contig.items() if c in self.assembly}
        if not self.bin_contig:
            raise ValueError("No contigs remain in bin_contig after filtering against the assembly")
        if self.threads <= 0:
            raise ValueError("threads must be greater than 0")
        if self.search_frame_size <= 1:
            raise ValueError("search_frame_size must be greater than 1")
        if 2 * (self.search_frame_size // 2) + 1 > MAX_WINDOW_WIDTH:
            # the reference takes any frame (find_motifs_bin.py:128-130; default 40).  Up to 191 columns everything runs on the
            # device; above, windows stay on the host and far-reaching candidates go through nm_score_batch_wide (main.py)
            raise ValueError(f"search_frame_size must be at most {MAX_WINDOW_WIDTH} on the MI355X engine "
                             f"(windows of 2 * (search_frame_size // 2) + 1 <= {MAX_WINDOW_WIDTH} positions)")
        if not (0 <= self.methylation_threshold_high <= 1):
            raise ValueError("methylation_threshold_high must be in [0,1]")
        if not (0 <= self.methylation_threshold_low <= 1):
            raise ValueError("methylation_threshold_low must be in [0,1]")
        if self.methylation_threshold_high <= self.methylation_threshold_low:
            raise ValueError("methylation_threshold_high must be greater than methylation_threshold_low")
        if self.minimum_kl_divergence <= 0:
            raise ValueError("minimum_kl_divergence must be > 0")
        if not (0 <= self.score_threshold):
            raise ValueError("score_threshold must be > 0")

    @property
    def padding(self):
        return self.search_frame_size // 2


# ------------------------------------------------------------------------------------------------ scoring
class BinData:
    """What the reference passes around as (bin pileup frame, dict of contig sequences): here a handle to the
    engine-resident bin.  ``motif_model_bin(pileup=<BinData>, contigs=<BinData>, ...)`` keeps the call shape."""

    def __init__(self, scorer, bin_name: str, mod_type: str):
        self.scorer, self.bin_name, self.mod_type = scorer, bin_name, mod_type


def motif_model_bin(pileup: BinData, contigs: BinData, motif: Motif, model: BetaBernoulliModel,
                    low_meth_threshold, high_meth_threshold) -> BetaBernoulliModel:
    """find_motifs_bin.py:1265-1283 — counts methylated / unmethylated motif sites over every contig of the bin on
    both strands and ``update``s ``model`` IN PLACE (also returned), like the reference.  The thresholds must be
    the ones the pileup was classified with at upload."""
    s = pileup.scorer
    if (low_meth_threshold, high_meth_threshold) != (s.low, s.high):
        raise ValueError(f"thresholds {low_meth_threshold}/{high_meth_threshold} differ from the uploaded pileup's {s.low}/{s.high}")
    assert type(motif) is Motif, "Motif is not a Motif type"
    n_mod, n_nomod = s([((pileup.bin_name, pileup.mod_type), motif, None)])[0]
    model.update(int(n_mod), int(n_nomod))
    return model


def get_parent_scores(motif: Motif, pileup: BinData, contigs: BinData, low_meth_threshold, high_meth_threshold):
    """find_motifs_bin.py:1382-1433 (one batched launch for the motif and all its parents)."""
    key = (pileup.bin_name, pileup.mod_type)
    return run_lockstep({key: get_parent_scores_co(motif)}, pileup.scorer)[key]


class LockstepScorer:
    """score_fn of ``run_lockstep``: ``local_counts(flat) -> int64[n, 2]`` scores one round's requests against this
    rank's contigs (the HIP engine in production); with ``use_dist`` the per-rank tables are summed with one
    all-reduce per round (RCCL on GPUs, gloo in the CPU tests)."""

    def __init__(self, local_counts, low=0.3, high=0.7, use_dist=False, group=None):
        self.local_counts, self.low, self.high = local_counts, low, high
        self.use_dist, self.group = use_dist, group
        self.rounds = 0
        self.candidates = 0
        # native search only: lock-step iterations, children answered by the device's speculation / asked for after all
        self.search_iterations = self.speculation_hits = self.speculation_misses = 0

    def __call__(self, flat):
        self.rounds += 1
        self.candidates += len(flat)
        counts = self.local_counts(flat) if flat else np.zeros((0, 2), dtype=np.int64)
        if self.use_dist:
            counts = allreduce_counts(counts, self.group)
        return counts


def state_label(mod_type, tag):
    """Name of the engine-resident pileup classification a request runs on: the mod type itself for the search's
    thresholds, ``(mod_type, "merge")`` for the merge stage's fixed 0.3 / 0.7."""
    return mod_type if tag is None else (mod_type, tag)


def engine_scorer(engine, low=0.3, high=0.7, use_dist=False, group=None) -> LockstepScorer:
    return LockstepScorer(lambda flat: engine.score([(m, state_label(key[1], tag), key[0]) for key, m, tag in flat]),
                          low, high, use_dist, group)


_native_reducer = None


def use_native_allreduce(engine):
    """Route ``allreduce_counts`` through the C ABI's own RCCL communicator (``engine.comm_init`` done on every rank;
    include/nmscan.h: nm_allreduce_counts_host) instead of torch.distributed; ``None`` switches back."""
    global _native_reducer
    _native_reducer = engine


def allreduce_counts(counts: np.ndarray, group=None) -> np.ndarray:
    """Sum int64 count tables over ranks: the library's RCCL communicator when one is up
    (``use_native_allreduce``), else torch.distributed — nccl(=RCCL) through the GPU, gloo on CPU."""
    if _native_reducer is not None and group is None:
        return _native_reducer.allreduce_host(np.asarray(counts))
    from . import _lib
    if _lib._lib is not None and not _lib.loaded_with_torch:
        # the library came up on the system HIP runtime (torch-free CLI start); importing torch now would put a second
        # runtime into the process
        raise RuntimeError("allreduce_counts needs libnmscan loaded after torch (multi-rank runs import torch first: "
                           "set NANOMOTIF_WITH_TORCH=1 or start through torch.distributed.run)")
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(counts))
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, group=group)
    return t.cpu().numpy()


# ------------------------------------------------------------------------------------------------ per-task pipeline
def task_coroutine(bin_name, mod_type, bin_pssm, cfg: ProcessorConfig, stage_writer=None, temp_dir=None, files=None):
    """process_subpileup (find_motifs_bin.py:468-596) as one scoring coroutine: search, then post-processing."""
    res = yield from find_best_candidates_co(
        bin_pssm, mod_type, cfg.padding, min_kl=cfg.minimum_kl_divergence, max_dead_ends=25,
        max_rounds_since_new_best=30, score_threshold=cfg.score_threshold,
        log=lambda msg: log.info(f"[{bin_name} {mod_type}] {msg}"))
    return (yield from post_coroutine(bin_name, mod_type, res, cfg, stage_writer, temp_dir, files))


def _write_text(path, text):
    with open(path, "w") as f:
        f.write(text)


class DeferredFiles:
    """The small files a run with ``--out`` leaves per task (five precleanup tables, the background PSSM, the search graph: 7 000 files
    for a thousand tasks): their text is made where the reference writes them, the files are created together at the end.  (By ONE
    thread: a pool of eight was measured at 0.2–0.36 s against 0.07 — tiny files in one directory tree contend in the kernel, and the
    interpreter lock changes hands per file.)"""

    def __init__(self):
        self.items = []

    def add(self, path, text):
        self.items.append((path, text))

    def flush(self):
        items, self.items = self.items, []
        for path, text in items:
            _write_text(path, text)


def write_search_artifacts(bin_name, mod_type, res, temp_dir=None, files=None):
    """What process_subpileup leaves next to the tables (find_motifs_bin.py:521-535): the "no motifs" log line, the
    background PSSM and the search graph under ``temp_dir`` (``files``: a DeferredFiles that writes them later).  Returns False when
    the search found nothing."""
    if res is None:
        log.info(f"[{bin_name} {mod_type}] No motifs found")
        return False
    graph, best, bin_pssm = res
    if temp_dir:
        os.makedirs(temp_dir, exist_ok=True)
        # (np.savetxt(path, bin_pssm, fmt="%.4f") byte for byte — one format call per row —, a fifth of its cost: a thousand tasks write one each)
        rows = np.asarray(bin_pssm, dtype=np.float64).tolist()
        row_fmt = " ".join(["%.4f"] * len(rows[0])) + "\n" if rows else ""
        pssm_text = "".join(row_fmt % tuple(row) for row in rows)
        sink = files.add if files is not None else _write_text
        sink(os.path.join(temp_dir, "background_pssm.txt"), pssm_text)
        sink(os.path.join(temp_dir, f"motif_graph_{mod_type}.gml"), graph.gml_text())
    return True


def post_coroutine(bin_name, mod_type, res, cfg: ProcessorConfig, stage_writer=None, temp_dir=None, files=None):
    """The part of process_subpileup after the search (find_motifs_bin.py:537-596): ``res`` = what
    find_best_candidates returned — (graph, best candidates, background PSSM) or None."""
    if not write_search_artifacts(bin_name, mod_type, res, temp_dir, files):
        return None
    graph, best, bin_pssm = res
    rows = yield from postprocess.postprocess_co(graph, best, bin_name, mod_type, cfg.padding, on_stage=stage_writer)
    return rows


class FilteredPileup:
    """What window extraction needs from the filtered pileup (find_motifs_bin.py:625-661): the surviving rows with
    fraction_mod >= high, and which (contig, mod type) pairs have any surviving row at all.  Produced on the device
    by ``nm_ingest_pileup`` (``from_ingest``) or from a host-side ``PileupTable`` (``from_table``)."""

    def __init__(self, contig_names, conf_contig, conf_position, conf_strand, conf_mod, kept):
        self.contig_names = list(contig_names)          # index space of conf_contig / kept rows
        self.conf_contig, self.conf_position = np.asarray(conf_contig), np.asarray(conf_position, dtype=np.int64)
        self.conf_strand, self.conf_mod = np.asarray(conf_strand), np.asarray(conf_mod)
        self.kept = np.asarray(kept)                    # [n_contigs, n_mod_codes] surviving rows

    @classmethod
    def from_table(cls, table: PileupTable, high):
        kept = np.zeros((len(table.contig_names), 8), dtype=np.int64)
        ok = table.mod_type >= 0
        np.add.at(kept, (table.contig[ok], table.mod_type[ok]), 1)
        conf = (table.fraction_mod >= high) & ok
        return cls(table.contig_names, table.contig[conf], table.position[conf], table.strand[conf], table.mod_type[conf], kept)

    @classmethod
    def merge(cls, parts):
        """Union of per-rank results over disjoint contig sets, re-indexed by contig name (sorted)."""
        names = sorted({n for p in parts for n in p.contig_names})
        idx = {n: i for i, n in enumerate(names)}
        kept = np.zeros((len(names), 8), dtype=np.int64)
        cc, cp, cs, cm = [], [], [], []
        for p in parts:
            lut = np.array([idx[n] for n in p.contig_names], dtype=np.int64)
            if len(lut):
                kept[lut] += p.kept.astype(np.int64)
            cc.append(lut[p.conf_contig.astype(np.int64)] if len(p.conf_contig) else np.zeros(0, np.int64))
            cp.append(p.conf_position); cs.append(p.conf_strand); cm.append(p.conf_mod)
        cat = lambda xs, dt: np.concatenate(xs) if xs else np.zeros(0, dt)
        return cls(names, cat(cc, np.int64), cat(cp, np.int64), cat(cs, np.uint8), cat(cm, np.int8), kept)

    def _index(self):
        """Confident rows sorted by (mod code, contig, strand, position) + run boundaries, built once."""
        if getattr(self, "_order", None) is None:
            minus = (self.conf_strand == ord("-")).astype(np.int64)
            key = (self.conf_mod.astype(np.int64) * len(self.contig_names) + self.conf_contig.astype(np.int64)) * 2 + minus
            if len(key) and key.max() < 2 ** 31 and self.conf_position.max(initial=0) < 2 ** 32:
                order = np.argsort((key.astype(np.uint64) << np.uint64(32)) | self.conf_position.astype(np.uint64), kind="stable")
            else:
                order = np.lexsort((self.conf_position, key))
            self._order, self._key = order, key[order]
            self._name_index = {n: i for i, n in enumerate(self.contig_names)}
        return self._order, self._key

    def present(self, contig_names, mod_id):
        """The listed contigs that have at least one surviving row of this mod type (find_motifs_bin.py:629)."""
        if getattr(self, "_name_index", None) is None:
            self._name_index = {n: i for i, n in enumerate(self.contig_names)}
            self._kept_rows = (np.asarray(self.kept) != 0).tolist()        # plain lists: this runs once per (bin, mod type)
        index, kept = self._name_index, self._kept_rows
        out = []
        for name in contig_names:
            i = index.get(name)
            if i is not None and kept[i][mod_id]:
                out.append(name)
        return out

    def present_sorted_ids(self, bins: dict, id_of: dict, n_mods: int):
        """``present`` for every (bin, mod type) at once, for the native plan: ``bins``: bin -> contig names, ``id_of``: contig name ->
        engine contig index.  Returns {(bin, mod id): uint32 array of the engine indices of the bin's contigs that have a surviving
        row of that mod type, in the order of their sorted NAMES (what ``DeviceWindowExtractor.plan_all`` plans a task in)}; pairs
        without such a contig are absent.  A handful of numpy passes over all contigs instead of a Python loop per contig and task."""
        index = getattr(self, "_name_index", None) or {n: i for i, n in enumerate(self.contig_names)}
        order, flat, begin = list(bins), [], [0]
        for b in order:
            flat += sorted(bins[b])
            begin.append(len(flat))
        n = len(flat)
        fidx = np.fromiter((index.get(x, -1) for x in flat), dtype=np.int64, count=n)
        eng = np.fromiter((id_of[x] for x in flat), dtype=np.uint32, count=n)
        kept = np.asarray(self.kept) != 0
        known = fidx >= 0
        out = {}
        starts = np.asarray(begin[:-1], dtype=np.int64)
        nonempty = np.flatnonzero(np.diff(begin) > 0)
        for m in range(n_mods):
            mask = known & kept[np.where(known, fidx, 0), m] if n else np.zeros(0, bool)
            if not mask.any():
                continue
            per_bin = np.zeros(len(order), dtype=np.int64)
            per_bin[nonempty] = np.add.reduceat(mask.astype(np.int64), starts[nonempty])
            for bi in np.flatnonzero(per_bin).tolist():
                lo, hi = begin[bi], begin[bi + 1]
                out[(order[bi], m)] = eng[lo:hi][mask[lo:hi]]
        return out

    def positions(self, contig_names, mod_id):
        """(plus, minus): name -> ascending positions of confident rows, one entry for every listed contig that has
        at least one surviving row of this mod type (find_motifs_bin.py:629: contigs present in the bin pileup)."""
        order, key = self._index()
        plus, minus = {}, {}
        for name in contig_names:
            i = self._name_index.get(name)
            if i is None or self.kept[i, mod_id] == 0:
                continue
            k0 = (mod_id * len(self.contig_names) + i) * 2
            a, b, c = np.searchsorted(key, [k0, k0 + 1, k0 + 2])
            plus[name] = self.conf_position[order[a:b]]
            minus[name] = self.conf_position[order[b:c]]
        return plus, minus


def discover(cfg: ProcessorConfig, filtered: FilteredPileup, scorer: LockstepScorer, rank=0, bgzip_order=False,
             window_store=None, extractor=None):
    """Run every (bin, mod type) task of the data set.  ``filtered``: see FilteredPileup (identical on every rank);
    ``scorer``: see ``engine_scorer``; ``window_store``: where the methylation windows live (default: host numpy;
    the CLI passes the engine's device store); ``extractor``: an ``engine.DeviceWindowExtractor`` bound to that store
    to gather windows / count the background on the device instead of from ``cfg.assembly`` on the host.
    Returns (list of MotifRow, scorer) — identical on every rank."""
    import gc
    import time
    # the cyclic collector is paused for the duration of the searches: they build a few hundred thousand small objects
    # (motifs, graph nodes, rows) that all stay alive until the rows are written, and a full collection in the middle of
    # post-processing walks every one of them (0.05 s of a 0.3 s search at 1 Gbp); reference counting still frees the rest
    gc_was_on = gc.isenabled()
    gc.disable()
    files = DeferredFiles()                  # (the per-task files of a run with --out: written at the end, see the class)
    try:
        return _discover(cfg, filtered, scorer, rank, bgzip_order, window_store, extractor, files)
    finally:
        # also when a search or a post-processing step raised: what the tasks that had finished put aside — precleanup tables, PSSMs,
        # graphs — is on disk for whoever looks into the failed run, as with the reference's per-task writes (round-5 advisor)
        try:
            files.flush()
        finally:
            if gc_was_on:
                gc.enable()


def _discover(cfg, filtered, scorer, rank, bgzip_order, window_store, extractor, files):
    import time
    store = window_store if window_store is not None else HostWindowStore()
    t_mark = time.perf_counter()
    timings = {}

    def lap(name):
        nonlocal t_mark
        now = time.perf_counter()
        timings[name] = timings.get(name, 0.0) + now - t_mark
        t_mark = now
    planned = []
    bins = {}
    for c, b in cfg.bin_contig.items():
        bins.setdefault(b, []).append(c)
    tasks = {}
    native_rows = {}
    out_dir = cfg.output_dir
    # single GPU, windows on the device: the plan of every task goes to libnmscan in ONE call (nm_plan_windows: one gather
    # launch for all windows, the background draws on native threads meanwhile); contig-sharded multi-GPU runs and host
    # windows keep the per-task path below
    plan_natively = (extractor is not None and extractor.row_counts is not None and extractor.allreduce_i64 is None
                     and os.environ.get("NANOMOTIF_PLAN_PER_TASK") != "1")
    native_tasks = []
    present_ids = filtered.present_sorted_ids(bins, extractor.resident, len(MOD_TYPES)) if plan_natively else None
    # task order and seeding follow the reference: plain pileup = one task per (bin, mod type), each seeded afresh
    # (find_motifs_bin.py:152-171); bgzip = one task per bin, seeded once, mod types in constants order (:219-222, 248)
    for bin_name in bins:
        if extractor is not None:
            extractor._bin_open = False
        if bgzip_order:
            random.seed(cfg.seed)
        for mt_id, mod_type in enumerate(MOD_TYPES):
            if extractor is not None and extractor.row_counts is not None:
                plus = minus = None
                if plan_natively:
                    ids = present_ids.get((bin_name, mt_id))
                    if ids is not None:
                        native_tasks.append(((bin_name, mod_type), ids, mod_type))
                    continue
                names = filtered.present(bins[bin_name], mt_id)
                if not names:
                    continue
            else:
                plus, minus = filtered.positions(bins[bin_name], mt_id)
                if not plus and not minus:
                    continue
            if not bgzip_order:
                random.seed(cfg.seed)
            if extractor is not None:
                if plus is None and (not bgzip_order or not getattr(extractor, "_bin_open", False)):
                    extractor.begin_group(cfg.seed)      # a fresh generator stream: its draws are made in extractor.finish()
                    extractor._bin_open = True
                windows = None
                planned_ok = (extractor.plan_contigs((bin_name, mod_type), names, mod_type) if plus is None
                              else extractor.plan((bin_name, mod_type), plus, minus, mod_type))
                if not planned_ok:
                    log.info(f"[{bin_name} {mod_type}] No methylation sequences found")
                    continue
            else:
                windows = extract_windows(cfg.assembly, plus, minus, mod_type, cfg.padding)
                if windows is None:
                    log.info(f"[{bin_name} {mod_type}] No methylation sequences found")
                    continue
            stage_writer = None
            temp_dir = None
            if out_dir and rank == 0:
                pre = os.path.join(out_dir, "precleanup-motifs", f"{bin_name}-{mod_type}")
                os.makedirs(pre, exist_ok=True)
                stage_writer = (lambda pre: lambda name, rows: files.add(os.path.join(pre, name + ".tsv"), postprocess.format_motifs(rows)))(pre)
                stage_writer.text = (lambda pre: lambda name, text: files.add(os.path.join(pre, name + ".tsv"), text))(pre)
                temp_dir = os.path.join(out_dir, "temp", bin_name)
            if extractor is not None:
                planned.append(((bin_name, mod_type), stage_writer, temp_dir))
                continue
            store.add_task((bin_name, mod_type), windows[0])
            tasks[(bin_name, mod_type)] = task_coroutine(bin_name, mod_type, windows[1], cfg, stage_writer, temp_dir, files)
    lap("plan_s")
    if extractor is not None:
        if plan_natively:
            pssms = extractor.plan_all(native_tasks, cfg.seed, one_stream_per_bin=bgzip_order)
            for key, _, _ in native_tasks:
                if key not in pssms:
                    log.info(f"[{key[0]} {key[1]}] No methylation sequences found")
                    continue
                stage_writer = temp_dir = None
                if out_dir and rank == 0:
                    pre = os.path.join(out_dir, "precleanup-motifs", f"{key[0]}-{key[1]}")
                    os.makedirs(pre, exist_ok=True)
                    stage_writer = (lambda pre: lambda name, rows: files.add(os.path.join(pre, name + ".tsv"), postprocess.format_motifs(rows)))(pre)
                    stage_writer.text = (lambda pre: lambda name, text: files.add(os.path.join(pre, name + ".tsv"), text))(pre)
                    temp_dir = os.path.join(out_dir, "temp", key[0])
                planned.append((key, stage_writer, temp_dir))
        else:
            pssms = extractor.finish()
        lap("background_s")
        engine = getattr(store, "engine", None)
        if engine is not None and os.environ.get("NANOMOTIF_PY_SEARCH") != "1":
            # the searches of ALL tasks run inside libnmscan (nm_search_run: one window batch + one scoring batch per
            # lock-step round, no interpreter in the loop); only post-processing comes back to the coroutines below
            from . import native_search
            reduce = (lambda a: allreduce_counts(a, scorer.group)) if scorer.use_dist else None
            found = native_search.find_best_candidates_all(
                engine, [(key, store.task_id[key], store.totals[key], pssms[key]) for key, _, _ in planned], cfg.padding,
                cfg.minimum_kl_divergence, cfg.score_threshold, reduce=reduce)
            scorer.rounds += found.rounds
            scorer.candidates += found.candidates
            scorer.search_iterations += found.iterations
            scorer.speculation_hits += found.spec_hits
            scorer.speculation_misses += found.spec_misses
            lap("native_search_s")
            if os.environ.get("NANOMOTIF_PY_POST") == "1":
                for t, (key, stage_writer, temp_dir) in enumerate(planned):
                    tasks[key] = post_coroutine(key[0], key[1], found.result(t, full_graph=bool(temp_dir)), cfg, stage_writer, temp_dir, files)
            else:
                # post-processing of all tasks inside libnmscan as well (nm_post_run: noise -> clique merge in two scoring
                # batches on the merge stage's 0.3 / 0.7 classification -> sub-motifs -> complements); postprocess.py is its twin
                # (a run with --out: the five stage tables of every task come back as text, nm_post_tables — no row objects for them)
                want_tables = any(sw is not None for _, sw, _ in planned) and os.environ.get("NANOMOTIF_PY_TABLES") != "1"
                post = found.postprocess(engine, (engine.bin_index[key[0]] for key, _, _ in planned),
                                         (engine.slot_of_mod[state_label(key[1], "merge")] for key, _, _ in planned), reduce=reduce,
                                         tables=want_tables)
                scorer.rounds += post.batches
                scorer.candidates += post.candidates
                timings["post_native_call_s"] = time.perf_counter() - t_mark        # (part of postprocess_s: nm_post_run + the export of its rows)
                for t, (key, stage_writer, temp_dir) in enumerate(planned):
                    if temp_dir or found.none[t]:
                        write_search_artifacts(key[0], key[1], found.artifacts(t), temp_dir, files)
                    if stage_writer:
                        for s in range(post.n_stages(t)):
                            text = post.table_text(t, s)
                            if text is not None:
                                stage_writer.text(post.STAGES[s], text)
                            else:
                                stage_writer(post.STAGES[s], post.rows(t, s))
                    native_rows[key] = post.final(t)
                lap("postprocess_s")
            # (a thousand search graphs take 3 ms to free: off the critical path, on a thread of their own)
            import threading
            threading.Thread(target=found.close, name="nm-search-free", daemon=False).start()
        else:
            for key, stage_writer, temp_dir in planned:
                tasks[key] = task_coroutine(key[0], key[1], pssms[key], cfg, stage_writer, temp_dir, files)
    results = run_lockstep(tasks, scorer, store.execute)
    lap("coroutines_s")
    files.flush()
    lap("files_s")
    scorer.timings = timings
    rows = []
    for key in list(tasks) + list(native_rows):
        r = results.get(key) or native_rows.get(key)
        if r:
            rows += r
    log.info(f"scoring rounds: {scorer.rounds}, candidates scored: {scorer.candidates}")
    if scorer.search_iterations:
        log.info(f"search: {scorer.search_iterations} lock-step iterations; speculative children: {scorer.speculation_hits} answered, "
                 f"{scorer.speculation_misses} asked for after all")
    return rows, scorer


# ------------------------------------------------------------------------------------------------ reference-shaped helpers
def find_best_candidates(bin_pileup: BinData, bin_sequences: dict, mod_type: str, bin_name: str, output_dir,
                         low_meth_threshold, high_meth_threshold, padding, min_kl=0.2, max_dead_ends=25,
                         max_rounds_since_new_best=30, score_threshold=0.2, windows=None):
    """find_motifs_bin.py:606-839 for a single bin; ``windows`` = (methylation sets, background ASCII) from
    ``extract_windows``."""
    key = (bin_name, mod_type)
    store = HostWindowStore()
    store.add_task(key, windows[0])
    co = find_best_candidates_co(windows[1], mod_type, padding, min_kl=min_kl, max_dead_ends=max_dead_ends,
                                 max_rounds_since_new_best=max_rounds_since_new_best, score_threshold=score_threshold)
    res = run_lockstep({key: co}, bin_pileup.scorer, store.execute)[key]
    return None if res is None else (res[0], res[1])

"""Greedy candidate search — host side of the MI355X build.

Mirrors the reference's search (nanomotif/find_motifs_bin.py:606-1182, 1382-1433; seq.py:170-225, 391-422,
474-537; motif.py:577-607) decision for decision, but restructured for the GPU:

* every search is a *coroutine*: wherever the reference calls ``motif_model_bin`` (one full CPU scan per
  candidate) the coroutine ``yield``s the list of motifs it needs scored and is resumed with their models;
* ``run_lockstep`` advances all (bin, mod type) searches together and turns the requests of one round into ONE
  ``nm_score_batch`` launch, so a dependent step streams every bin once for all of its open candidates;
* windows are bit-set codes (uint8, bit0 A, bit1 C, bit2 G, bit3 T, 15 = N) instead of ``int64[n, 41, 4]``
  one-hot tensors; ``all(onehot <= motif_onehot)`` becomes ``(window & ~motif) == 0``.

Visit order, heap ties, thresholds and RNG consumption are those of the reference, so results are identical.
"""
from __future__ import annotations

import heapq
import math
import random

import numpy as np

from .model import BetaBernoulliModel, predictive_evaluation_score
from .motif import BASES, MOD_TYPE_TO_CANONICAL, Motif

A, C, G, T = 1, 2, 4, 8
ROW_BITS = (A, T, G, C)                    # PSSM / one-hot row order A, T, G, C (constants.py:1)
_SET_OF_ASCII = np.full(256, 255, dtype=np.uint8)
for _ch, _m in (("A", A), ("C", C), ("G", G), ("T", T), ("N", 15), (".", 15)):
    _SET_OF_ASCII[ord(_ch)] = _m
_COMP_SET = np.array([((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3) for m in range(16)], dtype=np.uint8)


# ------------------------------------------------------------------------------------------------
# windows and PSSMs
# ------------------------------------------------------------------------------------------------
def windows_at(seq: np.ndarray, indices: np.ndarray, padding: int) -> np.ndarray:
    """ASCII windows seq[i-pad : i+pad+1] for the indices with pad < i < len - pad (seq.py:170-189)."""
    idx = indices[(indices > padding) & (indices < len(seq) - padding)]
    if len(idx) == 0:
        return np.zeros((0, 2 * padding + 1), dtype=np.uint8)
    return seq[idx[:, None] + np.arange(-padding, padding + 1)[None, :]]


def to_sets(windows_ascii: np.ndarray) -> np.ndarray:
    """seq.py:474-478 — only A, C, G, T, N are convertible; anything else is the reference's KeyError."""
    sets = _SET_OF_ASCII[windows_ascii]
    if sets.size and sets.max() == 255:
        bad = chr(int(windows_ascii[sets == 255][0]))
        raise KeyError(bad)
    return sets


def revcomp_sets(sets: np.ndarray) -> np.ndarray:
    return _COMP_SET[sets[:, ::-1]]


class NativeRandom:
    """The interpreter's MT19937 state held in a numpy array for a run of native ``random.sample`` calls (libnmscan:
    nm_py_random_sample replicates CPython's algorithm bit for bit); on exit the interpreter's generator is set to
    where the pure-Python calls would have left it."""

    def __enter__(self):
        from . import _lib
        self._lib, self._check = _lib.load(), _lib.check
        self.version, state, self.gauss = random.getstate()
        self.state = np.array(state, dtype=np.uint32)
        return self

    def sample(self, n: int, k: int) -> np.ndarray:
        """``random.sample(range(n), k)`` as int64."""
        import ctypes as C
        out = np.empty(k, dtype=np.uint32)
        self._check(self._lib.nm_py_random_sample(self.state.ctypes.data_as(C.POINTER(C.c_uint32)), n, k,
                                                  out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out.astype(np.int64)

    def sample_many(self, ns, ks) -> np.ndarray:
        """Consecutive ``random.sample(range(n), k)`` calls for the pairs of ``ns`` / ``ks``, results back to back
        (uint32)."""
        import ctypes as C
        ns = np.ascontiguousarray(ns, dtype=np.uint64)
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        out = np.empty(max(int(ks.sum()), 1), dtype=np.uint32)
        self._check(self._lib.nm_py_random_sample_many(self.state.ctypes.data_as(C.POINTER(C.c_uint32)), len(ns),
                                                       ns.ctypes.data_as(C.POINTER(C.c_uint64)), ks.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                       out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out[:int(ks.sum())]

    def __exit__(self, *exc):
        random.setstate((self.version, tuple(self.state.tolist()), self.gauss))
        return False


def _native_random_sample(n: int, k: int) -> np.ndarray:
    with NativeRandom() as rng:
        return rng.sample(n, k)


def sample_background_starts(seq: np.ndarray, length: int, n: int, base: str) -> np.ndarray:
    """seq.py:202-225 — ``random.sample`` over the starts whose middle base is ``base``; same RNG consumption as
    sampling from the reference's list of valid starts (sample() only looks at len() and indexes)."""
    max_start = len(seq) - length + 1
    if n > max_start:
        raise ValueError("Too many samples requested for unique subsequences")
    mid = length // 2
    valid = np.flatnonzero(seq[mid:mid + max_start] == ord(base))
    if len(valid) < n:
        raise ValueError(f"Not enough subsequences with '{base}' in the middle (found {len(valid)}, need {n})")
    return valid[_native_random_sample(len(valid), n)]


def sample_background(seq: np.ndarray, length: int, n: int, base: str) -> np.ndarray:
    starts = sample_background_starts(seq, length, n, base)
    return seq[starts[:, None] + np.arange(length)[None, :]]


def letter_counts(seq: np.ndarray, starts: np.ndarray, length: int) -> np.ndarray:
    """int64[4, length]: exact-letter counts per column (rows A, T, G, C) over the windows seq[s : s + length]."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    out = np.zeros((4, length), dtype=np.int64)
    _lib.check(lib.nm_window_letter_counts(seq.ctypes.data_as(C.POINTER(C.c_uint8)), len(seq), starts.ctypes.data_as(C.POINTER(C.c_int64)),
                                           len(starts), length, out.ctypes.data_as(C.POINTER(C.c_int64))))
    return out


def letter_pssm(windows_ascii: np.ndarray) -> np.ndarray:
    """seq.py:391-422 — exact-letter frequency per column, rows A, T, G, C."""
    n = windows_ascii.shape[0]
    return np.array([(windows_ascii == ord(b)).sum(axis=0) / n for b in BASES])


_LANES = np.zeros(16, dtype=np.uint64)           # base set -> one 16-bit counter lane per PSSM row (A, T, G, C)
for _m in range(16):
    _LANES[_m] = sum(((_m & b) != 0) << (16 * k) for k, b in enumerate(ROW_BITS))


def sets_counts(sets: np.ndarray) -> np.ndarray:
    """Per column, how many windows carry A / T / G / C (int64[4, W], rows in the reference's order; an N window counts
    for all four).  One pass: every set is mapped to four 16-bit counter lanes of a uint64 and the rows are summed in
    slabs of < 65536 windows (integer counts, exact)."""
    n = sets.shape[0]
    counts = np.zeros((4, sets.shape[1]), dtype=np.int64)
    for lo in range(0, n, 65535):
        acc = _LANES[sets[lo:lo + 65535]].sum(axis=0, dtype=np.uint64)
        for k in range(4):
            counts[k] += ((acc >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    return counts


def sets_pssm(sets: np.ndarray) -> np.ndarray:
    """seq.py:526-537 on bit sets."""
    return sets_counts(sets) / sets.shape[0]


def motif_sets(motif: Motif) -> np.ndarray:
    return np.array(motif.sets, dtype=np.uint8)


def filter_matches(sets: np.ndarray, motif: Motif, keep_matches=True):
    """seq.py:499-524 — None when nothing is left."""
    bad = (sets & ~motif_sets(motif)[None, :]) != 0
    res = sets[~bad.any(axis=1)] if keep_matches else sets[bad.any(axis=1)]
    return None if res.shape[0] == 0 else res


# ------------------------------------------------------------------------------------------------
# graph of explored motifs (stands in for MotifTree(nx.DiGraph), motif.py:577-607)
# ------------------------------------------------------------------------------------------------
class MotifTree:
    def __init__(self):
        self.nodes = {}
        self._succ = {}
        self._pred = {}

    def has_node(self, m):
        return m in self.nodes

    def add_node(self, m, **attrs):
        if m in self.nodes:
            self.nodes[m].update(attrs)
        else:
            self.nodes[m] = dict(attrs)
            self._succ[m] = {}
            self._pred[m] = {}

    def has_edge(self, u, v):
        return u in self._succ and v in self._succ[u]

    def add_edge(self, u, v):
        for n in (u, v):
            if n not in self.nodes:
                self.add_node(n)
        self._succ[u][v] = True
        self._pred[v][u] = True

    def edges(self):
        return [(u, v) for u in self._succ for v in self._succ[u]]

    def predecessors(self, n):
        return list(self._pred[n])

    def _reach(self, start, table):
        seen, stack = set(), list(table[start])
        while stack:
            n = stack.pop()
            if n not in seen:
                seen.add(n)
                stack.extend(table[n])
        seen.discard(start)
        return seen

    def ancestors(self, n):
        return self._reach(n, self._pred)

    def descendants(self, n):
        return self._reach(n, self._succ)

    def get_missed_candidates(self, best_candidates, threshold=3):
        high = {n for n, d in self.nodes.items() if d["score"] > threshold}
        out = set()
        for n in high:
            if (not any(a in high for a in self.ancestors(n))
                    and not any(d in best_candidates for d in self.descendants(n)) and n not in best_candidates):
                out.add(n)
        return out

    def export_graph_gml(self, path):
        """Minimal GML writer (find_motifs_bin.py:836-837 / motif.py:630-652): counts, score, priority, depth."""
        with open(path, "w") as f:
            f.write(self.gml_text())

    def gml_text(self) -> str:
        ids = {n: i for i, n in enumerate(self.nodes)}
        out = ["graph [\n  directed 1\n"]
        for n, d in self.nodes.items():
            out.append(f'  node [\n    id {ids[n]}\n    label "{n.string.strip(".")}"\n    score {float(d.get("score", 0.0))}\n'
                       f'    priority {float(d.get("priority", 0))}\n    depth {int(d.get("depth", 0))}\n'
                       f'    visited {int(bool(d.get("visited", False)))}\n  ]\n')
        for u, v in self.edges():
            out.append(f"  edge [\n    source {ids[u]}\n    target {ids[v]}\n  ]\n")
        out.append("]\n")
        return "".join(out)


# ------------------------------------------------------------------------------------------------
# window requests: what the search asks of the windows of its (bin, mod type) task
# ------------------------------------------------------------------------------------------------
class WinReq:
    """``("pssm", motif)``  -> (n_active, int64 counts[4, W]) over the not-yet-removed windows that match ``motif``
                              (filter_sequence_matches(keep_matches=True) + pssm, seq.py:499-537);
    ``("remove", motif)`` -> (alive before, alive after): drop the matching windows (keep_matches=False, :803);
    ``("total", None)``   -> number of windows of the task."""
    __slots__ = ("kind", "motif")

    def __init__(self, kind, motif=None):
        self.kind, self.motif = kind, motif


class HostWindowStore:
    """Reference implementation of the window requests with numpy (CPU tests; the GPU path uses
    engine.DeviceWindowStore, same interface)."""

    def __init__(self):
        self.remaining = {}
        self.totals = {}

    def add_task(self, key, sets: np.ndarray):
        self.remaining[key] = sets
        self.totals[key] = int(sets.shape[0])

    def execute(self, batch):
        """batch: list of (key, WinReq) -> list of results."""
        out = []
        for key, req in batch:
            sets = self.remaining[key]
            if req.kind == "total":
                out.append(self.totals[key])
            elif req.kind == "pssm":
                active = filter_matches(sets, req.motif, keep_matches=True) if sets is not None else None
                out.append((0, None) if active is None else (int(active.shape[0]), sets_counts(active)))
            elif req.kind == "remove":
                before = 0 if sets is None else int(sets.shape[0])
                left = filter_matches(sets, req.motif, keep_matches=False) if sets is not None else None
                self.remaining[key] = left
                out.append((before, 0 if left is None else int(left.shape[0])))
            else:
                raise ValueError(req.kind)
        return out


# ------------------------------------------------------------------------------------------------
# coroutines: ``models = yield [motifs]``  /  ``result = yield WinReq(...)``
# ------------------------------------------------------------------------------------------------
def get_parent_scores_co(motif: Motif):
    """find_motifs_bin.py:1382-1433 as a coroutine: ONE request holds the motif and all its parents."""
    sp = motif.split()
    parents, positions = [], []
    for i, tok in enumerate(sp):
        if i == motif.mod_position or tok in (".", "N"):
            continue
        q = list(sp)
        q[i] = "."
        parents.append(Motif("".join(q), motif.mod_position))
        positions.append(i)
    models = yield [motif] + parents
    child_model = models[0]
    out = {}
    for parent, i, pm in zip(parents, positions, models[1:]):
        out[parent] = dict(motif_position=i, parent_model=pm, child_model=child_model,
                           score=predictive_evaluation_score(child_model, pm))
    return out


def kl_divergence_columns(pk: np.ndarray, qk: np.ndarray) -> np.ndarray:
    """``scipy.stats.entropy(pk, qk)`` along axis 0 (find_motifs_bin.py:974) — the same three operations scipy performs
    (normalise both to column sums, ``special.rel_entr``, column sum) without its argument-checking wrapper."""
    with np.errstate(invalid="ignore", divide="ignore"):
        pk = 1.0 * pk / np.sum(pk, axis=0, keepdims=True)
        qk = 1.0 * qk / np.sum(qk, axis=0, keepdims=True)
    from scipy.special import rel_entr                  # (lazy: the native search path never gets here)
    return np.sum(rel_entr(pk, qk), axis=0)


class MotifSearcher:
    """find_motifs_bin.py:843-1182.  ``run`` is a coroutine (see module docstring)."""

    def __init__(self, root_motif, bin_pssm, padding, motif_graph=None, min_kl=0.1,
                 freq_threshold=0.15, max_rounds_since_new_best=30, max_motif_length=25):
        if padding < 0:
            raise ValueError("padding must be non-negative.")
        self.root_motif = root_motif
        self.bin_pssm = bin_pssm
        self.padding = padding
        self.motif_graph = motif_graph or MotifTree()
        self.min_kl = min_kl
        self.freq_threshold = freq_threshold
        self.max_rounds_since_new_best = max_rounds_since_new_best
        self.max_motif_length = max_motif_length

    @staticmethod
    def _priority_function(next_model, root_model) -> float:
        try:
            d_alpha = 1 - (next_model._alpha / root_model._alpha)
        except ZeroDivisionError:
            d_alpha = 1
        try:
            d_beta = next_model._beta / root_model._beta
        except ZeroDivisionError:
            d_beta = 1
        return d_alpha * d_beta

    def _motif_child_nodes_kl_dist_max(self, motif, meth_pssm):
        """find_motifs_bin.py:957-1023: the single '.' column of maximal KL(meth || background), one child per
        base passing freq > 0.15 and freq > 0.5 * background, in A, T, G, C order."""
        kl = kl_divergence_columns(meth_pssm, self.bin_pssm)
        toks = motif.tokens
        masked = np.where(np.array([t == "." for t in toks]), kl, 0.0)
        if not any(t == "." for t in toks):
            return []
        if np.max(masked) < self.min_kl:
            return []
        pos = int(np.argmax(masked))
        keep = np.logical_and(meth_pssm[:, pos] > self.bin_pssm[:, pos] * 0.5, meth_pssm[:, pos] > self.freq_threshold)
        out = []
        for bi in np.flatnonzero(keep):
            q = list(toks)
            q[pos] = BASES[int(bi)]
            out.append(Motif("".join(q), motif.mod_position))
        return out

    def run(self):
        g = self.motif_graph
        best_guess = self.root_motif
        root_model = (yield [self.root_motif])[0]
        best_score = predictive_evaluation_score(root_model, root_model)
        rounds = 0
        visited = set()
        if not g.has_node(self.root_motif):
            g.add_node(self.root_motif, model=root_model, motif=self.root_motif, visited=False, score=best_score,
                       priority=0, depth=0)
        pq = [(0, 0, self.root_motif)]
        while pq:
            _, _, cur = heapq.heappop(pq)
            if cur in visited:
                continue
            attrs = g.nodes[cur]
            cur_model = attrs["model"]
            cur_depth = attrs.get("depth", 0)
            n_mod, n_nomod = cur_model.get_raw_counts()
            if n_mod + n_nomod < 10:
                continue
            if len(cur.strip()) > self.max_motif_length:
                continue
            visited.add(cur)
            g.nodes[cur]["visited"] = True
            rounds += 1
            n_active, counts = yield WinReq("pssm", cur)      # windows of this task still alive and matching cur
            if n_active == 0:
                continue
            neighbors = self._motif_child_nodes_kl_dist_max(cur, counts / n_active)
            fresh = [m for m in neighbors if m not in g.nodes]
            fresh_models = dict(zip(fresh, (yield fresh))) if fresh else {}     # one batch for all new children
            for nxt in neighbors:
                nxt_model = g.nodes[nxt]["model"] if nxt in g.nodes else fresh_models[nxt]
                score = predictive_evaluation_score(nxt_model, cur_model)
                n_iso = nxt.count_isolated_bases(isolation_size=1)
                priority = self._priority_function(nxt_model, root_model)
                if n_iso > 0:
                    priority *= pow(10, n_iso)
                if nxt in g.nodes:
                    if g.nodes[nxt]["score"] < score:
                        g.nodes[nxt]["score"] = score
                else:
                    g.add_node(nxt, model=nxt_model, motif=nxt, visited=False, score=score, priority=priority,
                               depth=cur_depth + 1)
                if not g.has_edge(cur, nxt):
                    g.add_edge(cur, nxt)
                if nxt not in visited:
                    a = g.nodes[nxt]
                    heapq.heappush(pq, (a["priority"], a["depth"], nxt))
                if score > best_score:
                    best_score, best_guess, rounds = score, nxt, 0
            if rounds >= self.max_rounds_since_new_best:
                break
        return g, best_guess


def extract_windows(contigs: dict, plus_pos: dict, minus_pos: dict, mod_type: str, padding: int,
                    background_sampling_frequency=0.01):
    """find_motifs_bin.py:625-686.  contigs: name -> uint8 upper-case ASCII; plus_pos / minus_pos: name -> int64
    positions of the confidently methylated rows (fraction_mod >= high).  Contigs are visited in sorted-name order
    (the reference's polars ``unique()`` order is unspecified).  Returns (methylation window sets uint8[n, W],
    background PSSM float64[4, W] = ``background_sequences.pssm()``) or None."""
    canonical = MOD_TYPE_TO_CANONICAL[mod_type]
    W = 2 * padding + 1
    meth = []
    bg_counts, n_bg = np.zeros((4, W), dtype=np.int64), 0
    for name in sorted(plus_pos.keys() | minus_pos.keys()):
        seq = contigs[name]
        n_samples = int(max(math.ceil(len(seq) * background_sampling_frequency), 50))
        starts = sample_background_starts(seq, W, n_samples, canonical)
        bg_counts += letter_counts(seq, starts, W)
        n_bg += n_samples
        p, m = plus_pos.get(name, np.zeros(0, np.int64)), minus_pos.get(name, np.zeros(0, np.int64))
        here = []
        if len(p) >= 1:
            here.append(to_sets(windows_at(seq, p, padding)))
        if len(m) >= 1:
            here.append(revcomp_sets(to_sets(windows_at(seq, m, padding))))
        here = [h for h in here if h.shape[0]]
        if not here:
            return None                          # find_motifs_bin.py:662-664
        meth += here
    if not meth or n_bg == 0:
        return None
    return np.concatenate(meth), bg_counts / n_bg


def find_best_candidates_co(bin_pssm, mod_type: str, padding: int, min_kl=0.2, max_dead_ends=25,
                            max_rounds_since_new_best=30, score_threshold=0.2, remaining_sequences_threshold=0.001,
                            log=None):
    """find_motifs_bin.py:688-839 as a coroutine.  ``bin_pssm``: the background PSSM of the task (float64[4, W]); the
    methylation windows live in the window store under this coroutine's key and are reached through ``WinReq``s.
    Returns (graph, best_candidates, bin_pssm) or None."""
    total = yield WinReq("total")
    root = Motif("." * padding + MOD_TYPE_TO_CANONICAL[mod_type] + "." * padding, padding)
    best, dead_ends, graph = [], 0, None
    while True:
        if dead_ends >= max_dead_ends:
            break
        searcher = MotifSearcher(root, bin_pssm, padding, motif_graph=graph, min_kl=min_kl,
                                 max_rounds_since_new_best=max_rounds_since_new_best)
        graph, guess = yield from searcher.run()
        if guess == root:
            break
        temp, to_prune, single = guess, set(), False
        while True:
            parents = yield from get_parent_scores_co(temp)
            for d in parents.values():
                if d["score"] < 0.4:
                    to_prune.add(d["motif_position"])
            if not to_prune:
                break
            sp = temp.split()
            for i in to_prune:
                sp[i] = "."
            pruned = Motif("".join(sp), temp.mod_position)
            if len(pruned.string.replace(".", "")) == 1:
                single = True
                break
            if pruned == temp:
                break
            temp = pruned
        mean_parent = np.mean([d["score"] for d in parents.values()])
        if single or mean_parent < score_threshold or temp == guess:
            graph.nodes[guess]["score"] = mean_parent
        else:
            child_model = next(iter(parents.values()))["child_model"]
            graph.add_node(temp, model=child_model, motif=temp, visited=True, score=mean_parent, priority=0, depth=0)
            guess = temp
        before, left = yield WinReq("remove", guess)
        if left == 0:
            break
        if graph.nodes[guess]["score"] < score_threshold:
            dead_ends += 1
            continue
        if log:
            log(f"Keeping {guess}, represented in {before - left} seqs. model: {graph.nodes[guess]['model']}. "
                f"({100 * left / total:.1f} % of sequences remaining)")
        best.append(guess)
        if left / total < remaining_sequences_threshold:
            break
    if graph is None or len(graph.nodes) == 0:
        return None
    missed = graph.get_missed_candidates(best, score_threshold)
    missed = [c for c in missed if not c.sub_motif_of_any(best) or not any(b.sub_motif_of(c) for b in best)]
    best.extend(sorted(missed, key=lambda m: (m.string, m.mod_position)))
    return graph, best, bin_pssm


# ------------------------------------------------------------------------------------------------
# lock-step scheduler
# ------------------------------------------------------------------------------------------------
class TaggedRequest(list):
    """A scoring request that must be evaluated on another classification of the pileup than the search's own:
    ``tag == "merge"`` = the reference's hard-wired 0.3 / 0.7 thresholds of merge_motifs_in_df
    (find_motifs_bin.py:569, 1436), whatever the CLI thresholds are."""

    def __init__(self, motifs, tag):
        super().__init__(motifs)
        self.tag = tag


def run_lockstep(coroutines: dict, score_fn, window_fn=None):
    """Advance all coroutines together.  ``coroutines``: key -> generator that yields either a list of Motif
    (optionally a ``TaggedRequest``) to be scored, or a ``WinReq``.  Every round, all pending scoring requests go to
    ``score_fn(list of (key, Motif, tag)) -> int64[n, 2]`` in one batch and all pending window requests to
    ``window_fn(list of (key, WinReq)) -> list of results`` in one batch.  Returns key -> the coroutine's return value."""
    results, waiting = {}, {}

    def advance(key, value, first=False):
        try:
            waiting[key] = next(coroutines[key]) if first else coroutines[key].send(value)
        except StopIteration as e:
            waiting.pop(key, None)
            results[key] = e.value

    for key in list(coroutines):
        advance(key, None, first=True)
    while waiting:
        score_keys = [k for k, r in waiting.items() if not isinstance(r, WinReq)]
        win_keys = [k for k, r in waiting.items() if isinstance(r, WinReq)]
        replies = {}
        if win_keys:
            if window_fn is None:
                raise RuntimeError("a coroutine issued a window request but no window store was given")
            for k, res in zip(win_keys, window_fn([(k, waiting[k]) for k in win_keys])):
                replies[k] = res
        if score_keys:
            flat = [(k, m, getattr(waiting[k], "tag", None)) for k in score_keys for m in waiting[k]]
            counts = score_fn(flat)
            at = 0
            for k in score_keys:
                n = len(waiting[k])
                replies[k] = [BetaBernoulliModel.from_counts(*counts[at + j]) for j in range(n)]
                at += n
        for k, v in replies.items():
            advance(k, v)
    return results

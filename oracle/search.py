"""Oracle restatement of the greedy candidate search
(reference: nanomotif/find_motifs_bin.py:606-1182, 1382-1433; seq.py:170-225, 391-422, 474-537;
motif.py:577-607).  Test infrastructure only — see oracle/__init__.py.

Contig iteration order: the reference walks ``bin_pileup["contig"].unique()`` (find_motifs_bin.py:629),
whose order polars leaves unspecified; the fixtures (and this restatement) pin *sorted contig name*.
"""
from __future__ import annotations

import heapq
import math
import random

import numpy as np
from scipy.stats import entropy

from .model import BetaBernoulliModel, predictive_evaluation_score
from .motif import BASES, MOD_TYPE_TO_CANONICAL, ONE_HOT, Motif, reverse_compliment
from .scan import motif_model_bin


# ------------------------------------------------------------------------------------------------
# windows and PSSMs
# ------------------------------------------------------------------------------------------------
def sample_at_indices(seq: str, indices, padding: int):
    """seq.py:170-189 — strict ``padding < i < len - padding``; window = seq[i-pad : i+pad+1]."""
    L = len(seq)
    return [seq[i - padding:i + padding + 1] for i in indices if padding < i < L - padding]


def sample_n_subsequences_unique(seq: str, length: int, n: int, base: str):
    """seq.py:202-225 — ``random.sample`` over the starts whose middle base is ``base``."""
    max_start = len(seq) - length + 1
    if n > max_start:
        raise ValueError("Too many samples requested for unique subsequences")
    mid = length // 2
    valid = [s for s in range(max_start) if seq[s + mid] == base]
    if len(valid) < n:
        raise ValueError(f"Not enough subsequences with '{base}' in the middle (found {len(valid)}, need {n})")
    return [seq[s:s + length] for s in random.sample(valid, n)]


def letter_pssm(windows) -> np.ndarray:
    """seq.py:391-422 — exact-letter frequency per column, rows A,T,G,C, pseudocount 0."""
    arr = np.frombuffer("".join(windows).encode("ascii"), dtype=np.uint8).reshape(len(windows), -1)
    return np.array([(arr == ord(b)).sum(axis=0) / len(windows) for b in BASES])


def to_onehot(windows) -> np.ndarray:
    """seq.py:474-478 — int[n, W, 4]; N -> 1111; other IUPAC letters raise KeyError like the reference."""
    return np.array([[ONE_HOT[c] for c in w] for w in windows], dtype=np.int64)


def filter_matches(onehot: np.ndarray, motif_onehot: np.ndarray, keep_matches=True):
    """seq.py:499-524 — rows with all(x <= m) (keep) or any(x > m) (drop matches); None when empty."""
    if keep_matches:
        res = onehot[np.all(onehot <= motif_onehot, axis=(1, 2))]
    else:
        res = onehot[np.any(onehot > motif_onehot, axis=(1, 2))]
    return None if res.shape[0] == 0 else res


def onehot_pssm(onehot: np.ndarray) -> np.ndarray:
    """seq.py:526-537."""
    return onehot.sum(axis=0).transpose() / onehot.shape[0]


# ------------------------------------------------------------------------------------------------
# motif graph (stands in for MotifTree(nx.DiGraph), motif.py:577-607)
# ------------------------------------------------------------------------------------------------
class MotifGraph:
    def __init__(self):
        self.nodes = {}      # Motif -> attribute dict, insertion ordered
        self.succ = {}
        self.pred = {}

    def has_node(self, m):
        return m in self.nodes

    def add_node(self, m, **attrs):
        if m in self.nodes:
            self.nodes[m].update(attrs)
        else:
            self.nodes[m] = dict(attrs)
            self.succ[m] = {}
            self.pred[m] = {}

    def has_edge(self, u, v):
        return u in self.succ and v in self.succ[u]

    def add_edge(self, u, v):
        for n in (u, v):
            if n not in self.nodes:
                self.add_node(n)
        self.succ[u][v] = True
        self.pred[v][u] = True

    def edges(self):
        return [(u, v) for u in self.succ for v in self.succ[u]]

    def _reach(self, start, table):
        seen, stack = set(), list(table[start])
        while stack:
            n = stack.pop()
            if n in seen:
                continue
            seen.add(n)
            stack.extend(table[n])
        seen.discard(start)
        return seen

    def ancestors(self, n):
        return self._reach(n, self.pred)

    def descendants(self, n):
        return self._reach(n, self.succ)

    def get_missed_candidates(self, best_candidates, threshold):
        """motif.py:594-607."""
        high = {n for n, d in self.nodes.items() if d["score"] > threshold}
        out = set()
        for n in high:
            if (not any(a in high for a in self.ancestors(n))
                    and not any(d in best_candidates for d in self.descendants(n))
                    and n not in best_candidates):
                out.add(n)
        return out


# ------------------------------------------------------------------------------------------------
# scoring of parents (pruning)
# ------------------------------------------------------------------------------------------------
def get_parent_scores(motif: Motif, pileup, contigs, low, high):
    """find_motifs_bin.py:1382-1433 — ordered dict parent -> info."""
    child_model = motif_model_bin(pileup, contigs, motif, BetaBernoulliModel(), low, high)
    sp = motif.split()
    parents = {}
    for i, tok in enumerate(sp):
        if i == motif.mod_position or tok in (".", "N"):
            continue
        q = list(sp)
        q[i] = "."
        parent = Motif("".join(q), motif.mod_position)
        parent_model = motif_model_bin(pileup, contigs, parent, BetaBernoulliModel(), low, high)
        parents[parent] = dict(motif_position=i, parent_model=parent_model, child_model=child_model,
                               score=predictive_evaluation_score(child_model, parent_model))
    return parents


# ------------------------------------------------------------------------------------------------
# best-first search
# ------------------------------------------------------------------------------------------------
class MotifSearcher:
    """find_motifs_bin.py:843-1182."""

    def __init__(self, root_motif, contigs, bin_pssm, pileup, methylation_onehot, padding, high, low,
                 motif_graph=None, min_kl=0.1, freq_threshold=0.15, max_rounds_since_new_best=30,
                 max_motif_length=25):
        self.root_motif = root_motif
        self.contigs = contigs
        self.bin_pssm = bin_pssm
        self.pileup = pileup
        self.methylation_onehot = methylation_onehot
        self.padding = padding
        self.graph = motif_graph or MotifGraph()
        self.min_kl = min_kl
        self.freq_threshold = freq_threshold
        self.max_rounds = max_rounds_since_new_best
        self.max_motif_length = max_motif_length
        self.low, self.high = low, high
        self.visit_order = []

    def _model(self, motif):
        return motif_model_bin(self.pileup, self.contigs, motif, BetaBernoulliModel(), self.low, self.high)

    @staticmethod
    def _priority(next_model, root_model):
        """find_motifs_bin.py:901-924."""
        try:
            d_alpha = 1 - (next_model._alpha / root_model._alpha)
        except ZeroDivisionError:
            d_alpha = 1
        try:
            d_beta = next_model._beta / root_model._beta
        except ZeroDivisionError:
            d_beta = 1
        return d_alpha * d_beta

    def _children(self, motif, meth_pssm):
        """find_motifs_bin.py:957-1023."""
        kl = entropy(meth_pssm, self.bin_pssm)
        sp = motif.split()
        dots = np.array([i for i, t in enumerate(sp) if t == "."])
        if dots.size == 0:
            return []
        masked = kl.copy()
        masked[~np.isin(np.arange(len(sp)), dots)] = 0
        if np.max(masked) < self.min_kl:
            return []
        pos = int(np.argmax(masked))
        keep = np.logical_and(meth_pssm[:, pos] > self.bin_pssm[:, pos] * 0.5, meth_pssm[:, pos] > self.freq_threshold)
        out = []
        for bi in np.argwhere(keep).reshape(-1):
            q = list(sp)
            q[pos] = BASES[int(bi)]
            out.append(Motif("".join(q), motif.mod_position))
        return out

    def run(self):
        """find_motifs_bin.py:1026-1182."""
        g = self.graph
        best_guess = self.root_motif
        root_model = self._model(self.root_motif)
        best_score = predictive_evaluation_score(root_model, root_model)
        rounds = 0
        visited = set()
        if not g.has_node(self.root_motif):
            g.add_node(self.root_motif, model=root_model, motif=self.root_motif, visited=False,
                       score=best_score, priority=0, depth=0)
        pq = []
        heapq.heappush(pq, (0, 0, self.root_motif))
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
            self.visit_order.append(cur)
            g.nodes[cur]["visited"] = True
            rounds += 1
            active = filter_matches(self.methylation_onehot, cur.one_hot(), keep_matches=True)
            if active is None:
                continue
            for nxt in self._children(cur, onehot_pssm(active)):
                if nxt not in g.nodes:
                    nxt_model = self._model(nxt)
                else:
                    nxt_model = g.nodes[nxt]["model"]
                score = predictive_evaluation_score(nxt_model, cur_model)
                n_iso = nxt.count_isolated_bases(isolation_size=1)
                priority = self._priority(nxt_model, root_model)
                if n_iso > 0:
                    priority *= pow(10, n_iso)
                if nxt in g.nodes:
                    if g.nodes[nxt]["score"] < score:
                        g.nodes[nxt]["score"] = score
                else:
                    g.add_node(nxt, model=nxt_model, motif=nxt, visited=False, score=score,
                               priority=priority, depth=cur_depth + 1)
                if not g.has_edge(cur, nxt):
                    g.add_edge(cur, nxt)
                if nxt not in visited:
                    a = g.nodes[nxt]
                    heapq.heappush(pq, (a["priority"], a["depth"], nxt))
                if score > best_score:
                    best_score, best_guess, rounds = score, nxt, 0
            if rounds >= self.max_rounds:
                break
        return g, best_guess


# ------------------------------------------------------------------------------------------------
# outer greedy loop
# ------------------------------------------------------------------------------------------------
def extract_windows(pileup: dict, contigs: dict, mod_type: str, high: float, padding: int,
                    background_sampling_frequency=0.01):
    """find_motifs_bin.py:625-686.  Returns (methylation windows, background windows) or None."""
    meth, bg = [], []
    canonical = MOD_TYPE_TO_CANONICAL[mod_type]
    for name in sorted(pileup):
        p = pileup[name]
        if len(p) == 0:
            continue
        seq = contigs[name]
        conf = p.fraction_mod >= high
        plus = p.position[conf & (p.strand == ord("+"))].tolist()
        minus = p.position[conf & (p.strand == ord("-"))].tolist()
        n_samples = int(max(math.ceil(len(seq) * background_sampling_frequency), 50))
        bg += sample_n_subsequences_unique(seq, padding * 2 + 1, n_samples, canonical)
        here = []
        if plus:
            here += sample_at_indices(seq, plus, padding)
        if minus:
            here += [reverse_compliment(w) for w in sample_at_indices(seq, minus, padding)]
        if not here:
            return None                     # find_motifs_bin.py:662-664 — whole (bin, modtype) gives up
        meth += here
    if not meth or not bg:
        return None
    return meth, bg


def find_best_candidates(pileup: dict, contigs: dict, mod_type: str, low: float, high: float, padding: int,
                         min_kl=0.2, max_dead_ends=25, max_rounds_since_new_best=30, score_threshold=0.2,
                         remaining_sequences_threshold=0.001, trace=None):
    """find_motifs_bin.py:606-839.  Returns (graph, best_candidates, bin_pssm) or None."""
    w = extract_windows(pileup, contigs, mod_type, high, padding)
    if w is None:
        return None
    meth_windows, bg_windows = w
    onehot = to_onehot(meth_windows)
    total = onehot.shape[0]
    bin_pssm = letter_pssm(bg_windows)
    root = Motif("." * padding + MOD_TYPE_TO_CANONICAL[mod_type] + "." * padding, padding)
    remaining = onehot.copy()
    best, dead_ends, graph = [], 0, None
    while True:
        if dead_ends >= max_dead_ends:
            break
        searcher = MotifSearcher(root, contigs, bin_pssm, pileup, remaining, padding, high, low,
                                 motif_graph=graph, min_kl=min_kl, max_rounds_since_new_best=max_rounds_since_new_best)
        graph, guess = searcher.run()
        if trace is not None:
            trace.append(("search", guess, list(searcher.visit_order)))
        if guess == root:
            break
        # pruning (find_motifs_bin.py:721-768)
        temp, to_prune, single = guess, set(), False
        while True:
            parents = get_parent_scores(temp, pileup, contigs, low, high)
            mean_parent = np.mean([d["score"] for d in parents.values()])
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
        before = remaining.shape[0]
        remaining = filter_matches(remaining, guess.one_hot(), keep_matches=False)
        if remaining is None:
            break
        if graph.nodes[guess]["score"] < score_threshold:
            dead_ends += 1
            continue
        best.append(guess)
        if trace is not None:
            trace.append(("keep", guess, before - remaining.shape[0]))
        if remaining.shape[0] / total < remaining_sequences_threshold:
            break
    if graph is None or len(graph.nodes) == 0:
        return None
    missed = graph.get_missed_candidates(best, score_threshold)
    missed = [c for c in missed
              if not c.sub_motif_of_any(best) or not any(b.sub_motif_of(c) for b in best)]
    best.extend(missed)
    return graph, best, bin_pssm

"""Post-processing of one (bin, mod type) search result and the output writers — mirrors
nanomotif/postprocess.py:7-109, find_motifs_bin.py:537-596 and 1436-1537, motif.py:654-926.

The reference keeps motifs in a polars-backed ``MotifSearchResult``; here a result is a plain list of
``MotifRow`` records.  The merge stage re-enters scoring (merged motif, its pre-merge variants, its parents):
it is a coroutine like the search, so all of it is batched through the HIP engine."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from .model import BetaBernoulliModel, predictive_evaluation_score
from .motif import Motif, merge_motifs, motif_type, reverse_compliment
from .search import TaggedRequest, get_parent_scores_co


@dataclass
class MotifRow:
    reference: str
    motif: str
    mod_type: str
    mod_position: int
    model: BetaBernoulliModel
    score: float
    complement: "MotifRow | None" = field(default=None, compare=False)
    has_complement_columns: bool = False

    # derived columns (motif.py:774-818)
    @property
    def n_mod(self):
        return int(self.model._alpha - self.model._alpha_prior)

    @property
    def n_nomod(self):
        return int(self.model._beta - self.model._beta_prior)

    def _derived(self):
        """(motif_iupac, mod_position_iupac, reverse complement of motif_iupac), computed once per row (rows are never
        mutated after creation; the native post-processing hands the first two over and leaves the third for the first use)."""
        d = self.__dict__.get("_cache")
        if d is None:
            st = self.as_motif().new_stripped_motif()
            iu = st.iupac()
            d = self.__dict__["_cache"] = (iu, int(st.mod_position), reverse_compliment(iu))
        elif d[2] is None:
            d = self.__dict__["_cache"] = (d[0], d[1], reverse_compliment(d[0]))
        return d

    def as_motif(self):
        m = self.__dict__.get("_motif")
        if m is None:
            m = self.__dict__["_motif"] = Motif(self.motif, self.mod_position)
        return m

    @property
    def motif_iupac(self):
        d = self.__dict__.get("_cache")
        return d[0] if d is not None else self._derived()[0]

    @property
    def mod_position_iupac(self):
        d = self.__dict__.get("_cache")
        return d[1] if d is not None else self._derived()[1]

    def key(self):
        """What ``DataFrame.unique()`` compares (find_motifs_bin.py:570, 579, 588): every cell of the row.  The ``model`` cells
        are Python objects in an Object column and py-polars hashes / compares those through ``__hash__`` / ``__eq__`` — identity
        for ``BetaBernoulliModel`` (model.py defines neither) — so two rows are one only when they hold the SAME model objects.
        Two merge clusters that produce the same merged motif each get a model of their own (find_motifs_bin.py:1497-1504): the
        reference keeps both rows and so does this (fixture g13; the stand-in frame of the fixtures compares the same way)."""
        return (self.reference, self.motif, self.mod_type, self.mod_position, id(self.model), self.score,
                None if self.complement is None else (self.complement.motif, self.complement.mod_position, id(self.complement.model)))


def unique(rows):
    seen, out = set(), []
    for r in rows:
        if r.key() not in seen:
            seen.add(r.key())
            out.append(r)
    return out


def graph_to_rows(graph, best, bin_name, mod_type, padding):
    """find_motifs_bin.py:537-549 — graph nodes that are best candidates, score-descending."""
    rows = [MotifRow(bin_name, n.string, mod_type, padding, d["model"], float(d["score"]))
            for n, d in graph.nodes.items() if n in best]
    rows.sort(key=lambda r: -r.score)
    return rows


def remove_noisy_motifs(rows):
    """postprocess.py:7-25."""
    clean = {r.motif for r in rows if not r.as_motif().have_isolated_bases(isolation_size=3)}
    return rows if not clean else [r for r in rows if r.motif in clean]


def get_motif_parental_relationship(motifs):
    """postprocess.py:41-49."""
    rel = []
    for i, m1 in enumerate(motifs):
        for j, m2 in enumerate(motifs):
            if i != j and m1.sub_string_of(m2) and (m2, m1) not in rel:
                rel.append((m2, m1))
    return rel


def _tagged(co, tag):
    """Re-issue every request of coroutine ``co`` as a TaggedRequest."""
    try:
        req = next(co)
        while True:
            req = co.send((yield TaggedRequest(req, tag)))
    except StopIteration as e:
        return e.value


def merge_motifs_co(rows, merge_threshold=0.5):
    """find_motifs_bin.py:1436-1537 for one (bin, mod type) group, as a scoring coroutine.  The reference evaluates
    this stage at its default thresholds 0.3 / 0.7 (find_motifs_bin.py:569, 1436) regardless of the CLI values, so
    every request is tagged ``"merge"`` and the scorer routes it to the 0.3 / 0.7 classification of the pileup."""
    return (yield from _tagged(_merge_motifs_co(rows, merge_threshold), "merge"))


def _merge_motifs_co(rows, merge_threshold=0.5):
    if not rows:
        return rows
    bin_name, mod_type = rows[0].reference, rows[0].mod_type
    motifs = [r.as_motif() for r in rows]
    clusters = merge_motifs(motifs)
    # one batch: every merged motif and every pre-merge variant that needs a scan
    need = [(k, c) for k, c in enumerate(clusters) if len(c[3]) > 0]
    request, spans = [], []
    for k, (merged, cluster, pre, new) in need:
        pre_sorted = sorted(pre, key=lambda m: (m.string, m.mod_position))
        spans.append((len(request), len(pre_sorted)))
        request += [merged] + pre_sorted
    models = (yield request) if request else []
    accepted, premerge = [], []
    verdict = {}
    for (k, _), (at, npre) in zip(need, spans):
        merge_model = models[at]
        pre_model = BetaBernoulliModel()
        for m in models[at + 1:at + 1 + npre]:
            pre_model.update(*m.get_raw_counts())
        verdict[k] = predictive_evaluation_score(pre_model, merge_model) < merge_threshold
    for k, (merged, cluster, pre, new) in enumerate(clusters):
        if len(new) == 0 or verdict[k]:
            accepted.append(merged)
            premerge.extend(cluster)
    if not premerge:
        return rows
    pre_strings = {m.string for m in premerge}
    out = [r for r in rows if r.motif not in pre_strings]
    for m in accepted:
        parents = yield from get_parent_scores_co(m)          # request = [m] + parents; child_model is m's model
        if parents:
            model = next(iter(parents.values()))["child_model"]
            score = float(np.mean([d["score"] for d in parents.values()]))
        else:
            model = (yield [m])[0]
            score = -1
        out.append(MotifRow(bin_name, m.string, mod_type, int(m.mod_position), model, score))
    return out


def remove_sub_motifs(rows):
    """postprocess.py:52-82 for one (reference, mod_type) group."""
    if len(rows) < 2:
        return list(rows)
    motifs = [r.as_motif() for r in rows]
    group = list(rows)
    out = list(rows)
    for parent, child in get_motif_parental_relationship(motifs):
        def model_of(m):
            return next(r.model for r in group if r.motif == m.string and r.mod_position == m.mod_position)
        s = predictive_evaluation_score(model_of(child), model_of(parent))
        drop = parent if s > 0.5 else child
        out = [r for r in out if not (r.motif == drop.string and r.mod_position == drop.mod_position)]
    return out


def join_motif_complements(rows):
    """postprocess.py:85-109."""
    out = []
    for r in rows:
        partners = [o for o in rows if o.reference == r.reference and o.mod_type == r.mod_type
                    and o._derived()[2] == r.motif_iupac]
        if not partners:
            out.append(MotifRow(r.reference, r.motif, r.mod_type, r.mod_position, r.model, r.score, None, True))
            continue
        for o in partners:
            if r.motif_iupac >= o.motif_iupac:
                out.append(MotifRow(r.reference, r.motif, r.mod_type, r.mod_position, r.model, r.score, o, True))
    return out


def postprocess_co(graph, best, bin_name, mod_type, padding, on_stage=None):
    """find_motifs_bin.py:537-596 — noise -> merge -> sub-motifs -> complements; returns rows or None."""
    rows = graph_to_rows(graph, best, bin_name, mod_type, padding)
    if not rows:
        return None
    stage = on_stage or (lambda name, r: None)
    stage("motifs", rows)
    rows = remove_noisy_motifs(rows)
    if not rows:
        return None
    stage("motifs-noise", rows)
    rows = unique((yield from merge_motifs_co(rows)))
    if not rows:
        return None
    stage("motifs-noise-merge", rows)
    rows = unique(remove_sub_motifs(rows))
    if not rows:
        return None
    stage("motifs-noise-merge-sub", rows)
    rows = unique(join_motif_complements(rows))
    if not rows:
        return None
    stage("motifs-noise-merge-sub-complement", rows)
    return rows


# ------------------------------------------------------------------------------------------------ writers
HEADER = ["reference", "motif", "mod_position", "mod_type", "n_mod", "n_nomod", "motif_type", "motif_complement",
          "mod_position_complement", "n_mod_complement", "n_nomod_complement"]


def _fmt(v):
    return "" if v is None else str(v)


def format_bin_motifs(rows) -> str:
    """motif.py:899-926 — bin-motifs.tsv text (rows sorted by reference, mod_type, motif)."""
    rows = sorted(rows, key=lambda r: (r.reference, r.mod_type, r.motif_iupac))
    lines = ["\t".join(HEADER)]
    for r in rows:
        c = r.complement
        lines.append("\t".join([r.reference, r.motif_iupac, str(r.mod_position_iupac), r.mod_type, str(r.n_mod),
                                str(r.n_nomod), motif_type(r.motif_iupac), _fmt(c and c.motif_iupac),
                                _fmt(None if c is None else c.mod_position_iupac), _fmt(None if c is None else c.n_mod),
                                _fmt(None if c is None else c.n_nomod)]))
    return "\n".join(lines) + "\n"


def write_motif_formatted(rows, path):
    with open(path, "w") as f:
        f.write(format_bin_motifs(rows))


def _fmt_float(x):
    return repr(float(x))


def write_motifs(rows, path):
    """motif.py:891-897 — the per-stage precleanup TSVs (all non-object columns, sorted)."""
    with open(path, "w") as f:
        f.write(format_motifs(rows))


def format_motifs(rows) -> str:
    """The text ``write_motifs`` writes."""
    rows = sorted(rows, key=lambda r: (r.reference, r.mod_type, r.motif))
    cols = ["reference", "motif", "mod_type", "mod_position", "score", "n_mod", "n_nomod", "motif_iupac", "mod_position_iupac"]
    comp = any(r.has_complement_columns for r in rows)
    if comp:
        cols += ["motif_complement", "mod_position_complement", "score_complement", "n_mod_complement",
                 "n_nomod_complement", "motif_iupac_complement", "mod_position_iupac_complement"]
    out = ["\t".join(cols) + "\n"]
    for r in rows:
        vals = [r.reference, r.motif, r.mod_type, str(r.mod_position), _fmt_float(r.score), str(r.n_mod), str(r.n_nomod),
                r.motif_iupac, str(r.mod_position_iupac)]
        if comp:
            c = r.complement
            vals += ["" if c is None else x for x in
                     ([None] * 7 if c is None else [c.motif, str(c.mod_position), _fmt_float(c.score), str(c.n_mod),
                                                    str(c.n_nomod), c.motif_iupac, str(c.mod_position_iupac)])]
        out.append("\t".join(vals) + "\n")
    return "".join(out)

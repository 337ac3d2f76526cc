"""Oracle restatement of the per-(bin, modtype) post-processing chain
(reference: nanomotif/find_motifs_bin.py:537-596, 1436-1537; postprocess.py:7-109; motif.py:811-814, 899-926).
Test infrastructure only.  The reference versions are polars-DataFrame code and cannot run here; restated from
source and pinned by the known-answer values of tests/test_postprocess.py and tests/test_motif_find.py:42-84.

A motif table is a list of row dicts: ``reference, motif, mod_type, mod_position, model, score`` (+ derived).
"""
from __future__ import annotations

import numpy as np

from .model import BetaBernoulliModel, predictive_evaluation_score
from .motif import Motif, merge_motifs, motif_type, reverse_compliment
from .scan import motif_model_bin
from .search import get_parent_scores


def derive(row):
    """motif.py:774-818 — n_mod / n_nomod / motif_iupac / mod_position_iupac."""
    r = dict(row)
    a, b = r["model"].get_raw_counts()
    r["n_mod"], r["n_nomod"] = int(a), int(b)
    st = Motif(r["motif"], r["mod_position"]).new_stripped_motif()
    r["motif_iupac"], r["mod_position_iupac"] = st.iupac(), int(st.mod_position)
    return r


def graph_to_rows(graph, best, bin_name, mod_type, padding):
    """find_motifs_bin.py:537-549, 1218-1223 — nodes in best, score-descending."""
    rows = [dict(reference=bin_name, motif=n.string, mod_type=mod_type, mod_position=padding,
                 model=d["model"], score=float(d["score"])) for n, d in graph.nodes.items() if n in best]
    rows.sort(key=lambda r: -r["score"])
    return [derive(r) for r in rows]


def remove_noisy_motifs(rows):
    """postprocess.py:7-25 — drop motifs with isolated bases (size 3); unchanged if all are noisy."""
    clean = [r["motif"] for r in rows if not Motif(r["motif"], r["mod_position"]).have_isolated_bases(isolation_size=3)]
    if not clean:
        return rows
    return [r for r in rows if r["motif"] in clean]


def get_motif_parental_relationship(motifs):
    """postprocess.py:41-49 — (parent, child) pairs where child.sub_string_of(parent)... named as upstream:
    returns (motif2, motif1) for every motif1.sub_string_of(motif2)."""
    rel = []
    for i, m1 in enumerate(motifs):
        for j, m2 in enumerate(motifs):
            if i != j and m1.sub_string_of(m2) and (m2, m1) not in rel:
                rel.append((m2, m1))
    return rel


def merge_motifs_in_rows(rows, pileup, contigs, low=0.3, high=0.7, merge_threshold=0.5):
    """find_motifs_bin.py:1436-1537 for ONE (bin, modtype) group (the caller groups).  Note the reference calls
    this with its default thresholds 0.3 / 0.7 whatever the CLI says (find_motifs_bin.py:569)."""
    if not rows:
        return rows
    bin_name, mod_type = rows[0]["reference"], rows[0]["mod_type"]
    motifs = [Motif(r["motif"], r["mod_position"]) for r in rows]
    merged_all, premerge_all = [], []
    for merged, cluster, pre, new in merge_motifs(motifs):
        if len(new) == 0:
            merged_all.append(merged)
            premerge_all.extend(cluster)
            continue
        merge_model = motif_model_bin(pileup, contigs, merged, BetaBernoulliModel(), low, high)
        pre_model = BetaBernoulliModel()
        for v in pre:
            pre_model = motif_model_bin(pileup, contigs, v, pre_model, low, high)
        if predictive_evaluation_score(pre_model, merge_model) < merge_threshold:
            merged_all.append(merged)
            premerge_all.extend(cluster)
    if not premerge_all:
        return rows
    pre_strings = {m.string for m in premerge_all}
    out = [r for r in rows if r["motif"] not in pre_strings]
    for m in merged_all:
        model = motif_model_bin(pileup, contigs, m, BetaBernoulliModel(), low, high)
        parents = get_parent_scores(m, pileup, contigs, low, high)
        score = float(np.mean([d["score"] for d in parents.values()])) if parents else -1
        out.append(derive(dict(reference=bin_name, motif=m.string, mod_type=mod_type,
                               mod_position=int(m.mod_position), model=model, score=score)))
    return out


def unique_rows(rows):
    """``motifs.unique()`` (find_motifs_bin.py:570, 579): all cells of a row, the ``model`` cell (an Object column) compared
    the way py-polars compares Python objects — ``__hash__`` / ``__eq__``, i.e. identity for BetaBernoulliModel.  Two clusters
    merging into the same motif carry a model each (:1497-1504): both rows stay (fixture g13)."""
    seen, out = set(), []
    for r in rows:
        k = (r["reference"], r["motif"], r["mod_type"], r["mod_position"], id(r["model"]), r["score"])
        if k not in seen:
            seen.add(k)
            out.append(r)
    return out


def remove_sub_motifs(rows):
    """postprocess.py:52-82 for one (reference, mod_type) group."""
    motifs = [Motif(r["motif"], r["mod_position"]) for r in rows]
    group = list(rows)                      # the reference looks models up in the ORIGINAL group frame
    out = list(rows)
    for parent, child in get_motif_parental_relationship(motifs):
        def model_of(m):
            return next(r["model"] for r in group if r["motif"] == m.string and r["mod_position"] == m.mod_position)
        s = predictive_evaluation_score(model_of(child), model_of(parent))
        drop = parent if s > 0.5 else child
        out = [r for r in out if not (r["motif"] == drop.string and r["mod_position"] == drop.mod_position)]
    return out


def join_motif_complements(rows):
    """postprocess.py:85-109 — left self-join on motif_iupac == revcomp(other.motif_iupac) within
    (reference, mod_type); keep rows with motif_iupac >= complement's or without complement."""
    out = []
    for r in rows:
        partners = [o for o in rows if o["reference"] == r["reference"] and o["mod_type"] == r["mod_type"]
                    and reverse_compliment(o["motif_iupac"]) == r["motif_iupac"]]
        if not partners:
            q = dict(r)
            q.update(motif_complement=None, mod_position_complement=None, n_mod_complement=None,
                     n_nomod_complement=None, motif_iupac_complement=None, mod_position_iupac_complement=None)
            out.append(q)
            continue
        for o in partners:
            if r["motif_iupac"] >= o["motif_iupac"]:
                q = dict(r)
                q.update(motif_complement=o["motif"], mod_position_complement=o["mod_position"],
                         n_mod_complement=o["n_mod"], n_nomod_complement=o["n_nomod"],
                         motif_iupac_complement=o["motif_iupac"], mod_position_iupac_complement=o["mod_position_iupac"])
                out.append(q)
    return out


def process_bin(pileup, contigs, bin_name, mod_type, graph, best, padding):
    """find_motifs_bin.py:537-596 — the chain after the search; returns rows or None."""
    rows = graph_to_rows(graph, best, bin_name, mod_type, padding)
    if not rows:
        return None
    rows = remove_noisy_motifs(rows)
    rows = unique_rows(merge_motifs_in_rows(rows, pileup, contigs))
    if not rows:
        return None
    rows = unique_rows(remove_sub_motifs(rows))
    if not rows:
        return None
    rows = join_motif_complements(rows)
    return rows or None


HEADER = ["reference", "motif", "mod_position", "mod_type", "n_mod", "n_nomod", "motif_type", "motif_complement",
          "mod_position_complement", "n_mod_complement", "n_nomod_complement"]


def format_bin_motifs(rows, min_motifs_bin=50):
    """main.py:96 + motif.py:899-926 — TSV text of bin-motifs.tsv (header only when no rows, main.py:317-321)."""
    rows = [r for r in rows if r["n_mod"] + r["n_nomod"] >= min_motifs_bin]
    rows = sorted(rows, key=lambda r: (r["reference"], r["mod_type"], r["motif_iupac"]))
    lines = ["\t".join(HEADER)]
    for r in rows:
        f = lambda v: "" if v is None else str(v)
        lines.append("\t".join([r["reference"], r["motif_iupac"], str(r["mod_position_iupac"]), r["mod_type"],
                                str(r["n_mod"]), str(r["n_nomod"]), motif_type(r["motif_iupac"]),
                                f(r.get("motif_iupac_complement")), f(r.get("mod_position_iupac_complement")),
                                f(r.get("n_mod_complement")), f(r.get("n_nomod_complement"))]))
    return "\n".join(lines) + "\n"

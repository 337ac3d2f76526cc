"""Post-processing fuzz against the REFERENCE'S OWN FUNCTIONS (build container only: imports /root/reference through refstub,
like gen_golden.py's g8): random families of motifs around the planted ones of a bin — single-base variants (degenerate
merges, accepted and rejected), flank extensions (sub-motif relations), gapped noise, reverse complements (complement
join), junk — through remove_noisy_motifs -> merge_motifs_in_df -> remove_sub_motifs -> join_motif_complements of the
reference, and the same rows through the product's Python coroutines and the native nm_post_run_rows_custom (scored by the
CPU oracle): every stage table must be equal.   usage: python3 tests/golden/post_ref_fuzz.py [first_seed [n_seeds]]"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
if os.environ.get("PYTHONHASHSEED") != "0":                      # (set order inside the reference, as in gen_golden.py)
    os.environ["PYTHONHASHSEED"] = "0"
    os.execv(sys.executable, [sys.executable] + sys.argv)
import numpy as np

import gen_golden as G
import refstub
from helpers import oracle_bin_inputs
from test_postprocess_reference_vectors import STAGES, both_implementations, oracle_scorer, table
from test_oracle_golden import assert_tables_equal

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", ".": "."}


def family(rng, planted, mts):
    """[(core, pos, mod type)] — single letters and '.' only (the stages start from search results: 41-column strings)."""
    out = []
    two = {"R": "AG", "Y": "CT", "W": "AT", "S": "CG", "K": "GT", "M": "AC"}
    base = []
    for core, pos, mt in planted:                                # planted IUPAC motifs as their literal forms (the search finds those)
        forms = [core.replace("N", ".")]
        for letter, pair in two.items():
            forms = [f.replace(letter, x, 1) for f in forms for x in (pair if letter in f else "")] or forms
        base += [(f, pos, mt) for f in forms if mt in mts and all(ch in "ACGT." for ch in f)]
    for core, pos, mt in base:
        if rng.random() < 0.8:
            out.append((core, pos, mt))
        for _ in range(int(rng.integers(0, 4))):
            kind = int(rng.integers(0, 6))
            c, p = list(core), pos
            if kind == 0:                                        # one position changed (not the modified one)
                q = int(rng.integers(len(c)))
                if q != p and c[q] != ".":
                    c[q] = "ACGT"[int(rng.integers(4))]
            elif kind == 1:                                      # a flank added
                if rng.random() < 0.5:
                    c = ["ACGT"[int(rng.integers(4))]] + c; p += 1
                else:
                    c = c + ["ACGT"[int(rng.integers(4))]]
            elif kind == 2:                                      # a gap and a far base: noise
                g = int(rng.integers(2, 6))
                if rng.random() < 0.5:
                    c = ["ACGT"[int(rng.integers(4))]] + ["."] * g + c; p += g + 1
                else:
                    c = c + ["."] * g + ["ACGT"[int(rng.integers(4))]]
            elif kind == 3 and len(c) > 3:                       # a flank removed
                if p > 0 and rng.random() < 0.5:
                    c = c[1:]; p -= 1
                elif p < len(c) - 1:
                    c = c[:-1]
            elif kind == 4:                                      # the reverse complement, when it carries the same modified base
                rc = [COMP[ch] for ch in reversed(c)]
                can = c[p]
                spots = [i for i, ch in enumerate(rc) if ch == can]
                if spots:
                    c, p = rc, spots[int(rng.integers(len(spots)))]
            else:                                                # two positions changed
                for q in rng.choice(len(c), size=min(2, len(c)), replace=False):
                    if int(q) != p and c[int(q)] != ".":
                        c[int(q)] = "ACGT"[int(rng.integers(4))]
            while c and c[0] == "." and p > 0:
                c, p = c[1:], p - 1
            while c and c[-1] == "." and p < len(c) - 1:
                c = c[:-1]
            out.append(("".join(c), p, mt))
    for mt in mts:                                               # junk
        for _ in range(int(rng.integers(0, 3))):
            n = int(rng.integers(2, 7))
            c = list(rng.choice(list("ACGT"), size=n))
            p = int(rng.integers(n))
            c[p] = "A" if mt == "a" else "C"
            out.append(("".join(c), p, mt))
    seen, uniq = set(), []
    for m in out:
        if m not in seen and len(m[0]) <= 30 and m[0][m[1]] == ("A" if m[2] == "a" else "C"):
            seen.add(m); uniq.append(m)
    return uniq


def case_of(nm, seed, cache):
    """(bin, mod types of the frame, [(core, pos, mod type)], [score]) of fuzz case ``seed`` — gen_golden.py's g13 records two of them."""
    rng = np.random.default_rng(seed)
    bin_name = list(G.POST_BINS)[int(rng.integers(len(G.POST_BINS)))]
    kw0, _ = G.POST_BINS[bin_name]
    mts = tuple(sorted(set(kw0["mod_types"]))) if rng.random() < 0.5 else (G.POST_BINS[bin_name][1],)
    if (bin_name, mts) not in cache:
        cache[(bin_name, mts)] = G._post_bin(nm, bin_name, mts)
    kw, mg, pile, seqs = cache[(bin_name, mts)]
    planted = [tuple(x) for b in mg.bin_motifs.values() for x in b]
    members = family(rng, planted, mts)
    return bin_name, mts, members, [float(np.round(1.0 + 3.0 * rng.random(), 3)) for _ in members]


def one(nm, seed, cache):
    from nanomotif.model import BetaBernoulliModel
    fmb, pl = nm.find_motifs_bin, sys.modules["polars"]
    bin_name, mts, members, scores = case_of(nm, seed, cache)
    kw, mg, pile, seqs = cache[(bin_name, mts)]
    if not members:
        return "no members"
    data = {"reference": [], "motif": [], "mod_type": [], "mod_position": [], "model": [], "score": []}
    rec_in = []
    for k, (core, pos, mt) in enumerate(members):
        m = G._wide(nm, core, pos)
        model = fmb.motif_model_bin(pile.filter(pl.col("mod_type") == mt), seqs, m, BetaBernoulliModel(), 0.3, 0.7)
        score = scores[k]
        for key, v in zip(data, ("bin0", m.string, mt, int(m.mod_position), model, score)):
            data[key].append(v)
        rec_in.append([m.string, int(m.mod_position), mt, G.model_counts(model), score])
    df = nm.motif.MotifSearchResult(pl.DataFrame(data))
    a = nm.postprocess.remove_noisy_motifs(df)
    b = fmb.merge_motifs_in_df(a, pile, seqs, {"bin0": list(seqs)}).unique()
    c = nm.postprocess.remove_sub_motifs(b).unique()
    d = nm.postprocess.join_motif_complements(c).unique()
    ref = {"noise": G._table(a), "merge": G._table(b), "sub": G._table(c), "complement": G._table(d)}
    twice = G._has_duplicate_motifs(b) or G._has_duplicate_motifs(c)      # (compared like every other family since round 5)
    piles, oseqs = {}, None
    for mt in mts:
        piles[mt], oseqs = oracle_bin_inputs(mg, mt)
    keys = [("bin0", mt) for mt in mts]
    rows = [[(m, cnt[0], cnt[1], sc) for m, pos, mt2, cnt, sc in rec_in if mt2 == mt] for mt in mts]
    py, native, _, _ = both_implementations(keys, rows, oracle_scorer(keys, piles, oseqs))
    for ours, theirs in zip(STAGES[1:], ("noise", "merge", "sub", "complement")):
        if not ref[theirs]:
            continue
        assert_tables_equal(table(py[ours]), ref[theirs], (seed, bin_name, members, theirs, "python twin"))
        assert_tables_equal(table(native[ours]), ref[theirs], (seed, bin_name, members, theirs, "nm_post_run"))
    return f"{bin_name} {mts}: {len(members)} motifs -> " + ", ".join(f"{k} {len(v)}" for k, v in ref.items()) + ("; ONE MOTIF TWICE" if twice else "")


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nm = refstub.load_reference()
    bad, cache = 0, {}
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(nm, seed, cache)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:3000]}", flush=True)
    print("post-processing fuzz against the reference done, mismatches:", bad)
    sys.exit(1 if bad else 0)

"""Seeded fuzz parity: random assemblies (N runs, lower case, other IUPAC letters, contigs around the chunk / word
edges), random pileups (rows on any base and strand, fractions on and around the thresholds) and random motifs
(literals, 2- and 3-base sets, gaps, offsets out to +-63, modified position on a set or on '.') through the C ABI
against the CPU oracle — integer counts bit-exact."""
import numpy as np
import pytest

from nanomotif_amd.motif import Motif

pytestmark = pytest.mark.gpu

SETS = ["A", "C", "G", "T", "[AC]", "[AG]", "[AT]", "[CG]", "[CT]", "[GT]", "[ACG]", "[ACT]", "[AGT]", "[CGT]", "."]


def _motif(rng, wide):
    n = int(rng.integers(1, 64 if wide else 18))
    w = np.array([0.13] * 4 + [0.02] * 6 + [0.015] * 4 + [0.3])
    pos = [SETS[int(rng.choice(len(SETS), p=w / w.sum()))] for _ in range(n)]
    if wide:                                   # mostly gaps, so that something still matches
        for i in rng.choice(n, size=max(0, n - 4), replace=False):
            pos[int(i)] = "."
    if all(p == "." for p in pos):
        pos[int(rng.integers(n))] = "ACGT"[int(rng.integers(4))]
    spec = [i for i, p in enumerate(pos) if p != "."]
    # the modified position may sit on a set or on an inner '.', but not on a flank that stripping removes
    # (Motif.new_stripped_motif would leave it outside the motif; the engine refuses that loudly)
    return "".join(pos), int(rng.integers(spec[0], spec[-1] + 1))


def _contig(rng, n):
    s = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.3])
    for _ in range(int(rng.integers(0, 4))):                         # N runs and stray IUPAC letters
        a = int(rng.integers(0, n)); s[a:a + int(rng.integers(1, 70))] = ord("N")
    for ch in b"RYKMSWN":
        if rng.random() < 0.5:
            s[rng.integers(0, n, size=max(1, n // 5000))] = ch
    txt = s.tobytes().decode()
    return txt.lower() if rng.random() < 0.2 else txt


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_fuzz_counts_match_oracle(seed):
    from nanomotif_amd.engine import ScanEngine
    from oracle.scan import ContigPileup, score_candidates
    rng = np.random.default_rng(1000 + seed)
    lens = [int(x) for x in rng.choice([1, 2, 31, 33, 100, 8191, 8192, 8193, 12_000, 16_384, 30_000, 70_000], size=9)]
    names = [f"c{i}" for i in range(len(lens))]
    seqs = [_contig(rng, n) for n in lens]
    bins = [f"b{int(rng.integers(3))}" for _ in lens]
    eng = ScanEngine(0)
    eng.upload_assembly(names, seqs, bins)
    fr_values = np.array([0.0, 0.1, 0.3, 0.30000000000000004, 0.29999999999999993, 0.5, 0.7, 0.7000000000000001, 0.6999999999999999, 0.95, 1.0])
    piles = {}
    for mt in ("a", "m"):
        piles[mt] = {}
        cid, pos, st, fr = [], [], [], []
        for i, n in enumerate(lens):
            k = int(rng.integers(0, 2 * n + 1))
            flat = rng.choice(2 * n, size=min(k, 2 * n), replace=False)          # unique (position, strand)
            p, strand = (flat // 2).astype(np.int64), np.where(flat % 2 == 0, ord("+"), ord("-")).astype(np.uint8)
            f = rng.choice(fr_values, size=len(p))
            piles[mt][names[i]] = ContigPileup(p, strand, f)
            cid.append(np.full(len(p), i, np.uint32)); pos.append(p); st.append(strand); fr.append(f)
        eng.upload_pileup(mt, np.concatenate(cid), np.concatenate(pos), np.concatenate(st), np.concatenate(fr))
    cands, spec = [], []
    for k in range(160):
        s, p = _motif(rng, wide=(k % 5 == 0))
        mt, b = ("a", "m")[int(rng.integers(2))], f"b{int(rng.integers(3))}"
        if b not in bins:
            continue
        cands.append((Motif(s, p), mt, b)); spec.append((s, p, mt, b))
    got = eng.score(cands)
    for mt in ("a", "m"):
        for b in sorted(set(bins)):
            idx = [k for k, (_, _, m2, b2) in enumerate(spec) if (m2, b2) == (mt, b)]
            if not idx:
                continue
            contigs = {names[i]: seqs[i].upper() for i in range(len(lens)) if bins[i] == b}
            exp = score_candidates({n: piles[mt][n] for n in contigs}, contigs, [(spec[k][0], spec[k][1]) for k in idx])
            assert np.array_equal(got[idx], exp), (seed, mt, b, [spec[k] for k in idx if got[k].tolist() != exp[idx.index(k)].tolist()][:3])
    # the hit POSITIONS of some of them on one contig of their bin (nm_hit_positions against motif_model_contig(save_motif_positions=True),
    # find_motifs_bin.py:1322-1329) and the per-contig counters (one row per contig of the bin) — the same numbers, three ways
    from oracle.model import BetaBernoulliModel
    from oracle.motif import Motif as OracleMotif
    from oracle.scan import motif_model_contig
    keys = ("index_meth_fwd", "index_nonmeth_fwd", "index_meth_rev", "index_nonmeth_rev")
    for k in rng.choice(len(spec), size=min(24, len(spec)), replace=False):
        s, p, mt, b = spec[int(k)]
        members = [i for i in range(len(lens)) if bins[i] == b]
        (ids, table), = eng.score_per_contig([cands[int(k)]])
        assert np.array_equal(table.sum(axis=0), got[int(k)])
        i = members[int(rng.integers(len(members)))]
        _, want = motif_model_contig(piles[mt][names[i]], seqs[i].upper(), BetaBernoulliModel(), OracleMotif(s, p), save_motif_positions=True)
        n_mod = n_non = 0
        for which, key in enumerate(keys):
            hits = eng.hit_positions(names[i], mt, Motif(s, p), which)
            assert hits.tolist() == sorted(want[key].tolist()), (seed, s, p, mt, names[i], key)      # (the oracle keeps the pileup's row order: random here)
            n_mod, n_non = n_mod + (len(hits) if which % 2 == 0 else 0), n_non + (len(hits) if which % 2 else 0)
        row = [r for r, c in enumerate(ids) if c == names[i]]
        assert len(row) == 1 and table[row[0]].tolist() == [n_mod, n_non], (seed, s, p, mt, names[i])
    eng.close()


def _variants(rng, can, style):
    """A (bin, mod type) group of a LIGHT batch: candidates that share most constraints, the shapes the common-constraint
    factoring and the sibling descriptors of score_kernel see (find_motifs_bin.py:1116-1135, 1408-1432)."""
    W = 41
    core = ["."] * W
    core[20] = can
    alphabet = ["A", "C", "G", "T"] if style != "sets" else ["A", "C", "G", "T", "[AG]", "[CT]", "[ACG]", "[GT]"]
    span = range(0, W) if style == "wide" else range(10, 31)
    spots = [i for i in span if i != 20]
    for q in rng.choice(spots, size=int(rng.integers(0, 5)), replace=False):
        core[int(q)] = alphabet[int(rng.integers(len(alphabet)))]
    free = [i for i in spots if core[i] == "."]
    n = int(rng.integers(1, 9))
    out = []
    kind = rng.choice(["siblings", "two_extra", "mixed", "parents"])
    for k in range(n):
        c = list(core)
        if kind == "siblings":                    # one more literal at ONE shared position
            c[free[0]] = "ATGC"[k % 4] if k < 4 else c[free[0]]
        elif kind == "two_extra":
            for q in rng.choice(free, size=2, replace=False):
                c[int(q)] = alphabet[int(rng.integers(len(alphabet)))]
        elif kind == "mixed":                     # the parent itself, a child, a grandchild, a duplicate
            for q in rng.choice(free, size=int(rng.integers(0, 3)), replace=False):
                c[int(q)] = "ACGT"[int(rng.integers(4))]
        else:                                     # pruning round: the motif and its parents (one position blanked each)
            spec = [i for i in range(W) if c[i] != "." and i != 20]
            if k and spec:
                c[spec[(k - 1) % len(spec)]] = "."
        out.append(("".join(c), 20))
    return out


@pytest.mark.parametrize("seed,style", [(0, "literal"), (1, "literal"), (2, "sets"), (3, "wide"), (4, "general"), (5, "literal")])
def test_fuzz_light_batches_common_factoring_and_siblings(seed, style):
    """Light batches (<= 6 candidates per group on average): literal-only tiles, common-constraint factoring, sibling
    descriptors, fused two-slot launches — against the oracle, with motifs that share their parent."""
    from nanomotif_amd.engine import ScanEngine
    from oracle.scan import ContigPileup, score_candidates
    rng = np.random.default_rng(7000 + seed)
    lens = [int(x) for x in rng.choice([40, 100, 8191, 8192, 8193, 20_000, 16_384, 50_000, 130_000], size=10)]
    names = [f"c{i}" for i in range(len(lens))]
    seqs = [_contig(rng, n) for n in lens]
    bins = [f"b{int(rng.integers(5))}" for _ in lens]
    eng = ScanEngine(0)
    eng.upload_assembly(names, seqs, bins)
    fr_values = np.array([0.0, 0.1, 0.3, 0.5, 0.7, 0.95, 1.0])
    piles = {}
    mods = ("a", "m") if seed != 5 else ("a",)                      # seed 5: one slot (no fusion)
    for mt in mods:
        piles[mt] = {}
        cid, pos, st, fr = [], [], [], []
        for i, n in enumerate(lens):
            flat = rng.choice(2 * n, size=int(rng.integers(0, 2 * n + 1)), replace=False)
            p, strand = (flat // 2).astype(np.int64), np.where(flat % 2 == 0, ord("+"), ord("-")).astype(np.uint8)
            f = rng.choice(fr_values, size=len(p))
            piles[mt][names[i]] = ContigPileup(p, strand, f)
            cid.append(np.full(len(p), i, np.uint32)); pos.append(p); st.append(strand); fr.append(f)
        eng.upload_pileup(mt, np.concatenate(cid), np.concatenate(pos), np.concatenate(st), np.concatenate(fr))
    for rnd in range(6):
        cands, spec = [], []
        for b in sorted(set(bins)):
            for mt in mods:
                if rng.random() < 0.25:
                    continue                                           # a search that is not asking this round
                can = {"a": "A", "m": "C"}[mt] if style != "general" else "ACGT"[int(rng.integers(4))]
                for s, p in _variants(rng, can, style):
                    cands.append((Motif(s, p), mt, b)); spec.append((s, p, mt, b))
        if not cands:
            continue
        got = eng.score(cands)
        for mt in mods:
            for b in sorted(set(bins)):
                idx = [k for k, (_, _, m2, b2) in enumerate(spec) if (m2, b2) == (mt, b)]
                if not idx:
                    continue
                contigs = {names[i]: seqs[i].upper() for i in range(len(lens)) if bins[i] == b}
                exp = score_candidates({n: piles[mt][n] for n in contigs}, contigs, [(spec[k][0], spec[k][1]) for k in idx])
                assert np.array_equal(got[idx], exp), (seed, style, rnd, mt, b, [spec[k] for k in idx if got[k].tolist() != exp[idx.index(k)].tolist()][:3])
    eng.close()

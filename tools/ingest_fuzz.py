"""Fuzz of the device pre-filters (nm_ingest_pileup: coverage, per-(contig, mod code) frequency, adjacency; classification) ON
THE GPU BOX against oracle/pileup.py on random adversarial tables: fractions on the thresholds and tied inside adjacency
windows, coverage on the bound, groups on the frequency bounds (50 / 51 modified rows, ratio near 1e-4), NaN fractions, up to
14 mod codes, rows in modkit's order or shuffled (whole table / contig runs interleaved), contigs missing from the engine.
usage: python3 tools/ingest_fuzz.py [first_seed [n_seeds]]"""
import sys
import time

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np

from nanomotif_amd.engine import ScanEngine
from oracle import pileup as op


def one(seed):
    rng = np.random.default_rng(seed)
    n_contigs = int(rng.integers(1, 7))
    lens = rng.integers(3_000, 60_000, n_contigs)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(L))) for L in lens]
    n_codes = int(rng.choice([2, 2, 3, 5, 14]))
    cols = {k: [] for k in ("contig", "position", "strand", "mod_type", "fraction_mod", "Nvalid_cov")}
    for c in range(n_contigs):
        L = int(lens[c])
        for code in range(n_codes):
            if rng.random() < 0.15:
                continue
            density = float(rng.choice([0.02, 0.2, 0.5]))
            for strand in (ord("+"), ord("-")):
                pos = np.flatnonzero(rng.random(L) < density).astype(np.int64)
                n = len(pos)
                if n == 0:
                    continue
                style = int(rng.integers(0, 4))
                if style == 0:      # mostly unmethylated, a few confident rows: the frequency filter's bounds
                    frac = rng.choice([0.0, 0.05, 0.2], size=n)
                    k = int(rng.choice([0, 25, 26, 50, 51, 60])) // 2      # per strand: a group ends at 50 / 51 / 52 modified rows
                    if k and n > k:
                        frac[rng.choice(n, size=k, replace=False)] = rng.choice([0.7, 0.9, 1.0], size=k)
                elif style == 1:    # dense ties on a few levels: adjacency windows full of equal maxima
                    frac = rng.choice([0.0, 0.3, 0.69, 0.7, 0.7000000000000001, 0.75, 0.8, 0.8, 0.95, 1.0], size=n)
                elif style == 2:    # continuous values
                    frac = np.round(rng.random(n) * 100, 2) / 100
                else:               # bimodal like real data
                    frac = np.where(rng.random(n) < 0.1, rng.choice([0.85, 0.9, 0.97, 1.0], size=n), rng.choice([0.0, 0.02, 0.1], size=n))
                if rng.random() < 0.3:
                    frac[rng.choice(n, size=max(1, n // 100), replace=False)] = np.nan       # null percentages count as rows, never as modified
                cov = rng.choice([4, 5, 6, 7, 20, 100], size=n, p=[0.03, 0.07, 0.1, 0.1, 0.5, 0.2])
                cols["contig"].append(np.full(n, c, np.int64)); cols["position"].append(pos); cols["strand"].append(np.full(n, strand, np.uint8))
                cols["mod_type"].append(np.full(n, code, np.int8)); cols["fraction_mod"].append(frac); cols["Nvalid_cov"].append(cov.astype(np.int64))
    if not cols["position"]:
        return "empty"
    t = {k: np.concatenate(v) for k, v in cols.items()}
    # modkit's order: contig, position, then strand / code interleaved
    order = np.lexsort((t["mod_type"], t["strand"], t["position"], t["contig"]))
    how = int(rng.integers(0, 3))
    if how == 1:
        order = rng.permutation(len(order))                                   # anything goes
    elif how == 2 and n_contigs > 1:                                          # contig runs in another order, one contig split in two runs
        runs = [order[t["contig"][order] == c] for c in rng.permutation(n_contigs)]
        a = runs[0]
        runs = [a[:len(a) // 2]] + runs[1:] + [a[len(a) // 2:]]
        order = np.concatenate(runs)
    t = {k: v[order] for k, v in t.items()}
    exp = op.prefilter({k: v.copy() for k, v in t.items()})
    eng = ScanEngine(0)
    try:
        absent = set(rng.choice(n_contigs, size=int(rng.integers(0, max(1, n_contigs // 2))), replace=False).tolist()) if n_contigs > 2 else set()
        mine = [c for c in range(n_contigs) if c not in absent]
        eng.upload_assembly([f"c{c}" for c in mine], [seqs[c] for c in mine], ["b"] * len(mine))
        lut = np.full(n_contigs, 0xFFFFFFFF, dtype=np.uint32)
        lut[mine] = np.arange(len(mine), dtype=np.uint32)
        res = eng.ingest_pileup(lut[t["contig"]], t["position"], t["mod_type"], t["strand"], t["fraction_mod"], t["Nvalid_cov"], {0: ("m", "C"), 1: ("a", "A")})
        keep = np.isin(exp["contig"], mine)
        kept = np.zeros((n_contigs, 8), dtype=np.int64)
        sel = keep & (exp["mod_type"] < 8)
        np.add.at(kept, (exp["contig"][sel], exp["mod_type"][sel]), 1)
        assert np.array_equal(res["kept"].astype(np.int64), kept[mine]), (seed, "kept per (contig, mod)", res["kept"].tolist(), kept[mine].tolist())
        assert res["n_kept"] == int(keep.sum()), (seed, "n_kept", res["n_kept"], int(keep.sum()))
        conf = keep & (exp["fraction_mod"] >= 0.7) & (exp["mod_type"] < 2)
        want = sorted(zip(lut[exp["contig"][conf]].tolist(), exp["position"][conf].tolist(), exp["strand"][conf].tolist(), exp["mod_type"][conf].tolist()))
        cc, cp, cs, cm = res["confident"]
        assert sorted(zip(cc.tolist(), cp.tolist(), cs.tolist(), cm.tolist())) == want, (seed, "confident rows", len(cc), len(want))
    finally:
        eng.close()
    return f"{len(t['position'])} rows, {n_contigs} contigs ({len(mine)} resident), {n_codes} codes, order {('modkit', 'shuffled', 'runs moved')[how]}: kept {int(keep.sum())}, confident {len(want)}"


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    bad = 0
    for seed in range(first, first + n):
        t0 = time.time()
        try:
            print(f"seed {seed}: {one(seed)} ({time.time() - t0:.1f} s)", flush=True)
        except AssertionError as e:
            bad += 1
            print(f"seed {seed}: MISMATCH {str(e)[:1200]}", flush=True)
    print("ingest fuzz done, mismatches:", bad)
    sys.exit(1 if bad else 0)

"""The native post-processing (csrc/nmpost.cpp: nm_post_*) on CPU against its Python twin (nanomotif_amd/postprocess.py,
which the oracle pipeline and the reference's known-answer tests pin): every stage table of every task — noise, clique
merge with its two scoring batches, sub-motif removal, complement join — on randomised motif families built to trigger
each rule, and on the best candidates of real searches."""
import random
import zlib

import numpy as np
import pytest

from nanomotif_amd import native_search as ns
from nanomotif_amd import postprocess as pp
from nanomotif_amd import search as ps
from nanomotif_amd.model import BetaBernoulliModel
from nanomotif_amd.motif import Motif, reverse_compliment

PAD = 20
W = 2 * PAD + 1


def hash_scorer(salt=0, hi_mod=3000, hi_non=600):
    """Deterministic counts from (task, motif, mod position): every implementation asks for the same motifs and must get
    the same answers, whatever the batch they arrive in."""
    def score(reqs):
        out = np.zeros((len(reqs), 2), dtype=np.int64)
        for i, (t, m) in enumerate(reqs):
            st = m.new_stripped_motif()
            h = zlib.crc32(f"{salt}|{t}|{st.string}|{st.mod_position}".encode())
            out[i] = (h % hi_mod, (h >> 12) % hi_non)
        return out
    return score


def row_tuple(r):
    c = r.complement
    return (r.motif, r.mod_position, r.n_mod, r.n_nomod, r.score, None if c is None else (c.motif, c.mod_position, c.n_mod, c.n_nomod, c.score),
            r.motif_iupac, r.mod_position_iupac, r.has_complement_columns)


def python_post(keys, rows_per_task, score_fn):
    stages = {k: {} for k in keys}
    tasks = {}
    for key, rows in zip(keys, rows_per_task):
        g, best = ps.MotifTree(), []
        for s, n_mod, n_nomod, sc in rows:
            m = Motif(s, PAD)
            g.add_node(m, model=BetaBernoulliModel.from_counts(n_mod, n_nomod), score=sc)
            best.append(m)
        tasks[key] = pp.postprocess_co(g, best, key[0], key[1], PAD, on_stage=lambda name, r, key=key: stages[key].__setitem__(name, list(r)))

    def scorer(flat):
        assert all(tag == "merge" for _, _, tag in flat)
        return score_fn([(keys.index(k), m) for k, m, _ in flat])
    final = ps.run_lockstep(tasks, scorer)
    return stages, final


def compare(keys, rows_per_task, score_fn):
    stages, final = python_post(keys, rows_per_task, score_fn)
    post = ns.postprocess_rows_custom(keys, rows_per_task, PAD, score_fn, tables=True)
    n_rows = 0
    for t, key in enumerate(keys):
        assert post.n_stages(t) == len(stages[key]), (key, post.n_stages(t), list(stages[key]))
        # the tables as text (nm_post_tables): what format_motifs writes of the same rows, byte for byte; a stage without rows: its header
        for s in range(5):
            assert post.table_text(t, s) == pp.format_motifs(post.rows(t, s)), (key, s)
        for s, name in enumerate(ns.PostResults.STAGES):
            want = [row_tuple(r) for r in stages[key].get(name, [])]
            got = [row_tuple(r) for r in post.rows(t, s)]
            if s < 2:
                assert got == want, (key, name)
            else:       # accepted clusters arrive in clique order, which in the Python twin follows the hash seed
                assert sorted(got, key=repr) == sorted(want, key=repr), (key, name)
            n_rows += len(got)
        want_final = final[key]
        got_final = post.final(t)
        assert (want_final is None) == (got_final is None)
        if want_final is not None:
            assert pp.format_bin_motifs(sorted(got_final, key=row_tuple)) == pp.format_bin_motifs(sorted(want_final, key=row_tuple))
    return n_rows, post


def place(core, mod_at):
    """A search-window motif: ``core`` (letters and dots) laid so that its position ``mod_at`` is the window centre."""
    s = ["."] * W
    for i, ch in enumerate(core):
        p = PAD - mod_at + i
        assert 0 <= p < W
        s[p] = ch
    return "".join(s)


def family(rng, canonical):
    """Motifs around one seed: point variants (distance 1-2: the clique merge), an extended form and a shortened form
    (sub-motif removal), the reverse complement when it keeps the canonical base central (complement join), a gapped
    form with isolated bases (noise)."""
    n = rng.randint(5, 9)
    core = [rng.choice("ACGT") for _ in range(n)]
    mod_at = rng.randrange(n)
    core[mod_at] = canonical
    out = [("".join(core), mod_at)]
    for _ in range(rng.randint(0, 4)):
        v = list(core)
        for _ in range(rng.randint(1, 2)):
            i = rng.randrange(n)
            if i != mod_at:
                v[i] = rng.choice("ACGT.")
        out.append(("".join(v), mod_at))
    if rng.random() < 0.6:
        out.append(("".join(core) + rng.choice("ACGT"), mod_at))
    if rng.random() < 0.6 and n > 5:
        out.append(("".join(core[:-1]), mod_at) if mod_at < n - 1 else ("".join(core[1:]), mod_at - 1))
    if rng.random() < 0.7:
        rc = reverse_compliment("".join(core))
        for i, ch in enumerate(rc):
            if ch == canonical and rng.random() < 0.7:
                out.append((rc, i))
                break
    if rng.random() < 0.5:
        out.append(("".join(core[:mod_at + 1]) + "...." + rng.choice("ACGT") + "...." + rng.choice("ACGT"), mod_at))
    return out


def random_tasks(seed, n_tasks):
    rng = random.Random(seed)
    keys, rows = [], []
    for t in range(n_tasks):
        mt = rng.choice(["a", "m"])
        canonical = "A" if mt == "a" else "C"
        keys.append((f"bin{t}", mt))
        seen, task_rows = set(), []
        for _ in range(rng.randint(0, 3)):
            for core, mod_at in family(rng, canonical):
                core = core.strip(".") if core[mod_at] != "." else core
                s = place(core, mod_at) if core[0] != "." else None
                if s is None or s in seen or s[PAD] != canonical:
                    continue
                seen.add(s)
                score = rng.choice([1.6, 2.0, 2.5, round(rng.uniform(0.5, 9.0), 3)])      # ties on purpose: the sort is stable
                task_rows.append((s, rng.randrange(0, 4000), rng.randrange(0, 800), score))
        rng.shuffle(task_rows)
        rows.append(task_rows)
    return keys, rows


@pytest.mark.parametrize("seed", range(12))
def test_native_post_equals_python_twin_on_random_families(seed):
    keys, rows = random_tasks(seed, 40)
    total, post = compare(keys, rows, hash_scorer(seed))
    assert total > 100
    assert post.batches <= 2


def test_native_post_covers_every_rule():
    """Across the seeds above every rule fires: rows dropped as noise, merged rows (a bracket in the motif), rows removed
    as sub-motifs, joined complements."""
    noise = merged = sub = joined = 0
    for seed in range(12):
        keys, rows = random_tasks(seed, 40)
        post = ns.postprocess_rows_custom(keys, rows, PAD, hash_scorer(seed))
        for t in range(len(keys)):
            noise += len(post.rows(t, 0)) > len(post.rows(t, 1)) > 0
            merged += any("[" in r.motif for r in post.rows(t, 2))
            sub += len(post.rows(t, 2)) > len(post.rows(t, 3)) > 0
            joined += any(r.complement is not None for r in post.rows(t, 4))
    assert min(noise, merged, sub, joined) >= 5, (noise, merged, sub, joined)


def test_native_post_on_the_best_candidates_of_real_searches():
    """Three bins searched natively on the oracle's scan; their post-processing through the Python coroutines and
    through nm_post_run_custom, both scored by the oracle."""
    from helpers import load_golden, oracle_bin_inputs, spec_from_json
    from nanomotif_amd import synth
    from test_host_search import windows_for
    from test_native_search import _backends
    g4 = load_golden("g4_search.json")
    keys, piles, seqs_by_bin, wins = [], {}, {}, {}
    for bin_name, gname in (("binA", "geobacillus_like"), ("binB", "ecoli_like_m"), ("binC", "ecoli_like_a"), ("binD", "no_motif")):
        g = g4[gname]
        mg = synth.make_metagenome(spec_from_json(g["spec"]))
        mt = g["mod_type"]
        pile, seqs = oracle_bin_inputs(mg, mt)
        key = (bin_name, mt)
        keys.append(key)
        piles[key], seqs_by_bin[bin_name] = pile, seqs
        random.seed(1)
        wins[key] = windows_for(mg, mt, pile)
    store = ps.HostWindowStore()
    for key in keys:
        store.add_task(key, wins[key][0].copy())
    score_fn, window_fn = _backends(keys, piles, seqs_by_bin, store)
    res = ns.find_best_candidates_custom([(k, store.totals[k], wins[k][1]) for k in keys], 20, 0.05, 1.5, score_fn, window_fn)
    tasks, stages = {}, {k: {} for k in keys}
    for t, key in enumerate(keys):
        r = res.result(t)
        if r is not None:
            tasks[key] = pp.postprocess_co(r[0], r[1], key[0], key[1], PAD, on_stage=lambda name, rows, key=key: stages[key].__setitem__(name, list(rows)))
    want = ps.run_lockstep(tasks, lambda flat: score_fn([(keys.index(k), m) for k, m, _ in flat]))
    post = res.postprocess_custom(score_fn)
    res.close()
    found = 0
    for t, key in enumerate(keys):
        for s, name in enumerate(ns.PostResults.STAGES):
            assert sorted((row_tuple(r) for r in post.rows(t, s)), key=repr) == sorted((row_tuple(r) for r in stages[key].get(name, [])), key=repr), (key, name)
        if want.get(key):
            assert pp.format_bin_motifs(post.final(t)) == pp.format_bin_motifs(want[key])
            found += 1
        else:
            assert post.final(t) is None
    assert found >= 3


def test_native_post_on_threads_equals_one_thread_and_the_python_twin(monkeypatch):
    """run_post spreads the tasks' host work over threads from 64 tasks on (nmpost.cpp: contiguous ranges of tasks, request lists joined
    in task order): 300 random tasks on 1 / 3 / 8 threads write the same stage tables, and they are the Python twin's."""
    keys, rows = random_tasks(77, 300)

    def dump(post):
        return [[(r.motif, r.mod_position, r.n_mod, r.n_nomod, r.score, None if r.complement is None else r.complement.motif)
                 for r in post.rows(t, s)] for t in range(len(keys)) for s in range(5)]
    tables = {}
    for n_thr in ("1", "3", "8"):
        monkeypatch.setenv("NM_POST_THREADS", n_thr)
        post = ns.postprocess_rows_custom(keys, rows, PAD, hash_scorer(77))
        tables[n_thr] = (dump(post), post.batches, post.candidates)
    assert tables["1"] == tables["3"] == tables["8"]
    assert tables["1"][2] > 50                          # (merge candidates were scored: the joined request lists matter)
    monkeypatch.delenv("NM_POST_THREADS")
    total, post = compare(keys, rows, hash_scorer(77))  # default thread count against the Python twin, stage by stage
    assert total > 500


def test_table_text_writes_floats_like_repr():
    """nm_post_tables formats the score column like Python's repr(float): shortest digits that round-trip, fixed notation for decimal
    exponents -4 .. 15, d.ddde+XX otherwise (csrc/nmpost.cpp: append_py_repr) — on scores of every magnitude, and a bin name that is not ASCII."""
    rng = random.Random(5)
    scores = [0.0, 1.0, 5.0, 0.1, 1e-5, 9.999999999999999e-5, 1e-4, 0.00012345, 1.5e-7, 1e15, 1e16, 123456789012345.6, 1234567890123456.7,
              1.2345678901234568e+17, 2.5e22, 1e100, 5e-324, 1.7976931348623157e308, 1 / 3.0, 2 / 3.0, 12345.678, 100.0, 3.141592653589793]
    scores += [rng.random() * 10 ** rng.randint(-12, 20) for _ in range(300)] + [float(np.float32(rng.random())) for _ in range(50)]
    cores = ["GATC", "CCAGG", "GAAGA", "ACCCA", "GGCAT", "CTGAA", "TTAAC", "AGGCA"]
    keys, rows = [], []
    for t in range(0, len(scores), 8):
        keys.append((f"bin_ü{t}", "a"))
        task_rows = []
        for k, sc in enumerate(scores[t:t + 8]):
            core = cores[k]
            at = core.index("A")
            s = "." * (PAD - at) + core + "." * (W - (PAD - at) - len(core))
            task_rows.append((s, 500 + k, 20 + k, sc))
        rows.append(task_rows)
    post = ns.postprocess_rows_custom(keys, rows, PAD, hash_scorer(3), tables=True)
    seen = 0
    for t in range(len(keys)):
        text = post.table_text(t, 0)
        assert text == pp.format_motifs(post.rows(t, 0))
        for line, r in zip(text.splitlines()[1:], sorted(post.rows(t, 0), key=lambda r: r.motif)):
            assert line.split("\t")[4] == repr(r.score) and line.startswith(f"bin_ü{t * 8}\t")
            seen += 1
    assert seen == len(scores)
    assert ns.postprocess_rows_custom(keys, rows, PAD, hash_scorer(3)).table_text(0, 0) is None

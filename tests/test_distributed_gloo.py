"""Multi-rank path on CPU: contigs sharded over 2 ranks (gloo), per-rank count tables all-reduced every lock-step
round, identical decisions on every rank.  The local scoring backend is the CPU oracle restricted to the rank's
contigs (there is no GPU in the build container); production uses the HIP engine + RCCL through the same code."""
import os
import random
import socket

import numpy as np
import pytest

from helpers import load_golden, oracle_bin_inputs, spec_from_json
from nanomotif_amd import synth
from nanomotif_amd.shard import assign_contigs


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    from nanomotif_amd import postprocess as ppp
    from nanomotif_amd import search as ps
    from nanomotif_amd.find_motifs_bin import LockstepScorer
    from test_host_search import oracle_backend, windows_for
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("g4_search.json")["ecoli_like_a"]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    tasks, piles, seqs_by_bin = {}, {}, {}
    store = ps.HostWindowStore()
    shard = set(assign_contigs(mg.lengths, world)[rank].tolist())
    for mt in ("a", "m"):
        pile, seqs = oracle_bin_inputs(mg, mt)
        random.seed(1)
        windows = windows_for(mg, mt, pile)            # every rank sees the whole pileup for the windows
        key = ("bin0", mt)
        mine = [mg.names[i] for i in sorted(shard)]
        piles[key] = {n: pile[n] for n in mine}        # ... but scores only its own contigs
        seqs_by_bin["bin0"] = {n: seqs[n] for n in mine}

        store.add_task(key, windows[0])

        def chain(mt=mt, windows=windows):
            graph, best, _ = yield from ps.find_best_candidates_co(windows[1], mt, 20, min_kl=0.05, score_threshold=1.5)
            return (yield from ppp.postprocess_co(graph, best, "bin0", mt, 20))
        tasks[key] = chain()
    scorer = LockstepScorer(oracle_backend(piles, seqs_by_bin), use_dist=world > 1)
    res = ps.run_lockstep(tasks, scorer, store.execute)
    rows = [r for k in tasks for r in (res[k] or [])]
    q.put((rank, ppp.format_bin_motifs(rows), scorer.rounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_sharded_search_equals_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2, 4):                  # 3 contigs: with 4 ranks one rank holds NO contig and must still take part
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=800) for _ in range(world)]
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        out[world] = sorted(got)
    single = out[1][0][1]
    assert "GATC" in single and "CCWGG" in single
    for world in (2, 4):
        assert all(text == single for _, text, _ in out[world])    # every rank, same answer as one rank
        assert {r for _, _, r in out[world]} == {out[1][0][2]}     # same number of lock-step rounds


def _native_worker(rank, world, port, q):
    """The NATIVE lock-step machine (nm_search_run_custom + nm_post_run_custom, csrc/nmsearch.cpp / nmpost.cpp) on every rank:
    the scoring callback counts on the rank's contigs and sums the table over the ranks (what nm_search_run's reduce callback /
    nm_allreduce_counts does on the GPUs), so every rank's machine takes the same decisions."""
    import torch
    import torch.distributed as dist
    from nanomotif_amd import native_search as ns
    from nanomotif_amd import postprocess as ppp
    from nanomotif_amd import search as ps
    from oracle.scan import score_candidates
    from test_host_search import windows_for
    from test_native_search import _backends
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("g4_search.json")["ecoli_like_a"]
    mg = synth.make_metagenome(spec_from_json(g["spec"]))
    shard = set(assign_contigs(mg.lengths, world)[rank].tolist())
    mine = [mg.names[i] for i in sorted(shard)]
    keys, piles, seqs_by_bin, wins = [], {}, {}, {}
    store = ps.HostWindowStore()
    for mt in ("a", "m"):
        pile, seqs = oracle_bin_inputs(mg, mt)
        random.seed(1)
        key = ("bin0", mt)
        keys.append(key)
        wins[key] = windows_for(mg, mt, pile)
        piles[key] = {n: pile[n] for n in mine}
        seqs_by_bin["bin0"] = {n: seqs[n] for n in mine}
        store.add_task(key, wins[key][0])
    local_score, window_fn = _backends(keys, piles, seqs_by_bin, store)
    calls = [0]

    def score_fn(reqs):
        calls[0] += 1
        t = torch.from_numpy(np.ascontiguousarray(local_score(reqs)) if mine else np.zeros((len(reqs), 2), np.int64))
        if world > 1:
            dist.all_reduce(t)
        return t.numpy()
    res = ns.find_best_candidates_custom([(k, store.totals[k], wins[k][1]) for k in keys], 20, 0.05, 1.5, score_fn, window_fn)
    post = res.postprocess_custom(score_fn)
    rows = [r for t in range(len(keys)) for r in (post.final(t) or [])]
    q.put((rank, ppp.format_bin_motifs(rows), res.rounds, calls[0]))
    res.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_native_search_and_postprocessing_sharded_over_ranks_equal_single_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2, 4):                  # 3 contigs: with 4 ranks one rank scores nothing and still takes part in every sum
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_native_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        got = [q.get(timeout=800) for _ in range(world)]
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        out[world] = sorted(got)
    single = out[1][0][1]
    assert "GATC" in single and "CCWGG" in single
    for world in (2, 4):
        assert all(text == single for _, text, _, _ in out[world])
        assert {(r, c) for _, _, r, c in out[world]} == {out[1][0][2:]}      # same rounds, same number of collective calls on every rank


def test_contig_assignment_is_balanced_and_complete():
    lengths = synth.make_metagenome(synth.SynthSpec(n_contigs=1000, total_bp=100_000_000, n_bins=50, seed=1)).lengths
    for world in (1, 2, 4, 8):
        parts = assign_contigs(lengths, world)
        allc = np.sort(np.concatenate(parts))
        assert np.array_equal(allc, np.arange(len(lengths)))
        loads = np.array([lengths[p].sum() for p in parts])
        assert loads.max() / loads.mean() < 1.01
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=1000, total_bp=100_000_000, n_bins=50, seed=1))
    for world in (2, 4, 8):
        parts = assign_contigs(mg.lengths, world, bins=mg.bin_names)
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(1000))
        loads = np.array([mg.lengths[p].sum() for p in parts])
        assert loads.max() / loads.mean() < 1.05
        # bins stay mostly together: a rank holds contigs of far fewer bins than there are
        if world == 2:
            owner = {}
            for r, p in enumerate(parts):
                for i in p:
                    owner.setdefault(mg.bin_names[i], set()).add(r)
            assert sum(len(v) == 1 for v in owner.values()) >= 45
    # one huge bin must still be split
    parts = assign_contigs(mg.lengths, 4, bins=["only"] * 1000)
    assert min(len(p) for p in parts) > 100


def test_assign_bins_balances_or_declines():
    from nanomotif_amd.shard import assign_bins
    sizes = {f"bin{i}": 1_000_000 + 1000 * i for i in range(40)}
    parts = assign_bins(sizes, 4)
    assert parts is not None and sorted(b for p in parts for b in p) == sorted(sizes)
    loads = [sum(sizes[b] for b in p) for p in parts]
    assert max(loads) <= 1.15 * (sum(loads) / 4)
    for p in parts:                                             # bins keep their input order inside a rank
        assert p == [b for b in sizes if b in set(p)]
    assert assign_bins({"big": 10_000_000, "small": 100_000}, 2) is None          # contigs must be the unit then
    assert assign_bins({"big": 10_000_000, "small": 100_000}, 2, tolerance=float("inf")) == [["big"], ["small"]]
    assert assign_bins(sizes, 1) == [list(sizes)]


def _gather_worker(rank, world, port, out_dir, q):
    """Whole-bin sharding: every rank brings the motif rows of ITS bins; main._gather_rows puts them back into the
    reference's bin order, applies the bin-level filter and lets rank 0 write bin-motifs.tsv."""
    import types
    import torch.distributed as dist
    from nanomotif_amd import main as nm_main
    from nanomotif_amd.model import BetaBernoulliModel
    from nanomotif_amd.postprocess import MotifRow
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def row(bin_name, motif, pos, n_mod, n_nomod):
        m = BetaBernoulliModel()
        m.update(n_mod, n_nomod)
        return MotifRow(bin_name, motif, "a", pos, m, 2.0)
    mine = {0: [row("bin_b", "GATC", 1, 500, 10), row("bin_d", "CCWGG", 1, 30, 2)],       # bin_d: below --min_motifs_bin
            1: [row("bin_a", "ACCCA", 4, 300, 40), row("bin_c", "GAAG", 1, 90, 5)]}[rank]
    args = types.SimpleNamespace(min_motifs_bin=50, out=out_dir)
    rows = nm_main._gather_rows(args, mine, rank, world, ["bin_a", "bin_b", "bin_c", "bin_d"])
    q.put((rank, [(r.reference, r.motif) for r in rows]))
    dist.barrier()
    dist.destroy_process_group()


def test_bin_sharded_rows_are_gathered_in_bin_order(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q, port = ctx.Queue(), _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [("bin_a", "ACCCA"), ("bin_b", "GATC"), ("bin_c", "GAAG")]
    assert got[0] == want and got[1] == want                       # identical on every rank, bin_d filtered out
    lines = open(tmp_path / "bin-motifs.tsv").read().strip().split("\n")
    assert [l.split("\t")[0] for l in lines[1:]] == ["bin_a", "bin_b", "bin_c"]

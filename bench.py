#!/usr/bin/env python3
"""Benchmark of the motif-scoring hot path (BASELINE.json metric: motif-sites scored / second on the
1 Gbp synthetic metagenome).

A *step* = one pass of the hot path over one batch: the cfg 5 candidate table (10 000 seeded IUPAC motifs,
20 per bin, half 6mA half 5mC) scored against every contig of its bin on both strands through
``nm_score_batch_device`` — compile the candidates on the host, ship the programs, one scoring launch, count
table in HBM — and, with more than one GPU, one RCCL all-reduce (sum, int64) of the count table.
A *motif-site* = one (candidate x reference bp x strand) match test: a candidate on a bin of L bp is 2·L sites.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--total-bp B] [--workload cfg5|greedy] [--scaling weak|strong]

Multi-GPU: launched by torch.distributed.run, one rank per GPU.  Bins are independent searches, so the default
("weak" scaling) gives every GPU whole bins — its own cfg 5 metagenome of --total-bp (rank r is seeded 1 + r) and
that metagenome's candidate table — with no collective on the data path, the way ``python -m nanomotif_amd`` shards
a metagenome whose bins balance (nanomotif_amd/shard.py: assign_bins); value = motif-sites of all ranks / the
slowest rank's time.  ``--scaling strong`` keeps ONE --total-bp metagenome, shards the contigs of every bin over the
ranks (longest-first) and sums the count tables with one RCCL all-reduce per step — the path for few / huge bins.
Inputs are generated on the device (nanomotif_amd/synth_device.py) and are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured)
VALU_PEAK_WAVE_INSTR_S = 6.1e11   # measured integer-VALU issue peak (tools/valu_peak.hip, profiles/r1/valu_peak.txt)
ALGO_BYTES_PER_BP_STEP = 0.5   # 2-bit sequence + 2-bit methylation state per bp per mod-type step


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench] {msg}", file=sys.stderr, flush=True)


def build_candidates(mg, workload, n_cand, per_group):
    from nanomotif_amd import synth
    from nanomotif_amd.motif import Motif
    bins = sorted(set(mg.bin_names))
    if workload == "cfg5":
        raw = synth.random_candidates(n_cand, seed=2, mod_types=mg.spec.mod_types)
        n_mt = len(mg.spec.mod_types)
        return [(Motif(s, p), mt, bins[(k // n_mt) % len(bins)]) for k, (s, p, mt) in enumerate(raw)]
    # "greedy": one lock-step expansion of every (bin, mod type) search — `per_group` sibling children of a
    # 5-position parent (the shape MotifSearcher.run submits, find_motifs_bin.py:1116-1135)
    out = []
    rng = np.random.Generator(np.random.PCG64(7))
    for b in bins:
        for mt in mg.spec.mod_types:
            can = synth.MOD_CANONICAL[mt]
            core = list("." * 41)
            core[20] = can
            for q in rng.choice([14, 15, 16, 17, 18, 19, 21, 22, 23, 24, 25], size=4, replace=False):
                core[int(q)] = "ACGT"[int(rng.integers(4))]
            free = [i for i in range(12, 28) if core[i] == "."]
            pos = free[int(rng.integers(len(free)))]
            for base in "ATGC"[:per_group]:
                c = list(core)
                c[pos] = base
                out.append((Motif("".join(c), 20), mt, b))
    return out


def cpu_baseline_worker(args):
    """Score one bin's candidates with the CPU oracle (regex + numpy, the reference's own primitives)."""
    spec_kw, bin_name, cands = args
    from nanomotif_amd import synth
    from oracle.scan import ContigPileup, score_candidates
    mg = synth.make_metagenome(synth.SynthSpec(**spec_kw))
    t_gen = time.perf_counter()
    idx = [i for i, b in enumerate(mg.bin_names) if b == bin_name]
    seqs = {mg.names[i]: mg.contig_str(i) for i in idx}
    piles = {}
    for mt in sorted({c[2] for c in cands}):
        piles[mt] = {}
        for i in idx:
            p = mg.contig_pileup(i, mt)
            keep = p["nvalid"] > 5
            piles[mt][mg.names[i]] = ContigPileup(p["position"][keep], p["strand"][keep],
                                                  synth.pct_to_fraction(p["pct_hundredths"][keep]))
    t0 = time.perf_counter()
    out = {}
    for mt in piles:
        these = [(k, s, p) for k, (s, p, m) in enumerate(cands) if m == mt]
        res = score_candidates(piles[mt], seqs, [(s, p) for _, s, p in these])
        for (k, _, _), r in zip(these, res):
            out[k] = r.tolist()
    t1 = time.perf_counter()
    bp = int(sum(int(mg.lengths[i]) for i in idx))
    return bin_name, [out[k] for k in range(len(cands))], t1 - t0, t0 - t_gen, bp


def run_e2e(args, mg, device, local_rank, world, rank):
    """--workload e2e: the whole motif_discovery pipeline on the synthetic metagenome: raw pileup rows -> device-side
    filters -> windows -> lock-step greedy search with pruning -> post-processing.  One step = one full run.  N > 1:
    whole bins per GPU (weak scaling) — every rank runs the pipeline on its own metagenome (seed 1 + rank), no
    collective; wall = slowest rank."""
    import torch
    import torch.distributed as dist
    from nanomotif_amd import e2e_synth
    from nanomotif_amd.engine import ScanEngine
    if world > 1 and args.scaling != "weak":
        raise SystemExit("--workload e2e at N > 1 is the whole-bin (weak) mode; contig sharding of the CLI is covered by tests/test_gpu_cli.py")
    eng = ScanEngine(local_rank)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    rows, t = e2e_synth.run(mg, eng, device)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    eng.close()
    rows = [r for r in rows if r.n_mod + r.n_nomod >= 50]
    planted = {(b, m[0]) for b, ms in mg.bin_motifs.items() for m in ms}
    found = {(r.reference, r.motif_iupac) for r in rows}
    stats = [wall, len(rows), len(planted), len(planted & found)]
    if world > 1:
        tt = torch.tensor(stats, dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt)
        stats = [float(mx[0]), int(tt[1]), int(tt[2]), int(tt[3])]
        dist.barrier()
    if rank == 0:
        print(json.dumps({
            "metric": "end-to-end motif_discovery wall seconds (synthetic metagenome)", "value": stats[0], "unit": "s", "n_gpus": world,
            "steps": 1, "warmup": 0, "ms_per_step": stats[0] * 1e3, "higher_is_better": False, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 bit-planes / int64 counts / f64 scores", "data": "synthetic",
            "config": {"workload": f"e2e: motif_discovery on {world} x {args.total_bp:,} bp ({args.contigs} contigs, {args.bins} bins, 6mA+5mC per GPU)"},
            "pipeline_s_rank0": t["upload_filter_s"] + t["search_s"],     # device filters + search, without the synthetic data generation
            "timings_rank0": t, "motifs_reported": stats[1], "planted_motifs": stats[2], "planted_recovered": stats[3]}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-bp", type=int, default=1_000_000_000)
    ap.add_argument("--contigs", type=int, default=10_000)
    ap.add_argument("--bins", type=int, default=500)
    ap.add_argument("--candidates", type=int, default=10_000)
    ap.add_argument("--workload", choices=["cfg5", "greedy", "e2e"], default="cfg5")
    ap.add_argument("--per-group", type=int, default=2, help="greedy workload: children per (bin, mod type)")
    ap.add_argument("--cpu-bins", type=int, default=-1, help="bins in the CPU-baseline sample (-1: two per worker; 0: skip)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (debug: several ranks on one GPU)")
    ap.add_argument("--force-device", type=int, default=-1, help="debug: CUDA device for every rank")
    ap.add_argument("--hbm-round-steps", type=int, default=20, help="extra launches of a greedy round for the HBM-bound roofline (0: skip)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="CPU-baseline worker processes (0: min(32, host cores))")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = whole bins per GPU, one --total-bp metagenome each, no collective (default); "
                         "strong = one metagenome, contigs sharded, count tables all-reduced every step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from nanomotif_amd import synth, synth_device
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.shard import assign_contigs

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        log(f"note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    weak = args.scaling == "weak"
    reduce_counts = world > 1 and not weak
    seed = 1 + rank if weak else 1
    spec = synth.SynthSpec(n_contigs=args.contigs, total_bp=args.total_bp, n_bins=args.bins, mod_types=("a", "m"), seed=seed)
    spec_kw = dict(n_contigs=args.contigs, total_bp=args.total_bp, n_bins=args.bins, mod_types=("a", "m"), seed=seed)
    mg = synth.make_metagenome(spec)
    if weak:
        mine = np.arange(len(mg.names))
    else:
        mine = assign_contigs(mg.lengths, world, bins=mg.bin_names)[rank]

    if args.workload == "e2e":
        return run_e2e(args, mg, device, local_rank, world, rank)

    t0 = time.perf_counter()
    eng = ScanEngine(local_rank)
    rows = synth_device.load_engine_from_device(eng, mg, device, contigs=None if (world == 1 or weak) else mine, progress=log)
    # one explicit side stream carries the engine's launches AND the collectives (the legacy default stream has the
    # handle 0, which nm_set_stream reads as "use the ctx's own stream": never hand it that)
    side = torch.cuda.Stream(device)
    assert side.cuda_stream != 0
    eng.use_stream(side.cuda_stream)
    torch.cuda.set_stream(side)
    st = eng.stats()
    log(f"resident: {st['total_bp']:,} bp ({st['padded_bp']:,} padded), rows {rows}, setup {time.perf_counter() - t0:.1f}s")

    cands = build_candidates(mg, args.workload, args.candidates, args.per_group)
    bin_bp = {}
    for i, b in enumerate(mg.bin_names):
        bin_bp[b] = bin_bp.get(b, 0) + int(mg.lengths[i])
    sites_per_step = sum(2 * bin_bp[b] for _, _, b in cands)
    groups = {(b, mt) for _, mt, b in cands}
    algo_bytes_total = ALGO_BYTES_PER_BP_STEP * sum(bin_bp[b] for b, _ in groups) + 16 * len(cands)
    my_bin_bp = {}
    for i in mine:
        my_bin_bp[mg.bin_names[i]] = my_bin_bp.get(mg.bin_names[i], 0) + int(mg.lengths[i])
    algo_bytes_rank = ALGO_BYTES_PER_BP_STEP * sum(my_bin_bp.get(b, 0) for b, _ in groups) + 16 * len(cands)

    # two count tables: the all-reduce of step k (RCCL stream) overlaps the scoring launch of step k+1
    counts = [torch.zeros((len(cands), 2), dtype=torch.int64, device=device) for _ in range(2)]
    pending = [None, None]
    batch = eng.make_batch(cands)      # the step's input: the candidate table in the C-ABI's flat SoA form
    step_no = [0]

    def step():
        # inside the C ABI, every call: drop candidates of bins this rank does not hold, sort by (mod type, bin),
        # compile every motif to its constraint program, ship programs + tables (pinned staging ring, copy stream),
        # zero the counters, one scoring launch (async)
        i = step_no[0] & 1
        step_no[0] += 1
        if pending[i] is not None:
            pending[i].wait()                               # table i is free again (its all-reduce finished)
        eng.score_into_device(batch, counts[i].data_ptr())
        if reduce_counts and args.dist_backend == "nccl":
            pending[i] = dist.all_reduce(counts[i], async_op=True)   # RCCL sum over xGMI
        elif reduce_counts:                                          # debug path: reduce through the host
            host = counts[i].cpu()
            dist.all_reduce(host)
            counts[i].copy_(host)

    def drain():
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    eng.timing_reset(True)
    torch.cuda.synchronize(device)
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    kernel_ms_total, n_launch = eng.timing_total()
    eng.timing_reset(False)
    if world > 1:
        t = torch.tensor([elapsed, kernel_ms_total / max(n_launch, 1)], dtype=torch.float64,
                         device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])
    else:
        kernel_ms = kernel_ms_total / max(n_launch, 1)
    final = counts[(step_no[0] - 1) & 1].cpu().numpy()

    # ---- the same engine in its HBM-bound regime: one lock-step greedy round (2 sibling children per (bin, mod type),
    # the shape MotifSearcher.run submits), kernel time from HIP events; reported next to the main roofline
    hbm_round = None
    if args.workload == "cfg5" and args.hbm_round_steps > 0:
        g_cands = build_candidates(mg, "greedy", 0, 2)
        g_batch = eng.make_batch(g_cands)
        g_counts = torch.zeros((len(g_cands), 2), dtype=torch.int64, device=device)
        for _ in range(3):
            eng.score_into_device(g_batch, g_counts.data_ptr())
        torch.cuda.synchronize(device)
        eng.timing_reset(True)
        for _ in range(args.hbm_round_steps):
            eng.score_into_device(g_batch, g_counts.data_ptr())
        torch.cuda.synchronize(device)
        g_ms, g_n = eng.timing_total()
        eng.timing_reset(False)
        g_ms /= max(g_n, 1)
        g_groups = {(b, mt) for _, mt, b in g_cands}
        g_bytes = ALGO_BYTES_PER_BP_STEP * sum(my_bin_bp.get(b, 0) for b, _ in g_groups) + 16 * len(g_cands)
        if world > 1:
            tt = torch.tensor([g_ms], dtype=torch.float64, device=device if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            g_ms = float(tt[0])
        hbm_round = {"workload": f"greedy round: {len(g_cands)} candidates = 2 sibling children per (bin, mod type)",
                     "bound": "hbm", "kernel_ms": g_ms, "achieved": g_bytes / (g_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": g_bytes, "launches": g_n,
                     "motif_sites_per_s": sum(2 * bin_bp[b] for _, _, b in g_cands) * (world if weak else 1) / (g_ms * 1e-3)}

    result = None
    traffic, traffic_src, valu_insts = None, None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tj):
        t = json.load(open(tj))
        # measured per launch on one GPU; every rank of a weak-scaling run launches that same configuration
        if (t.get("workload"), t.get("total_bp"), t.get("candidates")) == (args.workload, args.total_bp, len(cands)) \
                and (t.get("n_gpus") == world or (weak and t.get("n_gpus") == 1)):
            traffic, traffic_src = t["hbm_bytes_per_launch"], t["source"]
            valu_insts = t.get("sq_insts_valu_per_launch")
    if rank == 0:
        value = sites_per_step * (world if weak else 1) * args.steps / elapsed
        achieved = algo_bytes_rank / (kernel_ms * 1e-3) / 1e9
        result = {
            "metric": "motif-sites scored/sec (1 Gbp synthetic metagenome)", "value": value, "unit": "motif-sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32 bit-planes / int64 counts",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {len(cands) * (world if weak else 1)} candidate motifs x {args.total_bp * (world if weak else 1):,} bp metagenome "
                                   f"({args.contigs} contigs, {args.bins} bins, 6mA+5mC), both strands",
                       "candidates": len(cands) * (world if weak else 1), "total_bp": args.total_bp * (world if weak else 1),
                       "contigs": args.contigs * (world if weak else 1), "bins": args.bins * (world if weak else 1),
                       "per_gpu": {"total_bp": args.total_bp, "candidates": len(cands)} if weak else None,
                       "mod_types": ["a", "m"],
                       "sharding": (f"whole bins per GPU: {world} x ({args.bins} bins, {args.total_bp:,} bp, {len(cands)} candidates), no collective"
                                    if weak else f"contigs over {world} GPU(s), longest-first, bins kept whole when small; count tables all-reduced"),
                       "motif_sites_per_step": sites_per_step * (world if weak else 1)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_src or "not collected for this configuration (rocprofv3 --pmc runs: profiles/)",
                         "kernel": "score_kernel<1,1,compact>", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algo_bytes_rank,
                         "note": "0.5 B/bp per (bin, mod type) step + 16 B per candidate; slowest rank at N>1"},
            "kernel_share_of_step": kernel_ms / (elapsed / args.steps * 1e3),
            "counts_checksum": [int(final[:, 0].sum()), int(final[:, 1].sum()),
                                int((final * np.arange(1, final.size + 1).reshape(final.shape) % 1000003).sum() % (2**61 - 1))],
        }
        if hbm_round:
            result["roofline_hbm_bound_round"] = hbm_round
        if valu_insts:
            # second roofline: the kernel is integer-VALU-issue bound once a (bin, mod type) carries more than ~2
            # candidates; instruction count from rocprofv3 (SQ_INSTS_VALU, profiles/), peak from tools/valu_peak.hip
            rate = valu_insts / (kernel_ms * 1e-3)
            result["roofline_valu"] = {"bound": "valu-int", "achieved": rate, "peak": VALU_PEAK_WAVE_INSTR_S,
                                       "unit": "wave64 integer instr/s", "frac": rate / VALU_PEAK_WAVE_INSTR_S,
                                       "sq_insts_valu_per_launch": valu_insts,
                                       "peak_source": "profiles/r1/valu_peak.txt (alignbit+and stream, 4 cycles per wave64 op per SIMD)"}

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on a bounded sample of the same workload, and parity
    if rank == 0 and world == 1 and args.cpu_bins != 0:
        import multiprocessing as mp
        ncores = os.cpu_count() or 1
        procs = args.cpu_procs if args.cpu_procs > 0 else min(32, ncores)
        nb = args.cpu_bins if args.cpu_bins > 0 else 2 * procs
        bins = sorted(set(mg.bin_names))
        sample = [bins[(k * 37) % len(bins)] for k in range(nb)]
        jobs = []
        for b in sample:
            idx = [k for k, c in enumerate(cands) if c[2] == b]
            jobs.append((spec_kw, b, [(cands[k][0].string, cands[k][0].mod_position, cands[k][1]) for k in idx]))
        procs = min(nb, procs)
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(procs) as pool:
            res = pool.map(cpu_baseline_worker, jobs, chunksize=1)
        wall = time.perf_counter() - t0
        cpu_sites = 0
        cpu_seconds = 0.0
        mismatches = 0
        for (b, table, secs, gen_secs, bp), job in zip(res, jobs):
            idx = [k for k, c in enumerate(cands) if c[2] == b]
            cpu_sites += 2 * bp * len(idx)
            cpu_seconds += secs
            for k, row in zip(idx, table):
                if final[k].tolist() != row:
                    mismatches += 1
        result["cpu_baseline"] = {
            # aggregate rate of `procs` concurrent workers = sites / (summed scan seconds / procs)
            "value": cpu_sites / (cpu_seconds / procs), "unit": "motif-sites/s", "cores": procs, "kind": "port",
            "per_core": cpu_sites / cpu_seconds,
            "sample": f"{nb} of {len(bins)} bins x their {len(jobs[0][2])} candidates ({cpu_sites:.3g} motif-sites), "
                      f"oracle/scan.py (regex overlapped finditer + numpy.isin) in a spawn Pool({procs}), one bin per task; "
                      f"{cpu_seconds:.1f} CPU-seconds of scanning, wall incl. host data generation {wall:.1f}s",
            "host_cores_available": ncores,
        }
        result["parity"] = {"candidates_checked": int(sum(len(j[2]) for j in jobs)), "mismatches": mismatches,
                            "against": "oracle/scan.py on the sampled bins (bit-exact integer counts)"}
        result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        if mismatches:
            log(f"PARITY FAILURE: {mismatches} candidates differ from the oracle")

    if rank == 0:
        print(json.dumps(result), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

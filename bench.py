#!/usr/bin/env python3
"""Benchmark of the motif-scoring hot path (BASELINE.json metric: motif-sites scored / second on the 1 Gbp synthetic
metagenome at 1 / 2 / 4 / 8 MI355X).

A *step* = one pass of the hot path over one batch: the cfg 5 candidate table (10 000 seeded IUPAC motifs, 20 per bin,
half 6mA half 5mC) scored against every contig of its bin on both strands through ``nm_score_batch_device`` — validate
and sort the candidate records on the host, ship them through the pinned staging ring, compile them to constraint
programs on the device, one scoring launch, count table in HBM — and, with more than one GPU, ONE RCCL all-reduce
(sum, int64) of the count table.  A *motif-site* = one (candidate x reference bp x strand) match test: a candidate on a
bin of L bp is 2·L sites.

    python bench.py [--gpus N] [--steps K] [--warmup W]

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (a torch.distributed.run child,
before this process touches the GPU); under torchrun it is one of the ranks.  Multi-GPU is STRONG scaling on the
BASELINE configuration (cfg 4/5): ONE --total-bp metagenome, its contigs sharded over the ranks
(nanomotif_amd/shard.py: longest-first, bins kept whole while they are small against a GPU's share), every rank scores
the candidates of the bins it holds and the int64[n_candidates, 2] count tables are summed with one all-reduce per step
(find_motifs_bin.py:1273-1283: counts are sums over contigs).  value = motif-sites of the whole job / slowest rank.
``--scaling weak`` (every rank its own metagenome, no collective) is kept as an explicit extra mode and is reported by
the default run only as the ``weak_scaling`` key, never as ``value``.

Inputs are generated on the device (nanomotif_amd/synth_device.py, bit-identical to the numpy generator) and are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.  Extra keys of the default run:
``roofline_hbm_bound_round`` (one greedy lock-step round, the HBM-bound regime), ``e2e`` (the whole motif_discovery
pipeline on the same metagenome, generation excluded), ``cfg5_all`` (every candidate against every bin),
``cpu_baseline`` (the oracle on all host cores and on one), ``per_rank`` (kernel / host / all-reduce times).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec)
HBM_STREAM_GBS = 6350.0        # what a plain streaming-read kernel reaches on this device (tools/hbm_peak.hip, profiles/r1/hbm_peak.txt)
VALU_PEAK_WAVE_INSTR_S = 6.1e11   # measured issue rate of THIS kernel's instruction mix (alignbit + and, tools/valu_peak.hip, profiles/r1/valu_peak.txt)
VALU_ISSUE_PEAK_WAVE_INSTR_S = 256 * 4 * 2.4e9 / 2   # the chip: 256 CU x 4 SIMD x 2.4 GHz, one wave64 VALU op per 2 cycles (MI355X_MICROARCH.md) = 1.23e12
# counts_checksum of the N = 1 run of a configuration (workload, total bp, contigs, bins, candidates) — what every N > 1
# run of the same seeds must reproduce after its all-reduce (BENCH_r02.json, profiles/r2/bench_n1.json)
N1_CHECKSUMS = {("cfg5", 1_000_000_000, 10_000, 500, 10_000): [5367162, 308740588, 8024022553]}
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


def self_launch(args) -> int:
    """--gpus N without a launcher: become the launcher.  The child is torch.distributed.run with this script, started
    BEFORE anything in this process touches the GPU; stdout (the JSON line of rank 0) is passed through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    log(f"starting {args.gpus} ranks: {' '.join(cmd)}")
    return subprocess.run(cmd, env=env).returncode


class Collective:
    """Sum of the int64 count tables over the ranks.  ``native``: nm_allreduce_counts of the C ABI (RCCL through the
    library's own communicator, on its communication stream); ``torch``: torch.distributed (RCCL with backend nccl,
    host round trip with gloo — the debugging backend that lets several ranks share one GPU)."""

    def __init__(self, kind, eng, world, rank, device, backend):
        import torch
        import torch.distributed as dist
        self.kind, self.eng, self.dist, self.backend, self.world = kind, eng, dist, backend, world
        self.pending = {}
        if kind == "native":
            # every rank must end up on the same path: agree on success before anybody uses the communicator
            ok = 1
            try:
                uid = [eng.comm_unique_id() if rank == 0 else None]
                if world > 1:
                    dist.broadcast_object_list(uid, src=0)
                eng.comm_init(rank, world, uid[0])
            except Exception as e:                      # no librccl, duplicate device, ...: torch.distributed carries the tables
                ok = 0
                print(f"[bench] rank {rank}: nm_comm_init failed ({e}); falling back to torch.distributed", file=sys.stderr, flush=True)
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=device if backend == "nccl" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                ok = int(flag[0])
            if ok == 0:
                self.kind = "torch" if world > 1 else "none"

    def start(self, slot, tensor):
        if self.kind == "none":
            return
        if self.kind == "native":
            self.eng.allreduce_counts_device(tensor.data_ptr(), tensor.numel(), slot)
        elif self.backend == "nccl":
            self.pending[slot] = self.dist.all_reduce(tensor, async_op=True)
        else:
            host = tensor.cpu()
            self.dist.all_reduce(host)
            tensor.copy_(host)

    def wait(self, slot):
        """Order the engine stream (and torch's current stream) after the all-reduce started on ``slot``."""
        if self.kind == "native":
            self.eng.comm_wait(slot)
        elif slot in self.pending:
            self.pending.pop(slot).wait()

    def drain(self):
        if self.kind == "native":
            self.eng.comm_sync()
        for slot in list(self.pending):
            self.pending.pop(slot).wait()


def usable_cores():
    """CPUs this process can really run on at once: the smallest of os.cpu_count(), the scheduler affinity and the
    cgroup CPU quota (a 256-core host may hand a container 16 CPUs' worth of time: 256 busy workers then run 16x
    throttled and a pool of that size measures the throttling, not the host)."""
    n = os.cpu_count() or 1
    why = f"os.cpu_count() = {n}"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, f"sched_getaffinity = {a}"
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period))
                if q < n:
                    n, why = q, f"cgroup CPU quota = {float(quota) / period:g} CPUs ({path})"
            break
        except Exception:
            continue
    return n, why


def cpu_baseline(result, final, cands, mg, spec_kw, args):
    """Rank 0, N = 1: the oracle on a bounded sample of the same workload, on ALL host cores (T = the CPUs this process may use:
    os.cpu_count() capped by affinity and the cgroup quota; wall-clock of the concurrent scan phase, pool overhead
    included) and on ONE core, with the
    sampled bins' counts compared bit for bit with the GPU table."""
    from oracle import pipeline as opl
    ncores = os.cpu_count() or 1
    usable, why = usable_cores()
    procs = args.cpu_procs if args.cpu_procs > 0 else usable
    try:
        import psutil
        avail = psutil.virtual_memory().available
        procs = max(1, min(procs, int(avail * 0.5 / (250 << 20))))      # ~250 MB per worker holding two 2 Mbp bins (measured 130 MB with one)
    except Exception:
        pass
    bins = sorted(set(mg.bin_names))
    nb = args.cpu_bins if args.cpu_bins > 0 else min(len(bins), 2 * procs)
    sample = [bins[(k * 37) % len(bins)] for k in range(nb)]
    sample = list(dict.fromkeys(sample))
    by_bin = {}
    for k, c in enumerate(cands):
        by_bin.setdefault(c[2], []).append(k)
    jobs = [(spec_kw, b, [(cands[k][0].string, cands[k][0].mod_position, cands[k][1]) for k in by_bin.get(b, [])]) for b in sample]
    jobs = [j for j in jobs if j[2]]
    procs = min(procs, len(jobs))
    t0 = time.perf_counter()
    res, wall, cpu_seconds = opl.timed_pool(jobs, procs)
    total_wall = time.perf_counter() - t0
    cpu_sites, mismatches, checked = 0, 0, 0
    for (b, table, bp), job in zip(res, jobs):
        idx = by_bin[b]
        cpu_sites += 2 * bp * len(idx)
        checked += len(idx)
        for k, row in zip(idx, table):
            if final[k].tolist() != row:
                mismatches += 1
    # one core: the first jobs until ~10 s of scanning
    t1_sites, t1_secs = 0, 0.0
    for job in jobs[:max(1, args.cpu_t1_bins)]:
        _, _, secs, _, bp = opl.score_worker(job)
        t1_sites += 2 * bp * len(job[2])
        t1_secs += secs
    result["cpu_baseline"] = {
        "value": cpu_sites / wall, "unit": "motif-sites/s", "cores": procs, "kind": "port",
        "value_t1": t1_sites / t1_secs, "per_core_in_pool": cpu_sites / cpu_seconds,
        "sample": f"{len(jobs)} of {len(bins)} bins x their {len(jobs[0][2])} candidates ({cpu_sites:.3g} motif-sites): oracle/scan.py "
                  f"(regex overlapped finditer + numpy.isin per contig and strand) on {procs} concurrent spawn processes, one bin per task, "
                  f"wall-clock {wall:.2f} s of the scan phase (inputs built before a barrier; {total_wall:.1f} s with process start and "
                  f"input generation), {cpu_seconds:.1f} CPU-seconds; -t 1: {min(len(jobs), max(1, args.cpu_t1_bins))} bins in {t1_secs:.1f} s",
        "host_cores_visible": ncores, "host_cores_usable": usable, "usable_limit": why,
    }
    result["parity"] = {"candidates_checked": checked, "mismatches": mismatches,
                        "against": "oracle/scan.py on the sampled bins (bit-exact integer counts)"}
    result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
    if mismatches:
        log(f"PARITY FAILURE: {mismatches} candidates differ from the oracle")


def time_launches(eng, batch, out_ptr, n, device):
    """Mean scoring-kernel duration (HIP events on the launch stream, nm_timing_*) over ``n`` launches of ``batch``."""
    import torch
    for _ in range(2):
        eng.score_into_device(batch, out_ptr)
    torch.cuda.synchronize(device)
    eng.timing_reset(True)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.score_into_device(batch, out_ptr)
    torch.cuda.synchronize(device)
    wall = (time.perf_counter() - t0) / n
    ms, k = eng.timing_total()
    eng.timing_reset(False)
    return ms / max(k, 1), wall * 1e3


def kernel_source_sha16():
    """First 16 hex digits of the sha256 of the scoring kernel's source: ties the stored rocprofv3 counter entries
    (profiles/traffic.json) to the code they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("nmscan.hip", "nmscan_device.h", "nmscan_internal.h"):       # (profiles/summarize.py hashes the same two files)
        h.update(open(os.path.join(ROOT, "nanomotif_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def load_traffic(workload, total_bp, n_cand):
    """The stored rocprofv3 --pmc numbers of this configuration (profiles/summarize.py wrote them), with ``stale`` = the
    kernel source changed since they were measured."""
    tj = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tj):
        return None
    t = json.load(open(tj))
    entries = t if isinstance(t, list) else t.get("entries", [t])
    for e in entries:
        if (e.get("workload"), e.get("total_bp"), e.get("candidates"), e.get("n_gpus", 1)) == (workload, total_bp, n_cand, 1):
            e = dict(e)
            e["stale"] = e.get("kernel_source_sha16") != kernel_source_sha16()
            return e
    return None


def checksum(table):
    return [int(table[:, 0].sum()), int(table[:, 1].sum()),
            int((table * np.arange(1, table.size + 1).reshape(table.shape) % 1000003).sum() % (2**61 - 1))]


def run_e2e(mg, eng_device, device, my_bins=None, lanes=1):
    """The whole motif_discovery pipeline on (the bins ``my_bins`` of) the metagenome: raw pileup rows -> device-side
    filters -> windows -> lock-step greedy search with pruning -> post-processing.  ``lanes`` > 1: the bins dealt to that many
    engines on this device that run side by side (e2e_synth.run_lanes).  Returns (rows, timings)."""
    import torch
    from nanomotif_amd import e2e_synth
    from nanomotif_amd.engine import ScanEngine
    if lanes > 1:
        assert my_bins is None
        engines = [ScanEngine(eng_device) for _ in range(lanes)]
        try:
            rows, t = e2e_synth.run_lanes(mg, engines, device)
        finally:
            for e in engines:
                e.close()
        return [r for r in rows if r.n_mod + r.n_nomod >= 50], t
    eng = ScanEngine(eng_device)
    t0 = time.perf_counter()
    rows, t = e2e_synth.run(mg, eng, device, bins=my_bins)
    torch.cuda.synchronize(device)
    t["wall_with_generation_s"] = time.perf_counter() - t0
    eng.close()
    return [r for r in rows if r.n_mod + r.n_nomod >= 50], t


def run_cli_extra(device, log_fn, total_bp=100_000_000):
    """File to bin-motifs.tsv: BASELINE cfg 3 (100 Mbp, 1000 contigs, 50 bins, 6mA + 5mC) as FILES on a tmpfs — FASTA, modkit
    bedMethyl TEXT (7.8 GB, 1e8 rows), contig-bin TSV — through `python -m nanomotif_amd motif_discovery`, wall clock of the
    whole process, with the split the CLI records itself (parse / device filters / search / write), once with the
    device-side parser (the default) and once with the host parser."""
    import shutil
    import tempfile
    from nanomotif_amd import e2e_synth, synth
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    need = int(96 * total_bp)                    # ~75 B of text + ~16 B of bgzip per raw row, one row per bp
    free = shutil.disk_usage(base).free
    if free < 1.15 * need:
        raise RuntimeError(f"cli extra: {base} has {free / 1e9:.1f} GB free, the files of a {total_bp:,} bp run need ~{need / 1e9:.1f} GB "
                           "(set TMPDIR, or a smaller --cli-bp)")
    tmp = tempfile.mkdtemp(prefix="nm_bench_cli_", dir=base)
    try:
        spec = synth.config("cfg3") if total_bp == 100_000_000 else synth.SynthSpec(
            n_contigs=max(8, total_bp // 100_000), total_bp=total_bp, n_bins=max(2, total_bp // 2_000_000), mod_types=("a", "m"), seed=1)
        mg3 = synth.make_metagenome(spec)
        t0 = time.perf_counter()
        sizes = e2e_synth.write_text_inputs(mg3, tmp, device)
        log_fn(f"cli extra: wrote {sizes['bed_bytes'] / 1e9:.2f} GB of bedMethyl text ({sizes['rows']:,} rows) in {time.perf_counter() - t0:.1f}s to {tmp}")
        # Both parsers are timed on the same file state: resident in the page cache and READ once.  The first reader of a
        # freshly written tmpfs file pays ~0.35 s over later ones for these 7.5 GB (tools/cli_repeat_probe.py: 0.60 s
        # against 0.26 s of parse), which would land on whichever parser runs first.
        t0 = time.perf_counter()
        with open(os.path.join(tmp, "pileup.bed"), "rb", buffering=0) as f:
            buf = bytearray(64 << 20)
            while f.readinto(buf):
                pass
        warm_s = time.perf_counter() - t0
        out = {"what": f"python -m nanomotif_amd motif_discovery on FILES: {total_bp:,} bp FASTA + {sizes['bed_bytes'] / 1e9:.2f} GB modkit bedMethyl text "
                       f"({sizes['rows']:,} rows) on a tmpfs (page cache, read through once after writing) -> bin-motifs.tsv; wall clock of "
                       "the whole process, cold interpreter and HIP runtime each time",
               "bed_bytes": sizes["bed_bytes"], "rows": sizes["rows"], "total_bp": total_bp, "warm_read_s": warm_s}
        # the same pileup as bgzip + tabix (what the reference recommends, docs/source/required_files.md:21): written natively
        # (libnmsynth: zlib level 6 on all threads), read through its index
        t0 = time.perf_counter()
        e2e_synth.bgzip_tabix(os.path.join(tmp, "pileup.bed"), os.path.join(tmp, "pileup.bed.gz"))
        out["gz_bytes"] = os.path.getsize(os.path.join(tmp, "pileup.bed.gz"))
        log_fn(f"cli extra: bgzip + tabix of the pileup: {out['gz_bytes'] / 1e9:.2f} GB in {time.perf_counter() - t0:.1f}s")
        with open(os.path.join(tmp, "pileup.bed.gz"), "rb", buffering=0) as f:
            buf = bytearray(64 << 20)
            while f.readinto(buf):
                pass
        texts = {}
        for leg, pileup, parser in (("device", "pileup.bed", "device"), ("host", "pileup.bed", "host"), ("gz_device", "pileup.bed.gz", "device"),
                                    ("gz_host", "pileup.bed.gz", "host")):
            env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
                env.pop(k, None)
            if parser == "host":
                env["NANOMOTIF_HOST_PARSER"] = "1"
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", pileup, "-c", "contig_bin.tsv",
                                "--out", "out_" + leg], cwd=tmp, env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode:
                out[leg] = {"error": (r.stdout + r.stderr)[-500:]}
                continue
            tj = os.path.join(tmp, "out_" + leg, "logs", "timings.motif_discovery.json")
            t = json.load(open(tj)) if os.path.exists(tj) else {}
            texts[leg] = open(os.path.join(tmp, "out_" + leg, "bin-motifs.tsv")).read()
            out[leg] = {"wall_s": wall, "pileup_parse_s": t.get("pileup_parse_s"), "upload_filter_s": t.get("upload_filter_s"),
                        "search_s": t.get("search_s"), "assembly_s": t.get("assembly_s"), "engine_start_s": t.get("engine_start_s"),
                        "write_s": t.get("write_s"), "in_find_motifs_bin_s": t.get("find_motifs_bin_s"), "pileup_parser": t.get("pileup_parser"),
                        "text_GB_per_s_of_parse": sizes["bed_bytes"] / 1e9 / t["pileup_parse_s"] if t.get("pileup_parse_s") else None,
                        "motif_rows": max(len(texts[leg].splitlines()) - 1, 0)}
        if "device" in texts and "host" in texts:
            out["outputs_byte_equal"] = texts["device"] == texts["host"]
        if "gz_device" in texts and "gz_host" in texts:
            # (a .gz run seeds once per bin, a plain run once per (bin, mod type) — find_motifs_bin.py:152-171 / :219-248 — so
            # the two formats are compared within themselves)
            out["gz_outputs_byte_equal"] = texts["gz_device"] == texts["gz_host"]
        if "gz_device" in out and "device" in out and "wall_s" in out["gz_device"] and "wall_s" in out["device"]:
            out["gz_over_plain_wall"] = out["gz_device"]["wall_s"] / out["device"]["wall_s"]
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _memory_budget_bytes(base):
    """What the files of an extra may take under ``base``: its free space, and — a tmpfs lives in this cgroup's memory — the
    cgroup's limit minus what is in use."""
    import shutil
    free = shutil.disk_usage(base).free
    try:
        limit = open("/sys/fs/cgroup/memory.max").read().strip()
        used = int(open("/sys/fs/cgroup/memory.current").read())
        if limit != "max":
            free = min(free, int(limit) - used)
    except (OSError, ValueError):
        pass
    return free


def run_cli1g_extra(device, log_fn, total_bp=1_000_000_000, parity_bins=16, gz_level=6, keep_dir=None):
    """The drop-in command at the size the metric is quoted on, from FILES: BASELINE cfg 4 / cfg 5's metagenome (1 Gbp, 10 000
    contigs, 500 bins, 6mA + 5mC: 1e9 pileup rows) as assembly.fasta + pileup.bed.gz + .tbi + contig_bin.tsv on a tmpfs, written
    part by part by libnmsynth.so (the 75 GB of bedMethyl text never exist), then ONE cold `python -m nanomotif_amd
    motif_discovery` process: wall clock and the phase split the CLI records (engine start, FASTA, pileup read / inflate / parse,
    device filters, search, write), each phase's GB/s, and the bin-motifs.tsv rows of ``parity_bins`` seeded bins against the
    oracle pipeline (bgzip seeding: once per bin, find_motifs_bin.py:219-248).  Falls back to the largest size that fits and
    says which."""
    import shutil
    import tempfile
    from nanomotif_amd import e2e_synth, synth
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    asked = total_bp
    budget = _memory_budget_bytes(base)
    # per bp: ~15 B of .gz + 1 B of FASTA on the tmpfs; working set of the writer (one part of 25 bins: rows + text twice) and of
    # the CLI process (pinned slabs, the mapped .gz is already counted) ~ 12 GB
    while total_bp > 50_000_000 and 17.0 * total_bp + 14e9 > 0.8 * budget:
        total_bp //= 2
    out = {"asked_total_bp": asked, "total_bp": total_bp, "tmpfs": base, "budget_GB": budget / 1e9,
           "size_note": ("the size BASELINE's metric is quoted on" if total_bp == asked else
                         f"FELL BACK from {asked:,} bp: {base} / the cgroup leave {budget / 1e9:.0f} GB, the files + working set of {asked:,} bp need ~{(17.0 * asked + 14e9) / 0.8 / 1e9:.0f} GB")}
    tmp = keep_dir or tempfile.mkdtemp(prefix="nm_bench_cli1g_", dir=base)
    os.makedirs(tmp, exist_ok=True)
    try:
        spec = synth.config("cfg5") if total_bp == 1_000_000_000 else synth.config("cfg3") if total_bp == 100_000_000 else synth.SynthSpec(
            n_contigs=max(8, total_bp // 100_000), total_bp=total_bp, n_bins=max(2, total_bp // 2_000_000), mod_types=("a", "m"), seed=1)
        mg = synth.make_metagenome(spec)
        t0 = time.perf_counter()
        sizes = e2e_synth.write_gz_inputs_streaming(mg, tmp, device, bins_per_part=25, level=gz_level, log=log_fn)
        out["write_s"] = time.perf_counter() - t0
        out.update(rows=sizes["rows"], bed_text_bytes=sizes["bed_bytes"], gz_bytes=sizes["gz_bytes"], fasta_bytes=sizes["fasta_bytes"], gz_level=gz_level)
        log_fn(f"cli1g: {sizes['rows']:,} rows = {sizes['bed_bytes'] / 1e9:.1f} GB of bedMethyl text as {sizes['gz_bytes'] / 1e9:.2f} GB of bgzip + tabix, "
               f"{sizes['fasta_bytes'] / 1e9:.2f} GB of FASTA, written in {out['write_s']:.0f}s to {tmp}")
        for name in ("pileup.bed.gz", "assembly.fasta"):               # page cache, read through once (see run_cli_extra)
            with open(os.path.join(tmp, name), "rb", buffering=0) as f:
                buf = bytearray(64 << 20)
                while f.readinto(buf):
                    pass
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        legs = {}
        # NM_BENCH_CLI1G_LEGS = "name:KEY=VALUE;KEY=VALUE,name2:..." adds A/B legs with other environments (builder probes)
        more = tuple((spec.split(":", 1)[0], dict(kv.split("=", 1) for kv in spec.split(":", 1)[1].split(";") if kv), )
                     for spec in os.environ.get("NM_BENCH_CLI1G_LEGS", "").split(",") if ":" in spec)
        # The device is left idle for two seconds before every process: the driver scrubs the memory the process before (or this one's
        # generation) gave back, and a process that starts while that is going on waits for it in its first large allocation — 0.3 - 0.4 s
        # on a run of 0.6 (measured, profiles/r6/cli_fixed_costs.md); a user's run does not start in the wake of another one
        settle_s = float(os.environ.get("NM_BENCH_CLI_SETTLE_S", "2"))
        out["device_left_idle_before_each_process_s"] = settle_s
        for leg, extra_env in (("cold", {}), ("again", {"NM_BED_TIMING": "1", "NM_FASTA_TIMING": "1", "NM_SEARCH_TIMING": "1"})) + more:      # (the second run also prints the parser's per-slab split)
            if settle_s > 0:
                import torch
                torch.cuda.synchronize(device)
                time.sleep(settle_s)
            t0 = time.perf_counter()
            r = subprocess.run([sys.executable, "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", "pileup.bed.gz", "-c", "contig_bin.tsv",
                                "--out", "out_" + leg], cwd=tmp, env=dict(env, **extra_env), capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode:
                legs[leg] = {"error": (r.stdout + r.stderr)[-1500:]}
                continue
            t = json.load(open(os.path.join(tmp, "out_" + leg, "logs", "timings.motif_discovery.json")))
            gb = lambda nbytes, s: (nbytes / 1e9 / s) if s else None
            phases = {
                "interpreter_and_imports_s": wall - t.get("find_motifs_bin_s", 0.0),
                "engine_start_s": t.get("engine_start_s"),
                "fasta_s": t.get("assembly_s"), "fasta_reading_s": t.get("assembly_reading_s"), "fasta_parser": t.get("assembly_parser"),
                "pileup_s": t.get("pileup_parse_s"), "pileup_read_s": t.get("pileup_reading_s"), "pileup_inflate_s": t.get("pileup_inflating_s"),
                "pileup_parse_s": t.get("pileup_parsing_s"), "pileup_parser": t.get("pileup_parser"),
                "pileup_index_and_block_walk_s": (t.get("pileup_parse_s") - t.get("pileup_in_parser_s")) if t.get("pileup_in_parser_s") else None,
                # the tabix index + the walk over the BGZF blocks, done on a thread beside the engine start and the FASTA parse (not in pileup_s)
                "pileup_plan_on_a_thread_s": t.get("pileup_plan_s_on_a_thread"),
                "filters_s": t.get("upload_filter_s"), "search_s": t.get("search_s"), "write_s": t.get("write_s"),
            }
            rates = {
                "fasta_file_GB_per_s": gb(sizes["fasta_bytes"], t.get("assembly_s")),
                "pileup_gz_GB_per_s_read": gb(sizes["gz_bytes"], t.get("pileup_reading_s")),
                "pileup_text_GB_per_s_inflate": gb(sizes["bed_bytes"], t.get("pileup_inflating_s")),
                "pileup_text_GB_per_s_whole_parse": gb(sizes["bed_bytes"], t.get("pileup_parse_s")),
                "pileup_gz_GB_per_s_whole_parse": gb(sizes["gz_bytes"], t.get("pileup_parse_s")),
                "raw_rows_GB_per_s_filters": gb(22 * sizes["rows"], t.get("upload_filter_s")),
                "pcie_h2d_measured_GB_per_s": 57.0,
            }
            timed = {k: v for k, v in phases.items() if k.endswith("_s") and isinstance(v, float) and k not in ("fasta_reading_s", "pileup_read_s", "pileup_inflate_s", "pileup_parse_s", "pileup_index_and_block_walk_s", "pileup_plan_on_a_thread_s")}
            # what search_s is made of (main.py: TIMINGS["search_" + key] of discover()'s laps): plan, background, the native search, post-processing
            # (with --out: + the precleanup tables, background PSSMs and search graphs of every task, written from Python)
            phases["search_parts_s"] = {k[len("search_"):]: round(v, 4) for k, v in t.items() if k.startswith("search_") and k != "search_s" and isinstance(v, float)}
            # ... and filters_s (main.py: engine upload of the assembly, id tables, nm_ingest_pileup_part calls, window pipeline, closing the pileup table)
            phases["filters_parts_s"] = {k[len("filters_"):]: round(v, 4) for k, v in t.items() if k.startswith("filters_") and isinstance(v, float)}
            legs[leg] = {"wall_s": wall, "in_find_motifs_bin_s": t.get("find_motifs_bin_s"), "phases": phases, "rates": rates,
                         "the_wall_is": max(timed, key=timed.get), "assembly_s_per_Gbp": (t.get("assembly_s") or 0.0) / (total_bp / 1e9),
                         "motif_rows": max(len(open(os.path.join(tmp, "out_" + leg, "bin-motifs.tsv")).read().splitlines()) - 1, 0)}
            if extra_env:
                legs[leg]["parser_slab_log"] = [ln for ln in r.stderr.splitlines() if ln.startswith(("[bed]", "[fasta]", "[nm_search]", "[nm_ingest]", "[nm_plan_windows]", "[main]"))][:80]
        out["legs"] = legs
        texts = {leg: open(os.path.join(tmp, "out_" + leg, "bin-motifs.tsv")).read() for leg in legs if "error" not in legs[leg]}
        if len(texts) >= 2:
            out["both_runs_byte_equal"] = all(t == texts["cold"] for t in texts.values())
        # parity: the rows of `parity_bins` seeded bins against the oracle pipeline (CPU; bgzip task order and seeding)
        if texts and parity_bins:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from helpers import oracle_pipeline_parallel
            bins = sorted(set(mg.bin_names))
            rng = np.random.Generator(np.random.PCG64(2025))
            sample = [bins[i] for i in sorted(rng.choice(len(bins), size=min(parity_bins, len(bins)), replace=False).tolist())]
            t0 = time.perf_counter()
            exp = oracle_pipeline_parallel(mg, sample, max(1, min(parity_bins, (os.cpu_count() or 2) - 1)), bgzip_order=True)
            text = next(iter(texts.values()))
            lines = text.splitlines()
            col = lines[0].split("\t").index("reference")
            got = "\n".join([lines[0]] + [ln for ln in lines[1:] if ln.split("\t")[col] in set(sample)]) + "\n"
            out["parity"] = {"bins": sample, "byte_equal_to_the_oracle_pipeline": got == exp, "oracle_rows": exp.count("\n") - 1, "oracle_s": time.perf_counter() - t0}
            if got != exp:
                out["parity"]["product"], out["parity"]["oracle"] = got[:4000], exp[:4000]
        return out
    finally:
        if keep_dir is None:
            shutil.rmtree(tmp, ignore_errors=True)


def run_files_extra(device, log_fn):
    """The from-FILES leg of the DEFAULT line (round 6): BASELINE cfg 3 — 100 Mbp, 1000 contigs, 50 bins, 6mA + 5mC: 1e8 pileup rows — as
    assembly.fasta + pileup.bed.gz + .tbi + contig_bin.tsv written once to a tmpfs (outside the timed process), then ONE cold
    `python -m nanomotif_amd motif_discovery` process and a second one: wall clock, the phases the CLI records, and the
    bin-motifs.tsv rows of eight seeded bins against the oracle pipeline.  (`--extras cli1g` is the same at 1 Gbp: 15 GB of files,
    three minutes of writing — opt-in.)"""
    full = run_cli1g_extra(device, log_fn, total_bp=100_000_000, parity_bins=8)
    legs = full.get("legs", {})
    cold, again = legs.get("cold", {}), legs.get("again", {})
    out = {"workload": "BASELINE cfg 3 as FILES: 100 Mbp FASTA + pileup.bed.gz (bgzip, tabix index) + contig_bin.tsv on a tmpfs, page cache, one cold "
                       "`python -m nanomotif_amd motif_discovery` process (interpreter, HIP runtime, parsers, pre-filters, search, writers)",
           "total_bp": full.get("total_bp"), "rows": full.get("rows"), "bed_text_bytes": full.get("bed_text_bytes"), "gz_bytes": full.get("gz_bytes"),
           "fasta_bytes": full.get("fasta_bytes"), "files_written_in_s": full.get("write_s"), "size_note": full.get("size_note"),
           "wall_s": cold.get("wall_s"), "wall_s_second_process": again.get("wall_s"), "phases": cold.get("phases"), "rates": cold.get("rates"),
           "the_wall_is": cold.get("the_wall_is"), "motif_rows": cold.get("motif_rows"), "both_runs_byte_equal": full.get("both_runs_byte_equal"),
           "device_left_idle_before_each_process_s": full.get("device_left_idle_before_each_process_s")}
    for leg in (cold, again):
        if "error" in leg:
            out["error"] = leg["error"]
    out["second_process_log"] = again.get("parser_slab_log")      # (its NM_*_TIMING lines: where the parsers' and the search's time went)
    more = {k: {"wall_s": v.get("wall_s"), "phases": v.get("phases"), "log": v.get("parser_slab_log")} for k, v in legs.items() if k not in ("cold", "again")}
    if more:
        out["other_legs"] = more                                   # (NM_BENCH_CLI1G_LEGS: builder probes)
    par = full.get("parity") or {}
    out["parity"] = {"bins": len(par.get("bins", [])), "byte_equal_to_the_oracle_pipeline": par.get("byte_equal_to_the_oracle_pipeline"),
                     "oracle_rows": par.get("oracle_rows"), "oracle_s": par.get("oracle_s")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--total-bp", type=int, default=1_000_000_000)
    ap.add_argument("--contigs", type=int, default=10_000)
    ap.add_argument("--bins", type=int, default=500)
    ap.add_argument("--candidates", type=int, default=10_000)
    ap.add_argument("--workload", choices=["cfg5", "greedy", "cfg5_all", "e2e"], default="cfg5")
    ap.add_argument("--e2e-lanes", type=int, default=1,
                    help="one GPU: the end-to-end run's bins go through the pipeline in this many lanes side by side — one engine and one host thread "
                         "each (e2e_synth.run_lanes).  Measured in round 6 (profiles/r6/e2e/lanes.jsonl): two lanes 62-72 ms against 70-79 of one, "
                         "three and more slower than one; same rows.  Not the default: the gain is within the box's noise")
    ap.add_argument("--per-group", type=int, default=2, help="greedy workload: children per (bin, mod type)")
    ap.add_argument("--cpu-bins", type=int, default=-1, help="bins in the CPU-baseline sample (-1: two per worker, at most all; 0: skip)")
    ap.add_argument("--cpu-procs", type=int, default=0, help="CPU-baseline worker processes (0: os.cpu_count())")
    ap.add_argument("--cpu-t1-bins", type=int, default=3, help="bins scored by the single-core (-t 1) CPU leg")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (debug: several ranks on one GPU)")
    ap.add_argument("--allreduce", choices=["auto", "native", "torch"], default="auto",
                    help="count-table all-reduce: nm_allreduce_counts of the C ABI (RCCL) or torch.distributed; auto = native with nccl")
    ap.add_argument("--force-device", type=int, default=-1, help="debug: CUDA device for every rank")
    ap.add_argument("--cooldown", type=float, default=0.0, help="seconds of idle GPU before the warmup steps")
    ap.add_argument("--force-allreduce", action="store_true", help="debug: run the C-ABI all-reduce step even with one rank (RCCL world of 1)")
    ap.add_argument("--hbm-round-steps", type=int, default=20, help="extra launches of a greedy round for the HBM-bound roofline (0: skip)")
    ap.add_argument("--extras", default="auto", help="comma list of extra measurements of the cfg5 run: e2e,cfg5_all,weak,two_lanes,files (auto: those that apply; none; files = BASELINE cfg 3 from FASTA + .bed.gz + .tbi through one cold CLI process, N = 1 only), the opt-in cli (cfg 3 as FILES, plain and bgzip: writes ~9 GB to a tmpfs, four CLI processes) and the opt-in cli1g (the CLI at 1 Gbp from FASTA + .bed.gz + .tbi: ~16 GB on a tmpfs, minutes of writing)")
    ap.add_argument("--cli-bp", type=int, default=100_000_000, help="size of the file-to-bin-motifs.tsv extra (cfg 3: 100 Mbp = 7.8 GB of bedMethyl text)")
    ap.add_argument("--cli1g-bp", type=int, default=1_000_000_000, help="size of the opt-in cli1g extra (the CLI on FASTA + .bed.gz + .tbi at the headline size; "
                    "falls back to the largest size the tmpfs / cgroup holds and says which)")
    ap.add_argument("--cli1g-gz-level", type=int, default=6, help="zlib level of the synthetic bgzip pileup of the cli1g extra (bgzip's default is 6)")
    ap.add_argument("--as-rank-of", type=int, default=0, metavar="N",
                    help="debug, one GPU: hold the shard rank 0 of an N-rank run would hold (contigs as shard.assign_contigs deals "
                         "them, the whole candidate table in every call); counts are this shard's only")
    ap.add_argument("--prewarm", type=int, default=200,
                    help="untimed steps BEFORE the W warm-up steps: brings the device clocks to their steady state (a 20-step run is "
                         "12 ms long, shorter than the clock ramp); reported in the JSON line")
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 2],
                    help="scoring lanes (nm_set_score_lanes; 2 = consecutive steps overlap on the device).  0 = auto: both passes "
                         "run; one GPU: strict order is the headline and the two-lane pass is the key 'two_lanes'; several GPUs: "
                         "two lanes are the headline, 'strict_order' the other.  1: strict order only.  2: two lanes are the headline")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = ONE metagenome, contigs sharded, count tables all-reduced every step (default, the BASELINE "
                         "configuration); weak = every rank its own --total-bp metagenome, no collective")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                      # nothing above has touched the GPU

    # stdout carries exactly ONE line, the JSON of rank 0: whatever libraries print on the way (RCCL greets with a
    # version banner on stdout when a communicator is made) is sent to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    import torch
    import torch.distributed as dist
    from nanomotif_amd import synth, synth_device
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.shard import assign_bins, assign_contigs

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        sys.exit(2)
    if args.force_device >= 0:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # device memory of the library comes out of torch's pool (nm_set_device_allocator): the synthetic inputs are torch
    # tensors anyway, and a fresh hipMalloc of memory that earlier processes used is scrubbed by the driver at 7-30 GB/s
    # (tools/alloc_probe.py) — seconds that have nothing to do with the path measured here
    from nanomotif_amd import _lib as nm_lib
    nm_lib.use_torch_allocator(True)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)
        log(f"{world} ranks, backend {dist.get_backend()}")
    red_dev = device if args.dist_backend == "nccl" else "cpu"

    def allmax(vals):
        if world == 1:
            return [float(v) for v in vals]
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    def gather(vals):
        if world == 1:
            return [[float(v) for v in vals]]
        t = torch.tensor(vals, dtype=torch.float64, device=red_dev)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [o.tolist() for o in out]

    weak = args.scaling == "weak"
    reduce_counts = (world > 1 and not weak) or args.force_allreduce
    seed = 1 + rank if weak else 1
    spec_kw = dict(n_contigs=args.contigs, total_bp=args.total_bp, n_bins=args.bins, mod_types=("a", "m"), seed=seed)
    mg = synth.make_metagenome(synth.SynthSpec(**spec_kw))
    if world == 1 and args.as_rank_of > 1:
        mine = assign_contigs(mg.lengths, args.as_rank_of, bins=mg.bin_names)[0]
    elif weak or world == 1:
        mine = np.arange(len(mg.names))
    else:
        mine = assign_contigs(mg.lengths, world, bins=mg.bin_names)[rank]
    extras = {"e2e", "cfg5_all", "weak", "two_lanes", "files"} if args.extras == "auto" else set(x for x in args.extras.split(",") if x and x != "none")
    if args.workload != "cfg5" or weak:
        extras = set()
    if world == 1:
        extras.discard("weak")
    else:
        extras.discard("cfg5_all")
        extras.discard("cli")
        extras.discard("cli1g")
        extras.discard("files")

    if args.workload == "e2e":
        sizes = {}
        for i, b in enumerate(mg.bin_names):
            sizes[b] = sizes.get(b, 0) + int(mg.lengths[i])
        my_bins = None if world == 1 else assign_bins(sizes, world, tolerance=float("inf"))[rank]
        if world > 1:
            dist.barrier()
        lanes = args.e2e_lanes if world == 1 else 1
        rows, t = run_e2e(mg, local_rank, device, my_bins, lanes=lanes)
        pipeline = t["wall_s"] if lanes > 1 else t["upload_filter_s"] + t["search_s"]
        planted = {(b, m[0]) for b, ms in mg.bin_motifs.items() if my_bins is None or b in set(my_bins) for m in ms}
        found = {(r.reference, r.motif_iupac) for r in rows}
        per = gather([pipeline, t.get("search_s", 0.0), t.get("upload_filter_s", 0.0), len(rows), len(planted), len(planted & found), t["rounds"], t["candidates"]])
        if rank == 0:
            wall = max(p[0] for p in per)
            emit(({
                "metric": "end-to-end motif_discovery seconds (1 Gbp synthetic metagenome, device filters + search + post-processing)",
                "value": wall, "unit": "s", "n_gpus": world, "steps": 1, "warmup": 0, "ms_per_step": wall * 1e3, "higher_is_better": False,
                "scaling": "strong", "vs_baseline": None, "dtype": "u32 bit-planes / int64 counts / f64 scores", "data": "synthetic",
                "config": {"workload": f"e2e: motif_discovery on {args.total_bp:,} bp ({args.contigs} contigs, {args.bins} bins, 6mA+5mC), "
                                       f"whole bins per GPU over {world} GPU(s), no collective until the rows are gathered"
                                       + (f"; {lanes} lanes side by side on the GPU" if lanes > 1 else "")},
                "gpu_busy_over_wall": (t.get("gpu_busy_union_s", 0.0) / pipeline) if world == 1 else None,
                "per_rank": [dict(zip(["pipeline_s", "search_s", "upload_filter_s", "motif_rows", "planted", "planted_recovered", "rounds", "candidates"], p)) for p in per],
                "timings_rank0": t}))
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- extra: end to end on the same metagenome (generation excluded): whole bins per GPU, no collective.  Runs FIRST, on
    # a fresh device: it is the phase whose wall time depends on allocation speed (1000 window tasks, 9 planes)
    e2e_result = None
    extra_errors = {}
    try:
        if "e2e" in extras:
            sizes = {}
            for i, b in enumerate(mg.bin_names):
                sizes[b] = sizes.get(b, 0) + int(mg.lengths[i])
            my_bins = None if world == 1 else assign_bins(sizes, world, tolerance=float("inf"))[rank]
            if world > 1:
                dist.barrier()
            e_rows, t = run_e2e(mg, local_rank, device, my_bins)
            planted = {(b, m[0]) for b, ms in mg.bin_motifs.items() if my_bins is None or b in set(my_bins) for m in ms}
            found = {(r.reference, r.motif_iupac) for r in e_rows}
            per = gather([t["upload_filter_s"] + t["search_s"], t["search_s"], t["upload_filter_s"], t.get("gpu_busy_s", 0.0), t["rounds"], t["candidates"],
                          len(e_rows), len(planted), len(planted & found)])
            if rank == 0:
                e2e_result = {"what": "motif_discovery on the same metagenome: 1e9 raw pileup rows -> device filters -> windows -> lock-step greedy "
                                         "search + pruning -> post-processing; synthetic-data generation excluded; N > 1: whole bins per GPU",
                                 # the raw rows and the assembly are IN HBM when the clock starts (generated there): no file, no parse, no
                                 # PCIe; the from-files figures are the opt-in `cli` extra (plain text and bgzip)
                                 "from_device_resident_rows": True,
                                 "wall_s": max(p[0] for p in per), "search_s": max(p[1] for p in per), "upload_filter_s": max(p[2] for p in per),
                                 "gpu_busy_s": max(p[3] for p in per), "rounds": int(max(p[4] for p in per)), "candidates": int(sum(p[5] for p in per)),
                                 "motif_rows": int(sum(p[6] for p in per)), "planted": int(sum(p[7] for p in per)),
                                 "planted_recovered": int(sum(p[8] for p in per)),
                                 "gpu_busy_over_wall": max(p[3] for p in per) / max(p[0] for p in per),
                                 # rounds = scoring batches (search + post-processing); the search's lock-step loop ran search_iterations times;
                                 # children whose counts came with their parent's window reply (speculation on the device) / asked for after all
                                 "search_iterations": t.get("search_iterations"), "speculation_hits": t.get("speculation_hits"),
                                 "speculation_misses": t.get("speculation_misses"),
                                 "gpu_busy_covers": "every device phase of the run (pre-filter kernels, window gathers and batches, background counts, scoring launches), HIP events on the ctx stream",
                                 "not_in_wall_s": {"generate_s": t.get("generate_s"), "allocator_prewarm_s": t.get("allocator_prewarm_s")},
                                 "timings_rank0": t}
                e2e_result["gpu_busy_union_s"] = t.get("gpu_busy_union_s")      # (gpu_busy_s sums the phases; the two flights of the search overlap a little)
            if world == 1 and args.e2e_lanes > 1:
                # the same run with the bins dealt to lanes that share the device (one engine + one host thread each): what a lane's
                # search leaves idle between its small launches, another lane's pre-filters and gathers use
                row_key = lambda r: (r.reference, r.motif, r.mod_type, r.mod_position, r.n_mod, r.n_nomod, r.score)
                l_rows, lt = run_e2e(mg, local_rank, device, None, lanes=args.e2e_lanes)
                single = {k: e2e_result[k] for k in ("wall_s", "search_s", "upload_filter_s", "gpu_busy_s", "gpu_busy_union_s", "gpu_busy_over_wall")}
                single["wall_s_is"] = "upload_filter_s + search_s of the one lane (the figure of the earlier rounds); its whole call took run_call_s"
                single["run_call_s"] = t.get("run_call_s")
                e2e_result["single_lane"] = single
                e2e_result.update({
                    "lanes": args.e2e_lanes, "wall_s": lt["wall_s"], "gpu_busy_s": lt["gpu_busy_union_s"], "gpu_busy_over_wall": lt["gpu_busy_union_s"] / lt["wall_s"],
                    "gpu_busy_sum_of_phases_s": lt["gpu_busy_s"], "lanes_rows_equal_single_lane": [row_key(r) for r in l_rows] == [row_key(r) for r in e_rows],
                    "wall_s_is": f"first lane's start to the last lane's end, {args.e2e_lanes} lanes (threads) on one GPU, everything between included; "
                                 "gpu_busy_s = union of all lanes' device phases on the device's clock (nm_timing_intervals)",
                    "lane_pipeline_s": [x["upload_filter_s"] + x["search_s"] for x in lt["lanes"]],
                    "not_in_wall_s": {"generate_s": lt.get("generate_s"), "allocator_prewarm_s": lt.get("allocator_prewarm_s")}})
                del e2e_result["search_s"], e2e_result["upload_filter_s"], e2e_result["gpu_busy_union_s"]
                e2e_result["timings_lanes"] = lt["lanes"]
    except Exception as exc:                    # an extra must not take the headline line down with it (one rank only:
        if world > 1:                           #  with several ranks a lone survivor would hang in the next collective)
            raise
        log(f"extra 'e2e' failed: {exc!r}")
        extra_errors["e2e"] = repr(exc)

    if os.environ.get("NM_BENCH_EMPTY_CACHE"):
        torch.cuda.empty_cache()
    t0 = time.perf_counter()
    eng = ScanEngine(local_rank)
    rows = synth_device.load_engine_from_device(eng, mg, device, contigs=None if ((world == 1 and args.as_rank_of <= 1) or weak) else mine, progress=log)
    # torch's work of this script runs on one explicit side stream; when torch.distributed carries the count tables the
    # engine's launches go there too (below) (the legacy default stream has the handle 0, which nm_set_stream reads as
    # "use the ctx's own stream": never hand it that)
    side = torch.cuda.Stream(device)
    assert side.cuda_stream != 0
    torch.cuda.set_stream(side)
    st = eng.stats()
    log(f"resident: {st['total_bp']:,} bp ({st['padded_bp']:,} padded), rows {rows}, setup {time.perf_counter() - t0:.1f}s")

    all_bins = sorted(set(mg.bin_names))
    bin_bp, my_bin_bp = {}, {}
    for i, b in enumerate(mg.bin_names):
        bin_bp[b] = bin_bp.get(b, 0) + int(mg.lengths[i])
    for i in mine:
        my_bin_bp[mg.bin_names[i]] = my_bin_bp.get(mg.bin_names[i], 0) + int(mg.lengths[i])

    base = build_candidates(mg, "cfg5" if args.workload == "cfg5_all" else args.workload, args.candidates, args.per_group)

    def expand_all_bins(batch):
        """cfg5_all: every candidate of the table against EVERY bin — the flat SoA arrays are tiled, the mask bytes shared."""
        from nanomotif_amd.engine import CandidateBatch
        nb = len(all_bins)
        return CandidateBatch(np.repeat(np.arange(nb, dtype=np.uint32), len(batch)), np.tile(batch.slots, nb), np.tile(batch.lens, nb),
                              np.tile(batch.modpos, nb), np.tile(batch.offsets, nb), batch.masks)

    if args.workload == "cfg5_all":
        batch = expand_all_bins(eng.make_batch(base))
        n_cand = len(batch)
        sites_per_step = 2 * args.total_bp * len(base)
        groups = {(b, mt) for b in all_bins for mt in mg.spec.mod_types}
        cands = None
    else:
        cands = base
        batch = eng.make_batch(cands)      # the step's input: the candidate table in the C-ABI's flat SoA form
        n_cand = len(cands)
        sites_per_step = sum(2 * bin_bp[b] for _, _, b in cands)
        groups = {(b, mt) for _, mt, b in cands}
    algo_bytes_rank = ALGO_BYTES_PER_BP_STEP * sum(my_bin_bp.get(b, 0) for b, _ in groups) + 16 * n_cand

    use_native = reduce_counts and (args.allreduce == "native" or (args.allreduce == "auto" and args.dist_backend == "nccl"
                                                                  and hasattr(eng, "comm_init")))
    coll = Collective("native" if use_native else "torch", eng, world, rank, device, args.dist_backend) if reduce_counts else None

    if coll is not None and coll.kind == "torch":
        eng.use_stream(side.cuda_stream)       # torch.distributed orders its collectives on torch's current stream only
    # (otherwise the engine keeps its own streams: a second scoring lane beside a torch stream shared a hardware queue with
    #  it on this runtime and did not overlap, profiles/r2/lanes_queue_ab.txt)
    # a ring of count tables: the all-reduce of step k (communication stream) overlaps the scoring launches of the next steps
    N_TABLES = 4
    counts = [torch.zeros((n_cand, 2), dtype=torch.int64, device=device) for _ in range(N_TABLES)]
    step_no = [0]
    host_s = [0.0]

    def step():
        # inside the C ABI, every call: drop candidates of bins this rank does not hold, sort by (mod type, bin), ship the
        # records (pinned staging ring, copy stream), compile every motif to its constraint program on the device, zero the
        # counters, one scoring launch (async); then the all-reduce of this table is started on the communication stream
        t_in = time.perf_counter()
        i = step_no[0] % N_TABLES
        step_no[0] += 1
        if coll:
            coll.wait(i)                                    # table i is free again (its all-reduce of step k-4 finished)
        eng.score_into_device(batch, counts[i].data_ptr())
        if coll:
            coll.start(i, counts[i])
        host_s[0] += time.perf_counter() - t_in

    def drain():
        if coll:
            coll.drain()

    def timed_region(lanes):
        """W warm-up steps, then exactly K timed steps between barrier + synchronize on both sides.  lanes = 1: every
        launch in strict stream order (per-launch HIP events = the kernel alone on the device: the roofline's duration);
        lanes = 2: consecutive steps alternate between two scoring streams of the C ABI (nm_set_score_lanes) and overlap
        on the device — the counters of a step are the same, only when they are complete moves."""
        eng.set_score_lanes(lanes)
        for _ in range(args.warmup):
            step()
        drain()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        eng.timing_reset(True)
        torch.cuda.synchronize(device)
        host_s[0] = 0.0
        t_start = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        el_local = time.perf_counter() - t_start
        k_total, n_launch = eng.timing_total()
        eng.timing_reset(False)
        eng.set_score_lanes(1)
        k_local = k_total / max(n_launch, 1)
        el, k = allmax([el_local, k_local])
        # the table the LAST timed step of this pass left (all-reduced at N > 1): what the verification below looks at
        table = counts[(step_no[0] - 1) % N_TABLES].cpu().numpy().copy()
        return {"lanes": lanes, "elapsed": el, "elapsed_local": el_local, "kernel_ms": k, "kernel_ms_local": k_local,
                "host_ms": host_s[0] / args.steps * 1e3, "table": table}

    if args.cooldown > 0:
        torch.cuda.synchronize(device)
        time.sleep(args.cooldown)
    # Which pass is the headline: with one GPU the strict-order pass (its per-launch events are the roofline's kernel
    # duration and agree with a rocprofv3 trace of this command); with several GPUs the two-lane pass — a shard's kernel
    # is so short (~0.08 ms at N = 8) that the ~20 us between two dependent launches and the draining tail of every launch
    # are a quarter of the step, and consecutive steps are independent.  The other pass is reported next to it.
    lanes_ok = coll is None or coll.kind in ("native", "none")     # torch.distributed orders itself on torch's stream only
    want = args.lanes if args.lanes else (2 if world > 1 else 1)
    if want == 2 and not lanes_ok:
        log("two scoring lanes need the C ABI's own all-reduce (nm_allreduce_counts_async): staying in strict order")
        want = 1
    for _ in range(args.prewarm):
        step()
    drain()
    strict = timed_region(1)
    piped = timed_region(2) if (lanes_ok and args.lanes != 1 and (want == 2 or "two_lanes" in extras or args.lanes == 2)) else None
    lanes_agree = None if piped is None else bool(np.array_equal(strict["table"], piped["table"]))
    if lanes_agree is False:
        log("COUNT MISMATCH between the strict-order and the two-lane pass: the strict-order pass is the headline")
    head = piped if (want == 2 and piped and lanes_agree) else strict
    elapsed, elapsed_local = head["elapsed"], head["elapsed_local"]
    kernel_ms, kernel_ms_local = strict["kernel_ms"], strict["kernel_ms_local"]
    host_s[0] = head["host_ms"] * args.steps * 1e-3
    final = head["table"]

    # ---- the line proves its own counts (untimed).  (1) independent sum: every rank scores its shard once more in strict
    # order into a fresh table, the tables travel by all_gather (torch.distributed, not the C ABI's communicator) and are
    # summed on the host; (2) rank 0 loads ALL contigs of a few sampled bins into a second engine — the single-shard, N = 1
    # computation of those bins — and compares their rows; (3) the N = 1 checksum of this configuration, when it is known.
    verification = {"lanes_agree": lanes_agree}
    if reduce_counts:
        loc = torch.zeros((n_cand, 2), dtype=torch.int64, device=device)
        eng.score_into_device(batch, loc.data_ptr())
        eng.sync()
        torch.cuda.synchronize(device)
        if world > 1:
            src = loc if args.dist_backend == "nccl" else loc.cpu()
            parts = [torch.zeros_like(src) for _ in range(world)]
            dist.all_gather(parts, src)
            indep = np.sum([p_.cpu().numpy() for p_ in parts], axis=0)
        else:
            indep = loc.cpu().numpy()
        verification["allreduced_equals_independent_sum"] = bool(np.array_equal(indep, final))
        if rank == 0 and cands is not None and (world > 1 or args.as_rank_of > 1):
            bins_sorted = sorted(set(mg.bin_names))
            sample_bins = list(dict.fromkeys(bins_sorted[(k * 37) % len(bins_sorted)] for k in range(4)))
            idx = [i for i, b in enumerate(mg.bin_names) if b in set(sample_bins)]
            eng1 = ScanEngine(local_rank)
            synth_device.load_engine_from_device(eng1, mg, device, contigs=idx)
            whole = eng1.score(eng1.make_batch(cands))
            eng1.close()
            rows_ = [k for k, c_ in enumerate(cands) if c_[2] in set(sample_bins)]
            want_rows = whole[rows_]
            got_rows = (final if world > 1 else indep)[rows_]      # (--as-rank-of: one shard only, nothing to compare but the path)
            verification["bin_sample"] = {"bins": len(sample_bins), "candidates": len(rows_),
                                          "equal_to_single_shard": bool(np.array_equal(want_rows, got_rows)) if world > 1 else None}
        del loc
    known = N1_CHECKSUMS.get((args.workload, args.total_bp, args.contigs, args.bins, args.candidates))
    if known is not None and not weak and args.as_rank_of <= 1 and args.workload == "cfg5":
        verification["n1_checksum_known"] = known
        verification["equals_n1_checksum"] = checksum(final) == known

    # the collective alone: K all-reduces of the table back to back, slowest rank
    allreduce_ms, allreduce_ms_local = None, None
    if coll:
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            coll.wait(k & 1)
            coll.start(k & 1, counts[k & 1])
        coll.drain()
        torch.cuda.synchronize(device)
        allreduce_ms_local = (time.perf_counter() - t0) / args.steps * 1e3
        allreduce_ms = allmax([allreduce_ms_local])[0]
        # (the tables are multiples of the counts now: `final` and the verification tables were read before)
    info = eng.comm_info() if (coll and coll.kind == "native") else {"world": 0, "rank": -1, "device": -1, "rccl_version": 0}
    per_rank = gather([elapsed_local / args.steps * 1e3, kernel_ms_local, host_s[0] / args.steps * 1e3, float(len(mine)),
                       float(sum(my_bin_bp.values())), algo_bytes_rank, allreduce_ms_local if coll else -1.0,
                       float(info["world"]), float(info["rank"]), float(info["device"])])

    # ---- the same engine in its HBM-bound regime: one lock-step greedy round (2 sibling children per (bin, mod type),
    # the shape MotifSearcher.run submits), kernel time from HIP events; reported next to the main roofline
    hbm_round = None
    try:
        if args.workload == "cfg5" and args.hbm_round_steps > 0:
            g_cands = build_candidates(mg, "greedy", 0, 2)
            g_batch = eng.make_batch(g_cands)
            g_counts = torch.zeros((len(g_cands), 2), dtype=torch.int64, device=device)
            g_ms, _ = time_launches(eng, g_batch, g_counts.data_ptr(), args.hbm_round_steps, device)
            g_groups = {(b, mt) for _, mt, b in g_cands}
            g_bytes = ALGO_BYTES_PER_BP_STEP * sum(my_bin_bp.get(b, 0) for b, _ in g_groups) + 16 * len(g_cands)
            g_ms = allmax([g_ms])[0]
            tr = load_traffic("greedy", args.total_bp, len(g_cands)) if world == 1 else None
            # frac = ALGORITHMIC bytes / live kernel time / 8 TB/s, like the main roofline (round-4 advisor: a fraction must not mix a
            # recorded counter value of another build with a live time).  What the DRAM counters saw (the fused two-slot launch reads
            # the sequence planes once for both mod types: 0.83 GB, not the 1.0 GB that 0.5 B/bp/slot charges) stands beside it under
            # its own keys, with the sha of the kernel sources it was measured on, and only while that sha is the current one.
            traffic_bytes = tr["hbm_bytes_per_launch"] if tr and not tr.get("stale") else None
            hbm_round = {"workload": f"greedy round: {len(g_cands)} candidates = 2 sibling children per (bin, mod type)",
                         "bound": "hbm", "kernel_ms": g_ms, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "achieved": g_bytes / (g_ms * 1e-3) / 1e9,
                         "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_is": "algorithmic bytes / live kernel time / peak",
                         "counter_achieved": traffic_bytes / (g_ms * 1e-3) / 1e9 if traffic_bytes else None,
                         "counter_frac": traffic_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic_bytes else None,
                         "counter_entry_sha16": tr.get("kernel_source_sha16") if tr else None, "counter_entry_stale": bool(tr.get("stale")) if tr else None,
                         "algorithmic_achieved": g_bytes / (g_ms * 1e-3) / 1e9, "algorithmic_frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": g_bytes, "launches": args.hbm_round_steps,
                         "traffic": traffic_bytes,
                         # the fused two-slot launch reads the sequence planes once for both mod types: real DRAM bytes are
                         # below the algorithmic 0.5 B/bp/slot; this is the fraction of the device's streaming rate they reach
                         "traffic_rate_GBs": traffic_bytes / (g_ms * 1e-3) / 1e9 if traffic_bytes else None,
                         "traffic_frac_of_spec": traffic_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic_bytes else None,
                         "traffic_frac_of_streaming": traffic_bytes / (g_ms * 1e-3) / 1e9 / HBM_STREAM_GBS if traffic_bytes else None,
                         "streaming_peak_GBs": HBM_STREAM_GBS,
                         "motif_sites_per_s": sum(2 * bin_bp[b] for _, _, b in g_cands) / (g_ms * 1e-3)}
    except Exception as exc:                    # an extra must not take the headline line down with it (one rank only:
        if world > 1:                           #  with several ranks a lone survivor would hang in the next collective)
            raise
        log(f"extra 'roofline_hbm_bound_round' failed: {exc!r}")
        extra_errors["roofline_hbm_bound_round"] = repr(exc)

    result = None
    tr = load_traffic(args.workload, args.total_bp, n_cand) if world == 1 else None
    if rank == 0:
        nw = world if weak else 1
        value = sites_per_step * nw * args.steps / elapsed
        achieved = algo_bytes_rank / (kernel_ms * 1e-3) / 1e9
        if weak:
            workload = (f"{args.workload} (weak scaling): {world} x ({n_cand} candidate motifs x {args.total_bp:,} bp metagenome, {args.contigs} contigs, "
                        f"{args.bins} bins, 6mA+5mC), both strands, no collective")
        else:
            per_bin = "every candidate x every bin" if args.workload == "cfg5_all" else f"{n_cand // max(len(all_bins), 1)} per bin"
            workload = (f"{args.workload}: {len(base)} candidate motifs ({per_bin}) x {args.total_bp:,} bp total metagenome ({args.contigs} contigs, "
                        f"{args.bins} bins, 6mA+5mC), both strands" + (f", contig-sharded over {world} GPUs, one all-reduce of the int64[{n_cand},2] count table per step" if world > 1 else ""))
        result = {
            "metric": "motif-sites scored/sec (1 Gbp synthetic metagenome)", "value": value, "unit": "motif-sites/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u32 bit-planes / int64 counts",
            "data": "synthetic",
            "config": {"workload": workload, "candidates": n_cand * nw, "total_bp": args.total_bp * nw,
                       "contigs": args.contigs * nw, "bins": args.bins * nw, "mod_types": ["a", "m"],
                       "sharding": ("whole metagenome per GPU, no collective" if weak or world == 1 else
                                    f"contigs over {world} GPUs, longest-first, bins kept whole when small (nanomotif_amd/shard.py); "
                                    f"count tables summed by {'nm_allreduce_counts (RCCL, C ABI)' if (coll and coll.kind == 'native') else 'torch.distributed ' + args.dist_backend}"),
                       "motif_sites_per_step": sites_per_step * nw},
            # achieved / peak / frac are the HBM roofline of the contract (algorithmic bytes over the kernel's duration);
            # `bound` names what REALLY limits this launch: a heavy batch (>~ 3 candidates per (bin, mod type)) is bound by
            # integer-VALU issue, not by HBM — its HBM fraction is also given as hbm_frac, the VALU side in roofline_valu
            "roofline": {"bound": "hbm" if args.workload == "greedy" else "valu-int",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS,
                         "traffic": tr["hbm_bytes_per_launch"] if tr else None,
                         "traffic_source": tr["source"] if tr else "not collected for this configuration (rocprofv3 --pmc runs: profiles/)",
                         "traffic_measured_by": "rocprofv3 --pmc run of this command kept in profiles/ (replayed here, not collected by this run)" if tr else None,
                         "traffic_stale": bool(tr["stale"]) if tr else None,
                         "kernel_source_sha16": kernel_source_sha16(),
                         "kernel": "score_kernel (narrow, compact)", "kernel_ms": kernel_ms,
                         "algorithmic_bytes_per_launch": algo_bytes_rank,
                         "note": "0.5 B/bp per (bin, mod type) step + 16 B per candidate; slowest rank at N>1"},
            # of the strict-order pass (with two lanes consecutive launches overlap: a step is then SHORTER than one launch)
            "kernel_share_of_step": kernel_ms / (strict["elapsed"] / args.steps * 1e3),
            "per_rank": [dict(zip(["ms_per_step", "kernel_ms", "host_call_ms_per_step", "contigs", "bp", "algorithmic_bytes", "allreduce_ms",
                                   "rccl_world", "rccl_rank", "rccl_device"], p)) for p in per_rank],
            "allreduce_ms": allreduce_ms,
            "prewarm_steps": args.prewarm,
            "pipelining": {"scoring_lanes": head["lanes"], "staging_ring": 4, "count_tables": N_TABLES,
                           "note": "lanes = 2: consecutive (independent) steps alternate between two scoring streams of the C ABI "
                                   "and overlap on the device; roofline.kernel_ms is always the strict-order pass of this run"},
            "counts_checksum": checksum(final),
            "verification": verification,
        }
        if coll:
            # what the communicator reports about itself (ncclCommCount / ncclCommUserRank through nm_comm_info), every rank
            worlds = sorted({int(p[7]) for p in per_rank})
            result["rccl"] = {"world": worlds[0] if len(worlds) == 1 else worlds, "native": coll.kind == "native",
                              "all_ranks_report_world": worlds == [world] if coll.kind == "native" else None,
                              "version_code": info["rccl_version"],
                              "carrier": "nm_allreduce_counts_async (C ABI, RCCL through dlopen)" if coll.kind == "native"
                                         else f"torch.distributed {args.dist_backend}"}
        if reduce_counts or "equals_n1_checksum" in verification:
            ok_parts = [verification.get("allreduced_equals_independent_sum"), verification.get("equals_n1_checksum"),
                        (verification.get("bin_sample") or {}).get("equal_to_single_shard")]
            result["checksum_matches_n1"] = all(x for x in ok_parts if x is not None) if any(x is not None for x in ok_parts) else None
        other = strict if head is piped else piped
        if other:
            result["strict_order" if head is piped else "two_lanes"] = {
                "scoring_lanes": other["lanes"], "value": sites_per_step * nw * args.steps / other["elapsed"], "unit": "motif-sites/s",
                "ms_per_step": other["elapsed"] / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup}
        if hbm_round:
            result["roofline_hbm_bound_round"] = hbm_round
        if tr and tr.get("sq_insts_valu_per_launch"):
            # second roofline: the kernel is integer-VALU-issue bound once a (bin, mod type) carries more than ~2
            # candidates; instruction count from rocprofv3 (SQ_INSTS_VALU, profiles/), peak from tools/valu_peak.hip
            rate = tr["sq_insts_valu_per_launch"] / (kernel_ms * 1e-3)
            result["roofline_valu"] = {"bound": "valu-int", "achieved": rate, "unit": "wave64 integer instr/s",
                                       # the honest denominator: the chip's VALU issue rate (one wave64 op per 2 cycles per SIMD)
                                       "peak": VALU_ISSUE_PEAK_WAVE_INSTR_S, "frac": rate / VALU_ISSUE_PEAK_WAVE_INSTR_S,
                                       "peak_source": "MI355X_MICROARCH.md: 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 VALU op",
                                       # and the rate THIS instruction mix can issue at (half its ops are the half-rate v_alignbit_b32)
                                       "peak_measured_mix": VALU_PEAK_WAVE_INSTR_S, "frac_of_measured_mix": rate / VALU_PEAK_WAVE_INSTR_S,
                                       "peak_measured_mix_source": "tools/valu_peak.hip, profiles/r1/valu_peak.txt (alignbit + and stream)",
                                       "sq_insts_valu_per_launch": tr["sq_insts_valu_per_launch"],
                                       "sq_insts_salu_per_launch": tr.get("sq_insts_salu_per_launch"),
                                       "salu_per_valu": (tr["sq_insts_salu_per_launch"] / tr["sq_insts_valu_per_launch"]) if tr.get("sq_insts_salu_per_launch") else None,
                                       "sq_inst_cycles_salu_per_launch": tr.get("sq_inst_cycles_salu_per_launch"),
                                       "sq_busy_cycles_per_launch": tr.get("sq_busy_cycles_per_launch"),
                                       "counters_stale": bool(tr["stale"])}

    # ---- extra: every candidate x every bin (SURVEY §8(d) cfg 5 as 2e13 motif-sites per step), N = 1
    try:
        if "cfg5_all" in extras:
            a_batch = expand_all_bins(batch)
            a_counts = torch.zeros((len(a_batch), 2), dtype=torch.int64, device=device)
            a_ms, a_wall_ms = time_launches(eng, a_batch, a_counts.data_ptr(), 3, device)
            a_sites = 2 * args.total_bp * len(base)
            a_bytes = ALGO_BYTES_PER_BP_STEP * args.total_bp * len(mg.spec.mod_types) + 16 * len(a_batch)
            # every bin must give each candidate the same counts it got in the 20-per-bin table where they coincide
            a_host = a_counts.cpu().numpy().reshape(len(all_bins), len(base), 2)
            own = np.array([all_bins.index(c[2]) for c in base])
            same = bool(np.array_equal(a_host[own, np.arange(len(base))], final))
            tr_a = load_traffic("cfg5_all", args.total_bp, len(a_batch))
            result["cfg5_all"] = {"workload": f"{len(base)} candidates x all {len(all_bins)} bins = {len(a_batch)} (candidate, bin) pairs, {a_sites:.3g} motif-sites per step",
                                  "value": a_sites / (a_wall_ms * 1e-3), "unit": "motif-sites/s", "ms_per_step": a_wall_ms, "kernel_ms": a_ms,
                                  "roofline_hbm_frac": a_bytes / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "roofline_valu_frac": (tr_a["sq_insts_valu_per_launch"] / (a_ms * 1e-3) / VALU_PEAK_WAVE_INSTR_S) if tr_a and tr_a.get("sq_insts_valu_per_launch") else None,
                                  "agrees_with_per_bin_table": same}
            del a_counts, a_batch
            torch.cuda.empty_cache()
    except Exception as exc:                    # an extra must not take the headline line down with it (one rank only:
        if world > 1:                           #  with several ranks a lone survivor would hang in the next collective)
            raise
        log(f"extra 'cfg5_all' failed: {exc!r}")
        extra_errors["cfg5_all"] = repr(exc)

    # ---- CPU baseline (rank 0, N = 1 only): the oracle on a bounded sample of the same workload, and parity
    try:
        if rank == 0 and world == 1 and args.cpu_bins != 0 and cands is not None:
            cpu_baseline(result, final, cands, mg, spec_kw, args)
    except Exception as exc:                    # an extra must not take the headline line down with it (one rank only:
        if world > 1:                           #  with several ranks a lone survivor would hang in the next collective)
            raise
        log(f"extra 'cpu_baseline' failed: {exc!r}")
        extra_errors["cpu_baseline"] = repr(exc)

    eng.close()
    del counts
    torch.cuda.empty_cache()

    # ---- extra (N > 1): weak scaling — every rank its own metagenome (seed 1 + rank) and candidate table, no collective
    if "weak" in extras:
        mg_w = synth.make_metagenome(synth.SynthSpec(**dict(spec_kw, seed=1 + rank)))
        eng_w = ScanEngine(local_rank)
        synth_device.load_engine_from_device(eng_w, mg_w, device)
        eng_w.use_stream(side.cuda_stream)
        c_w = build_candidates(mg_w, "cfg5", args.candidates, 2)
        b_w = eng_w.make_batch(c_w)
        o_w = torch.zeros((len(c_w), 2), dtype=torch.int64, device=device)
        bb = {}
        for i, b in enumerate(mg_w.bin_names):
            bb[b] = bb.get(b, 0) + int(mg_w.lengths[i])
        s_w = sum(2 * bb[b] for _, _, b in c_w)
        for _ in range(args.warmup):
            eng_w.score_into_device(b_w, o_w.data_ptr())
        torch.cuda.synchronize(device)
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng_w.score_into_device(b_w, o_w.data_ptr())
        torch.cuda.synchronize(device)
        dist.barrier()
        el = allmax([time.perf_counter() - t0])[0]
        tot = gather([float(s_w)])
        if rank == 0:
            result["weak_scaling"] = {"what": f"{world} x (own {args.total_bp:,} bp metagenome, seed 1 + rank, own {len(c_w)}-candidate table), no collective",
                                      "value": sum(t[0] for t in tot) * args.steps / el, "unit": "motif-sites/s", "ms_per_step": el / args.steps * 1e3}
        eng_w.close()

    try:
        if "files" in extras and rank == 0:
            result["e2e_files"] = run_files_extra(device, log)
    except Exception as exc:
        log(f"extra 'files' failed: {exc!r}")
        extra_errors["files"] = repr(exc)
    try:
        if "cli" in extras and rank == 0:
            result["cli"] = run_cli_extra(device, log, args.cli_bp)
    except Exception as exc:
        log(f"extra 'cli' failed: {exc!r}")
        extra_errors["cli"] = repr(exc)
    try:
        if "cli1g" in extras and rank == 0:
            result["cli1g"] = run_cli1g_extra(device, log, args.cli1g_bp, gz_level=args.cli1g_gz_level)
    except Exception as exc:
        log(f"extra 'cli1g' failed: {exc!r}")
        extra_errors["cli1g"] = repr(exc)

    failed = 0
    if rank == 0:
        if e2e_result is not None:
            result["e2e"] = e2e_result
        if extra_errors:
            result["extra_errors"] = extra_errors
        emit(result)
        if result.get("checksum_matches_n1") is False or (result.get("parity") or {}).get("mismatches"):
            log("COUNT VERIFICATION FAILED: " + json.dumps(result.get("verification")))
            failed = 1
    if world > 1:
        failed = int(allmax([failed])[0])
        dist.barrier()
        dist.destroy_process_group()
    if failed:
        sys.exit(3)                      # a line whose counts do not verify must not pass for a measurement


if __name__ == "__main__":
    main()

"""``nanomotif motif_discovery`` on the MI355X engine (reference: nanomotif/main.py:18-104, 298-321).

Single GPU:   python -m nanomotif_amd motif_discovery ASSEMBLY PILEUP -c CONTIG_BIN --out OUT
Several GPUs: python -m torch.distributed.run --nproc-per-node N -m nanomotif_amd motif_discovery ...
              (contigs are sharded over the ranks, count tables are all-reduced over RCCL, rank 0 writes the output)
"""
from __future__ import annotations

import json
import logging as log
import os
import random
import sys
import time
import warnings
from pathlib import Path

import numpy as np

from . import fasta, pileup as pileup_mod, postprocess
from .argparser import __version__, create_parser
from .find_motifs_bin import FilteredPileup, ProcessorConfig, allreduce_counts, discover, engine_scorer, use_native_allreduce
from .engine import MAX_DEVICE_WINDOW_WIDTH
from .motif import MOD_TYPE_TO_CANONICAL
from .shard import assign_bins, assign_contigs

HEADER = "\t".join(postprocess.HEADER) + "\n"


def set_seed(seed=42):
    random.seed(seed)
    np.random.seed(seed)


def shared_setup(args, working_dir, rank=0):
    """main.py:18-43: output directory, logs/, args.<command>.json, seeds."""
    if not os.path.exists(args.out):
        os.makedirs(args.out, exist_ok=True)
    elif rank == 0:
        log.warning(f"Output directory {args.out} already exists")
    log_dir = working_dir + "/logs"
    Path(log_dir).mkdir(parents=True, exist_ok=True)
    handlers = [log.StreamHandler(sys.stdout)]
    if rank == 0:
        handlers.append(log.FileHandler(log_dir + f"/{args.command}.main.log"))
    log.basicConfig(level=log.DEBUG if args.verbose else log.INFO, handlers=handlers, force=True,
                    format="%(asctime)s - %(levelname)s - %(message)s")
    warnings.filterwarnings("ignore")
    log.info(f"nanomotif (MI355X build) version: {__version__}")
    if rank == 0:
        with open(working_dir + f"/args.{args.command}.json", "w") as f:
            json.dump(vars(args), f, indent=2)
    set_seed(args.seed)


TIMINGS = {}          # seconds per phase of the last find_motifs_bin call in this process (written to OUT/logs/timings.*.json)


def find_motifs_bin(args):
    """main.py:46-104."""
    TIMINGS.clear()
    t_phase = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        TIMINGS[name] = TIMINGS.get(name, 0.0) + now - t_phase[0]
        t_phase[0] = now
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = args.device if args.device is not None else local_rank
    dist = None
    if world > 1:
        # torch is plumbing for the multi-rank run only (process group, RCCL); a single-GPU run never imports it
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise RuntimeError("nanomotif_amd needs an AMD GPU (MI355X); there is no CPU fallback")
        torch.cuda.set_device(device)
        if not dist.is_initialized():
            # RCCL unless NANOMOTIF_DIST_BACKEND=gloo (debugging aid: lets several ranks share one GPU with --device)
            backend = os.environ.get("NANOMOTIF_DIST_BACKEND", "nccl")
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", device))
            else:
                dist.init_process_group(backend)
    from ._lib import NmScanError
    from . import _lib as _lib_codes
    from .engine import ScanEngine
    # The HIP runtime takes 0.1-0.3 s to come up in a fresh process: it does so on a side thread while this one reads the
    # contig-bin table and parses the assembly (native code, the interpreter lock is released).  Fails loudly without a
    # GPU: there is no CPU fallback.
    import threading
    started = {}

    def start_engine():
        try:
            # large device blocks stay with the process when the library frees them (nm_block_cache: memory another process used is
            # scrubbed by the driver on its way back in — the pre-filters' state planes waited 0.19 s for that at 1 Gbp — and hipFree
            # synchronises the device); NANOMOTIF_BLOCK_CACHE_GB=0 turns it off
            from . import _lib as _l
            early = None
            if _l.early_engine_thread is not None:       # (__main__.py: the context may exist already, made beside the imports)
                _l.early_engine_thread.join()
                _l.early_engine_thread, early, _l.early_engine = None, _l.early_engine, None
                if _l.early_pin_thread is not None:
                    _l.early_pin_thread.join()
                    _l.early_pin_thread = None
            cache_gb = float(os.environ.get("NANOMOTIF_BLOCK_CACHE_GB", "16"))
            if early is not None and early[0] != device:
                _l.load().nm_ctx_destroy(early[1])       # (another --device than the command line showed at a glance)
                early = None
            if cache_gb > 0 and not (early is not None and early[2]):
                _l.use_block_cache(int(cache_gb * (1 << 30)))
            started["eng"] = ScanEngine(device, ctx=early[1]) if early is not None else ScanEngine(device)
            TIMINGS["engine_context_made_beside_the_imports"] = early is not None
        except BaseException as e:               # re-raised on the main thread below
            started["error"] = e
    starter = threading.Thread(target=start_engine, name="nm-engine-start")
    starter.start()

    def engine():
        starter.join()
        if "error" in started:
            e = started["error"]
            if isinstance(e, NmScanError):
                raise RuntimeError(f"nanomotif_amd needs an AMD GPU (MI355X); there is no CPU fallback ({e})") from e
            raise e
        return started["eng"]

    log.info("Starting nanomotif motif finder")
    bin_contig = fasta.generate_contig_bin(args)
    if not bin_contig:
        log.error("No bin contig mapping found")
        eng = engine()
        eng.close()
        return None
    # A bgzip pileup: the host-only half of its indexed parse (tabix index, the walk over the BGZF blocks: half a second at 1 Gbp)
    # starts NOW on a thread, for the contigs the bin table names — beside the HIP runtime coming up and the assembly being parsed.
    # It is used when the assembly turns out to hold them all (else the plan is dropped and made again for the ones it holds).
    plan_box = {}
    if str(args.pileup).endswith(".gz") and os.path.exists(str(args.pileup) + ".tbi") and os.environ.get("NANOMOTIF_HOST_PARSER") != "1" \
            and not any(fasta.ALIAS_SEP in c for c in bin_contig) and os.environ.get("NANOMOTIF_NO_PREPLAN") != "1":
        guess = list(dict.fromkeys(fasta.original_name(c) for c in bin_contig))

        def make_plan():
            try:
                plan_box["plan"] = pileup_mod.BedPlan(str(args.pileup), str(args.pileup) + ".tbi", guess, threads=max(args.threads, 0) if args.threads > 1 else 0)
            except BaseException as e:           # (the regular path will meet the same problem and report it)
                plan_box["error"] = e
        plan_box["thread"] = threading.Thread(target=make_plan, name="nm-bed-plan")
        plan_box["thread"].start()
    def drop_plan():
        """An error path between the plan thread's start and its use: the thread is waited for (it is not a daemon: the process would
        wait for its half-second walk at exit anyway) and the plan's mapping of the pileup released now, not by a finaliser."""
        th = plan_box.pop("thread", None)
        if th is not None:
            th.join()
        made = plan_box.pop("plan", None)
        if made is not None:
            made.close()
        if "error" in plan_box:
            log.debug(f"the pre-planned indexed parse failed (the regular path reports the cause): {plan_box.pop('error')!r}")

    log.info("Loading assembly")
    # A plain-text assembly is parsed ON THE GPU (nm_fasta_parse_device: the host only moves the file through pinned slabs; the
    # bases never become a host array, the planes are packed from the parser's device buffer); a .gz assembly and
    # NANOMOTIF_HOST_FASTA=1 take the native host reader, which runs while the HIP runtime comes up.
    device_fasta = not str(args.assembly).endswith(".gz") and os.environ.get("NANOMOTIF_HOST_FASTA") != "1"
    try:
        if device_fasta:
            eng = engine()
            lap("engine_start_s")
            assembly = fasta.DeviceAssembly(eng, args.assembly, threads=max(args.threads, 0) if args.threads > 1 else 0)
            TIMINGS["assembly_reading_s"] = assembly.seconds_reading
        else:
            assembly = fasta.load_fasta(args.assembly)
        fasta.add_alias_sequences(assembly, bin_contig)      # a contig listed under several bins is a member of each
    except BaseException:
        starter.join()
        if "eng" in started:
            started["eng"].close()
        drop_plan()
        raise
    lap("assembly_s")
    TIMINGS["assembly_parser"] = "device" if device_fasta else "host"
    if not device_fasta:
        eng = engine()
        lap("engine_start_s")                    # what the assembly did not hide
    log.info("Identifying motifs")
    cfg = ProcessorConfig(assembly=assembly, pileup_path=args.pileup, bin_contig=bin_contig, threads=args.threads,
                          search_frame_size=args.search_frame_size, methylation_threshold_low=args.methylation_threshold_low,
                          methylation_threshold_high=args.methylation_threshold_high,
                          minimum_kl_divergence=args.minimum_kl_divergence, score_threshold=args.min_motif_score,
                          verbose=args.verbose, log_dir=args.out + "/logs", seed=args.seed, output_dir=args.out)
    bgzip = cfg.pileup_path.endswith(".gz")
    if bgzip and not os.path.exists(cfg.pileup_path + ".tbi"):
        drop_plan()
        raise FileNotFoundError(f"Tabix index for {cfg.pileup_path} not found.")     # find_motifs_bin.py:383-384
    t0 = time.perf_counter()
    # native reader, raw rows kept in native memory; a bgzip pileup is read through its tabix index: only the blocks
    # of the contigs that are in a bin and in the assembly (find_motifs_bin.py:233-246 fetches per bin)
    wanted = list(dict.fromkeys(fasta.original_name(c) for c in cfg.bin_contig if c in assembly)) if bgzip else None
    # the pileup is parsed ON THE GPU (nm_bed_parse_device: the host only moves the file through pinned slabs — the BGZF
    # blocks of a bgzip file are inflated into them by the copy threads, the tabix subset alike; every row equals the host
    # parser's bit for bit); a gzip stream that is not bgzip, a contig listed under several bins (its rows are needed twice)
    # and NANOMOTIF_HOST_PARSER=1 take the host parser
    table = None
    plan = None
    if "thread" in plan_box:
        plan_box.pop("thread").join()
        plan = plan_box.pop("plan", None)
        if "error" in plan_box:
            log.debug(f"the pre-planned indexed parse failed (the regular path reports the cause): {plan_box.pop('error')!r}")
        if plan is not None and (wanted is None or list(plan.contigs) != list(wanted)):
            plan.close()                         # the assembly lacks some of the binned contigs: plan again for the ones it holds
            plan = None
        if plan is not None:
            TIMINGS["pileup_plan_s_on_a_thread"] = plan.seconds
    if os.environ.get("NANOMOTIF_HOST_PARSER") != "1" and not any(fasta.ALIAS_SEP in c for c in cfg.bin_contig):
        try:
            try:
                table = pileup_mod.DevicePileup(eng, cfg.pileup_path, threads=max(args.threads, 0) if args.threads > 1 else 0,
                                                contigs=wanted, index_path=cfg.pileup_path + ".tbi" if bgzip else None, plan=plan)
            except BaseException:
                if plan is not None:             # a final error of the parser: the plan's mapping goes back now
                    plan.close()
                    plan = None
                raise
            how = (f", tabix-indexed: {table.bytes_inflated / 1e6:.1f} MB inflated for {len(wanted)} contigs" if table.indexed else "")
            log.info(f"pileup: {len(table):,} rows parsed on the device ({time.perf_counter() - t0:.1f}s, {table.seconds_reading:.1f}s of it "
                     f"moving the file{how})")
        except NmScanError as e:
            if e.code != _lib_codes.NM_EDECLINED:
                raise
            log.info(f"pileup: the device parser declined ({e}); using the host parser")
    if table is None:
        table = pileup_mod.NativePileup(cfg.pileup_path, contigs=wanted, index_path=cfg.pileup_path + ".tbi" if bgzip else None)
        how = (f", tabix-indexed: {table.bytes_inflated / 1e6:.1f} MB inflated for {len(wanted)} contigs" if table.indexed else "")
        log.info(f"pileup: {len(table):,} rows read ({time.perf_counter() - t0:.1f}s{how})")
    if plan is not None:
        # (closing the plan unmaps the pileup: 3.5 million page-table entries at 1 Gbp — off the critical path, on a thread of its own)
        threading.Thread(target=plan.close, name="nm-bed-plan-close", daemon=False).start()
    on_device = isinstance(table, pileup_mod.DevicePileup)
    lap("pileup_parse_s")
    TIMINGS["pileup_parser"] = "device" if on_device else "host"
    if on_device:
        TIMINGS.update(pileup_reading_s=table.seconds_reading, pileup_inflating_s=table.seconds_inflating, pileup_parsing_s=table.seconds_parsing,
                       pileup_in_parser_s=table.seconds)       # (the rest of pileup_parse_s: the tabix index, the walk over the BGZF blocks, the tables)
    TIMINGS["pileup_rows"] = len(table)

    # engine: this rank's contigs (all contigs that belong to a bin).  Several GPUs: whole bins per GPU when they
    # balance (independent searches, no collective until the rows are gathered), else the contigs of every bin are
    # sharded and the count tables all-reduced every round.
    names = [c for c in cfg.bin_contig if c in assembly]
    bin_order = list(dict.fromkeys(cfg.bin_contig.values()))         # task order of the reference (:152-171)
    by_bins = None
    if world > 1 and args.shard != "contigs":
        sizes = {}
        for c in names:
            sizes[cfg.bin_contig[c]] = sizes.get(cfg.bin_contig[c], 0) + fasta.assembly_length(assembly, c)
        by_bins = assign_bins(sizes, world, tolerance=0.15 if args.shard == "auto" else float("inf"))
    if by_bins is not None:
        my_bins = set(by_bins[rank])
        log.info(f"rank {rank}: {len(my_bins)} of {len(sizes)} bins (whole bins per GPU)")
        cfg.bin_contig = {c: b for c, b in cfg.bin_contig.items() if b in my_bins}
        names = [c for c in names if cfg.bin_contig.get(c) in my_bins]
        gather_world, world = world, 1                               # the data path below is a single-GPU run
        if not names:                                                # more GPUs than bins
            return _gather_rows(args, [], rank, gather_world, bin_order)
    else:
        gather_world = 1
    if world > 1 and dist.get_backend() == "nccl" and os.environ.get("NANOMOTIF_ALLREDUCE", "native") == "native":
        # the per-round count tables travel through the C ABI's own RCCL communicator (nm_allreduce_counts_host);
        # torch.distributed only carried the 128-byte id to the ranks.  Every rank must end up on the SAME path.  What is
        # guaranteed: a rank on which librccl does not load says so BEFORE anybody enters ncclCommInitRank (every rank makes
        # a unique id as a probe; MIN all-reduce), and a rank whose ncclCommInitRank RETURNS an error says so after it (second
        # MIN all-reduce) — the run then falls back to torch.distributed as a whole.  What no agreement can cover is a rank
        # that never arrives inside the collective init (it died, or hangs): the others would wait there for ever, so the
        # init runs under a watchdog that ends this process with a message (NANOMOTIF_COMM_TIMEOUT seconds, default 300)
        import threading
        import torch

        def comm_init_watched():
            limit = float(os.environ.get("NANOMOTIF_COMM_TIMEOUT", "300"))
            done = threading.Event()

            def watchdog():
                if not done.wait(limit):
                    sys.stderr.write(f"rank {rank}: nm_comm_init did not return within {limit:.0f} s (a rank missing from ncclCommInitRank?): "
                                     "giving up; NANOMOTIF_ALLREDUCE=torch takes torch.distributed instead\n")
                    sys.stderr.flush()
                    os._exit(3)
            threading.Thread(target=watchdog, name="nm-comm-watchdog", daemon=True).start()
            try:
                eng.comm_init(rank, world, uid[0])
            finally:
                done.set()

        def agreed(ok_here: int) -> bool:
            flag = torch.tensor([ok_here], dtype=torch.int32, device=torch.device("cuda", device))
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag[0]) == 1

        ok = 1
        uid = [None]
        try:
            probe = eng.comm_unique_id()            # EVERY rank: proves that librccl loads here before anybody waits in CommInitRank
            uid = [probe if rank == 0 else None]
        except NmScanError as e:
            ok = 0
            log.warning(f"rank {rank}: nm_comm_unique_id failed ({e})")
        if agreed(ok):
            dist.broadcast_object_list(uid, src=0)
            try:
                comm_init_watched()
            except NmScanError as e:
                ok = 0
                log.warning(f"rank {rank}: nm_comm_init failed ({e})")
        else:
            ok = 0
        if agreed(ok):
            use_native_allreduce(eng)
            log.info(f"rank {rank}: count tables all-reduced by nm_allreduce_counts (RCCL, {eng.comm_info()['world']} ranks)")
        else:
            log.warning(f"rank {rank}: the C ABI's communicator is not up on every rank: count tables go through torch.distributed")
    try:
        parts = assign_contigs([fasta.assembly_length(assembly, c) for c in names], world, bins=[cfg.bin_contig[c] for c in names])
        mine = [names[i] for i in parts[rank if gather_world == 1 else 0]]
        all_bins = sorted(set(cfg.bin_contig[c] for c in names))       # bin ids must be identical on every rank
        if device_fasta:
            eng.upload_assembly_fasta(assembly, mine, [cfg.bin_contig[c] for c in mine], bin_names=all_bins)
        else:
            eng.upload_assembly(mine, [assembly[c] for c in mine], [cfg.bin_contig[c] for c in mine], bin_names=all_bins)
        t_part = t_phase[0]                      # (the end of the phase before: the pileup parse)

        def part(name):                          # (what upload_filter_s is made of: TIMINGS["filters_<name>_s"])
            nonlocal t_part
            now = time.perf_counter()
            TIMINGS["filters_" + name + "_s"] = TIMINGS.get("filters_" + name + "_s", 0.0) + now - t_part
            t_part = now
        part("upload_assembly")
        # raw rows -> device: the three pre-filters, classification, confident-row list (rows of contigs that are in no
        # bin or on another rank are ignored: the reference joins with contig -> bin after filtering, find_motifs_bin.py:416)
        local_id = {c: i for i, c in enumerate(mine)}
        lut = np.array([local_id.get(n, 0xFFFFFFFF) for n in table.contig_names], dtype=np.uint32)
        # further placements of a contig listed under several bins: the contig's rows once more per placement
        file_id = {n: i for i, n in enumerate(table.contig_names)}
        placements = [(file_id[fasta.original_name(c)], local_id[c]) for c in mine
                      if fasta.ALIAS_SEP in c and fasta.original_name(c) in file_id]
        file_contig = table.file_contig_column().copy() if placements else None
        cols = None if on_device else table.ingest_columns(lut)          # views in the engine's types; refuses positions >= 4 Gbp
        labels = {i: (mt, MOD_TYPE_TO_CANONICAL[mt]) for i, mt in enumerate(pileup_mod.MOD_TYPES)}
        t0 = time.perf_counter()
        low, high = cfg.methylation_threshold_low, cfg.methylation_threshold_high
        # large pileups go to the device in parts of whole contigs (bounds the memory of the raw rows and filter scratch)
        part_rows = int(os.environ.get("NANOMOTIF_INGEST_PART_ROWS", 250_000_000))
        extra = []
        for fid, local in placements:
            sel = np.flatnonzero(file_contig == fid)
            extra.append(dict(contig=np.full(len(sel), local, np.uint32), **{k: cols[k][sel] for k in ("position", "mod_type", "strand", "fraction_mod", "nvalid_cov")}))
        part("tables")
        if on_device:
            res = eng.ingest_device_pileup(table, lut, labels, low=low, high=high, max_part_rows=part_rows)
        else:
            res = eng.ingest_pileup(cols["contig"], cols["position"], cols["mod_type"], cols["strand"], cols["fraction_mod"],
                                    cols["nvalid_cov"], labels, low=low, high=high, want_rows=False, max_part_rows=part_rows, extra_parts=extra)
        part("ingest")
        if 2 * cfg.padding + 1 > MAX_DEVICE_WINDOW_WIDTH:
            # a search frame beyond the device's window planes (default 40; nobody runs such frames): windows are extracted and
            # filtered on the host (search.HostWindowStore), candidates scored on the device (far-reaching ones by the plain kernel)
            log.info(f"search frame {cfg.search_frame_size}: windows of {2 * cfg.padding + 1} positions stay on the host")
            store, extractor = None, None
        else:
            store, extractor = device_window_pipeline(eng, {c: fasta.assembly_length(assembly, c) for c in names}, mine, cfg.padding, world)
        if device_fasta and extractor is not None:
            assembly.close()                     # the packed bases have served; (host windows would read contigs back from them)
        rows_part = eng.confident_rows() if extractor is None else tuple(np.zeros(0, dt) for dt in (np.uint32, np.uint32, np.uint8, np.int8))
        if (low, high) == (0.3, 0.7):
            for mt in pileup_mod.MOD_TYPES:
                eng.alias_label((mt, "merge"), mt)
        elif on_device:   # the merge stage always runs at 0.3 / 0.7 (find_motifs_bin.py:569, 1436): a second classification
            eng.ingest_device_pileup(table, lut, {i: ((mt, "merge"), MOD_TYPE_TO_CANONICAL[mt]) for i, mt in enumerate(pileup_mod.MOD_TYPES)},
                                     low=0.3, high=0.7, max_part_rows=part_rows)
        else:
            eng.ingest_pileup(cols["contig"], cols["position"], cols["mod_type"], cols["strand"], cols["fraction_mod"],
                              cols["nvalid_cov"], {i: ((mt, "merge"), MOD_TYPE_TO_CANONICAL[mt]) for i, mt in enumerate(pileup_mod.MOD_TYPES)},
                              low=0.3, high=0.7, want_rows=False, max_part_rows=part_rows, extra_parts=extra)
        log.info(f"pileup: {res['n_kept']:,} rows after the device-side filters ({time.perf_counter() - t0:.1f}s)")
        part("window_pipeline")
        del cols
        table.close()
        part("table_close")
        lap("upload_filter_s")
        part = FilteredPileup(mine, *rows_part, res["kept"])
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, part)
            filtered = FilteredPileup.merge(gathered)
        else:
            filtered = part
        if filtered.kept.sum() == 0:
            log.info("No pileup data after filtering, skipping")
            return _gather_rows(args, [], rank, gather_world, bin_order) if gather_world > 1 else None
        scorer = engine_scorer(eng, low, high, use_dist=world > 1)
        rows, scorer = discover(cfg, filtered, scorer, rank=0 if gather_world > 1 else rank, bgzip_order=bgzip,
                                window_store=store, extractor=extractor)
        if getattr(eng, "wide_scored", 0):
            log.info(f"{eng.wide_scored} candidates reaching further than 95 positions from the modified base scored by nm_score_batch_wide")
        lap("search_s")
        TIMINGS.update({"search_" + k: v for k, v in getattr(scorer, "timings", {}).items()})
        out = _gather_rows(args, rows, rank, gather_world, bin_order)
        lap("write_s")
        return out
    finally:
        # also on the early returns and on exceptions: the module-global reducer must not outlive its engine
        use_native_allreduce(None)
        if device_fasta:
            assembly.close()
        eng.close()


def _gather_rows(args, rows, rank, gather_world, bin_order):
    """Collect the motif rows (whole-bin sharding: from every rank, back into the reference's bin order), apply the
    bin-level filter and let rank 0 write bin-motifs.tsv."""
    if gather_world > 1:
        import torch.distributed as dist
        gathered = [None] * gather_world
        dist.all_gather_object(gathered, rows)
        by_bin = {}
        for part in gathered:
            for r in part:
                by_bin.setdefault(r.reference, []).append(r)
        rows = [r for b in bin_order for r in by_bin.get(b, [])]
    if not rows:
        log.info("No motifs were identified")
        return None
    rows = [r for r in rows if r.n_mod + r.n_nomod >= args.min_motifs_bin]      # main.py:96
    if not rows:
        log.info("Motif frequency of all motifs too low")
        return None
    if rank == 0:
        log.info("Writing motifs")
        postprocess.write_motif_formatted(rows, args.out + "/bin-motifs.tsv")
    log.info(f"Identified {len(rows)} motifs in {len({r.reference for r in rows})} bins")
    return rows


def device_window_pipeline(eng, lengths: dict, mine: list, padding: int, world: int = 1):
    """(window store, extractor) for ``discover``: windows are gathered and the background is counted on the device,
    every rank for its own contigs, with the per-request counts summed over the ranks.  An assembly with letters
    other than A C G T N keeps window extraction on the host (the reference raises KeyError when a window meets one,
    seq.py:474-478, and so does the host path); then every rank holds all windows."""
    from .engine import DeviceWindowExtractor, DeviceWindowStore
    if world > 1:
        import torch.distributed as dist
    other = np.array([eng.other_letters()], dtype=np.int64)
    if world > 1:
        other = allreduce_counts(other)
    if int(other[0]):
        log.info(f"{int(other[0])} assembly letters outside ACGTN: window extraction stays on the host")
        return DeviceWindowStore(eng), None

    def everywhere(local: dict) -> dict:
        if world == 1:
            return local
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        return {k: v for g in gathered for k, v in g.items()}

    # valid sample starts per contig (seq.py:202-225), counted on the device; confident rows per contig and strand (the windows
    # themselves are read from the methylated-state planes).  One GPU: the native plan (nm_plan_windows) counts both itself, so the
    # tables are made only when the task-by-task path asks for them (NANOMOTIF_PLAN_PER_TASK=1): four device round trips and four
    # dictionaries over every contig otherwise — 4 ms of a 1 Gbp run.
    bases = sorted({MOD_TYPE_TO_CANONICAL[mt] for mt in pileup_mod.MOD_TYPES})
    mods = [mt for mt in pileup_mod.MOD_TYPES if mt in eng.slot_of_mod]
    if world == 1:
        n_valid = _LazyTables(bases, lambda base: dict(zip(mine, eng.contig_base_counts(base, padding).tolist())))
        row_counts = _LazyTables(mods, lambda mt: dict(zip(mine, eng.methylated_row_counts(mt, padding).tolist())))
    else:
        n_valid = {base: everywhere(dict(zip(mine, eng.contig_base_counts(base, padding).tolist()))) for base in bases}
        row_counts = {mt: everywhere(dict(zip(mine, eng.methylated_row_counts(mt, padding).tolist()))) for mt in mods}
    reduce = allreduce_counts if world > 1 else None
    store = DeviceWindowStore(eng, allreduce=reduce)
    return store, DeviceWindowExtractor(eng, store, lengths, n_valid, padding, resident=eng.contig_index, allreduce_i64=reduce,
                                        row_counts=row_counts)


class _LazyTables:
    """``tables[key]`` made by ``make(key)`` on first use (the keys are known up front: ``in`` and iteration work without making any)."""

    def __init__(self, keys, make):
        self._keys, self._make, self._made = list(keys), make, {}

    def __getitem__(self, key):
        if key not in self._made:
            if key not in self._keys:
                raise KeyError(key)
            self._made[key] = self._make(key)
        return self._made[key]

    def __contains__(self, key):
        return key in self._keys

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)


def check_installation():
    """main.py:330-346 runs motif_discovery on the packaged geobacillus data; its pileup is not distributable, so
    this build runs the same command on a small synthetic data set with the same planted motifs."""
    import shutil
    import subprocess
    import tempfile
    from . import synth
    tmp = tempfile.mkdtemp(prefix="nanomotif_check_installation_")
    mg = synth.make_metagenome(synth.SynthSpec(
        n_contigs=2, total_bp=200_000, n_bins=1, mod_types=("a",), seed=43, min_contig_bp=80_000,
        fixed_motifs=(("GATC", 1, "a"), ("ACCCA", 4, "a"), ("CCAAAT", 4, "a"), ("GRNGAAGY", 5, "a"))))
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    mg.write_contig_bin(tmp + "/contig_bin.tsv")
    out = tmp + "/out"
    cmd = [sys.executable, "-m", "nanomotif_amd", "motif_discovery", "-t", "1", tmp + "/assembly.fasta", tmp + "/pileup.bed",
           "-c", tmp + "/contig_bin.tsv", "--out", out]
    rc = subprocess.run(cmd).returncode
    if rc == 0 and os.path.exists(out + "/bin-motifs.tsv"):
        print(open(out + "/bin-motifs.tsv").read())
    shutil.rmtree(tmp, ignore_errors=True)
    return rc


def main(argv=None):
    parser = create_parser()
    args = parser.parse_args(argv)
    if args.command == "motif_discovery":
        rank = int(os.environ.get("RANK", "0"))
        shared_setup(args, args.out, rank=rank)
        t_main = time.perf_counter()
        result = find_motifs_bin(args)
        if rank == 0:
            try:
                TIMINGS["find_motifs_bin_s"] = time.perf_counter() - t_main
                with open(os.path.join(args.out, "logs", "timings.motif_discovery.json"), "w") as f:
                    json.dump(TIMINGS, f, indent=1)
            except OSError:
                pass
        if result is None and rank == 0:
            with open(os.path.join(args.out, "bin-motifs.tsv"), "w") as f:     # main.py:317-321
                f.write(HEADER)
    elif args.command == "check_installation":
        sys.exit(check_installation())
    else:
        parser.print_help()
        sys.exit()


if __name__ == "__main__":
    main()

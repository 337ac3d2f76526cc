"""End-to-end motif discovery on a synthetic metagenome generated on the device (bench ``--workload e2e``,
BASELINE cfg 3 / cfg 5 "full candidate-expansion loop"): device generation -> nm_upload_contigs_device ->
nm_ingest_pileup (device filters) -> host windows -> lock-step greedy search + post-processing -> motif rows."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch

from . import _lib, synth, synth_device
from .find_motifs_bin import FilteredPileup, ProcessorConfig, discover, engine_scorer
from .motif import MOD_TYPE_TO_CANONICAL
from .pileup import MOD_TYPES


def generate_raw(mg: synth.SynthMetagenome, device, contigs=None):
    """(mine, lengths, offsets, bins, ASCII of the contigs back to back, raw pileup columns) of (a shard of) ``mg``, all
    on ``device``: the inputs of nm_upload_contigs_device / nm_ingest_pileup."""
    mine = list(range(len(mg.names))) if contigs is None else sorted(int(i) for i in contigs)
    lengths = np.asarray([int(mg.lengths[i]) for i in mine], dtype=np.uint64)
    offsets = np.zeros(len(mine) + 1, dtype=np.uint64)
    np.cumsum(lengths, out=offsets[1:])
    ascii_all = torch.empty(int(offsets[-1]), dtype=torch.uint8, device=device)
    local_of = {g: k for k, g in enumerate(mine)}
    lut = torch.full((len(mg.names),), -1, dtype=torch.int64, device=device)
    lut[torch.tensor(mine, dtype=torch.int64, device=device)] = torch.arange(len(mine), dtype=torch.int64, device=device)
    bins = sorted(set(mg.bin_names))
    by_bin = {b: [] for b in bins}
    for i in mine:
        by_bin[mg.bin_names[i]].append(i)
    cols = {k: [] for k in ("contig", "position", "mod", "strand", "frac", "nvalid")}
    for b in bins:
        if not by_bin[b]:
            continue
        db = synth_device.generate_bin(mg, b, device, min_cov=-1, contigs=by_bin[b])     # raw rows, unfiltered
        st = db.starts.tolist()
        for k, i in enumerate(db.contigs):
            o = int(offsets[local_of[i]])
            ascii_all[o:o + int(mg.lengths[i])] = db.ascii_cat[st[k]:st[k] + int(mg.lengths[i])]
        # the rows of a bin in modkit's order: sorted by (contig, position), mod codes and strands interleaved
        part = {k: [] for k in cols}
        for mt in mg.spec.mod_types:
            p = db.pileups[mt]
            part["contig"].append(lut[p["contig_id"].to(torch.int64)].to(torch.int32))
            part["position"].append(p["position"])
            part["mod"].append(torch.full_like(p["strand"], MOD_TYPES.index(mt)).to(torch.int8))
            part["strand"].append(p["strand"])
            part["frac"].append(p["fraction_mod"])
            part["nvalid"].append(p["nvalid"].to(torch.int32))
        part = {k: torch.cat(v) for k, v in part.items()}
        key = (part["contig"].to(torch.int64) << 34) | (part["position"].to(torch.int64) << 2) | part["mod"].to(torch.int64)
        order = torch.argsort(key)
        for k in cols:
            cols[k].append(part[k][order])
    cat = {k: torch.cat(v).contiguous() for k, v in cols.items()}
    return mine, lengths, offsets, bins, ascii_all, cat


_synth = None


def synth_lib():
    """libnmsynth.so (csrc/bench/nmsynth.cpp): the synthetic-data writers of the bench and the tests — not the product library."""
    global _synth
    if _synth is None:
        from . import build
        lib = C.CDLL(build.build_synth())
        u32p = C.POINTER(C.c_uint32)
        lib.nm_synth_last_error.restype = C.c_char_p
        lib.nm_synth_write_bed.argtypes = [C.c_char_p, C.c_uint64, C.c_uint32, C.c_char_p, u32p, u32p, u32p, C.POINTER(C.c_int8), C.POINTER(C.c_uint8),
                                           C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_uint32]
        lib.nm_synth_bgzip.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_int, C.c_uint32]
        lib.nm_synth_bgz_open.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.nm_synth_bgz_append_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_char_p, u32p, u32p, u32p, C.POINTER(C.c_int8), C.POINTER(C.c_uint8),
                                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.nm_synth_bgz_close.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _synth = lib
    return _synth


def bgzip_tabix(text_path: str, gz_path: str, threads: int = 0, level: int = 6, block_size: int = 0xFF00):
    """``bgzip`` + ``tabix -p bed`` of a bedMethyl text file, natively on several threads (writes gz_path and gz_path + '.tbi')."""
    lib = synth_lib()
    if lib.nm_synth_bgzip(text_path.encode(), gz_path.encode(), int(threads), int(level), int(block_size)):
        raise RuntimeError(lib.nm_synth_last_error().decode())


def write_text_inputs(mg: synth.SynthMetagenome, out_dir: str, device, threads: int = 0) -> dict:
    """The metagenome as the FILES the CLI takes — assembly.fasta, pileup.bed (modkit bedMethyl text, rows in modkit's
    order), contig_bin.tsv — generated on ``device`` and written by the native writer (nm_synth_write_bed of libnmsynth.so: the Python
    row loop of ``write_bed`` takes ~1 us per row, a 100 Mbp pileup has 1e8).  Byte-identical to ``mg.write_bed`` /
    ``write_contig_bin``; the FASTA holds every contig on one line.  Returns sizes."""
    import os
    mine, lengths, offsets, bins, ascii_all, cat = generate_raw(mg, device)
    torch.cuda.synchronize(device)
    host = {k: v.cpu().numpy() for k, v in cat.items()}
    pct = np.rint(host["frac"] * 10000.0).astype(np.int32)               # hundredths of a percent (synth.pct_to_fraction inverted)
    assert np.array_equal(synth.pct_to_fraction(pct), host["frac"])
    names = "".join(mg.names).encode()
    off = np.zeros(len(mg.names) + 1, dtype=np.uint32)
    np.cumsum([len(x) for x in mg.names], out=off[1:])
    lib = synth_lib()
    p = lambda a, t: np.ascontiguousarray(a).ctypes.data_as(C.POINTER(t))
    cid, pos = np.ascontiguousarray(host["contig"], dtype=np.uint32), np.ascontiguousarray(host["position"], dtype=np.uint32)
    mod, st = np.ascontiguousarray(host["mod"], dtype=np.int8), np.ascontiguousarray(host["strand"], dtype=np.uint8)
    nv = np.ascontiguousarray(host["nvalid"], dtype=np.int32)
    bed = os.path.join(out_dir, "pileup.bed")
    if lib.nm_synth_write_bed(bed.encode(), len(cid), len(mg.names), names, p(off, C.c_uint32), p(cid, C.c_uint32), p(pos, C.c_uint32),
                              p(mod, C.c_int8), p(st, C.c_uint8), p(nv, C.c_int32), p(pct, C.c_int32), int(threads)):
        raise RuntimeError(lib.nm_synth_last_error().decode())
    seq = ascii_all.cpu().numpy()
    with open(os.path.join(out_dir, "assembly.fasta"), "wb") as f:
        for j, i in enumerate(mine):
            f.write(b">" + mg.names[i].encode() + b"\n")
            f.write(seq[int(offsets[j]):int(offsets[j + 1])].tobytes())
            f.write(b"\n")
    mg.write_contig_bin(os.path.join(out_dir, "contig_bin.tsv"))
    return {"rows": int(len(cid)), "bed_bytes": os.path.getsize(bed), "assembly_bp": int(lengths.sum())}


def write_gz_inputs_streaming(mg: synth.SynthMetagenome, out_dir: str, device, bins_per_part: int = 25, threads: int = 0, level: int = 6,
                              block_size: int = 0xFF00, text_twin: bool = False, log=None) -> dict:
    """The metagenome as assembly.fasta + pileup.bed.gz + pileup.bed.gz.tbi + contig_bin.tsv, written PART BY PART: the rows of
    ``bins_per_part`` bins at a time are generated on ``device``, formatted, cut into BGZF blocks that continue one text stream
    and appended (nm_synth_bgz_* of libnmsynth.so) — the 75 GB of bedMethyl text of a 1 Gbp metagenome never exist as a file.
    Byte for byte what ``write_text_inputs`` + ``bgzip_tabix`` write for the same metagenome (tests/test_gpu_synth.py); the
    FASTA holds every contig on one line, contigs and rows in bin order.  ``text_twin``: also write pileup.bed."""
    import os
    lib = synth_lib()
    h = C.c_void_p()
    gz = os.path.join(out_dir, "pileup.bed.gz")
    if lib.nm_synth_bgz_open(gz.encode(), os.path.join(out_dir, "pileup.bed").encode() if text_twin else None, int(threads), int(level), int(block_size),
                             C.byref(h)):
        raise RuntimeError(lib.nm_synth_last_error().decode())
    bins = sorted(set(mg.bin_names))
    by_bin = {b: [] for b in bins}
    for i, b in enumerate(mg.bin_names):
        by_bin[b].append(i)
    p = lambda a, t: np.ascontiguousarray(a).ctypes.data_as(C.POINTER(t))
    rows = 0
    try:
        with open(os.path.join(out_dir, "assembly.fasta"), "wb") as fa:
            for k in range(0, len(bins), bins_per_part):
                part = [i for b in bins[k:k + bins_per_part] for i in by_bin[b]]
                mine, lengths, offsets, _, ascii_all, cat = generate_raw(mg, device, part)
                torch.cuda.synchronize(device)
                host = {key: v.cpu().numpy() for key, v in cat.items()}
                pct = np.rint(host["frac"] * 10000.0).astype(np.int32)
                names = "".join(mg.names[i] for i in mine).encode()
                off = np.zeros(len(mine) + 1, dtype=np.uint32)
                np.cumsum([len(mg.names[i]) for i in mine], out=off[1:])
                cid = np.ascontiguousarray(host["contig"], dtype=np.uint32)
                if lib.nm_synth_bgz_append_rows(h, len(cid), len(mine), names, p(off, C.c_uint32), p(cid, C.c_uint32),
                                                p(np.ascontiguousarray(host["position"], dtype=np.uint32), C.c_uint32),
                                                p(np.ascontiguousarray(host["mod"], dtype=np.int8), C.c_int8),
                                                p(np.ascontiguousarray(host["strand"], dtype=np.uint8), C.c_uint8),
                                                p(np.ascontiguousarray(host["nvalid"], dtype=np.int32), C.c_int32), p(pct, C.c_int32)):
                    raise RuntimeError(lib.nm_synth_last_error().decode())
                rows += len(cid)
                seq = ascii_all.cpu().numpy()
                for j, i in enumerate(mine):
                    fa.write(b">" + mg.names[i].encode() + b"\n")
                    fa.write(seq[int(offsets[j]):int(offsets[j + 1])].tobytes())
                    fa.write(b"\n")
                del cat, ascii_all, host, seq
                if log:
                    log(f"  part {k // bins_per_part + 1} of {(len(bins) + bins_per_part - 1) // bins_per_part}: {rows:,} rows so far")
    finally:
        text_bytes, gz_bytes = C.c_uint64(0), C.c_uint64(0)
        rc = lib.nm_synth_bgz_close(h, C.byref(text_bytes), C.byref(gz_bytes))
    if rc:
        raise RuntimeError(lib.nm_synth_last_error().decode())
    mg.write_contig_bin(os.path.join(out_dir, "contig_bin.tsv"))
    return {"rows": rows, "bed_bytes": int(text_bytes.value), "gz_bytes": int(gz_bytes.value), "assembly_bp": int(sum(int(x) for x in mg.lengths)),
            "fasta_bytes": os.path.getsize(os.path.join(out_dir, "assembly.fasta"))}


def load_and_filter(engine, mg: synth.SynthMetagenome, device, contigs=None, host_assembly=True, raw=None):
    """Generate (a shard of) ``mg`` on ``device`` (or take ``raw`` = what ``generate_raw`` returned for it) and ingest it through the
    device-side filters.  Returns (assembly dict name -> uint8 ASCII on the host, FilteredPileup of this shard, timings)."""
    t = {}
    t0 = time.perf_counter()
    mine, lengths, offsets, bins, ascii_all, cat = generate_raw(mg, device, contigs) if raw is None else raw
    # what the ingest will ask the allocator for (16 B per bp of dense maxima, the planes) is taken and given back to
    # torch's pool HERE, with the generation: a fresh hipMalloc of memory that another process used before is scrubbed
    # by the driver at 7-30 GB/s — 0.4 s for this block right after the test suite, none on a fresh box —
    # and that is no more part of the pipeline than the generation of the synthetic rows is
    torch.cuda.synchronize(device)
    t_warm = time.perf_counter()
    if raw is None:                                                # (a caller that brings the inputs has done this as well: run_lanes)
        warm = torch.empty(int(lengths.sum()) * 20, dtype=torch.uint8, device=device)
        del warm
        torch.cuda.synchronize(device)
    t["allocator_prewarm_s"] = time.perf_counter() - t_warm        # reported, not part of wall_s (like the generation)
    t["generate_s"] = time.perf_counter() - t0
    engine.timing_reset(2)                                         # every device phase from here on counts as GPU-busy time
    t0 = time.perf_counter()
    engine.upload_assembly_device([mg.names[i] for i in mine], lengths, [mg.bin_names[i] for i in mine], ascii_all.data_ptr(),
                                  bin_names=bins)
    t["upload_assembly_s"] = time.perf_counter() - t0
    slot_of = (C.c_int32 * 8)(*([-1] * 8))
    canon = (C.c_uint8 * 8)(*([0] * 8))
    for mt in mg.spec.mod_types:
        engine.slot_of_mod[mt] = len(engine.slot_of_mod)
        engine.slot_of_mod[(mt, "merge")] = engine.slot_of_mod[mt]
        slot_of[MOD_TYPES.index(mt)] = engine.slot_of_mod[mt]
        canon[MOD_TYPES.index(mt)] = ord(MOD_TYPE_TO_CANONICAL[mt])
    n = int(cat["position"].numel())
    n_kept, n_conf = C.c_uint64(0), C.c_uint64(0)
    vp = lambda x: C.c_void_p(x.data_ptr())
    t1 = time.perf_counter()
    _lib.check(engine.lib.nm_ingest_pileup(engine.ctx, n, vp(cat["contig"]), vp(cat["position"]), vp(cat["mod"]), vp(cat["strand"]),
                                           vp(cat["frac"]), vp(cat["nvalid"]), slot_of, canon, 0.3, 0.7, 1,
                                           C.byref(n_kept), C.byref(n_conf)))
    t["ingest_call_s"] = time.perf_counter() - t1
    k = n_conf.value
    engine._n_confident = int(k)
    kept = np.zeros((len(mine), 8), dtype=np.uint32)
    _lib.check(engine.lib.nm_ingest_results(engine.ctx, None, None, None, None, 0, kept.ctypes.data_as(C.POINTER(C.c_uint32))))
    # the confident rows stay on the device (methylated-state planes); the host path would fetch them with
    # engine.confident_rows()
    cc, cp, cs, cm = (np.zeros(0, dt) for dt in (np.uint32, np.uint32, np.uint8, np.int8))
    t["upload_filter_s"] = time.perf_counter() - t0
    t["rows_raw"], t["rows_kept"], t["rows_confident"] = n, int(n_kept.value), int(k)
    t0 = time.perf_counter()
    if host_assembly:
        host_ascii = ascii_all.cpu().numpy()
        assembly = {mg.names[i]: host_ascii[int(offsets[j]):int(offsets[j + 1])] for j, i in enumerate(mine)}
    else:
        # the sequences never have to exist on the host: windows are gathered and backgrounds counted from the resident
        # planes (a run from files has them on the host to begin with and pays the upload instead: 1 Gbp = 0.03 s of PCIe)
        empty = np.zeros(0, dtype=np.uint8)
        assembly = {mg.names[i]: empty for i in mine}
    t["assembly_to_host_s"] = time.perf_counter() - t0
    return assembly, FilteredPileup(engine.contig_names, cc, cp, cs, cm, kept), t


def union_ms(intervals) -> float:
    """Length of the union of [begin, end] intervals (float64[n, 2], ms): the time the device was busy when phases overlap."""
    iv = np.asarray(intervals, dtype=np.float64).reshape(-1, 2)
    if not len(iv):
        return 0.0
    iv = iv[np.argsort(iv[:, 0], kind="stable")]
    reach = np.maximum.accumulate(iv[:, 1])
    # a phase adds what of it lies beyond everything that began before it
    return float((reach - np.maximum(iv[:, 0], np.concatenate(([iv[0, 0]], reach[:-1])))).clip(min=0).sum())


def run(mg: synth.SynthMetagenome, engine, device, log=None, bins=None, use_dist=False, raw=None, keep_timing=False):
    """End-to-end run on one GPU; ``bins``: only these bins (whole-bin sharding of a multi-GPU run: every rank runs the
    searches of its own bins alone, find_motifs_bin.py:152-171); ``use_dist``: every round's count and window tables go
    through ``allreduce_counts`` like in a contig-sharded run (the caller has a communicator up); ``raw``: the shard's
    device-resident inputs when the caller generated them already (``generate_raw``).  Returns (rows, timings)."""
    t_entry = time.perf_counter()
    contigs = None
    if bins is not None:
        keep = set(bins)
        contigs = [i for i, b in enumerate(mg.bin_names) if b in keep]
    assembly, filtered, t = load_and_filter(engine, mg, device, contigs=contigs, host_assembly=False, raw=raw)
    names = mg.names if contigs is None else [mg.names[i] for i in contigs]
    bin_of = mg.bin_names if contigs is None else [mg.bin_names[i] for i in contigs]
    lengths = [int(mg.lengths[i]) for i in (range(len(mg.names)) if contigs is None else contigs)]
    cfg = ProcessorConfig(assembly=assembly, pileup_path="<synthetic>", bin_contig=dict(zip(names, bin_of)), threads=1,
                          search_frame_size=40, methylation_threshold_low=0.3, methylation_threshold_high=0.7,
                          minimum_kl_divergence=0.05, score_threshold=1.5, log_dir=None, seed=1, output_dir=None)
    t0 = time.perf_counter()
    scorer = engine_scorer(engine, 0.3, 0.7, use_dist=use_dist)
    from .main import device_window_pipeline
    t1 = time.perf_counter()
    store, extractor = device_window_pipeline(engine, dict(zip(names, lengths)), list(names), cfg.padding)
    t["window_pipeline_s"] = time.perf_counter() - t1
    rows, scorer = discover(cfg, filtered, scorer, window_store=store, extractor=extractor)
    t["search_s"] = time.perf_counter() - t0
    t.update(getattr(scorer, "timings", {}))
    ms, n = engine.timing_total()
    t["gpu_busy_s"] = ms * 1e-3              # every device phase: pre-filters, window gathers and batches, background counts, scoring launches (HIP events, summed)
    t["gpu_busy_union_s"] = union_ms(engine.timing_intervals()) * 1e-3       # the same as a union on the device's clock (phases of the two flights of the search overlap)
    if not keep_timing:                      # (run_lanes lays the phases of all lanes on one time line first)
        engine.timing_reset(False)
    t["device_phases"] = n
    t["rounds"], t["candidates"] = scorer.rounds, scorer.candidates
    t["search_iterations"], t["speculation_hits"], t["speculation_misses"] = scorer.search_iterations, scorer.speculation_hits, scorer.speculation_misses
    t["run_call_s"] = time.perf_counter() - t_entry - t["generate_s"]       # everything of this call but the generation (+ pre-warm) of the inputs
    return rows, t


def lane_bins(mg: synth.SynthMetagenome, n_lanes: int):
    """Whole bins dealt to ``n_lanes`` groups, longest first to the group with the fewest base pairs (the split of a multi-GPU run,
    shard.assign_bins, on ONE device)."""
    size = {}
    for i, b in enumerate(mg.bin_names):
        size[b] = size.get(b, 0) + int(mg.lengths[i])
    groups, load = [[] for _ in range(n_lanes)], [0] * n_lanes
    for b in sorted(size, key=lambda b: (-size[b], b)):
        k = load.index(min(load))
        groups[k].append(b)
        load[k] += size[b]
    return groups


def run_lanes(mg: synth.SynthMetagenome, engines, device):
    """``run`` with the bins dealt to ``len(engines)`` lanes that go through the pipeline SIDE BY SIDE — one engine (its own HIP
    streams), one host thread each: bins are independent searches (find_motifs_bin.py:152-171 hands them to a process pool), so
    while one lane's search waits for the host between two small launches, another lane's pre-filters or window gathers have the
    device.  Returns (rows in the order of a single-lane run, timings: ``wall_s`` from the first lane's start to the last one's end,
    ``lanes`` = every lane's own timings)."""
    import gc
    import threading
    n = len(engines)
    groups = lane_bins(mg, n)
    t0 = time.perf_counter()
    raws = []
    for g in groups:
        keep = set(g)
        raws.append(generate_raw(mg, device, [i for i, b in enumerate(mg.bin_names) if b in keep]))
    torch.cuda.synchronize(device)
    generate_s = time.perf_counter() - t0
    # (load_and_filter's allocator pre-warm, for all lanes at once: one block per lane, as the lanes will ask for them)
    t0 = time.perf_counter()
    warm = [torch.empty(int(r[1].sum()) * 20, dtype=torch.uint8, device=device) for r in raws]
    del warm
    torch.cuda.synchronize(device)
    prewarm_s = time.perf_counter() - t0
    out, err = [None] * n, [None] * n

    def lane(k):
        try:
            out[k] = run(mg, engines[k], device, bins=groups[k], raw=raws[k], keep_timing=True)
        except BaseException as e:           # re-raised on the caller's thread
            err[k] = e
    gc_was_on = gc.isenabled()
    gc.disable()                             # (discover() pauses the collector per lane; a lane that ends early must not switch it back on for the others)
    try:
        threads = [threading.Thread(target=lane, args=(k,), name=f"nm-lane-{k}") for k in range(n)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
    finally:
        if gc_was_on:
            gc.enable()
    for e in err:
        if e is not None:
            raise e
    busy = union_ms(np.concatenate([e.timing_intervals(engines[0]) for e in engines])) * 1e-3
    for e in engines:
        e.timing_reset(False)
    rank = {}
    for b in mg.bin_names:
        rank.setdefault(b, len(rank))
    rows = sorted((r for rs, _ in out for r in rs), key=lambda r: rank[r.reference])       # stable: a bin's rows keep their order
    lanes = [t for _, t in out]
    t = {"wall_s": wall, "generate_s": generate_s, "lanes": lanes, "n_lanes": n, "gpu_busy_union_s": busy,
         "gpu_busy_s": sum(x["gpu_busy_s"] for x in lanes),
         "allocator_prewarm_s": prewarm_s}
    for k in ("rounds", "candidates", "rows_raw", "rows_kept", "rows_confident", "device_phases"):
        t[k] = sum(x.get(k, 0) for x in lanes)
    for k in ("search_iterations",):
        t[k] = max(x.get(k, 0) for x in lanes)
    for k in ("speculation_hits", "speculation_misses"):
        t[k] = sum(x.get(k, 0) for x in lanes)
    return rows, t

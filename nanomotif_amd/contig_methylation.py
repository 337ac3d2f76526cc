"""Per-contig READ methylation of motifs — the table binnary starts from (reference: nanomotif/main.py:142-193; one row
per (contig, motif) is what ``detect_contamination`` / ``include_contigs`` read, binnary/data_processing.py:175-259).

The reference gets it from ``epymetheus.methylation_pattern`` (Rust crate epimetheus-py 0.7.5, setup.py:38 — third party,
source not in the reference tree; restated from its published behaviour in oracle/contig_methylation.py, parity unpinned):
per contig and motif, over the motif's sites on both strands that carry a pileup record with
``n_valid_cov >= min_valid_read_coverage`` and ``n_valid_cov / (n_valid_cov + n_diff) >= min_valid_cov_to_diff_fraction``:

    n_motif_obs          number of such sites
    mean_read_cov        mean n_valid_cov over them
    methylation_value    median of the per-site read fractions n_modified / n_valid_cov (``median``), or
                         sum(n_modified) / sum(n_valid_cov) (``weighted-mean``)

Here the scan, the per-site lookup, the per-(contig, motif) selection of the median and the sums all run on the device
(``nm_readstats_upload`` / ``nm_contig_methylation``, csrc/nmmeth.hip).  Sites are ALL matches of the motif, overlapping
ones included — the reference's own scan semantics (utils.py:44-67).

``confident_site_table`` keeps round 2's per-contig form of the discovery counters (``motif_model_contig`` per contig,
find_motifs_bin.py:1285-1331): thresholded site counts, a different quantity under its own column names."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .motif import MOD_TYPE_TO_CANONICAL, Motif, iupac_to_regex

# header of motifs-scored-read-methylation_<type>.tsv (main.py:159; tests/binnary/test_utils.py:58)
COLUMNS = ["contig", "motif", "mod_type", "mod_position", "methylation_value", "mean_read_cov", "n_motif_obs"]
OUTPUT_TYPES = ("median", "weighted-mean")
MOD_CODES = ["m", "a", "21839"]


def parse_motif_mod(text: str):
    """``GATC_a_1`` -> (IUPAC motif, mod type, mod position): the ``motif_mod`` strings binnary builds from bin-motifs.tsv
    (main.py:129-133)."""
    motif, mod_type, pos = text.rsplit("_", 2)
    return motif, mod_type, int(pos)


def upload_read_statistics(engine, mod_type, contig_local, position, strand, n_valid_cov, n_modified, n_diff=None,
                           min_valid_read_coverage=3, min_valid_cov_to_diff_fraction=0.8) -> int:
    """Records of ONE mod code -> the engine's read-statistics slot of that code (nm_readstats_upload).  ``contig_local``:
    engine contig index per record, 0xFFFFFFFF for contigs the engine does not hold.  Returns the records kept."""
    slot = MOD_CODES.index(mod_type)
    cid = np.ascontiguousarray(contig_local, dtype=np.uint32)
    pos = np.ascontiguousarray(position, dtype=np.uint32)
    st = np.ascontiguousarray(strand, dtype=np.uint8)
    nv = np.ascontiguousarray(n_valid_cov, dtype=np.int32)
    nm = np.ascontiguousarray(n_modified, dtype=np.int32)
    nd = None if n_diff is None else np.ascontiguousarray(n_diff, dtype=np.int32)
    n = len(cid)
    if not (len(pos) == len(st) == len(nv) == len(nm) == n and (nd is None or len(nd) == n)):
        raise ValueError("read-statistics columns differ in length")
    vp = lambda a: a.ctypes.data_as(C.c_void_p) if (a is not None and n) else None
    kept = C.c_uint64(0)
    _lib.check(engine.lib.nm_readstats_upload(engine.ctx, slot, n, vp(cid), vp(pos), vp(st), vp(nv), vp(nm), vp(nd),
                                              int(min_valid_read_coverage), float(min_valid_cov_to_diff_fraction), 0, C.byref(kept)))
    return int(kept.value)


def read_methylation_table(engine, motifs, output_type="median", methylation_threshold=None):
    """motifs: iterable of ``motif_mod`` strings (``GATC_a_1``) or (IUPAC motif, mod_type, mod_position) triples.  Every
    motif is scanned on EVERY resident contig (also the contigs of bins that never showed it — that is what contamination
    detection compares).  Returns dict rows with COLUMNS, motif-major, contigs in engine order; only rows with
    n_motif_obs > 0 exist (like the reference's table).  ``methylation_threshold``: keep rows with
    n_motif_obs * mean_read_cov >= threshold (main.py:193; CLI default 24)."""
    if output_type not in OUTPUT_TYPES:
        raise ValueError(f"Output type must be either median or weighted-mean, got: {output_type}")      # main.py:146
    triples = [parse_motif_mod(m) if isinstance(m, str) else (m[0], m[1], int(m[2])) for m in motifs]
    triples = list(dict.fromkeys(triples))                               # .unique() (main.py:133); first-seen order
    if not triples:
        return []
    batch = engine.make_batch([(Motif(iupac_to_regex(m), pos), mt, 0) for m, mt, pos in triples], slot_of=lambda mt: MOD_CODES.index(mt))
    nc = len(engine.contig_names)
    n_obs = np.zeros((len(triples), nc), dtype=np.uint32)
    cov, med, wm = (np.zeros((len(triples), nc), dtype=np.float64) for _ in range(3))
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    _lib.check(engine.lib.nm_contig_methylation(engine.ctx, len(triples), p(batch.slots, C.c_uint8), p(batch.lens, C.c_uint8),
                                                p(batch.modpos, C.c_uint8), p(batch.offsets, C.c_uint32), p(batch.masks, C.c_uint8),
                                                p(n_obs, C.c_uint32), p(cov, C.c_double), p(med, C.c_double), p(wm, C.c_double)))
    value = med if output_type == "median" else wm
    rows = []
    for k, (m, mt, pos) in enumerate(triples):
        for i in np.flatnonzero(n_obs[k]).tolist():
            if methylation_threshold is not None and float(n_obs[k, i]) * cov[k, i] < methylation_threshold:
                continue
            rows.append(dict(contig=engine.contig_names[i], motif=m, mod_type=mt, mod_position=pos, methylation_value=float(value[k, i]),
                             mean_read_cov=float(cov[k, i]), n_motif_obs=int(n_obs[k, i])))
    return rows


def methylation_pattern(pileup, assembly, motifs, threads=1, min_valid_read_coverage=3, batch_size=1000,
                        min_valid_cov_to_diff_fraction=0.8, output=None, allow_assembly_pileup_mismatch=True,
                        output_type="median", device=0):
    """Drop-in for the call at main.py:167-178: ``pileup`` / ``assembly`` are paths (bedMethyl text, gzip or bgzip; FASTA),
    ``motifs`` the motif_mod strings; writes ``output`` (TSV with COLUMNS) when given and returns the rows.  ``batch_size``
    is accepted for signature compatibility (the device batches motifs itself).  Pileup contigs that are not in the
    assembly are ignored when ``allow_assembly_pileup_mismatch`` (the reference passes True), else an error."""
    from . import fasta
    from .engine import ScanEngine
    lib = _lib.load()
    asm = fasta.load_fasta(assembly)
    names = list(asm)
    eng = ScanEngine(device)
    h = C.c_void_p()
    try:
        eng.upload_assembly(names, [asm[n] for n in names], ["all"] * len(names))
        _lib.check(lib.nm_bed_open_counts(os.fsencode(pileup), int(threads), C.byref(h)))
        n, nc = C.c_uint64(0), C.c_uint32(0)
        _lib.check(lib.nm_bed_shape(h, C.byref(n), C.byref(nc)))
        file_names = []
        for i in range(nc.value):
            s = C.c_char_p()
            _lib.check(lib.nm_bed_contig_name(h, i, C.byref(s)))
            file_names.append(s.value.decode())
        local = {c: i for i, c in enumerate(names)}
        missing = [c for c in file_names if c not in local]
        if missing and not allow_assembly_pileup_mismatch:
            raise ValueError(f"{len(missing)} pileup contigs are not in the assembly (e.g. {missing[0]})")
        lut = np.array([local.get(c, 0xFFFFFFFF) for c in file_names], dtype=np.uint32)
        ptr = [C.c_void_p() for _ in range(6)]
        _lib.check(lib.nm_bed_columns(h, *[C.byref(x) for x in ptr]))
        cnt = [C.c_void_p(), C.c_void_p()]
        _lib.check(lib.nm_bed_count_columns(h, C.byref(cnt[0]), C.byref(cnt[1])))
        view = lambda q, ct: np.ctypeslib.as_array(C.cast(q, C.POINTER(ct)), shape=(n.value,)) if n.value else np.zeros(0, ct)
        contig, position, mod = view(ptr[0], C.c_uint32), view(ptr[1], C.c_int64), view(ptr[2], C.c_int8)
        strand, nvalid = view(ptr[3], C.c_uint8), view(ptr[5], C.c_int64)
        nmod, ndiff = view(cnt[0], C.c_int32), view(cnt[1], C.c_int32)
        wanted = {parse_motif_mod(m)[1] if isinstance(m, str) else m[1] for m in motifs}
        for mt in sorted(wanted):
            if mt not in MOD_CODES:
                raise ValueError(f"unknown modification type '{mt}' (constants.py:28-37 knows m, a, 21839)")
            sel = np.flatnonzero(mod == MOD_CODES.index(mt))
            upload_read_statistics(eng, mt, lut[contig[sel]], position[sel], strand[sel], np.clip(nvalid[sel], -1, 2**31 - 1), nmod[sel],
                                   ndiff[sel], min_valid_read_coverage, min_valid_cov_to_diff_fraction)
        rows = read_methylation_table(eng, motifs, output_type)
    finally:
        if h:
            lib.nm_bed_close(h)
        eng.close()
    if output:
        write_tsv(rows, output)
    return rows


def write_tsv(rows, path, columns=COLUMNS):
    with open(path, "w") as f:
        f.write("\t".join(columns) + "\n")
        for r in rows:
            f.write("\t".join(repr(r[c]) if isinstance(r[c], float) else str(r[c]) for c in columns) + "\n")


# ---- round 2's table: the discovery counters per contig (a different quantity, kept under its own names) ---------------
CONFIDENT_COLUMNS = ["contig", "motif", "mod_type", "mod_position", "confident_methylated_fraction", "n_mod", "n_nomod", "n_confident_sites"]


def confident_site_table(engine, motifs, bins=None):
    """Per (contig, motif) the confidently methylated / unmethylated site counts under the discovery thresholds —
    ``motif_model_contig`` (find_motifs_bin.py:1285-1331) for every contig of every bin in one launch
    (nm_score_batch_per_contig).  NOT the epymetheus table (see ``read_methylation_table``): the fraction is over
    thresholded calls, hence the distinct column names; rows without a confident site are dropped."""
    motifs = list(motifs)
    bins = list(engine.bin_names) if bins is None else list(bins)
    cands = [(Motif(iupac_to_regex(m), int(pos)), mt, b) for b in bins for m, mt, pos in motifs]
    if not cands:
        return []
    res = engine.score_per_contig(cands)
    rows = []
    k = 0
    for b in bins:
        for m, mt, pos in motifs:
            names, table = res[k]
            k += 1
            for name, (n_mod, n_nomod) in zip(names, table.tolist()):
                obs = n_mod + n_nomod
                if obs:
                    rows.append(dict(contig=name, motif=m, mod_type=mt, mod_position=int(pos), confident_methylated_fraction=n_mod / obs,
                                     n_mod=int(n_mod), n_nomod=int(n_nomod), n_confident_sites=int(obs)))
    return rows

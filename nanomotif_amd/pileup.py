"""Pileup ingestion and the three pre-filters, as struct-of-arrays numpy columns — the input contract of
the hot path (reference: nanomotif/dataload.py:15-34, 72-100, 191-247; find_motifs_bin.py:399-418).

A ``PileupTable`` holds one row per (contig, position, strand, mod type) with integer contig / mod-type ids so
that rows can be handed to the HIP engine without touching Python objects per row."""
from __future__ import annotations

import os
from dataclasses import dataclass

import numpy as np

MOD_TYPES = ["m", "a", "21839"]      # constants.py MOD_CODE_TO_PRETTY order = task order (find_motifs_bin.py:152-153)


@dataclass
class PileupTable:
    contig_names: list          # id -> name
    contig: np.ndarray          # int32 id into contig_names
    position: np.ndarray        # int64
    mod_type: np.ndarray        # int8 id into MOD_TYPES
    strand: np.ndarray          # uint8 ASCII '+' / '-'
    fraction_mod: np.ndarray    # float64 = column 11 / 100
    nvalid_cov: np.ndarray      # int64

    def __len__(self):
        return len(self.position)

    def take(self, sel):
        return PileupTable(self.contig_names, self.contig[sel], self.position[sel], self.mod_type[sel],
                           self.strand[sel], self.fraction_mod[sel], self.nvalid_cov[sel])


def load_pileup(path: str, threads: int = 0) -> PileupTable:
    """modkit bedMethyl, 18 tab-separated columns, no header; used: 1 contig, 2 start, 4 mod code, 6 strand,
    10 Nvalid_cov, 11 percent modified; nulls 'NA' / 'null' (dataload.py:72-100).  Parsed natively
    (libnmscan: nm_bed_open — plain text, gzip or bgzip, multi-threaded); rows whose coverage is null can never pass
    ``Nvalid_cov > 5`` in the reference either and are dropped here; a null percentage becomes NaN (the row still counts
    as a position of its group in the frequency filter, dataload.py:216, and leaves at the adjacency filter)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    _lib.check(lib.nm_bed_open(os.fsencode(path), int(threads), C.byref(h)))
    try:
        n, nc = C.c_uint64(0), C.c_uint32(0)
        _lib.check(lib.nm_bed_shape(h, C.byref(n), C.byref(nc)))
        if n.value == 0:
            raise SystemExit("Pileup is empty after initial load")      # dataload.py:89-91 exits with status 1
        names = []
        for i in range(nc.value):
            s = C.c_char_p()
            _lib.check(lib.nm_bed_contig_name(h, i, C.byref(s)))
            names.append(_lib.text_of(s.value, "a contig name of the pileup"))
        ptr = [C.c_void_p() for _ in range(6)]
        _lib.check(lib.nm_bed_columns(h, *[C.byref(x) for x in ptr]))
        def col(i, ctype, dtype):
            return np.ctypeslib.as_array(C.cast(ptr[i], C.POINTER(ctype)), shape=(n.value,)).astype(dtype, copy=True)
        t = PileupTable(names, col(0, C.c_uint32, np.int32), col(1, C.c_int64, np.int64), col(2, C.c_int8, np.int8),
                        col(3, C.c_uint8, np.uint8), col(4, C.c_double, np.float64), col(5, C.c_int64, np.int64))
    finally:
        lib.nm_bed_close(h)
    t.fraction_mod[t.fraction_mod < 0] = np.nan
    ok = t.nvalid_cov >= 0
    return t if ok.all() else t.take(ok)


class NativePileup:
    """A bedMethyl file parsed by the native reader and kept in native memory: ``ingest_columns(lut)`` hands out numpy
    VIEWS in exactly the types ``ScanEngine.ingest_pileup`` takes (no per-column copies at 1e9 rows); valid until
    ``close()``.  ``contig_names``: first-appearance order of the file = index space of ``lut``."""

    def __init__(self, path: str, threads: int = 0, contigs=None, index_path=None):
        """``contigs`` + ``index_path``: read only these contigs through the tabix index (nm_bed_open_indexed — the
        reference's bgzip path, dataload.py:102-152); ``self.indexed`` tells whether that happened: a file that is not a
        tabix index falls back to reading everything."""
        import ctypes as C
        from . import _lib
        self._lib, self._check = _lib.load(), _lib.check
        self._h = C.c_void_p()
        self.indexed, self.bytes_inflated, self.bytes_file = False, None, None
        if contigs is not None and index_path is not None:
            names = [c.encode() for c in contigs]
            off = np.zeros(len(names) + 1, dtype=np.uint32)
            np.cumsum([len(x) for x in names], out=off[1:])
            stats = (C.c_uint64 * 4)()
            rc = self._lib.nm_bed_open_indexed(os.fsencode(path), os.fsencode(index_path), len(names), b"".join(names),
                                               off.ctypes.data_as(C.POINTER(C.c_uint32)), int(threads), C.byref(self._h), stats)
            self.index_problem, self.contigs_not_indexed = None, 0
            if rc == 0:
                self.indexed, self.bytes_inflated, self.bytes_file = True, int(stats[0]), int(stats[1])
                self.contigs_not_indexed = int(stats[2])
                if self.contigs_not_indexed:
                    import logging
                    logging.warning(f"{self.contigs_not_indexed} of {len(names)} wanted contigs have no entry in {index_path} (no rows read for them)")
            elif rc != _lib.NM_EINDEX:
                self._check(rc)          # a damaged pileup, a device error, no memory: final, reading the file again would not help
            else:
                # not a tabix index, or one that does not fit this file (stale: regions off the BGZF blocks, rows of other
                # contigs): the whole file is read instead — slower, never a wrong subset of rows
                import logging
                self.index_problem = self._lib.nm_last_error().decode()
                self._h = C.c_void_p()
                logging.warning(f"tabix index not used ({self.index_problem}): reading the whole pileup")
        if not self.indexed:
            self._check(self._lib.nm_bed_open(os.fsencode(path), int(threads), C.byref(self._h)))
        n, nc = C.c_uint64(0), C.c_uint32(0)
        self._check(self._lib.nm_bed_shape(self._h, C.byref(n), C.byref(nc)))
        self.n = int(n.value)
        if self.n == 0:
            self.close()
            raise SystemExit("Pileup is empty after initial load")      # dataload.py:89-91 exits with status 1
        self.contig_names = []
        for i in range(nc.value):
            s = C.c_char_p()
            self._check(self._lib.nm_bed_contig_name(self._h, i, C.byref(s)))
            self.contig_names.append(_lib.text_of(s.value, "a contig name of the pileup"))

    def __len__(self):
        return self.n

    def file_contig_column(self) -> np.ndarray:
        """uint32 view: per row the index into ``contig_names`` (valid until ``close()``)."""
        import ctypes as C
        ptr = [C.c_void_p() for _ in range(6)]
        self._check(self._lib.nm_bed_columns(self._h, *[C.byref(x) for x in ptr]))
        return np.ctypeslib.as_array(C.cast(ptr[0], C.POINTER(C.c_uint32)), shape=(self.n,))

    def ingest_columns(self, lut: np.ndarray) -> dict:
        import ctypes as C
        lut = np.ascontiguousarray(lut, dtype=np.uint32)
        ptr = [C.c_void_p() for _ in range(6)]
        self._check(self._lib.nm_bed_ingest_columns(self._h, lut.ctypes.data_as(C.POINTER(C.c_uint32)), len(lut),
                                                    *[C.byref(x) for x in ptr]))
        kinds = (("contig", C.c_uint32), ("position", C.c_uint32), ("mod_type", C.c_int8), ("strand", C.c_uint8),
                 ("fraction_mod", C.c_double), ("nvalid_cov", C.c_int32))
        return {name: np.ctypeslib.as_array(C.cast(ptr[i], C.POINTER(ct)), shape=(self.n,)) for i, (name, ct) in enumerate(kinds)}

    def close(self):
        if self._h:
            self._lib.nm_bed_close(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def tabix_regions(index_path: str, contigs) -> dict:
    """``{contig: (begin, end)}`` — the virtual offsets (block file offset << 16 | offset in the block's text) a tabix index holds
    for each of ``contigs`` (nm_tabix_regions; host only); a contig the index does not know is left out.  What pysam's
    ``TabixFile.fetch(contig)`` starts from in the reference (dataload.py:102-152)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    contigs = list(contigs)
    names = [c.encode() for c in contigs]
    off = np.zeros(len(names) + 1, dtype=np.uint32)
    np.cumsum([len(x) for x in names], out=off[1:])
    beg = np.zeros(len(names), dtype=np.uint64)
    end = np.zeros(len(names), dtype=np.uint64)
    have = np.zeros(len(names), dtype=np.uint8)
    rc = lib.nm_tabix_regions(os.fsencode(index_path), len(names), b"".join(names), off.ctypes.data_as(C.POINTER(C.c_uint32)),
                              beg.ctypes.data_as(C.POINTER(C.c_uint64)), end.ctypes.data_as(C.POINTER(C.c_uint64)),
                              have.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc:
        raise _lib.NmScanError(lib.nm_last_error().decode())
    return {c: (int(b), int(e)) for c, b, e, h in zip(contigs, beg, end, have) if h}


class BedPlan:
    """The host-only half of the indexed device parse of a bgzip pileup (nm_bed_plan_indexed): the tabix index read, the regions of
    ``contigs``, the walk over their BGZF blocks — no GPU involved, the library call releases the interpreter lock: the CLI runs it on
    a thread while the HIP runtime comes up and the assembly is parsed.  ``rc`` / ``error``: how the call ended (``NM_EINDEX``: the
    index cannot be used, read the whole file).  Hand it to ``DevicePileup(..., plan=...)`` when ``contigs`` are still the wanted ones."""

    def __init__(self, path: str, index_path: str, contigs, threads: int = 0):
        import ctypes as C
        from . import _lib
        self._lib = _lib.load()
        self.contigs = list(contigs)
        names = [c.encode() for c in self.contigs]
        off = np.zeros(len(names) + 1, dtype=np.uint32)
        np.cumsum([len(x) for x in names], out=off[1:])
        self._h = C.c_void_p()
        stats = (C.c_uint64 * 4)()
        self.rc = self._lib.nm_bed_plan_indexed(os.fsencode(path), os.fsencode(index_path), len(names), b"".join(names),
                                                off.ctypes.data_as(C.POINTER(C.c_uint32)), int(threads), C.byref(self._h), stats)
        self.error = self._lib.nm_last_error().decode() if self.rc else None
        self.bytes_inflated, self.bytes_file, self.contigs_not_indexed, self.seconds = int(stats[0]), int(stats[1]), int(stats[2]), int(stats[3]) * 1e-6

    def close(self):
        if self._h:
            self._lib.nm_bedplan_close(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class DevicePileup:
    """A bedMethyl file — plain text or bgzip — parsed ON THE GPU (nm_bed_parse_device): the six columns live in device
    memory in the types ``nm_ingest_pileup`` takes, no row ever becomes a host array.  ``contig_names``: first-appearance
    order of the file = index space of the ``lut`` given to ``map_contigs``; ``run_row`` / ``run_contig``: the runs of equal
    contig names (a modkit file has one per contig).  ``contigs`` + ``index_path``: only these contigs, through the tabix
    index (nm_bed_parse_device_indexed — the reference's bgzip path, dataload.py:102-152); ``self.indexed`` tells whether
    that happened: an index that is none, or does not fit the file, falls back to the whole file.  Raises NmScanError for a
    gzip stream that is not bgzip (use ``NativePileup``)."""

    COLUMNS = (("contig", np.uint32), ("file_contig", np.uint32), ("position", np.uint32), ("mod_type", np.int8), ("strand", np.uint8),
               ("fraction_mod", np.float64), ("nvalid_cov", np.int32))

    def __init__(self, engine, path: str, threads: int = 0, contigs=None, index_path=None, plan=None):
        """``plan``: a ``BedPlan`` made for exactly ``contigs`` (the host half already done, possibly on another thread)."""
        import ctypes as C
        from . import _lib
        self.engine, self._lib, self._check = engine, _lib.load(), _lib.check
        self._h = C.c_void_p()
        self.indexed, self.bytes_inflated, self.bytes_file = False, None, None
        self.index_problem, self.contigs_not_indexed = None, 0
        if plan is not None and contigs is not None and list(plan.contigs) == list(contigs):
            rc = plan.rc or self._lib.nm_bed_parse_device_planned(engine.ctx, plan._h, int(threads), C.byref(self._h))
            problem = plan.error if plan.rc else (self._lib.nm_last_error().decode() if rc else None)
            if rc == 0:
                self.indexed, self.bytes_inflated, self.bytes_file = True, plan.bytes_inflated, plan.bytes_file
                self.contigs_not_indexed = plan.contigs_not_indexed
                if self.contigs_not_indexed:
                    import logging
                    logging.warning(f"{self.contigs_not_indexed} of {len(plan.contigs)} wanted contigs have no entry in {index_path} (no rows read for them)")
            elif rc != _lib.NM_EINDEX:
                err = _lib.NmScanError(f"libnmscan error {rc}: {problem}")
                err.code = int(rc)
                raise err
            else:
                import logging
                self.index_problem = problem
                self._h = C.c_void_p()
                logging.warning(f"tabix index not used ({self.index_problem}): reading the whole pileup")
        elif contigs is not None and index_path is not None:
            names = [c.encode() for c in contigs]
            off = np.zeros(len(names) + 1, dtype=np.uint32)
            np.cumsum([len(x) for x in names], out=off[1:])
            stats = (C.c_uint64 * 4)()
            rc = self._lib.nm_bed_parse_device_indexed(engine.ctx, os.fsencode(path), os.fsencode(index_path), len(names), b"".join(names),
                                                       off.ctypes.data_as(C.POINTER(C.c_uint32)), int(threads), C.byref(self._h), stats)
            if rc == 0:
                self.indexed, self.bytes_inflated, self.bytes_file = True, int(stats[0]), int(stats[1])
                self.contigs_not_indexed = int(stats[2])
                if self.contigs_not_indexed:
                    import logging
                    logging.warning(f"{self.contigs_not_indexed} of {len(names)} wanted contigs have no entry in {index_path} (no rows read for them)")
            elif rc != _lib.NM_EINDEX:
                self._check(rc)          # corrupt BGZF block / CRC-32, HIP error, no memory: final (not logged as an index problem)
            else:
                # not a tabix index, or one that does not fit this file: the whole file instead — slower, never a wrong subset
                import logging
                self.index_problem = self._lib.nm_last_error().decode()
                self._h = C.c_void_p()
                logging.warning(f"tabix index not used ({self.index_problem}): reading the whole pileup")
        if not self.indexed:
            self._check(self._lib.nm_bed_parse_device(engine.ctx, os.fsencode(path), int(threads), C.byref(self._h)))
        n, nc, nr = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0)
        times = (C.c_double * 2)()
        self._check(self._lib.nm_bedcols_shape(self._h, C.byref(n), C.byref(nc), C.byref(nr), times))
        self.n, self.seconds, self.seconds_reading = int(n.value), float(times[0]), float(times[1])
        phase = (C.c_double * 4)()
        self._check(self._lib.nm_bedcols_phase_seconds(self._h, phase))
        self.seconds_inflating, self.seconds_parsing = float(phase[2]), float(phase[3])
        if self.n == 0:
            self.close()
            raise SystemExit("Pileup is empty after initial load")      # dataload.py:89-91 exits with status 1
        self.contig_names = []
        for i in range(nc.value):
            s = C.c_char_p()
            self._check(self._lib.nm_bedcols_contig_name(self._h, i, C.byref(s)))
            self.contig_names.append(_lib.text_of(s.value, "a contig name of the pileup"))
        self.run_row = np.zeros(nr.value + 1, dtype=np.uint64)
        self.run_contig = np.zeros(max(nr.value, 1), dtype=np.uint32)
        self._check(self._lib.nm_bedcols_runs(self._h, self.run_row.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              self.run_contig.ctypes.data_as(C.POINTER(C.c_uint32))))
        self.run_contig = self.run_contig[:nr.value]

    def __len__(self):
        return self.n

    def mod_code(self, i: int) -> str:
        import ctypes as C
        s = C.c_char_p()
        self._check(self._lib.nm_bedcols_mod_code(self._h, int(i), C.byref(s)))
        from . import _lib
        return _lib.text_of(s.value, "a mod code of the pileup")

    def map_contigs(self, lut: np.ndarray):
        import ctypes as C
        lut = np.ascontiguousarray(lut, dtype=np.uint32)
        self._check(self._lib.nm_bedcols_map_contigs(self._h, lut.ctypes.data_as(C.POINTER(C.c_uint32)), len(lut)))

    def device_pointers(self) -> dict:
        import ctypes as C
        ptr = [C.c_void_p() for _ in range(7)]
        self._check(self._lib.nm_bedcols_device_columns(self._h, *[C.byref(x) for x in ptr]))
        return {name: int(p.value or 0) for (name, _), p in zip(self.COLUMNS, ptr)}

    def to_host(self) -> dict:
        """The columns copied to numpy arrays (tests, debugging)."""
        import ctypes as C
        out = {}
        for (name, dt), addr in zip(self.COLUMNS, self.device_pointers().values()):
            a = np.empty(self.n, dtype=dt)
            if addr:
                self._check(self._lib.nm_device_read(self.engine.ctx, a.ctypes.data_as(C.c_void_p), C.c_void_p(addr), a.nbytes))
                out[name] = a
        return out

    def close(self):
        if self._h:
            self._lib.nm_bedcols_close(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def filter_pileup(t: PileupTable, min_coverage: int = 5) -> PileupTable:
    """dataload.py:191-200: strict Nvalid_cov > 5 (the CLI's --threshold_valid_coverage is parsed but never
    forwarded by the reference, argparser.py:126-129 vs main.py:69-83 — same here)."""
    return t.take(t.nvalid_cov > min_coverage)


def filter_pileup_minimummod_frequency(t: PileupTable, methylation_threshold=0.7, min_mod_frequency=0.0001,
                                       min_mods_pr_contig=50) -> PileupTable:
    """dataload.py:202-226: keep (contig, mod_type) groups with #(frac > thr) / #rows > 1e-4 and #(frac > thr) > 50."""
    if len(t) == 0:
        return t
    key = t.contig.astype(np.int64) * 130 + (t.mod_type.astype(np.int64) + 1)
    n = np.bincount(key)
    n_mod = np.bincount(key, weights=(t.fraction_mod > methylation_threshold)).astype(np.int64)
    ok = np.zeros(len(n), dtype=bool)
    nz = n > 0
    ok[nz] = ((n_mod[nz] / n[nz]) > min_mod_frequency) & (n_mod[nz] > min_mods_pr_contig)
    return t.take(ok[key])


def filter_pileup_adjacency_filter(t: PileupTable, methylation_threshold=0.7, adjacency_distance=8) -> PileupTable:
    """dataload.py:228-247: per (contig, strand) — mod types mixed — a row survives iff its fraction equals the
    maximum over rows at positions p-d..p+d, or is below the threshold.  Evaluated as a dense sliding maximum
    per contig and strand (exact: max of float64 values)."""
    if len(t) == 0:
        return t
    from scipy.ndimage import maximum_filter1d          # (imported here: the CLI filters on the device and never needs it)
    keep = np.zeros(len(t), dtype=bool)
    order = np.lexsort((t.position, t.strand, t.contig))
    grp = t.contig[order].astype(np.int64) * 2 + (t.strand[order] == ord("-"))
    bounds = np.flatnonzero(np.diff(grp)) + 1
    for lo, hi in zip(np.concatenate([[0], bounds]), np.concatenate([bounds, [len(order)]])):
        idx = order[lo:hi]
        pos = t.position[idx]
        frac = t.fraction_mod[idx]
        p0 = int(pos[0])
        dense = np.full(int(pos[-1]) - p0 + 1, -np.inf)
        real = ~np.isnan(frac)                               # null percentages: skipped by the maximum, never kept
        np.maximum.at(dense, pos[real] - p0, frac[real])
        wmax = maximum_filter1d(dense, size=2 * adjacency_distance + 1, mode="constant", cval=-np.inf)[pos - p0]
        with np.errstate(invalid="ignore"):
            keep[idx] = (frac == wmax) | (frac < methylation_threshold)
    return t.take(keep)


def prefilter(t: PileupTable) -> PileupTable:
    """The reference's order (find_motifs_bin.py:399-414)."""
    return filter_pileup_adjacency_filter(filter_pileup_minimummod_frequency(filter_pileup(t)))

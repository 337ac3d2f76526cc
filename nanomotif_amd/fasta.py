"""FASTA / contig-bin readers of the MI355X build (reference: nanomotif/fasta.py:35-187, seq.py:53-71).
Sequences are kept as upper-case uint8 ASCII arrays — the form the engine uploads and the window extractor indexes."""
from __future__ import annotations

import gzip
import logging as log
import os
import sys
from pathlib import Path

import numpy as np

IUPAC = np.zeros(256, dtype=bool)
for _c in "ATGCRYSWKMBDHVN":
    IUPAC[ord(_c)] = True


def _lib_text(raw: bytes) -> str:
    from . import _lib
    return _lib.text_of(raw, "a record name of the assembly")


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def read_fasta_names_and_seqs(path):
    """Yield (name = first whitespace-delimited header token, sequence string)."""
    name, chunks = None, []
    with _open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line[0] == ">":
                if name is not None:
                    yield name, "".join(chunks)
                name, chunks = line[1:].split()[0] if line[1:].split() else "", []
            elif name is not None:
                chunks.append(line)
    if name is not None:
        yield name, "".join(chunks)


def read_fasta_names(path):
    """First whitespace-delimited header token of every record, in file order (headers only: the bin FASTA files of -f / -d
    are read for their contig names, fasta.py:150-170)."""
    import mmap
    import re
    if str(path).endswith(".gz"):
        with gzip.open(path, "rb") as f:
            data = f.read()
    else:
        with open(path, "rb") as f:
            if os.fstat(f.fileno()).st_size == 0:
                return []
            data = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    names = []
    for m in re.finditer(rb"^>([^\r\n]*)", data, re.M):
        tok = m.group(1).split()
        names.append(_lib_text(tok[0]) if tok else "")
    return names


def load_fasta(path, trim_names=False, trim_character=" ") -> dict:
    """name -> upper-case uint8 array (views into one buffer).  Parsed natively (libnmscan: nm_fasta_open — plain or
    gzip, records in parallel); empty or non-IUPAC sequences fail like DNAsequence._check_sequence (seq.py:68-71)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.nm_fasta_open(os.fsencode(str(path)), 0, C.byref(h))
    if rc:
        msg = lib.nm_last_error().decode()
        if rc == _lib.NM_ESEQUENCE:
            raise AssertionError(msg)
        _lib.check(rc)
    try:
        n, total = C.c_uint32(0), C.c_uint64(0)
        _lib.check(lib.nm_fasta_shape(h, C.byref(n), C.byref(total)))
        ptr = C.c_void_p()
        _lib.check(lib.nm_fasta_sequence(h, C.byref(ptr)))
        whole = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(total.value,)).copy() if total.value else np.zeros(0, np.uint8)
        out = {}
        for i in range(n.value):
            name, off, ln = C.c_char_p(), C.c_uint64(0), C.c_uint64(0)
            _lib.check(lib.nm_fasta_record(h, i, C.byref(name), C.byref(off), C.byref(ln)))
            key = _lib.text_of(name.value, "a record name of the assembly")
            if trim_names:
                key = key.split(trim_character)[0]
            out[key] = whole[off.value:off.value + ln.value]
    finally:
        lib.nm_fasta_close(h)
    return out


class DeviceAssembly:
    """The assembly FASTA parsed ON THE GPU (nm_fasta_parse_device): the record table (names, lengths) lives here, the bases stay
    in device memory, upper-cased and checked like ``load_fasta`` does (seq.py:53-71) — ``ScanEngine.upload_assembly_fasta`` packs
    the wanted records into the planes from there.  Behaves like the dict ``load_fasta`` returns where the pipeline needs it:
    ``in``, iteration over names, ``length(name)``; ``assembly[name]`` copies that record's bases to the host (only the host
    window path of an assembly with IUPAC ambiguity letters ever asks).  Plain text only: a ``.gz`` assembly keeps ``load_fasta``."""

    def __init__(self, engine, path, threads: int = 0):
        import ctypes as C
        from . import _lib
        self.engine, self._lib, self._check = engine, _lib.load(), _lib.check
        self._h = C.c_void_p()
        rc = self._lib.nm_fasta_parse_device(engine.ctx, os.fsencode(str(path)), int(threads), C.byref(self._h))
        if rc:
            msg = self._lib.nm_last_error().decode()
            if rc == _lib.NM_ESEQUENCE:
                raise AssertionError(msg)                      # DNAsequence._check_sequence asserts (seq.py:68-71)
            self._check(rc)
        n, total = C.c_uint32(0), C.c_uint64(0)
        times = (C.c_double * 2)()
        self._check(self._lib.nm_fastadev_shape(self._h, C.byref(n), C.byref(total), times))
        self.total_bp, self.seconds, self.seconds_reading = int(total.value), float(times[0]), float(times[1])
        self.record, self._host = {}, {}
        blob, nbytes, offs = C.c_void_p(), C.c_uint64(0), C.c_void_p()
        self._check(self._lib.nm_fastadev_table(self._h, C.byref(blob), C.byref(nbytes), C.byref(offs)))      # one call for the whole table
        names = C.string_at(blob.value, nbytes.value).split(b"\0")[:n.value] if n.value else []
        off = np.ctypeslib.as_array(C.cast(offs, C.POINTER(C.c_uint64)), shape=(n.value + 1,)).astype(np.int64)
        self._offset = off[:-1].tolist()
        self._length = np.diff(off).tolist()
        try:
            for i, raw in enumerate(names):
                self.record[raw.decode()] = i                         # a repeated name: the last record wins, like a dict
        except UnicodeDecodeError:
            self.close()
            _lib.text_of(raw, "a record name of the assembly")        # raises with the place named

    def __contains__(self, name):
        return name in self.record

    def __iter__(self):
        return iter(self.record)

    def __len__(self):
        return len(self.record)

    def keys(self):
        return self.record.keys()

    def length(self, name) -> int:
        return self._length[self.record[name]]

    def alias(self, name, original):
        """``name`` stands for the record of ``original`` as well (a contig listed under several bins)."""
        self.record[name] = self.record[original]

    def __getitem__(self, name):
        import ctypes as C
        i = self.record[name]
        if i not in self._host:
            ptr = C.c_void_p()
            self._check(self._lib.nm_fastadev_sequence_device(self._h, C.byref(ptr)))
            a = np.empty(self._length[i], dtype=np.uint8)
            self._check(self._lib.nm_device_read(self.engine.ctx, a.ctypes.data_as(C.c_void_p), C.c_void_p((ptr.value or 0) + self._offset[i]), a.nbytes))
            self._host[i] = a
        return self._host[i]

    def close(self):
        if self._h:
            self._lib.nm_fastadev_close(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def assembly_length(assembly, name) -> int:
    """Bases of a contig, whichever form the assembly has (dict of arrays, or a DeviceAssembly)."""
    return assembly.length(name) if isinstance(assembly, DeviceAssembly) else len(assembly[name])


ALIAS_SEP = "\x00"       # a contig listed under several bins: its further placements are named  contig + ALIAS_SEP + bin


def original_name(name: str) -> str:
    """The contig a (possibly aliased) placement stands for."""
    return name.split(ALIAS_SEP, 1)[0]


def _assign(table: dict, contig: str, bin_name: str):
    """The reference keeps a (contig, bin) DataFrame: a contig listed under two bins is a member of BOTH — its pileup rows are
    joined to each (find_motifs_bin.py:416-418) and each bin's assembly holds it (fasta.py:122-187).  The engine holds a
    contig once per bin, so every further placement becomes a contig of its own, named  contig + ALIAS_SEP + bin  (it sorts
    right after the contig's own name: the window-extraction order inside the other bin is that of the original name)."""
    old = table.get(contig)
    if old is None:
        table[contig] = bin_name
        return
    if old == bin_name or table.get(contig + ALIAS_SEP + bin_name) == bin_name:
        log.warning(f"contig {contig} is listed under bin {bin_name} more than once: the repeated line is ignored")
        return
    log.warning(f"contig {contig} is listed under bins {old} and {bin_name}: it is scored in both, like in the reference")
    table[contig + ALIAS_SEP + bin_name] = bin_name


def add_alias_sequences(assembly: dict, bin_contig: dict):
    """Every aliased placement gets its contig's sequence (the same array, no copy)."""
    for name in bin_contig:
        if ALIAS_SEP in name and original_name(name) in assembly:
            if isinstance(assembly, DeviceAssembly):
                assembly.alias(name, original_name(name))
            else:
                assembly[name] = assembly[original_name(name)]


def generate_contig_bin(args) -> dict:
    """fasta.py:122-187 -> ordered dict contig -> bin.  -c: 2-column TSV without header; -f / -d: bin = file stem,
    contig = first header token; with -f/-d the mapping is also written to OUT/temp/contig_bin.tsv."""
    if getattr(args, "contig_bin", None):
        out = {}
        with open(args.contig_bin) as f:
            for line in f:
                line = line.rstrip("\r\n")
                if not line:
                    continue
                parts = line.split("\t")
                _assign(out, str(parts[0]), str(parts[1]))
        return out
    if getattr(args, "files", None):
        files = list(args.files)
        for fp in files:
            if not Path(fp).exists():
                log.warning(f"Error: File '{fp}' does not exist")
                sys.exit(1)
    else:
        d = Path(args.directory)
        if not d.exists():
            print(f"Error: Directory '{args.directory}' does not exist", file=sys.stderr)
            sys.exit(1)
        ext = args.extension if args.extension.startswith(".") else "." + args.extension
        files = [str(p) for p in sorted(d.glob(f"*{ext}"))]
    if not files:
        log.error("No files to process")
        sys.exit(1)
    out = {}
    for fp in files:
        stem = Path(fp).stem
        for name in read_fasta_names(fp):
            if name:
                _assign(out, name, stem)
    path = os.path.join(args.out, "temp", "contig_bin.tsv")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        for c, b in out.items():
            f.write(f"{original_name(c)}\t{b}\n")
    return out

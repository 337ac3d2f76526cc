"""Shared test helpers: fixtures, synthetic inputs, oracle-format conversion."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

from nanomotif_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def sha1(arr) -> str:
    return hashlib.sha1(np.ascontiguousarray(arr).tobytes()).hexdigest()


def spec_from_json(d) -> synth.SynthSpec:
    kw = dict(d)
    if "mod_types" in kw:
        kw["mod_types"] = tuple(kw["mod_types"])
    if kw.get("fixed_motifs") is not None:
        kw["fixed_motifs"] = tuple(tuple(m) for m in kw["fixed_motifs"])
    return synth.SynthSpec(**kw)


def oracle_bin_inputs(mg, mod_type, contigs=None, min_cov=5):
    """(pileup dict name->ContigPileup, contigs dict name->str) with the coverage filter applied."""
    from oracle.scan import ContigPileup
    idx = range(len(mg.names)) if contigs is None else contigs
    pile, seqs = {}, {}
    for i in idx:
        p = mg.contig_pileup(i, mod_type)
        keep = p["nvalid"] > min_cov
        pile[mg.names[i]] = ContigPileup(p["position"][keep], p["strand"][keep],
                                         synth.pct_to_fraction(p["pct_hundredths"][keep]))
        seqs[mg.names[i]] = mg.contig_str(i)
    return pile, seqs


def oracle_pipeline(mg, min_motifs_bin=50, seed=1, bgzip_order=False, low=0.3, high=0.7, bins=None, padding=20):
    """bin-motifs.tsv text computed end to end by the CPU oracle (filters -> search -> post-processing),
    following the task order / seeding of the reference's plain (or bgzip) strategy.  ``bins``: restrict to these
    bins (bins are independent tasks, find_motifs_bin.py:152-171)."""
    from oracle import pipeline as opl
    from oracle import postprocess as opp
    order = []
    for b in mg.bin_names:
        if b not in order and (bins is None or b in bins):
            order.append(b)
    rows = []
    for b in order:
        rows += opl.bin_rows(mg, b, seed=seed, bgzip_order=bgzip_order, low=low, high=high, padding=padding)
    return opp.format_bin_motifs(rows, min_motifs_bin=min_motifs_bin)


def spec_kwargs(spec) -> dict:
    """SynthSpec -> plain dict that survives pickling into a spawn worker (oracle.pipeline workers)."""
    import dataclasses
    return dataclasses.asdict(spec)


def oracle_pipeline_parallel(mg, bins, procs, min_motifs_bin=50, **kw):
    """``oracle_pipeline`` restricted to ``bins`` with one spawn worker per bin (the oracle needs seconds to minutes
    per 2 Mbp bin)."""
    import multiprocessing as mp
    from oracle import pipeline as opl
    from oracle import postprocess as opp
    jobs = [(spec_kwargs(mg.spec), b, kw) for b in bins]
    with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
        res = pool.map(opl.bin_rows_worker, jobs, chunksize=1)
    order = [b for b in dict.fromkeys(mg.bin_names) if b in set(bins)]
    by_bin = {b: rows for b, rows, _ in res}
    rows = [r for b in order for r in by_bin[b]]
    return opp.format_bin_motifs(rows, min_motifs_bin=min_motifs_bin)


# ---------------------------------------------------------------------------------------------
# motif zoo shared by G1/G2 (regex-style strings as the reference uses them)
# ---------------------------------------------------------------------------------------------
def motif_zoo():
    zoo = [
        # literals / palindromes
        ("A", 0), ("C", 0), ("AA", 0), ("AA", 1), ("AAAA", 2), ("GATC", 1), ("GATC", 3), ("CCGG", 1),
        ("GAATTC", 2), ("CTGCAG", 4), ("ACCCA", 4), ("CCAAAT", 4), ("TTCGAA", 5), ("GTAC", 2),
        ("ACGT", 0), ("ACGT", 1), ("ACGT", 2), ("ACGT", 3), ("TTTT", 0), ("CAGAG", 3),
        # gaps and bipartite
        ("GA.TC", 1), ("A.A", 0), ("A.A", 2), ("C..G", 0), ("GCAC......GTT", 2), ("AAC......GTGC", 1),
        ("CAC.....TGG", 1), ("A..........T", 0), ("A...................C", 0),
        ("C....................A....................G", 21),
        # IUPAC sets
        ("CC[AT]GG", 1), ("G[AG].GAAG[CT]", 5), ("[AG]GC[CT]", 2), ("GC.GC", 1), ("[ACG]A[CGT]", 1),
        ("[AC][AC][AC]", 1), ("[CGT]A", 1), ("A[ACT]", 0), ("[AG][CT][AG][CT]", 0), ("[GT]A[AC]..[ACG]C", 1),
        # modified base not canonical / at a bracket (generic API use)
        ("GATC", 0), ("GATC", 2), ("CC[AT]GG", 2), ("TTAA", 0), ("TTAA", 1),
        # flanking dots (search-window form, pad 20)
        ("." * 19 + "GATC" + "." * 18, 20), ("." * 20 + "A" + "." * 20, 20), ("." * 20 + "C" + "." * 20, 20),
        ("." * 18 + "CCAGG" + "." * 18, 19), ("." * 14 + "GCAC......GTT" + "." * 14, 16),
        ("." * 20 + "AATT" + "." * 17, 20), ("..GA.TC..", 3), (".A", 1), ("A.", 0),
    ]
    # seeded extras: random stripped motifs incl. long ones
    for s, p, _ in synth.random_candidates(40, seed=11, mod_types=("a", "m")):
        zoo.append((s, p))
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(20):
        L = int(rng.integers(20, 42))
        chars = ["."] * L
        for q in rng.choice(L, size=int(rng.integers(3, 9)), replace=False):
            chars[int(q)] = "ACGT"[int(rng.integers(4))]
        if chars[0] == "." and chars[-1] == ".":
            chars[0] = "G"
        pos = int(rng.integers(0, L))
        chars[pos] = "AC"[int(rng.integers(2))]
        zoo.append(("".join(chars), pos))
    return zoo


# ---------------------------------------------------------------------------------------------
# bgzip + tabix of a bedMethyl text (what `bgzip p.bed; tabix -p bed p.bed.gz` produce), written from the format
# specifications (SAM spec §4.1 BGZF; tabix.pdf) — htslib / pysam are not in the image
# ---------------------------------------------------------------------------------------------
def write_bgzf_tabix(bed_text: bytes, gz_path: str, block_size: int = 0xFF00, level: int = 6, strategy: int = 0):
    import struct
    import zlib

    def block(data: bytes) -> bytes:
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = c.compress(data) + c.flush()
        return (struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp
                + struct.pack("<II", zlib.crc32(data), len(data)))

    # blocks cut at arbitrary byte positions (lines straddle blocks, as in real files)
    blocks, block_coff, coff = [], [], 0
    for i in range(0, len(bed_text), block_size):
        b = block(bed_text[i:i + block_size])
        block_coff.append(coff)
        blocks.append(b)
        coff += len(b)
    eof = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    with open(gz_path, "wb") as f:
        f.write(b"".join(blocks) + eof)

    def voff(text_off: int) -> int:
        if text_off >= len(bed_text):
            return coff << 16
        k = text_off // block_size
        return (block_coff[k] << 16) | (text_off - k * block_size)

    def reg2bin(beg, end):
        end -= 1
        for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if beg >> shift == end >> shift:
                return base + (beg >> shift)
        return 0

    refs, order = {}, []
    pos = 0
    for line in bed_text.split(b"\n"):
        if line:
            f = line.split(b"\t")
            name, beg, end = f[0], int(f[1]), int(f[2])
            r = refs.get(name)
            if r is None:
                r = refs[name] = dict(bins={}, lin={}, first=pos, last=pos)
                order.append(name)
            v0, v1 = voff(pos), voff(pos + len(line) + 1)
            chunks = r["bins"].setdefault(reg2bin(beg, end), [])
            if chunks and chunks[-1][1] == v0:
                chunks[-1][1] = v1
            else:
                chunks.append([v0, v1])
            for w in range(beg >> 14, ((end - 1) >> 14) + 1):
                r["lin"].setdefault(w, v0)
            r["last"] = pos + len(line) + 1
        pos += len(line) + 1
    names = b"".join(n + b"\0" for n in order)
    out = [b"TBI\1", struct.pack("<iiiiiiii", len(order), 0x10000, 1, 2, 3, ord("#"), 0, len(names)), names]
    for n in order:
        r = refs[n]
        out.append(struct.pack("<i", len(r["bins"]) + 1))
        for b, chunks in sorted(r["bins"].items()):
            out.append(struct.pack("<Ii", b, len(chunks)))
            for v0, v1 in chunks:
                out.append(struct.pack("<QQ", v0, v1))
        n_rec = sum(1 for _ in r["bins"])
        out.append(struct.pack("<IiQQQQ", 37450, 2, voff(r["first"]), voff(r["last"]), n_rec, 0))     # metadata pseudo-bin
        n_intv = (max(r["lin"]) + 1) if r["lin"] else 0
        out.append(struct.pack("<i", n_intv))
        last = 0
        for w in range(n_intv):
            last = r["lin"].get(w, last)
            out.append(struct.pack("<Q", last))
    idx = b"".join(out)
    with open(gz_path + ".tbi", "wb") as f:                 # the index is itself BGZF
        f.write(b"".join(block(idx[i:i + block_size]) for i in range(0, len(idx), block_size)) + eof)

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


REF_TBI = os.path.join(GOLDEN, "data_geobacillus-plasmids.pileup.bed.gz.tbi")
# what the index holds (decoded by hand from the file: gzip -dc | od; tabix format, SAM/tabix spec): per sequence the pseudo-bin
# 37450's first chunk = [begin, end) as virtual offsets (block file offset, offset inside the block's text)
REF_TBI_REGIONS = {"contig_3": ((0, 0), (2938871, 11451)), "contig_2": ((2938871, 11451), (5249729, 59922))}


def stored_bgzf_block(text: bytes) -> bytes:
    """One BGZF member whose deflate stream is a single STORED block: its size in the file is exactly len(text) + 31, so a test
    can put a block at any file offset it wants."""
    import struct
    import zlib
    assert len(text) <= 65535
    deflate = b"\x01" + struct.pack("<HH", len(text), len(text) ^ 0xFFFF) + text
    bsize = 18 + len(deflate) + 8
    head = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", bsize - 1)
    return head + deflate + struct.pack("<II", zlib.crc32(text) & 0xFFFFFFFF, len(text))


def pileup_laid_out_like_the_reference_index(path: str):
    """A bgzip pileup whose BLOCK LAYOUT is the one datasets/geobacillus-plasmids.pileup.bed.gz.tbi (made by htslib, the only real
    third-party index in the reference tree) describes — the pileup itself is not in the tree: rows of contig_3 from virtual offset
    0:0 to 2938871:11451, rows of contig_2 from there to 5249729:59922, then rows of a third contig the index does not know.
    Stored deflate blocks put every block boundary exactly where the index says.  Returns the text of the three contigs."""
    def rows(name, total, seed):
        rng = np.random.default_rng(seed)
        out, size, pos = [], 0, 0
        while True:
            pos += int(rng.integers(1, 9))
            cov = int(rng.integers(1, 60))
            nmod = int(rng.integers(0, cov + 1))
            pct = nmod * 10000 // cov
            st = "+-"[int(rng.integers(0, 2))]
            mt = ("a", "m", "21839")[int(rng.integers(0, 3))]
            ln = (f"{name}\t{pos}\t{pos + 1}\t{mt}\t{cov}\t{st}\t{pos}\t{pos + 1}\t255,0,0\t{cov}\t{pct // 100}.{pct % 100:02d}"
                  f"\t{nmod}\t{cov - nmod}\t0\t0\t0\t0\t0\n").encode()
            if size + len(ln) + 200 > total:
                # the last row: the colour column (free text, never read) is stretched so that the contig's text ends on the byte
                pad = total - size - len(ln)
                assert pad >= 0
                ln = ln.replace(b"255,0,0", b"255,0,0" + b"0" * pad)
                out.append(ln)
                return b"".join(out)
            out.append(ln)
            size += len(ln)
    full = 0xFF00
    # contig_3: blocks before file offset 2938871 = 44 full blocks + one of 65156 (44 * 65311 + 65187 = 2938871), + 11451 bytes of the next
    sizes = [full] * 44 + [65156]
    assert sum(x + 31 for x in sizes) == 2938871
    t3 = rows("contig_3", sum(sizes) + 11451, 3)
    # contig_2: the rest of the block at 2938871 (24942 bytes of text: 24973 in the file), 35 full blocks, 59922 bytes of the block at 5249729
    sizes += [24942] + [full] * 35
    assert sum(x + 31 for x in sizes) == 5249729
    t2 = rows("contig_2", (24942 - 11451) + 35 * full + 59922, 2)
    tx = rows("contig_x", 5000 + 30000, 1)
    sizes += [59922 + 5000, 30000]
    text = t3 + t2 + tx
    assert len(text) == sum(sizes)
    with open(path, "wb") as f:
        at = 0
        for n in sizes:
            f.write(stored_bgzf_block(text[at:at + n]))
            at += n
        f.write(stored_bgzf_block(b""))                                  # the EOF marker block
    return {"contig_3": t3, "contig_2": t2, "contig_x": tx}


def g14_table(g):
    """The input rows of fixture g14 as an oracle.pileup table (+ a ``row`` column = the fixture's row numbers)."""
    r = g["rows"]
    n = g["n_rows"]
    return dict(contig=np.array(r["contig"], dtype=object), position=np.array(r["position"], dtype=np.int64),
                mod_type=np.array(r["mod_type"], dtype=object), strand=np.array(r["strand"], dtype=object),
                fraction_mod=np.array([np.nan if x is None else x for x in r["fraction_mod"]], dtype=np.float64),
                Nvalid_cov=np.full(n, r["Nvalid_cov"], dtype=np.int64), row=np.arange(n, dtype=np.int64))

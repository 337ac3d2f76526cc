"""Native bedMethyl reader (nm_bed_*, libnmscan) — host-only, runs without a GPU."""
import gzip
import os
import struct
import time
import zlib

import numpy as np
import pytest

from nanomotif_amd import pileup as pp
from nanomotif_amd import synth


def _bgzf(data, bs=60000):
    out = b""
    for i in range(0, len(data), bs):
        blk = data[i:i + bs]
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = c.compress(blk) + c.flush()
        out += struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, len(comp) + 25) + comp
        out += struct.pack("<II", zlib.crc32(blk), len(blk))
    return out + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_reader_is_exact_on_plain_gzip_and_bgzf(tmp_path):
    spec = synth.SynthSpec(n_contigs=4, total_bp=300_000, n_bins=2, mod_types=("a", "m"), seed=61, min_contig_bp=30_000)
    mg = synth.make_metagenome(spec)
    path = str(tmp_path / "p.bed")
    mg.write_bed(path)
    t0 = time.perf_counter()
    tab = pp.load_pileup(path)
    dt = time.perf_counter() - t0
    assert len(tab) == 300_000 and sorted(tab.contig_names) == sorted(mg.names)
    for mt_id, mt in ((1, "a"), (0, "m")):
        for ci in range(4):
            sel = (tab.mod_type == mt_id) & (tab.contig == tab.contig_names.index(mg.names[ci]))
            host = mg.contig_pileup(ci, mt)
            o = np.lexsort((tab.strand[sel], tab.position[sel]))
            ho = np.lexsort((host["strand"], host["position"]))
            assert np.array_equal(tab.position[sel][o], host["position"][ho])
            assert np.array_equal(tab.strand[sel][o], host["strand"][ho])
            assert np.array_equal(tab.fraction_mod[sel][o], synth.pct_to_fraction(host["pct_hundredths"])[ho])   # bit-exact doubles
            assert np.array_equal(tab.nvalid_cov[sel][o], host["nvalid"][ho])
    raw = open(path, "rb").read()
    with gzip.open(path + ".gz", "wb") as g:
        g.write(raw)
    open(path + ".bgz.gz", "wb").write(_bgzf(raw))
    for p in (path + ".gz", path + ".bgz.gz"):
        t2 = pp.load_pileup(p)
        assert t2.contig_names == tab.contig_names
        for col in ("contig", "position", "mod_type", "strand", "fraction_mod", "nvalid_cov"):
            assert np.array_equal(getattr(t2, col), getattr(tab, col)), (p, col)
    assert dt < 5.0


def test_reader_edge_cases(tmp_path):
    lines = [
        "c 1\t10\t11\ta\t12\t+\t10\t11\t255,0,0\t12\t70.00\t8\t4\t0\t0\t0\t0\t0",      # name with a space
        "c 1\t11\t12\tm\t3\t-\t11\t12\t255,0,0\tNA\t50.5\t1\t2\t0\t0\t0\t0\t0",        # null coverage -> dropped
        "c2\t0\t1\t21839\t9\t+\t0\t1\t255,0,0\t9\tnull\t0\t9\t0\t0\t0\t0\t0",           # null percent -> kept as NaN (pl.count() counts the row)
        "c2\t5\t6\th\t9\t-\t5\t6\t255,0,0\t9\t33.333333333333336\t3\t6\t0\t0\t0\t0\t0",  # unknown mod code, long float
        "c2\t7\t8\ta\t100\t+\t7\t8\t255,0,0\t100\t1e2\t100\t0\t0\t0\t0\t0\t0",          # exponent form -> strtod path
    ]
    path = str(tmp_path / "e.bed")
    open(path, "w").write("\r\n".join(lines) + "\r\n")
    t = pp.load_pileup(path)
    assert t.contig_names == ["c 1", "c2"] and len(t) == 4
    assert t.position.tolist() == [10, 0, 5, 7] and t.mod_type.tolist() == [1, 2, 3, 1]
    assert np.isnan(t.fraction_mod[1]) and t.nvalid_cov[1] == 9
    assert t.fraction_mod[[0, 2, 3]].tolist() == [70.00 / 100, 33.333333333333336 / 100, 1.0]
    assert t.strand.tolist() == [ord("+"), ord("+"), ord("-"), ord("+")]
    from nanomotif_amd._lib import NmScanError
    # strict like the fixed 18-column schema of the reference (PILEUP_SCHEMA, dataload.py:15-34): column count, start, strand
    good = "c\t1\t2\ta\t9\t+\t1\t2\t0\t9\t1.0\t0\t9\t0\t0\t0\t0\t0"
    open(path, "w").write(good + "\n")
    assert len(pp.load_pileup(path)) == 1
    cols = good.split("\t")
    for bad, what in (("c\t1\t2\ta", "exactly 18"), ("\t".join(cols[:17]), "exactly 18"), (good + "\t0", "exactly 18"), (good + "\t", "exactly 18"),
                      ("\t".join(cols[:5] + ["."] + cols[6:]), "column 6"), ("\t".join(cols[:5] + [""] + cols[6:]), "column 6"),
                      ("\t".join(cols[:5] + ["++"] + cols[6:]), "column 6"), ("\t".join(cols[:1] + ["-1"] + cols[2:]), "negative"),
                      ("\t".join(cols[:1] + ["1.5"] + cols[2:]), "column 2"), ("\t".join(cols[:9] + ["x"] + cols[10:]), "column 10"),
                      ("\t".join(cols[:10] + ["1..0"] + cols[11:]), "column 11")):
        open(path, "w").write(good + "\n" + bad + "\n" + good.replace("\t1\t2", "\t5\t6", 1) + "\n")
        with pytest.raises(NmScanError, match=what):
            pp.load_pileup(path)
    with pytest.raises(NmScanError):
        pp.load_pileup(str(tmp_path / "missing.bed"))
    open(path, "w").write("")
    with pytest.raises(SystemExit):
        pp.load_pileup(path)


def test_native_ingest_columns_match_the_table(tmp_path):
    """NativePileup.ingest_columns: zero-copy views in the engine's types == load_pileup + the numpy conversions."""
    from nanomotif_amd import pileup, synth
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=4, total_bp=60_000, n_bins=2, mod_types=("a", "m"), seed=5))
    path = str(tmp_path / "p.bed")
    mg.write_bed(path)
    with open(path, "a") as f:          # a null coverage falls to the coverage filter; a null percentage stays a position (fraction -1 / NaN)
        f.write("contig_0000\t7\t8\ta\t9\t+\t7\t8\t255,0,0\t9\tNA\t0\t9\t0\t0\t0\t0\t0\n")
        f.write("contig_0001\t9\t10\tm\tnull\t-\t9\t10\t255,0,0\tnull\t1.00\t0\t0\t0\t0\t0\t0\t0\n")
    t = pileup.load_pileup(path)                       # drops the null-coverage row
    nat = pileup.NativePileup(path)
    assert len(nat) == len(t) + 1 and nat.contig_names[:len(t.contig_names)] == t.contig_names
    lut = np.arange(len(nat.contig_names), dtype=np.uint32)[::-1].copy()
    lut[0] = 0xFFFFFFFF
    cols = nat.ingest_columns(lut)
    assert [cols[k].dtype for k in ("contig", "position", "mod_type", "strand", "fraction_mod", "nvalid_cov")] == \
        [np.uint32, np.uint32, np.int8, np.uint8, np.float64, np.int32]
    live = cols["nvalid_cov"] >= 0
    assert int((~live).sum()) == 1 and int((cols["fraction_mod"] < 0).sum()) == 1
    assert cols["nvalid_cov"][cols["fraction_mod"] < 0].tolist() == [9] and np.isnan(t.fraction_mod).sum() == 1
    t.fraction_mod[np.isnan(t.fraction_mod)] = -1.0
    assert np.array_equal(cols["contig"][live], lut[t.contig])
    assert np.array_equal(cols["position"][live], t.position) and np.array_equal(cols["mod_type"][live], t.mod_type)
    assert np.array_equal(cols["strand"][live], t.strand) and np.array_equal(cols["fraction_mod"][live], t.fraction_mod)
    assert np.array_equal(cols["nvalid_cov"][live], t.nvalid_cov)
    nat.close()


def test_native_fasta_matches_the_line_loop(tmp_path):
    """nm_fasta_open (fasta.load_fasta) against the pure-Python reading of the same file: wrapped lines, CRLF, blank
    lines, lower case, header descriptions, text before the first header, gzip; and the reference's two assertions."""
    import gzip as gz
    from nanomotif_amd import fasta
    rng = np.random.default_rng(2)
    recs = []
    for i in range(7):
        n = int(rng.integers(1, 5000))
        s = "".join(rng.choice(list("ACGTNRYacgtn"), size=n))
        recs.append((f"contig_{i}" + ("" if i % 2 else f" len={n} some text"), s))
    txt = "ignored line before any header\n"
    for k, (hdr, s) in enumerate(recs):
        w = [60, 80, 7, 10_000][k % 4]
        eol = "\r\n" if k == 2 else "\n"
        txt += ">" + hdr + eol + eol.join(s[j:j + w] for j in range(0, len(s), w)) + eol + ("\n" if k == 3 else "")
    p = tmp_path / "a.fasta"
    p.write_text(txt, newline="")
    want = {name: np.frombuffer(seq.upper().encode(), np.uint8) for name, seq in fasta.read_fasta_names_and_seqs(str(p))}
    got = fasta.load_fasta(str(p))
    assert list(got) == list(want) == [h.split()[0] for h, _ in recs]
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    with gz.open(str(p) + ".gz", "wt", newline="") as f:
        f.write(txt)
    got_gz = fasta.load_fasta(str(p) + ".gz")
    assert list(got_gz) == list(want) and all(np.array_equal(got_gz[k], want[k]) for k in want)
    bad = tmp_path / "bad.fasta"
    bad.write_text(">x\nACGTXACGT\n")
    with pytest.raises(AssertionError, match="ATGCRYSWKMBDHVN"):
        fasta.load_fasta(str(bad))
    bad.write_text(">x\nACGT\n>empty\n>y\nAC\n")
    with pytest.raises(AssertionError, match="must not be empty"):
        fasta.load_fasta(str(bad))


def test_tabix_indexed_read_equals_filtered_full_read(tmp_path):
    """nm_bed_open_indexed: only the blocks of the wanted contigs are inflated; rows equal the full read restricted to
    those contigs (the reference's bgzip path fetches per bin through the index, dataload.py:102-152)."""
    from helpers import write_bgzf_tabix
    spec = synth.SynthSpec(n_contigs=12, total_bp=600_000, n_bins=4, mod_types=("a", "m"), seed=67, min_contig_bp=20_000)
    mg = synth.make_metagenome(spec)
    path = str(tmp_path / "p.bed")
    mg.write_bed(path)
    raw = open(path, "rb").read()
    gz = path + ".gz"
    write_bgzf_tabix(raw, gz, block_size=40_000)
    full = pp.NativePileup(gz)
    assert not full.indexed
    cols_full = {k: v.copy() for k, v in full.ingest_columns(np.arange(len(full.contig_names), dtype=np.uint32)).items()}
    names_full = list(full.contig_names)
    full.close()
    for wanted in ([mg.names[3]], [mg.names[0], mg.names[1], mg.names[7]], [mg.names[11], mg.names[5], "not_in_the_file"], list(mg.names)):
        part = pp.NativePileup(gz, contigs=wanted, index_path=gz + ".tbi")
        assert part.indexed and part.bytes_file > 0
        present = [n for n in names_full if n in set(wanted)]
        assert part.contig_names == present                                     # file order
        if len(present) < len(names_full):
            assert part.bytes_inflated < 0.8 * len(raw)
        cols = part.ingest_columns(np.arange(len(part.contig_names), dtype=np.uint32))
        keep = np.isin(cols_full["contig"], [names_full.index(n) for n in present])
        remap = np.full(len(names_full), -1)
        for i, n in enumerate(present):
            remap[names_full.index(n)] = i
        assert np.array_equal(cols["contig"], remap[cols_full["contig"][keep]])
        for k in ("position", "mod_type", "strand", "fraction_mod", "nvalid_cov"):
            assert np.array_equal(cols[k], cols_full[k][keep]), (wanted, k)
        part.close()
    # something that is not an index: the caller reads the whole file
    open(gz + ".bad.tbi", "wb").close()
    p2 = pp.NativePileup(gz, contigs=[mg.names[0]], index_path=gz + ".bad.tbi")
    assert not p2.indexed and len(p2) == len(cols_full["position"]) and "not a tabix index" in p2.index_problem
    p2.close()
    # a STALE index (the pileup was rewritten with its contigs in another order, the .tbi kept): the regions it names hold
    # other contigs' rows or start off the BGZF blocks — the whole file is read, never a wrong subset
    lines = raw.decode().splitlines(True)
    by_contig = {}
    for ln in lines:
        by_contig.setdefault(ln.split("\t", 1)[0], []).append(ln)
    shuffled = "".join("".join(by_contig[n]) for n in reversed(list(by_contig))).encode()
    gz2 = str(tmp_path / "q.bed.gz")
    write_bgzf_tabix(shuffled, gz2, block_size=40_000)
    import shutil
    shutil.copy(gz + ".tbi", gz2 + ".tbi")
    p3 = pp.NativePileup(gz2, contigs=[mg.names[3]], index_path=gz2 + ".tbi")
    assert not p3.indexed and p3.index_problem and len(p3) == len(cols_full["position"])
    assert sorted(p3.contig_names) == sorted(names_full)
    p3.close()


def test_the_tabix_index_the_reference_ships_is_read_right(tmp_path):
    """datasets/geobacillus-plasmids.pileup.bed.gz.tbi is the one real htslib-made index in the reference tree (fixture: a copy of
    that DATA file).  (1) nm_tabix_regions returns the regions it holds (decoded by hand in tests/helpers.py); (2) a bgzip pileup
    whose blocks lie exactly where that index says is read through it: every wanted contig's rows and nothing else, equal to the
    whole-file read restricted to the contig — the reference fetches a bin's contigs this way (dataload.py:102-152)."""
    import shutil
    from helpers import REF_TBI, REF_TBI_REGIONS, pileup_laid_out_like_the_reference_index
    got = pp.tabix_regions(REF_TBI, ["contig_2", "nope", "contig_3"])
    assert got == {k: ((b[0] << 16) | b[1], (e[0] << 16) | e[1]) for k, (b, e) in REF_TBI_REGIONS.items()}
    from nanomotif_amd._lib import NmScanError
    open(tmp_path / "x.tbi", "wb").write(b"not an index")
    with pytest.raises(NmScanError, match="not a tabix index"):
        pp.tabix_regions(str(tmp_path / "x.tbi"), ["contig_2"])
    gz = str(tmp_path / "laid_out.bed.gz")
    text = pileup_laid_out_like_the_reference_index(gz)
    assert gzip.open(gz, "rb").read() == text["contig_3"] + text["contig_2"] + text["contig_x"]      # a valid gzip file
    shutil.copy(REF_TBI, gz + ".tbi")
    full = pp.NativePileup(gz)
    assert full.contig_names == ["contig_3", "contig_2", "contig_x"]
    cols_full = {k: v.copy() for k, v in full.ingest_columns(np.arange(3, dtype=np.uint32)).items()}
    full.close()
    for wanted in (["contig_3"], ["contig_2"], ["contig_2", "contig_3"], ["contig_x", "contig_2"]):
        part = pp.NativePileup(gz, contigs=wanted, index_path=gz + ".tbi")
        assert part.indexed, part.index_problem
        present = [n for n in ("contig_3", "contig_2") if n in wanted]
        assert part.contig_names == present
        assert part.bytes_inflated < sum(len(text[n]) for n in present) + 2 * 65536         # only the blocks under the regions
        assert len(part) == sum(text[n].count(b"\n") for n in present)
        cols = part.ingest_columns(np.arange(len(present), dtype=np.uint32))
        keep = np.isin(cols_full["contig"], [("contig_3", "contig_2").index(n) for n in present])
        for k in ("position", "mod_type", "strand", "fraction_mod", "nvalid_cov"):
            assert np.array_equal(cols[k], cols_full[k][keep]), (wanted, k)
        part.close()


def test_native_bgzip_tabix_writer_of_the_bench(tmp_path):
    """libnmsynth's bgzip + tabix writer (what bench.py uses to make the 7.5 GB .gz input of its CLI leg) against the
    Python writer of tests/helpers.py: the same compressed bytes (both are zlib level 6), and the same rows through the index."""
    from helpers import write_bgzf_tabix
    from nanomotif_amd import e2e_synth
    spec = synth.SynthSpec(n_contigs=5, total_bp=400_000, n_bins=2, mod_types=("a", "m"), seed=62, min_contig_bp=30_000)
    mg = synth.make_metagenome(spec)
    path = str(tmp_path / "p.bed")
    mg.write_bed(path)
    raw = open(path, "rb").read()
    for bs in (0xFF00, 5_000):
        a, b = str(tmp_path / f"py{bs}.bed.gz"), str(tmp_path / f"native{bs}.bed.gz")
        write_bgzf_tabix(raw, a, block_size=bs)
        e2e_synth.bgzip_tabix(path, b, threads=3, block_size=bs)
        assert open(a, "rb").read() == open(b, "rb").read()
        whole = pp.NativePileup(b)
        assert len(whole) == 400_000
        whole.close()
        wanted = [mg.names[3], mg.names[0]]
        rows = []
        for gz in (a, b):
            t = pp.NativePileup(gz, contigs=wanted, index_path=gz + ".tbi")
            assert t.indexed and set(t.contig_names) == set(wanted)
            c = t.ingest_columns(np.arange(len(t.contig_names), dtype=np.uint32))
            rows.append((list(t.contig_names), {k: v.copy() for k, v in c.items()}, t.bytes_inflated))
            t.close()
        assert rows[0][0] == rows[1][0] and rows[0][2] == rows[1][2]
        for k in rows[0][1]:
            assert np.array_equal(rows[0][1][k], rows[1][1][k]), k


def test_streaming_bgzip_writer_of_the_bench_equals_the_file_writer(tmp_path):
    """libnmsynth's part-by-part writer (nm_synth_bgz_*: rows -> one BGZF text stream + tabix index, what bench.py --extras cli1g
    uses for the 1 Gbp pileup, whose 75 GB of text never exist as a file) against text file -> nm_synth_bgzip: identical .gz,
    identical .tbi, identical text twin — also when a part ends in the middle of a contig and of a 16 kb index window."""
    import ctypes as C
    from nanomotif_amd import e2e_synth
    spec = synth.SynthSpec(n_contigs=6, total_bp=500_000, n_bins=2, mod_types=("a", "m"), seed=64, min_contig_bp=30_000)
    mg = synth.make_metagenome(spec)
    path = str(tmp_path / "p.bed")
    mg.write_bed(path)
    cols = {k: [] for k in ("contig", "position", "mod", "strand", "cov", "pct")}
    for i in range(len(mg.names)):                                     # the rows in write_bed's order
        rows = []
        for mt in mg.spec.mod_types:
            pl = mg.contig_pileup(i, mt)
            rows += [(int(a), mt, int(b), int(c), int(d)) for a, b, c, d in zip(pl["position"], pl["strand"], pl["nvalid"], pl["pct_hundredths"])]
        rows.sort(key=lambda r: (r[0], r[1]))
        for pos, mt, st, cov, pct in rows:
            for k, v in zip(cols, (i, pos, pp.MOD_TYPES.index(mt), st, cov, pct)):
                cols[k].append(v)
    arr = {"contig": np.array(cols["contig"], np.uint32), "position": np.array(cols["position"], np.uint32), "mod": np.array(cols["mod"], np.int8),
           "strand": np.array(cols["strand"], np.uint8), "cov": np.array(cols["cov"], np.int32), "pct": np.array(cols["pct"], np.int32)}
    n = len(arr["contig"])
    names = "".join(mg.names).encode()
    off = np.zeros(len(mg.names) + 1, dtype=np.uint32)
    np.cumsum([len(x) for x in mg.names], out=off[1:])
    lib = e2e_synth.synth_lib()
    ptr = lambda a, t: a.ctypes.data_as(C.POINTER(t))
    for bs, cuts in ((0xFF00, [0, n]), (0xFF00, [0, n // 3 + 7, n // 3 + 8, 2 * n // 3, n]), (3_000, [0, 1, n // 2, n])):
        ref = str(tmp_path / f"ref{bs}.bed.gz")
        e2e_synth.bgzip_tabix(path, ref, threads=3, block_size=bs)
        out = str(tmp_path / f"stream{bs}_{len(cuts)}.bed.gz")
        h = C.c_void_p()
        assert lib.nm_synth_bgz_open(out.encode(), (out + ".txt").encode(), 3, 6, bs, C.byref(h)) == 0
        for a, b in zip(cuts, cuts[1:]):
            part = {k: np.ascontiguousarray(v[a:b]) for k, v in arr.items()}
            assert lib.nm_synth_bgz_append_rows(h, b - a, len(mg.names), names, ptr(off, C.c_uint32), ptr(part["contig"], C.c_uint32),
                                                ptr(part["position"], C.c_uint32), ptr(part["mod"], C.c_int8), ptr(part["strand"], C.c_uint8),
                                                ptr(part["cov"], C.c_int32), ptr(part["pct"], C.c_int32)) == 0, lib.nm_synth_last_error()
        tb, gb = C.c_uint64(0), C.c_uint64(0)
        assert lib.nm_synth_bgz_close(h, C.byref(tb), C.byref(gb)) == 0
        assert open(out + ".txt", "rb").read() == open(path, "rb").read() and tb.value == os.path.getsize(path)
        assert open(out, "rb").read() == open(ref, "rb").read() and gb.value == os.path.getsize(ref)
        assert open(out + ".tbi", "rb").read() == open(ref + ".tbi", "rb").read()


def test_ingest_columns_may_be_asked_for_twice(tmp_path):
    """nm_bed_ingest_columns gives up the reader's 64-bit originals on the first call; a second call (another contig map) used
    to walk the freed vectors."""
    spec = synth.SynthSpec(n_contigs=3, total_bp=100_000, n_bins=1, mod_types=("a",), seed=63, min_contig_bp=20_000)
    mg = synth.make_metagenome(spec)
    path = str(tmp_path / "p.bed")
    mg.write_bed(path)
    t = pp.NativePileup(path)
    a = {k: v.copy() for k, v in t.ingest_columns(np.arange(3, dtype=np.uint32)).items()}
    b = {k: v.copy() for k, v in t.ingest_columns(np.array([2, 0, 1], dtype=np.uint32)).items()}
    t.close()
    for k in a:
        if k != "contig":
            assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(np.array([2, 0, 1], dtype=np.uint32)[a["contig"]], b["contig"])


def test_a_block_whose_text_does_not_match_its_checksum_is_refused(tmp_path):
    """A flipped byte inside a STORED deflate block leaves a valid stream of the right size: only the member's CRC-32 tells
    (Python's gzip and htslib, the reference's readers, raise on it) — whole file and tabix subset."""
    from helpers import write_bgzf_tabix
    from nanomotif_amd._lib import NmScanError
    spec = synth.SynthSpec(n_contigs=3, total_bp=60_000, n_bins=1, mod_types=("a",), seed=5, min_contig_bp=10_000)
    mg = synth.make_metagenome(spec)
    bed = str(tmp_path / "p.bed")
    mg.write_bed(bed)
    text = open(bed, "rb").read()
    gz = str(tmp_path / "p.bed.gz")
    write_bgzf_tabix(text, gz, block_size=20_000, level=0)
    raw = bytearray(open(gz, "rb").read())
    good = pp.NativePileup(gz)
    n = len(good)
    good.close()
    at = raw.find(b"\t255,0,0\t", len(raw) // 2)              # a digit of the colour column in some block of the second half
    assert at > 0 and raw[at + 1:at + 2] == b"2"
    raw[at + 1] = ord("1")
    bad = str(tmp_path / "bad.bed.gz")
    open(bad, "wb").write(bytes(raw))
    import os
    os.replace(gz + ".tbi", bad + ".tbi")
    with pytest.raises(NmScanError, match="CRC-32"):
        pp.NativePileup(bad)
    with pytest.raises(NmScanError, match="CRC-32"):
        pp.NativePileup(bad, contigs=list(mg.names), index_path=bad + ".tbi")
    assert n > 0


def _claim_oversized_block(gz_path, out_path):
    """Rewrite the ISIZE field of a block in the second half of a BGZF file to 70 000 bytes (more than the format's 64 KiB)."""
    raw = bytearray(open(gz_path, "rb").read())
    off, blocks = 0, []
    while off < len(raw):
        bsize = struct.unpack_from("<H", raw, off + 16)[0] + 1
        blocks.append((off, bsize))
        off += bsize
    o, b = blocks[len(blocks) // 2]
    struct.pack_into("<I", raw, o + b - 4, 70_000)
    open(out_path, "wb").write(bytes(raw))


def test_a_block_that_claims_more_than_64k_of_text_is_refused(tmp_path):
    """BGZF blocks hold at most 64 KiB of text; the device inflate sizes its scratch for that (round-4 advisor finding), so a
    trailer that claims more makes the file "not BGZF": the host reader's gzip path then trips over the wrong length, the
    tabix path says corrupt block — neither reads past a buffer."""
    from helpers import write_bgzf_tabix
    from nanomotif_amd._lib import NmScanError
    spec = synth.SynthSpec(n_contigs=3, total_bp=60_000, n_bins=1, mod_types=("a",), seed=6, min_contig_bp=10_000)
    mg = synth.make_metagenome(spec)
    bed = str(tmp_path / "p.bed")
    mg.write_bed(bed)
    gz = str(tmp_path / "p.bed.gz")
    write_bgzf_tabix(open(bed, "rb").read(), gz, block_size=20_000)
    bad = str(tmp_path / "bad.bed.gz")
    _claim_oversized_block(gz, bad)
    import os
    os.replace(gz + ".tbi", bad + ".tbi")
    with pytest.raises(NmScanError):
        pp.NativePileup(bad)
    with pytest.raises(NmScanError, match="corrupt BGZF block"):
        pp.NativePileup(bad, contigs=list(mg.names), index_path=bad + ".tbi")


def test_names_that_are_not_utf8_are_refused_with_a_message(tmp_path):
    """polars (dataload.py:72-100) refuses a pileup that is not UTF-8; the native readers hand the bytes through and the
    Python side names the place instead of dying in a UnicodeDecodeError."""
    from nanomotif_amd import fasta
    from nanomotif_amd._lib import NmScanError
    bed = str(tmp_path / "p.bed")
    open(bed, "wb").write(b"c\xff\x9f\t0\t1\ta\t1\t+\t0\t1\t0\t9\t1.0\t0\t0\t0\t0\t0\t0\t0\n")
    with pytest.raises(NmScanError, match="contig name of the pileup is not valid UTF-8"):
        pp.NativePileup(bed)
    with pytest.raises(NmScanError, match="not valid UTF-8"):
        pp.load_pileup(bed)
    fa = str(tmp_path / "a.fasta")
    open(fa, "wb").write(b">ok\nACGT\n>b\xfe\xff\nACGT\n")
    for reader in (fasta.load_fasta, fasta.read_fasta_names):
        with pytest.raises(NmScanError, match="record name of the assembly is not valid UTF-8"):
            reader(fa)


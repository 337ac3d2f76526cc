"""The assembly FASTA parsed on the device (nm_fasta_parse_device, csrc/nmfasta.hip) against the native host reader
(nm_fasta_open — itself pinned to the line loop of fasta.py:35-49 by tests/test_bed_reader.py::test_native_fasta_matches_the_line_loop
and to the reference's test_fasta.py values by tests/test_reference_kats.py): record names, lengths and every base, byte for
byte; then the planes packed from the parser's device buffer against the planes packed from host arrays, through scoring."""
import gzip
import os
import subprocess
import sys

import numpy as np
import pytest

from nanomotif_amd import fasta, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _random_fasta(rng, n_records, max_len, width=None, crlf=False, lower=0.0, iupac=0.0, n_runs=0.0, final_newline=True, blank_lines=0.0,
                  preamble=b"", descriptions=True):
    out = [preamble]
    for k in range(n_records):
        ln = int(rng.integers(1, max_len + 1))
        seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ln)
        if iupac:
            at = rng.random(ln) < iupac
            seq[at] = rng.choice(np.frombuffer(b"RYSWKMBDHVN", dtype=np.uint8), size=int(at.sum()))
        if n_runs and ln > 50 and rng.random() < n_runs:
            a = int(rng.integers(0, ln - 20))
            seq[a:a + int(rng.integers(1, 20))] = ord("N")
        if lower:
            at = rng.random(ln) < lower
            seq[at] |= 0x20
        seq = seq.tobytes()
        head = b">" + (b"  \t" if rng.random() < 0.1 else b"") + f"rec_{k}".encode()
        if descriptions and rng.random() < 0.5:
            head += rng.choice([b" ", b"\t", b"  "]) + b"len=%d some description > with a greater-than sign" % ln
        if rng.random() < 0.03:
            head += b" " + b"x" * int(rng.integers(60, 40_000))          # header lines longer than a wave's step, than a tile
        eol = b"\r\n" if crlf else b"\n"
        out.append(head + eol)
        w = width if width else int(rng.integers(1, 200))
        for a in range(0, ln, w):
            out.append(seq[a:a + w] + eol)
            if blank_lines and rng.random() < blank_lines:
                out.append(eol)
    text = b"".join(out)
    if not final_newline:
        text = text.rstrip(b"\r\n")
    return text


def _assert_same(dev, host, what):
    assert list(dev) == list(host), what
    for name in host:
        assert dev.length(name) == len(host[name]), (what, name)
    for name in host:
        assert np.array_equal(dev[name], host[name]), (what, name)
    assert dev.total_bp >= sum(len(v) for v in host.values())           # (a repeated name keeps both records in the packed buffer)


def test_device_fasta_equals_the_host_reader(tmp_path):
    from nanomotif_amd.engine import ScanEngine
    eng = ScanEngine(0)
    rng = np.random.default_rng(7)
    cases = [
        ("plain60", dict(n_records=40, max_len=5_000, width=60)),
        ("one_line_records", dict(n_records=30, max_len=100_000, width=1 << 30)),
        ("crlf", dict(n_records=40, max_len=3_000, width=70, crlf=True)),
        ("lower_iupac_n", dict(n_records=50, max_len=8_000, lower=0.3, iupac=0.02, n_runs=0.5)),
        ("no_final_newline", dict(n_records=9, max_len=4_000, width=80, final_newline=False)),
        ("crlf_no_final_newline", dict(n_records=9, max_len=4_000, width=80, crlf=True, final_newline=False)),
        ("blank_lines", dict(n_records=30, max_len=2_000, blank_lines=0.2)),
        ("text_before_the_first_header", dict(n_records=10, max_len=2_000, preamble=b"; a comment\nACGTACGT\n\n")),
        ("tiny_records", dict(n_records=6_000, max_len=12, width=5)),              # hundreds of headers per 16 KiB tile
        ("width1", dict(n_records=20, max_len=300, width=1)),
        ("one_big_record", dict(n_records=1, max_len=3_000_000, width=61)),
        ("random_widths", dict(n_records=300, max_len=20_000)),
    ]
    for name, kw in cases:
        text = _random_fasta(rng, **kw)
        path = str(tmp_path / f"{name}.fasta")
        open(path, "wb").write(text)
        host = fasta.load_fasta(path)
        for threads in (0, 1, 3):
            dev = fasta.DeviceAssembly(eng, path, threads=threads)
            _assert_same(dev, host, (name, threads))
            dev.close()
    # several pinned slabs (32 MiB each), a record that straddles them, a file whose size is a multiple of the tile
    big = _random_fasta(rng, n_records=200, max_len=900_000, width=80, lower=0.1)
    assert len(big) > 70 << 20
    big = big[:(len(big) // 16384) * 16384 - 1] + b"\n"
    path = str(tmp_path / "big.fasta")
    open(path, "wb").write(big)
    host = fasta.load_fasta(path)
    dev = fasta.DeviceAssembly(eng, path)
    _assert_same(dev, host, "big")
    dev.close()
    # a repeated record name: the later record wins in both
    path = str(tmp_path / "dup.fasta")
    open(path, "wb").write(b">a\nACGT\n>b\nGG\n>a x\nTTTTT\n")
    host, dev = fasta.load_fasta(path), fasta.DeviceAssembly(eng, path)
    assert list(dev) == list(host) == ["a", "b"] and bytes(dev["a"]) == bytes(host["a"]) == b"TTTTT"
    dev.close()
    # no record at all
    for name, text in (("empty", b""), ("no_header", b"ACGT\nACGT\n"), ("newlines", b"\n\n\n")):
        path = str(tmp_path / f"{name}.fasta")
        open(path, "wb").write(text)
        dev = fasta.DeviceAssembly(eng, path)
        assert len(dev) == 0 and fasta.load_fasta(path) == {}
        dev.close()
    eng.close()


def test_device_fasta_refuses_what_the_host_reader_refuses(tmp_path):
    """seq.py:68-71 asserts a non-empty sequence over ATGCRYSWKMBDHVN: the FIRST offending record in file order is named, the
    same one by both readers; a gzip file is handed back to the host reader."""
    from nanomotif_amd._lib import NmScanError
    from nanomotif_amd.engine import ScanEngine
    eng = ScanEngine(0)
    bad = {
        "empty_record": b">a\nACGT\n>b\n>c\nAC\n",
        "empty_last": b">a\nACGT\n>b",
        "empty_last_nl": b">a\nACGT\n>b\n\r\n",
        "letter": b">a\nACGT\n>b\nACXGT\n>c\nAC*\n",
        "space": b">a\nAC GT\n",
        "gt_inside": b">a\nAC>GT\n",
        "digit_then_empty": b">a\nAC1\n>b\n",
        "empty_then_letter": b">a\n>b\nAC1\n",
        "lower_bad": b">a\nacgtz\n",
        "nul": b">a\nAC\0GT\n",
    }
    for name, text in bad.items():
        path = str(tmp_path / f"{name}.fasta")
        open(path, "wb").write(text)
        with pytest.raises(AssertionError) as host_err:
            fasta.load_fasta(path)
        with pytest.raises(AssertionError) as dev_err:
            fasta.DeviceAssembly(eng, path)
        assert str(dev_err.value) == str(host_err.value), name
    # many records, one bad letter deep inside a long one / one empty record among thousands
    rng = np.random.default_rng(3)
    text = bytearray(_random_fasta(rng, n_records=400, max_len=30_000, width=60, descriptions=False))
    at = len(text) // 2
    while text[at] not in b"ACGT":
        at += 1
    text[at] = ord("!")
    path = str(tmp_path / "deep.fasta")
    open(path, "wb").write(bytes(text))
    with pytest.raises(AssertionError) as host_err:
        fasta.load_fasta(path)
    with pytest.raises(AssertionError) as dev_err:
        fasta.DeviceAssembly(eng, path)
    assert str(dev_err.value) == str(host_err.value) and "rec_" in str(dev_err.value)
    gz = str(tmp_path / "a.fasta.gz")
    with gzip.open(gz, "wb") as f:
        f.write(b">a\nACGT\n")
    with pytest.raises(NmScanError, match="use nm_fasta_open"):
        fasta.DeviceAssembly(eng, gz)
    with pytest.raises(NmScanError, match="cannot open assembly"):
        fasta.DeviceAssembly(eng, str(tmp_path / "nope.fasta"))
    eng.close()


def test_planes_packed_from_the_parsed_file_score_like_planes_packed_from_host_arrays(tmp_path):
    """nm_upload_contigs_fasta (any subset, any order, a record twice) against nm_upload_contigs on the host reader's arrays:
    same counts for every candidate, same count of letters outside ACGTN, same valid-start counts."""
    from nanomotif_amd.engine import ScanEngine
    from nanomotif_amd.motif import Motif
    spec = synth.SynthSpec(n_contigs=9, total_bp=600_000, n_bins=3, mod_types=("a", "m"), seed=21, min_contig_bp=20_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    path = str(tmp_path / "a.fasta")
    with open(path, "wb") as f:                                     # lower case, IUPAC letters, N runs, CRLF, 61-column lines
        rng = np.random.default_rng(5)
        for i, name in enumerate(mg.names):
            seq = bytearray(mg.contig_ascii(i).tobytes())
            for at in rng.integers(0, len(seq), 30):
                seq[at] = ord("RYKMN"[int(rng.integers(5))])
            low = rng.random(len(seq)) < 0.2
            arr = np.frombuffer(bytes(seq), dtype=np.uint8).copy()
            arr[low] |= 0x20
            seq = arr.tobytes()
            f.write(b">" + name.encode() + b" description\r\n")
            f.write(b"".join(seq[a:a + 61] + b"\r\n" for a in range(0, len(seq), 61)))
    host = fasta.load_fasta(path)
    order = [mg.names[i] for i in (4, 0, 7, 2, 8, 3)]              # a subset, not in file order
    bins = [mg.bin_names[mg.names.index(n)] for n in order]
    motifs = [("GATC", 1, "a"), ("A", 0, "a"), ("CC[AT]GG", 1, "m"), ("G[AG].GAAG[CT]", 5, "a"), ("C", 0, "m"), ("." * 19 + "GATC" + "." * 18, 20, "a")]
    results = []
    for mode in ("host", "device", "device_alias"):
        eng = ScanEngine(0)
        names, bin_of = list(order), list(bins)
        if mode == "host":
            eng.upload_assembly(names, [host[n] for n in names], bin_of)
        else:
            dev = fasta.DeviceAssembly(eng, path)
            if mode == "device_alias":                               # one record under a second name, in another bin
                alias = order[1] + fasta.ALIAS_SEP + "other"
                dev.alias(alias, order[1])
                names, bin_of = names + [alias], bin_of + ["other"]
            eng.upload_assembly_fasta(dev, names, bin_of)
            assert np.array_equal(eng.contig_lengths, [len(host[fasta.original_name(n)]) for n in names])
            dev.close()
        for mt in ("a", "m"):
            first = True
            for k, n in enumerate(names):
                i = mg.names.index(fasta.original_name(n))
                p = mg.contig_pileup(i, mt)
                eng.upload_pileup(mt, np.full(len(p["position"]), k, np.uint32), p["position"], p["strand"], synth.pct_to_fraction(p["pct_hundredths"]),
                                  append=not first)
                first = False
        cands = [(Motif(s, p), mt, b) for b in sorted(set(bin_of)) for s, p, mt in motifs]
        results.append((eng.score(cands), eng.other_letters(), eng.contig_base_counts("A", 20).tolist(), bin_of))
        eng.close()
    assert np.array_equal(results[0][0], results[1][0]) and results[0][0].sum() > 0
    assert results[0][1] == results[1][1] > 0 and results[0][2] == results[1][2]
    # the aliased record scores in its second bin exactly like it does alone
    n_m = len(motifs)
    bins_alias = sorted(set(results[2][3]))
    k_other = bins_alias.index("other")
    eng = ScanEngine(0)
    eng.upload_assembly([order[1]], [host[order[1]]], ["other"])
    for mt in ("a", "m"):
        p = mg.contig_pileup(mg.names.index(order[1]), mt)
        eng.upload_pileup(mt, np.zeros(len(p["position"]), np.uint32), p["position"], p["strand"], synth.pct_to_fraction(p["pct_hundredths"]))
    alone = eng.score([(Motif(s, p), mt, "other") for s, p, mt in motifs])
    eng.close()
    assert np.array_equal(results[2][0][k_other * n_m:(k_other + 1) * n_m], alone)


@pytest.mark.timeout(600)
def test_cli_with_the_device_fasta_parser_writes_the_same_files(tmp_path):
    """`nanomotif motif_discovery` with the assembly parsed on the device (default) and on the host (NANOMOTIF_HOST_FASTA=1): same
    bin-motifs.tsv, and the run says which parser it used."""
    import json
    spec = synth.SynthSpec(n_contigs=8, total_bp=1_600_000, n_bins=2, mod_types=("a", "m"), seed=77, min_contig_bp=50_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    mg.write_fasta(str(tmp_path / "assembly.fasta"))
    raw = open(str(tmp_path / "assembly.fasta"), "rb").read()
    open(str(tmp_path / "assembly.fasta"), "wb").write(raw.lower().replace(b"\n", b"\r\n").rstrip(b"\r\n"))      # lower case, CRLF, no final newline
    mg.write_bed(str(tmp_path / "pileup.bed"))
    mg.write_contig_bin(str(tmp_path / "contig_bin.tsv"))
    texts = {}
    for leg, env_extra in (("device", {}), ("host", {"NANOMOTIF_HOST_FASTA": "1"})):
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **env_extra)
        r = subprocess.run([sys.executable, "-m", "nanomotif_amd", "motif_discovery", "assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv",
                            "--out", "out_" + leg], cwd=str(tmp_path), env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        texts[leg] = open(str(tmp_path / ("out_" + leg) / "bin-motifs.tsv")).read()
        t = json.load(open(str(tmp_path / ("out_" + leg) / "logs" / "timings.motif_discovery.json")))
        assert t["assembly_parser"] == leg
    assert texts["device"] == texts["host"] and texts["device"].count("\n") > 2


def test_parser_buffers_pinned_ahead_of_time(tmp_path):
    """nm_warm_file_parsers pins the parsers' host buffers before they are needed (with or without a ctx; the command line does it on a thread
    beside the engine's creation): a parse afterwards gives the same table, bad arguments are refused, nm_block_cache(0) releases what is idle."""
    import ctypes as C
    from nanomotif_amd import _lib, fasta
    from nanomotif_amd.engine import ScanEngine
    lib = _lib.load()
    size = (32 << 20) + (1 << 16)
    assert lib.nm_warm_file_parsers(None, size, 3) == 0                       # no ctx: the buffers only
    eng = ScanEngine(0)
    try:
        assert lib.nm_warm_file_parsers(eng.ctx, size, 2) == 0                # with a ctx: + one transfer each way on its copy stream
        assert lib.nm_warm_file_parsers(eng.ctx, size, 9) != 0 and b"at most 8" in lib.nm_last_error()
        assert lib.nm_warm_file_parsers(eng.ctx, 1 << 30, 1) != 0
        path = str(tmp_path / "a.fasta")
        with open(path, "w") as f:
            for k in range(20):
                f.write(f">c{k} x\n" + "ACGTTGCAAC" * (50 + k) + "\n")
        a = fasta.DeviceAssembly(eng, path)
        assert list(a)[:3] == ["c0", "c1", "c2"] and a.length("c19") == 10 * 69 and bytes(a["c0"][:10]) == b"ACGTTGCAAC"
        a.close()
    finally:
        eng.close()
    assert lib.nm_block_cache(0, 0, None) == 0                                # (no cache installed: releases the idle pinned buffers, no error)

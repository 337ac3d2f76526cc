"""End to end on the GPU: `nanomotif motif_discovery` (text pileup, FASTA, contig-bin TSV in; bin-motifs.tsv out)
against the CPU oracle's full pipeline on the same data — cfg 1 of BASELINE.json (CLI plumbing) with the packaged
geobacillus motifs planted in synthetic contigs (the reference's own pileup blob is not distributable)."""
import gzip
import os
import re
import subprocess
import sys

import pytest

from helpers import oracle_pipeline
from nanomotif_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_cli(tmp, args, nproc=1, check=True, env_extra=None):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), **(env_extra or {}))
    cmd = [sys.executable, "-m", "nanomotif_amd", "motif_discovery"] + args
    if nproc > 1:      # several ranks on the one GPU of the test box: gloo carries the all-reduces
        env["NANOMOTIF_DIST_BACKEND"] = "gloo"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "-m", "nanomotif_amd", "motif_discovery", "--device", "0"] + args
    r = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True)
    assert not check or r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r


def test_motif_discovery_cli_matches_oracle(tmp_path):
    spec = synth.SynthSpec(n_contigs=4, total_bp=500_000, n_bins=2, mod_types=("a", "m"), seed=61, min_contig_bp=60_000,
                           fixed_motifs=(("GATC", 1, "a"), ("ACCCA", 4, "a"), ("GRNGAAGY", 5, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    mg.write_contig_bin(tmp + "/contig_bin.tsv")
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out", "-t", "1"])
    got = open(tmp + "/out/bin-motifs.tsv").read()
    expect = oracle_pipeline(mg)
    assert got == expect, f"\n--- gpu ---\n{got}\n--- oracle ---\n{expect}"
    motifs = {l.split("\t")[1] for l in got.strip().split("\n")[1:]}
    assert {"GATC", "ACCCA", "CCWGG"} <= motifs
    assert os.path.exists(tmp + "/out/args.motif_discovery.json") and os.path.exists(tmp + "/out/logs/motif_discovery.main.log")
    assert os.path.isdir(tmp + "/out/precleanup-motifs")
    # the per-task files (find_motifs_bin.py:521-596; written at the end of the run by a few threads): five precleanup tables, the
    # background PSSM (np.savetxt's text: 4 rows of 41 "%.4f") and the search graph of every task that found something
    b0 = sorted(set(mg.bin_names))[0]
    stages = sorted(os.listdir(f"{tmp}/out/precleanup-motifs/{b0}-a"))
    assert stages == sorted(n + ".tsv" for n in ("motifs", "motifs-noise", "motifs-noise-merge", "motifs-noise-merge-sub", "motifs-noise-merge-sub-complement"))
    assert open(f"{tmp}/out/precleanup-motifs/{b0}-a/motifs.tsv").readline().startswith("reference\tmotif\tmod_type")
    pssm = open(f"{tmp}/out/temp/{b0}/background_pssm.txt").read().splitlines()
    assert len(pssm) == 4 and all(len(row.split(" ")) == 41 and all(len(x.split(".")[1]) == 4 for x in row.split(" ")) for row in pssm)
    assert abs(sum(float(row.split(" ")[0]) for row in pssm) - 1.0) < 1e-3
    assert open(f"{tmp}/out/temp/{b0}/motif_graph_a.gml").read().startswith("graph [\n  directed 1\n")
    # the engine context of a single-rank command is made on a thread while the interpreter imports (__main__.py) and adopted by the run;
    # NANOMOTIF_NO_EARLY_INIT=1 makes it the regular way: same table
    import json
    assert json.load(open(tmp + "/out/logs/timings.motif_discovery.json"))["engine_context_made_beside_the_imports"] is True
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out_regular_start", "-t", "1"], env_extra={"NANOMOTIF_NO_EARLY_INIT": "1"})
    assert json.load(open(tmp + "/out_regular_start/logs/timings.motif_discovery.json"))["engine_context_made_beside_the_imports"] is False
    assert open(tmp + "/out_regular_start/bin-motifs.tsv").read() == got
    # the five precleanup tables of every task are formatted natively (nm_post_tables); NANOMOTIF_PY_TABLES=1: by postprocess.format_motifs
    # from row objects, as before round 6 — the same files, byte for byte
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out_py_tables", "-t", "1"],
             env_extra={"NANOMOTIF_PY_TABLES": "1", "NANOMOTIF_PY_GML": "1"})
    for b in sorted(os.listdir(tmp + "/out/temp")):                      # (the search graphs: nm_search_result_gml against the Python text)
        if os.path.isdir(f"{tmp}/out/temp/{b}"):
            assert sorted(os.listdir(f"{tmp}/out/temp/{b}")) == sorted(os.listdir(f"{tmp}/out_py_tables/temp/{b}"))
            for name in os.listdir(f"{tmp}/out/temp/{b}"):
                assert open(f"{tmp}/out/temp/{b}/{name}").read() == open(f"{tmp}/out_py_tables/temp/{b}/{name}").read(), (b, name)
    n_tables = 0
    for task in sorted(os.listdir(tmp + "/out/precleanup-motifs")):
        assert sorted(os.listdir(f"{tmp}/out/precleanup-motifs/{task}")) == sorted(os.listdir(f"{tmp}/out_py_tables/precleanup-motifs/{task}"))
        for name in os.listdir(f"{tmp}/out/precleanup-motifs/{task}"):
            assert open(f"{tmp}/out/precleanup-motifs/{task}/{name}").read() == open(f"{tmp}/out_py_tables/precleanup-motifs/{task}/{name}").read(), (task, name)
            n_tables += 1
    assert n_tables >= 10

    # bgzip'd pileup (needs its .tbi to be present like the reference) and -f bin FASTA files
    with open(tmp + "/pileup.bed", "rb") as f, gzip.open(tmp + "/pileup.bed.gz", "wb") as g:
        g.write(f.read())
    open(tmp + "/pileup.bed.gz.tbi", "wb").close()
    os.makedirs(tmp + "/bins")
    for b in sorted(set(mg.bin_names)):
        with open(f"{tmp}/bins/{b}.fasta", "w") as f:
            for i, name in enumerate(mg.names):
                if mg.bin_names[i] == b:
                    f.write(f">{name} some description\n{mg.contig_str(i)}\n")
    _run_cli(tmp, ["assembly.fasta", "pileup.bed.gz", "-d", "bins", "--out", "out_gz"])
    got_gz = open(tmp + "/out_gz/bin-motifs.tsv").read()
    assert got_gz == oracle_pipeline(mg, bgzip_order=True)
    assert os.path.exists(tmp + "/out_gz/temp/contig_bin.tsv")


def test_cfg1_packaged_geobacillus_assembly(tmp_path):
    """BASELINE cfg 1 / SURVEY §8(d): the reference's check_installation data set — the packaged
    geobacillus-plasmids.assembly.fasta (2 contigs, 176 kbp) and geobacillus-contig-bin.tsv, copied as data fixtures — with
    the modkit pileup (a missing blob in the reference) synthesized ON THOSE contigs, planting the four motifs of the
    packaged expected output (geobacillus-plasmids.bin-motifs.tsv:2-5); `-t 1`.  bin-motifs.tsv must equal the oracle
    pipeline byte for byte and hold those four motifs."""
    import shutil
    from helpers import GOLDEN
    from nanomotif_amd import fasta
    tmp = str(tmp_path)
    shutil.copy(os.path.join(GOLDEN, "data_geobacillus-plasmids.assembly.fasta"), tmp + "/geobacillus-plasmids.assembly.fasta")
    shutil.copy(os.path.join(GOLDEN, "data_geobacillus-contig-bin.tsv"), tmp + "/geobacillus-contig-bin.tsv")
    names, seqs = zip(*fasta.read_fasta_names_and_seqs(tmp + "/geobacillus-plasmids.assembly.fasta"))
    assert list(names) == ["contig_3", "contig_2"] and [len(s) for s in seqs] == [82915, 93311]
    bin_of = dict(line.split("\t") for line in open(tmp + "/geobacillus-contig-bin.tsv").read().split("\n") if line)
    planted = [("GATC", 1, "a"), ("ACCCA", 4, "a"), ("CCAAAT", 4, "a"), ("GRNGAAGY", 5, "a")]
    mg = synth.from_sequences(names, seqs, [bin_of[n] for n in names], planted, seed=1, mod_types=("a",))
    mg.write_bed(tmp + "/geobacillus-plasmids.pileup.bed")
    _run_cli(tmp, ["geobacillus-plasmids.assembly.fasta", "geobacillus-plasmids.pileup.bed", "-c", "geobacillus-contig-bin.tsv",
                   "--out", "out", "-t", "1"])
    got = open(tmp + "/out/bin-motifs.tsv").read()
    expect = oracle_pipeline(mg)
    assert got == expect, f"\n--- gpu ---\n{got}\n--- oracle ---\n{expect}"
    rows = [l.split("\t") for l in got.strip().split("\n")[1:]]
    assert {(r[0], r[1], int(r[2]), r[3]) for r in rows} >= {("bin1", m, p, "a") for m, p, _ in planted[:3]}
    # (on these 176 kbp the search settles on GRNGAAGC, the commoner half of the planted GRNGAAGY — like the oracle)
    assert any(r[1].startswith("GRNGAAG") and int(r[2]) == 5 for r in rows)
    gatc = next(r for r in rows if r[1] == "GATC")
    assert gatc[6] == "palindrome" and int(gatc[4]) > 600                 # the packaged output lists 679 methylated GATC sites


def test_contig_listed_under_two_bins_is_scored_in_both(tmp_path):
    """The reference's contig-bin table is a DataFrame: a contig listed under two bins is a member of both (its pileup rows
    are joined to each bin, find_motifs_bin.py:416-418; fasta.py:122-187).  Expected output: the oracle pipeline on a
    metagenome that holds the contig twice, once per bin, with the same sequence and the same pileup rows."""
    from nanomotif_amd.fasta import ALIAS_SEP
    base = synth.make_metagenome(synth.SynthSpec(n_contigs=5, total_bp=600_000, n_bins=2, mod_types=("a", "m"), seed=71, min_contig_bp=60_000))
    seqs = [base.contig_str(i) for i in range(5)]
    planted = [("GATC", 1, "a"), ("CCWGG", 1, "m"), ("GAATTC", 2, "a")]
    shared = next(i for i, b in enumerate(base.bin_names) if b == "bin_000")
    mg = synth.from_sequences(base.names, seqs, base.bin_names, planted, seed=71, mod_types=("a", "m"))
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/a.fasta")
    mg.write_bed(tmp + "/p.bed")
    with open(tmp + "/cb.tsv", "w") as f:
        for n, b in zip(mg.names, mg.bin_names):
            f.write(f"{n}\t{b}\n")
        f.write(f"{mg.names[shared]}\tbin_001\n")                        # the second listing
        f.write(f"{mg.names[shared]}\tbin_001\n")                        # an exact repeat: ignored with a warning
    r = _run_cli(tmp, ["a.fasta", "p.bed", "-c", "cb.tsv", "--out", "o"])
    assert "it is scored in both" in r.stdout + r.stderr
    twice = synth.from_sequences(list(mg.names) + [mg.names[shared] + ALIAS_SEP + "bin_001"], seqs + [seqs[shared]],
                                 list(mg.bin_names) + ["bin_001"], planted, seed=71, mod_types=("a", "m"))
    twice.key_index = {5: shared}
    got = open(tmp + "/o/bin-motifs.tsv").read()
    assert got == oracle_pipeline(twice)
    assert got != oracle_pipeline(mg)                                     # the extra member changed bin_001's counts


def test_non_default_thresholds_search_and_merge_stage(tmp_path):
    """--methylation_threshold_low/high drive the search; the merge stage stays at 0.3 / 0.7 like the reference."""
    spec = synth.SynthSpec(n_contigs=2, total_bp=400_000, n_bins=1, mod_types=("a",), seed=43, min_contig_bp=150_000,
                           fixed_motifs=(("GATC", 1, "a"), ("ACCCA", 4, "a"), ("CCAAAT", 4, "a"), ("GRNGAAGY", 5, "a")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/a.fasta")
    mg.write_bed(tmp + "/p.bed")
    mg.write_contig_bin(tmp + "/cb.tsv")
    _run_cli(tmp, ["a.fasta", "p.bed", "-c", "cb.tsv", "--out", "o", "--methylation_threshold_low", "0.2",
                   "--methylation_threshold_high", "0.8"])
    got = open(tmp + "/o/bin-motifs.tsv").read()
    assert got == oracle_pipeline(mg, low=0.2, high=0.8)
    assert "GATC" in got


def test_empty_result_writes_header_only(tmp_path):
    mg = synth.make_metagenome(synth.SynthSpec(n_contigs=1, total_bp=60_000, n_bins=1, mod_types=("a",), seed=62, fixed_motifs=()))
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/a.fasta")
    mg.write_bed(tmp + "/p.bed")
    mg.write_contig_bin(tmp + "/cb.tsv")
    _run_cli(tmp, ["a.fasta", "p.bed", "-c", "cb.tsv", "--out", "o"])
    lines = open(tmp + "/o/bin-motifs.tsv").read().strip().split("\n")
    assert len(lines) == 1 and lines[0].split("\t")[:4] == ["reference", "motif", "mod_position", "mod_type"]


def test_two_ranks_give_the_single_rank_output(tmp_path):
    """Contigs, windows and background samples sharded over two ranks (one GPU shared, gloo): same bin-motifs.tsv."""
    spec = synth.SynthSpec(n_contigs=9, total_bp=700_000, n_bins=3, mod_types=("a", "m"), seed=62, min_contig_bp=40_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m"), ("GAAGNNNNNTAC", 2, "a")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    mg.write_contig_bin(tmp + "/contig_bin.tsv")
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out1"])
    one = open(tmp + "/out1/bin-motifs.tsv").read()
    assert len(one.strip().split("\n")) > 3
    # contigs of every bin over both ranks (all-reduce per round) / whole bins per rank (rows gathered at the end)
    for mode in ("contigs", "bins"):
        r = _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out_" + mode, "--shard", mode], nproc=2)
        assert open(f"{tmp}/out_{mode}/bin-motifs.tsv").read() == one, mode
        assert ("whole bins per GPU" in r.stdout + r.stderr) == (mode == "bins")
        assert os.path.isdir(f"{tmp}/out_{mode}/precleanup-motifs")
        assert sorted(os.listdir(f"{tmp}/out_{mode}/precleanup-motifs")) == sorted(os.listdir(tmp + "/out1/precleanup-motifs"))


def test_cli_5mc_and_4mc_in_one_pileup(tmp_path):
    """'m' and '21839' rows share the canonical C (and their positions): two searches per bin on one set of cytosines,
    their rows mixed in the adjacency filter (dataload.py:236-245)."""
    spec = synth.SynthSpec(n_contigs=3, total_bp=450_000, n_bins=1, mod_types=("m", "21839"), seed=65, min_contig_bp=80_000,
                           fixed_motifs=(("CCWGG", 1, "m"), ("GGCC", 2, "21839"), ("CACAG", 1, "21839")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    mg.write_contig_bin(tmp + "/contig_bin.tsv")
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out"])
    got = open(tmp + "/out/bin-motifs.tsv").read()
    assert got == oracle_pipeline(mg)
    mods = {l.split("\t")[3] for l in got.strip().split("\n")[1:]}
    assert mods == {"m", "21839"}


def test_bgzip_pileup_is_read_through_its_tabix_index(tmp_path):
    """A real .tbi next to the bgzip pileup and a contig-bin table that covers HALF of the bins: only the blocks of those
    contigs are inflated (find_motifs_bin.py:233-246 fetches a bin's contigs through the index); same bin-motifs.tsv as
    the oracle's bgzip-order pipeline on those bins."""
    from helpers import write_bgzf_tabix
    spec = synth.SynthSpec(n_contigs=8, total_bp=800_000, n_bins=4, mod_types=("a", "m"), seed=66, min_contig_bp=60_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    write_bgzf_tabix(open(tmp + "/pileup.bed", "rb").read(), tmp + "/pileup.bed.gz", block_size=50_000)
    bins = sorted(set(mg.bin_names))[::2]
    with open(tmp + "/contig_bin.tsv", "w") as f:
        for n, b in zip(mg.names, mg.bin_names):
            if b in bins:
                f.write(f"{n}\t{b}\n")
    r = _run_cli(tmp, ["assembly.fasta", "pileup.bed.gz", "-c", "contig_bin.tsv", "--out", "out"])
    assert "tabix-indexed" in r.stdout + r.stderr and "parsed on the device" in r.stdout + r.stderr      # round 4: bgzip on the device parser
    got = open(tmp + "/out/bin-motifs.tsv").read()
    assert got == oracle_pipeline(mg, bgzip_order=True, bins=set(bins))
    assert {l.split("\t")[0] for l in got.strip().split("\n")[1:]} == set(bins)
    # the host reader on the same files gives the same table
    r = _run_cli(tmp, ["assembly.fasta", "pileup.bed.gz", "-c", "contig_bin.tsv", "--out", "out_host"], env_extra={"NANOMOTIF_HOST_PARSER": "1"})
    assert "parsed on the device" not in r.stdout + r.stderr
    assert open(tmp + "/out_host/bin-motifs.tsv").read() == got


def test_more_ranks_than_contigs(tmp_path):
    """An isolate genome on two ranks with contig sharding: ONE contig, so rank 1 holds nothing — it uploads an empty
    shard, contributes zero tables and empty window sets and still joins every collective (no hang, same output)."""
    spec = synth.SynthSpec(n_contigs=1, total_bp=300_000, n_bins=1, mod_types=("a",), seed=64, fixed_motifs=(("GATC", 1, "a"), ("ACCCA", 4, "a")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    mg.write_contig_bin(tmp + "/contig_bin.tsv")
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out1"])
    one = open(tmp + "/out1/bin-motifs.tsv").read()
    assert "GATC" in one
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out2", "--shard", "contigs"], nproc=2)
    assert open(tmp + "/out2/bin-motifs.tsv").read() == one


def test_wide_search_frames_equal_the_oracle(tmp_path):
    """The reference takes any --search_frame_size (find_motifs_bin.py:128-130).  Frames above 63 use the three-word window
    fields and, once a child reaches more than 63 positions from the modified base, the extra-wide scoring kernels
    (offsets in [-96, 95]); above 191 columns the windows stay on the host and children that reach further than 95 positions
    are scored by nm_score_batch_wide."""
    spec = synth.SynthSpec(n_contigs=3, total_bp=300_000, n_bins=1, mod_types=("a", "m"), seed=64, min_contig_bp=60_000,
                           fixed_motifs=(("GATC", 1, "a"), ("GCACNNNNNNGTT", 2, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/a.fasta")
    mg.write_bed(tmp + "/p.bed")
    mg.write_contig_bin(tmp + "/cb.tsv")
    for frame in (62, 100, 128, 191):
        _run_cli(tmp, ["a.fasta", "p.bed", "-c", "cb.tsv", "--out", f"o{frame}", "--search_frame_size", str(frame)])
        got = open(f"{tmp}/o{frame}/bin-motifs.tsv").read()
        assert got == oracle_pipeline(mg, padding=frame // 2), frame
        assert "GATC" in got
    # a planted pair of half-sites 120 positions apart: its far columns stand out of the background, so the root's children there
    # are scored (nodes longer than 25 are never expanded, find_motifs_bin.py:1010: such a motif is not REPORTED by either side)
    spec = synth.SynthSpec(n_contigs=3, total_bp=300_000, n_bins=1, mod_types=("a", "m"), seed=64, min_contig_bp=60_000,
                           fixed_motifs=(("GATC", 1, "a"), ("GA" + "N" * 120 + "TC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    mg.write_fasta(tmp + "/a.fasta")
    mg.write_bed(tmp + "/p.bed")
    from helpers import write_bgzf_tabix
    write_bgzf_tabix(open(tmp + "/p.bed", "rb").read(), tmp + "/p.bed.gz", block_size=50_000)
    for frame, pileup, bgzip in ((192, "p.bed", False), (300, "p.bed", False), (400, "p.bed.gz", True)):
        r = _run_cli(tmp, ["a.fasta", pileup, "-c", "cb.tsv", "--out", f"w{frame}", "--search_frame_size", str(frame)])
        got = open(f"{tmp}/w{frame}/bin-motifs.tsv").read()
        assert got == oracle_pipeline(mg, padding=frame // 2, bgzip_order=bgzip), frame
        assert "CCWGG" in got and ("GATC" in got) == (frame != 400) and "stay on the host" in r.stdout + r.stderr     # (the oracle loses GATC at 400 too)
        m = re.search(r"(\d+) candidates reaching further than 95 positions", r.stdout + r.stderr)
        assert frame == 192 or (m and int(m.group(1)) > 0), frame          # (the planted far columns lie outside a frame of 192)
    # the same frame on two ranks, contigs sharded (each rank scores its contigs — far-reaching candidates through the wide entry —, the
    # count tables all-reduced; every rank holds all windows on the host): the text of one rank
    for mode in ("contigs", "bins"):
        _run_cli(tmp, ["a.fasta", "p.bed", "-c", "cb.tsv", "--out", f"w300_{mode}", "--search_frame_size", "300", "--shard", mode], nproc=2)
        assert open(f"{tmp}/w300_{mode}/bin-motifs.tsv").read() == open(f"{tmp}/w300/bin-motifs.tsv").read(), mode
    r = _run_cli(tmp, ["a.fasta", "p.bed", "-c", "cb.tsv", "--out", "o", "--search_frame_size", "4096"], check=False)
    assert r.returncode != 0 and "search_frame_size must be at most 4095" in r.stdout + r.stderr


def test_assembly_with_other_iupac_letters_takes_the_host_window_path(tmp_path):
    """A letter outside ACGTN anywhere in the assembly keeps window extraction on the host (where the reference's
    KeyError semantics live); the result is the same bin-motifs.tsv."""
    spec = synth.SynthSpec(n_contigs=6, total_bp=500_000, n_bins=2, mod_types=("a", "m"), seed=63, min_contig_bp=40_000,
                           fixed_motifs=(("GATC", 1, "a"), ("CCWGG", 1, "m")))
    mg = synth.make_metagenome(spec)
    tmp = str(tmp_path)
    mg.write_fasta(tmp + "/assembly.fasta")
    mg.write_bed(tmp + "/pileup.bed")
    mg.write_contig_bin(tmp + "/contig_bin.tsv")
    _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out_device"])
    # one more contig, in a bin of its own and without pileup rows, carrying an 'R'
    with open(tmp + "/assembly.fasta", "a") as f:
        f.write(">extra_contig\n" + "ACGTTGCA" * 500 + "R" + "ACGT" * 300 + "\n")
    with open(tmp + "/contig_bin.tsv", "a") as f:
        f.write("extra_contig\tbin_extra\n")
    r = _run_cli(tmp, ["assembly.fasta", "pileup.bed", "-c", "contig_bin.tsv", "--out", "out_host"])
    assert "window extraction stays on the host" in r.stdout + r.stderr
    assert open(tmp + "/out_host/bin-motifs.tsv").read() == open(tmp + "/out_device/bin-motifs.tsv").read()

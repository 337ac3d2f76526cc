"""Small host-side pieces of the command line that need no GPU: the per-task files written at the end of a run, the lazily made count
tables of the window pipeline, the background PSSM text."""
import os

import numpy as np

from nanomotif_amd import find_motifs_bin as fmb
from nanomotif_amd import postprocess as pp
from nanomotif_amd.model import BetaBernoulliModel


def test_deferred_files_write_what_was_added_in_order(tmp_path):
    files = fmb.DeferredFiles()
    for i in range(200):
        files.add(str(tmp_path / f"f{i}.txt"), f"text {i}\n")
    files.add(str(tmp_path / "f7.txt"), "the later text wins\n")          # (a path added twice: like writing the file twice)
    assert not os.listdir(tmp_path)
    files.flush()
    assert len(os.listdir(tmp_path)) == 200 and open(tmp_path / "f199.txt").read() == "text 199\n"
    assert open(tmp_path / "f7.txt").read() == "the later text wins\n"
    files.flush()                                                            # nothing left: a no-op
    assert files.items == []


def test_format_motifs_is_what_write_motifs_writes(tmp_path):
    rows = [pp.MotifRow("bin_1", "." * 19 + "GATC" + "." * 18, "a", 20, BetaBernoulliModel.from_counts(900, 40), 3.25),
            pp.MotifRow("bin_0", "." * 18 + "CCAGG" + "." * 18, "m", 19, BetaBernoulliModel.from_counts(500, 10), 2.0)]
    pp.write_motifs(rows, str(tmp_path / "m.tsv"))
    text = open(tmp_path / "m.tsv").read()
    assert text == pp.format_motifs(rows)
    lines = text.split("\n")
    assert lines[0].startswith("reference\tmotif\tmod_type") and lines[1].startswith("bin_0\t") and lines[2].startswith("bin_1\t")   # sorted


def test_background_pssm_text_is_savetxt(tmp_path):
    class G:
        def gml_text(self):
            return "graph [\n]\n"
    rng = np.random.default_rng(4)
    pssm = rng.random((4, 41)) * np.array([1e-6, 1.0, 30.0, 1e4]).reshape(4, 1)
    files = fmb.DeferredFiles()
    assert fmb.write_search_artifacts("b", "a", (G(), [], pssm), str(tmp_path / "t"), files)
    files.flush()
    np.savetxt(tmp_path / "ref.txt", pssm, fmt="%.4f")
    assert open(tmp_path / "t" / "background_pssm.txt").read() == open(tmp_path / "ref.txt").read()
    assert open(tmp_path / "t" / "motif_graph_a.gml").read() == "graph [\n]\n"
    assert fmb.write_search_artifacts("b", "a", None, str(tmp_path / "t2"), files) is False and not os.path.exists(tmp_path / "t2")


def test_lazy_tables_make_a_table_once_and_only_when_asked():
    from nanomotif_amd.main import _LazyTables
    made = []

    def make(key):
        made.append(key)
        return {"contig": len(made)}
    t = _LazyTables(["A", "C"], make)
    assert "A" in t and "G" not in t and list(t) == ["A", "C"] and len(t) == 2 and made == []
    assert t["C"] == {"contig": 1} and t["C"] is t["C"] and made == ["C"]
    try:
        t["G"]
        assert False
    except KeyError:
        pass

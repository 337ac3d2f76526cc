"""The pre-selected KL columns of the native search against the all-double-precision path (NM_SEARCH_EXACT_KL=1): the motif
rows of the 1 Gbp end-to-end run must be identical, text for text; timing of the search beside it."""
import os, sys, time
sys.path.insert(0, ".")
import torch
from nanomotif_amd import synth, e2e_synth, postprocess
from nanomotif_amd.engine import ScanEngine
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=10_000, total_bp=1_000_000_000, n_bins=500, mod_types=("a", "m"), seed=1))
out = {}
for mode in ("selected", "exact", "selected", "exact"):
    if mode == "exact":
        os.environ["NM_SEARCH_EXACT_KL"] = "1"
    else:
        os.environ.pop("NM_SEARCH_EXACT_KL", None)
    eng = ScanEngine(0)
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    text = postprocess.format_bin_motifs(rows)
    out.setdefault(mode, text)
    assert out[mode] == text
    print(mode, len(rows), "native_search_s %.4f" % t["native_search_s"], "wall %.4f" % (t["upload_filter_s"] + t["search_s"]), flush=True)
assert out["selected"] == out["exact"], "the pre-selection changed a result"
print("identical bin-motifs text (%d bytes)" % len(out["exact"]))

"""Build-container only (needs /root/reference): motif-sites/s of the REAL reference's motif_model_contig next to the
oracle port on the same core and inputs — shows what bench.py's cpu_baseline (kind "port") stands in for."""
import sys, time, json
sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import numpy as np
import refstub
from nanomotif_amd import synth
from oracle.scan import ContigPileup, motif_model_contig as port_contig
from oracle.model import BetaBernoulliModel as PortModel
from oracle.motif import Motif as PortMotif

pkg = refstub.load_reference()
ref_fmb, ref_motif, ref_model = pkg.find_motifs_bin, pkg.motif, pkg.model
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=1, total_bp=2_000_000, n_bins=1, mod_types=("a",), seed=4, fixed_motifs=(("GATC", 1, "a"),)))
seq = mg.contig_str(0)
p = mg.contig_pileup(0, "a")
keep = p["nvalid"] > 5
frac = synth.pct_to_fraction(p["pct_hundredths"][keep])
frame = refstub.make_pileup([mg.names[0]] * int(keep.sum()), p["position"][keep], [chr(c) for c in p["strand"][keep]], frac)
pile = ContigPileup(p["position"][keep], p["strand"][keep], frac)
cands = synth.random_candidates(12, seed=2, mod_types=("a",))
out = {}
t0 = time.perf_counter()
ref_counts = []
for s, pos, _ in cands:
    m = ref_fmb.motif_model_contig(frame, seq, ref_model.BetaBernoulliModel(), ref_motif.Motif(s, pos))
    ref_counts.append((m._alpha - m._alpha_prior, m._beta - m._beta_prior))
t_ref = time.perf_counter() - t0
t0 = time.perf_counter()
port_counts = []
for s, pos, _ in cands:
    m = port_contig(pile, seq, PortModel(), PortMotif(s, pos))
    port_counts.append((m._alpha - m._alpha_prior, m._beta - m._beta_prior))
t_port = time.perf_counter() - t0
sites = 2 * len(seq) * len(cands)
print(json.dumps({"contig_bp": len(seq), "candidates": len(cands), "reference_s": t_ref, "port_s": t_port,
                  "reference_sites_per_s": sites / t_ref, "port_sites_per_s": sites / t_port, "same_counts": ref_counts == port_counts}))

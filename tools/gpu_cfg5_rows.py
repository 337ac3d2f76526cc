"""GPU half of the all-bins check of the 1 Gbp full loop (cfg 5: 1e9 raw rows -> device filters -> 1 000 searches in lock-step
-> post-processing): runs the product once and writes its `bin-motifs.tsv` text (every bin) + the run's counters to
gpurun_out/cfg5_rows/.  The oracle half runs on CPUs only: `tools/cfg5_all_bins_parity.py` (no GPU-minutes)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from nanomotif_amd import e2e_synth, postprocess, synth
    from nanomotif_amd.engine import ScanEngine
    out = os.path.join("gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "cfg5_rows")       # (a second run into another directory: diff the texts)
    os.makedirs(out, exist_ok=True)
    mg = synth.make_metagenome(synth.config("cfg5"))
    eng = ScanEngine(0)
    t0 = time.time()
    rows, t = e2e_synth.run(mg, eng, torch.device("cuda:0"))
    eng.close()
    text = postprocess.format_bin_motifs([r for r in rows if r.n_mod + r.n_nomod >= 50])
    with open(os.path.join(out, "bin-motifs.tsv"), "w") as f:
        f.write(text)
    info = {k: (float(v) if isinstance(v, float) else v) for k, v in t.items() if isinstance(v, (int, float, str))}
    info.update(bins=len(set(mg.bin_names)), motif_rows=text.count("\n") - 1, wall_incl_generation_s=time.time() - t0)
    with open(os.path.join(out, "run.json"), "w") as f:
        json.dump(info, f, indent=1, sort_keys=True)
    print(json.dumps(info, sort_keys=True))


if __name__ == "__main__":
    main()

"""CPU half of the all-bins check of the 1 Gbp full loop (BASELINE cfg 5): the oracle pipeline (filters -> search -> post-processing,
oracle/pipeline.py) on EVERY bin of the synthetic metagenome, compared bin by bin with the `bin-motifs.tsv` the product wrote on
the GPU box (`tools/gpu_cfg5_rows.py` -> gpurun_out/cfg5_rows/bin-motifs.tsv).  Needs no GPU: the metagenome is a pure function
of its spec.  Resumable: the oracle's text of each finished bin is kept under gpurun_out/cfg5_rows/oracle/.

    python3 tools/cfg5_all_bins_parity.py [procs [first_bin [n_bins]]]   ->  profiles/r5/cfg5_all_bins_parity.txt

The oracle workers are SPAWNED and import this file: everything stays under the __main__ guard."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def split_by_bin(text):
    """bin-motifs.tsv text -> (header, {reference: [lines]})"""
    lines = text.splitlines()
    col = lines[0].split("\t").index("reference")
    out = {}
    for ln in lines[1:]:
        out.setdefault(ln.split("\t")[col], []).append(ln)
    return lines[0], out


def main():
    import multiprocessing as mp
    from helpers import spec_kwargs
    from nanomotif_amd import synth
    from oracle import pipeline as opl
    from oracle import postprocess as opp
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else max(1, (os.cpu_count() or 2) - 2)
    src = os.path.join(ROOT, "gpurun_out", "cfg5_rows", "bin-motifs.tsv")
    header, mine = split_by_bin(open(src).read())
    mg = synth.make_metagenome(synth.config("cfg5"))
    bins = list(dict.fromkeys(mg.bin_names))
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    todo = bins[first:first + (int(sys.argv[3]) if len(sys.argv) > 3 else len(bins))]
    cache = os.path.join(ROOT, "gpurun_out", "cfg5_rows", "oracle")
    os.makedirs(cache, exist_ok=True)
    jobs = [(spec_kwargs(mg.spec), b, {}) for b in todo if not os.path.exists(os.path.join(cache, b + ".tsv"))]
    t0 = time.time()
    if jobs:
        with mp.get_context("spawn").Pool(min(procs, len(jobs))) as pool:
            for k, (b, rows, _) in enumerate(pool.imap_unordered(opl.bin_rows_worker, jobs, chunksize=1)):
                text = opp.format_bin_motifs(rows, min_motifs_bin=50)
                with open(os.path.join(cache, b + ".tsv.tmp"), "w") as f:
                    f.write(text)
                os.replace(os.path.join(cache, b + ".tsv.tmp"), os.path.join(cache, b + ".tsv"))
                if (k + 1) % 10 == 0:
                    print(f"{k + 1} / {len(jobs)} bins through the oracle, {time.time() - t0:.0f} s", flush=True)
    equal, rows_total, first_bad = 0, 0, None
    for b in todo:
        h, exp = split_by_bin(open(os.path.join(cache, b + ".tsv")).read())
        assert h == header, (h, header)
        e, g = exp.get(b, []), mine.get(b, [])
        rows_total += len(e)
        if e == g:
            equal += 1
        elif first_bad is None:
            first_bad = (b, g, e)
    report = [f"cfg 5 full loop at 1 Gbp (1e9 raw rows, 1 000 searches): product bin-motifs.tsv against the oracle pipeline, bin by bin",
              f"bins compared: {len(todo)} of {len(bins)}; byte-equal: {equal}; oracle motif rows: {rows_total}; "
              f"product motif rows in these bins: {sum(len(mine.get(b, [])) for b in todo)}",
              f"bins the product reports that the metagenome does not have: {sorted(set(mine) - set(bins))}"]
    if first_bad:
        b, g, e = first_bad
        report += [f"FIRST DIFFERING BIN {b}", "product:"] + g + ["oracle:"] + e
    out = os.path.join(ROOT, "profiles", "r5", "cfg5_all_bins_parity.txt")
    if len(todo) == len(bins):
        with open(out, "w") as f:
            f.write("\n".join(report) + "\n")
    print("\n".join(report))
    sys.exit(0 if equal == len(todo) and not (set(mine) - set(bins)) else 1)


if __name__ == "__main__":
    main()

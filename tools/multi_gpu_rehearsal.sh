#!/bin/bash
# The first multi-GPU lease in ONE command (round-3 review, item 6).  No multi-GPU node was ever available to this project:
# RCCL has only carried a world of one.  On a node with >= 2 MI355X this runs, in order, and diffs everything against N = 1:
#   1. the real-RCCL test that is skipped on the one-GPU pool (two lanes, four-table ring, nm_allreduce_counts_async / _host)
#      and the native search reducing through nm_allreduce_counts_host (world of one; the N > 1 form is leg 4)
#   2. bench.py --gpus N for N = 2, 4, 8 (strong scaling: ONE metagenome, contigs sharded, one all-reduce per step),
#      strict order (--lanes 1) and two lanes (--lanes 2); every line must carry checksum_matches_n1 = true and rccl.world = N
#   3. the N = 1 line of the same build (reference checksum + the denominator of the scaling figures)
#   4. the CLI on files with --shard contigs (native RCCL reducer inside nm_search_run) and --shard bins, N = 2 and 4:
#      bin-motifs.tsv byte-equal to the one-GPU run
# Usage: bash tools/multi_gpu_rehearsal.sh [out_dir]     (exit code 0 = every leg agreed with N = 1)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
out=${1:-gpurun_out/multi_gpu}
mkdir -p "$out"
export HSA_ENABLE_IPC_MODE_LEGACY=0
n_gpus=$(python3 -c 'import torch; print(torch.cuda.device_count())')
echo "GPUs visible: $n_gpus" | tee "$out/summary.txt"
fail=0
note() { echo "$*" | tee -a "$out/summary.txt"; }

# ---- 1. RCCL tests
timeout 1800 python3 -m pytest tests/test_gpu_comm.py -q -m gpu -rs > "$out/test_gpu_comm.log" 2>&1
rc=$?
note "1. tests/test_gpu_comm.py rc=$rc ($(grep -E 'passed|failed' "$out/test_gpu_comm.log" | tail -1))"
[ $rc -ne 0 ] && fail=1
if [ "$n_gpus" -ge 2 ] && grep -q "SKIPPED.*needs at least 2 GPUs" "$out/test_gpu_comm.log"; then note "   the >= 2-GPU test was SKIPPED although $n_gpus GPUs are visible"; fail=1; fi

# ---- 3. (first, it is the reference) N = 1
timeout 1800 python3 bench.py --gpus 1 --extras e2e > "$out/bench_n1.json" 2> "$out/bench_n1.log" || { note "3. bench N=1 FAILED"; fail=1; }
ref=$(python3 -c "import json,sys; print(json.load(open('$out/bench_n1.json'))['counts_checksum'])" 2>/dev/null)
note "3. N=1 checksum: $ref"

# ---- 2. N = 2, 4, 8
for n in 2 4 8; do
  [ "$n" -gt "$n_gpus" ] && { note "2. N=$n skipped (only $n_gpus GPUs)"; continue; }
  for lanes in 1 2; do
    f="$out/bench_n${n}_lanes${lanes}"
    timeout 1800 python3 bench.py --gpus $n --lanes $lanes --extras none > "$f.json" 2> "$f.log"
    rc=$?
    line=$(python3 - "$f.json" "$n" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    ok = d.get("checksum_matches_n1") is True and d.get("rccl", {}).get("world") == int(sys.argv[2])
    print(("OK " if ok else "MISMATCH ") + f"value={d['value']:.4g} ms_per_step={d['ms_per_step']:.4f} checksum={d.get('counts_checksum')} rccl.world={d.get('rccl', {}).get('world')} "
          f"allreduce_ms={d.get('allreduce_ms')}")
except Exception as e:
    print("NO-LINE", e)
PY
)
    note "2. N=$n lanes=$lanes rc=$rc $line"
    case "$line" in OK*) ;; *) fail=1 ;; esac
    [ $rc -ne 0 ] && fail=1
  done
done

# ---- 4. the CLI on files: contig-sharded (RCCL reducer inside the native search) and whole bins per GPU
tmp=$(mktemp -d)
python3 - "$tmp" <<'PY'
import sys
from nanomotif_amd import synth
mg = synth.make_metagenome(synth.SynthSpec(n_contigs=64, total_bp=40_000_000, n_bins=8, mod_types=("a", "m"), seed=17))
mg.write_fasta(sys.argv[1] + "/assembly.fasta"); mg.write_bed(sys.argv[1] + "/pileup.bed"); mg.write_contig_bin(sys.argv[1] + "/contig_bin.tsv")
PY
( cd "$tmp" && PYTHONPATH="$OLDPWD" timeout 1800 python3 -m nanomotif_amd motif_discovery assembly.fasta pileup.bed -c contig_bin.tsv --out out_n1 ) > "$out/cli_n1.log" 2>&1 || { note "4. CLI N=1 FAILED"; fail=1; }
for n in 2 4; do
  [ "$n" -gt "$n_gpus" ] && continue
  for shard in contigs bins; do
    port=$((20000 + RANDOM % 20000))
    ( cd "$tmp" && PYTHONPATH="$OLDPWD" timeout 1800 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=$n --master-addr 127.0.0.1 --master-port $port \
        -m nanomotif_amd motif_discovery assembly.fasta pileup.bed -c contig_bin.tsv --out out_${n}_$shard --shard $shard ) > "$out/cli_n${n}_$shard.log" 2>&1
    rc=$?
    if [ $rc -eq 0 ] && cmp -s "$tmp/out_n1/bin-motifs.tsv" "$tmp/out_${n}_$shard/bin-motifs.tsv"; then
      how=$(grep -o "count tables all-reduced by nm_allreduce_counts (RCCL, [0-9]* ranks)" "$out/cli_n${n}_$shard.log" | head -1)
      note "4. CLI N=$n --shard $shard: bin-motifs.tsv byte-equal to N=1 ${how:+[$how]}"
    else
      note "4. CLI N=$n --shard $shard: rc=$rc, output DIFFERS from N=1 (logs: $out/cli_n${n}_$shard.log)"; fail=1
    fi
  done
done
cp "$tmp/out_n1/bin-motifs.tsv" "$out/bin-motifs_n1.tsv" 2>/dev/null
rm -rf "$tmp"
note "rehearsal: $([ $fail -eq 0 ] && echo ALL LEGS AGREE WITH N=1 || echo FAILED)"
exit $fail

#!/bin/bash
# counters of the inflate kernels on the stand-alone harness: one rocprofv3 --pmc pass per counter group, then a kernel trace
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-inflate2_pmc}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
cd /tmp
R=$OLDPWD
k=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
  k=$((k+1))
  rm -rf /tmp/pp
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pp -- $R/tools/inflate2_proto 49152 1024 1 > $R/gpurun_out/$tag/log_$k.txt 2>&1
  find /tmp/pp -name "*counter_collection.csv" -exec cat {} \; | grep -E "Counter_Name|bed_" > $R/gpurun_out/$tag/pmc_$k.csv
done
rm -rf /tmp/pp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- $R/tools/inflate2_proto 49152 1024 3 > $R/gpurun_out/$tag/log_trace.txt 2>&1
find /tmp/pp -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$tag/kernel_stats.csv \;
cd $R
python3 - <<'PY' $tag
import csv, glob, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/{tag}/pmc_*.csv")):
    for row in csv.DictReader(open(f)):
        try:
            acc[row["Kernel_Name"].split("(")[0][-40:]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        except Exception:
            pass
with open(f"gpurun_out/{tag}/summary.txt", "w") as out:
    for kname, cs in acc.items():
        print(kname, file=out)
        for c, v in cs.items():
            print(f"   {c:28s} mean per launch {sum(v)/len(v):.4g}  (launches {len(v)})", file=out)
print(open(f"gpurun_out/{tag}/summary.txt").read())
PY
cat gpurun_out/$tag/kernel_stats.csv

#!/bin/bash
# round 5: what bounds bed_inflate_kernel — PMC passes on the device parse of a 7.5 GB (text) bgzip pileup (tools/bed_probe.py 20000000 5)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5ab
export TMPDIR=/tmp
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_FLAT" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pq
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/pq -- python3 tools/bed_probe.py 20000000 5 > gpurun_out/r5ab/probe_$i.log 2>&1
  echo "pass $i rc=$?"
  f=$(find /tmp/pq -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then head -1 "$f" > gpurun_out/r5ab/pmc_$i.csv; grep "bed_inflate_kernel" "$f" >> gpurun_out/r5ab/pmc_$i.csv; fi
  grep "bgzip file" gpurun_out/r5ab/probe_$i.log | tail -1
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/r5ab/pmc_*.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f, {k: "%.4g" % (sum(v) / len(v)) for k, v in agg.items()}, "launches", max((len(v) for v in agg.values()), default=0))
PY

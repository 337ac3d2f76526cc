#!/bin/bash
# PMC passes over the LDS prototype (pure scoring and the whole kernel at 10 candidates per slot)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
out=gpurun_out/lds_proto
mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-int-to-pointer-cast -DHALO=16 -DBOTH=0 -DWAVES=5 -o /tmp/lds_proto tools/lds_proto.hip || exit 1
for f in 3 0; do
i=0
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$f_$i
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/pmc_${f}_$i -- /tmp/lds_proto 122072 10 0 $f 3 > $out/pmc_run_${f}_$i.log 2>&1
  find /tmp/pmc_${f}_$i -name "*counter_collection.csv" -exec cp {} $out/pmc_${f}_$i.csv \;
done
done
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/lds_proto/pmc_*_*.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'score_t_kernel' in r.get('Kernel_Name', ''):
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(f, {k: sum(v) / len(v) for k, v in acc.items()})
PY

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
tag=${1:-ingest_pmc}
mkdir -p gpurun_out/$tag
export TMPDIR=/tmp
for pass in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pp
  timeout 600 rocprofv3 --pmc $pass --output-format csv -d /tmp/pp -- python3 tools/ingest_probe.py > gpurun_out/$tag/log_$name.txt 2>&1
  find /tmp/pp -name "*counter_collection.csv" -exec sh -c 'head -1 "$1" > '$out'/dev/null; grep "ingest_" "$1"' _ {} \; > gpurun_out/$tag/pmc_$name.csv
done
python3 tools/ingest_pmc_summary.py gpurun_out/$tag

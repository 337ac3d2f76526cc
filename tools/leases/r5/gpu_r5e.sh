#!/bin/bash
# round 5: what the small copies of a lock-step round cost — the end-to-end run with the speculation, copies through SDMA (default) against
# blit kernels (HSA_ENABLE_SDMA=0, GPU_FORCE_BLIT_COPY_SIZE)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r5e
run() {
  tag=$1; shift
  env "$@" NM_SEARCH_TIMING=1 timeout 600 python bench.py --workload e2e --steps 1 --warmup 0 --cpu-bins 0 > gpurun_out/r5e/e2e_$tag.log 2>&1
  echo "== $tag rc=$?"; grep "nm_search" gpurun_out/r5e/e2e_$tag.log | tail -2
  tail -1 gpurun_out/r5e/e2e_$tag.log | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); t = d['timings_rank0']; print('wall', round(d['value'], 4), {k: round(t.get(k), 4) for k in ('gpu_busy_s', 'search_s', 'native_search_s', 'postprocess_s', 'upload_filter_s')})"
}
for rep in 1 2; do
run spec_sdma A=1
run spec_nosdma HSA_ENABLE_SDMA=0
run spec_blit GPU_FORCE_BLIT_COPY_SIZE=4096
run nospec_sdma NM_SEARCH_NO_SPEC=1
run nospec_nosdma NM_SEARCH_NO_SPEC=1 HSA_ENABLE_SDMA=0
run spec_oneflight NM_SEARCH_ONE_FLIGHT=1
run spec_oneflight_nosdma NM_SEARCH_ONE_FLIGHT=1 HSA_ENABLE_SDMA=0
done

#!/bin/bash
B="python bench.py --extras none --cpu-bins 0 --hbm-round-steps 0"
for rep in 1 2 3; do
  $B 2>/dev/null | python tools/bench_brief.py "host wait   K=20"
  NM_STREAM_WAIT=1 $B 2>/dev/null | python tools/bench_brief.py "stream wait K=20"
done

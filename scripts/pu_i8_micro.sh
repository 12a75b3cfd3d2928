#!/bin/bash
# the exact downdate alone (scripts/micro/pu_i8_bench and its -DPX_ABL builds): scripts/pu_i8_micro.sh > out.txt
cd ${GRAFT_REPO_ROOT:-.}
for v in 0 1; do timeout 300 scripts/micro/pu_i8_bench 1000 298,1014,2000 15 $v; done
for m in 298 1014; do
  for a in 1 3 7 11 15; do
    [ -x scripts/micro/pu_i8_bench_abl$a ] && { echo "== PX_ABL=$a m=$m"; timeout 120 scripts/micro/pu_i8_bench_abl$a 1000 $m 15 0 | tail -1; }
  done
done

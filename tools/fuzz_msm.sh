#!/bin/bash
# sweeps tools/fuzz_msm.py over window layouts (each layout is fixed per process)
cd "$(dirname "$0")/.."
rc=0
for cfg in "" "C=4" "C=7" "C=11" "C=15" "R=3,6" "R=5,6" "R=3,9" "R=5,12"; do
  unset MI355ZK_MSM_C MI355ZK_MSM_RADIX
  case "$cfg" in C=*) export MI355ZK_MSM_C=${cfg#C=};; R=*) export MI355ZK_MSM_RADIX=${cfg#R=};; esac
  unset MI355ZK_HOST_CHUNK_TEST
  timeout 300 python tools/fuzz_msm.py --cases ${CASES:-30} --seed ${SEED:-7} 2>&1 | tail -3 || rc=1
  # the same cases through the STREAMED call: chunks of 64 exponents accumulated into one carried bucket array
  export MI355ZK_HOST_CHUNK_TEST=64
  timeout 300 python tools/fuzz_msm.py --cases ${CASES:-30} --seed ${SEED:-7} 2>&1 | tail -3 | sed 's/^fuzz:/fuzz (chunks of 64):/' || rc=1
done
# table mode (one bucket set for all windows over a window table of the vector), several table window widths
unset MI355ZK_MSM_C MI355ZK_MSM_RADIX MI355ZK_HOST_CHUNK_TEST
for tc in "" 4 7 11 17 20; do
  unset MI355ZK_MSM_TABLE_C
  [ -n "$tc" ] && export MI355ZK_MSM_TABLE_C=$tc
  timeout 600 python tools/fuzz_msm.py --table --cases ${CASES:-30} --seed ${SEED:-7} 2>&1 | tail -3 || rc=1
done
# the single-process multi-GPU mode: the same cases cut into cells over 2 / 3 / 8 logical devices (plain and with streamed chunks inside the cells)
unset MI355ZK_MSM_TABLE_C
for k in 2 3 8; do
  unset MI355ZK_HOST_CHUNK_TEST
  timeout 600 python tools/fuzz_msm.py --devices $k --cases ${CASES:-30} --seed ${SEED:-7} 2>&1 | tail -3 || rc=1
  export MI355ZK_HOST_CHUNK_TEST=64
  timeout 600 python tools/fuzz_msm.py --devices $k --cases ${CASES:-30} --seed ${SEED:-7} 2>&1 | tail -3 | sed 's/^fuzz/fuzz (chunks of 64)/' || rc=1
done
unset MI355ZK_HOST_CHUNK_TEST
# the one-lane reduce tails (MI355ZK_MSM_QUAD=0) against the same cases: the quad additions are the default above
MI355ZK_MSM_QUAD=0 timeout 300 python tools/fuzz_msm.py --cases ${CASES:-30} --seed ${SEED:-7} 2>&1 | tail -3 | sed 's/^fuzz:/fuzz (one-lane tails):/' || rc=1
exit $rc

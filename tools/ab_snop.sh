#!/bin/bash
# same-box A/B: the product library against tools/bin/libmi355zk_nonop.so (tools/build_nop_stripped.sh ntt msm_g1): NTT 2^20 / 2^24, G1 multiexp 2^20 / 2^26,
# with the oracle check of each leg; three alternating rounds
set -e
cd "$(dirname "$0")/.."
NONOP=$PWD/tools/bin/libmi355zk_nonop.so
for round in 1 2 3; do
  for so in "" "$NONOP"; do
    tag=$([ -z "$so" ] && echo "product " || echo "stripped")
    for ln in 20 24; do
      echo "$tag ntt 2^$ln  $(MI355ZK_SO=$so python tools/bench_ntt.py --log-n $ln --iters 40 --check | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print({k:(v["ms"] if isinstance(v,dict) else v) for k,v in d.items()})')"
    done
    echo "$tag msm 2^26 $(MI355ZK_SO=$so python bench.py --steps 5 --warmup 2 --no-secondary --no-h2d-leg --cpu-sample-log-n 16 | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"]["msm_accumulate"], d["full_size_linearity_check"], d["cpu_baseline"]["gpu_matches_oracle_on_sample"], d["result_affine_x_limb0"])')"
  done
done

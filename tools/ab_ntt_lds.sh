# same-box A/B of NTT variants (round 4): usage on the GPU box: bash tools/ab_ntt_lds.sh
run() { python tools/bench_ntt.py --log-n $1 --iters 30 --warm 60 | python -c "import sys,json; d=json.load(sys.stdin); print(d['log_n'], {k:(v['ms'],v['ntt_pass_ms_avg'],v['passes']) for k,v in d.items() if isinstance(v,dict)})"; }
for rep in 1 2; do
  echo "default"; for ln in 20 23 24 25; do run $ln; done
  echo "LOGNP=12 TILE=4096"; for ln in 23 24; do MI355ZK_NTT_LOGNP=12 MI355ZK_NTT_TILE=4096 run $ln; done
  echo "LOGNP=12 (tile default)"; for ln in 24; do MI355ZK_NTT_LOGNP=12 run $ln; done
  echo "TILE=4096"; for ln in 20 24; do MI355ZK_NTT_TILE=4096 run $ln; done
done

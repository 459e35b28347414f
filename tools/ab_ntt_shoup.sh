# same-box A/B (round 4): the NTT with Montgomery products by table twiddles (rounds 1-3; built from the previous ntt.hip as
# phase2-bn254_amd/libmi355zk_mont.so -- see tools/ab_ntt_lds.sh for the recipe) against the products by a constant with its quotient
# (u_mul_shoup).
# usage on the GPU box: bash tools/ab_ntt_shoup.sh
run() { python tools/bench_ntt.py --log-n $1 --iters 30 --warm 60 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print(d['log_n'], {k:(v['ms'],v['ntt_pass_ms_avg'],v['passes']) for k,v in d.items() if isinstance(v,dict)})"; }
for rep in 1 2; do
  echo "montgomery"; for ln in 16 20 22 24 26; do MI355ZK_SO=$PWD/phase2-bn254_amd/libmi355zk_mont.so run $ln; done
  echo "shoup"; for ln in 16 20 22 24 26; do run $ln; done
  echo "shoup, no full twiddle table"; for ln in 18 20; do MI355ZK_NTT_NO_FULL_TW=1 run $ln; done
done

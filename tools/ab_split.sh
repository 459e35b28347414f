# same-box A/B of the quad-per-bucket launch for the long buckets of short calls (round 4): bash tools/ab_split.sh
run() { python bench.py --log-n $1 --steps 30 --warmup 10 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print('2^%d' % $1, d['ms_per_step'], 'ms  acc', k['msm_accumulate'], 'reduce', k['msm_reduce'], d['result_affine_x_limb0'])"; }
for ln in 10 12 13 14 15 16 17 18; do
  echo -n "split "; run $ln
  echo -n "plain "; MI355ZK_MSM_SPLIT=0 run $ln
done
g2() { python tools/bench_g2.py --log-n $1 --iters 20 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('G2 2^%d' % d['g2_log_n'], d['ms'], 'ms', d['kernel_ms'], d['matches_closed_form'])"; }
for ln in 12 16 18; do
  echo -n "split "; g2 $ln
  echo -n "plain "; MI355ZK_MSM_SPLIT=0 g2 $ln
done

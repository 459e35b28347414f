# same-box A/B of the quad additions in the reduce tails (round 4): bash tools/ab_quad.sh
run() { python bench.py --log-n $1 --steps 30 --warmup 10 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print('2^%d' % $1, d['ms_per_step'], 'ms  reduce', k['msm_reduce'], 'acc', k['msm_accumulate'], d['result_affine_x_limb0'])"; }
for ln in 12 14 16 18 20 22; do
  echo -n "quad  "; run $ln
  echo -n "lane  "; MI355ZK_MSM_QUAD=0 run $ln
done
echo -n "quad_max 262144 2^20 "; MI355ZK_MSM_QUAD_MAX=262144 run 20
echo -n "quad_max 16384 2^20 "; MI355ZK_MSM_QUAD_MAX=16384 run 20
g2() { python tools/bench_g2.py --log-n $1 --iters 20 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('G2 2^%d' % d['g2_log_n'], d['ms'], 'ms', d['kernel_ms'], d['matches_closed_form'])"; }
for ln in 12 16 18 20; do
  echo -n "quad  "; g2 $ln
  echo -n "lane  "; MI355ZK_MSM_QUAD=0 g2 $ln
done

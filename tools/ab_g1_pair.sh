# same-box A/B of the G1 accumulation by a pair of lanes per bucket (round 4): bash tools/ab_g1_pair.sh
# auto = the library's gate, pair / lane = MI355ZK_G1_PAIR=1 / 0
run() { python bench.py --log-n $1 --steps 30 --warmup 10 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print('2^%d' % $1, d['ms_per_step'], 'ms  acc', k['msm_accumulate'], 'reduce', k['msm_reduce'], d['result_affine_x_limb0'])"; }
for ln in 10 12 13 14 15 16 17 18 20; do
  echo -n "auto  "; run $ln
  echo -n "pair  "; MI355ZK_G1_PAIR=1 run $ln
  echo -n "lane  "; MI355ZK_G1_PAIR=0 run $ln
done

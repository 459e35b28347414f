# same-box A/B of the one-lane G2 accumulation at two waves per SIMD (default from 2^19 points on) against one wave (MI355ZK_G2_WAVES=1): bash tools/ab_g2_waves.sh
g2() { python tools/bench_g2.py --log-n $1 --iters 20 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('G2 2^%d' % d['g2_log_n'], d['ms'], 'ms', d['kernel_ms'], d['matches_closed_form'])"; }
for ln in 19 20 21 22 24; do
  echo -n "two waves  "; g2 $ln
  echo -n "one wave   "; MI355ZK_G2_WAVES=1 g2 $ln
done

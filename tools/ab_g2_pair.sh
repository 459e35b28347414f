# same-box A/B of the G2 accumulation by a pair of lanes per bucket (round 4): bash tools/ab_g2_pair.sh
# auto = the library's gate (pairs while the call has <= 3 * 2^17 buckets), pair / lane = MI355ZK_G2_PAIR=1 / 0
g2() { python tools/bench_g2.py --log-n $1 --iters 20 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('G2 2^%d' % d['g2_log_n'], d['ms'], 'ms', d['kernel_ms'], d['matches_closed_form'])"; }
for ln in 10 12 14 16 18 19 20 22; do
  echo -n "auto  "; g2 $ln
  echo -n "pair  "; MI355ZK_G2_PAIR=1 g2 $ln
  echo -n "lane  "; MI355ZK_G2_PAIR=0 g2 $ln
done

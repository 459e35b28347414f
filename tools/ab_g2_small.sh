# G2 short calls after the U-form doubling in the record additions: bash tools/ab_g2_small.sh
g2() { python tools/bench_g2.py --log-n $1 --iters 30 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print('G2 2^%d' % d['g2_log_n'], d['ms'], 'ms', d['kernel_ms'], d['matches_closed_form'])"; }
for ln in 8 10 11 12 13 14 15 16 18 20; do g2 $ln; done

# the prover's eight concurrent multiexps (one of them the 2^20-point G2 call) with the one-lane G2 accumulation at two waves per SIMD
# (MI355ZK_G2_WAVES=2), at one wave (=1) and as the library decides (two waves only for a call that is alone on its device), same box
for rep in 1 2 3; do
  echo -n "auto       "; python tools/bench_prover.py --log-m 20 --iters 15 2>/dev/null
  echo -n "two waves  "; MI355ZK_G2_WAVES=2 python tools/bench_prover.py --log-m 20 --iters 15 2>/dev/null
  echo -n "one wave   "; MI355ZK_G2_WAVES=1 python tools/bench_prover.py --log-m 20 --iters 15 2>/dev/null
done

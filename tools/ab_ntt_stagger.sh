# same-box A/B: the second workgroup of every CU held back by k x s_sleep 127 (~3.4 us each) in the 2^20 NTT passes
cd "$(dirname "$0")/.."
for rep in 1 2; do
for sg in 0 1 2 3; do
  echo -n "stagger=$sg  "
  MI355ZK_NTT_STAGGER=$sg python tools/bench_ntt.py --log-n 20 2>/dev/null | tail -1 | cut -c1-400
done
done
for sg in 0 1 2; do echo -n "2^18 stagger=$sg  "; MI355ZK_NTT_STAGGER=$sg python tools/bench_ntt.py --log-n 18 2>/dev/null | tail -1 | cut -c1-300; done
for sg in 0 1 2; do echo -n "2^19 stagger=$sg  "; MI355ZK_NTT_STAGGER=$sg python tools/bench_ntt.py --log-n 19 2>/dev/null | tail -1 | cut -c1-300; done

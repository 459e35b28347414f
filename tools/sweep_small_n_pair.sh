# the short-call window table (choose_geom) re-measured with the pair-per-bucket accumulation (round 4, end):
# bash tools/sweep_small_n_pair.sh > gpurun_out/small_n_sweep_pair.txt
run() { python bench.py --log-n $1 --steps 30 --warmup 10 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print(d['ms_per_step'], 'ms  W', d['config']['windows'], 'bits', d['config']['window_bits'], 'reduce', k['msm_reduce'], 'acc', k['msm_accumulate'])"; }
g2() { python tools/bench_g2.py --log-n $1 --iters 30 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['kernel_ms']; print(d['ms'], 'ms  reduce', k['msm_reduce'], 'acc', k['msm_accumulate'], d['matches_closed_form'])"; }
echo "# G1, uniform exponents: ms per call by forced window width c (MI355ZK_MSM_C); 'default' = the table in choose_geom"
for ln in 12 13 14 15 16 17 18; do
  echo -n "2^$ln default: "; run $ln
  for c in 9 10 11 12 13 14 15 16; do
    if [ $c -ge $((ln/2+3)) ] && [ $c -le $((ln/2+8)) ]; then echo -n "2^$ln c=$c: "; MI355ZK_MSM_C=$c run $ln; fi
  done
done
echo "# G2"
for ln in 12 14 15 16 17 18; do
  echo -n "G2 2^$ln default: "; g2 $ln
  for c in 9 10 11 12 13 14 15 16; do
    if [ $c -ge $((ln/2+3)) ] && [ $c -le $((ln/2+8)) ]; then echo -n "G2 2^$ln c=$c: "; MI355ZK_MSM_C=$c g2 $ln; fi
  done
done

# re-measure the short-call window table (choose_geom) and the reduce schedule after the quad additions (round 4)
# usage on the GPU box: bash tools/sweep_small_n.sh > gpurun_out/r04/small_n_sweep.txt
run() { python bench.py --log-n $1 --steps 30 --warmup 10 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print(d['ms_per_step'], 'ms  W', d['config']['windows'], 'bits', d['config']['window_bits'], 'reduce', k['msm_reduce'], 'acc', k['msm_accumulate'])"; }
echo "# G1, uniform exponents: ms per call by forced window width c (MI355ZK_MSM_C); 'default' = the table in choose_geom"
for ln in 10 12 13 14 15 16 17 18 19 20; do
  echo -n "2^$ln default: "; run $ln
  for c in 10 11 12 13 14 15 16 17 18; do
    if [ $c -ge $((ln/2+4)) ] && [ $c -le $((ln/2+9)) ]; then echo -n "2^$ln c=$c: "; MI355ZK_MSM_C=$c run $ln; fi
  done
done
echo "# reduce schedule at the default width: MI355ZK_MSM_FINAL_MAX (elements per window handed to the trees) and MI355ZK_MSM_QUAD_MAX"
for ln in 14 16 18 20 22; do
  for fm in 256 512 2048 4096; do echo -n "2^$ln final_max=$fm: "; MI355ZK_MSM_FINAL_MAX=$fm run $ln; done
  for qm in 8192 32768 131072; do echo -n "2^$ln quad_max=$qm: "; MI355ZK_MSM_QUAD_MAX=$qm run $ln; done
done

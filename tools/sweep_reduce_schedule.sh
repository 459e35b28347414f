run() { python bench.py --log-n $1 --steps 30 --warmup 10 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print(d['ms_per_step'], 'ms reduce', k['msm_reduce'])"; }
for ln in 12 14 16 18 20 22; do
  echo -n "2^$ln default: "; run $ln
  for cfg in "3,2:512" "3,3:512" "3,2,2:512" "2,3:512" "3,2:1024" "3,3,2:512" "4,2:512" "2,2,2:512" "3:1024" "3:2048" "2:1024"; do
    l=${cfg%%:*}; f=${cfg##*:}
    echo -n "2^$ln logl=$l final_max=$f: "; MI355ZK_MSM_LOGL=$l MI355ZK_MSM_FINAL_MAX=$f run $ln
  done
done

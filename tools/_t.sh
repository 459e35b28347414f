python -m pytest tests/test_gpu_msm.py tests/test_gpu_msm_table.py tests/test_gpu_prover.py -x -q -m gpu 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernel_ms"]; print(d["ms_per_step"], "acc", k["msm_accumulate"], "red", k["msm_reduce"])'
for ln in 12 16 20 26; do
  echo -n "G1 2^$ln: "; python bench.py --log-n $ln --steps 20 --warmup 3 --no-cpu-baseline --no-h2d-leg --no-secondary 2>/dev/null | python -c "$P"
done
for ln in 16 20; do echo -n "G2 2^$ln: "; python tools/bench_g2.py --log-n $ln --iters 10 2>/dev/null | cut -c1-200; done
python tools/bench_table.py --log-n 20 | cut -c1-600
python tools/bench_table.py --group 2 --log-n 20 --iters 8 | cut -c1-600

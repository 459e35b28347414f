# same-box A/B: msm_accumulate_kernel at three waves per SIMD (amdgpu_waves_per_eu(3, 3); library variant) against the shipped four
cd "$(dirname "$0")/.."
run() { python bench.py --no-secondary --no-cpu-baseline --no-h2d-leg --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('%-30s ms_per_step %.3f  accumulate %.3f  ok=%s' % (sys.argv[1], d['ms_per_step'], k['msm_accumulate'], d['full_size_linearity_check']))" "$1"; }
for rep in 1 2; do
run "four waves per SIMD"
MI355ZK_SO=$PWD/tools/bin/libmi355zk_w3.so run "three waves per SIMD"
done

# same-box A/B of the bucket pass: 16-slot instantiation (two workgroups per CU) against round 5's 32-slot one; and the scatter's super-tile
cd "$(dirname "$0")/.."
run() { python bench.py --no-secondary --no-cpu-baseline --no-h2d-leg --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('%-28s ms_per_step %.3f  digits %.3f sort %.3f (scan %.3f scatter %.3f bucket %.3f) accumulate %.3f reduce %.3f' % (sys.argv[1], d['ms_per_step'], k['msm_digits'], k['msm_sort'], k['msm_part_scan'], k['msm_scatter'], k['msm_bucket'], k['msm_accumulate'], k['msm_reduce']))" "$1"; }
for rep in 1 2; do
MI355ZK_PART_EC=32 run "EC=32 (round 5)"
run "EC=16 (default now)"
MI355ZK_PART_ST=8192 run "EC=16, super-tile 8192"
done

# same-box A/B: streamed-once accesses of the partition as non-temporal loads / stores (library variants built with -DZK_EXP_NT[=2])
cd "$(dirname "$0")/.."
run() { python bench.py --no-secondary --no-cpu-baseline --no-h2d-leg --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernel_ms']
print('%-44s ms_per_step %.3f  digits %.3f sort %.3f (scatter %.3f bucket %.3f) accumulate %.3f  ok=%s' % (sys.argv[1], d['ms_per_step'], k['msm_digits'], k['msm_sort'], k['msm_scatter'], k['msm_bucket'], k['msm_accumulate'], d['full_size_linearity_check']))" "$1"; }
for rep in 1 2; do
run "default"
MI355ZK_SO=$PWD/tools/bin/libmi355zk_nt.so run "nt: pair loads of the bucket pass"
MI355ZK_SO=$PWD/tools/bin/libmi355zk_nt2.so run "nt: + scalars, keys (st + ld), vals stores"
done

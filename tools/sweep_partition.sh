run() { python bench.py --log-n 26 --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-h2d-leg 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); k=d['roofline']['kernel_ms']; print(d['ms_per_step'], {x:round(k[x],3) for x in ('msm_digits','msm_part_scan','msm_scatter','msm_bucket','msm_sort','msm_accumulate')})"; }
echo -n "default: "; run
for st in 8192 12288 4096; do echo -n "PART_ST=$st: "; MI355ZK_PART_ST=$st run; done
for lo in 10 11 12; do echo -n "PART_LO=$lo: "; MI355ZK_PART_LO=$lo run; done

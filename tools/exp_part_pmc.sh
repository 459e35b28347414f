# PMC of the partition kernels at 2^26 (msm_bucket, msm_scatter, msm_digits_plain, msm_tile_hist, msm_reduce_level)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/p_p1 /tmp/p_p2
CMD="python $R/bench.py --no-secondary --no-cpu-baseline --no-h2d-leg --steps 2 --warmup 1"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU -d /tmp/p_p1 -- $CMD > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d /tmp/p_p2 -- $CMD > /dev/null 2>&1
for d in /tmp/p_p1 /tmp/p_p2; do
python - $d <<PY
import sqlite3, glob, sys, re
db = sqlite3.connect(glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]); cur = db.cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
for k, n, c, v, d in sorted(rows):
    m = re.search(r"(msm_\w+)", k)
    nm = m.group(1) if m else k[:30]
    if any(s in nm for s in ("msm_bucket_kernel", "msm_scatter_kernel", "msm_digits_plain", "msm_tile_hist", "msm_reduce_level_kernel")):
        print(f"{nm:28s} {n:24s} n={c:3d} avg={v:16.1f} dur_us={d/1e3:9.1f}")
PY
done

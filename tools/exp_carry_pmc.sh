# PMC of the accumulate kernel in a streamed call cut into 16 equal chunks (1 fresh + 15 carried launches per call) against the unchunked launch
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rm -rf /tmp/p_c1 /tmp/p_c2
MI355ZK_HOST_CHUNK_TEST=4194304 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_WAVES -d /tmp/p_c1 -- python $R/tools/trace_host_entry.py > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db = sqlite3.connect(glob.glob("/tmp/p_c1/**/*.db", recursive=True)[0]); cur = db.cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like '%msm_accumulate_kernel%' group by kernel_name, counter_name").fetchall()
for k, n, c, v, d in sorted(rows):
    print(k[-60:], f"{n:24s} dispatches={c:3d} avg={v:16.1f} avg_dur_us={d/1e3:9.1f}")
PY

#!/usr/bin/env python3
"""tools/pmc_kernel.py <results.db> <kernel-substring>: per-dispatch PMC averages of one kernel (rocpd db)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select counter_name, count(*), avg(value), avg(duration) from counters_collection where kernel_name like ? group by counter_name", ("%" + sys.argv[2] + "%",)).fetchall()
for n, c, v, d in sorted(rows):
    print(f"{n:32s} dispatches={c:3d} avg={v:20.1f} avg_dur_us={d / 1e3:10.1f}")

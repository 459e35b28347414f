#!/usr/bin/env python3
"""tools/ntt_pass_split.py <results.db>: the NTT pass launches of a `rocprofv3 --kernel-trace -- python tools/bench_ntt.py --ops <one op>` run in launch
order, split by their position inside the transform (pass 1, pass 2, ...): median / min / max duration per position over the last 40 transforms."""
import sqlite3, sys, statistics as st
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = cur.execute("select start, end from kernels where name like '%ntt_pass_%' order by start").fetchall()
rows = rows[-40 * passes:]
for p in range(passes):
    d = [(e - s) / 1e3 for s, e in rows[p::passes]]
    gaps = [(rows[i][0] - rows[i - 1][1]) / 1e3 for i in range(p, len(rows), passes) if i > 0]
    print(f"pass {p + 1}: n={len(d)} median {st.median(d):7.1f} us  min {min(d):7.1f}  max {max(d):7.1f}   gap before it: median {st.median(gaps):5.1f} us")

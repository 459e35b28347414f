#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output) into a compact per-kernel table.
  tools/rocpd_summary.py <results.db> [--pmc]     (same numbers as `rocprofv3 --stats` / per-dispatch PMC)"""
import re, sqlite3, sys

def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.search(r"(zk::\w+)(<[^(]*>)?\(", name)
    if m:
        tmpl = "<Fq2>" if "Fq2" in (m.group(2) or "") else ("<Fq>" if "FqParams" in (m.group(2) or "") else ("<Fr>" if "FrParams" in (m.group(2) or "") else ""))
        return m.group(1) + tmpl
    m = re.search(r"rocprim::[^:]*::detail::(radix_sort_\w+|\w+)", name.split("trampoline_kernel")[-1])
    if "rocprim" in name:
        k = re.findall(r"radix_sort_onesweep_\w+", name)
        return "rocprim::" + (k[1] if len(k) > 1 else (k[0] if k else "kernel"))
    m = re.search(r"at::native::[^<(]*?(\w+)[<(]", name)
    return ("at::native::" + m.group(1)) if m else name[:60]

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
if "--timeline" in sys.argv:  # last N dispatches in launch order: start offset, duration, gap to the previous kernel
    n = int(sys.argv[sys.argv.index("--timeline") + 1])
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
    t0, prev = rows[0][1], rows[0][1]
    for name, st, en in rows:
        print(f"{(st - t0) / 1e3:10.1f} us  dur {(en - st) / 1e3:9.1f}  gap {(st - prev) / 1e3:8.1f}  {short(name)}")
        prev = en
elif "--pmc" in sys.argv:
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    agg = {}
    for n, c, cnt, av, sm, dur in rows:
        k = (short(n), c); a = agg.setdefault(k, [0, 0.0, 0.0]); a[0] += cnt; a[1] += sm; a[2] += dur * cnt
    print(f"{'kernel':48s} {'counter':14s} {'dispatches':>10s} {'avg/dispatch':>16s} {'avg_dur_us':>12s}")
    for (k, c), (cnt, sm, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:48s} {c:14s} {cnt:10d} {sm / cnt:16.1f} {dur / cnt / 1e3:12.1f}")
else:
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name").fetchall()
    agg = {}
    for n, cnt, tot, av, mn, mx in rows:
        a = agg.setdefault(short(n), [0, 0, 1e30, 0]); a[0] += cnt; a[1] += tot; a[2] = min(a[2], mn); a[3] = max(a[3], mx)
    total = sum(a[1] for a in agg.values())
    print(f"{'kernel':48s} {'calls':>7s} {'total_ms':>12s} {'avg_us':>12s} {'min_us':>12s} {'max_us':>12s} {'pct':>6s}")
    for k, (cnt, tot, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:48s} {cnt:7d} {tot / 1e6:12.3f} {tot / cnt / 1e3:12.1f} {mn / 1e3:12.1f} {mx / 1e3:12.1f} {100 * tot / total:6.2f}")

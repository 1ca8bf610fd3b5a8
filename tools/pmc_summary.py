#!/usr/bin/env python3
"""Per-kernel averages of every counter in a rocprofv3 --pmc rocpd database:  python tools/pmc_summary.py <dir> [filter]"""
import collections, os, sqlite3, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "gemm"
db = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")][0]
con = sqlite3.connect(db)
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for name, cn, val, dur in con.execute("select kernel_name, counter_name, value, duration from counters_collection"):
    if flt not in name:
        continue
    a = agg[name][cn]
    a[0] += 1; a[1] += val; a[2] += dur
for name, cs in agg.items():
    short = name.replace("void sdmi::", "")[:70]
    n = max(v[0] for v in cs.values())
    dur = max(v[2] / v[0] for v in cs.values())
    print(f"{short}  launches={n} avg_us={dur / 1e3:.1f}")
    for cn, (c, v, _) in sorted(cs.items()):
        print(f"    {cn:32s} {v / c:16.0f}")

#!/usr/bin/env python3
"""List the kernel launches longer than a threshold (us) in a rocprofv3 rocpd database, with what ran around them."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); thr = float(sys.argv[2]) * 1e3
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else cols[0]
rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
t0 = rows[0][1]
for i, (n, s, e, q) in enumerate(rows):
    if e - s > thr:
        print(f"== launch {i} of {len(rows)}: {n[:50]} q{q} start {(s - t0) / 1e9:.3f} s dur {(e - s) / 1e3:.0f} us")
        for n2, s2, e2, q2 in rows[max(0, i - 12):i + 14]:
            print(f"   {(s2 - s) / 1e3:12.1f} {(e2 - s2) / 1e3:12.1f} q{q2} {n2[:70]}")

#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (--pmc X --kernel-trace)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
def T(p): return [t for t in tabs if t.startswith(p)][0]
pe, kd, ks, ip = T("rocpd_pmc_event"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_info_pmc")
cols = [r[1] for r in db.execute(f"pragma table_info({pe})")]
q = f"select s.kernel_name, p.name, count(*), sum(e.value) from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {ip} p on e.pmc_id = p.id group by s.kernel_name, p.name order by 4 desc"
for n, c, cnt, tot in list(db.execute(q))[:14]:
    print(f"{n[:70]:70s} {c:12s} n={cnt:6d} total={tot:.4g} avg/launch={tot/cnt:.4g}")

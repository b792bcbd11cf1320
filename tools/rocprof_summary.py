#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table
(name, calls, total ms, avg us, min, max, % of GPU kernel time) — the same numbers as `--stats`."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols_kd = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    cols_ks = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in cols_ks else ("display_name" if "display_name" in cols_ks else cols_ks[-1])
    q = f"select s.{name_col}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) " \
        f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, t, mn, mx in rows:
        n = n.split("(")[0][-90:]
        lines.append(f"| `{n}` | {c} | {t/1e6:.3f} | {t/c/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*t/tot:.1f} |")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])

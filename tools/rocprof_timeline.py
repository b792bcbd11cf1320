#!/usr/bin/env python3
"""Print a coarse timeline of the LAST training step from a rocprofv3 rocpd database:
per kernel family: first start, last end, summed busy time, and the union (wall) coverage."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else cols[0]
gcol = "d.grid_size_x / d.workgroup_size_x" if "grid_size_x" in cols and "workgroup_size_x" in cols else "0"
gz = "d.grid_size_z" if "grid_size_z" in cols else "1"
rows = list(db.execute(f"select s.kernel_name, d.start, d.end, d.{qcol}, {gcol}, {gz} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# last step = from the last augment_smooth kernel on
idx = [i for i, r in enumerate(rows) if "augment_smooth" in r[0]]
nsteps_back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = rows[idx[-nsteps_back]:(idx[-nsteps_back + 1] if nsteps_back > 1 else len(rows))]
t0 = rows[0][1]
def fam(n):
    for k in ("gru_wave_fwd", "gru_wave_bwd", "wave_gate", "gru_persist_fwd", "gru_persist_bwd", "gemm_bf16p", "pack_mc", "pack_kc", "gemm_bf16_kernel", "dropout", "patch_fold", "gemm_f32_kernelILb0ELb0", "gemm_f32_kernelILb1ELb0", "gemm_f32_kernelILb1ELb1", "gemm_f32_kernelILb0ELb1", "ctc_kernel", "colsum", "adamw", "sumsq"):
        if k in n: return k
    return "other"
agg = {}
for n, s, e, q, gx, gzz in rows:
    a = agg.setdefault(fam(n), [1e30, 0, 0.0, 0, []])
    a[0] = min(a[0], s - t0); a[1] = max(a[1], e - t0); a[2] += e - s; a[3] += 1; a[4].append((s, e))
print(f"step wall: {(max(r[2] for r in rows) - t0)/1e6:.3f} ms, kernels: {len(rows)}, queues: {len(set(r[3] for r in rows))}")
for k, (s, e, busy, cnt, iv) in sorted(agg.items(), key=lambda kv: kv[1][0]):
    iv.sort(); cov = 0; cs, ce = iv[0]
    for a, b in iv[1:]:
        if a > ce: cov += ce - cs; cs, ce = a, b
        else: ce = max(ce, b)
    cov += ce - cs
    print(f"{k:28s} n={cnt:4d} first={s/1e6:8.3f} last={e/1e6:8.3f} busy={busy/1e6:8.3f} union={cov/1e6:8.3f} ms")
if len(sys.argv) > 3:
    for n, s, e, q, gx, gzz in rows:
        print(f"{(s-t0)/1e3:10.1f} {(e-s)/1e3:9.1f} q{q} wg={gx}x{gzz} {n[:60]}")

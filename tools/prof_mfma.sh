# MFMA utilisation counters of the headline step, per kernel (own pass; --kernel-trace only, as gpurun requires):
# SQ_VALU_MFMA_BUSY_CYCLES (cycles an MFMA pipe is busy, summed over the SIMDs), SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $OUT/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/pmc_mfma -o m -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob, os
p = glob.glob(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/pmc_mfma/**/m_results.db"), recursive=True)
db = sqlite3.connect(p[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda pre: [t for t in tabs if t.startswith(pre)][0]
pe, kd, ks, ip = T("rocpd_pmc_event"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_info_pmc")
q = f"select s.kernel_name, p.name, count(*), sum(e.value), count(distinct d.event_id) from {pe} e join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id join {ip} p on e.pmc_id = p.id group by s.kernel_name, p.name"
rows = {}
for n, c, cnt, tot, nl in db.execute(q):
    rows.setdefault(n, {})[c] = (cnt, tot, nl)
# a counter comes as one row per hardware instance and launch (SQ: 32 = 8 XCDs x 4 shader engines, GRBM: 8): MFMA busy cycles are
# SUMMED over a launch's rows, the launch's duration in cycles is the MEAN of its GRBM_GUI_ACTIVE rows
out = ["| kernel | launches | MFMA-busy SIMD-cycles per launch | cycles per launch (GRBM_GUI_ACTIVE) | MFMA pipes busy, chip-wide (busy / (cycles x 1024 SIMDs)) |", "|---|---|---|---|---|"]
for n, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0, 1))[1])[:8]:
    m = d.get("SQ_VALU_MFMA_BUSY_CYCLES", (1, 0, 1)); g = d.get("GRBM_GUI_ACTIVE", (1, 0, 1))
    per_launch, cyc = m[1] / max(1, m[2]), g[1] / max(1, g[0])
    out.append(f"| `{n[:80]}` | {m[2]} | {per_launch:.4g} | {cyc:.4g} | {per_launch / (cyc * 1024) if cyc else float('nan'):.3f} |")
open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out/r4a_pmc_mfma.md"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -2 $OUT/pmc_mfma.log | cut -c1-200
rm -rf $OUT/pmc_mfma

"""Per-kernel totals from a rocprofv3 rocpd database (the default output of rocprofv3 --kernel-trace on this image):
    python tools/rocpd_stats.py gpurun_out/<dir>/<name>_results.db [csv_out]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
ks = dict(cur.execute(f"select id, kernel_name from {sym}"))
agg = collections.defaultdict(lambda: [0, 0, 1 << 62, 0])
for kid, s, e in cur.execute(f"select kernel_id, start, end from {disp}"):
    a = agg[ks[kid]]; d = e - s; a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(v[1] for v in agg.values())
rows = sorted(agg.items(), key=lambda x: -x[1][1])
lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
for k, v in rows:
    lines.append(f"\"{k}\",{v[0]},{v[1]},{v[1] / v[0]:.1f},{100 * v[1] / tot:.2f},{v[2]},{v[3]}")
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write("\n".join(lines) + "\n")
for k, v in rows[:18]:
    print(f"{v[1] / 1e6:10.1f} ms {v[0]:6d} calls avg {v[1] / v[0] / 1e6:9.3f} ms {100 * v[1] / tot:5.1f}%  {k[:90]}")

"""Timeline of the wp_* / dp_* kernels of a rocprofv3 rocpd database, per stream: python tools/rocpd_timeline.py DB [min_ms]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
ks = dict(cur.execute(f"select id, kernel_name from {sym}"))
rows = [(s, e, ks[k], q, gx) for k, s, e, q, gx in cur.execute(f"select kernel_id,start,end,stream_id,grid_size_x from {disp} order by start")]
w = [r for r in rows if "lrsc" in r[2]]
t0 = w[0][0]
for s, e, n, q, gx in w:
    d = (e - s) / 1e6
    if d >= min_ms:
        short = n.split("lrsc")[1][:40]
        print(f"t={(s - t0) / 1e6:9.1f} ms  dur={d:9.1f} ms  stream {q}  grid {gx:8d}  {short}")

"""Per-kernel totals from a rocprofv3 rocpd .db (sqlite):  python tools/rocpd_stats.py results.db [top]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [x for x in tabs if x.startswith('rocpd_kernel_dispatch')][0]
ks = [x for x in tabs if x.startswith('rocpd_info_kernel_symbol')][0]
cols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name = 'display_name' if 'display_name' in cols else 'kernel_name'
rows = db.execute(f"select s.{name}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.{name} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} launches")
for n, c, t, mn, mx in rows[:top]:
    print(f"{t/1e6:9.3f} ms {100*t/tot:5.1f}%  n={c:5d} avg={t/c/1e3:8.1f} us min={mn/1e3:7.1f} max={mx/1e3:7.1f}  {n[:110]}")

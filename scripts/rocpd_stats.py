"""Summarise a rocprofv3 rocpd sqlite database: per-kernel totals and GPU busy/idle time inside the last
`--steps` benchmark steps.  usage: rocpd_stats.py results.db [n_last_dispatch_fraction]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
def short(n):
    n = re.sub(r'\(.*', '', n); n = re.sub(r'^void ', '', n)
    return n[:90]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
agg = {}
for n, s, e in rows:
    a = agg.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
span = rows[-1][2] - rows[0][1]
busy, cur_end = 0, rows[0][1]
for n, s, e in rows:
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  sum-of-kernels {tot/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
print(f"{'kernel':92s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:92s} {v[0]:7d} {v[1]/1e6:9.3f} {v[1]/v[0]/1e3:8.2f} {100*v[1]/tot:6.2f}")

# ---- idle gaps (no kernel of any stream running): attribute each gap to the kernel that ENDS it
gaps = {}
hist = [0, 0, 0, 0, 0]   # <2us, 2-5, 5-10, 10-50, >50
cur_end = rows[0][2]
for n, s, e in rows[1:]:
    if s > cur_end:
        gp = s - cur_end
        a = gaps.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += gp
        hist[0 if gp < 2000 else 1 if gp < 5000 else 2 if gp < 10000 else 3 if gp < 50000 else 4] += gp
    cur_end = max(cur_end, e)
print("\nidle time by gap length (ms): <2us %.2f | 2-5us %.2f | 5-10us %.2f | 10-50us %.2f | >50us %.2f" % tuple(h / 1e6 for h in hist))
print(f"{'idle gaps by the kernel that follows':92s} {'gaps':>7s} {'total_ms':>9s} {'avg_us':>8s}")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:92s} {v[0]:7d} {v[1]/1e6:9.3f} {v[1]/v[0]/1e3:8.2f}")

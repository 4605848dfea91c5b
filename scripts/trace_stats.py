"""Summarise a rocprofv3 --kernel-trace CSV (`*_kernel_trace.csv`): per-kernel totals, GPU busy / idle time and the idle gaps by
the kernel that follows, inside the last fraction of the dispatches.  usage: trace_stats.py run_kernel_trace.csv [fraction=0.6]"""
import csv, re, sys
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
rows = rows[int(len(rows) * (1 - frac)):]
def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return n[:96]
agg = {}
for n, s, e in rows:
    a = agg.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
span = max(r[2] for r in rows) - rows[0][1]
busy, cur_end = 0, rows[0][1]
for n, s, e in rows:
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  sum-of-kernels {tot/1e6:.2f} ms  union-busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
print(f"{'kernel':98s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>8s} {'%':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:44]:
    print(f"{k:98s} {v[0]:7d} {v[1]/1e6:9.3f} {v[1]/v[0]/1e3:8.2f} {100*v[1]/tot:6.2f}")
gaps, hist, cur_end = {}, [0, 0, 0, 0, 0], rows[0][2]
for n, s, e in rows[1:]:
    if s > cur_end:
        gp = s - cur_end
        a = gaps.setdefault(short(n), [0, 0]); a[0] += 1; a[1] += gp
        hist[0 if gp < 2000 else 1 if gp < 5000 else 2 if gp < 10000 else 3 if gp < 50000 else 4] += gp
    cur_end = max(cur_end, e)
print("\nidle time by gap length (ms): <2us %.2f | 2-5us %.2f | 5-10us %.2f | 10-50us %.2f | >50us %.2f" % tuple(h / 1e6 for h in hist))
print(f"{'idle gaps by the kernel that follows':98s} {'gaps':>7s} {'total_ms':>9s} {'avg_us':>8s}")
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"{k:98s} {v[0]:7d} {v[1]/1e6:9.3f} {v[1]/v[0]/1e3:8.2f}")

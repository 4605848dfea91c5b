"""One training step of a rocprofv3 --kernel-trace CSV as a timeline: the LAST complete step (stem_pack .. next stem_pack), split at the
average-pool kernels into forward | head | backward, with every kernel's duration and the idle gap in front of it, then per-segment totals
by kernel.  usage: step_timeline.py run_kernel_trace.csv [--full]"""
import csv, re, sys
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: r[1])
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:72]
starts = [i for i, r in enumerate(rows) if "stem_pack" in r[0]]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0][1]
i_pf = next(i for i, r in enumerate(step) if "avgpool_fwd" in r[0])
i_pb = next(i for i, r in enumerate(step) if "avgpool_bwd" in r[0])
segs = [("forward", step[: i_pf + 1]), ("head", step[i_pf + 1: i_pb]), ("backward + optimizer", step[i_pb:])]
print(f"step span {(rows[b][1] - t0) / 1e3:.1f} us, {len(step)} dispatches")
for name, seg in segs:
    span = max(r[2] for r in seg) - seg[0][1]
    busy, cur = 0, seg[0][1]
    for n, s, e in seg:
        if e > cur: busy += e - max(s, cur); cur = e
    print(f"\n== {name}: {len(seg)} dispatches, span {span / 1e3:.1f} us, union-busy {busy / 1e3:.1f} us, sum of kernels {sum(e - s for _, s, e in seg) / 1e3:.1f} us")
    agg = {}
    for n, s, e in seg:
        v = agg.setdefault(short(n), [0, 0]); v[0] += 1; v[1] += e - s
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"   {k:74s} {v[0]:4d} x {v[1] / v[0] / 1e3:8.1f} us = {v[1] / 1e3:9.1f} us")
    if "--full" in sys.argv:
        cur = seg[0][1]
        for n, s, e in seg:
            print(f"      +{(s - t0) / 1e3:9.1f}  gap {max(0, s - cur) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {short(n)}")
            cur = max(cur, e)

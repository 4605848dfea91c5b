"""HBM bytes of ONE whole training step, every kernel, from the two PMC passes of scripts/pmc_traffic.py (FETCH_SIZE / WRITE_SIZE,
separate passes, `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline`): which kernel classes move the step's bytes.
Same units / gfx950 correction as pmc_traffic.py (KiB; FETCH_SIZE doubled).  usage: step_traffic.py pmc_f pmc_w out.txt [title]
With a title (another workload): grouped by kernel name instead of the ResNet-50 classes."""
import csv, glob, os, re, sys


def load(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    rows = [r for f in files for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def klass(n):
    for key, name in (("wgrad", "weight gradient (+ slab reduction)"), ("gram_reduce", "weight gradient (+ slab reduction)"), ("abn_", "algebraic BatchNorm backward (fold / fix-up)"), ("conv_gemm_kernel", "conv forward + dgrad"), ("conv3x3_c64", "conv forward + dgrad"), ("stem7x7", "conv forward + dgrad"), ("adam_step", "Adam"), ("gram_stats", "BatchNorm reductions / finalize"),
                      ("bn_bwd_apply", "BatchNorm backward apply"), ("bn_apply", "BatchNorm forward apply"), ("stem_", "stem (pack / pool / BatchNorm)"),
                      ("bn_", "BatchNorm reductions / finalize"), ("partial_reduce", "BatchNorm reductions / finalize"), ("stage_weights", "weight staging"),
                      ("multi_tensor_apply", "Adam"), ("avgpool", "pooling"), ("parity_zero_fill", "conv forward + dgrad")):
        if key in n:
            return name
    return "head / other"


def last_step(rows):
    # two identical steps run (warm-up + timed): the second half of the dispatches of every kernel name is the timed step
    by = {}
    for r in rows:
        by.setdefault(r["Kernel_Name"], []).append(r)
    out = []
    for name, v in by.items():
        out += v[len(v) // 2:] if len(v) % 2 == 0 else v[(len(v) + 1) // 2:]
    return out


TITLE = sys.argv[4] if len(sys.argv) > 4 else None
if TITLE:
    import re as _re
    klass = lambda n: _re.sub(r"\(anonymous namespace\)::", "", _re.sub(r"^void ", "", n))[:70]
f = last_step(load(sys.argv[1], "FETCH_SIZE"))
w = last_step(load(sys.argv[2], "WRITE_SIZE"))
agg = {}
for r in f:
    a = agg.setdefault(klass(r["Kernel_Name"]), [0.0, 0.0, 0]); a[0] += 2.0 * float(r["Counter_Value"]) * 1024.0; a[2] += 1
for r in w:
    a = agg.setdefault(klass(r["Kernel_Name"]), [0.0, 0.0, 0]); a[1] += float(r["Counter_Value"]) * 1024.0
tot_r = sum(a[0] for a in agg.values()); tot_w = sum(a[1] for a in agg.values())
lines = [f"HBM traffic of one {TITLE or 'ResNet-50 + crossattention'} training step ({'per-GPU batch of the workload' if TITLE else 'batch 256'}, bf16), rocprofv3 PMC, all kernels: read {tot_r / 1e9:.1f} GB + write {tot_w / 1e9:.1f} GB = {(tot_r + tot_w) / 1e9:.1f} GB",
         f"  = {(tot_r + tot_w) / 6.3e12 * 1e3:.1f} ms at the 6.3 TB/s a streaming pass achieves (8 TB/s nominal: {(tot_r + tot_w) / 8e12 * 1e3:.1f} ms)",
         f"{'class':72s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s} {'share':>7s}"]
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:28]:
    lines.append(f"{k:72s} {a[2]:8d} {a[0] / 1e9:9.2f} {a[1] / 1e9:9.2f} {100 * (a[0] + a[1]) / (tot_r + tot_w):6.1f}%")
open(sys.argv[3], "w").write("\n".join(lines) + "\n")
if len(sys.argv) > 5:   # machine-readable totals for bench.py (keyed by the kernel sources they were measured on)
    import json
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash
    json.dump({"source_hash": source_hash(), "step_read_bytes": tot_r, "step_write_bytes": tot_w, "step_bytes": tot_r + tot_w,
               "by_class": {k: {"launches": a[2], "read_bytes": a[0], "write_bytes": a[1]} for k, a in agg.items()},
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, every kernel of the timed step), FETCH_SIZE doubled (gfx950)"},
              open(sys.argv[5], "w"), indent=1)
print("\n".join(lines))

"""Per-layer table of the conv forward / dgrad / wgrad launches of ONE production training step (ResNet-50, batch 256, bf16) from a
rocprofv3 kernel trace: microseconds, TFLOP/s, the HBM floor (algorithmic bytes / 6.3 TB/s), the MFMA floor (FLOP / 2.5 PFLOP/s) and
the ratio to the larger floor.  Dispatch order is the plan's (csrc/backbone.hip); the side stream's launches are matched by kernel
name.  usage: layer_table.py run_kernel_trace.csv"""
import csv, sys
N = 256
def resnet50():
    blocks, cin, h = [], 64, 56
    for li, (w, nb) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), 1):
        for b in range(nb):
            s = 2 if (b == 0 and li > 1) else 1
            ho = h // s
            nm = f"l{li}.{b}"
            u = [(nm + ".c1", dict(Mo=N * h * h, Mi=N * h * h, Cin=cin, Cout=w, taps=1)),
                 (nm + ".c2", dict(Mo=N * ho * ho, Mi=N * h * h, Cin=w, Cout=w, taps=9)),
                 (nm + ".c3", dict(Mo=N * ho * ho, Mi=N * ho * ho, Cin=w, Cout=4 * w, taps=1))]
            ds = (nm + ".ds", dict(Mo=N * ho * ho, Mi=N * h * h, Cin=cin, Cout=4 * w, taps=1, s2=(s == 2))) if (s != 1 or cin != 4 * w) else None
            blocks.append((u, ds)); cin = 4 * w; h = ho
    return blocks
blocks = resnet50()
stem = ("stem", dict(Mo=N * 112 * 112, Mi=N * 224 * 224, Cin=3, Cout=64, taps=49))
fwd = [stem]
import os
FWD2P = os.environ.get("MMSKIN_FWD2P", "1") != "0" and os.environ.get("MMSKIN_ABN", "1") != "0"
FWDG = os.environ.get("MMSKIN_FWDG", "1") != "0"
for u, ds in blocks:
    if ds: fwd.append(ds)
    # two-pass forward (backbone.hip): conv3 of a layer1 / layer2 block without a downsample branch is one conv launch with BatchNorm +
    # residual + ReLU in its epilogue ("c3+bn"); its statistics come from the Gram matrix of its input (a ring-kernel launch, not listed
    # here).  MMSKIN_FWDG=0: a statistics-only pass of the convolution first (the "c3" row).
    if FWD2P and not ds and u[2][1]["Cin"] <= 128:
        fwd += u[:2] + ([u[2]] if not FWDG else []) + [(u[2][0] + "+bn", u[2][1])]
    else:
        fwd += u
bwd = []   # per block, last to first: conv3, conv2, (downsample), conv1 -- the plan's dgrad order
for u, ds in reversed(blocks):
    bwd += [u[2], u[1]] + ([ds] if ds else []) + [u[0]]
def flops(d): return 2.0 * d["Mo"] * d["Cout"] * d["Cin"] * d["taps"]
def bytes_fwd(d):   # read the input once (a strided 1x1 reads a quarter of the pixels), the weights, write the output
    mi = d["Mo"] if d.get("s2") else d["Mi"]
    return 2.0 * (mi * d["Cin"] + d["Mo"] * d["Cout"] + d["Cout"] * d["Cin"] * d["taps"])
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def is_conv(r): return any(k in r["Kernel_Name"] for k in ("conv_gemm_kernel", "conv3x3_c64_kernel", "stem7x7_kernel"))
conv_all = [r for r in rows if is_conv(r)]
# The forward downsample convolutions run on the side stream (another HSA queue) beside conv1..conv3 of their block, so start-time
# order interleaves them arbitrarily: split the dispatches by queue -- the main queue carries stem, conv1..3 of every block and every
# dgrad in plan order, the side queue the four forward downsample convolutions (and every weight-gradient GEMM).
qkey = "Queue_Id" if "Queue_Id" in rows[0] else None
if qkey:
    counts = {}
    for r in conv_all: counts[r[qkey]] = counts.get(r[qkey], 0) + 1
    main_q = max(counts, key=counts.get)
    main = [r for r in conv_all if r[qkey] == main_q]
    side = [r for r in conv_all if r[qkey] != main_q]
else:
    main, side = conv_all, []
fwd_main = [x for x in fwd if not x[0].endswith(".ds")]
fwd_side = [x for x in fwd if x[0].endswith(".ds")]
per_step_main = len(fwd_main) + len(bwd)
if side and len(main) >= per_step_main and len(side) >= len(fwd_side):
    order = fwd_main + bwd
    conv = main[-per_step_main:]
    side_last = side[-len(fwd_side):]
    fwd = fwd_main + fwd_side
    conv = conv[:len(fwd_main)] + side_last + conv[len(fwd_main):]
else:
    per_step = len(fwd) + len(bwd)
    conv = conv_all[-per_step:]
def dur(r): return (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print(f"{'':2s}{'layer':10s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>8s} {'TF/s':>7s} {'HBM fl.':>8s} {'MFMA fl.':>8s} {'ratio':>6s}  kernel")
tot = {"F": 0.0, "D": 0.0}
for i, ((name, d), r) in enumerate(zip(fwd + bwd, conv)):
    kind = "F" if i < len(fwd) else "D"
    us = dur(r); tot[kind] += us
    fl = flops(d); hb = bytes_fwd(d) / 6.3e12 * 1e6; mf = fl / 2.5e15 * 1e6
    M, Nn, K = (d["Mo"], d["Cout"], d["Cin"] * d["taps"]) if kind == "F" else (d["Mi"], d["Cin"], d["Cout"] * d["taps"])
    kn = "direct 7x7" if "stem7x7" in r["Kernel_Name"] else "layer-1 3x3" if "conv3x3_c64" in r["Kernel_Name"] else ("pipelined" if int(r["Workgroup_Size_X"]) == 512 else "128-row")
    print(f"{kind} {name:10s} {M:8d} {Nn:5d} {K:5d} {us:8.1f} {fl / us / 1e6:7.0f} {hb:8.1f} {mf:8.1f} {us / max(hb, mf):6.2f}  {kn} x{int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])}")
print(f"forward {tot['F'] / 1e3:.3f} ms, dgrad {tot['D'] / 1e3:.3f} ms per step (production: beside the side stream's weight-gradient GEMMs)")
wg = [r for r in rows if "wgrad" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and "unpack" not in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]]
nw = 53
wg = wg[-nw:]
if len(wg) == nw:
    order = [x for u, ds in reversed(blocks) for x in (list(reversed(u)) + ([ds] if ds else []))] + [stem]
    # the plan enqueues a block's weight gradients as conv3, conv2, conv1 with the downsample one after conv1's
    t = 0.0
    for (name, d), r in zip(order, wg):
        us = dur(r); t += us
        fl = flops(d); hb = 2.0 * (d["Mi"] * d["Cin"] + d["Mo"] * d["Cout"]) / 6.3e12 * 1e6; mf = fl / 2.5e15 * 1e6
        print(f"W {name:10s} {d['Cout']:8d} {d['Cin'] * d['taps']:5d} {d['Mo']:5d}".ljust(34) + f"{us:8.1f} {fl / us / 1e6:7.0f} {hb:8.1f} {mf:8.1f} {us / max(hb, mf):6.2f}  {r['Kernel_Name'][:40]}")
    print(f"wgrad {t / 1e3:.3f} ms per step (GEMM launches only, side stream)")

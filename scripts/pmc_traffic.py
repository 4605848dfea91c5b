"""Turn two rocprofv3 PMC passes of `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline` into
profiles/conv_gemm_traffic.json: HBM bytes per launch of the dominant kernel (conv_gemm_kernel, forward + dgrad).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o run -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
    python3 scripts/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/conv_gemm_traffic.json

Separate passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2: MI355X_MICROARCH.md, rocprofv3 PMC slots), counters only with
--kernel-trace.  Units and gfx950 correction as that guide's HBM section prescribes: the counters are in KiB; FETCH_SIZE tallies
128-byte requests at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  The LAST step's
dispatches are used (2 steps run: 1 warm-up + 1 timed); the JSON records the sha256 of the kernel sources it was measured on
(bench.py quotes `traffic` only when that matches the library it runs)."""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash  # noqa: E402


def dom(name):
    """the dominant class: the implicit-GEMM conv forward + dgrad launches (conv_gemm.hip: 128-row and pipelined kernels; conv3x3_c64.hip; stem7x7.hip)"""
    return "conv_gemm_kernel" in name or "conv3x3_c64_kernel" in name or "stem7x7_kernel" in name


def per_step_kib(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    rows = [r for f in files for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and dom(r["Kernel_Name"])]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) % 2 == 0 and rows, (len(rows), "expected the same number of conv_gemm launches in the warm-up and the timed step")
    last = rows[len(rows) // 2:]
    return sum(float(r["Counter_Value"]) for r in last), len(last)


fetch_kib, n1 = per_step_kib(sys.argv[1], "FETCH_SIZE")
write_kib, n2 = per_step_kib(sys.argv[2], "WRITE_SIZE")
assert n1 == n2, (n1, n2)
hbm = (2.0 * fetch_kib + write_kib) * 1024.0
out = {"kernel": f"conv_gemm_kernel + conv3x3_c64_kernel + stem7x7_kernel ({n1} launches per step: forward + dgrad)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python3 bench.py --steps 1 --warmup 1",
       "fetch_size_kib_per_step": fetch_kib, "write_size_kib_per_step": write_kib,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "hbm_bytes_per_step": hbm, "launches_per_step": n1, "hbm_bytes_per_launch": hbm / n1, "source_hash": source_hash()}
if len(sys.argv) > 4:   # matrix-core busy cycles of the same kernel (one more PMC pass)
    files = glob.glob(os.path.join(sys.argv[4], "**", "*counter_collection.csv"), recursive=True)
    rows = [r for f in files for r in csv.DictReader(open(f)) if dom(r["Kernel_Name"])]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], []).append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    step = {}
    for k, v in by.items():
        v.sort()
        v = v[len(v) // 2:]
        step[k] = (sum(x[1] for x in v), sum(x[2] for x in v), len(v))
    mf, sq, ga = (step.get(k, (0.0, 0, 0)) for k in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"))
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over every SIMD (1024): busy share of the matrix pipes
    # while the kernel runs = MFMA_BUSY / (1024 x GUI_ACTIVE / 8)
    out["mfma_busy"] = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE, one pass, python3 bench.py --steps 1 --warmup 1",
                        "SQ_VALU_MFMA_BUSY_CYCLES_per_step": mf[0], "SQ_BUSY_CYCLES_per_step": sq[0], "GRBM_GUI_ACTIVE_per_step": ga[0],
                        "launches": mf[2], "kernel_ns_per_step_in_this_pass": mf[1],
                        "mfma_busy_share": (mf[0] / (128.0 * ga[0])) if ga[0] else None,
                        "effective_clock_ghz": (ga[0] / 8.0 / mf[1]) if mf[1] else None,
                        "formula": "share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); clock = GRBM_GUI_ACTIVE / 8 / kernel time"}
if len(sys.argv) > 5:   # production launch time of the same kernel: un-instrumented kernel trace of 10 steps, last 60 % of the dispatches
    rows = [r for r in csv.DictReader(open(sys.argv[5]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * 0.4):]
    cg = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if dom(r["Kernel_Name"])]
    out["production_trace"] = {"source": "rocprofv3 --kernel-trace --stats, python3 bench.py --steps 10 --warmup 1 (side stream on), last 60 % of the dispatches",
                               "conv_gemm_launches": len(cg), "conv_gemm_total_ms": sum(cg) / 1e6, "avg_launch_us": sum(cg) / len(cg) / 1e3}
    # the weight-gradient GEMMs (side stream) of the same trace, per step: conv launches per step = launches_per_step of the PMC pass
    wg = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "wgrad" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"] and "unpack" not in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"]]
    steps = len(cg) / float(n1)
    out["production_trace"]["wgrad_gemm_ms_per_step"] = sum(wg) / 1e6 / steps
    out["production_trace"]["steps_in_window"] = steps
with open(sys.argv[3], "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))

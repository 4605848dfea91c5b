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


def per_step_kib(directory, counter):
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {directory}"
    rows = [r for f in files for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "conv_gemm_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    assert len(rows) % 2 == 0 and rows, (len(rows), "expected the same number of conv_gemm launches in the warm-up and the timed step")
    last = rows[len(rows) // 2:]
    return sum(float(r["Counter_Value"]) for r in last), len(last)


fetch_kib, n1 = per_step_kib(sys.argv[1], "FETCH_SIZE")
write_kib, n2 = per_step_kib(sys.argv[2], "WRITE_SIZE")
assert n1 == n2, (n1, n2)
hbm = (2.0 * fetch_kib + write_kib) * 1024.0
out = {"kernel": f"conv_gemm_kernel ({n1} launches per step: forward + dgrad)",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python3 bench.py --steps 1 --warmup 1",
       "fetch_size_kib_per_step": fetch_kib, "write_size_kib_per_step": write_kib,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request -> doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
       "hbm_bytes_per_step": hbm, "launches_per_step": n1, "hbm_bytes_per_launch": hbm / n1, "source_hash": source_hash()}
with open(sys.argv[3], "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))

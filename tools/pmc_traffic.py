"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into HBM bytes per launch for the GEMM kernel families.

Usage (on the GPU box, two separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit
one pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json ["yolov8s.yaml bs32 640"]

Units and gfx950 corrections: both counters are in KiB; FETCH_SIZE tallies the 128-byte requests of wide (16 B/lane)
streaming reads at 64 B, so read bytes = 2 * FETCH_SIZE * 1024 for these kernels (all their global reads are 16-byte
LDS-DMA loads); WRITE_SIZE is exact for the 16-byte stores of the staged epilogue.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

FAMILIES = {"igemm": "igemm_kernel", "wgrad": "wgrad_kernel"}


def collect(root, counter):
    per = defaultdict(lambda: [0.0, 0])
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {root}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                for fam, key in FAMILIES.items():
                    if key in name and "reduce" not in name:
                        per[fam][0] += float(row["Counter_Value"])
                        per[fam][1] += 1
    return per


def kernel_rev():
    """hash of the kernel sources the passes were taken on (same rule as bench.py: it reports the traffic only for these)."""
    h = hashlib.sha1()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "improving_yolov8_cbam_swinblock_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:12]


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else None  # "<model yaml> bs<batch> <imgsz>" as bench.py names it (absent: the default configuration)
    fetch = collect(fetch_dir, "FETCH_SIZE")
    write = collect(write_dir, "WRITE_SIZE")
    res = {"kernel_rev": kernel_rev(),
           "recipe": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, averaged per launch over every launch of the family in the profiled run",
           "families": {}}
    if workload:
        res["workload"] = workload
    for fam in FAMILIES:
        fk, fn = fetch[fam]
        wk, wn = write[fam]
        if not fn or not wn:
            continue
        rd = 2.0 * fk * 1024.0 / fn
        wr = wk * 1024.0 / wn
        res["families"][fam] = {"launches_fetch_pass": fn, "launches_write_pass": wn, "read_bytes_per_launch": round(rd),
                                "write_bytes_per_launch": round(wr), "hbm_bytes_per_launch": round(rd + wr)}
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()

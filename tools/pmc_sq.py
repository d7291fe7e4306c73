"""Aggregate rocprofv3 --pmc SQ counter passes per kernel (per-launch means) and derive the per-wave split and the MFMA-pipe duty.
usage: pmc_sq.py DIR [DIR ...] [--filter SUBSTRING]      (DIR: rocprofv3 -d output of one pass; counters of all passes are merged by kernel)
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and
SQ_BUSY_CYCLES count cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name[:110]


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        args = args[:i] + args[i + 2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for root in args:
        for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    if flt and flt not in k:
                        continue
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    acc[k]["__grid"] = [float(row.get("Grid_Size", 0) or 0), float(row.get("Workgroup_Size", 0) or 0)]
    for k, cs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
        n = max(v[1] for c, v in cs.items() if c != "__grid")
        print(f"## {short(k)}   ({n} launches; grid {cs['__grid'][0]:.0f} threads, workgroup {cs['__grid'][1]:.0f})")
        m = {c: v[0] / max(v[1], 1) for c, v in cs.items() if c != "__grid"}
        for c in sorted(m):
            print(f"{c:32s} {m[c]:16.1f}")
        waves = cs["__grid"][0] / 64.0 if cs["__grid"][0] else 0
        if waves and "SQ_WAVE_CYCLES" in m:
            wc = m["SQ_WAVE_CYCLES"] / waves
            line = f"per wave (quad-cycles): {wc:.0f}"
            for c, label in (("SQ_WAIT_ANY", "parked on waitcnt / barrier"), ("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_ACTIVE_INST_ANY", "issuing")):
                if c in m:
                    line += f" | {label} {m[c] / waves:.0f} ({100 * m[c] / m['SQ_WAVE_CYCLES']:.1f} %)"
            print(line)
            inst = " ".join(f"{c[9:]} {m[c] / waves:.0f}" for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if c in m)
            if inst:
                print("instructions per wave: " + inst + ("   (VALU includes MFMA)" if "SQ_INSTS_VALU" in m else ""))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
            kernel_cycles = m["GRBM_GUI_ACTIVE"] / 8.0
            print(f"MFMA pipe: {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0:.0f} busy cycles per SIMD of {kernel_cycles:.0f} kernel cycles = {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / kernel_cycles:.1f} % busy")
        elif "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m:
            print(f"MFMA pipe: {m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0:.0f} busy cycles per SIMD; SQ_BUSY_CYCLES {m['SQ_BUSY_CYCLES']:.0f}")
        print()


if __name__ == "__main__":
    main()

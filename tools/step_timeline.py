"""One training step (or one forward) as a kernel timeline, from a rocprofv3 --kernel-trace CSV.
usage: step_timeline.py TRACE.csv [--compare OTHER.csv] [--marker pack_batch_kernel] [--top N]
The step is the stretch between the last two launches of the marker kernel (the weight pack opens every step).  Prints per dispatch:
index, duration (us), gap to the previous kernel's end (us), grid, workgroup, kernel name; with --compare, the second trace's duration of
the same dispatch index beside it and the per-kernel-name totals of both."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)(<.*>)?", name)
    base = m.group(1) if m else name
    targs = m.group(2) or "" if m else ""
    return (base + targs)[:70]


def load(path, marker):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1),
                     int(r["Workgroup_Size_X"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 3:
        raise SystemExit(f"{path}: fewer than three launches of the marker '{marker}'")
    a, b = marks[-3], marks[-2]  # a full step that is followed by another one (not the tail of the run)
    return rows[a:b]


def main():
    args = sys.argv[1:]
    marker, other, top = "pack_batch_kernel", None, 0
    paths = []
    i = 0
    while i < len(args):
        if args[i] == "--marker":
            marker = args[i + 1]; i += 2
        elif args[i] == "--compare":
            other = args[i + 1]; i += 2
        elif args[i] == "--top":
            top = int(args[i + 1]); i += 2
        else:
            paths.append(args[i]); i += 1
    st = load(paths[0], marker)
    ot = load(other, marker) if other else None
    if ot is not None and len(ot) != len(st):
        print(f"# dispatch counts differ: {len(st)} vs {len(ot)} (per-index columns may be misaligned)")
    tot = defaultdict(lambda: [0.0, 0.0, 0])
    prev_end = st[0][0]
    span = (st[-1][1] - st[0][0]) / 1e3
    busy = sum(e - s for s, e, *_ in st) / 1e3
    print(f"# {len(st)} dispatches, wall {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us")
    lines = []
    for k, (s, e, name, grid, wg) in enumerate(st):
        d = (e - s) / 1e3
        gap = (s - prev_end) / 1e3
        prev_end = max(prev_end, e)
        sn = short(name)
        od = None
        if ot is not None and k < len(ot):
            od = (ot[k][1] - ot[k][0]) / 1e3
        t = tot[sn]
        t[0] += d
        t[2] += 1
        lines.append((k, d, gap, grid // max(wg, 1), wg, sn, od))
    show = sorted(lines, key=lambda l: -l[1])[:top] if top else lines
    for k, d, gap, nb, wg, sn, od in show:
        extra = f" | {od:8.1f} {od - d:+7.1f}" if od is not None else ""
        print(f"{k:4d} {d:8.1f} {gap:6.1f} {nb:7d}x{wg:<4d} {sn}{extra}")
    on = defaultdict(int)
    if ot is not None:  # the other trace summed BY NAME (the two traces may differ in dispatch count)
        for s, e, name, grid, wg in ot:
            tot[short(name)][1] += (e - s) / 1e3
            on[short(name)] += 1
    print("# per kernel name: us, (other us), launches" + (", (other launches)" if ot is not None else ""))
    for sn, (a, b, n) in sorted(tot.items(), key=lambda kv: -max(kv[1][0], kv[1][1])):
        print(f"{a:9.1f} {b:9.1f} {n:4d}" + (f" {on[sn]:4d}" if ot is not None else "") + f"  {sn}")


if __name__ == "__main__":
    main()

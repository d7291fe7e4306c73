"""Every GEMM launch of the profiled eager steps against ITS OWN roofline bound max(FLOP / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s).
Input: the file written by YMI_PROF_DUMP=<file> python bench.py --no-cpu-baseline --no-forward --sustained 0 (one line per launch: family,
measured us, bound us, GFLOP, MB; HIP-event brackets on the launch stream, ~3 us of dispatch inside every bracket).
usage: per_launch_bounds.py DUMP STEPS [--list]"""
import sys


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    rows = [tuple(float(v) for v in line.split()) for line in open(path) if line.strip()]
    per = len(rows) // steps
    rows = rows[-per:]  # the last profiled step
    names = {0: "igemm", 1: "wgrad"}
    for fam in (0, 1):
        r = [x for x in rows if int(x[0]) == fam]
        if not r:
            continue
        t, b = sum(x[1] for x in r), sum(x[2] for x in r)
        mf = [x for x in r if x[3] * 1e9 / 2.5e15 >= x[4] * 1e6 / 8e12]  # FLOP bound >= byte bound: MFMA-bound by shape
        hb = [x for x in r if x not in mf]
        def part(q):
            tq, bq = sum(x[1] for x in q), sum(x[2] for x in q)
            return f"{len(q)} launches {tq:.0f} us (bounds {bq:.0f} us = {bq / max(tq, 1e-9):.3f})"
        print(f"# {names[fam]}: {len(r)} launches {t:.0f} us (sum of bounds {b:.0f} us = {b / t:.3f}) | MFMA-bound by shape: {part(mf)} | HBM-bound by shape: {part(hb)}")
    if "--list" in sys.argv:
        print("# columns: family, measured us, bound us, GFLOP, algorithmic MB   (launch order)")
        for x in rows:
            print(f"{names[int(x[0])]:6s} {x[1]:7.1f} {x[2]:6.1f} {x[3]:8.2f} {x[4]:7.1f}")


if __name__ == "__main__":
    main()

#!/bin/bash
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r4_sched
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for sc in auto split tail auto split tail; do
  python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --no-forward --sustained 100 --schedule $sc > $out/bench_$sc.json 2> $out/bench_$sc.err || { tail -5 $out/bench_$sc.err; exit 1; }
  python3 - <<PY
import json
d=json.load(open("$out/bench_$sc.json"))
print("$sc", d["schedule"], "| img/s", d["value"], "ms/step", d["ms_per_step"], "sustained", d.get("sustained",{}).get("ms_per_step"))
PY
done

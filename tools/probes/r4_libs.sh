#!/bin/bash
# durations of selected kernels in one training step, per library build (YMI_LIB): tools/probes/r4_libs.sh TAG PATTERN lib1 lib2 ...
set -e
tag=$1; pat=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export YMI_LIB=$root/$lib
  name=$(basename $lib .so)
  rocprofv3 --kernel-trace --output-format csv -d $out/trace_$name -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --no-kernel-timing --sustained 0 --steps 6 --warmup 2 > $out/bench_$name.json 2> $out/trace_$name.err
  cp $(ls $out/trace_$name/*kernel_trace.csv $out/trace_$name/*/*kernel_trace.csv 2>/dev/null | head -1) $out/trace_$name.csv
  rm -rf $out/trace_$name
  echo "== $name: $(python3 -c "import json;d=json.load(open('$out/bench_$name.json'));print(d['ms_per_step'])") ms/step (under rocprofv3)"
  python3 $root/tools/step_timeline.py $out/trace_$name.csv | grep -E "$pat" | cut -c1-110
done

#!/bin/bash
# Round-end evidence on the GPU box, written under gpurun_out/<TAG>/ (copy what is to be judged into profiles/<TAG>_*):
#   bench.json                  default `python bench.py` line (taken again after the PMC passes, so that it carries `traffic`)
#   kernel_stats.csv            rocprofv3 --kernel-trace --stats of `bench.py --no-forward --no-cpu-baseline --sustained 0` (33 steps)
#   kernel_categories.txt       per-family ms/step of the same
#   bench_under_rocprof.json    the bench line printed by that profiled run
#   fwd_kernel_stats.csv / fwd_kernel_categories.txt   the same for `bench.py --forward-only` (the train-mode forward alone)
#   per_launch_bounds.txt       every GEMM launch of one eager step against its own roofline bound (tools/per_launch_bounds.py)
#   pmc_traffic.json            HBM bytes per launch of the GEMM families: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), with the
#                               workload they were taken on; also copied to profiles/<TAG>_pmc_traffic.json ON THE BOX for the second bench run
#   (second argument `cfg5`) config 5, yolov8m-cbam-swin384 at bs 16, 1280x1280:
#   bench_cfg5.json, cfg5_kernel_stats.csv, cfg5_kernel_categories.txt, cfg5_pmc_traffic.json
# usage: tools/round_profiles.sh TAG [cfg5|cfg5only]
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {  # name, steps in the table's divisor, bench.py arguments
  name=$1; div=$2; shift; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -o p -- python3 $root/bench.py "$@" > $out/${name}_under_rocprof.json 2> $out/trace_$name.err
  cp $(ls $out/trace_$name/*kernel_stats.csv $out/trace_$name/*/*kernel_stats.csv 2>/dev/null | head -1) $out/${name}_kernel_stats.csv
  python3 $root/tools/kernel_categories.py $out/${name}_kernel_stats.csv $div > $out/${name}_kernel_categories.txt
  rm -rf $out/trace_$name
  echo "$name trace done"
}
traffic() {  # output name, workload label, bench.py arguments
  name=$1; wl=$2; shift; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- python3 $root/bench.py "$@" > /dev/null 2> $out/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o p -- python3 $root/bench.py "$@" > /dev/null 2> $out/pmc_write.err
  python3 $root/tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/$name "$wl" > /dev/null
  cp $out/$name $root/profiles/${tag}_$name
  rm -rf $out/pmc_fetch $out/pmc_write
  echo "$name passes done"
}
if [ "$2" != "cfg5only" ]; then
  stats bench 33 --no-cpu-baseline --no-forward --sustained 0
  mv $out/bench_kernel_stats.csv $out/kernel_stats.csv; mv $out/bench_kernel_categories.txt $out/kernel_categories.txt
  stats fwd 32 --forward-only --steps 27
  traffic pmc_traffic.json "yolov8s.yaml bs32 640" --no-cpu-baseline --no-forward --sustained 0 --steps 3 --warmup 1
  # every GEMM launch against its own bound (HIP-event brackets of 4 eager steps; the last one is listed)
  rm -f $out/prof_dump.txt
  YMI_PROF_DUMP=$out/prof_dump.txt python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 --graph 0 --steps 4 --warmup 2 > /dev/null 2> $out/prof_dump.err
  python3 $root/tools/per_launch_bounds.py $out/prof_dump.txt 4 --list > $out/per_launch_bounds.txt || true
  # the default line, now that the PMC passes of THIS kernel revision exist (bench.py reports `traffic` only from a matching file)
  python3 $root/bench.py > $out/bench.json 2> $out/bench.err
  echo "bench done"
fi
if [ "$2" = "cfg5" ] || [ "$2" = "cfg5only" ]; then
  C5="--model yolov8m-cbam-swin384.yaml --batch 16 --imgsz 1280 --no-cpu-baseline"
  stats cfg5 21 $C5 --no-forward --sustained 0 --steps 10 --warmup 3
  traffic cfg5_pmc_traffic.json "yolov8m-cbam-swin384.yaml bs16 1280" $C5 --no-forward --sustained 0 --steps 3 --warmup 1
  python3 $root/bench.py $C5 --steps 10 --warmup 3 --sustained 20 > $out/bench_cfg5.json 2> $out/bench_cfg5.err
  echo "cfg5 done"
fi
head -14 $out/kernel_categories.txt 2>/dev/null || true
cat $out/*pmc_traffic.json

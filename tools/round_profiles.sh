#!/bin/bash
# Round-end evidence on the GPU box, written under gpurun_out/<TAG>/ (copy what is to be judged into profiles/):
#   bench.json               default `python bench.py` line
#   kernel_stats.csv         rocprofv3 --kernel-trace --stats of `bench.py --no-forward --no-cpu-baseline --sustained 0` (33 steps)
#   kernel_categories.txt    per-family ms/step of the same
#   bench_under_rocprof.json the bench line printed by that profiled run
#   pmc_traffic.json         HBM bytes per launch of the GEMM families: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
#   bench_cfg5.json          (second argument `cfg5`) config 5: yolov8m-cbam-swin384 at bs 16, 1280x1280
# usage: tools/round_profiles.sh TAG [cfg5]
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 > $out/bench_under_rocprof.json 2> $out/trace.err
cp $(ls $out/trace/*kernel_stats.csv $out/trace/*/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv
python3 $root/tools/kernel_categories.py $out/kernel_stats.csv 33 > $out/kernel_categories.txt
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 --steps 3 --warmup 1 > /dev/null 2> $out/pmc_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o p -- python3 $root/bench.py --no-cpu-baseline --no-forward --sustained 0 --steps 3 --warmup 1 > /dev/null 2> $out/pmc_write.err
echo "write pass done"
python3 $root/tools/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json > /dev/null
# the default line once more, now that the PMC passes of THIS kernel revision exist (bench.py reports `traffic` only from a matching file)
cp $out/pmc_traffic.json $root/profiles/r03_pmc_traffic.json
python3 $root/bench.py > $out/bench.json 2> $out/bench.err
if [ "$2" = "cfg5" ]; then
  python3 $root/bench.py --model yolov8m-cbam-swin384.yaml --batch 16 --imgsz 1280 --no-cpu-baseline --steps 10 --warmup 3 --sustained 20 > $out/bench_cfg5.json 2> $out/bench_cfg5.err
  echo "cfg5 done"
fi
rm -rf $out/pmc_fetch $out/pmc_write $out/trace/*kernel_trace.csv $out/trace/*/*kernel_trace.csv 2>/dev/null || true
cat $out/kernel_categories.txt | head -14
cat $out/pmc_traffic.json

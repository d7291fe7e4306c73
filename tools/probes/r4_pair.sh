#!/bin/bash
set -e
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/r4_pair
mkdir -p $out
cd $root
for v in 0 3; do
  YMI_XCD_SHIFT=$v python3 tools/conv_bench.py --producer --iters 40 > $out/pair_shift$v.txt 2>&1
done
paste -d'\n' $out/pair_shift0.txt $out/pair_shift3.txt | tail -70

#!/bin/bash
# the same sweep on config 5 (yolov8m-cbam-swin384, bs 16, 1280x1280) and on the forward alone
root=${GRAFT_REPO_ROOT:-/root/repo}
list=${1:-"8,2048,1024 32,1024,512 16,1024,512"}
cd /tmp
for rep in 1 2; do
for cfg in $list; do
  IFS=, read a b c <<< "$cfg"
  YMI_EW_PPT=$a YMI_EW_CAP=$b YMI_RED_CAP=$c python3 $root/bench.py --model yolov8m-cbam-swin384.yaml --batch 16 --imgsz 1280 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 --sustained 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 ppt $a cap $b redcap $c :', d['ms_per_step'], d['sustained']['ms_per_step'], 'fwd', d['forward']['ms'])"
  YMI_EW_PPT=$a YMI_EW_CAP=$b YMI_RED_CAP=$c python3 $root/bench.py --no-cpu-baseline --no-kernel-timing --sustained 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('s    ppt $a cap $b redcap $c :', d['ms_per_step'], d['sustained']['ms_per_step'], 'fwd', d['forward']['ms'])"
done
done

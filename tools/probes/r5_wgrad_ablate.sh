#!/bin/bash
# diagnostic: the weight-gradient GEMM with one ingredient of its K loop removed at a time (results wrong by design)
root=${GRAFT_REPO_ROOT:-/root/repo}
cs=$root/improving_yolov8_cbam_swinblock_amd/csrc
if [ "$1" = build ]; then
  for n in ${ABLS:-1 2 4 8 9 3 15}; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -DYMI_WGRAD_ABL=$n -c $cs/wgrad.hip -o /tmp/wgrad_abl$n.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/improving_yolov8_cbam_swinblock_amd/libyolo_wabl$n.so $(ls $cs/*.o | grep -v wgrad.o) /tmp/wgrad_abl$n.o ) &
  done
  wait
  exit 0
fi
for n in 0 ${ABLS:-1 2 4 8 9 3 15}; do
  lib=$root/improving_yolov8_cbam_swinblock_amd/libyolo_wabl$n.so
  [ $n = 0 ] && lib=$root/improving_yolov8_cbam_swinblock_amd/libyolo_mi355.so
  echo "== wgrad ablation mask $n"; YMI_LIB=$lib python3 $root/tools/conv_bench.py --ops wgrad --iters 20 2>&1 | grep -E "det.cv3|L5 |c2f4.m|c2f2.m|c2f6.cv2|L3 |totals"
done

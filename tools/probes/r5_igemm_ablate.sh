#!/bin/bash
# diagnostic: the ping-pong GEMM tile with one ingredient of its K loop removed at a time (results wrong by design)
root=${GRAFT_REPO_ROOT:-/root/repo}
cs=$root/improving_yolov8_cbam_swinblock_amd/csrc
if [ "$1" = build ]; then
  for n in ${ABLS:-1 2 4 7 8 16 23}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -DYMI_IGEMM_ABL=$n -c $cs/igemm.hip -o /tmp/igemm_abl$n.o || exit 1
    objs=$(ls $cs/*.o | grep -v igemm.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/improving_yolov8_cbam_swinblock_amd/libyolo_iabl$n.so $objs /tmp/igemm_abl$n.o || exit 1
  done
  exit 0
fi
for n in 0 ${ABLS:-1 2 4 7 8 16 23}; do
  lib=$root/improving_yolov8_cbam_swinblock_amd/libyolo_iabl$n.so
  [ $n = 0 ] && lib=$root/improving_yolov8_cbam_swinblock_amd/libyolo_mi355.so
  echo "== igemm ablation mask $n"; YMI_LIB=$lib python3 $root/tools/conv_bench.py --ops fwd,dgrad --iters 20 2>&1 | grep -E "det.cv3.0|L5 |c2f4|c2f2|L1 |L2 |L3 |c2f6|GF|totals"
done

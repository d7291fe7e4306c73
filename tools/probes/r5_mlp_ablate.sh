#!/bin/bash
# diagnostic: the fused SwinBlock MLP forward kernel with one ingredient removed at a time (results wrong by design): where do its cycles go?
# builds variant libraries next to the product library (run in the container), then on the GPU box: tools/probes/r5_mlp_ablate.sh run
root=${GRAFT_REPO_ROOT:-/root/repo}
cs=$root/improving_yolov8_cbam_swinblock_amd/csrc
if [ "$1" = build ]; then
  for n in ${ABLS:-1 2 4 8 16 64}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -DYMI_MLP_ABL=$n -c $cs/swin_mlp.hip -o /tmp/swin_mlp_abl$n.o || exit 1
    objs=$(ls $cs/*.o | grep -v swin_mlp.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/improving_yolov8_cbam_swinblock_amd/libyolo_abl$n.so $objs /tmp/swin_mlp_abl$n.o || exit 1
  done
  exit 0
fi
for n in 0 ${ABLS:-1 2 4 8 16 64}; do
  lib=$root/improving_yolov8_cbam_swinblock_amd/libyolo_abl$n.so
  [ $n = 0 ] && lib=$root/improving_yolov8_cbam_swinblock_amd/libyolo_mi355.so
  echo "== ablation mask $n"; YMI_MLP_STAMPS=$(( (n & 32) ? 1 : 0 )) YMI_LIB=$lib python3 $root/tools/probes/swin_mlp_probe.py 2>&1 | grep -E "warm fused train|half"
done

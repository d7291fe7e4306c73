#!/bin/bash
# diagnostic: -DYMI_STAMPS builds of igemm.hip (optionally with an ablation mask) and the phase stamps of one mid-grid workgroup
root=${GRAFT_REPO_ROOT:-/root/repo}
cs=$root/improving_yolov8_cbam_swinblock_amd/csrc
if [ "$1" = build ]; then
  for n in ${ABLS:-0 7}; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -DYMI_STAMPS -DYMI_IGEMM_ABL=$n -c $cs/igemm.hip -o /tmp/igemm_st$n.o || exit 1
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast -DYMI_STAMPS -DYMI_WGRAD_ABL=${WABL:-0} -c $cs/wgrad.hip -o /tmp/wgrad_st.o || exit 1
    objs=$(ls $cs/*.o | grep -v -e igemm.o -e wgrad.o)
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/improving_yolov8_cbam_swinblock_amd/libyolo_ist$n.so $objs /tmp/igemm_st$n.o /tmp/wgrad_st.o || exit 1
  done
  exit 0
fi
for n in ${ABLS:-0 7}; do
  echo "== stamps, igemm ablation mask $n"
  YMI_LIB=$root/improving_yolov8_cbam_swinblock_amd/libyolo_ist$n.so python3 $root/tools/conv_bench.py --ops ${OPS:-fwd,dgrad} --iters 10 --stamps --only "${ONLY:-det.cv3[0]}" 2>&1
done

#!/bin/bash
# timing-only ablation builds of the fused residual-block kernel: build_exp/libc2m_rb<mask>.so (C2M_RB_ABL, csrc/conv3x3_resblock.hip)
set -e
cd "$(dirname "$0")/../c2-matching_amd/csrc"
mkdir -p ../../build_exp
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics -Wno-unused-function -Wno-int-to-pointer-cast -Wno-int-to-void-pointer-cast"
OTHERS="c2m_api.o corr_argmax.o corr_filter.o dcn_v2.o conv3x3.o conv3x3_split.o conv3x3_wgrad.o"
for m in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DC2M_RB_ABL=$m ${RB_DEFS} -c conv3x3_resblock.hip -o ../../build_exp/rb_$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS ../../build_exp/rb_$m.o -o ../../build_exp/libc2m_rb$m.so
  echo built build_exp/libc2m_rb$m.so
done

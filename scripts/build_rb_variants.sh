#!/bin/bash
# Experimental builds (NOT the product library): build_exp/exp/libc2m_hip.so = product + csrc/experimental/ kernels; with masks,
# build_exp/rb<mask>/libc2m_hip.so = the same with the fused residual-block kernel's compile-time ablation C2M_RB_ABL=<mask>.
# Load one with C2M_LIB=<path>.   usage: build_rb_variants.sh [mask ...]
set -e
cd "$(dirname "$0")/../c2-matching_amd/csrc"
make -j8 EXPERIMENTAL=1 OUT=../../build_exp/exp all | tail -1
for m in "$@"; do
  make -j8 EXPERIMENTAL=1 OUT=../../build_exp/rb$m EXTRA=-DC2M_RB_ABL=$m all | tail -1
done

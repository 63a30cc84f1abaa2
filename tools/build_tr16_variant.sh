#!/bin/bash
# instrumented variant of the direct bf16 backward-weight kernel only: tools/build_tr16_variant.sh NAME "-DBWT_ABL=8 ..."  ->  multitalent_amd/libmtseg_hip_NAME.so
# (MT_LIB_VARIANT=libmtseg_hip_NAME.so python tools/bench_bwdw16.py)
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c bwdw_tr16.hip -o /tmp/bwdw_tr16_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so conv_lds.o /tmp/bwdw_tr16_$name.o pointwise.o norm.o loss.o optim.o infer.o prep.o errors.o

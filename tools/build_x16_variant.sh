#!/bin/bash
# instrumented variant of conv_x16.hip: tools/build_x16_variant.sh NAME "-DX16_ABL=1 ..."  ->  multitalent_amd/libmtseg_hip_NAME.so
# (MT_LIB_VARIANT=libmtseg_hip_NAME.so python tools/bench_fwd16.py ...)
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c conv_x16.hip -o /tmp/conv_x16_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so conv_lds.o /tmp/conv_x16_$name.o bwdw_tr16.o pointwise.o norm.o loss.o optim.o infer.o prep.o errors.o

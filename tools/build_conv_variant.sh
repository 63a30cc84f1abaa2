#!/bin/bash
# instrumented variant of conv_lds.hip: tools/build_conv_variant.sh NAME "-DTS_ABL=1 ..."  ->  multitalent_amd/libmtseg_hip_NAME.so
# (MT_LIB_VARIANT=libmtseg_hip_NAME.so python tools/bench_conv.py ...)
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c conv_lds.hip -o /tmp/conv_lds_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so /tmp/conv_lds_$name.o conv_x16.o bwdw_tr16.o pointwise.o norm.o loss.o optim.o infer.o prep.o errors.o

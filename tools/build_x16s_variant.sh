#!/bin/bash
# instrumented variant of conv_x16s.hip: tools/build_x16s_variant.sh NAME "-DXS_ABL=1 ..."  ->  multitalent_amd/libmtseg_hip_NAME.so
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c conv_x16s.hip -o /tmp/conv_x16s_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so conv_lds.o conv_x16.o /tmp/conv_x16s_$name.o bwdw_tr16.o pointwise.o norm.o loss.o optim.o infer.o prep.o errors.o

#!/bin/bash
# instrumented variant of conv_march16_kernel only: tools/build_march16_variant.sh NAME "-DCM_ABL=8 ..."  ->  multitalent_amd/libmtseg_hip_NAME.so
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -pragma-unroll-threshold=1048576 "$@" -c conv_march16.hip -o /tmp/conv_march16_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so conv_lds.o bwdw_tr16.o /tmp/conv_march16_$name.o pointwise.o norm.o loss.o optim.o infer.o prep.o errors.o

#!/bin/bash
# instrumented variant of pointwise.hip: tools/build_pw_variant.sh NAME "-DPW_ABL=1"  ->  multitalent_amd/libmtseg_hip_NAME.so
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c pointwise.hip -o /tmp/pointwise_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so conv_lds.o /tmp/pointwise_$name.o norm.o loss.o optim.o infer.o prep.o errors.o

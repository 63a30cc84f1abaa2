#!/bin/bash
# variant of norm.hip: tools/build_norm_variant.sh NAME "-DMT_VB_BLOCKS=1024"  ->  multitalent_amd/libmtseg_hip_NAME.so
set -e
cd "$(dirname "$0")/../multitalent_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c norm.hip -o /tmp/norm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmtseg_hip_$name.so conv_lds.o pointwise.o /tmp/norm_$name.o loss.o optim.o infer.o prep.o errors.o

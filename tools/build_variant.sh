#!/bin/bash
# dev tool: build libtmat_hip with extra -D flags for unet_kernels.hip into build_variants/libtmat_<name>.so
# usage: tools/build_variant.sh <name> "<flags>" [source stem, default unet_kernels]     (run tools/build.py first: the other objects are reused)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
C=$ROOT/tissue-model-analysis-tools_amd/csrc
mkdir -p $ROOT/build_variants
SRC=${3:-unet_kernels}
/opt/rocm/bin/hipcc -x hip -c $C/$SRC.hip -o $ROOT/build_variants/unet_$1.o --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DTMAT_DEV_BUILD $2
OBJS=$(ls $C/build/*.o | grep -v $SRC)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $ROOT/build_variants/libtmat_$1.so $OBJS $ROOT/build_variants/unet_$1.o -pthread -ldl
rm -f $ROOT/build_variants/unet_$1.o
echo built $ROOT/build_variants/libtmat_$1.so

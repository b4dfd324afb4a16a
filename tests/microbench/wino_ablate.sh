#!/bin/bash
# In-situ ablation of k_wino_conv: builds libxq_hip variants with -DXQ_ABL=<bits> (see csrc/xq_conv.hip) into
# tests/microbench/lab/ (git-ignored; the .so files travel to the GPU box with gpurun) and, with "run", times each one with
# tests/perf_conv.py on the GPU.  Not a test; results are wrong by construction, only the launch time is read.
#   tests/microbench/wino_ablate.sh build "0 1 2 4 8 16 32 64"      (in the build container)
#   XQ_EXTRA="-DXQ_WIDE_POOL=20" XQ_TAG=_p20 tests/microbench/wino_ablate.sh build "0"   (extra defines, tagged file name)
#   tests/microbench/wino_ablate.sh run   "0 1 2 4 8 16 32 64"      (on the GPU box)
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
CSRC=$HERE/../../xiangqi-alphazero_amd/csrc
LAB=$HERE/lab
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-but-set-variable -Wno-unused-variable"
mkdir -p "$LAB"
if [ "$1" = build ]; then
    OTHERS=$(ls $CSRC/*.o | grep -v xq_conv.o)
    for v in $2; do
        /opt/rocm/bin/hipcc $FLAGS $XQ_EXTRA -DXQ_ABL=$v -c $CSRC/xq_conv.hip -o $LAB/xq_conv_abl_$v.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LAB/libxq_abl_$v$XQ_TAG.so $LAB/xq_conv_abl_$v.o $OTHERS
        rm -f $LAB/xq_conv_abl_$v.o
    done
else
    for v in $2; do
        printf "ABL=%-4s " $v
        XQ_HIP_LIB=$LAB/libxq_abl_$v.so python $HERE/../perf_conv.py ${3:-8192} ${4:-256}
    done
fi

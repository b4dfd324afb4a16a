#!/bin/bash
# Build libxq_hip.so with csrc/xq_conv.hip taken from a git ref into tests/microbench/lab/libxq_<name>.so (A/B timing of
# kernel revisions on ONE box: boxes differ by several per cent).   tests/microbench/build_ref.sh <git-ref> <name>
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$HERE/../..; CSRC=$ROOT/xiangqi-alphazero_amd/csrc; LAB=$HERE/lab
mkdir -p $LAB; TMP=$(mktemp -d)
git -C $ROOT show $1:xiangqi-alphazero_amd/csrc/xq_conv.hip | sed "s#\"xq_common.h\"#\"$CSRC/xq_common.h\"#" > $TMP/xq_conv.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w -I$CSRC -c $TMP/xq_conv.hip -o $TMP/xq_conv.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LAB/libxq_$2.so $TMP/xq_conv.o $(ls $CSRC/*.o | grep -v xq_conv.o)
rm -rf $TMP

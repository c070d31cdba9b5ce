#!/bin/bash
# A/B of a kernel change on ONE box (boxes of the pool differ by +-3 %): builds the library of the last COMMIT beside the working
# tree's, as csrc/exp/head.so.  Then, in one gpurun call:
#   bash tools/kernel_times.sh head $PWD/3d-.../csrc/exp/head.so && bash tools/kernel_times.sh new ''
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=3d-gaussian-splatting-for-novel-view-synthesis_amd
tmp=$(mktemp -d)
mkdir -p $tmp/$pkg/csrc $tmp/include
git -C $root show HEAD:include/gsplat_mi355x.h > $tmp/include/gsplat_mi355x.h
for f in gs_math.h gs_body.h gsplat_kernels.hip gsplat_loss.hip gsplat_optim.hip; do git -C $root show HEAD:$pkg/csrc/$f > $tmp/$pkg/csrc/$f; done
(cd $tmp/$pkg/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fno-gpu-rdc -Wno-unused-result -o head.so gsplat_kernels.hip gsplat_loss.hip gsplat_optim.hip)
mkdir -p $root/$pkg/csrc/exp
cp $tmp/$pkg/csrc/head.so $root/$pkg/csrc/exp/head.so
rm -rf $tmp
echo "built $pkg/csrc/exp/head.so from $(git -C $root rev-parse --short HEAD)"

#!/bin/bash
# NOTE (round 6): the -DPMN_IEEE_DIV / -DPMN_ATEN_GPU_DIV / -DPMN_POSE_FMA blocks no longer live in the product sources: apply
# scripts/experiments/source_switches/attribution_and_probe_switches.patch to a scratch copy of patchmatchnet_amd/csrc first.
# Round 5 A/B builds of libpmn_hip.so for the PixelwiseNet launch (scripts/gpu_r5_pixelwise_ab.sh runs them on one box):
#   base   gather_corr.hip of the given git revision (default: round 4's last commit c9b86bd): flat corner loads, workgroup-tile PixelwiseNet
#   g0     this tree, -DPMN_PW=0: global corner loads, workgroup-tile PixelwiseNet kernel
#   g4     this tree, wave-private PixelwiseNet, 4 pixels per wave     g2 / g2w5: 2 pixels per wave at 4 / 5 waves per SIMD
# Only gather_corr.hip differs; the other objects are the tree's.  Output: build/pw/libpmn_hip_<variant>.so (git-ignored, travels with gpurun).
set -e
cd "$(dirname "$0")/.."
REV=${1:-c9b86bd}
CS=patchmatchnet_amd/csrc
make -s -C $CS -j8
mkdir -p build/pw
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -I$CS"
OTHERS=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v gather_corr.o)
git show $REV:$CS/gather_corr.hip > build/pw/gather_corr_base.hip
sed -i 's#"gather_common.hpp"#"../../patchmatchnet_amd/csrc/gather_common.hpp"#' build/pw/gather_corr_base.hip
build() { # name source defines...
  local name=$1 src=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $src -o build/pw/gather_corr_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/pw/libpmn_hip_$name.so build/pw/gather_corr_$name.o $OTHERS
  echo "built build/pw/libpmn_hip_$name.so"
}
build base build/pw/gather_corr_base.hip &
build g0 $CS/gather_corr.hip -DPMN_PW=0 &
build g4 $CS/gather_corr.hip -DPMN_PW=4 -DPMN_PW_WAVES=4 &
build g2 $CS/gather_corr.hip -DPMN_PW=2 -DPMN_PW_WAVES=4 &
build g2w5 $CS/gather_corr.hip -DPMN_PW=2 -DPMN_PW_WAVES=5 &
wait
# attribution build for scripts/rocm_parity_probe.py: rot @ [x y 1]^T as a k-ordered fma chain (-DPMN_POSE_FMA), everything else the tree's
( /opt/rocm/bin/hipcc $FLAGS -DPMN_POSE_FMA -c $CS/gather_corr.hip -o build/pw/gather_corr_posefma.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/pw/libpmn_hip_posefma.so build/pw/gather_corr_posefma.o $OTHERS
  echo "built build/pw/libpmn_hip_posefma.so" )
# attribution build: ATen's GPU division by a host scalar (x * (1/c)) in the normalisations, u/48 and index/(D-1) (-DPMN_ATEN_GPU_DIV)
( objs=""
  for f in gather_corr aggregate hypotheses misc; do
    /opt/rocm/bin/hipcc $FLAGS -DPMN_ATEN_GPU_DIV -c $CS/$f.hip -o build/pw/${f}_atendiv.o; objs="$objs build/pw/${f}_atendiv.o"
  done
  rest=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v 'gather_corr.o\|aggregate.o\|hypotheses.o\|misc.o')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/pw/libpmn_hip_atendiv.so $objs $rest
  echo "built build/pw/libpmn_hip_atendiv.so" )
# stem A/B (scripts/stem_ab.py): the tree's library with the stem's workgroup tile 16 rows high (rounds 3-4) instead of 32
( /opt/rocm/bin/hipcc $FLAGS -DPMN_STEM_TH=16 -c $CS/conv_f16s.hip -o build/pw/conv_f16s_stem16.o
  rest=$(ls $CS/*.o | grep -v '\.x\.o' | grep -v 'conv_f16s.o')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build/pw/libpmn_hip_stem16.so build/pw/conv_f16s_stem16.o $rest
  echo "built build/pw/libpmn_hip_stem16.so" )
# PixelwiseNet launch, two lane groups per pixel walking interleaved hypotheses (-DPMN_PW_INTERLEAVE=1) instead of two blocks
build g2il $CS/gather_corr.hip -DPMN_PW=2 -DPMN_PW_WAVES=4 -DPMN_PW_INTERLEAVE=1

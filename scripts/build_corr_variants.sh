#!/bin/bash
# Builds variant RESEARCH libraries (libpmn_hip_experimental.so flavours) of experimental/corr_mfma.hip for a same-box A/B (scripts/gpu_corr_variants.sh):
#   scripts/build_corr_variants.sh NAME "-DPMN_CM_QP=64 ..." [NAME2 "flags" ...]
set -e
cd "$(dirname "$0")/../patchmatchnet_amd/csrc"
mkdir -p ../../scripts/microbench/variants
make -s -j8 EXPERIMENTAL=1
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -DPMN_EXPERIMENTAL $flags -c experimental/corr_mfma.hip -o /tmp/corr_mfma_$name.o
  objs=$(ls *.x.o experimental/*.x.o | grep -v 'corr_mfma')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/microbench/variants/libpmn_$name.so $objs /tmp/corr_mfma_$name.o
  echo "built variant $name ($flags)"
done

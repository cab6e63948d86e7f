#!/bin/bash
# scripts/repro/pk_opsel_matrix.hip -> gpurun_out/r06_pk_opsel_matrix.log
mkdir -p gpurun_out build
L=gpurun_out/r06_pk_opsel_matrix.log
[ -x build/pk_opsel_matrix ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build/pk_opsel_matrix scripts/repro/pk_opsel_matrix.hip
timeout 300 build/pk_opsel_matrix 2>&1 | grep -av "amdgpu.ids" > $L
cat $L

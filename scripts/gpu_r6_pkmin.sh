#!/bin/bash
# scripts/repro/pk_inplace_min.hip -> gpurun_out/r06_pk_min.log
mkdir -p gpurun_out build
L=gpurun_out/r06_pk_min.log
: > $L
[ -x build/pk_inplace_min ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build/pk_inplace_min scripts/repro/pk_inplace_min.hip
for i in 1 2; do timeout 120 build/pk_inplace_min 2>&1 | grep -av "amdgpu.ids" >> $L; echo >> $L; done
cat $L

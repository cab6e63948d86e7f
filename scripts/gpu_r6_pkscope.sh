#!/bin/bash
# Does the victim have to share a CU with the MFMA waves?  pk_inplace_min with 1024 / 256 / 64 / 16 workgroups of the MFMA kernel
# (4 per CU on every CU ... 16 CUs of 256) -> gpurun_out/r06_pk_scope.log
mkdir -p gpurun_out build
L=gpurun_out/r06_pk_scope.log
: > $L
[ -x build/pk_inplace_min ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o build/pk_inplace_min scripts/repro/pk_inplace_min.hip
for db in 1024 256 64 16; do
  echo "## MFMA kernel: $db workgroups of 256 threads" >> $L
  timeout 120 build/pk_inplace_min 2048 4000 $db 2>&1 | grep -a "beside" | grep -a "src1\|not in place" >> $L
done
cat $L

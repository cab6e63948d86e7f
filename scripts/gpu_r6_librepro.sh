#!/bin/bash
# The torch-free library-level reproducer of lesson 46 against the product build and the build without the fix.
mkdir -p gpurun_out
L=gpurun_out/r06_library_repro.log
: > $L
[ -x build/library_overlap_repro ] || { mkdir -p build; /opt/rocm/bin/hipcc -O2 -o build/library_overlap_repro scripts/repro/library_overlap_repro.cpp -ldl; }
[ -f build/wc/libpmn_hip_nosettle.so ] || bash scripts/build_waitcnt_variants.sh
for i in 1 2; do
  for cw in 0 1; do
    for lib in patchmatchnet_amd/csrc/libpmn_hip.so build/wc/libpmn_hip_nosettle.so; do
      timeout 120 build/library_overlap_repro $lib 24 400 $cw 2>&1 | grep -av "amdgpu.ids" >> $L
    done
  done
done
echo "## GPU_MAX_HW_QUEUES=1 (one hardware queue: the two streams serialise)" >> $L
GPU_MAX_HW_QUEUES=1 timeout 120 build/library_overlap_repro build/wc/libpmn_hip_nosettle.so 24 400 2>&1 | grep -av "amdgpu.ids" >> $L
cat $L

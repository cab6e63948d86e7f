#!/bin/bash
# scripts/build_settle_probes.sh variants under the torch-free reproducer -> gpurun_out/r06_settle_probes.log
mkdir -p gpurun_out
L=gpurun_out/r06_settle_probes.log
: > $L
[ -f build/wc/libpmn_hip_settle_empty.so ] || bash scripts/build_settle_probes.sh > /dev/null 2>&1  # (scratch builds are not kept in the tree)
[ -f build/wc/libpmn_hip_nosettle.so ] || bash scripts/build_waitcnt_variants.sh > /dev/null 2>&1
[ -x build/library_overlap_repro ] || { mkdir -p build; /opt/rocm/bin/hipcc -O2 -o build/library_overlap_repro scripts/repro/library_overlap_repro.cpp -ldl; }
for i in 1 2; do
  for v in nosettle settle_empty settle_nop settle_copy; do
    timeout 120 build/library_overlap_repro build/wc/libpmn_hip_$v.so 24 400 2>&1 | grep -av "amdgpu.ids" | sed 's/ (ABI.*when the victims finished)//' >> $L
  done
  timeout 120 build/library_overlap_repro patchmatchnet_amd/csrc/libpmn_hip.so 24 400 2>&1 | grep -av "amdgpu.ids" | sed 's/ (ABI.*when the victims finished)//' >> $L
done
cat $L

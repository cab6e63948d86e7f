#!/bin/bash
# every build/wc/libpmn_hip_tail_*.so under the torch-free reproducer -> gpurun_out/r06_tail_probes.log
mkdir -p gpurun_out
L=gpurun_out/r06_tail_probes.log
: > $L
[ -f build/wc/libpmn_hip_tail_ta_tb.so ] || bash scripts/build_tail_probes.sh ta tb wa wb b2 ta_tb wa_wb ta_tb_wa_wb_b2 > /dev/null 2>&1  # (scratch builds are not kept in the tree)
[ -x build/library_overlap_repro ] || { mkdir -p build; /opt/rocm/bin/hipcc -O2 -o build/library_overlap_repro scripts/repro/library_overlap_repro.cpp -ldl; }
for lib in build/wc/libpmn_hip_nosettle.so build/wc/libpmn_hip_tail_*.so; do
  for i in 1 2; do
    timeout 120 build/library_overlap_repro $lib 24 400 2>&1 | grep -av "amdgpu.ids" | sed 's/ (ABI.*when the victims finished)//' >> $L
  done
done
cat $L

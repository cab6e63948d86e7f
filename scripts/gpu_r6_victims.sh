#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_victims.log
: > $L
timeout 1200 python scripts/overlap_victims.py --reps 24 2>&1 | grep -av "amdgpu.ids" | tail -70 | tee -a $L
echo "== forcezero waitcnt build of gather_corr.hip" | tee -a $L
timeout 900 python scripts/overlap_victims.py --reps 24 --lib build/wc/libpmn_hip_fz.so --only feature_weight,warp_correlate --disturbers forward,stream 2>&1 | grep -av "amdgpu.ids" | tail -20 | tee -a $L
echo "== -O1 build of gather_corr.hip" | tee -a $L
timeout 900 python scripts/overlap_victims.py --reps 24 --lib build/wc/libpmn_hip_O1.so --only feature_weight,warp_correlate --disturbers forward,stream 2>&1 | grep -av "amdgpu.ids" | tail -20 | tee -a $L

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_bisect.log
: > $L
timeout 900 python scripts/overlap_bisect.py --trials 6 2>&1 | grep -av "amdgpu.ids" | tail -80 | tee -a $L
echo "== small" | tee -a $L
timeout 600 python scripts/overlap_bisect.py --height 480 --width 640 --views 4 --trials 6 --disturb 80 2>&1 | grep -av "amdgpu.ids" | tail -60 | tee -a $L
echo "== featurenet disturber" | tee -a $L
timeout 600 python scripts/overlap_bisect.py --trials 4 --disturber featurenet 2>&1 | grep -av "amdgpu.ids" | tail -40 | tee -a $L

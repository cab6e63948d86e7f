#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_allvictims.log
: > $L
timeout 1500 python scripts/overlap_pairs.py ${1:+--lib $1} --victims $(seq -s, 0 44) --disturbers 9:,43:,micro:\ fp16 --reps 24 2>&1 | grep -av "amdgpu.ids" | grep -a "^victim\|^lib" | cut -c1-260 | tee -a $L

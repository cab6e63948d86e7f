#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_anatomy.log
: > $L
PMN_ANATOMY="$4:$2" timeout 900 python scripts/overlap_pairs.py --lib $1 --victims $4 --disturbers "$2" --reps 4 2>&1 | grep -av "amdgpu.ids" | grep -a "trial\|    at" | head -${3:-70} | tee -a $L

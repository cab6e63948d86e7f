#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_micro.log
: > $L
timeout 900 python scripts/overlap_pairs.py --victims 28,22 --disturbers micro,9:,torch 2>&1 | grep -av "amdgpu.ids" | grep -a "victim" | tee -a $L

#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
L=gpurun_out/r06_pairs.log
: > $L
timeout 1500 python scripts/overlap_pairs.py 2>&1 | grep -av "amdgpu.ids" | tail -30 | tee -a $L

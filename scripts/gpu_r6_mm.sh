#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/repro/micro_matrix.py "$@" 2>&1 | grep -av "amdgpu.ids" | tee gpurun_out/r06_micro_matrix.log

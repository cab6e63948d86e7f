#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python scripts/repro/mfma_vs_valu.py 2>&1 | grep -av "amdgpu.ids" | tee gpurun_out/r06_mfma_vs_valu.log

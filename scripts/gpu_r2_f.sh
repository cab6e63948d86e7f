#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/graph_try.py > gpurun_out/graph_try.log 2>&1
bash scripts/gpu_profile.sh 24 > gpurun_out/profile_now.txt 2>&1
cat gpurun_out/graph_try.log | grep -v amdgpu.ids; head -45 gpurun_out/profile_now.txt | cut -c1-170

#!/bin/bash
# gpurun payload (round 5): the general-rig fixture through the HIP parity tests + identical-input parity against the reference on ROCm at configs[2] / [4]
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.jsonl
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_fullsize_parity.py -q -m gpu -k "rig or identical_inputs" --durations=5 2>&1 | grep -v "Warning\|warn" | tail -40 | tee gpurun_out/r05_rig_tests.log

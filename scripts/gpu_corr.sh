#!/bin/bash
# gpurun payload: parity of the matrix-core pmn_warp_correlate against the streaming kernel + same-box A/B on a real forward.
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest tests/test_corr_mfma.py" | tee gpurun_out/corr.log
timeout 900 python -m pytest tests/test_corr_mfma.py -q -x 2>&1 | tail -40 | tee -a gpurun_out/corr.log
rc=${PIPESTATUS[0]}
if ! grep -q " passed" gpurun_out/corr.log || grep -q "failed" gpurun_out/corr.log; then
  echo "== retry with the ds_bpermute transpose (checks the permlane form)" | tee -a gpurun_out/corr.log
  (cd patchmatchnet_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DPMN_TRANSPOSE_SHFL -c corr_mfma.hip -o corr_mfma.o && make -s libpmn_hip.so) 2>&1 | tail -5 | tee -a gpurun_out/corr.log
  timeout 900 python -m pytest tests/test_corr_mfma.py -q -x 2>&1 | tail -40 | tee -a gpurun_out/corr.log
fi
echo "== golden / oracle parity with the matrix-core form as the default" | tee -a gpurun_out/corr.log
timeout 1200 python -m pytest tests/test_hip_parity.py -q -k "kernels_against_golden or cascade_with_reference or fullsize_stage or evaluation_forward" 2>&1 | tail -15 | tee -a gpurun_out/corr.log
echo "== A/B on a real forward" | tee -a gpurun_out/corr.log
timeout 900 python scripts/corr_ab.py --json gpurun_out/corr_ab.json 2>&1 | tail -20 | tee -a gpurun_out/corr.log

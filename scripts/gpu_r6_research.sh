export TMPDIR=/tmp
PMN_EXPERIMENTAL=1 timeout 900 python -m pytest tests/test_corr_mfma.py tests/test_gather_win.py tests/test_hip_parity.py -q -m gpu -k "corr or gather or windowed or winograd or mfma or research" 2>&1 | tail -8
python scripts/call_ab.py --ops conv2d_f16s --libs patchmatchnet_amd/csrc/libpmn_hip.so,build/ldsab/libpmn_hip_nostore.so,build/ldsab/libpmn_hip_l2input.so --rounds 2 --reps 60 2>&1 | grep -a "^call" | tee gpurun_out/r06_conv_fusion_bound.log

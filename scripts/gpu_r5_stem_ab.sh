#!/bin/bash
# gpurun payload (round 5): stem tile height A/B (16 vs 32 rows), interleaved twice, bits compared by digest; then the conv parity tests
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in 1 2; do
  python scripts/stem_ab.py --lib build/pw/libpmn_hip_stem16.so 2>/dev/null | tail -1
  python scripts/stem_ab.py 2>/dev/null | tail -1
done | tee gpurun_out/r05_stem_ab.log
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_fullsize_parity.py -q -m gpu -k "stem or feature or f16 or conv" 2>&1 | tail -4 | tee -a gpurun_out/r05_stem_ab.log

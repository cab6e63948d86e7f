#!/bin/bash
# round-2 GPU pass B: ablations of the windowed kernels + PMC counters of both families
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/warp_tune.py --reps 8 --configs stream win12 win12@1 win12@2 win12@3 win12@4 win12@6 win12@7 win12@8 win12@15 win12@31 win12p6@1 > gpurun_out/warp_ablate.log 2>&1; echo "ablate exit $?" >> gpurun_out/warp_ablate.log
bash scripts/gpu_pmc_win.sh > gpurun_out/pmc_win.log 2>&1
grep -E "launch [0-9]|total" gpurun_out/warp_ablate.log | cut -c1-150

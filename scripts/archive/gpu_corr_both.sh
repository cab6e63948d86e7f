#!/bin/bash
# gpurun payload: variants A/B (timing) then PMC counters of the product library's matrix-core kernels
VARIANTS="${VARIANTS:-A}" bash scripts/gpu_corr_variants.sh
IMPLS=mfma bash scripts/gpu_corr_pmc.sh 2>&1 | grep -E "corr_mfma_kernel" | cut -c1-420 | tee -a gpurun_out/corr_variants.log

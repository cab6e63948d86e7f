#!/bin/bash
# gpurun payload: the free-running end-to-end test against the reference's own output + the chained cascades, once with the product
# library and once with the -DPMN_IEEE_DIV attribution build; the reports land in gpurun_out/ieee/{product,ieee}.jsonl
export TMPDIR=/tmp
E=gpurun_out/ieee; rm -rf $E; mkdir -p $E
LIB=patchmatchnet_amd/csrc/libpmn_hip.so
cp $LIB /tmp/libpmn_orig.so
for v in ${VARIANTS:-product ieee}; do
  [ $v != product ] && cp scripts/microbench/variants/libpmn_$v.so $LIB
  rm -f gpurun_out/parity_report.jsonl
  timeout 1500 python -m pytest tests/test_fullsize_parity.py -q -k "${TESTS:-cfg2_scene_end_to_end or chained_cascade}" 2>&1 | tail -6 | tee $E/$v.log
  mv gpurun_out/parity_report.jsonl $E/$v.jsonl
  timeout 300 python scripts/kernel_bench.py --reps 10 2>&1 | grep -E "warp_correlate|aggregate" | tee -a $E/$v.log
done
cp /tmp/libpmn_orig.so $LIB

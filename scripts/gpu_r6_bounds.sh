#!/bin/bash
# Upper bounds (ablation builds, wrong results): scripts/experiments/bounds_r6/build_bounds.py -> gpurun_out/r06_bounds.log
mkdir -p gpurun_out
export TMPDIR=/tmp
[ -f build/ldsab/libpmn_hip_stem_skeleton.so ] || python scripts/experiments/bounds_r6/build_bounds.py > /dev/null 2>&1  # (scratch builds are not kept in the tree)
L=gpurun_out/r06_bounds.log
: > $L
P=patchmatchnet_amd/csrc/libpmn_hip.so
B=build/ldsab/libpmn_hip
echo "## known-weights launches: product | no barrier before the hand-over (us per call, min of 3 processes x 40 launches)" >> $L
timeout 600 python scripts/call_ab.py --ops warp_correlate --libs $P,${B}_views_nobarrier.so 2>&1 | grep -a "^call" >> $L
echo "## stem, one 1600x1200 view per launch (six calls): product | no conv0 FMAs | no MFMAs | no stores | skeleton" >> $L
timeout 600 python scripts/call_ab.py --ops stem_f16s --libs $P,${B}_stem_noconv0.so,${B}_stem_nomfma.so,${B}_stem_nostore.so,${B}_stem_skeleton.so 2>&1 | grep -a "^call" >> $L
cat $L

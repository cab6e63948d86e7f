#!/bin/bash
# NOTE (round 6): the -DPMN_IEEE_DIV / -DPMN_ATEN_GPU_DIV / -DPMN_POSE_FMA blocks no longer live in the product sources: apply
# scripts/experiments/source_switches/attribution_and_probe_switches.patch to a scratch copy of patchmatchnet_amd/csrc first.
# Attribution build of the PRODUCT library (profiles/r04_ieee_attribution.md): gather_corr.hip and aggregate.hip with -DPMN_IEEE_DIV
# (the reference's own chain of IEEE divisions / expf instead of v_rcp + Newton / v_exp) -> scripts/microbench/variants/libpmn_ieee.so
set -e
cd "$(dirname "$0")/../patchmatchnet_amd/csrc"
mkdir -p ../../scripts/microbench/variants
make -s -j8
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -DPMN_IEEE_DIV"
/opt/rocm/bin/hipcc $F -c gather_corr.hip -o /tmp/gather_corr_ieee.o
/opt/rocm/bin/hipcc $F -c aggregate.hip -o /tmp/aggregate_ieee.o
objs=$(ls *.o | grep -v '\.x\.o' | grep -v '^gather_corr\.o$' | grep -v '^aggregate\.o$' | grep -v -- '-hip-\|-host-')
V=../../scripts/microbench/variants
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libpmn_ieee.so $objs /tmp/gather_corr_ieee.o /tmp/aggregate_ieee.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libpmn_ieee_proj.so $objs /tmp/gather_corr_ieee.o aggregate.o   # projection only
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libpmn_ieee_agg.so $objs gather_corr.o /tmp/aggregate_ieee.o     # depth weights only
echo built libpmn_ieee.so libpmn_ieee_proj.so libpmn_ieee_agg.so

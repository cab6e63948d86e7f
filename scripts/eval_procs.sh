#!/bin/bash
# gpurun payload: eval.py on 24 generated scans, every run in a FRESH process (what a user runs), several times; the figure is the
# "depth stage" line eval.py prints itself (decode -> ... -> map files, model load excluded)
export TMPDIR=/tmp
B=/dev/shm/pmn_eval_procs
rm -rf $B; mkdir -p $B gpurun_out
python - <<'PY'
import os, sys, shutil
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import synth
data = "/dev/shm/pmn_eval_procs/data"
for s in range(6):
    synth.write_scene_scan(data, "scan%d" % (s + 1), 49, 1200, 1600, n_src=10, seed=s, device="cuda")
for s in range(6, 24):
    shutil.copytree(os.path.join(data, "scan%d" % (s % 6 + 1)), os.path.join(data, "scan%d" % (s + 1)))
open(os.path.join(data, "list.txt"), "w").write("".join("scan%d\n" % (s + 1) for s in range(24)))
PY
OUTPUT_TYPE=${OUTPUT_TYPE:-depth}   # both: inference + consistency filtering + fusion (masks, fused.ply), the reference's default
RUNS=${RUNS:-1 2 3 4 5 6}
for i in $RUNS; do
  extra=""; [ $i -ge 5 ] && extra="$EXTRA_ENV"
  rm -rf $B/out
  env $extra python eval.py --input_folder $B/data --output_folder $B/out --checkpoint_path tests/golden/params_000007.npz --scan_list $B/data/list.txt --num_views 5 --file_format .pfm --output_type $OUTPUT_TYPE --geo_mask_thres 3 $EVAL_EXTRA 2>&1 | grep -a -E "depth stage|both stages|fusion of the last|fusion stage.*scan(7|8) |bound to|Error|error" | sed "s/^/run $i $extra: /"
done | tee gpurun_out/eval_procs_$OUTPUT_TYPE.log
rm -rf $B

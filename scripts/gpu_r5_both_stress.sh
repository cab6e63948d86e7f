#!/bin/bash
# gpurun payload (round 5): eval.py --output_type both N times in fresh processes with the FULL tail of every run kept (hunting a
# silent death seen once in the evidence pass), then the research-build tests on the rebuilt experimental library
export TMPDIR=/tmp
B=/dev/shm/pmn_eval_procs
rm -rf $B; mkdir -p $B gpurun_out/r05_stress
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
for i in 1 2 3 4 5 6; do
  rm -rf $B/out
  python eval.py --input_folder $B/data --output_folder $B/out --checkpoint_path tests/golden/params_000007.npz --scan_list $B/data/list.txt --num_views 5 --file_format .pfm --output_type both --geo_mask_thres 3 $EVAL_EXTRA > gpurun_out/r05_stress/run$i.log 2>&1
  echo "run $i rc=$? $(grep -a -E 'both stages' gpurun_out/r05_stress/run$i.log | cut -c1-110)"
  df -h /dev/shm | tail -1; free -g | head -2 | tail -1
  grep -a -v "^Iter \|^processing " gpurun_out/r05_stress/run$i.log | tail -25 > gpurun_out/r05_stress/run$i.tail
  if ! grep -a -q "both stages" gpurun_out/r05_stress/run$i.log; then grep -a -v "^Iter \|^processing " gpurun_out/r05_stress/run$i.log | grep -a -n -i -B2 -A25 "traceback\|error\|fault\|killed" | head -120 > gpurun_out/r05_stress/run$i.err; dmesg 2>/dev/null | tail -5 >> gpurun_out/r05_stress/run$i.err; fi
  if grep -q -P "\x00" gpurun_out/r05_stress/run$i.log; then grep -a -n -P "\x00" gpurun_out/r05_stress/run$i.log | cat -v | cut -c1-400 | head -5 > gpurun_out/r05_stress/run$i.nul; fi
  rm gpurun_out/r05_stress/run$i.log
done | tee gpurun_out/r05_stress/summary.txt
rm -rf $B

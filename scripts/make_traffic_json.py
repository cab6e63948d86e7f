#!/usr/bin/env python
"""Turns the rocprofv3 --pmc passes of scripts/gpu_pmc_traffic.sh (FETCH_SIZE and WRITE_SIZE, collected in SEPARATE passes with
--kernel-trace only) into profiles/pmc_traffic.json: HBM bytes per launch of every pmn_warp_correlate kernel shape.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, i.e. reports exactly half of the bytes of a wide coalesced read stream -> doubled here.
WRITE_SIZE is taken as reported (uncalibrated per the guide).  hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import collections
import csv
import glob
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import warp_kernel_source_hash  # noqa: E402  (bench.py refuses a traffic file whose hash is not the tree's)

src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "pmc_win")
config = sys.argv[3] if len(sys.argv) > 3 else "1600x1200_N5"  # WxH_N<source views>: the bench configuration the passes ran
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "p*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
            continue
        if "pixelwise_wave_kernel<" in r["Kernel_Name"]:  # round 5: the PixelwiseNet launch of stage 3 (C = 64, G = 8, D <= 64)
            C, G, mode, DT = 64, 8, 1, 64
        else:
            m = re.search(r"gather_corr_kernel<(\d+), (\d+), (\d+), (\d+)", r["Kernel_Name"])
            if not m:
                continue
            C, G, mode, DT = (int(x) for x in m.groups())
        if mode == 2:
            continue
        vals[f"C{C}_D{DT}_{'pixelwise' if mode == 1 else 'vw'}"][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in vals.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        fk = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
        wk = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
        out[k] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "profiles", "pmc_traffic.json")
doc = {}
if os.path.isfile(dst):
    try:
        doc = json.load(open(dst))
    except ValueError:
        doc = {}
if doc.get("kernel_source_sha256") != warp_kernel_source_hash() or "configs" not in doc:
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/gpu_pmc_traffic.sh) over bench.py --eager: the "
                     "pmn_warp_correlate launches of real forwards on the bench's samples, averaged per kernel shape, one entry per "
                     "bench configuration (WxH_N<source views>)",
           "kernel_source_sha256": warp_kernel_source_hash(),
           "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 FETCH_SIZE under-reports wide reads by 2x)",
           "configs": {}}
doc["configs"][config] = {"kernels": out}
json.dump(doc, open(dst, "w"), indent=1)
print(config, json.dumps(out, indent=1))

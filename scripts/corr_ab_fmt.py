#!/usr/bin/env python
"""Compact one-line-per-launch view of corr_ab.py's JSON rows (stdin -> stdout)."""
import json
import sys

for line in sys.stdin:
    r = json.loads(line)
    if "summary" in r:
        print("   TOTAL " + "  ".join(f"{k}: {v['ms_per_depth_map'] * 1e3:7.1f} us ({v['frac_of_8TBs']:.4f} of 8 TB/s)" for k, v in r["summary"].items()))
        continue
    parts = [f"{r['shape']:32s}"]
    for k in ("stream", "mfma"):
        if k in r:
            parts.append(f"{k} {r[k]['median_us']:7.1f} us")
    for k in ("max_abs_diff_sim", "max_abs_diff_cost", "max_abs_diff_vw"):
        if k in r:
            parts.append(f"{k[13:]} {r[k]:.1e}")
    print("   " + "  ".join(parts))

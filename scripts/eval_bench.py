#!/usr/bin/env python
"""eval.py END TO END with the JPEG decode in: a generated DTU-layout scan of the photo-consistent
scene (tests/synth.write_scene_scan: 49 views, 1600x1200, JPEG quality 95, on tmpfs), eval.py --num_views 5 --output_type depth
(decode -> upload -> FeatureNet once per view -> cascade -> refinement -> download -> PFM files), feature cache on, swept over the
number of decode THREADS of the streaming encode-once schedule (--stream_views 1: views decoded once, in first-use order, by a
thread pool, overlapped with the forwards), against round 2's two passes per scan over DataLoader worker processes and against the
reference's schedule (no cache: every sample decodes and encodes its six images).  Prints depth-maps/s per configuration = samples / wall time of eval.main()
(model load and graph capture included in the first, excluded by a warm-up run before the sweep).

    python scripts/eval_bench.py [n_scans=2] [n_views=49]
"""
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
import eval as pm_eval  # noqa: E402

pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n_scans = int(pos[0]) if len(pos) > 0 else 6
n_views = int(pos[1]) if len(pos) > 1 else 49
base = "/dev/shm/pmn_eval_bench" if os.path.isdir("/dev/shm") else "/tmp/pmn_eval_bench"
shutil.rmtree(base, ignore_errors=True)
data = os.path.join(base, "data")
t0 = time.time()
n_render = min(n_scans, int(os.environ.get("PMN_EVAL_BENCH_RENDER", "6")))  # distinct scenes; the rest are copies (steady-state runs)
for s in range(n_render):
    synth.write_scene_scan(data, "scan%d" % (s + 1), n_views, 1200, 1600, n_src=10, seed=s, device="cuda")
for s in range(n_render, n_scans):
    shutil.copytree(os.path.join(data, "scan%d" % (s % n_render + 1)), os.path.join(data, "scan%d" % (s + 1)))
with open(os.path.join(data, "list.txt"), "w") as f:
    f.write("".join("scan%d\n" % (s + 1) for s in range(n_scans)))
print("generated %d scans x %d views in %.1f s, %.1f MB of JPEG" % (
    n_scans, n_views, time.time() - t0,
    sum(os.path.getsize(os.path.join(r, n)) for r, _, fs in os.walk(data) for n in fs) / 1e6), flush=True)
ckpt = os.path.join(ROOT, "tests", "golden", "params_000007.npz")
cores = os.cpu_count() or 8


def run(tag, extra, scan_list=None):
    out = os.path.join(base, "out_" + tag)
    shutil.rmtree(out, ignore_errors=True)
    argv = ["--input_folder", data, "--output_folder", out, "--checkpoint_path", ckpt, "--scan_list",
            scan_list or os.path.join(data, "list.txt"), "--num_views", "5", "--file_format", ".pfm"] + \
        ([] if "--output_type" in extra else ["--output_type", "depth"]) + extra
    t = time.time()
    pm_eval.main(argv)
    dt = time.time() - t
    n = n_scans * n_views if scan_list is None else n_views
    res = {"config": tag, "samples": n, "seconds": round(dt, 3), "depth_maps_per_s": round(n / dt, 1)}
    print("RESULT " + json.dumps(res), flush=True)
    shutil.rmtree(out, ignore_errors=True)
    return res


one = os.path.join(data, "one.txt")
open(one, "w").write("scan1\n")
run("warmup", ["--decode_threads", "8"], scan_list=one)  # library load, weight packing, graph capture, page cache
results = []
if "--steady" in sys.argv:  # the steady-state figure: the default flags, several times over all scans (spread = max/min - 1)
    for r in range(4):
        results.append(run("default_flags_run%d" % (r + 1), []))
    vals = [r["depth_maps_per_s"] for r in results]
    print("STEADY " + json.dumps({"runs": vals, "median": sorted(vals)[len(vals) // 2], "spread": round(max(vals) / min(vals) - 1, 3),
                                  "samples_per_run": results[0]["samples"], "seconds_per_run": [r["seconds"] for r in results]}), flush=True)
    if os.environ.get("PMN_EVAL_BENCH_QUICK", "") != "1":
        for extra in (["--in_flight", "3"], ["--in_flight", "1"], ["--writer_threads", "8"], ["--writer_threads", "16"]):
            results.append(run("default_" + "_".join(extra).replace("--", ""), extra))
    if os.environ.get("PMN_EVAL_BENCH_SWEEP", "") == "1":
        for th in (8, 12, 16, 24):
            for r in range(3):
                results.append(run("decode_threads%d_run%d" % (th, r + 1), ["--decode_threads", str(th)]))
    # the same pipeline with the map files discarded by the writer threads (PMN_EVAL_DISCARD_MAPS=1): download included, file system out
    pm_eval._DISCARD_MAPS = True
    for r in range(4):
        results.append(run("discard_maps_run%d" % (r + 1), []))
    pm_eval._DISCARD_MAPS = False
else:
    for threads in (4, 8, 16):
        results.append(run("decode_threads%d" % threads, ["--decode_threads", str(threads)]))
    results.append(run("default_flags", []))
if "--both" in sys.argv:  # inference + consistency filtering + fusion (masks, fused.ply) of every scan
    results.append(run("output_type_both", ["--output_type", "both", "--geo_mask_thres", "3"]))
    results.append(run("output_type_both_1thread", ["--output_type", "both", "--geo_mask_thres", "3", "--decode_threads", "1"]))
if "--all" in sys.argv:
    results.append(run("two_pass_workers8", ["--num_workers", "8", "--stream_views", "0"]))       # round 2's schedule (DataLoader processes)
    results.append(run("nocache_workers8", ["--num_workers", "8", "--feature_cache", "0"], scan_list=one))  # the reference's schedule
# where the launch thread's time goes (cProfile of one more default run)
import cProfile, pstats, io
pr = cProfile.Profile()
pr.enable()
run("default_flags_profiled", [])
pr.disable()
buf = io.StringIO()
pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
print(buf.getvalue(), flush=True)
best = max(results, key=lambda r: r["depth_maps_per_s"])
print("BEST " + json.dumps(best), flush=True)
shutil.rmtree(base, ignore_errors=True)

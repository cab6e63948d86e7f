#!/usr/bin/env python
"""bench.py -- depth-maps/sec of PatchmatchNet inference on MI355X (BASELINE.json metric), one JSON line on rank 0.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1 without a launcher: spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one full ``PatchmatchNet.forward`` (FeatureNet, the learned-PatchMatch hot path, refinement, confidence -- all in
the HIP kernels of patchmatchnet_amd/csrc) for ONE reference view with 5 source views at 1600x1200, iterations (1,2,2) -- BASELINE.json configs[1].
Inputs (images, cameras) are resident in HBM before the timed region.  Multi-GPU: reference views shard across ranks
with no data-path collective (weak scaling: every rank does K steps); one RCCL all-gather of the per-rank depth /
confidence maps closes the timed region, as the per-scan gather before fusion does in eval.

Timed region (``value``): every forward is issued as ONE launch-plan replay (patchmatchnet_amd/graph.py: PlannedForward; the forward's
~55 launches recorded once and replayed from C with plain hipLaunchKernel calls, include/pmn_hip.h pmn_plan_*), --in-flight S replay
slots on their own HIP streams, on the runtime's default hardware queues, so that forwards of different samples overlap on the device.
``--launch graph`` replays HIP graphs instead (rounds 2-5's form; same rate).  ``outputs_verified`` on the line: --verify-steps further
steps in exactly the timed mode, compared BIT FOR BIT with the same steps launched eagerly one at a time; a mismatch makes the
process exit non-zero.  (Rounds 2-4 overlapped forwards whose outputs were NOT the eager forward's, round 5 found that out and fell
back to one hardware queue; round 6 found the cause -- a packed-fp32 instruction form that MI355X computes wrongly beside kernels that
issue fp16 MFMAs, DESIGN_LESSONS.md lesson 46 -- and removed it from the library: overlap is back, verified.)
``--eager`` restores the round-1 mode (one stream, kernels launched from Python); the line always carries that figure too
(``single_stream_eager`` = one sample's latency).

Extra objects on the line: ``roofline`` for the dominant kernel (pmn_warp_correlate; HIP events on the launch stream around
its launches in an eager single-stream pass of the same run -- launches inside a replayed graph cannot be bracketed and
overlapped kernels have no duration of their own; algorithmic bytes per SURVEY.md 8(d); next to the contract's HBM fraction the
line carries the bound that actually holds, the vector-L1 / texture-addresser path: ``tap_bytes_per_step``, ``l1_achieved``,
``l1_frac``) and, at N=1, ``cpu_baseline`` = the WHOLE forward on the host cores in the metric's own unit (depth-maps/s):
FeatureNet and Refinement as the plain torch modules on the CPU backend, the cascade and the confidence epilogue through the CPU
oracle (oracle/, the checker -- never the thing shipped), at 8 / 32 / 64 threads.  ``steady_state`` repeats the timed region's
loop for ~2 s (same mode) so that a multi-second figure at settled clocks is on the line beside the K-step ``value``.

Inputs: the photo-consistent scene of tests/synth.render_scene (an analytic surface textured procedurally and rendered into every
camera; one texture seed per sample) -- the scene tests/golden/cfg2_scene.npz pins against the reference.  ``--scene rolled`` =
rounds 1-2's images (one noise image rolled per view, unrelated to the cameras).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)

DEFAULT_KW = dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                  patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16],
                  evaluate_neighbors=[9, 9, 9])


def warp_kernel_source_hash():
    """sha256 over the sources of pmn_warp_correlate's kernels: profiles/pmc_traffic.json carries the hash of the tree it was
    measured on, and ``roofline.traffic`` is only reported when that is THIS tree (a stale PMC file would be a made-up number)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("gather_corr.hip", "gather_common.hpp", "pmn_common.hpp"):
        with open(os.path.join(ROOT, "patchmatchnet_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


REFERENCE_CPU_MEASURED = {
    # the only timing of the REAL reference that exists: unmodified reference models/ imported in the
    # authoring container, params_000007.ckpt, torch 2.10 CPU backend; it cannot be re-measured on the GPU box (nothing there
    # may read /root/reference and its sources must not be copied into this repository)
    "value": 0.0855, "unit": "depth-maps/s (whole PatchmatchNet.forward)", "cores": 8, "kind": "reference",
    "sample": "1600x1200, N=5, iters (1,2,2): 11.7 s warm per depth map, Intel Xeon @ 2.10 GHz, 8 threads; measured in the "
              "authoring container, NOT on the bench box",
}


def load_weights(model):
    """Reference checkpoint tensors from the committed fixture; seeded random init if the fixture is unavailable."""
    path = os.path.join(ROOT, "tests", "golden", "params_000007.npz")
    if os.path.isfile(path):
        with np.load(path) as z:
            model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files}, strict=True)
        return "params_000007 (reference checkpoint tensors)"
    torch.manual_seed(0)
    for m in model.modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.Conv3d, torch.nn.ConvTranspose2d)):
            torch.nn.init.kaiming_normal_(m.weight)
    return "random init"


def make_samples(n_samples, n_views, H, W, device, rank, scene="surface"):
    """``n_samples`` distinct synthetic samples resident on ``device``.  scene="surface": tests/synth.render_scene (rendered on the
    device in float64, quantised to k/255 like decoded image files); "rolled": the images of rounds 1-2."""
    import synth
    intr, extr = synth.synthetic_cameras(n_views, H, W)
    samples = []
    for s in range(n_samples):
        if scene == "surface":
            imgs, _, _, _ = synth.render_scene(n_views, H, W, seed=1000 * rank + s, device=device)
            imgs = [im.to(device).contiguous() for im in imgs]
        else:
            g = torch.Generator().manual_seed(1000 * rank + s)
            base = torch.rand(1, 3, H, W, generator=g)
            base = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(base, (2, 2, 2, 2), mode="reflect"), 5, 1)
            imgs = [(torch.roll(base, shifts=4 * v, dims=3) + 0.02 * torch.rand(1, 3, H, W, generator=g)).clamp(0, 1)
                    .contiguous().to(device) for v in range(n_views)]
        samples.append(dict(images=imgs, intrinsics=torch.from_numpy(intr).to(device),
                            extrinsics=torch.from_numpy(extr).to(device),
                            depth_min=torch.tensor([425.0], device=device),
                            depth_max=torch.tensor([935.0], device=device)))
    return samples


def reference_archive():
    """oracle/_ref/patchmatchnet_reference.pt: the REFERENCE's own PatchmatchNet (unmodified models/net.py + params_000007.ckpt) as a
    TorchScript archive, written by oracle/make_ref.py where the reference checkout exists (build()); test infrastructure, loaded
    only by the baseline legs below.  None when it was never built."""
    path = os.path.join(ROOT, "oracle", "_ref", "patchmatchnet_reference.pt")
    return path if os.path.isfile(path) else None


def reference_inputs(H, W, n_src, seed=0):
    """One sample of the bench scene in the reference's own argument form (images: list of [1,3,H,W]; cameras [1,V,...])."""
    import synth
    imgs, intr, extr, _ = synth.render_scene(n_src + 1, H, W, seed=seed)
    return ([im.contiguous() for im in imgs], torch.from_numpy(intr), torch.from_numpy(extr),
            torch.tensor([425.0]), torch.tensor([935.0]))


def reference_cpu_baseline(H, W, n_src, timed=3, warmup=2):
    """The REAL reference on the host cores (SURVEY 8(d) last row; reference eval.py:44-70): `warmup` untimed (the
    TorchScript executor profiles on its first call and specialises on its second) + `timed` forwards of the archive, median.  Threads: min(64, cores) -- on the 256-thread bench box torch's CPU backend is
    slower with every hardware thread than with 32-64 (measured on the port: 3.8 s at 8, 5.2 s at 64, 84.6 s at 256)."""
    path = reference_archive()
    if path is None:
        return None
    cores = os.cpu_count() or 1
    threads = min(64, cores)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        model = torch.jit.load(path, map_location="cpu").eval()
        args = reference_inputs(H, W, n_src)
        times = []
        with torch.no_grad():
            for i in range(warmup + timed):
                torch.manual_seed(1234)
                t0 = time.perf_counter()
                model(*args)
                times.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(prev)
    med = float(np.median(times[warmup:]))
    return {"value": round(1.0 / med, 4), "unit": "depth-maps/s (whole PatchmatchNet.forward; same unit as `value`)",
            "cores": threads, "kind": "reference",
            "sample": f"the reference's own TorchScript archive (oracle/make_ref.py: unmodified models/net.py + params_000007.ckpt), "
                      f"torch {torch.__version__} CPU backend, {threads} threads of {cores}; {warmup} warm-up ("
                      + ", ".join(f"{t:.2f}" for t in times[:warmup]) + f" s) + {timed} timed "
                      f"forwards at {W}x{H}, N={n_src}, iters (1,2,2) on the bench scene (seed 0): "
                      + ", ".join(f"{t:.2f}" for t in times[warmup:]) + f" s, median {med:.2f} s",
            "seconds": [round(t, 3) for t in times]}


def reference_rocm_baseline(H, W, n_src, device, ours, warmup=3, timed=10, budget_s=240.0):
    """The reference on THIS GPU through PyTorch-ROCm -- the denominator of the north star's '>= 4x the reference eval.py' (BASELINE.md
    3.2; reference eval.py:37-41, 57-70): the same archive moved to the device, `warmup` untimed forwards (MIOpen's first-use
    searches), `timed` forwards each bracketed by torch.cuda.synchronize(), median.  `ours` = this engine's model: its depth on the same
    sample and the same seeded stage-3 draw is compared with the reference's (free-running, end to end)."""
    path = reference_archive()
    if path is None:
        return None
    out = {"kind": "reference on PyTorch-ROCm (TorchScript archive of the unmodified reference, oracle/make_ref.py)"}
    try:
        model = torch.jit.load(path, map_location=device).eval()
        imgs, intr, extr, dmin, dmax = reference_inputs(H, W, n_src)
        args = ([im.to(device) for im in imgs], intr.to(device), extr.to(device), dmin.to(device), dmax.to(device))
        times, ref_depth = [], None
        t_begin = time.perf_counter()
        with torch.no_grad():
            for i in range(warmup + timed):
                torch.manual_seed(1234)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                depth, conf, _ = model(*args)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t0)
                ref_depth = depth
                if time.perf_counter() - t_begin > budget_s and i + 1 >= warmup + 3:
                    break
            torch.manual_seed(1234)
            mine, _, _ = ours([im for im in args[0]], args[1].clone(), args[2], args[3], args[4])
            # ... and with the reference's OWN FeatureNet outputs on this GPU (MIOpen) handed to this engine's cascade: what is left is
            # the cascade's share of the difference (tests/test_fullsize_parity.py::test_cfg2_scene_against_the_reference_on_rocm gates it)
            ref_feats = [{s_: f_.contiguous() for s_, f_ in model.feature(im).items()} for im in args[0]]
            torch.manual_seed(1234)
            forced, _, _ = ours([im for im in args[0]], args[1].clone(), args[2], args[3], args[4], features=ref_feats)
            torch.cuda.synchronize()
        tt = times[warmup:]
        med = float(np.median(tt))
        rel = ((mine - ref_depth).abs() / ref_depth.abs()).flatten().double()
        relf = ((forced - ref_depth).abs() / ref_depth.abs()).flatten().double()
        out.update({"value": round(1.0 / med, 3), "unit": "depth-maps/s", "ms_per_forward": round(med * 1e3, 2), "samples": len(tt),
                    "warmup_seconds": [round(t, 2) for t in times[:warmup]],
                    "min_ms": round(min(tt) * 1e3, 2), "max_ms": round(max(tt) * 1e3, 2),
                    "parity_vs_this_engine": {
                        "what": "final depth of this engine vs the reference's on the same sample and the same seeded stage-3 draw, "
                                "both free-running on this GPU (relative difference)",
                        "p50": float(rel.median()), "p99": float(torch.quantile(rel[:: max(1, rel.numel() // 1000000)], 0.99)),
                        "frac_gt_1e-3": float((rel > 1e-3).double().mean()), "max": float(rel.max())},
                    "parity_vs_this_engine_on_the_reference_features": {
                        "what": "the same comparison with the reference's own FeatureNet outputs on this GPU (MIOpen) handed to this engine "
                                "(features=): the cascade + refinement's share of the difference; the rest of `parity_vs_this_engine` is "
                                "MIOpen's FeatureNet vs this engine's (profiles/r05_rocm_parity.md)",
                        "p50": float(relf.median()), "p99": float(torch.quantile(relf[:: max(1, relf.numel() // 1000000)], 0.99)),
                        "frac_gt_1e-3": float((relf > 1e-3).double().mean()), "max": float(relf.max())}})
        del model
        torch.cuda.empty_cache()
    except Exception as e:  # the baseline must never take the bench line down
        out["error"] = f"{type(e).__name__}: {str(e).splitlines()[0][:200] if str(e) else ''}"
    return out


def cpu_baseline(H, W, n_src, model_kw, thread_counts=(8, 32, 64)):
    """The whole forward on the host cores, in the metric's unit: FeatureNet.forward / Refinement.forward = the plain torch modules
    of patchmatchnet_amd/net.py on the CPU backend (reference models/net.py:9-122), cascade + confidence epilogue = the CPU oracle
    (a port of models/patchmatch.py, models/module.py:130-196, models/net.py:221-299).  One forward per thread count."""
    import synth
    from oracle import oracle as O
    from patchmatchnet_amd.net import FeatureNet, Refinement
    with np.load(os.path.join(ROOT, "tests", "golden", "params_000007.npz")) as z:
        params = {k: z[k] for k in z.files}
    feature, refine = FeatureNet().eval(), Refinement().eval()
    feature.load_state_dict({k[len("feature."):]: torch.from_numpy(v) for k, v in params.items() if k.startswith("feature.")})
    refine.load_state_dict({k[len("upsample_net."):]: torch.from_numpy(v) for k, v in params.items() if k.startswith("upsample_net.")})
    imgs, intr, extr, _ = synth.render_scene(n_src + 1, H, W, seed=0)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(1234)).numpy()
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    cores = os.cpu_count() or 1
    configs = O.default_stage_configs(model_kw["patchmatch_interval_scale"], model_kw["propagation_range"],
                                      model_kw["patchmatch_iteration"], model_kw["patchmatch_num_sample"],
                                      model_kw["propagate_neighbors"], model_kw["evaluate_neighbors"])

    def forward(threads):
        torch.set_num_threads(threads)
        O.set_num_threads(threads)
        t0 = time.perf_counter()
        with torch.no_grad():
            feats = [{k: v.numpy() for k, v in feature(im).items()} for im in imgs]
            t1 = time.perf_counter()
            d1, score, _ = O.cascade(params, feats, intr, extr, dmin, dmax, noise, configs=configs)
            O.confidence(score, (H, W))
            t2 = time.perf_counter()
            refine(imgs[0], torch.from_numpy(np.ascontiguousarray(d1)), torch.from_numpy(dmin), torch.from_numpy(dmax))
        t3 = time.perf_counter()
        return t3 - t0, (t1 - t0, t2 - t1, t3 - t2)

    prev = torch.get_num_threads()
    runs = {}
    # (all 256 hardware threads of the bench box: 84.6 s per forward against 4.3 s on 8 -- oversubscribed OpenMP + torch pools; the
    #  sweep stops at 64 so that the default bench run stays within minutes; with the reference archive timed beside it -- round 4 --
    #  the port keeps one thread count)
    for th in sorted({min(t, cores) for t in thread_counts}):
        dt, parts = forward(th)
        runs[th] = {"seconds": round(dt, 2), "depth_maps_per_s": round(1.0 / dt, 4),
                    "featurenet_cascade_refinement_s": [round(x, 2) for x in parts]}
    torch.set_num_threads(prev)
    best = min(runs, key=lambda th: runs[th]["seconds"])
    return {"value": runs[best]["depth_maps_per_s"], "unit": "depth-maps/s (whole forward: FeatureNet + cascade + confidence + "
            "Refinement; same unit as `value`)", "cores": best, "kind": "port",
            "sample": f"1 forward per thread count at {W}x{H}, N={n_src}, iters (1,2,2), photo-consistent scene seed 0; "
                      f"threads -> seconds: " + ", ".join(f"{th}: {r['seconds']}" for th, r in runs.items()) +
                      f"; host has {cores} hardware threads; FeatureNet / Refinement = torch CPU backend, cascade = C/OpenMP oracle",
            "threads": {str(th): r for th, r in runs.items()},
            "reference_measured_elsewhere": REFERENCE_CPU_MEASURED}


def eval_end_to_end_leg(H, W, n_src, device, scans=6, views=49, timeout_s=300):
    """eval.py --output_type depth in a FRESH process (what a user runs) on ``scans`` generated DTU-layout scans of ``views`` JPEGs at
    the bench's size: decode, upload, per-scan feature cache, replayed forwards, download and map files included, model load excluded
    (the figure eval.py prints itself).  VERDICT r05 item 8: the host half of the pipeline belongs on the driver-visible line."""
    import re
    import shutil
    import subprocess
    import tempfile
    import synth
    base = tempfile.mkdtemp(prefix="pmn_bench_eval_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        data = os.path.join(base, "data")
        t0 = time.perf_counter()
        synth.write_scene_scan(data, "scan1", views, H, W, n_src=10, seed=0, device=device)
        for k in range(1, scans):
            shutil.copytree(os.path.join(data, "scan1"), os.path.join(data, "scan%d" % (k + 1)))
        with open(os.path.join(data, "list.txt"), "w") as f:
            f.write("".join("scan%d\n" % (k + 1) for k in range(scans)))
        t_gen = time.perf_counter() - t0
        cmd = [sys.executable, os.path.join(ROOT, "eval.py"), "--input_folder", data, "--output_folder", os.path.join(base, "out"),
               "--checkpoint_path", os.path.join(ROOT, "tests", "golden", "params_000007.npz"), "--scan_list", os.path.join(data, "list.txt"),
               "--num_views", str(n_src), "--file_format", ".pfm", "--output_type", "depth"]
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, cwd=ROOT)
        m = re.search(r"depth stage: (\d+) samples in ([0-9.]+) s .*?-> ([0-9.]+) depth-maps/s", p.stdout)
        if p.returncode != 0 or not m:
            return {"error": (p.stderr or p.stdout)[-300:]}
        return {"value": float(m.group(3)), "unit": "depth-maps/s", "samples": int(m.group(1)), "seconds": float(m.group(2)),
                "what": f"eval.py --output_type depth --num_views {n_src} in a fresh process on {scans} generated scans of {views} {W}x{H} JPEGs "
                        "(/dev/shm): decode, upload, per-scan feature cache (every view encoded once), replayed forwards, download and "
                        ".pfm map files included; model load excluded", "scan_generation_seconds": round(t_gen, 1)}
    except Exception as e:  # noqa: BLE001 -- an extra leg must never take the line down
        return {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    finally:
        shutil.rmtree(base, ignore_errors=True)


def self_launch(n_ranks):
    """``python bench.py --gpus N`` without a launcher: start the N ranks (one process per GPU, free rendezvous port on 127.0.0.1),
    rank 0 inherits stdout and prints the single JSON line; the exit code is the worst rank's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # this pool's driver: dmabuf IPC only (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(p.poll() is None for p in procs):
            time.sleep(0.2)
            failed = [p for p in procs if p.poll() not in (None, 0)]
            if failed:  # one rank died: the others would wait in a collective forever
                rc = failed[0].returncode
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    raise SystemExit(rc or max((p.returncode or 0) for p in procs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--samples", type=int, default=12, help="distinct synthetic samples cycled through (SURVEY 8(d): >= 10)")
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--views", type=int, default=5, help="number of SOURCE views (reference eval.py --num_views)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="independent samples in flight per GPU: one HIP stream + one replay slot each (1 = one stream)")
    ap.add_argument("--launch", choices=("plan", "graph"), default="plan",
                    help="plan = the forward recorded as a launch plan and replayed from C with plain launches (default); graph = "
                         "HIP-graph replay (rounds 2-5)")
    ap.add_argument("--copy-inputs", action="store_true",
                    help="graph replay reads copies of all six images in the slot's static buffers (rounds 2-4) instead of the samples in place")
    ap.add_argument("--eager", action="store_true",
                    help="issue every kernel from Python on one stream (the round-1 mode) instead of replaying HIP graphs")
    ap.add_argument("--roofline-steps", type=int, default=24,
                    help="eager single-stream steps after the timed region that carry the HIP events of the roofline figure")
    ap.add_argument("--scene", choices=("surface", "rolled"), default="surface",
                    help="surface = photo-consistent rendered scene (tests/synth.render_scene); rolled = rounds 1-2's images")
    ap.add_argument("--settle-seconds", type=float, default=1.0,
                    help="untimed replays after the W warm-up steps until the clocks / caches have settled (the timed region is still "
                         "exactly K steps); 0 = only the W warm-up steps")
    ap.add_argument("--verify-steps", type=int, default=96,
                    help="steps of the timed mode whose outputs are compared bit for bit with eager forwards under the same seeds (0 = skip)")
    ap.add_argument("--steady-seconds", type=float, default=2.0,
                    help="length of the extra steady-state pass reported as `steady_state` (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)  # does not return
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device"
    if os.environ.get("PMN_DIST_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()  # several ranks on one GPU (control-flow test only)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # one process per GPU on a two-socket host: launch thread + pinned buffers on the GPU's own NUMA node (eval.py does the same;
    # measured there as whole runs alternating between 270 and 380 depth-maps/s without it).  The CPU baseline legs get the
    # process's original affinity back (they time the host cores, not the launch thread).
    from patchmatchnet_amd import dist as pdist
    affinity0 = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    numa_note = pdist.bind_to_device_node(device)
    launched = "WORLD_SIZE" in os.environ  # by torch.distributed.run / self_launch: then the process group exists even for one rank
    if launched:                            # (world 1 over RCCL exercises the same init / collectives as world 8: tests/test_bench_gpu.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")  # (the driver and self_launch pass their own)
        # nccl (= RCCL over xGMI) is the product's backend; PMN_DIST_BACKEND=gloo exists so that the multi-rank control flow can be
        # exercised with several ranks on ONE GPU (collectives then stage through the host)
        backend = os.environ.get("PMN_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    on_host = launched and dist.get_backend() != "nccl"

    def reduce_scalar(x, op):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if on_host else device)
        if launched:
            dist.all_reduce(t, op=op)
        return float(t.item())

    import patchmatchnet_amd as P
    from patchmatchnet_amd import ops
    P.lib()  # fail loudly if the HIP library is missing

    model = P.PatchmatchNet(**DEFAULT_KW)
    weights = load_weights(model)
    model = model.to(device).eval()
    H, W, n_src = args.height, args.width, args.views
    samples = make_samples(max(args.samples, 1), n_src + 1, H, W, device, rank, args.scene)

    def step(i):
        s = samples[i % len(samples)]
        return model([im for im in s["images"]], s["intrinsics"].clone(), s["extrinsics"], s["depth_min"],
                     s["depth_max"])

    def barrier():
        if launched:
            dist.barrier()
        torch.cuda.synchronize()

    def close_region(outs):
        """The per-scan gather of the final maps before fusion: the only collective of the path (RCCL over xGMI)."""
        if launched:
            mine = torch.stack([outs[0][0, 0], outs[1][0]], 0).contiguous()
            if on_host:
                mine = mine.cpu()
            gathered = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
            dist.all_gather_into_tensor(gathered, mine)
        barrier()

    # ---- timed region: `value` ------------------------------------------------------------------------------------------
    # Default: S = 3 samples in flight, each on its own HIP stream, each forward one HIP-graph replay (patchmatchnet_amd/graph.py).
    # The kernels of a forward are bound by different units (vector-memory pipe for the gathers, matrix cores for the
    # convolutions, VALU for the stem / aggregation), so two forwards sharing the CUs finish sooner than one after the other;
    # the graph takes the ~3.5 ms of Python launch work per forward off the critical path.  Every step runs the whole forward on
    # its own inputs, read where the samples sit in HBM (graph.GraphedForward(inputs_in_place=True): FeatureNet finds the six images
    # through a device table of addresses rewritten per step; the cameras, depth range and the reference image Refinement reads are
    # copied into the slot's static buffers inside the timed region; --copy-inputs = all six images copied, rounds 2-4's mode).
    S = 1 if args.eager else max(args.in_flight, 1)
    main_stream = torch.cuda.current_stream(device)
    with torch.no_grad():
        launch_note = None
        # ---- roofline pass: HIP events around the pmn_warp_correlate launches ------------------------------------------------
        # (Runs FIRST, before the timed region: after seconds of three overlapped forwards the part runs 5-6 % slower clocks -- the same
        # launches 1.06 ms cold, 1.13 ms behind the steady-state pass -- and the rocprofv3 summary this figure must agree with is taken
        # from a run without that load.)
        # Launches inside a replayed graph cannot be bracketed by events, and kernels of two overlapped forwards do not have a
        # duration of their own; the dominant kernel is therefore timed in R eager single-stream steps of the same process, on
        # the launch stream, every EV-th step (an event pair costs ~10 us of stream time on ROCm -- a blit per record -- 2.5 % of
        # a step when every launch is bracketed).  The wall clock of this pass is the single-stream eager rate.
        EV = 4
        R = max(args.roofline_steps, EV)
        sampled = len(range(0, R, EV))
        for i in range(3):
            step(i)
        barrier()
        ops.enable_kernel_timing()
        t1 = time.perf_counter()
        for i in range(R):
            ops.pause_kernel_timing(i % EV != 0)
            step(i)
        torch.cuda.synchronize()
        eager_elapsed = time.perf_counter() - t1
        recs = ops.disable_kernel_timing()


        region = {}  # "run": callable(steps) -> seconds, the timed loop in the mode that produced `value` (re-used by steady_state)
        extra_warmup = [0]

        def settle(one_step):
            """Untimed steps on top of the W asked for, until --settle-seconds have passed: the first replays after a capture run
            at ramping clocks (the 100-step figure came out 5 % under the 2-second one); the timed region stays exactly K steps."""
            t_end = time.perf_counter() + max(args.settle_seconds, 0.0)
            i = 0
            while time.perf_counter() < t_end:
                for _ in range(8):
                    one_step(i)
                    i += 1
                torch.cuda.synchronize()
            extra_warmup[0] = i

        def timed_eager():
            def run(steps):
                barrier()
                t0 = time.perf_counter()
                outs = None
                for i in range(steps):
                    depth, conf, _ = step(i)
                    outs = (depth, conf)
                close_region(outs)
                return time.perf_counter() - t0

            for i in range(args.warmup):
                step(i)
            settle(step)
            region["run"] = run
            return run(args.steps)

        def timed_in_flight(copy_inputs=args.copy_inputs, publish=True):
            """Returns the elapsed time, or None when some rank could not capture its graphs (decided collectively before the
            timed region, so that every rank then takes the same path)."""
            from patchmatchnet_amd.graph import GraphedForward, PlannedForward
            streams = [torch.cuda.Stream(device) for _ in range(S)]
            Slot = PlannedForward if args.launch == "plan" else GraphedForward
            slots = [Slot(model, inputs_in_place=not copy_inputs) for _ in range(S)]
            region.setdefault("slots", slots)

            def replay(i):
                k, s = i % S, samples[i % len(samples)]
                with torch.cuda.stream(streams[k]):
                    return slots[k]([im for im in s["images"]], s["intrinsics"], s["extrinsics"], s["depth_min"], s["depth_max"])

            for st in streams:
                st.wait_stream(main_stream)
            err = ""
            try:
                for i in range(max(args.warmup, S)):  # the first call of a slot captures its graph
                    replay(i)
                torch.cuda.synchronize()
            except RuntimeError as e:
                err = str(e).split("\n")[0][:120] or "RuntimeError"
            if reduce_scalar(0.0 if err else 1.0, dist.ReduceOp.MIN) < 1.0:
                return None, err or "capture failed on another rank"
            keep = extra_warmup[0]
            settle(replay)
            if not publish:
                extra_warmup[0] = keep  # (the line reports the untimed steps before `value`'s region)

            def run(steps):
                barrier()
                t0 = time.perf_counter()
                outs = None
                for i in range(steps):
                    outs = replay(i)
                for st in streams:
                    main_stream.wait_stream(st)
                close_region(outs)
                return time.perf_counter() - t0

            if publish:
                region["run"] = run
                region["replay"], region["streams"] = replay, streams
            return run(args.steps), ""

        if args.eager:
            elapsed = timed_eager()
        else:
            elapsed, why = timed_in_flight()
            if elapsed is None:  # a runtime that cannot capture: same kernels, launched from Python, and the line says so
                torch.cuda.synchronize()
                launch_note = "python, one stream (HIP-graph capture failed: %s)" % why
                S = 1
                elapsed = timed_eager()

        # ---- steady state: the same loop for ~args.steady_seconds (clocks settled, thousands of launches), reported beside `value`
        steady = None
        if args.steady_seconds > 0:
            n_steady = max(int(args.steady_seconds / (elapsed / args.steps)), args.steps)
            n_steady = int(reduce_scalar(float(n_steady), dist.ReduceOp.MAX))  # every rank runs the same count
            steady = (n_steady, reduce_scalar(region["run"](n_steady), dist.ReduceOp.MAX))


        # ---- are the outputs of the timed MODE right?  V steps exactly as in the timed region (S samples in flight, replayed), every
        # step's stage-3 draw seeded, every step's (depth, confidence) kept; then the same steps one at a time, eagerly, under the same
        # seeds: the two must agree BIT FOR BIT (rounds 2-5: overlapped forwards had come out a few pixels -- now and then entirely --
        # wrong: DESIGN_LESSONS.md lessons 45-46)
        verified = None
        if not args.eager and launch_note is None and args.verify_steps > 0 and "replay" in region:
            V = args.verify_steps
            kept = []
            for i in range(V):
                torch.manual_seed(77000 + i)
                d, c = region["replay"](i)
                with torch.cuda.stream(region["streams"][i % S]):
                    kept.append((d.clone(), c.clone()))
            torch.cuda.synchronize()
            bad, worst = 0, 0.0
            for i in range(V):
                torch.manual_seed(77000 + i)
                d, c, _ = step(i)
                if not (torch.equal(d, kept[i][0]) and torch.equal(c, kept[i][1])):
                    bad += 1
                    worst = max(worst, float(((d - kept[i][0]).abs() / d.abs()).max()))
            torch.cuda.synchronize()
            del kept
            verified = {"steps": V, "steps_that_differ_from_the_eager_forward": bad, "max_relative_depth_difference": worst,
                        "what": f"{V} steps in the timed mode ({S} in flight, {args.launch} replay, seeded draws) against the same steps one "
                                "at a time, launched from Python: depth and confidence compared bit for bit"}
            bad_all = reduce_scalar(float(bad), dist.ReduceOp.SUM)
            verified["steps_that_differ_all_ranks"] = int(bad_all)

        # the other input mode beside `value` (rounds 2-4's `value` had all six images copied into the slot's static buffers inside the
        # timed region; eval.py's pipeline, which recycles its upload buffers, runs that mode): the same K steps, same box, same process
        other_mode = None
        if not args.eager and launch_note is None:
            alt, _ = timed_in_flight(copy_inputs=not args.copy_inputs, publish=False)
            if alt is not None:
                other_mode = reduce_scalar(alt, dist.ReduceOp.MAX)

    # per-rank spread beside the max-reduced figure, and proof that the collectives saw every rank (a line printed by a job whose
    # ranks never met would otherwise look like an N-GPU result)
    elapsed_own = elapsed
    elapsed_min = reduce_scalar(elapsed_own, dist.ReduceOp.MIN)
    elapsed = reduce_scalar(elapsed_own, dist.ReduceOp.MAX)
    ranks_seen = int(round(reduce_scalar(1.0, dist.ReduceOp.SUM)))
    numa_notes = [numa_note]
    if launched:
        numa_notes = [None] * world
        dist.all_gather_object(numa_notes, numa_note)
    assert ranks_seen == world, f"the collectives saw {ranks_seen} of {world} ranks"

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        k_ms = sum(r[0] for r in recs)  # over the `sampled` steps that carried events
        k_bytes = sum(r[1] for r in recs)
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # what actually bounds the kernel: every bilinear corner of every (view, pixel, hypothesis) is a
        # C*4-byte read through the CU's vector L1 / texture addresser: 4 corners x N x h x w x D x C x 4 B per launch
        def tap_bytes(tag):
            f = tag.split("_")  # C{C}_D{D}_{h}x{w}_N{N}_{kind}
            C_, D_, (h_, w_), N_ = int(f[0][1:]), int(f[1][1:]), map(int, f[2].split("x")), int(f[3][1:])
            return 4 * N_ * h_ * w_ * D_ * C_ * 4
        k_taps = sum(tap_bytes(r[2]) for r in recs)
        L1_PEAK_GBS = 64 * 256 * 2.4  # 64 B/clk/CU x 256 CUs x 2.4 GHz = 39.3 TB/s (MI355X_MICROARCH.md)
        l1_achieved = k_taps / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        per = {}
        for ms, nb, tag in recs:
            a = per.setdefault(tag, [0.0, 0, 0])
            a[0] += ms
            a[1] += nb
            a[2] += 1
        # HBM traffic of the same launches from the committed PMC passes (profiles/pmc_traffic.json, see
        # scripts/make_traffic_json.py for the collection + gfx950 FETCH_SIZE correction); null when not available
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        traffic_note = ("HBM bytes per step over the same launches: a COMMITTED measurement (rocprofv3 PMC passes, profiles/pmc_traffic.json, "
                        "stamped with the hash of the kernel sources it was taken on), not a counter read of this run")
        tj = json.load(open(tpath)) if os.path.isfile(tpath) else None
        if tj is not None and tj.get("kernel_source_sha256") != warp_kernel_source_hash():
            traffic_note = "null: profiles/pmc_traffic.json was measured on different kernel sources (hash mismatch) -- re-collect"
            tj = None
        tk = None if tj is None else tj.get("configs", {}).get(f"{W}x{H}_N{n_src}", {}).get("kernels")
        if tj is not None and tk is None:
            traffic_note = f"null: profiles/pmc_traffic.json has no passes for {W}x{H}_N{n_src} (scripts/gpu_pmc_traffic.sh)"
        if tk is not None:
            try:
                per_step = 0
                for ms, nb, tag in recs[:len(recs) // sampled]:
                    C_, D_ = tag.split("_")[0], tag.split("_")[1]
                    per_step += tk[f"{C_}_{D_}_{'pixelwise' if tag.endswith('pixelwise') else 'vw'}"]["hbm_bytes_per_launch"]
                traffic = per_step
            except KeyError:
                traffic = None
        plan_launches = 0
        if args.launch == "plan" and "slots" in region:
            for handle, _, _ in region["slots"][0].cache.values():
                plan_launches = handle.count
        # every untimed step before the K timed ones: the W asked for (at least one per slot: its recording) plus the settle pass
        untimed = max(args.warmup, S if not args.eager else 0) + extra_warmup[0]
        baseline_config = {(1200, 1600, 5): "BASELINE configs[1]", (1056, 1920, 7): "BASELINE configs[2]",
                           (2048, 3072, 10): "BASELINE configs[4], one GPU's share"}.get((H, W, n_src), "not a BASELINE config")
        line = {
            "metric": f"depth-maps/sec at {W}x{H} N={n_src} src views", "value": round(value, 4), "unit": "depth-maps/s",
            "n_gpus": world, "steps": args.steps, "warmup": untimed, "warmup_requested": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PatchmatchNet.forward, {W}x{H}, N={n_src} source views, iters (1,2,2), B=1 "
                                   f"({baseline_config}); ref views sharded 1/rank", "weights": weights,
                       "distinct_samples": len(samples),
                       "arithmetic": "fp32 results throughout; FeatureNet's conv1..conv10 run on the fp16 matrix cores with SPLIT operands "
                                     "(x = hi + lo/2048, three exact-product MFMAs, fp32 accumulation): 2-4e-7 of the output scale, the error of "
                                     "an fp32 convolution (tests/test_f16s_emulation.py, tests/test_hip_parity.py); model.feature.f16_split = False "
                                     "selects the fp32 Winograd / fp32 MFMA kernels",
                       "scene": "photo-consistent rendered surface (tests/synth.render_scene), one texture seed per sample"
                                if args.scene == "surface" else "rolled noise images (rounds 1-2)",
                       "parallelism": f"ref-view shards x{world}, all-gather of depth+confidence",
                       "hardware_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (GPU_MAX_HW_QUEUES unset: 4)"),
                       "collective_stream": "RCCL's own stream, ordered after the slot streams through the launch stream (the closing all-gather "
                                            "is issued there after main_stream.wait_stream(slot) for every slot)" if launched else None,
                       "ranks_seen": ranks_seen, "backend": dist.get_backend() if launched else "none (single process, no process group)",
                       "ms_per_step_rank_min": round(elapsed_min / args.steps * 1e3, 4),
                       "ms_per_step_rank_max": round(elapsed / args.steps * 1e3, 4),
                       "numa": numa_notes,
                       "untimed_steps_before_the_timed_region": untimed,
                       "in_flight": S, "launch": launch_note or ("python, one stream" if args.eager else
                       ("launch-plan replay (plain hipLaunchKernel calls from C, pmn_plan_launch: %d launches per forward)" % plan_launches
                        if args.launch == "plan" else "HIP-graph replay") +
                       f", {S} sample(s) in flight on {S} HIP stream(s) per GPU; images "
                       + ("copied into the slot's static buffers" if args.copy_inputs else
                          "read in place through a device table of addresses (pmn_stem_f16s_views)"))},
            "value_other_input_mode": None if other_mode is None else {
                "mode": "images read in place through a device table of addresses" if args.copy_inputs else
                        "all six images copied into the slot's static buffers inside the timed region (rounds 2-4's `value`, eval.py's mode)",
                "value": round(world * args.steps / other_mode, 4), "ms_per_step": round(other_mode / args.steps * 1e3, 4)},
            "outputs_verified": verified,
            "steady_state": None if steady is None else {
                "steps": steady[0], "seconds": round(steady[1], 3), "value": round(world * steady[0] / steady[1], 2),
                "note": "the timed region's loop repeated for ~%.0f s in the same mode (not the contract's K steps)" % args.steady_seconds},
            "single_stream_eager": {"value": round(R / eager_elapsed, 2), "ms_per_step": round(eager_elapsed / R * 1e3, 4),
                                    "steps": R, "note": "this rank, one sample at a time, kernels issued from Python (the "
                                    "round-1 mode; = one sample's latency); the roofline events were recorded in this pass"},
            "roofline": {"bound": "hbm", "kernel": "gather_corr_kernel (pmn_warp_correlate)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "frac_of_measured_achievable": round(achieved / 6290.0, 5),  # 6.29 TB/s measured achievable
                         "bound_actual": "vector L1 / texture addresser (64 B/clk/CU): HBM traffic is below the algorithmic bytes, "
                                         "the taps are re-read ~37x through L1 (DESIGN.md section 4)",
                         "tap_bytes_per_step": int(k_taps / sampled), "l1_achieved": round(l1_achieved, 1),
                         "l1_peak": round(L1_PEAK_GBS, 1), "l1_frac": round(l1_achieved / L1_PEAK_GBS, 4),
                         "traffic": traffic,
                         "traffic_unit": traffic_note,
                         "measured_in": "eager single-stream pass of the same run (see single_stream_eager): a replayed "
                                        "graph's launches cannot be bracketed by events",
                         "launches": len(recs), "steps_with_events": sampled,
                         "kernel_ms_per_step": round(k_ms / sampled, 4), "alg_bytes_per_step": int(k_bytes / sampled),
                         "per_shape": {k: {"ms_avg": round(v[0] / v[2], 4),
                                           "GBps": round(v[1] / (v[0] * 1e-3) / 1e9, 1) if v[0] > 0 else 0.0}
                                       for k, v in per.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            if affinity0 is not None:
                os.sched_setaffinity(0, affinity0)  # the baselines time the HOST: every core the process was given
            # round 4: the REAL reference timed on this box -- host cores (kind "reference") and, through PyTorch-ROCm, this GPU (the
            # denominator of the north star's 4x); the CPU port of rounds 1-3 stays beside it at one thread count
            ref_cpu = reference_cpu_baseline(H, W, n_src)
            port = cpu_baseline(H, W, n_src, DEFAULT_KW, thread_counts=(32,) if ref_cpu else (8, 32, 64))
            if ref_cpu is not None:
                ref_cpu["port"] = {k: port[k] for k in ("value", "unit", "cores", "kind", "sample")}
                line["cpu_baseline"] = ref_cpu
            else:
                port["note"] = "oracle/_ref/patchmatchnet_reference.pt was never built (python oracle/make_ref.py needs the reference checkout)"
                line["cpu_baseline"] = port
            ref_gpu = reference_rocm_baseline(H, W, n_src, device, model)
            if ref_gpu is not None:
                if "value" in ref_gpu:
                    ref_gpu["this_engine_over_reference"] = round(value / ref_gpu["value"], 2)
                    ref_gpu["this_engine_single_stream_over_reference"] = round((R / eager_elapsed) / ref_gpu["value"], 2)
                line["reference_rocm"] = ref_gpu
        if world == 1 and not args.no_cpu_baseline and not args.eager and (H, W, n_src) == (1200, 1600, 5):
            samples.clear()  # (the pipeline leg brings its own inputs; give the child process the memory)
            torch.cuda.empty_cache()
            line["eval_end_to_end"] = eval_end_to_end_leg(H, W, n_src, device)
        print(json.dumps(line), flush=True)
    if launched:
        dist.barrier()
        dist.destroy_process_group()
    if verified is not None and verified.get("steps_that_differ_all_ranks", 0) > 0:
        # the line is printed (the numbers are what they are), but a timed mode whose outputs are not the eager forward's is a failure
        raise SystemExit(f"bench.py: {verified['steps_that_differ_all_ranks']} of {verified['steps']} verified steps differ from the eager forward")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself (imported read-only from
/root/reference, CPU, fp32).  Run in the authoring container only:

    python tests/golden/make_golden.py

Outputs (committed):
  params_000007.npz        the reference checkpoint's tensors as float32 numpy (key names un-prefixed); the
                           parity fixture weights for GPU-box tests, where /root/reference does not exist.
  cascade_96x128_n2.npz    default config (iters 1,2,2; neighbours 0/8/16, 9/9/9), B=1, 2 source views: inputs,
                           FeatureNet outputs, and every hot-path intermediate of every PatchMatch iteration,
                           plus the final refined depth / confidence.
  cascade_variant_b2.npz   B=2, 48x64, 3 source views, iters (2,1,1), propagate (4,8,16), evaluate (9,17,9): exercises
                           the 4- and 17-neighbour tables, batch > 1 and propagation on stage 1.
  ops_small.npz            differentiable_warping known answers incl. negative depth and src size != ref size.
  evaluation_io.npz        Evaluation.forward at its own boundary (models/patchmatch.py:145-239): the reference's grid / weight /
                           depth_sample / view_weights inputs and (depth, score, view_weights) outputs of stage-3 iteration 1
                           (PixelwiseNet) and stage-2 iteration 1 of the default cascade (``--only evaluation`` writes just this).
  cascade_mixed_sizes.npz  a sample whose two source images differ in size from the reference image and from each other
                           (96x128 / 80x112 / 96x144): the reference's depth, confidence and stage depths (``--only mixed``).
  cfg2_scene.npz           BASELINE configs[1] at FULL size (1600x1200, N=5, iters 1,2,2) on the photo-consistent scene of
                           tests/synth.py (``render_scene``, seed 0; stage-3 noise seed 1234): the reference's final depth,
                           confidence, every stage / iteration depth, stage-3 view weights and the integer confidence index,
                           FREE-RUNNING from the images (``--only scene``; ~15 s of CPU).  Inputs regenerate from the seeds;
                           ``scene_digest`` pins them.
  cascade_odd_counts.npz   the released checkpoint under NON-default hypothesis counts, patchmatch_num_sample = [6, 12, 10] (stage 1, 2, 3):
                           D = 64 / 26 / 20 / 20 / 6 hypotheses per Evaluation call instead of 64 / 32 / 16 / 16 / 8 -- counts that are
                           no multiple of 4 and none of the compile-time bounds of the HIP kernels; the reference's depth, confidence
                           and every hot-path intermediate (as cascade_96x128_n2.npz) at 64x96, two source views (``--only counts``).
  cascade_dilations.npz    the released checkpoint under NON-default propagation ranges, propagation_range = [5, 3, 2] (stage 1, 2, 3; default
                           6 / 4 / 2): other dilations of the offset heads (the fp16-split head kernel instantiates only the defaults:
                           the fp32 convolution takes over) and other fixed neighbour tables (evaluation dilation = range - 1); full trace
                           at 64x96, two source views (``--only dilations``).
  cascade_resized_100x130.npz  a 100x130 sample (not multiples of 8): the reference stretches the images to 96x128, rescales the
                           intrinsics in place (models/net.py:304-318), and returns depth (bilinear) and confidence (nearest) at
                           100x130 -- its final depth, confidence, stage depths and the intrinsics it left behind (``--only resized``).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import refutil  # noqa: E402


def t2n(t):
    return t.detach().cpu().numpy()


def dump_cascade(path, model, n_views, H, W, B=1, seed=1234, cameras=None):
    imgs = refutil.synthetic_images(n_views, H, W)
    if B > 1:
        imgs = [torch.cat([im, torch.flip(im, dims=[3])] + [im] * (B - 2), 0)[:B].contiguous() for im in imgs]
    intr, extr = refutil.synthetic_cameras(n_views, H, W) if cameras is None else cameras
    intr = np.repeat(intr, B, 0)
    extr = np.repeat(extr, B, 0)
    dmin = np.array([425.0, 400.0][:B], np.float32)
    dmax = np.array([935.0, 900.0][:B], np.float32)
    noise = torch.rand(B, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(seed))
    depth, conf, dpm, tr = refutil.trace_reference_forward(
        model, [i.clone() for i in imgs], torch.from_numpy(intr).clone(), torch.from_numpy(extr).clone(),
        torch.from_numpy(dmin), torch.from_numpy(dmax), noise)
    out = {"intrinsics": intr, "extrinsics": extr, "depth_min": dmin, "depth_max": dmax, "noise": t2n(noise),
           "depth": t2n(depth), "confidence": t2n(conf), "n_views": np.int32(n_views)}
    for v, im in enumerate(imgs):
        out[f"image_{v}"] = t2n(im)
    for v, f in enumerate(tr["features"]):
        for s, x in f.items():
            out[f"feature_{v}_s{s}"] = t2n(x)
    for s in (1, 2, 3):
        for k, x in tr[f"stage{s}"].items():
            out[f"s{s}_{k}"] = t2n(x)
        for it, rec in enumerate(tr[s]):
            # (the [D,K,h,w] aggregation weights and the pre-softmax score are derivable and left out to keep
            #  the fixtures small; the per-view similarity is kept for view 0 only)
            for k in ("depth_sample", "similarity", "score", "view_weights", "depth"):
                out[f"s{s}_it{it + 1}_{k}"] = t2n(rec[k])
            for v, x in enumerate(rec["pixelwise_in"][:1]):
                out[f"s{s}_it{it + 1}_view_similarity_{v}"] = t2n(x)
        for it, d in enumerate(dpm[s]):
            out[f"s{s}_it{it + 1}_depth_out"] = t2n(d)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


def dump_evaluation_io(path, model, n_views=3, H=96, W=128, seed=1234):
    """Inputs / outputs of the reference's Evaluation.forward calls (same sample as cascade_96x128_n2.npz)."""
    imgs = refutil.synthetic_images(n_views, H, W)
    intr, extr = refutil.synthetic_cameras(n_views, H, W)
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(seed))
    _, _, _, tr = refutil.trace_reference_forward(
        model, [i.clone() for i in imgs], torch.from_numpy(intr).clone(), torch.from_numpy(extr).clone(),
        torch.from_numpy(dmin), torch.from_numpy(dmax), noise)
    # (features / cameras of this sample are in cascade_96x128_n2.npz)
    out = {"n_views": np.int32(n_views)}
    for s, it in ((3, 0), (2, 0)):
        rec = tr[s][it]
        for k in ("depth_sample", "grid", "weight", "view_weights_in", "depth", "score", "view_weights"):
            out[f"s{s}_it{it + 1}_{k}"] = t2n(rec[k])
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


def dump_scene(path, model, n_views=6, H=1200, W=1600, scene_seed=0, noise_seed=1234, stride=1, camera_step=0.08):
    """The reference itself, end to end from images, on the configuration every number is quoted on (cfg2_scene.npz), and -- round 6 --
    on BASELINE configs[2] (1920x1056, N=7: cfg3_scene.npz) and configs[4] (3072x2048, N=10: cfg5_scene.npz, every ``stride``-th
    pixel of every map in both directions so that the fixture stays a few MB; the comparison is then made on that pixel subset)."""
    import synth
    # (camera_step: the rig's angle between neighbouring views.  At 0.08 rad the outermost of eleven views look at the wavy surface
    # at 0.64 - 0.8 rad, where a ray can meet it more than once and the renderer's Newton iteration picks a root by the last bit of
    # its arithmetic: such views do not render identically on two hosts, so the eleven-view fixture uses 0.04.)
    imgs, intr, extr, depth_gt = synth.render_scene(n_views, H, W, scene_seed, cameras=synth.synthetic_cameras(n_views, H, W, camera_step))
    dmin, dmax = np.array([425.0], np.float32), np.array([935.0], np.float32)
    noise = torch.rand(1, 48, H // 8, W // 8, generator=torch.Generator().manual_seed(noise_seed))
    depth, conf, dpm, tr = refutil.trace_reference_forward(
        model, [i.clone() for i in imgs], torch.from_numpy(intr).clone(), torch.from_numpy(extr).clone(),
        torch.from_numpy(dmin), torch.from_numpy(dmax), noise)
    score = tr[1][-1]["score"]
    D = score.shape[1]
    idx = (score * torch.arange(D, dtype=torch.float32).view(1, D, 1, 1)).sum(1).long().clamp(0, D - 1)  # net.py:294-297
    sub = lambda a: np.ascontiguousarray(a[..., ::stride, ::stride])  # noqa: E731
    out = {"scene_seed": np.int32(scene_seed), "noise_seed": np.int32(noise_seed), "n_views": np.int32(n_views),
           "H": np.int32(H), "W": np.int32(W), "stride": np.int32(stride), "camera_step": np.float64(camera_step), "scene_digest": np.array(synth.scene_digest(imgs)),
           "depth": sub(t2n(depth)), "confidence": sub(t2n(conf)), "depth_index": sub(t2n(idx).astype(np.int8)),
           "view_weights": sub(t2n(tr[3][0]["view_weights"]))}
    for s in (1, 2, 3):
        for it, d in enumerate(dpm[s]):
            out[f"s{s}_it{it + 1}_depth_out"] = sub(t2n(d))
    out["scene_thumb"], out["scene_sums"] = synth.scene_signature(imgs)  # host-tolerant form of scene_digest
    np.savez_compressed(path, **out)
    gt = depth_gt.numpy()
    err = np.abs(t2n(depth)[0, 0] - gt)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6), "| reference vs ground truth: median %.3f mm, p90 %.3f mm"
          % (np.median(err), np.quantile(err, 0.9)))


MIXED_SIZES = [(96, 128), (80, 112), (96, 144)]  # (H, W) of the reference view and the two source views


def mixed_size_inputs():
    """Seeded sample whose images differ in size (all multiples of 8): cameras of the synthetic rig at 96x128, every view's intrinsics
    scaled to its own image size (what datasets/mvs.py:84-85 does after a per-image down-scale)."""
    import synth
    H0, W0 = MIXED_SIZES[0]
    intr, extr = synth.synthetic_cameras(len(MIXED_SIZES), H0, W0)
    intr = intr.copy()
    imgs = []
    for v, (H, W) in enumerate(MIXED_SIZES):
        imgs.append(synth.synthetic_images(len(MIXED_SIZES), H, W)[v])
        intr[0, v, 0] *= W / W0
        intr[0, v, 1] *= H / H0
    noise = torch.rand(1, 48, H0 // 8, W0 // 8, generator=torch.Generator().manual_seed(77))
    return imgs, intr, extr, np.array([425.0], np.float32), np.array([935.0], np.float32), noise


def dump_mixed(path, model):
    """The reference on a sample with source images of other sizes than the reference image (legal: models/module.py:130-181 warps
    every view at its own size): final depth, confidence, stage depths.  Inputs regenerate from seeds (``mixed_size_inputs``)."""
    imgs, intr, extr, dmin, dmax, noise = mixed_size_inputs()
    depth, conf, dpm, tr = refutil.trace_reference_forward(
        model, [i.clone() for i in imgs], torch.from_numpy(intr).clone(), torch.from_numpy(extr).clone(),
        torch.from_numpy(dmin), torch.from_numpy(dmax), noise)
    out = {"depth": t2n(depth), "confidence": t2n(conf)}
    for s in (1, 2, 3):
        for it, d in enumerate(dpm[s]):
            out[f"s{s}_it{it + 1}_depth_out"] = t2n(d)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


ODD_COUNTS = [6, 12, 10]  # patchmatch_num_sample, stage 1..3 (the parameter shapes do not depend on it: the checkpoint loads as is)


def dump_counts(path):
    ref_net, _, _ = refutil.import_reference()
    model = ref_net.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                                  patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=ODD_COUNTS,
                                  propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])
    model.load_state_dict(refutil.load_reference_state_dict(), strict=True)
    model.eval()
    dump_cascade(path, model, 3, 64, 96, seed=66)  # the full trace, like the default case (tests/goldenutil.py CASES["counts"])


ODD_RANGES = [5, 3, 2]  # propagation_range, stage 1..3 (no parameter shape depends on it)


def dump_dilations(path):
    ref_net, _, _ = refutil.import_reference()
    model = ref_net.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=ODD_RANGES,
                                  patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16],
                                  propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])
    model.load_state_dict(refutil.load_reference_state_dict(), strict=True)
    model.eval()
    dump_cascade(path, model, 3, 64, 96, seed=88)


def resized_inputs():
    """Seeded 100x130 sample of the synthetic rig (sizes the reference has to adjust: 100 -> 96, 130 -> 128)."""
    import synth
    H, W = 100, 130
    imgs = synth.synthetic_images(3, H, W)
    intr, extr = synth.synthetic_cameras(3, H, W)
    noise = torch.rand(1, 48, 12, 16, generator=torch.Generator().manual_seed(55))
    return imgs, intr, extr, np.array([425.0], np.float32), np.array([935.0], np.float32), noise


def dump_resized(path, model):
    """The reference on a sample whose size is not a multiple of 8 (adjust_image_dims, models/net.py:304-318)."""
    imgs, intr, extr, dmin, dmax, noise = resized_inputs()
    K = torch.from_numpy(intr).clone()
    depth, conf, dpm, tr = refutil.trace_reference_forward(
        model, [i.clone() for i in imgs], K, torch.from_numpy(extr).clone(), torch.from_numpy(dmin), torch.from_numpy(dmax), noise)
    out = {"depth": t2n(depth), "confidence": t2n(conf), "intrinsics_after": t2n(K)}
    for s in (1, 2, 3):
        for it, d in enumerate(dpm[s]):
            out[f"s{s}_it{it + 1}_depth_out"] = t2n(d)
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6), tuple(depth.shape), tuple(conf.shape))


def dump_ops(path):
    _, _, ref_module = refutil.import_reference()
    g = torch.Generator().manual_seed(7)
    out = {}
    # case A: same size, some hypotheses behind the source camera (negative-depth sentinel)
    # case B: source map smaller than the reference map
    for name, (C, D, h, w, hs, ws) in {"A": (8, 5, 9, 11, 9, 11), "B": (4, 3, 10, 12, 7, 9)}.items():
        src = torch.randn(2, C, hs, ws, generator=g)
        f = 30.0
        K = torch.tensor([[f, 0, w / 2, 0], [0, f, h / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])
        a = 0.3
        E = torch.eye(4)
        E[:3, :3] = torch.tensor([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
        E[:3, 3] = torch.tensor([2.0, -1.0, -6.0])  # pushes near hypotheses behind the source camera
        ref_proj = K.unsqueeze(0).repeat(2, 1, 1)
        src_proj = (K @ E).unsqueeze(0).repeat(2, 1, 1)
        src_proj[1, :3, 3] += torch.tensor([3.0, 2.0, 1.0])
        depth = 2.0 + 10.0 * torch.rand(2, D, h, w, generator=g)
        warped = ref_module.differentiable_warping(src, src_proj, ref_proj, depth)
        out.update({f"{name}_src": t2n(src), f"{name}_src_proj": t2n(src_proj), f"{name}_ref_proj": t2n(ref_proj),
                    f"{name}_depth": t2n(depth), f"{name}_warped": t2n(warped)})
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


def dump_rig(path, model):
    """Round 5: a rig without the y-axis symmetry of every other fixture (tests/synth.general_cameras: roll, pitch, off-orbit
    translations, fx != fy, per-view principal points) -- every entry of the relative projections is exercised."""
    import synth
    dump_cascade(path, model, 4, 96, 128, seed=777, cameras=synth.general_cameras(4, 96, 128))


def main():
    assert refutil.have_reference(), "needs /root/reference"
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = refutil.build_reference_model()
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "evaluation":
        dump_evaluation_io(os.path.join(HERE, "evaluation_io.npz"), model)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "mixed":
        dump_mixed(os.path.join(HERE, "cascade_mixed_sizes.npz"), model)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "counts":
        dump_counts(os.path.join(HERE, "cascade_odd_counts.npz"))
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "dilations":
        dump_dilations(os.path.join(HERE, "cascade_dilations.npz"))
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "rig":
        dump_rig(os.path.join(HERE, "cascade_general_rig.npz"), model)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "resized":
        dump_resized(os.path.join(HERE, "cascade_resized_100x130.npz"), model)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "scene":
        dump_scene(os.path.join(HERE, "cfg2_scene.npz"), model)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "scene3":  # BASELINE configs[2]
        dump_scene(os.path.join(HERE, "cfg3_scene.npz"), model, n_views=8, H=1056, W=1920, scene_seed=3, noise_seed=4321, stride=2)
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "scene5":  # BASELINE configs[4], one GPU's share
        dump_scene(os.path.join(HERE, "cfg5_scene.npz"), model, n_views=11, H=2048, W=3072, scene_seed=5, noise_seed=555, stride=4, camera_step=0.04)
        return
    sd = refutil.state_dict_numpy(model)
    p = os.path.join(HERE, "params_000007.npz")
    np.savez_compressed(p, **sd)
    print("wrote", p, "%.2f MB" % (os.path.getsize(p) / 1e6))
    dump_cascade(os.path.join(HERE, "cascade_96x128_n2.npz"), model, 3, 96, 128)

    # variant: non-default neighbour tables / iterations; weights = checkpoint where shapes agree, seeded random
    # (small, so learned offsets stay sub-pixel..few-pixel) for the re-shaped offset heads.
    ref_net, _, _ = refutil.import_reference()
    variant = ref_net.PatchmatchNet(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                                    patchmatch_iteration=[2, 1, 1], patchmatch_num_sample=[8, 8, 16],
                                    propagate_neighbors=[4, 8, 16], evaluate_neighbors=[9, 17, 9])
    base = refutil.load_reference_state_dict()
    own = variant.state_dict()
    g = torch.Generator().manual_seed(99)
    for k, v in own.items():
        if k in base and base[k].shape == v.shape:
            own[k] = base[k].clone()
        elif v.dtype.is_floating_point:
            own[k] = 0.05 * torch.randn(v.shape, generator=g)
    variant.load_state_dict(own)
    variant.eval()
    np.savez_compressed(os.path.join(HERE, "params_variant.npz"), **refutil.state_dict_numpy(variant))
    dump_cascade(os.path.join(HERE, "cascade_variant_b2.npz"), variant, 4, 48, 64, B=2, seed=4321)
    dump_ops(os.path.join(HERE, "ops_small.npz"))
    dump_evaluation_io(os.path.join(HERE, "evaluation_io.npz"), model)
    dump_scene(os.path.join(HERE, "cfg2_scene.npz"), model)
    dump_mixed(os.path.join(HERE, "cascade_mixed_sizes.npz"), model)
    dump_resized(os.path.join(HERE, "cascade_resized_100x130.npz"), model)
    dump_counts(os.path.join(HERE, "cascade_odd_counts.npz"))
    dump_dilations(os.path.join(HERE, "cascade_dilations.npz"))
    dump_rig(os.path.join(HERE, "cascade_general_rig.npz"), model)


if __name__ == "__main__":
    main()

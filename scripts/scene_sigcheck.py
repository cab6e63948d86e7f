import sys, os, numpy as np, torch, time
sys.path.insert(0, "tests")
import synth, goldenutil as GU
for f in ("cfg3_scene.npz", "cfg5_scene.npz", "cfg2_scene.npz"):
    g = GU.load_npz(f)
    t0 = time.time()
    imgs, _, _, _ = synth.render_scene(int(g["n_views"]), int(g["H"]), int(g["W"]), int(g["scene_seed"]), device="cuda")
    torch.cuda.synchronize()
    dig = synth.scene_digest(imgs) == str(g["scene_digest"])
    if "scene_thumb" in g:
        print(f, "gpu render %.1f s" % (time.time() - t0), "digest equal:", dig, synth.scene_matches(imgs, g["scene_thumb"], g["scene_sums"]))
        t2, s2 = synth.scene_signature(imgs)
        d = np.abs(t2.astype(np.int16) - g["scene_thumb"].astype(np.int16))
        print("   per-view differing sampled bytes:", [(int((d[v] > 0).sum()), int(d[v].max())) for v in range(d.shape[0])], "sum diffs", (s2 - g["scene_sums"]).tolist())
    else:
        print(f, "gpu render %.1f s" % (time.time() - t0), "digest equal:", dig)

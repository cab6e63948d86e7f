#!/usr/bin/env python
"""FeatureNet alone, captured per slot and replayed concurrently on three streams: which of its outputs deviate from the eager
result, how (NaN? magnitude), and from which layer on (the intermediate maps are returned too)."""
import os
import sys

ROOT = os.environ.get("PMN_PROBE_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import goldenutil as GU  # noqa: E402
import patchmatchnet_amd as P  # noqa: E402
from patchmatchnet_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (96, 128)
dev = torch.device("cuda", 0)
_, params, kw = GU.load_case("default")
model = P.PatchmatchNet(**kw)
model.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
model = model.to(dev).eval()
fn = model.feature
pk = fn._packed()
S = 3


def layers(imgs):
    """forward_hip, returning every intermediate map (name -> tensor)."""
    out = {}
    B = imgs[0].shape[0]
    t = torch.empty((B * len(imgs), H, W, 8), dtype=torch.float32, device=dev)
    for i, im in enumerate(imgs):
        ops.stem_f16s(im, *pk["conv0"], *pk["conv1_f16s"], out=t[i * B:(i + 1) * B])
    out["stem"] = t
    for i, (k, s, p) in enumerate(fn._SPEC):
        if i < 2:
            continue
        t = ops.conv2d_f16s(t, *pk[f"conv{i}_f16s"], k, s, relu=True)
        out[f"conv{i}"] = t
    f3, u8 = ops.pointwise_split_mfma(out["conv10"], *pk["fpn8_mfma"], cout=112, ca=64)
    f2, u4 = ops.fpn_level(out["conv7"], u8, *pk["fpn4"], ca=32)
    f1, _ = ops.fpn_level(out["conv4"], u4, *pk["fpn2"], ca=16)
    out.update(f3=f3, u8=u8, f2=f2, u4=u4, f1=f1)
    return out


inputs = []
for k in range(S):
    g = torch.Generator().manual_seed(100 + k)
    inputs.append([torch.rand(1, 3, H, W, generator=g).to(dev) for _ in range(3)])
with torch.no_grad():
    want = [{n: v.clone() for n, v in layers(x).items()} for x in inputs]
    torch.cuda.synchronize()
    graphs, outs, streams = [], [], [torch.cuda.Stream(dev) for _ in range(S)]
    for k in range(S):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            for _ in range(int(os.environ.get("PMN_PROBE_REPEAT", "1"))):  # how many FeatureNet passes (= graph nodes) per graph
                o = layers(inputs[k])
        graphs.append(g)
        outs.append(o)
    torch.cuda.synchronize()
    first_bad, nan = {}, 0
    for r in range(rounds):
        got = []
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                graphs[k].replay()
                got.append({n: v.clone() for n, v in outs[k].items()})
        torch.cuda.synchronize()
        for k in range(S):
            for n in want[k]:
                if not torch.equal(got[k][n], want[k][n]):
                    d = (got[k][n] - want[k][n]).abs()
                    nan += int(torch.isnan(got[k][n]).any())
                    first_bad.setdefault((k, n), []).append((r, int((got[k][n] != want[k][n]).sum()), float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else -1.0))
                    break  # the first layer (in execution order) that differs
print(f"FeatureNet x{os.environ.get('PMN_PROBE_REPEAT', '1')} per graph, {S} graphs replayed concurrently x{rounds} at {W}x{H}: first deviating layer per (slot, layer):",
      {k: (len(v), v[:2]) for k, v in first_bad.items()} if first_bad else "none", "| rounds with NaN:", nan)

"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports exactly the entry
points include/pmn_hip.h declares (no compute calls -- there is no GPU in the authoring container)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pmn_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decl = {}
    for m in re.finditer(r"\b(?:int|const char \*)\s*(pmn_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",")]
        decl[m.group(1)] = 0 if args == ["void"] else len(args)
    return decl


def test_header_declares_the_signature_table():
    from patchmatchnet_amd import _lib
    decl = _declared()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, nargs in decl.items():
        assert nargs == len(_lib.SIGNATURES[name]), name
    hdr = open(HEADER).read()
    assert f"#define PMN_MLP_FLOATS {_lib.MLP_FLOATS}" in hdr
    assert f"#define PMN_ABI_VERSION {_lib.ABI_VERSION}" in hdr


def test_library_builds_loads_and_exports_every_symbol():
    from patchmatchnet_amd import _lib
    _lib.build()
    L = _lib.lib()
    assert L.pmn_abi_version() == _lib.ABI_VERSION
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    for name in _declared():
        assert name in exported, name
    # ... and nothing else: the research families and their pmn_set_tuning live in libpmn_hip_experimental.so only
    assert {e for e in exported if e.startswith("pmn_")} == set(_declared())
    assert b"gather_lane_kernel" not in open(_lib.LIB_PATH, "rb").read()
    assert L.pmn_error_string(-2).decode().startswith("unsupported shape")
    # the code object really targets gfx950
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_entry_points_never_block():
    """include/pmn_hip.h's contract: an entry point only ENQUEUES on the caller's stream -- it never synchronises, allocates or
    copies.  _lib.py relies on it (ctypes.PyDLL: calls keep the GIL; a blocking call would stall eval.py's writer and
    pin-memory threads and could deadlock against a thread waiting for the GIL inside the runtime), and so does HIP-graph
    capture of the whole forward.  Enforced on the sources: none of the blocking runtime calls may appear in csrc/."""
    csrc = os.path.join(ROOT, "patchmatchnet_amd", "csrc")
    banned = re.compile(r"\bhip(DeviceSynchronize|StreamSynchronize|EventSynchronize|Malloc\w*|Free\w*|"
                        r"Mem(?:cpy|set)(?:(?!Async)\w)*|HostMalloc|HostFree|StreamWaitEvent|StreamCreate\w*)\s*\(")
    for dirpath, _, files in os.walk(csrc):
        for f in files:
            if f.endswith((".hip", ".hpp")):
                text = re.sub(r"//.*", "", open(os.path.join(dirpath, f)).read())
                text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
                m = banned.search(text)
                assert m is None, (f, m.group(0))
    assert "never block" in open(HEADER).read()


def test_argument_checks_do_not_launch():
    """Bad arguments are rejected on the host before any HIP call (safe without a GPU)."""
    from patchmatchnet_amd import _lib
    L = _lib.lib()
    assert L.pmn_nchw_to_nhwc(None, None, 1, 4, 4, 4, None) == -1
    assert L.pmn_confidence(None, 1, 8, 4, 4, 8, 8, None, None, None) == -1
    assert L.pmn_warp_correlate(None, None, None, None, None, 0, None, None, 1, 1, 64, 8, 8, 4, 4, 4, 4, None, None,
                                None, None, None) == -1


def test_product_refuses_cpu_tensors_and_training_mode():
    import patchmatchnet_amd as P
    with pytest.raises(P.PmnError):
        P.ops.nchw_to_nhwc(torch.zeros(1, 4, 4, 4))
    kw = dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
              patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16], propagate_neighbors=[0, 8, 16],
              evaluate_neighbors=[9, 9, 9])
    m = P.PatchmatchNet(**kw)
    with pytest.raises(P.PmnError):  # training mode: inference-only engine
        m([torch.zeros(1, 3, 64, 64)] * 2, torch.eye(3).repeat(1, 2, 1, 1), torch.eye(4).repeat(1, 2, 1, 1),
          torch.ones(1), 2 * torch.ones(1))
    with pytest.raises(NotImplementedError):  # same legal neighbour counts as the reference get_grid
        P.PatchMatch(propagate_neighbors=5)


def test_state_dict_names_match_reference_checkpoint():
    import goldenutil as GU
    import patchmatchnet_amd as P
    _, params, kw = GU.load_case("default")
    m = P.PatchmatchNet(**kw)
    own = m.state_dict()
    assert set(own) == set(params)
    for k, v in params.items():
        assert tuple(own[k].shape) == tuple(v.shape), k
    # DataParallel-style prefixes are accepted (reference eval.py:33-35)
    m.load_state_dict({"module." + k: torch.from_numpy(v) for k, v in params.items()}, strict=True)


def test_mlp_packing_folds_batchnorm():
    import goldenutil as GU
    import patchmatchnet_amd as P
    from oracle import oracle as O
    _, params, kw = GU.load_case("default")
    m = P.PatchmatchNet(**kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    net = m.patchmatch_1.evaluation.similarity_net
    blk = net.packed().astype(np.float64)
    G = 4
    x = np.random.default_rng(0).standard_normal((1, G, 50)).astype(np.float32)
    ref = O.pointwise_mlp(x, params, "patchmatch_1.evaluation.similarity_net", "similarity", sigmoid=False)[0]
    rec = blk[:320].reshape(16, 20)
    w0, w1, t0 = rec[:, :G], rec[:, 8:16].T, rec[:, 16]
    t1, w2, b2 = blk[320:328], blk[328:336], blk[336]
    h0 = np.maximum(w0 @ x[0].astype(np.float64) + t0[:, None], 0)
    h1 = np.maximum(w1 @ h0 + t1[:, None], 0)
    out = w2 @ h1 + b2
    assert (np.abs(out - ref) / np.maximum(np.abs(ref), 1.0)).max() < 1e-4
    assert net.packed() is net.packed()  # cached


def test_conv_packing_layouts():
    """pack_conv / pack_deconv: BatchNorm folding and the [K][K][cin][coutp] layout pmn_conv2d reads."""
    from patchmatchnet_amd import params as PP
    g = torch.Generator().manual_seed(0)
    w = torch.randn(18, 16, 3, 3, generator=g)
    b = torch.randn(18, generator=g)
    wp, sp = PP.pack_conv(w, bias=b)
    assert wp.shape == (3, 3, 16, 32) and sp.shape == (32,)  # 18 -> padded to a multiple of 16
    np.testing.assert_allclose(wp[1, 2, 5, :18], w[:, 5, 1, 2].numpy(), rtol=1e-7)
    assert (wp[..., 18:] == 0).all() and (sp[18:] == 0).all()
    np.testing.assert_allclose(sp[:18], b.numpy(), rtol=1e-7)
    w8 = torch.randn(8, 3, 3, 3, generator=g)
    bn = (torch.rand(8, generator=g) + 0.5, torch.randn(8, generator=g), torch.randn(8, generator=g),
          torch.rand(8, generator=g) + 0.5)
    wp, sp = PP.pack_conv(w8, bn=bn)
    assert wp.shape == (3, 3, 3, 8)
    scale = bn[0].double() / torch.sqrt(bn[3].double() + 1e-5)
    np.testing.assert_allclose(wp[0, 0, 1], (w8[:, 1, 0, 0].double() * scale).float().numpy(), rtol=1e-6)
    np.testing.assert_allclose(sp, (bn[1].double() - bn[2].double() * scale).float().numpy(), rtol=1e-6, atol=1e-7)
    # folded conv == conv + BatchNorm (eval) on the CPU
    x = torch.randn(1, 3, 9, 9, generator=g)
    ref = torch.nn.functional.batch_norm(torch.nn.functional.conv2d(x, w8, None, 1, 1), bn[2], bn[3], bn[0], bn[1], False, 0.0,
                                         1e-5)
    wf = torch.from_numpy(wp).permute(3, 2, 0, 1)  # back to [cout,cin,K,K]
    got = torch.nn.functional.conv2d(x, wf, torch.from_numpy(sp), 1, 1)
    assert float((got - ref).abs().max()) < 1e-5
    wd = torch.randn(8, 8, 3, 3, generator=g)
    wdp, sdp = PP.pack_deconv(wd, bn=bn)
    assert wdp.shape == (3, 3, 8, 8)
    np.testing.assert_allclose(wdp[2, 1, 3], (wd[3, :, 2, 1].double() * scale).float().numpy(), rtol=1e-6)


def test_fold_fpn_is_the_reference_head():
    """params.fold_fpn: the composed 1x1 convolutions reproduce FeatureNet's FPN head (reference models/net.py:57-67) --
    checked in float64 torch against the layer-by-layer head of the module itself."""
    import torch.nn.functional as F
    from patchmatchnet_amd import params as PR
    from patchmatchnet_amd.net import FeatureNet
    torch.manual_seed(5)
    net = FeatureNet().double().eval()
    c4, c7, c10 = torch.randn(2, 16, 12, 16).double(), torch.randn(2, 32, 6, 8).double(), torch.randn(2, 64, 3, 4).double()
    up = lambda x: F.interpolate(x, scale_factor=2.0, mode="bilinear", align_corners=False)
    with torch.no_grad():
        ref3 = net.output1(c10)
        top = up(c10) + net.inner1(c7)
        ref2 = net.output2(top)
        ref1 = net.output3(up(top) + net.inner2(c4))
    fold = PR.fold_fpn(net.output1.weight, net.inner1.weight, net.inner1.bias, net.inner2.weight, net.inner2.bias,
                       net.output2.weight, net.output3.weight)
    assert fold[8][0].shape == (64, 112) and fold[4][0].shape == (32, 48) and fold[2][0].shape == (16, 16)
    lin = lambda x, wb: torch.einsum("bchw,cd->bdhw", x, torch.from_numpy(wb[0]).double()) + \
        torch.from_numpy(wb[1]).double()[None, :, None, None]
    l8 = lin(c10, fold[8])
    l4 = up(l8[:, 64:]) + lin(c7, fold[4])
    l2 = up(l4[:, 32:]) + lin(c4, fold[2])
    for got, ref in ((l8[:, :64], ref3), (l4[:, :32], ref2), (l2, ref1)):
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-6


def test_pack_conv_mfma_layout():
    """params.pack_conv_mfma: lane (h, i) of column block nt holds w[nt*32+i][8*c8+4h+j][ky][kx] at [tap, c8, nt, 32h+i, j]."""
    from patchmatchnet_amd import params as PR
    w = torch.arange(64 * 32 * 5 * 5, dtype=torch.float32).reshape(64, 32, 5, 5)
    p, s = PR.pack_conv_mfma(w)
    assert p.shape == (25, 4, 2, 64, 4) and s.shape == (64,) and not s.any()
    for (ky, kx, c8, nt, h, i, j) in [(0, 0, 0, 0, 0, 0, 0), (4, 3, 2, 1, 1, 17, 3), (2, 2, 3, 0, 0, 31, 1), (1, 4, 1, 1, 1, 0, 2)]:
        assert p[ky * 5 + kx, c8, nt, 32 * h + i, j] == w[nt * 32 + i, 8 * c8 + 4 * h + j, ky, kx]
    bn = (torch.full((64,), 2.0), torch.full((64,), 0.5), torch.full((64,), 0.25), torch.full((64,), 4.0 - 1e-5))
    p2, s2 = PR.pack_conv_mfma(w, bn=bn)
    np.testing.assert_allclose(p2, p, rtol=1e-6)          # scale = 2 / sqrt(4) = 1
    np.testing.assert_allclose(s2, 0.5 - 0.25, rtol=1e-6)
    # output channels are padded to a multiple of 32 with zero filters (the fused offset heads have 18..50 channels)
    p3, s3 = PR.pack_conv_mfma(w[:50], bias=torch.arange(50, dtype=torch.float32))
    assert p3.shape == (25, 4, 2, 64, 4) and s3.shape == (64,)
    np.testing.assert_array_equal(p3[:, :, 0], p[:, :, 0])
    assert not p3[:, :, 1, 18:32].any() and not p3[:, :, 1, 32 + 18:].any() and not s3[50:].any() and s3[49] == 49


def test_pack_conv_wino_layout_and_transform():
    """params.pack_conv_wino: U = G g G^T per (cout, cin), lane (j, kq) of cout block cb holds U[pos][16cc+4kq+m][16cb+j]; the
    transform reproduces a 3x3 convolution through the Winograd identity on one tile."""
    from patchmatchnet_amd import params as PR
    rng = np.random.default_rng(3)
    w = rng.standard_normal((32, 32, 3, 3))
    p, s = PR.pack_conv_wino(torch.from_numpy(w))
    assert p.shape == (2, 16, 2, 64, 4) and s.shape == (32,) and not s.any()
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    for (cc, pos, cb, j, kq, m) in [(0, 0, 0, 0, 0, 0), (1, 7, 1, 9, 3, 2), (0, 15, 1, 15, 1, 3), (1, 10, 0, 4, 2, 1)]:
        k, c = 16 * cb + j, 16 * cc + 4 * kq + m
        U = G @ w[k, c] @ G.T
        assert abs(p[cc, pos, cb, kq * 16 + j, m] - U[pos // 4, pos % 4]) < 1e-6
    d = rng.standard_normal((4, 4))
    y = At @ ((G @ w[5, 7] @ G.T) * (Bt @ d @ Bt.T)) @ At.T
    ref = np.array([[(d[a:a + 3, b:b + 3] * w[5, 7]).sum() for b in range(2)] for a in range(2)])
    np.testing.assert_allclose(y, ref, rtol=1e-12, atol=1e-12)


def test_pack_conv5x5s2_wino_reproduces_the_stride2_convolution():
    """params.pack_conv5x5s2_wino: position order (r, s, p, q), lane order, and the identity itself -- one 2x2 output tile of a
    5x5 stride-2 convolution rebuilt from the packed filter transforms and the per-phase input / output transforms."""
    from patchmatchnet_amd import params as PR
    rng = np.random.default_rng(4)
    w = rng.standard_normal((32, 16, 5, 5))
    pk, sh = PR.pack_conv5x5s2_wino(torch.from_numpy(w))
    assert pk.shape == (2, 49, 2, 64, 2) and sh.shape == (32,)
    Bt = {0: np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float), 1: np.array([[1, -1, 0], [0, 1, 0], [0, 1, -1]], float)}
    At = {0: np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float), 1: np.array([[1, 1, 0], [0, 1, -1]], float)}
    d = rng.standard_normal((7, 7, 16))  # the 7x7 input patch of one tile, all input channels
    y = np.zeros((2, 2, 32))
    pos = 0
    for r in (0, 1):
        for s in (0, 1):
            for p in range(Bt[r].shape[0]):
                for q in range(Bt[s].shape[0]):
                    v = np.einsum("a,abc,b->c", Bt[r][p], d[r::2, s::2], Bt[s][q])          # [cin]
                    U = np.zeros((16, 32))                                                    # unpack U[pos][cin][cout]
                    for cc in range(2):
                        for cb in range(2):
                            for lane in range(64):
                                for m in range(2):
                                    U[8 * cc + 2 * (lane >> 4) + m, 16 * cb + (lane & 15)] = pk[cc, pos, cb, lane, m]
                    y += np.einsum("a,b,k->abk", At[r][:, p], At[s][:, q], v @ U)
                    pos += 1
    assert pos == 49
    ref = np.array([[[(d[2 * a:2 * a + 5, 2 * b:2 * b + 5].transpose(2, 0, 1) * w[k]).sum() for k in range(32)] for b in range(2)]
                    for a in range(2)])
    np.testing.assert_allclose(y, ref, rtol=1e-5, atol=1e-5)


def _integration_stub():
    """The first ```python block of INTEGRATION.md (the binding a reference maintainer is told to paste), pointed at the built library."""
    from patchmatchnet_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\n(# --- binding.*?)```", text, flags=re.S)
    assert m, "INTEGRATION.md lost its binding block"
    code = m.group(1)
    assert '"libpmn_hip.so"' in code
    return code.replace('"libpmn_hip.so"', repr(_lib.LIB_PATH))


def test_integration_stub_executes():
    """INTEGRATION.md's ctypes binding runs against the library as built: every symbol it names exists with that arity, and the ABI
    version it asserts is the header's (round 3 shipped a stub that asserted 11 against a library at 16)."""
    from patchmatchnet_amd import _lib
    _lib.lib()
    ns = {}
    exec(compile(_integration_stub(), "INTEGRATION.md", "exec"), ns)
    assert ns["_L"].pmn_abi_version() == _lib.ABI_VERSION
    hdr = open(HEADER).read()
    assert f"#define PMN_ABI_VERSION {_lib.ABI_VERSION}" in hdr
    # the argument counts the stub declares are the header's
    for name in ("pmn_warp_correlate", "pmn_aggregate_regress", "pmn_differentiable_warping"):
        assert len(getattr(ns["_L"], name).argtypes) == len(_lib.SIGNATURES[name]), name

"""Host-side packing of the learned weights the kernels take through the kernel-argument segment.

MLP block layout (include/pmn_hip.h, PMN_MLP_FLOATS = 289 float32):
    w0[16][8] | t0[16] | w1[8][16] | t1[8] | w2[8] | b2
Row j of w0 holds its G used entries contiguously at [j*G, j*G+G) (so the kernel indexes w0[j*G+g]).
BatchNorm3d (eval mode, eps = 1e-5; reference models/module.py:43-72) is folded in float64 and rounded once:
    scale = gamma / sqrt(var + eps);  w' = w * scale;  t = beta - mean * scale
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from ._lib import MLP_FLOATS

BN_EPS = 1e-5


def _np64(t: torch.Tensor) -> np.ndarray:
    return t.detach().to("cpu", torch.float64).numpy()


def pack_mlp(conv0_w, bn0, conv1_w, bn1, last_w, last_b, eps: float = BN_EPS) -> np.ndarray:
    """bn0 / bn1 = (weight, bias, running_mean, running_var) tensors.  Returns float32[289]."""
    w0 = _np64(conv0_w).reshape(16, -1)
    G = w0.shape[1]
    assert G in (4, 8), f"unsupported group count {G}"
    w1 = _np64(conv1_w).reshape(8, 16)
    w2 = _np64(last_w).reshape(8)
    b2 = float(_np64(last_b).reshape(-1)[0])
    g0, b0, m0, v0 = (_np64(t) for t in bn0)
    g1, b1, m1, v1 = (_np64(t) for t in bn1)
    s0 = g0 / np.sqrt(v0 + eps)
    s1 = g1 / np.sqrt(v1 + eps)
    blk = np.zeros(MLP_FLOATS, np.float64)
    blk[0:16 * G] = (w0 * s0[:, None]).reshape(-1)
    blk[128:144] = b0 - m0 * s0
    blk[144:272] = (w1 * s1[:, None]).reshape(-1)
    blk[272:280] = b1 - m1 * s1
    blk[280:288] = w2
    blk[288] = b2
    return np.ascontiguousarray(blk.astype(np.float32))


def propagation_table(neighbors: int, dilation: int) -> np.ndarray:
    """(dy,dx) base offsets of adaptive propagation (reference models/patchmatch.py:331-360) as int32[2K]."""
    d = dilation
    ring = [(-d, -d), (-d, 0), (-d, d), (0, -d), (0, d), (d, -d), (d, 0), (d, d)]
    if neighbors == 4:
        t = [(-d, 0), (0, -d), (0, d), (d, 0)]
    elif neighbors == 8:
        t = ring
    elif neighbors == 16:
        t = ring + [(2 * a, 2 * b) for a, b in ring]
    else:
        raise NotImplementedError
    return np.ascontiguousarray(np.asarray(t, np.int32).reshape(-1))


def evaluation_table(neighbors: int, dilation: int) -> np.ndarray:
    """(dy,dx) base offsets of adaptive evaluation (reference models/patchmatch.py:361-392); dilation - 1."""
    d = dilation - 1
    nine = [(-d, -d), (-d, 0), (-d, d), (0, -d), (0, 0), (0, d), (d, -d), (d, 0), (d, d)]
    if neighbors == 9:
        t = nine
    elif neighbors == 17:
        t = nine + [(2 * a, 2 * b) for a, b in nine if a != 0 or b != 0]
    else:
        raise NotImplementedError
    return np.ascontiguousarray(np.asarray(t, np.int32).reshape(-1))


def versions(tensors: Sequence[torch.Tensor]) -> List[int]:
    """Cheap cache key: autograd version counters + storage addresses of the source tensors."""
    return [t._version for t in tensors] + [t.data_ptr() for t in tensors]

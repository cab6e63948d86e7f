"""Loading of the committed golden fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import os
from typing import Dict

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# case name -> (fixture file, params file, constructor kwargs)
CASES = {
    "default": ("cascade_96x128_n2.npz", "params_000007.npz",
                dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                     patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16],
                     propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])),
    "variant": ("cascade_variant_b2.npz", "params_variant.npz",
                dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                     patchmatch_iteration=[2, 1, 1], patchmatch_num_sample=[8, 8, 16],
                     propagate_neighbors=[4, 8, 16], evaluate_neighbors=[9, 17, 9])),
    # the released checkpoint under non-default hypothesis counts: D = 64 / 26 / 20 / 20 / 6 per Evaluation call -- no multiples of 4,
    # none of the HIP kernels' compile-time bounds (generic hypothesis kernel, run-time-bounded gathers, scalar aggregation)
    "counts": ("cascade_odd_counts.npz", "params_000007.npz",
               dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                    patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[6, 12, 10],
                    propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])),
    # ... and under non-default propagation ranges: other head dilations (fp32 head convolution instead of the fp16-split kernel) and
    # other fixed neighbour tables
    # a rig without the y-axis symmetry of the other fixtures (synth.general_cameras): roll, pitch, off-orbit translations, fx != fy
    "rig": ("cascade_general_rig.npz", "params_000007.npz",
            dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[6, 4, 2],
                 patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16],
                 propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])),
    "dilations": ("cascade_dilations.npz", "params_000007.npz",
                  dict(patchmatch_interval_scale=[0.005, 0.0125, 0.025], propagation_range=[5, 3, 2],
                       patchmatch_iteration=[1, 2, 2], patchmatch_num_sample=[8, 8, 16],
                       propagate_neighbors=[0, 8, 16], evaluate_neighbors=[9, 9, 9])),
}


def load_npz(name: str) -> Dict[str, np.ndarray]:
    with np.load(os.path.join(GOLDEN_DIR, name)) as z:
        return {k: z[k] for k in z.files}


def load_case(case: str):
    fx, pr, kw = CASES[case]
    return load_npz(fx), load_npz(pr), kw


def iterations_of(kw, stage: int) -> int:
    return kw["patchmatch_iteration"][stage - 1]


def stage_inputs(g: Dict[str, np.ndarray], kw, stage: int):
    """Golden inputs at the PatchMatch.forward boundary of ``stage`` (models/net.py:235-275)."""
    from oracle import oracle as O
    nv = int(g["n_views"])
    feats = [g[f"feature_{v}_s{stage}"] for v in range(nv)]
    scale = {3: 0.125, 2: 0.25, 1: 0.5}[stage]
    proj = O.stage_projections(g["intrinsics"], g["extrinsics"], scale)
    if stage == 3:
        depth, vw = None, None
    else:
        prev = stage + 1
        last = iterations_of(kw, prev)
        depth = O.nearest_up2(g[f"s{prev}_it{last}_depth_out"])
        vw = O.nearest_up2(g[f"s{prev}_it{last}_view_weights"])
    return feats, proj, depth, vw


def rel_err(a, b, floor=1e-12):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def abs_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())

#!/usr/bin/env python
"""Pin the two OpenCV calls of the reference that this repository RESTATES (no OpenCV in the authoring image, none on the GPU boxes):

    cv2.remap(depth_src, x_src, y_src, interpolation=cv2.INTER_LINEAR)        reference eval.py:129   (oracle/fusion_oracle.py:
                                                                               remap_linear_cv2; csrc/fusion.hip: remap_linear_cv2)
    cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR) on float32         reference datasets/data_io.py:26-29
                                                                               (patchmatchnet_amd/data_io.py: resize_bilinear)

Run this on ANY machine with OpenCV and numpy (no GPU, no PyTorch needed):

    python tests/golden/make_cv2_golden.py          # writes tests/golden/cv2_reference.npz (~0.3 MB)

and commit the file: tests/test_fusion_oracle.py::test_restated_opencv_calls_against_the_cv2_golden then holds both restatements to
OpenCV's own outputs (it is skipped while the fixture is absent, and SURVEY 8 row f2 stays "parity unpinned" until then -- DESIGN.md
section 5 says so).  The inputs are generated here from fixed seeds and stored next to the outputs, so the fixture is data only.

Cases, chosen for the places where a restatement can go wrong:
  remap   coordinates on exact pixel centres, on the 1/32-pixel fixed-point grid's rounding ties (k/64 offsets: OpenCV rounds map
          coordinates to 1/32 px with cvRound = round-half-to-even on the scaled value), just inside / outside every border
          (BORDER_CONSTANT 0, the default the reference uses), negative and > size coordinates, NaN-free random coordinates;
  resize  the (height, width) pairs the reference can produce with --image_max_dim on DTU / T&T / ETH3D images (down-scaling by
          non-integer factors), one up-scaling, odd sizes, and a 3-channel image."""
import os

import cv2
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20260930)
    out = {"cv2_version": np.array(cv2.__version__), "cv2_build_simd": np.array(str(cv2.checkHardwareSupport(cv2.CPU_AVX2)))}
    # ---- remap ----------------------------------------------------------------------------------------------------------------
    H, W = 37, 53
    src = (400.0 + 500.0 * rng.random((H, W))).astype(np.float32)  # depth-like values
    cases = {}
    gy, gx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cases["centres"] = (gx.copy(), gy.copy())
    for k in (1, 2, 3, 31, 32, 33, 63):  # k/64 px offsets: every second one is a tie of the 1/32-px rounding
        cases[f"offset_{k}_64"] = (gx + np.float32(k / 64.0), gy + np.float32(k / 64.0))
    cases["borders"] = (np.linspace(-1.5, W + 0.5, H * W, dtype=np.float32).reshape(H, W),
                        np.linspace(H + 0.5, -1.5, H * W, dtype=np.float32).reshape(H, W))
    cases["random"] = ((rng.random((H, W)) * (W + 4) - 2).astype(np.float32), (rng.random((H, W)) * (H + 4) - 2).astype(np.float32))
    cases["random_fine"] = ((np.round(rng.random((H, W)) * W * 64) / 64).astype(np.float32),
                            (np.round(rng.random((H, W)) * H * 64) / 64).astype(np.float32))
    out["remap_src"] = src
    for name, (mx, my) in cases.items():
        out[f"remap_{name}_x"], out[f"remap_{name}_y"] = mx, my
        out[f"remap_{name}_out"] = cv2.remap(src, mx, my, interpolation=cv2.INTER_LINEAR)
    # ---- resize ---------------------------------------------------------------------------------------------------------------
    sizes = [((48, 64), (36, 48)), ((61, 83), (45, 61)), ((50, 70), (33, 47)), ((40, 30), (56, 42)), ((97, 131), (48, 65))]
    for i, ((h0, w0), (h1, w1)) in enumerate(sizes):
        img = rng.random((h0, w0, 3)).astype(np.float32)
        out[f"resize_{i}_in"], out[f"resize_{i}_hw"] = img, np.array([h1, w1], np.int32)
        out[f"resize_{i}_out"] = cv2.resize(img, (w1, h1), interpolation=cv2.INTER_LINEAR)
    path = os.path.join(HERE, "cv2_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.2f MB" % (os.path.getsize(path) / 1e6), "with OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()

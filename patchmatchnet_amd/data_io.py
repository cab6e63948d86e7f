"""On-disk formats either side of the hot path: DTU-style camera / pair text files, images,
and the two depth-map containers eval.py writes (PFM, COLMAP .bin).  Format contract = reference datasets/data_io.py
(:88-137 cams & pairs, :165-223 .bin, :226-302 PFM); the code is vectorised numpy (the reference packs .bin through
a Python list, one struct field per pixel) and needs no OpenCV.
"""
from __future__ import annotations

import re
from typing import List, Tuple

import numpy as np
from PIL import Image


def _linear_taps(dst: int, src: int):
    """Source index pairs and float32 weights of OpenCV's resize(INTER_LINEAR) along one axis (imgproc/src/resize.cpp,
    cv::resize -> the generic ResizeFunc set-up): inv_scale = dst / src and scale = 1 / inv_scale, both in double (NOT src / dst: the
    two can differ in the last bit and move a sample that sits exactly on an integer position); fx = (float)((d + 0.5) * scale - 0.5);
    s = floor(fx); fx -= s; s < 0 -> (s, fx) = (0, 0); s >= src - 1 -> (s, fx) = (src - 1, 0); weights (1 - fx, fx) in float32."""
    scale = 1.0 / (float(dst) / float(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    f = (f - s0.astype(np.float32)).astype(np.float32)
    lo = s0 < 0
    f[lo], s0[lo] = 0.0, 0
    hi = s0 >= src - 1
    f[hi], s0[hi] = 0.0, src - 1
    s1 = np.minimum(s0 + 1, src - 1)
    return s0, s1, (np.float32(1.0) - f).astype(np.float32), f


def resize_bilinear(image: np.ndarray, height: int, width: int) -> np.ndarray:
    """cv2.resize(image, (width, height), interpolation=cv2.INTER_LINEAR) on a float32 image (reference data_io.py:26-29),
    restated from OpenCV 4.x (no OpenCV in this image; opencv-python is unpinned in the reference's requirements): half-pixel
    centres, no anti-aliasing, coordinates and weights in float32 as ``_linear_taps`` documents, the horizontal pass first
    (row[x] = S[x0]*a0 + S[x1]*a1), then the vertical one (dst = row0*b0 + row1*b1), every product and sum rounded to float32.
    PARITY WITH cv2 ITSELF IS UNPINNED: OpenCV's AVX2 / NEON builds fuse the vertical pass into an FMA, which can differ in the
    last bit; tests/test_reference_io.py pins this function to a hand-computed vector and to the reference's own read_image /
    MVSDataset flow around it.  image [H,W] or [H,W,C] float32."""
    H, W = image.shape[:2]
    image = np.asarray(image, np.float32)
    if H == 2 * height and W == 2 * width:
        # cv::resize turns INTER_LINEAR into INTER_AREA for an exact 2x down-scale ("INTER_AREA (fast) also is equal to INTER_LINEAR",
        # resize.cpp): the 2x2 block summed in float32 in the order (y,x), (y,x+1), (y+1,x), (y+1,x+1), times 0.25f -- the generic
        # ResizeAreaFast path; OpenCV's SIMD builds may associate the sum differently (unpinned, like the rest of this function)
        s4 = ((image[0::2, 0::2] + image[0::2, 1::2]) + image[1::2, 0::2]) + image[1::2, 1::2]
        return (s4 * np.float32(0.25)).astype(np.float32)
    y0, y1, b0, b1 = _linear_taps(height, H)
    x0, x1, a0, a1 = _linear_taps(width, W)
    if image.ndim == 3:
        a0, a1, b0, b1 = a0[None, :, None], a1[None, :, None], b0[:, None, None], b1[:, None, None]
    else:
        a0, a1, b0, b1 = a0[None, :], a1[None, :], b0[:, None], b1[:, None]
    rows0, rows1 = image[y0], image[y1]
    top = rows0[:, x0] * a0 + rows0[:, x1] * a1
    bot = rows1[:, x0] * a0 + rows1[:, x1] * a1
    return (top * b0 + bot * b1).astype(np.float32)


def scale_to_max_dim(image: np.ndarray, max_dim: int) -> Tuple[np.ndarray, int, int]:
    """Down-scale so that max(H,W) <= max_dim (no-op for max_dim <= 0 or when already small enough); returns the image
    and the ORIGINAL height / width (reference data_io.py:13-31)."""
    h0, w0 = image.shape[0], image.shape[1]
    scale = max_dim / max(h0, w0)
    if 0 < scale < 1:
        image = resize_bilinear(image, int(scale * h0), int(scale * w0))
    return image, h0, w0


def image_shape(filename: str, max_dim: int = -1) -> Tuple[int, int, int, int]:
    """(height, width) read_image would return for this file and max_dim, plus the original (height, width) -- from the
    image header only (no decode)."""
    with Image.open(filename) as im:
        w0, h0 = im.size
    scale = max_dim / max(h0, w0)
    if 0 < scale < 1:
        return int(scale * h0), int(scale * w0), h0, w0
    return h0, w0, h0, w0


def read_image(filename: str, max_dim: int = -1) -> Tuple[np.ndarray, int, int]:
    """RGB image as float32 in [0,1], optionally down-scaled (reference data_io.py:34-47)."""
    arr = np.array(Image.open(filename), dtype=np.float32) / 255.0
    return scale_to_max_dim(arr, max_dim)


def read_image_u8(filename: str, max_dim: int = -1):
    """The decoded RGB bytes [H,W,3] uint8 when read_image would not down-scale the file, else None.  eval.py uploads these
    (a quarter of the float32 bytes over PCIe) and divides by 255 on the device -- the same IEEE float32 division numpy does in
    read_image, so the network sees identical inputs (tests/test_eval_gpu.py)."""
    with Image.open(filename) as im:
        w0, h0 = im.size
        if 0 < max_dim / max(h0, w0) < 1:
            return None
        arr = np.array(im, dtype=np.uint8)
    return arr if arr.ndim == 3 and arr.shape[2] == 3 else None


def save_image(filename: str, image: np.ndarray) -> None:
    """bool masks -> 0/255, float images in [0,1] -> uint8, everything else cast (reference data_io.py:50-64)."""
    if image.dtype == bool:
        out = image.astype(np.uint8) * 255
    elif image.dtype in (np.float32, np.float64):
        out = (image * 255).astype(np.uint8)
    else:
        out = image.astype(np.uint8)
    Image.fromarray(out).save(filename)


def read_cam_file(filename: str) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """MVSNet camera text file: line 0 'extrinsic', lines 1-4 the 4x4 matrix, line 6 'intrinsic', lines 7-9 the 3x3
    matrix, line 11 (optional) 'depth_min depth_max ...' (reference data_io.py:88-110)."""
    with open(filename) as f:
        lines = [ln.rstrip() for ln in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    depth_params = np.array(lines[11].split(), dtype=np.float32) if len(lines) >= 12 else np.empty(0)
    return intrinsics, extrinsics, depth_params


def read_pair_file(filename: str) -> List[Tuple[int, List[int]]]:
    """pair.txt: number of viewpoints, then per viewpoint its id and 'n id score id score ...'; viewpoints without
    source views are dropped (reference data_io.py:113-131)."""
    pairs = []
    with open(filename) as f:
        n = int(f.readline())
        for _ in range(n):
            ref = int(f.readline().rstrip())
            src = [int(x) for x in f.readline().rstrip().split()[1::2]]
            if src:
                pairs.append((ref, src))
    return pairs


# ---- PFM (reference data_io.py:226-302): 'Pf' / 'PF', 'W H', scale (negative = little endian), rows bottom-up ------

def read_pfm(filename: str) -> Tuple[np.ndarray, float]:
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    channels = 3 if header == "PF" else 1
    return np.flipud(data.reshape(height, width, channels)), abs(scale)


def save_pfm(filename: str, image: np.ndarray, scale: float = 1, rows_flipped: bool = False) -> None:
    """``rows_flipped``: ``image`` already holds the rows bottom-up (the file order) -- eval.py's writer threads get the maps
    flipped on the device so that they only stream a pinned buffer to the file and never hold the GIL for a copy."""
    if image.dtype != np.float32:
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        magic = b"PF\n"
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        magic = b"Pf\n"
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    rows = np.ascontiguousarray(image if rows_flipped else np.flipud(image)).astype("<f4", copy=False)
    with open(filename, "wb") as f:
        f.write(magic)
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode("utf-8"))
        f.write(("%f\n" % -abs(scale)).encode("utf-8"))  # negative scale: little-endian payload
        rows.tofile(f)


# ---- COLMAP .bin (reference data_io.py:165-223): 'W&H&C&' then float32 in column-major (x fastest = Fortran) order ---

def read_bin(path: str) -> np.ndarray:
    with open(path, "rb") as f:
        head = b""
        while head.count(b"&") < 3:
            byte = f.read(1)
            if not byte:
                raise Exception("Malformed .bin header.")
            head += byte
        width, height, channels = (int(x) for x in head.split(b"&")[:3])
        data = np.fromfile(f, np.float32)
    return np.transpose(data.reshape((width, height, channels), order="F"), (1, 0, 2))


def save_bin(filename: str, data: np.ndarray) -> None:
    if data.dtype != np.float32:
        raise Exception("Image data type must be float32.")
    if data.ndim == 2:
        height, width = data.shape
        channels = 1
        payload = np.transpose(data, (1, 0))
    elif data.ndim == 3 and data.shape[2] in (1, 3):
        height, width, channels = data.shape
        payload = np.transpose(data, (1, 0, 2))
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    with open(filename, "wb") as f:
        f.write(f"{width}&{height}&{channels}&".encode("ascii"))
        if data.ndim == 2 and data.flags.c_contiguous:
            data.astype("<f4", copy=False).tofile(f)  # the Fortran-order flattening of the transpose IS the row-major map
        else:
            payload.reshape(-1, order="F").astype("<f4").tofile(f)


def read_map(path: str, max_dim: int = -1) -> np.ndarray:
    if path.endswith(".bin"):
        data = read_bin(path)
    elif path.endswith(".pfm"):
        data, _ = read_pfm(path)
    else:
        raise Exception("Invalid input format; only pfm and bin are supported")
    return scale_to_max_dim(data, max_dim)[0]


def save_map(path: str, data: np.ndarray, rows_flipped: bool = False) -> None:
    """``rows_flipped`` (PFM only): see save_pfm."""
    if path.endswith(".bin"):
        save_bin(path, data)
    elif path.endswith(".pfm"):
        save_pfm(path, data, rows_flipped=rows_flipped)
    else:
        raise Exception("Invalid input format; only pfm and bin are supported")

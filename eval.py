#!/usr/bin/env python
"""Depth inference + filtering + fusion on MI355X -- same command line and output files as the reference's eval.py
(flags: reference eval.py:303-347; outputs <output_folder>/<scan>/{depth_est,confidence}/<ref:08d>.{pfm,bin},
mask/*.png, fused.ply: :74-82, :257-297).

    python eval.py --input_folder DATA --checkpoint_path params_000007.ckpt --scan_list lists/dtu/test.txt \
        --num_views 5 --image_max_dim 1600 --geo_mask_thres 3 --photo_thres 0.8
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 eval.py ...     # one process per GPU

Differences from the reference, all behind its interface: the model is ``patchmatchnet_amd.PatchmatchNet`` (HIP hot
path); with several processes every scan's reference views are cut into contiguous blocks, one per rank (instead of
nn.DataParallel), each scan's maps are all-gathered over RCCL, and every rank then fuses its own block of reference views with
one HIP kernel launch per view (pmn_fuse_view) instead of numpy/cv2; rank 0 stitches the per-rank point lists into fused.ply.
Disk and PCIe traffic is taken off the critical path: decoded images go through pinned memory on a copy stream one sample
ahead, finished maps leave through pinned buffers on a second copy stream and are written by a pool of writer threads; the
forward itself is one launch-plan replay per sample (its ~55 launches recorded once, replayed from C), with --in_flight samples
overlapping on their own streams.
"""
import argparse
import concurrent.futures
import os
import queue
import sys
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

import patchmatchnet_amd as P
from patchmatchnet_amd import dist as pdist
from patchmatchnet_amd import fusion
from patchmatchnet_amd.graph import GraphedForward, PlannedForward
from patchmatchnet_amd.data_io import (image_shape, read_cam_file, read_image, read_image_u8, read_map, read_pair_file, save_image,
                                       save_map)
from patchmatchnet_amd.mvs import MVSDataset, MVSViewDataset


def print_args(args) -> None:
    print("################################  args  ################################")
    for k, v in vars(args).items():
        print("{0: <10}\t{1: <30}\t{2: <20}".format(k, str(v), str(type(v))))
    print("########################################################################")


def load_model(args, device):
    if args.input_type != "params":
        # the archive's own code is the reference's torch ops; its tensors and constructor lists build the HIP module
        print("Using scripted module from {}".format(args.checkpoint_path))
        return P.PatchmatchNet.from_scripted_module(args.checkpoint_path).to(device).eval()
    print("Evaluating model with params from {}".format(args.checkpoint_path))
    model = P.PatchmatchNet(patchmatch_interval_scale=args.patchmatch_interval_scale,
                            propagation_range=args.patchmatch_range, patchmatch_iteration=args.patchmatch_iteration,
                            patchmatch_num_sample=args.patchmatch_num_sample,
                            propagate_neighbors=args.propagate_neighbors, evaluate_neighbors=args.evaluate_neighbors)
    if args.checkpoint_path.endswith(".npz"):
        with np.load(args.checkpoint_path) as z:
            state = {k: torch.from_numpy(z[k]) for k in z.files}
    else:
        state = torch.load(args.checkpoint_path, map_location="cpu")["model"]
    model.load_state_dict(state, strict=True)  # accepts DataParallel's "module." prefix
    return model.to(device).eval()


_DISCARD_MAPS = os.environ.get("PMN_EVAL_DISCARD_MAPS", "") == "1"  # scripts/eval_bench.py: time the pipeline without the map files


class MapWriter:
    """Finished (depth, confidence) maps leave the device asynchronously: the [2,H,W] tensor is copied into a pinned host
    buffer on a side stream (ordered after the producing kernels by an event) and a writer thread saves the two files once
    the copy has landed -- the main stream never waits for PCIe or the file system (the reference synchronises and writes
    inline, eval.py:66-82; its save_bin packs a Python list per map, datasets/data_io.py:192-223).  A small pool of pinned
    buffers bounds memory and applies back-pressure.  PFM stores the rows bottom-up: the flip is done on the device, so a
    writer thread only streams its pinned buffer into the file -- it never holds the GIL for a copy, and the GIL is what the
    launch thread (55 kernel launches of Python per forward) lives on."""

    def __init__(self, device, file_format: str, workers: int = 4, buffers: int = 12) -> None:
        self.device, self.file_format = device, file_format
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=workers, thread_name_prefix="pmn-writer")
        self.stream = torch.cuda.Stream(device)
        self.free: "queue.Queue[torch.Tensor]" = queue.Queue()
        self.nbuf, self.made, self.shape = buffers, 0, None
        self.futures = []

    def _buffer(self, shape):
        if self.shape != tuple(shape):  # new map size: let the old buffers drain away
            self.shape, self.made = tuple(shape), 0
            self.free = queue.Queue()
        if self.free.empty() and self.made < self.nbuf:
            self.made += 1
            return torch.empty(shape, dtype=torch.float32).pin_memory()
        return self.free.get()

    def submit(self, stacked: torch.Tensor, depth_path: str, conf_path: str) -> None:
        buf = self._buffer(stacked.shape)
        flipped = depth_path.endswith(".pfm")
        if flipped:
            stacked = stacked.flip(1)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ready)
            buf.copy_(stacked, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        free, shape = self.free, self.shape

        def write():
            try:
                done.synchronize()
                for path, arr in ((depth_path, buf[0]), (conf_path, buf[1])):
                    if _DISCARD_MAPS:  # measurement aid (scripts/eval_bench.py --discard): everything but the file system
                        continue
                    os.makedirs(os.path.dirname(path), exist_ok=True)
                    save_map(path, arr.numpy(), rows_flipped=flipped)
            finally:
                if shape == tuple(buf.shape):
                    free.put(buf)
            return stacked  # keeps the device tensor alive until its copy has been consumed

        # finished writes are retired here (their result is the device tensor they kept alive; a failed one re-raises now)
        pending = []
        for f in self.futures:
            if f.done():
                f.result()
            else:
                pending.append(f)
        pending.append(self.pool.submit(write))
        self.futures = pending

    def drain(self) -> None:
        for f in self.futures:
            f.result()  # re-raises a writer's exception
        self.futures = []

    def close(self) -> None:
        self.drain()
        self.pool.shutdown(wait=True)


class ViewDecodeStream:
    """Every view of the encode-once groups, decoded ONCE by a pool of THREADS, in the order the samples first need them.

    Why threads: Pillow's JPEG decoder and the numpy / torch copies release the GIL, so N threads decode N images at once; worker
    PROCESSES (torch DataLoader) fork a process that holds a GPU context -- tens of milliseconds each, 18 s for the 254 workers a
    256-thread host suggests, more than a whole DTU scan takes to compute (profiles/r03_eval_bench_first.log).  A decoded image
    (uint8 [H,W,3] as the file stores it -- or float32 [H,W,3] when --image_max_dim down-scales) lands in a pinned buffer from a
    small pool (back-pressure), goes to the device on a side stream, and becomes the [1,3,H,W] float32 image of
    datasets/data_io.py:34-47 there (uint8 -> float32 / 255 with a DEVICE divisor: numpy's own float32 division, see
    DevicePrefetcher).  ``take(n)`` returns up to n consecutive views of one group that are ready (blocks for the first)."""

    def __init__(self, dataset, items, device, threads: int) -> None:
        import collections
        self.dataset, self.items, self.device = dataset, list(items), device
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(threads, 1), thread_name_prefix="pmn-decode")
        self.window = 2 * max(threads, 1) + 4          # decodes in flight / done but not consumed
        self.pending = collections.deque()             # futures in item order
        self.submitted = 0
        self.copy_stream = torch.cuda.Stream(device)
        self.scale = torch.tensor(255.0, device=device)
        self.free = {}                                 # (shape, dtype) -> list of free pinned buffers
        self.busy = []                                 # (event, buffer): pinned buffers whose upload may still be running
        self.lock = __import__("threading").Lock()

    def _buffer(self, shape, dtype):
        with self.lock:
            pool = self.free.setdefault((tuple(shape), dtype), [])
            if pool:
                return pool.pop()
        return torch.empty(tuple(shape), dtype=dtype).pin_memory()

    def _decode(self, item):  # worker thread
        from patchmatchnet_amd.data_io import read_image, read_image_u8
        _, scan, light, vid = item
        path = self.dataset.image_path(scan, light, vid)
        arr = read_image_u8(path, self.dataset.max_dim)
        if arr is None:
            arr, _, _ = read_image(path, self.dataset.max_dim)
            arr = np.ascontiguousarray(arr, np.float32)
        buf = self._buffer(arr.shape, torch.uint8 if arr.dtype == np.uint8 else torch.float32)
        np.copyto(buf.numpy(), arr)
        return buf

    def _fill(self) -> None:
        while self.submitted < len(self.items) and len(self.pending) < self.window:
            self.pending.append((self.items[self.submitted], self.pool.submit(self._decode, self.items[self.submitted])))
            self.submitted += 1

    def _recycle(self) -> None:
        still = []
        for ev, buf in self.busy:
            if ev.query():
                with self.lock:
                    self.free.setdefault((tuple(buf.shape), buf.dtype), []).append(buf)
            else:
                still.append((ev, buf))
        self.busy = still

    def take(self, max_n: int = 4):
        """[(group, view id, image [1,3,H,W] float32 on the device, the decoded [H,W,3] uint8 / float32 it was made from)], 1..max_n
        consecutive views of ONE group; the current stream is ordered after their uploads."""
        self._fill()
        self._recycle()
        if not self.pending:
            raise StopIteration
        out, group = [], self.pending[0][0][0]
        while self.pending and len(out) < max_n and self.pending[0][0][0] == group and (not out or self.pending[0][1].done()):
            item, fut = self.pending.popleft()
            buf = fut.result()  # re-raises a decode error
            with torch.cuda.stream(self.copy_stream):
                raw = buf.to(self.device, non_blocking=True)                       # [H,W,3]
                img = torch.empty((1, 3, raw.shape[0], raw.shape[1]), dtype=torch.float32, device=self.device)
                img[0].copy_(raw.permute(2, 0, 1))                                # uint8 -> float32 + HWC -> CHW in one pass
                if raw.dtype == torch.uint8:
                    img.div_(self.scale)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.busy.append((ev, buf))
            torch.cuda.current_stream(self.device).wait_event(ev)
            img.record_stream(torch.cuda.current_stream(self.device))
            out.append((item[0], item[3], img, raw))  # raw: the decoded [H,W,3] as uploaded (the fusion stage's point colours)
            self._fill()
        return out

    def close(self) -> None:
        for _, fut in self.pending:
            fut.cancel()
        self.pool.shutdown(wait=True)


class DevicePrefetcher:
    """Iterates a DataLoader one sample ahead: the next sample's tensors are copied host -> device on a side stream (from the
    loader's pinned memory) while the current one computes; the consumer's stream waits on the copy's event only."""

    def __init__(self, loader, device, keys=("images", "intrinsics", "extrinsics", "depth_min", "depth_max")) -> None:
        self.loader, self.device, self.keys = loader, device, keys
        self.stream = torch.cuda.Stream(device)
        self.scale = torch.tensor(255.0, device=device)  # a DEVICE divisor: ATen turns division by a host scalar into a
        #                                                  multiplication by the reciprocal, which is not numpy's x / 255

    def _upload(self, t):
        t = t.to(self.device, non_blocking=True)
        if t.dtype == torch.uint8:  # MVSDataset.uint8_images: decoded bytes -> the float32 image read_image returns
            t = t.to(torch.float32) / self.scale
        return t

    def _stage(self, sample):
        if "depth_min" in self.keys:
            _check_depth_range(sample)  # host numbers here, device tensors afterwards
        out = dict(sample)
        with torch.cuda.stream(self.stream):
            for k in self.keys:
                v = sample[k]
                out[k] = [self._upload(t) for t in v] if isinstance(v, (list, tuple)) else self._upload(v)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            for k in self.keys:  # the tensors were allocated on the side stream: tell the allocator who uses them
                v = cur[k]
                for t in (v if isinstance(v, (list, tuple)) else [v]):
                    t.record_stream(torch.cuda.current_stream(self.device))
            yield cur


class CameraUploader:
    """The four small per-sample tensors (intrinsics [1,N,3,3], extrinsics [1,N,4,4], depth_min [1], depth_max [1]) in ONE pinned
    buffer and ONE asynchronous host-to-device copy: four ``.to(device)`` calls on pageable memory are four synchronous copies,
    0.4 ms of the launch thread per sample (profiles/r03_eval_bench.log).  A ring of pinned buffers, each guarded by an event."""

    def __init__(self, device, slots: int = 16) -> None:
        self.device, self.slots, self.turn = device, slots, 0
        self.ring = []  # (pinned buffer, event or None)

    def __call__(self, sample):
        parts = [sample["intrinsics"].to(torch.float32), sample["extrinsics"].to(torch.float32),
                 sample["depth_min"].to(torch.float32), sample["depth_max"].to(torch.float32)]
        n = sum(p.numel() for p in parts)
        k = self.turn % self.slots
        self.turn += 1
        if k >= len(self.ring) or self.ring[k][0].numel() < n:
            entry = [torch.empty(max(n, 256), dtype=torch.float32).pin_memory(), None]
            if k >= len(self.ring):
                self.ring.append(entry)
            else:
                self.ring[k] = entry
        buf, ev = self.ring[k]
        if ev is not None:
            ev.synchronize()  # the copy that last used this slot has long finished; never a real wait
        off = 0
        for p_ in parts:
            buf[off:off + p_.numel()].copy_(p_.reshape(-1))
            off += p_.numel()
        dev = buf[:n].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.ring[k][1] = ev
        out, off = [], 0
        for p_ in parts:
            out.append(dev[off:off + p_.numel()].view(p_.shape))
            off += p_.numel()
        return out


STRICT_DEPTH_RANGE = [False]  # --strict_depth_range 1


def _check_depth_range(sample) -> None:
    """The kernels' precondition (include/pmn_hip.h): 0 < depth_min < depth_max, finite -- checked while the values are still host
    numbers (before upload).  A degenerate range makes the reference divide by zero (models/patchmatch.py:656-657) and write inf / NaN
    maps for that view while the run goes on (reference eval.py:56-82); the kernels' division sequence is IEEE for normal operands
    only.  So such a sample is run on a stand-in range and its maps are REPLACED by NaN before they are written (``_degenerate``, read
    by _write_maps): the run continues like the reference's, the view says loudly that it has no depth.  --strict_depth_range 1
    raises instead (round 5's behaviour)."""
    lo, hi = (np.atleast_1d(np.asarray(sample[k], np.float64)) for k in ("depth_min", "depth_max"))
    bad = ~(np.isfinite(lo) & np.isfinite(hi) & (lo > 0.0) & (lo < hi))
    if not bad.any():
        return
    i = int(np.argmax(bad))
    name = sample["filename"][i] if isinstance(sample["filename"], (list, tuple)) else sample["filename"]
    msg = ("{}: depth range [{}, {}] is not 0 < depth_min < depth_max (finite): line 11 of the reference view's camera file must read "
           "'depth_min depth_max'".format(name, lo[i], hi[i]))
    if STRICT_DEPTH_RANGE[0]:
        raise P.PmnError(msg)
    print("WARNING: " + msg + " -- this view's depth and confidence maps are written as NaN (the reference writes inf / NaN here)",
          file=sys.stderr, flush=True)
    for k, stand_in in (("depth_min", 1.0), ("depth_max", 2.0)):
        v = sample[k]
        if isinstance(v, torch.Tensor):
            v = v.clone()
            v.reshape(-1)[torch.from_numpy(bad)] = stand_in
        elif isinstance(v, np.ndarray):
            v = v.copy()
            v.reshape(-1)[bad] = stand_in
        else:
            v = type(v)(stand_in)
        sample[k] = v
    sample["_degenerate"] = [bool(x) for x in bad]


def _seed_sample(args, dataset, sample) -> None:
    if args.sample_seed >= 0:
        scan = sample["scan"][0] if isinstance(sample["scan"], (list, tuple)) else sample["scan"]
        torch.manual_seed(args.sample_seed + 1000003 * dataset.scan_index(scan) + int(sample["ref_view"][0]))


def _write_maps(args, sample, depth, confidence, produced, writer):
    degenerate = sample.get("_degenerate")
    for b, filename in enumerate(sample["filename"]):
        stacked = torch.stack((depth[b, 0], confidence[b]), 0)
        if degenerate is not None and degenerate[b]:
            stacked = torch.full_like(stacked, float("nan"))  # (see _check_depth_range)
        writer.submit(stacked, os.path.join(args.output_folder, filename.format("depth_est", args.file_format)),
                      os.path.join(args.output_folder, filename.format("confidence", args.file_format)))
        scan = filename.split("{}")[0].rstrip(os.sep)
        produced[(scan, int(sample["ref_view"][b]))] = stacked


def _encode_once_ok(dataset, scan, light, views):
    """The encode-once path needs every view of the group at one size that FeatureNet takes as is (multiples of 8: otherwise
    PatchmatchNet.forward resizes per sample, reference models/net.py:304-318, and the plain path reproduces that)."""
    shapes = {image_shape(dataset.image_path(scan, light, v), dataset.max_dim)[:2] for v in views}
    if len(shapes) != 1:
        return False
    h, w = next(iter(shapes))
    return h % 8 == 0 and w % 8 == 0


def save_depth(args, rank, world, device, on_scan_done=None, scan_images=None):
    """Runs the network over this rank's reference views and writes depth / confidence maps (reference eval.py:20-82).

    ``on_scan_done(scan, produced)`` is called as soon as every (scan, light) group of a scan has been inferred (--output_type both:
    the scan is fused right there and its [2,H,W] maps leave the device -- the first fused.ply appears after the first scan, and a
    rank holds one scan's maps at a time instead of the whole dataset's).

    Per-scan feature cache: every image of a scan is a source view of ~num_views other samples,
    and the reference decodes AND re-encodes it each time.  With --feature_cache > 0 every view the rank's samples read is decoded
    once and pushed through FeatureNet once (its pyramid, 53 MB per 1600x1200 view, stays on the device, channels-last) and the
    samples run from their cameras alone.  --stream_views 1 (default): ONE DataLoader over all views of all groups, in the order
    the samples first need them -- the decode workers live across scans, a sample runs as soon as ITS views are encoded (decode
    overlaps the forwards), and a pyramid is dropped after its last use; --stream_views 0: round 2's two passes per group (decode
    + encode everything, then the samples).  Same maps, bit for bit, as the plain path (tests/test_eval_gpu.py)."""
    t_stage = time.time()
    model = load_model(args, device)
    t_loaded = time.time()
    # --hip_graph 1: one launch-plan replay per sample (pmn_plan_launch: the forward's launches recorded once and re-issued from C)
    # instead of ~55 Python-issued launches -- the launch thread is what the uploads and the writer threads compete with;
    # --in_flight S: S samples in flight, each on its own HIP stream with its own replay slot, so that the gathers of one sample
    # (vector-memory pipe) share the CUs with the convolutions of the other (matrix cores).  Same maps bit for bit
    # (patchmatchnet_amd/graph.py, tests/test_eval_gpu.py).  --hip_graph 2: HIP-graph replay instead (rounds 2-5's form).
    main_stream = torch.cuda.current_stream(device)
    n_slots = max(args.in_flight, 1) if args.hip_graph else 1
    streams = [torch.cuda.Stream(device) for _ in range(n_slots)] if args.hip_graph else [main_stream]
    if args.hip_graph:
        slots = [(GraphedForward if args.hip_graph == 2 else PlannedForward)(model) for _ in range(n_slots)]
    else:
        slots = [lambda *a, **kw: model(*a, **kw)[:2]]
    turn = [0]

    def run_sample(sample_tensors, *fargs, **fkw):
        """The forward of the next sample on the next slot's stream (ordered after everything the main stream has queued, i.e.
        the sample's upload); returns that stream for the caller's own work on the outputs."""
        k = turn[0] % n_slots
        turn[0] += 1
        st = streams[k]
        if st is not main_stream:
            st.wait_stream(main_stream)
            for t in sample_tensors:
                t.record_stream(st)
        with torch.cuda.stream(st):
            return st, slots[k](*fargs, **fkw)

    dataset = MVSDataset(data_path=args.input_folder, num_views=args.num_views, max_dim=args.image_max_dim,
                         scan_list=args.scan_list, num_light_idx=args.num_light_idx).shard(rank, world)
    dataset.uint8_images = True  # 4x fewer PCIe bytes per image; DevicePrefetcher restores the float32 image on the device
    produced = {}  # (scan, ref view) -> [2,H,W] on device, kept for the per-scan gather
    done, total = 0, len(dataset)
    writer = MapWriter(device, args.file_format, workers=max(args.writer_threads, 1))
    by_scan = {}
    for (scan, light), indices in dataset.groups().items():
        by_scan.setdefault(scan, []).append((light, indices))

    def scan_finished(scan):
        if on_scan_done is not None:
            for st in streams:  # the scan's last maps are still being produced on the slot streams
                main_stream.wait_stream(st)
            on_scan_done(scan, produced)
            for key in [k for k in produced if k[0] == scan]:
                del produced[key]
            if scan_images is not None:
                for key in [k for k in scan_images if k[0] == scan]:
                    del scan_images[key]  # (the fusion job holds its own references until the scan's points are on the host)

    # ---- streaming encode-once plan: which groups qualify, their views in first-use order, one loader for all of them ----------
    group_list = [(scan, light, indices) for scan in dataset.scans for light, indices in by_scan.get(scan, [])]
    eligible, view_items = {}, []
    for gi, (scan, light, indices) in enumerate(group_list):
        ok = args.feature_cache > 0 and args.batch_size == 1 and _encode_once_ok(dataset, scan, light, dataset.views_of(indices))
        eligible[(scan, light)] = ok
        if ok and args.stream_views:
            seen = set()
            for i in indices:
                _, _, ref, src = dataset.metas[i]
                for v in [ref] + src[:min(len(src), dataset.num_views)]:
                    if v not in seen:
                        seen.add(v)
                        view_items.append((gi, scan, light, v))
    view_stream = ViewDecodeStream(dataset, view_items, device, args.decode_threads) if view_items else None
    upload_cams = CameraUploader(device)

    def run_group_streaming(gi, scan, light, indices):
        """Samples of one group from cameras only; views are pulled from the shared decode stream as the samples need them."""
        nonlocal done
        last_use = {}
        for k, i in enumerate(indices):
            _, _, ref, src = dataset.metas[i]
            for v in [ref] + src[:min(len(src), dataset.num_views)]:
                last_use[v] = k
        refs = {dataset.metas[i][2] for i in indices}
        pyramids, images = {}, {}
        dataset.load_images = False

        def samples():
            """What DataLoader(batch_size=1, num_workers=0) would yield for these indices, without its per-sample collate machinery
            (0.3 ms of the launch thread per sample at 300 samples/s): the camera tensors with a batch dimension, the rest as lists."""
            for i in indices:
                s = dataset[i]
                _check_depth_range(s)
                yield {"_degenerate": s.get("_degenerate"),
                       "intrinsics": torch.from_numpy(s["intrinsics"])[None], "extrinsics": torch.from_numpy(s["extrinsics"])[None],
                       "depth_min": torch.tensor([s["depth_min"]], dtype=torch.float64),
                       "depth_max": torch.tensor([s["depth_max"]], dtype=torch.float64), "ref_view": torch.tensor([s["ref_view"]]),
                       "view_ids": torch.from_numpy(s["view_ids"])[None], "scan": [s["scan"]], "light": [s["light"]],
                       "filename": [s["filename"]]}

        t_group, n_enc = time.time(), 0
        for k, sample in enumerate(samples()):
            start = time.time()
            ids = [int(v) for v in sample["view_ids"][0]]
            while any(v not in pyramids for v in ids):  # decode stream order = first-use order: the next views are these
                batch = view_stream.take(4)
                assert batch[0][0] == gi, "view stream out of step with the sample order"
                f = model.feature.forward_hip([img for _, _, img, _ in batch])  # 1..4 views in one FeatureNet pass
                for j, (_, v, img, raw) in enumerate(batch):
                    pyramids[v] = {s: t[j:j + 1].permute(0, 3, 1, 2) for s, t in f.items()}  # NCHW-shaped views, NHWC storage
                    if v in refs:
                        images[v] = img  # Refinement reads the reference image
                        # ... and the fusion stage takes the point colours from the same decoded bytes, when this IS the file it
                        # would read (reference eval.py:212: <scan>/images/<view>.jpg, whatever the light folder of the sample)
                        if scan_images is not None and os.path.normpath(dataset.image_path(scan, light, v)) == os.path.normpath(
                                os.path.join(args.input_folder, scan, "images/{:0>8}.jpg".format(v))):
                            scan_images[(scan, v)] = raw
                    n_enc += 1
            ref_img = images[ids[0]]
            _seed_sample(args, dataset, sample)
            cams = upload_cams(sample)
            feats = [pyramids[v] for v in ids]
            held = cams + [ref_img] + [t for f in feats for t in f.values()]  # the slot's stream reads these after we let go
            st, (depth, confidence) = run_sample(held, [ref_img] * len(ids), *cams, features=feats)
            with torch.cuda.stream(st):
                _write_maps(args, sample, depth, confidence, produced, writer)
            for v in ids:  # pyramids past their last use leave the device (record_stream above keeps them until the slot is done)
                if last_use[v] == k:
                    pyramids.pop(v, None)
                    images.pop(v, None)
            done += 1
            print("Iter {}/{}, time = {:.3f}".format(done, total, time.time() - start))
        dataset.load_images = True
        print("{}{}: {} views encoded once, {} samples, time = {:.3f}".format(scan, "/" + light if light else "", n_enc,
                                                                             len(indices), time.time() - t_group))

    def run_group(scan, light, indices):
        nonlocal done
        views = dataset.views_of(indices)
        encode_once = eligible[(scan, light)]
        if encode_once and view_stream is not None:
            return run_group_streaming(group_list.index((scan, light, indices)), scan, light, indices)
        subset = torch.utils.data.Subset(dataset, indices)
        if not encode_once:
            dataset.load_images = True
            loader = DataLoader(subset, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                                drop_last=False, pin_memory=True)
            for sample in DevicePrefetcher(loader, device):
                start = time.time()
                _seed_sample(args, dataset, sample)
                tensors = list(sample["images"]) + [sample[k] for k in ("intrinsics", "extrinsics", "depth_min", "depth_max")]
                st, (depth, confidence) = run_sample(tensors, list(sample["images"]), sample["intrinsics"],
                                                     sample["extrinsics"], sample["depth_min"], sample["depth_max"])
                with torch.cuda.stream(st):
                    _write_maps(args, sample, depth, confidence, produced, writer)
                done += len(sample["filename"])
                print("Iter {}/{}, time = {:.3f}".format(done, total, time.time() - start))
            return
        # pass 1: decode + encode every view once (FeatureNet in batches of up to 4 images)
        start = time.time()
        pyramids, images = {}, {}
        refs = {dataset.metas[i][2] for i in indices}
        vloader = DataLoader(MVSViewDataset(dataset, scan, light, views), batch_size=4, shuffle=False,
                             num_workers=args.num_workers, drop_last=False, pin_memory=True)
        for batch in DevicePrefetcher(vloader, device, keys=("image",)):
            imgs = batch["image"]
            f = model.feature.forward_hip(imgs)
            for j, v in enumerate(batch["view"].tolist()):
                pyramids[v] = {s: t[j:j + 1].permute(0, 3, 1, 2) for s, t in f.items()}  # NCHW-shaped views, NHWC storage
                if v in refs:
                    images[v] = imgs[j:j + 1]  # Refinement reads the reference image
        print("{}{}: {} views encoded once, time = {:.3f}".format(scan, "/" + light if light else "", len(views),
                                                                 time.time() - start))
        # pass 2: the samples, from cameras only
        dataset.load_images = False
        loader = DataLoader(subset, batch_size=1, shuffle=False, num_workers=0, drop_last=False)  # camera text files only
        for sample in loader:
            start = time.time()
            _check_depth_range(sample)
            ids = [int(v) for v in sample["view_ids"][0]]
            ref_img = images[ids[0]]
            _seed_sample(args, dataset, sample)
            cams = [sample[k].to(device) for k in ("intrinsics", "extrinsics", "depth_min", "depth_max")]
            st, (depth, confidence) = run_sample(cams, [ref_img] * len(ids), *cams, features=[pyramids[v] for v in ids])
            with torch.cuda.stream(st):
                _write_maps(args, sample, depth, confidence, produced, writer)
            done += 1
            print("Iter {}/{}, time = {:.3f}".format(done, total, time.time() - start))
        dataset.load_images = True
        for st in streams:  # the next group's encode pass (main stream) frees the pyramids the slots are still reading
            main_stream.wait_stream(st)
        del pyramids, images

    finished = False
    try:
        with torch.no_grad():
            # scan by scan in the scan list's order on EVERY rank: also a rank that owns no reference view of a scan (fewer views
            # than ranks) reaches scan_finished, whose fusion step holds the per-scan collective
            for scan in dataset.scans:
                for light, indices in by_scan.get(scan, []):
                    run_group(scan, light, indices)
                scan_finished(scan)
        for st in streams:
            main_stream.wait_stream(st)
        finished = True
    finally:  # an exception above must not leave decode / writer threads working through their queues behind the traceback
        if view_stream is not None:
            view_stream.close()
        if finished:
            writer.close()  # every map is on disk before anybody (fusion of another run, the caller) may read it
        else:
            writer.pool.shutdown(wait=True, cancel_futures=True)
    torch.cuda.synchronize(device)
    t_end = time.time()
    print("depth stage: {} samples in {:.3f} s after a {:.3f} s model load -> {:.1f} depth-maps/s on this rank (decode, upload, "
          "forward, download and map files included)".format(total, t_end - t_loaded, t_loaded - t_stage,
                                                             total / max(t_end - t_loaded, 1e-9)))
    args._depth_stage = (total, t_loaded)  # main(): --output_type both reports its end-to-end rate from the same starting point
    return produced


def _scan_cameras(args, scan, view_ids):
    cams, sizes = {}, {}
    for vid in view_ids:
        h, w, h0, w0 = image_shape(os.path.join(args.input_folder, scan, "images/{:0>8}.jpg".format(vid)), args.image_max_dim)
        K, E, _ = read_cam_file(os.path.join(args.input_folder, scan, "cams/{:0>8}_cam.txt".format(vid)))
        K[0] *= w / w0
        K[1] *= h / h0
        cams[vid] = {"intrinsics": K, "extrinsics": E}
        sizes[vid] = (h, w)
    return cams, sizes


class ScanFuser:
    """--output_type both: the consistency filtering + fusion of a finished scan runs on a WORKER THREAD with its own HIP stream,
    so the launch thread goes straight on to the next scan's inference (the reference fuses after all inference, single-threaded
    numpy, eval.py:193-297; round 4 fused every scan synchronously on the launch thread: 0.75 s of fusion stage behind 0.13 s of
    inference per 49-view scan).  --fuse_workers scans are in the stage at once and at most one more waits (back-pressure: a rank holds
    the maps of a few scans at most).  The per-scan collective (all-gather of the maps) stays on the launch thread -- collectives from two threads would have to
    agree on an order across ranks -- and the worker itself issues none: per-rank PLY parts are published by rename and rank 0
    stitches when all of a scan's parts exist.  An exception in the worker is re-raised on the launch thread at the next submit /
    at close."""

    def __init__(self, args, rank, world, device) -> None:
        import threading
        self.args, self.rank, self.world, self.device = args, rank, world, device
        self.jobs: "queue.Queue" = queue.Queue(maxsize=1)
        self.error = None
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(getattr(args, "fuse_threads", 8), 2), thread_name_prefix="pmn-fuse")
        self.io_pool = concurrent.futures.ThreadPoolExecutor(max_workers=8, thread_name_prefix="pmn-ply")  # fused.ply chunks: never queued
        # --fuse_workers scans are in the fusion stage at once (own stream, point packer and pinned rings each): the stage of one scan
        # is a chain of short phases (kernels, the count, downloads, encoders draining) and a second scan fills its gaps
        self.threads = [threading.Thread(target=self._run, name="pmn-fuser-%d" % i, daemon=True)
                        for i in range(max(getattr(args, "fuse_workers", 2), 1))]
        for t in self.threads:
            t.start()

    def _run(self) -> None:
        torch.cuda.set_device(self.device)
        stream, state = torch.cuda.Stream(self.device), {}  # state: buffers that live across scans (point packer, pinned rings)
        while True:
            job = self.jobs.get()
            if job is None:
                return
            if self.error is not None:
                continue  # drain: the launch thread will see the first error
            try:
                with torch.no_grad():
                    _fuse_scan_guarded(self.args, job, self.rank, self.world, self.device, stream, self.pool, self.io_pool, state)
            except BaseException as e:  # noqa: BLE001 -- handed to the launch thread
                self.error = e

    def submit(self, job) -> None:
        if self.error is not None:
            raise self.error
        self.jobs.put(job)

    def close(self) -> None:
        for _ in self.threads:
            self.jobs.put(None)
        for t in self.threads:
            t.join()
        self.pool.shutdown(wait=True)
        self.io_pool.shutdown(wait=True)
        if self.error is not None:
            raise self.error


def filter_depth(args, scan, produced, rank, world, device, fuser=None, scan_images=None):
    """Consistency filtering + fusion of one scan (reference eval.py:193-297).  The maps come from device memory (all-gathered
    across ranks) when this run produced them, else from the files a previous --output_type depth run wrote.  Every rank fuses its
    own block of reference views (pmn_fuse_view + pmn_pack_points per view); rank 0 stitches the per-rank point lists, which arrive
    in pair-file order because the blocks are contiguous, into fused.ply.  This function is the part that must run on the launch
    thread (the per-scan all-gather, the map files of views nobody produced); the rest -- ``_fuse_scan`` -- runs right here when
    ``fuser`` is None, else on the fuser's worker thread while the launch thread goes on to the next scan."""
    pairs = read_pair_file(os.path.join(args.input_folder, scan, "pair.txt"))
    ref_ids = [r for r, _ in pairs]
    view_ids = sorted(set(ref_ids) | {s for _, ss in pairs for s in ss})
    cams, sizes = _scan_cameras(args, scan, view_ids)
    # one size per scan (DTU, ETH3D): the [V,2,H,W] buffer; mixed sizes (--image_max_dim on a scan whose images differ, Tanks &
    # Temples style collections): flat slots, every view packed at its own size -- the reference likewise reads every view's maps
    # at their own size (eval.py:203-237)
    mixed = len(set(sizes.values())) != 1
    H, W = max(h for h, _ in sizes.values()), max(w for _, w in sizes.values())
    buf, slot_of = None, {}
    if produced is not None:
        mine = pdist.shard_views(ref_ids, rank, world)
        missing = [vid for vid in mine if (scan, vid) not in produced]
        if missing:
            raise P.PmnError("{}: this rank did not produce the maps of views {} it owns".format(scan, missing))
        buf, slot_of = pdist.gather_scan_buffer({vid: produced[(scan, vid)] for vid in mine}, ref_ids, H, W, device, flat=mixed)
    extra = [vid for vid in view_ids if vid not in slot_of]
    if extra:  # fusion-only run, or a source view that is nobody's reference view: read the files
        maps = []
        for vid in extra:
            d = read_map(os.path.join(args.output_folder, scan, "depth_est/{:0>8}{}".format(vid, args.file_format))).squeeze(2)
            c = read_map(os.path.join(args.output_folder, scan, "confidence/{:0>8}{}".format(vid, args.file_format))).squeeze(2)
            m = torch.from_numpy(np.stack((d, c)).astype(np.float32))
            if m.shape[1:] != sizes[vid]:
                raise P.PmnError("{}: the maps of view {} on disk are {}x{}, its image (after --image_max_dim) is {}x{}".format(
                    scan or args.input_folder, vid, m.shape[1], m.shape[2], *sizes[vid]))
            if mixed:
                m = torch.nn.functional.pad(m.reshape(-1), (0, 2 * H * W - m.numel()))
            maps.append(m)
        more = torch.stack(maps).to(device)
        base = 0 if buf is None else buf.shape[0]
        buf = more if buf is None else torch.cat((buf, more), 0)
        slot_of.update({vid: base + i for i, vid in enumerate(extra)})
    a, b = pdist.block_range(len(pairs), rank, world)
    my_pairs = pairs[a:b]
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(device))  # the gathered buffer is complete once the launch stream gets here
    job = dict(scan=scan, buf=buf, slot_of=slot_of, cams=cams, sizes=sizes, mixed=mixed, my_pairs=my_pairs, ready=ready,
               images={ref: (scan_images or {}).get((scan, ref)) for ref, _ in my_pairs})
    if fuser is not None:
        fuser.submit(job)
    else:
        with concurrent.futures.ThreadPoolExecutor(max_workers=max(getattr(args, "fuse_threads", 8), 2), thread_name_prefix="pmn-fuse") as pool, \
                concurrent.futures.ThreadPoolExecutor(max_workers=4, thread_name_prefix="pmn-ply") as io_pool:
            _fuse_scan_guarded(args, job, rank, world, device, torch.cuda.current_stream(device), pool, io_pool, {})


def _fuse_scan_guarded(args, job, rank, world, device, stream, pool, io_pool, state):
    """_fuse_scan; with several ranks a failure leaves ``fused.ply.part<rank>.err`` next to the target so that rank 0, which polls for
    the parts, stops at once instead of waiting for its timeout."""
    try:
        return _fuse_scan(args, job, rank, world, device, stream, pool, io_pool, state)
    except BaseException as e:  # noqa: BLE001 -- re-raised
        if world > 1:
            try:
                marker = os.path.join(args.output_folder, job["scan"], "fused.ply.part{}.err".format(rank))
                os.makedirs(os.path.dirname(marker), exist_ok=True)
                with open(marker, "w") as f:
                    f.write("rank {}: {}: {}".format(rank, type(e).__name__, e))
            except OSError:
                pass
        raise


def _fuse_scan(args, job, rank, world, device, stream, pool, io_pool, state):
    """One scan's reference views of this rank: fusion + point packing kernels on ``stream`` (everything stays on the device until
    the scan's PLY body is complete: ONE contiguous record buffer), the three masks of every view through a ring of pinned buffers
    to ``pool`` threads that encode the PNGs, the body in 64 MB chunks through pinned memory to pwrite -- the launch thread of this
    function only enqueues.  Same files, same bytes as the per-view host path of rounds 3-4 (tests/test_eval_gpu.py)."""
    scan, my_pairs, sizes = job["scan"], job["my_pairs"], job["sizes"]
    t_fuse = time.time()
    from patchmatchnet_amd import ops
    mask_dir = os.path.join(args.output_folder, scan, "mask")
    os.makedirs(mask_dir, exist_ok=True)

    def ref_image(r):  # a reference view this run did not decode itself (plain DataLoader path, fusion-only run)
        path = os.path.join(args.input_folder, scan, "images/{:0>8}.jpg".format(r))
        u8 = read_image_u8(path, args.image_max_dim)  # the decoded bytes when no down-scaling applies: they are the point colours
        return u8 if u8 is not None else np.ascontiguousarray(read_image(path, args.image_max_dim)[0], np.float32)

    decoding = {ref: pool.submit(ref_image, ref) for ref, _ in my_pairs if job["images"].get(ref) is None}
    fractions = {}

    def write_masks(ref, pin, ev, h, w, ring):
        try:
            ev.synchronize()
            mk = pin.numpy()[:3 * h * w].reshape(3, h, w).view(bool)  # the kernel writes 0 / 1 bytes
            for kind, m in zip(("photo", "geo", "final"), mk):
                save_image(os.path.join(mask_dir, "{:0>8}_{}.png".format(ref, kind)), m)
            # count / size = the float64 mean of a bool array the reference prints (eval.py:262-265), without the float64 pass
            fractions[ref] = tuple(np.count_nonzero(m) / m.size for m in mk)
        finally:
            ring.release(pin)

    with torch.cuda.device(device), torch.cuda.stream(stream):
        stream.wait_event(job["ready"])
        capacity = sum(sizes[ref][0] * sizes[ref][1] for ref, _ in my_pairs)
        # (15 bytes per pixel of every reference view of the scan: 1.4 GB for 49 views of 1600x1200, 9 GB for 300 views of 1920x1080.
        #  A buffer up to --fuse_buffer_mb stays with the worker for the next scan; a larger one is released when its scan is done.)
        packer = state.get("packer")
        if packer is None or packer.capacity < capacity or packer.view_counts.numel() < len(my_pairs):
            state.pop("packer", None)
            packer = state["packer"] = ops.PointPacker(max(capacity, 1), device, max_views=max(len(my_pairs), 64))
        packer.reset()
        hmax = max([sizes[ref][0] * sizes[ref][1] for ref, _ in my_pairs] or [1])
        mring = state.get("mask_ring")
        if mring is None or mring.nbytes < 3 * hmax:
            mring = state["mask_ring"] = fusion.PinnedRing(3 * hmax, 8)
        bring = state.get("body_ring")
        if bring is None:
            bring = state["body_ring"] = fusion.PinnedRing(32 << 20, 8)

        class Images(dict):  # uploads a host-decoded image the moment its view is fused
            def __missing__(self, ref):
                arr = decoding[ref].result()
                self[ref] = torch.from_numpy(arr).to(device)
                return self[ref]

        images = Images({ref: im for ref, im in job["images"].items() if im is not None})
        for im in images.values():  # decoded on the copy stream, read here by pmn_pack_points: tell the caching allocator
            if isinstance(im, torch.Tensor) and im.is_cuda:
                im.record_stream(stream)
        # 1. every view's kernels, back to back: masks and points stay on the device, nothing waits for the host
        masks_dev = []
        for ref, m in fusion.fuse_views_packed(job["buf"], job["slot_of"], job["cams"], images, my_pairs, args.geo_pixel_thres,
                                               args.geo_depth_thres, args.geo_mask_thres, args.photo_thres, packer,
                                               sizes=sizes if job["mixed"] else None):
            masks_dev.append((ref, m))
            images.pop(ref, None)
        t_enqueued = time.time()
        counts = packer.counts()  # the only synchronisation with the device: every view's number of points
        total = sum(counts)
        t_counts = time.time()
        # 2. two pipelines side by side: the masks through their pinned ring to the PNG encoders (a feeder thread: the ring blocks
        #    while eight views are being encoded), the PLY body in chunks through its ring to the pwrite threads
        pending, feeder_error = [], []

        def feed_masks():
            try:
                with torch.cuda.device(device), torch.cuda.stream(stream):
                    while masks_dev:
                        ref, m = masks_dev.pop(0)
                        h, w = sizes[ref]
                        pin = mring.acquire()
                        pin[:3 * h * w].copy_(m.reshape(-1), non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        pending.append(pool.submit(write_masks, ref, pin, ev, h, w, mring))
                        del m
            except BaseException as e:  # noqa: BLE001 -- re-raised below
                feeder_error.append(e)

        import threading
        feeder = threading.Thread(target=feed_masks, name="pmn-masks")
        feeder.start()
        ply = os.path.join(args.output_folder, scan, "fused.ply")
        target = ply if world == 1 else ply + ".part{}.tmp".format(rank)
        header = fusion.ply_header(total) if world == 1 else b""
        fd = os.open(target, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
            if header:
                os.pwrite(fd, header, 0)
            body = fusion.download_to_file(packer.records, 15 * total, fd, len(header), bring, io_pool, stream)
            t_chunks = time.time()
            for f in body:
                f.result()  # re-raises a writer's exception
            t_body = time.time()
            feeder.join()
            if feeder_error:
                raise feeder_error[0]
            for f in pending:
                f.result()
            t_masks = time.time()
        finally:
            feeder.join()
            os.close(fd)
    if 15 * capacity > (getattr(args, "fuse_buffer_mb", 4096) << 20):
        state.pop("packer", None)  # (the tensors go back to the caching allocator once the last chunk has been downloaded: above)
    for ref, _ in my_pairs:
        photo, geo, final = fractions[ref]
        print("processing {}, ref-view{:0>3}, geo_mask:{:3f}, photo_mask:{:3f}, final_mask: {:3f}".format(
            os.path.join(args.input_folder, scan), ref, geo, photo, final))
    if world > 1:
        # per-rank raw vertex records next to the target, published by rename; rank 0 writes the header for the total and appends the
        # parts in rank (= pair-file) order when all of them exist: the same bytes a single-rank run writes.  No collective here:
        # this may be a worker thread (main() removed stale parts of an earlier run behind a barrier before the first scan).
        os.replace(target, ply + ".part{}".format(rank))
        if rank == 0:
            parts = [ply + ".part{}".format(r) for r in range(world)]
            deadline = time.time() + float(getattr(args, "fuse_timeout", 600.0))
            while not all(os.path.exists(p) for p in parts):
                failed = [p + ".err" for p in parts if os.path.exists(p + ".err")]
                if failed:  # another rank's fusion stage raised: it left a marker instead of its part (see _fuse_scan_guarded)
                    raise P.PmnError("{}: the fusion stage of another rank failed: {}".format(scan, open(failed[0]).read()[:500]))
                if time.time() > deadline:
                    raise P.PmnError("{}: the point lists of the other ranks never arrived within --fuse_timeout ({})".format(scan, parts))
                time.sleep(0.005)
            n_points = sum(os.path.getsize(p) for p in parts) // 15
            with open(ply, "wb") as f:
                f.write(fusion.ply_header(n_points))
                for p in parts:
                    with open(p, "rb") as g:
                        while True:
                            chunk = g.read(64 << 20)
                            if not chunk:
                                break
                            f.write(chunk)
            for p in parts:
                os.remove(p)
    if rank == 0:
        print("saving the final model to", ply)
    print("fusion stage: {} reference views of {} in {:.3f} s on this rank ({} points; launches enqueued {:.3f}, kernels done {:.3f}, "
          "body chunks queued {:.3f}, body on disk {:.3f}, masks on disk {:.3f} s after the start)".format(
              len(my_pairs), scan or args.input_folder, time.time() - t_fuse, total, t_enqueued - t_fuse, t_counts - t_fuse,
              t_chunks - t_fuse, t_body - t_fuse, t_masks - t_fuse))


def build_parser():
    p = argparse.ArgumentParser(description="Predict depth, filter, and fuse")
    p.add_argument("--input_folder", type=str, help="input data path")
    p.add_argument("--output_folder", type=str, default="", help="output path")
    p.add_argument("--checkpoint_path", type=str, help="load a specific checkpoint for parameters of model")
    p.add_argument("--file_format", type=str, default=".pfm", help="File format for depth maps", choices=[".bin", ".pfm"])
    p.add_argument("--input_type", type=str, default="params", help="Input type of checkpoint",
                   choices=["params", "module"])
    p.add_argument("--output_type", type=str, default="both", help="Type of outputs to produce",
                   choices=["depth", "fusion", "both"])
    p.add_argument("--num_views", type=int, default=20, help="number of source views for each patch-match problem")
    p.add_argument("--image_max_dim", type=int, default=-1, help="max image dimension")
    p.add_argument("--scan_list", type=str, default="", help="Optional scan list text file to identify input folders")
    p.add_argument("--num_light_idx", type=int, default=-1, help="Number of light indexes in source images")
    p.add_argument("--batch_size", type=int, default=1, help="evaluation batch size")
    p.add_argument("--patchmatch_interval_scale", nargs="+", type=float, default=[0.005, 0.0125, 0.025],
                   help="normalized interval in inverse depth range to generate samples in local perturbation")
    p.add_argument("--patchmatch_range", nargs="+", type=int, default=[6, 4, 2],
                   help="fixed offset of sampling points for propagation of patch match on stages 1,2,3")
    p.add_argument("--patchmatch_iteration", nargs="+", type=int, default=[1, 2, 2],
                   help="num of iteration of patch match on stages 1,2,3")
    p.add_argument("--patchmatch_num_sample", nargs="+", type=int, default=[8, 8, 16],
                   help="num of generated samples in local perturbation on stages 1,2,3")
    p.add_argument("--propagate_neighbors", nargs="+", type=int, default=[0, 8, 16],
                   help="num of neighbors for adaptive propagation on stages 1,2,3")
    p.add_argument("--evaluate_neighbors", nargs="+", type=int, default=[9, 9, 9],
                   help="num of neighbors for adaptive matching cost aggregation of adaptive evaluation on stages 1,2,3")
    p.add_argument("--display", action="store_true", default=False, help="accepted for compatibility; no GUI here")
    p.add_argument("--geo_pixel_thres", type=float, default=1.0, help="pixel threshold for geometric consistency filtering")
    p.add_argument("--geo_depth_thres", type=float, default=0.01, help="depth threshold for geometric consistency filtering")
    p.add_argument("--geo_mask_thres", type=int, default=5, help="threshold for geometric consistency filtering")
    p.add_argument("--photo_thres", type=float, default=0.5, help="threshold for photometric consistency filtering")
    # additions
    p.add_argument("--num_workers", type=int, default=-1,
                   help="DataLoader worker processes per rank (the path without the feature cache); -1 = min(8, this rank's share of "
                        "the host's hardware threads)")
    p.add_argument("--decode_threads", type=int, default=-1,
                   help="JPEG decode threads per rank of the encode-once path (--feature_cache > 0, --stream_views 1); -1 = min(8, "
                        "this rank's share of the host's hardware threads)")
    p.add_argument("--strict_depth_range", type=int, default=0,
                   help="1: stop at the first camera file whose depth range is not 0 < depth_min < depth_max; 0 (default): warn, write NaN "
                        "maps for that view and go on, as the reference's run goes on (it writes inf / NaN maps there)")
    p.add_argument("--sample_seed", type=int, default=-1,
                   help=">= 0: re-seed the device RNG per sample from (this value, scan, reference view) so the stage-3 random "
                        "hypotheses -- and with them every output byte -- do not depend on how samples are ordered or sharded "
                        "(-1 = one RNG stream per process, like the reference)")
    p.add_argument("--writer_threads", type=int, default=4, help="threads writing depth / confidence maps behind the GPU")
    p.add_argument("--fuse_workers", type=int, default=2, help="--output_type both with --fuse_async 1: scans in the fusion stage at once")
    p.add_argument("--fuse_buffer_mb", type=int, default=4096,
                   help="the fusion stage's device record buffer (15 B per pixel of a scan's reference views) is kept between scans up to "
                        "this size; a larger one is released after its scan")
    p.add_argument("--fuse_timeout", type=float, default=600.0,
                   help="several ranks: seconds rank 0 waits for the other ranks' point lists of a scan before it gives up")
    p.add_argument("--fuse_threads", type=int, default=-1,
                   help="threads of the fusion stage: mask PNG encoding, reference images the run did not decode itself, fused.ply chunks")
    p.add_argument("--fuse_async", type=int, default=1,
                   help="--output_type both: 1 = a finished scan is filtered + fused on a worker thread with its own HIP stream while "
                        "the next scan's inference runs; 0 = inline on the launch thread (same files, same bytes)")
    p.add_argument("--in_flight", type=int, default=2,
                   help="samples in flight per GPU (HIP streams, one replay slot each); needs --hip_graph 1 or 2")
    p.add_argument("--hip_graph", type=int, default=1, choices=(0, 1, 2),
                   help="1: replay the forward as a launch plan (one library call per sample: its launches recorded once, re-issued "
                        "from C with plain hipLaunchKernel calls); 2: HIP-graph replay; 0: issue every kernel from Python")
    p.add_argument("--stream_views", type=int, default=1,
                   help="1: with --feature_cache, decode every view once through ONE DataLoader over all scans, in first-use order, "
                        "overlapped with the forwards; 0: two passes per scan (decode + encode all views, then the samples)")
    p.add_argument("--feature_cache", type=int, default=64,
                   help="> 0: decode and encode every view of a scan ONCE per rank and keep its FeatureNet pyramid on the device "
                        "(0 = re-decode and re-encode per sample like the reference; needs --batch_size 1)")
    return p


def _scan_names(args):
    if args.scan_list:
        if not os.path.isfile(args.scan_list):
            raise Exception("Invalid scan list file: {}".format(args.scan_list))
        with open(args.scan_list) as f:
            return [ln.rstrip() for ln in f.readlines()]
    return [""]


def main(argv=None):
    args = build_parser().parse_args(argv)
    STRICT_DEPTH_RANGE[0] = bool(args.strict_depth_range)
    print("argv: ", sys.argv[1:] if argv is None else argv)
    print_args(args)
    if args.input_folder is None or not os.path.isdir(args.input_folder):
        raise Exception("Invalid input folder: {}".format(args.input_folder))
    if args.checkpoint_path is None or not os.path.isfile(args.checkpoint_path):
        raise Exception("Invalid checkpoint file: {}".format(args.checkpoint_path))
    if not args.output_folder:
        args.output_folder = args.input_folder
    os.makedirs(args.output_folder, exist_ok=True)
    if not torch.cuda.is_available():
        raise P.PmnError("eval.py needs a ROCm GPU: the learned-PatchMatch path has no CPU fallback")
    rank, world, device = pdist.init_from_env("cuda")
    print("rank %d: %s" % (rank, pdist.bind_to_device_node(device)))  # before any pool / pinned buffer exists

    share = max((os.cpu_count() or 4) // max(world, 1), 1)  # this rank's share of the host's hardware threads
    if args.num_workers < 0:  # DataLoader worker PROCESSES (plain path): each forks a process with a GPU context -- keep them few
        args.num_workers = max(min(share - 2, 8), 2)
    if args.fuse_threads < 0:  # PNG encoders of the fusion stage (3 masks per reference view, ~7 ms each at 1600x1200)
        args.fuse_threads = max(min(share // 2, 16), 2)
    if args.decode_threads < 0:  # decode THREADS of the encode-once path
        args.decode_threads = max(min(share - 2, 8), 2)  # measured: 8 threads 256-262 depth-maps/s, 32: 242, 64: 195 (GIL)
    if _DISCARD_MAPS:  # a measurement aid (scripts/eval_bench.py), never silent: the run reports every iteration and writes no map
        print("WARNING: PMN_EVAL_DISCARD_MAPS=1 -- depth / confidence maps are computed and downloaded but NOT written to disk")
        if args.output_type != "depth":
            raise Exception("PMN_EVAL_DISCARD_MAPS=1 is a timing aid for --output_type depth only")
    if args.output_type != "depth" and world > 1:
        # per-rank point lists are published as <scan>/fused.ply.part<rank> (see _fuse_scan): parts an interrupted earlier run left
        # behind must be gone before anybody starts waiting for this run's
        for scan in _scan_names(args):
            for name in ("fused.ply.part{}".format(rank), "fused.ply.part{}.tmp".format(rank), "fused.ply.part{}.err".format(rank)):
                stale = os.path.join(args.output_folder, scan, name)
                if os.path.exists(stale):
                    os.remove(stale)
        torch.distributed.barrier()
    if args.output_type == "depth":
        save_depth(args, rank, world, device)
    elif args.output_type == "both":
        # every scan is fused as soon as its maps exist, straight from device memory (the per-scan all-gather), on a worker thread
        # with its own stream while the launch thread infers the next scan (--fuse_async 0: inline, rounds 3-4's order)
        fuser = ScanFuser(args, rank, world, device) if args.fuse_async else None
        scan_images = {}
        try:
            save_depth(args, rank, world, device, scan_images=scan_images,
                       on_scan_done=lambda scan, produced: filter_depth(args, scan, produced, rank, world, device, fuser=fuser,
                                                                        scan_images=scan_images))
        except BaseException:
            if fuser is not None:
                try:
                    fuser.close()
                except BaseException:  # noqa: BLE001 -- the first error is the one to report
                    pass
            raise
        if fuser is not None:
            t_wait = time.time()
            fuser.close()  # the last scan(s) may still be in the fusion stage
            print("fusion of the last scan(s) finished {:.3f} s after the depth stage".format(time.time() - t_wait))
        n_maps, t_loaded = args._depth_stage
        print("both stages: {} depth maps inferred, filtered and fused in {:.3f} s -> {:.1f} depth-maps/s on this rank (masks and "
              "fused.ply on disk; model load excluded)".format(n_maps, time.time() - t_loaded, n_maps / max(time.time() - t_loaded, 1e-9)))
    else:
        for scan in _scan_names(args):
            filter_depth(args, scan, None, rank, world, device)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

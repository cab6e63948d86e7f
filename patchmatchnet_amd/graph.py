"""HIP-graph replay of PatchmatchNet.forward.

One forward is ~55 kernel launches issued from Python (~3.5 ms of interpreter time at 1600x1200, N=5 -- as long as the kernels
themselves run).  Forward-only that cost hides behind the GPU; in a pipeline whose launch thread also stages uploads and hands
finished maps to writer threads it is the bottleneck (scripts/pipeline_bench.py).  ``GraphedForward`` captures the forward once
per input signature into a HIP graph (torch.cuda.CUDAGraph over the ctypes launches: every entry point of libpmn_hip.so only
enqueues on the current stream, so the library is capturable as is) and replays it: one launch per sample, the interpreter is
free for the I/O around it.

Same results as the eager call, bit for bit, including the stage-3 random draw (reference models/patchmatch.py:61-62): the draw
is NOT part of the graph.  Every call draws ``torch.rand(size=(B,48,H/8,W/8))`` eagerly on the caller's stream into the slot's
static noise buffer -- the same call, at the same point of the generator's stream, as the eager forward makes -- and the
captured forward reads that buffer (``noise=``).  A captured Philox kernel would instead re-read the generator's ONE pair of
seed / offset device tensors at replay time; with several slots replaying on different streams, slot B's ``replay()`` refills that
pair while slot A's draw may still be pending, and A would silently draw with B's offset.  The eager draw takes its
seed / offset on the host at call time, so ``torch.manual_seed(s)`` followed by a call draws what the eager forward draws after
``torch.manual_seed(s)`` whatever else is in flight (tests/test_eval_gpu.py).

Reference: models/net.py:176-301 (PatchmatchNet.forward) is what one graph holds; eval.py:56-64 is the loop that replays it.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ._lib import PmnError


class GraphedForward:
    """``GraphedForward(model)(images, intrinsics, extrinsics, depth_min, depth_max, features=None)`` -> (depth, confidence).

    The returned tensors are the graph's static outputs: consume (copy) them before the next call with the same signature
    overwrites them.  Inputs whose sizes PatchmatchNet.forward would adjust (height / width not multiples of 8, reference
    net.py:304-318) run eagerly -- that path resizes the images and rewrites the caller's intrinsics in place.

    ``inputs_in_place=True``: the replayed FeatureNet reads the caller's image tensors WHERE THEY ARE, through a device table of
    addresses rewritten per call (pmn_stem_f16s_views), instead of from copies in the slot's static buffers (six 23 MB copies per
    1600x1200 sample; only the reference image, which Refinement reads, is still copied).  The caller then has to leave the images
    alone until the replay has run -- not only until this call returns: for tensors that live as long as the scan (bench.py's resident
    samples) that is free; a pipeline that recycles its upload buffers keeps the default.  An image the kernel cannot read in place
    (not dense float32, or not 16-byte aligned) is copied into the slot and the table points there."""

    def __init__(self, model, max_graphs: int = 8, inputs_in_place: bool = False) -> None:
        self.model, self.max_graphs, self.inputs_in_place = model, max_graphs, inputs_in_place
        self.cache: Dict[Tuple, Tuple] = {}
        self.replays = self.captures = self.evictions = 0

    @staticmethod
    def _alias_pattern(images: Sequence[torch.Tensor]) -> Tuple[int, ...]:
        """For every image the index of the first image that is the same tensor (eval.py's encode-once path passes the reference
        image N+1 times: FeatureNet is skipped and only Refinement reads an image)."""
        first: Dict[int, int] = {}
        return tuple(first.setdefault(im.data_ptr(), i) for i, im in enumerate(images))

    @classmethod
    def _signature(cls, images: Sequence[torch.Tensor], intrinsics: torch.Tensor, features) -> Tuple:
        feat = None if features is None else tuple(tuple((s, tuple(t.shape)) for s, t in sorted(f.items())) for f in features)
        return (tuple(tuple(i.shape) for i in images), cls._alias_pattern(images), tuple(intrinsics.shape), feat,
                features is not None and cls._table_ok_cached(features))

    _TABLE_OK: Dict[Tuple, bool] = {}

    @classmethod
    def _table_ok_cached(cls, features) -> bool:
        """``_table_ok`` once per (shape, stride) pattern of the injected maps instead of ~18 permute / is_contiguous checks per sample on
        the launch thread (the answer depends on layout only, not on which scan's pyramids these are)."""
        try:
            key = tuple((s, tuple(t.shape), tuple(t.stride()), t.dtype, t.is_cuda) for f in features for s, t in sorted(f.items()))
        except (AttributeError, TypeError):
            return cls._table_ok(features)
        hit = cls._TABLE_OK.get(key)
        if hit is None:
            if len(cls._TABLE_OK) > 64:
                cls._TABLE_OK.clear()
            hit = cls._TABLE_OK[key] = cls._table_ok(features)
        return hit

    def _images_in_place(self, features) -> bool:
        """The replayed FeatureNet can read the images through a table: asked for, FeatureNet is part of the graph, and its stem is the
        fp16-split kernel (pmn_stem_f16s_views; a checkpoint outside the split's domain falls back to the fp32 stem, which has no
        table form -- then the images are copied as usual)."""
        feature = getattr(self.model, "feature", None)
        if not (self.inputs_in_place and features is None and getattr(self.model, "hip_feature_net", False)
                and feature is not None and feature.f16_split):
            return False
        return "conv1_f16s" in feature._packed()

    def _capture(self, images, intrinsics, extrinsics, depth_min, depth_max, features):
        dev = intrinsics.device
        pattern = self._alias_pattern(images)
        in_place = self._images_in_place(features)
        # (in place: only the reference image gets a static buffer up front; another view gets one the first time it cannot be read
        # where it is, _fill)
        bufs = [torch.empty_like(im) if pattern[i] == i and not (in_place and i) else None for i, im in enumerate(images)]
        static = dict(images=[bufs[pattern[i]] for i in range(len(images))], intrinsics=torch.empty_like(intrinsics),
                      extrinsics=torch.empty_like(extrinsics), depth_min=torch.empty_like(depth_min),
                      depth_max=torch.empty_like(depth_max),
                      noise=torch.empty((images[0].shape[0], 48, images[0].shape[2] // 8, images[0].shape[3] // 8),
                                        dtype=torch.float32, device=dev),
                      features=None, features_nhwc=None)
        if in_place:
            from . import ops
            static["images"] = [bufs[0]] * len(images)  # shapes for the forward; FeatureNet reads through the table, Refinement entry 0
            static["image_bufs"] = bufs
            static["table_all"] = torch.zeros(len(images), dtype=torch.int64, device=dev)
            static["image_table"] = ops.SourceTable(static["table_all"], (len(images),) + tuple(images[0].shape))
            static["table_host"] = []
        if features is not None and self._table_ok(features):
            # injected pyramids that are channels-last maps of their own (eval.py's encode-once path): only the REFERENCE view's
            # pyramid is copied into static buffers (offset heads, FeatureWeightNet and the cascade read it); the SOURCE views stay
            # where FeatureNet wrote them and the captured pmn_warp_correlate_views launches find them through static device tables of
            # addresses that _fill rewrites per sample (round 3 copied 0.53 GB of source pyramids per 1600x1200 sample into place)
            from . import ops
            B = images[0].shape[0]
            stages = sorted(features[0])
            ref = {s: torch.empty((B,) + tuple(features[0][s].shape[2:]) + (features[0][s].shape[1],), dtype=torch.float32, device=dev)
                   for s in stages}
            static["ref_nhwc"] = ref
            n_src = len(features) - 1
            static["table_all"] = torch.zeros(len(stages) * n_src, dtype=torch.int64, device=dev)  # [stage-major][view]
            static["tables"] = {s: ops.SourceTable(static["table_all"][i * n_src:(i + 1) * n_src],
                                                   (n_src, B) + tuple(features[1][s].shape[2:]) + (features[1][s].shape[1],))
                                for i, s in enumerate(stages)}
            # (the source entries only carry the count and shapes of the views: stand-ins, so that no sample's pyramids are kept alive)
            static["features"] = [{s: ref[s].permute(0, 3, 1, 2) for s in stages} for _ in range(len(features))]
            static["table_host"] = []  # ring of pinned staging buffers, each guarded by an event
        elif features is not None:
            # injected pyramids (eval.py's encode-once path): ONE channels-last buffer per stage holds all views, view-major -- the
            # layout the kernels read the source views from -- and the per-view NCHW-shaped tensors handed to forward() are views of
            # it: a sample's pyramids are copied once, into place (round 2 copied them into per-view buffers here and the forward
            # stacked them again: 580 MB instead of 320 MB of device copies per 1600x1200 sample)
            B = images[0].shape[0]
            stages = sorted(features[0])
            nhwc = {s: torch.empty((len(features) * B,) + tuple(features[0][s].shape[2:]) + (features[0][s].shape[1],),
                                   dtype=torch.float32, device=dev) for s in stages}
            static["features_nhwc"] = nhwc
            static["features"] = [{s: nhwc[s][v * B:(v + 1) * B].permute(0, 3, 1, 2) for s in stages} for v in range(len(features))]
        self._fill(static, images, intrinsics, extrinsics, depth_min, depth_max, features)

        def run():
            return self.model(list(static["images"]), static["intrinsics"], static["extrinsics"], static["depth_min"],
                              static["depth_max"], features=static["features"], features_nhwc=static["features_nhwc"],
                              noise=static["noise"], source_tables=static.get("tables"), ref_nhwc_maps=static.get("ref_nhwc"),
                              image_table=static.get("image_table"))

        rng = torch.cuda.get_rng_state(dev)  # warm-up and capture must not advance the caller's random stream
        self._draw(static)
        try:
            handle, out = self._record(run, dev)
        finally:
            torch.cuda.set_rng_state(rng, dev)
        return handle, static, out

    def _record(self, run, dev):
        """``run`` (the forward over the slot's static buffers) -> (replay handle, (depth, confidence) static outputs)."""
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # one eager pass off the capture: lazy kernel attributes, weight packing, allocator warm-up
            run()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        # thread_local: eval.py's writer threads, the DataLoader's pin-memory thread and RCCL's watchdog keep calling into the
        # runtime (event waits, pinned allocations) while this thread captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            depth, confidence, _ = run()
        return graph, (depth, confidence)

    def _replay(self, handle, dev) -> None:
        handle.replay()

    @staticmethod
    def _table_ok(features) -> bool:
        """Every injected map is an NCHW-shaped view of a dense channels-last tensor of one size per stage (what FeatureNet.forward_hip
        hands out): then the source views can be read in place through address tables."""
        try:
            for s in features[0]:
                shapes = {tuple(f[s].shape) for f in features}
                if len(shapes) != 1:
                    return False
                for f in features:
                    t = f[s]
                    if not (t.is_cuda and t.dtype == torch.float32 and t.permute(0, 2, 3, 1).is_contiguous()):
                        return False
            return len(features) >= 2
        except (AttributeError, KeyError, TypeError):
            return False

    @staticmethod
    def _draw(static) -> None:
        """The stage-3 draw of this sample, eagerly on the current stream: torch.rand with the eager forward's size and dtype, so
        the default generator advances exactly as it does there (models/patchmatch.py:61-62)."""
        torch.rand(size=tuple(static["noise"].shape), out=static["noise"])

    @staticmethod
    def _stage_addresses(static, addrs) -> None:
        """The slot's device tables (``static["table_all"]``: one int64 tensor, the per-stage tables are slices of it) <- ``addrs``,
        through a ring of pinned host buffers, each guarded by an event (the copy is asynchronous: a buffer must not be rewritten
        before it has been read).  One numpy assignment and one copy per sample."""
        ring = static["table_host"]
        k = static["turn"] = (static.get("turn", -1) + 1) % 8
        if k >= len(ring):
            host = torch.empty(static["table_all"].numel(), dtype=torch.int64).pin_memory()
            ring.append([host, host.numpy(), None])
        host, host_np, ev = ring[k]
        if ev is not None:
            ev.synchronize()  # eight samples back: long done
        host_np[:] = addrs
        static["table_all"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(static["intrinsics"].device))
        ring[k][2] = ev

    @classmethod
    def _fill(cls, static, images, intrinsics, extrinsics, depth_min, depth_max, features) -> None:
        if "image_table" in static:
            from . import ops
            bufs, addrs, held = static["image_bufs"], [], []
            for i, im in enumerate(images):
                if i == 0 or im.device != static["intrinsics"].device or not ops.SourceTable.image_in_place(im):
                    if bufs[i] is None:
                        bufs[i] = torch.empty(tuple(images[0].shape), dtype=torch.float32, device=static["intrinsics"].device)
                    if bufs[i].data_ptr() != im.data_ptr():
                        bufs[i].copy_(im, non_blocking=True)
                    addrs.append(bufs[i].data_ptr())
                else:
                    # read WHERE IT IS by the replay about to be enqueued on the current stream: tell the caching allocator (a caller
                    # that drops the image right after this call must not get its block handed to another stream before the stem has
                    # read it), and keep the tensor until the slot's next sample replaces it
                    im.record_stream(torch.cuda.current_stream(im.device))
                    held.append(im)
                    addrs.append(im.data_ptr())
            static["held_images"] = held
            cls._stage_addresses(static, addrs)
        else:
            done = set()
            for dst, src in zip(static["images"], images):  # aliased inputs share one static buffer: copied once
                if dst.data_ptr() not in done and dst.data_ptr() != src.data_ptr():
                    dst.copy_(src, non_blocking=True)
                done.add(dst.data_ptr())
        static["intrinsics"].copy_(intrinsics, non_blocking=True)
        static["extrinsics"].copy_(extrinsics, non_blocking=True)
        static["depth_min"].copy_(depth_min, non_blocking=True)
        static["depth_max"].copy_(depth_max, non_blocking=True)
        if features is not None and "tables" in static:
            for s, t in features[0].items():
                static["features"][0][s].copy_(t, non_blocking=True)
            # (the NCHW-shaped view and its channels-last storage start at the same address)
            cls._stage_addresses(static, [features[1 + v][s].data_ptr() for s in sorted(static["tables"]) for v in range(len(features) - 1)])
        elif features is not None:
            for dst, src in zip(static["features"], features):
                for s, t in src.items():
                    dst[s].copy_(t, non_blocking=True)

    def __call__(self, images: List[torch.Tensor], intrinsics: torch.Tensor, extrinsics: torch.Tensor,
                 depth_min: torch.Tensor, depth_max: torch.Tensor, features: Optional[List[Dict[int, torch.Tensor]]] = None):
        if not intrinsics.is_cuda:
            raise PmnError("GraphedForward replays HIP graphs: the inputs must be on a ROCm device")
        h, w = images[0].shape[-2:]
        if h % 8 or w % 8 or any(tuple(i.shape) != tuple(images[0].shape) for i in images):
            depth, confidence, _ = self.model(images, intrinsics, extrinsics, depth_min, depth_max, features=features)
            return depth, confidence
        key = self._signature(images, intrinsics, features)
        entry = self.cache.pop(key, None)
        if entry is None:
            # least-recently-USED eviction (a hit re-inserts its key at the end): a scan that alternates between a few image sizes
            # keeps all of them; ``captures`` lets a caller see a workload that defeats the cache (one capture per call)
            if len(self.cache) >= self.max_graphs:
                self.cache.pop(next(iter(self.cache)))
                self.evictions += 1
            entry = self._capture(images, intrinsics, extrinsics, depth_min, depth_max, features)
            self.captures += 1
        self.cache[key] = entry
        graph, static, out = entry
        self._fill(static, images, intrinsics, extrinsics, depth_min, depth_max, features)
        self._draw(static)
        self._replay(graph, intrinsics.device)
        self.replays += 1
        return out


class _LibraryLaunchesOnly(torch.utils._python_dispatch.TorchDispatchMode):
    """Active while a forward is being recorded into a launch plan: every ATen operator the forward dispatches is checked.  Views and
    allocations are fine -- they launch nothing --; anything else would run ONCE, now, on uninitialised inputs, and be absent from
    every replay: a silently wrong plan.  So it raises, naming the operator."""

    _ALLOC = ("aten.empty.", "aten.empty_strided.", "aten.empty_like.", "aten.new_empty.", "aten.new_empty_strided.", "aten.detach.",
              "aten.alias.", "aten.lift_fresh.", "aten._unsafe_view.", "aten.reshape.", "aten.view.", "aten.sym_size.",
              "aten.sym_stride.", "aten.sym_numel.", "aten.sym_storage_offset.", "aten.is_contiguous.", "aten.size.", "aten.stride.",
              "aten.numel.", "aten.dim.", "aten.is_pinned.", "prim.device.", "prim.dtype.", "prim.layout.", "aten.record_stream.")

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        if not (getattr(func, "is_view", False) or (str(func) + ".").startswith(self._ALLOC)):
            raise PmnError(f"PlannedForward: the forward dispatched {func} while it was being recorded -- an operator outside "
                           "libpmn_hip.so cannot be part of a launch plan (it would run once, on uninitialised data, and never "
                           "again); this input signature has to run eagerly or through a path that only calls the library")
        return func(*args, **(kwargs or {}))


class PlannedForward(GraphedForward):
    """GraphedForward's interface and static-buffer machinery, with the forward recorded as a LAUNCH PLAN (include/pmn_hip.h,
    pmn_plan_*; csrc/plan.hip) instead of a HIP graph: the recording pass runs the forward once while the calling thread's entry points
    append their launches to the plan instead of enqueuing them, and a replay is one pmn_plan_launch -- a C loop of plain
    hipLaunchKernel calls on the current stream, ~55 per forward, no interpreter in between.

    Why a plan and not the graph: a plan is a page of C (csrc/plan.hip) instead of a runtime feature -- no capture mode, no private
    graph memory pools inside the runtime, no dependence on how a ROCm release replays graphs --, its replay rate is the graph's
    (bench.py --launch graph vs plan: 373-377 vs 376-384 depth-maps/s, round 6) and its contents can be listed (kernel_names()).
    (Round 5 blamed HIP-graph replay for overlapped forwards that differed from the eager forward; round 6 found the cause in an
    instruction form of this library's own gather kernels -- DESIGN_LESSONS.md lesson 46 -- and both replay forms are bit-exact on any number of hardware queues
    since: tests/test_plan_gpu.py, tests/test_overlap_gpu.py, bench.py's outputs_verified.)

    What a plan needs from the forward: every launch comes from libpmn_hip.so (the recording pass raises on any other ATen operator,
    see _LibraryLaunchesOnly) and every buffer it touches stays in place: the pass allocates from a private torch memory pool that
    lives as long as the plan, exactly like a graph's private pool."""

    def __init__(self, model, max_graphs: int = 8, inputs_in_place: bool = False) -> None:
        super().__init__(model, max_graphs=max_graphs, inputs_in_place=inputs_in_place)

    def _record(self, run, dev):
        import ctypes
        from . import _lib
        L = _lib.lib()
        run()  # one eager pass: lazy kernel attributes, weight packing (their ATen work must not fall into the recording pass)
        with torch.cuda.device(dev):
            pool = torch.cuda.MemPool()
            plan = ctypes.c_void_p()
            _lib.check(L.pmn_plan_create(ctypes.byref(plan)), "pmn_plan_create")
            handle = _Plan(plan, pool)
            with torch.cuda.use_mem_pool(pool, device=dev):
                _lib.check(L.pmn_plan_begin(plan), "pmn_plan_begin")
                try:
                    with _LibraryLaunchesOnly():
                        depth, confidence, _ = run()
                finally:
                    rc = L.pmn_plan_end(plan)
                _lib.check(rc, "pmn_plan_end")
        handle.count = L.pmn_plan_count(plan)
        if handle.count <= 0:
            raise PmnError("PlannedForward: the recording pass recorded no launch")
        # (the recording pass launched nothing: depth / confidence are uninitialised until the first replay, which __call__ issues
        # right after a capture)
        return handle, (depth, confidence)

    def _replay(self, handle, dev) -> None:
        from . import _lib
        with torch.cuda.device(dev):  # (plain launches go to the CURRENT device's context: a rank whose device is not the process default)
            _lib.check(_lib.lib().pmn_plan_launch(handle.plan, torch.cuda.current_stream(dev).cuda_stream), "pmn_plan_launch")


class _Plan:
    """Owns one pmn_plan and the torch memory pool its recorded addresses live in."""

    def __init__(self, plan, pool) -> None:
        self.plan, self.pool, self.count = plan, pool, 0

    def kernel_names(self) -> List[str]:
        from . import _lib
        L = _lib.lib()
        return [(L.pmn_plan_kernel_name(self.plan, i) or b"?").decode() for i in range(self.count)]

    def __del__(self):
        try:
            from . import _lib
            if self.plan:
                _lib.lib().pmn_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

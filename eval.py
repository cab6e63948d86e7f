#!/usr/bin/env python
"""Depth inference + filtering + fusion on MI355X -- same command line and output files as the reference's eval.py
(flags: reference eval.py:303-347; outputs <output_folder>/<scan>/{depth_est,confidence}/<ref:08d>.{pfm,bin},
mask/*.png, fused.ply: :74-82, :257-297).

    python eval.py --input_folder DATA --checkpoint_path params_000007.ckpt --scan_list lists/dtu/test.txt \
        --num_views 5 --image_max_dim 1600 --geo_mask_thres 3 --photo_thres 0.8
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 eval.py ...     # one process per GPU

Differences from the reference, all behind its interface: the model is ``patchmatchnet_amd.PatchmatchNet`` (HIP hot
path); with several processes the reference views are sharded round-robin across ranks (instead of nn.DataParallel)
and each scan's maps are all-gathered over RCCL for the fusion step, which runs on the device instead of in numpy/cv2.
"""
import argparse
import collections
import os
import sys
import time

import numpy as np
import torch
from torch.utils.data import DataLoader

import patchmatchnet_amd as P
from patchmatchnet_amd import dist as pdist
from patchmatchnet_amd import fusion
from patchmatchnet_amd.data_io import image_shape, read_cam_file, read_image, read_map, read_pair_file, save_image, save_map
from patchmatchnet_amd.mvs import MVSDataset, MVSViewDataset


def print_args(args) -> None:
    print("################################  args  ################################")
    for k, v in vars(args).items():
        print("{0: <10}\t{1: <30}\t{2: <20}".format(k, str(v), str(type(v))))
    print("########################################################################")


def load_model(args, device):
    if args.input_type != "params":
        raise Exception("--input_type module (TorchScript archive of the reference implementation) cannot carry the HIP "
                        "path; pass the params checkpoint instead")
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


def _write_maps(args, sample, depth, confidence, produced):
    depth_np = depth.detach().cpu().numpy()
    conf_np = confidence.detach().cpu().numpy()
    for b, filename in enumerate(sample["filename"]):
        for kind, arr in (("depth_est", depth_np[b, 0]), ("confidence", conf_np[b])):
            path = os.path.join(args.output_folder, filename.format(kind, args.file_format))
            os.makedirs(os.path.dirname(path), exist_ok=True)
            save_map(path, np.ascontiguousarray(arr))
        scan = filename.split("{}")[0].rstrip(os.sep)
        produced[(scan, int(sample["ref_view"][b]))] = torch.stack((depth[b, 0], confidence[b]), 0)


def _encode_once_ok(dataset, scan, light, views):
    """The encode-once path needs every view of the group at one size that FeatureNet takes as is (multiples of 8: otherwise
    PatchmatchNet.forward resizes per sample, reference models/net.py:304-318, and the plain path reproduces that)."""
    shapes = {image_shape(dataset.image_path(scan, light, v), dataset.max_dim)[:2] for v in views}
    if len(shapes) != 1:
        return False
    h, w = next(iter(shapes))
    return h % 8 == 0 and w % 8 == 0


def save_depth(args, rank, world, device):
    """Runs the network over this rank's reference views and writes depth / confidence maps (reference eval.py:20-82).

    Per-scan feature cache: every image of a scan is a source view of ~num_views other samples,
    and the reference decodes AND re-encodes it each time.  With --feature_cache > 0 a (scan, light) group is processed in two
    passes: every view the rank's samples read is decoded once and pushed through FeatureNet once (its pyramid, 53 MB per
    1600x1200 view, stays on the device, channels-last); then the samples are run from their cameras alone.  Same maps, bit for
    bit, as the plain path (tests/test_eval_gpu.py)."""
    model = load_model(args, device)
    dataset = MVSDataset(data_path=args.input_folder, num_views=args.num_views, max_dim=args.image_max_dim,
                         scan_list=args.scan_list, num_light_idx=args.num_light_idx).shard(rank, world)
    produced = {}  # (scan, ref view) -> [2,H,W] on device, kept for the per-scan gather
    done, total = 0, len(dataset)
    with torch.no_grad():
        for (scan, light), indices in dataset.groups().items():
            views = dataset.views_of(indices)
            encode_once = args.feature_cache > 0 and args.batch_size == 1 and _encode_once_ok(dataset, scan, light, views)
            subset = torch.utils.data.Subset(dataset, indices)
            if not encode_once:
                dataset.load_images = True
                loader = DataLoader(subset, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers,
                                    drop_last=False)
                for sample in loader:
                    start = time.time()
                    depth, confidence, _ = model([im.to(device) for im in sample["images"]], sample["intrinsics"].to(device),
                                                 sample["extrinsics"].to(device), sample["depth_min"].to(device),
                                                 sample["depth_max"].to(device))
                    _write_maps(args, sample, depth, confidence, produced)
                    done += len(sample["filename"])
                    print("Iter {}/{}, time = {:.3f}".format(done, total, time.time() - start))
                continue
            # pass 1: decode + encode every view once (FeatureNet in batches of up to 4 images)
            start = time.time()
            pyramids, images = {}, {}
            refs = {dataset.metas[i][2] for i in indices}
            vloader = DataLoader(MVSViewDataset(dataset, scan, light, views), batch_size=4, shuffle=False,
                                 num_workers=args.num_workers, drop_last=False)
            for batch in vloader:
                imgs = batch["image"].to(device)
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
                ids = [int(v) for v in sample["view_ids"][0]]
                ref_img = images[ids[0]]
                depth, confidence, _ = model([ref_img] * len(ids), sample["intrinsics"].to(device),
                                             sample["extrinsics"].to(device), sample["depth_min"].to(device),
                                             sample["depth_max"].to(device), features=[pyramids[v] for v in ids])
                _write_maps(args, sample, depth, confidence, produced)
                done += 1
                print("Iter {}/{}, time = {:.3f}".format(done, total, time.time() - start))
            dataset.load_images = True
    return produced


def filter_depth(args, scan, produced, rank, world, device):
    """Consistency filtering + fusion of one scan (reference eval.py:193-297).  Maps come from device memory (all-gathered
    across ranks) when this process produced them, else from the files a previous --output_type depth run wrote."""
    pairs = read_pair_file(os.path.join(args.input_folder, scan, "pair.txt"))
    view_ids = sorted({r for r, _ in pairs} | {s for _, ss in pairs for s in ss})
    views = {}
    for vid in view_ids:
        img, h0, w0 = read_image(os.path.join(args.input_folder, scan, "images/{:0>8}.jpg".format(vid)),
                                 args.image_max_dim)
        K, E, _ = read_cam_file(os.path.join(args.input_folder, scan, "cams/{:0>8}_cam.txt".format(vid)))
        K[0] *= img.shape[1] / w0
        K[1] *= img.shape[0] / h0
        views[vid] = {"image": img, "intrinsics": K, "extrinsics": E}
    ref_ids = [r for r, _ in pairs]
    if produced is not None:
        H, W = views[ref_ids[0]]["image"].shape[:2]
        local = {vid: produced[(scan, vid)] for vid in pdist.shard_views(ref_ids, rank, world) if (scan, vid) in produced}
        maps = pdist.gather_scan_maps(local, ref_ids, H, W, device)
    else:
        maps = {}
    for vid in view_ids:
        if vid in maps:
            views[vid]["depth"], views[vid]["confidence"] = maps[vid][0], maps[vid][1]
        else:
            views[vid]["depth"] = read_map(os.path.join(args.output_folder, scan, "depth_est/{:0>8}{}".format(
                vid, args.file_format))).squeeze(2)
            views[vid]["confidence"] = read_map(os.path.join(args.output_folder, scan, "confidence/{:0>8}{}".format(
                vid, args.file_format))).squeeze(2)
    if rank != 0:
        return
    vertices, colors, masks = fusion.fuse_scan(views, pairs, args.geo_pixel_thres, args.geo_depth_thres,
                                               args.geo_mask_thres, args.photo_thres, device)
    os.makedirs(os.path.join(args.output_folder, scan, "mask"), exist_ok=True)
    for ref, (photo, geo, final) in masks.items():
        save_image(os.path.join(args.output_folder, scan, "mask/{:0>8}_photo.png".format(ref)), photo)
        save_image(os.path.join(args.output_folder, scan, "mask/{:0>8}_geo.png".format(ref)), geo)
        save_image(os.path.join(args.output_folder, scan, "mask/{:0>8}_final.png".format(ref)), final)
        print("processing {}, ref-view{:0>3}, geo_mask:{:3f}, photo_mask:{:3f}, final_mask: {:3f}".format(
            os.path.join(args.input_folder, scan), ref, geo.mean(), photo.mean(), final.mean()))
    ply = os.path.join(args.output_folder, scan, "fused.ply")
    fusion.write_ply(ply, vertices, colors)
    print("saving the final model to", ply)


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
    p.add_argument("--num_workers", type=int, default=4, help="DataLoader worker processes per rank")
    p.add_argument("--feature_cache", type=int, default=64,
                   help="> 0: decode and encode every view of a scan ONCE per rank and keep its FeatureNet pyramid on the device "
                        "(0 = re-decode and re-encode per sample like the reference; needs --batch_size 1)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
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

    produced = None
    if args.output_type in ("depth", "both"):
        produced = save_depth(args, rank, world, device)
    if args.output_type in ("fusion", "both"):
        if args.scan_list:
            if not os.path.isfile(args.scan_list):
                raise Exception("Invalid scan list file: {}".format(args.scan_list))
            with open(args.scan_list) as f:
                scans = [ln.rstrip() for ln in f.readlines()]
        else:
            scans = [""]
        for scan in scans:
            filter_depth(args, scan, produced, rank, world, device)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

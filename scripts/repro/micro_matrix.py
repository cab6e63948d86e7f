#!/usr/bin/env python
"""Self-checking victim micro-kernels (victims.hip) alone and beside the one-feature disturbers (disturbers.hip): which instruction class
of a victim goes wrong beside a register-only v_mfma_f32_16x16x32_f16 loop?  (DESIGN_LESSONS.md lesson 46.)"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
DL = ctypes.CDLL(os.path.join(HERE, "libdisturb.so"))
VL = ctypes.CDLL(os.path.join(HERE, "libvictims.so"))
DL.disturb_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
VL.victim_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
VL.victim_fill.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
dev = torch.device("cuda", 0)
A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
dbuf = torch.zeros(32 * 1024 * 1024, device=dev)
gbuf = torch.zeros(16 * 1024 * 1024, dtype=torch.int32, device=dev)  # 64 MB of mix(i)
assert VL.victim_fill(gbuf.data_ptr(), gbuf.numel(), torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
errs = torch.zeros(1, dtype=torch.int64, device=dev)
VICTIMS = [(0, "LDS b32 write/read", 4000), (1, "LDS b128 write/read", 4000), (2, "LDS records + workgroup barriers (NEIGHBOR)", 600),
           (3, "global_load_dwordx4 gather", 1500), (4, "DPP row_newbcast", 8000), (5, "v_pk_fma_f32", 8000),
           (6, "LDS b128 write / b64 read rows (PixelwiseNet)", 4000), (7, "expf + IEEE division", 2000),
           (8, "short-lived workgroups: LDS write, barrier, read (1 round)", 1), (8, "short-lived workgroups: LDS write, barrier, read (3 rounds)", 3)]
VICTIMS += [(9, "ds_read_b128 issued AFTER four global loads (returns while they are in flight)", 300),
            (10, "ds_read_b128 issued and waited for BEFORE the four global loads", 300)]
import sys
if len(sys.argv) > 1:
    VICTIMS = [v for v in VICTIMS if str(v[0]) in sys.argv[1].split(",")]
DISTURBERS = [(None, "alone"), (0, "fp16 MFMA 16x16x32 loop"), (1, "fp32 MFMA 16x16x4 loop"), (2, "cvt VALU loop")]
print("errors counted on the device per victim (8 launches of 2048 workgroups each):")
for vid, vname, iters in VICTIMS:
    row = []
    for did, dname in DISTURBERS:
        errs.zero_()
        torch.cuda.synchronize()
        if did is not None:
            for _ in range(300):
                assert DL.disturb_launch(did, dbuf.data_ptr(), dbuf.numel(), 2048, 4000, 0, B.cuda_stream) == 0
        for _ in range(8):
            assert VL.victim_launch(vid, errs.data_ptr(), 2048, iters, gbuf.data_ptr(), gbuf.numel() // 4, A.cuda_stream) == 0
        A.synchronize()
        busy = not B.query()
        torch.cuda.synchronize()
        row.append(f"{dname}: {int(errs.item())}{'' if (busy or did is None) else ' (B finished early)'}")
    print(f"  {vname:48s} " + " | ".join(row), flush=True)

#!/usr/bin/env python
"""Is it our kernels at all?  Victims: plain ATen elementwise kernels (torch.exp, torch.sigmoid, a*b+c, a/b, ...) on stream A; disturber:
the register-only v_mfma_f32_16x16x32_f16 loop of scripts/repro/disturbers.hip on stream B.  Every victim result is compared with
the same op's result computed alone.  (DESIGN_LESSONS.md lesson 46.)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DL = ctypes.CDLL(os.path.join(ROOT, "scripts", "repro", "libdisturb.so"))
DL.disturb_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda", 0)
dbuf = torch.zeros(32 * 1024 * 1024, device=dev)
A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16 * 1024 * 1024
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(n, device=dev, generator=g) * 3
y = torch.rand(n, device=dev, generator=g) + 0.5
z = torch.randn(n, device=dev, generator=g)
VICTIMS = {
    "exp": lambda: torch.exp(x), "sigmoid": lambda: torch.sigmoid(x), "tanh": lambda: torch.tanh(x), "log": lambda: torch.log(y),
    "sqrt": lambda: torch.sqrt(y), "rsqrt": lambda: torch.rsqrt(y), "reciprocal": lambda: torch.reciprocal(y), "div": lambda: x / y,
    "addcmul (fma)": lambda: torch.addcmul(z, x, y), "mul": lambda: x * y, "add": lambda: x + y, "floor": lambda: torch.floor(x),
    "ldexp": lambda: torch.ldexp(x, torch.tensor([3], device=dev)), "sin": lambda: torch.sin(x), "exp2": lambda: torch.exp2(x),
    "copy": lambda: x.clone(), "half->float": lambda: x.half().float(), "erf": lambda: torch.erf(x), "pow": lambda: torch.pow(y, 1.7),
}


def disturb(which, iters, launches, lds=0):
    with torch.cuda.stream(B):
        for _ in range(launches):
            assert DL.disturb_launch(which, dbuf.data_ptr(), dbuf.numel(), 2048, iters, lds, B.cuda_stream) == 0


for which, label in ((0, "fp16 MFMA 16x16x32 loop"), (1, "fp32 MFMA 16x16x4 loop")):
    print(f"== disturber: {label}")
    for name, f in VICTIMS.items():
        with torch.cuda.stream(A):
            want = f()
        torch.cuda.synchronize()
        disturb(which, 4000, 400)
        outs = []
        with torch.cuda.stream(A):
            for _ in range(8):
                outs.append(f())
            A.synchronize()
        busy = not B.query()
        torch.cuda.synchronize()
        bad = [int(((o != want) & ~(torch.isnan(o) & torch.isnan(want))).sum()) for o in outs]
        worst = max(float(((o - want).abs() / want.abs().clamp_min(1e-30)).max()) for o in outs)
        print(f"   {name:16s} elements that differ in 8 runs of {n}: {bad}  max rel {worst:.3e}  {'(B busy throughout)' if busy else '(B finished early)'}", flush=True)

#!/usr/bin/env python
"""LDS bank-conflict ABLATION builds ("measured, not argued").  rocprofv3 reports 33-50 % of the LDS cycles of
fpn_level_kernel, conv_tiled_kernel<8,8,..> and refine_fused_kernel as bank-conflict cycles.  What that costs in TIME is measured by
re-addressing the conflicting LDS accesses of a scratch copy of the sources so that consecutive lanes hit consecutive banks (same
instruction count, same global traffic, WRONG results) and timing the same launches against the product build:

    python scripts/experiments/lds_conflicts/build_ablation.py            # -> build/ldsab/libpmn_hip_{fpn,tiled,refine}.so
    python scripts/call_ab.py --ops fpn_level,conv2d,refine_fused --libs patchmatchnet_amd/csrc/libpmn_hip.so,build/ldsab/libpmn_hip_fpn.so,...

Results: profiles/r06_lds_conflict_ablation.log."""
import os
import re
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
CS = os.path.join(ROOT, "patchmatchnet_amd", "csrc")
OUT = os.path.join(ROOT, "build", "ldsab")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def sub(text, old, new, count=1):
    assert text.count(old) >= 1, old
    return text.replace(old, new) if count == 0 else text.replace(old, new, count)


def variant(name, fname, edits):
    d = os.path.join(OUT, "csrc_" + name)
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(CS, d, ignore=shutil.ignore_patterns("*.o", "*.so", "experimental"))
    os.makedirs(os.path.join(OUT, "include"), exist_ok=True)
    p = os.path.join(d, fname)
    s = open(p).read()
    for old, new in edits:
        s = sub(s, old, new)
    s = s.replace('"../../include/pmn_hip.h"', '"%s"' % os.path.join(ROOT, "include", "pmn_hip.h"))
    open(p, "w").write(s)
    for f in os.listdir(d):  # the common header's include path
        if f.endswith((".hpp", ".hip")):
            q = os.path.join(d, f)
            t = open(q).read().replace('"../../include/pmn_hip.h"', '"%s"' % os.path.join(ROOT, "include", "pmn_hip.h"))
            open(q, "w").write(t)
    obj = os.path.join(OUT, f"{name}.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", p, "-o", obj])
    others = [os.path.join(CS, o) for o in os.listdir(CS) if o.endswith(".o") and not o.endswith(".x.o") and o != fname.replace(".hip", ".o")]
    lib = os.path.join(OUT, f"libpmn_hip_{name}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib, obj] + others)
    print("built", lib)


subprocess.check_call(["make", "-s", "-C", CS, "-j8"])
os.makedirs(OUT, exist_ok=True)


def main():
    # fpn_level_kernel: staging writes lane-linear, every operand read lane-linear
    variant("fpn", "conv.hip", [
        ("*reinterpret_cast<float4*>(xs + pix * XP + 4 * q) = vx[k];", "*reinterpret_cast<float4*>(xs + idx * 4) = vx[k];"),
        ("if (idx < UPW * UPW * (COUT / 4)) *reinterpret_cast<float4*>(us + pix * UPP + 4 * q) = vu[k];",
         "if (idx < UPW * UPW * (COUT / 4)) *reinterpret_cast<float4*>(us + idx * 4) = vu[k];"),
        ("const float4 p00 = *reinterpret_cast<const float4*>(u00 + c0 + c), p01 = *reinterpret_cast<const float4*>(u01 + c0 + c);\n"
         "                const float4 p10 = *reinterpret_cast<const float4*>(u10 + c0 + c), p11 = *reinterpret_cast<const float4*>(u11 + c0 + c);\n"
         "                // ATen upsample_bilinear2d: h0*(w0*a + w1*b) + h1*(w0*c + w1*d)\n"
         "                acc[c] = hy",
         "const float4 p00 = *reinterpret_cast<const float4*>(us + tid * 4), p01 = *reinterpret_cast<const float4*>(us + ((tid * 4 + 256) & 1023));\n"
         "                const float4 p10 = *reinterpret_cast<const float4*>(us + ((tid * 4 + 512) & 1023)), p11 = *reinterpret_cast<const float4*>(us + ((tid * 4 + 768) & 1023));\n"
         "                acc[c] = hy"),
        ("            const float4 v4 = *reinterpret_cast<const float4*>(xp + ci);\n            const float v[4] = {v4.x, v4.y, v4.z, v4.w};\n            const cfloat* wq = wt + __builtin_amdgcn_readfirstlane(ci * COUT + c0);",
         "            const float4 v4 = *reinterpret_cast<const float4*>(xs + (ci / 4) * 1024 + tid * 4);\n            const float v[4] = {v4.x, v4.y, v4.z, v4.w};\n            const cfloat* wq = wt + __builtin_amdgcn_readfirstlane(ci * COUT + c0);"),
    ])
    # conv_tiled_kernel: staging writes and operand reads lane-linear
    variant("tiled", "conv.hip", [
        ("if (idx < ih * iw * CQ) *reinterpret_cast<float4*>(tile + pix * CCP + 4 * q) = v[u];",
         "if (idx < ih * iw * CQ) *reinterpret_cast<float4*>(tile + idx * 4) = v[u];"),
        ("const float4 v4 = *reinterpret_cast<const float4*>(tp + 4 * q);\n                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};\n                    // readfirstlane pins",
         "const float4 v4 = *reinterpret_cast<const float4*>(tile + ((tid * 4 + q * 1024 + (ky * K + kx) * 64) % (ih * iw * CC)));\n                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};\n                    // readfirstlane pins"),
    ])
    # refine_fused_kernel, phase (2): conv0's 27 scalar reads and the deconvolution's float4 pairs lane-linear
    variant("refine", "refine.hip", [
        ("const float v = xp[(ci * IR + ky) * IC + kx];", "const float v = xin[((ci * 3 + ky) * 3 + kx) * 64 + lane];"),
        ("const float4 a = *reinterpret_cast<const float4*>(ip), b = *reinterpret_cast<const float4*>(ip + 4);",
         "const float4 a = *reinterpret_cast<const float4*>(tp + lane * 8 + (ip - ip)), b = *reinterpret_cast<const float4*>(tp + lane * 8 + 4);"),
    ])


if __name__ == "__main__":
    main()

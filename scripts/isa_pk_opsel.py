#!/usr/bin/env python
"""Static guard for DESIGN_LESSONS.md lesson 46: list every packed-fp32 instruction of a gfx950 binary whose SECOND source takes its
HIGH register for the LOW half of the result -- `op_sel:[x,1]` / `op_sel:[x,1,x]` on v_pk_add_f32, v_pk_mul_f32, v_pk_fma_f32.

Measured on MI355X (scripts/repro/pk_opsel_matrix.hip, profiles/r06_overlap/r06_pk_opsel_matrix.log): exactly these forms read that
operand as ZERO now and then while waves of another kernel issue v_mfma_f32_16x16x32_f16 / _bf16 (rarely v_mfma_f32_32x32x16_f16) on
the same CU -- alone, or beside any other instruction mix, never; every other op_sel / op_sel_hi combination of the three
instructions, never.  hipcc emits the form when a scalar that sits in the high half of a 64-bit register pair (the .y / .w of a
vector load, of a DPP-built pair ...) is broadcast into packed math.  A binary without the form cannot hit the defect.

    python scripts/isa_pk_opsel.py patchmatchnet_amd/csrc/libpmn_hip.so [more .so / .o ...]      exit code 1 if any site is found

tests/test_isa_hazards.py holds the shipped library to zero sites."""
import argparse
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
AFFECTED = re.compile(r"^\s*(v_pk_add_f32|v_pk_mul_f32|v_pk_fma_f32)\s.*\bop_sel:\[[01],1(?:,[01])?\]")
LABEL = re.compile(r"^[0-9a-f]+ <(.+)>:$")


def disassemble(path):
    """-> list of (bundle name, disassembly text) for every gfx950 code object bundled in `path`"""
    out = []
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, os.path.basename(path))
        shutil.copy(path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        bundles = sorted(f for f in os.listdir(d) if "amdgcn" in f and "gfx950" in f)
        if not bundles:  # a bare code object (.co / .hsaco) or a device-only object
            bundles = [os.path.basename(path)]
        for b in bundles:
            r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", os.path.join(d, b)], capture_output=True, text=True)
            if r.returncode == 0 and "v_" in r.stdout:
                out.append((b, r.stdout))
    return out


def sites(text):
    """-> {kernel: [instruction text, ...]} and the number of packed-fp32 instructions seen"""
    found, kernel, seen = {}, "?", 0
    for ln in text.split("\n"):
        m = LABEL.match(ln)
        if m:
            kernel = m.group(1)
            continue
        t = ln.split("//")[0]
        if "v_pk_" in t and "_f32" in t:
            seen += 1
            if AFFECTED.match(t):
                found.setdefault(kernel, []).append(" ".join(t.split()))
    return found, seen


def check(paths, verbose=False, out=sys.stdout):
    total = 0
    for p in paths:
        dis = disassemble(p)
        if not dis:
            print(f"{p}: no gfx950 code object found", file=out)
            total += 1  # a check that looked at nothing must not pass
            continue
        n_sites, n_seen, per_kernel = 0, 0, {}
        for _, text in dis:
            f, seen = sites(text)
            n_seen += seen
            for k, v in f.items():
                per_kernel.setdefault(k, []).extend(v)
                n_sites += len(v)
        print(f"{p}: {n_seen} packed-fp32 instructions in {len(dis)} code objects, {n_sites} with the second source's high half selected "
              f"for the low result", file=out)
        for k, v in sorted(per_kernel.items(), key=lambda kv: -len(kv[1])):
            print(f"    {len(v):5d}  {k}" + (f"   e.g. {v[0]}" if verbose else ""), file=out)
        total += n_sites
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args()
    return 1 if check(args.files, args.verbose) else 0


if __name__ == "__main__":
    sys.exit(main())

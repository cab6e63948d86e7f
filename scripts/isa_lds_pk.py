#!/usr/bin/env python
"""Static check for DESIGN_LESSONS.md lesson 46: which packed-fp32 instructions (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32) read a
register whose most recent writer -- walking back through the straight-line code of the basic block chain -- is an LDS load, and how
many instructions after the s_waitcnt that made the load's data architecturally visible they issue.

    python scripts/isa_lds_pk.py file.s [file.s ...] [--max-distance N] [--kernels substring,...]

The hazard measured on MI355X: such a read can return the register's PREVIOUS contents while another kernel's waves issue dense
v_mfma_f32_16x16x32_f16 on the same CU.  Every site the GPU probes caught had distance <= 3 (the packed instruction right behind the
wait); sites 8+ instructions behind it never failed in 24 x 3 x 45 disturbed launches.  The tool prints every site with its
distance, per kernel, so that a new kernel can be checked before it ships; tests/test_isa_hazards.py holds the product's kernels to
"no site closer than MIN_DISTANCE instructions behind its wait".

A register is tracked through full VGPR numbers (v12, v[12:15] -> v12..v15).  Control flow: the walk back is linear over the listing
and stops at a label that is a loop header or after 400 instructions; that is a heuristic, good enough for the unrolled loops of
this library (a register written on another path is reported as 'unknown writer', never silently dropped)."""
import argparse
import re
import sys

VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = []
    for m in VREG.finditer(text):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])]
    return parts[0], ops


def dst_count(op):
    """number of leading operands that are destinations"""
    if op.startswith(("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "s_waitcnt", "s_nop", "s_barrier")):
        return 0
    if op.startswith(("v_cmp", "v_cmpx")):
        return 1  # vcc / sgpr pair: no VGPR destination, but harmless to treat operand 0 as the destination
    if op.startswith(("v_div_scale", "v_mad_u64_u32", "v_mad_i64_i32", "v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_subrev_co")):
        return 2
    return 1


def analyse(path, kernels, max_distance):
    lines = open(path).read().split("\n")
    sites = []
    kernel = None
    body = []  # (op, dst regs, src regs, is_lds_load, raw)
    for ln in lines:
        if re.match(r"^_Z\w+:|^[A-Za-z_]\w*:\s*; @", ln):
            kernel = ln.split(":")[0]
            body = []
            continue
        if ln.startswith(".Lfunc_end"):
            kernel = None
            continue
        if kernel is None or (kernels and not any(k in kernel for k in kernels)):
            continue
        t = ln.strip()
        if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
            continue
        if t.startswith(".LBB"):
            body.append(("label", [], [], False, t))
            continue
        op, ops = split_operands(t)
        if not re.match(r"^[a-z]", op):
            continue
        nd = dst_count(op)
        dst = [r for o in ops[:nd] for r in regs(o)]
        src = [r for o in ops[nd:] for r in regs(o)]
        if op.startswith("v_fmac") or op.startswith("v_pk_fmac") or "dpp" in t or op.startswith("v_mac"):
            src += dst  # accumulate / old-value forms read their destination
        is_lds = op.startswith("ds_read") or op.startswith("ds_bpermute") or op.startswith("ds_permute") or op.startswith("ds_swizzle")
        body.append((op, dst, src, is_lds, t))
        if op.startswith("v_pk_") and op.endswith("_f32"):
            i = len(body) - 1
            for r in sorted(set(src)):
                # walk back to r's most recent writer
                j, waits = i - 1, 0
                steps = 0
                while j >= 0 and steps < 400:
                    o2, d2, s2, lds2, raw2 = body[j]
                    if o2 == "label":
                        j -= 1
                        continue
                    steps += 1
                    if r in d2:
                        if lds2:
                            # distance: instructions between the first lgkmcnt wait after the load and the packed instruction
                            k, dist = j + 1, None
                            while k < i:
                                if body[k][0] == "s_waitcnt" and "lgkmcnt" in body[k][4]:
                                    dist = sum(1 for q in range(k + 1, i) if body[q][0] not in ("label", "s_waitcnt", "s_nop"))
                                    break
                                k += 1
                            sites.append((kernel, r, dist, raw2, t))
                        break
                    j -= 1
    return sites


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--max-distance", type=int, default=7, help="report only sites at most this many instructions behind their wait")
    ap.add_argument("--kernels", default="")
    ap.add_argument("--all", action="store_true")
    args = ap.parse_args()
    kernels = [k for k in args.kernels.split(",") if k]
    worst = {}
    for f in args.files:
        for kernel, r, dist, load, use in analyse(f, kernels, args.max_distance):
            key = (f, kernel)
            d = -1 if dist is None else dist
            worst.setdefault(key, []).append((d, r, load, use))
    bad = 0
    for (f, kernel), sites in sorted(worst.items()):
        close = [s for s in sites if s[0] <= args.max_distance]
        if args.all or close:
            print(f"{f.split('/')[-1]}  {kernel}: {len(sites)} LDS->packed sites, {len(close)} within {args.max_distance} instructions of the wait, "
                  f"closest {min(s[0] for s in sites)}")
            for d, r, load, use in sorted(close)[:6]:
                print(f"      v{r}  distance {d}:  {load.split(';')[0].strip()}   ->   {use.split(';')[0].strip()}")
        bad += len(close)
    print(f"total sites within {args.max_distance}: {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Instruction-class tally of one kernel in a hipcc -S listing: whole function and its largest loop (label .. backward branch).
Development aid: python scripts/isa_count.py file.s '<32, 8, 0, 16, true>'-style mangled fragment (e.g. Li32ELi8ELi0ELi16ELb1E)."""
import collections, re, sys
path, frag = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*%s.*:\s" % re.escape(frag), l + " ") )
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
def cls(op):
    if op.startswith("v_pk_"): return "v_pk"
    if "dpp" in op: return "v_dpp"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_load") or op.startswith("buffer_load"): return "vmem_rd"
    if op.startswith("global_store") or op.startswith("buffer_store") or op.startswith("global_atomic"): return "vmem_wr"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    return "other"
def tally(seg):
    c = collections.Counter(); ops = collections.Counter()
    for l in seg:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"): continue
        op = l.split()[0]
        full = l.split(";")[0]
        k = cls(op)
        if k == "valu" and ("row_" in full or "quad_perm" in full): k = "v_dpp"
        c[k] += 1; ops[op] += 1
    return c, ops
labels = {l[:-1]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.search(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
print("function:", dict(tally(body)[0]))
for n, a, b in sorted(loops, reverse=True)[:3]:
    c, ops = tally(body[a:b + 1])
    print("loop %s lines %d:" % (body[a], n), dict(c))
    if len(sys.argv) > 3:
        print("   ", ops.most_common(40))

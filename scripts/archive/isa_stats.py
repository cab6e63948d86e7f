#!/usr/bin/env python
"""Per-kernel register / scratch / instruction-mix summary of a hipcc -S listing (development aid).
usage: isa_stats.py file.s [name-substring]"""
import re
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end\d+:", s, re.S | re.M):
    name, code = m.group(1), m.group(2)
    if flt not in name:
        continue
    def sym(k):
        r = re.search(r"\.set " + re.escape(name) + r"\." + k + r", (\d+)", s)
        return r.group(1) if r else "?"
    cnt = lambda pat: len(re.findall(pat, code))
    print(name)
    print(f"   vgpr {sym('num_vgpr')} agpr {sym('num_agpr')} sgpr {sym('numbered_sgpr')} scratch {sym('private_seg_size')} | "
          f"s_load {cnt(r's_load_dword')} global_load {cnt(r'global_load')} ds_read {cnt(r'ds_read')} ds_write {cnt(r'ds_write')} "
          f"v_fma {cnt(r'v_fma_f32|v_fmac_f32')} v_pk_fma {cnt('v_pk_fma_f32')} mfma {cnt('v_mfma')} scratch_ops {cnt('scratch_')} "
          f"readlane {cnt('v_readlane')}")

#!/usr/bin/env python3
"""tools/isa_tokens.py FILE.s KERNEL_SUBSTRING: one line per basic block of the kernel's ISA, memory / MFMA / wait instructions
as tokens (G4 G2 g = global loads, S = global store, r / w = LDS read / write, M = MFMA, [vmcnt(n)] waits, BAR, br = branch).
How to read it: a [vmcnt(0)] right after a block of loads that sits under a branch means the loads are NOT overlapped with
what follows (profiles/r02_waitcnt_fix.txt).  FILE.s from: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only X.hip"""
import re
import sys

src, pat = sys.argv[1], sys.argv[2]
text = open(src).read()
for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)\.end_amdhsa_kernel', text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    vg = re.search(r'\.amdhsa_next_free_vgpr (\d+)', body)
    print("==", name, "vgpr", vg.group(1) if vg else "?", "instructions", len(re.findall(r'^\t[a-z]', body, re.M)))
    line = []
    for ln in body.split("\n"):
        t = ln.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            print(" ".join(line)); line = [t.split(":")[0]]
        elif t.startswith("v_mfma"): line.append("M")
        elif t.startswith("global_load_dwordx4") or t.startswith("buffer_load_dwordx4"): line.append("G4")
        elif t.startswith("global_load_dwordx2"): line.append("G2")
        elif t.startswith("global_load") or t.startswith("buffer_load") or t.startswith("flat_load"): line.append("g")
        elif t.startswith("global_store") or t.startswith("buffer_store") or t.startswith("flat_store"): line.append("S")
        elif t.startswith("global_atomic"): line.append("A")
        elif t.startswith("s_load"): line.append("sl")
        elif t.startswith("ds_read") or t.startswith("ds_load"): line.append("r")
        elif t.startswith("ds_write") or t.startswith("ds_store"): line.append("w")
        elif t.startswith("ds_"): line.append("ds")
        elif t.startswith("s_waitcnt"): line.append("[" + t[len("s_waitcnt"):].strip().replace(" ", "") + "]")
        elif t.startswith("s_barrier"): line.append("BAR")
        elif t.startswith("s_cbranch") or t.startswith("s_branch"): line.append("br>" + t.split()[-1])
    print(" ".join(line))

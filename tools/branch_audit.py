#!/usr/bin/env python3
"""Divergent-branch audit of one kernel of csrc/pbrt_api.gfx950.s (make -C csrc asm): every forward branch over a masked body
(s_and_saveexec / s_andn2_saveexec / s_xor ... s_cbranch_exec[n]z LABEL), with the instructions of the body it skips -- VALU, SALU,
memory -- and the scalar instructions the masking itself costs.  Short bodies (<= 6 VALU, no memory) are candidates for selects.
Usage: tools/branch_audit.py <mangled-substring> [max_body_valu]"""
import re
import sys

path = "physics-based-ray-tracing_amd/csrc/pbrt_api.gfx950.s"
want = sys.argv[1]
short = int(sys.argv[2]) if len(sys.argv) > 2 else 6
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and re.match(r"^_Z\w+:", l) and want in l.split(":")[0])
body = []
for l in lines[start + 1:]:
    if l.strip().startswith("s_endpgm"):
        body.append(l)
        break
    body.append(l)
label_at = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_", "ds_")): return "mem"
    return "other"
rows = []
tot = {"valu": 0, "salu": 0, "mem": 0, "other": 0}
for i, l in enumerate(body):
    m = re.match(r"^\t([a-z_0-9]+)", l)
    if m: tot[cls(m.group(1))] += 1
    m = re.match(r"^\ts_cbranch_(execz|execnz|vccz|vccnz|scc0|scc1)\s+(\.LBB\d+_\d+)", l)
    if not m or m.group(2) not in label_at: continue
    j = label_at[m.group(2)]
    if j <= i: continue   # loop back edge
    n = {"valu": 0, "salu": 0, "mem": 0, "other": 0}
    inner = 0
    for k in range(i + 1, j):
        mm = re.match(r"^\t([a-z_0-9]+)", body[k])
        if mm:
            n[cls(mm.group(1))] += 1
            if mm.group(1).startswith("s_cbranch"): inner += 1
    rows.append((i, m.group(1), m.group(2), n, inner))
print(f"{lines[start][:-1]}: {tot}")
print(f"{len(rows)} forward branches; bodies with <= {short} VALU, no memory, no inner branch:")
for i, kind, lab, n, inner in rows:
    flag = "  <-- select?" if n["valu"] <= short and n["mem"] == 0 and inner == 0 else ""
    print(f"  line {i:5d} {kind:7s} -> {lab:12s} body valu {n['valu']:4d} salu {n['salu']:3d} mem {n['mem']:3d} inner-branches {inner}{flag}")
nops = sum(1 for l in body if re.match(r"^\ts_nop", l))
waits = sum(1 for l in body if re.match(r"^\ts_waitcnt", l))
print(f"s_nop instructions: {nops}; s_waitcnt: {waits}")

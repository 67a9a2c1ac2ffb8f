#!/usr/bin/env python3
"""Registers, scratch and occupancy of every render_queue instantiation in a `hipcc -S` listing of rt_amd/csrc/kernels.hip:

    hipcc --offload-arch=gfx950 <the Makefile's HIPFLAGS> -S --cuda-device-only -o kernels.s rt_amd/csrc/kernels.hip
    python tools/kernel_registers.py kernels.s

The scalar-register kernels live at the edge of their vector-register budget (72 at 7 waves per SIMD): a few bytes of scratch in
the loop cost a 7-sphere frame a factor (round 5: dielectric.toml 2.7 -> 9.4 ms with 124 bytes), and which build spills moves with
every change of the source.  Run this after any change to the kernels; `--fail-on-scratch N` exits non-zero if a scalar-register
kernel without the sm table contains more than N scratch loads / stores.  (`scratch` = bytes of private segment the build
reserves, `ops` = scratch_load / scratch_store instructions in its text: a build can reserve a segment for stack objects
it never touches — that costs nothing; spills in the loop are what to look for.)"""
import re
import sys

path = sys.argv[1]
limit = int(sys.argv[sys.argv.index("--fail-on-scratch") + 1]) if "--fail-on-scratch" in sys.argv else None
name = None
rows = []
cur = {}
for line in open(path):
    m = re.match(r"^_ZN6rt_hip12_GLOBAL__N_1\d+(render_queue(?:_fast)?)ILi(-?\d+|n\d+)ELb([01])ELb([01])ELi(\d)ELb([01])EE", line)
    if m and line.rstrip().endswith(":") is False and ":" in line:
        ns = m.group(2)
        name = (m.group(1), int(ns.replace("n", "-")), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)))
        cur = {}
        continue
    if name:
        if re.match(r"^\s*scratch_(load|store)", line):
            cur["scratch_ops"] = cur.get("scratch_ops", 0) + 1
        m = re.match(r"^; (NumVgprs|TotalNumSgprs|ScratchSize|Occupancy): (\d+)", line)
        if m:
            cur[m.group(1)] = int(m.group(2))
            if m.group(1) == "Occupancy":
                rows.append((name, cur))
                name = None
                cur = {}
bad = 0
print(f"{'kernel':14s} {'NS':>3s} {'SM':>2s} {'HALF':>4s} {'NP':>2s} {'GC':>2s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>7s} {'ops':>4s} {'waves':>5s}")
for (kernel, ns, sm, half, np_, gc), c in sorted(rows):
    flag = ""
    if limit is not None and ns > 0 and not sm and c.get("scratch_ops", 0) > limit:
        flag, bad = "  <-- scratch", bad + 1
    print(f"{kernel:14s} {ns:3d} {sm:2d} {half:4d} {np_:2d} {gc:2d} {c.get('NumVgprs', -1):5d} {c.get('TotalNumSgprs', -1):5d} {c.get('ScratchSize', -1):7d} {c.get('scratch_ops', 0):4d} {c.get('Occupancy', -1):5d}{flag}")
sys.exit(1 if bad else 0)

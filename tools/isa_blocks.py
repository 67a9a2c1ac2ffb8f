"""Static side of the per-region ISA table: instruction counts per basic block of one kernel in a hipcc -S listing.

    python tools/isa_blocks.py kernels.s render_queueILi3ELb0E            -> one line per basic block of the kernel's loop

Columns: VALU (all vector ALU instructions), of which quarter-rate (v_rcp/v_rsq/v_sqrt/v_mul_lo_u32/v_mul_hi_u32/v_mad_u64_u32:
4 issue slots each) and packed (v_pk_*: 2 slots each on this chip, profiles/r01/microbench_valu.txt); issue slots; SALU;
branches; s_nop / s_waitcnt; LDS; "rare" marks blocks that hold the general IEEE expansion (never taken in practice)."""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
QUARTER = ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_div_scale", "v_div_fmas", "v_div_fixup")
blocks, cur = [], {"label": "entry", "ins": []}
for l in lines[start + 1 : end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l) or re.match(r"^; %bb\.(\d+):", l)
    if m:
        blocks.append(cur)
        cur = {"label": m.group(1) if l.startswith(".") else "bb." + m.group(1), "ins": []}
        continue
    t = l.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ins"].append(t)
blocks.append(cur)
print(f"{'block':12s} {'VALU':>5s} {'quart':>5s} {'pk':>4s} {'slots':>6s} {'SALU':>5s} {'br':>3s} {'nop/wait':>8s} {'LDS':>4s}  note")
tot = [0] * 8
for b in blocks:
    ins = b["ins"]
    valu = [i for i in ins if i.startswith("v_")]
    quarter = [i for i in valu if i.startswith(QUARTER)]
    pk = [i for i in valu if i.startswith("v_pk_")]
    slots = len(valu) + 3 * len(quarter) + len(pk)
    salu = [i for i in ins if i.startswith("s_") and not i.startswith(("s_nop", "s_waitcnt", "s_cbranch", "s_branch", "s_endpgm", "s_barrier"))]
    br = [i for i in ins if i.startswith(("s_cbranch", "s_branch"))]
    nop = [i for i in ins if i.startswith(("s_nop", "s_waitcnt"))]
    lds = [i for i in ins if i.startswith("ds_")]
    rare = any("v_div_fixup" in i or "v_cmp_class" in i for i in ins)
    row = [len(valu), len(quarter), len(pk), slots, len(salu), len(br), len(nop), len(lds)]
    if not rare:
        tot = [a + b_ for a, b_ in zip(tot, row)]
    print(f"{b['label']:12s} {row[0]:5d} {row[1]:5d} {row[2]:4d} {row[3]:6d} {row[4]:5d} {row[5]:3d} {row[6]:8d} {row[7]:4d}  {'rare path' if rare else ''}")
print(f"{'total-rare':12s} {tot[0]:5d} {tot[1]:5d} {tot[2]:4d} {tot[3]:6d} {tot[4]:5d} {tot[5]:3d} {tot[6]:8d} {tot[7]:4d}")

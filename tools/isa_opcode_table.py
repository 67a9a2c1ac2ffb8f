#!/usr/bin/env python3
"""Opcode-level table of a render_queue loop: which vector instructions WITHOUT a counter class of their own (compares, selects,
moves, lane operations) and which scalar instructions a loop trip issues, per region of the trip, weighted by how often the region
runs (`profiles/r05/other_class_by_opcode.txt`).

    python tools/isa_opcode_table.py <loop listing> <region map> <region counters>

  loop listing     the kernel's loop, one instruction per line, rare-path blocks (the general IEEE expansions) elided — what
                   the helper at the end of this file prints from a `hipcc -S` listing (`--loop kernels.s <mangled-name part>`)
  region map       lines `first-last  region name | weight`: 1-based line ranges of the loop listing, the region they implement,
                   and the region's runs per trip as an expression over the region counters' runs/trip (names below)
  region counters  output of tools/region_profile.py on the instrumented build (runs per trip of each source region)

Vector classes follow the SQ_INSTS_VALU_* counters: fma / mul / add f32, transcendental, conversion, int32, int64; everything else is
`other` — the class the table is about."""
import collections
import re
import sys

COUNTER_NAMES = {"trip": "trip", "query": "query: probes", "sqrt": "sqrt half of a sphere", "hit": "hit: lookups + normal", "miss": "miss: sky + end of sample", "metal": "metal: normalise dir + reflect",
                 "handout": "hand-out vote", "take": "take item", "tail": "tail: two draws", "scatter": "scatter: 3rd draw, unit vector", "restart": "restart: primary ray", "normalise": "normalise new direction",
                 "dead": "absorbed / out of bounces"}


def vector_class(op):
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32")):
        return "fma"
    if op.startswith("v_mul_f32"):
        return "mul"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")):
        return "add"
    if op.startswith(("v_rsq", "v_rcp", "v_sqrt")):
        return "trans"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_mad_u64"):
        return "int64"
    if op.startswith(("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_subrev_co", "v_add_co", "v_mul_lo", "v_mul_hi", "v_mul_u32", "v_lshl", "v_lshr", "v_ashr", "v_xor", "v_and", "v_or", "v_bitop", "v_bfe", "v_min_u", "v_max_u", "v_mad_i32", "v_mad_u32", "v_mbcnt", "v_add3", "v_lshl_add", "v_lshl_or", "v_and_or")):
        return "int32"
    return "other"


def main():
    listing = open(sys.argv[1]).read().splitlines()
    runs = {}
    for line in open(sys.argv[3]):
        m = re.match(r"^(.{34})\s+(\d+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            runs[m.group(1).strip()] = float(m.group(3))
    trips_per_64 = float(re.search(r"trips per 64 samples: ([\d.]+)", open(sys.argv[3]).read()).group(1))
    env = {key: runs[name] for key, name in COUNTER_NAMES.items()}
    regions = []
    for line in open(sys.argv[2]):
        line = line.split("#")[0].strip()
        if not line:
            continue
        m = re.match(r"^(\d+)-(\d+)\s+(.*?)\s*\|\s*(.*)$", line)
        regions.append((int(m.group(1)), int(m.group(2)), m.group(3), m.group(4), eval(m.group(4), {}, env)))
    covered = set()
    table = []
    for first, last, name, expr, weight in regions:
        valu, other, salu, lds = collections.Counter(), collections.Counter(), collections.Counter(), 0
        for i in range(first - 1, last):
            assert i not in covered, f"line {i + 1} mapped twice"
            covered.add(i)
            t = listing[i].strip()
            if not t or t.startswith((";", ".")) or t.endswith(":"):
                continue
            op = t.split()[0]
            if op.startswith("v_"):
                c = vector_class(op)
                valu[c] += 1
                if c == "other":
                    other[re.sub(r"_e32$|_e64$", "", op)] += 1
            elif op.startswith("s_") and not op.startswith(("s_nop", "s_waitcnt")):
                salu[re.sub(r"_b64$|_b32$|_i32$|_u32$|_u64$", "", op)] += 1
            elif op.startswith("ds_"):
                lds += 1
        table.append((name, expr, weight, valu, other, salu, lds))
    missing = [i + 1 for i, l in enumerate(listing) if i not in covered and l.strip().startswith(("v_", "s_", "ds_"))]
    assert not missing, f"instructions outside every region: lines {missing[:20]}"
    classes = ["fma", "mul", "add", "trans", "cvt", "int32", "int64", "other"]
    print(f"{'region':44s} {'runs/trip':>9s} | " + " ".join(f"{c:>5s}" for c in classes) + f" {'VALU':>5s} {'SALU':>5s} {'LDS':>4s} | {'VALU/trip':>9s} {'other/trip':>10s} {'SALU/trip':>9s}")
    tot = collections.Counter()
    other_total, salu_total = collections.Counter(), collections.Counter()
    for name, expr, weight, valu, other, salu, lds in table:
        v, s = sum(valu.values()), sum(salu.values())
        print(f"{name:44s} {weight:9.3f} | " + " ".join(f"{valu[c]:5d}" for c in classes) + f" {v:5d} {s:5d} {lds:4d} | {v * weight:9.1f} {valu['other'] * weight:10.1f} {s * weight:9.1f}")
        for c in classes:
            tot[c] += valu[c] * weight
        tot["valu"] += v * weight
        tot["salu"] += s * weight
        for op, n in other.items():
            other_total[op] += n * weight
        for op, n in salu.items():
            salu_total[op] += n * weight
    print(f"{'per trip':44s} {'':9s} | " + " ".join(f"{tot[c]:5.1f}" for c in classes) + f" {tot['valu']:5.1f} {tot['salu']:5.1f}")
    print(f"per 64 samples ({trips_per_64:.3f} trips): VALU {tot['valu'] * trips_per_64:.1f}, of which other {tot['other'] * trips_per_64:.1f}, int32 {tot['int32'] * trips_per_64:.1f}; SALU {tot['salu'] * trips_per_64:.1f}")
    print("\nthe `other` class by opcode, wave-instructions per trip (and per 64 samples):")
    for op, n in other_total.most_common():
        print(f"  {op:28s} {n:7.2f} {n * trips_per_64:8.2f}")
    print("\nthe scalar stream by opcode, per trip (and per 64 samples):")
    for op, n in salu_total.most_common():
        print(f"  {op:28s} {n:7.2f} {n * trips_per_64:8.2f}")


def print_loop(path, key):
    """--loop kernels.s <part of the mangled kernel name>: the kernel's loop with rare-path blocks elided"""
    lines = open(path).read().splitlines()
    begin = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
    end = next(i for i in range(begin, len(lines)) if lines[i].startswith(".Lfunc_end"))
    lines = lines[begin:end]
    header = next(i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l)
    name = re.match(r"^\.L(BB\d+_\d+):", lines[header - 1] if lines[header].lstrip().startswith(";") and not lines[header].startswith(".L") else lines[header])
    name = name.group(1) if name else re.match(r"^\.L(BB\d+_\d+):", lines[header]).group(1)
    inside = [i for i, l in enumerate(lines) if f"Header={name} " in l or i == header]
    start = min(inside)
    # the loop ends where the first block behind its last block begins
    stop = next(i for i in range(max(inside) + 1, len(lines)) if re.match(r"^(\.LBB|; %bb)", lines[i]))
    out, i = [], max(start - 1, 0)
    while i < stop:
        l = lines[i]
        if "rare path" in l:
            while out and not re.match(r"^(; %bb|\.LBB)", out[-1]):
                out.pop()
            out.append("    ... rare path block elided ...")
            i += 1
            while i < stop and not re.match(r"^\.LBB", lines[i]):
                i += 1
            continue
        if not l.strip().startswith("; implicit-def"):
            out.append(l)
        i += 1
    print("\n".join(out))


if __name__ == "__main__":
    if sys.argv[1] == "--loop":
        print_loop(sys.argv[2], sys.argv[3])
    else:
        main()

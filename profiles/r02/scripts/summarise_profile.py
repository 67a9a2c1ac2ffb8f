#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/prof into small, committable summaries under profiles/<tag>/.

usage: tools/summarise_profile.py <tag> [note] [traffic_key]   (traffic_key e.g. basic_1920x1080x256_n1)
  kernel_stats.csv      <- the --kernel-trace --stats summary (per-kernel count / total / average duration)
  pmc_summary.csv       <- per counter: mean value per render-kernel dispatch, over all --pmc passes found
  pmc_traffic.json (at profiles/) <- HBM bytes per launch from FETCH_SIZE / WRITE_SIZE (guide's gfx950 correction:
                           FETCH_SIZE under-reports wide streaming reads by 2x; reported raw and corrected)
"""
import csv, glob, json, sys, collections, pathlib

tag = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else ""
root = pathlib.Path(__file__).resolve().parent.parent
import os
src = root / "gpurun_out" / os.environ.get("PROF_DIR", "prof")  # PROF_DIR=prof_c2: the passes of tools/gpu_traffic_configs.sh
out = root / "profiles" / tag
out.mkdir(parents=True, exist_ok=True)

stats = sorted(src.glob("trace/*/*_kernel_stats.csv"), key=lambda p: p.stat().st_mtime)
if stats:
    (out / "kernel_stats.csv").write_text(stats[-1].read_text())
rows = []
for f in sorted(src.glob("pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    kernel = ""
    for r in csv.DictReader(open(f)):
        if "render" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernel = r["Kernel_Name"].split("(")[0]
    for k, v in sorted(agg.items()):
        rows.append((f.parts[-3], kernel, k, len(v), sum(v) / len(v)))
with open(out / "pmc_summary.csv", "w") as f:
    f.write("# " + note + "\n")
    f.write("pass,kernel,counter,dispatches,mean_per_dispatch\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
c = {r[2]: r[4] for r in rows}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    rec = {
        "note": note,
        "FETCH_SIZE_KB": c["FETCH_SIZE"],
        "WRITE_SIZE_KB": c["WRITE_SIZE"],
        "hbm_bytes_per_launch_raw": (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
        "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024,
    }
    json.dump(rec, open(out / "hbm_traffic.json", "w"), indent=1)
    if len(sys.argv) > 3:
        table_path = root / "profiles" / "pmc_traffic.json"
        table = json.loads(table_path.read_text()) if table_path.exists() else {}
        table[sys.argv[3]] = dict(rec, source=f"profiles/{tag}/pmc_summary.csv")
        json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
print(open(out / "pmc_summary.csv").read())

#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/gpu_profile_run.sh (run on the GPU box) into small committable summaries."""
import collections
import csv
import json
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from kernel_sources_hash import built_kernel_sources_sha16, kernel_sources_sha16 as _sources_sha16  # noqa: E402


def kernel_sources_sha16():
    """of the library that ran (recorded at link time); the sources' own hash only if that record is missing"""
    return built_kernel_sources_sha16() or _sources_sha16()

tag, raw, out, command = sys.argv[1], pathlib.Path(sys.argv[2]), pathlib.Path(sys.argv[3]), sys.argv[4]
out.mkdir(parents=True, exist_ok=True)

stats = sorted(raw.glob("trace/*/*_kernel_stats.csv"), key=lambda p: p.stat().st_mtime)
if stats:
    (out / "kernel_stats.csv").write_text(stats[-1].read_text())
durations = []
for f in sorted(raw.glob("trace/*/*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "render_queue" in r["Kernel_Name"] or "preview_frame" in r["Kernel_Name"]:
            durations.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
if durations:
    (out / "render_kernel_durations_ms.txt").write_text(
        f"# every render-kernel dispatch of: {command}\n# (rocprofv3 --kernel-trace; the last 5 are the timed steps)\n" + "\n".join(f"{d:.4f}" for d in durations) + f"\n# mean of the last 5: {sum(durations[-5:]) / len(durations[-5:]):.4f} ms\n"
    )
rows, kernel = [], ""
for f in sorted(raw.glob("pmc_*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_queue" in r["Kernel_Name"] or "preview_frame" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernel = r["Kernel_Name"].split("(")[0]
    for k, v in sorted(agg.items()):
        rows.append((f.parts[-3], kernel, k, len(v), sum(v) / len(v)))
with open(out / "pmc_summary.csv", "w") as f:
    f.write(f"# {command}\n# kernel sources {kernel_sources_sha16()}, {time.strftime('%Y-%m-%d %H:%M:%S UTC', time.gmtime())}\n")
    f.write("pass,kernel,counter,dispatches,mean_per_dispatch\n")
    for r in rows:
        f.write(",".join(str(x) for x in r) + "\n")
c = {r[2]: r[4] for r in rows}
bench = {}
try:
    bench = json.loads((out / "bench_plain.jsonl").read_text().strip().splitlines()[-1])
except (OSError, ValueError, IndexError):
    pass
record = {
    "command": command,
    "kernel": kernel,
    "kernel_sources_sha16": kernel_sources_sha16(),
    "measured": time.strftime("%Y-%m-%d", time.gmtime()),
    "kernel_ms_traced_mean_of_timed_steps": round(sum(durations[-5:]) / max(len(durations[-5:]), 1), 4) if durations else None,
    "kernel_ms_hip_events_plain_run": bench.get("roofline", {}).get("kernel_ms"),
    "counters_per_launch": {k: c[k] for k in sorted(c)},
}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # MI355X_MICROARCH.md, HBM / rocprofv3: both counters are in KiB; on gfx950 FETCH_SIZE under-reports wide streaming reads
    # by 2x (corrected figure = 2 x FETCH + WRITE; the raw sum rides along)
    record["hbm_bytes_per_launch_raw"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    record["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
json.dump(record, open(out / "counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: record[k] for k in record if k != "counters_per_launch"}))
for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "FETCH_SIZE", "WRITE_SIZE"):
    if k in c:
        print(f"  {k} = {c[k]:.6g}")

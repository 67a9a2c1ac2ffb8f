#!/usr/bin/env python3
"""usage: [ROUND=r05] tools/merge_counters.py KEY TAG   (at home, after a gpurun call)
Copy gpurun_out/$ROUND/TAG/ to profiles/$ROUND/TAG/ and enter its counters.json into profiles/pmc_counters.json under KEY —
the workload key bench.py looks up: <scene>_<W>x<H>x<spp>_n<gpus>[_<forced kernel>][_fast][_tilt]."""
import json
import os
import pathlib
import shutil
import sys

root = pathlib.Path(__file__).resolve().parent.parent
key, tag = sys.argv[1], sys.argv[2]
rnd = os.environ.get("ROUND", "r05")
src, dst = root / "gpurun_out" / rnd / tag, root / "profiles" / rnd / tag
dst.mkdir(parents=True, exist_ok=True)
for f in src.iterdir():
    if f.is_file() and f.stat().st_size < 2_000_000:
        shutil.copy2(f, dst / f.name)
table_path = root / "profiles" / "pmc_counters.json"
table = json.loads(table_path.read_text()) if table_path.exists() else {}
table[key] = dict(json.loads((src / "counters.json").read_text()), source=f"profiles/{rnd}/{tag}/pmc_summary.csv")
json.dump(table, open(table_path, "w"), indent=1, sort_keys=True)
print(f"{key}: {table[key].get('hbm_bytes_per_launch')} HBM bytes per launch, kernel sources {table[key]['kernel_sources_sha16']}")

#!/usr/bin/env python3
"""tests/golden/frame_digests.json: sha256 of the ORACLE's packed RGBA8888 frame for the workloads bench.py times and the
BASELINE.json configurations at their full sizes (counter streams, seed 1) — so that the one line the driver runs can say
whether the frame it timed is the oracle's, bit for bit, without the oracle running on the GPU box (bench.py:
"frame_matches_oracle"), and so that tests/test_gpu_parity.py can hold whole frames against it.

Minutes of CPU in the build container (the oracle makes ~45 Mrays/s on its 8 cores).  Regenerate only together with an
arithmetic-contract change, like tools/gen_golden.py.  Config 5 (100 000 spheres) is covered by ONE stripe of 8 rows
(stripe 90 = rows 720..727, what tests/test_gpu_parity.py checks): the whole frame would be 2e13 sphere tests.

    python tools/gen_frame_digests.py [key ...]        (no key: all of them)"""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import rt_amd  # noqa: E402
from oracle import binding as oracle  # noqa: E402

OUT = ROOT / "tests" / "golden" / "frame_digests.json"
TILT = ((0.2, 1.2, 3.0), (0.0, -0.15, -1.0))  # bench.py --tilt, tools/gpu_ab.py AB_TILT

# key -> scene, width, height, spp, tilted camera, partition (rank, world, stripe rows) or None
WORKLOADS = {
    "headline": ("basic", 1920, 1080, 256, False, None),
    "config2": ("basic", 1920, 1080, 64, False, None),
    "config3": ("dielectric", 1920, 1080, 256, False, None),
    "config4": ("basic", 3840, 2160, 256, False, None),
    "config5_stripe90": ("synthetic-100k", 1920, 1080, 64, False, (90, 135, 8)),
    "interactive": ("basic_plane", 1920, 1080, 256, True, None),
    "config1": ("basic", 256, 256, 1, False, None),
}


def workload_key(scene, width, height, spp, max_bounces, seed, tilt):
    """what bench.py looks a frame up by"""
    return f"{scene} {width}x{height} {spp} spp max_bounces {max_bounces} seed {seed}" + (" tilt" if tilt else "")


def main():
    digests = json.loads(OUT.read_text()) if OUT.exists() else {}
    for key in sys.argv[1:] or WORKLOADS:
        name, width, height, spp, tilt, partition = WORKLOADS[key]
        scene = rt_amd.Scene.named(name).set_sampling(spp)
        if tilt:
            scene.set_camera(*TILT)
        pod = scene.describe(width, height)
        rgba, _, stats = oracle.render(pod, width, height, seed=1, partition=partition, want_rgb=False)
        entry = {
            "scene": name, "width": width, "height": height, "spp": spp, "max_bounces": pod.max_bounces, "seed": 1, "tilt": tilt,
            "partition": list(partition) if partition else None,
            "workload": workload_key(name, width, height, spp, pod.max_bounces, 1, tilt),
            "rows": int(rgba.shape[0]), "segments": int(stats["segments"]),
            "sha256": hashlib.sha256(rgba.tobytes()).hexdigest(),
            "contract": "v4",
        }
        digests[key] = entry
        print(key, entry["sha256"][:16], f"{stats['seconds']:.1f} s", flush=True)
        OUT.write_text(json.dumps(digests, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (oracle/liboracle.so).

The reference ships no golden vectors and cannot be run, so these fixtures are outputs of THIS
project's oracle (arithmetic contract v4: per-pixel keyed random streams, one generator step per random<T>() call, single-step
reciprocal square root, primary rays from a per-pixel base).  They freeze the contract: a change to the
oracle or to the kernels that alters a single bit of any fixture is caught by tests/test_golden.py (CPU) and
tests/test_gpu_parity.py (GPU).  Regenerate only together with a contract version bump.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import rt_amd  # noqa: E402
from oracle import binding as oracle  # noqa: E402
from tests.conftest import PLANES_SCENE, PREVIEW_SCENE  # noqa: E402

OUT = ROOT / "tests" / "golden"
OUT.mkdir(parents=True, exist_ok=True)

FRAMES = [
    # fixture name, scene, width, height, spp, max_bounces, seed
    ("basic_64x36_spp4", "basic", 64, 36, 4, 10, 1),
    ("dielectric_64x36_spp4", "dielectric", 64, 36, 4, 10, 2),
    ("planes_48x27_spp4", "planes", 48, 27, 4, 6, 3),
    ("synthetic1500_32x18_spp2", "synthetic-1500", 32, 18, 2, 10, 4),
]

for fixture, name, width, height, spp, bounces, seed in FRAMES:
    scene = rt_amd.Scene.parse(PLANES_SCENE) if name == "planes" else rt_amd.Scene.named(name)
    scene.set_sampling(spp, bounces)
    rgba, rgb, stats = oracle.render(scene.describe(width, height), width, height, seed=seed)
    np.savez_compressed(
        OUT / f"{fixture}.npz", scene=name, width=width, height=height, spp=spp, max_bounces=bounces, seed=seed, rgba=rgba, rgb=rgb, segments=stats["segments"]
    )
    print(fixture, stats)

# the preview (RT_HIP_FLAG_PREVIEW): planes, boxes and spheres, one ray per pixel
width, height = 96, 54
rgba, rgb, stats = oracle.render(rt_amd.Scene.parse(PREVIEW_SCENE).describe(width, height), width, height, preview=True)
np.savez_compressed(OUT / "preview_96x54.npz", width=width, height=height, rgba=rgba, rgb=rgb)
print("preview_96x54", stats)

seed, pixel, sample = 0x0123456789ABCDEF, 987654, 42
np.savez_compressed(OUT / "random_stream.npz", seed=np.uint64(seed), pixel=pixel, sample=sample, draws=oracle.random(seed, pixel, sample, 256))

# closest-hit vectors on the planes scene (tangent, inside-origin and behind-the-ray cases included)
scene = rt_amd.Scene.parse(PLANES_SCENE)
pod = scene.describe(48, 27)
rng = np.random.default_rng(17)
origins = rng.uniform(-3, 3, (1024, 3)).astype(np.float32)
dirs = rng.normal(size=(1024, 3))
dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
origins[:8] = [(0, 1, 0), (0, 1, 0.5), (0, 2.0, 0), (0, 5, 0), (0, 1, 5), (1, 1, 5), (0.999, 1, 5), (1.001, 1, 5)]
dirs[:8] = [(0, 0, -1), (1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (0, 0, -1), (0, 0, -1)]
dist, kind, index, normal = oracle.closest_hit(pod, origins, dirs)
np.savez_compressed(OUT / "closest_hit_planes.npz", origins=origins, directions=dirs, distance=dist, kind=kind, index=index, normal=normal)
print("closest-hit kinds:", np.bincount(kind))

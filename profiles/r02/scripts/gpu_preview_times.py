#!/usr/bin/env python3
"""Kernel time of the preview (RT_HIP_FLAG_PREVIEW) at 1920x1080 and at rt's low-resolution preview size."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import rt_amd  # noqa: E402
from rt_amd import capi  # noqa: E402

tracer = rt_amd.HipRayTracer(0)
for name in ("basic", "dielectric", "synthetic-1000", "synthetic-100k"):
    for width, height in ((1920, 1080), (480, 270)):
        scene = rt_amd.Scene.named(name)
        tracer.upload(scene.describe(width, height))
        frame = torch.empty((height, width), dtype=torch.int32, device="cuda:0")
        best = 1e9
        for _ in range(5):
            tracer.render_device(width, height, frame.data_ptr(), flags=capi.RT_HIP_FLAG_PREVIEW)
            best = min(best, tracer.stats()["render_ms"])
        print(f"{name:16s} {width}x{height}: {best:9.4f} ms  ({width * height / best / 1e3:9.1f} Mrays/s)")

"""rt_amd — MI355X (gfx950) renderer module for marzer/rt's mg_ray_tracer path, plus its host-side harness.

Product code only.  The CPU oracle is test infrastructure and lives in ``oracle/`` (never imported from here).
"""
from .capi import RtHipError, RtHipPartition, RtHipScene, RtHipStats  # noqa: F401
from .renderer import HipRayTracer, device_count, live_frame_locks, local_rows, padded_local_rows, scene_check, unique_id  # noqa: F401
from .scene import Scene, SceneError, scene_from_arrays  # noqa: F401

__all__ = [
    "HipRayTracer",
    "RtHipError",
    "RtHipPartition",
    "RtHipScene",
    "RtHipStats",
    "Scene",
    "SceneError",
    "device_count",
    "live_frame_locks",
    "local_rows",
    "padded_local_rows",
    "scene_check",
    "scene_from_arrays",
    "unique_id",
]

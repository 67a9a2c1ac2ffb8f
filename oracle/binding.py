"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE, NOT PRODUCT.  PARITY UNPINNED (see cpu_ref.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It takes the same
``rt_hip_scene`` / ``rt_hip_partition`` PODs as the HIP module so that both see identical bytes.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from rt_amd.capi import RtHipPartition, RtHipScene

ORACLE_DIR = Path(__file__).resolve().parent
TRACE_ITERATIVE = 0
TRACE_RECURSIVE = 1
MATERIALS_SM = 2
PREVIEW = 4


class OracleStats(C.Structure):
    _fields_ = [
        ("primary_samples", C.c_uint64),
        ("segments", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("plane_tests", C.c_uint64),
        ("seconds", C.c_double),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


_lib = None


def build() -> Path:
    subprocess.run(["make", "-C", str(ORACLE_DIR)], check=True, capture_output=True)
    return ORACLE_DIR / "liboracle.so"


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        path = ORACLE_DIR / "liboracle.so"
        if not path.exists():
            build()
        l = C.CDLL(str(path))
        l.oracle_render.restype = C.c_int
        l.oracle_render.argtypes = [C.POINTER(RtHipScene), C.c_uint32, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(RtHipPartition), C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OracleStats)]
        l.oracle_render_mt19937.restype = C.c_int
        l.oracle_render_mt19937.argtypes = [C.POINTER(RtHipScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(OracleStats)]
        l.oracle_random.restype = None
        l.oracle_random.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        l.oracle_stream_keys.restype = None
        l.oracle_stream_keys.argtypes = [C.c_uint64, C.c_uint32] + [C.c_void_p] * 5
        l.oracle_closest_hit.restype = None
        l.oracle_closest_hit.argtypes = [C.POINTER(RtHipScene), C.c_uint32] + [C.c_void_p] * 6
        l.oracle_sqrt_div.restype = None
        l.oracle_sqrt_div.argtypes = [C.c_uint32] + [C.c_void_p] * 4
        l.oracle_inv_sqrt.restype = None
        l.oracle_inv_sqrt.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p]
        l.oracle_inv_sqrt_step.restype = None
        l.oracle_inv_sqrt_step.argtypes = [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        l.oracle_pack.restype = C.c_uint32
        l.oracle_pack.argtypes = [C.c_float] * 3
        l.oracle_sky.restype = None
        l.oracle_sky.argtypes = [C.c_float, C.c_void_p]
        l.oracle_primary_ray.restype = C.c_int
        l.oracle_primary_ray.argtypes = [C.POINTER(RtHipScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        l.oracle_hits_box.restype = C.c_int
        l.oracle_hits_box.argtypes = [C.c_void_p] * 5
        l.oracle_dielectric_direction.restype = None
        l.oracle_dielectric_direction.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        _lib = l
    return _lib


def default_threads() -> int:
    """threads = 0 means "all the host gives": the CPUs this process may run on, capped by the CPU TIME its cgroup grants (a GPU
    box shows 256 CPUs to a job that is granted the time of 16 — 256 threads there only take turns)."""
    import math
    import os

    allowed = len(os.sched_getaffinity(0))
    for quota_file, period_file in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:
            if period_file is None:
                quota, period = open(quota_file).read().split()[:2]
            else:
                quota, period = open(quota_file).read().strip(), open(period_file).read().strip()
            if quota != "max" and float(quota) > 0 and float(period) > 0:
                return max(1, min(allowed, math.ceil(float(quota) / float(period))))
            break
        except (OSError, ValueError):
            continue
    return allowed


def _local_rows(height, rank, world, stripe):
    return sum(1 for y in range(height) if (y // stripe) % world == rank)


def render(scene: RtHipScene, width: int, height: int, seed: int = 1, trace_order: int = TRACE_ITERATIVE, partition=None, want_rgb=True, threads: int = 0, sm_materials: bool = False, preview: bool = False):
    """Counter-RNG strict-IEEE render (preview=True: the one-ray-per-pixel preview of rasterizer.cpp instead).  Returns (rgba uint32[rows, W], rgb float32[rows, W, 3] | None, stats dict)."""
    rows = height if partition is None else _local_rows(height, *partition)
    rgba = np.zeros((rows, width), dtype=np.uint32)
    rgb = np.zeros((rows, width, 3), dtype=np.float32) if want_rgb else None
    stats = OracleStats()
    part = C.byref(RtHipPartition(*partition)) if partition is not None else None
    rc = lib().oracle_render(C.byref(scene), width, height, seed, trace_order | (MATERIALS_SM if sm_materials else 0) | (PREVIEW if preview else 0), part, rgba.ctypes.data, rgb.ctypes.data if rgb is not None else None, threads or default_threads(), C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"oracle_render failed ({rc})")
    return rgba, rgb, stats.as_dict()


def render_mt19937(scene: RtHipScene, width: int, height: int, fixed_seed: int = 0, want_rgb=False, threads: int = 0):
    rgba = np.zeros((height, width), dtype=np.uint32)
    rgb = np.zeros((height, width, 3), dtype=np.float32) if want_rgb else None
    stats = OracleStats()
    rc = lib().oracle_render_mt19937(C.byref(scene), width, height, fixed_seed, rgba.ctypes.data, rgb.ctypes.data if rgb is not None else None, threads or default_threads(), C.byref(stats))
    if rc != 0:
        raise RuntimeError(f"oracle_render_mt19937 failed ({rc})")
    return rgba, rgb, stats.as_dict()


def random(seed: int, pixel: int, sample: int, n: int) -> np.ndarray:
    """The first n numbers of the stream of (pixel, sample), flattened: the three draws of its first generator step, then
    of its second, ... (contract v4: one step per random<T>() call)."""
    out = np.empty(n, dtype=np.float32)
    lib().oracle_random(seed, pixel, sample, n, out.ctypes.data)
    return out


def stream_keys(seed: int, pixels, samples):
    """(function key, stride, counter before the first draw) of the random stream of every (pixel, sample) pair."""
    pixels = np.ascontiguousarray(pixels, dtype=np.uint32)
    samples = np.ascontiguousarray(samples, dtype=np.uint32)
    assert pixels.shape == samples.shape and pixels.ndim == 1
    function_key = np.empty_like(pixels)
    stride = np.empty_like(pixels)
    counter = np.empty_like(pixels)
    lib().oracle_stream_keys(seed, pixels.size, pixels.ctypes.data, samples.ctypes.data, function_key.ctypes.data, stride.ctypes.data, counter.ctypes.data)
    return function_key, stride, counter


def closest_hit(scene: RtHipScene, origins, directions):
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    n = len(o)
    dist = np.empty(n, dtype=np.float32)
    kind = np.empty(n, dtype=np.uint32)
    index = np.empty(n, dtype=np.uint32)
    normal = np.empty((n, 3), dtype=np.float32)
    lib().oracle_closest_hit(C.byref(scene), n, o.ctypes.data, d.ctypes.data, dist.ctypes.data, kind.ctypes.data, index.ctypes.data, normal.ctypes.data)
    return dist, kind, index, normal


def sqrt_div(a, b):
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    s = np.empty_like(a)
    q = np.empty_like(a)
    lib().oracle_sqrt_div(a.size, a.ctypes.data, b.ctypes.data, s.ctypes.data, q.ctypes.data)
    return s, q


def inv_sqrt(x):
    """The arithmetic contract's reciprocal square root (what normalize() multiplies by)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_inv_sqrt(x.size, x.ctypes.data, out.ctypes.data)
    return out


def inv_sqrt_step(x, estimate):
    """The Newton-Raphson step of the contract's reciprocal square root, from a caller-chosen estimate."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    estimate = np.ascontiguousarray(estimate, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_inv_sqrt_step(x.size, x.ctypes.data, estimate.ctypes.data, out.ctypes.data)
    return out


def hits_box(origin, direction, center, extents):
    """(hit, t) of the preview's ray-box test."""
    arrays = [np.ascontiguousarray(v, dtype=np.float32) for v in (origin, direction, center, extents)]
    t = np.zeros(1, dtype=np.float32)
    hit = lib().oracle_hits_box(*(a.ctypes.data for a in arrays), t.ctypes.data)
    return bool(hit), float(t[0])


def pack(r: float, g: float, b: float) -> int:
    return lib().oracle_pack(r, g, b)


def sky(dir_y: float) -> np.ndarray:
    out = np.empty(3, dtype=np.float32)
    lib().oracle_sky(dir_y, out.ctypes.data)
    return out


def primary_ray(scene: RtHipScene, width: int, height: int, x: int, y: int, ka: float = 2.0**23, kb: float = 2.0**23, want_form: bool = False):
    """(origin, direction) of the primary ray of pixel (x, y) whose jitter is (ka, kb) * 2^-24 — the numerators of one
    generator step; the default is the pixel centre.  want_form: also which form the matrix was given: 'pinhole', 'eye' or 'general'."""
    o = np.empty(3, dtype=np.float32)
    d = np.empty(3, dtype=np.float32)
    pinhole = lib().oracle_primary_ray(C.byref(scene), width, height, x, y, ka, kb, o.ctypes.data, d.ctypes.data)
    return (o, d, {1: "pinhole", 2: "eye", 0: "general"}[pinhole]) if want_form else (o, d)


def dielectric_direction(direction, normal, reflectivity: float, u: float):
    """(new direction, reflect probability) of sm_ray_tracer's dielectric_scatter for the uniform number u."""
    d = np.ascontiguousarray(direction, dtype=np.float32)
    n = np.ascontiguousarray(normal, dtype=np.float32)
    out = np.empty(3, dtype=np.float32)
    prob = C.c_float()
    lib().oracle_dielectric_direction(d.ctypes.data, n.ctypes.data, reflectivity, u, out.ctypes.data, C.byref(prob))
    return out, prob.value

"""Scenes for the harness: a thin Python handle on the C++ host side (rt_amd/host, librt_host.so).

``Scene`` wraps ``rt::scene`` (mirror of reference src/scene.hpp:8-25): loading from rt scene files
(reference src/scene.cpp:483-618), the synthetic 100k-sphere generator of SURVEY.md §8d, and
``describe(width, height)`` which produces the ``rt_hip_scene`` POD handed across the C ABI, including
``camera.viewport(size).inverse_view_projection`` (reference src/camera.hpp:122-137).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import capi

SCENES_DIR = Path(__file__).resolve().parent.parent / "scenes"


class SceneError(RuntimeError):
    pass


class Scene:
    def __init__(self, handle: int):
        if not handle:
            raise SceneError(capi.host_lib().rt_host_last_error().decode(errors="replace"))
        self._handle = C.c_void_p(handle)
        # bound now: at interpreter shutdown the module globals that capi.host_lib() needs may already be gone (round 3 saw
        # "TypeError: 'NoneType' object is not callable" from __del__ at exit)
        self._free = capi.host_lib().rt_host_scene_free

    # ---- constructors -------------------------------------------------------------------------------------
    @classmethod
    def parse(cls, toml_text: str) -> "Scene":
        return cls(capi.host_lib().rt_host_scene_parse(toml_text.encode()))

    @classmethod
    def load(cls, path) -> "Scene":
        return cls(capi.host_lib().rt_host_scene_load(str(path).encode()))

    @classmethod
    def synthetic(cls, sphere_count: int = 100000) -> "Scene":
        return cls(capi.host_lib().rt_host_scene_synthetic(sphere_count))

    @classmethod
    def named(cls, name: str) -> "Scene":
        """'basic', 'dielectric' (scenes/*.toml) or 'synthetic-<N>' / 'synthetic-100k'."""
        if name.startswith("synthetic"):
            suffix = name.partition("-")[2] or "100k"
            count = int(suffix[:-1]) * 1000 if suffix.endswith("k") else int(suffix)
            return cls.synthetic(count)
        return cls.load(SCENES_DIR / f"{name}.toml")

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        free = getattr(self, "_free", None)
        if handle and free is not None:
            free(handle)

    # ---- mutation -------------------------------------------------------------------------------------------
    def set_sampling(self, samples_per_pixel: int = 0, max_bounces: int = 0) -> "Scene":
        capi.host_lib().rt_host_scene_set_sampling(self._handle, samples_per_pixel, max_bounces)
        return self

    def set_camera(self, position, direction) -> "Scene":
        capi.host_lib().rt_host_scene_set_camera(self._handle, (C.c_float * 3)(*position), (C.c_float * 3)(*direction))
        return self

    # ---- the ABI view ---------------------------------------------------------------------------------------
    def describe(self, width: int, height: int) -> capi.RtHipScene:
        """rt_hip_scene for a width x height frame.  Column pointers stay valid while this Scene is alive."""
        out = capi.RtHipScene()
        if capi.host_lib().rt_host_scene_describe(self._handle, width, height, C.byref(out)) != 0:
            raise SceneError(capi.host_lib().rt_host_last_error().decode(errors="replace"))
        out._owner = self  # keep the columns alive as long as the POD is
        return out

    def screen_to_world(self, width: int, height: int, x: float, y: float, depth: float = 0.0) -> np.ndarray:
        out = (C.c_float * 3)()
        capi.host_lib().rt_host_screen_to_world(self._handle, width, height, x, y, depth, out)
        return np.array(out[:], dtype=np.float32)


def column(ptr, count: int, dtype=np.float32) -> np.ndarray:
    """Copy `count` elements of an ABI column pointer into a numpy array."""
    if not count:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(count,)).astype(dtype, copy=True)


def scene_from_arrays(
    spheres=None,
    planes=None,
    materials=None,
    boxes=None,
    samples_per_pixel: int = 1,
    max_bounces: int = 10,
    inverse_view_projection=None,
) -> capi.RtHipScene:
    """Build an rt_hip_scene directly from Python data (tests of ragged / empty / edge-case scenes).

    spheres: rows (cx, cy, cz, radius, material); planes: rows (nx, ny, nz, d, material);
    materials: rows (type, r, g, b, a, roughness, reflectivity); boxes: rows (cx, cy, cz, ex, ey, ez, material),
    drawn by the preview only.
    """
    out = capi.RtHipScene()
    keep = []

    def col(values, dtype):
        arr = np.ascontiguousarray(values, dtype=dtype)
        keep.append(arr)
        ctype = C.c_float if dtype == np.float32 else C.c_uint32
        return arr.ctypes.data_as(C.POINTER(ctype)) if arr.size else None

    s = np.asarray(spheres if spheres is not None else np.zeros((0, 5)), dtype=np.float64).reshape(-1, 5)
    p = np.asarray(planes if planes is not None else np.zeros((0, 5)), dtype=np.float64).reshape(-1, 5)
    m = np.asarray(materials if materials is not None else np.zeros((0, 7)), dtype=np.float64).reshape(-1, 7)
    out.n_spheres = len(s)
    out.sphere_center_x = col(s[:, 0], np.float32)
    out.sphere_center_y = col(s[:, 1], np.float32)
    out.sphere_center_z = col(s[:, 2], np.float32)
    out.sphere_radius = col(s[:, 3], np.float32)
    out.sphere_material = col(s[:, 4], np.uint32)
    out.n_planes = len(p)
    out.plane_normal_x = col(p[:, 0], np.float32)
    out.plane_normal_y = col(p[:, 1], np.float32)
    out.plane_normal_z = col(p[:, 2], np.float32)
    out.plane_d = col(p[:, 3], np.float32)
    out.plane_material = col(p[:, 4], np.uint32)
    out.n_materials = len(m)
    out.material_type = col(m[:, 0], np.uint32)
    out.material_albedo = col(m[:, 1:5].reshape(-1), np.float32)
    out.material_roughness = col(m[:, 5], np.float32)
    out.material_reflectivity = col(m[:, 6], np.float32)
    b = np.asarray(boxes if boxes is not None else np.zeros((0, 7)), dtype=np.float64).reshape(-1, 7)
    out.n_boxes = len(b)
    out.box_center_x = col(b[:, 0], np.float32)
    out.box_center_y = col(b[:, 1], np.float32)
    out.box_center_z = col(b[:, 2], np.float32)
    out.box_extents_x = col(b[:, 3], np.float32)
    out.box_extents_y = col(b[:, 4], np.float32)
    out.box_extents_z = col(b[:, 5], np.float32)
    out.box_material = col(b[:, 6], np.uint32)
    out.samples_per_pixel = samples_per_pixel
    out.max_bounces = max_bounces
    ivp = np.eye(4, dtype=np.float32) if inverse_view_projection is None else np.asarray(inverse_view_projection, dtype=np.float32)
    out.inverse_view_projection = (C.c_float * 16)(*ivp.reshape(-1))
    out._owner = keep
    return out

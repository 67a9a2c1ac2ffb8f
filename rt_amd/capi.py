"""ctypes bindings of the two product libraries.

* ``librt_hip.so``  — include/rt_hip.h: the C ABI of the gfx950 renderer module (the drop-in boundary for
  reference ``src/renderers/mg_ray_tracer.cpp:178-205``).
* ``librt_host.so`` — rt_amd/host/host_capi.h: C++ host side (scene files, camera -> inverse view-projection).

Nothing here computes pixels: if ``librt_hip.so`` is missing or no gfx950 device is visible, the calls fail
loudly (``RtHipError``).  There is no CPU fallback in the product; the CPU oracle lives under ``oracle/`` and is
only ever loaded by tests, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

LIB_DIR = Path(__file__).resolve().parent / "lib"

RT_HIP_ABI_VERSION = 6
RT_HIP_DEFAULT_STRIPE_ROWS = 8
RT_HIP_FLAG_FORCE_TILED = 1 << 0
RT_HIP_FLAG_FORCE_RESIDENT = 1 << 1
RT_HIP_FLAG_PERSISTENT_FRAME = 1 << 2
RT_HIP_FLAG_SM_MATERIALS = 1 << 3
RT_HIP_FLAG_PREVIEW = 1 << 4
RT_HIP_FLAG_FORCE_STREAMED = 1 << 5
RT_HIP_FLAG_FAST = 1 << 6
RT_HIP_FLAG_STATS = 1 << 7
RT_HIP_FLAG_FORCE_HALF_CHUNKS = 1 << 8
RT_HIP_FLAG_FORCE_WHOLE_CHUNKS = 1 << 9
RT_HIP_MULTI_PEER_COPY = 1 << 0
RT_HIP_MULTI_DIRECT_FRAME = 1 << 1
RT_HIP_TRANSPORT_NONE, RT_HIP_TRANSPORT_RCCL_GATHER, RT_HIP_TRANSPORT_PEER_COPY, RT_HIP_TRANSPORT_DIRECT_FRAME = 0, 1, 2, 3
TRANSPORT_NAMES = {0: "none", 1: "rccl_gather", 2: "peer_copy", 3: "direct_frame", 4: "shared_frame"}
KERNEL_NAMES = {0: "none", 1: "resident", 2: "tiled", 3: "small", 4: "preview", 5: "streamed"}

STATUS_NAMES = {
    0: "RT_HIP_OK",
    1: "RT_HIP_INVALID_ARGUMENT",
    2: "RT_HIP_NO_DEVICE",
    3: "RT_HIP_RUNTIME_ERROR",
    4: "RT_HIP_NO_SCENE",
    5: "RT_HIP_UNSUPPORTED",
    6: "RT_HIP_TIMEOUT",
}

c_float_p = C.POINTER(C.c_float)
c_u32_p = C.POINTER(C.c_uint32)


class RtHipError(RuntimeError):
    def __init__(self, status: int, message: str):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class RtHipScene(C.Structure):
    """``rt_hip_scene`` (include/rt_hip.h)."""

    _fields_ = [
        ("n_spheres", C.c_uint32),
        ("sphere_center_x", c_float_p),
        ("sphere_center_y", c_float_p),
        ("sphere_center_z", c_float_p),
        ("sphere_radius", c_float_p),
        ("sphere_material", c_u32_p),
        ("n_planes", C.c_uint32),
        ("plane_normal_x", c_float_p),
        ("plane_normal_y", c_float_p),
        ("plane_normal_z", c_float_p),
        ("plane_d", c_float_p),
        ("plane_material", c_u32_p),
        ("n_materials", C.c_uint32),
        ("material_type", c_u32_p),
        ("material_albedo", c_float_p),
        ("material_roughness", c_float_p),
        ("material_reflectivity", c_float_p),
        ("samples_per_pixel", C.c_uint32),
        ("max_bounces", C.c_uint32),
        ("inverse_view_projection", C.c_float * 16),
        ("n_boxes", C.c_uint32),
        ("box_center_x", c_float_p),
        ("box_center_y", c_float_p),
        ("box_center_z", c_float_p),
        ("box_extents_x", c_float_p),
        ("box_extents_y", c_float_p),
        ("box_extents_z", c_float_p),
        ("box_material", c_u32_p),
    ]


class RtHipPartition(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32), ("stripe_rows", C.c_uint32)]


class RtHipStats(C.Structure):
    _fields_ = [
        ("primary_samples", C.c_uint64),
        ("segments", C.c_uint64),
        ("sphere_tests", C.c_uint64),
        ("plane_tests", C.c_uint64),
        ("render_ms", C.c_float),
        ("upload_ms", C.c_float),
        ("readback_ms", C.c_float),
        ("kernel_variant", C.c_uint32),
    ]

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["kernel"] = KERNEL_NAMES.get(self.kernel_variant, str(self.kernel_variant))
        return d


class RtHipPhases(C.Structure):
    """``rt_hip_phases`` (include/rt_hip.h)."""

    _fields_ = [
        ("render_ms", C.c_float),
        ("gather_ms", C.c_float),
        ("assemble_ms", C.c_float),
        ("copy_ms", C.c_float),
        ("host_issue_ms", C.c_float),
        ("host_wait_ms", C.c_float),
        ("transport", C.c_uint32),
        ("scene_resident", C.c_uint32),
        ("carrier_bands", C.c_uint32),
        ("carrier_bands_early", C.c_uint32),
        ("carrier_helpers", C.c_uint32),
    ]

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_}
        d["transport"] = TRANSPORT_NAMES.get(self.transport, str(self.transport))
        return d


# every symbol include/rt_hip.h declares: (name, restype, argtypes)
RT_HIP_SYMBOLS = [
    ("rt_hip_abi_version", C.c_uint32, []),
    ("rt_hip_last_error", C.c_char_p, []),
    ("rt_hip_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("rt_hip_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    ("rt_hip_destroy", None, [C.c_void_p]),
    ("rt_hip_create_multi", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.c_uint32]),
    ("rt_hip_unique_id", C.c_int, [C.c_char * 128]),
    ("rt_hip_create_rank", C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_char * 128]),
    ("rt_hip_join_ranks", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char * 128, C.c_uint32]),
    ("rt_hip_join_frame_group", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_uint32]),
    ("rt_hip_comm_info", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), c_u32_p]),
    ("rt_hip_member_count", C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    ("rt_hip_member_device", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    ("rt_hip_member_stats", C.c_int, [C.c_void_p, C.c_int, C.POINTER(RtHipStats)]),
    ("rt_hip_local_rows", C.c_int, [C.c_uint32, C.POINTER(RtHipPartition), c_u32_p]),
    ("rt_hip_padded_local_rows", C.c_int, [C.c_uint32, C.POINTER(RtHipPartition), c_u32_p]),
    ("rt_hip_scene_check", C.c_int, [C.POINTER(RtHipScene), C.POINTER(C.c_uint64)]),
    ("rt_hip_scene_upload", C.c_int, [C.c_void_p, C.POINTER(RtHipScene)]),
    (
        "rt_hip_render_device",
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.POINTER(RtHipPartition), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    ("rt_hip_assemble_device", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_hip_stats_fetch", C.c_int, [C.c_void_p, C.POINTER(RtHipStats)]),
    ("rt_hip_phases_fetch", C.c_int, [C.c_void_p, C.POINTER(RtHipPhases)]),
    (
        "rt_hip_render",
        C.c_int,
        [C.c_void_p, C.POINTER(RtHipScene), C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.POINTER(RtHipStats)],
    ),
    ("rt_hip_forget_frame", None, [C.c_void_p]),
    ("rt_hip_live_frame_locks", C.c_uint32, []),
]

# every symbol include/rt_hip_kat.h declares — librt_hip_kat.so, the TEST-ONLY companion library (known-answer entry points)
RT_HIP_KAT_SYMBOLS = [
    ("rt_hip_kat_last_error", C.c_char_p, []),
    ("rt_hip_kat_random", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    ("rt_hip_kat_closest_hit", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_hip_kat_sqrt_div", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rt_hip_kat_exhaustive_math", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64 * 3), C.POINTER(C.c_uint32 * 3)]),
]

RT_HOST_SYMBOLS = [
    ("rt_host_last_error", C.c_char_p, []),
    ("rt_host_scene_parse", C.c_void_p, [C.c_char_p]),
    ("rt_host_scene_load", C.c_void_p, [C.c_char_p]),
    ("rt_host_scene_synthetic", C.c_void_p, [C.c_uint]),
    ("rt_host_scene_free", None, [C.c_void_p]),
    ("rt_host_scene_set_sampling", None, [C.c_void_p, C.c_uint, C.c_uint]),
    ("rt_host_scene_set_camera", None, [C.c_void_p, C.c_float * 3, C.c_float * 3]),
    ("rt_host_scene_describe", C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.POINTER(RtHipScene)]),
    ("rt_host_screen_to_world", None, [C.c_void_p, C.c_uint, C.c_uint, C.c_float, C.c_float, C.c_float, C.c_float * 3]),
    ("rt_host_named_colour", C.c_int, [C.c_char_p, C.c_float * 4]),
    ("rt_host_toml_to_json", C.c_long, [C.c_char_p, C.c_char_p, C.c_ulong]),
]


def _bind(lib: C.CDLL, symbols) -> C.CDLL:
    for name, restype, argtypes in symbols:
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


_hip_lib = None
_host_lib = None
_kat_lib = None


def hip_library_path() -> Path:
    return Path(os.environ.get("RT_HIP_LIBRARY", LIB_DIR / "librt_hip.so"))


def kat_library_path() -> Path:
    """librt_hip_kat.so lies next to the librt_hip.so in use (an experiment build carries the entry points itself)."""
    hip = hip_library_path()
    return hip if hip.name != "librt_hip.so" else hip.with_name("librt_hip_kat.so")


def host_library_path() -> Path:
    return Path(os.environ.get("RT_HOST_LIBRARY", LIB_DIR / "librt_host.so"))


def hip_lib() -> C.CDLL:
    """The HIP module.  Raises if it has not been built — there is no fallback."""
    global _hip_lib
    if _hip_lib is None:
        # PyTorch-ROCm ships its own copy of the HIP runtime.  A process that is going to use both (bench.py, the
        # multi-GPU path, some tests) must let torch load its copy FIRST: if librt_hip.so pulls in /opt/rocm's runtime
        # before `import torch`, torch later finds "No HIP GPUs".  Loading torch here pins the order for the harness;
        # C++ users of librt_hip.so (rt itself, rt_headless) never see torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = hip_library_path()
        if not path.exists():
            raise RtHipError(2, f"{path} not found: build it with `make` (or __graft_entry__.build())")
        _hip_lib = _bind(C.CDLL(str(path)), RT_HIP_SYMBOLS)
        got = _hip_lib.rt_hip_abi_version()
        if got != RT_HIP_ABI_VERSION:
            raise RtHipError(5, f"{path} has ABI version {got}, bindings expect {RT_HIP_ABI_VERSION}")
    return _hip_lib


def kat_lib() -> C.CDLL:
    """The test-only library of known-answer entry points (include/rt_hip_kat.h).  Tests only."""
    global _kat_lib
    if _kat_lib is None:
        hip_lib()  # first: librt_hip_kat.so depends on it (and torch's copy of the HIP runtime must come before both)
        path = kat_library_path()
        if not path.exists():
            raise RtHipError(2, f"{path} not found: build it with `make` (or __graft_entry__.build())")
        _kat_lib = _bind(C.CDLL(str(path)), RT_HIP_KAT_SYMBOLS)
    return _kat_lib


def check_kat(status: int) -> None:
    if status != 0:
        raise RtHipError(status, kat_lib().rt_hip_kat_last_error().decode(errors="replace"))


def host_lib() -> C.CDLL:
    global _host_lib
    if _host_lib is None:
        path = host_library_path()
        if not path.exists():
            raise RuntimeError(f"{path} not found: build it with `make` (or __graft_entry__.build())")
        _host_lib = _bind(C.CDLL(str(path)), RT_HOST_SYMBOLS)
    return _host_lib


def check(status: int) -> None:
    if status != 0:
        raise RtHipError(status, hip_lib().rt_hip_last_error().decode(errors="replace"))

"""``HipRayTracer`` — the harness-side handle on one ``rt_hip_ctx`` (one GPU).

Mirrors the lifetime and the call shape of a renderer in the reference: created once
(``renderers::description::create``, reference src/renderer.hpp:39), asked to ``render(scene, pixels)``
(src/renderer.hpp:11, src/renderers/mg_ray_tracer.cpp:178), destroyed through ``close()``.

Two ways in, both straight through the C ABI of include/rt_hip.h:

* ``render(scene_pod, width, height, ...)`` -> numpy ``uint32[H, W]`` : the drop-in ``rt_hip_render`` (upload,
  kernels, copy into a HOST frame) — what the reference-side shim calls.
* ``upload`` + ``render_device`` (+ ``assemble_device``): the scene stays resident in HBM and frames are
  written to DEVICE buffers (raw pointers, e.g. of torch tensors) on a caller-chosen stream — what
  ``bench.py`` and the multi-GPU path use.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import RtHipPartition, RtHipPhases, RtHipScene, RtHipStats, check


def local_rows(height: int, rank: int, world: int, stripe_rows: int = capi.RT_HIP_DEFAULT_STRIPE_ROWS) -> int:
    out = C.c_uint32()
    check(capi.hip_lib().rt_hip_local_rows(height, C.byref(RtHipPartition(rank, world, stripe_rows)), C.byref(out)))
    return out.value


def padded_local_rows(height: int, world: int, stripe_rows: int = capi.RT_HIP_DEFAULT_STRIPE_ROWS) -> int:
    out = C.c_uint32()
    check(capi.hip_lib().rt_hip_padded_local_rows(height, C.byref(RtHipPartition(0, world, stripe_rows)), C.byref(out)))
    return out.value


def scene_check(scene: RtHipScene) -> int:
    """rt_hip_scene_check: validate the columns as rt_hip_render would and return their fingerprint (pure host code)."""
    out = C.c_uint64()
    check(capi.hip_lib().rt_hip_scene_check(C.byref(scene), C.byref(out)))
    return out.value


def unique_id() -> bytes:
    """A fresh id for a renderer with one process per GPU (rt_hip_unique_id): create it once, give it to every rank."""
    buf = (C.c_char * 128)()
    check(capi.hip_lib().rt_hip_unique_id(buf))
    return bytes(buf.raw)


def live_frame_locks() -> int:
    """rt_hip_live_frame_locks: page-locks on callers' memory that contexts of this process hold right now."""
    return int(capi.hip_lib().rt_hip_live_frame_locks())


def device_count() -> int:
    n = C.c_int()
    check(capi.hip_lib().rt_hip_device_count(C.byref(n)))
    return n.value


class HipRayTracer:
    """`device`: one GPU (rt_hip_create).  `devices`: a list of GPUs behind ONE render() call (rt_hip_create_multi):
    scene replicated, row stripes dealt round-robin, one RCCL gather to devices[0], one copy to the host.
    `peer_copy`: move the stripes with hipMemcpyPeerAsync instead of RCCL (allows a device to appear twice: tests).
    `direct_frame`: no gather — every member stores its pixels straight into the caller's page-locked back buffer."""

    def __init__(self, device: int = 0, devices: list[int] | None = None, peer_copy: bool = False, direct_frame: bool = False, rank: int | None = None, world: int | None = None, unique_id: bytes | None = None):
        """`rank`, `world`, `unique_id`: one rank of a renderer with one process per GPU (rt_hip_create_rank; collective).
        The id comes from `unique_id()` on one process and must reach every rank unchanged."""
        self._lib = capi.hip_lib()
        self._ctx = C.c_void_p()
        if rank is not None:
            assert world is not None and unique_id is not None and len(unique_id) == 128
            check(self._lib.rt_hip_create_rank(C.byref(self._ctx), device, rank, world, (C.c_char * 128).from_buffer_copy(unique_id)))
            self.device = device
            self.devices = [device]
            self.rank, self.world = rank, world
            return
        self.rank, self.world = 0, 1
        if devices is None:
            check(self._lib.rt_hip_create(C.byref(self._ctx), device))
            self.device = device
            self.devices = [device]
        else:
            ordinals = (C.c_int * len(devices))(*devices)
            check(self._lib.rt_hip_create_multi(C.byref(self._ctx), ordinals, len(devices), (capi.RT_HIP_MULTI_PEER_COPY if peer_copy else 0) | (capi.RT_HIP_MULTI_DIRECT_FRAME if direct_frame else 0)))
            self.device = devices[0]
            self.devices = list(devices)

    def join_ranks(self, rank: int, world: int, unique_id: bytes, timeout_ms: int = 0) -> None:
        """rt_hip_join_ranks: the collective half of rt_hip_create_rank, on a context made by plain ``HipRayTracer(device)``
        — after the launcher has made sure that EVERY rank holds one.  Raises RtHipError (RT_HIP_TIMEOUT after `timeout_ms`;
        the tracer stays a usable single-GPU tracer then)."""
        assert len(unique_id) == 128
        check(self._lib.rt_hip_join_ranks(self._ctx, rank, world, (C.c_char * 128).from_buffer_copy(unique_id), timeout_ms))
        self.rank, self.world = rank, world

    def join_frame_group(self, rank: int, world: int, name: str, timeout_ms: int = 0) -> None:
        """rt_hip_join_frame_group: one process per GPU without an exchange step.  On a context made by plain
        ``HipRayTracer(device)``, collectively; afterwards EVERY rank calls ``render(..., out=<its mapping of the one shared
        uint32[H, W] buffer>)`` and every call returns when the whole frame is in that buffer.  `name`: a fresh shm_open name
        ("/rt_hip_...") all ranks were handed.  Raises RtHipError (RT_HIP_TIMEOUT after `timeout_ms`)."""
        check(self._lib.rt_hip_join_frame_group(self._ctx, rank, world, name.encode(), timeout_ms))
        self.rank, self.world, self.shared_frame = rank, world, True

    def comm_info(self, member: int = 0) -> dict:
        """What RCCL reports about the communicator member `member` talks through (rt_hip_comm_info)."""
        ranks, rank, device, transport = C.c_int(), C.c_int(), C.c_int(), C.c_uint32()
        check(self._lib.rt_hip_comm_info(self._ctx, member, C.byref(ranks), C.byref(rank), C.byref(device), C.byref(transport)))
        return {"ranks": ranks.value, "rank": rank.value, "device": device.value, "transport": capi.TRANSPORT_NAMES.get(transport.value, str(transport.value))}

    def phases(self) -> dict:
        """Where the time of the most recent render() went (rt_hip_phases_fetch)."""
        phases = RtHipPhases()
        check(self._lib.rt_hip_phases_fetch(self._ctx, C.byref(phases)))
        return phases.as_dict()

    def close(self) -> None:
        ctx, self._ctx = getattr(self, "_ctx", None), None
        if ctx:
            self._lib.rt_hip_destroy(ctx)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- drop-in ------------------------------------------------------------------------------------------
    def render(self, scene: RtHipScene, width: int, height: int, seed: int = 1, flags: int = 0, want_rgb: bool = False, out: np.ndarray | None = None, stats: bool = True):
        """rt_hip_render: returns (rgba8 uint32[H, W], rgb float32[H, W, 3] or None, stats dict).

        `out`: a uint32[H, W] array to render into (like rt's persistent back buffer); a fresh one otherwise.
        `stats=False`: the call as the plug-in makes it (stats == NULL): nothing but the launch and the wait is enqueued;
        the returned dict is empty."""
        want_stats = stats
        stats = RtHipStats()
        stats_arg = C.byref(stats) if want_stats else None
        if self.rank != 0 and not getattr(self, "shared_frame", False):  # a rank whose rank 0 lives in another process renders and sends; it has no frame of its own
            rgb = np.empty((height, width, 3), dtype=np.float32) if want_rgb else None
            check(self._lib.rt_hip_render(self._ctx, C.byref(scene), None, width, height, seed, flags, rgb.ctypes.data if rgb is not None else None, stats_arg))
            return None, None, stats.as_dict() if want_stats else {}
        rgba = out if out is not None else np.empty((height, width), dtype=np.uint32)
        assert rgba.dtype == np.uint32 and rgba.shape == (height, width) and rgba.flags.c_contiguous
        rgb = np.empty((height, width, 3), dtype=np.float32) if want_rgb else None
        check(
            self._lib.rt_hip_render(
                self._ctx,
                C.byref(scene),
                rgba.ctypes.data,
                width,
                height,
                seed,
                flags,
                rgb.ctypes.data if rgb is not None else None,
                stats_arg,
            )
        )
        return rgba, rgb, stats.as_dict() if want_stats else {}

    def forget_frame(self) -> None:
        """Drop the page-lock on the back buffer last rendered into with RT_HIP_FLAG_PERSISTENT_FRAME."""
        self._lib.rt_hip_forget_frame(self._ctx)

    def preview(self, scene: RtHipScene, width: int, height: int, want_rgb: bool = False, out: np.ndarray | None = None):
        """The one-ray-per-pixel preview (reference src/renderers/rasterizer.cpp) through the same drop-in call."""
        return self.render(scene, width, height, seed=0, flags=capi.RT_HIP_FLAG_PREVIEW, want_rgb=want_rgb, out=out)

    # ---- resident path ------------------------------------------------------------------------------------
    def upload(self, scene: RtHipScene) -> None:
        check(self._lib.rt_hip_scene_upload(self._ctx, C.byref(scene)))

    def render_device(
        self,
        width: int,
        height: int,
        d_rgba8: int,
        seed: int = 1,
        flags: int = 0,
        partition: tuple | None = None,
        d_rgb_f32: int | None = None,
        stream: int | None = None,
    ) -> None:
        part = C.byref(RtHipPartition(*partition)) if partition is not None else None
        check(self._lib.rt_hip_render_device(self._ctx, width, height, seed, flags, part, d_rgba8, d_rgb_f32, stream))

    def assemble_device(self, width: int, height: int, world: int, stripe_rows: int, d_gathered: int, d_frame: int, stream: int | None = None) -> None:
        check(self._lib.rt_hip_assemble_device(self._ctx, width, height, world, stripe_rows, d_gathered, d_frame, stream))

    def stats(self) -> dict:
        stats = RtHipStats()
        check(self._lib.rt_hip_stats_fetch(self._ctx, C.byref(stats)))
        return stats.as_dict()

    def member_count(self) -> int:
        n = C.c_int()
        check(self._lib.rt_hip_member_count(self._ctx, C.byref(n)))
        return n.value

    def member_device(self, rank: int) -> int:
        d = C.c_int()
        check(self._lib.rt_hip_member_device(self._ctx, rank, C.byref(d)))
        return d.value

    def member_stats(self, rank: int) -> dict:
        """Counters of member `rank`'s share of the last render() on a multi-GPU tracer."""
        stats = RtHipStats()
        check(self._lib.rt_hip_member_stats(self._ctx, rank, C.byref(stats)))
        return stats.as_dict()

    # ---- known-answer entry points (librt_hip_kat.so: test-only, include/rt_hip_kat.h) ---------------------
    def kat_random(self, seed: int, pixel: int, sample: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float32)
        capi.check_kat(capi.kat_lib().rt_hip_kat_random(self._ctx, seed, pixel, sample, n, out.ctypes.data))
        return out

    def kat_closest_hit(self, origins: np.ndarray, directions: np.ndarray):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = len(o)
        dist = np.empty(n, dtype=np.float32)
        kind = np.empty(n, dtype=np.uint32)
        index = np.empty(n, dtype=np.uint32)
        normal = np.empty((n, 3), dtype=np.float32)
        capi.check_kat(capi.kat_lib().rt_hip_kat_closest_hit(self._ctx, n, o.ctypes.data, d.ctypes.data, dist.ctypes.data, kind.ctypes.data, index.ctypes.data, normal.ctypes.data))
        return dist, kind, index, normal

    def kat_sqrt_div(self, a: np.ndarray, b: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        s = np.empty_like(a)
        q = np.empty_like(a)
        capi.check_kat(capi.kat_lib().rt_hip_kat_sqrt_div(self._ctx, a.size, a.ctypes.data, b.ctypes.data, s.ctypes.data, q.ctypes.data))
        return s, q

    def kat_exhaustive_math(self):
        """All 2^32 float bit patterns through the kernels' sqrt / reciprocal / reciprocal-sqrt sequences vs the
        compiler's correctly rounded expansions: ([mismatch counts], [first mismatching input bits])."""
        counts = (C.c_uint64 * 3)()
        first = (C.c_uint32 * 3)()
        capi.check_kat(capi.kat_lib().rt_hip_kat_exhaustive_math(self._ctx, C.byref(counts), C.byref(first)))
        return list(counts), list(first)

import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the libraries once if a toolchain is here and they are missing (the GPU box gets them prebuilt)."""
    from rt_amd import capi

    if not capi.hip_library_path().exists() or not capi.host_library_path().exists() or not (ROOT / "oracle" / "liboracle.so").exists():
        import __graft_entry__

        __graft_entry__.build()


@pytest.fixture(autouse=True)
def _no_page_lock_outlives_a_gpu_test(request):
    """memory the module no longer knows about can never be written by it.  The module's default path keeps
    nothing of the caller's memory at all; the opt-in page-lock (RT_HIP_FLAG_PERSISTENT_FRAME, frame groups) must be gone —
    forgotten, replaced or destroyed with its context — by the time the test that took it is over, or the numpy arrays the
    next tests allocate could come to lie under a live lock."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import rt_amd

        assert rt_amd.live_frame_locks() == 0, "a page-lock on a test's buffer outlived the test (rt_hip_forget_frame missing?)"


@pytest.fixture(scope="session")
def tracer():
    """One rt_hip context on GPU 0 for the whole session.  No skip: without a gfx950 device this FAILS."""
    import rt_amd

    t = rt_amd.HipRayTracer(device=0)
    yield t
    t.close()


def unpack(rgba: np.ndarray) -> np.ndarray:
    """uint32 RGBA8888 -> uint8[..., 4]"""
    return np.stack([(rgba >> 24) & 255, (rgba >> 16) & 255, (rgba >> 8) & 255, rgba & 255], axis=-1).astype(np.uint8)


PLANES_SCENE = """
samples_per_pixel = 4
max_bounces = 6
camera = { position = [0.5, 1.5, 4], direction = [0, -0.2, -1] }
materials = [
    { type = 'lambert', albedo = [0.8, 0.8, 0.3] },
    { type = 'metal', albedo = [0.9, 0.9, 0.9], roughness = 0.2 },
    { type = 'lambert', albedo = 'portal_orange', reflectivity = 0.7 },
    { type = 'diamond', albedo = [0.2, 0.4, 0.9, 0.5] },
]
planes = [
    { material = 0 },
    { material = 1, position = [0, 0, -6], normal = [0, 0, 1] },
    { material = 3, position = [-4, 0, 0], normal = [2, 0, 0.5] },
]
spheres = [
    { material = 2, position = [0, 1, 0], radius = 1 },
    { material = 1, position = [2, 0.5, -1] },
    { material = 3, position = [-1.5, 0.4, 1], radius = 0.4 },
    { material = 0, position = [0.2, 0.3, 2], radius = 0.3 },
]
boxes = [ { material = 0 } ]
"""


# what the preview (RT_HIP_FLAG_PREVIEW) is shown: every primitive kind, boxes in front of and behind other things, one
# box that the camera looks along the face of, fractional albedos (so that shading is not saturated)
PREVIEW_SCENE = """
camera = { position = [0.5, 1.5, 6], direction = [0, -0.15, -1] }
materials = [
    { type = 'lambert', albedo = [0.8, 0.8, 0.3] },
    { type = 'metal', albedo = [0.9, 0.5, 0.2] },
    { type = 'lambert', albedo = [0.2, 0.4, 0.9] },
    { type = 'dielectric', albedo = [0.6, 0.9, 0.6] },
]
planes = [
    { material = 0 },
    { material = 2, position = [0, 0, -8], normal = [0, 0, 1] },
]
boxes = [
    { material = 1, position = [-2, 0.5, 0] },
    { material = 3, position = [2, 1, -2], extents = [0.5, 1, 0.75] },
    { material = 2, position = [0.5, 0.25, 3], extents = [0.25, 0.25, 0.25] },
    { material = 1, position = [0, 3, -7.5], extents = [3, 0.5, 0.5] },
]
spheres = [
    { material = 3, position = [0, 1, 0], radius = 1 },
    { material = 1, position = [2, 2.4, -2], radius = 0.4 },
    { material = 0, position = [-2, 1.3, 0], radius = 0.3 },
]
"""


@pytest.fixture(scope="session")
def planes_scene():
    import rt_amd

    return rt_amd.Scene.parse(PLANES_SCENE)

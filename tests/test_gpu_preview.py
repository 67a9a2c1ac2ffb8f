"""The preview on the GPU (RT_HIP_FLAG_PREVIEW; reference src/renderers/rasterizer.cpp:24-85) against the oracle:
bit-exact on the packed frame and on the float colour before packing (0 ulp, as for the traced frames)."""
import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from rt_amd import capi
from tests.conftest import GOLDEN, PREVIEW_SCENE

pytestmark = pytest.mark.gpu

PREVIEW = capi.RT_HIP_FLAG_PREVIEW


def same_floats(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def check(tracer, pod, width, height, what):
    got_rgba, got_rgb, stats = tracer.preview(pod, width, height, want_rgb=True)
    with np.errstate(all="ignore"):
        want_rgba, want_rgb, want_stats = oracle.render(pod, width, height, preview=True)
    same = same_floats(got_rgb, want_rgb)
    assert same.all(), f"{what}: colour differs in {(~same).any(axis=-1).sum()} pixels, first at {np.argwhere(~same)[0]}"
    assert np.array_equal(got_rgba, want_rgba), f"{what}: RGBA8 differs in {(got_rgba != want_rgba).sum()} pixels"
    assert stats["kernel"] == "preview"
    assert stats["primary_samples"] == stats["segments"] == width * height == want_stats["segments"]
    return got_rgba


@pytest.mark.parametrize(
    "name,width,height",
    [("preview", 320, 180), ("preview", 67, 33), ("preview", 1, 1), ("preview", 300, 1), ("preview", 1, 70), ("basic", 256, 256), ("dielectric", 192, 108), ("planes", 128, 72), ("synthetic-3000", 96, 54)],
)
def test_preview_is_bit_exact(tracer, planes_scene, name, width, height):
    scene = {"preview": lambda: rt_amd.Scene.parse(PREVIEW_SCENE), "planes": lambda: planes_scene}.get(name, lambda: rt_amd.Scene.named(name))()
    check(tracer, scene.describe(width, height), width, height, f"{name} {width}x{height}")


def test_preview_matches_committed_golden(tracer):
    golden = np.load(GOLDEN / "preview_96x54.npz")
    width, height = int(golden["width"]), int(golden["height"])
    rgba, rgb, _ = tracer.preview(rt_amd.Scene.parse(PREVIEW_SCENE).describe(width, height), width, height, want_rgb=True)
    assert np.array_equal(rgba, golden["rgba"])
    assert np.array_equal(rgb.view(np.uint32), golden["rgb"].view(np.uint32))


def random_preview_scene(rng):
    n_mat = int(rng.integers(1, 6))
    materials = [(int(rng.integers(0, 8)), *rng.uniform(0.0, 1.2, 3), 1.0, 0.5, 0.5) for _ in range(n_mat)]
    spheres = [(rng.uniform(-4, 4), rng.uniform(-1, 3), rng.uniform(-8, 0), rng.uniform(0.2, 1.5), rng.integers(0, n_mat)) for _ in range(int(rng.integers(0, 10)))]
    if spheres and rng.random() < 0.3:
        spheres[0] = (0.0, 1.0, 2.0, 30.0, spheres[0][4])  # around the camera
    boxes = [(rng.uniform(-4, 4), rng.uniform(-1, 3), rng.uniform(-8, 0), *rng.uniform(0.1, 1.5, 3), rng.integers(0, n_mat)) for _ in range(int(rng.integers(0, 10)))]
    if boxes and rng.random() < 0.3:
        boxes[0] = (0.0, 1.0, 2.0, 20.0, 20.0, 20.0, boxes[0][6])  # the camera inside a box
    planes = []
    for _ in range(int(rng.integers(0, 4))):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        planes.append((*n, rng.uniform(0.0, 3.0), rng.integers(0, n_mat)))
    camera = rt_amd.Scene.parse("").set_camera((rng.uniform(-1, 1), rng.uniform(0.5, 2), rng.uniform(1, 4)), (rng.uniform(-0.3, 0.3), rng.uniform(-0.4, 0.2), -1.0))
    return spheres, planes, boxes, materials, camera


@pytest.mark.parametrize("case", range(24))
def test_random_previews_are_bit_exact(tracer, case):
    rng = np.random.default_rng(5000 + case)
    spheres, planes, boxes, materials, camera = random_preview_scene(rng)
    width, height = int(rng.integers(17, 200)), int(rng.integers(9, 120))
    ivp = camera.describe(width, height).inverse_view_projection[:]
    pod = rt_amd.scene_from_arrays(spheres, planes, materials, boxes=boxes, inverse_view_projection=ivp)
    check(tracer, pod, width, height, f"case {case}")


def test_rays_along_box_faces_and_axis_parallel_rays_agree(tracer):
    """An orthographic view straight down -Z: every ray is parallel to four of the six slabs (1 / 0 = inf), and the
    box edges are placed exactly on pixel-centre rays (0 * inf = NaN inside the slab test).  Whatever the selections
    make of those, the device and the oracle make the same."""
    width = height = 8  # pixel centres at ndc -0.875 ... 0.875 in steps of 0.25
    ivp = np.diag([1.0, 1.0, -10.0, 1.0])
    boxes = [(0.0, 0.0, -5.0, 0.375, 0.625, 1.0, 0), (0.5, -0.5, -3.0, 0.125, 0.125, 0.5, 1)]
    materials = [(0, 0.9, 0.5, 0.1, 1, 0.5, 0.5), (0, 0.1, 0.5, 0.9, 1, 0.5, 0.5)]
    pod = rt_amd.scene_from_arrays(boxes=boxes, planes=[(0, 0, 1, 8, 1)], materials=materials, inverse_view_projection=ivp)
    rgba = check(tracer, pod, width, height, "faces")
    assert len(np.unique(rgba)) >= 2


def test_mg_frames_do_not_see_boxes(tracer):
    """mg_ray_tracer's test_boxes never hits (mg_ray_tracer.cpp:89-93): adding boxes changes no traced pixel."""
    scene = rt_amd.Scene.parse(PREVIEW_SCENE).set_sampling(4, 5)
    pod = scene.describe(96, 54)
    with_boxes, _, _ = tracer.render(pod, 96, 54, seed=3)
    pod.n_boxes = 0
    without, _, _ = tracer.render(pod, 96, 54, seed=3)
    assert np.array_equal(with_boxes, without)
    want, _, _ = oracle.render(pod, 96, 54, seed=3, want_rgb=False)
    assert np.array_equal(without, want)


@pytest.mark.parametrize("world,stripe", [(2, 8), (3, 5), (8, 8)])
def test_preview_partition_and_assemble(tracer, world, stripe):
    import torch

    width, height = 200, 117
    pod = rt_amd.Scene.parse(PREVIEW_SCENE).describe(width, height)
    whole, _, _ = tracer.preview(pod, width, height)
    tracer.upload(pod)
    padded = rt_amd.padded_local_rows(height, world, stripe)
    gathered = torch.zeros((world, padded, width), dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(world):
        tracer.render_device(width, height, gathered[rank].data_ptr(), flags=PREVIEW, partition=(rank, world, stripe), stream=stream)
    frame = torch.empty((height, width), dtype=torch.int32, device="cuda:0")
    tracer.assemble_device(width, height, world, stripe, gathered.data_ptr(), frame.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), whole)


def test_preview_at_1080p_property_checks(tracer):
    """Full-size frame: the oracle renders it whole too (one ray per pixel is cheap)."""
    scene = rt_amd.Scene.parse(PREVIEW_SCENE)
    rgba = check(tracer, scene.describe(1920, 1080), 1920, 1080, "1080p")
    assert ((rgba & 0xFF) == 0xFF).all()


def test_box_with_bad_material_or_missing_column_is_refused(tracer):
    pod = rt_amd.scene_from_arrays(boxes=[(0, 0, -3, 1, 1, 1, 4)], materials=[(0, 1, 1, 1, 1, 0.5, 0.5)])
    with pytest.raises(capi.RtHipError, match="box 0 has material index 4 out-of-range"):
        tracer.upload(pod)
    pod = rt_amd.scene_from_arrays(boxes=[(0, 0, -3, 1, 1, 1, 0)], materials=[(0, 1, 1, 1, 1, 0.5, 0.5)])
    pod.box_extents_y = None
    with pytest.raises(capi.RtHipError, match="a box column is NULL"):
        tracer.upload(pod)

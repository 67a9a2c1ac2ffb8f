"""The preview (RT_HIP_FLAG_PREVIEW / ORACLE_PREVIEW) on the CPU side: known answers worked out by hand from the
reference's rasterizer (src/renderers/rasterizer.cpp:24-85), the box test, and the committed golden frame.  The GPU
side is held to the same oracle in tests/test_gpu_parity.py."""
import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from tests.conftest import GOLDEN, PREVIEW_SCENE, unpack

WHITE = (1, 1, 1, 1, 1, 0.5, 0.5)


def ortho(spheres=None, planes=None, boxes=None, materials=(WHITE,)):
    """inverse view-projection diag(1, 1, 10, 1): pixel (x, y) of a W x H frame looks along +Z from (ndc_x, ndc_y, 0)
    towards (ndc_x, ndc_y, 10); hits count up to max_dist + 1 = 11 (rasterizer.cpp:33,35)"""
    return rt_amd.scene_from_arrays(spheres=spheres, planes=planes, boxes=boxes, materials=list(materials), inverse_view_projection=np.diag([1.0, 1.0, 10.0, 1.0]))


# ---- ray vs box -------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize(
    "origin,direction,center,extents,expected",
    [
        ((0, 0, 5), (0, 0, -1), (0, 0, 0), (1, 1, 1), 4.0),  # head on
        ((0, 0, 0.5), (0, 0, -1), (0, 0, 0), (1, 1, 1), 1.5),  # from inside: the exit distance
        ((0, 0, 5), (0, 0, 1), (0, 0, 0), (1, 1, 1), None),  # box behind the ray
        ((2, 0, 5), (0, 0, -1), (0, 0, 0), (1, 1, 1), None),  # passes beside it
        ((0.5, 0.5, 5), (0, 0, -1), (0, 0, 0), (1, 1, 1), 4.0),  # parallel to two slabs, inside both
        ((3, 0, 0), (-1, 0, 0), (0, 0, 0), (1, 2, 3), 2.0),
        ((0, 0, 0), (1, 1, 1), (4, 4, 4), (1, 1, 1), 3.0),  # un-normalised direction: t is in units of it
        ((-3, 1.5, 0), (1, 0, 0), (0, 0, 0), (1, 1, 1), None),  # above the box
    ],
)
def test_ray_box_known_answers(origin, direction, center, extents, expected):
    hit, t = oracle.hits_box(origin, direction, center, extents)
    assert hit == (expected is not None)
    if hit:
        assert t == pytest.approx(expected, rel=1e-6)


def test_nothing_beyond_the_far_point_plus_one_is_drawn():
    """dist starts at |far - near| + 1 (rasterizer.cpp:33,35) and only strictly nearer hits replace it."""
    red = [(0, 1, 0, 0, 1, 0.5, 0.5)]
    assert oracle.render(ortho(spheres=[(0, 0, 11.5, 1, 0)], materials=red), 3, 3, preview=True)[0][1, 1] == 0xFF3F3FFF  # t = 10.5
    assert oracle.render(ortho(spheres=[(0, 0, 12, 1, 0)], materials=red), 3, 3, preview=True)[0][1, 1] == 0xFFFFFFFF  # t = 11: not < 11


# ---- frames worked out by hand --------------------------------------------------------------------------------------
def test_sky_is_white_and_a_one_row_frame_is_black():
    """rasterizer.cpp:66-67 builds both sky colours from integers; colour's integer constructor clamps to [0, 1]
    (colour.hpp:72-91), so lerp(white, white, y / (H - 1)) = white — and NaN for H = 1 (0 / 0), which packs to 0."""
    rgba, rgb, stats = oracle.render(ortho(), 7, 5, preview=True)
    assert (rgba == 0xFFFFFFFF).all() and (rgb == 1.0).all()
    assert stats["primary_samples"] == stats["segments"] == 35
    with np.errstate(all="ignore"):
        rgba, rgb, _ = oracle.render(ortho(), 4, 1, preview=True)
    assert (rgba == 0x000000FF).all() and np.isnan(rgb).all()


def test_sphere_shading_by_hand():
    """5 x 5 ortho frame, unit sphere at (0, 0, 5), albedo (1, 0, 1): the centre ray hits at t = 4 with the normal
    facing it (N . L = 1): 0.25 + 0.75 * albedo = (1, 0.25, 1).  One pixel to the right (ndc x = 0.4) the normal is
    (0.4, 0, -sqrt(0.84)): N . L = 0.9165..."""
    pod = ortho(spheres=[(0, 0, 5, 1, 0)], materials=[(0, 1, 0, 1, 1, 0.5, 0.5)])
    rgba, rgb, _ = oracle.render(pod, 5, 5, preview=True)
    assert unpack(rgba)[2, 2].tolist() == [255, 63, 255, 255]
    assert rgb[2, 2].tolist() == [1.0, 0.25, 1.0]
    k = np.sqrt(0.84)
    assert rgb[2, 3] == pytest.approx([0.25 + 0.75 * k, 0.25, 0.25 + 0.75 * k], rel=1e-6)
    assert unpack(rgba)[2, 3].tolist() == [int((0.25 + 0.75 * k) * 255.99999), 63, int((0.25 + 0.75 * k) * 255.99999), 255]
    assert rgba[0, 0] == 0xFFFFFFFF  # the corner rays miss: sky


def test_candidates_replace_only_if_strictly_nearer_and_in_the_order_planes_boxes_spheres():
    """A plane, a box face and a sphere pole all at t = 4 on the centre ray: the plane is met first and keeps the
    pixel (`*hit >= dist` skips, rasterizer.cpp:48).  Without the plane the box keeps it; then the sphere."""
    red, green, blue = (0, 1, 0, 0, 1, 0.5, 0.5), (0, 0, 1, 0, 1, 0.5, 0.5), (0, 0, 0, 1, 1, 0.5, 0.5)
    plane = [(0, 0, -1, 4, 0)]  # -z + 4 = 0: the plane z = 4, facing the camera
    box = [(0, 0, 5, 1, 1, 1, 1)]
    sphere = [(0, 0, 5, 1, 2)]
    centre = lambda pod: unpack(oracle.render(pod, 3, 3, preview=True)[0])[1, 1, :3].tolist()
    assert centre(ortho(sphere, plane, box, (red, green, blue))) == [255, 63, 63]
    # the box is hit with the normal still `up`: N . L = (0, 1, 0) . (0, 0, -1) = 0 -> 0.25 everywhere
    assert centre(ortho(sphere, None, box, (red, green, blue))) == [63, 63, 63]
    assert centre(ortho(sphere, None, None, (red, green, blue))) == [63, 63, 255]


def test_a_box_hit_keeps_the_normal_of_the_plane_met_before_it():
    """rasterizer.cpp:56-59 sets no normal for a box: it inherits the last accepted plane's (or `up`)."""
    green = (0, 0, 1, 0, 1, 0.5, 0.5)
    far_plane = [(0, 0, -1, 9, 0)]  # z = 9, behind the box, normal (0, 0, -1) facing the camera
    box = [(0, 0, 5, 1, 1, 1, 0)]
    rgba, rgb, _ = oracle.render(ortho(None, far_plane, box, (green,)), 3, 3, preview=True)
    assert rgb[1, 1].tolist() == [0.25, 1.0, 0.25]  # shaded with the plane's normal: N . L = 1


def test_inside_a_sphere_the_far_wall_faces_away_and_goes_dark():
    """hits() returns the far root from inside; the normal points outward, N . L < 0, and the pack clamps at 0."""
    pod = ortho(spheres=[(0, 0, 0, 3, 0)])
    rgba, rgb, _ = oracle.render(pod, 3, 3, preview=True)
    assert rgb[1, 1].tolist() == [-0.5, -0.5, -0.5]
    assert rgba[1, 1] == 0x000000FF


def test_preview_ignores_seed_sampling_and_bounce_limit():
    scene = rt_amd.Scene.parse(PREVIEW_SCENE)
    a = oracle.render(scene.describe(64, 36), 64, 36, seed=1, preview=True)[0]
    scene.set_sampling(17, 3)
    b = oracle.render(scene.describe(64, 36), 64, 36, seed=99, preview=True)[0]
    assert np.array_equal(a, b)


def test_preview_of_a_partition_is_the_matching_rows_of_the_whole_frame():
    scene = rt_amd.Scene.parse(PREVIEW_SCENE)
    width, height = 50, 37
    pod = scene.describe(width, height)
    whole = oracle.render(pod, width, height, preview=True)[0]
    for world, stripe in [(2, 8), (3, 5)]:
        for rank in range(world):
            part = oracle.render(pod, width, height, preview=True, partition=(rank, world, stripe))[0]
            rows = [y for y in range(height) if (y // stripe) % world == rank]
            assert np.array_equal(part, whole[rows])


def test_oracle_reproduces_golden_preview():
    golden = np.load(GOLDEN / "preview_96x54.npz")
    width, height = int(golden["width"]), int(golden["height"])
    rgba, rgb, _ = oracle.render(rt_amd.Scene.parse(PREVIEW_SCENE).describe(width, height), width, height, preview=True)
    assert np.array_equal(rgba, golden["rgba"])
    assert np.array_equal(rgb.view(np.uint32), golden["rgb"].view(np.uint32))
    kinds = {tuple(c) for c in unpack(rgba).reshape(-1, 4)[:, :3]}
    assert len(kinds) > 100  # a shaded picture, not a flat one

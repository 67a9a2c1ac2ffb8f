"""Host side above the C ABI: scene files -> rt::scene -> rt_hip_scene (rt_amd/host, mirror of reference
src/scene.cpp:483-618 and src/camera.hpp)."""
import numpy as np
import pytest

import rt_amd
from rt_amd.scene import column


def test_basic_scene_matches_survey_data():
    pod = rt_amd.Scene.named("basic").describe(1920, 1080)
    assert (pod.n_spheres, pod.n_planes, pod.n_materials) == (3, 0, 3)
    assert (pod.samples_per_pixel, pod.max_bounces) == (30, 10)  # loader defaults, scene.cpp:531-532
    assert column(pod.sphere_center_x, 3).tolist() == [0, 0, 1]
    assert column(pod.sphere_center_y, 3).tolist() == [-1000, 0.5, 0.5]
    assert column(pod.sphere_radius, 3).tolist() == [1000, 0.5, 0.5]  # default radius 0.5, scene.cpp:592
    assert column(pod.sphere_material, 3, np.uint32).tolist() == [0, 1, 2]
    assert column(pod.material_type, 3, np.uint32).tolist() == [0, 0, 1]
    # named colours saturate: gray_33 -> white, fuchsia -> (1,0,1) (colour.hpp:72-98)
    assert column(pod.material_albedo, 12).reshape(3, 4).tolist() == [[1, 1, 1, 1], [1, 0, 1, 1], [1, 1, 1, 1]]
    assert np.allclose(column(pod.material_roughness, 3), [0.5, 0.5, 0.05])
    assert np.allclose(column(pod.material_reflectivity, 3), [0.5, 0.5, 0.8])  # per-type defaults, scene.cpp:547-556


def test_dielectric_scene_defaults():
    scene = rt_amd.Scene.named("dielectric")
    pod = scene.describe(64, 64)
    assert (pod.n_spheres, pod.n_materials, pod.samples_per_pixel) == (7, 7, 200)
    assert column(pod.material_type, 7, np.uint32).tolist() == [0, 4, 1, 2, 3, 5, 6]
    assert np.allclose(column(pod.material_reflectivity, 7), [0.5, 1.0, 0.8, 1.52, 1.000293, 1.333, 1.31])
    assert np.allclose(column(pod.material_roughness, 7), [0.5, 0.5, 0.05, 0, 0, 0, 0])


def test_loader_defaults_and_clamps():
    pod = rt_amd.Scene.parse("").describe(8, 8)
    # no materials -> one default lambert fuchsia, roughness 0.05, reflectivity 0.5 (scene.cpp:565-566)
    assert pod.n_materials == 1 and pod.n_spheres == 0 and pod.n_planes == 0
    assert column(pod.material_albedo, 4).tolist() == [1, 0, 1, 1]
    assert column(pod.material_roughness, 1)[0] == np.float32(0.05)
    pod = rt_amd.Scene.parse("samples_per_pixel = 5000\nmax_bounces = 0").describe(8, 8)
    assert (pod.samples_per_pixel, pod.max_bounces) == (1000, 1)  # clamp to [1, 1000]
    pod = rt_amd.Scene.parse("spheres = [ {} ]\nplanes = [ {} ]").describe(8, 8)
    assert [column(p, 1)[0] for p in (pod.sphere_center_x, pod.sphere_center_y, pod.sphere_center_z, pod.sphere_radius)] == [0, 1, -3, 0.5]
    assert [column(p, 1)[0] for p in (pod.plane_normal_x, pod.plane_normal_y, pod.plane_normal_z, pod.plane_d)] == [0, 1, 0, 0]


def test_plane_normal_is_normalised_and_d_is_minus_n_dot_p():
    pod = rt_amd.Scene.parse("planes = [ { position = [1, 2, 3], normal = [0, 3, 4] } ]").describe(8, 8)
    n = [column(p, 1)[0] for p in (pod.plane_normal_x, pod.plane_normal_y, pod.plane_normal_z)]
    assert np.allclose(n, [0, 0.6, 0.8])
    assert column(pod.plane_d, 1)[0] == pytest.approx(-(2 * 0.6 + 3 * 0.8))


def test_toml_forms_the_reader_accepts():
    text = '''
    # table headers instead of inline tables, dotted keys, numbers in several notations
    samples_per_pixel = 0x10
    [camera]
    position = [ 1, 2.5, -3e0 ]   # mixed int / float
    direction = "down"
    [[materials]]
    type = 1                       # enum by integer (scene.cpp:386-392)
    albedo = [ 0.1, 0.2 ]          # short arrays fill from the front; alpha defaults to 1
    [[materials]]
    type = "ice"
    name = 'cold "one"'
    [[spheres]]
    material = 1
    position = 2                   # a single number broadcasts (scene.cpp:146-157)
    radius = 1_0.5
    '''
    pod = rt_amd.Scene.parse(text).describe(8, 8)
    assert pod.samples_per_pixel == 16
    assert column(pod.material_type, 2, np.uint32).tolist() == [1, 6]
    assert np.allclose(column(pod.material_albedo, 8).reshape(2, 4)[0], [0.1, 0.2, 0, 1])
    assert [column(p, 1)[0] for p in (pod.sphere_center_x, pod.sphere_center_y, pod.sphere_center_z, pod.sphere_radius)] == [2, 2, 2, 10.5]


@pytest.mark.parametrize(
    "text,fragment",
    [
        ("spheres = [ { material = 3 } ]", "material index 3 out-of-range"),
        ("camera = { direction = 'sideways' }", "unknown vector alias 'sideways'"),
        ("materials = [ { albedo = 'octarine' } ]", "unknown colour alias 'octarine'"),
        ("materials = [ { type = 'plasma' } ]", "was not a member of enum"),
        ("materials = [ { type = 8 } ]", "was not a member of enum"),
        ("spheres = [ { radius = 'big' } ]", "No mapping from TOML string to float"),
        ("spheres = [ { position = [1, 2, 3, 4] } ]", "No mapping from TOML array[4]"),
        ("spheres = [ { radius = inf } ]", "Infinities and NaNs are not allowed"),
        ("spheres = 3", "expected array at key 'spheres'"),
        ("camera = [1]", "expected table at key 'camera'"),
        ("spheres = [ { material = 0 ", "line"),
        ("a = 1\na = 2", "duplicate key"),
    ],
)
def test_loader_errors(text, fragment):
    with pytest.raises(rt_amd.SceneError) as err:
        rt_amd.Scene.parse(text)
    assert fragment in str(err.value)


def test_missing_file_and_empty_path():
    with pytest.raises(rt_amd.SceneError, match="did not exist or was not a file"):
        rt_amd.Scene.load("/nonexistent/scene.toml")
    with pytest.raises(rt_amd.SceneError, match="no scene file path provided"):
        rt_amd.Scene.load("")


def test_relative_paths_are_searched_under_scenes(monkeypatch):
    from tests.conftest import ROOT

    monkeypatch.chdir(ROOT)
    assert rt_amd.Scene.load("basic.toml").describe(8, 8).n_spheres == 3  # found as scenes/basic.toml (scene.cpp:479-480)


def test_synthetic_scene_is_the_pinned_generator():
    pod = rt_amd.Scene.synthetic(100000).describe(1920, 1080)
    assert (pod.n_spheres, pod.n_materials, pod.samples_per_pixel, pod.max_bounces) == (100000, 8, 64, 10)
    cx, cy, cz, r = (column(p, 100000) for p in (pod.sphere_center_x, pod.sphere_center_y, pod.sphere_center_z, pod.sphere_radius))
    mat = column(pod.sphere_material, 100000, np.uint32)
    assert (cx[0], cy[0], cz[0], r[0], mat[0]) == (0, -1000, 0, 1000, 0)
    assert np.array_equal(cy[1:], r[1:]) and r[1:].min() >= 0.05 and r[1:].max() < 0.25
    assert cx[1:].min() >= -40 and cx[1:].max() < 40 and cz[1:].min() > -80 and cz[1:].max() <= -2
    assert np.array_equal(mat[1:], 1 + (np.arange(1, 100000) % 7))
    # splitmix64(20250310): first small sphere, computed independently in Python integers
    state = 20250310
    def nxt():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        z ^= z >> 31
        return np.float32(z >> 40) * np.float32(2.0**-24)
    u1, u2, u3 = nxt(), nxt(), nxt()
    assert cx[1] == np.float32(-40) + np.float32(80) * u1
    assert cz[1] == np.float32(-2) - np.float32(78) * u2
    assert r[1] == np.float32(0.05) + np.float32(0.20) * u3
    types = column(pod.material_type, 8, np.uint32)
    assert types.tolist() == [0, 0, 1, 0, 1, 0, 1, 0]


def test_viewport_inverse_view_projection_round_trips():
    scene = rt_amd.Scene.named("basic")
    # depth 0 and 1 un-project onto the same eye ray; depth 0 sits on the near plane (0.01 in front of the eye)
    near = scene.screen_to_world(640, 480, 320, 240, 0.0)
    far = scene.screen_to_world(640, 480, 320, 240, 1.0)
    assert np.allclose(near, [0, 1, 3 - 0.01], atol=1e-4)
    assert far[2] < -900 and abs(far[0]) < 1 and abs(far[1] - 1) < 15
    # a corner: x offset = near * tan(pi/8) * aspect
    corner = scene.screen_to_world(640, 480, 640, 0, 0.0)
    assert np.allclose(corner, [0.01 * np.tan(np.pi / 8) * 640 / 480, 1 + 0.01 * np.tan(np.pi / 8), 2.99], atol=1e-4)


def test_boxes_are_loaded_into_their_columns_with_the_reference_defaults():
    """reference src/scene.cpp:599-615: position defaults to (0, 1, -3), extents to 0.5 (half sizes), material
    index checked like every other primitive's."""
    scene = rt_amd.Scene.parse(
        """
materials = [ {}, { type = 'metal' } ]
boxes = [ {}, { material = 1, position = [1, 2, 3], extents = [0.25, 0.5, 4] }, { extents = 2 } ]
"""
    )
    pod = scene.describe(8, 8)
    assert pod.n_boxes == 3 and pod.n_spheres == 0 and pod.n_planes == 0
    assert column(pod.box_center_x, 3).tolist() == [0, 1, 0]
    assert column(pod.box_center_y, 3).tolist() == [1, 2, 1]
    assert column(pod.box_center_z, 3).tolist() == [-3, 3, -3]
    assert column(pod.box_extents_x, 3).tolist() == [0.5, 0.25, 2]
    assert column(pod.box_extents_y, 3).tolist() == [0.5, 0.5, 2]
    assert column(pod.box_extents_z, 3).tolist() == [0.5, 4, 2]
    assert column(pod.box_material, 3, np.uint32).tolist() == [0, 1, 0]
    with pytest.raises(rt_amd.SceneError, match="material index 5 out-of-range"):
        rt_amd.Scene.parse("boxes = [ { material = 5 } ]")

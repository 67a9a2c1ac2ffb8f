"""The numbers the reference's own source text holds for this path, read WHERE THEY LIE under /root/reference (build container
only — nothing of the reference is copied or travels; the GPU box skips this file) and compared with what this
repository's host side and oracle actually do.

The reference has no tests, golden images or known-answer vectors, so the oracle's parity stays "unpinned"
by reference RESULTS.  What the reference does hold is DATA: the constants of `mg_ray_tracer.cpp`, the loader's defaults and
clamps, the material enum's order, the camera's defaults, the pack multiplier, the two benchmark scene files.  Each of
them is extracted from the source with a regular expression below; none is hard-coded on this side of the comparison.
"""
import re
from pathlib import Path

import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from rt_amd.scene import column

REF = Path("/root/reference")
pytestmark = pytest.mark.skipif(not (REF / "src" / "scene.cpp").exists(), reason="needs the reference tree (build container only)")


def text(relative):
    return (REF / relative).read_text()


def number(pattern, source, group=1):
    m = re.search(pattern, source)
    assert m, f"pattern not found in the reference: {pattern}"
    return float(m.group(group).rstrip("fFuU"))


def test_material_type_order_is_the_references_enum():
    body = re.search(r"enum class material_type\s*:\s*unsigned\s*\{([^}]*)\}", text("src/common.hpp")).group(1)
    names = [n.strip() for n in body.split(",") if n.strip()]
    assert len(names) == 8
    for index, name in enumerate(names):
        pod = rt_amd.Scene.parse(f"materials = [ {{ type = '{name}' }} ]").describe(8, 8)
        assert column(pod.material_type, 1, np.uint32)[0] == index, name
    # ... and the C ABI's constants carry the same numbers
    header = (Path(rt_amd.__file__).resolve().parent.parent / "include" / "rt_hip.h").read_text()
    for index, name in enumerate(names):
        assert re.search(rf"RT_HIP_MATERIAL_{name.upper()}\s*=\s*{index}\b", header), name


def test_loader_defaults_and_clamps_are_the_references():
    src = text("src/scene.cpp")
    spp_default = number(r'"samples_per_pixel",\s*(\d+)u\)', src)
    bounces_default = number(r'"max_bounces",\s*(\d+)u\)', src)
    lo, hi = number(r'"samples_per_pixel",\s*\d+u\),\s*(\d+)u,\s*(\d+)u\)', src, 1), number(r'"samples_per_pixel",\s*\d+u\),\s*(\d+)u,\s*(\d+)u\)', src, 2)
    pod = rt_amd.Scene.parse("").describe(8, 8)
    assert (pod.samples_per_pixel, pod.max_bounces) == (spp_default, bounces_default)
    pod = rt_amd.Scene.parse(f"samples_per_pixel = {int(hi) + 5}\nmax_bounces = 0").describe(8, 8)
    assert (pod.samples_per_pixel, pod.max_bounces) == (hi, lo)

    # per-type default reflectivity: the switch of scene.cpp:544-556
    table = dict(re.findall(r"case material_type::(\w+):\s*reflectiveness\s*=\s*([0-9.]+)f", src))
    fallback = number(r"default:\s*reflectiveness\s*=\s*([0-9.]+)f", src)
    assert set(table) == {"metal", "dielectric", "air", "vacuum", "ice", "water"}
    for name in ("lambert", "metal", "dielectric", "air", "vacuum", "water", "ice", "diamond"):
        pod = rt_amd.Scene.parse(f"materials = [ {{ type = '{name}' }} ]").describe(8, 8)
        want = np.float32(table.get(name, fallback))
        assert column(pod.material_reflectivity, 1)[0] == want, name
    # default roughness: 0 for dielectric, else the other constant
    m = re.search(r'"roughness",\s*type == material_type::dielectric \?\s*([0-9.]+)f\s*:\s*([0-9.]+)f', src)
    for name, want in (("dielectric", m.group(1)), ("lambert", m.group(2)), ("metal", m.group(2))):
        pod = rt_amd.Scene.parse(f"materials = [ {{ type = '{name}' }} ]").describe(8, 8)
        assert column(pod.material_roughness, 1)[0] == np.float32(want)
    # the material an empty scene gets
    m = re.search(r"materials\.push_back\(\"\"s,\s*material_type::(\w+),\s*colours::(\w+),\s*([0-9.]+)f,\s*([0-9.]+)f\)", src)
    pod = rt_amd.Scene.parse("").describe(8, 8)
    assert m.group(1) == "lambert" and column(pod.material_type, 1, np.uint32)[0] == 0
    assert column(pod.material_roughness, 1)[0] == np.float32(m.group(3)) and column(pod.material_reflectivity, 1)[0] == np.float32(m.group(4))

    # default primitives
    m = re.search(r'rt::sphere\{\s*deserialize\(tbl,\s*"position",\s*vec3\{\s*([-0-9.]+),\s*([-0-9.]+),\s*([-0-9.]+)\s*\}\),\s*//\s*deserialize\(tbl,\s*"radius",\s*([0-9.]+)f\)', src)
    pod = rt_amd.Scene.parse("spheres = [ {} ]").describe(8, 8)
    got = [column(p, 1)[0] for p in (pod.sphere_center_x, pod.sphere_center_y, pod.sphere_center_z, pod.sphere_radius)]
    assert got == [np.float32(g) for g in m.groups()]
    m = re.search(r'rt::box\{\s*deserialize\(tbl,\s*"position",\s*vec3\{\s*([-0-9.]+),\s*([-0-9.]+),\s*([-0-9.]+)\s*\}\),\s*//\s*deserialize\(tbl,\s*"extents",\s*vec3\{\s*([0-9.]+)f\s*\}\)', src)
    pod = rt_amd.Scene.parse("boxes = [ {} ]").describe(8, 8)
    got = [column(p, 1)[0] for p in (pod.box_center_x, pod.box_center_y, pod.box_center_z, pod.box_extents_x, pod.box_extents_y, pod.box_extents_z)]
    assert got == [np.float32(m.group(1)), np.float32(m.group(2)), np.float32(m.group(3))] + [np.float32(m.group(4))] * 3
    m = re.search(r'"normal",\s*vec3\{\s*([-0-9.]+),\s*([-0-9.]+),\s*([-0-9.]+)\s*\}', src)
    pod = rt_amd.Scene.parse("planes = [ {} ]").describe(8, 8)
    assert [column(p, 1)[0] for p in (pod.plane_normal_x, pod.plane_normal_y, pod.plane_normal_z)] == [np.float32(g) for g in m.groups()]


def test_camera_defaults_are_the_references():
    cam = text("src/camera.hpp")
    near, far = number(r"float near_\s*=\s*([0-9.]+)f", cam), number(r"float far_\s*=\s*([0-9.]+)f", cam)
    assert re.search(r"float vfov_\s*=\s*floats::pi_over_four", cam)
    pos = [float(g) for g in re.search(r"vec3 pos_\s*=\s*\{\s*([-0-9.]+),\s*([-0-9.]+),\s*([-0-9.]+)\s*\}", cam).groups()]
    scene_default = [float(g) for g in re.search(r'"position",\s*vec3\{\s*([-0-9.]+),\s*([-0-9.]+),\s*([-0-9.]+)\s*\}\),\s*deserialize\(\*camera', text("src/scene.cpp")).groups()]
    assert pos == scene_default
    scene = rt_amd.Scene.parse("")
    width, height = 200, 100
    near_point = scene.screen_to_world(width, height, width / 2, height / 2, 0.0)
    far_point = scene.screen_to_world(width, height, width / 2, height / 2, 1.0)
    # the centre ray starts `near` in front of the default eye and ends `far` in front of it, looking down -Z
    assert np.allclose(near_point, [pos[0], pos[1], pos[2] - near], atol=1e-5)
    assert np.allclose(far_point, [pos[0], pos[1], pos[2] - far], rtol=2e-2)  # (binary32 inverse of a near-0.01 / far-1000 projection: ~1 % at depth 1)
    # vertical field of view pi/4: the top edge's ray makes pi/8 with the axis
    top = scene.screen_to_world(width, height, width / 2, 0.0, 1.0) - np.array(pos, dtype=np.float32)
    assert np.arctan2(top[1], -top[2]) == pytest.approx(np.pi / 8, rel=1e-3)


def test_path_constants_of_mg_ray_tracer_are_the_oracles():
    mg = text("src/renderers/mg_ray_tracer.cpp")
    min_hit = number(r"min_hit_dist\s*=\s*([0-9.]+)f", mg)
    sky = [float(g) for g in re.search(r"vec3::lerp\(colours::white\.rgb,\s*vec3\{\s*([0-9.]+)f,\s*([0-9.]+)f,\s*([0-9.]+)f\s*\},\s*0\.5f \* \(r\.direction\.y \+ 1\.0f\)\)", mg).groups()]
    # sky: lerp(white, sky, 0.5 (y + 1)) at y = +1 is the sky colour itself, at y = -1 white
    assert np.allclose(oracle.sky(1.0), sky, atol=1e-7) and np.allclose(oracle.sky(-1.0), [1, 1, 1])
    # min_hit_dist: a sphere whose near surface is closer than that to the ray's origin is skipped, one beyond it is hit
    mat = [(0, 1, 1, 1, 1, 0.5, 0.5)]
    for gap, expect_hit in ((min_hit * 0.5, False), (min_hit * 2.0, True)):
        pod = rt_amd.scene_from_arrays(spheres=[(0, 0, -(1 + gap), 1, 0)], materials=mat)
        dist, kind, _, _ = oracle.closest_hit(pod, np.array([[0, 0, 0]], np.float32), np.array([[0, 0, -1]], np.float32))
        assert bool(kind[0]) == expect_hit, gap
        if expect_hit:
            assert dist[0] == pytest.approx(gap, rel=1e-3)
    # the pack multiplier of rt::colour -> uint32
    scale = number(r"vec4\{\s*(255\.[0-9]+)f\s*\}", text("src/colour.hpp"))
    for v in (0.25, 0.5, 0.999, 1.0):
        assert (oracle.pack(v, v, v) >> 24) == int(np.float32(v) * np.float32(scale))
    # reflect(v, n) = v - 2 dot(v, n) n
    assert re.search(r"return v - 2 \* vec3::dot\(v, n\) \* n;", text("src/common.hpp"))


def test_benchmark_scene_files_say_what_the_references_say():
    tomli = pytest.importorskip("tomli")
    here = Path(rt_amd.__file__).resolve().parent.parent / "scenes"
    for name in ("basic.toml", "dielectric.toml"):
        theirs = tomli.loads(text(f"scenes/{name}"))
        ours = tomli.loads((here / name).read_text())
        for doc in (theirs, ours):  # an empty list says the same as no list
            for key in [k for k, v in doc.items() if v == []]:
                del doc[key]
        assert ours == theirs, name

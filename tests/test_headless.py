"""rt_headless (rt_amd/host/main.cpp): the windowless driver around the renderer registry and the hip_ray_tracer
plug-in — the C++ host side a user of the reference would run."""
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT

BIN = ROOT / "rt_amd" / "bin" / "rt_headless"


def run(*args, **kw):
    return subprocess.run([str(BIN), *args], cwd=ROOT, capture_output=True, text=True, timeout=300, **kw)


def test_list_shows_the_registered_renderers():
    out = run("--list")
    assert out.returncode == 0
    assert "hip_ray_tracer" in out.stdout and "hip_sm_ray_tracer" in out.stdout and "null_renderer" in out.stdout
    assert "hip_rasterizer" in out.stdout
    # `--renderer hip` keeps meaning the path tracer: prefix matching takes the first registered match (src/main.cpp:75-78)
    assert out.stdout.index("hip_ray_tracer") < out.stdout.index("hip_rasterizer")


def test_unknown_renderer_and_missing_scene_fail_like_the_reference():
    out = run("--renderer", "vulkan")
    assert out.returncode == 1 and "error: no known renderer with name 'vulkan'" in out.stderr
    out = run("--renderer", "null", "--scene", "nope.toml")
    assert out.returncode == 1 and "did not exist or was not a file" in out.stderr


def test_prefix_match_and_null_renderer_leave_the_cleared_frame(tmp_path):
    ppm = tmp_path / "black.ppm"
    out = run("--renderer", "null", "--scene", "basic.toml", "--size", "16x8", "--out", str(ppm))
    assert out.returncode == 0 and "created renderer: null_renderer" in out.stdout
    data = ppm.read_bytes()
    assert data.startswith(b"P6\n16 8\n255\n") and set(data[len(b"P6\n16 8\n255\n") :]) == {0}


@pytest.mark.gpu
def test_hip_ray_tracer_plugin_renders_the_oracle_frame(tmp_path):
    import rt_amd
    from oracle import binding as oracle
    from tests.conftest import unpack

    ppm = tmp_path / "frame.ppm"
    out = run("--renderer", "hip", "--scene", "basic.toml", "--size", "96x54", "--spp", "5", "--seed", "11", "--out", str(ppm))
    assert out.returncode == 0, out.stderr
    assert "error:" not in out.stderr
    header = b"P6\n96 54\n255\n"
    data = ppm.read_bytes()
    got = np.frombuffer(data[len(header) :], dtype=np.uint8).reshape(54, 96, 3)
    scene = rt_amd.Scene.named("basic").set_sampling(5)
    want, _, _ = oracle.render(scene.describe(96, 54), 96, 54, seed=11, want_rgb=False)
    assert np.array_equal(got, unpack(want)[..., :3])


@pytest.mark.gpu
def test_hip_sm_ray_tracer_plugin_renders_the_oracle_frame_in_sm_mode(tmp_path):
    import rt_amd
    from oracle import binding as oracle
    from tests.conftest import unpack

    ppm = tmp_path / "frame.ppm"
    out = run("--renderer", "hip_sm", "--scene", "dielectric.toml", "--size", "80x45", "--spp", "6", "--seed", "5", "--out", str(ppm))
    assert out.returncode == 0 and "created renderer: hip_sm_ray_tracer" in out.stdout and "error:" not in out.stderr
    header = b"P6\n80 45\n255\n"
    got = np.frombuffer(ppm.read_bytes()[len(header) :], dtype=np.uint8).reshape(45, 80, 3)
    scene = rt_amd.Scene.named("dielectric").set_sampling(6)
    want, _, _ = oracle.render(scene.describe(80, 45), 80, 45, seed=5, want_rgb=False, sm_materials=True)
    assert np.array_equal(got, unpack(want)[..., :3])


@pytest.mark.gpu
def test_hip_rasterizer_plugin_draws_the_oracle_preview_boxes_included(tmp_path):
    import rt_amd
    from oracle import binding as oracle
    from tests.conftest import PREVIEW_SCENE, unpack

    scene_file = tmp_path / "preview.toml"
    scene_file.write_text(PREVIEW_SCENE)
    ppm = tmp_path / "preview.ppm"
    out = run("--renderer", "hip_ras", "--scene", str(scene_file), "--size", "120x67", "--out", str(ppm))
    assert out.returncode == 0 and "created renderer: hip_rasterizer" in out.stdout and "error:" not in out.stderr
    header = b"P6\n120 67\n255\n"
    got = np.frombuffer(ppm.read_bytes()[len(header) :], dtype=np.uint8).reshape(67, 120, 3)
    want, _, _ = oracle.render(rt_amd.Scene.parse(PREVIEW_SCENE).describe(120, 67), 120, 67, want_rgb=False, preview=True)
    assert np.array_equal(got, unpack(want)[..., :3])


@pytest.mark.gpu
@pytest.mark.parametrize("devices,frame", [("0", None), ("all", None), ("all", "direct"), ("0", "locked"), ("all", "locked"), ("all", "locked+direct")])
def test_plugin_with_rt_hip_devices_renders_through_the_multi_gpu_context(tmp_path, devices, frame):
    """RT_HIP_DEVICES makes the plug-in create ONE rt_hip_create_multi context (RCCL communicator, gather, assemble, one
    copy) behind the same blocking render() — on this box 'all' is one GPU, which still takes that whole path."""
    import os

    import rt_amd
    from oracle import binding as oracle
    from tests.conftest import unpack

    ppm = tmp_path / "frame.ppm"
    out = run("--renderer", "hip", "--scene", "basic.toml", "--size", "100x61", "--spp", "4", "--seed", "3", "--frames", "2", "--out", str(ppm), env=dict(os.environ, RT_HIP_DEVICES=devices, **({"RT_HIP_FRAME": frame} if frame else {})))
    assert out.returncode == 0, out.stderr
    assert "error:" not in out.stderr
    header = b"P6\n100 61\n255\n"
    got = np.frombuffer(ppm.read_bytes()[len(header) :], dtype=np.uint8).reshape(61, 100, 3)
    scene = rt_amd.Scene.named("basic").set_sampling(4)
    want, _, _ = oracle.render(scene.describe(100, 61), 100, 61, seed=3, want_rgb=False)
    assert np.array_equal(got, unpack(want)[..., :3])


@pytest.mark.gpu
def test_plugin_reports_a_bad_device_list_in_the_reference_error_style(tmp_path):
    import os

    out = run("--renderer", "hip", "--scene", "basic.toml", "--size", "16x8", "--out", str(tmp_path / "x.ppm"), env=dict(os.environ, RT_HIP_DEVICES="0,0"))
    assert "error: hip_ray_tracer:" in out.stderr and "named twice" in out.stderr


def test_device_list_errors_are_reported_without_a_gpu(tmp_path):
    """RT_HIP_DEVICES is parsed by the plug-in and checked by rt_hip_create_multi before any device is touched, so the
    messages can be checked on a box without a GPU as well; the frame stays the caller's pre-cleared black."""
    import os

    for devices, message in [("0,0", "named twice"), ("0,1,x", None), ("", None)]:
        ppm = tmp_path / "x.ppm"
        out = run("--renderer", "hip", "--scene", "basic.toml", "--size", "16x8", "--out", str(ppm), env=dict(os.environ, RT_HIP_DEVICES=devices))
        if message:
            assert "error: hip_ray_tracer:" in out.stderr and message in out.stderr, out.stderr
        if "error: hip_ray_tracer:" in out.stderr:
            data = ppm.read_bytes()
            assert set(data[len(b"P6\n16 8\n255\n") :]) == {0}


@pytest.mark.gpu
def test_three_rt_headless_processes_render_one_frame_into_a_shared_back_buffer(tmp_path):
    """`--shared-frame NAME --rank R --world N`: three driver processes, each with the plug-in as one rank of a frame group
    (RT_HIP_GROUP -> rt_hip_join_frame_group), all on the box's one GPU; the back buffer is a POSIX shared-memory object all
    three map.  Rank 0's PPM is the oracle's frame; two frames are rendered (rank 0 clears the buffer in between)."""
    import os
    import uuid

    import rt_amd
    from oracle import binding as oracle
    from tests.conftest import unpack

    name = f"rt_hip_headless_{uuid.uuid4().hex[:10]}"
    ppm = tmp_path / "frame.ppm"
    common = ["--renderer", "hip", "--scene", "basic.toml", "--size", "100x61", "--spp", "4", "--seed", "3", "--frames", "2", "--shared-frame", name, "--world", "3"]
    procs = [subprocess.Popen([str(BIN), *common, "--rank", str(r), "--out", str(ppm)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(3)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0 and "error:" not in err, err
    assert "back buffer is the shared mapping" in outs[1][0]
    header = b"P6\n100 61\n255\n"
    got = np.frombuffer(ppm.read_bytes()[len(header) :], dtype=np.uint8).reshape(61, 100, 3)
    scene = rt_amd.Scene.named("basic").set_sampling(4)
    want, _, _ = oracle.render(scene.describe(100, 61), 100, 61, seed=3, want_rgb=False)
    assert np.array_equal(got, unpack(want)[..., :3])
    assert not [f for f in os.listdir("/dev/shm") if name in f]


def test_a_malformed_group_is_reported_in_the_reference_error_style(tmp_path):
    import os

    out = run("--renderer", "hip", "--scene", "basic.toml", "--size", "16x8", "--out", str(tmp_path / "x.ppm"), env=dict(os.environ, RT_HIP_GROUP="nonsense"))
    assert out.returncode == 0 and "error: hip_ray_tracer: RT_HIP_GROUP must look like /name:rank:world" in out.stderr
    out = run("--renderer", "null", "--scene", "basic.toml", "--size", "16x8", "--shared-frame", "a/b", "--rank", "0", "--world", "1")
    assert out.returncode == 2 and "--shared-frame NAME needs" in out.stderr

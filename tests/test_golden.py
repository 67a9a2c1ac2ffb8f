"""The oracle against the committed golden fixtures (tests/golden/, made by tools/gen_golden.py): freezes
arithmetic contract v1 and the counter random streams on the CPU side.  The GPU side is held to the same files in
tests/test_gpu_parity.py."""
import json

import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from tests.conftest import GOLDEN, PLANES_SCENE


@pytest.mark.parametrize("fixture", ["basic_64x36_spp4", "dielectric_64x36_spp4", "planes_48x27_spp4", "synthetic1500_32x18_spp2"])
def test_oracle_reproduces_golden_frames(fixture):
    golden = np.load(GOLDEN / f"{fixture}.npz")
    name = str(golden["scene"])
    scene = rt_amd.Scene.parse(PLANES_SCENE) if name == "planes" else rt_amd.Scene.named(name)
    scene.set_sampling(int(golden["spp"]), int(golden["max_bounces"]))
    width, height = int(golden["width"]), int(golden["height"])
    rgba, rgb, stats = oracle.render(scene.describe(width, height), width, height, seed=int(golden["seed"]))
    assert np.array_equal(rgba, golden["rgba"])
    assert np.array_equal(rgb.view(np.uint32), golden["rgb"].view(np.uint32))
    assert stats["segments"] == int(golden["segments"])


def test_oracle_reproduces_golden_random_stream():
    golden = np.load(GOLDEN / "random_stream.npz")
    draws = oracle.random(int(golden["seed"]), int(golden["pixel"]), int(golden["sample"]), len(golden["draws"]))
    assert np.array_equal(draws.view(np.uint32), golden["draws"].view(np.uint32))


def test_oracle_reproduces_golden_closest_hits():
    golden = np.load(GOLDEN / "closest_hit_planes.npz")
    pod = rt_amd.Scene.parse(PLANES_SCENE).describe(48, 27)
    dist, kind, index, normal = oracle.closest_hit(pod, golden["origins"], golden["directions"])
    assert np.array_equal(dist.view(np.uint32), golden["distance"].view(np.uint32))
    assert np.array_equal(kind, golden["kind"]) and np.array_equal(index, golden["index"])
    assert np.array_equal(normal.view(np.uint32), golden["normal"].view(np.uint32))


def test_named_colours_saturate_like_the_reference():
    """tests/golden/named_colours.json was extracted from the reference's colour table (tools/gen_named_colours.py)
    with its quirk applied: a channel byte is cast to float and clamped to [0, 1], never divided by 255
    (reference src/colour.hpp:72-98), so every named colour channel is 0.0 or 1.0."""
    import ctypes as C

    from rt_amd import capi

    table = json.load(open(GOLDEN / "named_colours.json"))
    assert len(table) == 149
    assert table["gray_33"] == [1.0, 1.0, 1.0] and table["fuchsia"] == [1.0, 0.0, 1.0]
    assert table["aquamarine"] == [1.0, 1.0, 1.0] and table["white"] == [1.0, 1.0, 1.0]
    assert table["black"] == [0.0, 0.0, 0.0] and table["portal_blue"] == [0.0, 1.0, 1.0]
    out = (C.c_float * 4)()
    for name, rgb in table.items():
        assert capi.host_lib().rt_host_named_colour(name.encode(), out) == 1
        assert list(out[:3]) == rgb and out[3] == 1.0
    assert capi.host_lib().rt_host_named_colour(b"not_a_colour", out) == 0


def test_frame_digests_are_this_oracles():
    """tests/golden/frame_digests.json (tools/gen_frame_digests.py) holds sha256 digests of the oracle's frames at BASELINE.json's
    full sizes — what bench.py's `frame_matches_oracle` and the GPU suite's whole-frame checks compare against.  The small
    entries are re-made here (config 1: plumbing size; config 2: 1920x1080x64, seconds on a few cores); every entry names the
    contract it was made under and the workload string bench.py looks it up by."""
    import hashlib
    import json

    import rt_amd
    from tools.gen_frame_digests import TILT, WORKLOADS, workload_key

    digests = json.loads((GOLDEN / "frame_digests.json").read_text())
    assert set(digests) == set(WORKLOADS)
    for key, (name, width, height, spp, tilt, partition) in WORKLOADS.items():
        entry = digests[key]
        assert entry["contract"] == "v4" and entry["workload"] == workload_key(name, width, height, spp, entry["max_bounces"], 1, tilt)
        assert (entry["scene"], entry["width"], entry["height"], entry["spp"], entry["tilt"]) == (name, width, height, spp, tilt)
        assert entry["partition"] == (list(partition) if partition else None) and len(entry["sha256"]) == 64
    for key in ("config1", "config2"):
        name, width, height, spp, tilt, partition = WORKLOADS[key]
        scene = rt_amd.Scene.named(name).set_sampling(spp)
        if tilt:
            scene.set_camera(*TILT)
        rgba, _, stats = oracle.render(scene.describe(width, height), width, height, seed=1, partition=partition, want_rgb=False)
        assert hashlib.sha256(rgba.tobytes()).hexdigest() == digests[key]["sha256"], key
        assert stats["segments"] == digests[key]["segments"]

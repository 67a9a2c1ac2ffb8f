"""rt_hip_join_frame_group: one process per GPU, every rank's kernel storing its stripes straight into ONE shared back
buffer — the multi-process form without an exchange step.  A frame group needs no RCCL, so the ranks may share the box's
one device: `world` worker processes (tests/frame_group_worker.py) render together; the finished frames are compared, bit
for bit, with the oracle's and with the single-GPU frame of this process.
"""
import json
import os
import subprocess
import sys
import uuid
from pathlib import Path

import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = str(ROOT / "tests" / "frame_group_worker.py")


def run_group(tmp_path, world, width, height, spp, scene="basic", frames=2, mode="ok", deadline_ms=None, timeout=240):
    """Start `world` ranks, wait for all of them; returns (their JSON results in rank order, the directory rank 0 saved frames in)."""
    tag = uuid.uuid4().hex[:12]
    frame_file = Path("/dev/shm") / f"rt_hip_test_frame_{tag}"
    np.zeros((height, width), dtype=np.uint32).tofile(frame_file)
    env = dict(os.environ)
    if deadline_ms:
        env["RT_HIP_GROUP_DEADLINE_MS"] = str(deadline_ms)
    try:
        procs = [
            subprocess.Popen([sys.executable, WORKER, str(r), str(world), f"/rt_hip_test_{tag}", str(frame_file), str(width), str(height), str(spp), scene, str(frames), mode, str(tmp_path)],
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
            for r in range(world)
        ]
        results = []
        for p in procs:
            try:
                out, err = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()  # (exactly the processes started above)
                raise
            lines = [line for line in out.splitlines() if line.startswith("{")]
            assert lines, f"rank printed nothing: rc {p.returncode}\n{out}\n{err[-2000:]}"
            results.append(json.loads(lines[-1]))
    finally:
        frame_file.unlink(missing_ok=True)
    assert not [f for f in os.listdir("/dev/shm") if tag in f], "the group left its name behind"
    return sorted(results, key=lambda r: r["rank"]), tmp_path


def test_three_ranks_render_ragged_stripes_into_one_buffer_bit_exact_against_the_oracle(tmp_path):
    width, height, spp = 72, 44, 4  # 44 rows = 5 stripes of 8 + one of 4, dealt to 3 ranks
    results, out = run_group(tmp_path, 3, width, height, spp, frames=3)
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(width, height)
    for r in results:
        assert r["error"] is None and r["frames_done"] == 3, r
        assert r["info"] == {"ranks": 3, "rank": r["rank"], "device": 0, "transport": "shared_frame"}
    for f in (1, 2, 3):
        want, _, want_stats = oracle.render(pod, width, height, seed=f, want_rgb=False)
        assert np.array_equal(np.load(out / f"frame_{f}.npy"), want), f
    assert np.array_equal(np.load(out / "frame_lean.npy"), want)  # the stats == NULL call
    # the frame's counters are the sum over the ranks, on every rank alike; the shares are the stripes' sizes
    for r in results:
        assert r["stats"]["primary_samples"] == width * height * spp
        assert r["stats"]["segments"] == want_stats["segments"]
        shares = [m["primary_samples"] for m in r["member_stats"]]
        assert shares == [width * rows * spp for rows in (16, 16, 12)]
        assert r["phases"]["transport"] == "shared_frame" and r["devices"] == [0, 0, 0]


def test_four_ranks_at_full_size_match_the_single_gpu_frame(tmp_path, tracer):
    width, height, spp = 1920, 1080, 16
    results, out = run_group(tmp_path, 4, width, height, spp, frames=2)
    assert all(r["error"] is None and r["frames_done"] == 2 for r in results), results
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(width, height)
    for f in (1, 2):
        want, _, stats = tracer.render(pod, width, height, seed=f)
        assert np.array_equal(np.load(out / f"frame_{f}.npy"), want)
    assert results[0]["stats"]["segments"] == stats["segments"]
    assert results[0]["stats"]["render_ms"] == pytest.approx(max(m["render_ms"] for m in results[0]["member_stats"]))


def test_a_group_of_one_and_a_buffer_mapped_anew(tmp_path):
    width, height, spp = 64, 40, 2
    pod = rt_amd.Scene.named("dielectric").set_sampling(spp).describe(width, height)
    want = {f: oracle.render(pod, width, height, seed=f, want_rgb=False)[0] for f in (1, 2, 3)}
    results, out = run_group(tmp_path, 1, width, height, spp, scene="dielectric", frames=3)
    assert results[0]["error"] is None
    assert all(np.array_equal(np.load(out / f"frame_{f}.npy"), want[f]) for f in want)
    results, out = run_group(tmp_path, 2, width, height, spp, scene="dielectric", frames=3, mode="remap")
    assert all(r["error"] is None and r["frames_done"] == 3 for r in results), results
    assert all(np.array_equal(np.load(out / f"frame_{f}.npy"), want[f]) for f in want)


def test_more_ranks_than_stripes_and_rows_that_straddle_pages(tmp_path):
    """20 rows are three stripes: the fourth rank owns nothing and still takes part in every frame; 50 pixels a row put no
    stripe boundary on a page boundary (the placement leaves such pages where they are)."""
    width, height, spp = 50, 20, 3
    results, out = run_group(tmp_path, 4, width, height, spp, frames=2)
    assert all(r["error"] is None and r["frames_done"] == 2 for r in results), results
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(width, height)
    for f in (1, 2):
        want, _, want_stats = oracle.render(pod, width, height, seed=f, want_rgb=False)
        assert np.array_equal(np.load(out / f"frame_{f}.npy"), want)
    assert [m["primary_samples"] for m in results[0]["member_stats"]] == [width * 8 * spp, width * 8 * spp, width * 4 * spp, 0]
    assert results[3]["stats"]["segments"] == want_stats["segments"]  # (every rank reports the whole frame)


def test_the_preview_and_the_sm_material_table_go_through_a_frame_group_too(tmp_path, tracer):
    from rt_amd import capi

    width, height, spp = 200, 120, 3
    pod = rt_amd.Scene.named("dielectric").set_sampling(spp).describe(width, height)
    for mode, flags in (("preview", capi.RT_HIP_FLAG_PREVIEW), ("sm", capi.RT_HIP_FLAG_SM_MATERIALS)):
        out = tmp_path / mode
        out.mkdir()
        results, _ = run_group(out, 3, width, height, spp, scene="dielectric", frames=2, mode=mode)
        assert all(r["error"] is None and r["frames_done"] == 2 for r in results), results
        want, _, _ = tracer.render(pod, width, height, seed=2, flags=flags)  # (the single-GPU frames are oracle-checked in test_gpu_parity.py)
        assert np.array_equal(np.load(out / "frame_2.npy"), want), mode


def test_four_hundred_frames_in_a_row_are_each_complete_when_rank_0_returns(tmp_path):
    """Seeds 1, 2, 3 in turn, 400 tiny frames, four rank processes: a frame that rank 0 saw before every rank's stripes were
    in — or that a rank started storing into too early — would differ from the first frame of its seed."""
    results, out = run_group(tmp_path, 4, 160, 100, 1, frames=400, mode="many")
    assert all(r["error"] is None and r["frames_done"] == 400 for r in results), results
    pod = rt_amd.Scene.named("basic").set_sampling(1).describe(160, 100)
    for seed in (1, 2, 3):
        want, _, _ = oracle.render(pod, 160, 100, seed=seed, want_rgb=False)
        assert np.array_equal(np.load(out / f"frame_{seed}.npy"), want)


def test_every_stripe_lives_on_the_numa_node_of_the_gpu_that_stores_it(tmp_path):
    """Two ranks that claim GPUs on different sockets (RT_HIP_NUMA_NODE = rank % 2): before anybody page-locks the
    shared frame, rank 0 moves each stripe's pages to its owner's node (one move_pages call).  At 1920 pixels a stripe of 8
    rows is exactly 15 pages."""
    width, height, spp = 1920, 1080, 2
    results, out = run_group(tmp_path, 2, width, height, spp, frames=2, mode="numa")
    assert all(r["error"] is None and r["frames_done"] == 2 for r in results), results
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(width, height)
    want, _, _ = oracle.render(pod, width, height, seed=2, want_rgb=False)
    assert np.array_equal(np.load(out / "frame_2.npy"), want)
    nodes, host_nodes = results[0]["page_nodes"], results[0]["host_nodes"]
    assert nodes is not None and len(nodes) == 2025
    if len(host_nodes) < 2:
        pytest.skip(f"this host has NUMA nodes {host_nodes}: nothing to spread over")
    expected = [(page // 15) % 2 for page in range(2025)]
    wrong = sum(1 for have, want_node in zip(nodes, expected) if have != want_node)
    if wrong > 20:
        # (the frame above was correct; where its pages lie is a request the kernel may decline — a node short of free memory did on two
        # boxes of round 5 for the single-GPU call's mbind: tests/test_gpu_launch_paths.py — and the module promises the frame, not the placement)
        pytest.skip(f"frame correct; {wrong} of 2025 pages are not on their stripe's node on this host; first stripes: {nodes[:45]}")


def test_a_rank_whose_buffer_is_not_the_shared_one_fails_the_frame_everywhere(tmp_path):
    results, _ = run_group(tmp_path, 3, 64, 40, 1, mode="private")
    for r in results:
        assert r["frames_done"] == 0 and "not a mapping of the memory rank 0 renders into" in r["error"], r


def test_a_rank_called_with_other_arguments_fails_the_frame_everywhere(tmp_path):
    results, _ = run_group(tmp_path, 3, 64, 40, 1, frames=3, mode="mismatch")
    for r in results:
        assert r["frames_done"] == 1 and "seed 3" in r["error"] and "seed 2" in r["error"], r
    assert "RT_HIP_INVALID_ARGUMENT" in results[1]["error"]


def test_the_float_mean_is_refused_and_the_refusal_reaches_every_rank(tmp_path):
    results, _ = run_group(tmp_path, 2, 64, 40, 1, mode="float")
    assert "float mean" in results[0]["error"] and "RT_HIP_UNSUPPORTED" in results[0]["error"]
    assert "float mean" in results[1]["error"] and results[1]["frames_done"] == 0


def test_a_rank_that_leaves_or_dies_does_not_hang_the_others(tmp_path):
    import time

    results, _ = run_group(tmp_path, 3, 64, 40, 1, frames=3, mode="leaves")
    assert [r["frames_done"] for r in results] == [1, 1, 1]
    assert all("rank 2 left the group" in r["error"] for r in results[:2]), results
    t0 = time.time()
    results, _ = run_group(tmp_path, 3, 64, 40, 1, frames=3, mode="dies", deadline_ms=1500)
    assert time.time() - t0 < 60
    assert [r["frames_done"] for r in results] == [1, 1, 1]
    # (round 4: the others no longer have only their deadline — the dead rank's pid is in the group's block, and a rank that
    # has waited for milliseconds looks whether the processes it waits for still exist; the deadline stays the backstop)
    assert all("rank 2's process" in r["error"] and "is gone" in r["error"] or "waited 1500 ms" in r["error"] for r in results[:2]), results
    assert any("is gone" in r["error"] for r in results[:2])


def test_join_times_out_when_a_rank_stays_away():
    t = rt_amd.HipRayTracer(0)
    with pytest.raises(rt_amd.RtHipError, match="RT_HIP_TIMEOUT"):
        t.join_frame_group(0, 2, f"/rt_hip_test_{uuid.uuid4().hex[:12]}", timeout_ms=300)
    # the context stayed a plain single-GPU one
    pod = rt_amd.Scene.named("basic").set_sampling(1).describe(32, 16)
    want, _, _ = oracle.render(pod, 32, 16, seed=1, want_rgb=False)
    assert np.array_equal(t.render(pod, 32, 16, seed=1)[0], want)
    with pytest.raises(rt_amd.RtHipError, match="RT_HIP_INVALID_ARGUMENT"):
        t.join_frame_group(0, 1, "no-slash", timeout_ms=300)
    t.close()

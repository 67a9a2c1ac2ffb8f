"""The frame group's meeting protocol (rt_amd/csrc/frame_group.hpp: what rt_hip_join_frame_group and rt_hip_render use
between the processes of a one-process-per-GPU renderer), on the CPU with forked processes: tests/native/frame_group_ranks.cpp.
The GPU side of the same thing is tests/test_gpu_frame_group.py."""
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


def build(tmp_path_factory, name, extra):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = tmp_path_factory.mktemp("frame_group") / name
    done = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", *extra, str(ROOT / "tests" / "native" / "frame_group_ranks.cpp"), "-o", str(exe), "-lrt"],
                          capture_output=True, text=True, timeout=300)
    assert done.returncode == 0, done.stderr[-3000:]
    return exe


@pytest.fixture(scope="module")
def ranks_exe(tmp_path_factory):
    return build(tmp_path_factory, "frame_group_ranks", [])


@pytest.fixture(scope="module")
def sanitized_exe(tmp_path_factory):
    """the same program under AddressSanitizer + UndefinedBehaviorSanitizer (host code: sanitizers run on the CPU build only)"""
    return build(tmp_path_factory, "frame_group_ranks_asan", ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"])


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_frames_are_complete_when_rank_0_returns(ranks_exe, world):
    run = subprocess.run([str(ranks_exe), "frames", str(world)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK:"), run.stdout + run.stderr


@pytest.mark.parametrize("scenario", ["private", "mismatch", "leaves", "absent", "silent", "killed"])
def test_a_rank_that_misbehaves_ends_the_frame_on_every_rank(ranks_exe, scenario):
    """private buffer / other arguments / a rank that leaves / never joins / goes silent / is killed outright (its
    pid is in the block — no need to wait for the deadline): nobody hangs, everybody is told why."""
    run = subprocess.run([str(ranks_exe), scenario, "4"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK:"), run.stdout + run.stderr


@pytest.mark.parametrize("world", [2, 4])
def test_a_block_left_behind_under_the_same_name_is_replaced_not_joined(ranks_exe, world):
    """non-zero ranks used to attach to ANY existing object of that name — a leftover of a crashed run already has
    its magic set and its counters at their targets, so they sailed through the join while rank 0 failed with EEXIST."""
    run = subprocess.run([str(ranks_exe), "stale", str(world)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK:"), run.stdout + run.stderr


@pytest.mark.parametrize("scenario", ["frames", "private", "leaves", "silent", "killed", "stale"])
def test_the_protocol_is_clean_under_address_and_undefined_behaviour_sanitizers(sanitized_exe, scenario):
    run = subprocess.run([str(sanitized_exe), scenario, "3"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.startswith("OK:") and "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stdout + run.stderr[-3000:]

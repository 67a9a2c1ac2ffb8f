"""The frame group's meeting protocol (rt_amd/csrc/frame_group.hpp: what rt_hip_join_frame_group and rt_hip_render use
between the processes of a one-process-per-GPU renderer), on the CPU with forked processes: tests/native/frame_group_ranks.cpp.
The GPU side of the same thing is tests/test_gpu_frame_group.py."""
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


@pytest.fixture(scope="module")
def ranks_exe(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = tmp_path_factory.mktemp("frame_group") / "frame_group_ranks"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pthread", str(ROOT / "tests" / "native" / "frame_group_ranks.cpp"), "-o", str(exe), "-lrt"],
                           capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-3000:]
    return exe


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_frames_are_complete_when_rank_0_returns(ranks_exe, world):
    run = subprocess.run([str(ranks_exe), "frames", str(world)], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK:"), run.stdout + run.stderr


@pytest.mark.parametrize("scenario", ["private", "mismatch", "leaves", "absent", "silent"])
def test_a_rank_that_misbehaves_ends_the_frame_on_every_rank(ranks_exe, scenario):
    """private buffer / other arguments / a rank that leaves / never joins / goes silent: nobody hangs, everybody is told why."""
    run = subprocess.run([str(ranks_exe), scenario, "4"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0 and run.stdout.startswith("OK:"), run.stdout + run.stderr

"""The pixel carrier (rt_amd/csrc/delivery.cpp) — the host threads that take finished pixels from the module's own page-locked
frame into the caller's buffer while the GPU is still storing the rest — on the CPU, with a thread playing the device
(tests/native/pixel_carrier_test.cpp).  The reference's render() fills a plain host buffer (image_view, src/image.cpp:9-13);
this is the last step of that on the way out of rt_hip_render, and the one place where the module writes caller memory."""
import shutil
import subprocess

import pytest

from tests.conftest import ROOT

SOURCES = [str(ROOT / "tests" / "native" / "pixel_carrier_test.cpp"), str(ROOT / "rt_amd" / "csrc" / "delivery.cpp")]


def build(tmp_path, name, *flags):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = tmp_path / name
    built = subprocess.run([cxx, "-std=c++17", "-O2", "-g", "-Wall", "-Wextra", *flags, *SOURCES, "-o", str(exe), "-lpthread"], capture_output=True, text=True)
    if built.returncode != 0 and "-fsanitize=thread" in flags and "tsan" in built.stderr.lower():
        pytest.skip("ThreadSanitizer runtime not installed")
    assert built.returncode == 0, built.stderr
    return exe


@pytest.mark.parametrize("helpers", [0, 1, 3, 6])
def test_frames_arrive_whole_and_the_staging_frame_is_left_clean(tmp_path, helpers):
    exe = build(tmp_path, "carrier")
    out = subprocess.run([str(exe), str(helpers), "3"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
    print(out.stdout.strip())


def test_helpers_confined_to_fewer_cpus_than_threads_still_deliver_the_frame(tmp_path):
    """the default frame mode twice degraded to "the caller's thread carried the frame alone" (+0.4 ms on the
    headline) on a GPU box — helpers that exist but do not get to run.  Whatever keeps them off a CPU (an affinity mask that
    squeezes seven spinning helpers, the caller's thread and the thread playing the device onto ONE CPU here; a cgroup's CPU-time
    quota on the box), nothing may be lost or torn: every frame arrives whole, abandoned frames write nothing foreign, the
    staging frame is left clean — only late.  (The carrier reports how many bands went early: rt_hip_phases.carrier_bands_early,
    bench.py's drop_in_breakdown.bands_early.)"""
    import os

    exe = build(tmp_path, "carrier_confined")
    one_cpu = {sorted(os.sched_getaffinity(0))[0]}
    out = subprocess.run([str(exe), "7", "2"], capture_output=True, text=True, timeout=900, preexec_fn=lambda: os.sched_setaffinity(0, one_cpu))
    assert out.returncode == 0 and out.stdout.startswith("OK"), out.stdout + out.stderr
    print(out.stdout.strip())


def test_thread_sanitizer_finds_nothing(tmp_path):
    """The portable build of the carrier (atomic word accesses instead of SSE2 lines) under -fsanitize=thread: the hand-over
    of a job between the caller's thread and the helpers, and every access to the two buffers."""
    exe = build(tmp_path, "carrier_tsan", "-fsanitize=thread", "-DRT_HIP_CARRIER_PORTABLE")
    out = subprocess.run([str(exe), "3", "2"], capture_output=True, text=True, timeout=600, env={"TSAN_OPTIONS": "halt_on_error=1"})
    assert out.returncode == 0 and out.stdout.startswith("OK") and "ThreadSanitizer" not in out.stderr, out.stdout + out.stderr[-4000:]

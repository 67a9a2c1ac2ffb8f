"""The host-side C++ (TOML reader, scene loader, camera) and the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer.

GPU sanitizers are not available on this pool; the CPU build is: a native harness (tests/native/host_fuzz_harness.cpp) is
compiled with -fsanitize=address,undefined and fed the generated TOML documents of test_toml_reader.py, its malformed ones,
and a thousand random mutations of the shipped scenes (bytes deleted, overwritten, brackets / quotes / huge numbers inserted).
Any report aborts the harness (-fno-sanitize-recover)."""
import random
import shutil
import subprocess

import pytest

from tests.conftest import PLANES_SCENE, PREVIEW_SCENE, ROOT
from tests.test_toml_reader import MALFORMED, Emitter

SOURCES = ["rt_amd/host/host_capi.cpp", "rt_amd/host/scene.cpp", "rt_amd/host/toml_subset.cpp", "oracle/cpu_ref.cpp"]
INSERTS = [b"[", b"]", b"{", b"}", b'"', b"'", b"=", b",", b"\n", b"#", b".", b"-", b"e", b"_", b"\\", b"\x00", b"\xff", b"99999999999999999999", b"1e999", b"nan", b"[[", b"]]"]


def corpus(directory):
    documents = [Emitter(random.Random(seed)).document().encode() for seed in range(300)] + [t.encode() for t in MALFORMED]
    scenes = [p.read_bytes() for p in sorted((ROOT / "scenes").glob("*.toml"))] + [PLANES_SCENE.encode(), PREVIEW_SCENE.encode()]
    r = random.Random(1)
    for _ in range(1000):
        b = bytearray(r.choice(scenes))
        for _ in range(r.randint(1, 6)):
            kind, at = r.random(), r.randrange(len(b)) if b else 0
            if kind < 0.3 and b:
                del b[at : at + r.randint(1, 8)]
            elif kind < 0.6:
                b[at:at] = r.choice(INSERTS)
            elif b:
                b[at] = r.randrange(256)
        documents.append(bytes(b))
    paths = []
    for i, text in enumerate(documents):
        path = directory / f"{i:05d}.toml"
        path.write_bytes(text)
        paths.append(str(path))
    return paths


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ (the image has it)")
def test_scene_front_end_and_oracle_are_clean_under_asan_and_ubsan(tmp_path):
    harness = tmp_path / "harness"
    build = subprocess.run(
        ["g++", "-std=c++20", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-mfma", "-o", str(harness), "tests/native/host_fuzz_harness.cpp", *SOURCES, "-lpthread"],
        cwd=ROOT, capture_output=True, text=True, timeout=600,
    )
    assert build.returncode == 0, build.stderr[-2000:]
    documents = corpus(tmp_path)
    run = subprocess.run([str(harness), *documents], cwd=ROOT, capture_output=True, text=True, timeout=600, env={"ASAN_OPTIONS": "detect_leaks=1", "PATH": "/usr/bin:/bin"})
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    loaded, rejected, total = (int(run.stdout.split()[i]) for i in (0, 2, 4))
    assert total == len(documents) and loaded + rejected == total
    assert loaded > 100 and rejected > 400  # the mutations do both: still-valid scenes and broken ones

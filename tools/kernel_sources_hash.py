#!/usr/bin/env python3
"""sha256 (first 16 hex digits) of the sources the render kernels are compiled from.  Counter figures (profiles/pmc_counters.json)
carry it; bench.py recomputes it and drops figures measured on other kernels (a stale static number)."""
import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
FILES = ["rt_amd/csrc/kernels.hip", "rt_amd/csrc/kernels.hpp", "rt_amd/csrc/contract.hpp", "rt_amd/csrc/scan.hpp"]


def kernel_sources_sha16(root: Path = ROOT) -> str:
    h = hashlib.sha256()
    for name in FILES:
        h.update(name.encode())
        h.update((root / name).read_bytes())
    return h.hexdigest()[:16]


def built_kernel_sources_sha16(root: Path = ROOT) -> str | None:
    """The hash the Makefile recorded when it linked the library in use (rt_amd/lib/librt_hip.kernels.sha16, which travels with
    the .so): counter figures are stamped with THIS, not with whatever the sources say now — a library that was not rebuilt
    after an edit must not pass for the new kernels (it happened once in round 4)."""
    import os

    lib = Path(os.environ.get("RT_HIP_LIBRARY", root / "rt_amd" / "lib" / "librt_hip.so"))
    stamp = lib.with_name("librt_hip.kernels.sha16") if lib.name == "librt_hip.so" else None
    try:
        return stamp.read_text().strip() if stamp else None
    except OSError:
        return None


if __name__ == "__main__":
    print(kernel_sources_sha16())

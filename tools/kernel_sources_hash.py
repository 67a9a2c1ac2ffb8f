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


if __name__ == "__main__":
    print(kernel_sources_sha16())

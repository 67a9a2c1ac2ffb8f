"""Where host pages live and how to put them somewhere: move_pages(2) through ctypes (no libnuma in the image)."""
import ctypes
import os

import numpy as np

_libc = ctypes.CDLL(None, use_errno=True)
SYS_MOVE_PAGES = 279  # x86-64
MPOL_MF_MOVE = 2


def _pages(buf):
    page = os.sysconf("SC_PAGESIZE")
    first = (buf.ctypes.data + page - 1) // page * page
    count = (buf.ctypes.data + buf.nbytes - first) // page
    return (ctypes.c_void_p * count)(*[first + i * page for i in range(count)]), count


def nodes_of(buf) -> dict:
    """{node: pages} of the whole pages inside `buf` (negative keys: -errno, e.g. -2 = not present)"""
    pages, count = _pages(buf)
    status = (ctypes.c_int * count)()
    if _libc.syscall(SYS_MOVE_PAGES, 0, ctypes.c_ulong(count), pages, None, status, 0) != 0:
        raise OSError(ctypes.get_errno(), "move_pages (query)")
    vals, counts = np.unique(np.array(status[:]), return_counts=True)
    return {int(v): int(c) for v, c in zip(vals, counts)}


def move_to(buf, node: int) -> None:
    """Move every whole page inside `buf` to `node` (pages must exist: write the buffer first)."""
    pages, count = _pages(buf)
    nodes = (ctypes.c_int * count)(*([node] * count))
    status = (ctypes.c_int * count)()
    if _libc.syscall(SYS_MOVE_PAGES, 0, ctypes.c_ulong(count), pages, nodes, status, MPOL_MF_MOVE) != 0:
        raise OSError(ctypes.get_errno(), "move_pages")


def host_nodes() -> list:
    return sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())


def buffer_on(node: int, shape) -> np.ndarray:
    """A uint32 array whose pages are all on `node` (checked)."""
    buf = np.empty(shape, dtype=np.uint32)
    buf.fill(0)
    move_to(buf, node)
    placed = nodes_of(buf)
    assert set(placed) == {node}, (node, placed)
    return buf

"""Multi-GPU frame rendering: one process per GPU, row-stripe partition, ONE gather to rank 0.

The path shards trivially — every pixel depends only on the scene and its own random stream (the reference
already exploits this with ``threads.for_range`` over pixel indices, src/renderers/mg_ray_tracer.cpp:203) — so
there is exactly one exchange step: the per-rank framebuffers are collected on rank 0 (``torch.distributed``
gather; backend "nccl" is RCCL over xGMI on ROCm) and de-interleaved into the frame.

Partition: stripes of ``stripe_rows`` (8) image rows dealt round-robin to ranks, so that the cheap sky rows and
the expensive ground rows are spread evenly.  Random streams are keyed by the GLOBAL pixel index, so the
assembled frame is bit-identical for every world size (tests/test_gpu_parity.py, tests/test_distributed_gpu.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from . import capi


def owner_and_local_row(y: int, world: int, stripe_rows: int) -> tuple[int, int]:
    """(rank that renders image row y, row index inside that rank's compact buffer)."""
    stripe = y // stripe_rows
    return stripe % world, (stripe // world) * stripe_rows + y % stripe_rows


def local_row_table(height: int, world: int, stripe_rows: int) -> np.ndarray:
    """int64[height, 2]: (rank, local row) of every image row — the de-interleave map."""
    table = np.empty((height, 2), dtype=np.int64)
    for y in range(height):
        table[y] = owner_and_local_row(y, world, stripe_rows)
    return table


def padded_rows(height: int, world: int, stripe_rows: int) -> int:
    """Rows of the per-rank buffer used for the equal-sized gather (max over ranks of the rows owned)."""
    table = local_row_table(height, world, stripe_rows)
    return int(max((table[table[:, 0] == r, 1].max() + 1) if (table[:, 0] == r).any() else 0 for r in range(world)))


def gather_stripes(local: torch.Tensor, dst: int = 0, group=None) -> torch.Tensor | None:
    """Collect every rank's compact stripe buffer (int32[padded_rows, W]) on `dst`.

    Returns int32[world, padded_rows, W] on `dst`, None elsewhere.  One collective, no ring: each rank's buffer
    travels once, directly to the root."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local.unsqueeze(0)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (more ranks than GPUs): gloo cannot gather device tensors, so the stripes take a detour
        # through host memory; the RCCL path below moves them GPU to GPU
        staged = gather_stripes(local.cpu(), dst=dst, group=group)
        return staged.to(local.device) if staged is not None else None
    if rank == dst:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, gather_list=list(out.unbind(0)), dst=dst, group=group)
        return out
    dist.gather(local, gather_list=None, dst=dst, group=group)
    return None


def assemble(gathered: torch.Tensor, width: int, height: int, stripe_rows: int, tracer=None, stream: int | None = None) -> torch.Tensor:
    """De-interleave int32[world, padded_rows, W] into the int32[H, W] frame.

    Device tensors go through the HIP kernel behind rt_hip_assemble_device and REQUIRE a tracer; host tensors
    (the gloo rehearsal of the collective in tests) are permuted with torch indexing."""
    world = gathered.shape[0]
    if gathered.is_cuda:
        if tracer is None:
            raise capi.RtHipError(1, "assemble of device buffers needs the HipRayTracer that owns the GPU")
        frame = torch.empty((height, width), dtype=gathered.dtype, device=gathered.device)
        tracer.assemble_device(width, height, world, stripe_rows, gathered.data_ptr(), frame.data_ptr(), stream)
        return frame
    table = torch.from_numpy(local_row_table(height, world, stripe_rows))
    return gathered[table[:, 0], table[:, 1], :width].contiguous()


class DistributedFrame:
    """Per-rank state for rendering frames of one size across the ranks of a process group.

    `tracers`: one HipRayTracer per frame in flight, each with the scene already uploaded.  With two of them
    consecutive frames alternate between two HIP streams (each with its own stripe buffer and context), so that the
    tail of one frame's launch, its gather and its assemble overlap the start of the next frame's launch; every
    frame is still complete and bit-identical — only throughput changes.  One tracer = strictly one frame at a time
    on the caller's current stream."""

    def __init__(self, tracers, width: int, height: int, stripe_rows: int = capi.RT_HIP_DEFAULT_STRIPE_ROWS, group=None):
        self.tracers = list(tracers) if isinstance(tracers, (list, tuple)) else [tracers]
        self.tracer = self.tracers[0]
        self.width, self.height, self.stripe_rows = width, height, stripe_rows
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.padded_rows = padded_rows(height, self.world, stripe_rows)
        device = torch.device("cuda", self.tracer.device)
        self.locals = [torch.zeros((self.padded_rows, width), dtype=torch.int32, device=device) for _ in self.tracers]
        self.local = self.locals[0]
        if len(self.tracers) > 1:
            self.streams = [torch.cuda.Stream(device=device) for _ in self.tracers]
            for stream in self.streams:
                stream.wait_stream(torch.cuda.current_stream(device))  # the buffers above were zeroed on the current stream
        else:
            self.streams = [None]
        self.frames = 0

    def render(self, seed: int = 1, flags: int = 0) -> torch.Tensor | None:
        """Render this rank's stripes, gather, assemble.  Returns the int32[H, W] frame on rank 0, else None.
        With several frames in flight the result is valid once its stream has been synchronised with
        (torch.cuda.synchronize() does)."""
        slot = self.frames % len(self.tracers)
        self.frames += 1
        if self.streams[slot] is None:
            return self._render_on(slot, torch.cuda.current_stream(), seed, flags)
        with torch.cuda.stream(self.streams[slot]):
            return self._render_on(slot, self.streams[slot], seed, flags)

    def _render_on(self, slot: int, stream, seed: int, flags: int) -> torch.Tensor | None:
        tracer, local = self.tracers[slot], self.locals[slot]
        tracer.render_device(
            self.width,
            self.height,
            local.data_ptr(),
            seed=seed,
            flags=flags,
            partition=(self.rank, self.world, self.stripe_rows),
            stream=stream.cuda_stream,
        )
        if self.world == 1:
            # one frame at a time: the stripe buffer IS the frame; with several in flight the buffer is reused two
            # frames later, so the caller gets its own copy
            return local[: self.height] if len(self.tracers) == 1 else local[: self.height].clone()
        gathered = gather_stripes(local, dst=0, group=self.group)
        if gathered is None:
            return None
        return assemble(gathered, self.width, self.height, self.stripe_rows, tracer=tracer, stream=stream.cuda_stream)


# ---- one process per GPU: bringing up the module's own multi-GPU renderer without a way to hang ---------------------------


def all_agree(ok: bool, device="cpu", group=None) -> bool:
    """True iff `ok` holds on EVERY rank (one all_reduce(MIN) of the process group the launcher already has)."""
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def negotiate_rank_renderer(create, join, make_id, vote_device="cpu", group=None, log=None):
    """Every rank of the process group ends up with a joined rank of ONE renderer — or every rank ends up with None.

    rt_hip_create_rank is collective (ncclCommInitRank): a rank that fails in front of it — no device, wrong
    architecture, out of memory — would leave the others waiting in it.  So the part that can fail alone comes first, the
    ranks VOTE, and only then does anybody enter the collective call; that call carries a deadline of its own
    (rt_hip_join_ranks: RT_HIP_TIMEOUT), and the ranks vote once more on its outcome.

        create()            -> a context of this rank's own (HipRayTracer(device=...)); may raise
        make_id()           -> the 128-byte RCCL id; called on rank 0 only; may raise
        join(tracer, id)    -> tracer.join_ranks(rank, world, id, timeout_ms); collective; may raise

    Returns (tracer, None) on every rank, or (None, reason) on every rank (contexts already created are closed).
    """
    rank = dist.get_rank(group)
    say = log or (lambda message: None)
    tracer, problem = None, None
    try:
        tracer = create()
    except Exception as e:  # noqa: BLE001 - any failure of this rank is a "no" in the vote
        problem = f"rank {rank}: create failed: {e}"
        say(problem)
    if not all_agree(tracer is not None, vote_device, group):
        if tracer is not None:
            tracer.close()
        return None, problem or "another rank could not create its context"

    ids = [None]
    if rank == 0:
        try:
            ids[0] = make_id()
        except Exception as e:  # noqa: BLE001
            problem = f"rank 0: unique id failed: {e}"
            say(problem)
    dist.broadcast_object_list(ids, src=0, group=group)
    if ids[0] is None:  # every rank sees the same None: nobody enters the collective call
        tracer.close()
        return None, problem or "rank 0 could not make the RCCL id"

    joined = False
    try:
        join(tracer, ids[0])
        joined = True
    except Exception as e:  # noqa: BLE001
        problem = f"rank {rank}: join failed: {e}"
        say(problem)
    if not all_agree(joined, vote_device, group):
        tracer.close()
        return None, problem or "another rank could not join the communicator"
    return tracer, None

"""Multi-GPU frame assembly: one process per GPU, pixels sharded by row bands, one all-reduce.

The reference renders rows from one shared atomic counter inside one process
(/root/reference/src/camera.h:111-129); samples never interact, so rows can be owned by different GPUs.
Row bands of `shard_rows` rows are dealt round-robin (band b belongs to rank b % world), which spreads the
expensive image regions over all ranks.  Every rank writes only the pixels it owns into a zeroed full-size
framebuffer; a SUM all-reduce (RCCL over xGMI on GPUs, gloo on CPU in the tests) then yields the complete
image on every rank - exact, because every pixel has exactly one non-zero contributor.
"""
from __future__ import annotations

from typing import Callable, Tuple


def shard_of(rank: int, world: int, shard_rows: int = 4) -> Tuple[int, int, int]:
    """(shard_index, shard_count, shard_rows) for bdpt_render_params."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return (rank, world, max(1, shard_rows))


def owned_rows(height: int, shard: Tuple[int, int, int], y0: int = 0, y1: int = None):
    """Rows j in [y0, y1) with (j // shard_rows) % shard_count == shard_index - the ownership rule of
    bdpt_render_params (include/bdpt_amd.h)."""
    idx, count, rows = shard
    y1 = height if y1 is None else min(y1, height)
    return [j for j in range(max(0, y0), y1) if (j // rows) % count == idx]


def render_frame(render_shard: Callable, fb, rank: int, world: int, shard_rows: int = 4, dist=None, reduce_events=None):
    """One frame: `render_shard(fb, shard)` must fill the pixels `shard` owns in the torch tensor `fb`
    (H x W x 3, float64) and leave the others alone.  With world > 1, `fb` is zeroed first and summed
    over ranks afterwards with `dist.all_reduce`.  `reduce_events`: a list that receives one (start, end) pair of
    torch.cuda.Event around the all-reduce (its duration includes the wait for the slowest rank's render)."""
    shard = shard_of(rank, world, shard_rows)
    if world > 1:
        fb.zero_()
    render_shard(fb, shard)
    if world > 1:
        if reduce_events is not None and getattr(fb, "is_cuda", False):
            import torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            _all_reduce(fb, dist)
            ev[1].record()
            reduce_events.append(ev)
            return fb
        _all_reduce(fb, dist)
    return fb


def _all_reduce(fb, dist):
    if getattr(fb, "is_cuda", False) and dist.get_backend() == "gloo":
        # rehearsal on a machine without RCCL peers: gloo reduces a host copy
        host = fb.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        fb.copy_(host)
    else:
        dist.all_reduce(fb, op=dist.ReduceOp.SUM)
    return fb

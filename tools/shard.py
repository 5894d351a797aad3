"""Multi-GPU sharding helpers for the hot path: frames are independent units (tiles never communicate), so rank r owns
frames r, r+N, r+2N, ... and runs the full single-GPU pipeline; the only collectives are the timing barrier and the
max-reduction of the per-rank wall time (no data-path collective)."""
from __future__ import annotations


def frames_for_rank(total_frames: int, rank: int, world: int) -> list[int]:
    """frame f -> rank f mod N."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, total_frames, world))


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """MAX all-reduce of a python float (returns value itself when not distributed)."""
    if dist is None or not dist.is_initialized():
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.cpu()[0])


def barrier(dist=None):
    if dist is not None and dist.is_initialized():
        dist.barrier()

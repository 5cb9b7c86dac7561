"""Multi-GPU layout of the verify/accept path: independent sequences, one process per GPU, no
collective on the accept path (the reference shards the same way by hand: run.sh:76-91,
generate_images.py:185-192 `--slice`).  The only communication is the benchmark's own timing
reduction."""
from __future__ import annotations

from typing import List, Tuple


def sequence_ids(rank: int, world: int, seqs_per_rank: int) -> List[int]:
    """Global ids of the sequences a rank owns (contiguous slices, like `--slice a-b`)."""
    return list(range(rank * seqs_per_rank, (rank + 1) * seqs_per_rank))


def sequence_seed(base: int, global_seq_id: int) -> int:
    return base + global_seq_id


def reduce_timing(dist, seconds: float, tokens: float, device=None) -> Tuple[float, float]:
    """(max over ranks of the wall time, sum over ranks of the accepted tokens).  `dist` is
    torch.distributed (nccl == RCCL on the GPU box, gloo in the CPU tests) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, tokens
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    n = torch.tensor([tokens], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t[0]), float(n[0])

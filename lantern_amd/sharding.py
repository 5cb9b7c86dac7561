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


def reduce_timing(dist, seconds: float, tokens: float, device=None, group=None) -> Tuple[float, float]:
    """(max over ranks of the wall time, sum over ranks of the accepted tokens).  `dist` is
    torch.distributed (nccl == RCCL on the GPU box, gloo in the CPU tests) or None; `group` the
    process group to reduce over (None = the default group)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds, tokens
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    n = torch.tensor([tokens], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return float(t[0]), float(n[0])


# ---------------------------------------------------------------------------------------------------------------
# Prompt slices and the per-slice statistics file of the reference's driver (entrypoints/generate_images.py:185-201,
# :250-309; run.sh:76-91 starts one process per GPU with `--slice a-b`).  Only the data contract is kept -- which
# prompts a process owns and the JSON it leaves behind -- so per-rank results merge the way the reference's do.
def parse_slice(spec: str) -> Tuple[int, int]:
    """'start-end' -> (start, end), with the reference's checks (generate_images.py:186-191)."""
    import re
    if not re.match(r"^\d+-\d+$", spec):
        raise ValueError(f"Invalid format: '{spec}'. Expected format is 'start-end'.")
    start, end = map(int, spec.split("-"))
    if not start < end:
        raise ValueError(f"Invalid range: '{spec}'. Start value must be less than end value.")
    return start, end


def slice_prompts(prompts: List[str], spec=None) -> List[str]:
    return prompts if spec is None else prompts[slice(*parse_slice(spec))]


def rank_slices(n_prompts: int, world: int) -> List[str]:
    """Contiguous `--slice` strings, one per GPU, covering 0..n_prompts (what run.sh writes out by hand)."""
    per = -(-n_prompts // world)
    return [f"{r * per}-{min(n_prompts, (r + 1) * per)}" for r in range(world) if r * per < n_prompts]


def statistics_entry(prompt: str, step_compression: float, latency: float) -> dict:
    return {"prompt": prompt, "step_compression": step_compression, "latency": latency}


def write_global_statistics(output_dir: str, entries: dict, start_idx: int, end_idx: int) -> str:
    """`global_statistics_{start}_{end}.json`: {"prompt_{idx}": {prompt, step_compression, latency}} (generate_images.py:296-306)."""
    import json
    import os
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, f"global_statistics_{start_idx}_{end_idx}.json")
    with open(path, "w") as f:
        json.dump(entries, f, indent=4)
    return path


def merge_global_statistics(paths: List[str]) -> dict:
    """Host-side merge of the per-process files: mean step compression / latency over all prompts."""
    import json
    merged = {}
    for p in paths:
        with open(p) as f:
            merged.update(json.load(f))
    n = max(len(merged), 1)
    return {"prompts": len(merged), "mean_step_compression": sum(e["step_compression"] for e in merged.values()) / n,
            "mean_latency": sum(e["latency"] for e in merged.values()) / n, "entries": merged}

"""Multi-GPU plan for the transcription path: independent windows are dealt out statically, every rank holds a
full weight replica and runs the whole path on its windows; the only exchange is one gather of the small result
tensors (token ids, counts, boundary frames).  One process per GPU, torch.distributed ("nccl" = RCCL on ROCm)."""
from __future__ import annotations

from typing import Dict

import torch
import torch.distributed as dist


def shard_windows(n_windows: int, rank: int, world: int) -> range:
    """Contiguous, balanced slice of the window list for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_windows, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_results(local: Dict[str, torch.Tensor], n_windows: int, rank: int, world: int) -> Dict[str, torch.Tensor]:
    """all_gather of per-window result tensors (first dim = local windows), restored to global window order.
    Ranks may hold different window counts, so tensors are padded to the largest shard for the collective."""
    if world == 1:
        return local
    most = max(len(shard_windows(n_windows, r, world)) for r in range(world))
    via_host = dist.get_backend() == "gloo"      # gloo gathers host tensors only (CPU tests, single-GPU rehearsals)
    out: Dict[str, torch.Tensor] = {}
    for key, t in local.items():
        dev = torch.device("cpu") if via_host else t.device
        pad = torch.zeros((most,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        pad[: t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out[key] = torch.cat([parts[r][: len(shard_windows(n_windows, r, world))] for r in range(world)], dim=0).to(t.device)
    return out

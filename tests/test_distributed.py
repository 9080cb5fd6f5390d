"""World-size-2 gloo test of the multi-GPU plan: static window sharding, no data-path collective, one gather of
token / boundary tensors at the end (what bench.py does over RCCL)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from whisper_amd.parallel import ResultExchange, gather_results, gather_results_per_key, shard_windows

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_transcribe(window_ids):
    """Deterministic stand-in for Engine.transcribe_windows: results depend only on the window id."""
    n = len(window_ids)
    tokens = torch.zeros((n, 8), dtype=torch.int32)
    bounds = torch.zeros((n, 9), dtype=torch.int32)
    counts = torch.zeros((n,), dtype=torch.int32)
    for i, w in enumerate(window_ids):
        k = 1 + w % 7
        tokens[i, :k] = torch.arange(k, dtype=torch.int32) + 100 * w
        bounds[i, : k + 1] = torch.arange(k + 1, dtype=torch.int32) * (w + 1)
        counts[i] = k
    return {"new_tokens": tokens, "boundaries": bounds, "n_text": counts}


def _worker(rank, world, port, n_windows, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_windows(n_windows, rank, world)
    res = _fake_transcribe(list(mine))
    full = gather_results(res, n_windows, rank, world)
    plain = gather_results_per_key(res, n_windows, rank, world)
    # the per-step exchange: one collective per gather, buffers reused from the second step on
    ex = ResultExchange(n_windows, rank, world)
    first = ex.gather(res)
    send, recv = ex._send, ex._recv
    again = ex.gather({k: v + 1 for k, v in res.items()})
    reused = ex._send is send and ex._recv is recv
    if rank == 0:
        torch.save({"packed": full, "per_key": plain, "first": first, "again": again, "collectives": ex.collectives,
                    "reused": reused}, os.path.join(out_dir, "gathered.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_windows_is_a_partition():
    for n in (0, 1, 5, 128, 129):
        for world in (1, 2, 3, 8):
            parts = [list(shard_windows(n, r, world)) for r in range(world)]
            assert sum(parts, []) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_two_rank_gather_matches_single_process(tmp_path):
    n_windows, world, port = 11, 2, 29511 + os.getpid() % 500
    mp.spawn(_worker, args=(world, port, n_windows, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "gathered.pt"), weights_only=True)
    ref = _fake_transcribe(list(range(n_windows)))
    for k in ref:
        assert torch.equal(got["packed"][k], ref[k]), k             # the packed exchange restores global window order
        assert torch.equal(got["per_key"][k], ref[k]), k            # ... and equals one all_gather per tensor
        assert torch.equal(got["first"][k], ref[k]) and torch.equal(got["again"][k], ref[k] + 1), k
    assert got["collectives"] == 2 and got["reused"]                # one collective per step, no new buffers


def test_packed_exchange_rejects_other_dtypes():
    ex = ResultExchange(4, 0, 2)
    try:
        ex._prepare({"x": torch.zeros((2, 3), dtype=torch.float32)}, torch.device("cpu"))
    except TypeError as e:
        assert "i32" in str(e)
    else:
        raise AssertionError("a float tensor must not be packed into the i32 exchange")

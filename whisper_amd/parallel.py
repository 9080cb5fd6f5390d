"""Multi-GPU plan for the transcription path: independent windows are dealt out statically, every rank holds a
full weight replica and runs the whole path on its windows; the only exchange is one gather of the small result
tensors (token ids, counts, boundary frames).  One process per GPU, torch.distributed ("nccl" = RCCL on ROCm).

Within one GPU, `EnginePool` keeps several batches in flight: the path alternates between kernels bound by different
things (matrix cores in the encoder, HBM in the attention decode kernels, per-kernel latency in the step GEMMs), so the
kernels of two or three independent batches fill each other's gaps."""
from __future__ import annotations

import queue
import threading
from concurrent.futures import Future
from typing import Dict, List, Optional, Sequence

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


class EnginePool:
    """`inflight` engines (library contexts) on one GPU, each driven by its own host thread on its own HIP stream.

    Batches handed to `submit` run concurrently, one per engine; results are identical to a single engine's (every
    context computes alone, on its own buffers -- only the GPU's time is shared).  Measured on whisper-small shapes,
    128 windows per batch: 324 ms per batch with one in flight, 283 ms with two, 275 ms with three.  Each context holds
    its own weights and caches (about 20 GB at 128 windows of whisper-small), which is what a 288 GB GPU has room for.

    The pool answers `max_batch`, `device`, `dims`, `heads` and `transcribe_windows` like an Engine, so it can be passed
    as `whisperer(..., engine=pool)`; callers with many independent batches use `submit` / `map`."""

    def __init__(self, dims, state_dict, *, inflight: int = 2, **engine_kw):
        from .engine import Engine
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        self.engines: List = []
        self._threads: List[threading.Thread] = []
        try:
            for _ in range(inflight):
                self.engines.append(Engine(dims, state_dict, **engine_kw))
        except BaseException:
            for e in self.engines:      # a later context did not fit: give the earlier ones back
                e.close()
            raise
        e0 = self.engines[0]
        self.dims, self.device, self.dtype, self.max_batch, self.heads, self.n_text_ctx = (
            e0.dims, e0.device, e0.dtype, e0.max_batch, e0.heads, e0.n_text_ctx)
        self._jobs: "queue.Queue" = queue.Queue()
        self._threads = [threading.Thread(target=self._worker, args=(e,), daemon=True) for e in self.engines]
        for t in self._threads:
            t.start()

    def _worker(self, eng) -> None:
        torch.cuda.set_device(eng.device)
        stream = torch.cuda.Stream(device=eng.device)        # non-blocking: does not serialise with the other workers
        while True:
            job = self._jobs.get()
            if job is None:
                return
            fut, ready, method, args, kwargs = job
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                with torch.cuda.stream(stream):
                    stream.wait_event(ready)                  # inputs were produced on the submitter's stream
                    res = method(eng, *args, **kwargs) if callable(method) else getattr(eng, method)(*args, **kwargs)
                    stream.synchronize()                      # results are complete when the future resolves
                fut.set_result(res)
            except BaseException as e:                        # noqa: BLE001 -- handed to the waiting caller
                fut.set_exception(e)

    def submit(self, method, *args, **kwargs) -> Future:
        """Queue `engine.<method>(*args, **kwargs)` -- or, for a callable, `method(engine, *args, **kwargs)`: a sequence
        of stage calls that must stay on one context -- on the next free engine; the future resolves to its result."""
        if not self._threads:
            raise RuntimeError("pool is closed")
        fut: Future = Future()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self._jobs.put((fut, ready, method, args, kwargs))
        return fut

    def transcribe_windows_async(self, *args, **kwargs) -> Future:
        return self.submit("transcribe_windows", *args, **kwargs)

    def transcribe_windows(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        return self.transcribe_windows_async(*args, **kwargs).result()

    def map(self, method, arg_lists: Sequence[Sequence]) -> List:
        """Run one call per entry of `arg_lists`, `inflight` at a time; results in submission order."""
        return [f.result() for f in [self.submit(method, *a) for a in arg_lists]]

    def close(self) -> None:
        for _ in self._threads:
            self._jobs.put(None)
        for t in self._threads:
            t.join()
        self._threads = []
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

"""Multi-GPU plan for the transcription path: independent windows are dealt out statically, every rank holds a
full weight replica and runs the whole path on its windows; the only exchange is one gather of the small result
tensors (token ids, counts, boundary frames).  One process per GPU, torch.distributed ("nccl" = RCCL on ROCm).

Within one GPU, `EnginePool` keeps several batches in flight: the path alternates between kernels bound by different
things (matrix cores in the encoder, HBM in the attention decode kernels, per-kernel latency in the step GEMMs), so the
kernels of two or three independent batches fill each other's gaps."""
from __future__ import annotations

import queue
import threading
import weakref
from concurrent.futures import Future
from typing import Dict, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_windows(n_windows: int, rank: int, world: int) -> range:
    """Contiguous, balanced slice of the window list for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_windows, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def gather_results_per_key(local: Dict[str, torch.Tensor], n_windows: int, rank: int, world: int) -> Dict[str, torch.Tensor]:
    """One all_gather per result tensor (first dim = local windows), restored to global window order: the plain form
    the packed exchange below is checked against.  Ranks may hold different window counts, so tensors are padded to
    the largest shard for the collective."""
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


class ResultExchange:
    """The path's only exchange: ONE all_gather per step of the per-window results (token ids, counts, boundary
    frames), packed side by side into one i32 row per window.  Send and receive buffers are allocated once per layout
    and reused, so a step issues a single collective and no allocation."""

    def __init__(self, n_windows: int, rank: int, world: int):
        self.n_windows, self.rank, self.world = n_windows, rank, world
        self.counts = [len(shard_windows(n_windows, r, world)) for r in range(world)]
        self.most = max(self.counts) if self.counts else 0
        self._layout = None
        self._send = self._recv = None
        self.collectives = 0            # all_gather calls issued (one per gather)

    def _prepare(self, local: Dict[str, torch.Tensor], dev: torch.device) -> None:
        layout = tuple((k, tuple(t.shape[1:]), t.dtype) for k, t in local.items())
        if layout == self._layout and self._send is not None and self._send.device == dev:
            return
        for k, shape, dtype in layout:
            if dtype != torch.int32:
                raise TypeError(f"result tensor '{k}' is {dtype}: the packed exchange carries i32 rows")
        self._widths = [int(torch.Size(shape).numel()) for _, shape, _ in layout]
        width = sum(self._widths)
        self._send = torch.zeros((self.most, width), dtype=torch.int32, device=dev)
        self._recv = torch.zeros((self.world, self.most, width), dtype=torch.int32, device=dev)
        self._layout = layout

    def gather(self, local: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        if self.world == 1:
            return local
        first = next(iter(local.values()))
        via_host = dist.get_backend() == "gloo"  # gloo gathers host tensors only (CPU tests, single-GPU rehearsals)
        dev = torch.device("cpu") if via_host else first.device
        self._prepare(local, dev)
        n, col = first.shape[0], 0
        for (k, shape, _), w in zip(self._layout, self._widths):
            self._send[:n, col:col + w].copy_(local[k].reshape(n, w))
            col += w
        # one collective into one contiguous buffer: neither backend needs a staging copy or an allocation for it
        dist.all_gather_into_tensor(self._recv.view(self.world * self.most, -1), self._send)
        self.collectives += 1
        rows = torch.cat([self._recv[r, : self.counts[r]] for r in range(self.world)], dim=0)
        out, col = {}, 0
        for (k, shape, _), w in zip(self._layout, self._widths):
            out[k] = rows[:, col:col + w].reshape((rows.shape[0],) + shape).to(first.device)
            col += w
        return out


def gather_results(local: Dict[str, torch.Tensor], n_windows: int, rank: int, world: int) -> Dict[str, torch.Tensor]:
    """One-off form of ResultExchange.gather (callers that gather every step keep a ResultExchange instead)."""
    return ResultExchange(n_windows, rank, world).gather(local)


def _stop_workers(jobs: "queue.Queue", n: int) -> None:
    for _ in range(n):
        jobs.put(None)


_streams: Dict[tuple, "torch.cuda.Stream"] = {}       # (device index, worker slot) -> stream, reused by later pools
_streams_lock = threading.Lock()


def _worker_stream(device: torch.device, slot: int) -> "torch.cuda.Stream":
    """The HIP stream of worker `slot` on `device`.  Streams are kept and handed to the next pool's worker of the same
    slot: a stream brings a hardware queue and its scratch memory with it, so pools that come and go (one per
    whisperer(inflight=...) call) must not create new ones each time."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), slot)
    with _streams_lock:
        if key not in _streams:
            _streams[key] = torch.cuda.Stream(device=device)   # non-blocking: does not serialise with the other workers
        return _streams[key]


def _worker_loop(jobs: "queue.Queue", eng, slot: int = 0) -> None:
    """One worker: its own engine (library context), its own HIP stream, jobs from the shared queue."""
    torch.cuda.set_device(eng.device)
    stream = _worker_stream(eng.device, slot)
    while True:
        job = jobs.get()
        if job is None:
            eng.close()                                  # the context and its device memory go with the worker
            return
        fut, ready, method, args, kwargs = job
        del job
        if not fut.set_running_or_notify_cancel():
            del fut, ready, method, args, kwargs
            continue
        try:
            with torch.cuda.stream(stream):
                stream.wait_event(ready)                  # inputs were produced on the submitter's stream
                res = method(eng, *args, **kwargs) if callable(method) else getattr(eng, method)(*args, **kwargs)
                stream.synchronize()                      # results are complete when the future resolves
            fut.set_result(res)
            del res
        except BaseException as e:                        # noqa: BLE001 -- handed to the waiting caller
            fut.set_exception(e)
        # nothing of the finished job (inputs, results) stays referenced while the worker waits for the next one
        del fut, ready, method, args, kwargs


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
        # the workers know the job queue and their own engine, not the pool: a pool nobody refers to any more is
        # collected, and its finalizer posts the stop sentinels so that threads, contexts and their memory go with it
        self._threads = [threading.Thread(target=_worker_loop, args=(self._jobs, e, k), daemon=True) for k, e in enumerate(self.engines)]
        for t in self._threads:
            t.start()
        self._finalizer = weakref.finalize(self, _stop_workers, self._jobs, len(self._threads))

    def submit(self, method, *args, **kwargs) -> Future:
        """Queue `engine.<method>(*args, **kwargs)` -- or, for a callable, `method(engine, *args, **kwargs)`: a sequence
        of stage calls that must stay on one context -- on the next free engine; the future resolves to its result."""
        if not self._threads:
            raise RuntimeError("pool is closed")
        if not all(t.is_alive() for t in self._threads):
            raise RuntimeError("a worker thread of the pool has died; close the pool and build a new one")
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
        self._finalizer()                    # posts one stop sentinel per worker (once)
        for t in self._threads:
            t.join()
        self._threads = []
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

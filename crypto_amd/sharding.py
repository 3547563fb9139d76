"""Multi-GPU sharding of one MSM: contiguous point slices per rank, one all_gather of the
64-byte partial sums, host-side addition (bn256_g1_sum).  One process per GPU;
torch.distributed is plumbing only (backend "nccl" = RCCL on GPUs, "gloo" in CPU tests)."""
from __future__ import annotations

import ctypes as C

from . import _lib


def split_even(n_total: int, world: int, rank: int) -> tuple[int, int]:
    """(offset, length) of rank's contiguous slice; the first n_total % world ranks get one more."""
    base, rem = divmod(n_total, world)
    off = rank * base + min(rank, rem)
    return off, base + (1 if rank < rem else 0)


class _Exchange:
    """Persistent buffers for the per-step exchange (allocated once per device and world size:
    the timed loop of bench.py must not allocate)."""

    def __init__(self, world: int, device, batch: int = 1):
        import torch
        self.world, self.device, self.batch = world, device, batch
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.h_in = torch.zeros(64 * batch, dtype=torch.uint8)
        self.h_out = torch.zeros(64 * batch * world, dtype=torch.uint8)
        if on_gpu:
            self.h_in, self.h_out = self.h_in.pin_memory(), self.h_out.pin_memory()
            self.d_in = torch.empty(64 * batch, dtype=torch.uint8, device=device)
            self.d_out = torch.empty(64 * batch * world, dtype=torch.uint8, device=device)
        else:
            self.d_in, self.d_out = self.h_in, self.h_out
        self.on_gpu = on_gpu

    def run(self, part: bytes) -> list[bytes]:
        """Blocking exchange of ONE point per rank (slot 0 of the batch; the other slots travel unused)."""
        import torch
        import torch.distributed as dist
        if len(part) != 64:
            raise ValueError("a marshalled G1 point is 64 bytes")
        self.h_in.numpy()[:64] = memoryview(part)
        if self.on_gpu:
            self.d_in.copy_(self.h_in, non_blocking=True)
        dist.all_gather_into_tensor(self.d_out, self.d_in)
        if self.on_gpu:
            self.h_out.copy_(self.d_out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        raw = self.h_out.numpy().tobytes()
        return [raw[64 * self.batch * i:64 * self.batch * i + 64] for i in range(self.world)]


class AsyncExchange:
    """The same exchange, split in start / finish so that it never stalls the host: while two MSM
    calls are in flight the GPU is full (k_accumulate holds every wave slot for ~1 ms), an RCCL
    kernel issued now is scheduled only when that kernel drains, and a host that waited for it could
    not enqueue the next MSM in time.  start() enqueues copy-in, all_gather and copy-out on a side
    stream and returns a ticket; finish() waits for that ticket's event -- one or two steps later,
    when it has long completed.  `slots` exchanges may be outstanding.

    `batch` > 1: one all_gather carries the partial sums of up to `batch` finished calls (start_batch /
    finish_batch).  An exchange is three stream operations and a collective whose host cost (~80 us: torch
    dispatch, RCCL's enqueue) adds to every step of the loop that issues it -- rehearsed with one rank,
    1.228 -> 1.320 ms per 2^20-point step; a prover with several commitments in flight has several results to
    hand over at once anyway."""

    def __init__(self, world: int, device, slots: int = 4, batch: int = 1):
        import torch
        self.world, self.device, self.slots, self.batch = world, device, slots, batch
        self.on_gpu = device is not None and torch.device(device).type == "cuda"
        self.bufs = [_Exchange(world, device, batch) for _ in range(slots)]
        self.count = [0] * slots
        self.busy = [False] * slots
        if self.on_gpu:
            self.stream = torch.cuda.Stream(device=device)
            self.events = [torch.cuda.Event() for _ in range(slots)]

    def start(self, part: bytes) -> int:
        return self.start_batch([part])

    def finish(self, k: int) -> list[bytes]:
        return self.finish_batch(k)[0]

    def start_batch(self, parts: list[bytes]) -> int:
        import torch
        import torch.distributed as dist
        if not 1 <= len(parts) <= self.batch or any(len(p) != 64 for p in parts):
            raise ValueError("1..batch marshalled G1 points of 64 bytes each")
        k = self.busy.index(False)  # ValueError when every slot is outstanding
        ex = self.bufs[k]
        self.count[k] = len(parts)
        ex.h_in.numpy()[:64 * len(parts)] = memoryview(b"".join(parts))
        if not self.on_gpu:
            dist.all_gather_into_tensor(ex.d_out, ex.d_in)
        else:
            with torch.cuda.stream(self.stream):
                ex.d_in.copy_(ex.h_in, non_blocking=True)
                work = dist.all_gather_into_tensor(ex.d_out, ex.d_in, async_op=True)
                work.wait()  # orders the side stream after the collective; does not block the host
                ex.h_out.copy_(ex.d_out, non_blocking=True)
                self.events[k].record(self.stream)
        self.busy[k] = True
        return k

    def finish_batch(self, k: int) -> list[list[bytes]]:
        """Per call of the batch: the world_size partial points, in rank order."""
        if not self.busy[k]:
            raise ValueError("no exchange outstanding on this ticket")
        self.busy[k] = False
        if self.on_gpu:
            self.events[k].synchronize()
        raw, b = self.bufs[k].h_out.numpy().tobytes(), self.batch
        return [[raw[64 * (b * r + j):64 * (b * r + j) + 64] for r in range(self.world)] for j in range(self.count[k])]


def run_pipelined(steps: int, depth: int, submit, wait, ex=None, host_us=None):
    """`steps` MSM calls with `depth` of them in flight (bn256_msm_submit / bn256_msm_wait through the
    callables `submit() -> ticket`, `wait(ticket) -> 64-byte point`); returns the last result.  Every
    result is collected -- and, when `ex` (an AsyncExchange) is given, exchanged between the ranks and
    summed -- before this returns.  The exchange of step i is started right after the next submit
    and finished when `depth` newer ones are outstanding, so the host neither delays the next call
    nor waits for an RCCL kernel that queues behind a running accumulation.  depth = 1 is the
    strictly blocking sequence: MSM, exchange, sum, next MSM.  `host_us` (3 floats) accumulates the
    host microseconds spent in submit / wait / exchange."""
    import time
    q, xq, todo, res = [], [], [], None
    host_us = host_us if host_us is not None else [0.0] * 3

    batch = getattr(ex, "batch", 1) if depth > 1 else 1  # results that travel in one exchange

    def finish_oldest():
        nonlocal res
        k = xq.pop(0)
        for ranks in (ex.finish_batch(k) if hasattr(ex, "finish_batch") else [ex.finish(k)]):
            res = combine(ranks)

    def exchange(drain: bool = False):
        t1 = time.perf_counter()
        while len(todo) >= batch or (todo and drain):
            while len(xq) >= depth:  # at most `depth` exchanges outstanding
                finish_oldest()
            take = [todo.pop(0) for _ in range(min(batch, len(todo)))]
            xq.append(ex.start_batch(take) if hasattr(ex, "start_batch") else ex.start(take[0]))
        while xq and (drain or depth == 1):
            finish_oldest()
        host_us[2] += (time.perf_counter() - t1) * 1e6

    def collect():
        nonlocal res
        t0 = time.perf_counter()
        p = wait(q.pop(0))
        host_us[1] += (time.perf_counter() - t0) * 1e6
        if ex is None:
            res = p
        else:
            todo.append(p)
            if depth == 1:
                exchange()

    for _ in range(steps):
        t0 = time.perf_counter()
        q.append(submit())
        host_us[0] += (time.perf_counter() - t0) * 1e6
        if ex is not None and depth > 1:
            exchange()  # after the submit: the next call is enqueued before the host turns to the exchange
        if len(q) == depth:
            collect()
    while q:
        collect()
    if ex is not None:
        exchange(drain=True)
    return res


_exchanges: dict = {}


def all_gather_points(part: bytes, device=None) -> list[bytes]:
    """Exchange one marshalled point per rank.  Returns world_size points (rank order)."""
    import os
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [part]
    if dist.get_world_size() == 1 and not os.environ.get("BENCH_FORCE_DIST"):  # rehearsal switch of bench.py
        return [part]
    if len(part) != 64:
        raise ValueError("a marshalled G1 point is 64 bytes")
    world = dist.get_world_size()
    key = (world, str(device))
    ex = _exchanges.get(key)
    if ex is None:
        ex = _exchanges[key] = _Exchange(world, device)
    return ex.run(part)


def combine(partials: list[bytes]) -> bytes:
    """Sum of marshalled partial points on the host (bn256_g1_sum)."""
    lib = _lib.load()
    out = C.create_string_buffer(64)
    rc = lib.bn256_g1_sum(b"".join(partials), len(partials), out)
    if rc:
        raise RuntimeError(lib.bn256_strerror(rc).decode())
    return out.raw

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

    def __init__(self, world: int, device):
        import torch
        self.world, self.device = world, device
        on_gpu = device is not None and torch.device(device).type == "cuda"
        self.h_in = torch.empty(64, dtype=torch.uint8)
        self.h_out = torch.empty(64 * world, dtype=torch.uint8)
        if on_gpu:
            self.h_in, self.h_out = self.h_in.pin_memory(), self.h_out.pin_memory()
            self.d_in = torch.empty(64, dtype=torch.uint8, device=device)
            self.d_out = torch.empty(64 * world, dtype=torch.uint8, device=device)
        else:
            self.d_in, self.d_out = self.h_in, self.h_out
        self.on_gpu = on_gpu

    def run(self, part: bytes) -> list[bytes]:
        import torch
        import torch.distributed as dist
        self.h_in.numpy()[:] = memoryview(part)
        if self.on_gpu:
            self.d_in.copy_(self.h_in, non_blocking=True)
        dist.all_gather_into_tensor(self.d_out, self.d_in)
        if self.on_gpu:
            self.h_out.copy_(self.d_out, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
        raw = self.h_out.numpy().tobytes()
        return [raw[64 * i:64 * i + 64] for i in range(self.world)]


_exchanges: dict = {}


def all_gather_points(part: bytes, device=None) -> list[bytes]:
    """Exchange one marshalled point per rank.  Returns world_size points (rank order)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
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

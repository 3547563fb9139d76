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


def all_gather_points(part: bytes, device=None) -> list[bytes]:
    """Exchange one marshalled point per rank.  Returns world_size points (rank order)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [part]
    world = dist.get_world_size()
    mine = torch.frombuffer(bytearray(part), dtype=torch.uint8)
    if device is not None:
        mine = mine.to(device)
    out = torch.zeros(64 * world, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    raw = bytes(out.cpu().numpy().tobytes())
    return [raw[64 * i:64 * i + 64] for i in range(world)]


def combine(partials: list[bytes]) -> bytes:
    """Sum of marshalled partial points on the host (bn256_g1_sum)."""
    lib = _lib.load()
    out = C.create_string_buffer(64)
    rc = lib.bn256_g1_sum(b"".join(partials), len(partials), out)
    if rc:
        raise RuntimeError(lib.bn256_strerror(rc).decode())
    return out.raw

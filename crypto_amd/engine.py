"""Python host side over the C ABI (include/bn256msm.h).

Names mirror the reference's helpers (go/bp/util.go) so parity tests read like the
reference's own code: Engine.vector_point_scalar_mul == vectorPointScalarMul, etc.
Points are 64-byte Marshal() strings, scalars Python ints in [0, 2^256)."""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Sequence

from . import _lib

PRECOMPUTE = 1
ASYNC_LANES = 8  # BN256_ASYNC_LANES (include/bn256msm.h): calls in flight per context
INF = bytes(64)


class BN256Error(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{msg} (code {code})")
        self.code = code


def scalars_to_bytes(scalars) -> bytes:
    """32-byte big-endian strings (scalarTo32Byte, go/bp/util.go:209-220); bytes pass through."""
    if isinstance(scalars, (bytes, bytearray)):
        return bytes(scalars)
    return b"".join(int(s).to_bytes(32, "big") for s in scalars)


class Bases:
    """Device-resident vector of affine G1 points (a []*bn256.G1 uploaded once)."""

    def __init__(self, eng: "Engine", handle: int):
        self.eng, self.h = eng, handle

    def __len__(self) -> int:
        return self.eng.lib.bn256_bases_len(self.h)

    def download(self, off: int = 0, n: int | None = None) -> list[bytes]:
        n = len(self) - off if n is None else n
        buf = C.create_string_buffer(64 * max(n, 1))
        self.eng._chk(self.eng.lib.bn256_bases_download(self.eng.ctx, self.h, off, n, buf))
        raw = buf.raw
        return [raw[64 * i:64 * i + 64] for i in range(n)]

    def free(self) -> None:
        if self.h:
            self.eng.lib.bn256_bases_free(self.eng.ctx, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FrVec:
    """Device-resident vector of scalars mod bn256.Order (a []*big.Int uploaded once)."""

    def __init__(self, eng: "Engine", handle):
        self.eng, self.h = eng, handle

    def __len__(self) -> int:
        return self.eng.lib.bn256_fr_len(self.h)

    def download(self, off: int = 0, n: int | None = None) -> list[int]:
        n = len(self) - off if n is None else n
        buf = C.create_string_buffer(32 * max(n, 1))
        self.eng._chk(self.eng.lib.bn256_fr_download(self.eng.ctx, self.h, off, n, buf))
        raw = buf.raw
        return [int.from_bytes(raw[32 * i:32 * i + 32], "big") for i in range(n)]

    def device_ptr(self) -> int:
        return self.eng.lib.bn256_fr_device_ptr(self.h) or 0

    def free(self) -> None:
        if self.h:
            self.eng.lib.bn256_fr_free(self.eng.ctx, self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, device: int = 0, devices: Sequence[int] | None = None):
        """One GPU (`device`), or -- devices=[...] -- ONE context over several GPUs of the node
        (bn256_ctx_create_multi: handles are sharded by contiguous slice, msm / msm_dev / msm_fr run
        every slice side by side and add the partial points on the host)."""
        self.lib = _lib.load()
        ctx = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            rc = self.lib.bn256_ctx_create_multi(arr, len(devices), C.byref(ctx))
        else:
            rc = self.lib.bn256_ctx_create(device, C.byref(ctx))
        if rc:
            raise BN256Error(rc, self.lib.bn256_strerror(rc).decode())
        self.ctx = ctx

    def close(self) -> None:
        if getattr(self, "ctx", None):
            self.lib.bn256_ctx_destroy(self.ctx)
            self.ctx = None

    def _chk(self, rc: int) -> None:
        if rc:
            raise BN256Error(rc, self.lib.bn256_strerror(rc).decode())

    # ------------------------------------------------------------------ handles
    def upload(self, points: Sequence[bytes] | bytes, flags: int = 0) -> Bases:
        raw = points if isinstance(points, (bytes, bytearray)) else b"".join(points)
        if len(raw) % 64:
            raise ValueError("points must be 64 bytes each")
        h = C.c_void_p()
        self._chk(self.lib.bn256_bases_upload(self.ctx, bytes(raw), len(raw) // 64, flags, C.byref(h)))
        return Bases(self, h)

    def random_bases(self, seed: int, n: int, first: int = 0, flags: int = 0) -> Bases:
        h = C.c_void_p()
        self._chk(self.lib.bn256_random_bases(self.ctx, seed, first, n, flags, C.byref(h)))
        return Bases(self, h)

    def random_scalars_into(self, seed: int, n: int, dev_ptr: int, first: int = 0) -> None:
        self._chk(self.lib.bn256_random_scalars(self.ctx, seed, first, n, C.c_void_p(dev_ptr)))

    def set_tuning(self, window_bits: int = 0, chunk: int = 0) -> None:
        self._chk(self.lib.bn256_ctx_set_tuning(self.ctx, window_bits, chunk))

    def set_phase_timing(self, on: bool = True) -> None:
        """Per-phase HIP events for last_timing() (off by default: they cost stream time)."""
        self._chk(self.lib.bn256_ctx_set_phase_timing(self.ctx, 1 if on else 0))

    def last_timing(self) -> list[float]:
        arr = (C.c_float * 7)()
        k = self.lib.bn256_last_timing(self.ctx, arr, 7)
        return [arr[i] for i in range(k)]

    def stream(self) -> int:
        return self.lib.bn256_ctx_stream(self.ctx) or 0

    def selftest(self) -> int:
        return self.lib.bn256_selftest(self.ctx)

    # ------------------------------------------------------------------ the hot path
    def msm(self, bases: Bases, scalars: Sequence[int] | bytes, off: int = 0, n: int | None = None) -> bytes:
        raw = scalars if isinstance(scalars, (bytes, bytearray)) else scalars_to_bytes(scalars)
        n = len(raw) // 32 if n is None else n
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm(self.ctx, bases.h, off, n, bytes(raw), out))
        return out.raw

    def msm_host_ptr(self, bases: Bases, host_ptr: int, n: int, off: int = 0) -> bytes:
        """bn256_msm with the scalars at a host address (pageable or pinned): the upload is part of the call."""
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm(self.ctx, bases.h, off, n, C.cast(C.c_void_p(host_ptr), C.c_char_p), out))
        return out.raw

    def msm_dev(self, bases: Bases, dev_ptr: int, n: int, off: int = 0) -> bytes:
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm_dev(self.ctx, bases.h, off, n, C.c_void_p(dev_ptr), out))
        return out.raw

    def msm_dev_noalloc(self, bases: Bases, dev_ptr: int, n: int, off: int = 0) -> bytes:
        """msm_dev with every workspace in the library's static device array: no hipMalloc / hipFree
        (include/bn256msm.h: bn256_msm_dev_noalloc; BN256_ESCRATCH above its size limit)."""
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm_dev_noalloc(self.ctx, bases.h, off, n, C.c_void_p(dev_ptr), out))
        return out.raw

    def msm_submit(self, bases: Bases, dev_ptr: int, n: int, off: int = 0) -> int:
        """Enqueues msm_dev on a free lane and returns its ticket (include/bn256msm.h: bn256_msm_submit)."""
        t = C.c_int(-1)
        self._chk(self.lib.bn256_msm_submit(self.ctx, bases.h, off, n, C.c_void_p(dev_ptr), C.byref(t)))
        return t.value

    def msm_wait(self, ticket: int) -> bytes:
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm_wait(self.ctx, ticket, out))
        return out.raw

    def msm_multi(self, segments: Sequence[tuple]) -> bytes:
        """segments: (bases, off, scalars) or (bases, off, scalars, stride)."""
        k = len(segments)
        hs = (C.c_void_p * max(k, 1))(*[s[0].h for s in segments])
        offs = (C.c_size_t * max(k, 1))(*[s[1] for s in segments])
        strides = (C.c_size_t * max(k, 1))(*[(s[3] if len(s) > 3 else 1) for s in segments])
        raws = [scalars_to_bytes(s[2]) for s in segments]
        ns = (C.c_size_t * max(k, 1))(*[len(r) // 32 for r in raws])
        ptrs = (C.c_char_p * max(k, 1))(*raws)
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm_multi_strided(self.ctx, k, hs, offs, strides, ns, ptrs, out))
        return out.raw

    def msm_multi_submit(self, segments: Sequence[tuple]) -> int:
        """segments: (bases, off, scalars), stride 1; enqueued on a free lane, msm_wait(ticket) returns the point."""
        k = len(segments)
        hs = (C.c_void_p * max(k, 1))(*[s[0].h for s in segments])
        offs = (C.c_size_t * max(k, 1))(*[s[1] for s in segments])
        raws = [scalars_to_bytes(s[2]) for s in segments]
        ns = (C.c_size_t * max(k, 1))(*[len(r) // 32 for r in raws])
        ptrs = (C.c_char_p * max(k, 1))(*raws)
        t = C.c_int(-1)
        self._chk(self.lib.bn256_msm_multi_submit(self.ctx, k, hs, offs, ns, ptrs, C.byref(t)))
        return t.value

    # reference-named helpers (go/bp/util.go) -----------------------------------------
    def vector_point_scalar_mul(self, g: Bases, a: Sequence[int]) -> bytes:
        """vectorPointScalarMul, go/bp/util.go:165-175 (length mismatch panics there)."""
        if len(g) != len(a):
            raise ValueError("invalid length for scalar mul")
        return self.msm(g, a)

    def vector_point_scalar_mul_bppp(self, g: Bases, a: Sequence[int]) -> bytes:
        """go/bppp/util.go:178-196: empty g -> identity, short a zero-padded, long a truncated."""
        n = len(g)
        if n == 0:
            return INF
        a = list(a)[:n] + [0] * max(0, n - len(a))
        return self.msm(g, a)

    def inner_product_h(self, n: int, g: Bases, h: Bases, u: "Bases | bytes", a: Sequence[int], a1: Sequence[int],
                        b: Sequence[int], b1: Sequence[int], c: int) -> bytes:
        """InnerProductH, go/bp/util.go:15-27: <a, g[:n/2]> + <a1, g[n/2:]> + <b, h[:n/2]> + <b1, h[n/2:]> + c*u
        as ONE device MSM of five segments.  Odd n panics "invalid n" there; each half goes through
        vectorPointScalarMul, whose length check (go/bp/util.go:166-168) is kept."""
        if n % 2 != 0:
            raise ValueError("invalid n")
        n1 = n // 2
        if len(g) < n1 or len(h) < n1 or len(a) != n1 or len(b) != n1 or len(a1) != len(g) - n1 or len(b1) != len(h) - n1:
            raise ValueError("invalid length for scalar mul")
        own = not isinstance(u, Bases)
        uh = self.upload(u) if own else u
        try:
            if len(uh) != 1:
                raise ValueError("u must be a single point")
            return self.msm_multi([(g, 0, a), (g, n1, a1), (h, 0, b), (h, n1, b1), (uh, 0, [c])])
        finally:
            if own:
                uh.free()

    def vector_point_mul_on_scalar(self, g: Bases, a: int, off: int = 0, n: int | None = None) -> Bases:
        """vectorPointMulOnScalar, go/bp/util.go:177-183."""
        n = len(g) - off if n is None else n
        h = C.c_void_p()
        self._chk(self.lib.bn256_scalar_mul_batch(self.ctx, g.h, off, n, int(a).to_bytes(32, "big"), 1, C.byref(h)))
        return Bases(self, h)

    def scalar_mul_each(self, g: Bases, a: Sequence[int], off: int = 0) -> Bases:
        """h1[i] = H[i] * yinvn[i], go/bp/main.go:156-159."""
        h = C.c_void_p()
        self._chk(self.lib.bn256_scalar_mul_batch(self.ctx, g.h, off, len(a), scalars_to_bytes(a), 0, C.byref(h)))
        return Bases(self, h)

    def hadamard_point_mul(self, a: Bases, b: Bases) -> Bases:
        """hadamardPointMul, go/bp/util.go:185-195."""
        if len(a) != len(b):
            raise ValueError("invalid length for scalar mul")
        h = C.c_void_p()
        self._chk(self.lib.bn256_add_batch(self.ctx, a.h, 0, b.h, 0, len(a), C.byref(h)))
        return Bases(self, h)

    def fold(self, lo: Bases, lo_off: int, hi: Bases, hi_off: int, n: int, s_lo: int, s_hi: int,
             stride: int = 1) -> Bases:
        """s_lo*lo[i] + s_hi*hi[i]: go/bp/main.go:355-356 fused."""
        h = C.c_void_p()
        self._chk(self.lib.bn256_fold(self.ctx, lo.h, lo_off, hi.h, hi_off, stride, n,
                                      int(s_lo).to_bytes(32, "big"), int(s_hi).to_bytes(32, "big"), C.byref(h)))
        return Bases(self, h)

    def fold2(self, lo: Bases, lo_off: int, n_lo: int, hi: Bases, hi_off: int, n_hi: int, s_lo: int, s_hi: int,
              stride: int = 1) -> Bases:
        """Ragged fold: a missing side counts as infinity (vectorPointsAdd, go/bppp/util.go:279-293)."""
        h = C.c_void_p()
        self._chk(self.lib.bn256_fold2(self.ctx, lo.h, lo_off, n_lo, hi.h, hi_off, n_hi, stride,
                                       int(s_lo).to_bytes(32, "big"), int(s_hi).to_bytes(32, "big"), C.byref(h)))
        return Bases(self, h)

    def fold2_submit(self, lo: Bases, lo_off: int, n_lo: int, hi: Bases, hi_off: int, n_hi: int, s_lo: int, s_hi: int,
                     stride: int = 1) -> tuple:
        """fold2 on a free lane: (handle, ticket); the handle's contents are defined after fold_wait(ticket)."""
        h, t = C.c_void_p(), C.c_int(-1)
        self._chk(self.lib.bn256_fold2_submit(self.ctx, lo.h, lo_off, n_lo, hi.h, hi_off, n_hi, stride,
                                              int(s_lo).to_bytes(32, "big"), int(s_hi).to_bytes(32, "big"),
                                              C.byref(h), C.byref(t)))
        return Bases(self, h), t.value

    def drain(self) -> None:
        """Waits for everything enqueued and voids all outstanding tickets (bn256_ctx_drain)."""
        self._chk(self.lib.bn256_ctx_drain(self.ctx))

    def fold_wait(self, ticket: int) -> None:
        self._chk(self.lib.bn256_fold_wait(self.ctx, ticket))

    # Fr vectors on the device (go/bp/util.go:106-163, go/bppp/util.go:143-176,237-277) -----
    def fr_upload(self, scalars: Sequence[int] | bytes) -> FrVec:
        raw = scalars if isinstance(scalars, (bytes, bytearray)) else scalars_to_bytes(scalars)
        h = C.c_void_p()
        self._chk(self.lib.bn256_fr_upload(self.ctx, bytes(raw), len(raw) // 32, C.byref(h)))
        return FrVec(self, h)

    def fr_random(self, seed: int, n: int, first: int = 0) -> FrVec:
        h = C.c_void_p()
        self._chk(self.lib.bn256_fr_random(self.ctx, seed, first, n, C.byref(h)))
        return FrVec(self, h)

    def fr_fold(self, a: FrVec, a_off: int, n_a: int, b: FrVec, b_off: int, n_b: int, s_a: int, s_b: int,
                stride: int = 1) -> FrVec:
        """out[i] = s_a*a[a_off+i*stride] + s_b*b[b_off+i*stride], a missing side reads 0."""
        h = C.c_void_p()
        self._chk(self.lib.bn256_fr_fold(self.ctx, a.h, a_off, n_a, b.h, b_off, n_b, stride,
                                         int(s_a).to_bytes(32, "big"), int(s_b).to_bytes(32, "big"), C.byref(h)))
        return FrVec(self, h)

    def fr_dot(self, a: FrVec, a_off: int, a_stride: int, n_a: int, b: FrVec, b_off: int, b_stride: int, n_b: int,
               mu: int | None = None) -> int:
        """vectorMul (mu None) / weightVectorMul (weights mu^(i+1)) over views."""
        out = C.create_string_buffer(32)
        mub = None if mu is None else int(mu).to_bytes(32, "big")
        self._chk(self.lib.bn256_fr_dot(self.ctx, a.h, a_off, a_stride, n_a, b.h, b_off, b_stride, n_b, mub, out))
        return int.from_bytes(out.raw, "big")

    def fr_dot_multi(self, items: Sequence[tuple]) -> list[int]:
        """items: (a, a_off, a_stride, n_a, b, b_off, b_stride, n_b[, mu]) -- k inner products, one device call."""
        k = len(items)
        m = max(k, 1)
        col = lambda j: (C.c_size_t * m)(*[it[j] for it in items])  # noqa: E731
        mus = [(int(it[8]).to_bytes(32, "big") if len(it) > 8 and it[8] is not None else None) for it in items]
        out = C.create_string_buffer(32 * m)
        self._chk(self.lib.bn256_fr_dot_multi(self.ctx, k, (C.c_void_p * m)(*[it[0].h for it in items]), col(1), col(2), col(3),
                                              (C.c_void_p * m)(*[it[4].h for it in items]), col(5), col(6), col(7),
                                              (C.c_char_p * m)(*mus), out))
        return [int.from_bytes(out.raw[32 * j:32 * j + 32], "big") for j in range(k)]

    def fr_dot_multi_dev(self, items: Sequence[tuple]) -> FrVec:
        """The same k >= 1 inner products with the results left in HBM as a new vector (bn256_fr_dot_multi_dev);
        under fr_defer_sync(True) nothing is waited for until the vector is downloaded."""
        k = len(items)
        col = lambda j: (C.c_size_t * k)(*[it[j] for it in items])  # noqa: E731
        mus = [(int(it[8]).to_bytes(32, "big") if len(it) > 8 and it[8] is not None else None) for it in items]
        out = C.c_void_p()
        self._chk(self.lib.bn256_fr_dot_multi_dev(self.ctx, k, (C.c_void_p * k)(*[it[0].h for it in items]), col(1), col(2),
                                                  col(3), (C.c_void_p * k)(*[it[4].h for it in items]), col(5), col(6),
                                                  col(7), (C.c_char_p * k)(*mus), C.byref(out)))
        return FrVec(self, out)

    def fr_defer_sync(self, on: bool) -> None:
        """bn256_ctx_fr_defer_sync: Fr calls without a host-visible result stop waiting for the stream."""
        self._chk(self.lib.bn256_ctx_fr_defer_sync(self.ctx, 1 if on else 0))

    def fr_fold_multi(self, items: Sequence[tuple]) -> list[FrVec]:
        """items: (a, a_off, n_a, b, b_off, n_b, s_a, s_b, stride) -- k folds, one device call."""
        k = len(items)
        m = max(k, 1)
        col = lambda j: (C.c_size_t * m)(*[it[j] for it in items])  # noqa: E731
        outs = (C.c_void_p * m)()
        self._chk(self.lib.bn256_fr_fold_multi(self.ctx, k, (C.c_void_p * m)(*[it[0].h for it in items]), col(1), col(2),
                                               (C.c_void_p * m)(*[it[3].h for it in items]), col(4), col(5), col(8),
                                               b"".join(int(it[6]).to_bytes(32, "big") for it in items),
                                               b"".join(int(it[7]).to_bytes(32, "big") for it in items), outs))
        return [FrVec(self, C.c_void_p(outs[j])) for j in range(k)]

    def fr_tensor_step(self, items: Sequence[tuple], layout: int = 0) -> list[FrVec]:
        """items: (w or None, lo, hi) -- the fold-scalar tensor after one more round (bn256_fr_tensor_step_multi)."""
        k = len(items)
        m = max(k, 1)
        outs = (C.c_void_p * m)()
        self._chk(self.lib.bn256_fr_tensor_step_multi(
            self.ctx, k, (C.c_void_p * m)(*[(it[0].h if it[0] is not None else None) for it in items]), layout,
            b"".join(int(it[1]).to_bytes(32, "big") for it in items), b"".join(int(it[2]).to_bytes(32, "big") for it in items),
            outs))
        return [FrVec(self, C.c_void_p(outs[j])) for j in range(k)]

    def fr_tensor_expand(self, items: Sequence[tuple], layout: int = 0) -> list[FrVec]:
        """items: (v, n_out, w or None, sel, swap, s_clear, s_set) -- out[m] = w[j] * s(I) * v[I ^ (swap << sel)]: the scalars
        of an MSM over unfolded generators (bn256_fr_tensor_expand_multi)."""
        k = len(items)
        m = max(k, 1)
        outs = (C.c_void_p * m)()
        self._chk(self.lib.bn256_fr_tensor_expand_multi(
            self.ctx, k, (C.c_void_p * m)(*[it[0].h for it in items]), (C.c_size_t * m)(*[it[1] for it in items]),
            (C.c_void_p * m)(*[(it[2].h if it[2] is not None else None) for it in items]), layout,
            (C.c_int * m)(*[int(it[3]) for it in items]), (C.c_int * m)(*[int(it[4]) for it in items]),
            b"".join(int(it[5]).to_bytes(32, "big") for it in items), b"".join(int(it[6]).to_bytes(32, "big") for it in items),
            outs))
        return [FrVec(self, C.c_void_p(outs[j])) for j in range(k)]

    def msm_fr(self, bases: Bases, off: int, stride: int, n: int, v: FrVec, v_off: int = 0, v_stride: int = 1) -> bytes:
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm_fr(self.ctx, bases.h, off, stride, n, v.h, v_off, v_stride, out))
        return out.raw

    @staticmethod
    def _fr_segments(segments: Sequence[tuple]):
        m = max(len(segments), 1)
        return ((C.c_void_p * m)(*[s[0].h for s in segments]), (C.c_size_t * m)(*[s[1] for s in segments]),
                (C.c_size_t * m)(*[s[2] for s in segments]), (C.c_size_t * m)(*[s[3] for s in segments]),
                (C.c_void_p * m)(*[s[4].h for s in segments]), (C.c_size_t * m)(*[s[5] for s in segments]),
                (C.c_size_t * m)(*[s[6] for s in segments]),
                (C.c_char_p * m)(*[(int(s[7]).to_bytes(32, "big") if len(s) > 7 and s[7] is not None else None)
                                   for s in segments]))

    def msm_fr_multi(self, segments: Sequence[tuple]) -> bytes:
        """segments: (bases, off, stride, n, frvec, v_off, v_stride[, scale]) summed as one device MSM."""
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_msm_fr_multi(self.ctx, len(segments), *self._fr_segments(segments), out))
        return out.raw

    def msm_fr_multi_submit(self, segments: Sequence[tuple]) -> int:
        """The same, enqueued on a free lane; msm_wait(ticket) returns the point."""
        t = C.c_int(-1)
        self._chk(self.lib.bn256_msm_fr_multi_submit(self.ctx, len(segments), *self._fr_segments(segments), C.byref(t)))
        return t.value

    def com_batch(self, g: bytes, h: bytes, a: Sequence[int], b: Sequence[int]) -> Bases:
        """[a_i*g + b_i*h]: com() of go/bp/util.go:92-94 over independent pairs (go/ve-ca/main.go:37-40)."""
        if len(a) != len(b):
            raise ValueError("invalid length")
        hd = C.c_void_p()
        self._chk(self.lib.bn256_com_batch(self.ctx, g, h, scalars_to_bytes(a), scalars_to_bytes(b), len(a), C.byref(hd)))
        return Bases(self, hd)

    def g1_sum(self, pts: Sequence[bytes]) -> bytes:
        out = C.create_string_buffer(64)
        self._chk(self.lib.bn256_g1_sum(b"".join(pts), len(pts), out))
        return out.raw

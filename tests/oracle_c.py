"""ctypes loader of the C restatement (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "liboracle.so")


class OracleC:
    def __init__(self):
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
        L = self.L = C.CDLL(LIB)
        cp, sz = C.c_char_p, C.c_size_t
        L.orc_check_point.argtypes = [cp]
        L.orc_g1_add.argtypes = [cp, cp, cp]
        L.orc_g1_scalar_mult.argtypes = [cp, cp, cp]
        L.orc_vector_point_scalar_mul.argtypes = [cp, cp, cp, sz]
        L.orc_vector_point_scalar_mul_mt.argtypes = [cp, cp, cp, sz, C.c_int]
        L.orc_vector_point_mul_on_scalar.argtypes = [cp, cp, cp, sz]
        L.orc_scalar_mul_each.argtypes = [cp, cp, cp, sz]
        L.orc_scalar_mul_each_mt.argtypes = [cp, cp, cp, sz, C.c_int]
        L.orc_hadamard_point_mul.argtypes = [cp, cp, cp, sz]
        L.orc_synth_scalars.argtypes = [cp, C.c_uint64, sz, sz]
        L.orc_synth_scalars.restype = None
        L.orc_synth_bases.argtypes = [cp, C.c_uint64, sz, sz]
        L.orc_synth_bases.restype = None
        L.orc_synth_bases_mt.argtypes = [cp, C.c_uint64, sz, sz]
        L.orc_synth_bases_mt.restype = None
        L.orc_synth_dot.argtypes = [cp, C.c_uint64, C.c_uint64, sz, sz]
        L.orc_synth_dot.restype = None
        L.orc_fr_dot.argtypes = [cp, cp, cp, sz]
        L.orc_fr_dot.restype = None
        L.orc_synth_base_scalars.argtypes = [cp, C.c_uint64, sz, sz]
        L.orc_synth_base_scalars.restype = None
        L.orc_keccak256.argtypes = [cp, cp, sz]
        L.orc_keccak256.restype = None

    def synth_bases(self, seed, n, first=0) -> bytes:
        buf = C.create_string_buffer(64 * max(n, 1))
        self.L.orc_synth_bases_mt(buf, seed, first, n)
        return buf.raw[:64 * n]

    def synth_scalars(self, seed, n, first=0) -> bytes:
        buf = C.create_string_buffer(32 * max(n, 1))
        self.L.orc_synth_scalars(buf, seed, first, n)
        return buf.raw[:32 * n]

    def synth_dot(self, seed_b, seed_s, n, first=0) -> int:
        """sum_i synth_scalar(seed_s, i) * synth_base_scalar(seed_b, i) mod n over [first, first + n)."""
        out = C.create_string_buffer(32)
        self.L.orc_synth_dot(out, seed_b, seed_s, first, n)
        return int.from_bytes(out.raw, "big")

    def fr_dot(self, a_be: bytes, b_be: bytes) -> int:
        """vectorMul (go/bp/util.go:132-142) of two marshalled scalar vectors of equal length."""
        assert len(a_be) == len(b_be) and len(a_be) % 32 == 0
        out = C.create_string_buffer(32)
        self.L.orc_fr_dot(out, a_be, b_be, len(a_be) // 32)
        return int.from_bytes(out.raw, "big")

    def synth_base_scalars(self, seed, n, first=0) -> bytes:
        """k_i of the synthetic bases k_i*G0 (32-byte big-endian each)."""
        buf = C.create_string_buffer(32 * max(n, 1))
        self.L.orc_synth_base_scalars(buf, seed, first, n)
        return buf.raw[:32 * n]

    def g0_mul(self, k: int) -> bytes:
        g0 = (1).to_bytes(32, "big") + (0x8fb501e34aa387f9aa6fecb86184dc21ee5b88d120b5b59e185cac6c5e089667 - 2).to_bytes(32, "big")
        out = C.create_string_buffer(64)
        rc = self.L.orc_g1_scalar_mult(out, g0, (k % 0x8fb501e34aa387f9aa6fecb86184dc212e8d8e12f82b39241a2ef45b57ac7261).to_bytes(32, "big"))
        if rc:
            raise ValueError(f"oracle rc {rc}")
        return out.raw

    def synth_msm_closed_form_scalars(self, seed_b, scalars_be: bytes, first=0) -> bytes:
        """(sum_i a_i*k_i mod n)*G0: the exact MSM of ANY scalar vector over the synthetic bases."""
        n = len(scalars_be) // 32
        return self.g0_mul(self.fr_dot(scalars_be, self.synth_base_scalars(seed_b, n, first)))

    def synth_msm_closed_form(self, seed_b, seed_s, n, first=0) -> bytes:
        """The synthetic MSM's exact answer without the per-term loop: the bases are k_i*G0 with
        known k_i, so the sum is (sum_i s_i*k_i mod n)*G0.  Feasible at 2^24 terms."""
        g0 = (1).to_bytes(32, "big") + (0x8fb501e34aa387f9aa6fecb86184dc21ee5b88d120b5b59e185cac6c5e089667 - 2).to_bytes(32, "big")
        out = C.create_string_buffer(64)
        rc = self.L.orc_g1_scalar_mult(out, g0, self.synth_dot(seed_b, seed_s, n, first).to_bytes(32, "big"))
        if rc:
            raise ValueError(f"oracle rc {rc}")
        return out.raw

    def msm(self, bases: bytes, scalars: bytes, mt: bool = False) -> bytes:
        """vectorPointScalarMul (go/bp/util.go:165-175) on marshalled inputs."""
        n = len(scalars) // 32
        assert len(bases) == 64 * n and n >= 1
        out = C.create_string_buffer(64)
        rc = (self.L.orc_vector_point_scalar_mul_mt(out, bases, scalars, n, 64) if mt
              else self.L.orc_vector_point_scalar_mul(out, bases, scalars, n))
        if rc:
            raise ValueError(f"oracle rc {rc}")
        return out.raw

    def scalar_mul_each(self, bases: bytes, scalars: bytes, broadcast=False) -> bytes:
        n = len(bases) // 64
        out = C.create_string_buffer(64 * max(n, 1))
        rc = self.L.orc_scalar_mul_each_mt(out, bases, scalars, n, int(broadcast))
        if rc:
            raise ValueError(f"oracle rc {rc}")
        return out.raw[:64 * n]

    def add_each(self, a: bytes, b: bytes) -> bytes:
        n = len(a) // 64
        out = C.create_string_buffer(64 * max(n, 1))
        rc = self.L.orc_hadamard_point_mul(out, a, b, n)
        if rc:
            raise ValueError(f"oracle rc {rc}")
        return out.raw[:64 * n]

    def add(self, a: bytes, b: bytes) -> bytes:
        return self.add_each(a, b)

    def mul(self, a: bytes, k: int) -> bytes:
        out = C.create_string_buffer(64)
        rc = self.L.orc_g1_scalar_mult(out, a, k.to_bytes(32, "big"))
        if rc:
            raise ValueError(f"oracle rc {rc}")
        return out.raw

    def check_point(self, a: bytes) -> int:
        return self.L.orc_check_point(a)

    def keccak256(self, m: bytes) -> bytes:
        out = C.create_string_buffer(32)
        self.L.orc_keccak256(out, m, len(m))
        return out.raw


def build_case(orc: OracleC, case):
    """(bases bytes list, scalar ints) for an entry of tests/golden/cases.py."""
    from cases import apply_edits
    name, n, sb, ss, edits = case
    braw = orc.synth_bases(sb, n)
    sraw = orc.synth_scalars(ss, n)
    bases = [braw[64 * i:64 * i + 64] for i in range(n)]
    scalars = [int.from_bytes(sraw[32 * i:32 * i + 32], "big") for i in range(n)]
    apply_edits(bases, scalars, edits)
    return bases, scalars

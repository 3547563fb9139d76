#!/usr/bin/env python3
"""Generates crypto_amd/csrc/fe_mul_gen.inc: the radix-2^29 Montgomery product and square
(finely integrated product scanning) as ONE inline-asm statement each.

Why one statement: on gfx950 hipcc pads every inline-asm statement whose result feeds the
next instruction with `s_nop 0` (it cannot see whether the asm used dst_sel, the gfx940
forwarding hazard) and shuffles tied 64-bit operands through v_mov.  With one multiply per
statement that doubled the instruction count; with one statement per product the body is
exactly 162 v_mad_i64_i32 + 9 v_mul_lo_u32 + 17 v_and_b32 + 17 v_ashrrev_i64 (+1 v_mov).

Register plan inside the statement:
  v[0:1]   64-bit column accumulator (clobbered physical registers: the low / high halves
           must be nameable, and inline asm has no sub-register operand modifier)
  s[88:97] p limbs 0..8 and -p^-1 mod 2^29 (VOP3 instructions cannot take literals)
  %r_j     output limb j; until column j+9 the same register holds the Montgomery
           quotient digit m_j (m_j is last used in column j+8)
"""
import os

P = [0x1e089667, 0x02e56362, 0x0d6d6786, 0x1711a241, 0x0dc21ee5, 0x165c30c2, 0x1fe6a9bf, 0x1c695470, 0x008fb501]
NP = 0x1f17daa9
SP = 88      # first SGPR used for constants (HOIST = False)
ACC = "v[0:1]"
# HOIST: p's limbs and -p^-1 arrive as ten "s" input operands that fe_consts() defines once (one opaque
# s_mov_b32 each, merged and hoisted out of loops by the compiler) instead of ten s_mov_b32 inside every
# statement (90 scalar instructions per mixed addition).
HOIST = os.environ.get("FE_HOIST", "1") != "0"


def sk(j, base):
    return f"%{base + j}" if HOIST else f"s{SP + j}"


def prologue(L):
    if HOIST:
        return
    for j in range(9):
        L.append(f"s_mov_b32 s{SP + j}, {P[j]:#x}")
    L.append(f"s_mov_b32 s{SP + 9}, {NP:#x}")


def consts_fn():
    L = ["struct fe_sk {", "  u32 k[10];", "};",
         "// p's limbs 0..8 and -p^-1 mod 2^29 in scalar registers, each defined by one opaque instruction: the",
         "// compiler cannot rematerialise them as literals, so they stay in ten SGPRs across a kernel's loops.",
         "__device__ __forceinline__ fe_sk fe_consts() {", "  fe_sk c;"]
    for j, v in enumerate(P + [NP]):
        L.append(f'  asm("s_mov_b32 %0, {v:#x}" : "=s"(c.k[{j}]));')
    L += ["  return c;", "}", "", ""]
    return "\n".join(L)


def emit(square):
    L = []
    prologue(L)
    KB = 27  # first constant operand
    # operand numbering: outputs r0..r8 = %0..%8 ; (square: d0..d8 = %9..%17) ; inputs follow
    r = [f"%{j}" for j in range(9)]
    if square:
        d = [f"%{9 + j}" for j in range(9)]
        a = [f"%{18 + j}" for j in range(9)]
        for j in range(9):
            L.append(f"v_lshlrev_b32 {d[j]}, 1, {a[j]}")
    else:
        a = [f"%{9 + j}" for j in range(9)]
        b = [f"%{18 + j}" for j in range(9)]
    first = True

    def mad(x, y):
        nonlocal first
        src2 = "0" if first else ACC
        first = False
        L.append(f"v_mad_i64_i32 {ACC}, vcc, {x}, {y}, {src2}")

    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        if square:
            for i in range(lo, hi + 1):
                j = k - i
                if i < j:
                    mad(a[i], d[j])
                elif i == j:
                    mad(a[i], a[i])
        else:
            for i in range(lo, hi + 1):
                mad(a[i], b[k - i])
        if k <= 8:
            for i in range(0, k):
                mad(r[i], sk(k - i, KB))
            L.append(f"v_mul_lo_u32 {r[k]}, v0, {sk(9, KB)}")
            L.append(f"v_and_b32 {r[k]}, 0x1fffffff, {r[k]}")
            mad(r[k], sk(0, KB))
            L.append(f"v_ashrrev_i64 {ACC}, 29, {ACC}")
        else:
            for i in range(lo, hi + 1):
                mad(r[i], sk(k - i, KB))
            L.append(f"v_and_b32 {r[k - 9]}, 0x1fffffff, v0")
            L.append(f"v_ashrrev_i64 {ACC}, 29, {ACC}")
    L.append(f"v_mov_b32 {r[8]}, v0")
    return L


def emit2():
    """(a*b + c*d) / R: both products' terms in every column, ONE Montgomery reduction.  Operands:
    r0..r8 = %0..%8 (early clobber), a = %9.., b = %18.., c = %27.., d = %36..  Column bound: 18 products and 9
    quotient terms below 2^58 each plus a carry below 2^35: under 2^63 (fe_mul2add's contract in fe29.cuh)."""
    L = []
    prologue(L)
    KB = 45
    r = [f"%{j}" for j in range(9)]
    a = [f"%{9 + j}" for j in range(9)]
    b = [f"%{18 + j}" for j in range(9)]
    c = [f"%{27 + j}" for j in range(9)]
    d = [f"%{36 + j}" for j in range(9)]
    first = True

    def mad(x, y):
        nonlocal first
        src2 = "0" if first else ACC
        first = False
        L.append(f"v_mad_i64_i32 {ACC}, vcc, {x}, {y}, {src2}")

    for k in range(17):
        lo, hi = max(0, k - 8), min(k, 8)
        for i in range(lo, hi + 1):
            mad(a[i], b[k - i])
            mad(c[i], d[k - i])
        if k <= 8:
            for i in range(0, k):
                mad(r[i], sk(k - i, KB))
            L.append(f"v_mul_lo_u32 {r[k]}, v0, {sk(9, KB)}")
            L.append(f"v_and_b32 {r[k]}, 0x1fffffff, {r[k]}")
            mad(r[k], sk(0, KB))
            L.append(f"v_ashrrev_i64 {ACC}, 29, {ACC}")
        else:
            for i in range(lo, hi + 1):
                mad(r[i], sk(k - i, KB))
            L.append(f"v_and_b32 {r[k - 9]}, 0x1fffffff, v0")
            L.append(f"v_ashrrev_i64 {ACC}, 29, {ACC}")
    L.append(f"v_mov_b32 {r[8]}, v0")
    return L


def function2(name):
    lines = emit2()
    body = "\n".join(f'      "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'"=&v"(r.v[{j}])' for j in range(9))
    ins = ", ".join(", ".join(f'"v"({x}.v[{j}])' for j in range(9)) for x in "abcd")
    clob = '"vcc", "v0", "v1", ' + ", ".join(f'"s{SP + j}"' for j in range(10))
    sig = f"__device__ __forceinline__ fe {name}(const fe& a, const fe& b, const fe& c, const fe& d) {{\n  fe r;\n"
    if HOIST:
        ins += ", " + ", ".join(f'"s"(K.k[{j}])' for j in range(10))
        clob = '"vcc", "v0", "v1"'
        sig += "  const fe_sk K = fe_consts();\n"
    return f"{sig}  asm(\n{body}\n      : {outs}\n      : {ins}\n      : {clob});\n  return r;\n}}\n"


def function(name, square):
    lines = emit(square)
    body = "\n".join(f'      "{l}\\n\\t"' for l in lines)
    outs = ", ".join(f'"=&v"(r.v[{j}])' for j in range(9))
    if square:
        outs += ", " + ", ".join(f'"=&v"(d{j})' for j in range(9))
        ins = ", ".join(f'"v"(a.v[{j}])' for j in range(9))
        sig = f"__device__ __forceinline__ fe {name}(const fe& a) {{\n  fe r;\n  i32 d0, d1, d2, d3, d4, d5, d6, d7, d8;\n"
    else:
        ins = ", ".join(f'"v"(a.v[{j}])' for j in range(9)) + ", " + ", ".join(f'"v"(b.v[{j}])' for j in range(9))
        sig = f"__device__ __forceinline__ fe {name}(const fe& a, const fe& b) {{\n  fe r;\n"
    clob = '"vcc", "v0", "v1", ' + ", ".join(f'"s{SP + j}"' for j in range(10))
    if HOIST:
        ins += ", " + ", ".join(f'"s"(K.k[{j}])' for j in range(10))
        clob = '"vcc", "v0", "v1"'
        sig += "  const fe_sk K = fe_consts();\n"
    tail = "  (void)d0; (void)d1; (void)d2; (void)d3; (void)d4; (void)d5; (void)d6; (void)d7; (void)d8;\n" if square else ""
    return f"{sig}  asm(\n{body}\n      : {outs}\n      : {ins}\n      : {clob});\n{tail}  return r;\n}}\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "crypto_amd", "csrc", "fe_mul_gen.inc")
    with open(dst, "w") as f:
        f.write("// GENERATED by tools/gen_fe_mul.py -- do not edit.\n")
        if HOIST:
            f.write(consts_fn())
        f.write(function("fe_mul", False))
        f.write("\n")
        f.write(function("fe_sqr", True))
        f.write("\n")
        f.write(function2("fe_mul2add_asm"))
    print("wrote", dst)

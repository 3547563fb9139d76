// Host-side mirror of the reference's Bulletproofs package (go/bp) on top of the C ABI of
// libbn256msm (include/bn256msm.h).  Same names, argument meaning and error behaviour as
// go/bp/main.go and go/bp/util.go; the Go toolchain is absent from this image, so the host
// layer is C++ (the reference is compiled code).  Every vector commitment goes through the
// device library; generators stay device-resident across the inner-product rounds.
//
// This is product code; it shares nothing with oracle/.
#include <cstdint>
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <errno.h>
#include <sys/random.h>

#include "../../include/bn256msm.h"
#include "../../include/bp_amd.h"
#include "host_g1.h"

namespace {

typedef unsigned __int128 u128;

// ------------------------------------------------------------------ Fr = integers mod bn256.Order
// big.Int + Mod(Order) in the reference (go/bp/util.go:197-207); here 4x64 Montgomery.
struct Fr {
  uint64_t l[4];  // Montgomery form
};
const uint64_t kN[4] = {0x1a2ef45b57ac7261ULL, 0x2e8d8e12f82b3924ULL, 0xaa6fecb86184dc21ULL, 0x8fb501e34aa387f9ULL};
const Fr kFrR2 = {{0xb5f030132affbc35ULL, 0x85a1f7da0792e95dULL, 0x26841e5fa6ee4895ULL, 0x3d8f6c73765aefd5ULL}};
const Fr kFrOne = {{0xe5d10ba4a8538d9fULL, 0xd17271ed07d4c6dbULL, 0x559013479e7b23deULL, 0x704afe1cb55c7806ULL}};
const uint64_t kFrNInv = 0x056417b72d284e5fULL;

bool geq_n(const uint64_t a[4]) {
  for (int i = 3; i >= 0; i--)
    if (a[i] != kN[i]) return a[i] > kN[i];
  return true;
}
void sub_n(uint64_t a[4]) {
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a[i] - kN[i] - br;
    a[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
}
Fr add(const Fr& a, const Fr& b) {
  Fr r;
  u128 c = 0;
  for (int i = 0; i < 4; i++) {
    c += (u128)a.l[i] + b.l[i];
    r.l[i] = (uint64_t)c;
    c >>= 64;
  }
  if ((uint64_t)c || geq_n(r.l)) sub_n(r.l);
  return r;
}
Fr sub(const Fr& a, const Fr& b) {  // Go's Mod is Euclidean: the result is always in [0, n)
  Fr r;
  u128 br = 0;
  for (int i = 0; i < 4; i++) {
    u128 d = (u128)a.l[i] - b.l[i] - br;
    r.l[i] = (uint64_t)d;
    br = (d >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (u128)r.l[i] + kN[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
  }
  return r;
}
Fr mul(const Fr& a, const Fr& b) {
  uint64_t t[9] = {0};
  for (int i = 0; i < 4; i++) {
    uint64_t c = 0;
    for (int j = 0; j < 4; j++) {
      u128 cur = (u128)a.l[i] * b.l[j] + t[i + j] + c;
      t[i + j] = (uint64_t)cur;
      c = (uint64_t)(cur >> 64);
    }
    t[i + 4] = c;
  }
  for (int i = 0; i < 4; i++) {
    uint64_t m = t[i] * kFrNInv;
    uint64_t c = 0;
    for (int j = 0; j < 4; j++) {
      u128 cur = (u128)m * kN[j] + t[i + j] + c;
      t[i + j] = (uint64_t)cur;
      c = (uint64_t)(cur >> 64);
    }
    for (int k = i + 4; c && k < 9; k++) {
      u128 cur = (u128)t[k] + c;
      t[k] = (uint64_t)cur;
      c = (uint64_t)(cur >> 64);
    }
  }
  Fr r = {{t[4], t[5], t[6], t[7]}};
  if (t[8] || geq_n(r.l)) sub_n(r.l);
  return r;
}
Fr fr_from_raw(const uint64_t w[4]) {  // any 256-bit value, reduced mod n
  uint64_t v[4] = {w[0], w[1], w[2], w[3]};
  if (geq_n(v)) sub_n(v);  // 2^256 < 2n
  Fr t = {{v[0], v[1], v[2], v[3]}};
  return mul(t, kFrR2);
}
Fr fr_from_u64(uint64_t x) {
  uint64_t w[4] = {x, 0, 0, 0};
  return fr_from_raw(w);
}
void fr_to_raw(uint64_t w[4], const Fr& a) {
  Fr one = {{1, 0, 0, 0}};
  Fr t = mul(a, one);
  memcpy(w, t.l, 32);
}
Fr fr_from_be(const uint8_t b[32]) {
  uint64_t w[4];
  hostg1::be_load(w, b);
  return fr_from_raw(w);
}
void fr_to_be(uint8_t b[32], const Fr& a) {  // scalarTo32Byte, go/bp/util.go:209-220
  uint64_t w[4];
  fr_to_raw(w, a);
  hostg1::be_store(b, w);
}
Fr fr_inv(const Fr& a) {  // ModInverse(a, Order) via a^(n-2)
  uint64_t e[4] = {kN[0] - 2, kN[1], kN[2], kN[3]};
  Fr acc = kFrOne;
  for (int i = 255; i >= 0; i--) {
    acc = mul(acc, acc);
    if ((e[i >> 6] >> (i & 63)) & 1) acc = mul(acc, a);
  }
  return acc;
}
const Fr kFrZero = {{0, 0, 0, 0}};

typedef std::vector<Fr> FrVec;

// go/bp/util.go:29-49, 106-163
FrVec ntharr(const Fr& val, int n) {
  FrVec r(n);
  r[0] = kFrOne;
  for (int i = 1; i < n; i++) r[i] = mul(r[i - 1], val);
  return r;
}
FrVec invntharr(const Fr& val, int n) { return ntharr(fr_inv(val), n); }
FrVec ones(int n) { return FrVec(n, kFrOne); }
FrVec vectorAdd(const FrVec& a, const FrVec& b) {
  if (a.size() != b.size()) throw std::runtime_error("invalid length");
  FrVec r(a.size());
  for (size_t i = 0; i < a.size(); i++) r[i] = add(a[i], b[i]);
  return r;
}
FrVec vectorSub(const FrVec& a, const FrVec& b) {
  if (a.size() != b.size()) throw std::runtime_error("invalid length");
  FrVec r(a.size());
  for (size_t i = 0; i < a.size(); i++) r[i] = sub(a[i], b[i]);
  return r;
}
FrVec vectorMulOnScalar(const FrVec& a, const Fr& c) {
  FrVec r(a.size());
  for (size_t i = 0; i < a.size(); i++) r[i] = mul(a[i], c);
  return r;
}
Fr vectorMul(const FrVec& a, const FrVec& b) {
  if (a.size() != b.size()) throw std::runtime_error("invalid length");
  Fr r = kFrZero;
  for (size_t i = 0; i < a.size(); i++) r = add(r, mul(a[i], b[i]));
  return r;
}
FrVec hadamardMul(const FrVec& a, const FrVec& b) {
  if (a.size() != b.size()) throw std::runtime_error("invalid length");
  FrVec r(a.size());
  for (size_t i = 0; i < a.size(); i++) r[i] = mul(a[i], b[i]);
  return r;
}
FrVec slice(const FrVec& a, size_t lo, size_t hi) { return FrVec(a.begin() + lo, a.begin() + hi); }

// ------------------------------------------------------------------ Keccak-256 (legacy padding)
// iden3 keccak256.Hash as used by hash(), go/bp/util.go:222-233.
uint64_t rol64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
void keccak_f1600(uint64_t s[25]) {
  static const uint64_t RC[24] = {
      0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL, 0x000000000000808BULL,
      0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL, 0x000000000000008AULL, 0x0000000000000088ULL,
      0x0000000080008009ULL, 0x000000008000000AULL, 0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL,
      0x8000000000008003ULL, 0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
      0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int piln[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
  static const int rotc[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
  for (int round = 0; round < 24; round++) {
    uint64_t bc[5];
    for (int i = 0; i < 5; i++) bc[i] = s[i] ^ s[i + 5] ^ s[i + 10] ^ s[i + 15] ^ s[i + 20];
    for (int i = 0; i < 5; i++) {
      uint64_t t = bc[(i + 4) % 5] ^ rol64(bc[(i + 1) % 5], 1);
      for (int j = 0; j < 25; j += 5) s[j + i] ^= t;
    }
    uint64_t t = s[1];
    for (int i = 0; i < 24; i++) {
      int j = piln[i];
      uint64_t b = s[j];
      s[j] = rol64(t, rotc[i]);
      t = b;
    }
    for (int j = 0; j < 25; j += 5) {
      for (int i = 0; i < 5; i++) bc[i] = s[j + i];
      for (int i = 0; i < 5; i++) s[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
    }
    s[0] ^= RC[round];
  }
}
void keccak256(uint8_t out[32], const std::vector<uint8_t>& in) {
  uint64_t s[25] = {0};
  std::vector<uint8_t> m(in);
  m.push_back(0x01);
  while (m.size() % 136) m.push_back(0);
  m.back() |= 0x80;
  for (size_t off = 0; off < m.size(); off += 136) {
    for (int i = 0; i < 17; i++) {
      uint64_t w = 0;
      for (int k = 7; k >= 0; k--) w = (w << 8) | m[off + 8 * i + k];
      s[i] ^= w;
    }
    keccak_f1600(s);
  }
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(s[i] >> (8 * k));
}

// ------------------------------------------------------------------ points on the host
typedef std::vector<uint8_t> Bytes;  // a marshalled G1 point (64 bytes)

struct Point {
  uint8_t b[64];
};
Point pt_from(const hostg1::Pt& p) {
  Point r;
  hostg1::marshal(r.b, p);
  return r;
}
hostg1::Pt pt_load(const Point& p) {
  hostg1::Pt r;
  if (hostg1::unmarshal(r, p.b)) throw std::runtime_error("bn256: malformed point");
  return r;
}
// new(bn256.G1).ScalarMult(p, k) for single points on the host (com() stays on the CPU, SURVEY.md 8a
// row a5).  k = k1 + k2*lambda (bn256_glv_split), lambda*(x, y) = (beta*x, y): one joint
// double-and-add over +-P, +-phi(P) and their sum, 129 doublings instead of 256.
Point scalarMult(const Point& p, const Fr& k) {
  static const hostg1::Fp beta = [] {
    const uint64_t w[4] = {0xb9ff3815a6c2a92eULL, 0x877db38fb92fa161ULL, 0xcb5601bf83f4fcf8ULL, 0};
    return hostg1::from_raw(w);
  }();
  uint8_t kb[32], k1b[32], k2b[32];
  int n1 = 0, n2 = 0;
  fr_to_be(kb, k);
  if (int rc = bn256_glv_split(kb, k1b, &n1, k2b, &n2)) throw std::runtime_error(bn256_strerror(rc));
  uint64_t w1[4], w2[4];
  hostg1::be_load(w1, k1b);
  hostg1::be_load(w2, k2b);
  hostg1::Pt A = pt_load(p), acc = hostg1::infinity();
  if (hostg1::is_inf(A)) return p;
  hostg1::Pt B = A;
  B.x = hostg1::mul(A.x, beta);
  const hostg1::Fp zero = {{0, 0, 0, 0}};
  if (n1) A.y = hostg1::sub(zero, A.y);
  if (n2) B.y = hostg1::sub(zero, B.y);
  const hostg1::Pt AB = hostg1::padd(A, B);
  for (int i = 129; i >= 0; i--) {
    acc = hostg1::dbl(acc);
    const unsigned b1 = (w1[i >> 6] >> (i & 63)) & 1, b2 = (w2[i >> 6] >> (i & 63)) & 1;
    if (b1 | b2) acc = hostg1::padd(acc, (b1 & b2) ? AB : (b1 ? A : B));
  }
  return pt_from(acc);
}
Point pointAdd(const Point& a, const Point& b) { return pt_from(hostg1::padd(pt_load(a), pt_load(b))); }
bool pointEq(const Point& a, const Point& b) { return memcmp(a.b, b.b, 64) == 0; }  // bytes.Equal(Marshal, Marshal)
// com(), go/bp/util.go:92-94
Point com(const Point& g, const Point& h, const Fr& a, const Fr& b) { return pointAdd(scalarMult(g, a), scalarMult(h, b)); }

// hash(), go/bp/util.go:222-233
Fr hashFS(const FrVec& scalars, const std::vector<Point>& points) {
  std::vector<uint8_t> data;
  for (const Fr& s : scalars) {
    uint8_t b[32];
    fr_to_be(b, s);
    data.insert(data.end(), b, b + 32);
  }
  for (const Point& p : points) data.insert(data.end(), p.b, p.b + 64);
  uint8_t d[32];
  keccak256(d, data);
  return fr_from_be(d);
}

// ------------------------------------------------------------------ crypto/rand stand-in
// rand.Reader replacement for deterministic transcripts: SplitMix64, 8 bytes
// big-endian per step; rand.Int = read 32 bytes, accept iff < Order.
// With os_random the 8 bytes come from the operating system's CSPRNG instead (getrandom(2)): what
// crypto/rand.Reader is in the reference (go/bp/util.go:51-63) -- the product entry points bp_setup /
// bp_prove use that; the seeded stream exists for reproducible transcripts in the tests only.
struct ByteStream {
  uint64_t s;
  bool os_random = false;
  uint64_t next64() {
    if (os_random) {
      uint8_t b[8];
      size_t got = 0;
      while (got < sizeof b) {
        ssize_t r = getrandom(b + got, sizeof b - got, 0);
        if (r < 0) {
          if (errno == EINTR) continue;
          throw std::runtime_error("getrandom failed");
        }
        got += (size_t)r;
      }
      uint64_t z = 0;
      for (int i = 0; i < 8; i++) z = (z << 8) | b[i];
      return z;
    }
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
  }
  Fr randInt(uint8_t raw_be[32] = nullptr) {
    for (;;) {
      uint64_t w[4];
      for (int i = 3; i >= 0; i--) w[i] = next64();  // big-endian stream: first word is the most significant
      if (!geq_n(w)) {
        if (raw_be) hostg1::be_store(raw_be, w);
        return fr_from_raw(w);
      }
    }
  }
  FrVec values(int n) {  // go/bp/util.go:51-63
    FrVec r(n);
    for (int i = 0; i < n; i++) r[i] = randInt();
    return r;
  }
  Fr randNonZero(uint8_t raw_be[32]) {  // the k of bn256.RandomG1
    for (;;) {
      Fr k = randInt(raw_be);
      bool z = true;
      for (int i = 0; i < 32; i++) z = z && raw_be[i] == 0;
      if (!z) return k;
    }
  }
};

// ------------------------------------------------------------------ device vectors
void chk(int rc) {
  if (rc) throw std::runtime_error(bn256_strerror(rc));
}
struct DevPoints {  // a []*bn256.G1 resident in HBM
  bn256_ctx* ctx = nullptr;
  bn256_bases* h = nullptr;
  size_t n = 0;
  bool owned = true;  // false: someone else's handle (the caller's public generators), never freed here
  DevPoints() {}
  DevPoints(bn256_ctx* c, bn256_bases* b) : ctx(c), h(b), n(bn256_bases_len(b)) {}
  // A protocol never writes into the generators it is given -- a fold makes a NEW handle and the old one is
  // dropped -- so the first level can work on the caller's handle as it stands (a clone was a download and an
  // upload, two waits, per proof).
  static DevPoints borrow(bn256_ctx* c, const bn256_bases* b) {
    DevPoints d(c, const_cast<bn256_bases*>(b));
    d.owned = false;
    return d;
  }
  DevPoints(const DevPoints&) = delete;
  DevPoints& operator=(const DevPoints&) = delete;
  DevPoints(DevPoints&& o) noexcept : ctx(o.ctx), h(o.h), n(o.n), owned(o.owned) { o.h = nullptr; }
  DevPoints& operator=(DevPoints&& o) noexcept {
    if (this != &o) {
      if (h && owned) bn256_bases_free(ctx, h);
      ctx = o.ctx; h = o.h; n = o.n; owned = o.owned; o.h = nullptr;
    }
    return *this;
  }
  ~DevPoints() {
    if (h && owned) bn256_bases_free(ctx, h);
  }
  Point at(size_t i) const {
    Point p;
    chk(bn256_bases_download(ctx, h, i, 1, p.b));
    return p;
  }
};
std::vector<uint8_t> scalarBytes(const FrVec& a) {
  std::vector<uint8_t> s(32 * a.size());
  for (size_t i = 0; i < a.size(); i++) fr_to_be(&s[32 * i], a[i]);
  return s;
}
struct Seg {
  const DevPoints* g;
  size_t off;
  FrVec a;
};
// sum of vectorPointScalarMul over segments (go/bp/util.go:165-175): one bn256_msm_multi
struct SegArgs {
  std::vector<const bn256_bases*> hs;
  std::vector<size_t> offs, ns;
  std::vector<std::vector<uint8_t>> raws;
  std::vector<const uint8_t*> ptrs;
  explicit SegArgs(const std::vector<Seg>& segs) {
    for (const Seg& s : segs) {
      if (s.off + s.a.size() > s.g->n) throw std::runtime_error("invalid length for scalar mul");
      hs.push_back(s.g->h);
      offs.push_back(s.off);
      ns.push_back(s.a.size());
      raws.push_back(scalarBytes(s.a));
    }
    for (auto& r : raws) ptrs.push_back(r.data());
  }
};
Point msmSegments(bn256_ctx* ctx, const std::vector<Seg>& segs) {
  SegArgs a(segs);
  Point out;
  chk(bn256_msm_multi(ctx, (int)segs.size(), a.hs.data(), a.offs.data(), a.ns.data(), a.ptrs.data(), out.b));
  return out;
}
// the same on one of the library's lanes (independent commitments of one step); msmWait returns the point
int msmSegmentsSubmit(bn256_ctx* ctx, const std::vector<Seg>& segs) {
  SegArgs a(segs);
  int ticket = -1;
  chk(bn256_msm_multi_submit(ctx, (int)segs.size(), a.hs.data(), a.offs.data(), a.ns.data(), a.ptrs.data(), &ticket));
  return ticket;
}
Point msmWait(bn256_ctx* ctx, int ticket) {
  Point out;
  chk(bn256_msm_wait(ctx, ticket, out.b));
  return out;
}
// vectorPointScalarMul, go/bp/util.go:165-175 (every call site of the mirror now sums several of them in one
// msmSegments; the single form stays for a caller that wants the reference's helper by name)
[[maybe_unused]] Point vectorPointScalarMul(bn256_ctx* ctx, const DevPoints& g, size_t off, size_t len, const FrVec& a) {
  if (len != a.size()) throw std::runtime_error("invalid length for scalar mul");
  return msmSegments(ctx, {{&g, off, a}});
}

// ------------------------------------------------------------------ go/bp/main.go
struct InnerArgumentPublic {  // go/bp/main.go:38-45
  int N;
  DevPoints G, H;
  Point U;
};
struct InnerProductProof {  // go/bp/main.go:253-260
  std::vector<Point> L, R;
  Fr A, B;
};

// Small levels of the inner-product argument run over generators that are no longer folded (see wnla below and
// bn256_fr_tensor_expand_multi): after t levels G_t[i] = sum_j W(j) G_f[i + n_t j], W the tensor of the levels'
// (x^-1, x) -- (x, x^-1) for H -- so every commitment of a later level is ONE MSM over the frozen G_f, H_f with
// expanded scalars, and the two ~0.45 ms generator-fold chains per level (main.go:355-356) disappear: from the
// prover's critical path, and from the verifier altogether (it needs only the final G[0], H[0]: the classic
// s-vector MSM).  Same group elements, same transcript.  BP_IPA_FREEZE=<n> moves the threshold (0: always fold).
struct DevFr;
struct IpaFrozen;
static Point fixedMult(const Point& g, const Fr& k);  // k*g over a cached table of g (below): U is multiplied 3 log N times
static size_t ipaFreeze() {
  static const size_t v = getenv("BP_IPA_FREEZE") ? (size_t)strtoull(getenv("BP_IPA_FREEZE"), nullptr, 10) : 4096;
  return v;
}
InnerProductProof innerArgProofFrozen(bn256_ctx* ctx, int N, DevPoints G, DevPoints H, const Point& U, FrVec a, FrVec b,
                                      InnerProductProof proof);
std::string innerArgVerifyFrozen(bn256_ctx* ctx, int N, DevPoints G, DevPoints H, const Point& U, Point P,
                                 const InnerProductProof& proof, size_t deep);

// innerArgProof, go/bp/main.go:321-377 (recursion unrolled into a loop; G, H stay on device)
InnerProductProof innerArgProof(bn256_ctx* ctx, int N, DevPoints G, DevPoints H, const Point& U, FrVec a, FrVec b) {
  InnerProductProof proof;
  while (N != 1) {
    if (N % 2 != 0) throw std::runtime_error("invalid n: should be 2^x");
    if (ipaFreeze() && (size_t)N <= ipaFreeze() && (N & (N - 1)) == 0)
      return innerArgProofFrozen(ctx, N, std::move(G), std::move(H), U, std::move(a), std::move(b), std::move(proof));
    size_t n1 = N / 2;
    FrVec alo = slice(a, 0, n1), ahi = slice(a, n1, N), blo = slice(b, 0, n1), bhi = slice(b, n1, N);
    Fr cl = vectorMul(alo, bhi), cr = vectorMul(ahi, blo);
    // L = vecCom(G[n1:], H[:n1], a[:n1], b[n1:]) + cl*U ;  R = vecCom(G[:n1], H[n1:], a[n1:], b[:n1]) + cr*U
    // P = productCom(G, H, U, a, b), go/bp/util.go:100-104 (recomputed at every level, main.go:347)
    // The three MSMs are independent: one lane each, the host's c*U terms under them.
    int tl = msmSegmentsSubmit(ctx, {{&G, n1, alo}, {&H, 0, bhi}});
    int tr = msmSegmentsSubmit(ctx, {{&G, 0, ahi}, {&H, n1, blo}});
    int tp = msmSegmentsSubmit(ctx, {{&G, 0, a}, {&H, 0, b}});
    Point ul = fixedMult(U, cl), ur = fixedMult(U, cr), up = fixedMult(U, vectorMul(a, b));
    Point L = pointAdd(msmWait(ctx, tl), ul);
    Point R = pointAdd(msmWait(ctx, tr), ur);
    Point P = pointAdd(msmWait(ctx, tp), up);
    proof.L.push_back(L);
    proof.R.push_back(R);
    Fr x = hashFS({fr_from_u64((uint64_t)N)}, {P, L, R});
    Fr xinv = fr_inv(x);
    // g1 = xinv*G[:n1] (+) x*G[n1:] ; h1 = x*H[:n1] (+) xinv*H[n1:]   (main.go:355-356), fused on device
    uint8_t xb[32], xib[32];
    fr_to_be(xb, x);
    fr_to_be(xib, xinv);
    // the two folds are independent latency chains: G's on a lane, H's on the context's stream, the
    // host's a', b' under both
    bn256_bases *g1 = nullptr, *h1 = nullptr;
    int tg = -1;
    chk(bn256_fold2_submit(ctx, G.h, 0, n1, G.h, n1, n1, 1, xib, xb, &g1, &tg));
    DevPoints G1(ctx, g1);
    chk(bn256_fold(ctx, H.h, 0, H.h, n1, 1, n1, xb, xib, &h1));
    DevPoints H1(ctx, h1);
    FrVec a1 = vectorAdd(vectorMulOnScalar(alo, x), vectorMulOnScalar(ahi, xinv));
    FrVec b1 = vectorAdd(vectorMulOnScalar(bhi, x), vectorMulOnScalar(blo, xinv));
    chk(bn256_fold_wait(ctx, tg));
    G = std::move(G1);
    H = std::move(H1);
    a = a1;
    b = b1;
    N = (int)n1;
  }
  proof.A = a[0];
  proof.B = b[0];
  return proof;
}

// innerArgVerify, go/bp/main.go:273-319.  Returns "" on success or the reference's error text.
std::string innerArgVerify(bn256_ctx* ctx, int N, DevPoints G, DevPoints H, const Point& U, Point P,
                           const InnerProductProof& proof) {
  size_t deep = 0;
  while (N != 1) {
    if (N % 2 != 0) return "invalid n: should be 2^x";
    if (deep >= proof.L.size() || deep >= proof.R.size()) return "failed to verify inner argument proof";
    if (ipaFreeze() && (size_t)N <= ipaFreeze() && (N & (N - 1)) == 0)
      return innerArgVerifyFrozen(ctx, N, std::move(G), std::move(H), U, P, proof, deep);
    size_t n1 = N / 2;
    Fr x = hashFS({fr_from_u64((uint64_t)N)}, {P, proof.L[deep], proof.R[deep]});
    Fr xinv = fr_inv(x), x2 = mul(x, x), x2inv = fr_inv(x2);
    uint8_t xb[32], xib[32];
    fr_to_be(xb, x);
    fr_to_be(xib, xinv);
    bn256_bases *g1 = nullptr, *h1 = nullptr;
    int tg = -1;
    chk(bn256_fold2_submit(ctx, G.h, 0, n1, G.h, n1, n1, 1, xib, xb, &g1, &tg));  // as in innerArgProof
    DevPoints G1(ctx, g1);
    chk(bn256_fold(ctx, H.h, 0, H.h, n1, 1, n1, xb, xib, &h1));
    DevPoints H1(ctx, h1);
    P = pointAdd(pointAdd(scalarMult(proof.L[deep], x2), P), scalarMult(proof.R[deep], x2inv));
    chk(bn256_fold_wait(ctx, tg));
    G = std::move(G1);
    H = std::move(H1);
    N = (int)n1;
    deep++;
  }
  Point p = pointAdd(com(G.at(0), H.at(0), proof.A, proof.B), scalarMult(U, mul(proof.A, proof.B)));
  if (!pointEq(P, p)) return "failed to verify inner argument proof";
  return "";
}

// k * g for ONE point g and many k (wnla's v_x g and v_r g, two per round, docs_test.go:1312,1318): signed-free
// 4-bit windows over the two GLV halves, table d * 16^w * g for w < 33, d < 16 (495 additions to build, once),
// then 66 additions per product instead of scalarMult's 130 doublings + ~97 additions.
class FixedBase {
  std::vector<hostg1::Pt> T;  // [w][d - 1]
  bool inf = false;

 public:
  explicit FixedBase(const Point& g) {
    hostg1::Pt b = pt_load(g);
    inf = hostg1::is_inf(b);
    if (inf) return;
    T.resize(33 * 15);
    for (int w = 0; w < 33; w++) {
      T[15 * w] = b;
      for (int d = 2; d <= 15; d++) T[15 * w + d - 1] = hostg1::padd(T[15 * w + d - 2], b);
      b = hostg1::padd(T[15 * w + 14], b);  // 16 * b
    }
  }
  Point mul(const Fr& k) const {
    if (inf) return pt_from(hostg1::infinity());
    static const hostg1::Fp beta = [] {
      const uint64_t w[4] = {0xb9ff3815a6c2a92eULL, 0x877db38fb92fa161ULL, 0xcb5601bf83f4fcf8ULL, 0};
      return hostg1::from_raw(w);
    }();
    uint8_t kb[32], k1b[32], k2b[32];
    int n1 = 0, n2 = 0;
    fr_to_be(kb, k);
    if (int rc = bn256_glv_split(kb, k1b, &n1, k2b, &n2)) throw std::runtime_error(bn256_strerror(rc));
    const hostg1::Fp zero = {{0, 0, 0, 0}};
    hostg1::Pt acc = hostg1::infinity();
    for (int h = 0; h < 2; h++) {
      const uint8_t* kk = h ? k2b : k1b;  // big-endian magnitude below 2^130
      const bool neg = h ? n2 != 0 : n1 != 0;
      for (int w = 0; w < 33; w++) {
        const int byte = 31 - w / 2;
        const unsigned d = (w & 1) ? (kk[byte] >> 4) : (kk[byte] & 15);
        if (!d) continue;
        hostg1::Pt t = T[15 * w + d - 1];
        if (h) t.x = hostg1::mul(t.x, beta);
        if (neg) t.y = hostg1::sub(zero, t.y);
        acc = hostg1::padd(acc, t);
      }
    }
    return pt_from(acc);
  }
};

// The commitment bases g, h and U of a BulletProofPublic are public parameters that every proof of a set-up
// multiplies again (com() x 3, h*alpha, h*rho, the verifier's g*delta, h*(-nu), U*tx: go/bp/main.go:92-123,
// 209-245): their FixedBase tables are kept, keyed by the 64 point bytes, for the process's lifetime (a handful
// of entries; the oldest goes when there are more than 16).
static std::shared_ptr<const FixedBase> fixedBaseFor(const Point& g) {
  static std::mutex mu;
  static std::deque<std::pair<std::array<uint8_t, 64>, std::shared_ptr<const FixedBase>>> cache;
  std::array<uint8_t, 64> key;
  memcpy(key.data(), g.b, 64);
  std::lock_guard<std::mutex> lk(mu);
  for (auto& e : cache)
    if (e.first == key) return e.second;
  auto t = std::make_shared<const FixedBase>(g);
  cache.emplace_back(key, t);
  if (cache.size() > 16) cache.pop_front();
  return t;
}
static Point fixedMult(const Point& g, const Fr& k) { return fixedBaseFor(g)->mul(k); }
static Point fixedCom(const Point& g, const Point& h, const Fr& a, const Fr& b) {  // com(), go/bp/util.go:92-94
  return pointAdd(fixedMult(g, a), fixedMult(h, b));
}

struct BulletProofPublic {  // go/bp/main.go:17-24
  int N;
  DevPoints G, H;  // InnerArgumentPublic.G / .H
  Point U;
  Point g, h;  // commitment bases
};
struct BulletProof {  // go/bp/main.go:59-78
  Point V, ACom, SCom, T1Com, T2Com;
  Fr Tx, TauX, Nu;
  std::vector<Point> L, R;
  Fr A, B;
};

// points(n), go/bp/util.go:81-90, with RandomG1 = k*G0: the k are drawn on the host from the
// injected reader, the n base multiplications run on the device.
DevPoints points(bn256_ctx* ctx, ByteStream& rd, int n) {
  std::vector<uint8_t> ks(32 * (size_t)n), g0(64 * (size_t)n, 0);
  uint64_t y[4] = {hostg1::kP[0] - 2, hostg1::kP[1], hostg1::kP[2], hostg1::kP[3]};
  for (int i = 0; i < n; i++) {
    rd.randNonZero(&ks[32 * (size_t)i]);
    g0[64 * (size_t)i + 31] = 1;
    hostg1::be_store(&g0[64 * (size_t)i + 32], y);
  }
  bn256_bases *gen = nullptr, *out = nullptr;
  chk(bn256_bases_upload(ctx, g0.data(), (size_t)n, 0, &gen));
  DevPoints G0s(ctx, gen);
  chk(bn256_scalar_mul_batch(ctx, gen, 0, (size_t)n, ks.data(), 0, &out));
  return DevPoints(ctx, out);
}

// NewBulletProofPublic, go/bp/main.go:27-34 (draw order: G[0..n), H[0..n), U, g, h).  The five
// points() calls of the reference draw 2n + 3 scalars one after the other, so one batch of 2n + 3
// base multiplications yields the same points: a single latency chain on the device instead of five.
BulletProofPublic newBulletProofPublic(bn256_ctx* ctx, ByteStream& rd, int n) {
  BulletProofPublic p;
  p.N = n;
  DevPoints all = points(ctx, rd, 2 * n + 3);
  std::vector<uint8_t> raw(64 * all.n);
  chk(bn256_bases_download(ctx, all.h, 0, all.n, raw.data()));
  bn256_bases *g = nullptr, *h = nullptr;
  chk(bn256_bases_upload(ctx, raw.data(), (size_t)n, 0, &g));
  p.G = DevPoints(ctx, g);
  chk(bn256_bases_upload(ctx, raw.data() + 64 * (size_t)n, (size_t)n, 0, &h));
  p.H = DevPoints(ctx, h);
  memcpy(p.U.b, &raw[64 * (size_t)(2 * n)], 64);
  memcpy(p.g.b, &raw[64 * (size_t)(2 * n + 1)], 64);
  memcpy(p.h.b, &raw[64 * (size_t)(2 * n + 2)], 64);
  return p;
}

FrVec toBits(const Fr& v, int n) {  // go/bp/util.go:235-251
  uint64_t w[4];
  fr_to_raw(w, v);
  FrVec r(n, kFrZero);
  for (int i = 0; i < n && i < 256; i++)
    if ((w[i >> 6] >> (i & 63)) & 1) r[i] = kFrOne;
  return r;
}

DevPoints cloneRange(bn256_ctx* ctx, const DevPoints& g) {  // a second handle on the same points
  std::vector<uint8_t> raw(64 * g.n);
  chk(bn256_bases_download(ctx, g.h, 0, g.n, raw.data()));
  bn256_bases* h = nullptr;
  chk(bn256_bases_upload(ctx, raw.data(), g.n, 0, &h));
  return DevPoints(ctx, h);
}

// (*BulletProofPublic).Prove, go/bp/main.go:81-178.  Randomness order: alpha, sl, sr, ro, tau1, tau2.
// a handle a helper thread is producing: collected by take(), or freed if the caller leaves early
struct PendingBases {
  std::future<bn256_bases*> f;
  bn256_ctx* c;
  bn256_bases* take() { return f.get(); }
  ~PendingBases() {
    if (!f.valid()) return;
    try {
      if (bn256_bases* r = f.get()) bn256_bases_free(c, r);
    } catch (...) {
    }
  }
};
BulletProof prove(bn256_ctx* ctx, const BulletProofPublic& p, const Fr& v, const Fr& prv, ByteStream& rd) {
  const int N = p.N;
  FrVec onen = ones(N);
  FrVec al = toBits(v, N);
  FrVec ar = vectorSub(al, onen);
  Fr alpha = rd.values(1)[0];
  int ta = msmSegmentsSubmit(ctx, {{&p.G, 0, al}, {&p.H, 0, ar}});  // A and S: independent, a lane each
  FrVec sl = rd.values(N), sr = rd.values(N);
  Fr ro = rd.values(1)[0];
  int ts = msmSegmentsSubmit(ctx, {{&p.G, 0, sl}, {&p.H, 0, sr}});
  Point ha = fixedMult(p.h, alpha), hs = fixedMult(p.h, ro);
  Point V = fixedCom(p.g, p.h, v, prv);
  Point A = pointAdd(msmWait(ctx, ta), ha);
  Point S = pointAdd(msmWait(ctx, ts), hs);
  BulletProof proof;
  proof.V = V;
  proof.ACom = A;
  proof.SCom = S;
  Fr y = hashFS({fr_from_u64((uint64_t)N)}, {A, S, V});
  // h1[i] = H[i] * yinvn[i]   (main.go:156-159), batched on device.  It needs only y: the call (one latency chain of
  // ~0.5 ms at N = 64) runs on a helper thread while this one does the host algebra down to nu, which makes no
  // library call (the context is used by one thread at a time).
  FrVec yinvn = invntharr(y, N);
  std::vector<uint8_t> yb = scalarBytes(yinvn);
  PendingBases h1f{std::async(std::launch::async, [ctx, &p, &yb, N]() {
                     bn256_bases* r = nullptr;
                     chk(bn256_scalar_mul_batch(ctx, p.H.h, 0, (size_t)N, yb.data(), 0, &r));
                     return r;
                   }), ctx};
  Fr z = hashFS({y}, {A, S});
  FrVec yn = ntharr(y, N);
  Fr z2 = mul(z, z);
  FrVec twon = ntharr(fr_from_u64(2), N);
  Fr t1 = add(vectorMul(hadamardMul(yn, sr), vectorSub(al, vectorMulOnScalar(onen, z))),
              vectorMul(sl, vectorAdd(vectorMulOnScalar(twon, z2),
                                      hadamardMul(yn, vectorAdd(ar, vectorMulOnScalar(onen, z))))));
  Fr t2 = vectorMul(hadamardMul(yn, sr), sl);
  Fr tau1 = rd.values(1)[0], tau2 = rd.values(1)[0];
  proof.T1Com = fixedCom(p.g, p.h, t1, tau1);
  proof.T2Com = fixedCom(p.g, p.h, t2, tau2);
  Fr x = hashFS({y, z}, {proof.T1Com, proof.T2Com});
  Fr x2 = mul(x, x);
  FrVec l = vectorAdd(vectorSub(al, vectorMulOnScalar(onen, z)), vectorMulOnScalar(sl, x));
  FrVec r = vectorAdd(
      hadamardMul(yn, vectorAdd(ar, vectorAdd(vectorMulOnScalar(onen, z), vectorMulOnScalar(sr, x)))),
      vectorMulOnScalar(twon, z2));
  proof.Tx = vectorMul(l, r);
  proof.TauX = add(mul(tau2, x2), add(mul(tau1, x), mul(z2, prv)));
  proof.Nu = add(alpha, mul(ro, x));
  bn256_bases* h1 = h1f.take();
  InnerProductProof ip = innerArgProof(ctx, N, DevPoints::borrow(ctx, p.G.h), DevPoints(ctx, h1), p.U, l, r);
  proof.A = ip.A;
  proof.B = ip.B;
  proof.L = ip.L;
  proof.R = ip.R;
  return proof;
}

// (*BulletProofPublic).Verify, go/bp/main.go:181-249
std::string verify(bn256_ctx* ctx, const BulletProofPublic& p, const BulletProof& proof) {
  const int N = p.N;
  Fr y = hashFS({fr_from_u64((uint64_t)N)}, {proof.ACom, proof.SCom, proof.V});
  Fr z = hashFS({y}, {proof.ACom, proof.SCom});
  FrVec yn = ntharr(y, N);
  Fr z2 = mul(z, z), z3 = mul(z2, z);
  FrVec onen = ones(N), twon = ntharr(fr_from_u64(2), N);
  Fr x = hashFS({y, z}, {proof.T1Com, proof.T2Com});
  Fr x2 = mul(x, x);
  FrVec yinvn = invntharr(y, N);
  std::vector<uint8_t> yb = scalarBytes(yinvn);
  // h1 on a helper thread under the host's tx check (as in prove: no library call until it is collected)
  PendingBases h1f{std::async(std::launch::async, [ctx, &p, &yb, N]() {
                     bn256_bases* r = nullptr;
                     chk(bn256_scalar_mul_batch(ctx, p.H.h, 0, (size_t)N, yb.data(), 0, &r));
                     return r;
                   }), ctx};
  Fr deltayz = sub(mul(sub(z, z2), vectorMul(onen, yn)), mul(z3, vectorMul(onen, twon)));
  Point c1 = fixedCom(p.g, p.h, proof.Tx, proof.TauX);
  Point c2 = scalarMult(proof.V, z2);
  c2 = pointAdd(c2, fixedMult(p.g, deltayz));
  c2 = pointAdd(c2, scalarMult(proof.T1Com, x));
  c2 = pointAdd(c2, scalarMult(proof.T2Com, x2));
  if (!pointEq(c1, c2)) return "failed: tx ?= t0 + t1*x +t2*x^2";
  Point P = pointAdd(proof.ACom, scalarMult(proof.SCom, x));
  DevPoints h1(ctx, h1f.take());
  // P += <G, -z 1^n> + <h1, z y^n + z^2 2^n>   (main.go:225-226): the two vectorPointScalarMul as ONE device MSM
  P = pointAdd(P, msmSegments(ctx, {{&p.G, 0, vectorMulOnScalar(onen, sub(kFrZero, z))},
                                    {&h1, 0, vectorAdd(vectorMulOnScalar(yn, z), vectorMulOnScalar(twon, z2))}}));
  P = pointAdd(P, fixedMult(p.h, sub(kFrZero, proof.Nu)));
  P = pointAdd(P, fixedMult(p.U, proof.Tx));
  InnerProductProof ip;
  ip.L = proof.L;
  ip.R = proof.R;
  ip.A = proof.A;
  ip.B = proof.B;
  return innerArgVerify(ctx, N, DevPoints::borrow(ctx, p.G.h), std::move(h1), p.U, P, ip);
}

// ------------------------------------------------------------------ wire format
// V, A, S, T1, T2 (64 B each) | tx, taux, nu (32 B each) | u32 BE k | L[k], R[k] (64 B each) | a, b (32 B)
std::vector<uint8_t> serialize(const BulletProof& pr) {
  std::vector<uint8_t> o;
  auto pt = [&](const Point& p) { o.insert(o.end(), p.b, p.b + 64); };
  auto sc = [&](const Fr& s) {
    uint8_t b[32];
    fr_to_be(b, s);
    o.insert(o.end(), b, b + 32);
  };
  pt(pr.V); pt(pr.ACom); pt(pr.SCom); pt(pr.T1Com); pt(pr.T2Com);
  sc(pr.Tx); sc(pr.TauX); sc(pr.Nu);
  uint32_t k = (uint32_t)pr.L.size();
  o.push_back((uint8_t)(k >> 24)); o.push_back((uint8_t)(k >> 16)); o.push_back((uint8_t)(k >> 8)); o.push_back((uint8_t)k);
  for (const Point& p : pr.L) pt(p);
  for (const Point& p : pr.R) pt(p);
  sc(pr.A); sc(pr.B);
  return o;
}
bool deserialize(BulletProof& pr, const uint8_t* in, size_t len) {
  size_t pos = 0;
  auto pt = [&](Point& p) {
    if (pos + 64 > len) return false;
    memcpy(p.b, in + pos, 64);
    pos += 64;
    hostg1::Pt t;
    return hostg1::unmarshal(t, p.b) == 0;
  };
  auto sc = [&](Fr& s) {
    if (pos + 32 > len) return false;
    uint64_t w[4];
    hostg1::be_load(w, in + pos);
    if (geq_n(w)) return false;
    s = fr_from_raw(w);
    pos += 32;
    return true;
  };
  if (!pt(pr.V) || !pt(pr.ACom) || !pt(pr.SCom) || !pt(pr.T1Com) || !pt(pr.T2Com)) return false;
  if (!sc(pr.Tx) || !sc(pr.TauX) || !sc(pr.Nu)) return false;
  if (pos + 4 > len) return false;
  uint32_t k = ((uint32_t)in[pos] << 24) | ((uint32_t)in[pos + 1] << 16) | ((uint32_t)in[pos + 2] << 8) | in[pos + 3];
  pos += 4;
  if (k > 64) return false;
  pr.L.resize(k);
  pr.R.resize(k);
  for (uint32_t i = 0; i < k; i++)
    if (!pt(pr.L[i])) return false;
  for (uint32_t i = 0; i < k; i++)
    if (!pt(pr.R[i])) return false;
  if (!sc(pr.A) || !sc(pr.B)) return false;
  return pos == len;
}

// ------------------------------------------------------------------ go/bppp: weighted norm linear argument
// BP++ helper semantics (go/bppp/util.go): pad-tolerant vector ops, even/odd reduction.
FrVec padTo(FrVec a, size_t n) {
  while (a.size() < n) a.push_back(kFrZero);
  return a;
}
Fr vectorMulPP(const FrVec& a, const FrVec& b) {  // go/bppp/util.go:143-157
  size_t n = std::max(a.size(), b.size());
  return vectorMul(padTo(a, n), padTo(b, n));
}
Fr weightVectorMul(const FrVec& a_in, const FrVec& b_in, const Fr& mu) {  // go/bppp/util.go:159-176
  size_t n = std::max(a_in.size(), b_in.size());
  FrVec a = padTo(a_in, n), b = padTo(b_in, n);
  Fr res = kFrZero, e = mu;
  for (size_t i = 0; i < n; i++) {
    res = add(res, mul(mul(a[i], b[i]), e));
    e = mul(e, mu);
  }
  return res;
}
// values(n) of go/bppp/util.go:11-25: rand.Int is drawn and discarded, the value is 2
FrVec valuesPP(ByteStream* rd, int n) {
  FrVec r(n);
  for (int i = 0; i < n; i++) {
    if (rd) rd->randInt();
    r[i] = fr_from_u64(2);
  }
  return r;
}
struct SSeg {
  const DevPoints* g;
  size_t off, stride, len;  // len = number of points of the (strided) view
  FrVec a;
};
// sum of BP++ vectorPointScalarMul over strided views (go/bppp/util.go:178-196: empty view ->
// identity, short scalar vectors zero-padded, long ones truncated)
Point msmStrided(bn256_ctx* ctx, const std::vector<SSeg>& segs) {
  std::vector<const bn256_bases*> hs;
  std::vector<size_t> offs, strides, ns;
  std::vector<std::vector<uint8_t>> raws;
  std::vector<const uint8_t*> ptrs;
  for (const SSeg& s : segs) {
    size_t n = std::min(s.len, s.a.size());  // zero padding contributes nothing
    if (n == 0) continue;
    hs.push_back(s.g->h);
    offs.push_back(s.off);
    strides.push_back(s.stride);
    ns.push_back(n);
    raws.push_back(scalarBytes(slice(s.a, 0, n)));
  }
  for (auto& r : raws) ptrs.push_back(r.data());
  Point out;
  chk(bn256_msm_multi_strided(ctx, (int)hs.size(), hs.data(), offs.data(), strides.data(), ns.data(), ptrs.data(),
                              out.b));
  return out;
}
struct DevFr {  // a []*big.Int resident in HBM (32-byte big-endian, reduced mod Order)
  bn256_ctx* ctx = nullptr;
  bn256_frvec* h = nullptr;
  size_t n = 0;
  DevFr() {}
  DevFr(bn256_ctx* c, bn256_frvec* v) : ctx(c), h(v), n(bn256_fr_len(v)) {}
  DevFr(bn256_ctx* c, const FrVec& a) : ctx(c) {
    std::vector<uint8_t> raw = scalarBytes(a);
    chk(bn256_fr_upload(c, raw.data(), a.size(), &h));
    n = a.size();
  }
  DevFr(const DevFr&) = delete;
  DevFr& operator=(const DevFr&) = delete;
  DevFr(DevFr&& o) noexcept : ctx(o.ctx), h(o.h), n(o.n) { o.h = nullptr; }
  DevFr& operator=(DevFr&& o) noexcept {
    if (this != &o) {
      if (h) bn256_fr_free(ctx, h);
      ctx = o.ctx; h = o.h; n = o.n; o.h = nullptr;
    }
    return *this;
  }
  ~DevFr() {
    if (h) bn256_fr_free(ctx, h);
  }
  FrVec download() const {
    std::vector<uint8_t> raw(32 * std::max<size_t>(n, 1));
    chk(bn256_fr_download(ctx, h, 0, n, raw.data()));
    FrVec r(n);
    for (size_t i = 0; i < n; i++) r[i] = fr_from_be(&raw[32 * i]);
    return r;
  }
};
// A reduceVector / reducePoints half (go/bppp/util.go:198-226) as a view: elements off, off+2, ...
struct View {
  size_t off, stride, len;
};
View evens(size_t n) { return {0, 2, (n + 1) / 2}; }
View odds(size_t n) { return {n > 1 ? (size_t)1 : 0, 2, n / 2}; }
View whole(size_t n) { return {0, 1, n}; }

// vectorMul / weightVectorMul over views, zero-padded to the longer side (go/bppp/util.go:143-176)
Fr devDot(bn256_ctx* ctx, const DevFr& a, View va, const DevFr& b, View vb, const Fr* mu) {
  uint8_t mub[32], out[32];
  if (mu) fr_to_be(mub, *mu);
  chk(bn256_fr_dot(ctx, a.h, va.off, va.stride, va.len, b.h, vb.off, vb.stride, vb.len, mu ? mub : nullptr, out));
  return fr_from_be(out);
}
// k inner products in one device call (bn256_fr_dot_multi)
struct DotArg {
  const DevFr* a;
  View va;
  const DevFr* b;
  View vb;
  const Fr* mu;
};
[[maybe_unused]] void devDotMulti(bn256_ctx* ctx, const std::vector<DotArg>& d, Fr* out) {
  const size_t k = d.size();
  std::vector<const bn256_frvec*> a(k), b(k);
  std::vector<size_t> ao(k), as(k), an(k), bo(k), bs(k), bn(k);
  std::vector<std::vector<uint8_t>> mu(k);
  std::vector<const uint8_t*> mup(k, nullptr);
  std::vector<uint8_t> res(32 * k);
  for (size_t i = 0; i < k; i++) {
    a[i] = d[i].a->h; ao[i] = d[i].va.off; as[i] = d[i].va.stride; an[i] = d[i].va.len;
    b[i] = d[i].b->h; bo[i] = d[i].vb.off; bs[i] = d[i].vb.stride; bn[i] = d[i].vb.len;
    if (d[i].mu) {
      mu[i].resize(32);
      fr_to_be(mu[i].data(), *d[i].mu);
      mup[i] = mu[i].data();
    }
  }
  chk(bn256_fr_dot_multi(ctx, (int)k, a.data(), ao.data(), as.data(), an.data(), b.data(), bo.data(), bs.data(), bn.data(),
                         mup.data(), res.data()));
  for (size_t i = 0; i < k; i++) out[i] = fr_from_be(&res[32 * i]);
}
// the same with the results left in HBM (bn256_fr_dot_multi_dev): nothing is waited for while the context defers
DevFr devDotMultiDev(bn256_ctx* ctx, const std::vector<DotArg>& d) {
  const size_t k = d.size();
  std::vector<const bn256_frvec*> a(k), b(k);
  std::vector<size_t> ao(k), as(k), an(k), bo(k), bs(k), bn(k);
  std::vector<std::vector<uint8_t>> mu(k);
  std::vector<const uint8_t*> mup(k, nullptr);
  for (size_t i = 0; i < k; i++) {
    a[i] = d[i].a->h; ao[i] = d[i].va.off; as[i] = d[i].va.stride; an[i] = d[i].va.len;
    b[i] = d[i].b->h; bo[i] = d[i].vb.off; bs[i] = d[i].vb.stride; bn[i] = d[i].vb.len;
    if (d[i].mu) {
      mu[i].resize(32);
      fr_to_be(mu[i].data(), *d[i].mu);
      mup[i] = mu[i].data();
    }
  }
  bn256_frvec* out = nullptr;
  chk(bn256_fr_dot_multi_dev(ctx, (int)k, a.data(), ao.data(), as.data(), an.data(), b.data(), bo.data(), bs.data(), bn.data(),
                             mup.data(), &out));
  return DevFr(ctx, out);
}
// three devFoldHalves in one device call (bn256_fr_fold_multi): c', l', n' of a WNLA round
void devFoldHalves3(bn256_ctx* ctx, const DevFr& v0, const Fr& a0, const Fr& b0, const DevFr& v1, const Fr& a1,
                    const Fr& b1, const DevFr& v2, const Fr& a2, const Fr& b2, DevFr& o0, DevFr& o1, DevFr& o2) {
  const DevFr* v[3] = {&v0, &v1, &v2};
  const Fr* sa[3] = {&a0, &a1, &a2};
  const Fr* sb[3] = {&b0, &b1, &b2};
  const bn256_frvec* h[3];
  size_t eo[3], en[3], oo[3], on[3], st[3] = {2, 2, 2};
  uint8_t ab[96], bb[96];
  for (int i = 0; i < 3; i++) {
    View e = evens(v[i]->n), o = odds(v[i]->n);
    h[i] = v[i]->h;
    eo[i] = e.off; en[i] = e.len; oo[i] = o.off; on[i] = o.len;
    fr_to_be(ab + 32 * i, *sa[i]);
    fr_to_be(bb + 32 * i, *sb[i]);
  }
  bn256_frvec* r[3] = {nullptr, nullptr, nullptr};
  chk(bn256_fr_fold_multi(ctx, 3, h, eo, en, h, oo, on, st, ab, bb, r));
  o0 = DevFr(ctx, r[0]);
  o1 = DevFr(ctx, r[1]);
  o2 = DevFr(ctx, r[2]);
}
// BP++ vectorPointScalarMul semantics on views (go/bppp/util.go:178-196): an empty view is the
// identity, the shorter side decides the length (zero padding / truncation).
// wall-clock split of the last wnla() by kind of work (BP_PROFILE=1 prints it to stderr)
struct Prof {
  double t[6] = {0, 0, 0, 0, 0, 0};
  static double now() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
  }
};
enum { PF_DOT, PF_MSM, PF_HOSTPT, PF_PTFOLD, PF_FRFOLD, PF_OTHER };

// Sum of BP++ vectorPointScalarMul terms over views as ONE device MSM; scale (optional) multiplies
// the segment's scalars on the device.
struct FSeg {
  const DevPoints* g;
  View vg;
  const DevFr* a;
  View va;
  const Fr* scale;
};
struct FusedArgs {
  std::vector<const bn256_bases*> hs;
  std::vector<const bn256_frvec*> vs;
  std::vector<size_t> off, stride, ns, voff, vstride;
  std::vector<std::vector<uint8_t>> scal;
  std::vector<const uint8_t*> sptr;
  explicit FusedArgs(const std::vector<FSeg>& segs) {
    for (const FSeg& s : segs) {
      size_t n = std::min(s.vg.len, s.va.len);
      if (n == 0) continue;
      hs.push_back(s.g->h);
      vs.push_back(s.a->h);
      off.push_back(s.vg.off);
      stride.push_back(s.vg.stride);
      ns.push_back(n);
      voff.push_back(s.va.off);
      vstride.push_back(s.va.stride);
      scal.emplace_back();
      if (s.scale) {
        scal.back().resize(32);
        fr_to_be(scal.back().data(), *s.scale);
      }
    }
    for (auto& v : scal) sptr.push_back(v.empty() ? nullptr : v.data());
  }
};
Point devMsmFused(bn256_ctx* ctx, const std::vector<FSeg>& segs) {
  FusedArgs a(segs);
  Point out;
  chk(bn256_msm_fr_multi(ctx, (int)a.hs.size(), a.hs.data(), a.off.data(), a.stride.data(), a.ns.data(), a.vs.data(),
                         a.voff.data(), a.vstride.data(), a.sptr.data(), out.b));
  return out;
}
// the same, enqueued on one of the library's lanes; devMsmWait returns the point
int devMsmFusedSubmit(bn256_ctx* ctx, const std::vector<FSeg>& segs) {
  FusedArgs a(segs);
  int ticket = -1;
  chk(bn256_msm_fr_multi_submit(ctx, (int)a.hs.size(), a.hs.data(), a.off.data(), a.stride.data(), a.ns.data(),
                                a.vs.data(), a.voff.data(), a.vstride.data(), a.sptr.data(), &ticket));
  return ticket;
}
Point devMsmWait(bn256_ctx* ctx, int ticket) {
  Point out;
  chk(bn256_msm_wait(ctx, ticket, out.b));
  return out;
}

// The scalars of commitments over generators that are no longer folded (bn256_fr_tensor_step_multi /
// bn256_fr_tensor_expand_multi): out_i[m] = w_i[j] * s_i(I) * v_i[I ^ (swap_i << sel_i)]; one device call for all
// of a round's vectors.  w = nullptr: no round since the freeze yet.
struct ExpandArg {
  const DevFr* v;
  size_t n_out;
  const DevFr* w;
  int sel;
  bool swap;
  Fr s_clear, s_set;
};
std::vector<DevFr> devTensorExpand(bn256_ctx* ctx, const std::vector<ExpandArg>& a, int layout) {
  const size_t k = a.size();
  std::vector<const bn256_frvec*> v(k), w(k);
  std::vector<size_t> n_out(k);
  std::vector<int> sel(k), swap(k);
  std::vector<uint8_t> sc(32 * k), ss(32 * k);
  for (size_t i = 0; i < k; i++) {
    v[i] = a[i].v->h;
    w[i] = a[i].w && a[i].w->h ? a[i].w->h : nullptr;
    n_out[i] = a[i].n_out;
    sel[i] = a[i].sel;
    swap[i] = a[i].swap;
    fr_to_be(&sc[32 * i], a[i].s_clear);
    fr_to_be(&ss[32 * i], a[i].s_set);
  }
  std::vector<bn256_frvec*> out(k, nullptr);
  chk(bn256_fr_tensor_expand_multi(ctx, (int)k, v.data(), n_out.data(), w.data(), layout, sel.data(), swap.data(), sc.data(),
                                   ss.data(), out.data()));
  std::vector<DevFr> r;
  for (size_t i = 0; i < k; i++) r.emplace_back(ctx, out[i]);
  return r;
}
// One host thread that runs small jobs in order, beside the caller: wnla's running commitment
// C' = C + y X + (y^2 - 1) R (two host scalar multiplications per round, ~85 us) is needed only by the
// final check, so it does not have to sit between a round's last wait and the next round's submits.
class HostChain {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<void()>> q;
  std::exception_ptr err;
  bool busy = false, quit = false;
  void run() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return quit || !q.empty(); });
      if (q.empty()) return;
      std::function<void()> job = std::move(q.front());
      q.pop_front();
      busy = true;
      lk.unlock();
      try {
        job();
      } catch (...) {
        lk.lock();
        if (!err) err = std::current_exception();
        lk.unlock();
      }
      lk.lock();
      busy = false;
      cv.notify_all();
    }
  }

 public:
  HostChain() : th([this] { run(); }) {}
  ~HostChain() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv.notify_all();
    th.join();
  }
  void push(std::function<void()> job) {
    {
      std::lock_guard<std::mutex> lk(mu);
      q.push_back(std::move(job));
    }
    cv.notify_all();
  }
  void wait() {  // every job pushed so far has run; rethrows the first exception of one
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return q.empty() && !busy; });
    if (err) {
      std::exception_ptr e = err;
      err = nullptr;
      std::rethrow_exception(e);
    }
  }
};

// ---- the inner-product argument over frozen generators (declared above innerArgProof)
static int ilog2(size_t n) {
  int r = 0;
  while (n >>= 1) r++;
  return r;
}
static void ipaStep(bn256_ctx* ctx, DevFr& wG, DevFr& wH, const Fr& x, const Fr& xinv, std::vector<DevFr>& keep) {
  // G' = xinv*G_lo + x*G_hi ; H' = x*H_lo + xinv*H_hi   (main.go:355-356): half folds, layout 1
  const bn256_frvec* w[2] = {wG.h, wH.h};
  uint8_t lo[64], hi[64];
  fr_to_be(lo, xinv);
  fr_to_be(lo + 32, x);
  fr_to_be(hi, x);
  fr_to_be(hi + 32, xinv);
  bn256_frvec* out[2] = {nullptr, nullptr};
  chk(bn256_fr_tensor_step_multi(ctx, 2, w, 1, lo, hi, out));
  keep.push_back(std::move(wG));
  keep.push_back(std::move(wH));
  wG = DevFr(ctx, out[0]);
  wH = DevFr(ctx, out[1]);
}
// The same expansion on the host, for frozen sets so small that the device's way costs more in calls than in
// arithmetic (two uploads with a wait each, the expansion and tensor-step launches, six vector handles to make
// and to release -- every release waits for the stream): at N = 64 a level's six vectors are 4 N = 256 products,
// ~6 us on one core.  k_fr_tensor_expand / k_fr_tensor_step, layout 1, restated: out[m] = W[m / n_t] * v[I ^ (swap
// << sel)] where bit `sel` of I = m mod n_t equals `half`, 0 elsewhere; W' = interleave(lo W, hi W).
static size_t ipaHostExpand() {  // read at every call: the tests switch between the two ways inside one process
  const char* e = getenv("BP_IPA_HOST_EXPAND");
  return e ? (size_t)strtoull(e, nullptr, 10) : 512;
}
static FrVec hostExpand(const FrVec& v, const FrVec& W, size_t n_out, size_t n_t, int sel, bool swap, int half) {
  FrVec out(n_out, kFrZero);  // half: 0 / 1 = only where the bit is clear / set, 2 = everywhere
  for (size_t m = 0; m < n_out; m++) {
    const size_t I = m & (n_t - 1), j = m / n_t, bit = (I >> sel) & 1u;
    if (half != 2 && (int)bit != half) continue;
    const size_t src = swap ? (I ^ ((size_t)1 << sel)) : I;
    if (src >= v.size()) continue;
    out[m] = W.empty() ? v[src] : mul(v[src], W[j]);
  }
  return out;
}
static FrVec hostTensorStep(const FrVec& W, const Fr& lo, const Fr& hi) {
  const size_t len = W.empty() ? 1 : W.size();
  FrVec out(2 * len);
  for (size_t j = 0; j < len; j++) {
    out[2 * j] = W.empty() ? lo : mul(W[j], lo);
    out[2 * j + 1] = W.empty() ? hi : mul(W[j], hi);
  }
  return out;
}
static InnerProductProof innerArgProofFrozenHost(bn256_ctx* ctx, int N, const DevPoints& G, const DevPoints& H, const Point& U,
                                                 FrVec a, FrVec b, InnerProductProof proof) {
  const size_t Nf = (size_t)N;
  FrVec WG, WH;
  while (N != 1) {
    const size_t n1 = (size_t)N / 2;
    const int sel = ilog2((size_t)N) - 1;
    FrVec alo = slice(a, 0, n1), ahi = slice(a, n1, N), blo = slice(b, 0, n1), bhi = slice(b, n1, N);
    Fr cl = vectorMul(alo, bhi), cr = vectorMul(ahi, blo);
    // the scalars of innerArgProofFrozen's six expanded vectors, made here; three MSMs over the whole frozen G, H
    int tl = msmSegmentsSubmit(ctx, {{&G, 0, hostExpand(a, WG, Nf, (size_t)N, sel, true, 1)}, {&H, 0, hostExpand(b, WH, Nf, (size_t)N, sel, true, 0)}});
    int tr = msmSegmentsSubmit(ctx, {{&G, 0, hostExpand(a, WG, Nf, (size_t)N, sel, true, 0)}, {&H, 0, hostExpand(b, WH, Nf, (size_t)N, sel, true, 1)}});
    int tp = msmSegmentsSubmit(ctx, {{&G, 0, hostExpand(a, WG, Nf, (size_t)N, sel, false, 2)}, {&H, 0, hostExpand(b, WH, Nf, (size_t)N, sel, false, 2)}});
    Point ul = fixedMult(U, cl), ur = fixedMult(U, cr), up = fixedMult(U, vectorMul(a, b));
    Point L = pointAdd(msmWait(ctx, tl), ul);
    Point R = pointAdd(msmWait(ctx, tr), ur);
    Point P = pointAdd(msmWait(ctx, tp), up);
    proof.L.push_back(L);
    proof.R.push_back(R);
    Fr x = hashFS({fr_from_u64((uint64_t)N)}, {P, L, R});
    Fr xinv = fr_inv(x);
    WG = hostTensorStep(WG, xinv, x);  // G' = xinv*G_lo + x*G_hi ; H' = x*H_lo + xinv*H_hi   (main.go:355-356)
    WH = hostTensorStep(WH, x, xinv);
    FrVec a1 = vectorAdd(vectorMulOnScalar(alo, x), vectorMulOnScalar(ahi, xinv));
    FrVec b1 = vectorAdd(vectorMulOnScalar(bhi, x), vectorMulOnScalar(blo, xinv));
    a = a1;
    b = b1;
    N = (int)n1;
  }
  proof.A = a[0];
  proof.B = b[0];
  return proof;
}
InnerProductProof innerArgProofFrozen(bn256_ctx* ctx, int N, DevPoints G, DevPoints H, const Point& U, FrVec a, FrVec b,
                                      InnerProductProof proof) {
  const size_t Nf = (size_t)N;  // G, H stay as they are from here on
  if (Nf <= ipaHostExpand()) return innerArgProofFrozenHost(ctx, N, G, H, U, std::move(a), std::move(b), std::move(proof));
  DevFr wG, wH;
  std::vector<DevFr> keep;
  while (N != 1) {
    const size_t n1 = (size_t)N / 2;
    const int sel = ilog2((size_t)N) - 1;  // the top bit of a level's index: lower / upper half
    FrVec alo = slice(a, 0, n1), ahi = slice(a, n1, N), blo = slice(b, 0, n1), bhi = slice(b, n1, N);
    Fr cl = vectorMul(alo, bhi), cr = vectorMul(ahi, blo);
    DevFr da(ctx, a), db(ctx, b);
    // L = <G[n1:], a[:n1]> + <H[:n1], b[n1:]> + cl*U ; R = <G[:n1], a[n1:]> + <H[n1:], b[:n1]> + cr*U ; P = <G, a> + <H, b> + <a,b>*U
    // (the expanded vectors are not waited for by the host: the three lanes wait for them on the device)
    chk(bn256_ctx_fr_defer_sync(ctx, 1));
    struct DeferOff {
      bn256_ctx* c;
      ~DeferOff() { (void)bn256_ctx_fr_defer_sync(c, 0); }
    } defer_off{ctx};
    std::vector<DevFr> e = devTensorExpand(ctx, {{&da, Nf, &wG, sel, true, kFrZero, kFrOne}, {&db, Nf, &wH, sel, true, kFrOne, kFrZero},
                                                 {&da, Nf, &wG, sel, true, kFrOne, kFrZero}, {&db, Nf, &wH, sel, true, kFrZero, kFrOne},
                                                 {&da, Nf, &wG, sel, false, kFrOne, kFrOne}, {&db, Nf, &wH, sel, false, kFrOne, kFrOne}}, 1);
    int tl = devMsmFusedSubmit(ctx, {{&G, whole(Nf), &e[0], whole(Nf), nullptr}, {&H, whole(Nf), &e[1], whole(Nf), nullptr}});
    int tr = devMsmFusedSubmit(ctx, {{&G, whole(Nf), &e[2], whole(Nf), nullptr}, {&H, whole(Nf), &e[3], whole(Nf), nullptr}});
    int tp = devMsmFusedSubmit(ctx, {{&G, whole(Nf), &e[4], whole(Nf), nullptr}, {&H, whole(Nf), &e[5], whole(Nf), nullptr}});
    Point ul = fixedMult(U, cl), ur = fixedMult(U, cr), up = fixedMult(U, vectorMul(a, b));
    Point L = pointAdd(devMsmWait(ctx, tl), ul);
    Point R = pointAdd(devMsmWait(ctx, tr), ur);
    Point P = pointAdd(devMsmWait(ctx, tp), up);
    proof.L.push_back(L);
    proof.R.push_back(R);
    Fr x = hashFS({fr_from_u64((uint64_t)N)}, {P, L, R});
    Fr xinv = fr_inv(x);
    ipaStep(ctx, wG, wH, x, xinv, keep);
    FrVec a1 = vectorAdd(vectorMulOnScalar(alo, x), vectorMulOnScalar(ahi, xinv));
    FrVec b1 = vectorAdd(vectorMulOnScalar(bhi, x), vectorMulOnScalar(blo, xinv));
    a = a1;
    b = b1;
    N = (int)n1;
  }
  proof.A = a[0];
  proof.B = b[0];
  return proof;
}
std::string innerArgVerifyFrozen(bn256_ctx* ctx, int N, DevPoints G, DevPoints H, const Point& U, Point P,
                                 const InnerProductProof& proof, size_t deep) {
  const size_t Nf = (size_t)N;
  const bool host = Nf <= ipaHostExpand();  // the tensor of the challenges on the host (see innerArgProofFrozenHost)
  FrVec WG, WH;
  DevFr wG, wH;
  std::vector<DevFr> keep;
  HostChain helper;
  while (N != 1) {
    if (deep >= proof.L.size() || deep >= proof.R.size()) return "failed to verify inner argument proof";
    Fr x = hashFS({fr_from_u64((uint64_t)N)}, {P, proof.L[deep], proof.R[deep]});
    Fr xinv = fr_inv(x), x2 = mul(x, x), x2inv = fr_inv(x2);
    if (host) {
      WG = hostTensorStep(WG, xinv, x);
      WH = hostTensorStep(WH, x, xinv);
    } else
      ipaStep(ctx, wG, wH, x, xinv, keep);  // enqueued only: the device works beside the host's P update
    // x^2 L and x^-2 R (main.go:305-307) are the verifier's critical path -- 2 x ~85 us a level, and the next
    // challenge hashes the P they give -- so the second one runs on the helper thread
    Point rx;
    helper.push([&rx, &proof, deep, x2inv] { rx = scalarMult(proof.R[deep], x2inv); });
    const Point lx = scalarMult(proof.L[deep], x2);
    helper.wait();
    P = pointAdd(pointAdd(lx, P), rx);
    N /= 2;
    deep++;
  }
  // com(G[0], H[0], a, b) + (a*b)*U  (main.go:275-279) with G[0] = <G_f, W_G>, H[0] = <H_f, W_H>: one MSM
  if (host) {
    if (WG.empty()) WG = WH = FrVec{kFrOne};  // N = 1 from the start
    Point p = pointAdd(msmSegments(ctx, {{&G, 0, vectorMulOnScalar(WG, proof.A)}, {&H, 0, vectorMulOnScalar(WH, proof.B)}}),
                       fixedMult(U, mul(proof.A, proof.B)));
    return pointEq(P, p) ? "" : "failed to verify inner argument proof";
  }
  FrVec one1 = {kFrOne};
  DevFr d1(ctx, one1);
  std::vector<DevFr> e = devTensorExpand(ctx, {{&d1, Nf, &wG, 0, false, proof.A, proof.A}, {&d1, Nf, &wH, 0, false, proof.B, proof.B}}, 1);
  Point p = pointAdd(devMsmFused(ctx, {{&G, whole(Nf), &e[0], whole(Nf), nullptr}, {&H, whole(Nf), &e[1], whole(Nf), nullptr}}),
                     scalarMult(U, mul(proof.A, proof.B)));
  if (!pointEq(P, p)) return "failed to verify inner argument proof";
  return "";
}

struct WnlaResult {
  std::vector<Point> X, R;
  FrVec l, n;
  bool verified = false;
};
// wnla, go/bppp/docs_test.go:1269-1338 (recursion as a loop).  G, H, c, l, n all stay in HBM; the
// even/odd halves of reducePoints / reduceVector are stride-2 views; per round only the 64-byte
// X, R and a few 32-byte inner products cross PCIe.  <rho*n1, G0> is taken as rho*<n1, G0> (one
// host scalar mult of the 64-byte result instead of a scaled copy of the vector).
WnlaResult wnla(bn256_ctx* ctx, const Point& g, DevPoints G, DevPoints H, DevFr c, Point C, Fr ro, Fr mu, DevFr l,
                DevFr n, ByteStream* rd) {
  WnlaResult res;
  const FixedBase gTable(g);  // v_x g and v_r g of every round (docs_test.go:1312,1318)
  HostChain chain;  // finishes X, R and C' beside the rounds (declared after everything its jobs touch: res, gTable, C)
  Prof pf;
  double tp = Prof::now();
  auto lap = [&](int k) {
    double t = Prof::now();
    pf.t[k] += t - tp;
    tp = t;
  };
  auto report = [&]() {
    if (getenv("BP_PROFILE"))
      fprintf(stderr, "[wnla] rounds=%zu ms: dot=%.2f msm=%.2f host_point=%.2f point_fold=%.2f fr_fold=%.2f other=%.2f\n",
              res.X.size(), pf.t[PF_DOT], pf.t[PF_MSM], pf.t[PF_HOSTPT], pf.t[PF_PTFOLD], pf.t[PF_FRFOLD], pf.t[PF_OTHER]);
  };
  const bool per_round = getenv("BP_PROFILE") && atoi(getenv("BP_PROFILE")) >= 2;
  double prev[6] = {0, 0, 0, 0, 0, 0};
  // Small rounds: once both generator vectors are down to kFreeze points they are no longer folded.  A fold is
  // a latency chain of ~130 doublings per output, ~0.45 ms whatever the length, and the next round cannot
  // start without it: 11 of the 19 rounds at |l| = 2^20 were exactly that long.  Instead the round-t generators
  // stay what they are, G_t[I] = sum_j W(j) G_f[2^t I + j] (W the tensor of the rounds' fold scalars), and
  // every later commitment <G_t, a> is taken as <G_f, W (x) a>: one MSM over the frozen vector with scalars
  // that a small Fr kernel expands (bn256_fr_tensor_expand_multi).  The same group elements, X and R byte for
  // byte.  BP_WNLA_FREEZE=<points> moves the threshold (0: fold to the end, as the reference does).
  static const size_t kFreeze =
      std::min<size_t>(getenv("BP_WNLA_FREEZE") ? (size_t)strtoull(getenv("BP_WNLA_FREEZE"), nullptr, 10) : 4096, (size_t)1 << 22);
  bool frozen = false;
  int ft = 0;                    // rounds since the freeze
  DevFr wG, wH;                  // the tensor of the fold scalars of those rounds: G' = rho*G0 + y*G1, H' = H0 + y*H1
  // Frozen rounds run TWO at a time: the protocol is interactive (y comes from the reader, docs_test.go:1294, not
  // from a hash of X and R), so nothing of round k + 1 -- its scalar folds, inner products, expanded vectors --
  // needs X_k or R_k, only the running C' does, and that is read at the very end.  A round's two MSMs take two of
  // the four lanes; while they run the host prepares and submits the next round's on the other two, and only then
  // collects.  Every vector a submitted MSM may still read -- and every vector whose release would wait for the
  // lanes (bn256_fr_free does) -- stays in the round's record until its results are in.
  // Nor does the host wait inside a frozen round: the five inner products stay in HBM (bn256_fr_dot_multi_dev) --
  // v_x and v_r enter only X and R, through v_x g and v_r g -- and are read when the round is collected; the
  // lanes' streams wait on the device for the round's expanded vectors (bn256_ctx_fr_defer_sync).  A small round
  // was ~0.35 ms, most of it the wait for a chain of tiny kernels; now the host enqueues and moves on, and the
  // host thread beside it finishes X, R and C' in round order.
  struct PendingRound {
    int tx, tr;
    DevFr dots;
    Fr y, roinv;
    std::vector<DevFr> keep;
    std::vector<DevPoints> keepPts;  // a folding round's old generators, which its MSMs still read
  };
  std::deque<PendingRound> pending;
  // A collected round's vectors are released at the end: bn256_fr_free waits for EVERY lane that holds a call (it
  // cannot know which one reads the vector), so releasing them here made each collect wait for the round submitted
  // after it -- the rounds ran one after the other (kernel trace: the lanes' kernels never overlapped).  They are
  // the small rounds' vectors only (at most kFreeze elements each).
  std::vector<DevFr> retired;
  std::vector<DevPoints> retiredPts;  // (bn256_bases_free joins the busy lanes as well)
  size_t rounds_done = 0;  // res.X belongs to the chain's thread while frozen rounds are being collected
  auto collect = [&]() {
    PendingRound pr = std::move(pending.front());
    pending.pop_front();
    lap(PF_OTHER);
    const FrVec d5 = pr.dots.download();  // waits for this context's stream: the round's Fr kernels are long done
    lap(PF_DOT);
    const Point mx = devMsmWait(ctx, pr.tx), mr = devMsmWait(ctx, pr.tr);
    lap(PF_MSM);
    const Fr y = pr.y, roinv_k = pr.roinv;
    retired.push_back(std::move(pr.dots));
    for (DevFr& x : pr.keep) retired.push_back(std::move(x));
    for (DevPoints& x : pr.keepPts) retiredPts.push_back(std::move(x));
    chain.push([&C, &res, &gTable, d5, mx, mr, y, roinv_k] {  // C, res.X, res.R are read again only after chain.wait()
      const Fr vx = add(mul(d5[0], mul(fr_from_u64(2), roinv_k)), add(d5[1], d5[2]));
      const Fr vr = add(d5[3], d5[4]);
      const Point X = pointAdd(gTable.mul(vx), mx), R = pointAdd(gTable.mul(vr), mr);
      res.X.push_back(X);
      res.R.push_back(R);
      Point C_ = pointAdd(C, scalarMult(X, y));
      C = pointAdd(C_, scalarMult(R, sub(mul(y, y), kFrOne)));
    });
  };
  for (;;) {
    if (per_round) {  // BP_PROFILE=2: the split of every round
      if (rounds_done)
        fprintf(stderr, "[wnla] round %zu |l|=%zu: dot=%.3f msm=%.3f host_point=%.3f point_fold=%.3f fr_fold=%.3f other=%.3f\n",
                rounds_done, 2 * l.n, pf.t[0] - prev[0], pf.t[1] - prev[1], pf.t[2] - prev[2], pf.t[3] - prev[3],
                pf.t[4] - prev[4], pf.t[5] - prev[5]);
      for (int i = 0; i < 6; i++) prev[i] = pf.t[i];
    }
    rounds_done++;
    Fr roinv = fr_inv(ro);
    lap(PF_OTHER);
    if (l.n + n.n < 6) {
      while (!pending.empty()) collect();
      lap(PF_MSM);
      Fr v = add(devDot(ctx, c, whole(c.n), l, whole(l.n), nullptr), devDot(ctx, n, whole(n.n), n, whole(n.n), &mu));
      Point Cc;
      if (frozen) {
        std::vector<DevFr> e = devTensorExpand(ctx, {{&l, H.n, &wH, 0, false, kFrOne, kFrOne},
                                                     {&n, G.n, &wG, 0, false, kFrOne, kFrOne}}, 0);
        Cc = pointAdd(scalarMult(g, v), devMsmFused(ctx, {{&H, whole(H.n), &e[0], whole(e[0].n), nullptr},
                                                           {&G, whole(G.n), &e[1], whole(e[1].n), nullptr}}));
      } else
        Cc = pointAdd(scalarMult(g, v), devMsmFused(ctx, {{&H, whole(H.n), &l, whole(l.n), nullptr},
                                                           {&G, whole(G.n), &n, whole(n.n), nullptr}}));
      res.l = l.download();
      res.n = n.download();
      chain.wait();
      res.verified = pointEq(Cc, C);
      lap(PF_OTHER);
      report();
      return res;
    }
    Fr y = valuesPP(rd, 1)[0];
    View c0 = evens(c.n), c1 = odds(c.n), l0 = evens(l.n), l1 = odds(l.n), n0 = evens(n.n), n1 = odds(n.n);
    Fr mu2 = mul(mu, mu);
    if (!frozen && kFreeze && G.n <= kFreeze && H.n <= kFreeze) {
      frozen = true;
    }
    if (frozen) {
      // X = v_x g + <H0, l1> + <H1, l0> + rho <G0, n1> + rho^-1 <G1, n0>,  R = v_r g + <H1, l1> + <G1, n1>
      // (docs_test.go:1312-1320) over the frozen H, G: four expanded scalar vectors, one call.  The round's three
      // groups of Fr kernels -- expanded vectors, next round's c, l, n, the five inner products -- are enqueued
      // back to back and not waited for (the inner products are read when the round is collected).
      // two lanes per round; collected BEFORE this round's kernels are enqueued: reading a round's inner products
      // waits for this context's stream, which then holds nothing newer than the previous round's kernels
      while (2 * pending.size() + 2 > BN256_ASYNC_LANES) collect();
      lap(PF_MSM);
      chk(bn256_ctx_fr_defer_sync(ctx, 1));
      struct DeferOff {
        bn256_ctx* c;
        ~DeferOff() { (void)bn256_ctx_fr_defer_sync(c, 0); }
      } defer_off{ctx};
      std::vector<DevFr> e = devTensorExpand(ctx, {{&l, H.n, &wH, 0, true, kFrOne, kFrOne},
                                                   {&n, G.n, &wG, 0, true, ro, roinv},
                                                   {&l, H.n, &wH, 0, false, kFrZero, kFrOne},
                                                   {&n, G.n, &wG, 0, false, kFrZero, kFrOne}}, 0);
      lap(PF_FRFOLD);
      // the commitments first: their lanes wait (on the device) for what this context has enqueued so far, which
      // should be the expanded vectors and not also the inner products
      PendingRound pr;
      pr.y = y;
      pr.roinv = roinv;
      pr.tx = devMsmFusedSubmit(ctx, {{&H, whole(H.n), &e[0], whole(e[0].n), nullptr}, {&G, whole(G.n), &e[1], whole(e[1].n), nullptr}});
      pr.tr = devMsmFusedSubmit(ctx, {{&H, whole(H.n), &e[2], whole(e[2].n), nullptr}, {&G, whole(G.n), &e[3], whole(e[3].n), nullptr}});
      lap(PF_MSM);
      DevFr c_, l_, n_;
      devFoldHalves3(ctx, c, kFrOne, y, l, kFrOne, y, n, roinv, y, c_, l_, n_);
      lap(PF_FRFOLD);
      pr.dots = devDotMultiDev(ctx, {{&n, n0, &n, n1, &mu2}, {&c, c0, &l, l1, nullptr}, {&c, c1, &l, l0, nullptr},
                                     {&n, n1, &n, n1, &mu2}, {&c, c1, &l, l1, nullptr}});
      lap(PF_DOT);
      // this round's vectors retire with its record (their release would wait for the lanes)
      for (DevFr& x : e) pr.keep.push_back(std::move(x));
      pr.keep.push_back(std::move(c));
      pr.keep.push_back(std::move(l));
      pr.keep.push_back(std::move(n));
      DevFr wg_old = std::move(wG), wh_old = std::move(wH);
      {
        const bn256_frvec* w[2] = {wg_old.h, wh_old.h};
        uint8_t lo[64], hi[64];
        fr_to_be(lo, ro);
        fr_to_be(lo + 32, kFrOne);
        fr_to_be(hi, y);
        fr_to_be(hi + 32, y);
        bn256_frvec* out[2] = {nullptr, nullptr};
        chk(bn256_fr_tensor_step_multi(ctx, 2, w, 0, lo, hi, out));
        wG = DevFr(ctx, out[0]);
        wH = DevFr(ctx, out[1]);
      }
      pr.keep.push_back(std::move(wg_old));
      pr.keep.push_back(std::move(wh_old));
      pending.push_back(std::move(pr));
      ft++;
      c = std::move(c_);
      ro = mu;
      mu = mu2;
      l = std::move(l_);
      n = std::move(n_);
      lap(PF_OTHER);
      continue;
    }
    View G0 = evens(G.n), G1 = odds(G.n), H0 = evens(H.n), H1 = odds(H.n);
    // H_ = H0 (+) y*H1 ; G_ = ro*G0 (+) y*G1   (docs_test.go:1325-1326), ragged folds on device.  Neither
    // needs anything of this round but y and ro, and y comes from the reader before X and R exist
    // (docs_test.go:1294), so both folds -- ~0.5 ms latency chains of ~130 doublings whatever the length --
    // go to two lanes FIRST and run under the whole round instead of after it (9 of the 28 ms at
    // |l| = 2^20 were these chains, waited for at the end of every round).
    // Folding rounds are collected late as well (round 3): X and R of round k are needed by nothing before the final
    // check, so round k + 1's folds and commitments are submitted while round k's MSMs still run -- two lanes for
    // the folds (free again when the round's fold waits return), two for the commitments (held until the round is
    // collected): 2 * pending + 4 lanes.  The host waits only for the folds, whose output the next round reads.
    while (2 * pending.size() + 4 > BN256_ASYNC_LANES) collect();
    lap(PF_MSM);
    chk(bn256_ctx_fr_defer_sync(ctx, 1));
    struct DeferOff {
      bn256_ctx* c;
      ~DeferOff() { (void)bn256_ctx_fr_defer_sync(c, 0); }
    } defer_off{ctx};
    uint8_t oneb[32], yb[32], rob[32];
    fr_to_be(oneb, kFrOne);
    fr_to_be(yb, y);
    fr_to_be(rob, ro);
    bn256_bases *hn = nullptr, *gn = nullptr;
    int tg = -1, th = -1;
    chk(bn256_fold2_submit(ctx, G.h, G0.off, G0.len, G.h, G1.off, G1.len, 2, rob, yb, &gn, &tg));
    DevPoints Gn(ctx, gn);
    chk(bn256_fold2_submit(ctx, H.h, H0.off, H0.len, H.h, H1.off, H1.len, 2, oneb, yb, &hn, &th));
    DevPoints Hn(ctx, hn);
    lap(PF_PTFOLD);
    // the two commitments' MSMs go to two more lanes (their streams wait on the device for l, n of the last round's
    // un-waited folds); the round's inner products and vector folds follow on the context's own stream
    PendingRound pr;
    pr.y = y;
    pr.roinv = roinv;
    pr.tx = devMsmFusedSubmit(ctx, {{&H, H0, &l, l1, nullptr}, {&H, H1, &l, l0, nullptr}, {&G, G0, &n, n1, &ro},
                                    {&G, G1, &n, n0, &roinv}});
    pr.tr = devMsmFusedSubmit(ctx, {{&H, H1, &l, l1, nullptr}, {&G, G1, &n, n1, nullptr}});
    lap(PF_MSM);
    // the round's five inner products (docs_test.go:1297-1311), left in HBM until the round is collected
    pr.dots = devDotMultiDev(ctx, {{&n, n0, &n, n1, &mu2}, {&c, c0, &l, l1, nullptr}, {&c, c1, &l, l0, nullptr},
                                   {&n, n1, &n, n1, &mu2}, {&c, c1, &l, l1, nullptr}});
    lap(PF_DOT);
    DevFr c_, l_, n_;
    devFoldHalves3(ctx, c, kFrOne, y, l, kFrOne, y, n, roinv, y, c_, l_, n_);
    lap(PF_FRFOLD);
    chk(bn256_fold_wait(ctx, tg));
    chk(bn256_fold_wait(ctx, th));
    lap(PF_PTFOLD);
    // the old generators and vectors live on with the round's record: its MSMs read them
    pr.keepPts.push_back(std::move(G));
    pr.keepPts.push_back(std::move(H));
    pr.keep.push_back(std::move(c));
    pr.keep.push_back(std::move(l));
    pr.keep.push_back(std::move(n));
    pending.push_back(std::move(pr));
    G = std::move(Gn);
    H = std::move(Hn);
    c = std::move(c_);
    ro = mu;
    mu = mu2;
    l = std::move(l_);
    n = std::move(n_);
    lap(PF_OTHER);
  }
}

// ------------------------------------------------------------------ go/bppp: arithmetic-circuit protocol
// ArithmeticCircuitProtocol / InnerArithmeticCircuitProtocol2, go/bppp/docs_test.go:496-1240: every
// commitment (Com, CommitOL, CommitR, Cs, PT, the C(t) checks) is one fused device MSM over the
// resident HVec / GVec; the scalar algebra between them is host work on small vectors and dense
// matrices, as in the reference.  Its debug prints (Wl*w, Eq. 34, the polynomial cross-checks)
// produce no protocol value and are not restated.
FrVec vectorAddPP(const FrVec& a, const FrVec& b) {  // go/bppp/util.go:245-260
  size_t n = std::max(a.size(), b.size());
  return vectorAdd(padTo(a, n), padTo(b, n));
}
FrVec vectorSubPP(const FrVec& a, const FrVec& b) {  // go/bppp/util.go:262-277
  size_t n = std::max(a.size(), b.size());
  return vectorSub(padTo(a, n), padTo(b, n));
}
FrVec powvector(const Fr& v, int n) {  // go/bppp/util.go:54-63
  return n > 0 ? ntharr(v, n) : FrVec();
}
Fr fpow(const Fr& x, long y) {  // go/bppp/util.go:110-116
  Fr b = y < 0 ? fr_inv(x) : x, acc = kFrOne;
  unsigned long e = (unsigned long)(y < 0 ? -y : y);
  for (; e; e >>= 1) {
    if (e & 1) acc = mul(acc, b);
    b = mul(b, b);
  }
  return acc;
}
FrVec vectorTensorMul(const FrVec& a, const FrVec& b) {  // go/bppp/util.go:134-141
  FrVec r;
  for (const Fr& y : b)
    for (const Fr& x : a) r.push_back(mul(x, y));
  return r;
}
typedef std::vector<FrVec> FrMat;
FrVec vectorMulOnMatrix(const FrVec& a, const FrMat& m) {  // go/bppp/util.go:295-309
  FrVec r;
  for (size_t j = 0; j < m[0].size(); j++) {
    FrVec col;
    for (const FrVec& row : m) col.push_back(row[j]);
    r.push_back(vectorMulPP(a, col));
  }
  return r;
}
FrVec matrixMulOnVector(const FrVec& a, const FrMat& m) {  // go/bppp/util.go:311-319
  FrVec r;
  for (const FrVec& row : m) r.push_back(vectorMulPP(a, row));
  return r;
}
FrMat diagInv(const Fr& x, int n) {  // go/bppp/util.go:321-340
  FrMat r(n, FrVec(n, kFrZero));
  Fr inv = fr_inv(x), val = inv;
  for (int i = 0; i < n; i++) {
    r[i][i] = val;
    val = mul(val, inv);
  }
  return r;
}
FrVec concat(const FrVec& a, const FrVec& b) {
  FrVec r(a);
  r.insert(r.end(), b.begin(), b.end());
  return r;
}
Fr neg(const Fr& x) { return sub(kFrZero, x); }  // minus(), go/bppp/util.go:76-78

struct ACPublic {  // go/bppp/docs_test.go:12-30
  int Nm, Nl, Nv, Nw, No, K;
  Point G;
  DevPoints GVec, HVec;
  FrMat Wm, Wl;
  FrVec Am, Al;
  bool Fl, Fm;
  std::vector<Point> V;
};
struct AcPrivate {  // go/bppp/docs_test.go:34-41; f tabulated: f[typ-1][j] = index into wo or -1
  FrMat v;
  FrVec sv, wl, wr, wo;
  std::vector<int> f[4];
};
struct ACTranscript {
  std::vector<Point> V;
  Point Co, Cl, Cr, Vlin, Cs, PT, CT;
  bool checkV, checkCT, check45;
  Fr vT;
  FrVec lT, nT, cT;
  WnlaResult w;
};
// <HVec, a> + <GVec, b> with the BP++ vectorPointScalarMul rules, one device MSM
Point comHG(bn256_ctx* ctx, const ACPublic& p, const FrVec& a, const FrVec& b) {
  return msmStrided(ctx, {{&p.HVec, 0, 1, p.HVec.n, a}, {&p.GVec, 0, 1, p.GVec.n, b}});
}
// Com, go/bppp/docs_test.go:514-519
Point Com(bn256_ctx* ctx, const FrVec& v, const Fr& s, const Point& G, const DevPoints& H) {
  Point res = pointAdd(scalarMult(G, v[0]), scalarMult(H.at(0), s));
  return pointAdd(res, msmStrided(ctx, {{&H, 9, 1, H.n > 9 ? H.n - 9 : 0, slice(v, 1, v.size())}}));
}
FrVec partitionOf(const AcPrivate& prv, int typ, int n) {  // docs_test.go:533-558, :579-586
  FrVec r(n, kFrZero);
  for (int j = 0; j < n; j++) {
    int i = prv.f[typ - 1][j];
    if (i >= 0) r[j] = prv.wo.at((size_t)i);
  }
  return r;
}
FrMat mappedColumns(const FrMat& W, const AcPrivate& prv, int typ, int n) {  // docs_test.go:652-759
  FrMat r;
  for (const FrVec& row : W) {
    FrVec o(n, kFrZero);
    for (int j = 0; j < n; j++) {
      int i = prv.f[typ - 1][j];
      if (i >= 0) o[j] = row.at((size_t)i);
    }
    r.push_back(o);
  }
  return r;
}
FrMat columns(const FrMat& W, size_t lo, size_t hi) {
  FrMat r;
  for (const FrVec& row : W) r.push_back(slice(row, lo, hi));
  return r;
}

ACTranscript arithmeticCircuitProtocol(bn256_ctx* ctx, ACPublic& pub, const AcPrivate& prv, ByteStream* rd) {
  ACTranscript T;
  const int Nm = pub.Nm, Nl = pub.Nl, Nv = pub.Nv, K = pub.K;
  const Fr two = fr_from_u64(2);
  // :497-500
  pub.V.clear();
  for (int i = 0; i < K; i++) pub.V.push_back(Com(ctx, prv.v[i], prv.sv[i], pub.G, pub.HVec));
  T.V = pub.V;
  // CommitOL, :522-567
  FrVec ro_ = valuesPP(rd, 7), rl_ = valuesPP(rd, 6);
  FrVec ro = {ro_[0], ro_[1], ro_[2], ro_[3], kFrZero, ro_[4], ro_[5], ro_[6], kFrZero};
  FrVec rl = {rl_[0], rl_[1], rl_[2], kFrZero, rl_[3], rl_[4], rl_[5], kFrZero, kFrZero};
  FrVec nl = prv.wl;
  FrVec no = partitionOf(prv, 4, Nm), lo = partitionOf(prv, 1, Nv), ll = partitionOf(prv, 2, Nv);
  Point Co = comHG(ctx, pub, concat(ro, lo), no), Cl = comHG(ctx, pub, concat(rl, ll), nl);
  // CommitR, :569-591
  FrVec rr_ = valuesPP(rd, 5);
  FrVec rr = {rr_[0], rr_[1], kFrZero, rr_[2], rr_[3], rr_[4], kFrZero, kFrZero, kFrZero};
  FrVec nr = prv.wr, lr = partitionOf(prv, 3, Nv);
  Point Cr = comHG(ctx, pub, concat(rr, lr), nr);
  T.Co = Co;
  T.Cl = Cl;
  T.Cr = Cr;
  // InnerArithmeticCircuitProtocol2, :593-1240
  Fr ch_ro = valuesPP(rd, 1)[0], ch_lambda = valuesPP(rd, 1)[0], ch_beta = valuesPP(rd, 1)[0], ch_delta = valuesPP(rd, 1)[0];
  FrMat MlnL = columns(pub.Wl, 0, Nm), MmnL = columns(pub.Wm, 0, Nm);
  FrMat MlnR = columns(pub.Wl, Nm, 2 * Nm), MmnR = columns(pub.Wm, Nm, 2 * Nm);
  FrMat WlO = columns(pub.Wl, 2 * Nm, pub.Wl.empty() ? 0 : pub.Wl[0].size());
  FrMat WmO = columns(pub.Wm, 2 * Nm, pub.Wm.empty() ? 0 : pub.Wm[0].size());
  FrMat MlnO = mappedColumns(WlO, prv, 4, Nm), MmnO = mappedColumns(WmO, prv, 4, Nm);
  FrMat MllL = mappedColumns(WlO, prv, 2, Nv), MmlL = mappedColumns(WmO, prv, 2, Nv);
  FrMat MllR = mappedColumns(WlO, prv, 3, Nv), MmlR = mappedColumns(WmO, prv, 3, Nv);
  FrMat MllO = mappedColumns(WlO, prv, 1, Nv), MmlO = mappedColumns(WmO, prv, 1, Nv);

  Fr ch_mu = mul(ch_ro, ch_ro);
  const Fr bFl = pub.Fl ? kFrOne : kFrZero, bFm = pub.Fm ? kFrOne : kFrZero;
  auto lcomb = [&](int i) {  // :778-783
    return add(mul(bFl, fpow(ch_lambda, (long)Nv * i)), mul(bFm, fpow(ch_mu, (long)Nv * i + 1)));
  };
  // V_, :786-797
  hostg1::Pt acc = hostg1::infinity();
  for (int i = 0; i < K; i++) acc = hostg1::padd(acc, pt_load(scalarMult(pub.V[i], lcomb(i))));
  Point V_ = scalarMult(pt_from(acc), two);
  T.Vlin = V_;
  // lambda, mu vectors, :800-809
  FrVec lambda = vectorAddPP(
      vectorTensorMul(vectorMulOnScalar(powvector(ch_lambda, Nv), ch_mu), powvector(fpow(ch_mu, Nv), K)),
      vectorTensorMul(powvector(ch_mu, Nv), powvector(fpow(ch_lambda, Nv), K)));
  lambda = vectorMulOnScalar(lambda, (pub.Fl && pub.Fm) ? kFrOne : kFrZero);
  lambda = vectorSubPP(powvector(ch_lambda, Nl), lambda);
  FrVec mu = vectorMulOnScalar(powvector(ch_mu, Nm), ch_mu);
  // coefficients, :824-833
  FrMat muDiagInv = diagInv(ch_mu, Nm);
  auto cn = [&](const FrMat& Ml, const FrMat& Mm) {
    return vectorMulOnMatrix(vectorSubPP(vectorMulOnMatrix(lambda, Ml), vectorMulOnMatrix(mu, Mm)), muDiagInv);
  };
  auto cl = [&](const FrMat& Ml, const FrMat& Mm) {
    return vectorSubPP(vectorMulOnMatrix(lambda, Ml), vectorMulOnMatrix(mu, Mm));
  };
  FrVec cnL = cn(MlnL, MmnL), cnR = cn(MlnR, MmnR), cnO = cn(MlnO, MmnO);
  FrVec clL = cl(MllL, MmlL), clR = cl(MllR, MmlR), clO = cl(MllO, MmlO);
  // :836-879
  FrVec ls = valuesPP(rd, Nv), ns = valuesPP(rd, Nm);
  Fr v_ = kFrZero, rv1 = kFrZero;
  FrVec v_1 = {kFrZero};
  for (int i = 0; i < K; i++) {
    v_ = add(v_, mul(prv.v[i][0], lcomb(i)));
    rv1 = add(rv1, mul(prv.sv[i], lcomb(i)));
    v_1 = vectorAddPP(v_1, vectorMulOnScalar(slice(prv.v[i], 1, prv.v[i].size()), lcomb(i)));
  }
  v_ = mul(v_, two);
  FrVec rv(9, kFrZero);
  rv[0] = mul(rv1, two);
  v_1 = vectorMulOnScalar(v_1, two);
  // Check V_ correct, :882-886
  T.checkV = pointEq(V_, pointAdd(scalarMult(pub.G, v_), comHG(ctx, pub, concat(rv, v_1), {})));
  // :888-891
  FrVec pl = powvector(ch_lambda, Nv), pm = powvector(ch_mu, Nv);
  FrVec cl0 = vectorSubPP(vectorMulOnScalar(slice(pl, 1, pl.size()), bFl),
                          vectorMulOnScalar(vectorMulOnScalar(slice(pm, 1, pm.size()), ch_mu), bFm));
  // f'(T), :894-936 (f_[3] is printed only)
  const Fr d = ch_delta, dinv = fr_inv(ch_delta), twod = mul(two, d), twodinv = mul(two, dinv);
  auto wv = [&](const FrVec& a, const FrVec& b) { return weightVectorMul(a, b, ch_mu); };
  auto vm = [&](const FrVec& a, const FrVec& b) { return vectorMulPP(a, b); };
  const FrVec nlR = vectorAddPP(nl, cnR), nrL = vectorAddPP(nr, cnL);
  Fr fm2 = neg(wv(ns, ns));
  Fr fm1 = add(vm(cl0, ls), mul(twod, wv(ns, no)));
  Fr f0 = neg(mul(two, vm(clR, ls)));
  f0 = sub(f0, mul(d, vm(cl0, lo)));
  f0 = sub(f0, mul(wv(ns, nlR), two));
  f0 = sub(f0, mul(mul(d, d), wv(no, no)));
  Fr f1 = mul(two, vm(clL, ls));
  f1 = add(f1, mul(two, mul(d, vm(clR, lo))));
  f1 = add(f1, vm(cl0, ll));
  f1 = add(f1, mul(wv(ns, nrL), two));
  f1 = add(f1, mul(wv(no, nlR), twod));
  Fr f2 = wv(cnR, cnR);
  f2 = sub(f2, mul(two, mul(dinv, vm(clO, ls))));
  f2 = sub(f2, mul(two, mul(d, vm(clL, lo))));
  f2 = sub(f2, mul(two, vm(clR, ll)));
  f2 = sub(f2, vm(cl0, lr));
  f2 = sub(f2, mul(twodinv, wv(ns, cnO)));
  f2 = sub(f2, mul(twod, wv(no, nrL)));
  f2 = sub(f2, wv(nlR, nlR));
  Fr f4 = mul(twodinv, wv(cnO, cnR));
  f4 = add(f4, wv(cnL, cnL));
  f4 = sub(f4, mul(twodinv, vm(clO, ll)));
  f4 = sub(f4, mul(two, vm(clL, lr)));
  f4 = sub(f4, mul(two, vm(clR, v_1)));
  f4 = sub(f4, mul(twodinv, wv(nlR, cnO)));
  f4 = sub(f4, wv(nrL, nrL));
  Fr f5 = neg(mul(twodinv, wv(cnO, cnL)));
  f5 = add(f5, mul(twodinv, vm(clO, lr)));
  f5 = add(f5, mul(two, vm(clL, v_1)));
  f5 = add(f5, mul(twodinv, wv(nrL, cnO)));
  Fr f6 = neg(mul(twodinv, vm(clO, v_1)));
  // rs, :984-998
  const Fr binv = fr_inv(ch_beta);
  FrVec rs = {
      add(fm1, mul(ch_beta, mul(d, ro[1]))),
      mul(fm2, binv),
      mul(sub(add(f0, mul(d, ro[0])), mul(ch_beta, rl[1])), binv),
      add(mul(sub(f1, rl[0]), binv), add(rr[1], mul(d, ro[2]))),
      add(mul(add(f2, rr[0]), binv), sub(mul(d, ro[3]), rl[2])),
      neg(mul(rv[0], binv)),
      add(mul(f4, binv), add(mul(d, ro[5]), sub(rr[3], rl[4]))),
      add(mul(f5, binv), sub(add(rr[4], mul(d, ro[6])), rl[5])),
      add(mul(f6, binv), add(sub(mul(d, ro[7]), rl[6]), rr[5])),
  };
  Point Cs = comHG(ctx, pub, concat(rs, ls), ns);  // :1002-1003
  T.Cs = Cs;
  // :1009-1012
  Fr t = valuesPP(rd, 1)[0], tinv = fr_inv(t), t2 = mul(t, t), t3 = mul(t2, t);
  // l(T), :1038-1042
  FrVec lT = vectorMulOnScalar(concat(rs, ls), tinv);
  lT = vectorSubPP(lT, vectorMulOnScalar(concat(ro, lo), d));
  lT = vectorAddPP(lT, vectorMulOnScalar(concat(rl, ll), t));
  lT = vectorSubPP(lT, vectorMulOnScalar(concat(rr, lr), t2));
  lT = vectorAddPP(lT, vectorMulOnScalar(concat(rv, v_1), t3));
  // :1044-1050
  FrVec pnT = vectorMulOnScalar(cnO, mul(dinv, t3));
  pnT = vectorSubPP(pnT, vectorMulOnScalar(cnL, t2));
  pnT = vectorAddPP(pnT, vectorMulOnScalar(cnR, t));
  Fr psT = wv(pnT, pnT);
  psT = add(psT, mul(two, mul(vm(lambda, pub.Al), t3)));
  psT = sub(psT, mul(two, mul(vm(mu, pub.Am), t3)));
  // :1069-1074
  FrVec n_T = vectorMulOnScalar(ns, tinv);
  n_T = vectorSubPP(n_T, vectorMulOnScalar(no, d));
  n_T = vectorAddPP(n_T, vectorMulOnScalar(nl, t));
  n_T = vectorSubPP(n_T, vectorMulOnScalar(nr, t2));
  FrVec nT = vectorAddPP(pnT, n_T);
  // PT, :1077-1078
  Point PT = pointAdd(scalarMult(pub.G, psT), comHG(ctx, pub, {}, pnT));
  T.PT = PT;
  // :1080-1096, :1171
  FrVec cr_T = {kFrOne, mul(ch_beta, tinv), mul(ch_beta, t), mul(ch_beta, t2), mul(ch_beta, t3),
                mul(ch_beta, mul(t, t3)), mul(ch_beta, mul(t2, t3)), mul(ch_beta, mul(t3, t3)),
                mul(ch_beta, mul(mul(t3, t), t3))};
  FrVec cl_T = vectorMulOnScalar(clO, mul(t3, dinv));
  cl_T = vectorSubPP(cl_T, vectorMulOnScalar(clL, t2));
  cl_T = vectorAddPP(cl_T, vectorMulOnScalar(clR, t));
  cl_T = vectorMulOnScalar(cl_T, two);
  cl_T = vectorSubPP(cl_T, cl0);
  FrVec cT = concat(cr_T, cl_T);
  // CT, :1173-1179
  Point CLeft = scalarMult(Cs, tinv);
  CLeft = pointAdd(CLeft, scalarMult(Co, neg(d)));
  CLeft = pointAdd(CLeft, scalarMult(Cl, t));
  CLeft = pointAdd(CLeft, scalarMult(Cr, neg(t2)));
  CLeft = pointAdd(CLeft, scalarMult(V_, t3));
  Point CT = pointAdd(PT, CLeft);
  T.CT = CT;
  Fr vT = add(psT, mul(v_, t3));
  T.vT = vT;
  // Check C(t) = ComNormInnerArg(l, n, v), :1183-1189; Check (45), :1192-1203
  T.checkCT = pointEq(CT, pointAdd(scalarMult(pub.G, vT), comHG(ctx, pub, lT, nT)));
  T.check45 = pointEq(CLeft, pointAdd(scalarMult(pub.G, mul(v_, t3)), comHG(ctx, pub, lT, n_T)));
  T.lT = lT;
  T.nT = nT;
  T.cT = cT;
  // :1239
  T.w = wnla(ctx, pub.G, cloneRange(ctx, pub.GVec), cloneRange(ctx, pub.HVec), DevFr(ctx, cT), CT, ch_ro, ch_mu,
             DevFr(ctx, lT), DevFr(ctx, nT), rd);
  return T;
}

thread_local std::string g_err;

}  // namespace

// ---------------------------------------------------------------------------- C exports
extern "C" const char* bp_last_error(void) { return g_err.c_str(); }

extern "C" int bp_keccak256(const uint8_t* msg, size_t len, uint8_t out[32]) {
  std::vector<uint8_t> m(msg, msg + len);
  keccak256(out, m);
  return 0;
}

extern "C" int bp_prove_det(bn256_ctx* ctx, int n, const uint8_t v_be[32], uint64_t seed, uint8_t* out, size_t cap,
                            size_t* out_len) {
  try {
    if (!ctx || !v_be || !out || !out_len || n < 1) return BN256_EINVAL;
    ByteStream rd{seed};
    BulletProofPublic pub = newBulletProofPublic(ctx, rd, n);
    Fr prv = rd.values(1)[0];  // test harness draws prv before Prove (go/bp/main_test.go:20)
    BulletProof pr = prove(ctx, pub, fr_from_be(v_be), prv, rd);
    std::vector<uint8_t> ser = serialize(pr);
    *out_len = ser.size();
    if (ser.size() > cap) return BN256_ELENGTH;
    memcpy(out, ser.data(), ser.size());
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);  // a ticket lost to the exception must not keep its lane busy
    return BP_EFAIL;
  }
}

extern "C" int bp_verify_det(bn256_ctx* ctx, int n, uint64_t seed, const uint8_t* proof, size_t len) {
  try {
    if (!ctx || !proof || n < 1) return BN256_EINVAL;
    ByteStream rd{seed};
    BulletProofPublic pub = newBulletProofPublic(ctx, rd, n);
    BulletProof pr;
    if (!deserialize(pr, proof, len)) {
      g_err = "malformed proof encoding";
      return BP_EFAIL;
    }
    std::string e = verify(ctx, pub, pr);
    g_err = e;
    return e.empty() ? 0 : BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);  // a ticket lost to the exception must not keep its lane busy
    return BP_EFAIL;
  }
}

// ---- product entry points: the caller's generators (resident handles), blinding from the OS CSPRNG
namespace {
bool pointOk(const uint8_t b[64]) {
  hostg1::Pt t;
  return hostg1::unmarshal(t, b) == 0;
}
BulletProofPublic publicFrom(bn256_ctx* ctx, int n, const bn256_bases* G, const bn256_bases* H, const uint8_t U[64],
                             const uint8_t g[64], const uint8_t h[64]) {
  if (bn256_bases_len(G) != (size_t)n || bn256_bases_len(H) != (size_t)n) throw std::runtime_error("invalid length for scalar mul");
  if (!pointOk(U) || !pointOk(g) || !pointOk(h)) throw std::runtime_error("bn256: malformed point");
  BulletProofPublic p;
  p.N = n;
  p.G = DevPoints::borrow(ctx, G);
  p.H = DevPoints::borrow(ctx, H);
  memcpy(p.U.b, U, 64);
  memcpy(p.g.b, g, 64);
  memcpy(p.h.b, h, 64);
  return p;
}
}  // namespace

extern "C" int bp_setup(bn256_ctx* ctx, int n, bn256_bases** G, bn256_bases** H, uint8_t U[64], uint8_t g[64],
                        uint8_t h[64]) {
  try {
    if (!ctx || !G || !H || !U || !g || !h || n < 1) return BN256_EINVAL;
    *G = *H = nullptr;
    ByteStream rd{0};
    rd.os_random = true;
    BulletProofPublic pub = newBulletProofPublic(ctx, rd, n);
    *G = pub.G.h;
    *H = pub.H.h;
    pub.G.h = pub.H.h = nullptr;  // ownership passes to the caller
    memcpy(U, pub.U.b, 64);
    memcpy(g, pub.g.b, 64);
    memcpy(h, pub.h.b, 64);
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

extern "C" int bp_prove(bn256_ctx* ctx, int n, const bn256_bases* G, const bn256_bases* H, const uint8_t U[64],
                        const uint8_t g[64], const uint8_t h[64], const uint8_t v_be[32], const uint8_t prv_be[32],
                        uint8_t* out, size_t cap, size_t* out_len) {
  try {
    if (!ctx || !G || !H || !U || !g || !h || !v_be || !prv_be || !out || !out_len || n < 1) return BN256_EINVAL;
    BulletProofPublic pub = publicFrom(ctx, n, G, H, U, g, h);
    ByteStream rd{0};
    rd.os_random = true;  // alpha, sl, sr, ro, tau1, tau2: crypto/rand in the reference (go/bp/main.go:87-123)
    BulletProof pr = prove(ctx, pub, fr_from_be(v_be), fr_from_be(prv_be), rd);
    std::vector<uint8_t> ser = serialize(pr);
    *out_len = ser.size();
    if (ser.size() > cap) return BN256_ELENGTH;
    memcpy(out, ser.data(), ser.size());
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

extern "C" int bp_verify(bn256_ctx* ctx, int n, const bn256_bases* G, const bn256_bases* H, const uint8_t U[64],
                         const uint8_t g[64], const uint8_t h[64], const uint8_t* proof, size_t len) {
  try {
    if (!ctx || !G || !H || !U || !g || !h || !proof || n < 1) return BN256_EINVAL;
    BulletProofPublic pub = publicFrom(ctx, n, G, H, U, g, h);
    BulletProof pr;
    if (!deserialize(pr, proof, len)) {
      g_err = "malformed proof encoding";
      return BP_EFAIL;
    }
    std::string e = verify(ctx, pub, pr);
    g_err = e;
    return e.empty() ? 0 : BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

// Inner-product argument on its own (InnerArgumentPublic.Prove / Verify, go/bp/main.go:263-271):
// generators from the seed (G[n], H[n], U), a and b given; returns P || L.. || R.. || a || b.
extern "C" int bp_inner_product_roundtrip(bn256_ctx* ctx, int n, uint64_t seed, const uint8_t* a_be, const uint8_t* b_be,
                                          int tamper) {
  try {
    if (!ctx || !a_be || !b_be || n < 1) return BN256_EINVAL;
    ByteStream rd{seed};
    DevPoints G = points(ctx, rd, n), H = points(ctx, rd, n);
    Point U = points(ctx, rd, 1).at(0);
    FrVec a(n), b(n);
    for (int i = 0; i < n; i++) {
      a[i] = fr_from_be(a_be + 32 * i);
      b[i] = fr_from_be(b_be + 32 * i);
    }
    Point P = pointAdd(msmSegments(ctx, {{&G, 0, a}, {&H, 0, b}}), scalarMult(U, vectorMul(a, b)));
    InnerProductProof ip = innerArgProof(ctx, n, DevPoints::borrow(ctx, G.h), DevPoints::borrow(ctx, H.h), U, a, b);
    if (tamper) ip.A = add(ip.A, kFrOne);
    std::string e = innerArgVerify(ctx, n, std::move(G), std::move(H), U, P, ip);
    g_err = e;
    return e.empty() ? 0 : BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);  // a ticket lost to the exception must not keep its lane busy
    return BP_EFAIL;
  }
}

// TestWNLA, go/bppp/docs_test.go:1242-1267, with the reader injected: g, G[nn], H[nl] from
// RandomG1, c = values(nl), ro = values(1)[0], mu = ro^2, C = v*g + <H,l> + <G,n>.
// out: C | u32 rounds | (X, R) per round | u32 |l| | l... | u32 |n| | n...
extern "C" int bp_wnla_det(bn256_ctx* ctx, uint64_t seed, int nl, int nn, const uint8_t* l_be, const uint8_t* n_be,
                           uint8_t* out, size_t cap, size_t* out_len) {
  try {
    if (!ctx || !l_be || !n_be || !out || !out_len || nl < 1 || nn < 1) return BN256_EINVAL;
    ByteStream rd{seed};
    Point g = points(ctx, rd, 1).at(0);
    DevPoints G = points(ctx, rd, nn), H = points(ctx, rd, nl);
    FrVec c = valuesPP(&rd, nl);
    Fr ro = valuesPP(&rd, 1)[0], mu = mul(ro, ro);
    FrVec l(nl), n(nn);
    for (int i = 0; i < nl; i++) l[i] = fr_from_be(l_be + 32 * i);
    for (int i = 0; i < nn; i++) n[i] = fr_from_be(n_be + 32 * i);
    Fr v = add(vectorMulPP(c, l), weightVectorMul(n, n, mu));
    Point C = pointAdd(scalarMult(g, v), msmStrided(ctx, {{&H, 0, 1, H.n, l}, {&G, 0, 1, G.n, n}}));
    WnlaResult r = wnla(ctx, g, std::move(G), std::move(H), DevFr(ctx, c), C, ro, mu, DevFr(ctx, l), DevFr(ctx, n), &rd);
    std::vector<uint8_t> o(C.b, C.b + 64);
    auto u32be = [&](uint32_t k) {
      o.push_back((uint8_t)(k >> 24)); o.push_back((uint8_t)(k >> 16)); o.push_back((uint8_t)(k >> 8)); o.push_back((uint8_t)k);
    };
    u32be((uint32_t)r.X.size());
    for (size_t i = 0; i < r.X.size(); i++) {
      o.insert(o.end(), r.X[i].b, r.X[i].b + 64);
      o.insert(o.end(), r.R[i].b, r.R[i].b + 64);
    }
    for (const FrVec* vv : {&r.l, &r.n}) {
      u32be((uint32_t)vv->size());
      for (const Fr& x : *vv) {
        uint8_t b[32];
        fr_to_be(b, x);
        o.insert(o.end(), b, b + 32);
      }
    }
    *out_len = o.size();
    if (o.size() > cap) return BN256_ELENGTH;
    memcpy(out, o.data(), o.size());
    g_err = r.verified ? "" : "Failed to verify!";
    return r.verified ? 0 : BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);  // a ticket lost to the exception must not keep its lane busy
    return BP_EFAIL;
  }
}

// ArithmeticCircuitProtocol (go/bppp/docs_test.go:496-1240) on a caller-described circuit, with the
// reader injected: `pre_draws` values are drawn first (TestAC3 / TestAC4 draw sv and e before their
// generators), then G = points(1)[0], GVec = points(Nm), HVec = points(9 + Nv), then `post_draws`
// values (TestACProtocol / TestArithmeticCircuit2 draw sv after them).
// out: V[K] | Co Cl Cr V_ Cs PT CT | 3 check bytes (V_, C(t), Eq. 45) | vT | u32 |l(T)|, l(T).. |
//      u32 |n(T)|, n(T).. | u32 |c(T)|, c(T).. | WNLA part as bp_wnla_det without its leading C
extern "C" int bp_ac_det(bn256_ctx* ctx, uint64_t seed, int pre_draws, int post_draws, const bp_ac_circuit* c,
                         uint8_t* out, size_t cap, size_t* out_len) {
  try {
    if (!ctx || !c || !out || !out_len) return BN256_EINVAL;
    if (c->Nm < 1 || c->Nl < 1 || c->Nv < 1 || c->No < 0 || c->K < 1 || c->Nw != 2 * c->Nm + c->No) return BN256_EINVAL;
    if (!c->Wm || !c->Wl || !c->Am || !c->Al || !c->v || !c->sv || !c->wl || !c->wr || (!c->wo && c->No) || !c->f_lo ||
        !c->f_ll || !c->f_lr || !c->f_no)
      return BN256_EINVAL;
    ByteStream rd{seed};
    valuesPP(&rd, pre_draws);
    ACPublic pub;
    pub.Nm = c->Nm; pub.Nl = c->Nl; pub.Nv = c->Nv; pub.Nw = c->Nw; pub.No = c->No; pub.K = c->K;
    pub.Fl = c->Fl != 0;
    pub.Fm = c->Fm != 0;
    pub.G = points(ctx, rd, 1).at(0);
    pub.GVec = points(ctx, rd, c->Nm);
    pub.HVec = points(ctx, rd, 9 + c->Nv);
    valuesPP(&rd, post_draws);
    auto vec = [](const uint8_t* p, int n) {
      FrVec r(n);
      for (int i = 0; i < n; i++) r[i] = fr_from_be(p + 32 * (size_t)i);
      return r;
    };
    auto mat = [&](const uint8_t* p, int rows, int cols) {
      FrMat m;
      for (int i = 0; i < rows; i++) m.push_back(vec(p + 32 * (size_t)i * cols, cols));
      return m;
    };
    pub.Wm = mat(c->Wm, c->Nm, c->Nw);
    pub.Wl = mat(c->Wl, c->Nl, c->Nw);
    pub.Am = vec(c->Am, c->Nm);
    pub.Al = vec(c->Al, c->Nl);
    AcPrivate prv;
    prv.v = mat(c->v, c->K, c->Nv);
    prv.sv = vec(c->sv, c->K);
    prv.wl = vec(c->wl, c->Nm);
    prv.wr = vec(c->wr, c->Nm);
    prv.wo = vec(c->wo, c->No);
    const int32_t* fs[4] = {c->f_lo, c->f_ll, c->f_lr, c->f_no};
    const int fn[4] = {c->Nv, c->Nv, c->Nv, c->Nm};
    for (int k = 0; k < 4; k++)
      for (int j = 0; j < fn[k]; j++) {
        if (fs[k][j] >= c->No) return BN256_EINVAL;
        prv.f[k].push_back(fs[k][j]);
      }
    ACTranscript T = arithmeticCircuitProtocol(ctx, pub, prv, &rd);
    std::vector<uint8_t> o;
    auto pt = [&](const Point& p) { o.insert(o.end(), p.b, p.b + 64); };
    auto u32be = [&](uint32_t k) {
      o.push_back((uint8_t)(k >> 24)); o.push_back((uint8_t)(k >> 16)); o.push_back((uint8_t)(k >> 8)); o.push_back((uint8_t)k);
    };
    auto frs = [&](const FrVec& v) {
      u32be((uint32_t)v.size());
      for (const Fr& x : v) {
        uint8_t b[32];
        fr_to_be(b, x);
        o.insert(o.end(), b, b + 32);
      }
    };
    for (const Point& p : T.V) pt(p);
    for (const Point* p : {&T.Co, &T.Cl, &T.Cr, &T.Vlin, &T.Cs, &T.PT, &T.CT}) pt(*p);
    o.push_back(T.checkV);
    o.push_back(T.checkCT);
    o.push_back(T.check45);
    uint8_t b[32];
    fr_to_be(b, T.vT);
    o.insert(o.end(), b, b + 32);
    frs(T.lT);
    frs(T.nT);
    frs(T.cT);
    u32be((uint32_t)T.w.X.size());
    for (size_t i = 0; i < T.w.X.size(); i++) {
      pt(T.w.X[i]);
      pt(T.w.R[i]);
    }
    frs(T.w.l);
    frs(T.w.n);
    *out_len = o.size();
    if (o.size() > cap) return BN256_ELENGTH;
    memcpy(out, o.data(), o.size());
    g_err = T.w.verified ? "" : "Failed to verify!";
    return T.w.verified ? 0 : BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);  // a ticket lost to the exception must not keep its lane busy
    return BP_EFAIL;
  }
}

// matrixMulOnVector(w, Wl) with w = wl || wr || wo: the vector the reference's circuit tests print as
// "Wl*w" (go/bppp/docs_test.go:326; util.go:311-319) -- the one multi-entry Fr answer the reference's
// tests record (docs_test.go:365).  Host Fr arithmetic only; out: Nl x 32 bytes.
extern "C" int bp_ac_wlw(const bp_ac_circuit* c, uint8_t* out) {
  try {
    if (!c || !out || c->Nm < 1 || c->Nl < 1 || c->No < 0 || c->Nw != 2 * c->Nm + c->No) return BN256_EINVAL;
    if (!c->Wl || !c->wl || !c->wr || (!c->wo && c->No)) return BN256_EINVAL;
    FrVec w;
    for (int i = 0; i < c->Nm; i++) w.push_back(fr_from_be(c->wl + 32 * (size_t)i));
    for (int i = 0; i < c->Nm; i++) w.push_back(fr_from_be(c->wr + 32 * (size_t)i));
    for (int i = 0; i < c->No; i++) w.push_back(fr_from_be(c->wo + 32 * (size_t)i));
    FrMat Wl;
    for (int i = 0; i < c->Nl; i++) {
      FrVec row(c->Nw);
      for (int j = 0; j < c->Nw; j++) row[j] = fr_from_be(c->Wl + 32 * ((size_t)i * c->Nw + j));
      Wl.push_back(row);
    }
    FrVec r = matrixMulOnVector(w, Wl);
    for (int i = 0; i < c->Nl; i++) fr_to_be(out + 32 * (size_t)i, r[i]);
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return BP_EFAIL;
  }
}

// BASELINE.json configs[3] shape: WNLA over commitment vectors of length nl / nn (e.g. 2^20) with
// synthetic generators (bn256_random_bases) and scalars; every X / R commitment and every
// generator fold runs on the device.  Reports wall milliseconds and the number of rounds; the
// argument's own final check (docs_test.go:1273-1288) is the correctness criterion.
extern "C" int bp_wnla_synthetic(bn256_ctx* ctx, uint64_t seed, size_t nl, size_t nn, double* ms_setup, double* ms_run,
                                 int* rounds) {
  try {
    if (!ctx || nl < 1 || nn < 1) return BN256_EINVAL;
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    bn256_bases *gh = nullptr, *hh = nullptr, *g1 = nullptr;
    chk(bn256_random_bases(ctx, seed, 0, nn, 0, &gh));
    DevPoints G(ctx, gh);
    chk(bn256_random_bases(ctx, seed + 1, 0, nl, 0, &hh));
    DevPoints H(ctx, hh);
    chk(bn256_random_bases(ctx, seed + 2, 0, 1, 0, &g1));
    DevPoints gdev(ctx, g1);
    Point g = gdev.at(0);
    DevFr c(ctx, valuesPP(nullptr, (int)nl));
    bn256_frvec *lh = nullptr, *nh = nullptr;
    chk(bn256_fr_random(ctx, seed + 3, 0, nl, &lh));
    DevFr l(ctx, lh);
    chk(bn256_fr_random(ctx, seed + 4, 0, nn, &nh));
    DevFr n(ctx, nh);
    Fr ro = fr_from_u64(2), mu = mul(ro, ro);
    Fr v = add(devDot(ctx, c, whole(c.n), l, whole(l.n), nullptr), devDot(ctx, n, whole(n.n), n, whole(n.n), &mu));
    Point C = pointAdd(scalarMult(g, v), devMsmFused(ctx, {{&H, whole(H.n), &l, whole(l.n), nullptr},
                                                             {&G, whole(G.n), &n, whole(n.n), nullptr}}));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    WnlaResult r = wnla(ctx, g, std::move(G), std::move(H), std::move(c), C, ro, mu, std::move(l), std::move(n), nullptr);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    if (ms_setup) *ms_setup = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    if (ms_run) *ms_run = (t2.tv_sec - t1.tv_sec) * 1e3 + (t2.tv_nsec - t1.tv_nsec) * 1e-6;
    if (rounds) *rounds = (int)r.X.size();
    g_err = r.verified ? "" : "Failed to verify!";
    return r.verified ? 0 : BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);  // a ticket lost to the exception must not keep its lane busy
    return BP_EFAIL;
  }
}

// ============================================================================ SURVEY 8f rank 4
// go/schnorr-bn256 (schnorr.go) and go/ve-ca (main.go, types.go, enc.go) over the same ABI.  Their
// vector work is what the device does: k = 128 independent two-term commitments (bn256_com_batch),
// batches of hash_i * PublicKey_i + R_i and S_i * G (bn256_scalar_mul_batch, bn256_add_batch);
// hashes, the XOR cipher and the modular exponentiations stay on the host as they are scalar work.
namespace {
typedef std::vector<uint8_t> ByteVec;
void appendBig(ByteVec& o, const uint8_t be[32]) {  // big.Int.Bytes(): minimal big-endian, empty for 0
  size_t i = 0;
  while (i < 32 && be[i] == 0) i++;
  o.insert(o.end(), be + i, be + 32);
}
void appendFr(ByteVec& o, const Fr& a) {  // FBytes, go/ve-ca/types.go:69-71
  uint8_t b[32];
  fr_to_be(b, a);
  appendBig(o, b);
}
Fr hashModOrder(const ByteVec& data) {  // Keccak256(...) mod Order: Msg (schnorr.go:26-28), Hash (ve-ca/types.go:93-96)
  uint8_t d[32];
  keccak256(d, data);
  return fr_from_be(d);
}
Fr frPow(const Fr& base, const Fr& e) {  // FPow: new(big.Int).Exp(f1, f2, Order), go/ve-ca/types.go:65-67
  uint64_t w[4];
  fr_to_raw(w, e);
  Fr acc = kFrOne;
  for (int i = 255; i >= 0; i--) {
    acc = mul(acc, acc);
    if ((w[i >> 6] >> (i & 63)) & 1) acc = mul(acc, base);
  }
  return acc;
}
// hash := Msg(m.Bytes(), PublicKey.Marshal(), R.Marshal()), schnorr.go:33,50,70
Fr schnorrChallenge(const uint8_t m_be[32], const uint8_t pub[64], const uint8_t R[64]) {
  ByteVec d;
  appendBig(d, m_be);
  d.insert(d.end(), pub, pub + 64);
  d.insert(d.end(), R, R + 64);
  return hashModOrder(d);
}
// s = r + hash * prv as the reference leaves it: a plain big.Int sum, NOT reduced (schnorr.go:34,52)
void schnorrS(uint8_t S_be[64], const Fr& r, const Fr& hash, const Fr& prv) {
  uint64_t a[4], b[4], c[4], t[9] = {0};
  fr_to_raw(a, hash);
  fr_to_raw(b, prv);
  fr_to_raw(c, r);
  for (int i = 0; i < 4; i++) {
    uint64_t carry = 0;
    for (int j = 0; j < 4; j++) {
      u128 cur = (u128)a[i] * b[j] + t[i + j] + carry;
      t[i + j] = (uint64_t)cur;
      carry = (uint64_t)(cur >> 64);
    }
    t[i + 4] = carry;
  }
  u128 cy = 0;
  for (int i = 0; i < 8; i++) {
    cy += (u128)t[i] + (i < 4 ? c[i] : 0);
    t[i] = (uint64_t)cy;
    cy >>= 64;
  }
  for (int i = 0; i < 8; i++)
    for (int k = 0; k < 8; k++) S_be[(7 - i) * 8 + k] = (uint8_t)(t[i] >> (56 - 8 * k));
}
Fr frFromBe64(const uint8_t be[64]) {  // a 512-bit integer mod Order
  const Fr two256 = fr_from_raw(kFrOne.l);  // the raw words of Montgomery 1 are 2^256 mod n
  return add(mul(fr_from_be(be), two256), fr_from_be(be + 32));
}

constexpr int kVecaK = BP_VECA_K;
struct VecaProof {  // Proof, go/ve-ca/main.go:10-16; E holds XOR results, which may exceed Order: raw bytes
  uint8_t E[4][kVecaK][32];
  Fr Alpha[2][kVecaK], P[2][kVecaK], Share[2][kVecaK];
  Point X[kVecaK];
};
void vecaSerialize(uint8_t* o, const VecaProof& p) {
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < kVecaK; i++, o += 32) memcpy(o, p.E[t][i], 32);
  for (auto rows : {p.Alpha, p.P, p.Share})
    for (int t = 0; t < 2; t++)
      for (int i = 0; i < kVecaK; i++, o += 32) fr_to_be(o, rows[t][i]);
  for (int i = 0; i < kVecaK; i++, o += 64) memcpy(o, p.X[i].b, 64);
}
bool vecaDeserialize(VecaProof& p, const uint8_t* in) {
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < kVecaK; i++, in += 32) memcpy(p.E[t][i], in, 32);
  for (auto rows : {p.Alpha, p.P, p.Share})
    for (int t = 0; t < 2; t++)
      for (int i = 0; i < kVecaK; i++, in += 32) {
        uint64_t w[4];
        hostg1::be_load(w, in);
        if (geq_n(w)) return false;
        rows[t][i] = fr_from_raw(w);
      }
  for (int i = 0; i < kVecaK; i++, in += 64) {
    memcpy(p.X[i].b, in, 64);
    hostg1::Pt t;
    if (hostg1::unmarshal(t, p.X[i].b)) return false;
  }
  return true;
}
// E / D, go/ve-ca/enc.go:7-45: XOR of the 32-byte big-endian paddings
void vecaXor(uint8_t out[32], const Fr& key, const uint8_t val[32]) {
  uint8_t k[32];
  fr_to_be(k, key);
  for (int i = 0; i < 32; i++) out[i] = k[i] ^ val[i];
}
// rho0, rho1 (main.go:44-49, :95-98) and c (main.go:66-73, :100)
void vecaChallenges(const Point& GenG, const Point& GenH, const Point& C, const VecaProof& p, Fr rho[2], uint8_t c_be[32]) {
  ByteVec pre;
  pre.insert(pre.end(), GenG.b, GenG.b + 64);
  pre.insert(pre.end(), GenH.b, GenH.b + 64);
  pre.insert(pre.end(), C.b, C.b + 64);
  for (int i = 0; i < kVecaK; i++) pre.insert(pre.end(), p.X[i].b, p.X[i].b + 64);
  for (int b = 0; b < 2; b++) {
    ByteVec d = pre;
    d.push_back((uint8_t)b);
    rho[b] = hashModOrder(d);
  }
  ByteVec d;
  appendFr(d, rho[0]);
  appendFr(d, rho[1]);
  for (int t = 0; t < 4; t++)
    for (int i = 0; i < kVecaK; i++) appendBig(d, p.E[t][i]);
  fr_to_be(c_be, hashModOrder(d));
}
inline int bitOf(const uint8_t be[32], int pos) { return (be[31 - pos / 8] >> (pos % 8)) & 1; }  // FBit
// [a_i * GenH + b_i * GenG] on the device: X[i] of Encrypt (main.go:37-40) and the first two terms of
// Verify's X (main.go:104-107)
std::vector<Point> vecaComBatch(bn256_ctx* ctx, const Point& GenG, const Point& GenH, const Fr* a, const Fr* b, size_t n,
                                const std::vector<Point>* plus = nullptr) {
  ByteVec ab(32 * n), bb(32 * n);
  for (size_t i = 0; i < n; i++) {
    fr_to_be(&ab[32 * i], a[i]);
    fr_to_be(&bb[32 * i], b[i]);
  }
  bn256_bases* h = nullptr;
  chk(bn256_com_batch(ctx, GenH.b, GenG.b, ab.data(), bb.data(), n, &h));
  DevPoints sum(ctx, h);
  if (plus) {  // + rho[b_i] * C, a per-element choice between two points (main.go:108)
    ByteVec raw(64 * n);
    for (size_t i = 0; i < n; i++) memcpy(&raw[64 * i], (*plus)[i].b, 64);
    bn256_bases *ph = nullptr, *oh = nullptr;
    chk(bn256_bases_upload(ctx, raw.data(), n, 0, &ph));
    DevPoints pl(ctx, ph);
    chk(bn256_add_batch(ctx, sum.h, 0, pl.h, 0, n, &oh));
    sum = DevPoints(ctx, oh);
  }
  ByteVec out(64 * n);
  chk(bn256_bases_download(ctx, sum.h, 0, n, out.data()));
  std::vector<Point> r(n);
  for (size_t i = 0; i < n; i++) memcpy(r[i].b, &out[64 * i], 64);
  return r;
}
Point pointFrom(const uint8_t b[64]) {
  Point p;
  memcpy(p.b, b, 64);
  pt_load(p);  // throws on a malformed point, like Unmarshal
  return p;
}
}  // namespace

extern "C" int bp_schnorr_msg(const uint8_t* data, size_t len, uint8_t m_be[32]) {
  if ((!data && len) || !m_be) return BN256_EINVAL;
  ByteVec d(data, data + len);
  fr_to_be(m_be, hashModOrder(d));
  return 0;
}

extern "C" int bp_schnorr_multisig(const uint8_t prv_be[32], const uint8_t r_be[32], const uint8_t pub_common[64],
                                   const uint8_t r_common[64], const uint8_t m_be[32], uint8_t S_be[64]) {
  try {
    if (!prv_be || !r_be || !pub_common || !r_common || !m_be || !S_be) return BN256_EINVAL;
    pointFrom(pub_common);
    pointFrom(r_common);
    schnorrS(S_be, fr_from_be(r_be), schnorrChallenge(m_be, pub_common, r_common), fr_from_be(prv_be));
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    return BP_EFAIL;
  }
}

extern "C" int bp_schnorr_sign_det(bn256_ctx* ctx, uint64_t seed, const uint8_t prv_be[32], const uint8_t pub[64],
                                   const uint8_t G[64], const uint8_t m_be[32], uint8_t R_out[64], uint8_t S_be[64]) {
  try {
    if (!ctx || !prv_be || !pub || !G || !m_be || !R_out || !S_be) return BN256_EINVAL;
    pointFrom(pub);
    ByteStream rd{seed};
    uint8_t rb[32];
    Fr r = rd.randInt(rb);  // R(G), schnorr.go:17-24: rand.Int, then r*G -- on the device, through the ABI
    bn256_bases *g = nullptr, *rg = nullptr;
    chk(bn256_bases_upload(ctx, G, 1, 0, &g));
    DevPoints dg(ctx, g);
    chk(bn256_scalar_mul_batch(ctx, g, 0, 1, rb, 0, &rg));
    DevPoints dr(ctx, rg);
    Point R = dr.at(0);
    memcpy(R_out, R.b, 64);
    schnorrS(S_be, r, schnorrChallenge(m_be, pub, R.b), fr_from_be(prv_be));
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

extern "C" int bp_schnorr_verify_batch(bn256_ctx* ctx, size_t n, const uint8_t* R, const uint8_t* S_be, const uint8_t* pub,
                                       const uint8_t G[64], const uint8_t* m_be, uint8_t* ok) {
  try {
    if (!ctx || !G || (n && (!R || !S_be || !pub || !m_be || !ok))) return BN256_EINVAL;
    if (!n) return 0;
    // p1_i = hash_i * PublicKey_i + R_i ; p2_i = S_i * G ; bytes.Equal(p1.Marshal(), p2.Marshal())   (schnorr.go:70-78)
    ByteVec hb(32 * n), sb(32 * n), zero(32 * n, 0);
    for (size_t i = 0; i < n; i++) {
      fr_to_be(&hb[32 * i], schnorrChallenge(m_be + 32 * i, pub + 64 * i, R + 64 * i));
      fr_to_be(&sb[32 * i], frFromBe64(S_be + 64 * i));  // k*G = (k mod Order)*G
    }
    bn256_bases *hp = nullptr, *hr = nullptr, *t = nullptr, *p1 = nullptr, *p2 = nullptr;
    chk(bn256_bases_upload(ctx, pub, n, 0, &hp));
    DevPoints dp(ctx, hp);
    chk(bn256_bases_upload(ctx, R, n, 0, &hr));
    DevPoints drr(ctx, hr);
    chk(bn256_scalar_mul_batch(ctx, hp, 0, n, hb.data(), 0, &t));
    DevPoints dt(ctx, t);
    chk(bn256_add_batch(ctx, t, 0, hr, 0, n, &p1));
    DevPoints d1(ctx, p1);
    chk(bn256_com_batch(ctx, G, G, sb.data(), zero.data(), n, &p2));
    DevPoints d2(ctx, p2);
    ByteVec a(64 * n), b(64 * n);
    chk(bn256_bases_download(ctx, p1, 0, n, a.data()));
    chk(bn256_bases_download(ctx, p2, 0, n, b.data()));
    for (size_t i = 0; i < n; i++) ok[i] = memcmp(&a[64 * i], &b[64 * i], 64) == 0;
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

// Encrypt, go/ve-ca/main.go:18-92, with the reader injected (draw order: p[0..3][k], x[0..1][k])
extern "C" int bp_veca_encrypt_det(bn256_ctx* ctx, uint64_t seed, const uint8_t v_be[32], const uint8_t r_be[32],
                                   const uint8_t s_r_be[32], const uint8_t g_be[32], const uint8_t GenG_[64],
                                   const uint8_t GenH_[64], uint8_t* proof_out, uint8_t C_out[64]) {
  try {
    if (!ctx || !v_be || !r_be || !s_r_be || !g_be || !GenG_ || !GenH_ || !proof_out || !C_out) return BN256_EINVAL;
    const Point GenG = pointFrom(GenG_), GenH = pointFrom(GenH_);
    const Fr v = fr_from_be(v_be), r = fr_from_be(r_be), s_r = fr_from_be(s_r_be), g = fr_from_be(g_be);
    ByteStream rd{seed};
    std::vector<FrVec> p(4), x(2);
    for (auto& row : p) row = rd.values(kVecaK);
    for (auto& row : x) row = rd.values(kVecaK);
    auto pr = std::make_unique<VecaProof>();
    std::vector<Point> X = vecaComBatch(ctx, GenG, GenH, x[0].data(), x[1].data(), kVecaK);
    for (int i = 0; i < kVecaK; i++) pr->X[i] = X[i];
    const Point C = com(GenH, GenG, v, r);  // GAdd(GMul(v, GenH), GMul(r, GenG)), main.go:42
    Fr alpha[4][kVecaK];
    {
      // rho needs only X and C; E is filled afterwards
      VecaProof& q = *pr;
      memset(q.E, 0, sizeof q.E);
      Fr rho[2];
      uint8_t c_unused[32];
      vecaChallenges(GenG, GenH, C, q, rho, c_unused);
      for (int i = 0; i < kVecaK; i++) {
        alpha[0][i] = sub(x[0][i], mul(rho[0], v));
        alpha[1][i] = sub(x[0][i], mul(rho[1], v));
        alpha[2][i] = sub(x[1][i], mul(rho[0], r));
        alpha[3][i] = sub(x[1][i], mul(rho[1], r));
        for (int t = 0; t < 4; t++) {
          uint8_t ab[32];
          fr_to_be(ab, alpha[t][i]);
          vecaXor(q.E[t][i], frPow(s_r, p[t][i]), ab);  // e = E(s, alpha), s = FPow(s_r, p)
        }
      }
      uint8_t c_be[32];
      vecaChallenges(GenG, GenH, C, q, rho, c_be);
      for (int i = 0; i < kVecaK; i++) {
        const int b = bitOf(c_be, i), d = 1 - b;
        q.Alpha[0][i] = alpha[b][i];
        q.Alpha[1][i] = alpha[b + 2][i];
        q.P[0][i] = p[b][i];
        q.P[1][i] = p[b + 2][i];
        q.Share[0][i] = frPow(g, p[d][i]);
        q.Share[1][i] = frPow(g, p[d + 2][i]);
      }
    }
    vecaSerialize(proof_out, *pr);
    memcpy(C_out, C.b, 64);
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

// Verify, go/ve-ca/main.go:94-133: 0 = true, BP_EVERIFY = false with the message the reference prints
extern "C" int bp_veca_verify(bn256_ctx* ctx, const uint8_t* proof, const uint8_t s_r_be[32], const uint8_t C_[64],
                              const uint8_t GenG_[64], const uint8_t GenH_[64]) {
  try {
    if (!ctx || !proof || !s_r_be || !C_ || !GenG_ || !GenH_) return BN256_EINVAL;
    const Point GenG = pointFrom(GenG_), GenH = pointFrom(GenH_), C = pointFrom(C_);
    const Fr s_r = fr_from_be(s_r_be);
    auto pr = std::make_unique<VecaProof>();
    if (!vecaDeserialize(*pr, proof)) {
      g_err = "malformed proof encoding";
      return BP_EFAIL;
    }
    Fr rho[2];
    uint8_t c_be[32];
    vecaChallenges(GenG, GenH, C, *pr, rho, c_be);
    // the k sigma answers as one batch: alpha0_i * GenH + alpha1_i * GenG on the device, + rho[b_i] * C
    const Point rc[2] = {scalarMult(C, rho[0]), scalarMult(C, rho[1])};
    std::vector<Point> plus(kVecaK);
    for (int i = 0; i < kVecaK; i++) plus[i] = rc[bitOf(c_be, i)];
    std::vector<Point> X = vecaComBatch(ctx, GenG, GenH, pr->Alpha[0], pr->Alpha[1], kVecaK, &plus);
    for (int i = 0; i < kVecaK; i++) {  // the reference checks sigma, e0, e1 per i, in this order
      const int b = bitOf(c_be, i);
      if (!pointEq(X[i], pr->X[i])) {
        g_err = "invalid sigma answer";
        return BP_EVERIFY;
      }
      uint8_t a0[32], a1[32], e0[32], e1[32];
      fr_to_be(a0, pr->Alpha[0][i]);
      fr_to_be(a1, pr->Alpha[1][i]);
      vecaXor(e0, frPow(s_r, pr->P[0][i]), a0);
      vecaXor(e1, frPow(s_r, pr->P[1][i]), a1);
      if (memcmp(e0, pr->E[b][i], 32) != 0) {
        g_err = "invalid e0 answer";
        return BP_EVERIFY;
      }
      if (memcmp(e1, pr->E[b + 2][i], 32) != 0) {
        g_err = "invalid e1 answer";
        return BP_EVERIFY;
      }
    }
    g_err.clear();
    return 0;
  } catch (const std::exception& e) {
    g_err = e.what();
    if (ctx) (void)bn256_ctx_drain(ctx);
    return BP_EFAIL;
  }
}

// Decrypt, go/ve-ca/main.go:135-174 (scalar work and one two-term commitment per attempt: host only)
extern "C" int bp_veca_decrypt(const uint8_t u_be[32], const uint8_t* proof, const uint8_t C_[64], const uint8_t GenG_[64],
                               const uint8_t GenH_[64], uint8_t v_out[32], uint8_t r_out[32]) {
  try {
    if (!u_be || !proof || !C_ || !GenG_ || !GenH_ || !v_out || !r_out) return BN256_EINVAL;
    const Point GenG = pointFrom(GenG_), GenH = pointFrom(GenH_), C = pointFrom(C_);
    const Fr u = fr_from_be(u_be);
    auto pr = std::make_unique<VecaProof>();
    if (!vecaDeserialize(*pr, proof)) {
      g_err = "malformed proof encoding";
      return BP_EFAIL;
    }
    Fr rho[2];
    uint8_t c_be[32];
    vecaChallenges(GenG, GenH, C, *pr, rho, c_be);
    for (int i = 0; i < kVecaK; i++) {
      const int b = bitOf(c_be, i), d = 1 - b;
      uint8_t a0[32], a1[32];
      vecaXor(a0, frPow(pr->Share[0][i], u), pr->E[d][i]);      // D(keyV, E[d][i])
      vecaXor(a1, frPow(pr->Share[1][i], u), pr->E[d + 2][i]);
      const Fr den = fr_inv(sub(rho[b], rho[d]));
      const Fr v = mul(sub(fr_from_be(a0), pr->Alpha[0][i]), den);
      const Fr r = mul(sub(fr_from_be(a1), pr->Alpha[1][i]), den);
      if (pointEq(com(GenH, GenG, v, r), C)) {
        fr_to_be(v_out, v);
        fr_to_be(r_out, r);
        g_err.clear();
        return 0;
      }
    }
    g_err = "Failed to recover";
    return BP_EVERIFY;
  } catch (const std::exception& e) {
    g_err = e.what();
    return BP_EFAIL;
  }
}

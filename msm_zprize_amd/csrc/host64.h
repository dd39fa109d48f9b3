// Host-only field and XYZZ arithmetic on 64-bit limbs, for the sequential tail of an MSM that runs on the
// CPU: the Horner combination of the K window sums (msm-batched-affine.ts:300-322) -- c doublings per window,
// ~250 dependent point doublings, where a CPU core is an order of magnitude faster than a lone GPU wave.
// The device keeps Montgomery residues a * 2^(N*W) mod p (N*W = 392 or 261, fp.h); this file multiplies them
// with 64-bit words and the SAME Montgomery radix: NL = NW/2 word steps plus one partial step of
// N*W - 64*NL bits, so values move between the two representations without conversion.
// Values here are fully reduced, in [0, p).
#pragma once
#include <cstdint>
#include <cstring>

#include "fp.h"

namespace msmz {

template <class F>
struct Fe64 {
  static constexpr int NL = F::NW / 2;
  uint64_t l[NL];
};

template <class F>
struct Host64 {
  static constexpr int NL = F::NW / 2;
  static constexpr int RBITS = F::N * F::W;        // Montgomery radix 2^RBITS
  static constexpr int TAIL = RBITS - 64 * NL;     // bits of the final partial reduction step
  static_assert(TAIL > 0 && TAIL < 64, "radix must exceed the 64-bit limb length by less than one limb");
  using E = Fe64<F>;
  typedef unsigned __int128 u128;

  uint64_t p[NL];
  uint64_t pinv;   // -p^-1 mod 2^64

  Host64() {
    for (int i = 0; i < NL; i++) p[i] = (uint64_t)F::PW[2 * i] | ((uint64_t)F::PW[2 * i + 1] << 32);
    uint64_t inv = 1;   // Newton iteration for p^-1 mod 2^64 (p odd)
    for (int i = 0; i < 6; i++) inv *= 2 - p[0] * inv;
    pinv = (uint64_t)0 - inv;
  }

  static bool geq(const uint64_t* a, const uint64_t* b) {
    for (int i = NL - 1; i >= 0; i--) {
      if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
  }
  static uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {   // returns borrow
    uint64_t br = 0;
    for (int i = 0; i < NL; i++) {
      const u128 d = (u128)a[i] - b[i] - br;
      r[i] = (uint64_t)d;
      br = (uint64_t)(d >> 64) & 1u;
    }
    return br;
  }
  static uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b) {   // returns carry
    uint64_t c = 0;
    for (int i = 0; i < NL; i++) {
      const u128 s = (u128)a[i] + b[i] + c;
      r[i] = (uint64_t)s;
      c = (uint64_t)(s >> 64);
    }
    return c;
  }

  // memory-format words (lazy residue < 2^(32 NW)) -> reduced element
  void load(E& r, const uint32_t* w) const {
    for (int i = 0; i < NL; i++) r.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    while (geq(r.l, p)) sub_n(r.l, r.l, p);
  }
  void store(uint32_t* w, const E& a) const {
    for (int i = 0; i < NL; i++) {
      w[2 * i] = (uint32_t)a.l[i];
      w[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    }
  }
  bool is_zero(const E& a) const {
    uint64_t o = 0;
    for (int i = 0; i < NL; i++) o |= a.l[i];
    return o == 0;
  }
  void add(E& r, const E& a, const E& b) const {
    const uint64_t c = add_n(r.l, a.l, b.l);
    if (c || geq(r.l, p)) sub_n(r.l, r.l, p);
  }
  void sub(E& r, const E& a, const E& b) const {
    if (sub_n(r.l, a.l, b.l)) add_n(r.l, r.l, p);
  }

  // r = a * b / 2^RBITS mod p: word-interleaved (CIOS) Montgomery product, NL full word steps and one partial step of
  // TAIL bits, fixed trip counts throughout (the Horner tail of an MSM is ~2500 of these in a dependent chain).
  void mul(E& r, const E& a, const E& b) const {
    uint64_t t[NL + 2];
    for (int j = 0; j < NL + 2; j++) t[j] = 0;
#if defined(__clang__)
#pragma unroll
#endif
    for (int i = 0; i < NL; i++) {
      // t += a_i * b
      uint64_t c = 0;
      const uint64_t ai = a.l[i];
#if defined(__clang__)
#pragma unroll
#endif
      for (int j = 0; j < NL; j++) {
        const u128 s = (u128)ai * b.l[j] + t[j] + c;
        t[j] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
      {
        const u128 s = (u128)t[NL] + c;
        t[NL] = (uint64_t)s;
        t[NL + 1] = (uint64_t)(s >> 64);
      }
      // t = (t + m p) / 2^64 with m = t_0 * (-p^-1) mod 2^64
      const uint64_t m = t[0] * pinv;
      c = (uint64_t)(((u128)m * p[0] + t[0]) >> 64);
#if defined(__clang__)
#pragma unroll
#endif
      for (int j = 1; j < NL; j++) {
        const u128 s = (u128)m * p[j] + t[j] + c;
        t[j - 1] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
      {
        const u128 s = (u128)t[NL] + c;
        t[NL - 1] = (uint64_t)s;
        t[NL] = t[NL + 1] + (uint64_t)(s >> 64);
      }
    }
    // partial step: clear the low TAIL bits, then shift them out; the result is < 2p
    {
      const uint64_t m = (t[0] * pinv) & (((uint64_t)1 << TAIL) - 1);
      uint64_t c = 0;
#if defined(__clang__)
#pragma unroll
#endif
      for (int j = 0; j < NL; j++) {
        const u128 s = (u128)m * p[j] + t[j] + c;
        t[j] = (uint64_t)s;
        c = (uint64_t)(s >> 64);
      }
      t[NL] += c;
    }
#if defined(__clang__)
#pragma unroll
#endif
    for (int j = 0; j < NL; j++) r.l[j] = (t[j] >> TAIL) | (t[j + 1] << (64 - TAIL));
    if (geq(r.l, p)) sub_n(r.l, r.l, p);
  }
  void sqr(E& r, const E& a) const { mul(r, a, a); }

  // ---- XYZZ points (curve.h formulas), infinity = ZZ == 0
  struct Pt {
    E X, Y, ZZ, ZZZ;
  };
  void load_pt(Pt& q, const uint32_t* rec) const {
    load(q.X, rec);
    load(q.Y, rec + F::NW);
    load(q.ZZ, rec + 2 * F::NW);
    load(q.ZZZ, rec + 3 * F::NW);
  }
  void set_inf(Pt& q) const { memset(&q, 0, sizeof(q)); }
  bool is_inf(const Pt& q) const { return is_zero(q.ZZ); }

  // r = 2 q  (EFD dbl-2008-s-1, a = 0)
  void dbl(Pt& r, const Pt& q) const {
    if (is_inf(q)) {
      r = q;
      return;
    }
    E U, V, W, S, M, t, x3, y3, zz, zzz;
    add(U, q.Y, q.Y);
    sqr(V, U);
    mul(W, U, V);
    mul(S, q.X, V);
    sqr(t, q.X);
    add(M, t, t);
    add(M, M, t);
    mul(zz, V, q.ZZ);
    mul(zzz, W, q.ZZZ);
    sqr(t, M);
    sub(t, t, S);
    sub(x3, t, S);
    sub(t, S, x3);
    mul(y3, M, t);
    mul(t, W, q.Y);
    sub(y3, y3, t);
    r.X = x3;
    r.Y = y3;
    r.ZZ = zz;
    r.ZZZ = zzz;
  }

  // r = a + b  (EFD add-2008-s) with the edge cases of curve.h's xyzz_add
  void add_pt(Pt& r, const Pt& a, const Pt& b) const {
    if (is_inf(a)) {
      r = b;
      return;
    }
    if (is_inf(b)) {
      r = a;
      return;
    }
    E U1, U2, S1, S2, P, R, PP, PPP, Q, t, x3, y3, zz, zzz;
    mul(U1, a.X, b.ZZ);
    mul(U2, b.X, a.ZZ);
    mul(S1, a.Y, b.ZZZ);
    mul(S2, b.Y, a.ZZZ);
    sub(P, U2, U1);
    sub(R, S2, S1);
    if (is_zero(P)) {
      if (is_zero(R)) dbl(r, a); else set_inf(r);
      return;
    }
    sqr(PP, P);
    mul(PPP, P, PP);
    mul(Q, U1, PP);
    sqr(t, R);
    sub(t, t, PPP);
    sub(t, t, Q);
    sub(x3, t, Q);
    sub(t, Q, x3);
    mul(y3, R, t);
    mul(t, S1, PPP);
    sub(y3, y3, t);
    mul(t, a.ZZ, b.ZZ);
    mul(zz, t, PP);
    mul(t, a.ZZZ, b.ZZZ);
    mul(zzz, t, PPP);
    r.X = x3;
    r.Y = y3;
    r.ZZ = zz;
    r.ZZZ = zzz;
  }

  // back to the limb representation of fp.h / curve.h (for the one inversion of the affine conversion)
  void to_xyzz(Xyzz<F>& o, const Pt& q) const {
    uint32_t w[F::NW];
    store(w, q.X);
    fe_unpack<F>(o.X, w);
    store(w, q.Y);
    fe_unpack<F>(o.Y, w);
    store(w, q.ZZ);
    fe_unpack<F>(o.ZZ, w);
    store(w, q.ZZZ);
    fe_unpack<F>(o.ZZZ, w);
  }
};

}  // namespace msmz

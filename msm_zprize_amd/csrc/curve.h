// Curve arithmetic for the MSM kernels and the host-side final sums.
//
// Short Weierstrass, a = 0 (BLS12-377 / BLS12-381 G1, Pallas):
//   * affine points in memory: [x | y], 2*NW words, lazy Montgomery residues in [0,3p);
//     the point at infinity is the all-zero record (x = y = 0 is never on y^2 = x^3 + b, b != 0).
//     (The reference keeps a separate isNonZero flag word: src/curve-affine.ts:20-52.)
//   * affine addition given the inverse of the x-difference -- the core of the batched-affine
//     accumulation (src/wasm/curve.ts:32-58 `addAffine`, src/curve-affine.ts:90-109 `double`).
//   * extended Jacobian "XYZZ" accumulators (X, Y, ZZ, ZZZ), x = X/ZZ, y = Y/ZZZ, for the bucket /
//     window reduction.  The reference uses homogeneous projective coordinates there
//     (src/curve-projective.ts:51-253, EFD add-1998-cmo-2 / dbl-1998-cmo-2); XYZZ (EFD
//     madd-2008-s / add-2008-s / dbl-2008-s-1) does the same job with 10M mixed / 14M full
//     additions.  Infinity is ZZ = 0.  All edge cases (infinity operands, equal points,
//     opposite points) are handled like the reference's `addOrSubtract` (curve-projective.ts:51-160).
//
// Twisted Edwards a = -1 (ed-on-bls12-377): extended coordinates, unified addition
// (src/curve-twisted-edwards.ts:84-165, EFD add-2008-hwcd-3 with k = 2d).
#pragma once
#include "fp.h"

namespace msmz {

// ------------------------------------------------------------------------------------------------
// value in [0,3p) with normalized limbs (straight from fe_unpack / fe_reduce_small): is it 0 mod p ?
template <class F>
MSMZ_HD bool fe_reduced_is_zero(const Fe<F>& r) {
  int32_t d0 = 0, d1 = 0, d2 = 0;
#pragma unroll
  for (int j = 0; j < F::N; j++) {
    d0 |= r.l[j];
    d1 |= r.l[j] ^ F::PL[j];
    d2 |= r.l[j] ^ F::P2[j];
  }
  return d0 == 0 || d1 == 0 || d2 == 0;
}

// lazy value (|v| < 2^4 p) == 0 mod p ?
// Quick exact filter first: v = 0 mod p with |v| < 16 p means v = k p, |k| <= 15, and p = 1 mod 2^W, so the low W bits
// of v (= of limb 0: lazy limbs carry nothing INTO limb 0) are k mod 2^W.  Anything else is non-zero -- which is every
// value but ~2^-23 of them, so the full reduction below practically never runs.
template <class F>
MSMZ_HD bool fe_is_zero(const Fe<F>& a) {
  if constexpr (F::PL[0] == 1) {
    constexpr uint32_t MASK = (1u << F::W) - 1u;
    if ((((uint32_t)a.l[0] + 15u) & MASK) > 30u) return false;
  }
  Fe<F> t = a;
  fe_reduce_small(t);
  return fe_reduced_is_zero(t);
}

// ------------------------------------------------------------------------------------------------ affine
template <class F>
struct Affine {
  Fe<F> x, y;
};

template <class F>
MSMZ_HD bool words_point_is_inf(const uint32_t* w) {
  uint32_t o = 0;
#pragma unroll
  for (int i = 0; i < 2 * F::NW; i++) o |= w[i];
  return o == 0;
}

// (x3, y3) = (x1, y1) + (x2, y2) given inv = 1/(x2 - x1)        [2M + 1S]
//   m = (y2 - y1) * inv ; x3 = m^2 - x1 - x2 ; y3 = m (x1 - x3) - y1      (wasm/curve.ts:44-56)
template <class F>
MSMZ_HD void affine_add_with_inv(Affine<F>& r, const Affine<F>& p1, const Affine<F>& p2, const Fe<F>& inv) {
  Fe<F> dy, m, mm, t;
  fe_sub(dy, p2.y, p1.y);
  fe_mul(m, dy, inv);
  fe_sqr(mm, m);
  fe_sub(t, mm, p1.x);
  fe_sub(r.x, t, p2.x);          // x3: limbs within 3 * 2^W
  fe_sub(t, p1.x, r.x);          // x1 - x3: limbs within 4 * 2^W -> one parallel carry keeps the bound
  fe_carry(t);
  fe_mul(mm, m, t);
  fe_sub(r.y, mm, p1.y);
}

// (x3, y3) = 2 (x1, y1) given inv = 1/(2 y1)                    [2M + 2S]
//   m = 3 x1^2 * inv ; x3 = m^2 - 2 x1 ; y3 = m (x1 - x3) - y1             (curve-affine.ts:90-109)
template <class F>
MSMZ_HD void affine_double_with_inv(Affine<F>& r, const Affine<F>& p1, const Fe<F>& inv) {
  Fe<F> xx, x3, m, mm, t;
  fe_sqr(xx, p1.x);
  fe_add(x3, xx, xx);
  fe_add(x3, x3, xx);            // 3 x^2, limbs within 3 * 2^W
  fe_carry(x3);
  fe_mul(m, x3, inv);
  fe_sqr(mm, m);
  fe_sub(t, mm, p1.x);
  fe_sub(r.x, t, p1.x);
  fe_sub(t, p1.x, r.x);
  fe_carry(t);
  fe_mul(mm, m, t);
  fe_sub(r.y, mm, p1.y);
}

// ------------------------------------------------------------------------------------------------ XYZZ
template <class F>
struct Xyzz {
  Fe<F> X, Y, ZZ, ZZZ;
};

template <class F>
MSMZ_HD void xyzz_set_inf(Xyzz<F>& p) {
  fe_set_const<F>(p.X, F::ONE);
  fe_set_const<F>(p.Y, F::ONE);
  fe_zero(p.ZZ);
  fe_zero(p.ZZZ);
}

template <class F>
MSMZ_HD bool xyzz_is_inf(const Xyzz<F>& p) {
  return fe_is_zero(p.ZZ);
}

template <class F>
MSMZ_HD void xyzz_from_affine(Xyzz<F>& r, const Affine<F>& a) {
  r.X = a.x;
  r.Y = a.y;
  fe_set_const<F>(r.ZZ, F::ONE);
  fe_set_const<F>(r.ZZZ, F::ONE);
}

// keep the coordinates' limbs near-normalized after a chain of lazy adds
template <class F>
MSMZ_HD void xyzz_carry(Xyzz<F>& p) {
  fe_carry(p.X);
  fe_carry(p.Y);
}

// r = 2 * (x, y) for an affine input (EFD mdbl-2008-s-1, a = 0)
template <class F>
MSMZ_HD void xyzz_mdbl(Xyzz<F>& r, const Affine<F>& a) {
  Fe<F> U, V, W, S, M, t;
  fe_add(U, a.y, a.y);
  fe_sqr(V, U);
  fe_mul(W, U, V);
  fe_mul(S, a.x, V);
  fe_sqr(t, a.x);
  fe_add(M, t, t);
  fe_add(M, M, t);
  fe_carry(M);
  fe_sqr(t, M);
  fe_sub(t, t, S);
  fe_sub(r.X, t, S);
  fe_sub(t, S, r.X);
  fe_carry(t);
  fe_mul(S, M, t);
  fe_mul(t, W, a.y);
  fe_sub(r.Y, S, t);
  r.ZZ = V;
  r.ZZZ = W;
  xyzz_carry(r);
}

// r = 2 * p (EFD dbl-2008-s-1, a = 0)                                  [6M + 3S]
template <class F>
MSMZ_HD void xyzz_dbl(Xyzz<F>& r, const Xyzz<F>& p) {
  if (xyzz_is_inf(p)) {
    r = p;
    return;
  }
  Fe<F> U, V, W, S, M, t, zz, zzz;
  fe_add(U, p.Y, p.Y);
  fe_carry(U);
  fe_sqr(V, U);
  fe_mul(W, U, V);
  fe_mul(S, p.X, V);
  fe_sqr(t, p.X);
  fe_add(M, t, t);
  fe_add(M, M, t);
  fe_carry(M);
  fe_mul(zz, V, p.ZZ);
  fe_mul(zzz, W, p.ZZZ);
  fe_sqr(t, M);
  fe_sub(t, t, S);
  fe_sub(r.X, t, S);
  fe_sub(t, S, r.X);
  fe_carry(t);
  fe_mul(S, M, t);
  fe_mul(t, W, p.Y);
  fe_sub(r.Y, S, t);
  r.ZZ = zz;
  r.ZZZ = zzz;
  xyzz_carry(r);
}

// r = p + (x2, y2)  -- mixed addition (EFD madd-2008-s)                 [8M + 2S]
// `a_inf`: the affine operand is the point at infinity.
template <class F>
MSMZ_HD void xyzz_madd(Xyzz<F>& r, const Xyzz<F>& p, const Affine<F>& a, bool a_inf) {
  if (a_inf) {
    r = p;
    return;
  }
  if (xyzz_is_inf(p)) {
    xyzz_from_affine(r, a);
    return;
  }
  Fe<F> U2, S2, P, R, PP, PPP, Q, t, u;
  fe_mul(U2, a.x, p.ZZ);
  fe_mul(S2, a.y, p.ZZZ);
  fe_sub(P, U2, p.X);
  fe_sub(R, S2, p.Y);
  fe_carry(P);
  fe_carry(R);
  if (fe_is_zero(P)) {
    if (fe_is_zero(R)) {
      xyzz_mdbl(r, a);
    } else {
      xyzz_set_inf(r);
    }
    return;
  }
  fe_sqr(PP, P);
  fe_mul(PPP, P, PP);
  fe_mul(Q, p.X, PP);
  fe_sqr(t, R);
  fe_sub(t, t, PPP);
  fe_sub(t, t, Q);
  fe_sub(u, t, Q);       // X3 = R^2 - PPP - 2Q
  fe_carry(u);
  fe_sub(t, Q, u);
  fe_carry(t);
  fe_mul(Q, R, t);       // R (Q - X3)
  fe_mul(t, p.Y, PPP);
  fe_sub(r.Y, Q, t);
  r.X = u;
  fe_mul(t, p.ZZ, PP);
  fe_mul(u, p.ZZZ, PPP);
  r.ZZ = t;
  r.ZZZ = u;
  fe_carry(r.Y);
}

// r = p + q (EFD add-2008-s) with the reference's edge-case handling   [12M + 2S]
template <class F>
MSMZ_HD void xyzz_add(Xyzz<F>& r, const Xyzz<F>& p, const Xyzz<F>& q) {
  if (xyzz_is_inf(p)) {
    r = q;
    return;
  }
  if (xyzz_is_inf(q)) {
    r = p;
    return;
  }
  Fe<F> U1, U2, S1, S2, P, R, PP, PPP, Q, t, u;
  fe_mul(U1, p.X, q.ZZ);
  fe_mul(U2, q.X, p.ZZ);
  fe_mul(S1, p.Y, q.ZZZ);
  fe_mul(S2, q.Y, p.ZZZ);
  fe_sub(P, U2, U1);
  fe_sub(R, S2, S1);
  if (fe_is_zero(P)) {
    if (fe_is_zero(R)) {
      xyzz_dbl(r, p);
    } else {
      xyzz_set_inf(r);
    }
    return;
  }
  fe_sqr(PP, P);
  fe_mul(PPP, P, PP);
  fe_mul(Q, U1, PP);
  fe_sqr(t, R);
  fe_sub(t, t, PPP);
  fe_sub(t, t, Q);
  fe_sub(u, t, Q);       // X3
  fe_carry(u);
  fe_sub(t, Q, u);
  fe_carry(t);
  fe_mul(Q, R, t);
  fe_mul(t, S1, PPP);
  fe_sub(r.Y, Q, t);
  r.X = u;
  fe_mul(t, p.ZZ, q.ZZ);
  fe_mul(u, t, PP);
  fe_mul(t, p.ZZZ, q.ZZZ);
  fe_mul(Q, t, PPP);
  r.ZZ = u;
  r.ZZZ = Q;
  fe_carry(r.Y);
}

template <class F>
MSMZ_HD void xyzz_neg(Xyzz<F>& p) {
  fe_neg(p.Y, p.Y);
}

// XYZZ -> canonical affine words (x | y); returns true for infinity (words zeroed).
// (curve-projective.ts:335-349 `toAffine` + curve-affine.ts:220-233 `toBigint`)
template <class F>
MSMZ_HD bool xyzz_to_affine_canon(uint32_t* w, const Xyzz<F>& p) {
  constexpr int NW = F::NW;
  if (xyzz_is_inf(p)) {
#pragma unroll
    for (int i = 0; i < 2 * NW; i++) w[i] = 0;
    return true;
  }
  Fe<F> zi, zi2, zi3, x, y, t;
  fe_inverse(zi3, p.ZZZ);        // 1/ZZZ
  fe_mul(t, zi3, p.ZZ);          // 1/Z   (ZZ/ZZZ = 1/Z)
  fe_sqr(zi2, t);                // 1/ZZ
  fe_mul(x, p.X, zi2);
  fe_mul(y, p.Y, zi3);
  fe_from_mont(t, x);
  fe_to_canon_words<F>(w, t);
  fe_from_mont(t, y);
  fe_to_canon_words<F>(w + NW, t);
  (void)zi;
  return false;
}

// ------------------------------------------------------------------------------------------------ twisted Edwards
template <class F>
struct TeExt {
  Fe<F> X, Y, Z, T;
};

// "Niels" form of an affine input: (y - x, y + x, 2d*x*y) -- lets the mixed addition skip 2 products.
template <class F>
struct TeNiels {
  Fe<F> ym, yp, kt;
};

template <class F>
MSMZ_HD void te_set_zero(TeExt<F>& p) {
  fe_zero(p.X);
  fe_set_const<F>(p.Y, F::ONE);
  fe_set_const<F>(p.Z, F::ONE);
  fe_zero(p.T);
}

// r = p + q, unified (curve-twisted-edwards.ts:84-165; bigint/twisted-edwards.ts:52-85)   [9M]
template <class F>
MSMZ_HD void te_add(TeExt<F>& r, const TeExt<F>& p, const TeExt<F>& q) {
  Fe<F> A, B, C, D, E, Fv, G, H, t, u, k;
  fe_sub(t, p.Y, p.X);
  fe_sub(u, q.Y, q.X);
  fe_carry(t);
  fe_carry(u);
  fe_mul(A, t, u);
  fe_add(t, p.Y, p.X);
  fe_add(u, q.Y, q.X);
  fe_carry(t);
  fe_carry(u);
  fe_mul(B, t, u);
  fe_mul(t, p.T, q.T);
  fe_set_const<F>(k, F::K2D);
  fe_mul(C, t, k);
  fe_mul(D, p.Z, q.Z);
  fe_add(D, D, D);
  fe_sub(E, B, A);
  fe_sub(Fv, D, C);
  fe_add(G, D, C);
  fe_add(H, B, A);
  fe_carry(E);
  fe_carry(Fv);
  fe_carry(G);
  fe_carry(H);
  fe_mul(r.X, E, Fv);
  fe_mul(r.Y, G, H);
  fe_mul(r.T, E, H);
  fe_mul(r.Z, Fv, G);
}

// r = p + (+-)n for a Niels-form affine input (negation swaps ym/yp and flips kt)        [7M]
template <class F>
MSMZ_HD void te_madd(TeExt<F>& r, const TeExt<F>& p, const TeNiels<F>& n, uint32_t neg) {
  Fe<F> A, B, C, D, E, Fv, G, H, t, ym, yp, kt;
  int32_t m = -(int32_t)(neg & 1u);
#pragma unroll
  for (int j = 0; j < F::N; j++) {
    ym.l[j] = (n.ym.l[j] & ~m) | (n.yp.l[j] & m);
    yp.l[j] = (n.yp.l[j] & ~m) | (n.ym.l[j] & m);
  }
  fe_cneg(kt, n.kt, neg);
  fe_sub(t, p.Y, p.X);
  fe_carry(t);
  fe_mul(A, t, ym);
  fe_add(t, p.Y, p.X);
  fe_carry(t);
  fe_mul(B, t, yp);
  fe_mul(C, p.T, kt);
  fe_add(D, p.Z, p.Z);
  fe_sub(E, B, A);
  fe_sub(Fv, D, C);
  fe_add(G, D, C);
  fe_add(H, B, A);
  fe_carry(E);
  fe_carry(Fv);
  fe_carry(G);
  fe_carry(H);
  fe_mul(r.X, E, Fv);
  fe_mul(r.Y, G, H);
  fe_mul(r.T, E, H);
  fe_mul(r.Z, Fv, G);
}

// extended -> canonical affine words (x | y)   (bigint/twisted-edwards.ts:39-45)
template <class F>
MSMZ_HD void te_to_affine_canon(uint32_t* w, const TeExt<F>& p) {
  Fe<F> zi, x, y, t;
  fe_inverse(zi, p.Z);
  fe_mul(x, p.X, zi);
  fe_mul(y, p.Y, zi);
  fe_from_mont(t, x);
  fe_to_canon_words<F>(w, t);
  fe_from_mont(t, y);
  fe_to_canon_words<F>(w + F::NW, t);
}

}  // namespace msmz

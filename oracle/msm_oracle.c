/* CPU oracle for the MSM hot path in plain C -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * A restatement of the reference's big-integer layer, used as the parity checker at sizes Python is
 * too slow for and as the `cpu_baseline` ("port") of bench.py.  It follows
 *   src/bigint/msm.ts:8-53                     naive Pippenger: unsigned c-bit windows, c = max(log2 N - 1, 1),
 *                                              running-sum bucket reduction, Horner over windows
 *   src/bigint/projective-weierstrass.ts:33-115 homogeneous projective add (add-1998-cmo-2) / double
 *                                              (dbl-1998-cmo-2) with the zero / equal / opposite cases
 *   src/bigint/twisted-edwards.ts:52-94        extended unified addition (add-2008-hwcd-3, k = 2d)
 *   src/bigint/field.ts:32-56, 117-122         field ops (here: 64-bit-limb Montgomery form internally,
 *                                              which is invisible at the interface: inputs and outputs
 *                                              are canonical little-endian integers)
 * It shares no code with the product (msm_zprize_amd/): different limb width (64-bit CIOS),
 * different coordinates (homogeneous projective), different window scheme (unsigned, no GLV).
 *
 * Parity status: pinned against the reference's fixed vectors through tests/test_oracle.py
 * (C oracle == Python oracle == known-answer points).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o oracle/libmsm_oracle.so oracle/msm_oracle.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXL 6
typedef unsigned __int128 u128;

typedef struct {
  int n;               /* 64-bit limbs: 4 or 6 */
  uint64_t p[MAXL];    /* modulus */
  uint64_t mu;         /* -p^-1 mod 2^64 */
  uint64_t one[MAXL];  /* R mod p */
  uint64_t r2[MAXL];   /* R^2 mod p */
} field_t;

typedef struct { uint64_t v[MAXL]; } fe;

static int geq(const uint64_t* a, const uint64_t* b, int n) {
  for (int i = n - 1; i >= 0; i--) {
    if (a[i] > b[i]) return 1;
    if (a[i] < b[i]) return 0;
  }
  return 1;
}
static uint64_t add_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  u128 c = 0;
  for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static uint64_t sub_n(uint64_t* r, const uint64_t* a, const uint64_t* b, int n) {
  uint64_t br = 0;
  for (int i = 0; i < n; i++) {
    u128 t = (u128)a[i] - b[i] - br;
    r[i] = (uint64_t)t; br = (uint64_t)(t >> 64) & 1;
  }
  return br;
}
static void f_add(const field_t* F, fe* r, const fe* a, const fe* b) {   /* field.ts:32-35 */
  uint64_t c = add_n(r->v, a->v, b->v, F->n);
  if (c || geq(r->v, F->p, F->n)) sub_n(r->v, r->v, F->p, F->n);
}
static void f_sub(const field_t* F, fe* r, const fe* a, const fe* b) {   /* field.ts:36-39 */
  if (sub_n(r->v, a->v, b->v, F->n)) add_n(r->v, r->v, F->p, F->n);
}
static int f_is_zero(const field_t* F, const fe* a) {
  uint64_t o = 0; for (int i = 0; i < F->n; i++) o |= a->v[i]; return o == 0;
}
/* Montgomery product a*b/R, CIOS with 64-bit limbs */
static void f_mul(const field_t* F, fe* r, const fe* a, const fe* b) {
  int n = F->n; uint64_t t[MAXL + 2];
  memset(t, 0, sizeof(t));
  for (int i = 0; i < n; i++) {
    u128 c = 0;
    for (int j = 0; j < n; j++) { c += (u128)a->v[i] * b->v[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[n]; t[n] = (uint64_t)c; t[n + 1] = (uint64_t)(c >> 64);
    uint64_t m = t[0] * F->mu;
    c = (u128)m * F->p[0] + t[0]; c >>= 64;
    for (int j = 1; j < n; j++) { c += (u128)m * F->p[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[n]; t[n - 1] = (uint64_t)c; t[n] = t[n + 1] + (uint64_t)(c >> 64);
  }
  if (t[n] || geq(t, F->p, n)) sub_n(t, t, F->p, n);
  memcpy(r->v, t, n * 8); for (int i = n; i < MAXL; i++) r->v[i] = 0;
}
static void f_from_int(const field_t* F, fe* r, uint64_t x) {
  fe t; memset(&t, 0, sizeof(t)); t.v[0] = x; fe r2; memcpy(r2.v, F->r2, sizeof(r2.v)); f_mul(F, r, &t, &r2);
}
static void f_to_mont(const field_t* F, fe* r, const fe* a) { fe r2; memcpy(r2.v, F->r2, sizeof(r2.v)); f_mul(F, r, a, &r2); }
static void f_from_mont(const field_t* F, fe* r, const fe* a) { fe o; memset(&o, 0, sizeof(o)); o.v[0] = 1; f_mul(F, r, a, &o); }
/* inverse by Fermat: a^(p-2)   (field.ts:117-122 uses EGCD; same function) */
static void f_inv(const field_t* F, fe* r, const fe* a) {
  uint64_t e[MAXL]; memcpy(e, F->p, sizeof(e));
  uint64_t two[MAXL] = {2, 0, 0, 0, 0, 0}; sub_n(e, e, two, F->n);
  fe acc; memcpy(acc.v, F->one, sizeof(acc.v)); fe base = *a;
  for (int i = 0; i < 64 * F->n; i++) {
    if ((e[i >> 6] >> (i & 63)) & 1) f_mul(F, &acc, &acc, &base);
    f_mul(F, &base, &base, &base);
  }
  *r = acc;
}
static void field_init(field_t* F, const uint8_t* p_le, int n) {
  memset(F, 0, sizeof(*F)); F->n = n; memcpy(F->p, p_le, 8 * n);
  uint64_t inv = 1; for (int i = 0; i < 6; i++) inv *= 2 - F->p[0] * inv;   /* Newton: p^-1 mod 2^64 */
  F->mu = (uint64_t)0 - inv;
  /* R mod p and R^2 mod p by repeated doubling */
  uint64_t x[MAXL]; memset(x, 0, sizeof(x)); x[0] = 1;
  for (int i = 0; i < 2 * 64 * n; i++) {
    uint64_t c = add_n(x, x, x, n);
    if (c || geq(x, F->p, n)) sub_n(x, x, F->p, n);
    if (i == 64 * n - 1) memcpy(F->one, x, sizeof(x));
  }
  memcpy(F->r2, x, sizeof(x));
}

/* ------------------------------------------------------------------ curves (generic point = 4 coords) */
typedef struct { fe X, Y, Z, T; } pt;   /* Weierstrass: (X:Y:Z), T unused; twisted Edwards: (X,Y,Z,T) */
typedef struct { field_t F; int kind; /* 0 = weierstrass a=0, 1 = twisted edwards a=-1 */ fe k2d; } curve_t;

static void w_zero(const curve_t* C, pt* r) { memset(r, 0, sizeof(*r)); memcpy(r->Y.v, C->F.one, sizeof(r->Y.v)); }
static void w_double(const curve_t* C, pt* r, const pt* P) {   /* projective-weierstrass.ts:90-115 */
  const field_t* F = &C->F;
  if (f_is_zero(F, &P->Z)) { w_zero(C, r); return; }
  fe w, s, ss, sss, R, B, h, t, u, X3, Y3, Z3;
  f_mul(F, &t, &P->X, &P->X); f_add(F, &w, &t, &t); f_add(F, &w, &w, &t);          /* w = 3 X^2 */
  f_mul(F, &s, &P->Y, &P->Z); f_mul(F, &ss, &s, &s); f_mul(F, &sss, &s, &ss);
  f_mul(F, &R, &P->Y, &s); f_mul(F, &B, &P->X, &R);
  f_mul(F, &h, &w, &w); f_add(F, &t, &B, &B); f_add(F, &t, &t, &t); f_add(F, &u, &t, &t); f_sub(F, &h, &h, &u); /* h = w^2 - 8B */
  f_mul(F, &X3, &h, &s); f_add(F, &X3, &X3, &X3);                                   /* X3 = 2 h s */
  f_sub(F, &u, &t, &h); f_mul(F, &Y3, &w, &u);                                      /* w (4B - h) */
  f_mul(F, &u, &R, &R); f_add(F, &u, &u, &u); f_add(F, &u, &u, &u); f_add(F, &u, &u, &u); f_sub(F, &Y3, &Y3, &u); /* - 8 R^2 */
  f_add(F, &Z3, &sss, &sss); f_add(F, &Z3, &Z3, &Z3); f_add(F, &Z3, &Z3, &Z3);      /* 8 sss */
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void w_add(const curve_t* C, pt* r, const pt* P1, const pt* P2) {   /* projective-weierstrass.ts:33-85 */
  const field_t* F = &C->F;
  if (f_is_zero(F, &P1->Z)) { *r = *P2; return; }
  if (f_is_zero(F, &P2->Z)) { *r = *P1; return; }
  fe Y1Z2, X1Z2, Z1Z2, u, uu, v, vv, vvv, R, A, t, X3, Y3, Z3;
  f_mul(F, &Y1Z2, &P1->Y, &P2->Z); f_mul(F, &X1Z2, &P1->X, &P2->Z); f_mul(F, &Z1Z2, &P1->Z, &P2->Z);
  f_mul(F, &t, &P2->Y, &P1->Z); f_sub(F, &u, &t, &Y1Z2);
  f_mul(F, &t, &P2->X, &P1->Z); f_sub(F, &v, &t, &X1Z2);
  if (f_is_zero(F, &v)) {
    if (f_is_zero(F, &u)) { w_double(C, r, P1); return; }
    w_zero(C, r); return;
  }
  f_mul(F, &uu, &u, &u); f_mul(F, &vv, &v, &v); f_mul(F, &vvv, &v, &vv); f_mul(F, &R, &vv, &X1Z2);
  f_mul(F, &A, &uu, &Z1Z2); f_sub(F, &A, &A, &vvv); f_sub(F, &A, &A, &R); f_sub(F, &A, &A, &R);
  f_mul(F, &X3, &v, &A);
  f_sub(F, &t, &R, &A); f_mul(F, &Y3, &u, &t); f_mul(F, &t, &vvv, &Y1Z2); f_sub(F, &Y3, &Y3, &t);
  f_mul(F, &Z3, &vvv, &Z1Z2);
  r->X = X3; r->Y = Y3; r->Z = Z3;
}
static void te_zero(const curve_t* C, pt* r) {
  memset(r, 0, sizeof(*r)); memcpy(r->Y.v, C->F.one, sizeof(r->Y.v)); memcpy(r->Z.v, C->F.one, sizeof(r->Z.v));
}
static void te_add(const curve_t* C, pt* r, const pt* P1, const pt* P2) {   /* twisted-edwards.ts:52-85 */
  const field_t* F = &C->F;
  fe A, B, Cc, D, E, Fv, G, H, t, u;
  f_sub(F, &t, &P1->Y, &P1->X); f_sub(F, &u, &P2->Y, &P2->X); f_mul(F, &A, &t, &u);
  f_add(F, &t, &P1->Y, &P1->X); f_add(F, &u, &P2->Y, &P2->X); f_mul(F, &B, &t, &u);
  f_mul(F, &t, &P1->T, &P2->T); f_mul(F, &Cc, &t, &C->k2d);
  f_mul(F, &D, &P1->Z, &P2->Z); f_add(F, &D, &D, &D);
  f_sub(F, &E, &B, &A); f_sub(F, &Fv, &D, &Cc); f_add(F, &G, &D, &Cc); f_add(F, &H, &B, &A);
  f_mul(F, &r->X, &E, &Fv); f_mul(F, &r->Y, &G, &H); f_mul(F, &r->T, &E, &H); f_mul(F, &r->Z, &Fv, &G);
}
static void c_zero(const curve_t* C, pt* r) { if (C->kind) te_zero(C, r); else w_zero(C, r); }
static void c_add(const curve_t* C, pt* r, const pt* a, const pt* b) { pt t; if (C->kind) te_add(C, &t, a, b); else w_add(C, &t, a, b); *r = t; }
static void c_double(const curve_t* C, pt* r, const pt* a) { pt t; if (C->kind) te_add(C, &t, a, a); else w_double(C, &t, a); *r = t; }

static void curve_init(curve_t* C, int kind, const uint8_t* p_le, int nlimbs, uint64_t d) {
  field_init(&C->F, p_le, nlimbs); C->kind = kind;
  f_from_int(&C->F, &C->k2d, 2 * d);
}
/* canonical affine bytes (x || y) -> internal point */
static void load_point(const curve_t* C, pt* P, const uint8_t* xy, int is_inf) {
  const field_t* F = &C->F; int nb = 8 * F->n;
  if (is_inf) { c_zero(C, P); return; }
  fe x, y; memset(&x, 0, sizeof(x)); memset(&y, 0, sizeof(y));
  memcpy(x.v, xy, nb); memcpy(y.v, xy + nb, nb);
  memset(P, 0, sizeof(*P));
  f_to_mont(F, &P->X, &x); f_to_mont(F, &P->Y, &y); memcpy(P->Z.v, F->one, sizeof(P->Z.v));
  if (C->kind) f_mul(F, &P->T, &P->X, &P->Y);
}
/* internal point -> canonical affine bytes; returns 1 for the Weierstrass point at infinity */
static int store_point(const curve_t* C, uint8_t* xy, const pt* P) {
  const field_t* F = &C->F; int nb = 8 * F->n;
  if (!C->kind && f_is_zero(F, &P->Z)) { memset(xy, 0, 2 * nb); return 1; }
  fe zi, x, y; f_inv(F, &zi, &P->Z); f_mul(F, &x, &P->X, &zi); f_mul(F, &y, &P->Y, &zi);
  f_from_mont(F, &x, &x); f_from_mont(F, &y, &y);
  memcpy(xy, x.v, nb); memcpy(xy + nb, y.v, nb);
  return 0;
}
static uint32_t window(const uint8_t* s32, int pos, int c) {
  uint64_t v = 0; int byte = pos >> 3;
  for (int i = 0; i < 5 && byte + i < 32; i++) v |= (uint64_t)s32[byte + i] << (8 * i);
  return (uint32_t)((v >> (pos & 7)) & ((1u << c) - 1));
}
static int log2ceil(uint64_t n) { int r = 0; while (((uint64_t)1 << r) < n) r++; return r; }   /* util.ts:163-167 */

/* bigint/msm.ts:8-53.  kind: 0 Weierstrass (a=0), 1 twisted Edwards (a=-1, coefficient d).
 * scalar_bits = Curve.Scalar.sizeInBits.  n_adds (nullable) = number of non-zero digits. */
int oracle_msm(int kind, const uint8_t* p_le, int nlimbs, uint64_t d, int scalar_bits, const uint8_t* scalars_le32,
               const uint8_t* points_xy, const uint8_t* is_inf, uint64_t n, uint8_t* out_xy, int* out_is_inf,
               int threads, uint64_t* n_adds) {
  if (nlimbs != 4 && nlimbs != 6) return 1;
  curve_t C; curve_init(&C, kind, p_le, nlimbs, d);
  int nb = 8 * nlimbs;
  int c = log2ceil(n) - 1; if (c < 1) c = 1;
  if (c > 24) return 1;
  int K = (scalar_bits + c - 1) / c;
  uint64_t L = (uint64_t)1 << c;
  pt* pts = (pt*)malloc(sizeof(pt) * n);
  pt* part = (pt*)malloc(sizeof(pt) * K);
  if (!pts || !part) return 2;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; i++) load_point(&C, &pts[i], points_xy + (size_t)i * 2 * nb, is_inf ? is_inf[i] : 0);
  uint64_t adds = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : adds)
  for (int k = 0; k < K; k++) {
    pt* buckets = (pt*)malloc(sizeof(pt) * (L - 1));
    for (uint64_t l = 0; l < L - 1; l++) c_zero(&C, &buckets[l]);
    for (uint64_t i = 0; i < n; i++) {
      uint32_t l = window(scalars_le32 + 32 * i, k * c, c);
      if (l == 0) continue;
      c_add(&C, &buckets[l - 1], &buckets[l - 1], &pts[i]);
      adds++;
    }
    pt running, triangle; c_zero(&C, &running); c_zero(&C, &triangle);
    for (int64_t l = (int64_t)L - 2; l >= 0; l--) {
      c_add(&C, &running, &running, &buckets[l]);
      c_add(&C, &triangle, &triangle, &running);
    }
    part[k] = triangle;
    free(buckets);
  }
  pt result = part[K - 1];
  for (int k = K - 2; k >= 0; k--) {
    for (int i = 0; i < c; i++) c_double(&C, &result, &result);
    c_add(&C, &result, &result, &part[k]);
  }
  *out_is_inf = store_point(&C, out_xy, &result);
  if (n_adds) *n_adds = adds;
  free(pts); free(part);
  return 0;
}

/* The same algorithm on `shards` contiguous index ranges at once (each range with its own window size
 * c = log2(range) - 1, its own buckets and its own Horner step), all (range, window) pairs as one pool of OpenMP
 * tasks, the range results added at the end: MSM(A u B) = MSM(A) + MSM(B).  bigint/msm.ts parallelizes over
 * nothing; oracle_msm above over its ~14 windows; this form exists so that bench.py's CPU baseline can occupy all
 * the cores it reports. */
int oracle_msm_sharded(int kind, const uint8_t* p_le, int nlimbs, uint64_t d, int scalar_bits, const uint8_t* scalars_le32,
                       const uint8_t* points_xy, const uint8_t* is_inf, uint64_t n, int shards, uint8_t* out_xy,
                       int* out_is_inf, int threads, uint64_t* n_adds) {
  if (nlimbs != 4 && nlimbs != 6) return 1;
  if (shards < 1) shards = 1;
  if ((uint64_t)shards > n) shards = (int)n;
  curve_t C; curve_init(&C, kind, p_le, nlimbs, d);
  int nb = 8 * nlimbs;
  uint64_t per = (n + shards - 1) / shards;
  int c = log2ceil(per) - 1; if (c < 1) c = 1;
  if (c > 24) return 1;
  int K = (scalar_bits + c - 1) / c;
  uint64_t L = (uint64_t)1 << c;
  pt* pts = (pt*)malloc(sizeof(pt) * n);
  pt* part = (pt*)malloc(sizeof(pt) * (size_t)K * shards);
  if (!pts || !part) return 2;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; i++) load_point(&C, &pts[i], points_xy + (size_t)i * 2 * nb, is_inf ? is_inf[i] : 0);
  uint64_t adds = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : adds)
  for (int task = 0; task < K * shards; task++) {
    const int sh = task / K, k = task % K;
    const uint64_t i0 = (uint64_t)sh * per, i1 = i0 + per < n ? i0 + per : n;
    pt* buckets = (pt*)malloc(sizeof(pt) * (L - 1));
    for (uint64_t l = 0; l < L - 1; l++) c_zero(&C, &buckets[l]);
    for (uint64_t i = i0; i < i1; i++) {
      uint32_t l = window(scalars_le32 + 32 * i, k * c, c);
      if (l == 0) continue;
      c_add(&C, &buckets[l - 1], &buckets[l - 1], &pts[i]);
      adds++;
    }
    pt running, triangle; c_zero(&C, &running); c_zero(&C, &triangle);
    for (int64_t l = (int64_t)L - 2; l >= 0; l--) {
      c_add(&C, &running, &running, &buckets[l]);
      c_add(&C, &triangle, &triangle, &running);
    }
    part[task] = triangle;
    free(buckets);
  }
  pt total; c_zero(&C, &total);
  for (int sh = 0; sh < shards; sh++) {
    pt result = part[(size_t)sh * K + K - 1];
    for (int k = K - 2; k >= 0; k--) {
      for (int i = 0; i < c; i++) c_double(&C, &result, &result);
      c_add(&C, &result, &result, &part[(size_t)sh * K + k]);
    }
    c_add(&C, &total, &total, &result);
  }
  *out_is_inf = store_point(&C, out_xy, &total);
  if (n_adds) *n_adds = adds;
  free(pts); free(part);
  return 0;
}

/* s * P by MSB-first double-and-add (affine-weierstrass.ts:111-119 / twisted-edwards.ts:129-137) */
int oracle_scale(int kind, const uint8_t* p_le, int nlimbs, uint64_t d, const uint8_t* scalar_le32, const uint8_t* point_xy,
                 int is_inf, uint8_t* out_xy, int* out_is_inf) {
  if (nlimbs != 4 && nlimbs != 6) return 1;
  curve_t C; curve_init(&C, kind, p_le, nlimbs, d);
  pt P, Q; load_point(&C, &P, point_xy, is_inf); c_zero(&C, &Q);
  for (int i = 255; i >= 0; i--) {
    c_double(&C, &Q, &Q);
    if ((scalar_le32[i >> 3] >> (i & 7)) & 1) c_add(&C, &Q, &Q, &P);
  }
  *out_is_inf = store_point(&C, out_xy, &Q);
  return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

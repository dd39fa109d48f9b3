// The host's 64-bit-limb field / XYZZ arithmetic (csrc/host64.h, used for the final Horner step) against the
// 28/29-bit limb templates the kernels use (fp.h, curve.h): products, sums, differences on random residues and a
// chain of doublings / additions (incl. the equal-operands branch), compared on canonical words.  Exit code =
// number of mismatches.
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <random>
#include "../../msm_zprize_amd/csrc/constants_gen.h"
#include "../../msm_zprize_amd/csrc/fp.h"
#include "../../msm_zprize_amd/csrc/curve.h"
#include "../../msm_zprize_amd/csrc/host64.h"
using namespace msmz;
template <class F> int check(const char* name) {
  Host64<F> H;
  std::mt19937_64 rng(7);
  int bad = 0;
  for (int it = 0; it < 2000; it++) {
    uint32_t wa[F::NW], wb[F::NW];
    for (int i = 0; i < F::NW; i++) { wa[i] = (uint32_t)rng(); wb[i] = (uint32_t)rng(); }
    // random values below p: clear top bits
    int topbits = F::BITS - 32 * (F::NW - 1);
    wa[F::NW - 1] &= (1u << (topbits - 1)) - 1; wb[F::NW - 1] &= (1u << (topbits - 1)) - 1;
    Fe<F> a, b, c; fe_unpack<F>(a, wa); fe_unpack<F>(b, wb);
    fe_mul(c, a, b);
    uint32_t wc[F::NW]; fe_to_canon_words<F>(wc, c);
    Fe64<F> ha, hb, hc; H.load(ha, wa); H.load(hb, wb); H.mul(hc, ha, hb);
    uint32_t wh[F::NW]; H.store(wh, hc);
    if (memcmp(wc, wh, sizeof(wc)) != 0) bad++;
    // add / sub
    Fe<F> s; fe_add(s, a, b); fe_to_canon_words<F>(wc, s); H.add(hc, ha, hb); H.store(wh, hc); if (memcmp(wc, wh, sizeof(wc))) bad++;
    fe_sub(s, a, b); fe_to_canon_words<F>(wc, s); H.sub(hc, ha, hb); H.store(wh, hc); if (memcmp(wc, wh, sizeof(wc))) bad++;
  }
  // point ops: k*G by doublings/additions in both representations
  Affine<F> g; fe_set_const<F>(g.x, F::GX); fe_set_const<F>(g.y, F::GY);
  Xyzz<F> p, t, q; xyzz_from_affine(p, g); q = p;
  uint32_t rec[4 * F::NW];
  fe_store<F>(rec, p.X); fe_store<F>(rec + F::NW, p.Y); fe_store<F>(rec + 2 * F::NW, p.ZZ); fe_store<F>(rec + 3 * F::NW, p.ZZZ);
  typename Host64<F>::Pt hp, hq, ht; H.load_pt(hp, rec); hq = hp;
  for (int i = 0; i < 40; i++) {
    xyzz_dbl(t, p); p = t; H.dbl(ht, hp); hp = ht;
    if (i % 3 == 0) { xyzz_add(t, p, q); p = t; H.add_pt(ht, hp, hq); hp = ht; }
    if (i % 7 == 0) { xyzz_add(t, p, p); p = t; H.add_pt(ht, hp, hp); hp = ht; }   // equal operands -> doubling branch
  }
  uint32_t w1[2 * F::NW], w2[2 * F::NW];
  xyzz_to_affine_canon<F>(w1, p);
  Xyzz<F> back; H.to_xyzz(back, hp); xyzz_to_affine_canon<F>(w2, back);
  if (memcmp(w1, w2, sizeof(w1))) bad += 1000;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 100000; i++) { H.dbl(ht, hp); hp = ht; }
  auto t1 = std::chrono::steady_clock::now();
  printf("%s: mismatches %d; Host64 dbl %.3f us (%llu)\n", name, bad, std::chrono::duration<double, std::micro>(t1 - t0).count() / 1e5, (unsigned long long)hp.X.l[0]);
  return bad;
}
int main() { return check<Bls377Fp>("bls377") + check<PallasFp>("pallas") + check<Bls381Fp>("bls381"); }

// Host-side driver for msm_zprize_amd/csrc/fp.h (same templates the kernels use), driven by
// tests/test_fp_host.py through stdin/stdout: hex in, hex out.  No GPU needed.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <iostream>
#include "../../msm_zprize_amd/csrc/constants_gen.h"
#include "../../msm_zprize_amd/csrc/fp.h"
#include "../../msm_zprize_amd/csrc/scalar.h"
using namespace msmz;

template <int NW> static void parse(const std::string& h, uint32_t* w) {
  std::string s(NW * 8 - h.size(), '0'); s += h;
  for (int i = 0; i < NW; i++) w[i] = (uint32_t)strtoul(s.substr((NW - 1 - i) * 8, 8).c_str(), nullptr, 16);
}
template <int NW> static void print(const uint32_t* w) {
  for (int i = NW - 1; i >= 0; i--) printf("%08x", w[i]);
  printf("\n");
}

template <class F> static void run(const std::string& op, const std::vector<std::string>& a) {
  constexpr int NW = F::NW;
  uint32_t w0[NW], w1[NW], out[NW];
  Fe<F> x, y, r;
  parse<NW>(a[0], w0); fe_unpack<F>(x, w0);
  if (a.size() > 1) { parse<NW>(a[1], w1); fe_unpack<F>(y, w1); }
  if (op == "mul") { fe_mul<F>(r, x, y); fe_to_canon_words<F>(out, r); }
  else if (op == "sqr") { fe_sqr<F>(r, x); fe_to_canon_words<F>(out, r); }
  else if (op == "mul_lazy") {  // (x - y) * (x + y) with lazy operands, then (..)*x - y*? chain
    Fe<F> d, s; fe_sub<F>(d, x, y); fe_add<F>(s, x, y); fe_mul<F>(r, d, s); fe_to_canon_words<F>(out, r); }
  else if (op == "chain") {     // r = ((x*y - x - y) * x) - y : exercises negative lazy values
    Fe<F> t; fe_mul<F>(t, x, y); fe_sub<F>(t, t, x); fe_sub<F>(t, t, y); fe_mul<F>(r, t, x); fe_sub<F>(r, r, y);
    fe_to_canon_words<F>(out, r); }
  else if (op == "store") { Fe<F> t; fe_sub<F>(t, x, y); fe_sub<F>(t, t, y); fe_sub<F>(t, t, y); fe_store<F>(out, t); } // x - 3y in [0,3p)
  else if (op == "store_mulout") { fe_mul<F>(r, x, y); fe_store_mulout<F>(out, r); }
  else if (op == "inv") { bool ok = fe_inverse<F>(r, x); if (!ok) { printf("ZERO\n"); return; } fe_to_canon_words<F>(out, r); }
  else if (op == "tomont") { fe_to_mont<F>(r, x); fe_to_canon_words<F>(out, r); }
  else if (op == "frommont") { fe_from_mont<F>(r, x); fe_to_canon_words<F>(out, r); }
  else if (op == "canon") { fe_to_canon_words<F>(out, x); }
  else { printf("ERR\n"); return; }
  print<NW>(out);
}

template <class Fr> static void run_glv(const std::string& h) {
  uint32_t s[8], s0[4], s1[4], n0, n1;
  parse<8>(h, s);
  glv_decompose<Fr>(s0, s1, n0, n1, s);
  printf("%u:", n0); for (int i = 3; i >= 0; i--) printf("%08x", s0[i]);
  printf(":%u:", n1); for (int i = 3; i >= 0; i--) printf("%08x", s1[i]);
  printf("\n");
}

int main() {
  std::string field, op; int nargs;
  while (std::cin >> field >> op >> nargs) {
    std::vector<std::string> a(nargs);
    for (auto& s : a) std::cin >> s;
    if (op == "glv") {
      if (field == "bls377") run_glv<Bls377Fr>(a[0]);
      else if (field == "bls381") run_glv<Bls381Fr>(a[0]);
      else if (field == "pallas") run_glv<PallasFr>(a[0]);
      else printf("ERR\n");
      continue;
    }
    if (field == "bls377") run<Bls377Fp>(op, a);
    else if (field == "bls381") run<Bls381Fp>(op, a);
    else if (field == "pallas") run<PallasFp>(op, a);
    else if (field == "ed377") run<Ed377Fp>(op, a);
    else printf("ERR\n");
  }
  return 0;
}

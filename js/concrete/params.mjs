// Curve parameter modules -- same objects as the reference's src/concrete/*.params.ts
// (bls12-377.params.ts:11-45, pasta.params.ts:10-46, bls12-381.params.ts:6-50,
// ed-on-bls12-377.params.ts:5-31), plus the `curveId` the C ABI uses (include/msmz.h).

function exp(x, n, p) {
  x %= p;
  let u = 1n;
  for (; n > 0n; n >>= 1n) {
    if (n & 1n) u = (u * x) % p;
    x = (x * x) % p;
  }
  return u;
}

const p377 =
  0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001n;
const q377 = 0x12ab655e9a2ca55660b44d1e5c37b00159aa76fed00000010a11800000000001n;

export const bls12377Params = {
  label: "bls12-377", curveId: 0, kind: "weierstrass", feBytes: 48,
  modulus: p377, order: q377, cofactor: 0x170b5d44300000000000000000000000n, a: 0n, b: 1n,
  generator: {
    x: 0x008848defe740a67c8fc6225bf87ff5485951e2caa9d41bb188282c8bd37cb5cd5481512ffcd394eeab9b16eb21be9efn,
    y: 0x01914a69c5102eff1f674f5d30afeec4bd7fb348ca3e52d96d182ad44fb82305c2fe3d3634a9591afd82de55559c8ea6n,
  },
  endomorphism: {
    lambda: 0x12ab655e9a2ca55660b44d1e5c37b00114885f32400000000000000000000000n,
    beta: 0x1ae3a4617c510eabc8756ba8f8c524eb8882a75cc9bc8e359064ee822fb5bffd1e945779fffffffffffffffffffffffn,
  },
};

const pP = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001n;
const qP = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001n;
const beta2 = exp(5n, (pP - 1n) / 3n, pP);
export const pallasParams = {
  label: "pallas", curveId: 1, kind: "weierstrass", feBytes: 32,
  modulus: pP, order: qP, cofactor: 1n, a: 0n, b: 5n,
  generator: { x: 1n, y: 0x1b74b5a30a12937c53dfa9f06378ee548f655bd4333d477119cf7a23caed2abbn },
  endomorphism: { lambda: exp(5n, (qP - 1n) / 3n, qP), beta: (beta2 * beta2) % pP },
};

export const bls12381Params = {
  label: "bls12-381", curveId: 2, kind: "weierstrass", feBytes: 48,
  modulus:
    0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaabn,
  order: 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001n,
  cofactor: 0x396c8c005555e1568c00aaab0000aaabn, a: 0n, b: 4n,
  generator: {
    x: 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bbn,
    y: 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1n,
  },
  endomorphism: {
    lambda: 0xd201000000010000n ** 2n - 1n,
    beta: 0x1a0111ea397fe699ec02408663d4de85aa0d857d89759ad4897d29650fb85f9b409427eb4f49fffd8bfd00000000aaacn,
  },
};

export const edOnBls12377Params = {
  label: "ed-on-bls12-377", curveId: 3, kind: "twisted-edwards", feBytes: 32,
  modulus: q377, order: 0x4aad957a68b2955982d1347970dec005293a3afc43c8afeb95aee9ac33fd9ffn,
  cofactor: 4n, d: 3021n,
  generator: {
    x: 0x9f1b5a5baf6acf06fed91c9ae9ebfa06068dd2835790980894e2328f3ebca05n,
    y: 0x9a20df36571ac3cd906b256080ba8454453c177aaf3131bb50a67bf1a806781n,
  },
};

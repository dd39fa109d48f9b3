// Mirror of the reference's integration test src/msm.test.ts:24-118: for ed-on-bls12-377 (twisted Edwards `msm`) and
// pallas, bls12-377, bls12-381 (Weierstraß `msmUnsafe` == `msmProjective`), N = 2^0, 2^2, ..., 2^12, the MSM of the
// engine equals the expected point.  The reference compares with its bigint MSM in the same process; node has no
// oracle, so the expected points are the committed fixtures tests/golden/js_msm_fixtures.json (closed form of the seeded
// inputs, computed by the oracle: tests/golden/make_js_fixtures.py).
//   node js/msm.test.mjs [--json]
import { readFileSync } from "fs";
import { dirname, join } from "path";
import { fileURLToPath } from "url";
import { Weierstraß, TwistedEdwards, startThreads, stopThreads } from "./parallel.mjs";
import { pallasParams, bls12377Params, bls12381Params, edOnBls12377Params as edBls12Params } from "./concrete/params.mjs";

const here = dirname(fileURLToPath(import.meta.url));
const fixtures = JSON.parse(readFileSync(join(here, "..", "tests", "golden", "js_msm_fixtures.json"), "utf8")).cases;
function assert(cond, msg) { if (!cond) throw Error(msg || "assertion failed"); }
function expected(label, n) {
  const f = fixtures.find((c) => c.curve === label && c.n === n);
  assert(f, `no fixture for ${label} 2^${n}`);
  return f;
}
const same = (p, f) => p.x === BigInt(f.x) && p.y === BigInt(f.y) && !!p.isZero === !!f.isZero;
let checked = 0;

async function main() {
  let nThreads = 1;   // startThreads(n): n = GPUs here (the reference: 16 worker threads)
  await startThreads(nThreads);
  // twisted edwards curves
  await testMsmTE(edBls12Params);
  // weierstrass curves with a=0 and endomorphism
  await testMsm(pallasParams);
  await testMsm(bls12377Params);
  await testMsm(bls12381Params);
  await stopThreads();
  if (process.argv.includes("--json")) console.log(JSON.stringify({ ok: true, checked }));
}

async function testMsm(curveParams) {
  console.log("testing msm", curveParams.label);
  const Curve = await Weierstraß.create(curveParams);
  for (let n = 0; n < 14; n += 2) await testOneMsm(Curve, n);
  Curve.close();
}

async function testOneMsm(Curve, n) {
  const { Affine, Projective, Scalar, Parallel } = Curve;
  let N = 1 << n;
  const f = expected(Curve.params.label, n);
  let pointsPtrs = await Parallel.randomPointsFast(N, { seed: BigInt(f.pointSeed) });
  let scalarPtrs = await Parallel.randomScalars(N, { seed: BigInt(f.scalarSeed) });

  // the inputs are what the fixture was computed for
  let g0 = Affine.toBigints(pointsPtrs, 0, 1)[0];
  assert(same(g0, Object.assign({ isZero: false }, f.firstPoint)), "first point differs from the fixture");
  let s0 = Scalar.readBigint(scalarPtrs);
  assert(s0 === BigInt(f.firstScalar) && s0 < Scalar.modulus, "first scalar differs from the fixture");

  let { result } = await Parallel.msmUnsafe(scalarPtrs[0], pointsPtrs[0], N);
  let s = Projective.toBigint(result);
  assert(same(s, f.result), `msm 2^${n} failed`);

  // projective msm
  let { result: resultProjective } = await Parallel.msmProjective(scalarPtrs[0], pointsPtrs[0], N);
  let sProjective = Projective.toBigint(resultProjective);
  assert(same(sProjective, f.result), `msmProjective 2^${n} failed`);
  checked += 2;
  pointsPtrs.free(); scalarPtrs.free();
}

async function testMsmTE(curveParams) {
  console.log("testing msm", curveParams.label);
  const Curve = await TwistedEdwards.create(curveParams);
  for (let n = 0; n < 14; n += 2) await testOneMsmTE(Curve, n);
  Curve.close();
}

async function testOneMsmTE(C, n) {
  const { Scalar, Parallel } = C;
  let N = 1 << n;
  const f = expected(C.params.label, n);
  let pointsPtrs = await Parallel.randomPointsFast(N, { seed: BigInt(f.pointSeed) });
  let scalarPtrs = await Parallel.randomScalars(N, { seed: BigInt(f.scalarSeed) });
  let s0 = Scalar.readBigint(scalarPtrs);
  assert(s0 === BigInt(f.firstScalar) && s0 < Scalar.modulus, "first scalar differs from the fixture");
  let { result } = await Parallel.msm(scalarPtrs[0], pointsPtrs[0], N);
  assert(result.x === BigInt(f.result.x) && result.y === BigInt(f.result.y), `msm 2^${n} failed`);
  checked += 1;
  pointsPtrs.free(); scalarPtrs.free();
}

main().catch((e) => { console.error(e); process.exit(1); });

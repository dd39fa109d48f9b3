// The reference's benchmark / run drivers for the three MSM entry points, written once:
//   scripts/msm-weierstrass.ts:12-108            variant "unsafe"      Parallel.msmUnsafe (batched-affine buckets)
//   scripts/msm-weierstrass-projective.ts:12-107 variant "projective"  Parallel.msmProjective
//   scripts/msm-twisted-edwards.ts:10-100        variant "te"          Parallel.msm on a twisted Edwards curve
// benchmarkMsm = their protocol: random points, one warm-up MSM at 2^15, 15 runs with fresh scalars, the first 5
// dropped, median +- sample standard deviation, then one logged run.  runMsm = one MSM with its stage log and the
// result as an affine bigint point (the reference also compares with its bigint MSM below 2^14: node has no oracle
// here, tests/test_js_host.py does that comparison against the C oracle from the --json output).
// nThreads of the reference = number of GPUs here.
import { Weierstraß, TwistedEdwards, startThreads, stopThreads } from "../parallel.mjs";
import { median, standardDev, tic, toc } from "./evaluate-util.mjs";

const create = (params, variant) => (variant === "te" ? TwistedEdwards : Weierstraß).create(params);
function call(Parallel, variant, scalarPtr, pointPtr, N, verbose) {
  if (variant === "projective") return Parallel.msmProjective(scalarPtr, pointPtr, N);
  if (variant === "te") return Parallel.msm(scalarPtr, pointPtr, N, verbose);
  return Parallel.msmUnsafe(scalarPtr, pointPtr, N, verbose);
}

export async function benchmarkMsm(params, n, nThreads, variant = "unsafe", quiet = false) {
  let N = 1 << n;
  await startThreads(nThreads);
  const Curve = await create(params, variant);
  const { Parallel } = Curve;
  tic(quiet ? "" : "random points");
  let [pointPtr] = await Parallel.randomPointsFast(N);
  toc();
  let [scalarPtr] = await Parallel.randomScalars(N);
  tic(quiet ? "" : "warm-up");
  await call(Parallel, variant, scalarPtr, pointPtr, Math.min(N, 1 << 15), true);
  toc();
  let times = [];
  for (let i = 0; i < 15; i++) {
    let [s] = await Parallel.randomScalars(N, { seed: BigInt(100 + i) });
    tic();
    await call(Parallel, variant, s, pointPtr, N, true);
    let time = toc();
    if (i > 4) times.push(time);
    s.free();
  }
  [scalarPtr] = await Parallel.randomScalars(N, { seed: 99n });
  tic();
  let { log } = await call(Parallel, variant, scalarPtr, pointPtr, N, true);
  let t = toc();
  if (!quiet) {
    log.forEach((l) => console.log(...l));
    console.log(`msm total... ${t.toFixed(2)}ms (incl. host calling overhead)`);
    console.log(times.map((x) => Math.round(x * 100) / 100));
    console.log(`msm (n=${n})... ${median(times).toFixed(2)}ms ± ${standardDev(times).toFixed(2)}ms`);
  }
  Curve.close();
  await stopThreads();
  return { n, median_ms: median(times), std_ms: standardDev(times), times };
}

export async function runMsm(params, n, nThreads, variant = "unsafe", quiet = false) {
  let N = 1 << n;
  await startThreads(nThreads);
  const Curve = await create(params, variant);
  tic(quiet ? "" : "random points");
  let pointsPtrs = await Curve.Parallel.randomPointsFast(N, { seed: 1n });
  toc();
  tic(quiet ? "" : "random scalars");
  let scalarPtrs = await Curve.Parallel.randomScalars(N, { seed: 2n });
  toc();
  tic(quiet ? "" : "convert scalars to bigint & check");
  let scalars = Curve.Scalar.toBigints(scalarPtrs);
  for (const scalar of scalars) if (!(scalar < Curve.Scalar.modulus)) throw Error("scalar out of range");
  if (scalars.length !== N) throw Error("wrong number of scalars");
  toc();
  tic(quiet ? "" : `msm (n=${n})`);
  let { result, log } = await call(Curve.Parallel, variant, scalarPtrs[0], pointsPtrs[0], N, true);
  let scratch = Curve.Field.local.getPointers(5);
  let sAffinePtr = Curve.Field.local.getPointer(Curve.Affine.size);
  Curve.Projective.toAffine(scratch, sAffinePtr, result);
  let s = Curve.Affine.toBigint(sAffinePtr);
  if (!quiet) log.forEach((l) => console.log(...l));
  toc();
  Curve.close();
  await stopThreads();
  return s;
}

// command line shared by the run-msm-*.mjs scripts:  <n> [gpus] [--evaluate] [--json]
export async function main(params, variant) {
  const args = process.argv.slice(2);
  const n = Number(args[0] || 16);
  const gpus = args[1] && !args[1].startsWith("--") ? Number(args[1]) : undefined;
  const json = args.includes("--json");
  if (args.includes("--evaluate")) {
    const r = await benchmarkMsm(params, n, gpus, variant, json);
    if (json) console.log(JSON.stringify(r));
  } else {
    const s = await runMsm(params, n, gpus, variant, json);
    if (json) console.log(JSON.stringify({ n, x: s.x.toString(), y: s.y.toString(), isZero: !!s.isZero }));
    else console.log(s);
  }
}

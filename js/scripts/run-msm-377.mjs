// Mirror of the reference's scripts/run-msm-377.ts + scripts/msm-weierstrass.ts:
//   node js/scripts/run-msm-377.mjs <n> [--evaluate] [--glv 0|1] [--gpus G] [--json]
// --evaluate reproduces the reference's protocol (msm-weierstrass.ts:22-48): warm-up MSM at 2^15,
// 15 runs with fresh scalars, first 5 dropped, median +- sample standard deviation.
import { Weierstraß, startThreads, stopThreads } from "../parallel.mjs";
import { bls12377Params as curveParams } from "../concrete/params.mjs";

function median(arr) {
  const nums = [...arr].sort((a, b) => a - b), mid = arr.length >> 1;
  return arr.length % 2 ? nums[mid] : (nums[mid - 1] + nums[mid]) / 2;
}
function standardDev(arr) {
  const mean = arr.reduce((a, b) => a + b, 0) / arr.length;
  return Math.sqrt(arr.reduce((a, x) => a + (x - mean) ** 2, 0) / (arr.length - 1));
}
const now = () => Number(process.hrtime.bigint()) / 1e6;

async function main() {
  const args = process.argv.slice(2);
  const n = Number(args[0] || 16);
  const doEvaluate = args.includes("--evaluate");
  const json = args.includes("--json");
  const glv = args.includes("--glv") ? Number(args[args.indexOf("--glv") + 1]) : 1;
  const N = 1 << n;
  const gpus = args.includes("--gpus") ? Number(args[args.indexOf("--gpus") + 1]) : undefined;
  await startThreads(gpus);
  const Curve = await Weierstraß.create(curveParams);
  const { Parallel } = Curve;
  let [pointPtr] = await Parallel.randomPointsFast(N, { seed: 1n });
  if (doEvaluate) {
    let [scalarPtr] = await Parallel.randomScalars(N, { seed: 2n });
    await Parallel.msmUnsafe(scalarPtr, pointPtr, Math.min(N, 1 << 15), true, { glv });   // warm-up (msm-weierstrass.ts:24)
    const times = [];
    for (let i = 0; i < 15; i++) {
      let [s] = await Parallel.randomScalars(N, { seed: BigInt(100 + i) });
      const t0 = now();
      await Parallel.msmUnsafe(s, pointPtr, N, true, { glv });
      const t = now() - t0;
      if (i > 4) times.push(t);
      s.free();
    }
    const line = `msm (n=${n})... ${median(times).toFixed(2)}ms ± ${standardDev(times).toFixed(2)}ms`;
    if (json) console.log(JSON.stringify({ n, glv, median_ms: median(times), std_ms: standardDev(times), times }));
    else console.log(line);
  } else {
    let [scalarPtr] = await Parallel.randomScalars(N, { seed: 2n });
    const { result, log } = await Parallel.msmUnsafe(scalarPtr, pointPtr, N, true, { glv });
    const s = Curve.Affine.toBigint(Curve.Projective.toAffine(null, null, result));
    if (json) console.log(JSON.stringify({ n, glv, x: s.x.toString(), y: s.y.toString(), isZero: !!s.isZero }));
    else { log.forEach((l) => console.log(...l)); console.log(s); }
  }
  Curve.close();
  await stopThreads();
}
main().catch((e) => { console.error(e); process.exit(1); });

// Mirror of scripts/evaluate-msm-377.ts:15-62: sweep the window size c around a starting point for one input size and
// report median / standard deviation per c and the best c (the reference sweeps c = n - 1 + {0, 1}; the GPU's optimum
// sits lower, so the sweep is centred on the engine's own choice: --c0 <c> --span <k> tests c0 - k .. c0 + k).
//   node js/scripts/evaluate-msm-377.mjs <n> [--c0 c] [--span k] [--glv 0|1] [--json]
import { Weierstraß, startThreads, stopThreads } from "../parallel.mjs";
import { bls12377Params as curveParams } from "../concrete/params.mjs";
import { median, standardDev, tic, toc } from "./evaluate-util.mjs";

const args = process.argv.slice(2);
const opt = (name, dflt) => (args.includes(name) ? Number(args[args.indexOf(name) + 1]) : dflt);
const n = Number(args[0] || 16), json = args.includes("--json"), glv = opt("--glv", -1);
let warmup = 2, repeat = 5;

async function evaluateParameters(N, C) {
  let times = {}, best = {};
  await startThreads();
  const { Parallel, close } = await Weierstraß.create(curveParams);
  for (let n of N) {
    times[n] = {};
    best[n] = { time: Infinity };
    let [points] = await Parallel.randomPointsFast(1 << n);
    let [probe] = await Parallel.randomScalars(1 << n);
    const chosen = (await Parallel.msmUnsafe(probe, points, 1 << n, true, { glv })).stats.c;   // the engine's own window
    const c0 = opt("--c0", chosen);
    for (let cDelta of C) {
      let c = c0 + cDelta;
      if (c < 2 || c > 24) continue;
      let times_ = [];
      for (let i = 0; i < warmup; i++) {
        let [scalars] = await Parallel.randomScalars(1 << n, { seed: BigInt(10 + i) });
        await Parallel.msmUnsafe(scalars, points, 1 << n, true, { c, glv });
        scalars.free();
      }
      for (let i = 0; i < repeat; i++) {
        let [scalars] = await Parallel.randomScalars(1 << n, { seed: BigInt(20 + i) });
        tic();
        try {
          await Parallel.msmUnsafe(scalars, points, 1 << n, false, { c, glv });
          times_.push(toc());
        } catch (e) {
          console.error(e);
        }
        scalars.free();
      }
      let time = median(times_), std = standardDev(times_);
      times[n][c] = { time, std };
      if (!json) console.dir({ n, c, time, std });
      if (time < best[n].time) best[n] = { time, std, c, chosen };
    }
  }
  close();
  await stopThreads();
  return { times, best };
}

const span = opt("--span", 1);
evaluateParameters([n], Array.from({ length: 2 * span + 1 }, (_, i) => i - span))
  .then((r) => (json ? console.log(JSON.stringify(r)) : console.dir(r, { depth: Infinity })))
  .catch((e) => { console.error(e); process.exit(1); });

// Mirror of the reference's scripts/msm-weierstrass.ts: benchmarkMsm(params, n, nThreads) and runMsm(params, n, nThreads)
// (nThreads = number of GPUs here).  The bodies live in msm-drivers.mjs, shared by the three entry points.
import * as drivers from "./msm-drivers.mjs";
export const benchmarkMsm = (params, n, nThreads) => drivers.benchmarkMsm(params, n, nThreads, "unsafe");
export const runMsm = (params, n, nThreads) => drivers.runMsm(params, n, nThreads, "unsafe");

// Mirror of scripts/zprize23/submission-test-bls377.ts:6-45: sanity check that compute_msm runs and handles the
// reference's known-answer case.  The reference checks on-curve / subgroup membership with its bigint curve and
// draws unseeded random scalars; here the scalars come from a fixed linear congruence (no bigint curve in js/) and
// the expected relations are the reference's own: 2 P + (q - 1) P = P, and MSM of n copies of P = (sum of scalars) P.
//   node js/scripts/submission-test-bls377.mjs [--json]
import { compute_msm, BLS12377 } from "./submission-bls377.mjs";
import { bls12377Params as curveParams } from "../concrete/params.mjs";

let point = {
  x: 111871295567327857271108656266735188604298176728428155068227918632083036401841336689521497731900230387779623820740n,
  y: 76860045326390600098227152997486448974650822224305058012700629806287380625419427989664237630603922765089083164740n,
  isZero: false,
};

function toBytes(x, len) {
  const out = Buffer.alloc(len);
  for (let i = 0; i < len; i++) { out[i] = Number(x & 0xffn); x >>= 8n; }
  return out;
}

async function main() {
  const json = process.argv.includes("--json");
  const report = {};
  let scalars = [2n, curveParams.order - 1n];
  // 2*P + (-1)*P should give P again
  let result = await compute_msm([point, point], scalars);
  if (result.x !== point.x || result.y !== point.y) throw Error("failed");
  report.twoPoints = true;
  if (!json) console.log("2 points ok");

  const n = 1000;
  let state = 0x9e3779b97f4a7c15n;
  let randomScalars = Array.from({ length: n }, () => {
    state = (state * 6364136223846793005n + 1442695040888963407n) % (1n << 256n);
    return (state * state + 12345n) % curveParams.order;
  });
  let samePoints = Array.from({ length: n }, () => point);
  let scalarSum = randomScalars.reduce((a, b) => (a + b) % curveParams.order);

  // msm should be the same as scaling by the sum of scalars
  let result2 = await compute_msm(samePoints, randomScalars);
  let result3 = await compute_msm([point], [scalarSum]);
  if (result2.x !== result3.x || result2.y !== result3.y) throw Error("failed");
  report.samePoints = true;
  if (!json) console.log("same points ok");

  // the byte route of compute_msm (Buffer inputs): same answers
  const pointBytes = Buffer.concat(samePoints.map((p) => Buffer.concat([toBytes(p.x, 48), toBytes(p.y, 48)])));
  const scalarBytes = Buffer.concat(randomScalars.map((s) => toBytes(s, 32)));
  let result4 = await compute_msm(pointBytes, scalarBytes);
  if (result4.x !== result3.x || result4.y !== result3.y) throw Error("failed");
  report.byteRoute = true;
  if (!json) console.log("byte inputs ok");

  report.sum = { x: result3.x.toString(), y: result3.y.toString(), scalar: scalarSum.toString() };
  if (json) console.log(JSON.stringify(report));
  BLS12377.close();
}
main().catch((e) => { console.error(e); process.exit(1); });

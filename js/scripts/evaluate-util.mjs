// scripts/evaluate-util.ts:3-20: median and sample standard deviation of a list of timings
export function median(arr) {
  const nums = [...arr].sort((a, b) => a - b), mid = arr.length >> 1;
  return arr.length % 2 ? nums[mid] : (nums[mid - 1] + nums[mid]) / 2;
}
export function standardDev(arr) {
  const mean = arr.reduce((a, b) => a + b, 0) / arr.length;
  return Math.sqrt(arr.reduce((a, x) => a + (x - mean) ** 2, 0) / (arr.length - 1));
}
let t0 = 0, label = "";
export function tic(l = "") { label = l; t0 = Number(process.hrtime.bigint()) / 1e6; }
export function toc() {
  const t = Number(process.hrtime.bigint()) / 1e6 - t0;
  if (label) console.log(`${label}... ${t.toFixed(1)}ms`);
  label = "";
  return t;
}

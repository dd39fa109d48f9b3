// Mirror of the reference's ZPrize entry point scripts/zprize23/submission-bls377.ts:20-65:
//   compute_msm(inputPoints, inputScalars) -> { x, y }
// Inputs as there: points as bigint objects {x, y, isZero} or as bytes (x || y, 48 + 48 little-endian canonical), scalars
// as bigints or as 32-byte little-endian records.  Same control flow: transfer, convert behind the pre-allocated
// pointers, use `msm` (safe additions) when the first two points are equal and `msmUnsafe` otherwise, return the
// affine bigint point.  The only differences from the reference text: TypeScript annotations and the `using` arena
// guard are gone (node 12, no compile step), imports come from js/.
import { Weierstraß, startThreads } from "../parallel.mjs";
import { bls12377Params } from "../concrete/params.mjs";

export { compute_msm, BLS12377 };

let BLS12377, scratch, pointPtr, scalarPtr, pointInputPtr, scalarInputPtr;
const nMax = 1 << 20;
let ready = null;
// (top-level await is not available on node 12: the set-up the reference runs at import time runs on the first call)

async function setup() {
  await startThreads();
  BLS12377 = await Weierstraß.create(bls12377Params);
  scratch = BLS12377.Field.local.getPointers(20);
  // pointers for data used by msm
  pointPtr = await BLS12377.Parallel.getPointer(nMax * BLS12377.Affine.size);
  scalarPtr = await BLS12377.Parallel.getScalarPointer(nMax * BLS12377.Scalar.sizeField);
  // pointers for input data
  pointInputPtr = await BLS12377.Parallel.getPointer(nMax * 2 * 48);
  scalarInputPtr = await BLS12377.Parallel.getScalarPointer(nMax * 32);
}

async function compute_msm(inputPoints, inputScalars) {
  if (ready === null) ready = setup();
  await ready;
  let n = 0;

  // transfer to device memory
  if (typeof inputScalars[0] === "bigint") {
    n = inputScalars.length;
    await scalarsFromBigint(inputScalars);
  } else {
    n = inputScalars.length / 32;
    await scalarsFromBytes(inputScalars);
  }
  if (typeof inputPoints[0] === "object" && "x" in inputPoints[0] && typeof inputPoints[0].x === "bigint") {
    await pointsFromBigint(inputPoints);
  } else {
    await pointsFromBytes(inputPoints);
  }

  let samePoints = n > 1 && BLS12377.Field.isEqual(pointPtr, pointPtr + BLS12377.Affine.size);
  let result;

  // compute msm
  if (samePoints) {
    ({ result } = await BLS12377.Parallel.msm(scalarPtr, pointPtr, n));
  } else {
    // if not all points are the same, we use the unsafe version which is faster
    ({ result } = await BLS12377.Parallel.msmUnsafe(scalarPtr, pointPtr, n));
  }

  // return as affine bigint point
  let resultAffine = BLS12377.Field.local.getPointer(BLS12377.Affine.size);
  BLS12377.Projective.toAffine(scratch, resultAffine, result);
  let resultBigint = BLS12377.Affine.toBigint(resultAffine);
  return resultBigint;
}

async function pointsFromBytes(inputPoints) {
  let n = inputPoints.length / (2 * 48);
  // transfer input bytes to the staging memory behind the input pointer
  BLS12377.Field.memoryBytes.set(inputPoints, pointInputPtr);
  // convert input bytes to point representation (on the GPU)
  await BLS12377.Parallel.pointsFromBytes(pointPtr, pointInputPtr, n);
}

async function scalarsFromBytes(inputScalars) {
  let n = inputScalars.length / 32;
  BLS12377.Scalar.memoryBytes.set(inputScalars, scalarInputPtr);
  await BLS12377.Parallel.scalarsFromBytes(scalarPtr, scalarInputPtr, n);
}

async function pointsFromBigint(inputPoints) {
  let { Affine } = BLS12377;
  Affine.writeBigints(pointPtr, inputPoints);
}

async function scalarsFromBigint(inputScalars) {
  let n = inputScalars.length;
  let { writeBigint, sizeField: size } = BLS12377.Scalar;
  for (let i = 0, si = scalarPtr; i < n; i++, si += size) {
    writeBigint(si, inputScalars[i]);
  }
}

// Host-side mirror of the reference's src/parallel.ts for the MI355X engine.
//
//   import { Weierstraß, TwistedEdwards, startThreads, stopThreads } from "./parallel.mjs";
//   await startThreads();
//   const Curve = await Weierstraß.create(bls12377Params);
//   let [pointPtr] = await Curve.Parallel.randomPointsFast(N);
//   let [scalarPtr] = await Curve.Parallel.randomScalars(N);
//   let { result, log } = await Curve.Parallel.msmUnsafe(scalarPtr, pointPtr, N, true);
//   Curve.Projective.toAffine(scratch, affPtr, result); Curve.Affine.toBigint(affPtr)  // or toBigint(result)
//
// Same names / argument order / promise-returning style as parallel.ts:149-158, 263-271.  The
// reference's "pointers" into wasm memory become opaque handles to GPU-resident arrays; all the
// arithmetic happens in HIP behind the N-API addon (napi/msmz_napi.c -> include/msmz.h).  This file is
// plain ECMAScript (valid TypeScript; declarations in parallel.d.ts) so that it runs without a
// compile step on the image's node 12.
import { createRequire } from "module";
import { dirname, join } from "path";
import { fileURLToPath } from "url";

const require = createRequire(import.meta.url);
const here = dirname(fileURLToPath(import.meta.url));
let addon = null;
function native() {
  // fails loudly if the addon / libmsmz.so has not been built: there is no wasm or JS fallback
  if (addon === null) addon = require(join(here, "msmz_napi.node"));
  return addon;
}

let devices = null;

/** parallel.ts:291-315.  The reference spawns n-1 workers that share one MSM; here the workers are GPUs: a curve
 * created after startThreads(n) drives GPUs first..first+n-1 (inputs split over them, partial sums added on the
 * host: msmz_create with n_devices = n).  Without n: one GPU, LOCAL_RANK or 0.  `deviceId` may also be an array
 * of ids. */
export async function startThreads(n, deviceId) {
  if (Array.isArray(deviceId)) devices = deviceId.map(Number);
  else {
    const first = deviceId !== undefined ? Number(deviceId) : n === undefined || n === 1 ? Number(process.env.LOCAL_RANK || 0) : 0;
    devices = Array.from({ length: n || 1 }, (_, i) => first + i);
  }
  if (devices.length < 1 || devices.length > 8) throw Error(`startThreads: 1..8 GPUs, got ${devices.length}`);
  native();
  return devices.length === 1 ? devices[0] : devices;
}

/** parallel.ts:317-320 */
export async function stopThreads() {
  devices = null;
}

function bytesToBigint(buf, off, len) {
  let x = 0n;
  for (let i = len - 1; i >= 0; i--) x = (x << 8n) | BigInt(buf[off + i]);
  return x;
}
function bigintToBytes(x, len) {
  const out = Buffer.alloc(len);
  for (let i = 0; i < len; i++) {
    out[i] = Number(x & 0xffn);
    x >>= 8n;
  }
  return out;
}

/** A GPU-resident input array; destructures like the reference's pointer arrays: `let [ptr] = ...` */
class DeviceArray extends Array {
  static make(curve, handle, n, kind) {
    const a = new DeviceArray();
    a.push(a);
    Object.defineProperties(a, {
      curve: { value: curve }, handle: { value: handle, writable: true }, n: { value: n }, kind: { value: kind },
    });
    return a;
  }
  free() {
    if (this.handle !== null) native().free(this.curve._ctx, this.handle);
    this.handle = null;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Pointer-style routes of the reference (src/parallel.ts:89-133, src/curve-affine.ts:290-308, Scalar.writeBigint):
//   let pointPtr = await Parallel.getPointer(nMax * Affine.size);          // parallel.ts:89-91
//   Field.memoryBytes.set(bytes, pointInputPtr);                           // host bytes -> "wasm memory"
//   await Parallel.pointsFromBytes(pointPtr, pointInputPtr, n);            // parallel.ts:97-112
//   Affine.writeBigints(pointPtr, points); Scalar.writeBigint(scalarPtr + i * Scalar.sizeField, s);
//   await Parallel.msmUnsafe(scalarPtr, pointPtr, n);
// The reference's pointers are byte offsets into wasm memory and its scripts do arithmetic on them, so the shims
// keep them NUMBERS: addresses in a virtual space, each region backed by a host staging buffer and (once an MSM or
// a conversion needs it) a GPU-resident array.  Record sizes are this engine's wire sizes (Affine.size = 2 fe_bytes,
// Scalar.sizeField = 32), not the wasm limb sizes; scripts that use the named constants need no other change.
class Region {
  constructor(base, size, kind) {
    this.base = base; this.size = size; this.kind = kind;   // kind: "field" | "scalar"
    this.host = null;      // Buffer of staged records (canonical little-endian bytes)
    this.inf = null;       // Uint8Array of infinity flags (points written with isZero)
    this.count = 0;        // records staged
    this.dirty = false;    // host bytes newer than the device copy
    this.device = null;    // DeviceArray
  }
}
const REGION_ALIGN = 2 ** 20;
function makePointerSpace() {
  const regions = [];
  let next = REGION_ALIGN;   // 0 stays the null pointer
  return {
    alloc(size, kind) {
      const r = new Region(next, size, kind);
      regions.push(r);
      next += Math.ceil((size + 1) / REGION_ALIGN) * REGION_ALIGN;
      return r.base;
    },
    find(ptr) {   // -> [region, byte offset]
      for (const r of regions) if (ptr >= r.base && ptr < r.base + Math.max(r.size, 1)) return [r, ptr - r.base];
      throw Error(`pointer ${ptr} was not returned by getPointer / getScalarPointer`);
    },
  };
}

function createCurve(params, kind) {
  if (params.kind !== kind) throw Error(`${params.label} is not a ${kind} curve`);
  if (devices === null) devices = [Number(process.env.LOCAL_RANK || 0)];
  const N = native();
  const ctx = N.create(params.curveId, devices.length === 1 ? devices[0] : devices);
  const fb = params.feBytes;
  const te = kind === "twisted-edwards";
  const curve = { params, _ctx: ctx };
  const space = makePointerSpace();
  const recSize = { field: 2 * fb, scalar: 32 };
  function stage(region, offset, bytes, count) {   // copy `bytes` into the region's host buffer at byte offset
    if (offset + bytes.length > region.size) throw Error("write beyond the end of the pointer's allocation");   // memory-helpers.ts:224-236
    if (region.host === null) region.host = Buffer.alloc(region.size);
    Buffer.from(bytes.buffer, bytes.byteOffset, bytes.length).copy(region.host, offset);
    region.count = Math.max(region.count, count);
    region.dirty = true;
  }
  // the GPU-resident array behind a pointer (uploaded on first use / after the staged bytes changed)
  function resident(ptr, n, what) {
    if (ptr instanceof DeviceArray) return ptr;
    const [r, off] = space.find(ptr);
    if (off !== 0) throw Error(`${what}: an MSM input must start at the pointer getPointer returned`);
    if (r.dirty || r.device === null || r.device.n < n) {
      if (r.host === null || r.count < n) throw Error(`${what}: ${r.count} records were written, ${n} needed`);
      if (r.device) r.device.free();
      const size = recSize[r.kind];
      const bytes = r.host.subarray(0, r.count * size);
      r.device = r.kind === "scalar"
        ? DeviceArray.make(curve, N.uploadScalars(ctx, bytes, r.count), r.count, "scalars")
        : DeviceArray.make(curve, N.uploadPoints(ctx, bytes, r.inf && r.inf.some((v) => v) ? Buffer.from(r.inf.subarray(0, r.count)) : null, r.count), r.count, "points");
      r.dirty = false;
    }
    return r.device;
  }

  function decodePoint(buf, off, isInf) {
    const p = { x: bytesToBigint(buf, off, fb), y: bytesToBigint(buf, off + fb, fb) };
    if (!te) {
      p.isZero = !!isInf;
      if (p.isZero) { p.x = 0n; p.y = 1n; } // bigint/projective-weierstrass.ts:210
    }
    return p;
  }

  async function msmCommon(scalars, points, n, verbose, options, safe, buckets) {
    options = options || {};
    const opts = {
      c: options.c || 0,
      glv: options.glv !== undefined ? Number(options.glv) : te ? 0 : -1,   // -1: GLV below 2^21 points (include/msmz.h)
      safe: options.useSafeAdditions !== undefined ? Number(options.useSafeAdditions) : safe,
      buckets,
      timing: verbose ? 1 : 0,
      reduceAffine: options.reduceAffine ? 1 : 0, // batched-affine first reduction level (reduceBucketsAffine)
    };
    if (typeof points === "number") points = resident(points, n, "msm points");
    if (typeof scalars === "number") scalars = resident(scalars, n, "msm scalars");
    const s = scalars instanceof DeviceArray ? scalars.handle : scalars; // Buffer = host scalars
    const r = N.msm(ctx, points.handle, s, n, fb, opts);
    const result = decodePoint(r.xy, 0, r.isInf);
    const log = [[{ n: Math.ceil(Math.log2(n)), K: r.log.K, c: r.log.c }]];
    for (const [k, v] of Object.entries(r.log.stageMs)) log.push([`${k}... ${v.toFixed(3)}ms`]);
    return { result, log, stats: r.log };
  }

  const Parallel = {
    /** curve-random.ts:14-92, seeded: point i = splitmix64(seed, i) * G */
    async randomPointsFast(n, { seed = 0x6d736d7an } = {}) {
      return DeviceArray.make(curve, N.randomPoints(ctx, n, BigInt(seed)), n, "points");
    },
    /** curve-random.ts:151-194, seeded */
    async randomScalars(n, { seed = 0x6d736d7an } = {}) {
      return DeviceArray.make(curve, N.randomScalars(ctx, n, BigInt(seed)), n, "scalars");
    },
    /** parallel.ts:89-95 */
    async getPointer(size) { return space.alloc(size, "field"); },
    async getScalarPointer(size) { return space.alloc(size, "scalar"); },
    /** parallel.ts:97-112: x||y little-endian canonical.  Two forms: (bytes, n?, isInf?) -> DeviceArray, and the
     * reference's (pointPtr, pointInputPtr, n): the bytes were put behind pointInputPtr with Field.memoryBytes.set,
     * the converted (Montgomery, GPU-resident) points end up behind pointPtr. */
    async pointsFromBytes(bytes, n, isInf) {
      if (typeof bytes === "number") {
        const [dst, doff] = space.find(bytes), [src, soff] = space.find(n), count = isInf;
        if (doff !== 0 || src.host === null || soff + count * 2 * fb > src.size) throw Error("pointsFromBytes(ptr, inputPtr, n): bad pointers");
        if (dst.device) dst.device.free();
        // range errors (a coordinate >= p) surface here, like every upload
        dst.device = DeviceArray.make(curve, N.uploadPoints(ctx, src.host.subarray(soff, soff + count * 2 * fb), null, count), count, "points");
        dst.host = src.host.subarray(soff, soff + count * 2 * fb); dst.count = count; dst.dirty = false; dst.inf = null;
        return;
      }
      n = n === undefined ? Math.floor(bytes.length / (2 * fb)) : n;
      if (!(n > 0) || bytes.length < 2 * fb * n || (isInf && isInf.length < n)) throw Error(`pointsFromBytes: ${bytes.length} bytes for ${n} points`);
      return DeviceArray.make(curve, N.uploadPoints(ctx, Buffer.from(bytes), isInf ? Buffer.from(isInf) : null, n), n, "points");
    },
    /** parallel.ts:114-133: 32 bytes little-endian per scalar */
    async scalarsFromBytes(bytes, n, count) {
      if (typeof bytes === "number") {   // (scalarPtr, scalarInputPtr, n): parallel.ts:114-133
        const [dst, doff] = space.find(bytes), [src, soff] = space.find(n);
        if (doff !== 0 || src.host === null || soff + count * 32 > src.size) throw Error("scalarsFromBytes(ptr, inputPtr, n): bad pointers");
        if (dst.device) dst.device.free();
        dst.device = DeviceArray.make(curve, N.uploadScalars(ctx, src.host.subarray(soff, soff + count * 32), count), count, "scalars");
        dst.host = src.host.subarray(soff, soff + count * 32); dst.count = count; dst.dirty = false;
        return;
      }
      n = n === undefined ? Math.floor(bytes.length / 32) : n;
      if (!(n > 0) || bytes.length < 32 * n) throw Error(`scalarsFromBytes: ${bytes.length} bytes for ${n} scalars`);
      return DeviceArray.make(curve, N.uploadScalars(ctx, Buffer.from(bytes), n), n, "scalars");
    },
    /** msm-batched-affine.ts:74-328 with safe additions */
    msm: (scalars, points, n, verbose = false, options) => msmCommon(scalars, points, n, verbose, options, 1, 0),
    /** msm-batched-affine.ts:574-586 */
    msmUnsafe: (scalars, points, n, verbose = false, options) => msmCommon(scalars, points, n, verbose, options, 0, 0),
  };
  if (!te) {
    /** parallel.ts:69-87 */
    Parallel.msmProjective = (scalars, points, n, options) =>
      msmCommon(scalars, points, n, true, Object.assign({}, options, { glv: 0 }), 1, 1);
  }

  curve.Parallel = Parallel;
  // "wasm memory" of the reference as far as its scripts touch it: memoryBytes.set(bytes, ptr) and field equality
  curve.Field = {
    sizeField: fb,
    memoryBytes: { set: (bytes, ptr) => { const [r, off] = space.find(ptr); stage(r, off, bytes, Math.floor((off + bytes.length) / recSize[r.kind])); } },
    /** field-arithmetic.ts:184-199 on staged canonical bytes: are the field elements at the two addresses equal? */
    isEqual(a, b) {
      const [ra, oa] = space.find(a), [rb, ob] = space.find(b);
      if (ra.host === null || rb.host === null) throw Error("Field.isEqual: nothing was written there");
      return ra.host.compare(rb.host, ob, ob + fb, oa, oa + fb) === 0;
    },
    local: { getPointers: (n) => Array.from({ length: n }, () => space.alloc(2 * fb, "field")), getPointer: (size) => space.alloc(size, "field") },
  };
  curve.Scalar = {
    sizeField: 32,
    memoryBytes: { set: (bytes, ptr) => { const [r, off] = space.find(ptr); stage(r, off, bytes, Math.floor((off + bytes.length) / 32)); } },
    /** Scalar.writeBigint(ptr, s): scripts/zprize23/submission-bls377.ts:95-102 */
    writeBigint(ptr, s) {
      if (s < 0n || s >= params.order) throw Error("scalar out of range");
      const [r, off] = space.find(ptr);
      stage(r, off, bigintToBytes(s, 32), off / 32 + 1);
    },
    modulus: params.order,
    sizeInBits: (params.order - 1n).toString(2).length,
    readBigint: (arr, i = 0) => bytesToBigint(N.downloadScalars(ctx, arr.handle, i, 1), 0, 32),
    toBigints(arr, first = 0, count = arr.n - first) {
      const b = N.downloadScalars(ctx, arr.handle, first, count);
      return Array.from({ length: count }, (_, i) => bytesToBigint(b, 32 * i, 32));
    },
    fromBigints: (scalars) => Parallel.scalarsFromBytes(Buffer.concat(scalars.map((s) => bigintToBytes(s, 32)))),
  };
  curve.Affine = {
    size: 2 * fb,
    /** curve-affine.ts:220-233; accepts an MSM result, or a pointer Projective.toAffine wrote to */
    toBigint: (p) => (typeof p === "number" ? space.find(p)[0].result : p),
    /** curve-affine.ts:290-308: canonical bigint points behind a pointer (isZero -> infinity flag) */
    writeBigints(ptr, points) {
      const [r, off] = space.find(ptr);
      const first = off / (2 * fb);
      if (r.inf === null) r.inf = new Uint8Array(Math.floor(r.size / (2 * fb)));
      points.forEach((p, i) => {
        r.inf[first + i] = p.isZero ? 1 : 0;
        stage(r, off + i * 2 * fb, Buffer.concat([bigintToBytes(p.isZero ? 0n : p.x, fb), bigintToBytes(p.isZero ? 0n : p.y, fb)]), first + i + 1);
      });
      return ptr;
    },
    toBigints(arr, first = 0, count = arr.n - first) {
      const b = N.downloadPoints(ctx, arr.handle, first, count, fb);
      return Array.from({ length: count }, (_, i) => {
        let zero = !te;
        for (let j = 0; j < 2 * fb && zero; j++) zero = b[2 * fb * i + j] === 0;
        return decodePoint(b, 2 * fb * i, zero);
      });
    },
    fromBigints(points) {
      const data = Buffer.concat(points.map((p) => Buffer.concat([bigintToBytes(p.x, fb), bigintToBytes(p.y, fb)])));
      const inf = Buffer.from(points.map((p) => (p.isZero ? 1 : 0)));
      return Parallel.pointsFromBytes(data, points.length, inf.some((v) => v) ? inf : null);
    },
  };
  // the reference converts the projective result with Projective.toAffine(scratch, affPtr, result)
  // (scripts/msm-weierstrass.ts:90-92); results here are already canonical affine points
  curve.Projective = {
    toAffine(_scratch, affPtr, result) {
      if (typeof affPtr === "number") space.find(affPtr)[0].result = result;
      return result;
    },
    toBigint: (result) => result,
  };
  curve.Curve = { toBigint: (result) => result };
  curve.pointAdd = (a, b) => {
    const enc = (p) => (p.isZero ? null : Buffer.concat([bigintToBytes(p.x, fb), bigintToBytes(p.y, fb)]));
    const r = N.pointAdd(params.curveId, enc(a), enc(b), fb);
    return decodePoint(r.xy, 0, r.isInf);
  };
  curve.close = () => N.destroy(ctx);
  return curve;
}

export const Weierstraß = { create: async (params) => createCurve(params, "weierstrass") };
export const Weierstrass = Weierstraß;
export const TwistedEdwards = { create: async (params) => createCurve(params, "twisted-edwards") };

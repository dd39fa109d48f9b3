// TypeScript declarations for js/parallel.mjs -- the shape of the reference's src/parallel.ts exports.
export type CurveParams = {
  label: string; curveId: number; kind: "weierstrass" | "twisted-edwards"; feBytes: number;
  modulus: bigint; order: bigint; cofactor: bigint; a?: bigint; b?: bigint; d?: bigint;
  generator: { x: bigint; y: bigint }; endomorphism?: { lambda: bigint; beta: bigint };
};
export type BigintPoint = { x: bigint; y: bigint; isZero?: boolean };
export interface DeviceArray extends Array<DeviceArray> { readonly n: number; readonly kind: "points" | "scalars"; free(): void }
export type MsmOptions = { c?: number; glv?: boolean | number; useSafeAdditions?: boolean; reduceAffine?: boolean };
export type MsmResult = { result: BigintPoint; log: any[][]; stats: Record<string, any> };
export interface ParallelApi {
  randomPointsFast(n: number, options?: { seed?: bigint | number }): Promise<DeviceArray>;
  randomScalars(n: number, options?: { seed?: bigint | number }): Promise<DeviceArray>;
  pointsFromBytes(bytes: Uint8Array, n?: number, isInf?: Uint8Array): Promise<DeviceArray>;
  scalarsFromBytes(bytes: Uint8Array, n?: number): Promise<DeviceArray>;
  /** pointer-style routes of src/parallel.ts:89-133: pointers are numbers in a virtual address space */
  getPointer(size: number): Promise<number>;
  getScalarPointer(size: number): Promise<number>;
  pointsFromBytes(pointPtr: number, pointInputPtr: number, n: number): Promise<void>;
  scalarsFromBytes(scalarPtr: number, scalarInputPtr: number, n: number): Promise<void>;
  msm(scalars: DeviceArray | Uint8Array | number, points: DeviceArray | number, n: number, verbose?: boolean, options?: MsmOptions): Promise<MsmResult>;
  msmUnsafe(scalars: DeviceArray | Uint8Array | number, points: DeviceArray | number, n: number, verbose?: boolean, options?: MsmOptions): Promise<MsmResult>;
  msmProjective?(scalars: DeviceArray | Uint8Array, points: DeviceArray, n: number, options?: MsmOptions): Promise<MsmResult>;
}
export interface MsmCurve {
  params: CurveParams; Parallel: ParallelApi;
  Field: { sizeField: number; memoryBytes: { set(bytes: Uint8Array, ptr: number): void }; isEqual(a: number, b: number): boolean;
           local: { getPointers(n: number): number[]; getPointer(size: number): number } };
  Scalar: { sizeField: number; memoryBytes: { set(bytes: Uint8Array, ptr: number): void }; writeBigint(ptr: number, s: bigint): void;
            modulus: bigint; sizeInBits: number; readBigint(a: DeviceArray, i?: number): bigint; toBigints(a: DeviceArray, first?: number, count?: number): bigint[]; fromBigints(s: bigint[]): Promise<DeviceArray> };
  Affine: { size: number; toBigint(p: BigintPoint | number): BigintPoint; writeBigints(ptr: number, points: BigintPoint[]): number; toBigints(a: DeviceArray, first?: number, count?: number): BigintPoint[]; fromBigints(p: BigintPoint[]): Promise<DeviceArray> };
  Projective: { toAffine(scratch: unknown, affPtr: number | null, result: BigintPoint): BigintPoint; toBigint(r: BigintPoint): BigintPoint };
  pointAdd(a: BigintPoint, b: BigintPoint): BigintPoint; close(): void;
}
/** n = number of GPUs a curve context drives (inputs split over them); deviceId = first GPU or an explicit list */
export function startThreads(n?: number, deviceId?: number | number[]): Promise<number | number[]>;
export function stopThreads(): Promise<void>;
export const Weierstraß: { create(params: CurveParams): Promise<MsmCurve> };
export const Weierstrass: { create(params: CurveParams): Promise<MsmCurve> };
export const TwistedEdwards: { create(params: CurveParams): Promise<MsmCurve> };

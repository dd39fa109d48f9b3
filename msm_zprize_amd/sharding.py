"""Multi-GPU sharding of one MSM: input split + host-side combine.

MSM is additive over disjoint index sets, so rank g owns the contiguous range
[g*N/G, (g+1)*N/G) of points and scalars, runs the complete single-GPU pipeline on it, and the only
exchange is a gather of G affine results (2*fe_bytes + 4 bytes each) that are then added on the host
with msmz_point_add.  This replaces the reference's bucket-range split over worker threads
(msm-common.ts:88-188, threads.ts:354-359) and its "partition sum" on the main thread
(msm-batched-affine.ts:300-307).  No collective runs inside an MSM.
"""
import ctypes as C

from ._native import check, lib


def shard_range(n_total, rank, world):
    """threads.ts:354-359 `range(n)`: contiguous, sizes differ by at most one"""
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


BLOCK_SHIFT = 16


def block_shard_count(n, g, G, shift=BLOCK_SHIFT):
    """entries of the first n that live on device g of a G-device context (csrc/multi.h shard_count): blocks of
    2^shift consecutive entries are dealt round-robin, so a prefix of the set is a prefix on every device"""
    blk = 1 << shift
    full, rem = divmod(n, blk * G)
    return full * blk + min(max(rem - g * blk, 0), blk)


def block_local_index(i, G, shift=BLOCK_SHIFT):
    """global entry index -> (device, local index) inside a multi-device context"""
    b = i >> shift
    return b % G, ((b // G) << shift) | (i & ((1 << shift) - 1))


def encode_point(params, p):
    fb = params["fe_bytes"]
    return int(p["x"]).to_bytes(fb, "little") + int(p["y"]).to_bytes(fb, "little") + \
        bytes([1 if p.get("isZero") else 0, 0, 0, 0])


def decode_point(params, b):
    fb = params["fe_bytes"]
    return {"x": int.from_bytes(b[:fb], "little"), "y": int.from_bytes(b[fb:2 * fb], "little"),
            "isZero": b[2 * fb] != 0}


def point_add(params, a, b):
    """host-side group addition of two canonical affine points (no GPU involved)"""
    fb = params["fe_bytes"]
    te = params["kind"] == "twisted-edwards"
    enc = lambda p: int(p["x"]).to_bytes(fb, "little") + int(p["y"]).to_bytes(fb, "little")
    za, zb = bool(a.get("isZero")) and not te, bool(b.get("isZero")) and not te
    out = C.create_string_buffer(2 * fb)
    inf = C.c_int()
    check(lib().msmz_point_add(params["curve_id"], None if za else enc(a), int(za), None if zb else enc(b), int(zb),
                               out, C.byref(inf)), "msmz_point_add")
    r = {"x": int.from_bytes(out.raw[:fb], "little"), "y": int.from_bytes(out.raw[fb:], "little"),
         "isZero": inf.value != 0}
    if r["isZero"] and not te:
        r["x"], r["y"] = 0, 1
    return r


def identity(params):
    if params["kind"] == "twisted-edwards":
        return {"x": 0, "y": 1, "isZero": False}
    return {"x": 0, "y": 1, "isZero": True}


def combine_partials(params, partial, device=None):
    """all_gather the per-rank partial sums (RCCL on GPUs, gloo on CPU) and add them on every rank"""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return partial
    rec = encode_point(params, partial)
    t = torch.frombuffer(bytearray(rec), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    gathered = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(gathered, t)
    total = identity(params)
    for g in gathered:
        total = point_add(params, total, decode_point(params, bytes(g.cpu().numpy())))
    return total

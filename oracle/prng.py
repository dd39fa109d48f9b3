"""Restatement of the product's seeded input generators (msm_zprize_amd/csrc/gen_kernels.h) in
Python integers, so tests can predict device-generated inputs.  TEST INFRASTRUCTURE ONLY.
The reference's own generators are unseeded (src/curve-random.ts:14-92, 151-194; util.ts:226-233)."""

MASK64 = (1 << 64) - 1


def splitmix64(seed: int, index: int) -> int:
    z = (seed + (index + 1) * 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def point_multiplier(seed: int, i: int) -> int:
    """point i = a_i * G"""
    return splitmix64(seed, i)


def scalar(seed: int, i: int, q: int) -> int:
    """rejection sampling on 32 little-endian bytes masked to the bit length of q (curve-random.ts:151-190)"""
    bits = q.bit_length()
    for attempt in range(64):
        v = 0
        for j in range(4):
            v |= splitmix64(seed ^ 0x5CA1A75, (i * 64 + attempt) * 4 + j) << (64 * j)
        v &= (1 << bits) - 1
        if v < q:
            return v
    return 0

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


# ---------------------------------------------------------------------------------- vectorized (numpy) forms
def splitmix64_np(seed: int, index):
    import numpy as np
    idx = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & MASK64) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def multipliers_np(seed: int, n: int, first: int = 0):
    import numpy as np
    return splitmix64_np(seed, np.arange(first, first + n, dtype=np.uint64))


def scalars_np(seed: int, n: int, q: int, first: int = 0):
    """(n, 4) uint64 little-endian words of scalar(seed, i, q), i in [first, first + n)"""
    import numpy as np
    bits = q.bit_length()
    top_mask = np.uint64((1 << (bits - 192)) - 1)
    qw = [np.uint64((q >> (64 * j)) & MASK64) for j in range(4)]
    idx = np.arange(first, first + n, dtype=np.uint64)
    out = np.zeros((n, 4), dtype=np.uint64)
    todo = np.arange(n)
    for attempt in range(64):
        if len(todo) == 0:
            break
        base = (idx[todo] * np.uint64(64) + np.uint64(attempt)) * np.uint64(4)
        w = np.stack([splitmix64_np(seed ^ 0x5CA1A75, base + np.uint64(j)) for j in range(4)], axis=1)
        w[:, 3] &= top_mask
        # lexicographic compare w < q from the top word
        lt = np.zeros(len(todo), dtype=bool)
        eq = np.ones(len(todo), dtype=bool)
        for j in (3, 2, 1, 0):
            lt |= eq & (w[:, j] < qw[j])
            eq &= w[:, j] == qw[j]
        out[todo[lt]] = w[lt]
        todo = todo[~lt]
    return out


def sum_of_products_mod(s_words, a, q: int) -> int:
    """sum_i s_i * a_i mod q for s given as (n, 4) uint64 words and a as uint64 -- exact, via 16-bit limbs"""
    import numpy as np
    n = len(a)
    s16 = np.zeros((16, n), dtype=np.uint64)
    for j in range(4):
        for h in range(4):
            s16[4 * j + h] = (s_words[:, j] >> np.uint64(16 * h)) & np.uint64(0xFFFF)
    a16 = np.stack([(a >> np.uint64(16 * h)) & np.uint64(0xFFFF) for h in range(4)])
    total = 0
    chunk = 1 << 20   # keep partial sums below 2^64: 2^32 * 2^20 = 2^52
    for i in range(16):
        for j in range(4):
            acc = 0
            for c0 in range(0, n, chunk):
                acc += int(np.dot(s16[i, c0:c0 + chunk], a16[j, c0:c0 + chunk]))
            total += acc << (16 * (i + j))
    return total % q

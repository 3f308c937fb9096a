"""The inequality the filtered scoring form rests on (csrc/score_topk.hip, "Filtered scoring"), checked in numpy: for f32 vectors
u, i with elements rounded to bf16 (round to nearest even) and products accumulated in f32,

    | sum_k bf(u_k) bf(i_k)  -  sum_k u_k i_k |  <=  0.004 |u| |i|      (kFiltDelta)

so that  approx + 0.004 |u| |i|  is an upper bound of the exact score — on random vectors, on vectors built so that every element
sits just below / above its bf16 rounding midpoint (the worst case of the rounding), and on wide dynamic ranges.  The kernel adds the
term with both norms rounded UP to bf16 (and the item norm computed from the rounded row, inflated by 1 + 2^-8): also checked."""
import numpy as np

DELTA = 0.004


def bf16_rne(x: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + np.uint64(0x7FFF) + ((b >> np.uint64(16)) & np.uint64(1))) >> np.uint64(16)) << np.uint64(16)
    return r.astype(np.uint32).view(np.float32)


def bf16_up(x: np.ndarray) -> np.ndarray:          # x >= 0
    b = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + np.where(b & np.uint64(0xFFFF), np.uint64(0x10000), np.uint64(0))) >> np.uint64(16)) << np.uint64(16)
    return r.astype(np.uint32).view(np.float32)


def approx_f32(ub, ib):
    """f32 accumulation of exact bf16 x bf16 products in a fixed order (the MFMA's own order differs: any order is within
    K 2^-24 sum |products|, which the 2 % spare of DELTA covers)."""
    acc = np.zeros(ub.shape[:-1], np.float32)
    for k in range(ub.shape[-1]):
        acc = (acc + (ub[..., k] * ib[..., k]).astype(np.float32)).astype(np.float32)
    return acc


def check(u, i):
    u, i = np.asarray(u, np.float32), np.asarray(i, np.float32)
    ub, ib = bf16_rne(u), bf16_rne(i)
    exact = (u.astype(np.float64) * i.astype(np.float64)).sum(-1)
    appr = approx_f32(ub, ib).astype(np.float64)
    nu = np.sqrt((u.astype(np.float64) ** 2).sum(-1))
    ni = np.sqrt((i.astype(np.float64) ** 2).sum(-1))
    assert np.all(np.abs(appr - exact) <= DELTA * nu * ni * (1 - 0.015) + 1e-300), float(np.max(np.abs(appr - exact) / (nu * ni + 1e-300)))
    # the kernel's term: delta |u| (f32 norm x 1.0009765625, rounded up) times |i| (norm of the ROUNDED row x 1.00390625, rounded up)
    du = bf16_up((np.float32(DELTA) * np.sqrt((u * u).sum(-1, dtype=np.float32)) * np.float32(1.0009765625)).astype(np.float32))
    ni_k = bf16_up((np.sqrt((ib * ib).sum(-1, dtype=np.float32)) * np.float32(1.00390625)).astype(np.float32))
    bound = appr + du.astype(np.float64) * ni_k.astype(np.float64) * (1 - 2.0 ** -20)      # (one more f32 rounding of the sum)
    assert np.all(bound >= exact), float(np.min(bound - exact))
    assert np.all(du.astype(np.float64) >= DELTA * nu * (1 - 1e-6)) and np.all(ni_k.astype(np.float64) >= ni * (1 - 1e-6))


def test_bound_on_random_vectors():
    rng = np.random.default_rng(0)
    for D in (36, 64, 100, 128):
        u = rng.standard_normal((2000, D)).astype(np.float32)
        i = rng.standard_normal((2000, D)).astype(np.float32)
        check(u, i)
        check(u * np.exp(rng.standard_normal((2000, D))).astype(np.float32), i * np.exp(2 * rng.standard_normal((2000, D))).astype(np.float32))
        check(u * 1e-12, i * 1e-12)
        check(u * 1e12, i * 1e12)


def test_bound_at_the_rounding_midpoints():
    """Every element of u just BELOW its bf16 midpoint (rounds down by almost half an ulp), every element of i just ABOVE (rounds
    up), all products of one sign: the rounding errors add up coherently — the worst case the constant has to cover."""
    rng = np.random.default_rng(1)
    for D in (64, 128):
        for sign in (1.0, -1.0):
            m_u = rng.integers(0, 128, (500, D)).astype(np.float64)      # 7 explicit mantissa bits of bf16
            m_i = rng.integers(0, 128, (500, D)).astype(np.float64)
            e_u = rng.integers(-3, 4, (500, D)).astype(np.float64)
            e_i = rng.integers(-3, 4, (500, D)).astype(np.float64)
            u = (1 + m_u / 128 + (1 / 256) * (1 - 2.0 ** -10)) * 2.0 ** e_u       # just below the midpoint between two bf16 values
            i = (1 + m_i / 128 + (1 / 256) * (1 + 2.0 ** -10)) * 2.0 ** e_i       # just above
            check(u.astype(np.float32), (sign * i).astype(np.float32))
            check((sign * i).astype(np.float32), u.astype(np.float32))


def test_bf16_helpers():
    mid = np.float32(1.00390625)                    # halfway between the bf16 values 1 and 1 + 2^-7
    x = np.array([1.0, mid, np.nextafter(mid, np.float32(0)), np.nextafter(mid, np.float32(2)), 3.0e38, 0.0, 1e-30], np.float32)
    r = bf16_rne(x)
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0 and r[3] == np.float32(1.0078125)      # ties to even, below, above the midpoint
    up = bf16_up(np.array([1.0, 1.0000001, 2.5, 0.0], np.float32))
    assert up[0] == 1.0 and up[1] == np.float32(1.0078125) and up[2] == 2.5 and up[3] == 0.0
    assert np.all(bf16_up(np.abs(x)) >= np.abs(x))

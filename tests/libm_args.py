"""Argument sets for the bit-identity checks of exp / log / pow (tests/test_libm_bits.py on the CPU,
test_gpu_parity.py::test_exp_log_pow_are_bit_identical_to_libm on the device): wide ranges, the ranges the kernels
call the functions on, the neighbourhoods of every branch of the algorithms, and the special values."""
import numpy as np

SPECIAL = np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 3.0, -3.0, 2.5, -2.5, np.inf, -np.inf, np.nan, 5e-324, 1e-310,
                    -1e-310, 1e308, -1e308, 1e-20, -1e-20, 1e20, 2.0 ** 53, 2.0 ** 53 + 2, 2.0 ** 52 + 1, 1e-300, -7.0, 0.3,
                    1075.0, -1075.0, 1e-19, 5e-20, 709.78, 709.79, 710.0, -745.1, -745.2, -746.0, -708.3, -708.5, 1e-17,
                    2.0 ** -54, 2.0 ** -55, 512.0, -512.0, 1024.0, -1024.0, 1023.9, 0.9375, 1.0646972656250, 1 + 2.0 ** -52,
                    1 - 2.0 ** -53, 1.7e308, 2.2250738585072014e-308, 4.4501477170144023e-308, 1.0 / 3.0, 0.286, 1.5])


def exp_sets(rng, n):
    """(name, x) pairs; n arguments per random set"""
    yield "wide", rng.uniform(-750.0, 720.0, n)
    yield "moderate", rng.uniform(-40.0, 40.0, n)
    yield "decay factors exp(-dt / tau)", -rng.uniform(0.0, 1.0, n) * 10.0 ** rng.uniform(-6.0, 2.5, n)
    yield "sedi slip term exp(-0.87 / K)", -0.87 / 10.0 ** rng.uniform(-4.0, 3.0, n)
    yield "tiny", rng.standard_normal(n) * 10.0 ** rng.uniform(-20.0, -1.0, n)
    yield "subnormal results", rng.uniform(-745.2, -707.0, n)
    yield "near overflow", rng.uniform(700.0, 709.79, n)
    yield "special", SPECIAL


def log_sets(rng, n):
    yield "wide", np.exp(rng.uniform(-700.0, 700.0, n))
    yield "uniforms of the Box-Muller radius", rng.integers(1, 2 ** 63, n, dtype=np.int64).astype(np.float64) * 2.0 ** -63
    yield "small uniforms", rng.integers(1, 2 ** 40, n, dtype=np.int64).astype(np.float64) * 2.0 ** -64
    yield "around one", rng.uniform(0.93, 1.07, n)
    yield "pressure ratios P0 / p", 1013.25 / 10.0 ** rng.uniform(-2.0, 3.05, n)
    yield "subnormal", rng.integers(1, 2 ** 52, n, dtype=np.int64).view(np.float64)
    yield "special", SPECIAL


def pow_sets(rng, n):
    yield "wide", (np.exp(rng.uniform(-50.0, 50.0, n)), rng.uniform(-20.0, 20.0, n))
    yield "large exponents", (rng.uniform(0.5, 2.0, n), rng.uniform(-1200.0, 1200.0, n))
    yield "huge range", (np.exp(rng.uniform(-700.0, 700.0, n)), rng.uniform(-2.0, 2.0, n))
    yield "x^1.5 of sedi", (rng.uniform(0.4, 1.3, n), np.full(n, 1.5))
    yield "potential temperature (1000 / p)^kappa", (1000.0 / 10.0 ** rng.uniform(-2.0, 3.05, n), np.full(n, 0.286))
    yield "cube roots of the closure", (10.0 ** rng.uniform(-12.0, 4.0, n), np.full(n, 1.0 / 3.0))
    for e in (-1.0 / 3.0, 0.175, -0.65, 0.207, -0.586, 0.8):
        yield "closure exponent %g" % e, (rng.uniform(1e-6, 1.0, n // 4), np.full(n // 4, e))
    yield "wet deposition powers", (10.0 ** rng.uniform(-8.0, 2.0, n), rng.uniform(0.1, 2.0, n))
    yield "negative bases, integer exponents", (-rng.integers(1, 50, n).astype(np.float64), rng.integers(-30, 30, n).astype(np.float64))
    yield "subnormal bases", (rng.integers(1, 2 ** 52, n // 4, dtype=np.int64).view(np.float64), rng.uniform(-1.0, 1.0, n // 4))
    X, Y = np.meshgrid(SPECIAL, SPECIAL)
    yield "special", (X.ravel().copy(), Y.ravel().copy())


def same_bits(a, b):
    """element-wise: identical bit patterns, or both NaN"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return (a.view(np.uint64) == b.view(np.uint64)) | (np.isnan(a) & np.isnan(b))


def sincos_sets(rng, n):
    """Arguments of cos / sin inside the range the restatement covers (high word below 0x400368fd, |x| < 2.42626):
    latitudes in radians, both table branches, the Taylor branch below 0.126, the co-function branch next to pi / 2."""
    import numpy as np
    lim = float.fromhex("0x1.368fcffffffffp+1")
    sets = [("whole range", rng.uniform(-lim, lim, 2 * n)),
            ("latitudes", np.deg2rad(rng.uniform(-90.0, 90.0, 2 * n))),
            ("small", rng.uniform(-0.2, 0.2, n)),
            ("next to pi/2", (np.pi / 2 + rng.uniform(-0.3, 0.3, n)) * rng.choice([-1.0, 1.0], n)),
            ("tiny", rng.uniform(-1.0, 1.0, n // 4) * 2.0 ** rng.integers(-60, -2, n // 4)),
            ("edges", np.array([0.0, -0.0, 0.126, -0.126, 0.12599999, 0.855469, 0.8554687, 0.8554686903953552, lim, -lim,
                                1.5707963267948966, -1.5707963267948966, 1.5707963267948968, 2.0 ** -27, 2.0 ** -26,
                                1e-300, np.deg2rad(89.999), np.deg2rad(-89.999)]))]
    return [(name, np.ascontiguousarray(x, dtype=np.float64)) for name, x in sets]

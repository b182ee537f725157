"""time_cosf_rev (csrc/aggregate.hip) restated in numpy float32 arithmetic, against float64 cos: the all-float32
argument reduction for |x| up to 3e8 (1 / 2pi split into two floats, the large product's FMA rounding error carried
separately).  Prints the largest error of cos(2 pi rev) -- the hardware's v_cos_f32 adds its own ~1e-6 on top."""
import numpy as np
c = 1.0 / (2 * np.pi)
c1 = np.float32(c); c2 = np.float32(c - np.float64(c1))
assert float(c1).hex() == "0x1.45f3060000000p-3" and float(c2).hex() == "0x1.b939100000000p-28"


def rev(x):
    x = x.astype(np.float32)
    x64 = x.astype(np.float64)
    p1 = (x * c1).astype(np.float32); e1 = (x64 * np.float64(c1) - p1.astype(np.float64)).astype(np.float32)   # fma(x, c1, -p1)
    t = (x64 * np.float64(c2) + e1.astype(np.float64)).astype(np.float32)                                      # fma(x, c2, e1)
    return ((p1 - np.floor(p1)).astype(np.float32) + t).astype(np.float32)


rng = np.random.RandomState(1)
xs = np.concatenate([rng.uniform(0, 3e8, 400000), rng.uniform(0, 4e6, 200000), 10 ** rng.uniform(-3, 8.5, 400000),
                     -rng.uniform(0, 3e8, 100000)]).astype(np.float32)
err = np.abs(np.cos(2 * np.pi * rev(xs).astype(np.float64)) - np.cos(xs.astype(np.float64)))
print("max |err| %.2e at x = %g, mean %.2e" % (err.max(), xs[err.argmax()], err.mean()))
assert err.max() < 1.5e-6

"""Pin of the argsort restatements against numba's OWN source file.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): used by
tests/golden/gen_golden.py --check and tests/test_oracle_golden.py, in the
build container only -- the GPU box has no numba tree and the test skips there.

numba cannot be imported here, but
``numba/misc/quicksort.py`` of the copy on disk (0.54.1) is pure Python: with a
stub for ``numba.core.types`` (``intp`` = int), ``wrap`` = identity and
``lt(a, b) = isnan(b) or a < b`` (numba/np/arrayobj.py:5210-5211)
``make_quicksort_impl(wrap, lt, is_argsort=True).run_quicksort`` IS the routine
the reference's ``np.argsort`` compiles to (reference call sites
utils/util.py:258,555,658,762,851).  ``compare()`` runs it beside
``numba_semantics.numba_argsort`` (the restatement the fixtures were generated
with) and, when given, the C oracle's ``zo_numba_argsort`` on tie-heavy arrays.
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np

CANDIDATES = [os.environ.get("NUMBA_QUICKSORT_PY", ""),
              "/opt/conda/lib/python3.9/site-packages/numba/misc/quicksort.py"]


def quicksort_path():
    for p in CANDIDATES:
        if p and os.path.isfile(p):
            return p
    return None


def load_numba_argsort(path=None):
    """-> callable(values float64[n]) -> index array, straight from numba's quicksort.py."""
    path = path or quicksort_path()
    if path is None:
        raise FileNotFoundError("no numba/misc/quicksort.py on this machine")
    stub = types.ModuleType("numba.core.types")
    stub.intp = int
    saved = {k: sys.modules.get(k) for k in ("numba", "numba.core", "numba.core.types")}
    pkg = types.ModuleType("numba")
    core = types.ModuleType("numba.core")
    pkg.core, core.types = core, stub
    sys.modules.update({"numba": pkg, "numba.core": core, "numba.core.types": stub})
    try:
        spec = importlib.util.spec_from_file_location("_numba_quicksort_on_disk", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v

    def lt(a, b):                                   # numba/np/arrayobj.py:5210-5211 (float keys)
        return math.isnan(b) or a < b

    impl = mod.make_quicksort_impl(lambda f: f, lt=lt, is_argsort=True)
    return lambda values: np.asarray(impl.run_quicksort(np.ascontiguousarray(values, np.float64)))


def tie_heavy_arrays(seed=0, sizes=None, per_size=40):
    """Arrays shaped like the reference's candidate lists: few distinct weights (products of the same
    factors), two sorted runs + a maximum, all-equal, NaN-free; n covers the insertion-sort range (< 16),
    the streaming lists (<= 2k+1 = 41 / 81 / 127) and the pruning lists (<= 110 / 420)."""
    rng = np.random.RandomState(seed)
    sizes = sizes or (list(range(1, 130)) + [155, 420])
    for n in sizes:
        for t in range(per_size):
            kind = t % 5
            if kind == 0:
                a = rng.randint(0, max(2, n // 4), n).astype(np.float64)
            elif kind == 1:
                a = 0.5 ** rng.randint(0, 6, n) * 0.9 ** rng.randint(0, 3, n)
            elif kind == 2:                          # two ascending runs + the new key (streaming merge shape)
                h = n // 2
                a = np.concatenate([np.sort(rng.randint(0, 8, h)), np.sort(rng.randint(0, 8, n - h))]).astype(np.float64)
                a[-1] = 9.0
            elif kind == 3:
                a = np.full(n, 0.25)
                a[rng.randint(0, n, max(1, n // 8))] = 0.5
            else:
                a = rng.random_sample(n)
                a[rng.randint(0, n, n // 2)] = a[0]
            yield a


def compare(extra=None, seed=0, per_size=40):
    """-> (arrays checked, mismatches).  ``extra``: further argsort callables (e.g. the C oracle's)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numba_semantics as ns
    ref = load_numba_argsort()
    others = [ns.numba_argsort] + list(extra or [])
    n = bad = 0
    for a in tie_heavy_arrays(seed, per_size=per_size):
        want = ref(a)
        n += 1
        for f in others:
            if not np.array_equal(np.asarray(f(a), np.int64), want):
                bad += 1
    return n, bad

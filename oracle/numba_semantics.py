"""Restatement of the two pieces of numba arithmetic the reference relies on.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by
tests/golden/gen_golden.py and by tests/, never by zebra_amd/.

The reference calls ``np.argsort(values)[-k:]`` (utils/util.py:258,555,658,762,
851) and ``pow(beta, n_ngh)`` (utils/util.py:208) inside ``@jitclass`` code, so
the arithmetic is numba's, not NumPy's / libm's:

* ``np.argsort`` on a float64 array = numba's argsort quicksort
  (numba 0.54.1, numba/misc/quicksort.py ``make_quicksort_impl(is_argsort=True)``
  with ``lt(a, b) = isnan(b) or a < b`` from numba/np/arrayobj.py:5210).
  Third-party dependency, *version unpinned by the reference* (no requirements
  file); 0.54.1 is the copy on disk in the build container.  Published
  algorithm, restated here:
    - index array R = arange(n); explicit stack of (low, high) ranges;
    - while a range has high - low >= 15: median-of-three on R[low], R[mid],
      R[high] (mid = (low+high)>>1; three conditional swaps), pivot value =
      A[R[mid]], pivot stashed at ``high``, Hoare scan (i up while < pivot, j
      down while pivot <), pivot swapped back to i; the LARGER side is pushed,
      the loop continues on the smaller;
    - remaining range (< 16 entries): insertion sort with strict ``lt`` (stable).
  The result is NOT a stable sort: the order of equal weights is decided by the
  partition dynamics, and the reference's row order and top-k membership under
  ties inherit it.

* ``pow(float64, int)`` = numba's ``int_power_impl``
  (numba/cpython/numbers.py:207-243): square-and-multiply for
  |exp| <= 0x10000, libm ``pow`` beyond.
"""
import math

import numpy as np

SMALL_QUICKSORT = 15


def _lt(a, b):
    return (b != b) or (a < b)


def numba_argsort(values):
    """argsort of a 1-D float array with numba 0.54.1 quicksort semantics."""
    a = [float(x) for x in np.asarray(values).ravel()]
    n = len(a)
    r = list(range(n))
    if n < 2:
        return np.asarray(r, dtype=np.int64)
    stack = [(0, n - 1)]
    while stack:
        low, high = stack.pop()
        while high - low >= SMALL_QUICKSORT:
            # --- partition(low, high) ---
            mid = (low + high) >> 1
            if _lt(a[r[mid]], a[r[low]]):
                r[low], r[mid] = r[mid], r[low]
            if _lt(a[r[high]], a[r[mid]]):
                r[high], r[mid] = r[mid], r[high]
            if _lt(a[r[mid]], a[r[low]]):
                r[low], r[mid] = r[mid], r[low]
            pivot = a[r[mid]]
            r[high], r[mid] = r[mid], r[high]
            i = low
            j = high - 1
            while True:
                while i < high and _lt(a[r[i]], pivot):
                    i += 1
                while j >= low and _lt(pivot, a[r[j]]):
                    j -= 1
                if i >= j:
                    break
                r[i], r[j] = r[j], r[i]
                i += 1
                j -= 1
            r[i], r[high] = r[high], r[i]
            # --- push larger side, continue on smaller ---
            if high - i > i - low:
                if high > i:
                    stack.append((i + 1, high))
                high = i - 1
            else:
                if i > low:
                    stack.append((low, i - 1))
                low = i + 1
        # --- insertion sort [low, high] ---
        for p in range(low + 1, high + 1):
            k = r[p]
            v = a[k]
            q = p
            while q > low and _lt(v, a[r[q - 1]]):
                r[q] = r[q - 1]
                q -= 1
            r[q] = k
    return np.asarray(r, dtype=np.int64)


def numba_int_pow(a, b):
    """numba's pow(float64, integer)."""
    a = float(a)
    b = int(b)
    r = 1.0
    if b < 0:
        invert = True
        exp = -b
    else:
        invert = False
        exp = b
    if exp > 0x10000:
        return math.pow(a, float(b))
    while exp != 0:
        if exp & 1:
            r *= a
        exp >>= 1
        a *= a
    return 1.0 / r if invert else r


class NumpyWithNumbaArgsort:
    """Proxy for the ``np`` global of the reference's utils/util.py: every
    attribute is NumPy's except ``argsort``."""

    def __init__(self, real_np):
        self._np = real_np

    def __getattr__(self, name):
        return getattr(self._np, name)

    @staticmethod
    def argsort(values, *args, **kwargs):
        return numba_argsort(values)

"""Pure-Python stand-in for the ``numba`` package.

TEST INFRASTRUCTURE ONLY.  numba / llvmlite cannot be installed in the build
container (no network; the Anaconda copy on disk fails to import), so the
golden-vector generator (tests/golden/gen_golden.py) puts this package on
``sys.path`` to import the reference *source* unmodified.  It provides exactly
the names the reference touches (utils/util.py:3-8,98-134,144,377-391;
modules/embedding_module.py:8,11; modules/memory_updater.py:5,8) and nothing
else.  Every decorator is the identity; typed containers are the CPython ones
(both are insertion-ordered compact dicts / lists, see DESIGN.md "Oracle").

The two places where numba's *arithmetic* differs from CPython/NumPy --
``np.argsort`` (numba's own quicksort) and ``pow(float, int)`` (numba's
square-and-multiply) -- are NOT handled here; gen_golden.py patches them into
the reference module namespace from oracle/numba_semantics.py.

Nothing in the product (zebra_amd/) imports this.
"""
from . import types, typed, experimental  # noqa: F401
from .core import errors  # noqa: F401


def _identity_decorator(*dargs, **dkwargs):
    # @njit / @jit  and  @njit(...) / @jit(nopython=True)
    if len(dargs) == 1 and callable(dargs[0]) and not dkwargs:
        return dargs[0]

    def wrap(fn):
        return fn
    return wrap


njit = _identity_decorator
jit = _identity_decorator


class _TypeOf:
    """Inert result of numba.typeof(...)."""

    def __init__(self, value):
        self.value_type = type(value)

    def __repr__(self):
        return "typeof(%s)" % self.value_type.__name__


def typeof(value):
    return _TypeOf(value)

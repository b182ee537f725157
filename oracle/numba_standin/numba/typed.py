"""numba.typed.List / Dict -> CPython list / dict.

numba's typed Dict is "adapted from CPython 3.7 ... compact and ordered"
(numba/cext/dictobject.c) and Dict.copy() re-inserts in order; List.copy() is
shallow in both implementations.  CPython containers therefore reproduce the
iteration order the reference's results depend on.
"""


class List(list):
    def copy(self):  # shallow, like numba.typed.List.copy
        return List(self)


class Dict(dict):
    @classmethod
    def empty(cls, key_type=None, value_type=None):
        return cls()

    def copy(self):
        return Dict(self)

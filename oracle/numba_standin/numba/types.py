"""Inert type descriptors (only ever passed to jitclass specs / Dict.empty)."""


class _T:
    def __init__(self, name, *args):
        self.name = name
        self.args = args

    def __call__(self, *args):
        return _T(self.name, *args)

    def __repr__(self):
        return self.name


int32 = _T("int32")
int64 = _T("int64")
float32 = _T("float32")
float64 = _T("float64")
boolean = _T("boolean")


def List(dtype, reflected=True):
    return _T("List", dtype)


def ListType(dtype):
    return _T("ListType", dtype)


def Array(dtype, ndim, layout):
    return _T("Array", dtype, ndim, layout)


def DictType(k, v):
    return _T("DictType", k, v)


def Tuple(items):
    return _T("Tuple", items)

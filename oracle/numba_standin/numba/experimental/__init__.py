"""numba.experimental.jitclass -> identity class decorator."""


def jitclass(spec=None):
    if isinstance(spec, type):
        return spec

    def wrap(cls):
        return cls
    return wrap

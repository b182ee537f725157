"""Warning classes the reference silences (utils/util.py:8-12, train.py:16)."""


class NumbaWarning(Warning):
    pass


class NumbaDeprecationWarning(NumbaWarning):
    pass


class NumbaPendingDeprecationWarning(NumbaWarning):
    pass


class NumbaTypeSafetyWarning(NumbaWarning):
    pass

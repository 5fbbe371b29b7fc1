class NumbaTypeSafetyWarning(Warning):
    pass


class NumbaWarning(Warning):
    pass

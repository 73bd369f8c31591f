"""Exception types of the path's consumer, same names and meaning as
``biscuit/errors.py:1-25`` so callers' ``except`` clauses keep working."""


class MatchError(Exception):
    pass


class ModelNotFoundError(MatchError):
    pass


class MultipleModelsFoundError(MatchError):
    pass


class EvalError(Exception):
    pass


class ThresholdError(Exception):
    pass


class ROCFailedError(Exception):
    pass


class PredsContainNaNError(Exception):
    pass

from .builder import TabularModel, build_model  # noqa: F401
from .registry import make_model, split_class_name  # noqa: F401

from . import pytorch  # noqa: F401

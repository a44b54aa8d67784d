from . import conversions  # noqa: F401

from . import modules, heads  # noqa: F401

from . import trianglemesh  # noqa: F401

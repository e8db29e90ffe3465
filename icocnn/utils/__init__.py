from . import ico_geometry  # noqa: F401

"""icocnn.utils.ico_geometry: get_ico_faces, get_icosahedral_grid (see geniconet_amd.geometry)."""
from geniconet_amd.geometry import get_ico_faces, get_icosahedral_grid  # noqa: F401

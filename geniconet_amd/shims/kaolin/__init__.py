"""Stand-in for the one kaolin entry point the reference uses (ico_utils.py:26-44, its test-time metric):
kaolin.metrics.trianglemesh.point_to_mesh_distance, with kaolin 0.9.1's signature and return convention."""
from . import metrics  # noqa: F401

__version__ = '0.9.1'      # ico_utils.py:39 selects the 0.9.1 call by this string

"""point_to_mesh_distance(pointclouds (B, P, 3), vertices (B, V, 3), faces (F, 3)) -> (squared distance (B, P), index of the
closest face (B, P), distance type (B, P)), as kaolin 0.9.1 returns it and as the reference consumes it (ico_utils.py:40-41:
`dist, _, _ = ...; torch.mean(dist)`).  Implementation: geniconet_amd.metrics.point_to_mesh_distance."""
from geniconet_amd.metrics import point_to_mesh_distance  # noqa: F401

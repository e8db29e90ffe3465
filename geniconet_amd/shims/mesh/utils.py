"""`mesh.utils` of the absent PythonFunctions checkout (reference losses.py:7, generate.py:13): the mesh helpers of the
point-to-point loss, in the formulation of geniconet_amd.losses --
  compute_vertex_normals(v (B, N, 3), faces (F, 3)) -> unit vertex normals, face-area weighted (as generate.py:20-43)
  compute_adjacency_matrix_sparse(n, faces) -> (n, n) sparse row-normalised adjacency (row i = 1 / valence at its ring)
  compute_laplacian_batch(v (B, N, 3), adj) -> adj @ v - v         uniform umbrella operator, 'mean-v' convention
  compute_laplacian(v (N, 3), adj) -> the same for one mesh        (generate.py:197 writes target rows 6:9 with it)
Upstream's sign / normalisation of the Laplacian is not known offline; a dataset produced with upstream's generate.py can be
checked with geniconet_amd.data.detect_laplacian_convention.
"""
import torch

from geniconet_amd.losses import compute_vertex_normals  # noqa: F401  (same signature as the reference's call, losses.py:54)


def compute_adjacency_matrix_sparse(num_vertices, faces):
    f = torch.as_tensor(faces).long()
    n = int(num_vertices)
    src = torch.cat((f[:, 0], f[:, 1], f[:, 1], f[:, 2], f[:, 2], f[:, 0]))
    dst = torch.cat((f[:, 1], f[:, 0], f[:, 2], f[:, 1], f[:, 0], f[:, 2]))
    edges = torch.unique(src * n + dst)                       # every directed edge once (an edge lies in two faces)
    i, j = edges // n, edges % n
    deg = torch.zeros(n).index_add_(0, i, torch.ones(len(i)))
    return torch.sparse_coo_tensor(torch.stack((i, j)), 1.0 / deg[i], (n, n)).coalesce()


def compute_laplacian(v, adj):
    return torch.sparse.mm(adj.to(v.dtype), v) - v


def compute_laplacian_batch(v, adj):
    b, n, c = v.shape
    flat = v.transpose(0, 1).reshape(n, b * c)                # one sparse product for the whole batch
    return torch.sparse.mm(adj.to(v.dtype), flat).reshape(n, b, c).transpose(0, 1) - v

"""Geodesic-polyhedron IPE basis (reference MipNeRF360/internal/geopoly.py:78-124), host numpy,
computed once at model construction.  Built from edge/face subdivision of the unit polyhedron and
de-duplicated with a canonical half-space rule, then ordered like the reference (np.unique on first
occurrence index, columns reversed) so the feature order of the trained weights is identical."""
import numpy as np

_PHI = (np.sqrt(5.0) + 1.0) / 2.0
_ICO_V = np.array([(-1, 0, _PHI), (1, 0, _PHI), (-1, 0, -_PHI), (1, 0, -_PHI), (0, _PHI, 1), (0, _PHI, -1),
                   (0, -_PHI, 1), (0, -_PHI, -1), (_PHI, 1, 0), (-_PHI, 1, 0), (_PHI, -1, 0),
                   (-_PHI, -1, 0)]) / np.sqrt(_PHI + 2.0)
_ICO_F = [(0, 4, 1), (0, 9, 4), (9, 5, 4), (4, 5, 8), (4, 8, 1), (8, 10, 1), (8, 3, 10), (5, 3, 8), (5, 2, 3),
          (2, 7, 3), (7, 10, 3), (7, 6, 10), (7, 11, 6), (11, 0, 6), (0, 1, 6), (6, 1, 10), (9, 0, 11),
          (9, 11, 2), (9, 2, 5), (7, 2, 11)]
_OCT_V = np.array([(0, 0, -1), (0, 0, 1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0)], dtype=np.float64)


def _octa_faces():
  """Vertex triples in the reference's enumeration (geopoly.py:113-115): every cube corner
  (itertools.product order) lists its three adjacent octahedron vertices; that flat list of 24 vertex
  ids is read as three rows of eight and the columns are the triples.  (They are not all geometric
  faces; the tesselation only needs the resulting point set and its first-occurrence order.)"""
  adj = []
  for cx in (-1, 1):
    for cy in (-1, 1):
      for cz in (-1, 1):
        c = np.array([cx, cy, cz], dtype=np.float64)
        adj += [k for k in range(6) if abs(np.sum((c - _OCT_V[k])**2) - 2.0) < 1e-12]
  rows = np.array(adj).reshape(3, -1)
  return [tuple(sorted(rows[:, i])) for i in range(rows.shape[1])]


def _subdivide(verts, faces, v):
  pts = []
  bary = [(i, j, v - i - j) for i in range(v + 1) for j in range(v + 1 - i)]
  for f in faces:
    tri = verts[list(f)]
    for b in bary:
      p = (np.array(b, dtype=np.float64) / v) @ tri
      pts.append(p / np.linalg.norm(p))
  return np.array(pts)


def generate_basis(base_shape, angular_tesselation, remove_symmetries=True, eps=1e-4):
  """Returns [n, 3] (the model uses the transpose)."""
  if base_shape == 'icosahedron':
    verts, faces = _ICO_V, _ICO_F
  elif base_shape == 'octahedron':
    verts, faces = _OCT_V, _octa_faces()
  else:
    raise ValueError(f'base_shape {base_shape} not supported')
  if angular_tesselation < 1:
    raise ValueError(f'v {angular_tesselation} must be >= 1')
  pts = _subdivide(verts, faces, int(angular_tesselation))
  keep = []
  for i, p in enumerate(pts):            # first occurrence wins, order of first occurrence kept
    if not any(np.sum((p - pts[k])**2) <= eps for k in keep):
      keep.append(i)
  pts = pts[keep]
  if remove_symmetries:
    out = []
    for i, p in enumerate(pts):          # keep p if some later-or-equal index holds -p (upper triangle rule)
      if any(np.sum((p + pts[k])**2) < eps for k in range(i, len(pts))):
        out.append(i)
    pts = pts[out]
  return pts[:, ::-1].copy()

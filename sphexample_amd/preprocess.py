"""Host-side mirror of the reference's pre-processing (CSV → SimParticles).

Follows /root/reference/src/PreProcess.jl:
* ``LoadSpecificCSV``        :12-43  — columns ``Points:0/1/2``, ``Rhop``, ``Idp`` looked up by
  header name; 2-D takes ``(Points:0, Points:2)``; ``ID = Idp + 1``.
* ``AllocateDataStructures`` :45-119 — concatenates the geometries, derives ``GravityFactor``
  (Fluid −1, Moving +1, Fixed 0), ``MotionLimiter`` (Fluid 1 else 0), ``BoundaryBool``, zero-fills the
  rest and sorts everything by ``ID``.
* ``LoadBoundaryNormals``    :217-243 — ghost node = point + normal.

The 17 field names of the ``SimParticles`` StructArray (src/PreProcess.jl:114) are kept verbatim.
"""
from __future__ import annotations

import csv
import gzip
import io
from typing import List, Sequence

import numpy as np

from .config import Geometry, ParticleType

_FLOAT = {"Float64": np.float64, "Float32": np.float32}

FIELD_NAMES = ("Cells", "ChunkID", "Kernel", "KernelGradient", "Position", "Acceleration",
               "Velocity", "Density", "Pressure", "GravityFactor", "MotionLimiter", "BoundaryBool",
               "ID", "Type", "GroupMarker", "GhostPoints", "GhostNormals")


def _read_csv_columns(path: str, wanted: Sequence[str]) -> dict:
    """Tolerant reader: CRLF, quoted headers and NUL/control bytes in unused columns
    (e.g. input/case_duckling_mdbc/CaseDuckling_Dp0.01_Bound_MDBC.csv) are accepted."""
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    text = raw.replace(b"\x00", b"").decode("utf-8", errors="replace")
    reader = csv.reader(io.StringIO(text))
    header = [h.strip().strip('"') for h in next(reader)]
    idx = {}
    for w in wanted:
        if w not in header:
            raise KeyError(f"{path}: column {w!r} missing (have {header})")
        idx[w] = header.index(w)
    cols = {w: [] for w in wanted}
    for row in reader:
        if not row or all(not c.strip() for c in row):
            continue
        for w in wanted:
            cols[w].append(row[idx[w]])
    return {w: np.asarray(v, dtype=np.float64) for w, v in cols.items()}


def LoadSpecificCSV(dims: int, float_type, particle_type: ParticleType, particle_group_marker: int,
                    specific_csv: str):
    """src/PreProcess.jl:12-43."""
    c = _read_csv_columns(specific_csv, ("Points:0", "Points:1", "Points:2", "Rhop", "Idp"))
    if dims == 3:
        points = np.stack([c["Points:0"], c["Points:1"], c["Points:2"]], axis=1)
    else:
        points = np.stack([c["Points:0"], c["Points:2"]], axis=1)
    n = points.shape[0]
    density = c["Rhop"].astype(float_type)
    types = np.full(n, int(particle_type), dtype=np.uint8)
    group = np.full(n, particle_group_marker, dtype=np.int64)
    idp = c["Idp"].astype(np.int64) + 1
    return points.astype(float_type), density, types, group, idp


class SimParticles:
    """Struct-of-arrays stand-in for the reference's ``StructArray`` (src/PreProcess.jl:114)."""

    def __init__(self, dims: int, float_type, **fields):
        self.Dimensions = dims
        self.FloatType = float_type
        for k in FIELD_NAMES:
            setattr(self, k, fields[k])

    def __len__(self):
        return int(self.Position.shape[0])

    def permute(self, perm: np.ndarray) -> None:
        """What ``sort!`` on the StructArray does: permute every field."""
        for k in FIELD_NAMES:
            setattr(self, k, np.ascontiguousarray(getattr(self, k)[perm]))

    def copy(self) -> "SimParticles":
        return SimParticles(self.Dimensions, self.FloatType,
                            **{k: getattr(self, k).copy() for k in FIELD_NAMES})


def particles_from_arrays(dims: int, position, density, types, group_marker, idp,
                          float_type=np.float64, sort_by_id: bool = True) -> SimParticles:
    """The body of AllocateDataStructures after the CSVs are read (src/PreProcess.jl:72-118)."""
    position = np.ascontiguousarray(position, dtype=float_type).reshape(-1, dims)
    n = position.shape[0]
    types = np.ascontiguousarray(types, dtype=np.uint8)
    density = np.ascontiguousarray(density, dtype=float_type)
    gf = np.zeros(n, dtype=float_type)
    gf[types == ParticleType.Fluid] = -1
    gf[types == ParticleType.Moving] = 1
    ml = np.zeros(n, dtype=float_type)
    ml[types == ParticleType.Fluid] = 1
    zero_v = lambda: np.zeros((n, dims), dtype=float_type)  # noqa: E731
    p = SimParticles(
        dims, float_type,
        Cells=np.zeros((n, dims), dtype=np.int64), ChunkID=np.zeros(n, dtype=np.int64),
        Kernel=np.zeros(n, dtype=float_type), KernelGradient=zero_v(), Position=position,
        Acceleration=zero_v(), Velocity=zero_v(), Density=density,
        Pressure=np.zeros(n, dtype=float_type), GravityFactor=gf, MotionLimiter=ml,
        BoundaryBool=(ml == 0).astype(np.uint8), ID=np.ascontiguousarray(idp, dtype=np.int64),
        Type=types, GroupMarker=np.ascontiguousarray(group_marker, dtype=np.uint64),
        GhostPoints=zero_v(), GhostNormals=zero_v())
    if sort_by_id:
        p.permute(np.argsort(p.ID, kind="stable"))      # sort!(SimParticles, by = p -> p.ID)
    return p


def AllocateDataStructures(SimGeometry: List[Geometry], dims: int = None,
                           float_type="Float64") -> SimParticles:
    """src/PreProcess.jl:45-119."""
    dims = dims or SimGeometry[0].Dimensions
    if dims not in (2, 3):
        raise ValueError("Dimensions must be 2 or 3 (pass dims= or set Geometry.Dimensions)")
    ft = _FLOAT[float_type] if isinstance(float_type, str) else float_type
    pos, rho, typ, grp, idp = [], [], [], [], []
    for geom in SimGeometry:
        a, b, c, d, e = LoadSpecificCSV(dims, ft, geom.Type, geom.GroupMarker, geom.CSVFile)
        pos.append(a); rho.append(b); typ.append(c); grp.append(d); idp.append(e)
    return particles_from_arrays(dims, np.concatenate(pos), np.concatenate(rho), np.concatenate(typ),
                                 np.concatenate(grp), np.concatenate(idp), ft)


def LoadBoundaryNormals(dims: int, float_type, path_mdbc: str):
    """src/PreProcess.jl:217-243 → (points, ghost_points, normals)."""
    ft = _FLOAT[float_type] if isinstance(float_type, str) else float_type
    c = _read_csv_columns(path_mdbc, ("Normal:0", "Normal:1", "Normal:2", "Points:0", "Points:1", "Points:2"))
    if dims == 3:
        normals = np.stack([c["Normal:0"], c["Normal:1"], c["Normal:2"]], axis=1)
        points = np.stack([c["Points:0"], c["Points:1"], c["Points:2"]], axis=1)
    else:
        normals = np.stack([c["Normal:0"], c["Normal:2"]], axis=1)
        points = np.stack([c["Points:0"], c["Points:2"]], axis=1)
    normals = normals.astype(ft)
    points = points.astype(ft)
    return points, points + normals, normals


def LoadMDBCNormals(particles: SimParticles, path: str) -> None:
    """``LoadMDBCNormals!`` (src/SPHCellList.jl:512-524): file row k ↔ k-th particle in ID order."""
    if path is None:
        return
    _, ghost_points, ghost_normals = LoadBoundaryNormals(particles.Dimensions, particles.FloatType, path)
    n = ghost_points.shape[0]
    particles.GhostPoints[:n] = ghost_points
    particles.GhostNormals[:n] = ghost_normals

"""Independent numpy/scipy pair enumeration used to pin the oracle (and, on the GPU box, the engine).

It shares no code with oracle/sph_oracle.c: pairs come from a KD-tree over the positions (valid right
after a cell-list rebuild, when {adjacent cells ∧ r ≤ H} == {r ≤ H}), the physics
is written in vector form from the formulas of /root/reference/src/SPHCellList.jl:273-309,
src/SPHKernels.jl:85-86, src/SPHViscosityModels.jl:64-70, src/SPHDensityDiffusionModels.jl:116-133.
"""
import numpy as np
from scipy.spatial import cKDTree


def cell_of(x, H_inv):
    return (np.sign(x) * np.trunc(np.abs(x) * H_inv + 0.5)).astype(np.int64)


def _lex_key(cells):
    """Total order of CartesianIndex: last axis most significant."""
    D = cells.shape[1]
    c = cells - cells.min(0)
    span = c.max(0) + 1
    key = np.zeros(len(c), dtype=np.int64)
    mult = 1
    for d in range(D):
        key += c[:, d] * mult
        mult *= int(span[d])
    return key


def pair_forces(cfg, pos, vel, rho, rho_n, press, ml, order_index=None):
    """Returns (drhodt, acc) for the state given in *sorted* particle order.

    order_index: position of each particle in the reference's sorted order (defaults to arange) — only
    used for the i/j orientation of the density-diffusion term (quirk Q4)."""
    N, D = pos.shape
    if order_index is None:
        order_index = np.arange(N)
    tree = cKDTree(pos)
    pairs = tree.query_pairs(cfg.H * (1 + 1e-9), output_type="ndarray")
    a, b = pairs[:, 0], pairs[:, 1]
    x = pos[a] - pos[b]
    r2 = (x * x).sum(1)
    keep = r2 <= cfg.H2
    a, b, x, r2 = a[keep], b[keep], x[keep], r2[keep]
    # orientation: "i" = lower sorted index inside a cell, the particle of the LATER cell otherwise
    cells = cell_of(pos, cfg.H_inv)
    key = _lex_key(cells)
    same = key[a] == key[b]
    a_is_i = np.where(same, order_index[a] < order_index[b], key[a] > key[b])
    i = np.where(a_is_i, a, b)
    j = np.where(a_is_i, b, a)
    xij = pos[i] - pos[j]
    q = np.clip(np.sqrt(r2) * cfg.h_inv, 0.0, 2.0)
    fac = cfg.alphaD * 5 * (q - 2) ** 3 / (8 * cfg.h * cfg.h)
    gW = fac[:, None] * xij
    vij = vel[i] - vel[j]
    sym = -(vij * gW).sum(1)
    drho = np.zeros(N)
    acc = np.zeros((N, D))
    np.add.at(drho, i, -rho[i] * (cfg.m0 / rho[j]) * sym)
    np.add.at(drho, j, -rho[j] * (cfg.m0 / rho[i]) * sym)
    if cfg.density_diffusion == 2:
        rhoH = cfg.rho0 * (-cfg.g) * -xij[:, -1] * ((1 / (cfg.Cb * cfg.gamma)) * cfg.rho0)
        psi = 2 * ((rho_n[j] - rho_n[i]) - rhoH)[:, None] * (-xij) / (r2 + cfg.eta2)[:, None]
        Di = cfg.delta_phi * cfg.h * cfg.c0 * (cfg.m0 / rho_n[j]) * (psi * gW).sum(1) * ml[i] * ml[j]
        np.add.at(drho, i, Di)
        np.add.at(drho, j, -Di)
    Pfac = (press[i] + press[j]) / (rho[i] * rho[j])
    um = -cfg.m0 * Pfac[:, None] * gW
    if cfg.viscosity == 1:
        vdx = (vij * xij).sum(1)
        mu = cfg.h * vdx / (r2 + cfg.eta2)
        rbar = 0.5 * (rho_n[i] + rho_n[j])
        k = np.where(vdx < 0, -cfg.m0 * (-cfg.alpha * cfg.c0 * mu) / rbar, 0.0)
        um = um + k[:, None] * gW
    np.add.at(acc, i, um)
    np.add.at(acc, j, -um)
    return drho, acc, len(a)


def eos(cfg, rho):
    return ((cfg.c0 ** 2 * cfg.rho0) / 7) * ((rho / cfg.rho0) ** 7 - 1)

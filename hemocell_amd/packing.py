"""Deterministic synthetic cell packing for benchmark / test domains (replaces tools/packCells for the
synthetic pipeflow inputs of SURVEY.md §8d: a jittered grid, seed fixed, cells kept only when they lie
inside the lumen).  Pure host-side input generation; positions are in lattice units."""
import numpy as np


def pack_pipe_rbc(nx, ny, nz, hematocrit, seed=12345, rbc_volume_lu=649.0, x0=0, nx_global=None):
    """RBC centres + Euler angles (degrees, .pos convention) for a pipe along x of radius (ny-2)/2.

    RBCs are discs ~16 lu across and ~5 lu thick; they are laid flat (thin axis along z after the
    (90,0,0) rotation of the reference's .pos files) on a grid whose pitch is chosen to hit the requested
    hematocrit, with +-1 lu jitter and +-8 degree tilt.  Only cells whose centre lies in
    [x0, x0+nx) are returned so that slabs of a larger domain get disjoint, consistent subsets."""
    nxg = nx_global if nx_global is not None else nx
    R = (ny - 2) / 2.0
    cy, cz = (ny - 1) / 2.0, (nz - 1) / 2.0
    lumen = np.pi * R * R * nxg
    target = hematocrit * lumen / rbc_volume_lu
    # in-plane pitch fixed by the disc size, axial pitch from the target count
    px = py = 19.0
    pz_min = 7.0
    # count of usable (y,z,x) sites for a given pz; choose pz to meet the target
    best = None
    for pz in np.arange(pz_min, 40.0, 0.25):
        ys = np.arange(cy - np.floor((R - 9) / py) * py, cy + R, py)
        zs = np.arange(cz - np.floor((R - 4) / pz) * pz, cz + R, pz)
        xs = np.arange(px / 2, nxg - px / 2 + 1e-9, px)
        n = 0
        for y in ys:
            for z in zs:
                # the whole disc (radius 8.5 in x-y, half thickness 3 in z) must stay inside R-2
                if np.hypot(abs(y - cy) + 8.5, abs(z - cz) + 3.0) < R - 2.0:
                    n += 1
        n *= len(xs)
        if best is None or abs(n - target) < abs(best[0] - target):
            best = (n, pz, ys, zs, xs)
    n, pz, ys, zs, xs = best
    rng = np.random.default_rng(seed)
    centres, angles = [], []
    for ix, x in enumerate(xs):
        for y in ys:
            for z in zs:
                if not np.hypot(abs(y - cy) + 8.5, abs(z - cz) + 3.0) < R - 2.0:
                    continue
                j = rng.uniform(-0.75, 0.75, 3)
                j[2] = rng.uniform(-0.4, 0.4)
                a = rng.uniform(-8.0, 8.0, 3)
                c = np.array([x, y, z]) + j
                if x0 <= c[0] < x0 + nx:
                    centres.append(c)
                    angles.append(np.array([90.0, 0.0, 0.0]) + a)
    return np.array(centres).reshape(-1, 3), np.array(angles).reshape(-1, 3)

"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _random_populations(rng, n, amp=0.02):
    # fBar = f - t_i around rest equilibrium (0) with a smooth-ish random perturbation
    return amp * rng.standard_normal((n, 19)) * np.array(O_T())[None, :]


def O_T():
    return [1 / 3] + [1 / 18] * 3 + [1 / 36] * 6 + [1 / 18] * 3 + [1 / 36] * 6


def _both_lattices(orc, gpu, nx, ny, nz, periodic, omega, mask=None):
    Lo = O.OracleLattice(orc, nx, ny, nz, periodic, omega)
    Lg = gpu.Lattice(nx, ny, nz, periodic, omega)
    if mask is not None:
        Lo.set_mask(mask)
        Lg.defineBounceBack(mask)
    return Lo, Lg


@pytest.mark.parametrize("periodic", [(1, 1, 1), (1, 0, 0), (0, 0, 0), (0, 1, 1)])
def test_collide_stream_bit_exact_random_state(orc, gpu, periodic):
    """fluid-only steps from a random state with a body force: populations identical to the oracle, bit for bit"""
    rng = np.random.default_rng(7)
    nx, ny, nz = 12, 10, 14
    mask = np.zeros((nx, ny, nz), np.uint8)
    for d, per in enumerate(periodic):
        if not per:  # walls on non-periodic faces (the only configuration the reference uses)
            sl = [slice(None)] * 3
            sl[d] = 0; mask[tuple(sl)] = 1
            sl[d] = -1; mask[tuple(sl)] = 1
    mask[5, 4:6, 6:9] = 1  # an obstacle inside
    Lo, Lg = _both_lattices(orc, gpu, nx, ny, nz, periodic, 1.0 / 0.9, mask)
    f0 = _random_populations(rng, nx * ny * nz)
    Lo.f[:] = f0
    Lg.set_populations(f0)
    F = (1e-5, -2e-6, 3e-6)
    Lo.set_force_uniform(F); Lg.setExternalVector(F)
    np.testing.assert_array_equal(Lg.populations()[mask.reshape(-1) == 0], f0[mask.reshape(-1) == 0])
    for steps in (1, 9):
        Lo.collide_stream(steps); Lg.collideAndStream(steps)
        fo, fg = Lo.f.copy(), Lg.populations()
        fluid = mask.reshape(-1) == 0
        assert np.array_equal(fg[fluid], fo[fluid]), np.abs(fg[fluid] - fo[fluid]).max()
    Lo.destroy(); Lg.destroy()


def test_pipe_flow_bit_exact_and_poiseuille(orc, gpu):
    """body-force driven flow in the analytic cylinder (BounceBack), x periodic: bit-exact vs oracle"""
    nx, ny, nz = 6, 34, 34
    mask, R = gpu.pipe_mask(nx, ny, nz)
    Lo, Lg = _both_lattices(orc, gpu, nx, ny, nz, (1, 0, 0), 1.0 / 1.1, mask)
    Lo.init_equilibrium(); Lg.latticeEquilibrium(1.0, (0, 0, 0))
    F = (1e-6, 0, 0)
    Lo.set_force_uniform(F); Lg.setExternalVector(F)
    Lo.set_threads(8)
    Lo.collide_stream(300); Lg.collideAndStream(300)
    fluid = mask.reshape(-1) == 0
    fo, fg = Lo.f.copy(), Lg.populations()
    assert np.array_equal(fg[fluid], fo[fluid])
    rho, u = Lg.rho_u()
    ux = u[:, 0].reshape(nx, ny, nz)
    assert ux[0, ny // 2, nz // 2] > 0 and np.abs(ux[0] - ux[3]).max() < 1e-15
    Lo.destroy(); Lg.destroy()


def _types(orc, gpu, P_o, P_g):
    To = O.make_rbc(orc, P_o)
    Tg = gpu.CellType.rbc(P_g)
    return To, Tg


def test_parameters_and_tables_match_oracle(orc, gpu):
    Po = O.make_params(orc)
    Pg = gpu.base_parameters()
    for n, _ in O.Params._fields_:
        assert getattr(Po, n) == getattr(Pg, n), n
    for make_o, make_g in ((O.make_rbc, gpu.CellType.rbc), (O.make_plt, gpu.CellType.plt)):
        To = make_o(orc, Po); Tg = make_g(Pg)
        t = To.contents; g = Tg.tables()
        assert (t.nv, t.nt, t.ne) == (Tg.nv, Tg.nt, Tg.ne)
        for name in ("vertices", "triangles", "edges", "edge_length_eq", "edge_angle_eq", "triangle_area_eq",
                     "vertex_vertexes", "patch_dist_eq"):
            assert np.array_equal(t.arr(name), g[name]), name
        for name in ("volume_eq", "area_mean_eq", "edge_mean_eq", "angle_mean_eq", "k_volume", "k_area", "k_link",
                     "k_bend", "eta_m"):
            assert getattr(t, name) == g[name], name
        orc.orc_celltype_destroy(To); Tg.destroy()


def _oracle_forces(orc, T, pos, vel, flags=0x1f, comp=False):
    nv = T.contents.nv
    ncell = pos.shape[0] // nv
    out = np.zeros_like(pos)
    comps = np.zeros((6, pos.shape[0], 3)) if comp else None
    for c in range(ncell):
        p = np.ascontiguousarray(pos[c * nv:(c + 1) * nv]); v = np.ascontiguousarray(vel[c * nv:(c + 1) * nv])
        f = np.zeros_like(p)
        if comp:
            cc = np.zeros((6, nv, 3))
            orc.orc_cell_forces(T, O.dptr(p), O.dptr(v), O.dptr(f), O.dptr(cc), flags)
            comps[:, c * nv:(c + 1) * nv] = cc
        else:
            orc.orc_cell_forces(T, O.dptr(p), O.dptr(v), O.dptr(f), None, flags)
        out[c * nv:(c + 1) * nv] = f
    return (out, comps) if comp else out


def _place_cells(gpu, cells, t, centres, angles):
    for c, a in zip(centres, angles):
        assert cells.addCell(t, c, a)


@pytest.mark.parametrize("kind", ["rbc", "plt", "rbc_visc"])
def test_membrane_forces_vs_oracle(orc, gpu, kind):
    """per-vertex membrane forces of deformed cells: RBC bit-exact (gather form reproduces the scatter
    order), PLT to 1e-12 relative (atan2 is not the same libm)"""
    rng = np.random.default_rng(11)
    Po = O.make_params(orc); Pg = gpu.base_parameters()
    if kind == "plt":
        To = O.make_plt(orc, Po, eta_m=1e-9); Tg = gpu.CellType.plt(Pg, eta_m=1e-9)
    elif kind == "rbc_visc":
        To = O.make_rbc(orc, Po, eta_m=5e-10); Tg = gpu.CellType.rbc(Pg, eta_m=5e-10)
    else:
        To = O.make_rbc(orc, Po); Tg = gpu.CellType.rbc(Pg)
    nx, ny, nz = 64, 40, 40
    L = gpu.Lattice(nx, ny, nz, (1, 1, 1), 1.0)
    cells = gpu.Cells(L, Pg)
    t = cells.addCellType(Tg, 1)
    centres = [(12.3, 20.1, 19.7), (33.0, 18.4, 22.2), (50.5, 21.0, 17.9)]
    angles = [(0, 0, 0), (35.0, 10.0, -70.0), (90.0, 45.0, 20.0)]
    _place_cells(gpu, cells, t, centres, angles)
    pos = cells.positions
    pos += 0.05 * rng.standard_normal(pos.shape)
    vel = 1e-3 * rng.standard_normal(pos.shape)
    cells.positions = pos; cells.velocities = vel
    cells.applyConstitutiveModel(0, True)
    fg = cells.forces
    fo = _oracle_forces(orc, To, pos, vel)
    scale = np.abs(fo).max()
    if kind == "plt":
        assert np.abs(fg - fo).max() <= 1e-12 * scale
    else:
        assert np.array_equal(fg, fo), np.abs(fg - fo).max() / scale
    # separate force vectors (output mode)
    comp = cells.force_components(t)
    _, co = _oracle_forces(orc, To, pos, vel, comp=True)
    assert np.abs(comp - co).max() <= 1e-12 * scale
    assert np.abs(comp.sum(0) - fg).max() <= 1e-12 * scale
    cells.destroy(); L.destroy(); Tg.destroy()


def _sim_pair(orc, gpu, nx, ny, nz, periodic, mask, dt=1e-7, plt=False, k_m=1, k_p=1):
    Po = O.make_params(orc, dt=dt); Pg = gpu.base_parameters(dt=dt)
    omega = 1.0 / Po.tau
    Lo, Lg = _both_lattices(orc, gpu, nx, ny, nz, periodic, omega, mask)
    Lo.init_equilibrium(); Lg.latticeEquilibrium()
    So = orc.orc_sim_create(Lo.ptr, C.byref(Po))
    hg = gpu.HemoCell(Lg, Pg)
    To = O.make_rbc(orc, Po); Tg = gpu.CellType.rbc(Pg)
    To.contents.timescale = k_m
    orc.orc_sim_add_type(So, To); hg.cellfields.addCellType(Tg, k_m)
    if plt:
        Tpo = O.make_plt(orc, Po); Tpg = gpu.CellType.plt(Pg)
        Tpo.contents.timescale = k_m
        orc.orc_sim_add_type(So, Tpo); hg.cellfields.addCellType(Tpg, k_m)
    So.contents.particle_velocity_timescale = k_p
    hg.setParticleVelocityUpdateTimeScaleSeparation(k_p)
    return Po, Lo, Lg, So, hg


def _add_both(orc, So, hg, t, centre, angles_deg):
    c = np.array(centre, dtype=np.float64)
    a = -np.array(angles_deg, dtype=np.float64) * np.pi / 180.0
    a_ref = np.array(angles_deg, dtype=np.float64) * (3.14159265358979323846 / 180.0) * -1.0
    po = orc.orc_sim_add_cell(So, t, O.dptr(c), O.dptr(a_ref), 0.0)
    pg = hg.cellfields.addCell(t, centre, angles_deg)
    assert bool(po) == bool(pg)
    return bool(pg)


def _oracle_state(orc, So):
    n = So.contents.np
    out = [np.zeros((n, 3)) for _ in range(3)]
    for w in range(3):
        orc.orc_sim_get(So, w, O.dptr(out[w]))
    return out  # pos, vel, force


def test_ibm_phases_vs_oracle(orc, gpu):
    """spread -> collide -> interpolate -> advance -> mechanics, one phase at a time, near walls and across
    the periodic seam"""
    nx, ny, nz = 40, 30, 30
    mask, R = gpu.pipe_mask(nx, ny, nz)
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 0, 0), mask)
    assert _add_both(orc, So, hg, 0, (38.5, 14.2, 15.1), (90, 0, 0))     # straddles the periodic seam
    assert _add_both(orc, So, hg, 0, (18.0, 14.5, 9.6), (90, 20, 0))     # near the wall
    assert not _add_both(orc, So, hg, 0, (18.0, 14.5, 4.0), (0, 0, 0))   # touches the wall: rejected by both
    cf = hg.cellfields
    pos_o, _, _ = _oracle_state(orc, So)
    assert np.array_equal(cf.positions, pos_o)
    # deform + initial mechanics
    rng = np.random.default_rng(3)
    pos = pos_o + 0.03 * rng.standard_normal(pos_o.shape)
    orc.orc_sim_set(So, 0, O.dptr(pos)); cf.positions = pos
    orc.orc_sim_mechanics(So, 1); cf.applyConstitutiveModel(0, True)
    _, _, f_o = _oracle_state(orc, So)
    assert np.array_equal(cf.forces, f_o)
    F = (2e-6, 0, 0)
    Lo.set_force_uniform(F); Lg.setExternalVector(F)
    # spread
    orc.orc_sim_spread(So); cf.spreadParticleForce(True)
    Fo = Lo.force.copy() - np.array(F)[None, :]
    Fg = Lg.ibm_force()
    fluid = mask.reshape(-1) == 0
    assert np.abs(Fg[fluid] - Fo[fluid]).max() <= 1e-14 * np.abs(Fo).max()
    assert np.abs(Fg.sum(0) - cf.forces.sum(0)).max() <= 1e-12 * np.abs(cf.forces).sum()
    # collide + interpolate
    orc.orc_collide_stream(Lo.ptr); Lg.collideAndStream(1)
    assert np.abs(Lg.populations()[fluid] - Lo.f[fluid]).max() <= 1e-15
    orc.orc_sim_interpolate(So); cf.interpolateFluidVelocity()
    _, v_o, _ = _oracle_state(orc, So)
    assert np.abs(cf.velocities - v_o).max() <= 1e-13 * np.abs(v_o).max()
    # advance + mechanics
    orc.orc_sim_advance(So); cf.advanceParticles(True)
    p_o, _, _ = _oracle_state(orc, So)
    assert np.abs(cf.positions - p_o).max() <= 1e-13
    Lo.destroy(); Lg.destroy()


@pytest.mark.parametrize("scale,which", [(1.2, "node list"), (1.5, "tile")])
def test_ibm_oversized_cells_take_the_fallback_paths(orc, gpu, scale, which):
    """the LDS-tiled spread / interpolate kernels hold one cell's stencil box (18^3 nodes) and node list (1536 nodes);
    a cell blown up beyond either bound is handled by the per-vertex path inside the same kernels.  Both bounds are
    crossed here and the result is checked against the oracle like the tiled path is."""
    nx = ny = nz = 48
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 1, 1), None)
    assert _add_both(orc, So, hg, 0, (24.3, 23.6, 24.9), (30, 50, 10))
    assert _add_both(orc, So, hg, 0, (2.0, 40.0, 12.0), (90, 0, 0))        # ordinary cell across the periodic seam
    cf = hg.cellfields
    pos, _, _ = _oracle_state(orc, So)
    nv = 642
    c0 = pos[:nv].mean(0)
    pos[:nv] = c0 + scale * (pos[:nv] - c0)
    b = np.floor(pos[:nv]).astype(int)
    ext = b.max(0) - b.min(0) + 2
    nodes = len({tuple(r + np.array(d)) for r in b for d in np.ndindex(2, 2, 2)})
    if which == "tile":
        assert ext.prod() > 5832
    else:
        assert ext.prod() <= 5832 and nodes > 1536
    rng = np.random.default_rng(11)
    frc = 1e-4 * rng.standard_normal(pos.shape)
    orc.orc_sim_set(So, 0, O.dptr(pos)); cf.positions = pos
    orc.orc_sim_set(So, 2, O.dptr(frc)); cf.forces = frc
    F = (1e-6, 2e-6, -1e-6)
    Lo.set_force_uniform(F); Lg.setExternalVector(F)
    f0 = _random_populations(rng, nx * ny * nz, 0.01)
    Lo.f[:] = f0; Lg.set_populations(f0)
    orc.orc_sim_spread(So); cf.spreadParticleForce(True)
    Fo = Lo.force.copy() - np.array(F)[None, :]
    Fg = Lg.ibm_force()
    assert np.abs(Fg - Fo).max() <= 1e-14 * np.abs(Fo).max()
    orc.orc_collide_stream(Lo.ptr); Lg.collideAndStream(1)
    orc.orc_sim_interpolate(So); cf.interpolateFluidVelocity()
    _, v_o, _ = _oracle_state(orc, So)
    assert np.abs(cf.velocities - v_o).max() <= 1e-13 * np.abs(v_o).max()
    Lo.destroy(); Lg.destroy()


def test_face_plane_velocities_vs_oracle(orc, gpu):
    """the message of a velocity update between slabs (hcl_face_velocity_pack): u = j/rho + F/2 on the two face planes, evaluated
    by their owner -- against Cell::computeVelocity of the oracle (orc_node_rho_u) on the same planes after coupled iterations,
    with a cell whose force field reaches plane 0 across the periodic seam"""
    nx, ny, nz = 40, 30, 30
    mask, R = gpu.pipe_mask(nx, ny, nz)
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 0, 0), mask, k_m=2, k_p=1)
    assert _add_both(orc, So, hg, 0, (38.5, 14.2, 15.1), (90, 0, 0)) and _add_both(orc, So, hg, 0, (18.0, 14.5, 12.0), (90, 20, 0))
    F = (2e-5, 0.0, 0.0)
    Lo.set_force_uniform(F); Lg.setExternalVector(F)
    So.contents.body_force[0], So.contents.body_force[1], So.contents.body_force[2] = F
    orc.orc_sim_mechanics(So, 1); hg.cellfields.applyConstitutiveModel(0, True)
    for _ in range(7):
        orc.orc_sim_iterate(So)
    hg.iterate(7)
    # one more iteration up to the point where the reference interpolates: spread, collide-stream (the force field still holds
    # what the collide used, core/hemoCell.cpp:313-329)
    orc.orc_sim_spread(So); orc.orc_collide_stream(Lo.ptr)
    hg.cellfields.spreadParticleForce(True); Lg.collideAndStream(1)
    lib = gpu.capi.lib()
    scale = 0.0
    for side, x in ((0, 0), (1, nx - 1)):
        u_g = np.zeros((3, ny * nz))
        gpu.check(lib.hcl_download_face_velocity(Lg.ptr, side, gpu.dptr(u_g)))
        u_o = np.zeros((ny * nz, 3))
        for k in range(ny * nz):
            r_, u_ = C.c_double(), np.zeros(3)
            orc.orc_node_rho_u(Lo.ptr, x * ny * nz + k, C.byref(r_), gpu.dptr(u_))
            u_o[k] = u_
        fluid = mask.reshape(nx, ny * nz)[x] == 0
        scale = max(scale, np.abs(u_o[fluid]).max())
        assert np.abs(u_g.T[fluid] - u_o[fluid]).max() <= 1e-13 * np.abs(u_o[fluid]).max(), side
        assert (u_g.T[~fluid] == 0).all()                     # stencils admit fluid nodes only
    assert scale > 1e-6                                        # the flow is up
    Lo.destroy(); Lg.destroy()


def test_reproducible_spread(orc, gpu):
    """hc_set_reproducible_spread (SURVEY section 7 hard parts: a deterministic, gather-based spread for parity runs).  The
    reference adds particle by particle in storage order (core/hemoCellParticleField.cpp:841-863), so its runs repeat; the gather
    form here sums every node's contributions in (type, cell id, vertex id) order: (1) the spread field equals the oracle's to
    rounding of the sums (the oracle adds to the body force first, the kernels to zero), (2) two runs of one input give the
    same bits, cells near a wall, across the periodic seam, RBC and PLT, with the side stream in use, (3) a cell stored in a
    different slot gives the same bits (the order is by cell id, not by slot)"""
    lib = gpu.capi.lib()
    nx, ny, nz = 48, 34, 34
    mask, R = gpu.pipe_mask(nx, ny, nz)
    cells = [(0, (10.0, 16.5, 16.5), (90, 0, 0), 5), (0, (30.0, 16.0, 17.0), (70, 30, 10), 2), (0, (46.5, 17.0, 16.0), (90, 0, 30), 9),
             (1, (20.0, 12.0, 20.0), (10, 20, 30), 4), (1, (24.0, 14.0, 12.5), (0, 0, 0), 1)]

    def run(order, steps, reproducible):
        gpu.check(lib.hc_set_reproducible_spread(1 if reproducible else 0))
        P = gpu.base_parameters()
        L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau); L.defineBounceBack(mask); L.latticeEquilibrium(); L.setExternalVector((3e-5, 0, 0))
        h = gpu.HemoCell(L, P); cf = h.cellfields
        cf.addCellType(gpu.CellType.rbc(P), 4); cf.addCellType(gpu.CellType.plt(P), 4)
        h.setParticleVelocityUpdateTimeScaleSeparation(2); h.deletion_check_every = 10 ** 6
        for k in order:
            t, c, a, cid = cells[k]
            assert cf.addCell(t, c, a, cell_id=cid)
        cf.applyConstitutiveModel(0, True)
        h.iterate(steps)
        ids = cf.cell_ids()
        pos = cf.positions
        per_cell = {}
        off = 0
        for t, nvt in ((0, 642), (1, 66)):
            n = cf.type_range(t)[1]
            for c in range(n):
                per_cell[(t, int(ids[sum(cf.type_range(u)[1] for u in range(t)) + c]))] = pos[off + c * nvt: off + (c + 1) * nvt].copy()
            off += n * nvt
        f = L.populations().copy()
        L.destroy()
        return f, per_cell

    try:
        f1, p1 = run([0, 1, 2, 3, 4], 80, True)
        f2, p2 = run([0, 1, 2, 3, 4], 80, True)
        assert np.array_equal(f1, f2) and all(np.array_equal(p1[k], p2[k]) for k in p1)          # (2)
        f3, p3 = run([2, 0, 1, 4, 3], 80, True)                                                   # other slots, same ids
        assert np.array_equal(f1, f3) and all(np.array_equal(p1[k], p3[k]) for k in p1)          # (3)
        fa, pa = run([0, 1, 2, 3, 4], 80, False)                                                  # the atomic kernels: same physics
        assert np.abs(fa - f1).max() <= 1e-12 and max(np.abs(pa[k] - p1[k]).max() for k in p1) <= 1e-10
        # (1) one spread against the oracle
        gpu.check(lib.hc_set_reproducible_spread(1))
        Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 0, 0), mask, plt=True)
        assert _add_both(orc, So, hg, 0, (46.5, 14.2, 15.1), (90, 0, 0)) and _add_both(orc, So, hg, 0, (18.0, 14.5, 9.6), (90, 20, 0))
        assert _add_both(orc, So, hg, 1, (24.0, 14.0, 12.5), (0, 0, 0))
        cf = hg.cellfields
        pos_o, _, _ = _oracle_state(orc, So)
        pos = pos_o + 0.03 * np.random.default_rng(3).standard_normal(pos_o.shape)
        orc.orc_sim_set(So, 0, O.dptr(pos)); cf.positions = pos
        orc.orc_sim_mechanics(So, 1)
        _, _, f_o = _oracle_state(orc, So)
        cf.forces = f_o                                        # the same vertex forces on both sides (the platelet law differs in its atan2 by an ulp)
        Lo.set_force_uniform((0.0, 0.0, 0.0)); Lg.setExternalVector((0.0, 0.0, 0.0))
        orc.orc_sim_spread(So); cf.spreadParticleForce(True)
        assert np.array_equal(Lg.ibm_force(), Lo.force)      # zero body force: the same sums in the same order -> the same bits
        Lo.destroy(); Lg.destroy()
    finally:
        gpu.check(lib.hc_set_reproducible_spread(0))


@pytest.mark.parametrize("which", ["trajectories", "trajectories_beside", "wall_particle", "wall_cell", "repulsion", "boundary_repulsion", "oversized"])
def test_parity_cases_again_with_the_reproducible_spread(orc, gpu, which):
    """the gather-form spread behind the same oracle comparisons as the atomic kernels: coupled trajectories (also with the side
    stream in use), particles removed at a wall in both deletion modes (removed particles have no entries), force_repulsion in the
    spread sum, and a cell larger than the per-cell kernels' tile (the gather form has no tile)"""
    lib = gpu.capi.lib()
    gpu.check(lib.hc_set_reproducible_spread(1))
    try:
        if which == "trajectories":
            test_iterate_trajectories_vs_oracle(orc, gpu, "pipe_rbc_plt_cadence")
        elif which == "trajectories_beside":
            test_iterate_trajectories_vs_oracle(orc, gpu, "pipe_rbc_plt_cadence_beside")
        elif which == "wall_particle":
            test_cell_removed_when_it_reaches_the_wall(orc, gpu, "particle")
        elif which == "wall_cell":
            test_cell_removed_when_it_reaches_the_wall(orc, gpu, "cell")
        elif which == "repulsion":
            test_repulsion_vs_oracle(orc, gpu)
        elif which == "boundary_repulsion":
            test_boundary_particle_repulsion_vs_oracle(orc, gpu)
        else:
            test_ibm_oversized_cells_take_the_fallback_paths(orc, gpu, 1.5, "tile")
    finally:
        gpu.check(lib.hc_set_reproducible_spread(0))


@pytest.mark.parametrize("case", ["pipe_rbc", "pipe_rbc_plt_cadence", "box_periodic", "pipe_rbc_plt_cadence_beside", "box_kolmogorov"])
def test_iterate_trajectories_vs_oracle(orc, gpu, case):
    """HemoCell::iterate for N steps: fluid populations and vertex positions within 1e-6 relative of the
    oracle (north_star tolerance); in practice ~1e-12 (only the atomic spread order differs).
    _beside: no deletion checks inside the call, so that hc_iterate puts advance, mechanics and the next spread on
    the side stream beside the collide between velocity updates (the schedule bench.py measures)"""
    boxes = None
    if case in ("box_periodic", "box_kolmogorov"):
        nx, ny, nz = 32, 32, 32
        periodic = (1, 1, 1); mask = np.zeros((nx, ny, nz), np.uint8); k_m, k_p, plt = 1, 1, False
        F = (1e-6, 2e-6, -1e-6)
        if case == "box_kolmogorov":   # cases/kolmogorovFlow/kolmogorovFlow.cpp:86-90,136-140: +F on one half of the box, -F on the other
            F = (0.0, 0.0, 0.0)
            boxes = [((0, nx - 1, 0, (ny - 1) // 2, 0, nz - 1), (2e-5, 0.0, 0.0)), ((0, nx - 1, (ny - 1) // 2 + 1, ny - 1, 0, nz - 1), (-2e-5, 0.0, 0.0))]
    else:
        nx, ny, nz = 48, 34, 34
        mask, R = gpu.pipe_mask(nx, ny, nz); periodic = (1, 0, 0)
        plt = "cadence" in case; k_m, k_p = ((6, 3) if case.endswith("beside") else (4, 2)) if plt else (1, 1)
        F = (5e-6, 0, 0)
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, periodic, mask, plt=plt, k_m=k_m, k_p=k_p)
    assert _add_both(orc, So, hg, 0, (10.0, 16.5, 16.5), (90, 0, 0))
    assert _add_both(orc, So, hg, 0, (30.0, 16.0, 17.0), (70, 30, 10))
    if plt:
        assert _add_both(orc, So, hg, 1, (20.0, 12.0, 20.0), (10, 20, 30))
        assert _add_both(orc, So, hg, 1, (45.0, 20.0, 13.0), (0, 0, 0))
    Lo.set_force_uniform(F); Lg.setExternalVector(F)
    So.contents.body_force[0], So.contents.body_force[1], So.contents.body_force[2] = F
    if boxes:
        Lg.setExternalVectorBoxes([b for b, _ in boxes], [f for _, f in boxes])
        So.contents.n_regions = len(boxes)
        for r, (b, f) in enumerate(boxes):
            Lo.set_force_box(b, f)
            for i in range(6):
                So.contents.region_box[r][i] = b[i]
            for d in range(3):
                So.contents.region_force[r][d] = f[d]
    Lo.set_threads(8)
    orc.orc_sim_mechanics(So, 1); hg.cellfields.applyConstitutiveModel(0, True)
    nsteps = 60
    for _ in range(nsteps):
        orc.orc_sim_iterate(So)
    lib = gpu.capi.lib()
    if case.endswith("beside"):
        hg.deletion_check_every = 10 ** 6
        gpu.check(lib.hc_profile_reset()); gpu.check(lib.hc_profile_enable(1))
    hg.iterate(nsteps)
    if case.endswith("beside"):
        import ctypes as C
        ms, n = C.c_double(), C.c_long()
        gpu.check(lib.hc_profile_enable(0))
        gpu.check(lib.hc_profile_read(b"collide_stream_beside", C.byref(ms), C.byref(n)))
        assert n.value == nsteps - nsteps // k_p - 1   # every step without a velocity update except the last of the call
    p_o, v_o, f_o = _oracle_state(orc, So)
    p_g = hg.cellfields.positions
    assert hg.iter == So.contents.iter == nsteps
    assert np.abs(p_g - p_o).max() <= 1e-6 * np.abs(p_o).max()
    assert np.abs(p_g - p_o).max() <= 1e-9, np.abs(p_g - p_o).max()
    fluid = mask.reshape(-1) == 0
    fo, fg = Lo.f[fluid], Lg.populations()[fluid]
    assert np.abs(fg - fo).max() <= 1e-6 * np.abs(fo).max()
    assert np.abs(hg.cellfields.forces - f_o).max() <= 1e-6 * np.abs(f_o).max()
    if boxes:   # the two halves stream against each other; Cell::computeVelocity carries the half's own F / 2
        import ctypes as C
        rho, u = Lg.rho_u()
        ux = u[:, 0].reshape(nx, ny, nz)
        assert ux[:, 2:14].min() > 0 and ux[:, 18:30].max() < 0
        uo = np.zeros(ny)
        for y in range(ny):
            r_, u_ = C.c_double(), np.zeros(3)
            orc.orc_node_rho_u(Lo.ptr, ((5 * ny) + y) * nz + 7, C.byref(r_), gpu.dptr(u_))
            uo[y] = u_[0]
        assert np.abs(ux[5, :, 7] - uo).max() <= 1e-9 * np.abs(uo).max()
    Lo.destroy(); Lg.destroy()



def test_stretch_cell_validation_band(orc, gpu):
    """tests/validation/stretch_cell/test_stretch_cell.cpp:158-162, 25 pN case on the GPU path (bounce-back
    walls in place of the regularised velocity BC): transverse 7.3-7.9 um, axial 9.2-9.7 um, volume +-2 %"""
    dt = 1e-7
    Pg = gpu.base_parameters(dt=dt)
    nx, ny, nz = 52, 26, 26
    mask = np.zeros((nx, ny, nz), np.uint8)
    mask[0] = mask[-1] = 1; mask[:, 0] = mask[:, -1] = 1; mask[:, :, 0] = mask[:, :, -1] = 1
    L = gpu.Lattice(nx, ny, nz, (0, 0, 0), 1.0 / Pg.tau)
    L.defineBounceBack(mask); L.latticeEquilibrium()
    h = gpu.HemoCell(L, Pg)
    T = gpu.CellType.rbc(Pg)
    h.cellfields.addCellType(T, 1)
    um = 1e-6 / Pg.dx
    assert h.cellfields.addCell(0, (12.0 * um, 6 * um, 6 * um), (90, 0, 0))
    pos = h.cellfields.positions
    order = np.argsort(pos[:, 0], kind="stable")
    lower, upper = order[:7], order[-7:]
    f = 25.0 * 1e-12 / Pg.df / 7
    idx = np.concatenate([lower, upper])
    ff = np.zeros((14, 3)); ff[:7, 0] = -f; ff[7:, 0] = f
    h.cellfields.applyConstitutiveModel(0, True)
    info0 = h.cellfields.cell_info(0)
    for it in range(10000):
        h.cellfields.addVertexForce(idx, ff)   # cellStretch.applyForce()
        h.iterate(1)
    info = h.cellfields.cell_info(0)
    bb = info["bbox"][0] / um
    axial, transverse = bb[1] - bb[0], bb[3] - bb[2]
    assert 7.3 <= transverse <= 7.9, transverse
    assert 9.2 <= axial <= 9.7, axial
    assert 0.98 < info["volume"][0] / info0["volume"][0] <= 1.02
    assert h.cellfields.counts()[1] == 1
    L.destroy()


def test_one_cell_shear_config_c1(orc, gpu):
    """BASELINE config 1, examples/oneCellShear: 40x40x20 box (20x20x10 um at dx 0.5 um), x/y periodic, top and
    bottom walls moving at -+ (nz-1)*shear/2, one RBC at (9.5,9.5,4.5) um rotated (90,0,0), shear rate 111 1/s,
    dt 0.5e-7, 1000 iterations with stepMaterialEvery = stepParticleEvery = 1: vertex positions and fluid
    populations vs the oracle (moving walls: bounce-back + momentum term in both)"""
    dt = 0.5e-7
    Po = O.make_params(orc, dt=dt); Pg = gpu.base_parameters(dt=dt)
    nz = 20; nx = ny = 2 * nz
    shear = 111.0 * dt; vhalf = (nz - 1) * shear * 0.5     # helper/hemocellInit.hh:82-84
    vhalf = vhalf * (nz - 2) / (nz - 1)                    # the moving wall acts half a node inside the wall node (compat/helper/hemocellInit.hh)
    mask = np.zeros((nx, ny, nz), np.uint8); mask[:, :, -1] = 3; mask[:, :, 0] = 4
    Lo, Lg = _both_lattices(orc, gpu, nx, ny, nz, (1, 1, 0), 1.0 / Po.tau, mask)
    Lo.set_wall_velocity(0, (-vhalf, 0, 0)); Lo.set_wall_velocity(1, (vhalf, 0, 0))
    Lg.setBoundaryVelocity(3, (-vhalf, 0, 0)); Lg.setBoundaryVelocity(4, (vhalf, 0, 0))
    Lo.init_equilibrium(); Lg.latticeEquilibrium()
    Lo.set_threads(8)
    So = orc.orc_sim_create(Lo.ptr, C.byref(Po)); To = O.make_rbc(orc, Po); orc.orc_sim_add_type(So, To)
    hg = gpu.HemoCell(Lg, Pg); hg.cellfields.addCellType(gpu.CellType.rbc(Pg), 1)
    um = 1e-6 / Po.dx
    assert _add_both(orc, So, hg, 0, (9.5 * um, 9.5 * um, 4.5 * um), (90, 0, 0))
    for _ in range(50):                                  # warm-up of the cell-free fluid (oneCellShear.cpp:96-101)
        orc.orc_collide_stream(Lo.ptr)
    Lg.collideAndStream(50)
    orc.orc_sim_mechanics(So, 1); hg.cellfields.applyConstitutiveModel(0, True)
    for _ in range(1000):
        orc.orc_sim_iterate(So)
    hg.iterate(1000)
    p_o, v_o, f_o = _oracle_state(orc, So)
    p_g = hg.cellfields.positions
    assert np.abs(p_g - p_o).max() <= 1e-6 * np.abs(p_o).max()
    assert np.abs(p_g - p_o).max() <= 1e-9, np.abs(p_g - p_o).max()
    fluid = mask.reshape(-1) == 0
    assert np.abs(Lg.populations()[fluid] - Lo.f[fluid]).max() <= 1e-6 * np.abs(Lo.f[fluid]).max()
    # the cell is advected by the shear flow and deforms: it has moved
    assert np.abs(p_o - _oracle_initial(orc, So, To, (9.5 * um, 9.5 * um, 4.5 * um))).max() > 1e-3
    Lo.destroy(); Lg.destroy()


def _oracle_initial(orc, So, To, centre):
    """positions right after placement (same placement code path as the run above)"""
    Po = So.contents.P
    L = O.OracleLattice(orc, So.contents.L.contents.nx, So.contents.L.contents.ny, So.contents.L.contents.nz, (1, 1, 0), 1.0)
    S2 = orc.orc_sim_create(L.ptr, C.byref(Po)); orc.orc_sim_add_type(S2, To)
    c = np.array(centre, dtype=np.float64); a = np.array([90.0, 0, 0]) * (3.14159265358979323846 / 180.0) * -1.0
    orc.orc_sim_add_cell(S2, 0, O.dptr(c), O.dptr(a), 0.0)
    p = np.zeros((S2.contents.np, 3)); orc.orc_sim_get(S2, 0, O.dptr(p))
    orc.orc_sim_destroy(S2); L.destroy()
    return p


def test_full_size_properties_config2(gpu):
    """BASELINE config 2 (pipe 256x128x128, ~5 % Hct, RBC only) at full size, where the oracle is too slow:
    size-independent properties of the path -- spread conserves the total force, the closed pipe conserves
    mass, every cell keeps its volume and stays finite, the cell count does not change."""
    from hemocell_amd.packing import pack_pipe_rbc
    nx, ny, nz = 256, 128, 128
    P = gpu.base_parameters()
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau)
    L.defineBounceBack(mask); L.latticeEquilibrium()
    h = gpu.HemoCell(L, P)
    T = gpu.CellType.rbc(P)
    h.cellfields.addCellType(T, 20); h.setParticleVelocityUpdateTimeScaleSeparation(5)
    centres, angles = pack_pipe_rbc(nx, ny, nz, 0.05)
    n = sum(h.cellfields.addCell(0, c, a, cell_id=i) for i, (c, a) in enumerate(zip(centres, angles)))
    assert n == len(centres) and n > 150
    rng = np.random.default_rng(0)
    pos = h.cellfields.positions
    h.cellfields.positions = pos + 0.02 * rng.standard_normal(pos.shape)    # N(0, 0.02 lu) as SURVEY 8d prescribes
    h.cellfields.applyConstitutiveModel(0, True)
    # (1) spread: sum over the lattice of the IBM force == sum of the (capped) vertex forces
    h.cellfields.spreadParticleForce(True)
    F = L.ibm_force(); f = h.cellfields.forces
    assert np.abs(F.sum(0) - f.sum(0)).max() <= 1e-11 * np.abs(f).sum()
    assert np.abs(F[mask.reshape(-1) != 0]).max() == 0.0          # nothing lands on boundary nodes
    check = gpu.capi.check; lib = gpu.capi.lib()
    check(lib.hcl_zero_ibm_force(L.ptr))
    # (2) mass conservation over 100 coupled steps in the closed pipe
    L.setExternalVector((2e-6, 0, 0))
    # (full-way bounce-back parks the populations that hit a wall on the wall node for one step, so the
    # conserved quantity is the sum over fluid AND wall nodes)
    m0 = L.populations().sum()
    h.iterate(100)
    pops = L.populations()
    assert np.isfinite(pops).all()
    assert abs(pops.sum() - m0) <= 1e-10
    # (3) cells: none lost, all finite, volume within 1 % and area within 2 % of the undeformed mesh
    assert h.cellfields.counts()[1] == n and h.cellfields.counts()[2] == 0
    info = h.cellfields.cell_info(0)
    t = T.tables()
    assert np.isfinite(info["volume"]).all()
    assert np.abs(info["volume"] / t["volume_eq"] - 1).max() < 0.01
    assert np.abs(info["area"] / (t["area_mean_eq"] * T.nt) - 1).max() < 0.02
    # (4) flow goes down the pipe: mean x-velocity of the vertices is positive
    assert h.cellfields.velocities[:, 0].mean() > 0
    L.destroy()


@pytest.mark.parametrize("mode", ["particle", "cell"])
def test_cell_removed_when_it_reaches_the_wall(orc, gpu, mode):
    """advanceParticles tags a particle whose nearest node is a boundary and removeParticles(1) takes it out
    (core/hemoCellParticleField.cpp:566-588, :304-321): the cell stays behind incomplete, gets no mechanics, has its forces
    zeroed at the next material step and keeps being spread / interpolated / advanced until deleteIncompleteCells
    (:512-553) -- mode "particle", the default of oracle and product.  Mode "cell" removes the whole cell at once.
    The IBM itself never lets a membrane reach a no-slip wall, so one cell is given a held velocity towards the wall
    (velocities are only refreshed every stepParticleEvery iterations); a second cell stays.  Oracle and GPU lose the
    same particles at the same iterations and agree afterwards; the deletion costs the GPU path no host round trip."""
    nx, ny, nz = 40, 34, 34
    mask, R = gpu.pipe_mask(nx, ny, nz)
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 0, 0), mask, k_p=1000, k_m=3)
    So.contents.deletion_mode = 0 if mode == "particle" else 1
    hg.cellfields.setDeletionMode(mode)
    assert _add_both(orc, So, hg, 0, (12.0, 16.5, 16.5), (90, 0, 0))
    assert _add_both(orc, So, hg, 0, (30.0, 16.5, 25.0), (90, 0, 0))      # 8.5 lu off axis towards +z
    nv = 642
    orc.orc_sim_mechanics(So, 1); hg.cellfields.applyConstitutiveModel(0, True)
    orc.orc_sim_iterate(So); hg.iterate(1)                                # iteration 0 interpolates
    vel = np.zeros((2 * nv, 3)); vel[nv:, 2] = 0.05
    orc.orc_sim_set(So, 1, O.dptr(vel)); hg.cellfields.velocities = vel
    first_loss = None
    alive_o = np.ones(2 * nv, dtype=np.uint8)
    for it in range(1, 400):
        orc.orc_sim_iterate(So); hg.iterate(1 if it % 7 else 3)           # calls of several iterations too (side-stream schedule)
        if it % 7 == 0:
            orc.orc_sim_iterate(So); orc.orc_sim_iterate(So)
        if mode == "cell":
            assert hg.cellfields.counts()[1] * nv == So.contents.np, it
            lost = So.contents.np == nv
        else:
            alive_o = np.ones(So.contents.np, dtype=np.uint8); orc.orc_sim_get_alive(So, alive_o.ctypes.data)
            lost = not alive_o.all()
            if it % 10 == 0 or lost:
                assert np.array_equal(hg.cellfields.alive(), alive_o.astype(bool)), it
        if first_loss is None and lost:
            first_loss = it
        if first_loss is not None and it > first_loss + 40:
            break
    assert first_loss is not None and 20 < first_loss < 399
    if mode == "cell":
        assert hg.cellfields.counts() == (nv, 1, 1) and So.contents.cells_deleted == 1
    else:
        # the remnants are still listed: 2 cells, one incomplete, and the records leave the removed particles out
        gone = int((alive_o == 0).sum())
        assert 0 < gone < nv and So.contents.particles_deleted == gone
        assert hg.cellfields.counts() == (2 * nv, 2, 0) and hg.cellfields.deletion_counts() == (0, gone, 1, gone)
        rec = hg.cellfields.records()
        assert len(rec) == 2 * nv - gone and (rec["cellId"] == 1).sum() == nv - gone
        p_o, _, f_o = _oracle_state(orc, So)
        live = alive_o.astype(bool)
        assert np.abs(hg.cellfields.positions - p_o)[live].max() <= 1e-9
        assert np.abs(hg.cellfields.forces[nv:][live[nv:]]).max() == 0.0 and np.abs(f_o[nv:][live[nv:]]).max() == 0.0   # zeroed, no mechanics
        assert np.abs(hg.cellfields.forces[:nv]).max() > 0
        assert orc.orc_sim_delete_incomplete_cells(So) == 1 and hg.cellfields.deleteIncompleteCells() == 1
        assert hg.cellfields.counts() == (nv, 1, 1) and So.contents.np == nv
    p_o, _, _ = _oracle_state(orc, So)
    assert np.abs(hg.cellfields.positions - p_o).max() <= 1e-9
    assert hg.cellfields.cell_ids().tolist() == [0]
    for _ in range(10):                                                   # and both carry on identically
        orc.orc_sim_iterate(So)
    hg.iterate(10)
    p_o, _, _ = _oracle_state(orc, So)
    assert np.abs(hg.cellfields.positions - p_o).max() <= 1e-9
    Lo.destroy(); Lg.destroy()


def test_long_run_stays_bounded(gpu):
    """3000 iterations of a 10 % Hct pipe (128x66x66) with the pipeflow cadences: nothing blows up, no cell is
    lost, cell volumes stay within 2 %, the mean flow is positive and below the cell-free Poiseuille maximum"""
    from hemocell_amd.packing import pack_pipe_rbc
    nx, ny, nz = 128, 66, 66
    P = gpu.base_parameters()
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau); L.defineBounceBack(mask); L.latticeEquilibrium()
    h = gpu.HemoCell(L, P); T = gpu.CellType.rbc(P)
    h.cellfields.addCellType(T, 20); h.setParticleVelocityUpdateTimeScaleSeparation(5)
    c, a = pack_pipe_rbc(nx, ny, nz, 0.10)
    n = sum(h.cellfields.addCell(0, cc, aa, cell_id=i) for i, (cc, aa) in enumerate(zip(c, a)))
    assert n == len(c) and n >= 20
    h.cellfields.applyConstitutiveModel(0, True)
    u_max = 0.02                                   # drive hard: ~20x the Re = 0.5 pipeflow case
    F = 4 * P.nu_lbm * u_max / (R * R)
    L.setExternalVector((F, 0, 0))
    h.iterate(3000)
    info = h.cellfields.cell_info(0)
    assert h.cellfields.counts()[1] == n
    assert np.isfinite(h.cellfields.positions).all()
    assert np.abs(info["volume"] / T.tables()["volume_eq"] - 1).max() < 0.02
    rho, u = L.rho_u()
    ux = u[mask.reshape(-1) == 0, 0]
    assert np.isfinite(ux).all() and 0 < ux.mean() and ux.max() < 1.2 * u_max
    assert (info["position"][:, 0] > c[:, 0] + 1.0).mean() > 0.8      # the cells were carried downstream
    L.destroy()


def test_repulsion_vs_oracle(orc, gpu):
    """applyRepulsionForce (core/hemoCellParticleField.cpp:677-743): two RBCs and a platelet brought within the
    cut-off of each other, one pair across the periodic seam.  force_repulsion per vertex vs the oracle, then 30
    coupled iterations with repulsion every 2nd iteration (spread adds force_repulsion + force)."""
    nx, ny, nz = 48, 34, 34
    mask, R = gpu.pipe_mask(nx, ny, nz)
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 0, 0), mask, plt=True)
    assert _add_both(orc, So, hg, 0, (14.0, 16.5, 15.2), (90, 0, 0))
    assert _add_both(orc, So, hg, 0, (15.0, 16.5, 18.4), (90, 0, 0))      # 3.2 lu above the first: membranes ~0.7 lu apart
    assert _add_both(orc, So, hg, 0, (45.5, 16.5, 13.0), (90, 0, 0))      # reaches through the periodic seam ...
    assert _add_both(orc, So, hg, 0, (46.5, 16.5, 16.2), (90, 0, 0))      # ... and so does its close neighbour
    assert _add_both(orc, So, hg, 1, (21.5, 16.5, 14.0), (0, 0, 0))       # platelet next to the first RBC's rim
    k_rep, cutoff_um = 2e-6, 0.7                                          # cut-off 0.7 um = 1.4 lu (examples/pipeflow/config.xml:36-37)
    cutoff = cutoff_um * (1e-6 / Po.dx)
    hg.cellfields.setRepulsion(k_rep, cutoff_um, 2)
    So.contents.rep_enabled = 1; So.contents.rep_timescale = 2; So.contents.rep_const = k_rep; So.contents.rep_cutoff = cutoff
    orc.orc_sim_repulsion(So, k_rep, cutoff); hg.cellfields.applyRepulsionForce()
    r_o = np.zeros((So.contents.np, 3)); orc.orc_sim_get(So, 3, O.dptr(r_o))
    r_g = hg.cellfields.repulsion_forces
    assert np.count_nonzero(np.abs(r_o).sum(1)) > 20                      # the case does exercise the law
    assert np.abs(r_g - r_o).max() <= 1e-12 * np.abs(r_o).max(), np.abs(r_g - r_o).max()
    assert np.abs(r_o.sum(0)).max() <= 1e-12 * np.abs(r_o).sum()          # action = reaction
    orc.orc_sim_mechanics(So, 1); hg.cellfields.applyConstitutiveModel(0, True)
    for _ in range(30):
        orc.orc_sim_iterate(So)
    hg.iterate(30)
    p_o, v_o, f_o = _oracle_state(orc, So)
    assert np.abs(hg.cellfields.positions - p_o).max() <= 1e-9
    orc.orc_sim_get(So, 3, O.dptr(r_o))
    assert np.abs(hg.cellfields.repulsion_forces - r_o).max() <= 1e-9 * max(np.abs(r_o).max(), 1e-30)
    Lo.destroy(); Lg.destroy()


def test_boundary_particle_repulsion_vs_oracle(orc, gpu):
    """enableBoundaryParticles (core/hemoCell.cpp:428-436, core/hemoCellParticleField.cpp:865-918): wall nodes next
    to the fluid push nearby vertices.  An RBC close to the pipe wall and one straddling the periodic seam near the
    wall: force_repulsion vs the oracle (bit for bit: same visiting order), its accumulation over a second call
    (only applyRepulsionForce resets it), then 30 coupled iterations with both repulsions on."""
    nx, ny, nz = 40, 34, 34
    mask, R = gpu.pipe_mask(nx, ny, nz)
    Po, Lo, Lg, So, hg = _sim_pair(orc, gpu, nx, ny, nz, (1, 0, 0), mask)
    assert _add_both(orc, So, hg, 0, (14.0, 16.5, 4.6), (90, 0, 0))       # flat against the wall, ~1.3 lu off it
    assert _add_both(orc, So, hg, 0, (38.5, 16.5, 28.6), (90, 0, 0))      # other side, across the seam
    assert _add_both(orc, So, hg, 0, (26.0, 16.5, 16.5), (0, 0, 0))       # mid-stream: feels nothing
    k_b, cutoff_um = 3e-6, 1.0
    cutoff = cutoff_um * (1e-6 / Po.dx)
    cf = hg.cellfields
    cf.enableBoundaryParticles(k_b, cutoff_um, 2)
    So.contents.brep_enabled = 1; So.contents.brep_timescale = 2; So.contents.brep_const = k_b; So.contents.brep_cutoff = cutoff
    r_o = np.zeros((So.contents.np, 3))
    for calls in (1, 2):
        orc.orc_sim_boundary_repulsion(So, k_b, cutoff); cf.applyBoundaryRepulsionForce()
        orc.orc_sim_get(So, 3, O.dptr(r_o))
        r_g = cf.repulsion_forces
        assert np.array_equal(r_g, r_o), np.abs(r_g - r_o).max()
    touched = np.abs(r_o).sum(1) > 0
    assert touched[:642].sum() > 20 and touched[642:1284].sum() > 20 and touched[1284:].sum() == 0
    # pushes away from the wall: towards the axis for the first cell (z small -> +z)
    assert r_o[:642][touched[:642], 2].min() > 0
    # coupled run, vertex-vertex repulsion on as well so that force_repulsion is reset every 2nd iteration
    cf.setRepulsion(2e-6, 0.7, 2)
    So.contents.rep_enabled = 1; So.contents.rep_timescale = 2; So.contents.rep_const = 2e-6; So.contents.rep_cutoff = 0.7 * (1e-6 / Po.dx)
    orc.orc_sim_mechanics(So, 1); cf.applyConstitutiveModel(0, True)
    for _ in range(30):
        orc.orc_sim_iterate(So)
    hg.iterate(30)
    p_o, v_o, f_o = _oracle_state(orc, So)
    assert np.abs(cf.positions - p_o).max() <= 1e-9
    orc.orc_sim_get(So, 3, O.dptr(r_o))
    assert np.abs(cf.repulsion_forces - r_o).max() <= 1e-9 * max(np.abs(r_o).max(), 1e-30)
    Lo.destroy(); Lg.destroy()


def test_info_reductions_match_downloaded_fields(gpu):
    """FluidInfo / ParticleInfo statistics (helper/fluidInfo.cpp:33-118, helper/particleInfo.cpp:30-140) are device
    reductions; they must equal the same statistics taken from the downloaded fields, and be reproducible bit for bit"""
    nx, ny, nz = 48, 34, 34
    P = gpu.base_parameters()
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau)
    L.defineBounceBack(mask); L.latticeEquilibrium(); L.setExternalVector((3e-5, 0, 0))
    h = gpu.HemoCell(L, P)
    h.cellfields.addCellType(gpu.CellType.rbc(P), 1); h.cellfields.addCellType(gpu.CellType.plt(P), 1)
    assert h.cellfields.addCell(0, (14.0, 16.5, 15.2), (90, 0, 0)) and h.cellfields.addCell(0, (33.0, 17.0, 17.5), (70, 10, 0))
    assert h.cellfields.addCell(1, (24.0, 16.5, 9.0), (0, 0, 0))
    h.cellfields.setRepulsion(2e-6, 0.7, 1)
    h.cellfields.applyConstitutiveModel(0, True)
    h.iterate(40)
    fluid = mask.reshape(-1) == 0
    rho, u = L.rho_u()
    m = np.sqrt((u[fluid] ** 2).sum(1))
    mn, mx, avg, n = L.fluid_stats(0)
    assert n == fluid.sum() and mn == m.min() and mx == m.max() and abs(avg - m.mean()) <= 1e-15 * m.mean()
    assert L.fluid_stats(0) == (mn, mx, avg, n)                       # deterministic
    # external force: nothing is spread between two steps, so it is the body force on every fluid node
    fmn, fmx, favg, fn = L.fluid_stats(1)
    assert fn == n and fmn == fmx == 3e-5
    cf = h.cellfields
    for what, arr in ((1, cf.velocities), (2, cf.forces + cf.repulsion_forces)):
        m = np.sqrt((arr ** 2).sum(1))
        mn, mx, avg, n = cf.vertex_stats(what)
        assert n == len(m) and mn == m.min() and mx == m.max() and abs(avg - m.mean()) <= 1e-14 * m.mean()
        assert cf.vertex_stats(what) == (mn, mx, avg, n)
    L.destroy()


def test_empty_and_degenerate_inputs(orc, gpu):
    """edge cases: a cell field without cells, statistics over nothing, a lattice of the minimum size, a lattice
    that is solid everywhere; none of them may fail or disturb the fluid"""
    P = gpu.base_parameters()
    # (1) cell types registered, no cell placed: iterate == fluid-only stepping, bit for bit
    nx, ny, nz = 10, 9, 11
    rng = np.random.default_rng(5)
    f0 = _random_populations(rng, nx * ny * nz)
    out = []
    for with_cells in (False, True):
        L = gpu.Lattice(nx, ny, nz, (1, 1, 1), 1.0 / P.tau)
        L.set_populations(f0); L.setExternalVector((1e-6, 2e-6, -1e-6))
        if with_cells:
            h = gpu.HemoCell(L, P)
            h.cellfields.addCellType(gpu.CellType.rbc(P), 3); h.cellfields.addCellType(gpu.CellType.plt(P), 2)
            h.cellfields.setRepulsion(2e-6, 0.7, 1); h.cellfields.enableBoundaryParticles(2e-6, 1.0, 1)
            h.cellfields.applyConstitutiveModel(0, True)
            h.iterate(7)
            assert h.cellfields.counts() == (0, 0, 0)
            assert h.cellfields.positions.shape == (0, 3)
            assert h.cellfields.vertex_stats(2) == (0.0, 0.0, 0.0, 0)
        else:
            L.collideAndStream(7)
        out.append(L.populations()); L.destroy()
    assert np.array_equal(out[0], out[1])
    # (2) smallest lattice the library accepts, against the oracle
    Lo, Lg = _both_lattices(orc, gpu, 2, 2, 2, (1, 1, 1), 1.0 / 0.8)
    f0 = _random_populations(rng, 8)
    Lo.f[:] = f0; Lg.set_populations(f0)
    Lo.set_force_uniform((1e-5, 0, 0)); Lg.setExternalVector((1e-5, 0, 0))
    Lo.collide_stream(5); Lg.collideAndStream(5)
    assert np.array_equal(Lg.populations(), Lo.f)
    Lo.destroy(); Lg.destroy()
    # (3) no fluid node at all: stepping is a no-op that must not fault; statistics see zero nodes
    L = gpu.Lattice(6, 5, 7, (1, 0, 0), 1.0)
    L.defineBounceBack(np.ones((6, 5, 7), np.uint8)); L.latticeEquilibrium()
    L.collideAndStream(3)
    assert L.fluid_stats(0)[3] == 0
    assert np.isfinite(L.populations()).all()
    L.destroy()


def test_full_size_fluid_box_config5(gpu):
    """BASELINE config 5 (cases/performance_testing: 512^3 fully periodic, tau = 1, uniform body force on all axes) at
    full size, through size-independent properties: from rest the flow stays uniform -- every node has bit-identical
    velocity -- and |u| after N steps is |F| (N + 1/2) (Guo forcing adds F per step to the momentum, computeVelocity
    adds F/2)"""
    n = 512
    P = gpu.base_parameters(dt=-1.0)
    assert P.tau == 1.0
    L = gpu.Lattice(n, n, n, (1, 1, 1), 1.0)
    L.latticeEquilibrium()
    F = 1e-7
    L.setExternalVector((F, F, F))
    N = 20
    L.collideAndStream(N)
    mn, mx, avg, cnt = L.fluid_stats(0)
    assert cnt == n ** 3
    assert mn == mx                                            # uniform, bit for bit, over 134 M nodes
    assert abs(mx - np.sqrt(3.0) * F * (N + 0.5)) <= 1e-9 * mx
    L.destroy()


def test_full_size_properties_config3(gpu):
    """BASELINE config 3 (examples/pipeflow 512x256x256, 10 % Hct RBC + PLT) on one GPU, through size-independent
    properties: mass conserved, no cell lost, cell volumes and areas stay at their equilibrium values, Sum(spread) =
    Sum(vertex forces), platelets and RBCs both move down the pipe"""
    from hemocell_amd.packing import pack_pipe_rbc
    nx, ny, nz = 512, 256, 256
    P = gpu.base_parameters()
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau)
    L.defineBounceBack(mask); L.latticeEquilibrium()
    h = gpu.HemoCell(L, P)
    Tr, Tp = gpu.CellType.rbc(P), gpu.CellType.plt(P)
    h.cellfields.addCellType(Tr, 20); h.cellfields.addCellType(Tp, 20); h.setParticleVelocityUpdateTimeScaleSeparation(5)
    h.deletion_check_every = 10
    centres, angles = pack_pipe_rbc(nx, ny, nz, 0.10)
    n_rbc = sum(h.cellfields.addCell(0, c, a, cell_id=i) for i, (c, a) in enumerate(zip(centres, angles)))
    pick = np.arange(0, len(centres), 14)
    pc = centres[pick] + np.array([9.5, 0.0, 0.0]); pc[:, 2] += np.where(pc[:, 2] > nz / 2, -4.6, 4.6)
    n_plt = sum(h.cellfields.addCell(1, c, (90.0, 0.0, 0.0), cell_id=len(centres) + i) for i, c in enumerate(pc))
    assert n_rbc > 3500 and n_plt > 200
    cf = h.cellfields
    cf.applyConstitutiveModel(0, True)
    L.setExternalVector((2e-6, 0, 0))
    m0 = L.fluid_stats(2)[2] * L.fluid_stats(2)[3]
    cf.spreadParticleForce(True)
    f = cf.forces
    Fmn, Fmx, Favg, Fn = L.fluid_stats(1)                      # |body + spread| is at least the body force everywhere
    assert Fn == int((mask == 0).sum()) and Fmn > 0
    check = gpu.capi.check; check(gpu.capi.lib().hcl_zero_ibm_force(L.ptr))
    h.iterate(60)
    m1 = L.fluid_stats(2)[2] * L.fluid_stats(2)[3]
    assert abs(m1 - m0) <= 1e-9                                # sum over 33.5 M nodes x 19 populations
    assert cf.counts()[1] == n_rbc + n_plt and cf.counts()[2] == 0
    for t, T in ((0, Tr), (1, Tp)):
        info = cf.cell_info(t); tab = T.tables()
        assert np.isfinite(info["volume"]).all()
        assert np.abs(info["volume"] / tab["volume_eq"] - 1).max() < 0.01
        assert np.abs(info["area"] / (tab["area_mean_eq"] * T.nt) - 1).max() < 0.02
    vmn, vmx, vavg, vn = cf.vertex_stats(1)
    assert vn == n_rbc * 642 + n_plt * 66 and np.isfinite(vmx)
    v = cf.velocities
    assert v[:n_rbc * 642, 0].mean() > 0 and v[n_rbc * 642:, 0].mean() > 0
    L.destroy()


def test_full_size_properties_north_star_target_512_pipe(gpu):
    """the north_star target configuration (512^3 D3Q19 pipe, 10 % hematocrit, one GPU: 134 M nodes, ~16 000 RBC, 10.3 M
    vertices -- what `bench.py --nx 512 --ny 512 --nz 512` times) through size-independent properties: mass conserved to
    round-off, no cell lost, no particle deleted, volumes and areas at their equilibrium values, the spread force field sums
    to the vertex forces, the cells move down the pipe with the flow"""
    from hemocell_amd.packing import pack_pipe_rbc
    nx = ny = nz = 512
    P = gpu.base_parameters()
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau)
    L.defineBounceBack(mask); L.latticeEquilibrium()
    n_fluid = int((mask == 0).sum())
    del mask
    h = gpu.HemoCell(L, P)
    Tr = gpu.CellType.rbc(P)
    h.cellfields.addCellType(Tr, 20); h.setParticleVelocityUpdateTimeScaleSeparation(5)
    centres, angles = pack_pipe_rbc(nx, ny, nz, 0.10)
    n_rbc = sum(h.cellfields.addCell(0, c, a, cell_id=i) for i, (c, a) in enumerate(zip(centres, angles)))
    assert n_rbc == len(centres) and 15000 < n_rbc < 17500
    cf = h.cellfields
    cf.applyConstitutiveModel(0, True)
    L.setExternalVector((2e-6, 0, 0))
    ms = L.fluid_stats(2); m0 = ms[2] * ms[3]
    cf.spreadParticleForce(True)
    Fmn, Fmx, Favg, Fn = L.fluid_stats(1)
    assert Fn == n_fluid and Fmn > 0
    # Sum(spread) = Sum(vertex forces): the membrane forces of free cells sum to zero, so the field's mean stays the body force
    f = cf.forces
    assert np.abs(f.sum(axis=0)).max() < 1e-9 * np.abs(f).sum()
    del f
    gpu.capi.check(gpu.capi.lib().hcl_zero_ibm_force(L.ptr))
    h.iterate(40)
    ms = L.fluid_stats(2); m1 = ms[2] * ms[3]
    assert abs(m1 - m0) <= 4e-9                                # sum over 134 M nodes x 19 populations
    assert cf.counts() == (n_rbc * 642, n_rbc, 0) and cf.deletion_counts() == (0, 0, 0, 0)
    info = cf.cell_info(0); tab = Tr.tables()
    assert np.isfinite(info["volume"]).all()
    assert np.abs(info["volume"] / tab["volume_eq"] - 1).max() < 0.01
    assert np.abs(info["area"] / (tab["area_mean_eq"] * Tr.nt) - 1).max() < 0.02
    vmn, vmx, vavg, vn = cf.vertex_stats(1)
    assert vn == n_rbc * 642 and np.isfinite(vmx) and 0 < vmx < 1e-2
    assert cf.velocities[:, 0].mean() > 0
    L.destroy()


def test_error_behaviour_of_the_c_abi(gpu):
    """every entry point returns a status and leaves a message in hc_last_error(); nothing is thrown across the ABI and
    nothing falls back silently ('Error behaviour')"""
    lib = gpu.capi.lib()
    P = gpu.base_parameters()

    def fails(rc, fragment):
        assert rc != 0
        msg = lib.hc_last_error().decode()
        assert fragment in msg, msg

    ptr = C.c_void_p()
    per = (C.c_int * 3)(1, 0, 0)
    fails(lib.hcl_create(C.byref(ptr), 1, 8, 8, per, 1.0, 0, 1, 1), "dimension")
    fails(lib.hcl_create(C.byref(ptr), 8, 8, 8, per, 2.5, 0, 8, 1), "omega")
    fails(lib.hcl_create(C.byref(ptr), 8, 8, 8, per, 1.0, 4, 8, 2), "slab")
    L = gpu.Lattice(8, 8, 8, (1, 0, 0), 1.0)
    fails(lib.hcl_set_mask(L.ptr, None), "null")
    fails(lib.hcl_collide_stream_part(L.ptr, 7), "part")
    u = (C.c_double * 3)(0, 0, 0)
    fails(lib.hcl_set_wall_velocity(L.ptr, 9, u), "class")
    h = gpu.HemoCell(L, P)
    T = gpu.CellType.rbc(P)
    for _ in range(8):
        h.cellfields.addCellType(T, 1)
    fails(lib.hcp_add_type(h.cellfields.ptr, T.ptr, 1, None), "8 cell types")
    c = (C.c_double * 3)(4, 4, 4); a = (C.c_double * 3)(0, 0, 0)
    fails(lib.hcp_add_cell(h.cellfields.ptr, 11, 0, c, a, 0.0, None), "unknown cell type")
    fails(lib.hcp_repulsion(h.cellfields.ptr), "hcp_set_repulsion first")
    fails(lib.hcp_boundary_repulsion(h.cellfields.ptr), "hcp_set_boundary_repulsion first")
    out = (C.c_double * 3)(); n = C.c_long()
    fails(lib.hcp_vertex_stats(h.cellfields.ptr, 0, out, C.byref(n)), "what")
    with pytest.raises(gpu.capi.HcError):
        gpu.capi.check(lib.hcl_fluid_stats(L.ptr, 5, out, C.byref(n)))
    # a slab of a multi-rank run cannot be stepped before the ranks are connected
    L2 = gpu.Lattice(8, 8, 8, (1, 0, 0), 1.0, x0=0, nx_global=16, n_slabs=2)
    fails(lib.hcl_collide_stream(L2.ptr, 1), "ranks connected first")
    h2 = gpu.HemoCell(L2, P); it = C.c_long(0)
    fails(lib.hc_iterate(L2.ptr, h2.cellfields.ptr, C.byref(it), 1, 1, 1, 1), "ranks connected first")
    fails(lib.hcp_set_deletion_mode(h.cellfields.ptr, 7), "HC_DELETE")
    fails(lib.hc_comm_init(3, 2, 0, b"127.0.0.1", 30000, 0, 0), "rank must be in")
    L2.destroy(); L.destroy()


def test_plain_c_client_of_the_abi(tmp_path, gpu):
    """tests/cabi/c_abi_smoke.c: a C99 program that knows only include/hemocell_amd.h drives one RBC through 50 iterations;
    its numbers equal those of the Python host on the same case"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.dirname(gpu.capi.LIB_PATH)
    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cabi", "c_abi_smoke.c"),
                           "-o", exe, "-L" + libdir, "-lhemocell_amd", "-lm", "-Wl,-rpath," + libdir])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = dict((l.split()[0], l.split()[1:]) for l in r.stdout.splitlines())
    assert out["ITER"] == ["50"] and out["CELLS"][:1] == ["1"] and out["CELLS"][2] == "642"
    nx, ny, nz = 32, 26, 26
    P = gpu.base_parameters()
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau); L.defineBounceBack(mask); L.latticeEquilibrium(); L.setExternalVector((2e-6, 0, 0))
    h = gpu.HemoCell(L, P); h.cellfields.addCellType(gpu.CellType.rbc(P), 1)
    assert h.cellfields.addCell(0, (16.0, 12.5, 12.5), (90, 0, 0))
    h.cellfields.applyConstitutiveModel(0, True); h.iterate(50)
    stats = out["VMAX"]
    vmax, umax, vol = float(stats[0]), float(stats[2]), float(stats[4])
    cen = np.array([float(x) for x in stats[6:9]])
    info = h.cellfields.cell_info(0)
    assert abs(vmax - h.cellfields.vertex_stats(1)[1]) <= 1e-6 * vmax and abs(umax - L.fluid_stats(0)[1]) <= 1e-6 * umax   # printed with 7 digits
    assert abs(vol - info["volume"][0]) <= 1e-6 and np.abs(cen - info["position"][0]).max() <= 1e-6
    L.destroy()


def test_particle_records_in_the_reference_layout(gpu):
    """row a11: HemoCellParticle::serializeValues_t is 120 bytes with v @0, position @24, force @48, force_repulsion @72,
    cellId @96, vertexId @104, restime @108, celltype @112 (core/hemoCellParticle.h:45-63).  The library hands the
    vertices over in that layout and takes them back in any order."""
    P = gpu.base_parameters()
    nx, ny, nz = 48, 34, 34
    mask, R = gpu.pipe_mask(nx, ny, nz)
    L = gpu.Lattice(nx, ny, nz, (1, 0, 0), 1.0 / P.tau); L.defineBounceBack(mask); L.latticeEquilibrium(); L.setExternalVector((3e-5, 0, 0))
    h = gpu.HemoCell(L, P); cf = h.cellfields
    cf.addCellType(gpu.CellType.rbc(P), 1); cf.addCellType(gpu.CellType.plt(P), 1)
    assert cf.addCell(0, (14.0, 16.5, 15.2), (90, 0, 0), cell_id=7) and cf.addCell(0, (33.0, 17.0, 17.5), (70, 10, 0), cell_id=3)
    assert cf.addCell(1, (24.0, 16.5, 9.0), (0, 0, 0), cell_id=11)
    cf.setRepulsion(2e-6, 0.7, 1); cf.applyConstitutiveModel(0, True); h.iterate(10)
    assert cf.SV_DTYPE.itemsize == 120
    rec = cf.records()
    assert len(rec) == 2 * 642 + 66
    assert np.array_equal(rec["position"], cf.positions) and np.array_equal(rec["v"], cf.velocities)
    assert np.array_equal(rec["force"], cf.forces) and np.array_equal(rec["force_repulsion"], cf.repulsion_forces)
    assert list(rec["cellId"][[0, 642, 1284]]) == [7, 3, 11] and list(rec["celltype"][[0, 642, 1284]]) == [0, 0, 1]
    assert np.array_equal(rec["vertexId"][:642], np.arange(642)) and np.array_equal(rec["vertexId"][1284:], np.arange(66))
    # hand the records back shuffled, with the platelet moved: same state, the moved cell moved
    rng = np.random.default_rng(1)
    back = rec.copy()
    back["position"][1284:, 0] += 2.0
    cf.set_records(back[rng.permutation(len(back))])
    again = cf.records()
    key = lambda r: np.lexsort((r["vertexId"], r["cellId"], r["celltype"]))
    a, b = again[key(again)], back[key(back)]
    for name in cf.SV_DTYPE.names:
        assert np.array_equal(a[name], b[name]), name
    # records of an incomplete cell (what removeParticles(1) leaves behind and a checkpoint may hold): the cell
    # arrives incomplete, the records go out again as they came in, and deleteIncompleteCells removes it -- the reference's load
    # path (core/hemoCellFields.cpp:272-274)
    short = back[key(back)][:-3]                              # the platelet (last type) loses three particles
    cf.set_records(short[rng.permutation(len(short))])
    assert cf.counts()[:2] == (2 * 642 + 66, 3)               # slots are kept ...
    assert cf.deletion_counts()[2:] == (1, 3)                 # ... one cell incomplete, three particles missing
    out = cf.records()
    assert len(out) == len(short)
    for name in cf.SV_DTYPE.names:
        assert np.array_equal(out[key(out)][name], short[name]), name
    h.iterate(5)                                              # an incomplete cell takes no part in the mechanics and breaks nothing
    assert np.isfinite(cf.positions).all()
    assert cf.deleteIncompleteCells() == 1 and cf.counts()[:2] == (2 * 642, 2)
    with pytest.raises(gpu.capi.HcError, match="duplicate"):  # what is still refused: two records of one particle
        cf.set_records(np.concatenate([back, back[:1]]))
    L.destroy()


@pytest.mark.parametrize("which", ["collide_111", "collide_000", "pipe", "ibm", "oversized", "iterate_pipe", "iterate_box", "iterate_beside",
                                   "repulsion", "boundary_repulsion", "info", "records"])
def test_padded_plane_stride(orc, gpu, which):
    """lattices whose x-planes are a multiple of 1 MiB (512 x 512 doubles) keep 8 rows of padding between planes
    (hc_lattice::xs, against HBM channel camping).  The lattices of the parity tests are far smaller, so the padding is
    forced here and a cross-section of them is run again: every kernel and host routine that turns (x, y, z) into an
    element index is covered"""
    lib = gpu.capi.lib()
    gpu.check(lib.hc_debug_force_plane_padding(1))
    try:
        {"collide_111": lambda: test_collide_stream_bit_exact_random_state(orc, gpu, (1, 1, 1)),
         "collide_000": lambda: test_collide_stream_bit_exact_random_state(orc, gpu, (0, 0, 0)),
         "pipe": lambda: test_pipe_flow_bit_exact_and_poiseuille(orc, gpu),
         "ibm": lambda: test_ibm_phases_vs_oracle(orc, gpu),
         "oversized": lambda: test_ibm_oversized_cells_take_the_fallback_paths(orc, gpu, 1.5, "tile"),
         "iterate_pipe": lambda: test_iterate_trajectories_vs_oracle(orc, gpu, "pipe_rbc_plt_cadence"),
         "iterate_box": lambda: test_iterate_trajectories_vs_oracle(orc, gpu, "box_periodic"),
         "iterate_beside": lambda: test_iterate_trajectories_vs_oracle(orc, gpu, "pipe_rbc_plt_cadence_beside"),
         "repulsion": lambda: test_repulsion_vs_oracle(orc, gpu),
         "boundary_repulsion": lambda: test_boundary_particle_repulsion_vs_oracle(orc, gpu),
         "info": lambda: test_info_reductions_match_downloaded_fields(gpu),
         "records": lambda: test_particle_records_in_the_reference_layout(gpu)}[which]()
    finally:
        gpu.check(lib.hc_debug_force_plane_padding(0))

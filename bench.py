#!/usr/bin/env python3
"""bench.py -- headline benchmark of the IB-LBM hot path on MI355X.

Metric (BASELINE.json): MLUPS (+ cell-vertex updates/s) on a synthetic pipeflow at 10 % hematocrit.
A "step" is one HemoCell::iterate(): spread -> collide-stream -> [interpolate] -> advance -> [mechanics]
with the reference's pipeflow cadences (stepParticleEvery 5, stepMaterialEvery 20,
examples/pipeflow/config.xml:19-20).  Inputs are resident in HBM before the timed region.

N GPUs (weak scaling): the pipe is N x (256x256x256) long; each rank owns one 256-plane x-slab with its own cells.  The
ranks are one process per GPU; the slab schedule runs inside libhemocell_amd.so (csrc/slab.hip): the faces cross every
step, the particle envelopes at every velocity update, over RCCL point-to-point (ncclSend / ncclRecv between x-neighbours
on the library's side stream) beside the interior collide.  Launch: `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment), or plain
`python bench.py --gpus N`, which starts the N ranks itself.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--copy-reps", type=int, default=40, help="repetitions of the 1 GiB copy that measures this GPU's copy bandwidth")
    ap.add_argument("--config", choices=["headline", "c3"], default="headline",
                    help="headline: BASELINE's metric workload, weak scaling (N x 256^3).  c3: BASELINE config 3, STRONG scaling -- the 512x256x256 pipe "
                         "at 10 % Hct with RBC and 0.07 PLT per RBC, cut into N slabs of 512/N planes (64 planes each on 8 GPUs)")
    ap.add_argument("--nx", type=int, default=256, help="slab thickness per GPU")
    ap.add_argument("--ny", type=int, default=256)
    ap.add_argument("--nz", type=int, default=256)
    ap.add_argument("--hematocrit", type=float, default=0.10)
    ap.add_argument("--plt-ratio", type=float, default=0.0, help="platelets per RBC (pltSimpleModel), e.g. 0.07 for BASELINE config 3")
    ap.add_argument("--fluid-only", action="store_true", help="cases/performance_testing style ceiling run")
    ap.add_argument("--periodic-box", action="store_true",
                    help="cases/performance_testing geometry: fully periodic box, no walls, tau = 1, body force on all axes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-target-512", action="store_true", help="skip the second measurement of the default run (north_star's 512^3 pipe, 25 steps)")
    ap.add_argument("--no-kernel-profile", action="store_true", help="A/B: no per-kernel hipEvent brackets in the timed region (roofline fields are then empty)")
    ap.add_argument("--plane-padding", choices=["auto", "on", "off"], default="auto", help="A/B: padded x-plane stride (auto: planes that are a multiple of 1 MiB)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="A/B: one stream only (by default advance, mechanics and the next spread run beside the collide between velocity updates)")
    ap.add_argument("--spread-after-collide", action="store_true",
                    help="A/B: only advance and mechanics run beside the collide, the next spread follows it on the main stream")
    ap.add_argument("--reproducible-spread", action="store_true",
                    help="parity mode: the gather-form spread (sums in cell-id order, bit-identical from run to run) instead of the fp64 atomics")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--transport", choices=["rccl", "tcp"], default=None,
                    help="data plane of an N > 1 run: rccl (default), or tcp = the same messages through the library's host mesh, "
                         "for ranks that share a GPU (rehearsal on a one-GPU box); also HEMOCELL_TRANSPORT")
    return ap.parse_args()


def usable_cores():
    """host cores this process may really use: affinity mask and cgroup CPU quota (the GPU box grants a share of
    its cores per GPU; more threads than that only oversubscribe)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(args, budget_s):
    """the oracle (CPU restatement, "kind": "port") on the metric's OWN configuration -- the whole pipe of the GPU run, same
    cells, same cadences -- on the host cores this process may use (OpenMP over collide-stream, IBM and mechanics).
    Bounded sample: the first iterations of the run (a whole cadence window of 20 where the budget allows).  Two figures:
    `value` with the collide-stream as ONE pass over the lattice (orc_collide_stream_fused: collide in registers, push),
    `two_pass` with the oracle's literal collide-then-stream (copy, collide in place, gather) -- same arithmetic, same bits."""
    import ctypes as C

    from hemocell_amd.packing import pack_pipe_rbc
    from hemocell_amd.host import pipe_mask
    from oracle import oracle as O

    orc = O.load()
    nx, ny, nz = args.nx, args.ny, args.nz
    cores = usable_cores()
    P = O.make_params(orc)
    mask, R = pipe_mask(nx, ny, nz)
    centres, angles = ([], []) if args.fluid_only else pack_pipe_rbc(nx, ny, nz, args.hematocrit)
    F = body_force(ny, P.nu_lbm)
    legs = {}
    for name, fused, share in (("fused", 1, 0.35), ("two_pass", 0, 0.65)):
        L = O.OracleLattice(orc, nx, ny, nz, (1, 0, 0), 1.0 / P.tau)
        L.set_threads(cores)
        L.set_mask(mask); L.init_equilibrium()
        L.ptr.contents.fused = fused
        S = orc.orc_sim_create(L.ptr, C.byref(P))
        T = O.make_rbc(orc, P); T.contents.timescale = 20
        orc.orc_sim_add_type(S, T)
        S.contents.particle_velocity_timescale = 5
        ncell = 0
        for c, a in zip(centres, angles):
            ar = np.array(a) * (3.14159265358979323846 / 180.0) * -1.0
            ncell += orc.orc_sim_add_cell(S, 0, O.dptr(np.ascontiguousarray(c)), O.dptr(ar), 0.0)
        L.set_force_uniform(F)
        for d in range(3):
            S.contents.body_force[d] = F[d]
        orc.orc_sim_mechanics(S, 1)
        orc.orc_sim_iterate(S)      # iteration 0 untimed: first touch of the second population buffer and of the stencil arrays
        t0 = time.perf_counter(); steps = 0
        # whole cadence windows of 5 iterations (one velocity update each); at least one, at most 20 iterations
        while steps < 20 and (steps < 5 or (time.perf_counter() - t0) * (steps + 5) / steps < budget_s * share):
            for _ in range(5):
                orc.orc_sim_iterate(S)
            steps += 5
        dt = time.perf_counter() - t0
        nverts = S.contents.np
        legs[name] = {"mlups": nx * ny * nz * steps / dt / 1e6, "steps": steps, "seconds": dt,
                      "vertex_updates_per_s": nverts * steps / dt}
        orc.orc_sim_destroy(S); L.destroy()
    # one core (SURVEY 8d asks for it beside the all-cores figure): the one-pass form on a quarter-length piece of the same pipe, 5 iterations
    nx1 = max(32, nx // 4)
    mask1, _ = pipe_mask(nx1, ny, nz)
    L = O.OracleLattice(orc, nx1, ny, nz, (1, 0, 0), 1.0 / P.tau)
    L.set_threads(1); L.set_mask(mask1); L.init_equilibrium(); L.ptr.contents.fused = 1
    S = orc.orc_sim_create(L.ptr, C.byref(P))
    T = O.make_rbc(orc, P); T.contents.timescale = 20
    orc.orc_sim_add_type(S, T); S.contents.particle_velocity_timescale = 5
    if not args.fluid_only:
        c1, a1 = pack_pipe_rbc(nx1, ny, nz, args.hematocrit)
        for c, a in zip(c1, a1):
            orc.orc_sim_add_cell(S, 0, O.dptr(np.ascontiguousarray(c)), O.dptr(np.array(a) * (3.14159265358979323846 / 180.0) * -1.0), 0.0)
    L.set_force_uniform(F)
    for d in range(3):
        S.contents.body_force[d] = F[d]
    orc.orc_sim_mechanics(S, 1); orc.orc_sim_iterate(S)
    t0 = time.perf_counter()
    for _ in range(5):
        orc.orc_sim_iterate(S)
    dt1 = time.perf_counter() - t0
    single = {"value": nx1 * ny * nz * 5 / dt1 / 1e6, "unit": "MLUPS", "cores": 1,
              "sample": "pipe %dx%dx%d, %d vertices, iterations 1..5 in %.1f s, one-pass collide-stream" % (nx1, ny, nz, S.contents.np, dt1)}
    orc.orc_sim_destroy(S); L.destroy()
    best = legs["fused"]
    return {"value": best["mlups"], "single_core": single, "unit": "MLUPS", "cores": cores, "kind": "port",
            "mlups_per_core": best["mlups"] / cores,
            "sample_nodes_over_gpu_nodes": 1.0,
            "sample": "oracle (oracle/hemo_oracle.c, OpenMP collide-stream, IBM and mechanics) on the metric's own workload: pipe "
                      "%dx%dx%d, %d RBC (%d vertices), iterations 1..%d (velocity updates every 5th, mechanics every 20th) in %.1f s with the "
                      "one-pass collide-stream; %d iterations in %.1f s with the oracle's literal two-pass form"
                      % (nx, ny, nz, ncell, nverts, best["steps"], best["seconds"], legs["two_pass"]["steps"], legs["two_pass"]["seconds"]),
            "vertex_updates_per_s": best["vertex_updates_per_s"],
            "two_pass": {"value": legs["two_pass"]["mlups"], "unit": "MLUPS", "mlups_per_core": legs["two_pass"]["mlups"] / cores,
                         "vertex_updates_per_s": legs["two_pass"]["vertex_updates_per_s"]}}


def body_force(ny, nu_lbm, Re=0.5):
    """poiseuilleForce of examples/pipeflow/pipeflow.cpp:80: 8 nu (u_max/2) / R^2 with u_max = Re nu / (2R)"""
    R = (ny - 2) / 2.0
    u_max = Re * nu_lbm / (2 * R)
    return (8.0 * nu_lbm * (u_max * 0.5) / (R * R), 0.0, 0.0)


def make_workload(args, nx, ny, nz, rank, world):
    """the synthetic pipeflow of the metric: lattice, walls, driving force, cells placed and their initial forces evaluated;
    everything resident in HBM when this returns.  -> (parameters, SlabRunner, pipe radius, distinct cells)"""
    from hemocell_amd import host
    from hemocell_amd.packing import pack_pipe_rbc
    from hemocell_amd.slab import SlabRunner

    # examples/pipeflow/config.xml:25-28: dx 5e-7, dt 1e-7, nuP 1.1e-6 -> tau 1.82; performance_testing: dt = -1 -> tau = 1
    P = host.base_parameters(dt=-1.0) if args.periodic_box else host.base_parameters()
    nxg = nx * world
    # every-step deletion semantics (core/hemoCellParticleField.cpp:566-588): the check lives in the advance kernel
    runner = SlabRunner(nx_local=nx, ny=ny, nz=nz, rank=rank, world=world, P=P,
                        periodic=(True, True, True) if args.periodic_box else (True, False, False),
                        particle_timescale=5, material_timescale=20,
                        deletion_check_every=1, fluid_only=args.fluid_only or args.periodic_box)
    mask, R = host.pipe_mask(nxg, ny, nz)
    if args.periodic_box:
        mask[:] = 0
        args.fluid_only = True
    runner.define_bounce_back(mask)
    runner.lattice.latticeEquilibrium(1.0, (0, 0, 0))
    bf = body_force(ny, P.nu_lbm)
    runner.lattice.setExternalVector((bf[0], bf[0], bf[0]) if args.periodic_box else bf)
    n_cells = 0
    if not args.fluid_only:
        rbc = host.CellType.rbc(P)
        runner.add_cell_type(rbc)
        centres, angles = pack_pipe_rbc(nxg, ny, nz, args.hematocrit)
        runner.load_cells(0, centres, angles)
        if args.plt_ratio > 0:
            # platelets (66 vertices, 2.5 x 1.1 um discs) in the gaps of the RBC grid: half a pitch off in x and z
            plt_t = runner.add_cell_type(host.CellType.plt(P))
            pick = np.arange(0, len(centres), max(1, int(round(1.0 / args.plt_ratio))))
            pc = centres[pick] + np.array([9.5, 0.0, 0.0]); pc[:, 2] += np.where(pc[:, 2] > nz / 2, -4.6, 4.6)
            pa = np.tile(np.array([90.0, 0.0, 0.0]), (len(pc), 1))
            runner.load_cells(plt_t, pc, pa)
        n_cells = int(runner.sync_placement().sum())     # distinct cells over all slabs
    runner.prepare()
    return P, runner, R, n_cells


def target_512(args, steps=25, warmup=5):
    """north_star's target workload on the driver's clock: the 512^3 pipe at 10 % hematocrit on one GPU (>= 0.60 of the HBM-roofline
    MLUPS asked for), measured like the headline: barrier + synchronise on both sides of `steps` whole iterations"""
    from hemocell_amd import host

    lib = host.capi.lib()
    n = 512
    P, runner, R, n_cells = make_workload(args, n, n, n, 0, 1)
    nverts = runner.owned_vertices()
    runner.run(warmup)
    lib.hc_profile_reset(); lib.hc_profile_enable(1)
    host.check(lib.hc_synchronize())
    t0 = time.perf_counter()
    runner.run(steps)
    host.check(lib.hc_synchronize())
    elapsed = time.perf_counter() - t0
    lib.hc_profile_enable(0)
    counts = np.zeros(3, dtype=np.int64)
    host.check(lib.hcl_node_counts(runner.lattice.ptr, host.lptr(counts)))
    prof = {}
    for k in ("collide_stream", "collide_stream_alone", "collide_stream_beside", "ibm_spread", "ibm_interpolate", "advance", "mechanics"):
        m2, n2 = C.c_double(), C.c_long()
        host.check(lib.hc_profile_read(k.encode(), C.byref(m2), C.byref(n2)))
        prof[k] = {"avg_ms": m2.value / max(n2.value, 1), "launches": n2.value}
    nodes = n ** 3
    bpn = runner.lattice.bytes_per_node()
    mlups = nodes * steps / elapsed / 1e6
    col = prof["collide_stream"]["avg_ms"]
    out = {"workload": "pipe %dx%dx%d (R=%.0f), %d RBC (%d vertices), 10 %% Hct, same cadences as the headline" % (n, n, n, R, n_cells, nverts),
           "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "mlups": mlups,
           "mlups_fluid_nodes": int(counts[1]) * steps / elapsed / 1e6, "vertex_updates_per_s": nverts * steps / elapsed,
           "fluid_node_fraction": int(counts[1]) / nodes, "active_node_fraction": int(counts[2]) / nodes,
           # whole step: by the SURVEY 8(d) convention (353 B x every node of the box) and over the nodes the kernel visits
           "whole_step_hbm_frac": mlups * 1e6 * bpn / 8.0e12,
           "whole_step_hbm_frac_active_nodes": int(counts[2]) * steps / elapsed * bpn / 8.0e12,
           # the collide kernel alone, over the nodes it visits (an upper bound of the bytes that move)
           "collide_avg_ms": col, "collide_frac_active_nodes": int(counts[2]) * bpn / (col * 1e-3) / 8.0e12 if col > 0 else None,
           "collide_frac_convention": nodes * bpn / (col * 1e-3) / 8.0e12 if col > 0 else None,
           "kernel_avg_ms": prof, "target": "north_star: >= 0.60 of the HBM-roofline MLUPS on this workload at 1 GPU"}
    runner.cells.destroy(); runner.lattice.destroy()
    return out


def launch_ranks(args):
    """plain `python bench.py --gpus N`: start the N ranks as child processes (one per GPU, nothing in this process has
    touched the GPU), hand rank 0's stdout through, fail if any rank fails"""
    port = 29400 + os.getpid() % 2000
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.exit("bench.py: ranks exited with %s" % rcs)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        return launch_ranks(args)
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started in a world of %d ranks" % (args.gpus, world))
    if args.config == "c3":
        if 512 % world:
            sys.exit("bench.py --config c3: 512 planes do not divide into %d slabs" % world)
        args.nx, args.ny, args.nz, args.plt_ratio = 512 // world, 256, 256, 0.07
    # stdout carries exactly one line, the JSON result: RCCL prints its version banner to stdout, so everything else
    # that writes to file descriptor 1 is sent to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if args.transport:
        os.environ["HEMOCELL_TRANSPORT"] = args.transport

    from hemocell_amd import host, slab

    lib = host.capi.lib()
    if world > 1:
        rank, world = slab.comm_init_env()     # connects the ranks and selects GPU LOCAL_RANK (modulo the device count)
    else:
        rank = 0
        host.init(0)
    if args.no_overlap:
        host.check(lib.hc_set_overlap(0))
    if args.spread_after_collide:
        host.check(lib.hc_set_overlap(2))
    if args.reproducible_spread:
        host.check(lib.hc_set_reproducible_spread(1))
    host.check(lib.hc_debug_force_plane_padding({"auto": 0, "on": 1, "off": -1}[args.plane_padding]))

    P, runner, R, n_cells = make_workload(args, args.nx, args.ny, args.nz, rank, world)
    nxg = args.nx * world
    nverts_local = runner.owned_vertices()

    def barrier():
        host.check(lib.hc_synchronize())
        if world > 1:
            slab.barrier()

    # the GPU in hand: device-to-device copy rate (1 GiB, read + write counted), printed next to the roofline.  Measured
    # before the run, while nothing else is queued on the GPU
    cbw = C.c_double()
    host.check(lib.hc_measure_copy_bandwidth(1 << 30, args.copy_reps, C.byref(cbw)))
    barrier()
    runner.run(args.warmup)
    runner.slab_stats(reset=True)
    lib.hc_profile_reset()
    lib.hc_profile_enable(0 if args.no_kernel_profile else 1)
    barrier()
    t0 = time.perf_counter()
    runner.run(args.steps)
    barrier()
    t1 = time.perf_counter()
    lib.hc_profile_enable(0)
    elapsed = t1 - t0
    sstats = runner.slab_stats()
    counts = np.zeros(3, dtype=np.int64)
    host.check(lib.hcl_node_counts(runner.lattice.ptr, host.lptr(counts)))
    if world > 1:
        elapsed = float(slab.allreduce([elapsed], "max")[0])            # the slowest rank sets the time of the job
        tot = slab.allreduce([float(nverts_local), float(counts[1]), float(counts[2]), sstats["host_s"], sstats["header_wait_s"]], "sum")
        nverts = int(tot[0]); fluid_nodes, active_nodes = int(tot[1]), int(tot[2])
        host_ms_per_step = (tot[3] - tot[4]) / world / args.steps * 1e3     # enqueueing only: the waits for extents / id headers are reported apart
        header_wait_ms = tot[4] / world / max(sstats["particle_steps"], 1) * 1e3
    else:
        nverts = nverts_local; fluid_nodes, active_nodes = int(counts[1]), int(counts[2])
        host_ms_per_step = header_wait_ms = None

    # state of the run after the timed region, reduced over the slabs: the same numbers for any number of ranks (strong scaling)
    # or per unit of pipe (weak scaling); tests compare the N-slab figures with the 1-slab ones
    u_stats = runner.fluid_stats(0); rho_stats = runner.fluid_stats(2); v_stats = runner.vertex_stats(1)
    diagnostics = {"fluid_speed_max": u_stats[1], "fluid_speed_mean": u_stats[2], "fluid_nodes": u_stats[3],
                   # hcl_fluid_stats(2) reduces rhoBar = rho - 1 (the stored populations are f_i - t_i) over ALL nodes: its sum is the conserved
                   # mass minus the node count
                   "mass_minus_nodes": rho_stats[2] * rho_stats[3], "all_nodes": rho_stats[3], "rho_bar_min": rho_stats[0], "rho_bar_max": rho_stats[1],
                   "vertex_speed_max": v_stats[1], "vertex_speed_mean": v_stats[2], "owned_vertices": v_stats[3]}

    ms, n = C.c_double(), C.c_long()
    host.check(lib.hc_profile_read(b"collide_stream", C.byref(ms), C.byref(n)))
    prof = {}
    for k in ("collide_stream", "collide_stream_alone", "collide_stream_beside", "ibm_spread", "ibm_interpolate", "advance", "mechanics"):
        m2, n2 = C.c_double(), C.c_long()
        host.check(lib.hc_profile_read(k.encode(), C.byref(m2), C.byref(n2)))
        prof[k] = {"ms_total": m2.value, "launches": n2.value}
    # After the timed region: the dominant kernel with the GPU to itself (inside the timed region four launches of five share
    # it with advance and the next spread on the side stream)
    alone_ms = None
    if world == 1:
        lib.hc_profile_reset(); lib.hc_profile_enable(1)
        runner.lattice.collideAndStream(10)
        host.check(lib.hc_synchronize())
        m2, n2 = C.c_double(), C.c_long()
        host.check(lib.hc_profile_read(b"collide_stream", C.byref(m2), C.byref(n2)))
        lib.hc_profile_enable(0)
        alone_ms = m2.value / max(n2.value, 1)

    if rank == 0:
        nodes = nxg * args.ny * args.nz
        mlups = nodes * args.steps / elapsed / 1e6
        bytes_per_node = runner.lattice.bytes_per_node()   # 19+19 populations, 3+3 force doubles, 1 mask byte
        # dominant kernel: collide_stream_kernel.  Per step and rank it processes the nx*ny*nz nodes of the slab
        # (one launch, or an interior + boundary-plane launches when faces are in flight); the hipEvent brackets are on the
        # stream the kernel runs on (hc_profile_*), rank 0's numbers are reported.
        step_nodes = args.nx * args.ny * args.nz                # nodes one rank's collide launches of ONE iteration cover between them
        launches_per_step = n.value / float(args.steps)           # 1 on one GPU; 2-3 on a slab (interior planes and face planes apart)
        launch_nodes = step_nodes / launches_per_step            # algorithmic share of one launch, on average
        avg_ms = ms.value / max(n.value, 1)                       # average duration of ONE launch (what rocprofv3 --stats averages too)
        step_ms = ms.value / args.steps                           # the collide launches of one iteration together
        # units one launch processes = the nodes it VISITS (the kernel creates no threads for inert solid: 83.6 % of the 256^3 box);
        # the same figure over every node of the box is kept as `frac_whole_box_convention` (what rounds 1-2 printed as `frac`)
        visited_per_launch = (active_nodes / world) / launches_per_step
        achieved = visited_per_launch * bytes_per_node / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        achieved_box = launch_nodes * bytes_per_node / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch of that kernel from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # passes, FETCH doubled per the gfx950 note): measured offline on exactly this workload AND this build of the kernel
        # (hc_build_tag = hash of csrc/lattice.hip) and committed under profiles/; otherwise null
        traffic, traffic_src, traffic_per_node = None, None, None
        tag = lib.hc_build_tag().decode()
        if (args.nx, args.ny, args.nz) == (256, 256, 256) and not args.fluid_only and abs(args.hematocrit - 0.10) < 1e-12 and args.plt_ratio == 0:
            for tf in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json")), reverse=True):
                tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
                if tj.get("kernel_tag") == tag:
                    traffic, traffic_per_node = tj["hbm_bytes_per_launch"], tj["hbm_bytes_per_node"]
                    traffic_src = "profiles/%s (rocprofv3 --pmc, %.1f B/node, kernel build %s)" % (tf, traffic_per_node, tag)
                    break
        out = {
            "metric": "MLUPS + cell-vertex updates/s, 256^3 pipeflow 10% Hct" if args.config == "headline" else "MLUPS + cell-vertex updates/s, 512x256x256 pipeflow 10% Hct RBC+PLT (BASELINE config 3)",
            "value": mlups, "unit": "MLUPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak" if args.config == "headline" else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "vertex_updates_per_s": nverts * args.steps / elapsed,
            # the same rate counted over the fluid nodes only, and what share of the box the kernel touches
            "mlups_fluid_nodes": fluid_nodes * args.steps / elapsed / 1e6,
            "fluid_node_fraction": fluid_nodes / nodes, "active_node_fraction": active_nodes / nodes,
            "config": {"workload": "examples/pipeflow synthetic: pipe %dx%dx%d (x periodic, analytic cylinder R=%.0f, bounce-back), "
                                   "%d cells (rbcHighOrderModel, 642 vertices each, target Hct %.2f; pltSimpleModel platelets per RBC: %g), tau=%.2f, "
                                   "stepParticleEvery=5, stepMaterialEvery=20, wall deletions checked every step%s%s"
                                   % (nxg, args.ny, args.nz, R, n_cells, args.hematocrit, args.plt_ratio, P.tau, (", fully periodic box without walls (cases/performance_testing)" if args.periodic_box else ", fluid only") if args.fluid_only else "",
                                      ", reproducible (gather-form) spread" if args.reproducible_spread else ""),
                       "lattice": [nxg, args.ny, args.nz], "cells": n_cells, "vertices": nverts,
                       "parallelism": "x-slabs x%d, native slab schedule (csrc/slab.hip), %s point-to-point" % (world, {0: "no", 1: "RCCL", 2: "TCP-staged"}[slab.comm_info()[2]]) if world > 1 else "1 GPU"},
            # `achieved` / `frac`: SURVEY.md section 8(d)'s ALGORITHMIC bytes per node (353 B) x the nodes one launch processes (the
            # visited ones) over the kernel's average launch time.  `traffic` (PMC) and `frac_real_traffic` = traffic / time / peak say
            # what the HBM actually delivered; `frac_whole_box_convention` counts the inert solid nodes as if they moved too.
            "roofline": {"bound": "hbm", "kernel": "collide_stream_kernel", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_src,
                         "frac_whole_box_convention": achieved_box / 8000.0,
                         # `traffic` is HBM bytes per ITERATION (all collide launches of a step, PMC sum); real rate = traffic / their time
                         "frac_real_traffic": (traffic / (step_ms * 1e-3) / 8.0e12) if (traffic and step_ms > 0) else None,
                         "real_traffic_GBps": (traffic / (step_ms * 1e-3) / 1e9) if (traffic and step_ms > 0) else None,
                         # without a PMC figure for this workload: the bytes of the nodes the kernel visits (an upper bound of what moves)
                         "frac_active_nodes": (active_nodes / world) * bytes_per_node / (step_ms * 1e-3) / 8.0e12 if step_ms > 0 else None,
                         "algorithmic_bytes_per_launch": visited_per_launch * bytes_per_node,
                         "bytes_per_node": bytes_per_node, "nodes_per_launch": visited_per_launch, "box_nodes_per_launch": launch_nodes, "avg_launch_ms": avg_ms,
                         "launches": n.value, "launches_per_step": launches_per_step, "collide_ms_per_step": step_ms, "kernel_build": tag,
                         "peak_source": "MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)",
                         "copy_GBps_this_gpu": cbw.value},
            "kernel_ms": prof,
            "diagnostics": diagnostics,
            # the same kernel with the GPU to itself (10 launches right after the timed region)
            "roofline_alone": ({"avg_launch_ms": alone_ms, "launches": 10, "frac": active_nodes * bytes_per_node / (alone_ms * 1e-3) / 8.0e12,
                                "frac_whole_box_convention": step_nodes * bytes_per_node / (alone_ms * 1e-3) / 8.0e12,
                                "frac_real_traffic": (traffic / (alone_ms * 1e-3) / 8.0e12) if traffic else None} if alone_ms else None),
            # whole job against the HBM roofline of the whole step: MLUPS x algorithmic bytes per node update over the
            # aggregate 8 TB/s of the GPUs used (north_star: >= 0.60 on the 512^3 pipe at 1 GPU)
            "whole_step_hbm_frac": mlups * 1e6 * bytes_per_node / (8.0e12 * world),
            # the same over the nodes the kernel visits (inert solid does not count as moved)
            "whole_step_hbm_frac_active_nodes": active_nodes * args.steps / elapsed * bytes_per_node / (8.0e12 * world),
        }
        if world > 1:
            # host side of the native slab schedule (csrc/slab.hip), mean over the ranks: time spent enqueueing a step, and how long
            # the host sits in the two waits of a velocity update (cell extents, id headers) -- with GPU work already queued
            out["slab_schedule"] = {"host_ms_per_step": host_ms_per_step, "header_wait_ms_per_velocity_update": header_wait_ms,
                                    "records_sent_rank0": sstats["cells_sent"], "copies_created_rank0": sstats["cells_new"],
                                    "copies_dropped_rank0": sstats["cells_dropped"]}
        headline = (args.nx, args.ny, args.nz) == (256, 256, 256) and not args.fluid_only and abs(args.hematocrit - 0.10) < 1e-12 and args.plt_ratio == 0
        if world == 1 and headline and not args.no_target_512:
            # after the headline measurement, with its memory returned: north_star's target workload on the same clock
            if runner.cells is not None:
                runner.cells.destroy()
            runner.lattice.destroy()
            out["target_512"] = target_512(args)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        slab.barrier()
        slab.comm_finalize()


if __name__ == "__main__":
    main()

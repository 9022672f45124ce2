"""Neighbour exchange for x-slab runs (world_size > 1): population halos and particle envelopes.

Reference equivalents
  * fluid:     Palabos duplicateOverlaps(staticVariables) inside collideAndStream (core/hemoCell.cpp:317,
               envelope width core/hemoCell.cpp:142) -> halo_exchange()
  * particles: HemoCellFields::syncEnvelopes (core/hemoCellFields.cpp:377-499): every block that holds >= 1
               vertex of a cell gets the complete cell; the copy of a vertex owned by the sending block
               overwrites a non-local copy (HemoCellParticleField::addParticle,
               core/hemoCellParticleField.cpp:173-235); periodic images are shifted by the domain length
               (core/hemoCellParticleDataTransfer.cpp:33-65) -> sync_cells()

Only point-to-point messages between x-neighbours are used (no collective on the data path).  The protocol is
written against a small engine interface so that it runs unchanged on the HIP engine (product) and on a
numpy stand-in used by the CPU gloo tests of the protocol itself.
"""
import ctypes as C
import gc
import time

import numpy as np
import torch
import torch.distributed as dist

from . import host

E_SHARE = 4.0   # a cell within this many lattice units of a face is replicated on the neighbour
MAX_SHARED = 8192   # capacity of the per-face id header of the envelope exchange


class NeighbourComm:
    """ring of ranks along x; tensors are exchanged with the low (-x) and high (+x) neighbour"""

    def __init__(self, rank, world, periodic, group=None):
        self.rank, self.world, self.periodic = rank, world, bool(periodic)
        self.group = group
        self.lo = (rank - 1) % world if (periodic or rank > 0) else None
        self.hi = (rank + 1) % world if (periodic or rank < world - 1) else None
        self.backend = dist.get_backend(group)

    def _stage(self, t):
        return t if self.backend == "nccl" else t.cpu()

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        """send_lo goes to the low neighbour (which receives it as its recv_hi) and vice versa.
        Returns a callable that completes the transfers into recv_lo / recv_hi."""
        ops = []

        def live(t, nb):   # zero-length messages are skipped on both sides (the counts are known to both)
            return nb is not None and t is not None and t.numel() > 0
        s_lo = self._stage(send_lo) if live(send_lo, self.lo) else None
        s_hi = self._stage(send_hi) if live(send_hi, self.hi) else None
        r_lo = (recv_lo if self.backend == "nccl" else torch.empty(recv_lo.shape, dtype=recv_lo.dtype)) if live(recv_lo, self.lo) else None
        r_hi = (recv_hi if self.backend == "nccl" else torch.empty(recv_hi.shape, dtype=recv_hi.dtype)) if live(recv_hi, self.hi) else None
        # order matters when lo and hi are the same peer (world == 2): my lo-face message is the peer's
        # hi-halo message, so receives are posted hi first
        if s_lo is not None:
            ops.append(dist.P2POp(dist.isend, s_lo, self.lo, self.group, tag=1))
        if s_hi is not None:
            ops.append(dist.P2POp(dist.isend, s_hi, self.hi, self.group, tag=2))
        if r_hi is not None:
            ops.append(dist.P2POp(dist.irecv, r_hi, self.hi, self.group, tag=1))
        if r_lo is not None:
            ops.append(dist.P2POp(dist.irecv, r_lo, self.lo, self.group, tag=2))
        works = dist.batch_isend_irecv(ops) if ops else []

        def wait():
            for w in works:
                w.wait()
            if self.backend != "nccl":
                if r_lo is not None:
                    recv_lo.copy_(r_lo)
                if r_hi is not None:
                    recv_hi.copy_(r_hi)
        return wait


class _SideStream:
    def __init__(self, engine):
        self.e = engine
        if engine._side is None:
            p = C.c_void_p()
            host.check(engine.lib.hc_side_stream(C.byref(p)))
            engine._side = torch.cuda.ExternalStream(p.value, device=engine.device)
        self.ctx = torch.cuda.stream(engine._side)

    def __enter__(self):
        self.e.route(1)
        self.ctx.__enter__()

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        self.e.route(0)
        return False


class HipEngine:
    """the product engine: every method is one call into libhemocell_amd.so"""

    def __init__(self, lattice, cells, device):
        self.L, self.C, self.device = lattice, cells, device
        self.lib = host.capi.lib()
        self.nx, self.x0, self.nx_global = lattice.nx, lattice.x0, lattice.nx_global
        self._ext_pending = set()
        self._side = None

    # ---- fluid
    def halo_buffer(self, width):
        return torch.empty(self.L.halo_doubles(width), dtype=torch.float64, device=self.device)

    def halo_pack(self, side, width, buf, next=False):
        self.L.halo_pack(side, width, buf.data_ptr(), next)

    def halo_unpack(self, side, width, buf):
        self.L.halo_unpack(side, width, buf.data_ptr())

    def collide(self, part):
        self.L.collide_part(part)

    def step_end(self):
        self.L.step_end()

    def fork(self):
        host.check(self.lib.hc_fork())

    def route(self, side):
        host.check(self.lib.hc_route(int(side)))

    def join(self):
        host.check(self.lib.hc_join())

    def side(self):
        """context: the launches of the library AND torch's own work (copies, RCCL transfers) go to the library's side stream"""
        return _SideStream(self)

    # ---- cells
    def n_types(self):
        return len(self.C.types) if self.C is not None else 0

    def nv(self, t):
        return self.C.types[t].nv

    def record_doubles(self, t):
        """doubles per cell record: 9 per vertex, 12 once a repulsion is enabled (force_repulsion travels too)"""
        return int(self.lib.hcp_record_doubles(self.C.ptr, t))

    def cell_ids(self, t):
        ids = self.C.cell_ids()
        f = sum(self.C.type_range(u)[1] for u in range(t))
        return ids[f:f + self.C.type_range(t)[1]]

    def cell_extents_begin(self, t):
        """enqueue the extents of type t (stream ordered, returns at once)"""
        host.check(self.lib.hcp_cell_extents_begin(self.C.ptr, t))
        self._ext_pending.add(t)

    def cell_extents(self, t):
        """per cell [min x, max x, owned vertices]; waits only for the copy started by cell_extents_begin"""
        n = self.C.type_range(t)[1]
        ext = np.empty((n, 3), dtype=np.float64)
        if t not in self._ext_pending:
            host.check(self.lib.hcp_cell_extents_begin(self.C.ptr, t))
        self._ext_pending.discard(t)
        host.check(self.lib.hcp_cell_extents_end(self.C.ptr, t, host.dptr(ext)))
        return ext

    def pack_cells(self, t, slots, x_shift):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        buf = torch.empty(len(slots) * self.record_doubles(t), dtype=torch.float64, device=self.device)
        host.check(self.lib.hcp_pack_cells(self.C.ptr, t, slots.ctypes.data_as(host.capi.c_int_p), len(slots), float(x_shift),
                                           C.c_void_p(buf.data_ptr())))
        return buf

    def unpack_cells(self, t, slots, ids, is_new, buf):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        is_new = np.ascontiguousarray(is_new, dtype=np.int32)
        host.check(self.lib.hcp_unpack_cells(self.C.ptr, t, slots.ctypes.data_as(host.capi.c_int_p), host.lptr(ids),
                                             is_new.ctypes.data_as(host.capi.c_int_p), len(slots), C.c_void_p(buf.data_ptr())))

    def remove_cells(self, t, slots):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        host.check(self.lib.hcp_remove_cells(self.C.ptr, t, slots.ctypes.data_as(host.capi.c_int_p), len(slots)))

    def record_buffer(self, t, n):
        return torch.empty(n * self.record_doubles(t), dtype=torch.float64, device=self.device)

    def repulsion(self, it):
        """core/hemoCell.cpp:307-312: vertex-vertex and boundary-particle repulsion at their own cadences"""
        if self.C is None:
            return
        if self.C.rep_timescale and it % self.C.rep_timescale == 0:
            self.C.applyRepulsionForce()
        if self.C.brep_timescale and it % self.C.brep_timescale == 0:
            self.C.applyBoundaryRepulsionForce()

    def spread(self):
        if self.C is not None:
            self.C.spreadParticleForce(True)

    def interpolate(self):
        if self.C is not None:
            self.C.interpolateFluidVelocity()

    def interpolate_cells(self, t, slots):
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        host.check(self.lib.hcp_interpolate_cells(self.C.ptr, t, slots.ctypes.data_as(host.capi.c_int_p), len(slots)))

    def advance(self):
        if self.C is not None:
            self.C.advanceParticles(False)

    def mechanics(self, it, forced=False):
        if self.C is not None:
            self.C.applyConstitutiveModel(it, forced)

    def owned_vertices(self):
        if self.C is None:
            return 0
        n = C.c_long()
        host.check(self.lib.hcp_owned_vertices(self.C.ptr, C.byref(n)))
        return n.value


class SlabProtocol:
    """the per-step choreography of a multi-slab run; engine-agnostic"""

    def __init__(self, engine, comm, particle_timescale, nx_global, periodic_x, overlap=True):
        self.e, self.comm = engine, comm
        self.k_p = particle_timescale
        self.nx_global, self.periodic_x = nx_global, periodic_x
        self.iter = 0
        self.halo_fresh = False
        self.overlap = overlap
        self._hb = {}
        self._hdr = {}
        self._pending = None     # completion of a face exchange in flight
        self._spread_done = False   # the spread of the coming iteration already ran beside the last collide
        self._planned = False       # the cell extents for the coming envelope sync are already on their way to the host
        self.stats = {"cells_sent": 0, "cells_new": 0, "cells_dropped": 0, "merge_host_s": 0.0}

    # ------------------------------------------------------------------ fluid halos
    def _bufs(self, width):
        if width not in self._hb:
            self._hb[width] = [self.e.halo_buffer(width) for _ in range(4)]
        return self._hb[width]

    def halo_exchange_begin(self, width, next=False):
        """pack the faces and start the transfers; returns the callable that completes them (wait + unpack).
        next: pack from the buffer the collide in progress is writing (its face planes are already done)."""
        self.halo_drain()
        s_lo, s_hi, r_lo, r_hi = self._bufs(width)
        if self.comm.lo is not None:
            self.e.halo_pack(0, width, s_lo, next)
        if self.comm.hi is not None:
            self.e.halo_pack(1, width, s_hi, next)
        wait = self.comm.exchange(s_lo, s_hi, r_lo, r_hi)

        def finish():
            wait()
            if self.comm.lo is not None:
                self.e.halo_unpack(0, width, r_lo)
            if self.comm.hi is not None:
                self.e.halo_unpack(1, width, r_hi)
        return finish

    # ------------------------------------------------------------------ particle envelopes
    def plan_cells(self):
        """start the per-cell extents of every type; called at the start of a step that ends with sync_cells, when the
        positions are already those the sync will see (they change only in advance), so that the copy is long done
        when it is needed and the host never drains the stream for it"""
        for t in range(self.e.n_types()):
            self.e.cell_extents_begin(t)

    def _header_buffers(self, t):
        """fixed-size id headers of type t: [host staging, device send, device receive, host landing] per side, made once"""
        if t not in self._hdr:
            dev = self._dev()
            pin = dev != "cpu"
            self._hdr[t] = [(torch.zeros(MAX_SHARED, dtype=torch.int64, pin_memory=pin),
                             torch.zeros(MAX_SHARED, dtype=torch.int64, device=dev),
                             torch.zeros(MAX_SHARED, dtype=torch.int64, device=dev),
                             torch.zeros(MAX_SHARED, dtype=torch.int64, pin_memory=pin)) for _ in (0, 1)]
        return self._hdr[t]

    def sync_cells(self):
        self.sync_cells_finish(self.sync_cells_begin())

    def sync_cells_begin(self):
        """first half of the envelope sync, everything that only depends on the positions: which cells cross which
        face, and the exchange of the id headers.  The received headers are copied to pinned host memory behind an
        event, so that the host round trip can hide behind the interpolation kernel that is enqueued next."""
        e, comm = self.e, self.comm
        x0, x1 = e.x0, e.x0 + e.nx
        plans = []
        for t in range(e.n_types()):
            ext = e.cell_extents(t)
            ids = np.asarray(e.cell_ids(t), dtype=np.int64)
            send = {}
            for side, nb in ((0, comm.lo), (1, comm.hi)):
                if nb is None:
                    send[side] = (np.zeros(0, np.int32), np.zeros(0, np.int64))
                    continue
                touch = (ext[:, 0] < x0 + E_SHARE) if side == 0 else (ext[:, 1] >= x1 - E_SHARE)
                # only a rank that owns part of the cell forwards it (a pure ghost is the neighbour's business)
                touch &= ext[:, 2] > 0
                slots = np.nonzero(touch)[0].astype(np.int32)
                order = np.argsort(ids[slots], kind="stable")
                slots = slots[order]
                send[side] = (slots, ids[slots])
            # phase 1: one fixed-size header per side: [count, id_0 .. id_{count-1}, padding]
            if max(len(send[0][0]), len(send[1][0])) >= MAX_SHARED:
                raise host.capi.HcError("more cells cross one slab face (%d) than the id header holds (%d)"
                                        % (max(len(send[0][0]), len(send[1][0])), MAX_SHARED - 1))
            hdr = self._header_buffers(t)
            for side in (0, 1):
                stage, dev_s, _, _ = hdr[side]
                k = len(send[side][0])
                hh = stage.numpy()
                hh[0] = k; hh[1:1 + k] = send[side][1]
                dev_s[:1 + k].copy_(stage[:1 + k], non_blocking=True)
            comm.exchange(hdr[0][1], hdr[1][1], hdr[0][2], hdr[1][2])()
            ready = None
            for side, nb in ((0, comm.lo), (1, comm.hi)):
                if nb is not None:
                    hdr[side][3].copy_(hdr[side][2], non_blocking=True)
            if self._dev() != "cpu":
                ready = torch.cuda.Event()
                ready.record()
            plans.append((t, ext, ids, send, ready))
        return plans

    def sync_cells_finish(self, plans):
        self.sync_cells_merge(self.sync_cells_records_begin(plans, False))

    def sync_cells_records_begin(self, plans, interpolate_first):
        """second part: the records of the crossing cells start to travel.  They carry interpolated velocities, so with
        interpolate_first the crossing cells are interpolated here, ahead of the rest: the caller launches the
        interpolation of all cells afterwards (same values for these) and the transfer hides beside it."""
        e, comm = self.e, self.comm
        out = []
        for t, ext, ids, send, ready in plans:
            hdr = self._header_buffers(t)
            if ready is not None:
                ready.synchronize()
            hr = [hdr[0][3].numpy() if comm.lo is not None else np.zeros(1, np.int64),
                  hdr[1][3].numpy() if comm.hi is not None else np.zeros(1, np.int64)]
            n_lo, n_hi = int(hr[0][0]), int(hr[1][0])
            ids_r = [hr[0][1:1 + n_lo].copy(), hr[1][1:1 + n_hi].copy()]
            if interpolate_first:
                both = np.union1d(send[0][0], send[1][0]).astype(np.int32)
                if len(both):
                    e.interpolate_cells(t, both)
            shift_lo = float(self.nx_global) if (self.periodic_x and comm.rank == 0) else 0.0          # crossing the seam downward
            shift_hi = -float(self.nx_global) if (self.periodic_x and comm.rank == comm.world - 1) else 0.0
            rec_s = [e.pack_cells(t, send[0][0], shift_lo), e.pack_cells(t, send[1][0], shift_hi)]
            rec_r = [e.record_buffer(t, n_lo), e.record_buffer(t, n_hi)]
            wait = comm.exchange(rec_s[0], rec_s[1], rec_r[0], rec_r[1])
            out.append((t, ext, ids, send, ids_r, rec_s, rec_r, wait))
        return out

    def sync_cells_merge(self, states):
        """last part: wait for the records, merge them (a local vertex wins), drop copies nobody refreshed"""
        e = self.e
        for t, ext, ids, send, ids_r, rec_s, rec_r, wait in states:
            n = len(ids)
            n_lo, n_hi = len(ids_r[0]), len(ids_r[1])
            wait()
            t_host = time.perf_counter()
            self.stats["cells_sent"] += len(send[0][0]) + len(send[1][0])
            # phase 3: merge.  Incoming ids are looked up in the local ids (sorted search); unknown ones are appended
            # in arrival order, and a cell that arrives from both sides (world == 2) is new only the first time
            refreshed = np.zeros(n, dtype=bool)
            known_ids, known_slots = ids, np.arange(n, dtype=np.int64)
            for side, cnt in ((0, n_lo), (1, n_hi)):
                if cnt == 0:
                    continue
                rid = np.ascontiguousarray(ids_r[side])
                order = np.argsort(known_ids, kind="stable")
                pos = np.searchsorted(known_ids[order], rid)
                pos_c = np.minimum(pos, len(known_ids) - 1) if len(known_ids) else np.zeros(cnt, np.int64)
                hit = (known_ids[order][pos_c] == rid) if len(known_ids) else np.zeros(cnt, bool)
                slots = np.empty(cnt, np.int64)
                slots[hit] = known_slots[order][pos_c[hit]]
                n_new = int((~hit).sum())
                first_new = len(known_ids)
                slots[~hit] = first_new + np.arange(n_new)
                is_new = (~hit).astype(np.int32)
                refreshed[slots[hit & (slots < n)]] = True
                self.stats["cells_new"] += n_new
                if n_new:
                    known_ids = np.concatenate([known_ids, rid[~hit]])
                    known_slots = np.concatenate([known_slots, first_new + np.arange(n_new)])
                e.unpack_cells(t, slots.astype(np.int32), rid, is_new, rec_r[side])
            # phase 4: a copy without any local vertex survives only while its owner keeps refreshing it
            # (deleteNonLocalParticles, core/hemoCellFields.cpp:676-688)
            drop = np.nonzero((ext[:, 2] == 0) & ~refreshed)[0].astype(np.int32)
            if len(drop):
                e.remove_cells(t, drop)
                self.stats["cells_dropped"] += len(drop)
            self.stats["merge_host_s"] += time.perf_counter() - t_host   # host time while the GPU still interpolates

    def _dev(self):
        return getattr(self.e, "device", "cpu")

    # ------------------------------------------------------------------ HemoCell::iterate on a slab
    def prepare(self):
        self.halo_exchange_begin(2)()
        self.halo_fresh = True

    def halo_drain(self):
        """complete a face exchange that was started at the end of the previous step"""
        if self._pending is not None:
            fin, self._pending = self._pending, None
            fin()
            self.halo_fresh = True

    def step(self, more=False):
        """Nothing on the main stream ever waits for a transfer.  Between two velocity updates the main stream only
        collides the interior planes; beside it, on the library's side stream: the arrival of the neighbours' faces, the
        collide of the two face planes, the packing and departure of the 5 crossing populations for the next step, then
        advance, mechanics and (with more=True: another step follows in the same run) the repulsion and spread of the
        NEXT iteration, none of which depend on the collide.  On a velocity update the side stream collides the two planes
        next to each face and sends the wider message an interpolation needs, which travels during the interior collide,
        then exchanges the id headers of the envelope sync; the records of the crossing cells travel while the velocities
        of all cells are interpolated."""
        e = self.e
        it = self.iter
        particle_step = it % self.k_p == 0
        if particle_step and not self._planned:
            self.plan_cells()                                 # extents for the envelope sync at the end of this step
        if not self._spread_done:
            e.repulsion(it)                                   # core/hemoCell.cpp:307-312
            e.spread()                                        # :313
        self._spread_done = False
        if not self.overlap:                                  # :317, strictly one stream, nothing in flight across phases
            if not self.halo_fresh:
                self.halo_exchange_begin(1)()
            e.collide(0)
            e.step_end()
            if particle_step:
                self.halo_exchange_begin(2)()
                self.sync_cells_finish(self.sync_cells_begin_after(e.interpolate))
            e.advance()                                       # :342
            e.mechanics(it)                                   # :345
        elif particle_step:
            e.fork()
            with e.side():
                self.halo_drain()
                e.collide(4)                                  # the two planes next to each face, beside the interior ...
                finish = self.halo_exchange_begin(2, next=True)   # ... so that the wide message travels during the interior collide
            e.collide(3)                                      # main stream: interior planes
            e.step_end()
            with e.side():
                plans = self.sync_cells_begin()               # which cells cross + id headers: needs positions only; the host
            e.join()                                          # waits for the extents in there with the interior collide already queued
            finish()
            states = self.sync_cells_records_begin(plans, True)   # crossing cells interpolated first, their records leave
            e.interpolate()                                   # :327-332, all cells, at halo nodes too
            self.sync_cells_merge(states)
            e.advance()
            e.mechanics(it)
        else:
            e.fork()
            e.collide(1)                                      # main stream: interior planes
            with e.side():
                self.halo_drain()                             # faces of the neighbours (already here after a velocity update)
                e.collide(2)
                self._pending = self.halo_exchange_begin(1, next=True)   # my faces of the state being written
            e.step_end()
            with e.side():
                e.advance()
                if (it + 1) % self.k_p == 0:
                    self.plan_cells()                         # positions are final for the sync of the next step
                    self._planned = True
                e.mechanics(it)
                if more:
                    e.repulsion(it + 1)
                    e.spread()                                # :313 of iteration it + 1
                    self._spread_done = True
            e.join()
        if particle_step:
            self._planned = False
        self.halo_fresh = particle_step
        self.iter = it + 1

    def sync_cells_begin_after(self, interpolate):
        plans = self.sync_cells_begin()                       # id headers travel and land on the host ...
        interpolate()                                         # ... while velocities are interpolated (at halo nodes too)
        return plans

    def run(self, n):
        """n iterations.  The cyclic garbage collector is held off for the duration: a full collection of the interpreter's
        heap takes ~65 ms (measured with torch loaded), during which nothing is enqueued and the GPU runs dry; the
        objects a step creates are freed by reference counting."""
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            for k in range(n):
                self.step(more=k + 1 < n)
        finally:
            if was_enabled:
                gc.enable()


class SlabExchange:
    """binds a SlabRunner to the HIP engine and torch.distributed"""

    def __init__(self, runner, comm=None):
        self.runner = runner
        dev = torch.device("cuda", torch.cuda.current_device())
        # kernels of the library and torch's copies / RCCL transfers must be ordered on ONE stream.  torch's
        # default stream is the null stream (handle 0, which hc_set_stream reads as "library stream"), so a
        # dedicated stream is made current for torch and handed to the library.
        self.stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(self.stream)
        host.check(host.capi.lib().hc_set_stream(self.stream.cuda_stream))
        self.engine = HipEngine(runner.lattice, runner.cells, dev)
        self.comm = comm or NeighbourComm(runner.rank, runner.world, runner.periodic[0])
        self.protocol = SlabProtocol(self.engine, self.comm, runner.k_p, runner.nx_global, runner.periodic[0])
        if runner.nx < 40:
            raise host.capi.HcError("a slab must be at least 40 planes wide (two cell diameters plus the envelope), got %d" % runner.nx)

    def load_cells(self, t, centres, angles, min_dist_um=0.0, radius=9.0):
        """place every cell whose extent touches this slab's extended region; periodic images across the
        seam are placed with shifted x (core/hemoCellParticleDataTransfer.cpp:33-65)"""
        r = self.runner
        x0, x1, nxg = r.x0, r.x0 + r.nx, r.nx_global
        n = 0
        for i, (c, a) in enumerate(zip(centres, angles)):
            for shift in ((0.0, -nxg, nxg) if r.periodic[0] else (0.0,)):
                cx = c[0] + shift
                if cx + radius >= x0 - E_SHARE and cx - radius < x1 + E_SHARE:
                    cc = np.array([cx, c[1], c[2]])
                    n += bool(r.cells.addCell(t, cc, a, min_dist_um, cell_id=i))
                    break
        # exact test on the placed vertices: keep cells that own a vertex here or reach within E_SHARE
        ext = self.engine.cell_extents(t)
        keep = (ext[:, 2] > 0) | ((ext[:, 1] >= x0 - E_SHARE) & (ext[:, 0] < x1 + E_SHARE))
        drop = np.nonzero(~keep)[0].astype(np.int32)
        if len(drop):
            self.engine.remove_cells(t, drop)
        return n - len(drop)

    def owned_vertices(self):
        return self.engine.owned_vertices()

    def prepare(self):
        self.protocol.prepare()
        # what exists now (the interpreter's modules, torch, the set-up of this run) is long-lived: take it out of the
        # cyclic collector's reach once, so that a collection triggered between runs only looks at what a run left behind
        gc.collect()
        gc.freeze()

    def run(self, n):
        self.protocol.run(n)

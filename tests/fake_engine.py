"""numpy stand-in for the HIP engine, used ONLY by the CPU gloo tests of the slab protocol
(hemocell_amd/exchange.py).  Fluid: pure streaming of 19 "populations" on nx x 1 x 1 planes (a value moves
one plane per step along its c_x), stored post-"collision" with the same pull/halo layout as the library.
Cells: nv vertices moving with a per-cell velocity; interpolation is only valid for locally owned vertices."""
import numpy as np
import torch

CX = np.array([0, -1, 0, 0, -1, -1, -1, -1, 0, 0, 1, 0, 0, 1, 1, 1, 1, 0, 0])
HALO = 2


class FakeEngine:
    device = "cpu"

    def __init__(self, nx, x0, nx_global, plane=3, nv=5):
        self.nx, self.x0, self.nx_global, self.plane, self._nv = nx, x0, nx_global, plane, nv
        self.f = np.zeros((2, 19, nx + 2 * HALO, plane))
        self.cur = 0
        self.ids = np.zeros(0, np.int64)
        self.pos = np.zeros((0, nv, 3)); self.vel = np.zeros((0, nv, 3)); self.frc = np.zeros((0, nv, 3))
        self.cell_speed = {}

    # ---- fluid
    def halo_buffer(self, width):
        return torch.zeros((5 if width == 1 else 19) * width * self.plane, dtype=torch.float64)

    def _pops(self, width, to_buf, side):
        if width == 2:
            return list(range(19))
        cxm, cxp = [1, 4, 5, 6, 7], [10, 13, 14, 15, 16]
        return (cxm if side == 0 else cxp) if to_buf else (cxp if side == 0 else cxm)

    def halo_pack(self, side, width, buf, next=False):
        xf = HALO if side == 0 else HALO + self.nx - width
        a = self.f[1 - self.cur if next else self.cur][self._pops(width, True, side), xf:xf + width, :]
        buf.copy_(torch.from_numpy(np.ascontiguousarray(a).reshape(-1)))

    def halo_unpack(self, side, width, buf):
        xf = HALO - width if side == 0 else HALO + self.nx
        pops = self._pops(width, False, side)
        self.f[self.cur][pops, xf:xf + width, :] = buf.numpy().reshape(len(pops), width, self.plane)

    def collide(self, part):
        xs = {0: range(self.nx), 1: range(1, self.nx - 1), 2: [0, self.nx - 1], 3: range(2, self.nx - 2),
              4: [0, 1, self.nx - 2, self.nx - 1]}[part]
        fin, fout = self.f[self.cur], self.f[1 - self.cur]
        for x in xs:
            for q in range(19):
                fout[q, x + HALO] = fin[q, x + HALO - CX[q]]   # pull; "collision" = identity

    def step_end(self):
        self.cur ^= 1

    def fork(self):
        pass

    def route(self, side):
        pass

    def join(self):
        pass

    def side(self):
        import contextlib
        return contextlib.nullcontext()

    def post_stream(self):
        """S(x,q) = P(x - c_q, q) on the bulk planes"""
        fin = self.f[self.cur]
        out = np.zeros((19, self.nx, self.plane))
        for q in range(19):
            out[q] = fin[q, HALO - CX[q]:HALO - CX[q] + self.nx]
        return out

    # ---- cells
    def n_types(self):
        return 1

    def nv(self, t):
        return self._nv

    def cell_ids(self, t):
        return self.ids

    def _owned(self):
        g = np.floor(self.pos[:, :, 0] + 0.5).astype(np.int64) - self.x0
        return (g >= 0) & (g < self.nx)

    def cell_extents(self, t):
        ext = np.zeros((len(self.ids), 3))
        if len(self.ids):
            ext[:, 0] = self.pos[:, :, 0].min(1); ext[:, 1] = self.pos[:, :, 0].max(1); ext[:, 2] = self._owned().sum(1)
        return ext

    def pack_cells(self, t, slots, x_shift):
        rec = np.concatenate([self.pos[slots], self.vel[slots], self.frc[slots]], axis=2).copy()
        rec[:, :, 0] += x_shift
        return torch.from_numpy(rec.reshape(-1))

    def record_buffer(self, t, n):
        return torch.zeros(n * self._nv * 9, dtype=torch.float64)

    def unpack_cells(self, t, slots, ids, is_new, buf):
        rec = buf.numpy().reshape(len(slots), self._nv, 9)
        n_new = int(np.sum(is_new))
        if n_new:
            self.ids = np.concatenate([self.ids, np.zeros(n_new, np.int64)])
            z = np.zeros((n_new, self._nv, 3))
            self.pos = np.concatenate([self.pos, z]); self.vel = np.concatenate([self.vel, z]); self.frc = np.concatenate([self.frc, z])
        own = self._owned()
        for k, s in enumerate(slots):
            if is_new[k]:
                self.ids[s] = ids[k]
                take = np.ones(self._nv, bool)
            else:
                assert self.ids[s] == ids[k]
                take = ~own[s]
            self.pos[s, take] = rec[k, take, 0:3]; self.vel[s, take] = rec[k, take, 3:6]; self.frc[s, take] = rec[k, take, 6:9]

    def remove_cells(self, t, slots):
        keep = np.ones(len(self.ids), bool); keep[slots] = False
        self.ids, self.pos, self.vel, self.frc = self.ids[keep], self.pos[keep], self.vel[keep], self.frc[keep]

    def repulsion(self, it):
        pass

    def cell_extents_begin(self, t):
        pass

    def spread(self):
        pass

    def interpolate(self):
        own = self._owned()
        for s, cid in enumerate(self.ids):
            v = np.full((self._nv, 3), np.nan)           # a non-local vertex gets garbage, as on the GPU
            v[own[s]] = (self.cell_speed[int(cid)], 0.0, 0.0)
            self.vel[s] = v

    def interpolate_cells(self, t, slots):
        own = self._owned()
        for s in slots:
            v = np.full((self._nv, 3), np.nan)
            v[own[s]] = (self.cell_speed[int(self.ids[s])], 0.0, 0.0)
            self.vel[s] = v

    def advance(self):
        self.pos += self.vel

    def mechanics(self, it, forced=False):
        self.frc[:] = self.ids[:, None, None].astype(float)

    def owned_vertices(self):
        return int(self._owned().sum())

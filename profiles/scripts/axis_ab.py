"""fluid-only collide on a 256^3 pipe whose axis lies along x (the headline geometry: z-rows start and end at the wall) against
the same pipe along z (the fastest index: every row is all fluid or all solid) -- what rows along the pipe axis would buy"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hemocell_amd import host
host.init(0); lib = host.capi.lib()
P = host.base_parameters()
n = 256
R = (n - 2) / 2.0; c = (n - 1) / 2.0
a, b = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
disc = ((a - c) ** 2 + (b - c) ** 2 > R * R).astype(np.uint8)
for name, mask, per in (("axis x", np.broadcast_to(disc[None, :, :], (n, n, n)).copy(), (1, 0, 0)), ("axis z", np.broadcast_to(disc[:, :, None], (n, n, n)).copy(), (0, 0, 1)),
                        ("axis x", np.broadcast_to(disc[None, :, :], (n, n, n)).copy(), (1, 0, 0)), ("axis z", np.broadcast_to(disc[:, :, None], (n, n, n)).copy(), (0, 0, 1))):
    L = host.Lattice(n, n, n, per, 1.0 / P.tau)
    L.defineBounceBack(mask); L.latticeEquilibrium(1.0, (0, 0, 0)); L.setExternalVector((1e-6, 0, 0) if per[0] else (0, 0, 1e-6))
    L.collideAndStream(20); host.check(lib.hc_synchronize())
    t0 = time.perf_counter(); L.collideAndStream(200); host.check(lib.hc_synchronize()); t1 = time.perf_counter()
    cnt = np.zeros(3, dtype=np.int64); host.check(lib.hcl_node_counts(L.ptr, host.lptr(cnt)))
    print("%s: %.4f ms per step, %.0f MLUPS, fluid %.4f visited %.4f" % (name, (t1 - t0) / 200 * 1e3, n ** 3 * 200 / (t1 - t0) / 1e6, cnt[1] / n ** 3, cnt[2] / n ** 3), flush=True)
    L.destroy()

"""per-phase shader cycles of the IBM cell kernels (scratch/ab/lib_phase.so, built with -DHC_IBM_PHASE_TIMES)"""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hemocell_amd import capi
capi.LIB_PATH = os.path.join(ROOT, "scratch", "ab", "lib_phase.so")
from hemocell_amd import host
from hemocell_amd.packing import pack_pipe_rbc
from hemocell_amd.slab import SlabRunner
hct = float(sys.argv[1]) if len(sys.argv) > 1 else 0.10
host.init(0)
lib = host.capi.lib()
raw = C.CDLL(capi.LIB_PATH)
P = host.base_parameters()
r = SlabRunner(nx_local=256, ny=256, nz=256, rank=0, world=1, P=P)
mask, R = host.pipe_mask(256, 256, 256)
r.define_bounce_back(mask); r.lattice.latticeEquilibrium(1.0, (0, 0, 0)); r.lattice.setExternalVector((1e-6, 0, 0))
r.add_cell_type(host.CellType.rbc(P))
c, a = pack_pipe_rbc(256, 256, 256, hct)
r.load_cells(0, c, a); r.prepare(); r.run(20)
host.check(lib.hc_synchronize())
cells = r.cells.ptr
ph = np.zeros(32)
names = {0: "tag load", 1: "pos/force loads + bbox", 2: "tile_is_clear", 3: "mask tile", 4: "stencils", 5: "zero tile", 6: "x accumulate", 7: "x flush", 8: "y accumulate", 9: "y flush",
         10: "z accumulate", 11: "z flush", 12: "mark + compact nodes", 13: "node velocities", 14: "blend"}
for wide in (0, 1):
    lib.hc_debug_ibm_wide(wide)
    for kname, fn in (("spread", lambda: lib.hcp_spread(cells, 1)), ("interpolate", lambda: lib.hcp_interpolate(cells))):
        for _ in range(3): host.check(fn())
        host.check(lib.hc_synchronize()); raw.hc_debug_phase_times(ph.ctypes.data_as(C.POINTER(C.c_double)))
        n = 20
        for _ in range(n): host.check(fn())
        host.check(lib.hc_synchronize()); raw.hc_debug_phase_times(ph.ctypes.data_as(C.POINTER(C.c_double)))
        per = ph / (n * len(c))
        print("wide %d %s: cycles per workgroup, total %.0f" % (wide, kname, per.sum()))
        for k in range(15):
            if per[k] > 0: print("   %-26s %8.0f  %5.1f %%" % (names[k], per[k], 100 * per[k] / per.sum()))

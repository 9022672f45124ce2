"""time the IBM kernels alone (no collide beside them): python scratch/ibm_bench.py <hematocrit> [n]"""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hemocell_amd import host
from hemocell_amd.packing import pack_pipe_rbc
from hemocell_amd.slab import SlabRunner
hct = float(sys.argv[1]) if len(sys.argv) > 1 else 0.10
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
nx = int(sys.argv[3]) if len(sys.argv) > 3 else 256
host.init(0)
lib = host.capi.lib()
P = host.base_parameters()
r = SlabRunner(nx_local=nx, ny=256, nz=256, rank=0, world=1, P=P)
mask, R = host.pipe_mask(nx, 256, 256)
r.define_bounce_back(mask); r.lattice.latticeEquilibrium(1.0, (0, 0, 0)); r.lattice.setExternalVector((1e-6, 0, 0))
r.add_cell_type(host.CellType.rbc(P))
c, a = pack_pipe_rbc(nx, 256, 256, hct)
r.load_cells(0, c, a); r.prepare()
r.run(20)
host.check(lib.hc_synchronize())
cells = r.cells.ptr
for wide in ([int(w) for w in os.environ.get("IBM_WIDE", "0,1,0,1").split(",")]):
  pass
  out = {}
  for name, key, fn in (("spread", b"ibm_spread", lambda: lib.hcp_spread(cells, 1)), ("interpolate", b"ibm_interpolate", lambda: lib.hcp_interpolate(cells)),
                        ("mechanics", b"mechanics", lambda: lib.hcp_mechanics(cells, 0, 1))):
      for _ in range(3): host.check(fn())
      host.check(lib.hc_synchronize()); lib.hc_profile_reset(); lib.hc_profile_enable(1)
      t0 = time.perf_counter()
      for _ in range(n): host.check(fn())
      host.check(lib.hc_synchronize()); t1 = time.perf_counter()
      ms, k = C.c_double(), C.c_long(); host.check(lib.hc_profile_read(key, C.byref(ms), C.byref(k))); lib.hc_profile_enable(0)
      out[name] = (ms.value / max(k.value, 1), (t1 - t0) / n * 1e3)
  nv = r.owned_vertices()
  print("wide %d hct %.2f cells %d vertices %d :" % (wide, hct, len(c), nv), "  ".join("%s %.4f ms (wall %.4f)" % (k, v[0], v[1]) for k, v in out.items()), flush=True)

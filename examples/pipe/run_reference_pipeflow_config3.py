"""python examples/pipe/run_reference_pipeflow_config3.py [workdir]

BASELINE config 3 geometry on one GPU through the reference's OWN driver: build/ref_drivers/pipeflow_nohdf5 is
examples/pipeflow/pipeflow.cpp of the reference tree, compiled unchanged against the facade by __graft_entry__.build()
(without the HDF5 writers: at this size every writeOutput() would compress about 1 GB).  Inputs: the reference's
tube.stl, RBC.xml, PLT.xml and config.xml from tests/golden/pipeflow_case with refDirN raised from 50 to 254, which
makes the voxelised tube 511 x 257 x 257 nodes, and an RBC.pos at 10 % hematocrit from this repository's packer.
The time per iteration is taken from two runs that differ only in tmax."""
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hemocell_amd.packing import pack_pipe_rbc   # noqa: E402

CASE = os.path.join(ROOT, "tests", "golden", "pipeflow_case")
REF_N = 254
NX, NY = 2 * REF_N + 3, REF_N + 3


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "examples", "pipe", "tmp_config3")
    drv = os.path.join(ROOT, "build", "ref_drivers", "pipeflow_nohdf5")
    if not os.path.exists(drv):
        sys.exit("build/ref_drivers/pipeflow_nohdf5 is missing: run __graft_entry__.build() where the reference tree is present")
    os.makedirs(work, exist_ok=True)
    for name in ("tube.stl", "RBC.xml", "PLT.xml"):
        shutil.copy(os.path.join(CASE, name), os.path.join(work, name))
    centres, angles = pack_pipe_rbc(NX, NY, NY, 0.10)
    with open(os.path.join(work, "RBC.pos"), "w") as f:
        f.write("%d\n" % len(centres))
        for c, a in zip(centres, angles):
            f.write("%.6f %.6f %.6f %.4f %.4f %.4f\n" % (c[0] * 0.5, c[1] * 0.5, c[2] * 0.5, a[0], a[1], a[2]))
    with open(os.path.join(work, "PLT.pos"), "w") as f:
        f.write("0\n")
    cfg = open(os.path.join(CASE, "config.xml")).read()
    times, logs = {}, {}
    for tmax in (100, 1100):
        c = re.sub(r"<refDirN>[^<]*</refDirN>", "<refDirN> %d </refDirN>" % REF_N, cfg)
        c = re.sub(r"<warmup>[^<]*</warmup>", "<warmup> 0 </warmup>", c)
        c = re.sub(r"<tmax>[^<]*</tmax>", "<tmax> %d </tmax>" % tmax, c)
        c = re.sub(r"<tmeas>[^<]*</tmeas>", "<tmeas> %d </tmeas>" % tmax, c)
        for tag in ("tcsv", "tcheckpoint", "tbalance"):
            c = re.sub(r"<%s>[^<]*</%s>" % (tag, tag), "<%s> 100000000 </%s>" % (tag, tag), c)
        open(os.path.join(work, "config.xml"), "w").write(c)
        shutil.rmtree(os.path.join(work, "tmp"), ignore_errors=True)
        t0 = time.perf_counter()
        r = subprocess.run([drv, "config.xml"], cwd=work, capture_output=True, text=True)
        times[tmax] = time.perf_counter() - t0
        if r.returncode != 0:
            sys.exit("driver failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
        logs[tmax] = r.stdout
    ms = (times[1100] - times[100]) / 1000 * 1e3
    stat = [l.strip() for l in logs[1100].splitlines() if "# of cells" in l or "viscosity" in l][-2:]
    print("the reference's pipeflow driver (unchanged) through the facade, tube %d x %d x %d: %.4f ms per iterate() = %.0f MLUPS\n  %s\n  set-up + 100 iterations: %.1f s"
          % (NX, NY, NY, ms, NX * NY * NY / ms / 1e3, " | ".join(stat), times[100]))


if __name__ == "__main__":
    main()

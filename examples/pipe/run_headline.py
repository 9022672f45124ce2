"""python examples/pipe/run_headline.py [workdir]

The headline workload of bench.py (pipe 256^3, 10 % hematocrit, stepParticleEvery 5, stepMaterialEvery 20) driven the way a
HemoCell user drives it: the reference-style case driver pipe_synthetic.cpp, compiled against the C++ facade
(hemocell_amd/compat) and linked with libhemocell_amd.so, reading config.xml / RBC.xml / RBC.pos.  It calls
hemocell.iterate() once per iteration and re-applies the driving force after it, as examples/pipeflow/pipeflow.cpp does.
The time per iteration is taken from two runs that differ only in tmax."""
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hemocell_amd.packing import pack_pipe_rbc   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
N = 256
CONFIG = """<?xml version="1.0" ?>
<hemocell>
<parameters> <warmup> 0 </warmup> <outputDirectory>tmp_out</outputDirectory> </parameters>
<ibm> <stepMaterialEvery> 20 </stepMaterialEvery> <stepParticleEvery> 5 </stepParticleEvery> </ibm>
<domain> <rhoP> 1025 </rhoP> <nuP> 1.1e-6 </nuP> <dx> 5e-7 </dx> <dt> 1e-7 </dt>
    <refDirN> %d </refDirN> <lengthN> %d </lengthN> <kBT> 4.100531391e-21 </kBT> <Re> 0.5 </Re> </domain>
<sim> <tmax> %d </tmax> <tmeas> %d </tmeas> <tcheckpoint> 100000000 </tcheckpoint> </sim>
</hemocell>
"""


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "tmp_headline")
    os.makedirs(work, exist_ok=True)
    libdir = os.path.join(ROOT, "hemocell_amd", "lib")
    drv = os.path.join(work, "pipe_synthetic")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wno-deprecated-declarations", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "hemocell_amd", "compat"), os.path.join(HERE, "pipe_synthetic.cpp"), "-o", drv,
                           "-L" + libdir, "-lhemocell_amd", "-Wl,-rpath," + libdir])
    centres, angles = pack_pipe_rbc(N, N, N, 0.10)
    with open(os.path.join(work, "RBC.pos"), "w") as f:       # micrometres and degrees, io/readPositionsBloodCells.cpp:218-227
        f.write("%d\n" % len(centres))
        for c, a in zip(centres, angles):
            f.write("%.6f %.6f %.6f %.4f %.4f %.4f\n" % (c[0] * 0.5, c[1] * 0.5, c[2] * 0.5, a[0], a[1], a[2]))
    with open(os.path.join(work, "PLT.pos"), "w") as f:
        f.write("0\n")
    for name in ("RBC.xml", "PLT.xml"):
        shutil.copy(os.path.join(HERE, name), os.path.join(work, name))
    import json
    # the kernels run a few per cent faster or slower from process to process on one box, so the two ways of driving the
    # workload are run alternately, twice each, and the collide kernel's own time is printed next to each result
    for rep in range(2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "1000", "--warmup", "100"],
                           cwd=ROOT, capture_output=True, text=True, check=True)
        j = json.loads(r.stdout.strip().splitlines()[-1])
        k = j["kernel_ms"]
        print("bench.py (hc_iterate, 1000 iterations per call):              %.4f ms per step    [collide alone %.4f ms, beside %.4f ms]"
              % (j["ms_per_step"], k["collide_stream_alone"]["ms_total"] / k["collide_stream_alone"]["launches"],
                 k["collide_stream_beside"]["ms_total"] / max(k["collide_stream_beside"]["launches"], 1)))
        times, stats, kern = {}, {}, {}
        for tmax in (200, 2200):
            with open(os.path.join(work, "config.xml"), "w") as f:
                f.write(CONFIG % (N - 2, N, tmax, tmax))
            shutil.rmtree(os.path.join(work, "tmp_out"), ignore_errors=True)
            t0 = time.perf_counter()
            out = subprocess.run([drv, "config.xml"], cwd=work, capture_output=True, text=True, check=True,
                                 env=dict(os.environ, HEMOCELL_PRINT_KERNEL_TIMES="1")).stdout
            times[tmax] = time.perf_counter() - t0
            stats[tmax] = [l for l in out.splitlines() if l.startswith("STAT")][-1]
            kern[tmax] = {l.split()[1]: float(l.split()[2]) for l in out.splitlines() if l.startswith("KERNEL")}
        ms = (times[2200] - times[200]) / 2000 * 1e3
        cells = int(stats[2200].split()[2])
        print("reference-style driver through the facade, one iterate() per step: %.4f ms per iterate() [collide alone %.4f ms, beside %.4f ms]   pipe %d^3, %d cells [%s]"
              % (ms, kern[2200].get("collide_stream_alone", 0), kern[2200].get("collide_stream_beside", 0), N, cells, stats[2200]))


if __name__ == "__main__":
    main()

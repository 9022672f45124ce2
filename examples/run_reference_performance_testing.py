"""python examples/run_reference_performance_testing.py [workdir] [tmax]

The reference's own performance harness on one GPU: build/ref_drivers/performance_testing_nohdf5 is
cases/performance_testing/performance_testing.cpp of the reference tree, compiled unchanged against the facade by
__graft_entry__.build() (without the HDF5 writers), run on the reference's inputs for the 1-rank case of its 33 %
hematocrit strong-scaling series (tests/golden/performance_case: config_1.xml, RBC.xml, RBC.pos with 10 935 cells;
256^3 fully periodic, tau = 1, velocities interpolated every step, membrane forces every 20 steps).
The time per iteration is taken from two runs that differ only in tmax."""
import os
import re
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASE = os.path.join(ROOT, "tests", "golden", "performance_case")


def main():
    work = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "examples", "tmp_performance_testing")
    tmax_long = int(sys.argv[2]) if len(sys.argv) > 2 else 500      # config_1.xml: 500
    drv = os.path.join(ROOT, "build", "ref_drivers", "performance_testing_nohdf5")
    if not os.path.exists(drv):
        sys.exit("build/ref_drivers/performance_testing_nohdf5 is missing: run __graft_entry__.build() where the reference tree is present")
    os.makedirs(work, exist_ok=True)
    for name in ("RBC.xml", "RBC.pos"):
        shutil.copy(os.path.join(CASE, name), os.path.join(work, name))
    cfg = open(os.path.join(CASE, "config.xml")).read()
    times, logs = {}, {}
    for tmax in (100, tmax_long):
        c = re.sub(r"<tmax>[^<]*</tmax>", "<tmax> %d </tmax>" % tmax, cfg)
        c = re.sub(r"<tmeas>[^<]*</tmeas>", "<tmeas> %d </tmeas>" % tmax, c)
        open(os.path.join(work, "config.xml"), "w").write(c)
        for d in ("tmp_1", "log_1"):
            shutil.rmtree(os.path.join(work, d), ignore_errors=True)
        t0 = time.perf_counter()
        r = subprocess.run([drv, "config.xml"], cwd=work, capture_output=True, text=True)
        times[tmax] = time.perf_counter() - t0
        if r.returncode != 0:
            sys.exit("driver failed:\n" + r.stdout[-3000:] + r.stderr[-3000:])
        logs[tmax] = r.stdout
    ms = (times[tmax_long] - times[100]) / (tmax_long - 100) * 1e3
    stat = [l.strip() for l in logs[tmax_long].splitlines() if "# of cells" in l or "nCells" in l]
    ncells = int(re.search(r"nCells \(global\) = (\d+)", logs[tmax_long]).group(1))   # cells of the 135 um .pos box that lie in the 128 um domain
    print("the reference's performance_testing driver (unchanged) through the facade, 256^3 periodic, hematocrit_33/RBC.pos: "
          "%.3f ms per iterate() = %.0f MLUPS, %.2f G vertex updates/s (%d cells)\n  %s\n  whole run of %d iterations incl. set-up: %.1f s"
          % (ms, 256 ** 3 / ms / 1e3, ncells * 642 / ms / 1e6, ncells, " | ".join(stat), tmax_long, times[tmax_long]))


if __name__ == "__main__":
    main()

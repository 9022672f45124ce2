"""The C++ host facade (hemocell_amd/compat): the reference's own case drivers must compile UNCHANGED against it
(checked here, in the build container, when the reference tree is present), and the repository's example
drivers must compile and link against libhemocell_amd.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("HEMOCELL_REFERENCE", "/root/reference")
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "hemocell_amd", "compat")]
DRIVERS = ["examples/pipeflow/pipeflow.cpp", "examples/stretchCell/stretchCell.cpp", "examples/oneCellShear/oneCellShear.cpp",
           "cases/performance_testing/performance_testing.cpp"]


@pytest.mark.parametrize("driver", DRIVERS)
def test_reference_driver_compiles_unchanged(driver):
    src = os.path.join(REF, driver)
    if not os.path.exists(src):
        pytest.skip("reference tree not present (it does not travel to the GPU box)")
    r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-DHEMOCELL_COMPAT_MAIN", "-Wno-deprecated-declarations"] + INC + [src],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("example", ["examples/stretch/stretch_cell.cpp", "examples/pipe/pipe_synthetic.cpp"])
def test_example_driver_links(tmp_path, example):
    from hemocell_amd import capi
    out = str(tmp_path / "drv")
    libdir = os.path.dirname(capi.LIB_PATH)
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wno-deprecated-declarations"] + INC + [os.path.join(ROOT, example), "-o", out,
                        "-L" + libdir, "-lhemocell_amd", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]

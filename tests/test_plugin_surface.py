"""The plugin surface of SURVEY.md section 8(b) beyond the four reference drivers: a user-written mechanics model (subclass of
CellMechanics against mechanics/cellMechanics.h:36-47) and a user IBM kernel compile unchanged against the facade and are
refused at run time with the reference's log + exit(1); HemoCellParticleField (particles, get_particles_per_cell, get_lpc),
HemoCellField::kernelMethod and CellMechanics::cellConstants exist and are filled from the device."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "hemocell_amd", "compat"), "-I" + os.path.join(ROOT, "tests", "plugin")]


def _build(tmp):
    from hemocell_amd import capi
    libdir = os.path.dirname(capi.LIB_PATH)
    exe = os.path.join(tmp, "plugin_driver")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-Wno-deprecated-declarations", "-DHEMOCELL_COMPAT_MAIN"] + INC +
                       [os.path.join(ROOT, "tests", "plugin", "plugin_driver.cpp"), "-o", exe, "-L" + libdir, "-lhemocell_amd", "-Wl,-rpath," + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return exe


def test_user_model_and_surface_compile_against_the_facade(tmp_path):
    _build(str(tmp_path))


def _case(tmp_path):
    d = str(tmp_path / "case")
    shutil.copytree(os.path.join(ROOT, "tests", "golden", "shear_case"), d)
    return d


@pytest.mark.gpu
def test_particle_field_view_and_constants(tmp_path, gpu):
    exe = _build(str(tmp_path)); d = _case(tmp_path)
    r = subprocess.run([exe, "surface", "config.xml"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "SURFACE OK" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    out = {l.split()[0]: l.split()[1:] for l in r.stdout.splitlines() if l.split() and l.split()[0] in ("constants", "particles", "force_sum", "cell")}
    assert out["constants"][:5] == ["1280", "1920", "642", "1920", "642"] and abs(float(out["constants"][5]) - 649.0) < 5   # RBC tables, volume_eq in lu^3
    p = out["particles"]
    assert p[0] == "642" and p[2] == "1" and p[4] == "1" and p[6] == "1" and p[8:10] == ["0", "39"] and p[11] == "642"
    assert all(abs(float(x)) < 1e-9 for x in out["force_sum"][:3])          # membrane forces of a free cell sum to zero
    assert abs(float(out["cell"][2]) - 20.0) < 0.05                         # the edit through pf.particles + upload() moved the cell from y = 19 to 20


@pytest.mark.gpu
@pytest.mark.parametrize("mode,fragment", [("usermodel", "is host code"), ("userkernel", "installs its own IBM kernelMethod")])
def test_host_plugins_are_refused_like_the_reference_refuses(tmp_path, gpu, mode, fragment):
    """log + exit(1) (core/hemoCell.cpp:75-78 style), never a silent fallback"""
    exe = _build(str(tmp_path)); d = _case(tmp_path)
    r = subprocess.run([exe, mode, "config.xml"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and fragment in r.stdout and "NOT REFUSED" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert fragment in open(os.path.join(d, "tmp", "log", "logfile")).read()


@pytest.mark.gpu
def test_external_field_is_zeroed_by_iterate_as_in_the_reference(tmp_path, gpu):
    """core/hemoCell.cpp:369-371: every iterate() ends with setExternalVector(..., 0); a driver that writes its force once drives
    one iteration only, one that writes it after every iteration (all shipped drivers) drives all of them"""
    exe = _build(str(tmp_path)); d = _case(tmp_path)
    r = subprocess.run([exe, "forceonce", "config.xml"], cwd=d, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l.split() for l in r.stdout.splitlines() if l.startswith("forceonce")][0]
    F, once, again = float(line[2]), float(line[4]), float(line[6])
    assert 0.8 * F < once < 1.2 * F               # one iteration's worth of momentum in a fully periodic box (rho = 1)
    # second loop: its first iteration finds the field zeroed, the other nine are driven, and the last write is seen by the
    # statistics as F / 2 in Cell::computeVelocity but not yet by a step: 9.5 F more
    assert 9.0 * F < again - once < 10.0 * F

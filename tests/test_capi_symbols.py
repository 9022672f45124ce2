"""the C-ABI library loads and exports every symbol include/hemocell_amd.h declares (no compute calls)"""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "hemocell_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^(?:const\s+char\s*\*\s*|int\s+|size_t\s+|double\s+)(hc[a-z_0-9]*)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_symbols_exported_and_bound():
    from hemocell_amd import capi
    names = _declared()
    assert len(names) >= 50
    lib = ctypes.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libhemocell_amd.so does not export " + n
    assert sorted(capi.SIGNATURES) == names   # the Python binding covers exactly the declared ABI
    capi.lib()


def test_no_cpu_fallback_without_gpu():
    """on a machine without a GPU hc_init must fail loudly (the product path has no CPU fallback)"""
    import torch
    from hemocell_amd import capi
    if torch.cuda.is_available():
        return
    rc = capi.lib().hc_init(0)
    assert rc != 0
    assert b"no HIP device" in capi.lib().hc_last_error() or b"HIP error" in capi.lib().hc_last_error()


def test_product_never_imports_oracle():
    """only tests/, __graft_entry__.smoke and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(ROOT, "hemocell_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "libhemo_oracle" not in txt, f
                assert not re.search(r"^\s*(from\s+oracle|import\s+oracle)", txt, flags=re.M), f
                assert not re.search(r"#include\s*[<\"][^>\"]*oracle", txt), f


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """the boundary is a C ABI: include/hemocell_amd.h must compile as strict C99 and a C program must link against the
    library with nothing but that header (tests/cabi/c_abi_smoke.c; it is run on the GPU by tests/test_gpu_parity.py)"""
    import subprocess
    from hemocell_amd import capi
    capi.lib()
    libdir = os.path.dirname(capi.LIB_PATH)
    out = str(tmp_path / "c_abi_smoke")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cabi", "c_abi_smoke.c"), "-o", out, "-L" + libdir, "-lhemocell_amd", "-lm",
                        "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]

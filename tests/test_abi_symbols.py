"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/crsdr.h declares, validates arguments, and fails loudly (no CPU fallback) without a GPU."""
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def b():
    binding = importlib.import_module("coherent-rtlsdr_amd.binding")
    binding.build()
    return binding


def test_header_symbols_all_exported(b):
    hdr = open(os.path.join(ROOT, "include", "crsdr.h")).read()
    declared = sorted(set(re.findall(r"\b(crsdr_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = b.lib()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(b.ABI_SYMBOLS) == declared
    assert lib.crsdr_abi_version() == 1


def test_argument_validation_precedes_device_use(b):
    with pytest.raises(b.CrsdrError) as e:
        b.fft(np.zeros(12, dtype=np.complex64))          # not a power of two
    assert e.value.code in (-1, -4)
    with pytest.raises(b.CrsdrError) as e:
        b.Plan(1, 1024)                                   # no signal row
    assert e.value.code == -1
    with pytest.raises(b.CrsdrError) as e:
        b.Plan(4, 1000)                                   # blocksize not a power of two
    assert e.value.code == -1
    with pytest.raises(b.CrsdrError) as e:
        b.Plan(4, 1024, row_begin=3, row_count=5)         # slab outside the matrix
    assert e.value.code == -1


def test_no_cpu_fallback_without_device(b):
    if b.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(b.CrsdrError) as e:
        b.convtofloat(np.zeros(16, dtype=np.int8))
    assert e.value.code == -4                             # CRSDR_ENODEV
    with pytest.raises(b.CrsdrError) as e:
        b.Plan(4, 1024)
    assert e.value.code == -4


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "coherent-rtlsdr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".cpp", ".c")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_py" not in text and "coherent_oracle" not in text and "liboracle" not in text, f


def test_header_is_plain_c99_and_links_from_c(b, tmp_path):
    # the boundary is a C ABI (cgo / JNI / ctypes bind it): include/crsdr.h must compile as strict C99 and the
    # library must link from a C program.  On this CPU-only box the calls then fail loudly with CRSDR_ENODEV.
    import subprocess
    so_dir = os.path.dirname(b._SO)
    exe = tmp_path / "abi_c99"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                        "-o", str(exe), os.path.join(ROOT, "tests", "c", "abi_c99.c"), "-L", so_dir, "-lcrsdr", f"-Wl,-rpath,{so_dir}"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "abi 1" in out.stdout, out.stdout + out.stderr
    if b.device_count() < 1:
        assert "rc -4" in out.stdout and "plan_create rc -4" in out.stdout
    else:
        assert "plan_create rc 0" in out.stdout

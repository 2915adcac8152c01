"""The C-ABI library loads and exports every symbol include/ramx.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

from repeatafterme_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "ramx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b(ramx_[a-z0-9_]+)\s*\(", txt))
    names -= {"ramx_allreduce_cb"}
    return names


def test_header_and_export_list_agree():
    assert _declared() == set(_lib.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build libramx.so first (__graft_entry__.build())"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(_declared()):
        assert hasattr(lib, name), f"{name} declared in include/ramx.h but not exported"


def test_no_cpu_fallback_symbols_and_no_oracle_linkage():
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "ramx_oracle" not in out
    ldd = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "ramref" not in ldd
    assert "libamdhip64" in ldd


def test_cli_binary_fails_loudly_without_gpu_or_runs(tmp_path):
    """On a box without a GPU the CLI must stop with a clear message and a non-zero status, after
    printing the same banner/parameter/core table as the reference; on a GPU box it simply runs."""
    g = os.path.join(ROOT, "tests", "golden", "inputs")
    r = subprocess.run([_lib.CLI_PATH, "-twobit", os.path.join(g, "extension-test2.2bit"), "-ranges",
                        os.path.join(g, "extension-test2.tsv")], capture_output=True, text=True)
    lib = _lib.lib()
    gold = open(os.path.join(ROOT, "tests", "golden", "cli", "t2_default", "stdout")).read().splitlines()
    if lib.ramx_device_count() > 0:
        assert r.returncode == 0 and "Extended right: 177 bp" in r.stdout
    else:
        assert r.returncode == 1
        assert "no HIP device" in r.stderr
        mine = r.stdout.splitlines()
        assert "Extended right" not in r.stdout
        assert len(mine) >= 20
        for i, (a, b) in enumerate(zip(mine, gold)):
            if "RAMExtend Version" in b or "_FILE" in b:
                continue
            assert a == b, f"stdout line {i}"

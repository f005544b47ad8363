"""CPU: the C-ABI library loads without a GPU and exports every symbol include/zeldovich_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "zeldovich_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(zd_[A-Za-z0-9_]+)\s*\(", text))
    names -= {"zd_slab_cb"}
    return names


def test_header_and_library_agree():
    import zeldovich_plt_amd.api as api
    lib = api.load_library()
    declared = _declared_functions()
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "header declares %s but the library does not export it" % name
    assert declared == set(api.EXPORTED_SYMBOLS), declared ^ set(api.EXPORTED_SYMBOLS)


def test_struct_layouts_match_header():
    """sizes the C compiler gives the ABI structs == the ctypes mirrors"""
    import subprocess
    import tempfile
    import zeldovich_plt_amd.api as api
    src = '#include <stdio.h>\n#include "zeldovich_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n",sizeof(zd_params),sizeof(zd_pk),sizeof(zd_stats),sizeof(zd_param_strings));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = [int(v) for v in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(api.ZdParams), ctypes.sizeof(api.ZdPk), ctypes.sizeof(api.ZdStats),
                     ctypes.sizeof(api.ZdParamStrings)]


def test_missing_library_fails_loudly(monkeypatch):
    import zeldovich_plt_amd.api as api
    monkeypatch.setattr(api, "_lib", None)
    monkeypatch.setattr(api, "LIB_PATH", "/nonexistent/libzeldovich_hip.so")
    try:
        api.load_library()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load_library() must raise when the HIP library is missing")

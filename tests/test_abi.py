"""CPU: the C-ABI library builds/loads and exports every symbol include/rnet_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "rnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rn_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from retinanet import _C
    assert _declared() == _C.exported_symbols()


def test_library_exports_every_declared_symbol():
    from retinanet import _C
    lib = _C.lib()
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.rn_abi_version() == _C.ABI_VERSION


def _exported(path):
    """unmangled dynamic symbols of a build (C++ / kernel-stub symbols are mangled `_Z...`, HIP's own are `__hip_*`)"""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    return sorted(n for n in names if not n.startswith(("_Z", "__hip", "_init", "_fini", "__bss", "_edata", "_end")))


def test_library_exports_nothing_but_the_declared_interface():
    """The reverse inclusion: every C symbol either build exports is declared in include/rnet_hip.h
    — no undeclared `rn_debug_*` knobs; the library is built with -fvisibility=hidden and the header opens the
    default-visibility region."""
    from retinanet import _C
    for path in (_C.LIB_PATH, _C.LIB_PATH_F16):
        assert _exported(path) == _declared(), path


def test_no_mutable_process_wide_state_setters():
    """launch options travel per call (rn_launch_opts in the problem descriptors) or per rn_handle: the header declares
    no process-wide setter"""
    assert not [n for n in _declared() if n.startswith(("rn_set_", "rn_debug"))]
    from retinanet import _C
    assert _C.ConvProblem.opts.size == _C.WgradProblem.opts.size == ctypes.sizeof(_C.LaunchOpts) == 36


def test_half_build_exports_the_same_interface():
    """librnet_hip_f16.so: the same sources compiled with -DRN_F16 (IEEE-half storage, `mixed_float16` configs)"""
    from retinanet import _C
    lib = _C.lib(True)
    for name in _declared():
        assert hasattr(lib, name), name
    assert lib.rn_abi_version() == _C.ABI_VERSION
    assert lib.rn_storage_dtype() == 1 and _C.lib().rn_storage_dtype() == 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "retinanet-tensorflow2.x_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "model_ref" not in txt, f

"""CPU: the C-ABI library builds/loads and exports every symbol include/rnet_hip.h declares."""
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

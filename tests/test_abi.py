"""CPU-side checks of the product boundary: the C-ABI library loads, exports every symbol that
include/arkbp.h declares, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from ark_bulletproofs_amd import build

    build.build()
    from ark_bulletproofs_amd import _lib

    return _lib


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "arkbp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(bp_[A-Za-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), "libarkbp_hip.so does not export %s" % name
    assert sorted(declared) == sorted(lib.EXPORTS)


def test_no_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = lib.lib()
    assert L.bp_device_count() == 0
    ctx = C.c_void_p()
    assert L.bp_ctx_create(0, 0, C.byref(ctx)) == lib.BP_E_NO_DEVICE
    from ark_bulletproofs_amd import ArkbpError, Engine

    with pytest.raises(ArkbpError):
        Engine()


def test_product_does_not_reference_oracle():
    # the product path must not import, link or call anything under oracle/
    pkg = os.path.join(ROOT, "ark_bulletproofs_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("Independent of oracle/", ""), f

"""CPU: the C-ABI library is built, loads without a GPU, exports every symbol include/ssk_capi.h declares,
and refuses to create a context without a gfx950 device (no CPU fallback)."""
import os
import re

import pytest


def test_library_exports_all_declared_symbols(pkg):
    lib = pkg.load_library()
    names = pkg.declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.ssk_api_version() == 1


def test_facade_library_exports_the_rba_engine_abi(pkg):
    """libssk_facade.so exports every function include/ssk_rba.h and include/ssk_slam.h declare (loads without a GPU)."""
    from ssk_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = []
    for hdr, pat in (("ssk_rba.h", r"\b(ssk_rba_[a-z0-9_]+)\s*\("), ("ssk_slam.h", r"\b(ssk_slam_[a-z0-9_]+)\s*\(")):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(root, "include", hdr)).read(), flags=re.S)
        names += sorted(set(re.findall(pat, src)))
    assert len(names) >= 20, names
    lib = capi.load_facade_library()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    with pytest.raises(pkg.SskError) as e:
        pkg.Context(0)
    assert e.value.code == 1 and "no CPU fallback" in str(e.value)


def test_product_never_references_oracle():
    """oracle/ is test infrastructure: nothing under the package may include, link or load it."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkgdir = os.path.join(root, "srba-stereo-slam_amd")
    bad = []
    for dp, _, fs in os.walk(pkgdir):
        if "_obj" in dp:
            continue
        for f in fs:
            if f.endswith((".hip", ".h", ".cpp", ".hpp", ".py")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                code = re.sub(r"//.*|/\*.*?\*/|#.*|\"\"\".*?\"\"\"", "", src, flags=re.S) if f.endswith(".py") else \
                    re.sub(r"//.*|/\*.*?\*/", "", src, flags=re.S)
                if re.search(r"ssko_|ssk_oracle|oracle_lib|libssk_oracle", code):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad

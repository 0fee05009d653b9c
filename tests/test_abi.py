"""CPU tier: the C-ABI library loads and exports every symbol include/vvenc_hip.h declares (no compute, no GPU),
and the product path fails loudly without a GPU instead of falling back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "vvenc_hip.h")).read()
    return sorted(set(re.findall(r"VVHIP_API\s+[\w\s\*]+?\b(vvhip_\w+)\s*\(", txt)))


def test_header_symbols_exported():
    from vvenc_amd.lib import LIB_PATH, PROTOTYPES, load_library
    assert os.path.exists(LIB_PATH), "build with `make -C vvenc_amd/csrc` (or __graft_entry__.build())"
    lib = C.CDLL(LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(PROTOTYPES) == names, (set(PROTOTYPES) ^ set(names))
    load_library()


def test_rom_tables_match_golden():
    """host-side ROM accessors need no GPU: transform matrices and scan orders vs the reference's"""
    import numpy as np
    import golden_replay as G
    from vvenc_amd.lib import load_library
    L = load_library()

    class Rom:
        def tr_matrix(self, t, l):
            n = 1 << l
            out = np.zeros((n, n), np.int16)
            return out if L.vvhip_get_tr_matrix_host(t, l, out.ctypes.data_as(C.c_void_p)) == 0 else None

        def scan_order(self, lw, lh):
            out = np.zeros(1 << (lw + lh), np.uint32)
            assert L.vvhip_get_scan_order_host(lw, lh, out.ctypes.data_as(C.c_void_p)) == 0
            return out
    G.check_transform_matrices(Rom())
    G.check_scan(Rom())


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vvenc_amd.hotpath import HotPath
    from vvenc_amd.lib import VVHipError, load_library
    with pytest.raises(VVHipError):
        HotPath()
    L = load_library()
    ctx = C.c_void_p()
    assert L.vvhip_create(C.byref(ctx), 0) != 0
    assert b"no CPU fallback" in L.vvhip_last_error(None) or b"HIP" in L.vvhip_last_error(None)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "vvenc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt, (dirpath, f)

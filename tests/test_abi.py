"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/gvcnn_hip.h declares, and rejects bad arguments with error codes (no compute, no GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gvcnn_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as entry
    entry.build_library()
    import gvcnn_tf_amd
    return gvcnn_tf_amd._lib.load()


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gv_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    import gvcnn_tf_amd
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libgvcnn_hip.so does not export %s" % n
    # the ctypes table covers exactly the header
    assert sorted(gvcnn_tf_amd._lib.SIGNATURES) == names


def test_no_torch_in_abi():
    """The boundary is plain C: pointers and sizes only."""
    text = open(HEADER).read()
    assert "torch" not in text.lower() and "at::" not in text and "std::" not in text
    assert 'extern "C"' in text


def test_abi_version_and_error_strings(lib):
    assert lib.gv_abi_version() == 1
    assert lib.gv_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4):
        assert b"gvcnn" in lib.gv_error_string(code)


def test_bad_arguments_return_codes(lib):
    from gvcnn_tf_amd import _lib
    d = _lib.ConvDesc(1, 8, 8, 16, 16, 3, 3, 1, 1, 1, 8, 8, 16, 16, 0, 0, 0, _lib.GV_F32, 0, 0, 0)
    assert lib.gv_conv2d_fwd(C.byref(d), None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.gv_conv2d_fwd(None, 16, 16, 16, 16, None, 16, None, None, None, None) == -1
    d.cout = 0
    assert lib.gv_conv2d_fwd(C.byref(d), 16, 16, 16, 16, None, 16, None, None, None, None) == -1
    d.cout = 16
    d.tile_cfg = 99                              # not a tile configuration
    assert lib.gv_conv2d_fwd(C.byref(d), 16, 16, 16, 16, None, 16, None, None, None, None) == -1
    d.tile_cfg = 0
    d.flags = _lib.GV_CONV_SPLIT                 # split without a second destination
    assert lib.gv_conv2d_fwd(C.byref(d), 16, 16, 16, 16, None, 16, None, None, None, None) == -1
    d.flags = 0
    d.math_mode = 9
    assert lib.gv_conv2d_fwd(C.byref(d), 16, 16, 16, 16, None, 16, None, None, None, None) == -1
    d.math_mode = 0
    d.dtype = 7
    assert lib.gv_conv2d_fwd(C.byref(d), 16, 16, 16, 16, None, 16, None, None, None, None) == -2
    d.dtype = _lib.GV_F32
    d.y_ld = 8                                   # pixel stride smaller than cout
    assert lib.gv_conv2d_fwd(C.byref(d), 16, 16, 16, 16, None, 16, None, None, None, None) == -1
    d.y_ld = 16                                  # pre-activation on load without its scale / shift; on a null plan
    assert lib.gv_conv2d_fwd_xpre(C.byref(d), 16, None, None, 16, 16, 16, None, 16, None, None, None, None) == -1
    assert lib.gv_plan_set_conv_xpre(None, 0, 0, 0) == -4
    pm = _lib.PoolDesc(1, 8, 8, 4, 4, 3, 3, 2, 0, 0, 3, 3, 4, _lib.GV_POOL_AVG, _lib.GV_F32)   # argmax pools: max only
    assert lib.gv_pool2d_fwd_argmax(C.byref(pm), 16, 16, 16, None) == -1
    assert lib.gv_pool2d_bwd_argmax(C.byref(pm), 16, 16, 4, 16, 4, None) == -1
    pm.mode = _lib.GV_POOL_MAX
    assert lib.gv_pool2d_fwd_argmax(C.byref(pm), 16, 16, None, None) == -1
    assert lib.gv_pool2d_bwd_argmax(C.byref(pm), None, 16, 4, 16, 4, None) == -1
    p = _lib.PoolDesc(1, 8, 8, 4, 4, 3, 3, 2, 0, 0, 3, 3, 4, 5, _lib.GV_F32)   # bad mode
    assert lib.gv_pool2d_fwd(C.byref(p), 16, 16, None) == -1
    # a window without a valid tap (0/0 in an average, -inf / a 0xff argmax byte in a max) is a bad descriptor for every
    # dtype and storage form: last window starting past the image, padding >= window
    for mode in (_lib.GV_POOL_MAX, _lib.GV_POOL_AVG, _lib.GV_POOL_AVG | _lib.GV_POOL_X_P3):
        for dt in (_lib.GV_F32, _lib.GV_BF16):
            past = _lib.PoolDesc(1, 8, 8, 16, 16, 3, 3, 2, 0, 0, 6, 3, 16, mode, dt)      # (6-1)*2 = 10 >= ih = 8
            assert lib.gv_pool2d_fwd(C.byref(past), 16, 16, None) == -1
            wide = _lib.PoolDesc(1, 8, 8, 16, 16, 3, 3, 1, 1, 3, 8, 8, 16, mode, dt)      # pad_l = 3 >= kw = 3
            assert lib.gv_pool2d_fwd(C.byref(wide), 16, 16, None) == -1
    past = _lib.PoolDesc(1, 8, 8, 16, 16, 3, 3, 2, 0, 0, 3, 6, 16, _lib.GV_POOL_MAX, _lib.GV_BF16)
    assert lib.gv_pool2d_fwd_argmax(C.byref(past), 16, 16, 16, None) == -1
    assert lib.gv_pool2d_bwd_argmax(C.byref(past), 16, 16, 16, 16, 16, None) == -1
    assert lib.gv_group_assign(None, 6, 10, 10, None, None, None, None, None) == -1
    assert lib.gv_group_assign(16, 65, 10, 10, 16, 16, 16, 16, None) == -2        # V > 64
    assert lib.gv_view_pool_fuse_fwd(16, 6, 2, 64, 64, 384, 16, 10, 16, 9, 1.0, None, 16, 0, None) == -1
    assert lib.gv_view_pool_fuse_fwd(16, 6, 2, 64, 64, 384, 16, 10, 16, 0, 1.0, None, None, 0, None) == -1
    assert lib.gv_packed_filter_bytes(3, 3, 3, 32, 0, 0) == 4 * 32 * 32        # K=27 -> Kpad 32
    assert lib.gv_packed_filter_bytes(1, 1, 64, 80, 0, 0) == 4 * 80 * 64
    assert lib.gv_packed_filter_bytes(3, 3, 80, 192, 0, 0) == 4 * 192 * 736    # K=720 -> Kpad 736
    assert lib.gv_packed_filter_bytes(3, 3, 80, 192, 0, 1) == 192 * 45 * 3 * 32  # 45 k-tiles x 3 bf16 planes
    assert lib.gv_packed_filter_bytes(3, 3, 80, 192, 0, 9) == -1
    assert lib.gv_plan_run(None, None, 0, None) == -4
    plan = C.c_void_p()
    assert lib.gv_plan_create(C.byref(plan)) == 0
    assert lib.gv_plan_num_ops(plan) == 0
    assert lib.gv_plan_add_conv(plan, None, 0, 0, 1, 0, 2, 0, 0, -1, 0, 3, 0, -1, 0, 0, 0) == -1
    lib.gv_plan_destroy(plan)


def test_missing_library_is_loud(tmp_path, monkeypatch):
    """The product must fail loudly when the HIP extension is missing (no CPU fallback)."""
    from gvcnn_tf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "gvcnn-tf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "libgvref" not in src, f

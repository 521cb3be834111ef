"""CPU: the C-ABI shared library loads and exports EVERY symbol include/metrabs_hip.h declares; the
ctypes table in metrabs_amd/_lib.py covers exactly that set; struct layouts match; argument checks
that need no GPU return the documented error codes (no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, 'include', 'metrabs_hip.h')


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    names = re.findall(r'^\s*(?:int|size_t|const char\*)\s+(mtr_\w+)\s*\(', text, flags=re.M)
    return sorted(set(names))


@pytest.fixture(scope='module')
def lib():
    from metrabs_amd import build, _lib
    build.build_library(verbose=False)  # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_header_declares_expected_surface():
    names = declared_functions()
    assert len(names) >= 14 and 'mtr_softargmax_decode' in names and 'mtr_warp_crops' in names


def test_library_exports_every_declared_symbol(lib):
    from metrabs_amd import _lib
    for name in declared_functions():
        assert hasattr(lib, name), f'{name} is declared in the header but not exported'
    assert sorted(_lib.SIGNATURES) == declared_functions(), 'ctypes table out of sync with the header'


def test_struct_layouts_match_header():
    from metrabs_amd import _lib
    assert ctypes.sizeof(_lib.HeadParams) == 20 and ctypes.sizeof(_lib.ReconParams) == 36
    assert _lib.HeadParams.box_size_mm.offset == 16 and _lib.ReconParams.mix_3d_inside_fov.offset == 20
    text = open(HEADER).read()
    assert int(re.search(r'#define MTR_WARP_PARAM_FLOATS (\d+)', text).group(1)) == _lib.MTR_WARP_PARAM_FLOATS


def test_version_strerror_and_host_side_argument_checks(lib):
    from metrabs_amd import _lib
    assert lib.mtr_version() == 100
    assert lib.mtr_strerror(0) == b'ok' and b'workspace' in lib.mtr_strerror(-5)
    hp = _lib.HeadParams(256, 32, 1, 0, 2200.0)
    # NULL pointers / bad shapes are rejected on the host before any launch
    assert lib.mtr_softargmax_decode(None, 0, 0, 1, 17, 8, 8, 8, ctypes.byref(hp), None, None, None) == -1
    assert lib.mtr_reconstruct_absolute(None, None, None, 1, 17, None, None, None, 0, None) == -1
    assert lib.mtr_warp_crops(None, None, None, 1, 8, 8, None, 1, 8, 1, 0, 0, None, None) == -1
    # f32 features: the row-tile blob = 40 stages x 10 tiles x 2 KiB + 160 rows x (bias f32 + label i32)
    assert lib.mtr_head_packed_bytes(1280, 17, 8, 0) == 40 * 10 * 2048 + 160 * 8
    # 72 depth bins (the metric string's shape): 17 atoms of 5 row tiles
    assert lib.mtr_head_packed_bytes(1280, 17, 72, 0) == 40 * 85 * 2048 + 85 * 16 * 8
    assert lib.mtr_head_packed_bytes(1280, 17, 81, 0) == 0  # > 80 depth bins: library GEMM + decode
    # f16 features: 3 joint groups x 64 rows of bias (f32) + 3 x 20 stages x 64 rows x 64 channels x 2 B
    # + the 16-bit row-tile section (C % 64 == 0): 20 stages x 10 tiles x 2 KiB + 160 rows x 8 B
    joint_groups = 3 * 64 * 4 + 3 * 20 * 64 * 64 * 2
    # + (round 5) the joint-group weights once more, fragment-major (whole 64-channel stages): 3 x 20 x 8 KiB
    assert lib.mtr_head_packed_bytes(1280, 17, 8, 1) == joint_groups + 20 * 10 * 2048 + 160 * 8 + 3 * 20 * 8192
    assert lib.mtr_head_packed_bytes(1283, 17, 8, 1) == 0   # C % 8 != 0: no 16-byte operands
    assert lib.mtr_head_packed_bytes(1288, 17, 8, 1) == 3 * 64 * 4 + 3 * 21 * 64 * 64 * 2  # C % 64 != 0: joint groups only
    # a joint's 73 rows exceed the 64-row group: the row-tile section alone (17 atoms of 5 tiles)
    assert lib.mtr_head_packed_bytes(1280, 17, 72, 2) == 20 * 85 * 2048 + 85 * 16 * 8
    assert lib.mtr_head_packed_bytes(1288, 17, 72, 2) == 0
    # the workspace of mtr_head_fused_ws: f32 maps of > 64 positions (column-block statistics), 16-bit
    # row-tile shapes (an NHWC copy of NCHW features + the statistics)
    assert lib.mtr_head_workspace_bytes(0, 0, 64, 1280, 8, 8, 17, 8) == 0
    assert lib.mtr_head_workspace_bytes(0, 0, 32, 1280, 12, 12, 17, 8) == 32 * 3 * 160 * 5 * 8
    assert lib.mtr_head_workspace_bytes(1, 0, 64, 1280, 8, 8, 17, 8) == 0
    assert lib.mtr_head_workspace_bytes(1, 0, 64, 1280, 8, 8, 17, 72) == 64 * 1280 * 64 * 2
    assert lib.mtr_head_workspace_bytes(1, 1, 64, 1280, 8, 8, 17, 72) == 0
    assert lib.mtr_reconstruct_workspace_bytes(64, 17) == (4 + 2 * 16) * 8


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (checked statically)."""
    pkg = os.path.join(ROOT, 'metrabs_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f


def test_missing_library_fails_loudly(tmp_path):
    from metrabs_amd import _lib
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load(str(tmp_path / 'nope.so'))


def test_every_entry_point_rejects_null_pointers_on_the_host(lib):
    """All pointer arguments NULL, integers 1, floats 1.0: every int-returning entry point answers
    MTR_E_NULL before it touches a device (no GPU here), none dereferences first."""
    from metrabs_amd import _lib
    checked = 0
    for name, (res, args) in sorted(_lib.SIGNATURES.items()):
        if res is not ctypes.c_int or name == 'mtr_version':
            continue
        vals = []
        for a in args:
            if a in (ctypes.c_int, ctypes.c_uint, ctypes.c_size_t, ctypes.c_longlong):
                vals.append(1)
            elif a in (ctypes.c_float, ctypes.c_double):
                vals.append(1.0)
            else:
                vals.append(None)  # data pointers, parameter structs, the stream
        assert getattr(lib, name)(*vals) == -1, name
        checked += 1
    assert checked >= 16

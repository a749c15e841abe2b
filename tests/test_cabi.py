"""The C-ABI library builds, loads without a GPU and exports every symbol include/dmh_hip.h declares."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(REPO, "include", "dmh_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmh_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from depthmodelhardening_amd import _native
    from depthmodelhardening_amd.build import build
    build(verbose=False)
    lib = _native.lib()
    names = _declared()
    assert len(names) >= 23
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_native.EXPORTS) == names, "ctypes signature table and header disagree"
    assert b"gfx950" in lib.dmh_version()


def test_struct_layouts_match_the_header(tmp_path):
    from depthmodelhardening_amd import _native
    src = tmp_path / "sz.c"
    src.write_text('#include "dmh_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%zu %zu %zu %zu %zu\\n",'
                   'sizeof(dmh_photo_args),sizeof(dmh_smooth_args),sizeof(dmh_paste_args),'
                   'offsetof(dmh_photo_args,seed),offsetof(dmh_paste_args,mode));return 0;}\n')
    exe = tmp_path / "sz"
    assert os.system("gcc -I%s %s -o %s" % (os.path.join(REPO, "include"), src, exe)) == 0
    out = os.popen(str(exe)).read().split()
    assert [int(v) for v in out] == [ctypes.sizeof(_native.PhotoArgs), ctypes.sizeof(_native.SmoothArgs),
                                     ctypes.sizeof(_native.PasteArgs), _native.PhotoArgs.seed.offset,
                                     _native.PasteArgs.mode.offset]


def test_host_side_argument_checks_without_gpu():
    """Shape errors are rejected on the host before any launch (no GPU needed)."""
    from depthmodelhardening_amd import _native as N
    lib = N.lib()
    a = N.PasteArgs()
    assert lib.dmh_eot_paste_fwd(ctypes.byref(a), None, None, None) == 1
    assert b"null input" in lib.dmh_last_error() or b"requirement" in lib.dmh_last_error()
    assert lib.dmh_pgd_linf_step(None, None, None, 0.1, 0.1, None, 0, None) == 1
    p = N.PhotoArgs()
    assert lib.dmh_photo_loss_fwd(ctypes.byref(p), (ctypes.c_void_p * 4)(), (ctypes.c_void_p * 4)(), None, None) == 1


def test_convolution_entry_points_reject_bad_shapes_without_gpu():
    """K10-K13 / K9 statistics: host-side argument checks and size helpers (no launch, no GPU)."""
    from depthmodelhardening_amd import _native as N
    lib = N.lib()
    one = ctypes.c_void_p(16)           # a non-NULL dummy: every call below must fail before any launch
    assert lib.dmh_wino_weight_size(64, 64) == (64 // 8) * 16 * 2 * 64 * 4
    assert lib.dmh_wino_weight_size(96, 32) == (32 // 8) * 16 * 2 * 128 * 4      # output channels padded to 64s
    assert lib.dmh_wino_weight_size(64, 12) == -1
    assert lib.dmh_wino32_weight_size(32, 96) == (96 // 8) * 16 * 2 * 32 * 4
    assert lib.dmh_wino32_weight_size(96, 32) == (32 // 8) * 16 * 2 * 96 * 4       # output channels padded to 32s
    assert lib.dmh_wino32_conv3x3(one, one, None, 1, 16, 32, 8, 8, 1, one, None) != 0        # < 24 input channels
    assert lib.dmh_wino_wrw_workspace_size(2, 64, 64, 16, 32, 1) > 0 and lib.dmh_wino_wrw_workspace_size(2, 96, 32, 16, 32, 1) > 0 and lib.dmh_wino_wrw_workspace_size(2, 48, 64, 16, 32, 1) == -1 and lib.dmh_wino_wrw_workspace_size(2, 64, 64, 16, 24, 1) == -1
    assert lib.dmh_wino_wrw(one, one, 1, 64, 48, 8, 8, 1, one, one, None) != 0                # K not a multiple of 32
    assert lib.dmh_wino_conv3x3(one, one, None, 1, 16, 64, 8, 8, 1, one, None) != 0          # < 24 input channels
    assert b"multiple of 8" in lib.dmh_last_error()
    assert lib.dmh_wino_conv3x3(one, one, None, 1, 32, 64, 9, 8, 1, one, None) != 0          # odd output height
    assert lib.dmh_wino_conv3x3_act(one, one, None, None, 1, 1, 32, 64, 8, 8, 3, one, None) != 0   # pad
    assert lib.dmh_conv3x3_small(one, one, None, 1, 64, 64, 8, 8, 1, 0, one, None) != 0
    assert b"channel counts" in lib.dmh_last_error()
    assert lib.dmh_conv3x3_head(one, one, None, 1, 24, 8, 8, 1, one, None) != 0
    assert lib.dmh_conv7x7s2_bwd_data(one, one, 1, 64, 5, 8, 8, one, None) != 0               # > 4 image channels
    assert lib.dmh_conv7x7s2_bwd_data(one, one, 1, 64, 3, 7, 8, one, None) != 0               # odd height
    assert lib.dmh_bn_stats_partials_size(2, 8, 100) == 8 * 1 * 3
    assert lib.dmh_bn_stats_partials_size(32, 64, 160 * 512) == 64 * 64 * 3
    assert lib.dmh_bn_stats_partials_size(0, 8, 100) == -1
    assert lib.dmh_stem_bn_relu_pool_fwd(one, one, one, 1, 2, 3, 4, one, one, one, None) != 0
    assert b"even" in lib.dmh_last_error()

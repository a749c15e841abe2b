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
    src.write_text('#include "dmh_hip.h"\n#include <stdio.h>\n#include <stddef.h>\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(dmh_photo_args),sizeof(dmh_smooth_args),sizeof(dmh_paste_args),'
                   'offsetof(dmh_photo_args,seed),offsetof(dmh_paste_args,mode),sizeof(dmh_roi_glue_args),'
                   'offsetof(dmh_roi_glue_args,B),offsetof(dmh_roi_glue_args,elu));return 0;}\n')
    exe = tmp_path / "sz"
    assert os.system("gcc -I%s %s -o %s" % (os.path.join(REPO, "include"), src, exe)) == 0
    out = os.popen(str(exe)).read().split()
    assert [int(v) for v in out] == [ctypes.sizeof(_native.PhotoArgs), ctypes.sizeof(_native.SmoothArgs),
                                     ctypes.sizeof(_native.PasteArgs), _native.PhotoArgs.seed.offset,
                                     _native.PasteArgs.mode.offset, ctypes.sizeof(_native.RoiGlueArgs),
                                     _native.RoiGlueArgs.B.offset, _native.RoiGlueArgs.elu.offset]


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
    # the workspace forms check their shapes before anything else (a NULL workspace is legal: whole items)
    assert lib.dmh_wino_conv3x3_ws(one, one, None, 1, 16, 64, 8, 8, 1, one, None, 0, None) != 0
    assert lib.dmh_wino_conv3x3_ws(one, one, None, 1, 32, 64, 9, 8, 1, one, one, 1 << 23, None) != 0
    assert lib.dmh_wino_conv3x3_act_ws(one, one, None, None, 1, 1, 32, 64, 8, 8, 3, one, one, 1 << 23, None) != 0
    assert lib.dmh_wino32_conv3x3_ws(one, one, None, 1, 16, 32, 8, 8, 1, one, one, 1 << 23, None) != 0
    assert lib.dmh_avg_pyramid(one, 3, 321, 1024, one, one, one, None) != 0                 # sizes must be multiples of 8
    # K15's image form: sizes, and the 16-byte alignment the wide loaders need
    assert lib.dmh_down_conv_image_size(128, 64) == (128 // 32) * (64 // 8) * 2560
    assert lib.dmh_down_conv_image_size(48, 64) == -1 and lib.dmh_down_conv_image_size(64, 12) == -1
    assert lib.dmh_down_conv_weight_image(one, None, 48, 64, one, None) != 0
    assert lib.dmh_down_conv_fwd_img(one, one, 1, None, None, 0, 1, 64, 128, 8, 10, one, one, None) != 0      # W % 4
    assert b"multiple of 4" in lib.dmh_last_error()
    assert lib.dmh_down_conv_fwd_img(one, one, 1, None, None, 0, 1, 64, 128, 8, 8, one, None, None) != 0       # yd missing
    assert lib.dmh_down_conv_bwd_data_img(one, one, one, None, 1, 64, 128, 8, 12, one, None) != 0              # W % 8
    assert lib.dmh_down_conv_bwd_data_img(one, one, one, None, 1, 48, 128, 8, 16, one, None) != 0              # C_in % 64
    assert lib.dmh_gt_depth_mse_fwd(one, one, one, 0, one, 0, 64, 0.1, 100.0, one, one, None) != 0   # empty batch
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


def test_window_entry_points_reject_bad_shapes_without_gpu():
    """K19 (windowed glue / cost / encoder-head passes) and the stand-alone layers kernels: host-side argument checks and
    size helpers, no launch."""
    from depthmodelhardening_amd import _native as N
    lib = N.lib()
    one = ctypes.c_void_p(16)
    a = N.RoiGlueArgs()
    assert lib.dmh_roi_glue_fwd(ctypes.byref(a), one, None) != 0 and b"null pointer" in lib.dmh_last_error()
    a.y, a.dst_org = one, one
    a.B, a.C1, a.C2, a.sh, a.sw, a.hc, a.wc, a.H, a.W = 2, 8, 0, 10, 10, 6, 5, 16, 16      # odd window width
    assert lib.dmh_roi_glue_fwd(ctypes.byref(a), one, None) != 0 and b"even width" in lib.dmh_last_error()
    a.wc = 20                                                                            # window wider than its frame
    assert lib.dmh_roi_glue_bwd(ctypes.byref(a), one, one, None, None, 0, 0, None, 0, 0, None) != 0
    a.wc, a.y_org = 6, one                                                               # a region needs a whole-frame y
    assert lib.dmh_roi_glue_bwd(ctypes.byref(a), one, one, None, one, 4, 4, None, 0, 0, None) != 0
    assert b"whole-frame y" in lib.dmh_last_error()
    assert lib.dmh_roi_cost_partials_size(12, 174, 208) == 12 * 36
    assert lib.dmh_roi_cost_fwd(one, one, one, 2, 40, 40, 32, 64, one, one, one, None) != 0        # window taller than the frame
    assert lib.dmh_roi_crop(one, None, None, one, 2, 8, 16, 16, 6, 5, 0, one, None) != 0           # odd width
    assert lib.dmh_roi_crop(one, None, one, one, 2, 8, 16, 16, 6, 6, 0, one, None) != 0            # both src and g
    assert lib.dmh_roi_paste(one, one, 4, 4, one, 2, 8, 16, 16, 6, 6, one, None) != 0              # source smaller than the window
    assert lib.dmh_stem_bn_relu_pool_bwd_win(one, one, None, one, one, one, one, 2, 8, 16, 16, 6, 5, 4, 4, one, None) != 0
    assert lib.dmh_conv7x7s2_bwd_data_win(one, one, one, one, 2, 64, 3, 32, 64, 12, 15, 8, 10, one, None) != 0   # odd window
    assert lib.dmh_stem_conv_norm_fwd_win(one, one, one, 2, 32, 64, 20, 10, 0.45, 0.225, one, None) != 0        # taller than H/2
    assert lib.dmh_ssim_map(one, one, 6, 1, 8, one, None) != 0 and lib.dmh_edge_smooth_partials_size(2, 32, 96) == 2 * 2 * 3
    assert lib.dmh_edge_smooth(one, one, 2, 3, 1, 8, one, one, None) != 0


def test_strided_and_stem_weight_gradient_entry_points_without_gpu():
    """K20 / K21: workspace sizes (-1 for shapes the kernels do not take) and host-side argument checks, no launch."""
    from depthmodelhardening_amd import _native as N
    lib = N.lib()
    one = ctypes.c_void_p(16)
    # layer2.0 at batch 32: 2 channel-block pairs x 256 pixel slices x 10 taps x 64 x 64
    assert lib.dmh_down_wrw_workspace_size(32, 64, 128, 80, 256) == 2 * 256 * 10 * 64 * 64
    assert lib.dmh_down_wrw_workspace_size(3, 64, 64, 6, 24) == 9 * 10 * 64 * 64            # nine row tiles: nine slices
    for bad in ((32, 48, 128, 80, 256), (32, 64, 96, 80, 256), (32, 64, 128, 81, 256), (32, 64, 128, 80, 100), (0, 64, 64, 8, 8)):
        assert lib.dmh_down_wrw_workspace_size(*bad) == -1, bad
    assert lib.dmh_down_wrw(one, None, None, 2, 64, 64, 8, 16, one, one, None, None) != 0 and b"null pointer" in lib.dmh_last_error()
    assert lib.dmh_down_wrw(one, one, one, 2, 64, 64, 8, 16, one, one, None, None) != 0            # gd without dwd
    assert lib.dmh_down_wrw(one, one, None, 2, 64, 64, 8, 12, one, one, None, None) != 0 and b"multiple of 8" in lib.dmh_last_error()
    # stem at batch 32, 320 x 1024: 20,480 row tiles -> 512 workgroups x 4 waves x 10 blocks of 32 x 32
    assert lib.dmh_stem_wrw_workspace_size(32, 320, 1024) == 512 * 4 * 10 * 1024
    assert lib.dmh_stem_wrw_workspace_size(3, 38, 72) == 30 * 4 * 10 * 1024
    assert lib.dmh_stem_wrw_workspace_size(2, 37, 72) == -1 and lib.dmh_stem_wrw_workspace_size(2, 38, 76) == -1
    assert lib.dmh_stem_wrw(one, one, 2, 38, 76, 0.45, 0.225, one, one, None) != 0 and b"multiple of 8" in lib.dmh_last_error()
    assert lib.dmh_stem_wrw(one, one, 2, 38, 72, 0.45, 0.0, one, one, None) != 0 and b"std" in lib.dmh_last_error()

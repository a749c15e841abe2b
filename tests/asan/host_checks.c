/* Host side of libdmh_hip under AddressSanitizer (CPU build only: GPU ASan is not available on this pool).
 *
 * Linked against depthmodelhardening_amd/lib/libdmh_hip_asan.so -- every .hip source compiled with
 * `hipcc --offload-host-only -fsanitize=address`: the argument checks, the workspace-size arithmetic and the error
 * formatting of every entry point, no device code.  Each call below must be refused on the host BEFORE any launch
 * (DMH_EINVAL, message in dmh_last_error()) or return a size; ASan aborts the process on any out-of-bounds access
 * of the argument structs (e.g. num_scales / num_frames beyond the arrays), of the error buffer, or of the static
 * tables the checks index.  Compiled as C: the header must be a plain C header.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dmh_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                            \
    do {                                                                        \
        if (!(cond)) {                                                          \
            fprintf(stderr, "host_checks: %s (line %d): %s\n", #cond, __LINE__, dmh_last_error()); \
            ++failures;                                                         \
        }                                                                       \
    } while (0)

int main(void) {
    float* one = (float*)(uintptr_t)64; /* a non-NULL dummy "device pointer": every call must fail before a launch */
    void* ptrs[DMH_MAX_SCALES] = {0, 0, 0, 0};
    EXPECT(strstr(dmh_version(), "gfx950") != NULL);

    /* K1: struct with out-of-range counts must be rejected, not indexed */
    dmh_photo_args p;
    memset(&p, 0, sizeof p);
    EXPECT(dmh_photo_loss_fwd(&p, NULL, (float* const*)ptrs, NULL, NULL) == DMH_EINVAL);
    p.target = one; p.K = one; p.inv_K = one; p.B = 2; p.H = 64; p.W = 192; p.min_depth = 0.1f; p.max_depth = 100.f;
    for (int bad = -1; bad <= 9; ++bad) {
        if (bad >= 1 && bad <= DMH_MAX_SCALES) continue;
        p.num_scales = bad; p.num_frames = 1;
        EXPECT(dmh_photo_loss_fwd(&p, (uint8_t*)one, (float* const*)ptrs, one, NULL) == DMH_EINVAL);
        EXPECT(dmh_photo_stage_size(&p) <= 0 || bad >= 1);
        p.num_scales = 1; p.num_frames = bad;
        EXPECT(dmh_photo_loss_fwd(&p, (uint8_t*)one, (float* const*)ptrs, one, NULL) == DMH_EINVAL);
        EXPECT(dmh_photo_loss_bwd(&p, (const uint8_t*)one, one, one, one, (float* const*)ptrs, NULL) == DMH_EINVAL);
        EXPECT(dmh_photo_pose_partials_size(&p) <= 0 || bad >= 1);
    }
    for (int s = 1; s <= 4; ++s)
        EXPECT(dmh_photo_partials_size(32, 320, 1024, s) > 0);
    EXPECT(dmh_photo_partials_size(0, 320, 1024, 4) <= 0 || 1);   /* any value, but no crash */
    EXPECT(dmh_unpack_selection(NULL, 10, 0, NULL, NULL) == DMH_EINVAL);
    EXPECT(dmh_unpack_selection((const uint8_t*)one, 10, 7, one, NULL) == DMH_EINVAL);

    /* K2 */
    dmh_smooth_args sm;
    memset(&sm, 0, sizeof sm);
    for (int bad = -1; bad <= 9; ++bad) {
        sm.num_scales = bad; sm.B = 2;
        (void)dmh_smooth_partials_size(&sm);
        if (bad < 1 || bad > DMH_MAX_SCALES) EXPECT(dmh_smooth_loss_fwd(&sm, one, NULL) == DMH_EINVAL);
    }

    /* K3 */
    dmh_paste_args pa;
    memset(&pa, 0, sizeof pa);
    EXPECT(dmh_eot_paste_fwd(&pa, NULL, NULL, NULL) == DMH_EINVAL);
    pa.scene = one; pa.patch = one; pa.pmask = one; pa.coeffs = one;
    pa.N = -3; pa.SH = 375; pa.SW = 1242; pa.PH = 260; pa.PW = 300; pa.OH = 320; pa.OW = 1024;
    EXPECT(dmh_eot_paste_fwd(&pa, one, one, NULL) == DMH_EINVAL);
    pa.N = 2; pa.mode = 99;
    EXPECT(dmh_eot_paste_fwd(&pa, one, one, NULL) == DMH_EINVAL);
    EXPECT(dmh_eot_paste_bwd(&pa, one, one, NULL) == DMH_EINVAL);

    /* K4 - K6, K6b */
    EXPECT(dmh_pgd_linf_step(NULL, NULL, NULL, 0.1f, 0.1f, NULL, 0, NULL) == DMH_EINVAL);
    EXPECT(dmh_l0_compose_fwd(one, one, one, 0, 10, 0.1f, 0, one, NULL, NULL) == DMH_EINVAL);
    EXPECT(dmh_l0_mask_partials_size(78000) > 0);
    EXPECT(dmh_sq_mean_partials_size(1) == 1 && dmh_sq_mean_partials_size((int64_t)1 << 40) > 0);
    EXPECT(dmh_masked_sq_mean_fwd(one, NULL, 0, one, one, NULL) == DMH_EINVAL);
    EXPECT(dmh_gt_depth_mse_fwd(one, one, one, 10, one, 2, 100, 0.1f, 100.f, one, one, NULL) == DMH_EINVAL);  /* stride < HW */
    EXPECT(dmh_gt_depth_mse_bwd(one, one, one, 100, one, 2, 100, 100.f, 0.1f, one, one, NULL) == DMH_EINVAL); /* depths swapped */

    /* convolution kernels: size helpers over odd shapes, refusals */
    for (int c = -8; c <= 520; c += 7)
        for (int k = -8; k <= 520; k += 61) {
            (void)dmh_wino_weight_size(k, c);
            (void)dmh_wino32_weight_size(k, c);
            (void)dmh_wino_wrw_workspace_size(2, c, k, 16, 32, 1);
            (void)dmh_down_wrw_workspace_size(2, c, k, 16, 32);
        }
    EXPECT(dmh_wino_conv3x3(one, one, NULL, 1, 16, 64, 8, 8, 1, one, NULL) != DMH_OK);
    EXPECT(dmh_wino_conv3x3(one, one, NULL, 1, 32, 64, 9, 8, 1, one, NULL) != DMH_OK);
    EXPECT(dmh_wino_conv3x3_act(one, one, NULL, NULL, 1, 1, 32, 64, 8, 8, 3, one, NULL) != DMH_OK);
    EXPECT(dmh_wino32_conv3x3(one, one, NULL, 1, 16, 32, 8, 8, 1, one, NULL) != DMH_OK);
    EXPECT(dmh_conv3x3_small(one, one, NULL, 1, 64, 64, 8, 8, 1, 0, one, NULL) != DMH_OK);
    EXPECT(dmh_conv3x3_head(one, one, NULL, 1, 24, 8, 8, 1, one, NULL) != DMH_OK);
    EXPECT(dmh_conv7x7s2_bwd_data(one, one, 1, 64, 5, 8, 8, one, NULL) != DMH_OK);
    EXPECT(dmh_bn_stats_partials_size(0, 8, 100) == -1 && dmh_bn_stats_partials_size(32, 64, 160 * 512) > 0);
    EXPECT(dmh_stem_wrw_workspace_size(2, 37, 72) == -1 && dmh_stem_wrw_workspace_size(32, 320, 1024) > 0);
    EXPECT(dmh_stem_wrw(one, one, 2, 38, 72, 0.45f, 0.0f, one, one, NULL) != DMH_OK);

    /* K19 windows */
    dmh_roi_glue_args g;
    memset(&g, 0, sizeof g);
    EXPECT(dmh_roi_glue_fwd(&g, one, NULL) != DMH_OK);
    g.y = one; g.dst_org = (const int32_t*)one;
    g.B = 2; g.C1 = 8; g.C2 = 0; g.sh = 10; g.sw = 10; g.hc = 6; g.wc = 5; g.H = 16; g.W = 16;
    EXPECT(dmh_roi_glue_fwd(&g, one, NULL) != DMH_OK);
    EXPECT(dmh_roi_cost_partials_size(12, 174, 208) > 0);
    EXPECT(dmh_roi_cost_fwd(one, one, (const int32_t*)one, 2, 40, 40, 32, 64, one, one, one, NULL) != DMH_OK);

    /* the error buffer: a long message must be truncated, not overflow (the formatted text carries the entry point's name) */
    EXPECT(strlen(dmh_last_error()) < 512);
    if (failures) {
        fprintf(stderr, "host_checks: %d expectation(s) failed\n", failures);
        return 1;
    }
    printf("host_checks: ok\n");
    return 0;
}

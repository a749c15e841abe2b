/*
 * dmh_hip.h -- C ABI of libdmh_hip.so: the MI355X (gfx950) hot path of the
 * DepthModelHardening adversarial-training loop.
 *
 * The reference (Bob-cheng/DepthModelHardening) is pure Python; its seam for this path
 * is a set of Python call signatures.  Every entry point below
 * replaces one PyTorch op chain of the reference, cited as file:line relative to the
 * reference root (MD2 = DepthNetworks/monodepth2, DH = DepthNetworks/depth-hints).
 * The binding a maintainer would add on the reference side is a ctypes stub -- see
 * INTEGRATION.md.
 *
 * Conventions
 *  - plain C: pointers are DEVICE pointers to fp32, NCHW, contiguous; sizes are ints.
 *    No torch types.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - the library allocates nothing, retains nothing, never synchronises; every call only
 *    enqueues kernels on `stream` (safe under hipGraph capture).
 *  - return value: DMH_OK or an error code; dmh_last_error() gives the message of the
 *    calling thread's last failure.  Shape errors are caught on the host BEFORE launch.
 */
#ifndef DMH_HIP_H
#define DMH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMH_OK 0
#define DMH_EINVAL 1  /* bad argument (null pointer, unsupported shape) */
#define DMH_ELAUNCH 2 /* hipLaunch / runtime error                      */

#define DMH_MAX_SCALES 4
#define DMH_MAX_FRAMES 4

#define DMH_VARIANT_MD2 0 /* MD2/trainer.py:647-660  min over {identity, reprojection}, mean over all pixels */
#define DMH_VARIANT_DH 1  /* DH/trainer.py:557-590,700-708  argmin mask, masked sum / mask count            */

#define DMH_NOISE_NONE 0   /* no tie-break term                                                     */
#define DMH_NOISE_TENSOR 1 /* caller passes the already-scaled term (randn*1e-5, MD2/trainer.py:642-645) */
#define DMH_NOISE_PHILOX 2 /* N(0,1)*1e-5 generated in-kernel from (seed, offset): no HBM traffic    */

/* Slots of the `fin` vector written by dmh_loss_finalize (floats). */
#define DMH_FIN_LOSS 0       /* sum_s loss_s / NS                         MD2/trainer.py:670 */
#define DMH_FIN_LOSS_S 1     /* [4] loss/{s}                              MD2/trainer.py:668 */
#define DMH_FIN_REPROJ_S 5   /* [4] mean(to_optimise) | DH reproj_loss/{s}                   */
#define DMH_FIN_COUNT_S 9    /* [4] number of pixels where reprojection was selected         */
#define DMH_FIN_SMOOTH_S 13  /* [4] get_smooth_loss(norm_disp, color)     MD2/trainer.py:664 */
#define DMH_FIN_HINT_S 20    /* [4] depth_hint_loss/{s} (0 without depth hints)   DH/trainer.py:713-725 */
#define DMH_FIN_HINTCOUNT_S 24 /* [4] number of pixels where the depth hint won the argmin */
#define DMH_FIN_SIZE 28

const char* dmh_version(void);
const char* dmh_last_error(void);
/* Diagnostics (tools/rccl_overlap.py): a stand-in with a ring collective's launch geometry -- `channels` persistent workgroups of
 * 256 threads copy n floats src -> dst `rounds` times -- to observe, on ONE GPU, how such a kernel on a side stream is scheduled
 * against the persistent convolution workgroups (a 1-rank RCCL all-reduce launches no device kernel).  Not on any product path. */
int dmh_debug_channel_copy(const float* src, float* dst, int64_t n, int channels, int rounds, void* stream);

/* ------------------------------------------------------------------------------------
 * K1  fused photometric loss: bilinear-upsample(disp_s) -> disp_to_depth -> BackprojectDepth
 *     -> Project3D -> grid_sample(border, align_corners=True) -> SSIM(3x3, reflect) + L1
 *     -> identity term -> per-pixel min / argmin mask -> block partial sums,
 *     for all scales in ONE launch (target/source tiles are read once).
 * Replaces: MD2/trainer.py:472-523 (generate_images_pred), :525-537
 *           (compute_reprojection_loss), :589-660 (compute_losses body),
 *           MD2/layers.py:16-25,139-198,223-253; DH/trainer.py:638-708.
 * ---------------------------------------------------------------------------------- */
typedef struct dmh_photo_args {
    const float* target;                 /* [B,3,H,W] inputs[("color",0,0)]                        */
    const float* source[DMH_MAX_FRAMES]; /* [B,3,H,W] inputs[("color",f,0)], f = frame_ids[1:]     */
    const float* T[DMH_MAX_FRAMES];      /* [B,4,4]   stereo_T or cam_T_cam                        */
    const float* K;                      /* [B,4,4]   inputs[("K",0)]                              */
    const float* inv_K;                  /* [B,4,4]   inputs[("inv_K",0)]                          */
    const float* disp[DMH_MAX_SCALES];   /* [B,1,Hs,Ws] outputs[("disp",s)]                        */
    int Hs[DMH_MAX_SCALES], Ws[DMH_MAX_SCALES];
    int B, H, W;
    int num_frames;                      /* 1..DMH_MAX_FRAMES                                      */
    int num_scales;                      /* 1..DMH_MAX_SCALES                                      */
    float min_depth, max_depth;          /* MD2/options.py:69-76                                   */
    int variant;                         /* DMH_VARIANT_*                                          */
    int automask;                        /* 0 = --disable_automasking                              */
    int no_ssim;                         /* 1 = --no_ssim                                          */
    int noise_mode;                      /* DMH_NOISE_*                                            */
    const float* noise[DMH_MAX_SCALES];  /* TENSOR mode: [B,NF,H,W], NF = num_frames (MD2) or 1 (DH) */
    uint64_t seed, offset;               /* PHILOX mode                                            */
    /* DepthHints --use_depth_hints (DH/trainer.py:510-525,629-636,700-725); both NULL = off.  Needs variant DH,
     * automask and ONE source frame (the reference builds the hint view for the stereo frame only).              */
    const float* depth_hint;             /* [B,1,H,W] inputs["depth_hint"], 0 where there is no hint */
    const float* depth_hint_mask;        /* [B,1,H,W] inputs["depth_hint_mask"]                      */
} dmh_photo_args;

/* number of floats the photometric partial-sum workspace needs */
int64_t dmh_photo_partials_size(int B, int H, int W, int num_scales);   /* 4 floats per (scale, strip) */
/* number of floats of the backward staging workspace (per-strip partial low-resolution gradients) */
int64_t dmh_photo_stage_size(const dmh_photo_args* a);

/* Forward.  sel      : out [B,H,W] uint8, the selection of every scale packed 2 bits per scale: bits [2s, 2s+1] =
 *                      0 identity chosen, 1+f reprojection of source frame f chosen (== outputs["identity_selection/s"]
 *                      for one source frame, MD2/trainer.py:656-658); with depth hints 3 = the hint won (reprojection
 *                      of frame 0 applies as well, DH/trainer.py:583-584).  At most 3 source frames.
 *           to_opt[s]: out [B,H,W] per-pixel selected loss, or NULL
 *           partials : out, dmh_photo_partials_size floats
 * disp[s] must be [B,1,H/f,W/f] with f in {1,2,4,8,16}.                                                     */
int dmh_photo_loss_fwd(const dmh_photo_args* a, uint8_t* sel, float* const to_opt[DMH_MAX_SCALES], float* partials,
                       void* stream);

/* Backward (recompute).  gvec: [DMH_FIN_SIZE] upstream gradient of the fin vector; fin: the finalized forward
 * vector; stage: dmh_photo_stage_size floats of scratch.  g_disp[s]: out [B,1,Hs,Ws] gradient w.r.t. disp[s]
 * (overwritten; the adjoint of the bilinear up-sampling, MD2/trainer.py:481-482, is applied in the kernel).   */
int dmh_photo_loss_bwd(const dmh_photo_args* a, const uint8_t* sel, const float* gvec, const float* fin, float* stage,
                       float* const g_disp[DMH_MAX_SCALES], void* stream);
/* The same + the gradient w.r.t. the camera poses (outputs[("cam_T_cam", 0, f)] of monocular source frames,
 * MD2/trainer.py:276-330,487-519): pose_partials[num_scales][strips][num_frames][12] receives per-strip sums
 *   [ S_i0, S_i1, S_i2, s_i ]  (i = 0, 1, 2)   with   d loss / d P_f[i][j<3] = sum_k inv_K[j][k] S_ik,   d loss / d P_f[i][3] = s_i,
 * P_f = (K T_f)[:3,:] (MD2/layers.py:188); the caller sums over scales and an image's strips (strip index = image-major:
 * strips / B per image) and forms d loss / d T_f = K[:3,:]^T dP_f.  dmh_photo_pose_partials_size(a) floats. */
int64_t dmh_photo_pose_partials_size(const dmh_photo_args* a);
int dmh_photo_loss_bwd_pose(const dmh_photo_args* a, const uint8_t* sel, const float* gvec, const float* fin, float* stage,
                            float* const g_disp[DMH_MAX_SCALES], float* pose_partials, void* stream);

/* out[i] = field `scale` of the packed selection map as float (0 identity, 1+f frame f), n = B*H*W.          */
int dmh_unpack_selection(const uint8_t* sel, int64_t n, int scale, float* out, void* stream);

/* Adjoint of F.interpolate(disp,[H,W],bilinear,align_corners=False) (MD2/trainer.py:481-482):
 * g_disp [B,1,Hs,Ws] (+)= gather(g_up [B,H,W]).  accumulate != 0 adds to g_disp.             */
int dmh_upsample_bilinear_adjoint(const float* g_up, float* g_disp, int B, int H, int W, int Hs, int Ws,
                                  int accumulate, void* stream);

/* Materialise generate_images_pred's tensors for one (scale, frame) (MD2/trainer.py:481-519):
 * depth [B,1,H,W], sample [B,H,W,2], color [B,3,H,W]; any output may be NULL.                */
int dmh_warp_view_fwd(const float* source, const float* disp, const float* K, const float* inv_K,
                      const float* T, int B, int H, int W, int Hs, int Ws, float min_depth,
                      float max_depth, float* depth, float* sample, float* color, void* stream);

/* Backward of dmh_warp_view_fwd w.r.t. the upsampled disparity given grad_color [B,3,H,W]
 * (and optionally grad_depth [B,1,H,W], may be NULL): g_up [B,H,W].                           */
int dmh_warp_view_bwd(const float* source, const float* disp, const float* K, const float* inv_K,
                      const float* T, int B, int H, int W, int Hs, int Ws, float min_depth,
                      float max_depth, const float* grad_color, const float* grad_depth, float* g_up,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * K2  edge-aware smoothness on the mean-normalised disparity, all scales in one launch.
 * Replaces: MD2/trainer.py:662-666 + MD2/layers.py:207-220.
 * ---------------------------------------------------------------------------------- */
typedef struct dmh_smooth_args {
    const float* disp[DMH_MAX_SCALES];  /* [B,1,Hs,Ws] outputs[("disp",s)]   */
    const float* color[DMH_MAX_SCALES]; /* [B,3,Hs,Ws] inputs[("color",0,s)] */
    int Hs[DMH_MAX_SCALES], Ws[DMH_MAX_SCALES];
    int B, num_scales;
} dmh_smooth_args;

int64_t dmh_smooth_partials_size(const dmh_smooth_args* a);
int dmh_smooth_loss_fwd(const dmh_smooth_args* a, float* partials, void* stream);
/* g_disp[s] [B,1,Hs,Ws]; accumulate != 0 adds.  sstats from dmh_loss_finalize. */
int dmh_smooth_loss_bwd(const dmh_smooth_args* a, const float* gvec, const float* sstats,
                        float smooth_wt, float* const g_disp[DMH_MAX_SCALES], int accumulate, void* stream);

/* Deterministic final reduction of K1+K2 partials:  fin[DMH_FIN_SIZE], sstats[NS][B][2] =
 * (mean_disp_b, R_b).  loss_s = reproj_s + smooth_wt * smooth_s / 2^s  (MD2/trainer.py:660-668). */
int dmh_loss_finalize(const float* photo_partials, const float* smooth_partials, int B, int H, int W,
                      const dmh_smooth_args* sm, int variant, float smooth_wt, float* fin, float* sstats,
                      void* stream);

/* ------------------------------------------------------------------------------------
 * K3  EOT paste: Pad -> perspective warp of patch and mask -> composite -> Resize, fused.
 * Replaces: physicalTrans.py:107-166 (padding_img, project) +
 *           torchattacks/attacks/phy_obj_atk.py:87-90 (composite + 2x Resize).
 * coeffs [N,8]: torchvision-0.8.2 perspective coefficients per sample (host computes them from
 * the integer pixel quads, physicalTrans.py:62-81).  scene_bstride = 0 broadcasts one scene.
 * ---------------------------------------------------------------------------------- */
#define DMH_PASTE_COMPOSITE 0 /* adv = resize(scene*(1-m) + obj*m), mask_out = resize(m)   phy_obj_atk.py:88-90 */
#define DMH_PASTE_WARP_ONLY 1 /* adv = resize(obj), mask_out = resize(m); scene ignored    physicalTrans.py:156-165 */

typedef struct dmh_paste_args {
    const float* scene; /* [N or 1,3,SH,SW] */
    int64_t scene_bstride;
    const float* patch; /* [1,3,PH,PW] */
    const float* pmask; /* [1,1,PH,PW] */
    const float* coeffs; /* [N,8] */
    int N, SH, SW, PH, PW, OH, OW;
    int l_pad, t_pad; /* physicalTrans.py:110-113 */
    int mode;         /* DMH_PASTE_COMPOSITE or DMH_PASTE_WARP_ONLY */
    const int32_t* flip; /* [N] or NULL: != 0 mirrors sample n horizontally (do_flip of MD2/datasets/mono_dataset.py:288,
                          :222-225: the loader flips the frame, prep_adv_data flips the projected object and mask; the
                          scene passed here is the UN-flipped frame and the whole composite is written mirrored) */
    const int32_t* scene_index; /* [N] or NULL: sample n reads frame scene_index[n] of `scene` (a pool of frames with batch
                                   stride scene_bstride) instead of frame n -- the loader's index_select / side pick without
                                   a copy of the frames.  Every index must lie inside the pool (checked by the caller). */
} dmh_paste_args;

/* adv [N,3,OH,OW], mask_out [N,1,OH,OW] (either may be NULL) */
int dmh_eot_paste_fwd(const dmh_paste_args* a, float* adv, float* mask_out, void* stream);
/* g_patch [1,3,PH,PW] is overwritten (gather per patch texel over the inverse homography: no atomics, bitwise
 * reproducible; the caller does not have to zero it). */
int dmh_eot_paste_bwd(const dmh_paste_args* a, const float* g_adv, float* g_patch, void* stream);

/* ------------------------------------------------------------------------------------
 * K4  PGD-L_inf update  x <- clamp(x0 + clamp(x + alpha*sign(g) - x0, -eps, eps), 0, 1)
 * Replaces: phy_obj_atk.py:98-101, pgd_depth.py:76-78.  out may alias x.
 * ---------------------------------------------------------------------------------- */
int dmh_pgd_linf_step(const float* x, const float* x0, const float* g, float alpha, float eps, float* out,
                      int64_t n, void* stream);

/* ------------------------------------------------------------------------------------
 * K5  L0 attack pieces (phy_obj_atk_l0.py).
 * compose (:94-99,:43-52): adv = clamp(obj + clamp(pos,0,1) - clamp(neg,0,1), 0, 1);
 *   l0_count (int32, zeroed by caller) += #pixels whose thresholded pattern is non-zero.
 *   finalize != 0 applies the final thresholding of :143-150 to the composed patch too.
 * ---------------------------------------------------------------------------------- */
int dmh_l0_compose_fwd(const float* obj, const float* pos, const float* neg, int C, int HW, float l0_clip,
                       int finalize, float* adv, int32_t* l0_count, void* stream);
int dmh_l0_compose_bwd(const float* obj, const float* pos, const float* neg, const float* g_adv, int C, int HW,
                       float* g_pos, float* g_neg, int accumulate, void* stream);
/* mask cost (:130-132): mean_hw max_c(tanh(p/10)/(2-1e-7)+0.5) for pos and neg; cost: float[1] out.
 * partials: 2*nblk floats (dmh_l0_mask_partials_size). */
int64_t dmh_l0_mask_partials_size(int HW);
int dmh_l0_mask_cost_fwd(const float* pos, const float* neg, int C, int HW, float* partials, float* cost,
                         void* stream);
/* g_pos/g_neg (+)= gscale[0]*weight[0] * d cost/d p ; gscale, weight: device scalars */
int dmh_l0_mask_cost_bwd(const float* pos, const float* neg, int C, int HW, const float* gscale,
                         const float* weight, float* g_pos, float* g_neg, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * K6  masked squared mean  cost = mean((disp*mask)^2)  (MSELoss against zeros,
 *     phy_obj_atk.py:94, phy_obj_atk_l0.py:127).  mask may be NULL (pgd_depth.py:68).
 * ---------------------------------------------------------------------------------- */
int64_t dmh_sq_mean_partials_size(int64_t n);
int dmh_masked_sq_mean_fwd(const float* disp, const float* mask, int64_t n, float* partials, float* cost,
                           void* stream);
/* g_disp = gscale[0] * 2 * disp * mask^2 / n */
int dmh_masked_sq_mean_bwd(const float* disp, const float* mask, int64_t n, const float* gscale, float* g_disp,
                           void* stream);

/* ------------------------------------------------------------------------------------
 * K6b --gt_depth supervised term (MD2/trainer.py:551-557, options.py:227-229): with
 *     depth(d) = clamp(5.4 / (1/max_depth + (1/min_depth - 1/max_depth) d), 1e-3, 80)   (disp_to_depth, layers.py:16-25)
 *     cost = mean_{b,hw} ( m objdepth[b] + depth(disp_gt) (1 - m) - depth(disp) )^2 ,   m = objmask[b * mask_bstride + hw]
 *     disp, disp_gt: [B,1,H,W]; objmask: channel 0 of inputs[("color_objmask",0,0)] ([B,3,H,W]: mask_bstride = 3 H W);
 *     objdepth: [B] metres.  partials: dmh_sq_mean_partials_size(B*HW) floats; cost: float[1].
 *     bwd: g_disp = gscale[0] * d cost / d disp (0 where depth(disp) sits on a clamp bound, as torch.clamp does).
 * ---------------------------------------------------------------------------------- */
int dmh_gt_depth_mse_fwd(const float* disp, const float* disp_gt, const float* objmask, int64_t mask_bstride,
                         const float* objdepth, int B, int64_t HW, float min_depth, float max_depth, float* partials,
                         float* cost, void* stream);
int dmh_gt_depth_mse_bwd(const float* disp, const float* disp_gt, const float* objmask, int64_t mask_bstride,
                         const float* objdepth, int B, int64_t HW, float min_depth, float max_depth, const float* gscale,
                         float* g_disp, void* stream);

/* ------------------------------------------------------------------------------------
 * K20 weight gradients of the pair of convolutions that opens a down-sampling ResNet block (K15's train-pass counterpart):
 *     dw3[K][C][3][3] of nn.Conv2d(C, K, 3, stride 2, padding 1) and dwd[K][C][1][1] of nn.Conv2d(C, K, 1, stride 2) on the
 *     same x[B,C,H,W], from g3 / gd [B,K,H/2,W/2] (gd and dwd may be NULL: the 3x3 filter only).  Pixel axis on the fp32 MFMA,
 *     one partial block set per workgroup, fixed-order reduction: no atomics, bitwise reproducible (MIOpen's kernels for these
 *     shapes accumulate with float atomics).  C and K multiples of 64, H even, W a multiple of 8.
 *     workspace: dmh_down_wrw_workspace_size(B, C, K, H, W) floats (-1: shape not supported).
 * K21 weight gradient of the encoder's first convolution on the normalised image, nn.Conv2d(3, 64, 7, stride 2, padding 3)
 *     applied to (x - mean) / std (MD2/networks/resnet_encoder.py:89-90): dw[64][3][7][7] from x[B,3,H,W] and g[B,64,H/2,W/2];
 *     the normalisation is applied while the image tile is staged (zero in the padding), as in K14.  Same reduction scheme.
 *     H even, W a multiple of 8.  workspace: dmh_stem_wrw_workspace_size(B, H, W) floats.
 * ---------------------------------------------------------------------------------- */
int64_t dmh_down_wrw_workspace_size(int B, int C, int K, int H, int W);
int dmh_down_wrw(const float* x, const float* g3, const float* gd, int B, int C, int K, int H, int W, float* workspace, float* dw3,
                 float* dwd, void* stream);
int64_t dmh_stem_wrw_workspace_size(int B, int H, int W);
int dmh_stem_wrw(const float* x, const float* g, int B, int H, int W, float mean, float std, float* workspace, float* dw,
                 void* stream);

/* ------------------------------------------------------------------------------------
 * Stand-alone forms of the two loss layers of MD2/layers.py (Trainer.compute_losses runs the fused K1 + K2 instead; these
 * serve callers of the reference surface `SSIM()(x, y)` / `get_smooth_loss(disp, img)`):
 *   ssim_map   : out[planes,H,W] = clamp((1 - SSIM(x, y)) / 2, 0, 1), 3x3 means over the ReflectionPad2d(1) planes
 *                (MD2/layers.py:223-253).  bwd: g_x / g_y (either may be NULL); workspace: 5 * planes * H * W floats.
 *   edge_smooth: out = mean(|d_x disp| exp(-mean_c |d_x img|)) + mean(|d_y disp| exp(-mean_c |d_y img|)) for disp[B,1,H,W],
 *                img[B,C,H,W] (MD2/layers.py:207-220).  bwd: g_disp = gscale[0] * d out / d disp.
 * ---------------------------------------------------------------------------------- */
int dmh_ssim_map(const float* x, const float* y, int planes, int H, int W, float* out, void* stream);
int dmh_ssim_map_bwd(const float* x, const float* y, const float* g_out, int planes, int H, int W, float* workspace, float* g_x,
                     float* g_y, void* stream);
/* colour pyramid of the synthesised frame (inputs[("color", f, s)], s = 1..3; the reference's loader resizes per sample on the
 * CPU, MD2/datasets/mono_dataset.py:119-144): out_s[planes, H/2^s, W/2^s] = F.avg_pool2d(x, 2^s) for s = 1, 2, 3 in ONE pass
 * over x[planes, H, W], bit-identical to three ATen avg_pool2d calls.  H, W multiples of 8. */
int dmh_avg_pyramid(const float* x, int planes, int H, int W, float* out1, float* out2, float* out3, void* stream);
int64_t dmh_edge_smooth_partials_size(int B, int H, int W);
int dmh_edge_smooth(const float* disp, const float* img, int B, int C, int H, int W, float* partials, float* out, void* stream);
int dmh_edge_smooth_bwd(const float* disp, const float* img, int B, int C, int H, int W, const float* gscale, float* g_disp,
                        void* stream);

/* ------------------------------------------------------------------------------------
 * K8  attack-evaluation metrics of Trainer.val -> evaluate_attacks (MD2/evaluate_depth.py:57-99,193-197):
 *     depth = clamp(5.4 / (min_disp + (max_disp-min_disp)*|disp|), 1e-3, 80) for both disparities, then the
 *     (mask-weighted) abs_err, abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3 -> out8.  mask may be NULL.
 * ---------------------------------------------------------------------------------- */
int64_t dmh_depth_errors_partials_size(int64_t n);
int dmh_masked_depth_errors(const float* disp_gt, const float* disp_pred, const float* mask, int64_t n,
                            float min_depth, float max_depth, float scale, float clamp_lo, float clamp_hi,
                            float* partials, float* out8, void* stream);

/* ------------------------------------------------------------------------------------
 * Decoder glue (SURVEY.md section 8f direction "fuse around the convolutions"): the element-wise passes between
 * the MIOpen convolutions of the depth decoder, one pass per stage boundary.
 * Replaces MD2/networks/depth_decoder.py:54-60 (ELU, nearest upsample, torch.cat) + the ReflectionPad2d(1) of the
 * following Conv3x3 (MD2/layers.py:133-136).
 *   up_cat_pad: out[B,C1+C2,2h+2,2w+2] = pad1_reflect(cat(up2_nearest(ELU(y[B,C1,h,w])), skip[B,C2,2h,2w]))
 *   elu_pad   : out[B,C,H+2,W+2]       = pad1_reflect(ELU(z[B,C,H,W]))      (apply_elu = 0: pad only)
 * skip / g_skip may be NULL when C2 == 0.
 * ---------------------------------------------------------------------------------- */
int dmh_dec_up_cat_pad_fwd(const float* y, const float* skip, int B, int C1, int C2, int h, int w, float* out,
                           void* stream);
int dmh_dec_up_cat_pad_bwd(const float* y, const float* g_out, int B, int C1, int C2, int h, int w, float* g_y,
                           float* g_skip, void* stream);
int dmh_elu_pad_fwd(const float* z, int B, int C, int H, int W, int apply_elu, float* out, void* stream);
int dmh_elu_pad_bwd(const float* z, const float* g_out, int B, int C, int H, int W, int apply_elu, float* g_z,
                    void* stream);

/* ------------------------------------------------------------------------------------
 * K19 windowed decoder glue + windowed attack cost: the attack evaluated only where its loss lives.
 *     cost = -mean((disp * mask)^2) (torchattacks/attacks/phy_obj_atk.py:88-97, phy_obj_atk_l0.py:118-134) reads the
 *     disparity under the pasted object only, so inside an attack the high-resolution tail of the decoder
 *     (MD2/networks/depth_decoder.py:51-63, upconv(1,0) ... dispconv(0)) runs -- exactly -- on one window per sample
 *     around the object's bounding box: the convolution kernels take compact [B,C,hc+2,wc+2] windows with padding 0,
 *     and this is the pass between them:
 *       out[b, :, i, j] = pad1_reflect(cat(up2_nearest(ELU(y)), skip))  at frame position dst_org[b] + (i, j) - 1
 *     for 0 <= i < hc + 2, 0 <= j < wc + 2: decoder_glue's up_cat_pad / elu_pad restricted to a window of the H x W
 *     destination frame (the reflection is the FRAME's).  y [B,C1,sh,sw] and skip [B,C2,kh,kw] are windows of their own
 *     frames with per-sample origins y_org / skip_org ([B,2] int32: row, column), or whole frames (origin table NULL).
 *     up: y lives at half the destination resolution; elu: ELU is applied to y.  wc must be even.
 *     Every read must fall inside the source windows (the caller's window plan guarantees it; indices are clamped).
 *   bwd: g_y / g_skip over the whole source planes (0 where no window entry reads the element) or, for whole-frame
 *     sources, over a per-sample rectangle (below); g_skip may be NULL.
 *   roi_cost: cost = sum over the windows of (sigmoid(d_pre) * mask)^2 / (B H W); d_pre [B,1,hd,wd] is the disparity
 *     head's output on the window at org[b] of the H x W frame, mask [B,1,H,W] the full-frame K3 mask; sig receives the
 *     sigmoid.  bwd: g_pre = gscale[0] * 2 sig mask^2 / (B H W) * sig (1 - sig).
 * ---------------------------------------------------------------------------------- */
typedef struct dmh_roi_glue_args {
    const float* y;
    const float* skip;      /* NULL when C2 == 0 */
    const int* y_org;       /* [B,2] or NULL (y is the whole frame) */
    const int* skip_org;    /* [B,2] or NULL */
    const int* dst_org;     /* [B,2] window origin in the destination frame (unpadded coordinates) */
    int B, C1, C2;
    int sh, sw;             /* plane size of y */
    int kh, kw;             /* plane size of skip */
    int hc, wc;             /* destination window (unpadded); out is [B, C1 + C2, hc + 2, wc + 2] */
    int H, W;               /* destination frame (unpadded) */
    int up, elu;
} dmh_roi_glue_args;
int dmh_roi_glue_fwd(const dmh_roi_glue_args* a, float* out, void* stream);
/* y_reg_org / skip_reg_org ([B,2] int32 or NULL): for a whole-frame source, write only the y_reg_h x y_reg_w rectangle at
 * that per-sample origin (it must hold everything the window reaches; the caller owns the rest of the plane). */
int dmh_roi_glue_bwd(const dmh_roi_glue_args* a, const float* g_out, float* g_y, float* g_skip, const int* y_reg_org,
                     int y_reg_h, int y_reg_w, const int* skip_reg_org, int skip_reg_h, int skip_reg_w, void* stream);
int64_t dmh_roi_cost_partials_size(int B, int hd, int wd);
int dmh_roi_cost_fwd(const float* d_pre, const float* mask, const int* org, int B, int hd, int wd, int H, int W, float* sig,
                     float* partials, float* cost, void* stream);
int dmh_roi_cost_bwd(const float* sig, const float* mask, const int* org, int B, int hd, int wd, int H, int W,
                     const float* gscale, float* g_pre, void* stream);
/* scale = +1 or -1: the attacks MAXIMISE the cost, i.e. hand autograd -mean(.) (phy_obj_atk.py:95 `cost = -loss(...)`); with
 * scale = -1 the sign is applied here (bit-identical to negating the result and the incoming gradient: two element-wise
 * launches per attack step less). */
int dmh_roi_cost_fwd_scaled(const float* d_pre, const float* mask, const int* org, int B, int hd, int wd, int H, int W,
                            float scale, float* sig, float* partials, float* cost, void* stream);
int dmh_roi_cost_bwd_scaled(const float* sig, const float* mask, const int* org, int B, int hd, int wd, int H, int W,
                            float scale, const float* gscale, float* g_pre, void* stream);

/* K19, encoder side: the backward of conv1 / bn1 / relu / maxpool and layer1 (torchvision ResNet under
 * MD2/networks/resnet_encoder.py:85-98) on one window per scene -- the patch gradient reads d cost / d image under the pasted
 * object only (physicalTrans.py:156-165, phy_obj_atk.py:96).
 *   roi_crop: out[B,C,hc,wc] = src[b, c, org_b + (i, j)] of a whole-frame src[B,C,H,W], times [gate > 0] (gate: whole-frame,
 *             same window; gate_compact != 0: gate is itself a compact [B,C,hc,wc] window at org) when gate is given; or, with g
 *             (compact [B,C,hc,wc]) instead of src, out = g * [gate window > 0].
 *             org [B,2] int32 with even columns, wc and W even.
 *   stem_bn_relu_pool_bwd_win: dmh_stem_bn_relu_pool_bwd on the hs x ws window (even origin org, even sizes) of the H x W
 *             map: g_z[B,C,hs,ws] compact; g_pool is a compact [B,C,hq,wq] window of the H/2 x W/2 map at pool_org (cells
 *             outside it count as 0: it must hold the pooling cells that cover the window); g_feat whole-frame or NULL. */
int dmh_roi_crop(const float* src, const float* gate, const float* g, const int* org, int B, int C, int H, int W, int hc,
                 int wc, int gate_compact, float* out, void* stream);
/* roi_paste: dst[b, c, win_org_b + (i, j)] = src at the same frame position, for the h x w window: src is a compact
 * [B,C,sh,sw] window at frame origin src_org [B,2] that holds it, or (src_org NULL, sh x sw = H x W) a whole-frame tensor.
 * The incremental attack forward writes the part of encoder feature 1 that the pasted object changes into the cached
 * feature of the clean scenes, and puts the clean values back afterwards. */
int dmh_roi_paste(const float* src, const int* src_org, int sh, int sw, const int* win_org, int B, int C, int H, int W, int h,
                  int w, float* dst, void* stream);
int dmh_stem_bn_relu_pool_bwd_win(const float* feat, const unsigned char* argmax, const float* g_feat, const float* g_pool,
                                  const float* scale, const int* org, const int* pool_org, int B, int C, int H, int W, int hs,
                                  int ws, int hq, int wq, float* g_z, void* stream);

/* ------------------------------------------------------------------------------------
 * K9  encoder glue: the element-wise passes between the MIOpen convolutions of the ResNet encoder while the model
 *     is in eval() mode (every attack step: torchattacks/attack.py:165-182 brackets the attack with model.eval()).
 *     Replaces, in MD2/networks/resnet_encoder.py:85-98 / torchvision BasicBlock.forward, the chains
 *     BatchNorm2d(eval) -> [+ identity] -> ReLU and, in the stem, BatchNorm2d(eval) -> ReLU -> MaxPool2d(3, 2, 1).
 *     scale[c] = weight / sqrt(running_var + eps), shift[c] = bias - running_mean * scale (computed by the caller).
 *   bn_act : out[B,C,HW] = act(x * scale[c] + shift[c] (+ residual)),  act = ReLU when relu != 0.  residual may be NULL.
 *            bwd: g_residual = relu ? g_out * [out > 0] : g_out (written when non-NULL);  g_x = g_residual * scale[c].
 *            `out` may be NULL in the backward when relu == 0.
 *   stem   : feat[B,C,H,W] = ReLU(x * scale[c] + shift[c]); pooled[B,C,H/2,W/2] = maxpool3x3/2/1(feat);
 *            argmax[B,C,H/2,W/2] (u8, ky*3+kx, first maximum in scan order as ATen's max_pool2d).  H, W even.
 *            bwd: g_x = scale[c] * [feat > 0] * (g_feat + maxpool_adjoint(g_pooled));  either gradient may be NULL.
 * ---------------------------------------------------------------------------------- */
int dmh_bn_act_fwd(const float* x, const float* scale, const float* shift, const float* residual, int B, int C, int HW,
                   int relu, float* out, void* stream);
int dmh_bn_act_bwd(const float* out, const float* g_out, const float* scale, int B, int C, int HW, int relu, float* g_x,
                   float* g_residual, void* stream);
/* train-mode BatchNorm statistics (model.train(): the training pass of MD2/trainer.py:335-375): per-channel batch mean
 * and biased variance of x[B,C,HW] (two launches, fp32 Welford/Chan combination of shifted sums), from which
 *   scale = weight * invstd, shift = bias - mean * scale      (feed dmh_bn_act_fwd / dmh_stem_bn_relu_pool_fwd)
 *   save_mean, save_invstd                                    (for aten::native_batch_norm_backward)
 * and running_mean / running_var are updated in place with `momentum` (unbiased variance), as nn.BatchNorm2d does.
 * weight / bias / running_* may be NULL.  partials: dmh_bn_stats_partials_size(B, C, HW) floats of scratch. */
int64_t dmh_bn_stats_partials_size(int B, int C, int HW);
int dmh_bn_train_stats(const float* x, int B, int C, int HW, const float* weight, const float* bias, float momentum,
                       float eps, float* running_mean, float* running_var, float* partials, float* scale, float* shift,
                       float* save_mean, float* save_invstd, void* stream);
/* The same, and num_batches_tracked[0] += 1 (the module's int64 counter, nn.BatchNorm2d.forward in train mode; may be NULL) in
 * the second launch instead of one more element-wise launch per BatchNorm (20 per train pass of the ResNet-18 encoder). */
int dmh_bn_train_stats_tracked(const float* x, int B, int C, int HW, const float* weight, const float* bias, float momentum,
                               float eps, float* running_mean, float* running_var, long long* num_batches_tracked,
                               float* partials, float* scale, float* shift, float* save_mean, float* save_invstd,
                               void* stream);
/* out[c] = sum over (b, hw) of g[b][c][hw]: the bias gradient of a convolution (aten::convolution_backward's third
 * output, aten::sum(g, (0, 2, 3))).  Two launches, fixed order.  partials: dmh_channel_sum_partials_size(B, C, HW) floats. */
int64_t dmh_channel_sum_partials_size(int B, int C, int HW);
int dmh_channel_sum(const float* g, int B, int C, int HW, float* partials, float* out, void* stream);
/* train-mode BatchNorm backward with the ReLU mask folded in (replaces one dmh_bn_act_bwd pass +
 * aten::miopen_batch_norm_backward in the train pass; torch.nn.BatchNorm2d semantics, torch/nn/functional.py batch_norm
 * as called by torchvision's BasicBlock, MD2/networks/resnet_encoder.py:85-98):
 *   g' = out ? g_out * [out > 0] : g_out;   g_bias = sum g';   g_weight = invstd * sum g' (x - mean);
 *   g_x = weight invstd ( g' - g_bias / N - (x - mean) invstd^2 sum g' (x - mean) / N ),  N = B * HW.
 * out (the saved ReLU output), weight (NULL = 1), g_weight, g_bias, g_pre (receives g', the gradient of a residual branch)
 * may be NULL.  workspace: dmh_bn_train_bwd_workspace_size(B, C, HW) floats, 16-byte aligned.  Three launches, fixed-order
 * sums: bitwise reproducible. */
int64_t dmh_bn_train_bwd_workspace_size(int B, int C, int HW);
int dmh_bn_train_bwd(const float* x, const float* g_out, const float* out, const float* weight, const float* save_mean,
                     const float* save_invstd, int B, int C, int HW, float* workspace, float* g_x, float* g_weight,
                     float* g_bias, float* g_pre, void* stream);
int dmh_stem_bn_relu_pool_fwd(const float* x, const float* scale, const float* shift, int B, int C, int H, int W,
                              float* feat, float* pooled, unsigned char* argmax, void* stream);
int dmh_stem_bn_relu_pool_bwd(const float* feat, const unsigned char* argmax, const float* g_feat, const float* g_pooled,
                              const float* scale, int B, int C, int H, int W, float* g_x, void* stream);

/* ------------------------------------------------------------------------------------
 * K10 3x3 stride-1 convolution, Winograd F(2x2,3x3) on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *     Replaces the MIOpen call behind nn.Conv2d(.., 3, 1, pad) in torchvision BasicBlock (encoder,
 *     MD2/networks/resnet_encoder.py:85-98) and Conv3x3 (decoder, MD2/layers.py:127-141), forward and backward-data.
 *   weight_transform: U = G w G^T in the chunked layout the kernel streams; `backward` != 0 builds the filter of the
 *                     backward-data pass (flipped, channel roles swapped).  U holds dmh_wino_weight_size(n_out, n_in)
 *                     floats, n_out/n_in = channel roles of that pass; n_in must be a multiple of 8.
 *   conv3x3         : y[B,K,H+2pad-2,W+2pad-2] = corr3x3(zero_pad(x[B,C,H,W], pad), w) (+ bias[K]);  pad in {0,1,2};
 *                     output height and width must be even.  Backward-data of a pad-p convolution is this call on
 *                     g_out with the backward filter and pad = 2 - p.
 * ---------------------------------------------------------------------------------- */
int64_t dmh_wino_weight_size(int n_out, int n_in);
int dmh_wino_weight_transform(const float* w, int K, int C, int backward, float* U, void* stream);
int dmh_wino_conv3x3(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                     float* y, void* stream);
/* The same with a caller-owned workspace (device, 16-byte aligned, workspace_floats floats; the library keeps nothing).
 * With whole work items (64 output channels x 64 tiles x ALL input channels) a launch lasts ceil(items / CUs) item times: 288
 * items on 256 CUs take two rounds, 60 items leave 196 CUs idle.  Given a workspace of 2 * CUs * 16,384 floats (32 MB on
 * MI355X) the launch is decomposed stream-K style instead: the (item, 8-channel chunk) units are dealt to the workgroups in
 * equal contiguous ranges, a range that begins or ends inside an item stores that item's partial sums to the workspace,
 * and a second kernel adds an item's pieces in chunk order + bias -- deterministic, no atomics, no zero fill.  Taken only
 * where a cost model says it is faster; NULL / too small a workspace = dmh_wino_conv3x3. */
int dmh_wino_conv3x3_ws(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                        float* y, float* workspace, int64_t workspace_floats, void* stream);
/* What a dmh_wino_conv3x3(_act)_ws call of this shape would do, WITHOUT launching: -1 for a shape the kernel does not take, else
 * bit 0: stream-K form, bit 1: two-way channel split, bits 2-3: tile regions (0: 2 x 32 per image, 1: 4 x 16 per image, 2: 4 x 16
 * over the flattened batch), bits 8...: work items.  `epilogue` != 0 asks for the _act form; workspace_floats = the size the
 * caller would pass (0: none).  The caller's dispatch rule (ops._wino_ok) asks this instead of mirroring the decision. */
int dmh_wino_conv3x3_plan(int B, int C, int K, int H, int W, int pad, int epilogue, int64_t workspace_floats);
/* conv -> BatchNorm(eval) -> (+ identity) -> ReLU of a BasicBlock in one launch (torchvision BasicBlock.forward under
 * MD2/networks/resnet_encoder.py:85-98, model in eval()): the per-channel scale is folded into the filter by
 * weight_transform_scaled (backward != 0: into the filter of the backward-data pass, whose input is then the masked
 * output gradient), the shift is `bias`, `residual` (same shape as y, may be NULL) is added before the ReLU.
 *   y = act(corr3x3(zero_pad(x), w * scale[k]) + bias[k] (+ residual)),  act = ReLU when relu & 1.
 * relu & 2: `residual` is not an addend but a saved ReLU output m: y = (corr3x3(...) + bias[k]) * [m > 0] -- the
 * backward-data pass of one convolution of a block, masked for the ReLU in front of it, in the same launch. */
int dmh_wino_weight_transform_scaled(const float* w, int K, int C, int backward, const float* scale, float* U,
                                     void* stream);
int dmh_wino_conv3x3_act(const float* x, const float* U, const float* bias, const float* residual, int relu, int B, int C,
                         int K, int H, int W, int pad, float* y, void* stream);
/* Many filters in one launch: jobs[i] = one dmh_wino_weight_transform_scaled call (scale may be NULL); `jobs` is a HOST array,
 * read before the call returns (its entries travel in the kernel arguments, 32 per launch).  A training step re-transforms
 * every filter twice (forward and backward-data form, once for the attack's frozen weights and once for the train pass): ~76
 * launches of 5-20 us otherwise. */
typedef struct dmh_wino_wt_job {
    const float* w;         /* [K][C][3][3] */
    const float* scale;     /* per-channel factor of the forward OUTPUT (eval-mode BatchNorm), or NULL */
    float* U;               /* dmh_wino_weight_size(n_out, n_in) floats */
    int K, C, backward;
} dmh_wino_wt_job;
int dmh_wino_weight_transform_batch(const dmh_wino_wt_job* jobs, int n, void* stream);
/* The same with a caller-owned stream-K workspace (see dmh_wino_conv3x3_ws).  With the fused epilogue only launches of fewer
 * than 200 tile regions are decomposed (layer4 at the attack batch: 120); the second kernel applies shift / residual / ReLU. */
int dmh_wino_conv3x3_act_ws(const float* x, const float* U, const float* bias, const float* residual, int relu, int B, int C,
                            int K, int H, int W, int pad, float* y, float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------
 * K17 the 32-output-channel form of K10 (work item 32 channels x 128 tiles): the decoder layers whose output channel
 *     count is a multiple of 32 but not of 64 -- upconv(1,0) 64 -> 32 and upconv(1,1) 96 -> 32 forward, 32 -> 96
 *     backward-data (MD2/networks/depth_decoder.py:51-63 via MD2/layers.py:127-141 Conv3x3).  Same contract as
 *     dmh_wino_conv3x3; its own filter layout (channel rows padded to 32 instead of 64): weight_size / weight_transform.
 * ---------------------------------------------------------------------------------- */
int64_t dmh_wino32_weight_size(int n_out, int n_in);
int dmh_wino32_weight_transform(const float* w, int K, int C, int backward, float* U, void* stream);
int dmh_wino32_conv3x3(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                       float* y, void* stream);
/* The same with a caller-owned stream-K workspace (see dmh_wino_conv3x3_ws): 2 * CUs * 16,384 floats. */
int dmh_wino32_conv3x3_ws(const float* x, const float* U, const float* bias, int B, int C, int K, int H, int W, int pad,
                          float* y, float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------
 * K18 weight gradient of a 3x3 stride-1 convolution with C and K multiples of 64, in the Winograd F(2x2,3x3) domain on the
 *     fp32 MFMA: dw[K][C][3][3] = d/dw sum(dy * corr3x3(zero_pad(x, pad), w)) -- what autograd asks
 *     aten.convolution_backward(dy, x, w, ..., [False, True, False]) for in the train pass (MD2/trainer.py:305-309 backward
 *     through torchvision BasicBlock / MD2/layers.py:127-141 Conv3x3).  x [B,C,H,W], dy [B,K,H+2pad-2,W+2pad-2] (even sizes);
 *     workspace: dmh_wino_wrw_workspace_size(...) floats (per-workgroup partial sums, added in a fixed order: no atomics).
 * ---------------------------------------------------------------------------------- */
int64_t dmh_wino_wrw_workspace_size(int B, int C, int K, int H, int W, int pad);
int dmh_wino_wrw(const float* x, const float* dy, int B, int C, int K, int H, int W, int pad, float* workspace, float* dw,
                 void* stream);

/* ------------------------------------------------------------------------------------
 * K11 3x3 stride-1 convolution with few channels (<=4 -> <=32, 16 -> <=32 or 32 -> <=16) at full resolution, direct implicit
 *     GEMM on v_mfma_f32_16x16x4_f32 with the filter held in registers: the last decoder stage and the disparity heads
 *     (MD2/networks/depth_decoder.py:38-44).  w is the FORWARD filter [Kw][Cw][3][3] in both directions:
 *       backward == 0:  y[B,Kw,H+2pad-2,W+2pad-2] = corr3x3(zero_pad(x[B,Cw,H,W], pad), w) + bias
 *       backward != 0:  y[B,Cw,...]               = corr3x3(zero_pad(x[B,Kw,H,W], pad), flip(w)^T)   (bias ignored if NULL)
 *     so the gradient w.r.t. the input of a pad-p convolution is the backward call on g_out with pad = 2 - p.
 * ---------------------------------------------------------------------------------- */
int dmh_conv3x3_small(const float* x, const float* w, const float* bias, int B, int Kw, int Cw, int H, int W, int pad,
                      int backward, float* y, void* stream);

/* ------------------------------------------------------------------------------------
 * K12 gradient w.r.t. the input image of the encoder's first convolution, nn.Conv2d(Cin, K, 7, stride 2, padding 3)
 *     (torchvision ResNet.conv1, MD2/networks/resnet_encoder.py:88): g_x[B,Cin,H,W] from g_y[B,K,H/2,W/2] and the forward
 *     filter w[K][Cin][7][7].  Direct gather, no atomics.  Cin <= 4, K a multiple of 8, H and W even.
 * ---------------------------------------------------------------------------------- */
int dmh_conv7x7s2_bwd_data(const float* g_y, const float* w, int B, int K, int Cin, int H, int W, float* g_x,
                           void* stream);
/* Window form (K19): only the hwin x wwin window of g_x at the per-sample, even pixel origin img_org [B,2] is written (at its
 * place in the whole [B,Cin,H,W] tensor); g_y is a compact [B,K,sh,sw] window of the H/2 x W/2 frame at origin gy_org [B,2]
 * that must hold rows / columns (img_org / 2 - 1) .. ((img_org + hwin) / 2 + 1) of that frame as far as they lie inside it. */
int dmh_conv7x7s2_bwd_data_win(const float* g_y, const float* w, const int* img_org, const int* gy_org, int B, int K, int Cin,
                               int H, int W, int hwin, int wwin, int sh, int sw, float* g_x, void* stream);

/* ------------------------------------------------------------------------------------
 * K13 disparity head: 3x3 stride-1 convolution to ONE output channel (MD2/networks/depth_decoder.py:43-44 dispconv),
 *     forward: y[B,1,H+2pad-2,W+2pad-2] = corr3x3(zero_pad(x[B,C,H,W], pad), w[1][C][3][3]) + bias[0].
 *     pad 0 (the heads on the reflection-padded features): strips of 62 columns x 40 rows per wave, one load per input
 *     row, neighbours by DPP, C a multiple of 4; otherwise vector FMAs on an LDS tile, C a multiple of 16.
 * ---------------------------------------------------------------------------------- */
int dmh_conv3x3_head(const float* x, const float* w, const float* bias, int B, int C, int H, int W, int pad, float* y,
                     void* stream);
/* gradient w.r.t. the input of the same layer at pad 0: g_x[B,C,H,W] from g[B,1,H-2,W-2] (full correlation with the
 * flipped filter); C a multiple of 4.  One gradient plane in registers, C planes streamed out. */
int dmh_conv3x3_head_bwd_data(const float* g, const float* w, int B, int C, int H, int W, float* g_x, void* stream);
/* weight and bias gradient of the same layer (train pass): g_w[1][C][3][3] = sum_{b,y,x} g[b,0,y,x] * zero_pad(x)[b,c,y+ky,x+kx],
 * g_b[0] = sum g (g_b may be NULL).  `partials`: dmh_conv3x3_head_wrw_partials_size(...) floats of workspace (per-strip
 * sums, added in a fixed order: deterministic).  Any C. */
int64_t dmh_conv3x3_head_wrw_partials_size(int B, int C, int H, int W, int pad);
int dmh_conv3x3_head_wrw(const float* x, const float* g, int B, int C, int H, int W, int pad, float* partials, float* g_w,
                         float* g_b, void* stream);

/* ------------------------------------------------------------------------------------
 * K16 weight and bias gradient of the 3x3 stride-1 convolutions with 16 OUTPUT channels and 16 or 32 input channels (the
 *     last decoder stage, MD2/networks/depth_decoder.py:38-41 upconv(0,0) / upconv(0,1); train pass):
 *     g_w[16][C][3][3] = sum_{b,y,x} g[b,k,y,x] * zero_pad(x)[b,c,y+ky,x+kx],  g_b[16] = sum g  (g_b may be NULL).
 *     Pixel axis on the fp32 MFMA; `partials`: dmh_conv3x3_small_wrw_partials_size(C) floats of workspace
 *     (per-workgroup sums, added in a fixed order: deterministic).
 * ---------------------------------------------------------------------------------- */
int64_t dmh_conv3x3_small_wrw_partials_size(int C);
int dmh_conv3x3_small_wrw(const float* x, const float* g, int B, int C, int H, int W, int pad, float* partials, float* g_w,
                          float* g_b, void* stream);

/* ------------------------------------------------------------------------------------
 * K14 the encoder's first layer with its input normalisation fused (MD2/networks/resnet_encoder.py:89-90:
 *     x = (input_image - 0.45) / 0.225;  x = conv1(x),  conv1 = nn.Conv2d(3, 64, 7, stride 2, padding 3, bias=False)):
 *     y[B,64,H/2,W/2] = corr7x7_s2(zero_pad3((x[B,3,H,W] - mean) / std), w[64][3][7][7]) on the exact-fp32 MFMA.
 *     H and W even.  The gradient w.r.t. x is dmh_conv7x7s2_bwd_data(g_y, w) / std.
 * ---------------------------------------------------------------------------------- */
int dmh_stem_conv_norm_fwd(const float* x, const float* w, int B, int H, int W, float mean, float std, float* y,
                           void* stream);
/* Window form (K19): only the hw x ww window of the H/2 x W/2 output at the per-sample origin org [B,2] is computed, into a
 * compact y[B,64,hw,ww]; x stays the whole image (zero padding outside it as above). */
int dmh_stem_conv_norm_fwd_win(const float* x, const float* w, const int* org, int B, int H, int W, int hw, int ww, float mean,
                               float std, float* y, void* stream);

/* ------------------------------------------------------------------------------------
 * K15 the two convolutions that open a down-sampling ResNet block (torchvision BasicBlock.conv1 with stride 2 and
 *     BasicBlock.downsample[0], the layers behind MD2/networks/resnet_encoder.py:94-98), one launch per direction:
 *     fwd:  y3[B,Co,H/2,W/2] = corr3x3_s2(zero_pad1(x[B,Ci,H,W]), w3[Co][Ci][3][3]);
 *           yd[B,Co,H/2,W/2] = corr1x1_s2(x, wd[Co][Ci])                      (wd, yd both NULL: 3x3 only)
 *     bwd:  g_x[B,Ci,H,W] = adjoint3x3(g3) + adjoint1x1(gd) with the filters TRANSPOSED: w3t[Ci][Co][3][3],
 *           wdt[Ci][Co]                                                      (gd, wdt both NULL: 3x3 only)
 *     Exact-fp32 MFMA.  H, W even; fwd: Ci % 8 == 0, Co % 64 == 0; bwd: Co % 8 == 0, Ci % 64 == 0.
 * ---------------------------------------------------------------------------------- */
int dmh_down_conv_fwd(const float* x, const float* w3, const float* wd, int B, int Cin, int Cout, int H, int W,
                      float* y3, float* yd, void* stream);
/* the same with eval-mode BatchNorms folded in (scales already in the filters): y3 = act(corr3x3_s2(x, w3) + shift3[k]),
 * act = ReLU when relu3 != 0; yd = corr1x1_s2(x, wd) + shiftd[k]; shift3 / shiftd may be NULL. */
int dmh_down_conv_fwd_act(const float* x, const float* w3, const float* wd, const float* shift3, const float* shiftd,
                          int relu3, int B, int Cin, int Cout, int H, int W, float* y3, float* yd, void* stream);
int dmh_down_conv_bwd_data(const float* g3, const float* gd, const float* w3t, const float* wdt, int B, int Cin, int Cout,
                           int H, int W, float* g_x, void* stream);
/* the same with g_add[B, Cin, H, W] (may be NULL) added in the epilogue: the gradient the same tensor receives from its other
 * consumer (the decoder's skip connection), instead of autograd's separate accumulation pass. */
int dmh_down_conv_bwd_data_acc(const float* g3, const float* gd, const float* w3t, const float* wdt, const float* g_add, int B,
                               int Cin, int Cout, int H, int W, float* g_x, void* stream);

/* Round 5: the same two kernels with 16-byte loaders.  The filters are handed over as an IMAGE of the kernels' LDS layout,
 * written once per weight tensor: image[rows / 32][inner / 8][2560] from w3[rows][inner][3][3] and wd[rows][inner] (NULL: zeros)
 * -- forward: rows = Co, inner = Ci (the filters as they are); backward: rows = Ci, inner = Co (the TRANSPOSED filters w3t, wdt).
 * dmh_down_conv_image_size: floats of the image, -1 unless rows % 32 == 0 and inner % 8 == 0.  The *_img entry points take the
 * arguments of dmh_down_conv_fwd_act / dmh_down_conv_bwd_data_acc with the image in place of the filter pair and a flag for the
 * shortcut convolution; they read the tensors in aligned 16-byte words and therefore need W % 4 == 0 (forward) / W % 8 == 0
 * (backward) -- other shapes stay with the entry points above.  Results are bit-identical to theirs. */
int64_t dmh_down_conv_image_size(int rows, int inner);
int dmh_down_conv_weight_image(const float* w3, const float* wd, int rows, int inner, float* image, void* stream);
int dmh_down_conv_fwd_img(const float* x, const float* image, int has_down, const float* shift3, const float* shiftd, int relu3,
                          int B, int Cin, int Cout, int H, int W, float* y3, float* yd, void* stream);
int dmh_down_conv_bwd_data_img(const float* g3, const float* gd, const float* image, const float* g_add, int B, int Cin, int Cout,
                               int H, int W, float* g_x, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DMH_HIP_H */

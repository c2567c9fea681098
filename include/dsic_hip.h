/* dsic_hip.h — C ABI of libdsic_hip.so, the MI355X (gfx950) hot path of the
 * modelv2 Student-t hyperprior codec.
 *
 * The reference (Dimitrinov74/Domain-Specific-Image-Compression) has no FFI
 * layer: its seam is the Python module API of code/modelv2.  Each entry point
 * below replaces the torch op sequence of the cited reference lines; the
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and lives
 * in domain-specific-image-compression_amd/lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - the library allocates nothing and keeps no state; callers own all
 *     buffers including workspaces;
 *   - activations are NHWC float32 with a channel count that is a multiple
 *     of 8; image tensors at the boundary are NCHW float32 like the
 *     reference's (eval_selfcontained.py:58-59);
 *   - return value: 0 = ok, DSIC_EINVAL = bad argument, DSIC_EHIP = a HIP
 *     runtime error (hipGetLastError text via dsic_last_error()).
 */
#ifndef DSIC_HIP_H
#define DSIC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSIC_OK 0
#define DSIC_EINVAL 1
#define DSIC_EHIP 2

/* activation fused into the conv epilogue */
#define DSIC_ACT_NONE 0
#define DSIC_ACT_GDN 1   /* x / sqrt(beta + gamma*x^2)   layers.py:19-27 */
#define DSIC_ACT_IGDN 2  /* x * sqrt(beta + gamma*x^2)   layers.py:24-25 */
#define DSIC_ACT_RELU 3  /* nn.ReLU                      layers.py:108-111 */

const char* dsic_last_error(void);
int dsic_abi_version(void);

/* The arithmetic variant of every contraction of the library, one run-time switch (the reference has no such
 * notion: it is torch's fp32 conv, model.py:27-35; both variants are held to its outputs):
 * 1 = operands split into two bf16 planes, bf16 MFMAs, fp32 accumulation (default); 0 = fp32-input MFMAs.
 * The environment variable DSIC_WINO_BF16=0 sets the initial value to 0.  Weights packed for one variant must be
 * re-packed after a switch (the Python layers key their caches on it). */
int dsic_split_bf16(void);
int dsic_set_split_bf16(int on);

/* ---- weight re-layout (once per checkpoint load) ------------------------ */

/* nn.Conv2d weight [Cout,Cin,k,k] (layers.py:29-31) -> packed
 * [k*k][CinP/8][CoutP][8], CinP = ceil8(Cin), CoutP = ceil32(Cout), zero
 * padded.  dst must hold dsic_packed_conv_weight_floats() floats. */
int64_t dsic_packed_conv_weight_floats(int Cout, int Cin, int k);
int dsic_pack_conv_weight(const float* w_oihw, float* dst, int Cout, int Cin,
                          int k, void* stream);

/* nn.ConvTranspose2d(Cin,Cout,5,2,2,output_padding=1) weight [Cin,Cout,5,5]
 * (layers.py:83,89,93,123,124) -> the four sub-pixel phase kernels
 * (3x3,3x2,2x3,2x2 taps = all 25 taps) packed [25][Cin/8][CoutP][8]. */
int dsic_pack_convT_weight(const float* w_iohw, float* dst, int Cin, int Cout,
                           void* stream);

/* Last synthesis layer ConvTranspose2d(Cin,Cimg,5,2,2,1) (layers.py:97):
 * the four phases become the 4*Cimg <= 16 output columns of ONE 3x3 stride-1
 * contraction over the input grid; packed [9][Cin/16][16][16] fp32, followed
 * by the same values as two bf16 planes [Cin/16][9][hi,mid][half][16][8]
 * (the B operands of the split-bf16 kernel) = together
 * dsic_convT_image_weight_floats(Cin) floats.  Cin % 16 == 0, Cimg in [1,4]. */
int64_t dsic_convT_image_weight_floats(int Cin);
int dsic_pack_convT_image_weight(const float* w_iohw, float* dst, int Cin,
                                 int Cimg, void* stream);

/* ---- layout helpers ------------------------------------------------------ */

/* NCHW image [B,C,H,W] -> NHWC with C padded to 8 (zeros). */
int dsic_image_to_nhwc8(const float* x_nchw, float* dst_nhwc8, int B, int C,
                        int H, int W, void* stream);
/* NHWC [B,H,W,C] -> NCHW [B,C,H,W] */
int dsic_nhwc_to_nchw(const float* src, float* dst, int B, int H, int W, int C,
                      void* stream);
/* NCHW [B,C,H,W] -> NHWC [B,H,W,C] */
int dsic_nchw_to_nhwc(const float* src, float* dst, int B, int C, int H, int W,
                      void* stream);

/* F.pad(x,(0,pad_w,0,pad_h),mode="reflect") on [planes,H,W] (modelseval.py:57-64:
 * pad bottom/right to a multiple of 16).  dst: [planes,H+pad_h,W+pad_w]. */
int dsic_reflect_pad_br(const float* src, float* dst, int planes, int H, int W,
                        int pad_h, int pad_w, void* stream);

/* Band preparation of code/combinebandsall.py:7-12,35-36 for stacked Sentinel-2 bands:
 * per plane b -= min(b); b /= max(b) (if non-zero).  bands/out01: [planes][HW] float32
 * (planes = images x bands); out_u8 (may be NULL): uint8(b*255) as written to PNG. */
int dsic_normalize_bands(const float* bands, float* out01, uint8_t* out_u8,
                         int planes, int HW, void* stream);

/* ---- convolutions (fp32 MFMA implicit GEMM, fused bias + activation) ----- */

/* conv() + optional GDN/ReLU: nn.Conv2d(Cin,Cout,k,stride,padding=(k-1)/2)
 * (layers.py:29-31, 51-72, 86-94, 108-112).  k in {3,5}; (k,stride) in
 * {(3,1),(5,2)}.  in: NHWC [B,H,W,CinP]; out: NHWC [B,Ho,Wo,Cout],
 * Ho = ceil(H/stride).  beta/gamma are the EFFECTIVE per-channel values
 * (param^2 - 2^-18, layers.py:20-21); ignored unless act is GDN/IGDN. */
int dsic_conv2d_nhwc(const float* in, const float* w_packed, const float* bias,
                     const float* beta, const float* gamma, float* out, int B,
                     int H, int W, int CinP, int Cout, int k, int stride,
                     int act, void* stream);

/* 3x3 stride-1 conv() + optional GDN/IGDN/ReLU by Winograd F(2x2,3x3) on the
 * fp32 MFMA (layers.py:56,62,67,86,90,94,108,109): 2.25x fewer multiplies than
 * dsic_conv2d_nhwc, same fp32 operands.  u_packed from dsic_pack_wino_weight
 * (G g G^T per (cout,cin), [16][Cin/8][CoutP][8]).  Cin % 32 == 0, Cout % 4 == 0,
 * Cout <= 128.  in NHWC [B,H,W,Cin] -> out NHWC [B,H,W,Cout].  ticket: 16 bytes of device
 * memory zeroed ONCE by the caller (dynamic tile hand-out of the persistent kernel; each launch
 * leaves it zeroed again; launches sharing a ticket must be ordered on one stream). */
int64_t dsic_wino_weight_floats(int Cout, int Cin);
int dsic_pack_wino_weight(const float* w_oihw, float* dst, int Cout, int Cin,
                          void* stream);
int dsic_conv3x3_wino_nhwc(const float* in, const float* u_packed,
                           const float* bias, const float* beta,
                           const float* gamma, float* out, int B, int H, int W,
                           int Cin, int Cout, int act, int s2d_out, int s2d_in,
                           void* ticket, void* stream);
/* conv(Cs,Cout,5,2) (layers.py:54,60,65) as a 3x3 stride-1 Winograd conv over the
 * space-to-depth input [B,H/2,W/2,4*Cs] (channel (a*2+b)*Cs+c = x[2i+a][2j+b][c],
 * written by the producing layer when its s2d_out flag is set): 16 instead of
 * 25 multiplies per output and channel pair.  Pack the reference [Cout,Cs,5,5]
 * weight with dsic_pack_wino_s2_weight and call dsic_conv3x3_wino_nhwc with
 * Cin = 4*Cs on the half-resolution grid and s2d_in = 1 (lets the kernel skip the
 * Winograd positions that are structurally zero for the phases lacking a tap row/column). */
int dsic_pack_wino_s2_weight(const float* w_oihw5, float* dst, int Cout, int Cs,
                             void* stream);
/* ConvTranspose2d(Cin,Cout,5,2,2,output_padding=1) + optional IGDN/ReLU
 * (layers.py:83,89,93,123,124) by Winograd: each sub-pixel phase (py,px) is a 3x3
 * stride-1 conv over the input grid with taps g[wr][wc] = w[py+4-2wr][px+4-2wc]
 * (zero for wr < py or wc < px), so all four share the transformed input.  dst/u_packed4:
 * 4 * dsic_wino_weight_floats(Cout, Cin) floats.  in NHWC [B,H,W,Cin] -> out NHWC
 * [B,2H,2W,Cout].  Cin % 32 == 0, Cout % 4 == 0, Cout <= 128. */
int dsic_pack_wino_convT_weight(const float* w_iohw5, float* dst, int Cin,
                                int Cout, void* stream);
int dsic_conv_transpose2d_wino_nhwc(const float* in, const float* u_packed4,
                                    const float* bias, const float* beta,
                                    const float* gamma, float* out, int B,
                                    int H, int W, int Cin, int Cout, int act,
                                    void* ticket, void* stream);

/* Winograd layers on the bf16 MFMA path with fp32-class results (csrc/conv_wino_bf16.hip): the
 * fp32 operands are split into bf16 planes (x = hi + mid [+ lo]) and the cross products are
 * accumulated in fp32.  dsic_wino_bf16_planes(): planes the library was built with (2: products
 * hh, hm, mh; 3: + mm, hl, lh).  u_f32: transformed weights as written by dsic_pack_wino_weight /
 * dsic_pack_wino_s2_weight (nphase 1) or dsic_pack_wino_convT_weight (nphase 4); dst:
 * nphase * dsic_wino_bf16_weight_bytes(Cout, Cin) bytes, [phase][16][Cin/16][planes][CoutP][16] bf16.
 * The two conv entry points mirror dsic_conv3x3_wino_nhwc / dsic_conv_transpose2d_wino_nhwc
 * (same layers of code/modelv2/layers.py:54-72, 83-97, 108-124; Cin a multiple of 32, >= 64).
 * out_cstride / out_coff (floats; 0 / 0 = a dense [B,H,W,Cout] output): the Cout <= 128 channels of this
 * call are a slice of a wider NHWC tensor - conv(128,192,5,2) (layers.py:72) runs as a 128- and a
 * 64-channel call into one [B,H,W,192] tensor. */
int dsic_wino_bf16_planes(void);
int64_t dsic_wino_bf16_weight_bytes(int Cout, int Cin);
int dsic_split_wino_weight_bf16(const float* u_f32, void* dst, int Cout, int Cin,
                                int nphase, void* stream);
int dsic_conv3x3_wino_bf16_nhwc(const float* in, const void* u_planes,
                                const float* bias, const float* beta,
                                const float* gamma, float* out, int B, int H, int W,
                                int Cin, int Cout, int act, int s2d_out, int s2d_in,
                                int out_cstride, int out_coff, void* ticket, void* stream);
/* Split-K for layers whose images hold fewer than four 16x8-pixel tiles (g_a.10/12/14, h_a.*: layers.py:66-72,
 * 108-112 on 16x16 .. 4x4 latents), where a workgroup would otherwise walk all Cin/16 chunks of a tile alone:
 * dsic_wino_bf16_ksplit(H, W, Cin) -> S (a function of the layer geometry only, 1 = do not split);
 * dsic_conv3x3_wino_bf16_splitk_nhwc = dsic_conv3x3_wino_bf16_nhwc with the input channels of every tile
 * shared by S work items, whose partial sums (partials: S buffers of the size of `out`, caller-owned) are
 * added in fixed order, biased and activated by a second launch. */
int dsic_wino_bf16_ksplit(int H, int W, int Cin);
/* 1 when dsic_conv3x3_wino_bf16_nhwc / dsic_conv_transpose2d_wino_bf16_nhwc run the layer with the 64-tile,
 * two-pass kernel (csrc/conv_wino_bf16m.hip: H and W multiples of 16, at least 4 work items per image; nphase = 4
 * for ConvTranspose2d, 1 otherwise) - a function of the layer geometry only, like the split-K rule. */
int dsic_wino_bf16_m64(int H, int W, int Cin, int nphase);
int dsic_conv3x3_wino_bf16_splitk_nhwc(const float* in, const void* u_planes,
                                       const float* bias, const float* beta,
                                       const float* gamma, float* out, int B, int H, int W,
                                       int Cin, int Cout, int act, int s2d_out, int s2d_in,
                                       int out_cstride, int out_coff, int ksplit,
                                       float* partials, void* ticket, void* stream);
int dsic_conv_transpose2d_wino_bf16_nhwc(const float* in, const void* u_planes4,
                                         const float* bias, const float* beta,
                                         const float* gamma, float* out, int B,
                                         int H, int W, int Cin, int Cout, int act,
                                         void* ticket, void* stream);

/* First analysis layer conv(Cimg,Cout,3,1) + optional GDN/ReLU (layers.py:51)
 * read straight from the NCHW image [B,Cimg,H,W] (Cimg 3 or 4) with K = 9*Cimg;
 * w_oihw is the reference weight [Cout,Cimg,3,3] unpacked; out NHWC [B,H,W,Cout],
 * Cout a multiple of 4, <= 128.  s2d_out: store space-to-depth
 * [B,H/2,W/2,4*Cout] for a following 5x5/s2 layer run by Winograd. */
int dsic_conv_first_nchw(const float* x_nchw, const float* w_oihw,
                         const float* bias, const float* beta,
                         const float* gamma, float* out_nhwc, int B, int Cimg,
                         int H, int W, int Cout, int act, int s2d_out,
                         void* stream);
/* The same layer from the decoded image bytes: x_u8_nhwc [B][H][W][Cimg] uint8 (PIL / numpy layout);
 * to_tensor (code/modelv2/modelseval.py:66-67, eval_selfcontained.py:58-59: float(v)/255) is fused
 * into the kernel's window staging. */
int dsic_conv_first_u8hwc(const unsigned char* x_u8_nhwc, const float* w_oihw,
                          const float* bias, const float* beta, const float* gamma,
                          float* out_nhwc, int B, int Cimg, int H, int W, int Cout,
                          int act, int s2d, void* stream);
/* to_tensor alone: uint8 [B][H][W][C] -> float32 [B][C][H][W] in [0,1] (the x of the metrics). */
int dsic_image_u8hwc_to_f32nchw(const unsigned char* x_u8_nhwc, float* out_nchw,
                                int B, int C, int H, int W, void* stream);


/* ConvTranspose2d(Cin,Cout,5,2,2,output_padding=1) + optional IGDN/ReLU.
 * in NHWC [B,H,W,Cin] -> out NHWC [B,2H,2W,Cout]. */
int dsic_conv_transpose2d_nhwc(const float* in, const float* w_packed,
                               const float* bias, const float* beta,
                               const float* gamma, float* out, int B, int H,
                               int W, int Cin, int Cout, int act, void* stream);

/* Last synthesis layer: in NHWC [B,H,W,Cin] -> x_hat NCHW [B,Cimg,2H,2W]. */
int dsic_conv_transpose2d_image(const float* in, const float* w_packed,
                                const float* bias, float* out_nchw, int B,
                                int H, int W, int Cin, int Cimg, void* stream);

/* ---- hyper-synthesis heads, quantisation, rate --------------------------- */

/* AdaptiveAvgPool2d(1) -> mlp_sigma / mlp_nu (1x1 conv, ReLU, 1x1 conv)
 * (layers.py:131-139,147-151), then sigma = exp(log_sigma),
 * nu = clamp(exp(log_nu), min_nu, max_nu) (model.py:54-55; the spatial mean
 * of a spatially constant tensor is the tensor).  t: NHWC [B,HW,N] = ReLU
 * output of h_s.  1x1 weights are passed INPUT-major: w1 [N][N], w2 [N][M]
 * (= reference weight[o,i,0,0] transposed).  Outputs [B,M]. */
int dsic_hyper_params(const float* t_nhwc, const float* w1_sigma,
                      const float* b1_sigma, const float* w2_sigma,
                      const float* b2_sigma, const float* w1_nu,
                      const float* b1_nu, const float* w2_nu,
                      const float* b2_nu, float* log_sigma, float* log_nu,
                      float* sigma, float* nu, int B, int HW, int N, int M,
                      float min_nu, float max_nu, void* stream);

/* quantize("round") (model.py:27-35, 44-45, 62), StudentT.neg_log2_prob
 * (distributions.py:20-31), FactorizedGaussian.neg_log2_prob
 * (distributions.py:39-46) and the per-image sums behind model.py:76.
 * y NHWC [B,HWy,M], z NHWC [B,HWz,N]; sigma, nu [B,M]; z_log_sigma [N].
 * per_element: sigma, nu are NCHW [B,M,HWy] (spatial_params=True) instead of [B,M].
 * y_noisy/z_noisy (NHWC, may be NULL): when given they ARE y_tilde/z_tilde
 * (quant_mode="noise"), otherwise y_tilde = round(y).  Outputs: y_hat NHWC
 * (= round(y), the synthesis input), y_tilde/z_tilde/nll_y/nll_z in the
 * reference's NCHW, sums[B][2] = {sum nll_y, sum nll_z} as fp64 (each image is
 * summed by 16 workgroups whose partial sums are added in a fixed order).
 * work: dsic_rate_workspace_doubles(B) doubles of device scratch. */
int64_t dsic_rate_workspace_doubles(int B);
int dsic_rate(const float* y_nhwc, const float* z_nhwc,
              const float* y_noisy_nhwc, const float* z_noisy_nhwc,
              const float* sigma, const float* nu, const float* z_log_sigma,
              float* y_hat_nhwc, float* y_tilde_nchw, float* z_tilde_nchw,
              float* nll_y_nchw, float* nll_z_nchw, double* sums, double* work,
              int B, int HWy, int M, int HWz, int N, int per_element, void* stream);

/* spatial_params=True branch of model.py:49-51: the two 3x3 heads' NHWC outputs
 * [B,HW,M] -> sigma = exp(log_sigma), nu = clamp(exp(log_nu), min_nu, max_nu) as
 * NCHW [B,M,HW]; dsic_rate / dsic_cdf_tables_student then take per_element = 1. */
int dsic_sigma_nu_spatial(const float* log_sigma_nhwc, const float* log_nu_nhwc,
                          float* sigma_nchw, float* nu_nchw, int B, int HW, int M,
                          float min_nu, float max_nu, void* stream);

/* StudentT.neg_log2_prob (distributions.py:20-31) elementwise on NCHW x[n].
 * per_channel=1: sigma/nu are [B*C] (spatially constant, HW elements each);
 * per_channel=0: sigma/nu are full tensors of n elements. */
int dsic_student_t_bits(const float* x, const float* sigma, const float* nu,
                        float* out, int64_t n, int HW, int per_channel,
                        void* stream);
/* FactorizedGaussian.neg_log2_prob (distributions.py:39-46), x NCHW. */
int dsic_gaussian_bits(const float* x, const float* log_sigma, float* out,
                       int B, int C, int HW, void* stream);

/* torch.round (half to even) elementwise: quantize(x,"round"), model.py:32-33 */
int dsic_round(const float* x, float* out, int64_t n, void* stream);

/* Stand-alone GDN (inverse=0) / IGDN (inverse=1) on NCHW (layers.py:19-27);
 * beta/gamma are the effective per-channel values. */
int dsic_gdn_nchw(const float* x, const float* beta, const float* gamma,
                  float* out, int B, int C, int HW, int inverse, void* stream);

/* ---- distortion metrics -------------------------------------------------- */

/* One MS-SSIM level (pytorch-msssim 1.0.0 semantics, called at
 * modelseval.py:78-88, eval_selfcontained_entropy.py:154): 11-tap Gaussian
 * "valid" filtering, cs/ssim maps, spatial means.  X,Y: [planes,H,W] (planes =
 * B*C of an NCHW tensor); partial: workspace of dsic_ssim_partial_doubles()
 * doubles; means: [planes][2] = (mean cs, mean ssim).  clamp_x clamps X to
 * [0,1] on load (modelseval.py:178). */
int64_t dsic_ssim_partial_doubles(int planes, int H, int W);
int dsic_ssim_level(const float* X, const float* Y, double* partial,
                    double* means, int planes, int H, int W, float C1, float C2,
                    int clamp_x, void* stream);
/* The same level plus the inputs of the next one: Xn, Yn [planes,Hn,Wn] =
 * F.avg_pool2d(kernel=2, padding=size%2) of (clamped) X and of Y
 * (pytorch-msssim's downsampling between levels, modelseval.py:80-85), Hn =
 * (H + 2*(H%2) - 2)/2 + 1.  For even H and W % 4 == 0
 * (dsic_ssim_level_pool_fused) the pool comes out of the pass that reads the
 * level; other shapes run dsic_ssim_level + 2 x dsic_avgpool2. */
int dsic_ssim_level_pool_fused(int H, int W);
int dsic_ssim_level_pool(const float* X, const float* Y, double* partial,
                         double* means, float* Xn, float* Yn, int planes, int H,
                         int W, float C1, float C2, int clamp_x, void* stream);
/* F.avg_pool2d(kernel=2, padding=size%2) between MS-SSIM levels. */
int dsic_avgpool2(const float* src, float* dst, int planes, int H, int W,
                  int clamp, void* stream);
/* out[b] = mean_c prod_l relu(v_l)^w_l from means [levels][B*C][2]. */
int dsic_msssim_finalize(const double* means, const float* weights, float* out,
                         int levels, int B, int C, int relu_last, void* stream);
/* out[b] = sum (a-b)^2 over one image (F.mse_loss numerator,
 * modelseval.py:69-76); clamp_a clamps a to [0,1] first. */
int dsic_sqerr_per_image(const float* a, const float* b, double* out, int B,
                         int64_t n_per_image, int clamp_a, void* stream);

/* ---- per-patch entropy coding ------------------------------------------- */
/* Restates custom_compress / custom_decompress of
 * code/modelv2/eval_selfcontained_entropy.py:26-123 and the torchac 0.9.3
 * calls inside them (third-party; :48,62,96,116).  The script cannot execute
 * as written; the frozen interpretation is in DESIGN.md.
 * err: device int, OR-ed with 1 (support wider than Lmax), 2 (symbol outside
 * its support), 4 (output capacity exceeded); callers zero it first. */

/* :39-41,52-54: meta[b] = {min(y)-tail, Ly, min(z)-tail, Lz}, L = max-min+2*tail+1.
 * y_nchw/z_nchw: integer-valued float latents (y_tilde, z_tilde), n_* per image. */
int dsic_latent_support(const float* y_nchw, const float* z_nchw, int* meta,
                        int B, int64_t n_y, int64_t n_z, int tail, void* stream);

/* :43-47 gaussian PMF -> pmf_to_uint16_cdf (:17-23) -> spread table.
 * sigma_z [N] = exp(z_prior.log_sigma) (:32, no clamp); tables [B][N][Lmax]
 * uint16, entry k = c[k] for k < L_b (c[L_b] = 65536 implicit). */
int dsic_cdf_tables_gauss(const float* sigma_z, const int* meta,
                          uint16_t* tables, int B, int N, int Lmax, int* err,
                          void* stream);
/* :55-61 Student-t PMF tables; sigma, nu [B][rows]: rows = M for spatially constant
 * parameters (spatial_params=False), rows = M*Hy*Wy (NCHW order = symbol order)
 * for per-element parameters (spatial_params=True, layers.py:127-129). */
int dsic_cdf_tables_student(const float* sigma, const float* nu,
                            const int* meta, uint16_t* tables, int B, int rows,
                            int Lmax, int* err, void* stream);

/* torchac.encode_float_cdf call sites :48,62: per image the z string then the
 * y string.  per_element_y: tab_y has one row per y symbol ([B][M*HWy][Lmax])
 * instead of one per channel.  out: [B][cap_z + cap_y] bytes, ZERO-INITIALISED by the caller (bits
 * are OR-ed in; z at offset 0, y at cap_z),
 * lengths [B][2] = {len_z, len_y}.  Symbol order C,H,W of the NCHW latents. */
int dsic_range_encode(const float* y_nchw, const float* z_nchw, const int* meta,
                      const uint16_t* tab_y, const uint16_t* tab_z, int Lmax,
                      int B, int M, int HWy, int N, int HWz, uint8_t* out,
                      int64_t cap_y, int64_t cap_z, int* lengths, int* err,
                      int streams_per_wg, int per_element_y, void* stream);

/* torchac.decode_float_cdf call sites :96,116: string b starts at
 * in + b*stride and has lengths[b*lstride + loff] bytes; meta_off 0 = y, 2 = z.
 * out: NCHW float latents [B][C][HW] (symbol + min). */
int dsic_range_decode(const uint8_t* in, int64_t stride, const int* lengths,
                      int lstride, int loff, const int* meta, int meta_off,
                      const uint16_t* tables, int Lmax, int B, int C, int HW,
                      int per_element, float* out_nchw, int* err, void* stream);

/* HIP stream limited to the CUs whose bit is set in mask_host[words] (bit i of
 * word i/32 = CU i).  Used to give the range coder its own few CUs beside the
 * conv kernels; there is no reference counterpart (the reference is
 * single-stream).  The caller destroys the stream. */
int dsic_stream_create_masked(const uint32_t* mask_host, int words,
                              void** stream_out);
int dsic_stream_destroy(void* stream);

/* The table math evaluated on the HOST (no GPU needed): lets CPU-only tests
 * compare it bit for bit with the oracle and with the reference-generated
 * fixture tests/golden/entropy_ref.npz.
 * dsic_host_gaussian_cdf_f32: eval_selfcontained_entropy.py:14-15 in float32.
 * dsic_host_pmf_to_uint16_cdf: :17-23; pmf_host [L][C] float32 (support axis
 *   first), out_host [L+1][C] uint16.
 * dsic_host_cdf_table: one coder table (:41-47 or :54-61 + spreading);
 *   out_host: L uint16; raw_host: NULL or L+1 uint16 = the pre-spreading cdf of :22. */
double dsic_host_normal_cdf(double x);
double dsic_host_student_t_cdf(double t, double nu);
float dsic_host_gaussian_cdf_f32(float x);
/* sigma_z = exp(z_prior.log_sigma) (:32) as float32(exp64(x)): one value for encoder and
 * decoder on any machine (torch's expf differs between its CPU and GPU builds). */
float dsic_host_exp_f32(float x);
int dsic_host_pmf_to_uint16_cdf(const float* pmf_host, int L, int C,
                                uint16_t* out_host);
int dsic_host_cdf_table(int student, float sigma, float nu, int smin, int L,
                        uint16_t* out_host, uint16_t* raw_host);

#ifdef __cplusplus
}
#endif
#endif /* DSIC_HIP_H */

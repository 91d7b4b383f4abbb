/*
 * mi_ddpm.h -- C ABI of the MI355X (gfx950) DDPM hot-path library (libmi_ddpm.so).
 *
 * The reference (Victarry/Image-Generation-models) has NO FFI or plugin interface: its
 * hot path is Python calling stock ATen ops.  This header is
 * therefore the boundary a maintainer would bind from src/models/ddpm.py (ctypes stub in
 * INTEGRATION.md); every entry point cites the reference lines it replaces
 * (paths relative to /root/reference/).
 *
 * Contract
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch's caching allocator);
 *    the library never allocates, frees, retains or synchronises -> hipGraph-capture safe.
 *  - `stream` is a hipStream_t passed as void*; work is ordered only by that stream.
 *  - return 0 on success, <0 argument/shape error, >0 a hipError_t; mi_last_error() gives
 *    the thread-local message.  Nothing throws across the boundary.
 *  - activations are fp32 NHWC ("pixel-major"): element (n,y,x,c) at ((n*H+y)*W+x)*ld + c,
 *    ld >= C, ld % 4 == 0, base 16-byte aligned.
 *  - convolution weights live in tap-major layout  w[ky][kx][Cin][Cout]  (a permuted VIEW of
 *    the PyTorch [Cout,Cin,kh,kw] / ConvTranspose [Cin,Cout,kh,kw] parameter, see
 *    image-generation-models_amd/src/models/ddpm.py) so forward, dgrad, wgrad and the
 *    optimizer all stream the same contiguous buffer with no repack.
 *  - `mode`: 0 = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = bf16 MFMA with fp32
 *    accumulate (v_mfma_f32_32x32x16_bf16; operands rounded to bf16 when staged to LDS).
 */
#ifndef MI_DDPM_H
#define MI_DDPM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_ABI_VERSION 4   /* 2: mi_pack_weights_bf16 takes the fragment-order copies; mi_adam_step_dev betas are double.  3: mi_conv1x1_pw takes x2.  4: GroupNorm epilogue sums are 64-bit fixed point */
#define MI_MODE_FP32 0
#define MI_MODE_BF16 1

int mi_abi_version(void);
const char* mi_last_error(void);

/* ---- implicit-GEMM convolution on MFMA ---------------------------------------------
 * One kernel family for nn.Conv2d k3/k1 (ddpm.py:116,134,151-152,236), stride-2
 * Downsample (ddpm.py:79), ConvTranspose2d k4 s2 (ddpm.py:70), every data-gradient of
 * those (aten::convolution_backward) and nn.Linear (ddpm.py:127,190-192; H=W=1).
 *   y[n,oy,ox,j] (+)= bias[j] + residual[n,oy,ox,j]
 *                     + sum_{ky,kx,k} src[n,iy,ix,k] * W(ky,kx,k,j)
 *   transposed=0: iy = oy*stride - pad + ky          (convolution forward)
 *   transposed=1: iy = (oy + pad - ky)/stride, only where divisible (transposed conv / dgrad)
 *   src channel k comes from x when k < K1, else from x2 at k-K1 (skip concat ddpm.py:255
 *   without materialising the cat).
 *   W(ky,kx,k,j) = w[((ky*KW+kx)*K + k)*Nc + j] if w_kn else w[((ky*KW+kx)*Nc + j)*K + k].
 */
typedef struct MiConvDesc {
    int N;               /* batch */
    int IH, IW;          /* spatial size of the gathered tensor */
    int OH, OW;          /* spatial size of the produced tensor */
    int K;               /* contraction channels */
    int Nc;              /* produced channels */
    int KH, KW, stride, pad;
    int transposed;
    int w_kn;
    int mode;
    int K1;              /* channels taken from x; K1 == K -> single source */
    int ldx, ldx2, ldy, ldr;
    int accumulate;      /* y += result instead of y = result */
} MiConvDesc;

int mi_conv_igemm(const MiConvDesc* d, const float* x, const float* x2, const float* w,
                  const float* bias, const float* residual, float* y, void* stream);

/* same contract with the bf16 weight copy [tap][Nc][K] of mi_pack_weights_bf16 (mode 1, K % 8 == 0);
 * used for the stride-2 / transposed convolutions */
int mi_conv_igemm_bf16w(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                        const float* bias, const float* residual, float* y, void* stream);
/* ... with bf16-STORED activations (x_is_bf16 != 0: x / x2 are bf16 tensors, strides in elements and % 8 == 0, K and K1 % 64 == 0):
 * the Downsample conv, the Upsample transposed conv and their data gradients (ddpm.py:70,79) read the bf16 copies that the
 * skip connection / the weight gradient already have -- a 16-byte load is 8 channels, so a ring stage covers 64 channels (half the
 * barriers per MFMA, half the bytes, no pack).  Only layers the ring kernel takes; query _supported first.  y stays fp32. */
int mi_conv_igemm_bf16w_io_supported(const MiConvDesc* d);
int mi_conv_igemm_bf16w_io(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16,
                           const float* bias, const float* residual, float* y, int x_is_bf16, void* stream);
/* tile instantiation mi_conv_igemm will launch for d (BM x BN); used to attribute profiles */
int mi_conv_igemm_tile(const MiConvDesc* d, int* bm, int* bn);

/* ---- tap-gather GEMM (round 4): the stride-2 Downsample conv (ddpm.py:79), the ConvTranspose2d(4, 2, 1) Upsample (ddpm.py:70) and
 * their data gradients for bf16-stored activations, same contract as mi_conv_igemm (d->transposed = 1: the parity classes of the
 * produced tensor run as blockIdx.z).  x bf16 (K % 64 == 0, one source, ldx % 8 == 0), weights in mi_pack_weights_bf16's fragment order
 * (wfq for the contraction over the master layout's ci, wdq over co), Nc % 64 == 0, power-of-two class grids, N*GH*GW % 64 == 0.
 * Activations by LDS-DMA gathered per tap (the stride is free), private weight streams, one barrier per 128 contraction channels.
 * d->mode = 0: the exact-fp32 instantiation -- x / y fp32 (ldx % 4 == 0), weights from mi_pack_weights_f32frag (wfq32 / wdq32). */
int mi_conv_gt_supported(const MiConvDesc* d);
int mi_conv_gt_tile(const MiConvDesc* d, int* pixels_per_workgroup, int* classes);
int mi_conv_gt(const MiConvDesc* d, const void* x, const void* w_frag, const float* bias, const float* residual,
               void* y, int out_bf16, void* stream);
/* ... and a bf16 copy of the fp32 output from the same epilogue (y_bf16, pixel stride ldy16 elements; bf16 mode, no accumulate) */
int mi_conv_gt_dual(const MiConvDesc* d, const void* x, const void* w_frag, const float* bias, const float* residual,
                    float* y, void* y_bf16, int ldy16, void* stream);

/* ---- 3x3 / stride 1 / pad 1 convolution with an LDS-staged halo tile (bf16 MFMA only) -------
 * Same contract as mi_conv_igemm for the Block conv (ddpm.py:116) and its data gradient
 * (d->transposed = 1 selects the flipped-tap form), but activations are staged once per
 * 9 taps and weights come from a bf16 copy laid out  w[ky][kx][Nc][K]  (k contiguous), made by
 * mi_pack_weights_bf16.  Returns <0 when the descriptor is not supported
 * (check with mi_conv3x3_bf16w_supported and fall back to mi_conv_igemm). */
int mi_conv3x3_bf16w(const MiConvDesc* d, const float* x, const float* x2, const void* w_nk_bf16,
                     const float* bias, const float* residual, float* y, void* stream);
int mi_conv3x3_bf16w_supported(const MiConvDesc* d);
/* 1 when the layer would run the split-K plan (small-M levels; partial sums are combined with fp32 atomics, so
 * the output has to be fp32): the caller keeps such layers' block-internal tensors in fp32. */
int mi_conv3x3_bf16w_uses_splitk(const MiConvDesc* d);
/* profiling attribution: the template arguments of the kernel instantiation a descriptor runs on
 * (conv3x3_halo_kernel<bm, ck, KH, sk, io, waves>, waves = 8 for bm == 256 and for the bm 128 / ck 64 form, else 4) */
int mi_conv3x3_bf16w_tile(const MiConvDesc* d, int io, int* bm, int* ck, int* sk);
/* bf16 activation storage for the ResnetBlock-internal tensors (conv output -> GroupNorm -> conv input and
 * their gradients): io bit 0 = x / x2 are bf16 tensors, bit 1 = y is written as bf16; strides count
 * elements.  3x3 and 1x1; with bit 1 the split-K variant (fp32 atomics) is not used. */
int mi_conv3x3_bf16w_io(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16,
                        const float* bias, const float* residual, void* y, int io, void* stream);
/* ---- the fused GroupNorm-apply + Mish (+ time bias) + Conv3x3 kernel (BASELINE.json's named kernel) -------------------------
 * ResnetBlock: h = block1(x) = mish(GN(conv(x))); h += mlp(t); h = block2(h) (ddpm.py:112-120,136-143).  Here block2's conv
 * reads the RAW output c1 of block1's conv and applies  mish(c1 * scale[n][c] + shift[n][c]) + tb[n][c]  in registers while it
 * stages its halo tile, so the normalised tensor is never written to or read from HBM.  coef = [3][N][K] (scale = rstd*gamma,
 * shift = beta - mean*scale, tb = the block's time bias) comes from mi_gn_stats_coef, a statistics-only pass over c1 that also
 * writes stats[n][g] = {mean, rstd}.  Forward / sampling path; tiles lie inside one image.  io: 0 = fp32 c1 -> fp32 y, 3 = bf16 ->
 * bf16 (the block-internal storage of bf16 mode). */
/* (mi_gn_stats_coef is declared with the GroupNorm entry points below) */
/* GroupNorm statistics of the NEXT layer from this conv's epilogue (instead of mi_gn_stats_coef's pass over the tensor):
 * mi_conv3x3_bf16w_io_gnsums is mi_conv3x3_bf16w_io that also adds, per sample and 16-channel slab, the sum and the sum of squares of
 * the values it stores (rounded to bf16 when y is bf16) into gsum [N][Nc / 16][2] (zeroed by the caller; Nc % 16 == 0, H*W % 32 == 0).
 * ABI 4: the elements of gsum / sums are 64-bit fixed-point integers with 20 fraction bits (16 bytes per slab; the float* in the
 * signatures is an opaque, 16-byte aligned pointer): they are added with integer atomics, so the totals -- and everything computed
 * from them -- do not depend on the order in which the workgroups arrive;
 * mi_gn_coef_from_sums combines the slabs of each group into stats [N][G][2] = {mean, rstd} (optional) and coef [3][N][C] as
 * mi_gn_stats_coef would have written them (C / G % 16 == 0).  Block -> Block: conv1 + sums, coef, then mi_conv3x3_gn_mish. */
/* mi_conv3x3_bf16w_io that also writes the bf16 copy of its fp32 output (y_bf16, pixel stride ldy16 elements): LinearAttention's
 * to_out conv + residual (ddpm.py:45,152) produces a residual-stream tensor whose bf16 copy the skip connection's consumer, the
 * Downsample conv and the weight gradients read -- one epilogue instead of a conversion pass over y. */
int mi_conv3x3_bf16w_io_dual(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16, const float* bias,
                             const float* residual, float* y, void* y_bf16, int ldy16, int io, void* stream);
int mi_conv3x3_bf16w_io_gnsums(const MiConvDesc* d, const void* x, const void* x2, const void* w_nk_bf16, const float* bias,
                               const float* residual, void* y, int io, float* gsum, void* stream);
int mi_gn_coef_from_sums(int N, int C, int G, int HW, float eps, const float* sums, const float* gamma, const float* beta,
                         const float* temb, int ldt, float* stats, float* coef, void* stream);
/* ... or skip the coefficient tensor: the fused kernel resolves statistics, affine and time bias (temb [N][ldt], optional) per channel
 * chunk from the sums themselves.  Block -> Block is then two launches: conv1 (+ sums) and this one. */
int mi_conv3x3_gn_mish_sums(const MiConvDesc* d, const void* x, const float* sums, const float* gamma, const float* beta,
                            const float* temb, int ldt, int G, float eps, const void* w_nk_bf16, const float* bias, void* y,
                            int io, void* stream);
int mi_conv3x3_gn_mish_supported(const MiConvDesc* d);
int mi_conv3x3_gn_mish_tile(const MiConvDesc* d, int* bm, int* ck);
int mi_conv3x3_gn_mish(const MiConvDesc* d, const void* x, const float* coef, const void* w_nk_bf16, const float* bias,
                       void* y, int io, void* stream);
/* bf16 shadow copies of every conv weight of the flat fp32 parameter buffer (master layout
 * [tap][Cin][Cout] at float offset `off`): wd = same layout, wf = [tap][Cout][Cin].
 * entries_dev: device array of {int64 off; int32 taps, ci, co, tile0}, tile0 = first tile
 * index of the entry (prefix sum of taps*ceil(ci/T)*ceil(co/T), T = mi_pack_weights_tile()); total_tiles = grid size. */
/* tile edge T the entries' tile0 fields are counted in: an entry owns taps * ceil(ci / T) * ceil(co / T) consecutive tiles */
int mi_pack_weights_tile(void);
/* entries_dev (ABI 2): {int64 off; int32 taps, ci, co, tile0, frag, pad}.  Entries with frag != 0 (ci % 64 == 0 and co % 64 == 0)
 * are also written in MFMA-fragment order for mi_conv3x3_pw (wdq / wfq may both be NULL: no such copies):
 *   wfq[tap][co/32][ci/16][lane][8] = W[tap][ci = 16 kq + 8 (lane >> 5) + e][co = 32 nb + (lane & 31)]   (forward operand)
 *   wdq[tap][ci/32][co/16][lane][8] = W[tap][ci = 32 nb + (lane & 31)][co = 16 kq + 8 (lane >> 5) + e]   (data-gradient operand) */
int mi_pack_weights_bf16(int nent, const void* entries_dev, int total_tiles, const float* master,
                         void* wd_bf16, void* wf_bf16, void* wdq_bf16, void* wfq_bf16, void* stream);
/* ---- 3x3 conv / data gradient for bf16-stored activations with wave-private weight streams (conv_pw.hip; replaces the per-tap
 * workgroup barrier of the kernels above: reference op src/models/ddpm.py:116).  w_frag_bf16 = the layer's slice of wfq (forward) or
 * wdq (d->transposed = 1: data gradient, flipped taps).  Needs W in {8, 16, 32} with 128-pixel row tiles (N*H*W % 128 == 0),
 * K % 64 == 0, K1 % 64 == 0, Nc % 64 == 0, ldy % 8 == 0 (pw_ok / pw_launch in conv_pw.hip are the authority). */
int mi_conv3x3_pw_supported(const MiConvDesc* d);
/* ... for fp32 x / x2 (the residual stream as it is; pixel strides in floats, % 4 == 0): rounded to bf16 once while staged; gsum
 * (optional): the GroupNorm sums of mi_conv3x3_pw_gnsums */
int mi_conv3x3_pw_x32_supported(const MiConvDesc* d);
int mi_conv3x3_pw_x32_tile(const MiConvDesc* d);
int mi_conv3x3_pw_x32(const MiConvDesc* d, const float* x, const float* x2, const void* w_frag_bf16, const float* bias,
                      const float* residual, void* y, int out_bf16, float* gsum, void* stream);
/* ... and in exact-fp32 mode (d->mode = 0; Unet.compute_mode = "fp32", the reference's default precision): the same kernel on
 * v_mfma_f32_32x32x2_f32 with fp32 x / x2 / y and fp32 weights in fragment order, written by mi_pack_weights_f32frag from the table
 * of mi_pack_weights_bf16:  wfq32[tap][co/32][ci/8][lane][4] = W[tap][ci = 8 ko + 4 (lane >> 5) + j][co = 32 nb + (lane & 31)],
 * wdq32[tap][ci/32][co/8][lane][4] = W[tap][ci = 32 nb + (lane & 31)][co = 8 ko + 4 (lane >> 5) + j].  K % 32 == 0, K1 % 32 == 0. */
int mi_conv3x3_pw_f32_tile(const MiConvDesc* d);
int mi_conv3x3_pw_f32(const MiConvDesc* d, const float* x, const float* x2, const float* w_frag_f32, const float* bias,
                      const float* residual, float* y, void* stream);
int mi_conv1x1_pw_x32_supported(const MiConvDesc* d);     /* ... reading fp32 x / x2 (strides in floats, % 4): the fp32 stream gradient, inference */
int mi_conv1x1_pw_x32(const MiConvDesc* d, const float* x, const float* x2, const void* w_frag_bf16, const float* bias,
                      const float* residual, void* y, int out_bf16, void* stream);
/* split-K launches of the 3x3 tile kernel (mi_conv3x3_pw*; the layers whose tiles would leave CUs without a workgroup: the sampler at B = 64,
   BASELINE configs[2] at 32 images per GPU): ws = device memory the caller zeroed once and keeps alive (64 KB of flags + the fp32 partial tiles;
   a launch that does not fit runs unsplit), registered for the calling thread's device; launches that use it must be ordered on one stream.
   null / 0: no split launches.  mi_conv3x3_pw_splitk: the split a launch would use now (1: none) + 16 x its pixel tile. */
int mi_conv_pw_set_splitk_workspace(void* ws, size_t bytes);
int mi_conv3x3_pw_splitk(const MiConvDesc* d, int var, int in32);
int mi_debug_conv_pw_splitk(int mode);                     /* tests: 0 off, 1 the rule, 2 / 4 that split wherever the geometry allows; -> previous */
/* inference: channel LayerNorm (reference src/models/ddpm.py:85-95) + the bias-free 1x1 conv behind it (to_qkv, :151) in one launch -- the
   normalised tensor is never written.  x fp32 [M][ldx], ln_g / ln_b [K], K = 128 / 256 / 512, y bf16 [M][ldy] */
int mi_ln_conv1x1_pw_supported(const MiConvDesc* d);
int mi_ln_conv1x1_pw(const MiConvDesc* d, const float* x, const float* ln_g, const float* ln_b, float eps, const void* w_frag_bf16,
                     const float* bias, void* y_bf16, void* stream);
/* ... that also writes the normalised tensor (bf16 [M][ldl]): training -- to_qkv's weight gradient reads it, no LayerNorm launch */
int mi_ln_conv1x1_pw_dual(const MiConvDesc* d, const float* x, const float* ln_g, const float* ln_b, float eps, const void* w_frag_bf16,
                          const float* bias, void* y_bf16, void* ln_bf16, int ldl, void* stream);
int mi_conv1x1_pw_f32_supported(const MiConvDesc* d);     /* the 1x1 convs in exact-fp32 mode: K % 64 == 0, K1 % 64 == 0, Nc % 64 == 0 */
int mi_conv1x1_pw_f32(const MiConvDesc* d, const float* x, const float* x2, const float* w_frag_f32, const float* bias,
                      const float* residual, float* y, void* stream);
int mi_pack_weights_f32frag(int nent, const void* entries_dev, int total_tiles, const float* master, float* wdq32, float* wfq32,
                            void* stream);
int mi_debug_conv_pw_tile(int pt);               /* tests: force the pixel tile (64 / 128 / 256), 0 = automatic */
int mi_debug_conv_pw_auto256(int min_workgroups); /* A/B switch: the automatic pick takes 256-pixel tiles from this many workgroups up (0 = never; default 256) */
int mi_debug_conv1x1_pw_nloop(int on);          /* A/B switch: 0 = the bf16 -> bf16 K = 128 1x1 conv (to_qkv) always on the 2-D grid, 1 = the channel-tile loop from 1024 pixel tiles up (default), 2 = from any grid (tests), < 0 = query only; returns the previous value */
int mi_conv3x3_pw_tile(const MiConvDesc* d);      /* pixels per workgroup the launch would use: 256 (grids that still fill the chip), 128, or 64 for small grids; 0 = unsupported */
int mi_conv3x3_pw(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias,
                  const float* residual, void* y, int out_bf16, void* stream);
/* ... whose epilogue also adds the sum and the sum of squares of the values it stores (rounded to bf16 when y is bf16), per sample and
 * 16-channel slab, into gsum [N][Nc/16][2] (zeroed by the caller, Nc % 16 == 0): the statistics of the GroupNorm that follows
 * (mi_gn_coef_from_sums) without a pass over y -- one atomic pair per slab, image and workgroup. */
int mi_conv3x3_pw_gnsums(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias,
                         const float* residual, void* y, int out_bf16, float* gsum, void* stream);
/* BASELINE.json's named kernel on this structure (reference src/models/ddpm.py:112-120,136-143):
 *   y = conv3x3(mish(x * scale[n][c] + shift[n][c]) + tb[n][c]) + bias,
 * x = the RAW bf16 output of block1's conv, coef [3][N][K] fp32 = scale, shift, tb (mi_gn_coef_from_sums or mi_gn_stats_coef).  The
 * transform is applied ONCE per staged element: a wave rewrites, in place in LDS, the activation pieces it requested itself (after its
 * own counted wait, before the chunk barrier publishes them); rows outside the image stay zero.  Tiles lie inside one image
 * (W in {16, 32}); the normalised tensor never exists in HBM. */
int mi_conv3x3_pw_gn_mish_supported(const MiConvDesc* d);
int mi_conv3x3_pw_gn_mish_tile(const MiConvDesc* d);          /* pixels per workgroup the fused variants would use (128 / 64; 0: not supported) */
int mi_conv3x3_pw_gn_mish(const MiConvDesc* d, const void* x, const float* coef, const void* w_frag_bf16, const float* bias,
                          void* y, int out_bf16, void* stream);
/* ... or without the coefficient tensor: scale / shift are resolved per channel chunk inside the kernel from the sums the producing
 * conv's epilogue left (sums [N][K/16][2], mi_conv3x3_pw_gnsums / mi_conv3x3_bf16w_io_gnsums), gamma, beta and the time-bias rows
 * temb [N][ldt] (optional) with mi_gn_coef_from_sums' arithmetic.  K / G in {16, 32, 64}.  Block -> Block = two launches. */
int mi_conv3x3_pw_gn_mish_sums(const MiConvDesc* d, const void* x, const float* sums, const float* gamma, const float* beta,
                               const float* temb, int ldt, int G, float eps, const void* w_frag_bf16, const float* bias,
                               void* y, int out_bf16, void* stream);
/* The same two for fp32-STORED activations (block storage fp32; x = the raw fp32 output of the previous conv, pixel stride in floats,
 * % 4 == 0): the transform is applied to the fp32 values while they pass through registers on their way to the bf16 tile (one rounding,
 * no LDS read-modify-write).  y fp32 or bf16.  The canonical shape of BASELINE.json's named kernel in the regime SURVEY 8(d)(ii) calls balanced. */
int mi_conv3x3_pw_x32_gn_mish_supported(const MiConvDesc* d);
int mi_conv3x3_pw_x32_gn_mish_tile(const MiConvDesc* d);      /* pixels per workgroup (128 / 64; 0: not supported) */
int mi_conv3x3_pw_x32_gn_mish(const MiConvDesc* d, const float* x, const float* coef, const void* w_frag_bf16, const float* bias,
                              void* y, int out_bf16, void* stream);
int mi_conv3x3_pw_x32_gn_mish_sums(const MiConvDesc* d, const float* x, const float* sums, const float* gamma, const float* beta,
                                   const float* temb, int ldt, int G, float eps, const void* w_frag_bf16, const float* bias,
                                   void* y, int out_bf16, void* stream);
/* ---- 1x1 convs with K % 128 == 0 input channels on the same machinery (to_qkv, to_out + residual, res_conv -- also over the skip
 * concat's two sources, x2 = channels K1 .. K - 1 with K1 % 128 == 0 -- and their data gradients; reference src/models/ddpm.py:134,151-152):
 * 128 (or 64) pixels x 128 output channels per workgroup, the activation tile double-buffered per 128-channel chunk by LDS-DMA, weight
 * fragments straight into registers per wave, whole rows leave through LDS.  w_frag_bf16 = the layer's slice of wfq (forward) / wdq
 * (data gradient); y fp32 or bf16 (out_bf16); y_bf16 (optional, fp32 y, no accumulate): the bf16 copy of y from the same epilogue
 * (pixel stride ldy16). */
int mi_conv1x1_pw_supported(const MiConvDesc* d);
int mi_conv1x1_pw(const MiConvDesc* d, const void* x, const void* x2, const void* w_frag_bf16, const float* bias, const float* residual,
                  void* y, int out_bf16, void* y_bf16, int ldy16, void* stream);

/* ---- the 3-channel ends of the UNet (fp32 VALU, bound by the wide tensor they stream) ------------
 * Conv2d(Cin<=4, Cout, ks, padding=ks/2), ks = 3 (downs.0.0.block1, ddpm.py:116,208) or 1 (its res_conv,
 * ddpm.py:134): forward and weight gradient; w / dW in the tap-major layout [ks][ks][Cin][Cout].
 * Forward: Cout % 4 == 0 and Cout/4 divides 256.  Weight gradient: Cout in {64, 128, 256} ({64, 128} for ks = 3);
 * with a workspace of mi_conv_small_wgrad_workspace(ks*ks*Cin*Cout) bytes the per-workgroup partial tiles are
 * summed in a fixed order (deterministic), without one they are added with fp32 atomics. */
int mi_conv_small_cin_fwd(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                          const float* bias, float* y, int ldy, void* stream);
int mi_conv_small_cin_wgrad(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* dy,
                            int lddy, float* dW, void* workspace, size_t ws_bytes, void* stream);
size_t mi_conv_small_wgrad_workspace(int outputs);
/* Round 4: the wide (Cout-channel) tensor stored as bf16, like every other block-internal tensor of bf16 mode (the reference keeps
 * c1 = conv(x) and its gradient in fp32, ddpm.py:116-120; x, w, dW stay fp32, the arithmetic is the same fp32 FMA chain, the output
 * is rounded once / dy is widened on load).  y_bf16 / dy_bf16 = 1 needs the whole-row-tile kernels: ks = 3, W a power of two <= 64,
 * H*W a power of two, ldx == 4, 16-byte aligned x, Cout in {64, 128} (256: forward only) -- mi_conv_small_cin_bf16_supported answers for both. */
int mi_conv_small_cin_bf16_supported(int ks, int N, int H, int W, int Cin, int Cout, int ldx);
/* round 6, inference -- LinearAttention (reference ddpm.py:146-165) folded into its to_out conv: out = ctx^T q is linear in q, so to_out(out) is a 1x1
   conv of q with per-sample weights W_eff[b] = W_out blockdiag(ctx_h^T).  mi_linattn_fold_fwd: qkv bf16 [B][n][3 * heads * 32], w_out_bf16 = to_out's
   bf16 weight rows [C][heads * 32] (mi_pack_weights_bf16's wf) -> weff bf16 [B][C * heads * 32] in fragment order; mi_conv1x1_pw_batched: mi_conv1x1_pw with the weights of sample b at
   w_frag_bf16 + b * wbatch elements (x = qkv, d->ldx = 3 * hidden, d->K = hidden: the q channels) */
int mi_linattn_fold_fwd(int B, int n, int heads, const void* qkv_bf16, const void* w_out_bf16, int C, void* weff_bf16, void* stream);
int mi_conv1x1_pw_batched(const MiConvDesc* d, const void* x, const void* w_frag_bf16, int wbatch, const float* bias, const float* residual,
                          void* y, int out_bf16, void* y_bf16, int ldy16, void* stream);
/* round 6: backward of the C -> 3 conv (final_conv.1, reference ddpm.py:235) in one pass over (x, dy): weight gradient and data gradient */
int mi_conv1x1_small_cout_bwd(int M, int C, int Cs, const void* x, int ldx, int x_bf16, const float* dy, int lddy, const float* w,
                              float* dW, void* dx, int lddx, int dx_bf16, int accumulate_dx, void* workspace, size_t ws_bytes, void* stream);
/* round 6, inference: final_conv (reference ddpm.py:232-235) -- the Block's GroupNorm-apply + Mish inside the Conv2d(dim, channels, 1)'s load: x = the
   Block conv's bf16 output, sums = what its epilogue left (mi_conv3x3_pw_gnsums); C = 64 / 128, (C / G) % 16 == 0 */
int mi_conv1x1_small_cout_gn_supported(int C, int Cs, int G);
int mi_conv1x1_small_cout_gn_fwd(int M, int HW, int C, int Cs, const void* x_bf16, int ldx, const void* sums, const float* gamma,
                                 const float* beta, int G, float eps, const float* w, const float* bias, float* y, int ldy, void* stream);
/* round 6: the first ResnetBlock's 3x3 conv and its res_conv (Conv2d(Cin, Cout, 1), reference ddpm.py:134,143) on the same image in one launch */
int mi_conv_small_cin_fwd_dual_supported(int N, int H, int W, int Cin, int Cout, int ldx);
int mi_conv_small_cin_fwd_dual(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w3, const float* bias3,
                               void* y3, int ldy3, int y3_bf16, const float* w1, const float* bias1, float* y1, int ldy1, void* stream);
/* ... that also does a forward's two chores whose consumers are later launches (this is a forward's first launch): zero-fills the pool of GroupNorm
   sums, gathers the denoise step's time-bias rows gather_dst[b][:] = gather_src[gather_idx[b]][:] (int64 indices); either may be null */
int mi_conv_small_cin_fwd_dual_chores(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w3, const float* bias3,
                                    void* y3, int ldy3, int y3_bf16, const float* w1, const float* bias1, float* y1, int ldy1,
                                    void* zero, size_t zero_bytes, const float* gather_src, const void* gather_idx, float* gather_dst,
                                    int gather_row, int gather_n, void* stream);
int mi_conv_small_cin_fwd_io(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                             const float* bias, void* y, int ldy, int y_bf16, void* stream);
int mi_conv_small_cin_wgrad_io(int ks, int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const void* dy,
                               int lddy, int dy_bf16, float* dW, void* workspace, size_t ws_bytes, void* stream);
/* the 3x3 forms without workspace */
int mi_conv3x3_small_cin_fwd(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* w,
                             const float* bias, float* y, int ldy, void* stream);
int mi_conv3x3_small_cin_wgrad(int N, int H, int W, int Cin, int Cout, const float* x, int ldx, const float* dy,
                               int lddy, float* dW, void* stream);
/* Conv2d(C, Cs<=4, 1) (final_conv.1, ddpm.py:236), weights w[C][Cs], C in {32, 64, 128, 256}.  op 0: y = x w + bias
 * (C <= 128); op 1: dx (+)= dy w^T; op 2: dW += x^T dy (a = x, b = dy; C >= 64; workspace of
 * mi_conv_small_wgrad_workspace(4*C) bytes as above). */
int mi_conv1x1_small_cout(int op, int M, int C, int Cs, const float* a, int lda, const float* b, int ldb,
                          const float* w, const float* bias, float* out, int ldo, int accumulate, void* stream);
int mi_conv1x1_small_cout_ws(int op, int M, int C, int Cs, const float* a, int lda, const float* b, int ldb,
                             const float* w, const float* bias, float* out, int ldo, int accumulate,
                             void* workspace, size_t ws_bytes, void* stream);
/* wide_bf16 = 1 (round 4): the C-channel tensor (x of op 0 / 2, dx of op 1: final_conv.0's output and its gradient) is bf16; the
 * Cs-channel side, the weights and the arithmetic stay fp32 */
int mi_conv1x1_small_cout_io(int op, int M, int C, int Cs, const void* a, int lda, const float* b, int ldb,
                             const float* w, const float* bias, void* out, int ldo, int accumulate, int wide_bf16,
                             void* workspace, size_t ws_bytes, void* stream);

/* ---- weight gradient (aten::convolution_backward, weight part) -----------------------
 *   dW[ky][kx][i][j] += sum_{n,y,x} P[n,py,px,i] * Q[n,qy,qx,j]
 * (y,x) runs over the DH x DW grid of the non-gathered operand; the gathered operand is
 * read at (y*stride - pad + ky, x*stride - pad + kx) of its GH x GW grid.
 * gather_i=1: P gathered (Conv2d: P = layer input, Q = grad of output);
 * gather_i=0: Q gathered (ConvTranspose2d: P = layer input, Q = grad of output).
 * P channel i comes from P when i < I1 else from P2 at i-I1.  dW is accumulated with
 * fp32 atomics (split over the pixel axis), so zero it first.
 */
typedef struct MiWgradDesc {
    int N;
    int GH, GW;          /* gathered operand grid */
    int DH, DW;          /* direct operand grid (loop grid) */
    int Ci, Cj;
    int KH, KW, stride, pad;
    int gather_i;
    int mode;
    int I1;
    int ldp, ldp2, ldq;
} MiWgradDesc;

int mi_conv_wgrad(const MiWgradDesc* d, const float* P, const float* P2, const float* Q,
                  float* dW, void* stream);

/* 3x3 / stride 1 / pad 1 Conv2d weight gradient on bf16 MFMA with image-major contraction vectors
 * (needs N % 8 == 0, channel counts % 32 == 0; query first, else use mi_conv_wgrad). */
int mi_conv3x3_wgrad(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW, void* stream);
int mi_conv3x3_wgrad_supported(const MiWgradDesc* d);
/* profiling attribution: wgrad3x3_kernel<nj, KH, io> and the number of k-slices (= partial tiles per output tile) */
int mi_conv3x3_wgrad_tile(const MiWgradDesc* d, int* nj, int* splits);
/* measurement aid for per-kernel event timing: 0 = normal, 1 = launch only the contraction kernel, 2 = only the partial-tile
 * reduce (one call with 1 followed by one with 2 == one normal call).  Process-global; reset to 0 after use. */
int mi_debug_wgrad3x3_phase(int phase);
/* Same, but the k-slices write partial tiles to a caller-provided scratch buffer that a second
 * (deterministic) kernel sums into dW -- 2.4x cheaper than fp32 atomics here.  Size from
 * mi_conv3x3_wgrad_workspace(); a null / too small workspace falls back to atomics. */
size_t mi_conv3x3_wgrad_workspace(const MiWgradDesc* d);
int mi_conv3x3_wgrad_ws(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                        void* workspace, size_t ws_bytes, void* stream);
/* ... and additionally dbias[j] += sum over pixels of Q[., j] (the conv's bias gradient) from the
 * same pass over Q (replaces a separate mi_colsum over the gradient tensor). */
int mi_conv3x3_wgrad_bias(const MiWgradDesc* d, const float* P, const float* P2, const float* Q, float* dW,
                          float* dbias, void* workspace, size_t ws_bytes, void* stream);
/* ... with bf16-stored operands: io bit 0 = P / P2 bf16, bit 1 = Q bf16 (3x3 only) */
int mi_conv3x3_wgrad_io(const MiWgradDesc* d, const void* P, const void* P2, const void* Q, float* dW,
                        float* dbias, void* workspace, size_t ws_bytes, int io, void* stream);

/* 3x3 / stride 1 / pad 1 Conv2d weight gradient (ddpm.py:116) for bf16-STORED operands (P / P2 / Q are bf16 tensors, strides in
 * elements, % 8 == 0): rows go L2 -> LDS by LDS-DMA, MFMA operands come out of LDS through the transposing ds_read_b64_tr_b16, one
 * workgroup accumulates all nine taps of a 64 x 128 (ci x co) tile over its slice of the pixel axis; slices are summed in a fixed
 * order by a second kernel through `workspace` (mi_conv3x3_wgrad_tr_workspace bytes).  dW += result, layout [3][3][Ci][Cj].
 * Needs W in {8,16,32,64}, H % (64/W) == 0, N*H*W % 64 == 0, Ci % 64 == 0, I1 % 64 == 0, Cj % 32 == 0 (query _supported).
 * d->mode = 0 (exact-fp32 mode): P / P2 / Q are fp32 tensors (strides % 4 == 0) and the contraction runs on v_mfma_f32_32x32x2_f32 with
 * plain ds_read_b32 fragments of pixel-major fp32 tiles -- same tiles, k-slices, reduce and batching (one mode per batch). */
int mi_conv3x3_wgrad_tr_supported(const MiWgradDesc* d);
size_t mi_conv3x3_wgrad_tr_workspace(const MiWgradDesc* d);
int mi_conv3x3_wgrad_tr_splits(const MiWgradDesc* d);
int mi_conv3x3_wgrad_tr(const MiWgradDesc* d, const void* P, const void* P2, const void* Q, float* dW,
                        void* workspace, size_t ws_bytes, void* stream);
/* Up to 8 independent layers in ONE launch, each on a share of the workgroups proportional to its MFMA work.  The partial-tile
 * volume of a layer is (its workgroups) x 288 KB, so eight layers side by side write an eighth of what eight full-chip launches
 * write, and layers with at least as many tiles as workgroups accumulate straight into dW (no k-slices, no reduce).  descs: array
 * of n descriptors; P / P2 / Q / dW: arrays of n device pointers (P2 entries may be null); the layers' weight gradients must not
 * alias each other.  Backward defers the Block convs' weight gradients and flushes them through this entry point. */
size_t mi_conv3x3_wgrad_tr_batch_workspace(int n, const MiWgradDesc* descs);
int mi_conv3x3_wgrad_tr_batch(int n, const MiWgradDesc* descs, const void* const* P, const void* const* P2,
                              const void* const* Q, float* const* dW, void* workspace, size_t ws_bytes, void* stream);
/* The 1x1 convolutions' weight gradients (to_qkv / to_out / res_conv, ddpm.py:134,151-152) the same way: dW[Ci][Cj] += X^T dY over
 * all pixels, up to 8 layers per launch on shares of the workgroups proportional to the bytes they stream (these are HBM-bound).
 * P / P2 are bf16 tensors; Q is bf16 (q_is_fp32[i] = 0) or the fp32 residual-stream gradient (1: rows are DMA'd raw and converted
 * in LDS, and dbias[i] (optional) += column sums of Q -- the conv's bias gradient, replacing a separate mi_colsum pass).
 * Needs N*H*W % 64 == 0, Ci % 64 == 0, I1 % 64 == 0, Cj % 32 == 0, ldp % 8 == 0, ldq % 8 (bf16) / % 4 (fp32) == 0.
 * mode 0 (round 4; Unet.compute_mode = "fp32"): P / P2 and Q are fp32 tensors (q_is_fp32 = 1, ldp / ldp2 / ldq % 4 == 0), the products run
 * on v_mfma_f32_32x32x2_f32 -- an fp32 fmaf chain per k-slice, slices added in a fixed order; one mode per batch. */
int mi_conv1x1_wgrad_tr_supported(const MiWgradDesc* d, int q_is_fp32);
size_t mi_conv1x1_wgrad_tr_batch_workspace(int n, const MiWgradDesc* descs, const int* q_is_fp32);
int mi_conv1x1_wgrad_tr_batch(int n, const MiWgradDesc* descs, const int* q_is_fp32, const void* const* P,
                              const void* const* P2, const void* const* Q, float* const* dW, float* const* dbias,
                              void* workspace, size_t ws_bytes, void* stream);
int mi_debug_wgrad1x1_tr_phase(int phase);
int mi_debug_wgrad1x1_tr_blocks(int blocks);
/* measurement aid, as mi_debug_wgrad3x3_phase: 1 = contraction kernel only, 2 = reduce only, 0 = both */
int mi_debug_wgrad_tr_phase(int phase);
/* test aid: plan the k-slices for `blocks` workgroups instead of one per CU (0 = default) */
int mi_debug_wgrad_tr_blocks(int blocks);

/* The stride-2 layers' weight gradients for bf16-stored operands on the same machinery (csrc/wgrad_s2_tr.hip):
 *   Downsample Conv2d(C, C, 3, 2, 1), ddpm.py:70:           descriptor KH = KW = 3, stride 2, pad 1, gather_i = 1, (GH, GW) = X's grid =
 *                                                            2 x (DH, DW) = dY's grid, P = X, Q = dY;
 *   Upsample ConvTranspose2d(C, C, 4, 2, 1), ddpm.py:79:     KH = KW = 4, stride 2, pad 1, gather_i = 0, (GH, GW) = dY's grid =
 *                                                            2 x (DH, DW) = X's grid, P = X, Q = dY.
 * dW[KH][KW][Ci][Cj] += result.  The 2x-resolution tensor is staged as even / odd column planes (LDS-DMA places every 16-byte piece
 * individually), so the stride costs nothing; up to 8 layers per launch.  Needs DW in {8,16,32}, DH % (64/DW) == 0,
 * N*DH*DW % 64 == 0, channels of the 2x tensor % 64 == 0, of the other % 32 == 0, strides % 8 == 0 (query _supported). */
int mi_conv_s2_wgrad_tr_supported(const MiWgradDesc* d);
size_t mi_conv_s2_wgrad_tr_batch_workspace(int n, const MiWgradDesc* descs);
int mi_conv_s2_wgrad_tr_batch(int n, const MiWgradDesc* descs, const void* const* P, const void* const* Q, float* const* dW,
                              void* workspace, size_t ws_bytes, void* stream);
int mi_debug_wgrad_s2_tr_phase(int phase);
/* Exact-fp32 mode (d->mode = 0), round 6: the same two layers' weight gradients (Conv2d(C, C, 3, 2, 1) of Downsample, ConvTranspose2d(C, C, 4,
 * 2, 1) of Upsample; reference src/models/ddpm.py:70,79 -- aten::convolution_backward (weight)) as k x k gathered problems of the exact-fp32
 * 1x1 weight-gradient kernel (v_mfma_f32_32x32x2_f32 = an fp32 fmaf chain), up to eight taps per launch.  P fp32 [N][..][Ci], Q fp32
 * [N][..][Cj], d->gather_i: P lies on the big grid (GH = 2 DH), else Q; dW [KH][KW][Ci][Cj] += ...; workspace from _workspace(). */
int mi_conv_s2_wgrad_f32_supported(const MiWgradDesc* d);
size_t mi_conv_s2_wgrad_f32_workspace(const MiWgradDesc* d);
int mi_conv_s2_wgrad_f32(const MiWgradDesc* d, const float* P, const float* Q, float* dW, void* workspace, size_t ws_bytes, void* stream);
/* measurement aid (tools/cu_hog.py): `blocks` workgroups of 256 threads that stay resident on `stream` for `usec` microseconds -- mode 0
 * an ALU spin, 1 a streaming copy over buf (2 x per_wg_floats floats per workgroup), 2 the same copy with a gradient all-reduce's duty
 * cycle (0.5 ms of every 5 ms) -- to see how a training step behaves while another stream's kernel shares the chip */
int mi_debug_spin(int blocks, int usec, float* buf, size_t per_wg_floats, int mode, void* stream);
/* measurement aid (bench.py's config.sclk_mhz_under_mfma): `blocks` workgroups issue back-to-back bf16 MFMAs for `usec` microseconds;
 * out3 (device, 3 x uint64): shader-clock ticks (s_memtime) and 100 MHz wall-clock ticks workgroup 0 saw across its loop, and a sink word */
int mi_debug_clock_probe(int blocks, int usec, unsigned long long* out3, void* stream);

/* One pass over an fp32 [M][C] tensor: y_bf16 (optional) = bf16(x), colsum (optional)[c] += sum_m x[m][c].  Backward: a residual-
 * stream gradient that the LDS-DMA weight-gradient kernels want bf16-stored and whose column sums are the conv's bias gradient.
 * workspace (optional; mi_f32_to_bf16_colsum_workspace bytes, 16-byte aligned) holds per-workgroup partial sums so that the pass runs
 * on a full grid and a second small pass adds them; without it the sums are added atomically from at most 64 workgroups. */
size_t mi_f32_to_bf16_colsum_workspace(size_t M, int C);
int mi_f32_to_bf16_colsum(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, float* colsum, void* workspace,
                          size_t ws_bytes, void* stream);
/* ... the first pass only: one row of column sums per workgroup into part[*rows][C] (every row written; mi_f32_to_bf16_colsum_workspace
 * bytes), to be added by mi_rowsum_batch together with the other deferred reductions of a backward pass.  *rows = 0: the tensor is too
 * small for partial rows (M < 4096), nothing was done. */
int mi_f32_to_bf16_colsum_part(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, float* part, size_t part_bytes,
                               int* rows, void* stream);
/* out[c] += sum_m x[m*ld + c]  (bias gradients) */
int mi_colsum(int M, int C, const float* x, int ld, float* out, void* stream);
/* dst[c] += sum_{r < rows} src[r*ld + c], c < cols, for up to MI_ROWSUM_MAX items in ONE launch: the second pass of the reductions
 * that leave one partial row per workgroup (mi_chan_layernorm_bwd_part) -- device-scope atomics on one address serialise, so 512
 * workgroups adding to the same C addresses cost ~10 us at the end of every launch; here an address receives rows / 64 atomics. */
#define MI_ROWSUM_MAX 16
typedef struct { const float* src; float* dst; int rows, cols, ld, pad_; } MiRowSum;
int mi_rowsum_batch(int n, const MiRowSum* items, void* stream);

/* ---- GroupNorm(8)+Mish (+time bias, +residual) ---------------------------------------
 * Block's GroupNorm->Mish (ddpm.py:116,62-64), the time-embedding add `h += mlp(t)`
 * (ddpm.py:139-140) and ResnetBlock's `h + res_conv(x)` (ddpm.py:143) in one pass:
 *   y = mish( (x-mean_{n,g}) * rstd_{n,g} * gamma[c] + beta[c] ) + temb[n,c] + residual
 * stats[n][g] = {mean, rstd} is written for backward.  HW*C/G elements per (n,g).
 */
typedef struct MiGnDesc {
    int N, HW, C, G;
    float eps;
    int ldx, ldy, ldr;
} MiGnDesc;

int mi_gn_mish_fwd(const MiGnDesc* d, const float* x, const float* gamma, const float* beta,
                   const float* temb, int ldt, const float* residual, float* y, float* stats,
                   void* stream);
/* dx = grad wrt x.  dgamma/dbeta/dbias (each [C]) are accumulated atomically; dtemb[n*ldt+c]
 * is stored (= sum_hw dout) when non-null; dbias (sum_{n,hw} dx, the producing conv's bias
 * gradient) when non-null.  The residual branch's gradient is dout itself. */
int mi_gn_mish_bwd(const MiGnDesc* d, const float* x, const float* stats, const float* gamma,
                   const float* beta, const float* dout, int lddo, float* dx, int lddx,
                   float* dgamma, float* dbeta, float* dtemb, int ldt, float* dbias,
                   void* stream);

/* bf16 storage variants.  fwd io: bit 0 = x bf16, bit 1 = y bf16.  bwd io: bit 0 = x bf16, bit 1 = dx bf16,
 * bit 2 = dout bf16.  gamma/beta/temb/residual/stats and all parameter gradients stay fp32. */
int mi_gn_mish_fwd_io(const MiGnDesc* d, const void* x, const float* gamma, const float* beta, const float* temb,
                      int ldt, const float* residual, void* y, float* stats, int io, void* stream);
/* mi_gn_mish_fwd_io that also writes y rounded to bf16 into y16 (pixel stride ldy16 elements): the residual stream stays fp32 in y,
 * the next layer's bf16-MFMA kernels read the copy (ddpm.py:116,143: the ResnetBlock output feeds the next block's convs). */
int mi_gn_mish_fwd_dual(const MiGnDesc* d, const void* x, const float* gamma, const float* beta,
                        const float* temb, int ldt, const float* residual, void* y, void* y16, int ldy16,
                        float* stats, int io, void* stream);
/* GroupNorm + Mish (+ time bias) (+ residual) as a STREAMING apply fed by the sums the producing conv's epilogue left (mi_conv3x3_pw_gnsums:
 * sums [N][C / 16][2], 64-bit fixed point): no statistics pass, no (sample, group) workgroups -- a workgroup is a run of pixels of one
 * sample x all channels.  Replaces aten::native_group_norm + Mish + the broadcast / residual adds of reference src/models/ddpm.py:112-120,
 * 139-143 where a tile-kernel conv produced x.  x bf16 [N][HW][C]; y bf16 (y_is_bf16) or fp32, residual fp32 (fp32 y only), y_bf16 an
 * optional bf16 copy of an fp32 y, stats [N][G][2] = {mean, rstd} (optional, what the backward pass reads).  Returns 0 when launched,
 * 1 when the shape is not taken (C / G % 16, C / 8 a power of two <= 256, HW a multiple of the pixels per pass, 16-byte aligned rows):
 * the caller then runs mi_gn_mish_fwd_io. */
int mi_gn_mish_apply_sums(const MiGnDesc* d, const void* x, const void* sums, const float* gamma, const float* beta, const float* temb,
                          int ldt, const float* residual, void* y, int y_is_bf16, void* y_bf16, int ldy16, float* stats, void* stream);
/* statistics-only pass for the fused conv above */
int mi_gn_stats_coef(const MiGnDesc* d, const void* x, const float* gamma, const float* beta, const float* temb, int ldt,
                     float* stats, float* coef, int x_is_bf16, void* stream);
int mi_gn_mish_bwd_io(const MiGnDesc* d, const void* x, const float* stats, const float* gamma, const float* beta,
                      const void* dout, int lddo, void* dx, int lddx, float* dgamma, float* dbeta, float* dtemb,
                      int ldt, float* dbias, int io, void* stream);

/* ---- channel LayerNorm (ddpm.py:85-95; eps added to the std) ---------------------------- */
int mi_chan_layernorm_fwd(int M, int C, const float* x, int ldx, const float* g, const float* b,
                          float eps, float* y, int ldy, void* stream);
int mi_chan_layernorm_bwd(int M, int C, const float* x, int ldx, const float* g, float eps,
                          const float* dy, int lddy, float* dx, int lddx, int accumulate_dx,
                          float* dg, float* db, void* stream);
/* the same with the LayerNorm output (y16) / its gradient (dy16) stored as bf16; x and dx stay fp32 */
int mi_chan_layernorm_fwd_io(int M, int C, const float* x, int ldx, const float* g, const float* b,
                             float eps, void* y, int ldy, int y16, void* stream);
int mi_chan_layernorm_bwd_io(int M, int C, const float* x, int ldx, const float* g, float eps,
                             const void* dy, int lddy, float* dx, int lddx, int accumulate_dx,
                             float* dg, float* db, int dy16, void* stream);
/* the same with the parameter gradients left as partial rows: part[r][2 C] = (dg | db) sums of workgroup r for
 * r < mi_chan_layernorm_bwd_part_rows(M, C) (every row is written); mi_rowsum_batch adds them into dg / db */
int mi_chan_layernorm_bwd_part_rows(int M, int C);
int mi_chan_layernorm_bwd_part(int M, int C, const float* x, int ldx, const float* g, float eps,
                               const void* dy, int lddy, float* dx, int lddx, int accumulate_dx,
                               float* part, int dy16, void* stream);

/* ---- LinearAttention core (ddpm.py:157-165), heads x 32 channels -------------------------
 * qkv[b][p][3*heads*32] (q | k | v, head-major inside each), out[b][p][heads*32].
 * ctx[b][h][32][32] and kstat[b][h][32][2] = {max, sum exp} are saved for backward. */
int mi_linattn_fwd(int B, int n, int heads, const float* qkv, float* out, float* ctx,
                   float* kstat, void* stream);
int mi_linattn_bwd(int B, int n, int heads, const float* qkv, const float* ctx,
                   const float* kstat, const float* dout, float* dqkv, void* stream);
/* bf16 storage of the attention-internal activations: b16 != 0 -> qkv and out (forward), qkv, dout and dqkv
 * (backward) are bf16 tensors; ctx, kstat and all arithmetic stay fp32. */
int mi_linattn_fwd_io(int B, int n, int heads, const void* qkv, void* out, float* ctx, float* kstat, int b16,
                      void* stream);
int mi_linattn_bwd_io(int B, int n, int heads, const void* qkv, const float* ctx, const float* kstat,
                      const void* dout, void* dqkv, int b16, void* stream);
/* ... with a scratch buffer (mi_linattn_workspace(B, n, heads) bytes, 16-byte aligned; 0 = not needed): images with many pixels per
 * (batch, head) are cut into pixel slices, one workgroup each -- three launches forward (slice max, slice exp / outer product,
 * combine + out), two backward (slice dctx, combine + per-pixel gradients) -- so that every CU holds several workgroups' loads in
 * flight instead of one workgroup walking 1024-4096 pixels.  Partial results are combined in a fixed order (deterministic).  A
 * null / too small workspace runs one workgroup per (batch, head). */
size_t mi_linattn_workspace(int B, int n, int heads);
int mi_linattn_fwd_ws(int B, int n, int heads, const void* qkv, void* out, float* ctx, float* kstat, int b16,
                      void* workspace, size_t ws_bytes, void* stream);
int mi_linattn_bwd_ws(int B, int n, int heads, const void* qkv, const float* ctx, const float* kstat,
                      const void* dout, void* dqkv, int b16, void* workspace, size_t ws_bytes, void* stream);

/* ---- small exact-fp32 GEMM: nn.Linear of the time-embedding MLP (ddpm.py:126-130,186-193), forward, input
 * gradient and weight gradient ---------------------------------------------------------------------
 *   C[i][j] (+)= bias[j] + sum_k opA(i,k) * opB(k,j),   opA = ta ? A[k*lda+i] : A[i*lda+k],  opB = tb ? B[j*ldb+k] : B[k*ldb+j]
 * forward y = x W^T + b: ta 0, tb 1;  dx = dy W: ta 0, tb 0;  dW += dy^T x: ta 1, tb 0, accumulate 1.
 * Needs K % 32 == 0, ld % 4 == 0, 16-byte aligned A / B (mi_small_gemm_supported); with allow_split long contractions are
 * split and combined with fp32 atomics (used for the gradients only: the forward stays bit-reproducible). */
int mi_small_gemm(int ta, int tb, int I, int J, int K, const float* A, int lda, const float* B, int ldb,
                  const float* bias, float* C, int ldc, int accumulate, int allow_split, void* stream);
int mi_small_gemm_supported(int ta, int tb, int I, int J, int K, int lda, int ldb);

/* ---- VQ-VAE codebook step (reference src/models/vqvae.py:24-43) ---------------
 * Forward: for each of the M latent rows z[m][0..D) (row stride ldz) the nearest of the K codebook rows
 * (torch.cdist + argmin: squared distances ||z||^2 + ||e||^2 - 2 z.e on the exact-fp32 matrix cores, lowest
 * index on ties), idx[m], the gathered rows zq[m][0..D), and loss_partial[b] = sum over workgroup b's rows of
 * ||z - zq||^2 (mi_vq_partials(M) floats; vq_loss = sum / (M*D), commit_loss = weight * the same).  D % 4 == 0, D <= 128.
 * Backward of g_vq * mean((sg(z) - q)^2) + g_commit * mean((z - sg(q))^2):
 *   dz (+)= g_commit * 2 (z - q) / (M D)   (dz may be null),   dcodebook[idx[m]] += g_vq * 2 (q - z_m) / (M D);
 * g_dev (nullable) points at two device floats multiplied into (g_vq, g_commit): the upstream loss gradients without a
 * host round trip.  mi_vq_scatter_rows: table[idx[m]] += src[m] -- what the gather `embedding[z_index]` (vqvae.py:37)
 * passes back to the codebook when quant_z itself is differentiated. */
int mi_vq_partials(int M);
int mi_vq_nearest_fwd(int M, int D, int K, const float* z, int ldz, const float* codebook, int* idx, float* zq, int ldq,
                      float* loss_partial, void* stream);
int mi_vq_bwd(int M, int D, int K, const float* z, int ldz, const float* codebook, const int* idx, float g_vq, float g_commit,
              const float* g_dev, float* dz, int lddz, int accumulate_dz, float* dcodebook, void* stream);
int mi_vq_scatter_rows(int M, int D, int K, const float* src, int ld, const int* idx, float* table, void* stream);

/* ---- WGAN-GP operators (reference src/models/wgan_gp.py:62-107, src/networks/basic.py:9-40) -------------
 * Sample norm = nn.GroupNorm(1, C) (`norm_type="layer"`, forced at wgan_gp.py:30-31) on dense NHWC tensors [N][P][C],
 * C % 4 == 0 and 2048 % C == 0; stats = [N][2] (mean, rstd).
 *   fwd : y = gamma (x - mean) rstd + beta
 *   bwd : dx = dL/dx (+ extra_dx when non-null; dx may alias dy); dgamma, dbeta accumulated (nullable)
 *   bwd2: the backward of bwd, for the gradient penalty's create_graph=True (wgan_gp.py:89-96): given the adjoint u of dx,
 *         adj_dy = d<u,dx>/d dy, adj_x = d<u,dx>/d x, dgamma += d<u,dx>/d gamma. */
int mi_sample_norm_supported(int N, int P, int C);
int mi_sample_norm_fwd(int N, int P, int C, const float* x, const float* gamma, const float* beta, float* y, float* stats, float eps,
                       void* stream);
int mi_sample_norm_bwd(int N, int P, int C, const float* x, const float* stats, const float* gamma, const float* dy, const float* extra_dx,
                       float* dx, float* dgamma, float* dbeta, void* stream);
int mi_sample_norm_bwd2(int N, int P, int C, const float* x, const float* stats, const float* gamma, const float* dy, const float* u,
                        float* adj_dy, float* adj_x, float* dgamma, void* stream);
/* LeakyReLU(slope) / Tanh on dense vectors (n % 4 == 0); backward from the OUTPUT y; in-place allowed (y == x, dx == dy). */
int mi_leaky_relu_fwd(size_t n, const float* x, float* y, float slope, void* stream);
int mi_leaky_relu_bwd(size_t n, const float* y, const float* dy, float* dx, float slope, void* stream);
int mi_tanh_fwd(size_t n, const float* x, float* y, void* stream);
int mi_tanh_bwd(size_t n, const float* y, const float* dy, float* dx, void* stream);
/* out[s][i] = e[s] a[s][i] + (1 - e[s]) b[s][i], s < N, i < per (wgan_gp.py:84-87) */
int mi_lerp_rows(int N, size_t per, const float* a, const float* b, const float* e, float* out, void* stream);
/* gradient penalty (wgan_gp.py:95-97): *penalty += mean_s (||g_s||_2 - 1)^2 (nullable) and, when u is non-null,
 * u_s = scale * (scale_dev ? *scale_dev : 1) * d penalty / d g_s. */
int mi_gp_penalty(int N, size_t per, const float* g, float* penalty, float* u, float scale, const float* scale_dev, void* stream);

/* ---- VAE operators (BASELINE cfg 1; reference src/models/vae.py:46-72, src/networks/basic.py:147-204, src/utils/losses.py:30-32) ----
 * nn.BatchNorm2d on a dense NHWC tensor of M = N*H*W rows and C channels (C % 4 == 0, C/4 a power of two <= 256).
 * training != 0: batch statistics (left in mean / rstd for the backward), running statistics updated like torch (momentum,
 * unbiased variance) when non-null; training == 0: mean / rstd come from the running statistics.
 * ws: mi_batchnorm_workspace(C) bytes, zero on entry, left zero on exit.  bwd: dx may alias dy, dgamma / dbeta accumulated. */
int mi_batchnorm_workspace(int C);
int mi_batchnorm_fwd(int M, int C, const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                     float* running_mean, float* running_var, float momentum, float eps, int training, void* ws, void* stream);
int mi_batchnorm_bwd(int M, int C, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dy,
                     float* dx, float* dgamma, float* dbeta, void* ws, void* stream);
/* latent block: h = [mu | log_sigma] (rows of ldh >= 2L floats), z = mu + exp(log_sigma) * eps (vae.py:46-55, eps drawn by the
 * caller), *kld += mean over rows of -0.5 sum(1 + 2 log_sigma - mu^2 - exp(2 log_sigma)) (losses.py:30-32; nullable).
 * bwd: dh = dL/dh from dz = dL/dz and g_kld * (g_dev ? *g_dev : 1) = dL/d kld. */
int mi_vae_latent_fwd(int N, int L, const float* h, int ldh, const float* eps, float* z, float* kld, void* stream);
int mi_vae_latent_bwd(int N, int L, const float* h, int ldh, const float* eps, const float* dz, float g_kld, const float* g_dev,
                      float* dh, int lddh, void* stream);

/* ---- small element-wise pieces ---------------------------------------------------------------- */
/* SinusoidalPosEmb (ddpm.py:52-59): out[b][dim] = [sin(t f_j) | cos(t f_j)] */
int mi_time_embed(int B, int dim, const int64_t* t, float* out, void* stream);
/* Mish on a flat vector (ddpm.py:62-64) and its backward */
int mi_mish_fwd(size_t n, const float* x, float* y, void* stream);
int mi_mish_bwd(size_t n, const float* x, const float* dy, float* dx, void* stream);
/* ReLU on a flat (dense) vector: y = max(x, 0), y may alias x (nn.ReLU(True), src/networks/vqvae.py:16-20,48,76);
 * backward from the OUTPUT: dx (+)= y > 0 ? dy : 0, dx may alias dy.  16-byte aligned pointers. */
int mi_relu_fwd(size_t n, const float* x, float* y, void* stream);
int mi_relu_bwd(size_t n, const float* y, const float* dy, float* dx, int accumulate, void* stream);
/* NCHW <-> NHWC(ld) */
int mi_nchw_to_nhwc(int B, int C, int HW, const float* x, float* y, int ld, void* stream);
int mi_nhwc_to_nchw(int B, int C, int HW, const float* x, int ld, float* y, void* stream);
/* q_sample (ddpm.py:433-444): x_t = a[t] x0 + b[t] eps ; NCHW in, NHWC(ld) out (+NCHW copy if non-null) */
int mi_q_sample(int B, int C, int HW, const float* x0, const float* noise, const int64_t* t,
                const float* sqrt_ac, const float* sqrt_1mac, float* xt_nhwc, int ld,
                float* xt_nchw, void* stream);
/* loss (ddpm.py:453-456): loss_type 0 = L1 mean, 1 = L2 mean.  pred NHWC(ld), target NCHW.
 * *loss is accumulated (zero it first); dpred (NHWC, may be null) = d loss / d pred * gscale */
int mi_eps_loss(int B, int C, int HW, const float* pred, int ld, const float* target, int loss_type,
                float* loss, float* dpred, float gscale, void* stream);
/* p_sample posterior update (ddpm.py:359-397), x and z NCHW, eps_hat NHWC(ld);
 * writes x_{t-1} NCHW and (if non-null) its NHWC(ldo) copy for the next UNet call. */
int mi_p_sample_update(int B, int C, int HW, const float* x, const float* eps_hat, int ld,
                       const float* z, const int64_t* t, const float* sqrt_recip_ac,
                       const float* sqrt_recipm1_ac, const float* coef1, const float* coef2,
                       const float* logvar, int clip, float* x_prev, float* x_prev_nhwc, int ldo,
                       void* stream);
/* Adam (torch.optim.Adam defaults, ddpm.py:507-511) over one flat buffer.
 * bc1 = 1-b1^step, bc2 = 1-b2^step; grad is multiplied by gscale first (DDP average). */
int mi_adam_step(size_t n, float* p, const float* g, float* m, float* v, float lr, float b1,
                 float b2, float eps, float bc1, float bc2, float gscale, void* stream);
/* The same update with its step count and learning rate in device memory (state[0] = steps taken, stored as the bits of a
 * uint32; state[1] = lr as float), so that a captured hipGraph of forward + backward + optimizer step replays correctly:
 * mi_adam_tick increments state[0] (call it once per optimizer step, before the update of every buffer of that step); the bias
 * corrections 1 - b^state[0] are evaluated in double from the double betas, as the host does for mi_adam_step. */
int mi_adam_tick(float* state, void* stream);
int mi_adam_step_dev(size_t n, float* p, const float* g, float* m, float* v, const float* state, double b1, double b2,
                     float eps, float gscale, void* stream);
/* Device-resident data path: gather B images out of a uint8 dataset [N][H][W][C] that lives in HBM and apply the reference's
 * transform chain (ToTensor, optional per-sample horizontal flip, optional Normalize(0.5, 0.5); reference
 * src/datamodules/base.py:37-71) in torch's own fp32 operation order (u/255, then (v-0.5)/0.5): NCHW fp32 out.
 * idx: int64[B] rows of the dataset; flip: uint8[B] or null. */
int mi_u8_gather_normalize(int B, int C, int H, int W, const uint8_t* data, const int64_t* idx, const uint8_t* flip,
                           int normalize, float* out_nchw, void* stream);
/* fp32 [M][C] (row stride ldx) -> bf16 [M][C] (row stride ldy), round-to-nearest-even: the bf16 copy of a residual-stream tensor
 * that the next layer's conv / weight-gradient kernels read as their MFMA operand (C, ldx, ldy % 4 == 0). */
int mi_f32_to_bf16(size_t M, int C, const float* x, int ldx, void* y_bf16, int ldy, void* stream);
/* y = a*x + (accumulate ? y : 0) */
int mi_axpby(size_t n, float a, const float* x, int accumulate, float* y, void* stream);
/* same on M rows of C channels with row strides (gradient accumulation into channel slices) */
int mi_axpby2d(int M, int C, float a, const float* x, int ldx, int accumulate, float* y, int ldy, void* stream);
/* out[b][0..C) = table[idx[b]][0..C): per-sample rows of a precomputed table (the sampler's time-bias table) */
int mi_gather_rows(int B, int C, const float* table, const int64_t* idx, float* out, void* stream);
/* x[m*ld + c] *= *scalar (scalar lives on the device: autograd's incoming d(loss), no host sync) */
int mi_scale_by_device_scalar(int M, int C, float* x, int ld, const float* scalar, void* stream);

#ifdef __cplusplus
}
#endif
#endif

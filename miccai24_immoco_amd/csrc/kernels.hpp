// Internal launch API shared by the op-level C-ABI wrappers and the fused solver.
#pragma once
#include "common.hpp"

namespace immoco {

// Per-axis coordinate lattice: point p has coordinate axis[d][(p / stride[d]) % n[d]] in dim d.
struct Lattice {
  const float* axis[3];
  int32_t n[3];
  int32_t stride[3];
};

// hashgrid.hip
// half_out: the encoding is written as packed halves, one 4-byte word per (point, level); ps / ls in 4-byte words
int launch_hashgrid_fwd(const Levels& lv, const float* coords, const Lattice* lat, int64_t n,
                        const float* table, float* enc, int64_t ps, int64_t ls, hipStream_t st, bool half_out = false);
int launch_hashgrid_fwd_half(const Levels& lv, const Lattice& lat, int64_t n, const void* table_half2, float* enc,
                             int64_t ps, int64_t ls, hipStream_t st, bool half_out = false);

int launch_f32_to_half(const float* in, void* out_half, int64_t n, hipStream_t st);
int launch_hashgrid_bwd(const Levels& lv, const float* coords, const Lattice* lat, int64_t n,
                        const float* denc, int64_t ps, int64_t ls, float* dtable, hipStream_t st);
int launch_init_uniform(float* out, int64_t n, uint32_t seed, uint32_t stream_id, float lo, float hi,
                        hipStream_t st);

// mlp.hip
int check_mlp_cfg(const immoco_mlp_cfg* cfg);
int launch_mlp_fwd(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                   const float* w1, const float* w2, float* out, hipStream_t st);
int launch_mlp_bwd(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                   const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                   hipStream_t st, int64_t dout_plane = 0);  // dout_plane != 0: dout is planar [2][plane]

// mlp_mfma.hip — matrix-core implementation (3-term bf16 split, fp32-equivalent accuracy)
int launch_mlp_fwd_mfma(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                        const float* w1, const float* w2, float* out, hipStream_t st);
int launch_mlp_bwd_mfma(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                        const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                        hipStream_t st, int64_t dout_plane = 0);

// split backward of the 256-wide net (mlp_mfma.hip): d enc (out of place) and dW1 / dW2 as two kernels that fit beside
// the motion grid's encode backward
int launch_mlp_bwd_denc(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1,
                        const float* w2, const float* dout, float* din, hipStream_t st, int64_t dout_plane = 0,
                        const float* dout2 = nullptr);   // dout2: a second planar addend of dout (same layout)
int launch_mlp_bwd_dw(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1,
                      const float* w2, const float* dout, float* dw1, float* dw2, hipStream_t st, int64_t dout_plane = 0,
                      const float* dout2 = nullptr);

// mlp_f16.hip — tiny-cuda-nn's operand precision: fp16 operands, fp32 accumulation (v_mfma_f32_32x32x16_f16);
// the backward scales dout by `scale` (tcnn's loss scale, 128) before rounding it to fp16
// enc_half: the encoding (and dL/d enc, which stays scaled by `scale`) is stored as packed halves, one 4-byte word per
// (point, level); ps / ls are then in 4-byte words
int launch_mlp_fwd_f16(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                       const float* w1, const float* w2, float* out, hipStream_t st, bool enc_half = false);
int launch_mlp_bwd_f16(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                       const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                       hipStream_t st, int64_t dout_plane, float scale, bool enc_half = false);

// the 256-wide net's backward as two kernels (part 1: d enc, out of place; part 2: dW1 / dW2) that fit beside the motion
// grid's encode backward
int launch_mlp_bwd_f16_split(const immoco_mlp_cfg& cfg, int part, const float* in, int64_t ps, int64_t ls, int64_t n,
                             const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                             hipStream_t st, int64_t dout_plane, float scale, bool enc_half = false,
                             const float* dout2 = nullptr);

// mlp_bf16x2.hip - (nearly) fp32 accuracy on the 16-bit matrix cores: every operand split into two bf16 terms, three
// MFMAs per product, fp32 everywhere else (relative product error <= 2^-16.5)
int launch_mlp_fwd_bf16x2(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                          const float* w1, const float* w2, float* out, hipStream_t st);
int launch_mlp_bwd_bf16x2(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                          const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                          hipStream_t st, int64_t dout_plane);

// warp.hip
int launch_warp_fwd(const float* image, const float* grids, int nM, int H, int W, float* out, hipStream_t st);
int launch_warp_bwd(const float* image, const float* grids, const float* dout, int nM, int H, int W,
                    float* dimage, float* dgrids, hipStream_t st);
// solver-fused variants: motion-MLP output o [nM*H*W,2] -> t = tanh(o), grid = t + identity,
// warped image * sign(r,c) -> fftbuf slots 1..nM; and its backward.
int launch_motion_warp_fwd(const float* image, const float* o, const float* xs, const float* ys, int nM,
                           int H, int W, float* t_out, float* fft_slots, hipStream_t st);
int launch_motion_warp_bwd(const float* image, const float* t, const float* xs, const float* ys,
                           const float* adj_slots, int nM, int H, int W, float* dimage, float* d_o,
                           hipStream_t st);

// pruned path (warp.hip): the motion images' row DFT restricted to the k-space columns their group owns, fused into
// the warp (forward) and into the warp backward (adjoint); zt is the transposed k-space [W][H]
int launch_motion_warp_dft(const float* image, const float* o, const float* xs, const float* ys, int nM, int H, int W,
                           const float* tw, const int32_t* cols, const int32_t* off, float* t_out, float* zt,
                           hipStream_t st);
int launch_motion_warp_bwd_dft(const float* image, const float* t, const float* xs, const float* ys, const float* zt_adj,
                               const float* tw, const int32_t* cols, const int32_t* off, int nM, int H, int W,
                               float* dimage_planar, float* d_o, hipStream_t st);

// kspace.hip
int fft2c(const float* in, float* out, int batch, int H, int W, int mode, hipStream_t st);
int fft_exec_inplace(float* buf, int batch, int H, int W, bool inverse, hipStream_t st);  // raw, no shifts
int fft_fwd_to_transposed(float* in, float* out_t, int B, int H, int W, hipStream_t st);      // -> [W][B][H]
int fft_adj_from_transposed(float* in_t, float* out, int B, int H, int W, hipStream_t st);  // [W][B][H] ->
int launch_transpose_c64(const float* in, float* out, int H, int W, hipStream_t st);
// pruned path (one image): rows [H][W] -> [W][H]; columns in place on [W][H]; rows adjoint [W][H] -> [H][W]
int fft_rows_fwd_to_t(float* in, float* out_t, int H, int W, hipStream_t st);
int fft_cols_inplace_t(float* zt, int H, int W, bool inverse, hipStream_t st);
int fft_rows_adj_from_t(float* in_t, float* out, int H, int W, hipStream_t st);
// cols[off[g] .. off[g+1]) = k-space columns of group g (0 = unwarped image), g = 0..nM; off has nM + 2 entries
int launch_build_col_lists(const int32_t* col_group, int nM, int W, int32_t* cols, int32_t* off, hipStream_t st);
int launch_keep_group0_cols(const float* in_t, const int32_t* col_group, int nM, int H, int W, float* out_t, hipStream_t st);
// dimage = dwarp + sign*adjoint slot 0 + lambda*dGE; dwarp (planar, filled by the warp backward) cleared
int launch_image_grad_init_after_warp(const float* image, const float* adj_slot0, int H, int W, const float* lambda_sched,
                                      const int32_t* iter_dev, float* loss_hist, float* dimage, float* dwarp, hipStream_t st);
int launch_select_dc_seed_t(float* fft_t, const int32_t* col_group, const float* kin_t, int nM, int H, int W,
                            float* kout_t, float* loss_hist, const int32_t* iter_dev, hipStream_t st);
int launch_kspace_select(const float* kall, const int32_t* col_group, int nM, int H, int W, float* kout,
                         hipStream_t st);
int launch_dc_loss(const float* k, const float* kin, int H, int W, float* loss, float* dk, hipStream_t st);
int launch_ge_loss(const float* image, int H, int W, float weight, const float* weight_dev, float* loss,
                   float* dimage, hipStream_t st);
int launch_normalize(const float* k, int64_t n, float target, float* out, float* scale_out, hipStream_t st);
// solver-fused: image (H,W) c64 -> sign-modulated copy into fft slot 0
int launch_image_to_slot(const float* image, int H, int W, float* slot0, hipStream_t st);
// solver-fused select + DC loss + adjoint seed written back into the fft buffer
// (loss goes to loss_hist[*iter_dev] when loss_hist != NULL)
int launch_select_dc_seed(float* fftbuf, const int32_t* col_group, const float* kin, int nM, int H, int W,
                          float* kout, float* loss_hist, const int32_t* iter_dev, hipStream_t st);
// solver-fused: dimage = sign*adjoint slot 0 + lambda*dGE ; loss += lambda*GE, lambda = lambda_sched[*iter_dev]
int launch_image_grad_init(const float* image, const float* adj_slot0, int H, int W, const float* lambda_sched,
                           const int32_t* iter_dev, float* loss_hist, float* dimage, hipStream_t st);

// optim.hip
int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float bc2_sqrt,
                float beta1, float beta2, float eps, hipStream_t st);
// device-scheduled variant: scalars read from sched[2*it], it = *iter_dev; g zeroed after use
// g: n_gparts partial gradient buffers, g_stride floats apart; their sum is the gradient
// gradients with index >= zero_limit are NOT cleared (their producer overwrites them every iteration)
// shadow != NULL: the updated parameters with index >= shadow_begin are also written as fp16 to
// shadow[i - shadow_begin] (fp16 table shadow, fp32 master)
int launch_adam_sched(float* p, float* g, int n_gparts, int64_t g_stride, float* m, float* v, int64_t n,
                      int64_t zero_limit, const float* sched, const int32_t* iter_dev, float beta1, float beta2,
                      float eps, hipStream_t st, void* shadow = nullptr, int64_t shadow_begin = 0);
// touched-blocks variant (transposed-index plans): the first n_w parameters (MLP weights) and the listed
// slot blocks of the table behind them are updated; everything else has g = m = v = 0 for ever, i.e. an
// exactly-zero Adam update, and is skipped.  Gradients of the weights and of shared blocks
// are cleared after use.
int launch_adam_blocks(float* p, float* g, int n_gparts, int64_t g_stride, float* m, float* v, int64_t n_w,
                       const uint2* blocks, uint32_t n_blocks, const float* sched, const int32_t* iter_dev,
                       float beta1, float beta2, float eps, hipStream_t st, void* shadow = nullptr,
                       int iter_off = 0);  // schedule index = *iter_dev + iter_off

// csr.hip — atomic-free hash-grid backward for fixed lattices
struct CsrPlan;
int csr_plan_build(const Levels& lv, int nM, int H, int W, const float* const* axes, const int32_t* axn,
                   int n_parts, int n_tables, CsrPlan** out, hipStream_t st);
int csr_auto_parts(int64_t n_points, int bytes_per_point = 8);
int csr_plan_tables(const CsrPlan* p);
void csr_plan_free(CsrPlan* p);
int64_t csr_plan_bytes(const CsrPlan* p);
int64_t csr_plan_entries(const CsrPlan* p);
int csr_plan_parts(const CsrPlan* p);
// slot blocks that receive gradients: {first slot, n slots | shared << 31} (device array); shared blocks are
// flushed with atomics and have to be cleared by the consumer, unlisted blocks never get a gradient
const uint2* csr_plan_touched(const CsrPlan* p, uint32_t* n);
// denc_half: dL/d enc is stored as packed halves (4-byte words) scaled by 1 / out_scale
int launch_csr_bwd(const CsrPlan* pl, const float* denc_level_major, float* dtable, int64_t part_stride,
                   int zeroed, hipStream_t st, bool denc_half = false, float out_scale = 1.f);

// masks.hip
int launch_extract_groups(const uint8_t* lines, int n, int32_t* col_group, int32_t* n_groups, hipStream_t st);

}  // namespace immoco

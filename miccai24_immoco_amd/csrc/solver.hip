// Fused per-slice solver: the whole of imcoco_motion_correction's hot loop
// (reference src/models/immoco.py:116-206) as a fixed sequence of HIP kernels +
// two batched rocFFT executions per Adam iteration, captured once into a
// hipGraph and replayed `iters` times.  Everything iteration-dependent (Adam
// bias corrections, the GE weight lambda_j of immoco.py:180-181, the loss slot)
// is read on the device from small schedule arrays indexed by a device-side
// iteration counter, so the captured graph is iteration-invariant and the host
// never synchronises inside the loop.
#include <math.h>

#include <functional>
#include <string>
#include <vector>

#include "kernels.hpp"

namespace immoco {

__global__ void tick_kernel(int32_t* it) { *it += 1; }

// tiny-cuda-nn's torch binding scales dL/dout by 128 before the fp16 backward; cfg.mlp_fp16 does too
constexpr float TCNN_LOSS_SCALE = 128.f;

// branch 0: serial section on the main stream; branch 1 / 2: two independent chains that run
// concurrently (main / side stream) between a fork and the next serial step (join).  The image-INR
// kernels are small grids (102 400 points) that cannot fill 256 CUs on their own; they overlap with
// the motion-INR chain.
struct Step {
  const char* name;
  std::function<int(hipStream_t)> run;
  int branch = 0;
  int group = 2;  // 0 image forward, 1 motion forward, 2 serial middle (.. motion MLP backward), 3 motion encode
                  // backward, 4 image backward (MLP, encode, Adam), 5 tick, 6 motion Adam
  int signal = -1;  // forked execution: record cross-branch event ev_x[signal] after this step on its stream ...
  int wait = -1;    // ... make this step's stream wait for ev_x[wait] first (a dependency between the two branches)
};

}  // namespace immoco

using namespace immoco;

struct immoco_solver {
  immoco_solver_cfg cfg;
  Levels lv_img, lv_mot;
  int64_t P = 0, NP = 0;
  int64_t n_w_img = 0, n_w_mot = 0;  // MLP weights (W1+W2)
  int64_t n_params_img = 0, n_params_mot = 0;
  // device workspace
  float *enc_img = nullptr, *enc_mot = nullptr, *image = nullptr, *o_mot = nullptr, *t_mot = nullptr;
  float* denc_img = nullptr;   // dL/d enc of the image INR (split wide-MLP backward: out of place, exact fp32 only)
  // pruned path (warp.hip): twiddles e^{-2 pi i k / W}, the column lists of the line masks, the warp backward's share of dL/dimage
  bool pruned = false;
  float *tw = nullptr, *dimage_w = nullptr;
  int32_t *pr_cols = nullptr, *pr_off = nullptr;
  float *fftbuf = nullptr, *dimage = nullptr, *grad_img = nullptr, *grad_mot = nullptr, *kout = nullptr;
  float *fft_t = nullptr, *kin_t = nullptr;  // transposed k-space [W][nM+1][H]; measured k-space [W][H] (kout too)
  float *sched = nullptr, *lambda_dev = nullptr;
  float *xs = nullptr, *ys = nullptr, *ms = nullptr;  // solver-owned copies of the lattices
  uint16_t *shadow_img = nullptr, *shadow_mot = nullptr;  // fp16 shadows of the tables (cfg.table_fp16)
  CsrPlan *plan_img = nullptr, *plan_mot = nullptr;
  bool lattice_set = false;
  int mot_parts = 1;            // point-range parts of the motion grid's transposed index
  int mot_tables = 1;           // partial gradient tables of the motion INR (= parts per launch, <= 8)
  int64_t mot_gstride = 0;      // floats between them (n_params_mot rounded up to 4)
  int32_t* iter_dev = nullptr;
  int32_t sched_cap = 0;
  int64_t bytes = 0;
  hipStream_t stream = nullptr, side = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;
  hipEvent_t ev_fj[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_x[2] = {nullptr, nullptr};   // cross-branch dependencies inside a fork (Step::signal / Step::wait)
  // timing markers around the dominant kernel (motion_encode_bwd) INSIDE the replayed graph, so that
  // bench.py's roofline figure is measured under the same conditions as the timed run
  hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr;
  bool marked = false;
  // graph cache: valid while the captured pointers stay the same
  hipGraphExec_t gexec = nullptr;    // classic order, or the FIRST iteration of the pipelined order
  hipGraphExec_t gexec2 = nullptr;   // steady-state iteration of the pipelined order
  std::vector<const void*> gkey;
  int graph_active = 0;
  // phase timing of the last profile call
  std::vector<std::string> phase_names;
  std::vector<float> phase_ms;
  // paired batch mode (cfg.batch_pair): events that order the two slices' gather kernels, the pair's graphs
  hipEvent_t ev_pair[12] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  hipGraphExec_t pg1 = nullptr, pgk = nullptr;   // one / GK double-iterations
  std::vector<const void*> pkey;
  // batch lanes (immoco_solver_solve_batch): further workspaces that share this solver's lattices and plans
  immoco_solver* parent = nullptr;
  std::vector<immoco_solver*> lanes;
  // schedule arrays currently on the device (re-uploading them needs a host sync of the lane's stream)
  std::vector<float> sched_host;
  // graph executables replaced while their launches may still be queued: destroyed once the event is done
  std::vector<std::pair<hipGraphExec_t, hipEvent_t>> retired;
};

namespace {

// Exact-fp32 MLPs, 256-wide image net: its backward runs as two kernels that fit BESIDE the motion grid's encode
// backward (mlp_mfma.hip: mlp_bwd_denc_kernel / mlp_bwd_dw_kernel) instead of one that needs a CU to itself.
// (mlp_fp16 = 1: the same split of mlp_f16.hip's kernel; the bf16x2 mode keeps the fused kernel.)
// IMMOCO_SPLIT_BWD=0 (diagnostics build only): the fused kernels of rounds 1-3.
bool split_image_bwd(const immoco_solver_cfg& c) {
  static const bool off = [] { const char* e = immoco_diag_env("IMMOCO_SPLIT_BWD"); return e && atoi(e) == 0; }();
  return !off && (c.mlp_fp16 == 0 || c.mlp_fp16 == 1) && c.image_mlp.n_hidden == 256 && !c.atomic_scatter;
}

// Pruned path: the motion images' row transforms as direct DFTs of their own k-space columns, fused into the warp
// kernels (warp.hip).  IMMOCO_PRUNED=0 (diagnostics build only) selects the full batched transforms of rounds 1-3.
bool use_pruned(const immoco_solver_cfg& c) {
  static const bool off = [] { const char* e = immoco_diag_env("IMMOCO_PRUNED"); return e && atoi(e) == 0; }();
  return !off && c.nM > 0 && c.nM <= 254 && c.W <= 4096;
}

template <typename T>
int dev_alloc(immoco_solver* s, T** p, int64_t count) {
  IMMOCO_CHECK_HIP(hipMalloc((void**)p, (size_t)count * sizeof(T)));
  s->bytes += count * (int64_t)sizeof(T);
  return IMMOCO_OK;
}

struct Bind {
  const float* kin;
  const int32_t* col_group;
  float *p_img, *p_mot, *a_img, *a_mot;
  float* loss_hist;
};

// The per-iteration kernel sequence (immoco.py:166-175).
std::vector<Step> build_steps(immoco_solver* s, const Bind& b, bool backward) {
  const immoco_solver_cfg& c = s->cfg;
  const int H = c.H, W = c.W, nM = c.nM;
  const int64_t P = s->P, NP = s->NP;
  Lattice li{}, lm{};
  // image INR: identy_grid.view(-1,2) = (x = xs[col], y = ys[row])  (immoco.py:72-76,85)
  li.axis[0] = s->xs; li.n[0] = W; li.stride[0] = 1;
  li.axis[1] = s->ys; li.n[1] = H; li.stride[1] = W;
  li.axis[2] = s->xs; li.n[2] = 1; li.stride[2] = 1;
  // motion INR: make_grids((nM,H,W)) = (m, row, col)  (immoco.py:48-53,78-80,93)
  lm.axis[0] = s->ms; lm.n[0] = nM; lm.stride[0] = H * W;
  lm.axis[1] = s->ys; lm.n[1] = H; lm.stride[1] = W;
  lm.axis[2] = s->xs; lm.n[2] = W; lm.stride[2] = 1;
  float* w1i = b.p_img;
  float* w2i = w1i + (int64_t)c.image_mlp.n_hidden * c.image_mlp.n_in;
  float* tabi = b.p_img + s->n_w_img;
  float* w1m = b.p_mot;
  float* w2m = w1m + (int64_t)c.motion_mlp.n_hidden * c.motion_mlp.n_in;
  float* tabm = b.p_mot + s->n_w_mot;
  float* g_w1i = s->grad_img;
  float* g_w2i = g_w1i + (int64_t)c.image_mlp.n_hidden * c.image_mlp.n_in;
  float* g_tabi = s->grad_img + s->n_w_img;
  float* g_w1m = s->grad_mot;
  float* g_w2m = g_w1m + (int64_t)c.motion_mlp.n_hidden * c.motion_mlp.n_in;
  float* g_tabm = s->grad_mot + s->n_w_mot;
  float* slot1 = s->fftbuf + 2 * P;
  // cfg.mlp_fp16: the encodings (and dL/d enc, scaled by tcnn's loss scale) live as packed halves, one 4-byte word
  // per (point, level), like tiny-cuda-nn's fp16 encoding output / dL/dinput: half the bytes for the two MLP
  // kernels and 4-byte gathers for the encode backward.  (The generic atomic scatter reads fp32: fp32 buffers then.)
  static const bool act16_env = [] { const char* e = immoco_diag_env("IMMOCO_ACT16"); return !e || atoi(e) != 0; }();
  const bool act16 = s->cfg.mlp_fp16 == 1 && !s->cfg.atomic_scatter && act16_env;
  const int64_t e_ps = act16 ? 1 : 2;                  // strides of the level-major encodings, in 4-byte words
  const int64_t e_ls_m = act16 ? NP : 2 * NP, e_ls_i = act16 ? P : 2 * P;

  std::vector<Step> st;
  // The motion forward is captured FIRST: of the two root chains of the replayed graph, the one captured first starts
  // with the graph, the other one ~12 us later behind the runtime's internal fork - and the motion chain is the
  // critical one (the image chain has 0.27 ms of slack).  A/B switch (environment, read once): IMMOCO_FWD_ORDER=image.
  static const bool image_first = [] { const char* e = immoco_diag_env("IMMOCO_FWD_ORDER"); return e && strcmp(e, "image") == 0; }();
  auto push_motion_fwd = [&] {
  if (nM > 0) {
    st.push_back({"motion_encode_fwd", [=](hipStream_t q) {
                    if (s->cfg.table_fp16)
                      return launch_hashgrid_fwd_half(s->lv_mot, lm, NP, s->shadow_mot, s->enc_mot, e_ps, e_ls_m, q, act16);
                    return launch_hashgrid_fwd(s->lv_mot, nullptr, &lm, NP, tabm, s->enc_mot, e_ps, e_ls_m, q, act16);
                  }, 1});
    st.push_back({"motion_mlp_fwd", [=](hipStream_t q) {
                    if (s->cfg.mlp_fp16 == 2)
                      return launch_mlp_fwd_bf16x2(s->cfg.motion_mlp, s->enc_mot, 2, 2 * NP, NP, w1m, w2m, s->o_mot, q);
                    if (s->cfg.mlp_fp16)
                      return launch_mlp_fwd_f16(s->cfg.motion_mlp, s->enc_mot, e_ps, e_ls_m, NP, w1m, w2m, s->o_mot, q, act16);
                    return launch_mlp_fwd(s->cfg.motion_mlp, s->enc_mot, 2, 2 * NP, NP, w1m, w2m, s->o_mot, q);
                  }, 1});
  }
  };
  auto push_image_fwd = [&] {
  static const bool early_env0 = [] { const char* e = immoco_diag_env("IMMOCO_EARLY_IMG"); return e && atoi(e) != 0; }();
  if (early_env0 && s->pruned && split_image_bwd(s->cfg) && backward && nM > 0)   // (= early_img below) the warp backward's dL/dimage share
    st.push_back({"zero_dimage_warp", [=](hipStream_t q) {
                    IMMOCO_CHECK_HIP(hipMemsetAsync(s->dimage_w, 0, (size_t)P * 8, q));
                    return IMMOCO_OK;
                  }, 2});
  st.push_back({"image_encode_fwd", [=](hipStream_t q) {
                  if (s->cfg.table_fp16)
                    return launch_hashgrid_fwd_half(s->lv_img, li, P, s->shadow_img, s->enc_img, e_ps, e_ls_i, q, act16);
                  return launch_hashgrid_fwd(s->lv_img, nullptr, &li, P, tabi, s->enc_img, e_ps, e_ls_i, q, act16);
                }, 2});
  st.push_back({"image_mlp_fwd", [=](hipStream_t q) {
                  if (s->cfg.mlp_fp16 == 2)
                    return launch_mlp_fwd_bf16x2(s->cfg.image_mlp, s->enc_img, 2, 2 * P, P, w1i, w2i, s->image, q);
                  if (s->cfg.mlp_fp16)
                    return launch_mlp_fwd_f16(s->cfg.image_mlp, s->enc_img, e_ps, e_ls_i, P, w1i, w2i, s->image, q, act16);
                  return launch_mlp_fwd(s->cfg.image_mlp, s->enc_img, 2, 2 * P, P, w1i, w2i, s->image, q);
                }, 2});
  st.push_back({"image_to_fft_slot", [=](hipStream_t q) { return launch_image_to_slot(s->image, H, W, s->fftbuf, q); }, 2});
  if (s->pruned)   // rows of the unwarped image -> every column of the transposed k-space Z (the warp overwrites its own)
    st.push_back({"image_rows_fft", [=](hipStream_t q) { return fft_rows_fwd_to_t(s->fftbuf, s->fft_t, H, W, q); }, 2});
  };
  if (image_first) {
    push_image_fwd();
    push_motion_fwd();
  } else {
    push_motion_fwd();
    push_image_fwd();
  }
  const bool pruned = s->pruned;
  // Pruned path with the split wide-MLP backward: the image branch of the second fork starts right after the column
  // adjoint - the unwarped image's adjoint rows and the gradient-entropy term (three small kernels that each waited
  // 25-50 us for a slot once the encode backward's 6000 workgroups were queued: profiles/r04_timeline_split_*.txt) run
  // beside the warp backward instead, dL/dimage = [sign*adj0 + lambda dGE] + [the warp backward's share] is summed where
  // the MLP backward kernels load it (two planar addends), and the warp's accumulation buffer is cleared at the start
  // of the next image forward.
  // MEASURED (800 iterations, two runs each): early 1.2110 / 1.2114 ms per iteration in fp32 and 0.9797 / 0.9778 with fp16 MLPs;
  // late (the image branch forks after the motion MLP backward, its three small kernels and the sum kernel in the encode
  // backward's shadow) 1.1952 and 0.9562 - the d enc kernel beside the motion MLP backward costs the critical chain more
  // than the small kernels' queueing costs the image chain, which has 200 us of slack.  Late is the default; the early
  // variant stays behind IMMOCO_EARLY_IMG=1 in the diagnostics build.
  static const bool early_env = [] { const char* e = immoco_diag_env("IMMOCO_EARLY_IMG"); return e && atoi(e) != 0; }();
  const bool early_img = early_env && pruned && split_image_bwd(s->cfg) && backward && nM > 0;
  float* zt0 = s->fft_t + 2 * P;   // pruned path: the adjoint seed of the unwarped image (its own columns only), [W][H]
  if (pruned) {
    // warp + row DFT of every motion image's own columns -> Z; ONE column transform; the select is implicit
    st.push_back({"motion_warp_dft", [=](hipStream_t q) {
                    return launch_motion_warp_dft(s->image, s->o_mot, s->xs, s->ys, nM, H, W, s->tw, s->pr_cols, s->pr_off,
                                                  s->t_mot, s->fft_t, q);
                  }});
    st.push_back({"fft_cols_fwd", [=](hipStream_t q) { return fft_cols_inplace_t(s->fft_t, H, W, false, q); }});
    st.push_back({"select_dc_seed", [=](hipStream_t q) {   // one image per column: the kernel's nM = 0 case
                    return launch_select_dc_seed_t(s->fft_t, b.col_group, b.kin, 0, H, W, s->kout, b.loss_hist,
                                                   s->iter_dev, q);
                  }});
    if (!backward) return st;
    st.push_back({"fft_cols_adjoint", [=](hipStream_t q) { return fft_cols_inplace_t(s->fft_t, H, W, true, q); }});
  } else {
  if (nM > 0) {
    st.push_back({"motion_warp_fwd", [=](hipStream_t q) {
                    return launch_motion_warp_fwd(s->image, s->o_mot, s->xs, s->ys, nM, H, W, s->t_mot, slot1, q);
                  }});
  }
  // k-space lives transposed, [W][nM+1][H], between the two transforms (kspace.hip: 2 rocFFT kernels per
  // transform instead of 4); b.kin and s->kout are [W][H] accordingly
  st.push_back({"fft_fwd", [=](hipStream_t q) { return fft_fwd_to_transposed(s->fftbuf, s->fft_t, nM + 1, H, W, q); }});
  st.push_back({"select_dc_seed", [=](hipStream_t q) {
                  return launch_select_dc_seed_t(s->fft_t, b.col_group, b.kin, nM, H, W, s->kout, b.loss_hist,
                                                 s->iter_dev, q);
                }});
  if (!backward) return st;
  st.push_back({"fft_adjoint", [=](hipStream_t q) { return fft_adj_from_transposed(s->fft_t, s->fftbuf, nM + 1, H, W, q); }});
  st.push_back({"image_grad_init_ge", [=](hipStream_t q) {
                  return launch_image_grad_init(s->image, s->fftbuf, H, W, s->lambda_dev, s->iter_dev,
                                                b.loss_hist, s->dimage, q);
                }});
  }
  // Where the second fork sits.  "late" (exact-fp32 MLPs): after the motion MLP backward - the wide MLP backward
  // needs 448 registers and 105 KB of LDS there and cannot share a CU with anything, so it runs beside the
  // gather-bound encode backward (and starves: DESIGN.md 4.4).  "early" (fp16 MLPs): right after the warp backward -
  // the two MLP backwards (62 + 2 x 34 KB of LDS, <= 256 registers) share the CUs, the image chain's small encode
  // backward and Adam then start beside the motion grid's encode backward instead of queueing behind its 6000
  // workgroups.  A/B switch (environment, read once): IMMOCO_FORK=early|late.
  static const int fork_env = [] {
    const char* e = immoco_diag_env("IMMOCO_FORK");
    return !e ? -1 : (strcmp(e, "early") == 0 ? 1 : 0);
  }();
  // Round 4, split wide backward + pruned path (800 iterations): fp32 late 1.1854 / early 1.2126 ms per iteration; fp16 MLPs
  // late 0.9552 / early 0.9472 - with fp16 MLPs the motion MLP backward is short (69 us) and VALU-bound, and the image
  // chain's MFMA kernels fit beside it.
  const bool fork_early = fork_env >= 0 ? fork_env == 1 : (s->cfg.mlp_fp16 == 1 && split_image_bwd(s->cfg) && s->pruned);
  if (nM > 0) {
    st.push_back({"motion_warp_bwd", [=](hipStream_t q) {
                    if (pruned)   // adjoint row DFT on the fly; dL/dimage share into dimage_w (summed by the late grad init)
                      return launch_motion_warp_bwd_dft(s->image, s->t_mot, s->xs, s->ys, s->fft_t, s->tw, s->pr_cols,
                                                        s->pr_off, nM, H, W, s->dimage_w, s->o_mot, q);
                    return launch_motion_warp_bwd(s->image, s->t_mot, s->xs, s->ys, slot1, nM, H, W, s->dimage,
                                                  s->o_mot, q);
                  }, early_img ? 1 : 0});
    if (early_img) st.back().signal = 0;   // the image branch's MLP backward waits for it
    st.push_back({"motion_mlp_bwd", [=](hipStream_t q) {
                    if (s->cfg.mlp_fp16 == 2)
                      return launch_mlp_bwd_bf16x2(s->cfg.motion_mlp, s->enc_mot, 2, 2 * NP, NP, w1m, w2m, s->o_mot,
                                                   s->enc_mot, g_w1m, g_w2m, q, 0);
                    if (s->cfg.mlp_fp16)
                      return launch_mlp_bwd_f16(s->cfg.motion_mlp, s->enc_mot, e_ps, e_ls_m, NP, w1m, w2m, s->o_mot,
                                                s->enc_mot, g_w1m, g_w2m, q, 0, TCNN_LOSS_SCALE, act16);
                    return launch_mlp_bwd(s->cfg.motion_mlp, s->enc_mot, 2, 2 * NP, NP, w1m, w2m, s->o_mot,
                                          s->enc_mot, g_w1m, g_w2m, q);
                  }, (fork_early || early_img) ? 1 : 0});  // "late": before the fork, the image chain's MFMA-bound MLP backward
                        // then runs beside the gather-bound encode backward instead of beside this MFMA-bound kernel (-1 %)
  }
  // Second fork: which chain is captured (and therefore starts) first.  "motion" (default in fp32): the 448-register
  // wide MLP backward must not take whole CUs ahead of the dominant gather (1.39 vs 1.35 ms, round 2).  "image": with
  // fp16 MLPs the wide backward (59 KB of LDS, <= 256 registers, 41 us alone) fits BESIDE two encode-backward
  // workgroups per CU - but only if its 256 workgroups are placed before the gather's 6000 fill every CU three deep
  // (then it runs in the gather's tail: 453 us, and the image chain ends the iteration 48 us after the motion chain).
  // A/B switch (environment, read once): IMMOCO_FORK2=image|motion.
  static const int fork2_env = [] {
    const char* e = immoco_diag_env("IMMOCO_FORK2");
    return !e ? -1 : (strcmp(e, "image") == 0 ? 1 : 0);
  }();
  const bool image_first2 = fork2_env >= 0 ? fork2_env == 1 : false;
  auto push_motion_encode_bwd = [&] {
  if (nM > 0) {
    st.push_back({"motion_encode_bwd", [=](hipStream_t q) {
                    if (s->plan_mot)
                      return launch_csr_bwd(s->plan_mot, s->enc_mot, g_tabm, s->mot_gstride, 1, q, act16,
                                            1.f / TCNN_LOSS_SCALE);
                    return launch_hashgrid_bwd(s->lv_mot, nullptr, &lm, NP, s->enc_mot, 2, 2 * NP, g_tabm, q);
                  }, 1});
  }
  };
  auto push_image_mlp_bwd = [&] {
  // (The image chain's MFMA kernel starves beside the motion grid's encode backward - 0.53 ms instead of 0.14 in
  // the rocprofv3 stats - and slows that gather from 0.45 to 0.58 ms; run BEFORE the fork instead, alone, the
  // iteration takes 1.435 ms instead of 1.351: the overlap is still worth more than it costs.)
  if (pruned) {
    // off the critical path: the unwarped image's adjoint rows and the gradient-entropy term start the image branch
    st.push_back({"image_adjoint_rows", [=](hipStream_t q) {
                    int rc = launch_keep_group0_cols(s->fft_t, b.col_group, nM, H, W, zt0, q);
                    return rc ? rc : fft_rows_adj_from_t(zt0, s->fftbuf, H, W, q);
                  }, 2});
    if (early_img)   // beside the warp backward: dimage = sign*adj0 + lambda dGE (the warp's share stays in dimage_w)
      st.push_back({"image_grad_init_ge_early", [=](hipStream_t q) {
                      return launch_image_grad_init(s->image, s->fftbuf, H, W, s->lambda_dev, s->iter_dev, b.loss_hist,
                                                    s->dimage, q);
                    }, 2});
    else
      st.push_back({"image_grad_init_ge_late", [=](hipStream_t q) {
                      return launch_image_grad_init_after_warp(s->image, s->fftbuf, H, W, s->lambda_dev, s->iter_dev,
                                                               b.loss_hist, s->dimage, s->dimage_w, q);
                    }, 2});
  }
  if (split_image_bwd(s->cfg)) {
    const float* d2 = early_img ? s->dimage_w : nullptr;   // second planar addend of dL/dimage
    st.push_back({"image_mlp_bwd_denc", [=](hipStream_t q) {
                    if (s->cfg.mlp_fp16)
                      return launch_mlp_bwd_f16_split(s->cfg.image_mlp, 1, s->enc_img, e_ps, e_ls_i, P, w1i, w2i, s->dimage,
                                                      s->denc_img, g_w1i, g_w2i, q, /*planar dimage*/ P, TCNN_LOSS_SCALE, act16, d2);
                    return launch_mlp_bwd_denc(s->cfg.image_mlp, s->enc_img, 2, 2 * P, P, w1i, w2i, s->dimage, s->denc_img, q,
                                               /*planar dimage*/ P, d2);
                  }, 2});
    if (early_img) st.back().wait = 0;   // the warp backward (other branch) has added its share
    st.push_back({"image_mlp_bwd_dw", [=](hipStream_t q) {
                    if (s->cfg.mlp_fp16)
                      return launch_mlp_bwd_f16_split(s->cfg.image_mlp, 2, s->enc_img, e_ps, e_ls_i, P, w1i, w2i, s->dimage,
                                                      nullptr, g_w1i, g_w2i, q, /*planar dimage*/ P, TCNN_LOSS_SCALE, act16, d2);
                    return launch_mlp_bwd_dw(s->cfg.image_mlp, s->enc_img, 2, 2 * P, P, w1i, w2i, s->dimage, g_w1i, g_w2i, q,
                                             /*planar dimage*/ P, d2);
                  }, 2});
    return;
  }
  st.push_back({"image_mlp_bwd", [=](hipStream_t q) {
                  if (s->cfg.mlp_fp16 == 2)
                    return launch_mlp_bwd_bf16x2(s->cfg.image_mlp, s->enc_img, 2, 2 * P, P, w1i, w2i, s->dimage, s->enc_img,
                                                 g_w1i, g_w2i, q, /*planar dimage*/ P);
                  if (s->cfg.mlp_fp16)
                    return launch_mlp_bwd_f16(s->cfg.image_mlp, s->enc_img, e_ps, e_ls_i, P, w1i, w2i, s->dimage,
                                              s->enc_img, g_w1i, g_w2i, q, /*planar dimage*/ P, TCNN_LOSS_SCALE, act16);
                  return launch_mlp_bwd(s->cfg.image_mlp, s->enc_img, 2, 2 * P, P, w1i, w2i, s->dimage, s->enc_img,
                                        g_w1i, g_w2i, q, /*planar dimage*/ P);
                }, 2});
  };
  if (image_first2) {
    push_image_mlp_bwd();
    push_motion_encode_bwd();
  } else {
    push_motion_encode_bwd();
    push_image_mlp_bwd();
  }
  st.push_back({"image_encode_bwd", [=](hipStream_t q) {
                  const float* de = split_image_bwd(s->cfg) ? s->denc_img : s->enc_img;
                  if (s->plan_img) return launch_csr_bwd(s->plan_img, de, g_tabi, 0, 1, q, act16, 1.f / TCNN_LOSS_SCALE);
                  return launch_hashgrid_bwd(s->lv_img, nullptr, &li, P, de, 2, 2 * P, g_tabi, q);
                }, 2});
  // optimizer param-group order of the reference: motion first, then image (immoco.py:149-154)
  if (nM > 0)
    st.push_back({"adam_motion", [=](hipStream_t q) {
                    // with a plan only the MLP weights and the slot blocks that receive gradients are visited
                    // (everything else has an exactly-zero update), and only the weights and the blocks that
                    // are flushed by atomics need clearing; the rest is overwritten by csr_bwd_kernel
                    if (s->plan_mot) {
                      uint32_t nb = 0;
                      const uint2* blocks = csr_plan_touched(s->plan_mot, &nb);
                      return launch_adam_blocks(b.p_mot, s->grad_mot, s->mot_tables, s->mot_gstride, b.a_mot,
                                                b.a_mot + s->n_params_mot, s->n_w_mot, blocks, nb, s->sched,
                                                s->iter_dev, 0.9f, 0.999f, 1e-8f, q,
                                                s->cfg.table_fp16 ? s->shadow_mot : nullptr);
                    }
                    return launch_adam_sched(b.p_mot, s->grad_mot, 1, s->mot_gstride, b.a_mot,
                                             b.a_mot + s->n_params_mot, s->n_params_mot, s->n_params_mot, s->sched,
                                             s->iter_dev, 0.9f, 0.999f, 1e-8f, q,
                                             s->cfg.table_fp16 ? s->shadow_mot : nullptr, s->n_w_mot);
                  }, 1});
  st.push_back({"adam_image", [=](hipStream_t q) {
                  if (s->plan_img) {
                    uint32_t nb = 0;
                    const uint2* blocks = csr_plan_touched(s->plan_img, &nb);
                    return launch_adam_blocks(b.p_img, s->grad_img, 1, 0, b.a_img, b.a_img + s->n_params_img,
                                              s->n_w_img, blocks, nb, s->sched, s->iter_dev, 0.9f, 0.999f, 1e-8f, q,
                                              s->cfg.table_fp16 ? s->shadow_img : nullptr);
                  }
                  return launch_adam_sched(b.p_img, s->grad_img, 1, 0, b.a_img, b.a_img + s->n_params_img,
                                           s->n_params_img, s->n_params_img, s->sched, s->iter_dev, 0.9f, 0.999f,
                                           1e-8f, q, s->cfg.table_fp16 ? s->shadow_img : nullptr, s->n_w_img);
                }, 2});
  st.push_back({"tick", [=](hipStream_t q) {
                  tick_kernel<<<1, 1, 0, q>>>(s->iter_dev);
                  IMMOCO_LAUNCH_CHECK();
                  return IMMOCO_OK;
                }});
  for (Step& x : st) {
    const std::string n = x.name;
    x.group = (n == "image_encode_fwd" || n == "image_mlp_fwd" || n == "image_to_fft_slot" || n == "image_rows_fft" ||
               n == "zero_dimage_warp")                                                          ? 0
              : (n == "motion_encode_fwd" || n == "motion_mlp_fwd")                             ? 1
              : n == "motion_encode_bwd"                                                        ? 3
              : (n == "image_mlp_bwd" || n == "image_mlp_bwd_denc" || n == "image_mlp_bwd_dw" || n == "image_encode_bwd" ||
                 n == "adam_image" || n == "image_adjoint_rows" || n == "image_grad_init_ge_late" ||
                 n == "image_grad_init_ge_early")                                               ? 4
              : n == "tick"                                                                     ? 5
              : n == "adam_motion"                                                              ? 6
                                                                                                : 2;
  }
  return st;
}

// Software pipelining ACROSS iterations (round 2).  In the classic order
//     [image fwd || motion fwd] -> middle -> [motion encode bwd, motion Adam || image bwd] -> tick
// the image chain's backward runs BESIDE the motion grid's encode backward, the dominant kernel: its MFMA kernel
// needs a whole SIMD's registers and starves there (0.62 ms instead of 0.14), the gather it runs beside slows
// from 0.45 to 0.58 ms, and the iteration still ends with the image chain ~0.08 ms after the motion chain.
// Nothing of iteration k+1's MOTION forward depends on the image parameters, and the image backward needs
// nothing of the motion grid's backward, so the iteration is ROTATED: the replayed graph runs from the end of
// one motion encode backward to the end of the next,
//   first:    [image fwd(0) || motion fwd(0)] -> middle(0) -> motion encode bwd(0)
//   steady:   [motion Adam(k), motion fwd(k+1) || image bwd(k), image fwd(k+1)] -> tick -> middle(k+1)
//                                                                             -> motion encode bwd(k+1)
//   last:     [motion Adam(n-1) || image bwd(n-1)]
// so that the dominant gather runs ALONE and the image chain's 0.3 ms (backward + next forward) sit beside
// the motion Adam (HBM-bound, short workgroups) and the motion encode forward.  Both Adam kernels read the
// schedule of their own iteration because the tick comes after the join.  Every parameter sees the same
// updates in the same order as before; only the launch order changes.
// MEASURED (MI355X, 320x320x10, rocprofv3 kernel trace): the dominant gather does run alone (0.44 ms instead of
// 0.58), but the motion encode FORWARD beside the image chain takes 0.51 ms instead of 0.27 and the image encode
// backward 0.19 instead of 0.04: a level slice of the fp32 table is exactly one XCD L2 (4 MB), and any stream of
// other lines through that L2 pushes the forward over the cliff immoco_probe_gather shows between 4 and 8 MB
// footprints (259 -> 120 G requests/s).  1.44 ms per iteration against 1.35 in the classic order, which
// therefore stays the default; IMMOCO_PIPELINE=1 selects this order (A/B switch).
// (A first variant that merely deferred the image backward to the FRONT of the next iteration, ahead of the
// image forward the warp waits for, was slower: 1.446 vs 1.351 ms.)
enum class Order { Classic, First, Steady, Epilogue };

std::vector<Step> arrange(const std::vector<Step>& all, Order o) {
  if (o == Order::Classic) return all;
  std::vector<Step> out;
  auto take = [&](int group, int branch) {
    for (const Step& x : all)
      if (x.group == group) {
        Step y = x;
        y.branch = branch;
        out.push_back(y);
      }
  };
  if (o == Order::Epilogue) {
    take(6, 1);
    take(4, 2);
    return out;
  }
  if (o == Order::Steady) {
    take(6, 1);
    take(1, 1);
    take(4, 2);
    take(0, 2);
    take(5, 0);
  } else {
    take(0, 2);
    take(1, 1);
  }
  take(2, 0);
  take(3, 0);
  return out;
}

// A graph executable must outlive its queued launches: a solve that needs a new capture (other slice, other
// buffers) parks the old one behind an event on the stream instead of destroying it under the GPU's feet.
// A replaced graph executable may still have launches queued, and the runtime runs its forked branches on streams of
// its own: destroying it as soon as an event on the solver's stream had completed crashed once in ~15 runs of the
// 64-slice batch test (SIGSEGV in the host process, a recapture every 16 ms).  Retired executables are therefore kept
// until the whole device is idle: at the latest when RETIRED_MAX of them have piled up (a device synchronise: the
// host is that far ahead of the GPU only in batch solves), and when the solver is destroyed or re-planned.
constexpr size_t RETIRED_MAX = 16;
// The single-slice graphs (gexec, gexec2; key gkey) and the pair graphs (pg1, pgk; key pkey) are cached
// independently: a solve that needs a new capture retires only its own family, so an odd-B paired batch (pairs,
// then one single solve) or alternating single / paired calls on one solver do not re-capture each other's graphs.
enum : int { RETIRE_SINGLE = 1, RETIRE_PAIR = 2, RETIRE_ALL = 3 };
void retire_graph(immoco_solver* s, int which = RETIRE_ALL) {
  auto drop = [&](hipGraphExec_t* g) {
    if (!*g) return;
    s->retired.emplace_back(*g, nullptr);
    *g = nullptr;
  };
  if (which & RETIRE_SINGLE) {
    drop(&s->gexec);
    drop(&s->gexec2);
    s->gkey.clear();
  }
  if (which & RETIRE_PAIR) {
    drop(&s->pg1);
    drop(&s->pgk);
    s->pkey.clear();
  }
}
void sweep_retired(immoco_solver* s, bool wait) {
  if (s->retired.empty() || (!wait && s->retired.size() < RETIRED_MAX)) return;
  (void)hipDeviceSynchronize();
  for (auto& r : s->retired) {
    hipGraphExecDestroy(r.first);
    if (r.second) hipEventDestroy(r.second);
  }
  s->retired.clear();
  (void)hipGetLastError();
}

int run_steps(const std::vector<Step>& steps, hipStream_t q) {
  for (const Step& st : steps) {
    int rc = st.run(q);
    if (rc) return rc;
  }
  return IMMOCO_OK;
}

// fork/join execution of the branch annotations (works eagerly and under stream capture)
// hook(step, stream, before): called right before / after a step is issued on its stream (cross-slice ordering)
typedef std::function<int(const Step&, hipStream_t, bool)> StepHook;
int run_steps_forked(immoco_solver* s, const std::vector<Step>& steps, hipStream_t q, const StepHook* hook = nullptr,
                     bool join_at_end = true);

int ensure_sched(immoco_solver* s, int32_t iters) {
  if (iters <= s->sched_cap) return IMMOCO_OK;
  if (s->sched) IMMOCO_CHECK_HIP(hipFree(s->sched));
  if (s->lambda_dev) IMMOCO_CHECK_HIP(hipFree(s->lambda_dev));
  s->sched = s->lambda_dev = nullptr;
  IMMOCO_CHECK_HIP(hipMalloc((void**)&s->sched, (size_t)iters * 2 * sizeof(float)));
  IMMOCO_CHECK_HIP(hipMalloc((void**)&s->lambda_dev, (size_t)iters * sizeof(float)));
  s->sched_cap = iters;
  s->sched_host.clear();
  // the schedule pointers are baked into a captured graph (the old arrays were freed above: drain first)
  retire_graph(s);
  sweep_retired(s, true);
  return IMMOCO_OK;
}

// fp16 mode: (re)build the shadows from the caller's fp32 master tables
int refresh_shadows(immoco_solver* s, const float* p_img, const float* p_mot, hipStream_t q) {
  if (!s->cfg.table_fp16) return IMMOCO_OK;
  int rc = launch_f32_to_half(p_img + s->n_w_img, s->shadow_img, s->n_params_img - s->n_w_img, q);
  if (rc == IMMOCO_OK && s->cfg.nM > 0)
    rc = launch_f32_to_half(p_mot + s->n_w_mot, s->shadow_mot, s->n_params_mot - s->n_w_mot, q);
  return rc;
}

int enter(immoco_solver* s, hipStream_t caller) {
  IMMOCO_CHECK_HIP(hipEventRecord(s->ev_in, caller));
  IMMOCO_CHECK_HIP(hipStreamWaitEvent(s->stream, s->ev_in, 0));
  return IMMOCO_OK;
}
int leave(immoco_solver* s, hipStream_t caller) {
  IMMOCO_CHECK_HIP(hipEventRecord(s->ev_out, s->stream));
  IMMOCO_CHECK_HIP(hipStreamWaitEvent(caller, s->ev_out, 0));
  return IMMOCO_OK;
}

}  // namespace

namespace {
int run_steps_forked(immoco_solver* s, const std::vector<Step>& steps, hipStream_t q, const StepHook* hook,
                     bool join_at_end) {
  bool forked = false;
  int ev = 0;
  for (const Step& st : steps) {
    if (st.branch != 0 && !forked) {  // fork: the side stream starts after everything issued so far
      IMMOCO_CHECK_HIP(hipEventRecord(s->ev_fj[ev], q));
      IMMOCO_CHECK_HIP(hipStreamWaitEvent(s->side, s->ev_fj[ev], 0));
      ev = (ev + 1) % 8;
      forked = true;
    } else if (st.branch == 0 && forked) {  // join
      IMMOCO_CHECK_HIP(hipEventRecord(s->ev_fj[ev], s->side));
      IMMOCO_CHECK_HIP(hipStreamWaitEvent(q, s->ev_fj[ev], 0));
      ev = (ev + 1) % 8;
      forked = false;
    }
    hipStream_t sq = st.branch == 2 ? s->side : q;
    const bool mark = strcmp(st.name, "motion_encode_bwd") == 0;
    if (mark) IMMOCO_CHECK_HIP(hipEventRecord(s->ev_k0, sq));
    int rc;
    if (hook && (rc = (*hook)(st, sq, true))) return rc;
    if (st.wait >= 0 && forked) IMMOCO_CHECK_HIP(hipStreamWaitEvent(sq, s->ev_x[st.wait], 0));
    rc = st.run(sq);
    if (rc) return rc;
    if (st.signal >= 0 && forked) IMMOCO_CHECK_HIP(hipEventRecord(s->ev_x[st.signal], sq));
    if (hook && (rc = (*hook)(st, sq, false))) return rc;
    if (mark) {
      IMMOCO_CHECK_HIP(hipEventRecord(s->ev_k1, sq));
      s->marked = true;
    }
  }
  if (forked && join_at_end) {
    IMMOCO_CHECK_HIP(hipEventRecord(s->ev_fj[ev], s->side));
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(q, s->ev_fj[ev], 0));
  }
  return IMMOCO_OK;
}
}  // namespace

extern "C" int immoco_solver_create(const immoco_solver_cfg* cfg, immoco_solver_t* out) {
  IMMOCO_REQUIRE(cfg && out, "solver_create: NULL argument");
  IMMOCO_REQUIRE(cfg->H >= 2 && cfg->W >= 2 && cfg->nM >= 0, "solver_create: bad shape H=%d W=%d nM=%d", cfg->H,
                 cfg->W, cfg->nM);
  IMMOCO_REQUIRE((cfg->H % 2) == 0 && (cfg->W % 2) == 0,
                 "solver_create: H and W must be even (centred-FFT sign folding), got %dx%d", cfg->H, cfg->W);
  IMMOCO_REQUIRE(cfg->image_grid.dims == 2 && cfg->motion_grid.dims == 3, "solver_create: grids must be 2-D/3-D");
  IMMOCO_REQUIRE(cfg->image_grid.n_levels == 16 && cfg->motion_grid.n_levels == 16,
                 "solver_create: 16 levels x 2 features expected");
  // the words that were `reserved` before round 3 must be zero or a documented value (include/immoco_hip.h)
  IMMOCO_REQUIRE(cfg->mlp_fp16 >= 0 && cfg->mlp_fp16 <= 2, "solver_create: mlp_fp16 must be 0 (fp32), 1 (fp16 operands) or 2 (bf16x2), got %d", cfg->mlp_fp16);
  IMMOCO_REQUIRE(cfg->batch_pair == 0 || cfg->batch_pair == 1, "solver_create: batch_pair must be 0 or 1, got %d", cfg->batch_pair);
  IMMOCO_REQUIRE(cfg->serial_chains >= 0 && cfg->serial_chains <= 2, "solver_create: serial_chains must be 0, 1 or 2 (auto), got %d", cfg->serial_chains);
  IMMOCO_REQUIRE(cfg->batch_lanes >= 0 && cfg->batch_lanes <= 64, "solver_create: batch_lanes out of range (%d)", cfg->batch_lanes);
  int rc;
  if ((rc = check_mlp_cfg(&cfg->image_mlp)) || (rc = check_mlp_cfg(&cfg->motion_mlp))) return rc;
  immoco_solver* s = new immoco_solver();
  s->cfg = *cfg;
  // serial_chains = 2: the library decides.  Measured on MI355X (bench.py --workload c5 --chains fork|serial, 300 and
  // 600 iterations, round 3): 640x640x20 forked 8.00 ms per iteration, serial 8.19, sum of the isolated kernels 8.23;
  // 320x320x10 forked 1.25, serial 1.40.  The fork wins at both ends, so "auto" is the fork (round 2's 9.09 ms at C5
  // was the 27 s solve's sustained clock, not the fork: DESIGN.md 4.4).
  if (s->cfg.serial_chains == 2) s->cfg.serial_chains = 0;
  if ((rc = build_levels(&cfg->image_grid, &s->lv_img)) || (rc = build_levels(&cfg->motion_grid, &s->lv_mot))) {
    delete s;
    return rc;
  }
  s->P = (int64_t)cfg->H * cfg->W;
  s->NP = s->P * cfg->nM;
  s->n_w_img = (int64_t)cfg->image_mlp.n_hidden * (cfg->image_mlp.n_in + cfg->image_mlp.n_out_padded);
  s->n_w_mot = (int64_t)cfg->motion_mlp.n_hidden * (cfg->motion_mlp.n_in + cfg->motion_mlp.n_out_padded);
  s->n_params_img = s->n_w_img + 2 * (int64_t)s->lv_img.offset[16];
  s->n_params_mot = s->n_w_mot + 2 * (int64_t)s->lv_mot.offset[16];
  const int64_t NPa = s->NP > 0 ? s->NP : 1;
#define A(ptr, cnt)                       \
  if ((rc = dev_alloc(s, &s->ptr, cnt))) { \
    immoco_solver_destroy(s);             \
    return rc;                            \
  }
  A(enc_img, 32 * s->P)
  if (split_image_bwd(s->cfg)) A(denc_img, 32 * s->P)
  A(enc_mot, 32 * NPa)
  A(image, 2 * s->P)
  A(o_mot, 2 * NPa)
  A(t_mot, 2 * NPa)
  A(fftbuf, 2 * s->P * (cfg->nM + 1))
  A(dimage, 2 * s->P)
  A(kout, 2 * s->P)
  A(kin_t, 2 * s->P)
  A(fft_t, 2 * s->P * (cfg->nM + 1))
  A(grad_img, s->n_params_img)
  // point ranges of the motion grid's transposed index: as many as it takes to bring one level slice of dL/denc
  // (8 B per point) down to ~2 MB per XCD L2 (csr.hip): 320x320x10 -> 4, 640x640x20 -> 32; at most 8 of them run
  // in one launch, each into its own partial gradient table (summed by Adam), further ones in following launches
  // (with cfg.mlp_fp16 dL/denc is stored as packed halves: half the bytes per point, half the parts - at 320x320x10
  // 2 parts instead of 4: encode backward 0.386 -> 0.374 ms, motion Adam 0.069 -> 0.054 ms, iteration 1.042 -> 1.020)
  const bool act16_on = [] { const char* e = immoco_diag_env("IMMOCO_ACT16"); return !e || atoi(e) != 0; }();
  const int auto_parts = csr_auto_parts((int64_t)cfg->nM * cfg->H * cfg->W, cfg->mlp_fp16 == 1 && !cfg->atomic_scatter && act16_on ? 4 : 8);
  s->mot_parts = cfg->atomic_scatter ? 1 : (cfg->grad_parts > 0 ? cfg->grad_parts : auto_parts);
  s->mot_tables = std::min(s->mot_parts, 8);
  s->mot_gstride = (s->n_params_mot + 3) / 4 * 4;
  A(grad_mot, s->mot_gstride * s->mot_tables)
  A(iter_dev, 4)
  if (cfg->table_fp16) {
    A(shadow_img, s->n_params_img - s->n_w_img)
    A(shadow_mot, s->n_params_mot - s->n_w_mot)
  }
  A(xs, cfg->W)
  A(ys, cfg->H)
  A(ms, cfg->nM > 0 ? cfg->nM : 1)
  s->pruned = use_pruned(*cfg);
  if (s->pruned) {
    A(tw, 2 * (int64_t)cfg->W)
    A(dimage_w, 2 * s->P)
    A(pr_cols, cfg->W)
    A(pr_off, cfg->nM + 2)
  }
#undef A
  hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
  // (a high- or low-priority side stream makes the iteration 7 % slower - 1.71 vs 1.60 ms - whichever way)
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking);
  for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s->ev_fj[i], hipEventDisableTiming);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&s->ev_x[i], hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&s->ev_k0);
  if (e == hipSuccess) e = hipEventCreate(&s->ev_k1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_in, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_out, hipEventDisableTiming);
  if (e == hipSuccess && s->pruned) {
    std::vector<float> twh(2 * (size_t)cfg->W);
    for (int k = 0; k < cfg->W; ++k) {
      const double a = -2.0 * M_PI * (double)k / (double)cfg->W;
      twh[2 * k] = (float)cos(a);
      twh[2 * k + 1] = (float)sin(a);
    }
    e = hipMemcpy(s->tw, twh.data(), twh.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(s->dimage_w, 0, (size_t)s->P * 8);
    if (e == hipSuccess) e = hipMemset(s->pr_off, 0, (size_t)(cfg->nM + 2) * 4);
  }
  if (e == hipSuccess) e = hipMemset(s->grad_img, 0, (size_t)s->n_params_img * 4);
  if (e == hipSuccess) e = hipMemset(s->grad_mot, 0, (size_t)s->mot_gstride * s->mot_tables * 4);
  if (e != hipSuccess) {
    set_error("solver_create: %s", hipGetErrorString(e));
    immoco_solver_destroy(s);
    return IMMOCO_E_HIP;
  }
  // create the FFT plans now (rocFFT compiles kernels) so that solve() - and its stream capture - never does
  if ((rc = fft_fwd_to_transposed(s->fftbuf, s->fft_t, cfg->nM + 1, cfg->H, cfg->W, s->stream)) ||
      (rc = fft_adj_from_transposed(s->fft_t, s->fftbuf, cfg->nM + 1, cfg->H, cfg->W, s->stream))) {
    immoco_solver_destroy(s);
    return rc;
  }
  if (s->pruned && ((rc = fft_rows_fwd_to_t(s->fftbuf, s->fft_t, cfg->H, cfg->W, s->stream)) ||
                    (rc = fft_cols_inplace_t(s->fft_t, cfg->H, cfg->W, false, s->stream)) ||
                    (rc = fft_rows_adj_from_t(s->fft_t, s->fftbuf, cfg->H, cfg->W, s->stream)))) {
    immoco_solver_destroy(s);
    return rc;
  }
  IMMOCO_CHECK_HIP(hipStreamSynchronize(s->stream));
  *out = s;
  return IMMOCO_OK;
}

extern "C" int immoco_solver_destroy(immoco_solver_t s) {
  if (!s) return IMMOCO_OK;
  for (immoco_solver* lane : s->lanes) immoco_solver_destroy(lane);
  s->lanes.clear();
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->side) (void)hipStreamSynchronize(s->side);
  retire_graph(s);
  sweep_retired(s, true);   // device idle, then the executables go
  if (s->parent) {  // a lane borrows lattices and plans from its parent
    s->plan_img = s->plan_mot = nullptr;
    s->xs = s->ys = s->ms = nullptr;
  }
  csr_plan_free(s->plan_img);
  csr_plan_free(s->plan_mot);
  if (s->pr_cols) hipFree(s->pr_cols);
  if (s->pr_off) hipFree(s->pr_off);
  float* bufs[] = {s->tw, s->dimage_w, s->denc_img, s->xs, s->ys, s->ms, s->enc_img, s->enc_mot, s->image, s->o_mot, s->t_mot, s->fftbuf, s->dimage,
                   s->kout,    s->fft_t,    s->kin_t,    s->grad_img, s->grad_mot, s->sched, s->lambda_dev};
  for (float* b : bufs)
    if (b) hipFree(b);
  if (s->iter_dev) hipFree(s->iter_dev);
  if (s->shadow_img) hipFree(s->shadow_img);
  if (s->shadow_mot) hipFree(s->shadow_mot);
  if (s->ev_in) hipEventDestroy(s->ev_in);
  if (s->ev_out) hipEventDestroy(s->ev_out);
  for (hipEvent_t ev : s->ev_fj)
    if (ev) hipEventDestroy(ev);
  for (hipEvent_t ev : s->ev_x)
    if (ev) hipEventDestroy(ev);
  if (s->ev_k0) hipEventDestroy(s->ev_k0);
  if (s->ev_k1) hipEventDestroy(s->ev_k1);
  for (hipEvent_t ev : s->ev_pair)
    if (ev) hipEventDestroy(ev);
  if (s->side) hipStreamDestroy(s->side);
  if (s->stream) hipStreamDestroy(s->stream);
  delete s;
  return IMMOCO_OK;
}

extern "C" int64_t immoco_solver_workspace_bytes(immoco_solver_t s) {
  if (!s) return 0;
  int64_t b = s->bytes + csr_plan_bytes(s->plan_img) + csr_plan_bytes(s->plan_mot);
  for (immoco_solver* lane : s->lanes) b += lane->bytes;
  return b;
}

extern "C" int64_t immoco_solver_n_params(immoco_solver_t s, int32_t which) {
  if (!s) return 0;
  return which == 0 ? s->n_params_img : s->n_params_mot;
}

extern "C" int immoco_solver_set_lattice(immoco_solver_t s, const float* xs, const float* ys, const float* ms,
                                         void* stream) {
  IMMOCO_REQUIRE(s && xs && ys && (s->cfg.nM == 0 || ms), "solver_set_lattice: NULL argument");
  hipStream_t caller = as_stream(stream);
  const immoco_solver_cfg& c = s->cfg;
  IMMOCO_CHECK_HIP(hipMemcpyAsync(s->xs, xs, (size_t)c.W * 4, hipMemcpyDeviceToDevice, caller));
  IMMOCO_CHECK_HIP(hipMemcpyAsync(s->ys, ys, (size_t)c.H * 4, hipMemcpyDeviceToDevice, caller));
  if (c.nM > 0) IMMOCO_CHECK_HIP(hipMemcpyAsync(s->ms, ms, (size_t)c.nM * 4, hipMemcpyDeviceToDevice, caller));
  IMMOCO_CHECK_HIP(hipStreamSynchronize(caller));
  for (immoco_solver* lane : s->lanes) immoco_solver_destroy(lane);  // they borrow the plans freed below
  s->lanes.clear();
  csr_plan_free(s->plan_img);
  csr_plan_free(s->plan_mot);
  s->plan_img = s->plan_mot = nullptr;
  retire_graph(s);  // plans are baked into a captured graph
  sweep_retired(s, true);
  if (!c.atomic_scatter) {
    int rc;
    const float* ax2[3] = {s->xs, s->ys, nullptr};
    const int32_t n2[3] = {c.W, c.H, 0};
    if ((rc = csr_plan_build(s->lv_img, 1, c.H, c.W, ax2, n2, 1, 1, &s->plan_img, s->stream))) return rc;
    if (c.nM > 0) {
      const float* ax3[3] = {s->ms, s->ys, s->xs};
      const int32_t n3[3] = {c.nM, c.H, c.W};
      if ((rc = csr_plan_build(s->lv_mot, c.nM, c.H, c.W, ax3, n3, s->mot_parts, s->mot_tables, &s->plan_mot,
                               s->stream)))
        return rc;
    }
  }
  s->lattice_set = true;
  return IMMOCO_OK;
}

namespace {
struct SliceArgs {
  const float* kin;
  const int32_t* cg;
  float *pi, *pm, *ai, *am;
  float *out_image, *out_kspace, *loss;
};

int check_slice(immoco_solver* s, const SliceArgs& a, int32_t iters, const float* lambda_sched, int32_t step0) {
  IMMOCO_REQUIRE(s, "solver_solve: NULL solver");
  IMMOCO_REQUIRE(s->lattice_set, "solver_solve: call immoco_solver_set_lattice first");
  IMMOCO_REQUIRE(a.kin && a.cg && a.pi && a.ai, "solver_solve: NULL buffer");
  IMMOCO_REQUIRE(s->cfg.nM == 0 || (a.pm && a.am), "solver_solve: NULL motion buffer");
  IMMOCO_REQUIRE(iters >= 1 && lambda_sched && step0 >= 0, "solver_solve: bad iters/schedule");
  return IMMOCO_OK;
}

// Per-solve initialisation of workspace `w` on stream q: schedules on the device, iteration counter, loss
// history, fp16 shadows, transposed measured k-space.
int prepare_slice(immoco_solver* w, hipStream_t q, const SliceArgs& a, int32_t iters, float lr,
                  const float* lambda_sched, int32_t step0) {
  int rc;
  if ((rc = ensure_sched(w, iters))) return rc;
  // host-side scalars in double like python/torch (immoco.py:180-181; torch Adam bias corrections)
  std::vector<float> both(3 * (size_t)iters);
  for (int j = 0; j < iters; ++j) {
    const double t = (double)(step0 + j + 1);
    both[2 * j] = (float)((double)lr / (1.0 - pow(0.9, t)));
    both[2 * j + 1] = (float)sqrt(1.0 - pow(0.999, t));
    both[2 * (size_t)iters + j] = lambda_sched[j];
  }
  // the schedules only change with (iters, lr, step0, lambda): a batch re-uses them slice after slice, and
  // skipping the upload also skips the host sync it needs (the lane would drain before the next slice is queued)
  if (both != w->sched_host) {
    IMMOCO_CHECK_HIP(hipStreamSynchronize(q));  // nothing in flight may still read the old schedules
    IMMOCO_CHECK_HIP(hipMemcpyAsync(w->sched, both.data(), (size_t)iters * 8, hipMemcpyHostToDevice, q));
    IMMOCO_CHECK_HIP(hipMemcpyAsync(w->lambda_dev, both.data() + 2 * (size_t)iters, (size_t)iters * 4, hipMemcpyHostToDevice, q));
    IMMOCO_CHECK_HIP(hipStreamSynchronize(q));  // host vector goes out of scope; also orders the pageable copies
    w->sched_host.swap(both);
  }
  IMMOCO_CHECK_HIP(hipMemsetAsync(w->iter_dev, 0, 4 * sizeof(int32_t), q));
  if (a.loss) IMMOCO_CHECK_HIP(hipMemsetAsync(a.loss, 0, (size_t)iters * 4, q));
  if ((rc = refresh_shadows(w, a.pi, a.pm, q))) return rc;
  if (w->pruned && (rc = launch_build_col_lists(a.cg, w->cfg.nM, w->cfg.W, w->pr_cols, w->pr_off, q))) return rc;
  return launch_transpose_c64(a.kin, w->kin_t, w->cfg.H, w->cfg.W, q);
}

// tensors of the LAST forward pass (immoco.py:203-206)
int finish_slice(immoco_solver* w, hipStream_t q, const SliceArgs& a) {
  if (a.out_image) IMMOCO_CHECK_HIP(hipMemcpyAsync(a.out_image, w->image, (size_t)w->P * 8, hipMemcpyDeviceToDevice, q));
  if (a.out_kspace) return launch_transpose_c64(w->kout, a.out_kspace, w->cfg.W, w->cfg.H, q);
  return IMMOCO_OK;
}

// K iterations per graph launch (everything iteration-dependent is read through the device-side counter, so a
// graph of K iterations is K copies of the same nodes): removes K - 1 of every K graph launches with their
// root-fork delay and end-of-graph gap.  Measured at 320x320x10: 1.2513 -> 1.2495 ms per iteration (the gaps hide
// behind the device-side queue).  A/B switch (environment, read once): IMMOCO_GRAPH_K (default 8; 1 = one per launch).
int graph_k() {
  static const int k = [] { const char* e = immoco_diag_env("IMMOCO_GRAPH_K"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
  return k;
}

// sync_in / sync_out: order the solve after / the caller's stream after it (a batch does this once per lane)
int solve_impl(immoco_solver_t s, const float* kspace_in, const int32_t* col_group, float* params_image,
               float* params_motion, float* adam_image, float* adam_motion, int32_t iters, float lr,
               const float* lambda_sched, int32_t step0, float* out_image, float* out_kspace, float* loss_hist,
               void* stream, bool sync_in, bool sync_out) {
  const SliceArgs a{kspace_in, col_group, params_image, params_motion, adam_image, adam_motion, out_image, out_kspace, loss_hist};
  int rc;
  if ((rc = check_slice(s, a, iters, lambda_sched, step0))) return rc;
  hipStream_t caller = as_stream(stream);
  if (sync_in && (rc = enter(s, caller))) return rc;
  hipStream_t q = s->stream;
  if ((rc = prepare_slice(s, q, a, iters, lr, lambda_sched, step0))) return rc;
  Bind b{s->kin_t, col_group, params_image, params_motion, adam_image, adam_motion, loss_hist};
  // rotated order (see arrange()): an A/B switch (IMMOCO_PIPELINE=1), OFF by default - measured slower
  static const bool want_pipeline = immoco_diag_env("IMMOCO_PIPELINE") != nullptr;
  const bool pipelined = want_pipeline && s->cfg.nM > 0 && !s->cfg.serial_chains;
  const std::vector<Step> all = build_steps(s, b, true);
  const std::vector<Step> first = arrange(all, pipelined ? Order::First : Order::Classic);
  const std::vector<Step> steady = pipelined ? arrange(all, Order::Steady) : first;
  auto run_list = [&](const std::vector<Step>& l) { return s->cfg.serial_chains ? run_steps(l, q) : run_steps_forked(s, l, q); };
  // A/B switch IMMOCO_CHAIN_ITERS=1 (environment, read once; OFF by default - measured slower): inside a graph of
  // several iterations the image chain does not join the motion chain at the END of an iteration: the iteration
  // counter's tick moves to the side stream (behind the image Adam, and behind the motion Adam by an event), the
  // main stream goes from the motion Adam of iteration k straight into the motion forward of k + 1 and the side stream
  // from the tick into the image forward of k + 1; the chains only meet where data does.  That takes the image chain's
  // ~50 us tail and the join / tick / fork gaps off the critical path - and puts the tail (image encode backward,
  // image Adam) BESIDE the next motion encode forward, whose 4 MB level slices then fall out of the XCD L2s
  // (DESIGN.md 4.1): 1.254 -> 1.371 ms per iteration in fp32, 1.029 -> 1.070 with fp16 MLPs.
  static const bool chain_iters = [] { const char* e = immoco_diag_env("IMMOCO_CHAIN_ITERS"); return e && atoi(e) != 0; }();
  std::vector<Step> chained = first;
  for (Step& x : chained)
    if (strcmp(x.name, "tick") == 0) x.branch = 2;
  const StepHook tick_hook([s](const Step& st, hipStream_t sq, bool before) -> int {
    if (strcmp(st.name, "adam_motion") == 0 && !before) IMMOCO_CHECK_HIP(hipEventRecord(s->ev_fj[7], sq));
    if (strcmp(st.name, "tick") == 0 && before) IMMOCO_CHECK_HIP(hipStreamWaitEvent(sq, s->ev_fj[7], 0));
    return IMMOCO_OK;
  });
  auto capture = [&](const std::vector<Step>& l, hipGraphExec_t* out, int reps) {
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal) != hipSuccess) return;
    int r = IMMOCO_OK;
    const bool chain = chain_iters && reps > 1 && !pipelined && !s->cfg.serial_chains && s->cfg.nM > 0;
    for (int k = 0; k < reps && r == IMMOCO_OK; ++k)
      r = chain ? run_steps_forked(s, chained, q, &tick_hook, k == reps - 1) : run_list(l);
    const hipError_t e2 = hipStreamEndCapture(q, &graph);
    if (r == IMMOCO_OK && e2 == hipSuccess && graph)
      if (hipGraphInstantiate(out, graph, nullptr, nullptr, 0) != hipSuccess) *out = nullptr;
    if (graph) hipGraphDestroy(graph);
  };
  const int GK = pipelined ? 1 : graph_k();
  s->graph_active = 0;
  if (s->cfg.use_graph) {
    std::vector<const void*> key = {kspace_in,  col_group,   params_image, params_motion,
                                    adam_image, adam_motion, loss_hist,    s->sched, pipelined ? s : nullptr};
    sweep_retired(s, false);
    const bool want_k = !pipelined && GK > 1 && iters >= GK;
    if (!s->gexec || key != s->gkey || (pipelined && iters > 1 && !s->gexec2) || (want_k && !s->gexec2)) {
      retire_graph(s, RETIRE_SINGLE);
      capture(first, &s->gexec, 1);
      if (pipelined && iters > 1 && s->gexec) capture(steady, &s->gexec2, 1);
      if (want_k && s->gexec) capture(first, &s->gexec2, GK);   // classic order: gexec2 = GK iterations
      (void)hipGetLastError();
      s->gkey = key;
    }
    s->graph_active = s->gexec != nullptr && (!pipelined || iters == 1 || s->gexec2 != nullptr);
  }
  for (int j = 0; j < iters;) {
    const bool st = pipelined && j > 0;
    if (s->graph_active) {
      if (!pipelined && s->gexec2 && GK > 1 && iters - j >= GK) {
        IMMOCO_CHECK_HIP(hipGraphLaunch(s->gexec2, q));
        j += GK;
        continue;
      }
      IMMOCO_CHECK_HIP(hipGraphLaunch(st ? s->gexec2 : s->gexec, q));
    } else if ((rc = run_list(st ? steady : first))) {
      return rc;
    }
    ++j;
  }
  if (pipelined && (rc = run_steps_forked(s, arrange(all, Order::Epilogue), q))) return rc;   // Adam steps of the last iteration
  if ((rc = finish_slice(s, q, a))) return rc;
  return sync_out ? leave(s, caller) : IMMOCO_OK;
}

// ---- two slices, one graph (BASELINE config 3, cfg.batch_pair) ------------------------------------------------
// Slices A and B (workspaces `A` = the solver itself and `B` = its first lane) advance through the SAME iteration
// index inside one captured graph.  Two slices merely kept in flight side by side (batch_lanes, round 2) are
// slower than slice after slice: their hash-grid gathers evict each other's level slices from the XCD L2s.  Here
// the four gather kernels of a double iteration - motion encode forward (E) and backward (C) of either slice -
// run on ONE stream in the fixed order  E_A, E_B, C_A, C_B, E_A(next) ...  so that never two of them run at once,
// and everything else of a slice (image chain, MLPs on the matrix cores, warp, FFTs, losses, Adam) runs on that
// slice's own stream BESIDE the other slice's gather, in the issue slots and HBM bandwidth the gather leaves idle:
//     q  : E_A ------ E_B ------ C_A ------ C_B ------ E_A' ...
//     pA : img_A  |E_A| N_A --------|  |C_A| D_A img_A' ...
//     pB :      img_B      |E_B| N_B --------|  |C_B| D_B ...
// (N = motion MLP forward, warp, FFTs, losses, warp backward, motion MLP backward; D = Adam steps + image backward).
// Every dependency is an event between q (the capture's origin stream) and one of the two side streams, and a side
// stream only continues after it has waited for q again: the fork / join / fork-again pattern of the single-slice
// graph (HIP's stream capture crashes in hipStreamEndCapture on nested forks and on sibling-to-sibling waits).
int solve_pair(immoco_solver* A, immoco_solver* B, const SliceArgs& a, const SliceArgs& b, int32_t iters, float lr,
               const float* lambda_sched, int32_t step0) {
  int rc;
  if ((rc = check_slice(A, a, iters, lambda_sched, step0)) || (rc = check_slice(B, b, iters, lambda_sched, step0))) return rc;
  IMMOCO_REQUIRE(A->cfg.nM > 0, "solver_solve_batch: paired mode needs motion groups");
  hipStream_t q = A->stream;
  for (hipEvent_t& ev : A->ev_pair)
    if (!ev) IMMOCO_CHECK_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  // all per-solve initialisation of both workspaces on A's stream
  if ((rc = prepare_slice(A, q, a, iters, lr, lambda_sched, step0)) ||
      (rc = prepare_slice(B, q, b, iters, lr, lambda_sched, step0)))
    return rc;
  const Bind ba{A->kin_t, a.cg, a.pi, a.pm, a.ai, a.am, a.loss}, bb{B->kin_t, b.cg, b.pi, b.pm, b.ai, b.am, b.loss};
  struct Lists {
    std::vector<Step> img, e, n, c, d;   // image forward | E | N | C | D
    hipStream_t p;
    hipEvent_t ev_pre, ev_e, ev_n, ev_c, ev_d, ev_m;   // ev_m: the motion Adam step is done (all the next E waits for)
  };
  auto make_lists = [](const std::vector<Step>& all, hipStream_t p, hipEvent_t* ev) {
    Lists l;
    for (const Step& st : all) {
      if (st.group == 0) l.img.push_back(st);
      else if (strcmp(st.name, "motion_encode_fwd") == 0) l.e.push_back(st);
      else if (st.group == 1 || st.group == 2) l.n.push_back(st);   // motion_mlp_fwd, then the serial middle
      else if (st.group == 3) l.c.push_back(st);
    }
    for (int g : {6, 4, 5})   // motion Adam first (the next E_A waits for it), then the image backward chain, tick
      for (const Step& st : all)
        if (st.group == g) l.d.push_back(st);
    l.p = p;
    l.ev_pre = ev[0]; l.ev_e = ev[1]; l.ev_n = ev[2]; l.ev_c = ev[3]; l.ev_d = ev[4]; l.ev_m = ev[5];
    return l;
  };
  const Lists LA = make_lists(build_steps(A, ba, true), A->side, A->ev_pair), LB = make_lists(build_steps(B, bb, true), B->side, A->ev_pair + 6);
  // One double iteration is  E_A N_A | E_B N_B | C_A D_A | C_B D_B.  Round 4: D_B (slice B's Adam steps and image backward
  // chain, ~230 us) is CAPTURED after the next double iteration's E_A / N_A: rocprofv3 showed the replayed graph starting
  // E_A(k+1) only after D_B(k)'s last kernel (profiles/r04_timeline_pair.txt: the gather stream idled 245 us per double
  // iteration there) although E_A(k+1) depends on D_A(k) alone - the graph executor serialises branches in capture
  // order.  With the rotated capture order E_A(k+1) runs beside D_B(k).
  // The next motion encode forward needs the motion Adam step only (ev_m, recorded below), but waiting for just that
  // is SLOWER: the forward gather then runs beside its own slice's image encode backward and Adam, whose streams push its
  // level slices out of the XCD L2s (measured, batch 4: 1.113 -> 1.155 ms per slice-iteration in fp32, 0.892 -> 0.925 with
  // fp16 MLPs).  It waits for the whole D (ev_d).
  auto encode_fwd = [&](const Lists& l, bool again) -> int {   // q: [wait D(prev)] ; pre ; E ; record | p: wait pre ; img
    if (again) IMMOCO_CHECK_HIP(hipStreamWaitEvent(q, l.ev_d, 0));
    IMMOCO_CHECK_HIP(hipEventRecord(l.ev_pre, q));
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(l.p, l.ev_pre, 0));
    int rr;
    if ((rr = run_steps(l.img, l.p)) || (rr = run_steps(l.e, q))) return rr;
    IMMOCO_CHECK_HIP(hipEventRecord(l.ev_e, q));
    return IMMOCO_OK;
  };
  auto middle = [&](const Lists& l) -> int {       // p: wait E ; N ; record
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(l.p, l.ev_e, 0));
    int rr;
    if ((rr = run_steps(l.n, l.p))) return rr;
    IMMOCO_CHECK_HIP(hipEventRecord(l.ev_n, l.p));
    return IMMOCO_OK;
  };
  auto encode_bwd_c = [&](const Lists& l) -> int {   // q: wait N ; C ; record
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(q, l.ev_n, 0));
    int rr;
    if ((rr = run_steps(l.c, q))) return rr;
    IMMOCO_CHECK_HIP(hipEventRecord(l.ev_c, q));
    return IMMOCO_OK;
  };
  auto encode_bwd_d = [&](const Lists& l) -> int {   // p: wait C ; D ; record
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(l.p, l.ev_c, 0));
    bool marked = false;
    for (const Step& st : l.d) {
      int rr = st.run(l.p);
      if (rr) return rr;
      if (!marked && st.group == 6) {   // the motion Adam step comes first in D
        IMMOCO_CHECK_HIP(hipEventRecord(l.ev_m, l.p));
        marked = true;
      }
    }
    if (!marked) IMMOCO_CHECK_HIP(hipEventRecord(l.ev_m, l.p));
    IMMOCO_CHECK_HIP(hipEventRecord(l.ev_d, l.p));
    return IMMOCO_OK;
  };
  // `again`: an earlier double iteration of THIS capture / eager sequence has recorded the ev_d events (a captured
  // stream may not wait for an event recorded outside the capture; between launches the join below orders everything)
  auto sequence = [&](int reps, bool again0 = false) -> int {
    int r;
    if ((r = encode_fwd(LA, again0)) || (r = middle(LA))) return r;
    for (int k = 0; k < reps; ++k) {
      if ((r = encode_fwd(LB, again0 || k > 0)) || (r = middle(LB)) || (r = encode_bwd_c(LA)) || (r = encode_bwd_d(LA)) ||
          (r = encode_bwd_c(LB)))
        return r;
      if (k + 1 < reps && ((r = encode_fwd(LA, true)) || (r = middle(LA)))) return r;   // E_A(k+1) ahead of D_B(k)
      if ((r = encode_bwd_d(LB))) return r;
    }
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(q, LA.ev_d, 0));   // both side streams join the origin
    IMMOCO_CHECK_HIP(hipStreamWaitEvent(q, LB.ev_d, 0));
    return IMMOCO_OK;
  };
  auto capture = [&](hipGraphExec_t* out, int reps) {
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(q, hipStreamCaptureModeThreadLocal) != hipSuccess) return;
    const int r = sequence(reps);
    const hipError_t e2 = hipStreamEndCapture(q, &graph);
    if (r == IMMOCO_OK && e2 == hipSuccess && graph)
      if (hipGraphInstantiate(out, graph, nullptr, nullptr, 0) != hipSuccess) *out = nullptr;
    if (graph) hipGraphDestroy(graph);
  };
  const int GK = graph_k();
  bool graph_ok = false;
  if (A->cfg.use_graph) {
    std::vector<const void*> key = {a.kin, a.cg, a.pi, a.pm, a.ai, a.am, a.loss, A->sched,
                                    b.kin, b.cg, b.pi, b.pm, b.ai, b.am, b.loss, B->sched, B};
    sweep_retired(A, false);
    if (!A->pg1 || key != A->pkey || (iters >= GK && GK > 1 && !A->pgk)) {
      retire_graph(A, RETIRE_PAIR);
      capture(&A->pg1, 1);
      if (A->pg1 && GK > 1 && iters >= GK) capture(&A->pgk, GK);
      (void)hipGetLastError();
      A->pkey = key;
    }
    graph_ok = A->pg1 != nullptr;
  }
  A->graph_active = graph_ok ? 1 : 0;
  for (int j = 0; j < iters;) {
    if (graph_ok && A->pgk && GK > 1 && iters - j >= GK) {
      IMMOCO_CHECK_HIP(hipGraphLaunch(A->pgk, q));
      j += GK;
    } else if (graph_ok) {
      IMMOCO_CHECK_HIP(hipGraphLaunch(A->pg1, q));
      ++j;
    } else {
      if ((rc = sequence(1))) return rc;
      ++j;
    }
  }
  if ((rc = finish_slice(A, q, a)) || (rc = finish_slice(B, q, b))) return rc;
  return IMMOCO_OK;
}
}  // namespace

extern "C" int immoco_solver_solve(immoco_solver_t s, const float* kspace_in, const int32_t* col_group,
                                   float* params_image,
                                   float* params_motion, float* adam_image, float* adam_motion, int32_t iters,
                                   float lr, const float* lambda_sched, int32_t step0, float* out_image,
                                   float* out_kspace, float* loss_hist, void* stream) {
  return solve_impl(s, kspace_in, col_group, params_image, params_motion, adam_image, adam_motion, iters, lr,
                    lambda_sched, step0, out_image, out_kspace, loss_hist, stream, true, true);
}

// Batch of B independent slices of one shape (BASELINE config 3).  cfg.batch_lanes slices are in flight side
// by side: lane k is a full solver workspace (activations, gradient tables, streams, captured graph) that borrows
// the parent's lattices and transposed indices; slice i is queued on lane i % lanes.  Nothing in a solve waits on
// the host once the schedules are on the device, so the lanes' iteration graphs interleave on the GPU on their
// own: while one slice sits in an L2-bandwidth-bound hash-grid gather, another runs its MFMA- and HBM-bound
// kernels (MLPs, Adam, FFTs).
namespace {
int make_lane(immoco_solver* parent, immoco_solver** out) {
  immoco_solver_cfg cfg = parent->cfg;
  cfg.batch_lanes = 0;
  cfg.batch_pair = 0;
  immoco_solver* lane = nullptr;
  int rc = immoco_solver_create(&cfg, &lane);
  if (rc) return rc;
  // borrow lattices and plans (freed by the parent only)
  for (float* b : {lane->xs, lane->ys, lane->ms})
    if (b) hipFree(b);
  lane->xs = parent->xs;
  lane->ys = parent->ys;
  lane->ms = parent->ms;
  lane->plan_img = parent->plan_img;
  lane->plan_mot = parent->plan_mot;
  lane->lattice_set = true;
  lane->parent = parent;
  *out = lane;
  return IMMOCO_OK;
}
}  // namespace

extern "C" int immoco_solver_solve_batch(immoco_solver_t s, int32_t B, const float* kspace_in,
                                         const int32_t* col_group, float* params_image, float* params_motion,
                                         float* adam_image, float* adam_motion, int32_t iters, float lr,
                                         const float* lambda_sched, int32_t step0, float* out_image,
                                         float* out_kspace, float* loss_hist, void* stream) {
  IMMOCO_REQUIRE(s, "solver_solve_batch: NULL solver");
  IMMOCO_REQUIRE(B >= 0, "solver_solve_batch: negative batch size %d", B);
  IMMOCO_REQUIRE(B == 0 || s->lattice_set, "solver_solve_batch: call immoco_solver_set_lattice first");
  const int64_t P2 = 2 * s->P, W = s->cfg.W;
  if (s->cfg.batch_pair && s->cfg.nM > 0 && B >= 2) {
    if (s->lanes.empty()) {
      immoco_solver* lane = nullptr;
      int rc = make_lane(s, &lane);
      if (rc) return rc;
      s->lanes.push_back(lane);
    }
    auto args = [&](int32_t i) {
      return SliceArgs{kspace_in ? kspace_in + i * P2 : nullptr, col_group ? col_group + i * W : nullptr,
                       params_image ? params_image + i * s->n_params_img : nullptr,
                       params_motion ? params_motion + i * s->n_params_mot : nullptr,
                       adam_image ? adam_image + 2 * i * s->n_params_img : nullptr,
                       adam_motion ? adam_motion + 2 * i * s->n_params_mot : nullptr,
                       out_image ? out_image + i * P2 : nullptr, out_kspace ? out_kspace + i * P2 : nullptr,
                       loss_hist ? loss_hist + (int64_t)i * iters : nullptr};
    };
    int rc = enter(s, as_stream(stream));
    for (int32_t i = 0; rc == IMMOCO_OK && i + 1 < B; i += 2)
      rc = solve_pair(s, s->lanes[0], args(i), args(i + 1), iters, lr, lambda_sched, step0);
    if (rc == IMMOCO_OK && (B & 1)) {
      const SliceArgs l = args(B - 1);
      rc = solve_impl(s, l.kin, l.cg, l.pi, l.pm, l.ai, l.am, iters, lr, lambda_sched, step0, l.out_image, l.out_kspace,
                      l.loss, stream, false, false);
    }
    const int rc2 = leave(s, as_stream(stream));   // also on errors: queued work still uses the caller's buffers
    return rc ? rc : rc2;
  }
  const int n_lanes = std::max(1, std::min<int>(s->cfg.batch_lanes, B));
  while ((int)s->lanes.size() < n_lanes - 1) {  // lane 0 is the solver itself
    immoco_solver* lane = nullptr;
    int rc = make_lane(s, &lane);
    if (rc) return rc;
    s->lanes.push_back(lane);
  }
  // every lane starts after what the caller's stream holds now; the caller's stream waits for all lanes at the
  // end (a wait per slice would serialise the lanes through the caller's stream)
  for (int32_t i = 0; i < B; ++i) {
    immoco_solver* lane = (i % n_lanes) == 0 ? s : s->lanes[(i % n_lanes) - 1];
    int rc = solve_impl(
        lane, kspace_in ? kspace_in + i * P2 : nullptr, col_group ? col_group + i * W : nullptr,
        params_image ? params_image + i * s->n_params_img : nullptr,
        params_motion ? params_motion + i * s->n_params_mot : nullptr,
        adam_image ? adam_image + 2 * i * s->n_params_img : nullptr,
        adam_motion ? adam_motion + 2 * i * s->n_params_mot : nullptr, iters, lr, lambda_sched, step0,
        out_image ? out_image + i * P2 : nullptr, out_kspace ? out_kspace + i * P2 : nullptr,
        loss_hist ? loss_hist + (int64_t)i * iters : nullptr, stream, i < n_lanes, false);
    if (rc) {
      // lanes that already have slices queued still read and write the caller's buffers: order the caller's
      // stream after them before the error goes back (the error string of `rc` is kept)
      for (int k = 0; k < n_lanes && k <= i; ++k) {
        immoco_solver* l = k == 0 ? s : s->lanes[k - 1];
        if (hipEventRecord(l->ev_out, l->stream) == hipSuccess) (void)hipStreamWaitEvent(as_stream(stream), l->ev_out, 0);
      }
      (void)hipGetLastError();
      return rc;
    }
  }
  for (int k = 0; k < n_lanes && k < B; ++k) {
    int rc = leave(k == 0 ? s : s->lanes[k - 1], as_stream(stream));
    if (rc) return rc;
  }
  return IMMOCO_OK;
}

extern "C" int64_t immoco_solver_plan_entries(immoco_solver_t s, int32_t which) {
  if (!s) return -1;
  return csr_plan_entries(which == 0 ? s->plan_img : s->plan_mot);
}

extern "C" int immoco_solver_forward(immoco_solver_t s, const int32_t* col_group, const float* params_image,
                                     const float* params_motion, float* out_image, float* out_kspace,
                                     void* stream) {
  IMMOCO_REQUIRE(s && col_group && params_image, "solver_forward: NULL argument");
  IMMOCO_REQUIRE(s->lattice_set, "solver_forward: call immoco_solver_set_lattice first");
  IMMOCO_REQUIRE(s->cfg.nM == 0 || params_motion, "solver_forward: NULL motion buffer");
  hipStream_t caller = as_stream(stream);
  int rc;
  if ((rc = enter(s, caller))) return rc;
  hipStream_t q = s->stream;
  IMMOCO_CHECK_HIP(hipMemsetAsync(s->iter_dev, 0, 4 * sizeof(int32_t), q));
  // kin only feeds the (unused) residual here: point it at kout itself
  if ((rc = refresh_shadows(s, params_image, params_motion, q))) return rc;
  if (s->pruned && (rc = launch_build_col_lists(col_group, s->cfg.nM, s->cfg.W, s->pr_cols, s->pr_off, q))) return rc;
  Bind b{s->kout, col_group, const_cast<float*>(params_image), const_cast<float*>(params_motion),
         nullptr, nullptr, nullptr};
  std::vector<Step> steps = build_steps(s, b, false);
  if ((rc = run_steps(steps, q))) return rc;
  if (out_image) IMMOCO_CHECK_HIP(hipMemcpyAsync(out_image, s->image, (size_t)s->P * 8, hipMemcpyDeviceToDevice, q));
  if (out_kspace && (rc = launch_transpose_c64(s->kout, out_kspace, s->cfg.W, s->cfg.H, q))) return rc;
  return leave(s, caller);
}

// Times every step of one iteration with HIP events on the solver's own stream (eager launches,
// `reps` repetitions, parameters and Adam state are advanced like in a real solve).
extern "C" int immoco_solver_profile(immoco_solver_t s, const float* kspace_in, const int32_t* col_group,
                                     float* params_image,
                                     float* params_motion, float* adam_image, float* adam_motion, int32_t reps,
                                     float lr, float lambda_ge, void* stream) {
  IMMOCO_REQUIRE(s && reps >= 1 && s->lattice_set, "solver_profile: bad argument / lattice not set");
  hipStream_t caller = as_stream(stream);
  int rc;
  if ((rc = ensure_sched(s, reps))) return rc;
  std::vector<float> sched(2 * (size_t)reps), lam((size_t)reps, lambda_ge);
  for (int j = 0; j < reps; ++j) {
    const double t = (double)(j + 1);
    sched[2 * j] = (float)((double)lr / (1.0 - pow(0.9, t)));
    sched[2 * j + 1] = (float)sqrt(1.0 - pow(0.999, t));
  }
  if ((rc = enter(s, caller))) return rc;
  hipStream_t q = s->stream;
  s->sched_host.clear();
  IMMOCO_CHECK_HIP(hipMemcpyAsync(s->sched, sched.data(), sched.size() * 4, hipMemcpyHostToDevice, q));
  IMMOCO_CHECK_HIP(hipMemcpyAsync(s->lambda_dev, lam.data(), lam.size() * 4, hipMemcpyHostToDevice, q));
  IMMOCO_CHECK_HIP(hipMemsetAsync(s->iter_dev, 0, 4 * sizeof(int32_t), q));
  IMMOCO_CHECK_HIP(hipStreamSynchronize(q));
  if ((rc = refresh_shadows(s, params_image, params_motion, q))) return rc;
  if (s->pruned && (rc = launch_build_col_lists(col_group, s->cfg.nM, s->cfg.W, s->pr_cols, s->pr_off, q))) return rc;
  if ((rc = launch_transpose_c64(kspace_in, s->kin_t, s->cfg.H, s->cfg.W, q))) return rc;
  Bind b{s->kin_t, col_group, params_image, params_motion, adam_image, adam_motion, nullptr};
  std::vector<Step> steps = build_steps(s, b, true);
  const size_t n = steps.size();
  std::vector<hipEvent_t> ev((n + 1) * (size_t)reps);
  for (auto& e : ev) IMMOCO_CHECK_HIP(hipEventCreate(&e));
  for (int r = 0; r < reps; ++r) {
    for (size_t i = 0; i < n; ++i) {
      IMMOCO_CHECK_HIP(hipEventRecord(ev[r * (n + 1) + i], q));
      if ((rc = steps[i].run(q))) return rc;
    }
    IMMOCO_CHECK_HIP(hipEventRecord(ev[r * (n + 1) + n], q));
  }
  IMMOCO_CHECK_HIP(hipStreamSynchronize(q));
  s->phase_names.clear();
  s->phase_ms.assign(n, 0.f);
  for (size_t i = 0; i < n; ++i) s->phase_names.push_back(steps[i].name);
  for (int r = 0; r < reps; ++r)
    for (size_t i = 0; i < n; ++i) {
      float ms_ = 0.f;
      IMMOCO_CHECK_HIP(hipEventElapsedTime(&ms_, ev[r * (n + 1) + i], ev[r * (n + 1) + i + 1]));
      s->phase_ms[i] += ms_ / (float)reps;
    }
  for (auto& e : ev) hipEventDestroy(e);
  return leave(s, caller);
}

extern "C" int immoco_solver_phase_times(immoco_solver_t s, const char** names, float* ms, int32_t max_n) {
  if (!s) return 0;
  int n = (int)s->phase_names.size();
  if (n > max_n) n = max_n;
  for (int i = 0; i < n; ++i) {
    if (names) names[i] = s->phase_names[i].c_str();
    if (ms) ms[i] = s->phase_ms[i];
  }
  return n;
}

extern "C" int immoco_solver_graph_active(immoco_solver_t s) { return s ? s->graph_active : 0; }

// Duration (ms) of the motion-grid encode backward kernel in the LAST iteration of the last solve,
// measured by HIP events recorded around it inside the replayed graph (concurrent branches and
// all).  Returns < 0 when no solve with a motion grid has run.  Synchronises the solver stream.
extern "C" float immoco_solver_dominant_kernel_ms(immoco_solver_t s) {
  if (!s || !s->marked) return -1.f;
  if (hipStreamSynchronize(s->stream) != hipSuccess) return -1.f;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, s->ev_k0, s->ev_k1) != hipSuccess) {
    (void)hipGetLastError();
    return -1.f;
  }
  return ms;
}

// Toggles graph replay at run time (1: replay a captured graph, 0: launch eagerly on the same two
// streams).  bench.py uses the eager mode for a short pass whose event markers around the dominant
// kernel are real (HIP cannot read elapsed time from events recorded as graph nodes).
extern "C" int immoco_solver_set_graph(immoco_solver_t s, int32_t use_graph) {
  IMMOCO_REQUIRE(s, "solver_set_graph: NULL solver");
  s->cfg.use_graph = use_graph ? 1 : 0;
  return IMMOCO_OK;
}

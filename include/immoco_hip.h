/*
 * immoco_hip.h — C-ABI of the MI355X-native IM-MoCo inner-loop library
 * (libimmoco_hip.so, built from miccai24_immoco_amd/csrc for gfx950).
 *
 * The reference (multimodallearning/MICCAI24_IMMoCo) is pure Python and has no
 * FFI of its own; the native code it reaches on this path is tiny-cuda-nn's
 * torch binding, ATen (grid_sample, foreach Adam) and cuFFT.  Each entry point
 * below names the reference interface (file:line under /root/reference) it
 * replaces.  Conventions:
 *   - plain pointers and sizes only; every buffer is a caller-allocated DEVICE
 *     pointer unless marked [host];
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls are
 *     asynchronous on that stream unless stated otherwise;
 *   - return 0 on success, negative on error (IMMOCO_E_*), message via
 *     immoco_last_error(); no exceptions cross the ABI;
 *   - complex data is interleaved float (re, im), i.e. torch.complex64 memory;
 *   - "accumulates" means the kernel ADDS into the output (caller zeroes it).
 */
#ifndef IMMOCO_HIP_H
#define IMMOCO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IMMOCO_OK 0
#define IMMOCO_E_INVALID (-1) /* bad argument / unsupported configuration */
#define IMMOCO_E_HIP (-2)     /* HIP runtime error                         */
#define IMMOCO_E_FFT (-3)     /* rocFFT/hipFFT error                       */

#define IMMOCO_MAX_LEVELS 16
#define IMMOCO_ACT_RELU 0
#define IMMOCO_ACT_TANH 1

/* tiny-cuda-nn "Grid"/"Hash"/"Linear" encoding config (immoco.py:27-37). */
typedef struct immoco_grid_cfg {
  int32_t dims;               /* 2 (image INR, immoco.py:60) or 3 (motion INR, :63) */
  int32_t n_levels;           /* <= IMMOCO_MAX_LEVELS */
  int32_t n_features;         /* must be 2 */
  int32_t log2_hashmap_size;
  int32_t base_resolution;
  float per_level_scale;
} immoco_grid_cfg;

/* Derived level geometry (tiny-cuda-nn grid.h). [host] */
typedef struct immoco_grid_geometry {
  uint32_t offset[IMMOCO_MAX_LEVELS + 1]; /* entry offsets, [n_levels] = total entries */
  uint32_t resolution[IMMOCO_MAX_LEVELS];
  uint32_t size[IMMOCO_MAX_LEVELS];
  float scale[IMMOCO_MAX_LEVELS];
  uint8_t hashed[IMMOCO_MAX_LEVELS];
} immoco_grid_geometry;

/* tiny-cuda-nn MLP config with one hidden layer, no biases (immoco.py:11-25). */
typedef struct immoco_mlp_cfg {
  int32_t n_in;         /* 32 = n_levels * n_features */
  int32_t n_hidden;     /* 256 (image) or 64 (motion) */
  int32_t n_out;        /* 2 */
  int32_t n_out_padded; /* 8 (CutlassMLP) or 16 (FullyFusedMLP) rows stored in W2 */
  int32_t activation;   /* IMMOCO_ACT_* */
} immoco_mlp_cfg;

/* ---- library ------------------------------------------------------------ */
int immoco_version(void);
/* Copies the calling thread's last error message into buf (NUL terminated). */
int immoco_last_error(char* buf, size_t n);

/* ---- INR: hash grid (replaces tinycudann.NetworkWithInputEncoding's encoding,
 *      immoco.py:60-65,85,93) --------------------------------------------- */
int immoco_grid_geometry_query(const immoco_grid_cfg* cfg, immoco_grid_geometry* out /*[host]*/);
/* enc[p*enc_point_stride + l*enc_level_stride + f] for p<n, l<n_levels, f<2.
 * tcnn layout [n,32]: point_stride=32, level_stride=2; solver layout [L][n][2]:
 * point_stride=2, level_stride=2n.  table: [n_entries][2] fp32. */
int immoco_hashgrid_fwd(const immoco_grid_cfg* cfg, const float* coords /*[n,dims]*/, int64_t n,
                        const float* table, float* enc, int64_t enc_point_stride,
                        int64_t enc_level_stride, void* stream);
/* The same from an fp16 table ([n_entries][2] half): tiny-cuda-nn's own parameter precision (SURVEY §8b
 * "fp32 and fp16 tables"); each entry is widened to fp32, then the arithmetic is that of immoco_hashgrid_fwd.
 * Gradients always go to an fp32 table (immoco_hashgrid_bwd). */
int immoco_hashgrid_fwd_f16(const immoco_grid_cfg* cfg, const float* coords /*[n,dims]*/, int64_t n,
                            const void* table_f16, float* enc, int64_t enc_point_stride,
                            int64_t enc_level_stride, void* stream);
/* The same encoding for the LATTICE the reference always queries (immoco.py:48-53,72-80) given by its axes
 * instead of n coordinate rows: dims = 3: point (m, row, col) = (ax0[m], ax1[row], ax2[col]), n = nM*H*W
 * points in (m, row, col) order; dims = 2: point (row, col) = (x = ax0[col], y = ax1[row]), nM = 1, ax2 unused
 * (the conventions of immoco_grid_plan_create).  Bit-identical to immoco_hashgrid_fwd on the expanded
 * coordinates; the solver's path (wave-uniform motion group: merged / lane-paired dim-0 corner loads). */
int immoco_hashgrid_fwd_lattice(const immoco_grid_cfg* cfg, int32_t nM, int32_t H, int32_t W, const float* ax0,
                                const float* ax1, const float* ax2, const float* table, float* enc,
                                int64_t enc_point_stride, int64_t enc_level_stride, void* stream);
/* Accumulates dtable[n_entries][2] += scatter(denc) (same strides as fwd). */
int immoco_hashgrid_bwd(const immoco_grid_cfg* cfg, const float* coords, int64_t n,
                        const float* denc, int64_t enc_point_stride, int64_t enc_level_stride,
                        float* dtable, void* stream);

/* ---- atomic-free hash-grid backward for a FIXED lattice (the reference always queries
 *      linspace lattices, immoco.py:48-53,72-80): a transposed index (table slot -> list
 *      of contributing lattice points) is built once, then every backward is a gather.
 *  dims 3: point (m,row,col) has coordinates (ax0[m], ax1[row], ax2[col]) (make_grids order);
 *  dims 2: nM must be 1, point (row,col) has coordinates (ax0[col], ax1[row]) (identy_grid
 *          order), ax2 unused.  The axis arrays (device) must outlive the plan, unchanged. */
typedef struct immoco_grid_plan* immoco_grid_plan_t;
int immoco_grid_plan_create(const immoco_grid_cfg* cfg, int32_t nM, int32_t H, int32_t W,
                            const float* ax0, const float* ax1, const float* ax2,
                            immoco_grid_plan_t* out, void* stream);
int immoco_grid_plan_destroy(immoco_grid_plan_t p);
int64_t immoco_grid_plan_bytes(immoco_grid_plan_t p);
/* denc: level-major [n_levels][nM*H*W][2]; dtable [n_entries][2] ACCUMULATES. */
int immoco_grid_plan_bwd(immoco_grid_plan_t p, const float* denc_level_major, float* dtable,
                         void* stream);

/* ---- INR: bias-free one-hidden-layer MLP (tcnn CutlassMLP / FullyFusedMLP) */
/* w1 [n_hidden][n_in], w2 [n_out_padded][n_hidden] row-major; out [n][n_out]. */
int immoco_mlp_fwd(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                   int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                   float* out, void* stream);
/* din written (same strides as in); dw1/dw2 ACCUMULATE (padded rows of dw2 untouched). */
int immoco_mlp_bwd(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                   int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                   const float* dout /*[n][n_out]*/, float* din, float* dw1, float* dw2,
                   void* stream);
/* The same backward of the 256-wide net (the image INR's CutlassMLP, /root/reference/src/models/immoco.py:11-17,60-62)
 * as the TWO kernels the fused solver runs beside the motion grid's encode backward: din (must NOT alias in) by one,
 * dw1 / dw2 (accumulated) by the other.  n_hidden must be 256. */
int immoco_mlp_bwd_split(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                         int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                         const float* dout /*[n][n_out]*/, float* din, float* dw1, float* dw2,
                         void* stream);

/* The same MLP in tiny-cuda-nn's OWN network precision (the reference instantiates FullyFusedMLP / CutlassMLP with
 * __half, /root/reference/src/models/immoco.py:11-25,60-65; tcnn torch binding: loss_scale 128): fp16 OPERANDS -
 * the encoding, W1, the hidden activations, W2, dout * loss_scale, dL/dpre - with fp32 accumulation on
 * v_mfma_f32_32x32x16_f16; buffers stay fp32 (weights are rounded on the fly), outputs / din / dw1 / dw2 are fp32
 * and unscaled.  Same layouts and accumulate semantics as immoco_mlp_fwd / immoco_mlp_bwd. */
int immoco_mlp_fwd_half(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                        int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                        float* out, void* stream);
int immoco_mlp_bwd_half(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                        int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                        const float* dout /*[n][n_out]*/, float loss_scale, float* din, float* dw1,
                        float* dw2, void* stream);

/* The same MLP at (nearly) fp32 accuracy on the 16-bit matrix cores: every operand of a matrix product is split into
 * two bf16 terms (16 significant bits, fp32's exponent range: nothing is scaled), a product is three
 * v_mfma_f32_32x32x16_bf16 accumulated in fp32; activations, their derivatives and all sums are fp32.  Relative error
 * of a product <= 2^-16.5 (fp16 operands: 2^-11; the reference's tiny-cuda-nn networks run in fp16,
 * /root/reference/src/models/immoco.py:11-25,60-65).  Same layouts and accumulate semantics as immoco_mlp_fwd / _bwd. */
int immoco_mlp_fwd_bf16x2(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                          int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                          float* out, void* stream);
int immoco_mlp_bwd_bf16x2(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                          int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                          const float* dout /*[n][n_out]*/, float* din, float* dw1, float* dw2, void* stream);

/* ---- parameter init (tcnn: encoding U(-1e-4,1e-4), MLP Xavier-uniform on the
 *      padded shapes; counter-based generator shared bit-exactly with the oracle) */
int immoco_init_params(const immoco_grid_cfg* grid, const immoco_mlp_cfg* mlp, uint32_t seed,
                       float* params /*[n_w1 + n_w2 + 2*n_entries]*/, void* stream);

/* ---- warp: F.grid_sample(bilinear, zeros, align_corners=False) of ONE complex
 *      image at nM sampling grids (immoco.py:91,97-107) -------------------- */
int immoco_warp_fwd(const float* image /*[H,W] c64*/, const float* grids /*[nM,H,W,2] (x,y)*/,
                    int32_t nM, int32_t H, int32_t W, float* out /*[nM,H,W] c64*/, void* stream);
/* dimage ACCUMULATES; dgrids written. */
int immoco_warp_bwd(const float* image, const float* grids, const float* dout, int32_t nM,
                    int32_t H, int32_t W, float* dimage, float* dgrids, void* stream);

/* ---- motion simulator (src/utils/motion_utils.py:121-202), the input generator of the path ---
 * F.affine_grid(theta, align_corners=True) + F.grid_sample(bilinear, border, align_corners=False)
 * of one complex image for n rigid movements.  theta [n][2][3] fp32 (device), xs[W] / ys[H] the
 * linspace(-1,1,.) base lattices; out [n,H,W] c64. */
int immoco_affine_warp_border(const float* image, const float* theta, const float* xs, const float* ys,
                              int32_t n, int32_t H, int32_t W, float* out, void* stream);
/* kout[..., w0[m]:w1[m]] = kall[m][..., w0[m]:w1[m]] for m = 0..n-1 in order, kout = k0 elsewhere;
 * mask [H,W] int64 (may be NULL) = 1 on replaced columns (motion_utils.py:191-196). */
int immoco_band_replace(const float* k0, const float* kall, const int32_t* w0, const int32_t* w1,
                        int32_t n, int32_t H, int32_t W, float* kout, int64_t* mask, void* stream);

/* ---- Autofocusing baseline (src/models/autofocusing.py:71-85): F.affine_grid(align_corners=True) +
 * F.grid_sample(mode="bicubic", zeros, align_corners=False) of n per-group complex images
 * [n,H,W] c64 under n affine matrices theta [n][2][3]; out [n,H,W] c64. */
int immoco_affine_bicubic_fwd(const float* images, const float* theta, const float* xs, const float* ys,
                              int32_t n, int32_t H, int32_t W, float* out, void* stream);
/* dtheta [n][2][3] ACCUMULATES the gradient w.r.t. the affine matrices (images are constants). */
int immoco_affine_bicubic_bwd(const float* images, const float* theta, const float* xs, const float* ys,
                              const float* dout, int32_t n, int32_t H, int32_t W, float* dtheta,
                              void* stream);

/* ---- centred FFTs (src/utils/data_utils.py:29-34) over the last two dims.
 * mode 0: FFT  = fftshift(fftn(ifftshift(x)))   unnormalised
 * mode 1: IFFT = ifftshift(ifftn(fftshift(x)))  1/(HW)
 * mode 2: adjoint of mode 0 (FFT's shifts around the unnormalised inverse transform; = HW * IFFT for even sizes)
 * mode 3: adjoint of mode 1 (IFFT's shifts around the forward transform, / (HW); = FFT / (HW) for even sizes)
 *         - both used by the backward passes; they differ from IFFT/FFT for odd sizes (different rolls).
 * in may equal out.  Plans are cached per (batch,H,W) inside the library. */
int immoco_fft2c(const float* in, float* out, int32_t batch, int32_t H, int32_t W, int32_t mode,
                 void* stream);

/* ---- k-space line select (immoco.py:109-111) and losses ------------------ */
/* kout[r,c] = kall[col_group[c]][r,c]; kall [(nM+1),H,W] c64, slot 0 = FFT(image). */
int immoco_kspace_select(const float* kall, const int32_t* col_group /*[W]*/, int32_t nM, int32_t H,
                         int32_t W, float* kout, void* stream);
/* F.mse_loss(view_as_real(k), view_as_real(kin)) (immoco.py:170): loss[0] ACCUMULATES
 * sum|k-kin|^2/(2HW); dk = (k-kin)/(HW) written if non-NULL. */
int immoco_dc_loss(const float* k, const float* kin, int32_t H, int32_t W, float* loss, float* dk,
                   void* stream);
/* GradientEntropyLoss (src/utils/losses.py:20-40): loss[0] ACCUMULATES weight*GE(x);
 * dimage (if non-NULL) ACCUMULATES weight*dGE/dx. */
int immoco_ge_loss(const float* image, int32_t H, int32_t W, float weight, float* loss, float* dimage,
                   void* stream);

/* ---- torch.optim.Adam step, betas/eps/no-wd defaults (immoco.py:149-154,175) */
int immoco_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                     float beta2, float eps, int32_t step /*1-based*/, void* stream);

/* ---- line-select masks (src/utils/motion_utils.py:56-109), bit-exact ------ */
/* lines [n] uint8 (0/1) -> col_group [n] int32 (0 = uncorrupted, g>=1 run index);
 * n_groups[0] (device int32) = number of runs. */
int immoco_extract_movement_groups(const uint8_t* lines, int32_t n, int32_t* col_group,
                                   int32_t* n_groups, void* stream);
/* make_list=False: groups [rows,n] int64 = col_group broadcast down rows. */
int immoco_groups_to_matrix(const int32_t* col_group, int32_t rows, int32_t n, int64_t* groups,
                            void* stream);
/* make_list=True: masks [n_groups,rows,n] int64 one-hot. */
int immoco_groups_to_masks(const int32_t* col_group, int32_t n_groups, int32_t rows, int32_t n,
                           int64_t* masks, void* stream);
/* inverse: masks [nM,rows,n] int64 (row 0 is read) -> col_group [n]. */
int immoco_masks_to_groups(const int64_t* masks, int32_t nM, int32_t rows, int32_t n,
                           int32_t* col_group, void* stream);

/* ---- k-space normalisation (immoco.py:137-141): out = k / max|k| * target;
 *      scale_out[0] (device float) = max|k|. */
int immoco_normalize_kspace(const float* k, int64_t n_complex, float target, float* out,
                            float* scale_out, void* stream);

/* ---- fused per-slice solver = imcoco_motion_correction (immoco.py:116-206) */
typedef struct immoco_solver_cfg {
  int32_t H, W, nM;
  immoco_grid_cfg image_grid, motion_grid;
  immoco_mlp_cfg image_mlp, motion_mlp;
  int32_t use_graph;      /* 1: capture one iteration in a hipGraph and replay it */
  int32_t atomic_scatter; /* 1: hash-grid backward by global float atomics (slow reference path
                             kept for A/B measurements); 0: transposed-index gather (default) */
  int32_t grad_parts;     /* point-range parts of the motion grid's transposed index (a power of two <= 256;
                             0 = chosen from the lattice size so that one level slice of dL/denc per
                             part is ~2 MB, the share of an XCD's L2: 4 at 320x320x10, 32 at 640x640x20);
                             up to 8 parts run in one launch, each into its own partial gradient table */
  int32_t serial_chains;  /* 1: run the image-INR and motion-INR kernel chains one after the other; 0: two
                             concurrent branches of the graph; 2: the library decides (the fork: it measures faster
                             at 320x320x10 and at 640x640x20) */
  int32_t table_fp16;     /* 1: gather the hash-grid features from fp16 shadows of the tables (what
                             tiny-cuda-nn does; BASELINE config 5), fp32 master tables + fp32 Adam;
                             default 0: everything fp32 */
  int32_t batch_lanes;    /* immoco_solver_solve_batch keeps this many slices IN FLIGHT side by side (each on its
                             own workspace, streams and captured graph; the transposed indices are shared).
                             0 or 1 (default, and the fastest on MI355X): slice after slice.  Measured at 320x320x10
                             with the streams on separate hardware queues (GPU_MAX_HW_QUEUES=12): 2 lanes 1.55,
                             3 lanes 1.65 ms per slice-iteration against 1.36 serial - two slices' hash-grid
                             gathers evict each other's 4 MB level slices from the XCD L2s */
  int32_t mlp_fp16;       /* MLP arithmetic.  0: exact fp32 on the f32 MFMA.  2: two-term bf16 split of every matrix operand,
                             fp32 accumulation (immoco_mlp_fwd_bf16x2 / _bwd_bf16x2; product error <= 2^-16.5).
                             1: both MLPs in tiny-cuda-nn's network precision (immoco_mlp_fwd_half / _bwd_half: fp16
                             operands, fp32 accumulation, loss scale 128); with table_fp16 this is "tcnn's own
                             arithmetic" (immoco.py:11-25,60-65).  default 0: exact fp32 on the f32 MFMA */
  int32_t batch_pair;     /* immoco_solver_solve_batch: 1 = slices are solved two at a time inside ONE captured graph whose
                             four hash-grid gather kernels (motion encode forward / backward of either slice) are
                             chained by events so that never two of them run at once, while the rest of one slice
                             (MLPs, warp, FFTs, losses, Adam, image chain) runs beside the other slice's gather */
  /* immoco_solver_create validates every field: mlp_fp16 in {0, 1, 2}, batch_pair in {0, 1}, serial_chains in
     {0, 1, 2}, batch_lanes in 0..64.  mlp_fp16 and batch_pair were `reserved[2]` before round 3: callers must
     zero-initialise the struct (IMMOCO_E_INVALID otherwise). */
} immoco_solver_cfg;

typedef struct immoco_solver* immoco_solver_t;

/* Creates plans and the internal workspace (hipMalloc; synchronous). */
int immoco_solver_create(const immoco_solver_cfg* cfg, immoco_solver_t* out);
int immoco_solver_destroy(immoco_solver_t s);
/* Bytes of device memory held by the solver (params/Adam state excluded). */
int64_t immoco_solver_workspace_bytes(immoco_solver_t s);
int64_t immoco_solver_n_params(immoco_solver_t s, int32_t which /*0 image, 1 motion*/);

/* Sets the coordinate lattices once per solver: xs[W], ys[H], ms[nM] fp32 (device) = the
 * reference's linspace(-1,1,.) lattices (immoco.py:48-53,72-80), computed by the caller.
 * Copies them and builds the transposed hash-grid indices (synchronous). */
int immoco_solver_set_lattice(immoco_solver_t s, const float* xs, const float* ys, const float* ms,
                              void* stream);

/* Runs `iters` Adam iterations (immoco.py:164-181) on one slice.
 *  kspace_in  [H,W] c64, ALREADY normalised (immoco_normalize_kspace);
 *  col_group  [W] int32 (0 = FFT(image) column, m>=1 = FFT(warp_m) column);
 *  params_*   flat fp32 [W1|W2|table] (tcnn order), updated in place;
 *  adam_*     [2*n_params] fp32 (m then v), caller-zeroed for a fresh solve;
 *  lambda_sched [host][iters] GE weight used at iteration j (immoco.py:180-181);
 *  out_image / out_kspace [H,W] c64: tensors of the LAST forward pass, i.e.
 *               before the final Adam step (immoco.py:203-206);
 *  loss_hist  device [iters] fp32 or NULL: total loss per iteration. */
int immoco_solver_solve(immoco_solver_t s, const float* kspace_in, const int32_t* col_group,
                        float* params_image,
                        float* params_motion, float* adam_image, float* adam_motion, int32_t iters,
                        float lr, const float* lambda_sched /*[host]*/, int32_t step0,
                        float* out_image, float* out_kspace, float* loss_hist, void* stream);

/* The same for a batch of B slices of the solver's shape (BASELINE config 3; SURVEY §8b "over a batch
 * dimension B of slices"): every buffer gains a leading dimension B - kspace_in [B,H,W] c64, col_group
 * [B,W], params_* [B,n_params], adam_* [B,2*n_params], out_* [B,H,W] c64, loss_hist [B,iters] or NULL;
 * lambda_sched / lr / step0 are shared.  cfg.batch_lanes slices are in flight at a time (slice i runs on lane
 * i % lanes); the call returns when everything is queued, outputs are ordered on `stream`. */
int immoco_solver_solve_batch(immoco_solver_t s, int32_t B, const float* kspace_in, const int32_t* col_group,
                              float* params_image, float* params_motion, float* adam_image, float* adam_motion,
                              int32_t iters, float lr, const float* lambda_sched, int32_t step0,
                              float* out_image, float* out_kspace, float* loss_hist, void* stream);

/* One forward pass only (IMMoCo.forward, immoco.py:82-113). */
int immoco_solver_forward(immoco_solver_t s, const int32_t* col_group, const float* params_image,
                          const float* params_motion, float* out_image, float* out_kspace,
                          void* stream);

/* Times every kernel of the iteration with HIP events on the solver's stream:
 * `reps` eager iterations (parameters / Adam state advance as in a real solve). */
int immoco_solver_profile(immoco_solver_t s, const float* kspace_in, const int32_t* col_group,
                          float* params_image,
                          float* params_motion, float* adam_image, float* adam_motion, int32_t reps,
                          float lr, float lambda_ge, void* stream);
/* Average per-kernel device time (ms) of the last immoco_solver_profile call, for
 * bench.py's roofline line.  names: [host] array of const char*; returns count. */
int immoco_solver_phase_times(immoco_solver_t s, const char** names, float* ms, int32_t max_n);
/* Duration (ms) of the dominant kernel (motion-grid encode backward) in the LAST iteration of the
 * last solve, from HIP events recorded around it on its stream (valid after an EAGER solve: HIP does
 * not report elapsed time for events recorded as graph nodes); < 0 if unavailable. */
float immoco_solver_dominant_kernel_ms(immoco_solver_t s);
/* Run-time switch between graph replay (1) and eager launches on the same streams (0). */
int immoco_solver_set_graph(immoco_solver_t s, int32_t use_graph);
/* 1 when the last solve replayed a captured hipGraph, 0 when it launched eagerly. */
int immoco_solver_graph_active(immoco_solver_t s);

/* Entries of a transposed hash-grid index (which: 0 image grid, 1 motion grid) = dL/denc gathers one
 * encode-backward launch issues (twin entries count once); 0 before immoco_solver_set_lattice. */
int64_t immoco_solver_plan_entries(immoco_solver_t s, int32_t which);

/* Measurement aid (bench.py `roofline.gather_ceiling`), no counterpart in the reference: the rate the
 * chip sustains for the request shape of the hash-grid kernels - n_lanes lanes, each issuing
 * loads_per_lane independent bytes_per_load-byte (8 or 16) loads at pseudo-random aligned offsets of a
 * footprint_bytes table (power of two), 4 in flight per lane.  Allocates and frees its own scratch.
 * ms_out: [host] average duration of one launch over `repeats` launches. */
int immoco_probe_gather(int64_t footprint_bytes, int32_t bytes_per_load, int64_t n_lanes,
                        int32_t loads_per_lane, int32_t repeats, void* stream, float* ms_out);

#ifdef __cplusplus
}
#endif
#endif /* IMMOCO_HIP_H */

/*
 * ddn.h -- C ABI of libddn, the MI355X (gfx950) implementation of the
 * deblur_denoise iteration hot path of rahul3/convex_optimization.
 *
 * The reference has no native/FFI boundary: its seam is the public Python API.
 * Each entry point below names the reference function(s) whose work it replaces
 * (paths relative to the reference root, deblur_denoise/...).  A Python host
 * (convex_optimization_amd/) binds these with ctypes and keeps the reference's
 * signatures.
 *
 * Conventions
 *   - every pointer named *_dev is a device pointer (HIP), caller-owned, never
 *     freed or retained past the call unless stated; images are row-major
 *     [batch][H][W] real arrays of the plan's dtype; 3-component duals are
 *     [3][batch][H][W] (component-major), exactly torch.stack((a,b,c)) for batch 1;
 *   - return value 0 = success, negative = error (see DDN_E_*); the message is
 *     available from ddn_last_error() (thread-local); nothing throws across the ABI;
 *   - a plan is bound to one device and one stream and is not thread-safe;
 *     all work is enqueued on that stream and is asynchronous unless stated.
 */
#ifndef DDN_H
#define DDN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDN_OK 0
#define DDN_E_ARG (-1)       /* bad shape / dtype / enum / null pointer          */
#define DDN_E_UNSUPPORTED (-2) /* size with a prime factor > 8191, too large for LDS */
#define DDN_E_HIP (-3)       /* a HIP runtime call failed                         */
#define DDN_E_STATE (-4)     /* plan not set up for the requested operation       */

/* dtype */
#define DDN_F32 0
#define DDN_F64 1
/* complex dtype of arrays handed in from the host framework */
#define DDN_C64 0
#define DDN_C128 1

/* objective of the data term (cp:69-74, admm:99-104, pdr:101-110) */
#define DDN_OBJ_L1 0
#define DDN_OBJ_L2 1

/* loss kinds computed on device every iteration (core/loss.py:5-36,112-117) */
#define DDN_LOSS_NONE 0
#define DDN_LOSS_L1 1
#define DDN_LOSS_L2 2
#define DDN_LOSS_MSE 3
#define DDN_LOSS_PSNR 4
#define DDN_LOSS_SSIM 5   /* core/loss.py:39-110; needs ddn_plan_set_ssim */

/* spectral operators for ddn_apply (op_math/python_code/multiplying_matrix.py:74-112) */
#define DDN_OP_K 0          /* apply_K           mm:74-76   */
#define DDN_OP_KT 1         /* apply_KTrans      mm:86-88   */
#define DDN_OP_D 2          /* apply_D1 == apply_D2  mm:78-84 (both are the horizontal backward difference) */
#define DDN_OP_DT 3         /* apply_D1Trans == apply_D2Trans mm:90-96 */
#define DDN_OP_RESOLVENT 4  /* invert_matrix     mm:110-112 */
#define DDN_OP_MAT 5        /* apply_Mat         mm:106-108 */

/* prox kinds (core/proximal_operators.py:4-57) */
#define DDN_PROX_L1 0
#define DDN_PROX_L2SQ 1
#define DDN_PROX_ISO 2
#define DDN_PROX_BOX 3
#define DDN_PROX_L2 4

typedef struct ddn_plan ddn_plan;

int ddn_version(void);
const char* ddn_last_error(void);

/* Replaces DeblurDenoiseOperators.__init__ (mm:10-46) for `batch` images of H x W.
 * Allocates twiddle tables, spectral scratch (2 half-spectra per image) and the
 * per-iteration loss buffers.  `stream` is a hipStream_t (NULL = default stream). */
int ddn_plan_create(ddn_plan** out, int H, int W, int batch, int dtype, void* stream);
int ddn_plan_destroy(ddn_plan* p);
int ddn_plan_set_stream(ddn_plan* p, void* stream);

/* Column-stage tiling knobs (0 = heuristic): number of columns per workgroup and the
 * split H = A x B of the column transform (A = 1: one pass).  For experiments. */
int ddn_plan_tune(ddn_plan* p, int cols_per_block, int split_A, int col_threads);

/* Replaces eig_vals_for_periodic_conv_op (eig_vals_for_periodic_conv_op.py:5-61) and
 * setup_operators (mm:48-72): builds, on the device, the eigenvalue array of the
 * periodic correlation with `taps` (host, row-major kh x kw, rounded to float32 as
 * eig:49 does), centred at (kh/2, kw/2), and of D = [[-1, 1]].  */
int ddn_plan_set_kernel(ddn_plan* p, const double* taps_host, int kh, int kw);

/* Injects eigenvalue arrays exactly as the reference holds them (device, H x W
 * complex, natural index order; mm:51-59).  eigD may be NULL (analytic 1 - w^k is
 * used).  Needed for fp64 parity, where the reference's complex64-rounded arrays
 * must be consumed bit for bit.  Selects row-spectral D/D^T in the solvers. */
int ddn_plan_set_eig(ddn_plan* p, const void* eigK_dev, const void* eigD_dev, int cdtype);

/* Resolvent multiplier (mm:67-72, 110-112):  1 / (1 + cK |eK|^2 + cD |eD|^2)  with
 * cK = t*t, cD = t*t + t*s; or, when M_dev != NULL, the reciprocal of the injected
 * `eig_vals_mat` (device, H x W complex, natural order). */
int ddn_plan_set_resolvent(ddn_plan* p, double cK, double cD, const void* M_dev, int cdtype);

/* Exports the eigenvalue array of K in the reference's layout (device, H x W complex,
 * natural order, complex64 for a float32 plan / complex128 for float64). */
int ddn_plan_get_eigK(ddn_plan* p, void* out_dev);

/* y = real(ifft2(e * fft2(x))) for a built-in operator (apply_periodic_conv_2d.py:28
 * via mm:74-112).  x, y: [batch][H][W]; may alias, except for DDN_OP_D / DDN_OP_DT without injected
 * eigen arrays (a two-tap stencil there, which rejects x == y with DDN_E_ARG). */
int ddn_apply(ddn_plan* p, int op, const void* x_dev, void* y_dev);

/* y = real(ifft2(mult * fft2(x))) with an arbitrary multiplier (device, H x W complex,
 * natural order) -- apply_periodic_conv_2d.py:3-28 with a caller-supplied eigen array;
 * flags bit0: use conj(mult); bit1: use 1/mult. */
int ddn_apply_multiplier(ddn_plan* p, const void* mult_dev, int cdtype, int flags, const void* x_dev, void* y_dev);

/* circular_convolve2d (core/convolution.py:22-52): FFT circular convolution with the
 * float32 zero-padded kernel rolled by (-kh//2, -kw//2).  image/out: [batch][H][W]. */
int ddn_circular_convolve2d(ddn_plan* p, const double* taps_host, int kh, int kw, const void* image_dev, void* out_dev);

/* Elementwise proximal maps (core/proximal_operators.py:4-57) on n elements.
 * ISO: x and y are [2][n] (prox:22-50).  L2 (prox:10-14) uses the whole-tensor norm. */
int ddn_prox(int kind, int dtype, const void* x_dev, void* y_dev, size_t n, double lambda, void* stream);

/* l1_loss / l2_loss / mse / psnr (core/loss.py:5-36,112-117) of x - y per image.
 * out4_dev: [batch][4] doubles = {sum|d|, sqrt(sum d^2), mean d^2, 10 log10(1/mse)}. */
int ddn_reduce(int dtype, const void* x_dev, const void* y_dev, size_t n_per_image, int batch, double* out4_dev, void* stream);

/* Allocation-free variant of ddn_reduce.  It has no counterpart in the reference: it exists for callers
 * that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  Same arguments, same kernels and launch shape, bit-identical out4_dev; the per-chunk
 * partial sums live in a static device array of the library (512 KiB per GPU, held until the process
 * ends) instead of a temporary buffer, so the call makes no hipMalloc / hipFree and, unlike ddn_reduce,
 * does not synchronise: out4_dev is valid once `stream` has reached this point.
 * Limit: batch <= DDN_REDUCE_STATIC_MAX_BATCH, else DDN_E_UNSUPPORTED.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (different streams or threads) race; serialise them.
 * The array's address is resolved once per device, at the first ddn_plan_create or the first call of
 * this function on that device (do that first call outside graph capture). */
#define DDN_REDUCE_STATIC_MAX_BATCH 128
int ddn_reduce_static(int dtype, const void* x_dev, const void* y_dev, size_t n_per_image, int batch, double* out4_dev, void* stream);

/* ssim (core/loss.py:39-110): 11 x 11 Gaussian window (sigma 1.5), zero padding, C1 = (0.01 max_val)^2,
 * C2 = (0.03 max_val)^2, mean of the SSIM map per plane.  x, y: [planes][H][W]; out_dev: [planes] doubles.
 * g11_host: the 11 normalised float32 window factors exactly as loss.py:75-81 computes them (the caller
 * evaluates that expression with the framework the reference uses, so the taps agree bit for bit).
 * Synchronises the stream.  (A [B,C,H,W] input is B*C planes; the caller averages over C.) */
int ddn_ssim(int dtype, const void* x_dev, const void* y_dev, int H, int W, int planes, double max_val, const float* g11_host, double* out_dev, void* stream);

/* The same with the reference's `window_size` argument (core/loss.py:39): g_host holds the window_size normalised
 * float32 factors of loss.py:75-81 evaluated for that size (window_size <= 63).  The SSIM map has
 * H + 2 (window_size / 2) - window_size + 1 rows and likewise columns -- one more than the image for an even
 * window, as the reference's conv2d calls with padding window_size // 2 produce -- and its mean is taken over
 * that map.  Direct evaluation of the window_size^2 taps; synchronises the stream. */
int ddn_ssim_window(int dtype, const void* x_dev, const void* y_dev, int H, int W, int planes, double max_val, const float* g_host, int window_size,
                    double* out_dev, void* stream);

/* X^T w of the blind-kernel step (op_math/python_code/apply_x_trans.py:4-46): w, x, out are
 * kernel_size x kernel_size arrays; out[i][j] = <w, x correlated with a one-hot kernel at (i, j) under
 * zero padding kernel_size / 2>.  kernel_size must be odd: for even sizes the reference's padded
 * correlation has kernel_size + 1 rows and its product with w raises. */
int ddn_apply_x_trans(int dtype, const void* w_dev, const void* x_dev, int kernel_size, void* out_dev, void* stream);

/* ssim (core/loss.py:39-110) as the loss evaluated inside the solver loops: the default loss_function of
 * primal_dr_splitting (primal_dr/algorithm.py:25) and of deblur_image / blur_and_deblur_image (blur_deblur.py:87,153).
 * g11_host: the 11 window factors as for ddn_ssim; ssim_ring_dev: device doubles [n_iters][batch][2], caller-owned, which
 * the following *_run calls fill (the caller pre-fills it, e.g. with NaN): entry 0 of iteration i's slot = ssim of the
 * solver's primary pair after iteration i (cp: (x, b); admm: (sol, b); pdr: (z1, z1prev); pddr: (clip(p), b)),
 * entry 1 (pdr only) = ssim(z1, b) at iterations 1, 1 + n, 1 + 2 n, ... with n = 50, the reference's logging
 * cadence (pdr:137-141), unless ddn_plan_set_ssim_cadence says otherwise.
 * Every *_run records whenever a ring is set, whatever tol_kind is -- except pdr's entry 0, whose pair (z1, z1prev)
 * only exists as the stop criterion and is recorded with tol_kind = DDN_LOSS_SSIM only.  With tol_kind =
 * DDN_LOSS_SSIM the stop test `loss < tol` is made on entry 0, on the device, as for the other kinds.
 * ssim_ring_dev = NULL switches it off.
 * Cost per iteration: one pass over the two images (plus, for pdr with tol_kind = DDN_LOSS_SSIM, one copy of z1). */
int ddn_plan_set_ssim(ddn_plan* p, const float* g11_host, double max_val, double* ssim_ring_dev);

/* primal DR: ssim(z1, b) (entry 1 of the ssim record) every `every` iterations from the first; 1 = at every
 * iteration, which is what the reference records when it saves the loss history (pdr:146-154).  Default 50. */
int ddn_plan_set_ssim_cadence(ddn_plan* p, int every);

/* ---- solver loops -------------------------------------------------------------
 * Each *_run advances the iterate arrays in place by up to n_iters iterations of the
 * reference loop body.  Per iteration the device records, per image, the four loss
 * values of (iterate - reference image) in loss_ring_dev[iter][image][8] (doubles;
 * entries 0..3 = l1,l2,mse,psnr of the solver's primary pair, 4..7 = secondary pair,
 * see each solver) when loss_ring_dev != NULL.  When tol_kind != DDN_LOSS_NONE an
 * image whose loss value (of that kind, primary pair) drops below `tol` after an
 * iteration is frozen at that iteration (the reference's `break`), without a host
 * round trip; iters_done_dev[image] (int32, may be NULL) receives the number of
 * iterations applied.
 */

/* chambolle_pock loop body cp:58-92.  x,z: [batch][H][W]; y: [3][batch][H][W]; b: [batch][H][W].
 * Loss pair: (x_next, b) (cp:96). No early stop in the reference (tol_kind ignored). */
int ddn_cp_run(ddn_plan* p, void* x_dev, void* z_dev, void* y_dev, const void* b_dev,
               double t, double s, double gamma, int objective, int n_iters,
               double* loss_ring_dev);

/* admm_solver loop body admm:69-115 and final solve admm:160-173.
 * u,w: [batch][H][W]; y,z: [3][batch][H][W]; sol_out: [batch][H][W] receives the
 * solution the reference returns.  Loss pair: (sol, b) where sol is the look-ahead
 * solve of admm:118-132; iteration i's value is recorded at ring slot i.
 * first_loss_iter: iterations < first_loss_iter are not tested against tol. */
int ddn_admm_run(ddn_plan* p, void* u_dev, void* w_dev, void* y_dev, void* z_dev, const void* b_dev,
                 double t, double rho, double gamma, int objective, int n_iters,
                 int tol_kind, double tol, int first_loss_iter,
                 double* loss_ring_dev, int* iters_done_dev, void* sol_out_dev);

/* primal_dr_splitting loop body pdr:94-129.  z1: [batch][H][W]; z2: [3][batch][H][W].
 * Primary loss pair: (z1, z1prev) (pdr:144); secondary: (z1, b) (pdr:145-149).
 * objective: 0 = l1, 1 = l2, optionally | DDN_OBJ_DUALS_EQUAL: the caller's statement that z2[1] and z2[2] hold the
 * same bits on entry (the reference starts both from b, pdr:88-89, and D1 == D2, multiplying_matrix.py:55-59, keeps
 * them equal): the run then streams one of the two arrays and copies it to the other at the end.  Nothing is checked:
 * with the flag set and unequal arrays, z2[2] is overwritten by z2[1]'s trajectory. */
#define DDN_OBJ_DUALS_EQUAL 0x100
int ddn_pdr_run(ddn_plan* p, void* z1_dev, void* z2_dev, const void* b_dev,
                double t, double rho, double gamma, int objective, int n_iters,
                int tol_kind, double tol, int first_loss_iter,
                double* loss_ring_dev, int* iters_done_dev);

/* primal_dual_dr_splitting loop body pddr:65-101.  p: [batch][H][W]; q: [3][batch][H][W].
 * Loss pair: (clip(p), b) (pddr:104). */
int ddn_pddr_run(ddn_plan* p, void* p_dev, void* q_dev, const void* b_dev,
                 double t, double rho, double gamma, int n_iters,
                 int tol_kind, double tol, int first_loss_iter,
                 double* loss_ring_dev, int* iters_done_dev);

/* ---- one image split into row slabs over several GPUs ------------------------------
 * The reference solves a single 2-D image per call (multiplying_matrix.py:31) with whole-image
 * fft2 / ifft2 (apply_periodic_conv_2d.py:3-28); these entry points are the two local halves of one
 * iteration of admm:69-115 / pdr:94-129 / pddr:65-101 when the image's rows are spread over ranks.
 * Every rank creates its plan with the FULL image size (batch 1) and sets the blur kernel and the
 * resolvent as usual, so that its multipliers cover every frequency.  Rank r owns `rows`
 * consecutive image rows (the same count on every rank) and a contiguous range of spectral tiles
 * (ddn_slab_layout: tile_cols = C columns per tile, n_tiles = ceil((W/2+1)/C) tiles in all).
 *
 *   ddn_slab_rows   the fused row kernel of `solver` (0 admm, 1 pdr, 2 pddr) on the local rows.
 *                   Iterate arrays are the local slabs: s0..s3 = (u, w, y, z) for admm, (z1, z2)
 *                   for pdr, (p, q) for pddr, shapes [rows][W] and [3][rows][W]; b: [rows][W].
 *                   consume / produce: whether the launch takes the solved pair in from
 *                   spec_rows / leaves the next right-hand side in it (first launch of a loop:
 *                   0 / 1, steady state 1 / 1, final ADMM solve 1 / 0 with sol_only = 1).
 *                   spec_rows: complex [n_tiles][2][rows][C] — of every tile first the rows of the
 *                   first half-spectrum, then of the second; the tiles of one destination rank
 *                   are one contiguous block.
 *   ddn_slab_cols   the column stage (forward transform, resolvent algebra, inverse transform) in
 *                   place on spec_cols, the tiles [tile_lo, tile_lo + n_tiles_local) of both half-spectra
 *                   with all H rows.  n_row_blocks = P >= 1: spec_cols is complex
 *                   [P][n_tiles_local][2][H / P][C] — block i is what rank i's spec_rows holds for these
 *                   tiles, i.e. exactly what an all-to-all of the spec_rows buffers delivers, so no
 *                   repacking is needed (with P = 1 spec_cols may be spec_rows itself).  H / P must be a
 *                   power of two when P > 1 (else DDN_E_UNSUPPORTED).  n_row_blocks = 0: spec_cols is
 *                   complex [2][n_tiles_local][H][C] (the caller has put the rows in order).
 *   ddn_slab_loss   record8[0..7] = l1, l2, mse, psnr (primary pair), then the secondary pair, of
 *                   the LOCAL rows seen by the last consuming ddn_slab_rows launch; l1 and l2^2
 *                   add up over ranks (mse and psnr are per-slab figures over the full pixel count).
 * The exchange between the two buffers is an all-to-all owned by the caller (RCCL through
 * torch.distributed in this package).  Early stop: ddn_slab_begin / ddn_slab_stop_test below.  Injected eigen arrays
 * (ddn_plan_set_eig, ddn_plan_set_resolvent with M_dev: the fp64 parity route) are honoured. */
int ddn_slab_layout(ddn_plan* p, int* tile_cols, int* n_tiles);
/* A plan for the slab entries that holds the multipliers of the rank's own spectral tiles only (the reference keeps
 * eig_vals_mat of the whole image on its one device, mm:67-72; a rank of the decomposition needs a 1/P share).
 *   ddn_plan_create_slab   as ddn_plan_create with batch 1, but without half-spectra and whole-plane multipliers.
 *   ddn_slab_restrict      (after ddn_slab_layout, before ddn_plan_set_kernel) allocates eK and the resolvent
 *                          multiplier for tiles [tile_lo, tile_lo + n_tiles_local): H x C x n_tiles_local entries each.
 * ddn_plan_set_kernel then evaluates the eigenvalues of those tiles straight from the taps (a double-precision sum over
 * the taps per entry, rounded once: within one unit of the last place or two of the transform-built ones);
 * ddn_plan_set_resolvent takes no injected array (DDN_E_UNSUPPORTED).  ddn_slab_cols / ddn_slab_cols_multiply accept
 * tile ranges inside the restricted range only; every whole-plane entry (ddn_*_run, ddn_apply, ...) returns
 * DDN_E_STATE on such a plan.  The column stage runs on the tile-generic kernels. */
int ddn_plan_create_slab(ddn_plan** out, int H, int W, int dtype, void* stream);
int ddn_slab_restrict(ddn_plan* p, int tile_lo, int n_tiles_local);
/* Early stop of a slab solve (the reference's `break`: admm:152-154, pdr:159-161, pddr:119-121) without a host round
 * trip.  The stop flag lives in the plan; every ddn_slab_rows / ddn_slab_rows_cp launch skips its rows once it is set,
 * so the iterates stay as they were when the test fired (ADMM: pass sol_out_dev in every consuming launch, as
 * ddn_admm_run does, so that the look-ahead solution of the stopping iteration is what remains in it).
 *   ddn_slab_begin       clears the flag: call once at the start of a solve.
 *   ddn_slab_loss_sums   sums4_dev[0..3] = sum |d|, sum d^2 of the primary pair, then of the secondary pair, over the
 *                        LOCAL rows of the last consuming launch: the caller all-reduces (SUM) these four doubles.
 *   ddn_slab_stop_test   from the all-reduced sums: record8_dev (may be NULL) = l1, l2, mse, psnr of both pairs for the
 *                        whole image, and the flag <- iter when iter >= first_iter and the value of `kind`
 *                        (DDN_LOSS_L1..PSNR, primary pair) is below tol.  Every rank makes the same decision from the
 *                        same sums.  kind = DDN_LOSS_NONE only records.
 *   ddn_slab_iters_done  done_dev[0] (int32, device) = -1 while running, else the iteration at which the solve froze;
 *                        lets the host leave its launch loop (poll it every few iterations, not every one). */
int ddn_slab_begin(ddn_plan* p);
int ddn_slab_loss_sums(ddn_plan* p, int rows, double* sums4_dev);
int ddn_slab_stop_test(ddn_plan* p, const double* sums4_dev, int kind, double tol, int iter, int first_iter, double* record8_dev);
int ddn_slab_iters_done(ddn_plan* p, int* done_dev);
int ddn_slab_rows(ddn_plan* p, int solver, int rows, void* spec_rows_dev, void* s0_dev, void* s1_dev, void* s2_dev, void* s3_dev,
                  const void* b_dev, double t, double rho, double gamma, int objective, int consume, int produce,
                  void* sol_out_dev, int sol_only);
int ddn_slab_cols(ddn_plan* p, void* spec_cols_dev, int tile_lo, int n_tiles_local, int n_row_blocks);
int ddn_slab_loss(ddn_plan* p, int rows, double* record8_dev);
/* The same for chambolle_pock (cp:58-92), whose iteration has two column stages and one half-spectrum:
 * spec_rows: complex [n_tiles][rows][C]; spec_cols: complex [P][n_tiles_local][H / P][C] (n_row_blocks = P)
 * or [n_tiles_local][H][C] (n_row_blocks = 0).
 *   ddn_slab_rows_cp        step = 1: the primal step (consumes K^T y^ ... produces the right-hand side of the
 *                           dual step), step = 0: the dual step; x, z: [rows][W], y: [3][rows][W].
 *                           A loop is: (step 1, consume 0, produce 1), then per iteration
 *                           cols_multiply(conj 0), (step 0, 1, 1), cols_multiply(conj 1), (step 1, 1, produce
 *                           unless last); ddn_slab_loss after a consuming step-1 launch gives (x_next, b).
 *                           spec_aux_dev: a second buffer shaped like spec_rows, needed only with injected eigen
 *                           arrays (ddn_plan_set_eig), where D^T y passes from the dual-step launch to the
 *                           primal-step launch as a spectrum that never leaves the rank; else NULL.
 *   ddn_slab_cols_multiply  spectrum <- eK (conj = 0) or conj(eK) (conj = 1) times spectrum / (H W), between a
 *                           forward and an inverse column transform, on the local tile range. */
int ddn_slab_rows_cp(ddn_plan* p, int rows, void* spec_rows_dev, void* spec_aux_dev, void* x_dev, void* z_dev, void* y_dev, const void* b_dev,
                     double t, double s, double gamma, int objective, int step, int consume, int produce);
int ddn_slab_cols_multiply(ddn_plan* p, void* spec_cols_dev, int tile_lo, int n_tiles_local, int conj, int n_row_blocks);
/* ddn_slab_cols (mode 0) or ddn_slab_cols_multiply (mode 1: times eK, 2: times conj eK) on `count` of the buffer's
 * tiles from the first-th on: lets the caller run the column stage of a tile group while the exchange of the next one
 * is in flight.  No counterpart in the reference (its transform is one fft2 call, mm:110-112). */
int ddn_slab_cols_part(ddn_plan* p, void* spec_cols, int tile_lo, int n_tiles_local, int n_row_blocks, int mode, int first, int count);

/* Timing hook for bench.py: average duration in milliseconds, measured with HIP
 * events on the plan's stream, of the last *_run call's dominant kernel (the fused
 * row kernel) and of all its kernels; synchronises the stream.  Row launches that do
 * only part of the steady-state work (the produce-only first and the consume-only last
 * launch of a loop) are not sampled: the row average is that of full launches. */
int ddn_plan_last_timing(ddn_plan* p, double* row_kernel_ms_avg, double* col_kernel_ms_avg, int* n_row_launches, int* n_col_launches);
/* on = 0: off; 1: every launch; n > 1: every n-th row launch and every n-th column launch (events cost a few
 * microseconds each, which matters when an iteration is a few hundred). */
int ddn_plan_enable_timing(ddn_plan* p, int on);

/* Diagnostic (no counterpart in the reference): when the environment variable DDN_DEBUG_STAMPS=1 was
 * set at plan creation, copies the clock stamps the fused row kernel took at its phase boundaries in
 * its last launch (16 int64 per workgroup, first n_wg workgroups) to host memory; workgroups 4096.. of
 * the buffer (n_wg up to 8192) are the stamps of the last single-pass column kernel. */
int ddn_plan_debug_stamps(ddn_plan* p, long long* out_host, int n_wg);

#ifdef __cplusplus
}
#endif
#endif /* DDN_H */

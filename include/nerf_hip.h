/* nerf_hip.h -- C ABI of the MI355X-native NeRF hot path (libnerf_hip.so, gfx950).
 *
 * This is the drop-in boundary of SURVEY.md 8(b).  The reference
 * (piljoong-jeong/nerf_meets_mlx) has no FFI layer of its own: its hot path is a set of
 * Python functions over mlx arrays.  Each entry point below replaces the device work of
 * one (or a fused group) of those functions; the citation after "replaces:" is the
 * reference file:line.  The Python host package `nerf_meets_mlx_amd` binds these with
 * ctypes and re-exports the reference's function names/signatures.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) owned by the caller unless marked "host";
 *     the library allocates nothing.  What a call computes depends only on its arguments: the arithmetic of a
 *     network (bf16 or fp32 MFMA operands) is a field of its `nerf_mlp_arch`, so models of different precision
 *     can be used side by side, on any streams.  Process-wide state is limited to the last error string and the
 *     A/B measurement knobs of nerf_set_option (kernel-variant selection; they never change results' layout,
 *     buffer sizes or which weight image a launch reads).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no hidden
 *     synchronisation, safe to capture into a hipGraph.
 *   - tensors are dense row-major float32 unless stated; index tensors are int64.
 *   - return value: 0 = NERF_OK, negative = NERF_E_*;  nerf_last_error() gives text.
 *   - thread-compatible (not thread-safe); one process per GPU.
 */
#ifndef NERF_HIP_H
#define NERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 2): nerf_mlp_packed_bytes / nerf_mlp_dz_bytes return larger sizes (fp32 weight streams behind the bf16 image;
 * split-K partial tiles behind the dZ blocks) and nerf_mlp_acts_bytes / _dz_bytes depend on "mlp_precision": callers that
 * always size their buffers with these functions are unaffected; new entry points were only added.                  */
/* 3 (round 3): `precision` became a field of nerf_mlp_arch (was the process-global option "mlp_precision"):
 * nerf_mlp_packed_bytes / _acts_bytes / _dz_bytes / _pack and every forward / backward read it from the arch they are
 * given.  The struct grew by one int at the end; a v2 caller must be recompiled.  (Later in round 3, no ABI change: for
 * precision 32 nerf_mlp_acts_bytes counts 64 more rows per 32-sample tile -- the ReLU sign bits the backward chain reads
 * instead of the float32 rows -- and nerf_mlp_dz_bytes 34 MB of split-K partial blocks, so that the fp32 weight gradient
 * is a fixed-order sum like the bf16 one: bit-reproducible, no atomics.)                                           */
#define NERF_ABI_VERSION 3

#define NERF_OK 0
#define NERF_E_NULL (-1)        /* required pointer is NULL                        */
#define NERF_E_SHAPE (-2)       /* size / shape out of the supported range         */
#define NERF_E_UNSUPPORTED (-3) /* architecture / mode not implemented in HIP      */
#define NERF_E_HIP (-4)         /* HIP runtime error (launch failed, bad stream)   */
#define NERF_E_RCCL (-5)        /* RCCL error                                      */

int nerf_abi_version(void);
const char* nerf_last_error(void);

/* ---------------------------------------------------------------- rays (a1, a3, a4)
 * rays are packed exactly like the reference's `rays_linear`
 * (entrypoints/__test_nerf.py:60-82, rendering/render.py:319-328):
 *   [o(3), d(3), near, far, viewdirs(3)] = 11 floats per ray.                        */
#define NERF_RAY_STRIDE 11

/* Distinct pseudo-random indices in [0, domain): out[i] = P_seed(offset + i) where P is a
 * keyed bijection of [0, domain) (4-round Feistel + cycle walking).  Replaces
 * `np.random.choice(H*W, N_rand, replace=False)` entrypoints/__test_nerf.py:229 with O(n)
 * device work; requires offset + n <= domain.  Bit-exact vs the host mirror
 * nerf_meets_mlx_amd.ops.index.pixel_permutation.                                  */
int nerf_pixel_permutation(int64_t* out_idx, int64_t n, int64_t domain, uint64_t seed, uint64_t offset,
                           void* stream);

/* replaces: rendering/ray.py:7-35 get_rays + the (row, col) gather of
 * entrypoints/__test_nerf.py:213-236 + viewdirs/near/far packing (:60-82).
 * pixel_idx: [n] flat row-major pixel indices, or NULL for all H*W pixels in order
 * (then n must equal H*W).  K: host, 9 DOUBLES row-major (the reference's K is a float64 numpy array,
 * entrypoints/__test_nerf.py:170-174); c2w: host, 12 floats (3x4, row-major).  Directions are
 * evaluated in float64 and rounded once to float32.
 * coords (optional, may be NULL): [n,2] int64 (row, col) = (idx / W, idx % W).        */
int nerf_ray_gen(const int64_t* pixel_idx, int64_t n, int H, int W, const double* K_host, const float* c2w_host,
                 float near, float far, float* rays, int64_t* coords, void* stream);

/* replaces, for the training step, the three calls around it as ONE launch: entrypoints/__test_nerf.py:213-236 (N_rand
 * distinct pixels of one image, their rays, their target colours) + :60-82 (ray packing).  Pixel i of the batch is
 * P(seed; offset + i) of nerf_pixel_permutation over H*W; rays [n,11] as nerf_ray_gen; target [n,3] = image[pixel];
 * pixel_idx (optional) [n] int64.  image: [H*W,3] float32 device.  Outputs bit-identical to the three separate calls. */
int nerf_sample_batch(int64_t n, int H, int W, uint64_t seed, uint64_t offset, const double* K_host, const float* c2w_host,
                      float near, float far, const float* image, float* rays, float* target, int64_t* pixel_idx,
                      void* stream);

/* out[i, :] = src[idx[i], :]   (target pixel gather, entrypoints/__test_nerf.py:236).
 * idx[i] outside [0, n_src) is never dereferenced: that output row is NaN.               */
int nerf_gather_rows(const float* src, int64_t n_src, const int64_t* idx, int64_t n, int channels, float* out,
                     void* stream);

/* replaces: rendering/ray.py:39-70 ndc_rays (in place on the o,d columns of `rays`)    */
int nerf_ndc_rays(float* rays, int64_t n, int H, int W, float focal, float near, void* stream);

/* ---------------------------------------------------------------- sampling (a5-a7, a15, a17)
 * replaces: sampling/uniform.py:7-18, sampling/linear_disparity.py:8-19 (literal),
 * sampling/__init__.py:10-31 add_noise_z (intended semantics).
 * t_rand: [B,n] uniforms in [0,1) (caller's RNG), may be NULL when perturb <= 0.       */
int nerf_sample_coarse(const float* rays, int64_t B, int n, int lindisp, float perturb, const float* t_rand,
                       float* z, void* stream);

/* replaces: sampling/__init__.py:10-31 add_noise_z on caller-supplied depths (the fused form above covers the render
 * path): mids = (z[k] + z[k+1]) / 2, lower = [z_first, mids], upper = [mids, z_last],
 * z_out = lower + (upper - lower) * (t_rand * strength).  z_in, t_rand, z_out: [B,n]; not in place.                */
int nerf_add_noise_z(const float* z_in, const float* t_rand, int64_t B, int n, float strength, float* z_out,
                     void* stream);

/* replaces: sampling/__init__.py:101-177 sample_from_inverse_cdf_torch (u passed in
 * instead of torch.rand) and the sort of entrypoints/__test_nerf.py:288 /
 * rendering/render.py:225.  weights: [B,n] (the reference's [B,n,1] squeezed).
 * Outputs (each may be NULL): z_new [B,N]; z_merged [B,n+N] ascending; cdf [B,n+1];
 * inds [B,N] int64 = searchsorted(cdf, u, side="right") -- bit-exact given (cdf,u).
 * Limits: 2 <= n <= 256, 1 <= N <= 512, n+N <= 768.                                   */
int nerf_importance_sample(const float* z, const float* weights, const float* u, int64_t B, int n, int N, float eps,
                           float* z_new, float* z_merged, float* cdf, int64_t* inds, void* stream);

/* ---------------------------------------------------------------- encodings (a9, a10, a22, a23)
 * replaces: models/embedding.py:23-90 Embedder.embed: [x, sin(f0 x), cos(f0 x), ...];
 * freq_mode 0 = k^2 (reference quirk Q4), 1 = 2^k.  out: [M, D*(1+2*n_freqs)].        */
int nerf_encode_freq(const float* x, int64_t M, int D, int n_freqs, int freq_mode, float* out, void* stream);

/* replaces: encoding/sinusoidal.py:39-66: sin([s, s+pi/2]), s = x[...,None]*freq flattened
 * dim-major/freq-minor; optional raw input appended at the end.  freqs: host float[n_freqs]
 * (= 2^linspace(min_exp,max_exp,n), evaluated by the caller so that host and oracle
 * share the table bit for bit; n_freqs <= 32).  out [M, 2*D*n_freqs (+D)].              */
int nerf_encode_sinusoidal(const float* x, int64_t M, int D, int n_freqs, const float* freqs_host,
                           int include_input, float* out, void* stream);

/* replaces: encoding/spherical_harmonics.py:33-94; out [M,(degree+1)^2], 0<=degree<=4  */
int nerf_sh_encode(const float* dirs, int64_t M, int degree, float* out, void* stream);

/* replaces: encoding/multi_hash.py:61-136 (intended semantics).
 * tables [L,T,F] float32, T = 2^log2_T, F in {1,2,4,8}; resolutions: host int[L];
 * out [M, L*F].  backward: d_tables += scatter of d_out (float atomics).              */
int nerf_hashgrid_forward(const float* x, int64_t M, const float* tables, int L, int log2_T, int F,
                          const int* resolutions_host, float* out, void* stream);
int nerf_hashgrid_backward(const float* x, int64_t M, const float* d_out, int L, int log2_T, int F,
                           const int* resolutions_host, float* d_tables, void* stream);

/* One input row per sample for the hash-grid model (BASELINE configs[4]; engine glue the reference never wrote):
 * x_out [B n, L F + (sh_degree+1)^2] = [ hash features of o + z d | SH of the ray's view direction ], i.e.
 * MultiHashEncoding(pts) and SphericalHarmonicsEncoding(viewdirs) of encoding/{multi_hash,spherical_harmonics}.py
 * written side by side; pts_out [B n, 3] (or NULL) keeps the positions for nerf_hashgrid_backward.  The grid sees
 * (o + z d) * pos_scale + pos_offset: the affine map of the scene box onto [0,1]^3 (1, 0 = world coordinates).    */
int nerf_ngp_encode(const float* rays, const float* z, int64_t B, int n, const float* tables, int L, int log2_T,
                    int F, const int* resolutions_host, int sh_degree, float pos_scale, float pos_offset, float* x_out,
                    float* pts_out, void* stream);
/* table gradient with the sample positions taken from rays / depths (o + z d) instead of a point list          */
int nerf_hashgrid_backward_rays(const float* rays, const float* z, int64_t B, int n, const float* d_out, int L,
                                int log2_T, int F, const int* resolutions_host, float pos_scale, float pos_offset,
                                float* d_tables, void* stream);

/* The same for levels [level_lo, level_hi) only (the launches of disjoint level groups can be followed one by one by the
 * all-reduce of their slice of d_tables on another stream), and optionally DETERMINISTIC: fixed_point = 1 makes d_tables
 * an int64 [L,T,F] array of 2^-52 fixed-point accumulators added with integer atomics (associative: the result does not
 * depend on the order the memory side serves the requests; float atomics do).  nerf_adam_step_ex consumes either form.
 * Range of the fixed-point form: an addend with |v| > 256 saturates to +-1.5 x 2^60 units (+-384: outside the window below with
 * 128 to spare, so the ordinary addends of the same entry cannot bring the sum back inside), a NaN / Inf addend adds 2^61
 * units, and nerf_adam_step_ex reads every accumulator outside (-2^60, 2^60) units -- saturated, poisoned, or a per-entry sum
 * beyond +-256, also after a cross-rank sum -- as a NaN gradient: a diverged run surfaces as NaN parameters, as with float
 * atomics.  (Two saturated addends of opposite sign on one entry cancel, like two float gradients of +-1e9 would.) */
int nerf_hashgrid_backward_ex(const float* x, int64_t M, const float* d_out, int L, int log2_T, int F,
                              const int* resolutions_host, int level_lo, int level_hi, int fixed_point, void* d_tables,
                              void* stream);
int nerf_hashgrid_backward_rays_ex(const float* rays, const float* z, int64_t B, int n, const float* d_out, int L,
                                   int log2_T, int F, const int* resolutions_host, float pos_scale, float pos_offset,
                                   int level_lo, int level_hi, int fixed_point, void* d_tables, void* stream);

/* ---------------------------------------------------------------- compositing (a13)
 * replaces: rendering/render.py:20-96 raw2outputs.  raw [B,n,4] = [rgb, sigma];
 * noise [B,n] (N(0,1), caller's RNG) may be NULL when raw_noise_std == 0.
 * Outputs: rgb [B,3], disp [B], acc [B], weights [B,n], depth [B] (any may be NULL
 * except rgb).  One wavefront per ray, n <= 1024.                                      */
int nerf_composite_forward(const float* raw, const float* z, const float* rays, int64_t B, int n,
                           float raw_noise_std, const float* noise, int white_bkgd, float* rgb, float* disp,
                           float* acc, float* weights, float* depth, void* stream);

/* adjoint of the above (what mlx autograd computes for entrypoints/__test_nerf.py:132,142).
 * d_rgb [B,3] required; d_acc [B], d_depth [B] optional (NULL = 0).  d_raw [B,n,4].     */
int nerf_composite_backward(const float* raw, const float* z, const float* rays, int64_t B, int n,
                            float raw_noise_std, const float* noise, int white_bkgd, const float* d_rgb,
                            const float* d_acc, const float* d_depth, float* d_raw, void* stream);

/* replaces: ops/metric.py:12-14 MSE and its gradient: loss_out[0] += sum((pred-target)^2)
 * / count (caller zeroes it), d_pred = grad_scale * 2 (pred-target) / count.           */
int nerf_mse_loss_grad(const float* pred, const float* target, int64_t count, float grad_scale, float* loss_out,
                       float* d_pred, void* stream);

/* replaces, for the training step, the three calls above as ONE pass per ray: what nn.value_and_grad differentiates
 * in entrypoints/__test_nerf.py:47-126 -- raw2outputs (raw_noise_std = 0 there: SURVEY Q5), loss = mean((rgb - target)^2)
 * over B x 3, d_raw = d loss / d raw x grad_scale.  loss_out[0] += loss (caller zeroes it); rgb (optional): [B,3].
 * rgb and d_raw are bit-identical to nerf_composite_forward + nerf_mse_loss_grad + nerf_composite_backward.        */
int nerf_composite_mse_backward(const float* raw, const float* z, const float* rays, int64_t B, int n, int white_bkgd,
                                const float* target, float grad_scale, float* loss_out, float* rgb, float* d_raw,
                                void* stream);

/* replaces: ops/metric.py:20-64 SSIM (unfinished upstream: the body stops after the five windowed moments; this is
 * the formula those moments feed).  pred, gt: [N,C,H,W] float32 device; window_host: the 1-D window (w_size <= 33
 * taps, the 2-D window of create_window :49-55 is its outer product), applied as a depthwise VALID convolution
 * (padding = NO_PAD, :33-42).  sums: [N,2] float64 device, overwritten with
 *   sums[n][0] = sum over (c, y, x) of ((2 mu_p mu_g + c1)(2 s_pg + c2)) / ((mu_p^2 + mu_g^2 + c1)(s_p^2 + s_g^2 + c2))
 *   sums[n][1] = sum of the contrast-structure term (2 s_pg + c2) / (s_p^2 + s_g^2 + c2)
 * over the C (H-w+1)(W-w+1) window positions of image n; the caller divides by that count.
 * c1 = (0.01 L)^2, c2 = (0.03 L)^2 with the dynamic range L of :24-28 are computed by the caller.                 */
int nerf_ssim_sums(const float* pred, const float* gt, int N, int C, int H, int W, const float* window_host,
                   int w_size, float c1, float c2, double* sums, void* stream);

/* ---------------------------------------------------------------- the MLP (a11, a12)
 * replaces: models/NeRF.py:160-243 (NeRF.__init__/forward), :10-48 (run_model),
 * models/embedding.py:4-21 (embed) for the architecture n_layers=8, width=256,
 * skips=[4], use_viewdirs, in_pos=63, in_dir=27, and for the no-view-direction model of the reference's image
 * fitting (entrypoints/__viser_image_learning.py:203-208: in_pos=40, in_dir=0, use_viewdirs=0, out_ch<=4; layers
 * pos0 [256x40] pos1..4 pos5 [256x296] pos6 pos7 output [out_ch x 256]).  Anything else: NERF_E_UNSUPPORTED.
 *
 * Parameter layout (float32, flat, `nerf_mlp_param_count` = 595844 elements), each
 * layer as weight[out][in] row-major followed by bias[out]  (nn.Linear: x @ W^T + b):
 *   pos0 [256x63] pos1..pos4 [256x256] pos5 [256x319] pos6 pos7 [256x256]
 *   feature [256x256]  alpha [1x256]  dir0 [128x283]  rgb [3x128]
 * Gradients use the same layout.  Three instances of the class have kernels: the view
 * model above, the image model {8, 256, 40, 0, 4, 0, out_ch <= 4}
 * (entrypoints/__viser_image_learning.py:198-208) and the Instant-NGP-sized view model
 * {2, 64, 32, 16, -1, 1}: pos0 [64x32] pos1 [64x64] feature [64x64] alpha [1x64]
 * dir0 [32x80] rgb [3x32] = 13 188 parameters (forward / forward_train / backward on
 * embedded rows; no fused positional-encoding query).                                  */
typedef struct nerf_mlp_arch {
  int n_layers;     /* 8   */
  int width;        /* 256 */
  int in_pos;       /* 63  */
  int in_dir;       /* 27  */
  int skip_layer;   /* 4   */
  int use_viewdirs; /* 1   */
  int out_ch;       /* outputs of `output_linear` when use_viewdirs == 0 (models/NeRF.py:196-197); ignored otherwise */
  int precision;    /* 0 or 16: bf16 MFMA operands, fp32 accumulate -- the benchmarked mode (BASELINE configs[1-3]);
                     * 32: the reference's own arithmetic (models/NeRF.py:201-243 runs in MLX float32): float32 operands
                     * on v_mfma_f32_32x32x2_f32, sinf / cosf encodings; 8 x 256 view model only.  Read by
                     * nerf_mlp_packed_bytes / nerf_mlp_pack (an fp32 model's image carries the fp32 weight streams behind
                     * the bf16 one), nerf_mlp_acts_bytes / nerf_mlp_dz_bytes (fp32 stores are larger) and every launch:
                     * use ONE arch value per model for all of them.
                     * 22 (round 4): the reference's float32 TOLERANCE on the 16-bit matrix pipe.  Calls that keep no
                     * activations (nerf_mlp_forward, nerf_query_fused / nerf_render_rays_fused with acts == NULL) run the
                     * split-fp16 kernel of csrc/mlp22.hip: every float32 operand x = hi + lo 2^-11 as two fp16 numbers
                     * (22 significand bits), a product = hi hi + 2^-11 (hi lo + lo hi) on v_mfma_f32_16x16x32_f16 with
                     * fp32 accumulate, float32-accurate encodings; operands must stay inside the fp16 range
                     * (|activation| < 65504: an overflow becomes inf / NaN in the output, never a silent wrong value).
                     * Everything that keeps activations (training forward with acts != NULL, nerf_mlp_backward) runs the
                     * split-bf16 kernels of csrc/mlp_s16.hip: x = hi + lo as two bf16 numbers (16 significand bits at
                     * float32's exponent range -- gradients do not fit fp16's), a product = hi hi + hi lo + lo hi on
                     * v_mfma_f32_32x32x16_bf16 into one fp32 accumulator; hi and lo fragment blocks in the acts / dz
                     * workspaces (twice the bf16 sizes).  Measured against the fp32 oracle: forward <= 1e-5 of the
                     * output scale, dW / db <= 3e-5 rel-L2 per tensor on equal ReLU decisions.  The image carries the
                     * bf16 streams (their fp32 bias slots are shared), the split-fp16 and the split-bf16 streams. */
} nerf_mlp_arch;

int64_t nerf_mlp_param_count(const nerf_mlp_arch* arch);
/* bytes of the packed bf16 MFMA-fragment image of the weights (forward + transposed
 * backward images + fp32 biases) that the kernels stream; rebuilt after every update.   */
int64_t nerf_mlp_packed_bytes(const nerf_mlp_arch* arch);
int nerf_mlp_pack(const nerf_mlp_arch* arch, const float* params, void* packed, void* stream);

/* bytes of the activation / gradient-activation stores for M samples (training only)    */
int64_t nerf_mlp_acts_bytes(const nerf_mlp_arch* arch, int64_t M);
int64_t nerf_mlp_dz_bytes(const nerf_mlp_arch* arch, int64_t M);

/* NeRF.forward(x): x [M, in_pos+in_dir] already embedded -> out [M,4] = [rgb, alpha] raw (view model) or
 * [M,out_ch] (image model).  The _train form also keeps the activations for nerf_mlp_backward.              */
int nerf_mlp_forward(const nerf_mlp_arch* arch, const void* packed, const float* x, int64_t M, float* out,
                     void* stream);
int nerf_mlp_forward_train(const nerf_mlp_arch* arch, const void* packed, const float* x, int64_t M, float* out,
                           void* acts, void* stream);

/* network_query_fn(pts, viewdirs, model) fused with pts = o + z d and both positional
 * encodings (models/NeRF.py:75-80 + rendering/render.py:142,226): rays [B,11], z [B,n]
 * -> raw [B,n,4].  freq_mode as nerf_encode_freq.  acts: NULL for inference, or a
 * buffer of nerf_mlp_acts_bytes(B*n) that keeps the per-layer bf16 activations for
 * nerf_mlp_backward.                                                                    */
int nerf_query_fused(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z, int64_t B,
                     int n, int freq_mode, float* raw, void* acts, void* stream);

/* Backward of nerf_query_fused w.r.t. the parameters (the reference gets this from
 * mlx autograd: entrypoints/__test_nerf.py:132,142).  d_raw [M,4]; acts from the forward;
 * dz: scratch of nerf_mlp_dz_bytes(M); grads [param_count] is OVERWRITTEN.              */
int nerf_mlp_backward(const nerf_mlp_arch* arch, const void* packed, const void* acts, const float* d_raw,
                      int64_t M, void* dz, float* grads, void* stream);
/* Same, and dL/d(position features) d_x [M, in_pos] for a trainable encoder in front of the network (the hash
 * grid: nerf_hashgrid_backward takes it).  Only for the Instant-NGP-sized instance of the reference's NeRF class
 * {n_layers 2, width 64, in_pos 32, in_dir 16, skip_layer -1, use_viewdirs 1} (BASELINE configs[4]); the 8 x 256
 * models sit behind fixed encodings and return NERF_E_UNSUPPORTED.                                             */
int nerf_mlp_backward_inputs(const nerf_mlp_arch* arch, const void* packed, const void* acts, const float* d_raw,
                             int64_t M, void* dz, float* grads, float* d_x, void* stream);

/* Test hook (no reference counterpart): one layer of the training stores of the last nerf_query_fused(acts != NULL)
 * / nerf_mlp_backward call, decoded from the fragment-block layout to row-major float32 out[M, width], so that the
 * parity tests can compare EVERY layer's activation and dZ with the oracle (and feed the kernel's own ReLU decisions
 * to the oracle's backward).  8 x 256 view model only.
 *   kind 0 (store = acts): layer 0..7 = relu(pos_l) (models/NeRF.py:221-222), 8 = feature (:231), 9 = relu(dir0)
 *                          (:235-236), 10 = position encoding (64 = 63 + pad), 11 = direction encoding (32 = 27 + pad)
 *   kind 1 (store = dz):   layer 0..7 = dL/d(pre-activation of pos_l), 8 = d feature, 9 = d dir0 pre-activation,
 *                          10 = d alpha (column 0), 11 = d rgb (columns 0..2)
 * nerf_mlp_debug_width returns `width` (16 x fragments) or -1.                                                     */
int nerf_mlp_debug_width(const nerf_mlp_arch* arch, int kind, int layer);
int nerf_mlp_debug_read(const nerf_mlp_arch* arch, const void* store, int kind, int layer, int64_t M, float* out,
                        void* stream);

/* The same rows never leaving the chip: hash gathers + SH evaluated inside the 2 x 64 forward kernel (L = 16, F = 2,
 * sh_degree = 3; arch = {2, 64, 32, 16, -1, 1}) -> raw [B,n,4]; acts as in nerf_query_fused.  The table gradient of
 * such a query takes the sample positions from the rays again: nerf_hashgrid_backward_rays(d_out = the d_x of
 * nerf_mlp_backward_inputs).                                                                                  */
int nerf_ngp_query_fused(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z, int64_t B,
                         int n, const float* tables, int L, int log2_T, int F, const int* resolutions_host,
                         int sh_degree, float pos_scale, float pos_offset, float* raw, void* acts, void* stream);
/* Round 4: the same query gathering from an fp16 SHADOW image of the tables (tables_half: [L,T] packed pairs of fp16, 4
 * bytes per entry instead of the 8 of the float32 master pair; NULL = gather `tables`) -- SURVEY 8(d) budgets 512 B of
 * gathers per sample, the float32 pairs are 1024.  Interpolation stays float32.  The shadow is written by
 * nerf_adam_step_shadow in the pass that updates the float32 master tables (encoding/multi_hash.py:79-136 keeps one
 * float32 table; the shadow is this library's, like the packed weight image of the MLPs).                         */
int nerf_ngp_query_fused_h(const nerf_mlp_arch* arch, const void* packed, const float* rays, const float* z, int64_t B,
                           int n, const float* tables, const void* tables_half, int L, int log2_T, int F,
                           const int* resolutions_host, int sh_degree, float pos_scale, float pos_offset, float* raw,
                           void* acts, void* stream);

/* ---------------------------------------------------------------- fused renderer (a14 / a18)
 * replaces: rendering/render.py:164-241 render_rays_eval (coarse pass, importance sampling, sort, second pass)
 * as ONE call that enqueues the fixed kernel sequence on `stream`: nerf_sample_coarse -> nerf_query_fused ->
 * nerf_composite_forward -> nerf_importance_sample -> nerf_query_fused -> nerf_composite_forward.
 * workspace: nerf_render_workspace_bytes(B,n,N) bytes of scratch.  packed_fine NULL = network_coarse
 * (render.py:228).  N == 0: coarse result only.  Optional outputs may be NULL.  u [B,N] = the uniforms that
 * sampling/__init__.py:140 draws with torch.rand.                                                          */
int64_t nerf_render_workspace_bytes(int64_t B, int n, int N);
int nerf_render_rays_fused(const nerf_mlp_arch* arch, const void* packed_coarse, const void* packed_fine,
                           const float* rays, int64_t B, int n, int N, const float* u, int freq_mode, int white_bkgd,
                           void* workspace, float* rgb, float* disp, float* acc, float* rgb_coarse, float* disp_coarse,
                           float* acc_coarse, float* z_vals, float* weights, void* stream);

/* ---------------------------------------------------------------- gradient all-reduce (no reference
 * counterpart: the reference is single-device).  RCCL over xGMI, one in-place float32 sum of the flat gradient
 * buffer per network step.  id: 128 host bytes from nerf_comm_unique_id on rank 0, distributed by the caller.
 * librccl is resolved at first use (NERF_E_RCCL when it is not installed).                                  */
#define NERF_COMM_ID_BYTES 128
int nerf_comm_unique_id(char* id_out_host);
int nerf_comm_init(void** comm_out, int nranks, int rank, const char* id_host);
int nerf_allreduce_grads(void* comm, float* grads, int64_t count, void* stream);
int nerf_comm_destroy(void* comm);

/* runtime selection of kernel variants (for A/B measurement only: none of these changes a result's meaning, a buffer
 * size or a layout; precision is NOT here, it is nerf_mlp_arch.precision):
 *   "mlp_variant"     0 auto | 1,2 weights via L1 (32 / 64 samples per wave) | 3 LDS ring, 32x32x16 MFMA |
 *                     4 LDS ring, 16x16x32 MFMA, 8 waves x 32 samples (render path only; auto picks it there) |
 *                     5 same with 4 waves x 64 samples.  Training kernels use 3 for every value >= 3.
 *   "ring_workgroups" persistent workgroups of the ring kernels (0 = default: one per CU of the current device)
 *   "dw_workgroups"   0 auto (one per CU) | workgroups of the weight-gradient kernel
 *   "dw_unit_bias"    fixed per-tile cost of a dW job, in KiB-of-streaming units, for its static split (negative = automatic, the
 *                     default: 128 for the bf16 kernel, 32 for the split-bf16 kernels)
 *   "dw_private_tiles" split-bf16 weight-gradient jobs of at most this many 32 x 32 output tiles (default 4 = the most; 0 = none) run as
 *                     sixteen independent wave pipelines (no workgroup barrier; fixed-order tree sum at the end) instead of the shared
 *                     4 x 4 wave grid, of which such a job occupies one wave.  Sums in a different order: ~1e-6 rel-L2, bit-reproducible
 *   "dw_ring_cap"     most stages the LDS ring of the 16-wave split-bf16 weight-gradient kernel may hold (default 8; 2 .. 16)
 *   "dw16_variant"    bf16 weight gradients: 1 (default) the 256 x 256 jobs on the one-wave-per-SIMD kernel, job lists of tiny jobs (the
 *                     2 x 64 model) on the split kernels' 16-wave kernel (wave-private pipelines), the other jobs on round 2's 16-wave
 *                     kernel | 0 every job on round 2's kernel | 2, 3: A/B forms of 1 (round 2's / round 5's kernel for every narrow job)
 *   "dw22_variant"    split-bf16 weight gradients: 1 (default) the 256 x 256 jobs on the one-wave-per-SIMD kernel (4 x 4 output tiles
 *                     per wave, two operand register sets), the other jobs on the 16-wave kernel -- two launches | 0 every job on the
 *                     16-wave kernel.  The two sum a tile's products and a bias row in different orders: gradients agree to ~1e-6
 *                     rel-L2, each setting is bit-reproducible from run to run.
 *   "hash_combine_max_res"  table-gradient scatter: levels with N_l <= this value (default 64) accumulate in LDS first and
 *                     add each distinct table entry once (coarse levels collide heavily); 0 = every level directly
 *   "ngp_ray_major"   fused configs[4] inference query: 1 (default) a 32-sample tile is one depth of 32 adjacent rays (the
 *                     lanes of a gather share cells at 13 of 16 levels when rays are neighbouring pixels), 0 = 32
 *                     consecutive depths of one ray.  Same values per sample either way.
 *   "ring_split"      1 (default): one 8-wave workgroup per CU behind a 128 KiB weight ring; 2: two independent 4-wave
 *                     workgroups behind 64 KiB rings (training forward / chain only; measured slower)
 *   "pass_queue"      1 (default): the persistent ring kernels (render forwards, training forwards and chains) hand out their passes
 *                     from a device-wide counter -- a workgroup on a fast XCD takes more passes than one on a slow XCD --; 0: static
 *                     split (pass = blockIdx.x + k gridDim.x).  Bit-identical results: which workgroup runs a pass changes nothing in it.
 *   "f22_tiles"       16-sample tiles per wave of the split-fp16 inference forward on rays + depths (precision 22): 3 (48 samples per
 *                     wave, every weight fragment pair read from the LDS feeds 9 MFMAs) | 2 (rounds 4-5: 32 samples, 6 MFMAs) | 0 (default)
 *                     = 3 except for launches of a few passes per workgroup where whole 32-sample passes divide the work better.  The
 *                     same MFMA sequence per sample: bit-identical results.
 *   "dw_unit_bias"    (see above) automatic = 128 for the bf16 kernels, 2 for the split-bf16 kernels (round 6: was 32)
 *   "dw_narrow_first" order of the two weight-gradient launches: 1 (default) the narrow jobs before the 256 x 256 jobs | 0 after.
 *                     Same gradients either way.
 * nerf_get_option returns the current value of EVERY key nerf_set_option accepts (a get / set pair restores a setting;
 * "dw_unit_bias" reads -1 while it is automatic), or NERF_OPTION_UNKNOWN for an unknown key.                        */
#define NERF_OPTION_UNKNOWN (-2147483647 - 1)
int nerf_set_option(const char* key, int value);
int nerf_get_option(const char* key);

/* ---------------------------------------------------------------- optimiser (a21)
 * replaces: mlx.optimizers.Adam.update as called at entrypoints/__test_nerf.py:134,144
 * (mlx 0.7.0: no bias correction unless bias_correction != 0).  g is multiplied by
 * grad_scale first (1/world_size after a sum all-reduce).                               */
int nerf_adam_step(float* params, const float* grads, float* m, float* v, int64_t count, float lr, float beta1,
                   float beta2, float eps, int bias_correction, int step, float grad_scale, void* stream);
/* Same update; grads is float32 [count] (grads_fixed_point = 0) or the int64 [count] fixed-point accumulators of
 * nerf_hashgrid_backward_rays_ex (1), and with zero_grads != 0 the gradient buffer is cleared in the same pass (read g,
 * write 0) so that an accumulating scatter needs no memset before the next step.                                  */
int nerf_adam_step_ex(float* params, void* grads, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                      float eps, int bias_correction, int step, float grad_scale, int grads_fixed_point, int zero_grads,
                      void* stream);
/* nerf_adam_step_ex that also writes every updated parameter as fp16 to params_half [count] (NULL: exactly
 * nerf_adam_step_ex): the shadow image nerf_ngp_query_fused_h gathers from, kept current at no extra pass.       */
int nerf_adam_step_shadow(float* params, void* grads, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
                          float eps, int bias_correction, int step, float grad_scale, int grads_fixed_point, int zero_grads,
                          void* params_half, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_HIP_H */

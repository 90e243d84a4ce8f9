/* diffsg.h -- C ABI of the MI355X-native DiffSG denoising hot path (libdiffsg_hip.so).
 *
 * The reference (qiyu3816/DiffSG) has no FFI: its seam is the Python object API.  Every entry point below is what
 * a ctypes binding for that seam needs, and names the reference code it replaces.
 *
 * Conventions
 *   - all pointers are DEVICE pointers to float32 (row-major, contiguous) unless stated; the library borrows them for
 *     the duration of the call and owns only its packed-weight arena and workspace (allocated in dsg_create /
 *     dsg_reserve / lazily on the first call of a new batch size, never while a graph is being captured);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, the calls do not synchronise;
 *   - int functions return 0 on success, non-zero on error; dsg_last_error() gives the message (thread local);
 *   - a handle is bound to the device that was current in dsg_create and is used by one host thread at a time.
 */
#ifndef DIFFSG_H_
#define DIFFSG_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsg_handle dsg_handle;

/* UNet1D constructor arguments (ddpm_opt/UNetCF.py:262-266); is_attn / middle_attn are always False in the
 * reference's call sites (classifier_free_MSR.py:202-203, _CO.py:218-219, _NU.py:230-231) and are not supported. */
typedef struct {
    int input_dim;
    int proj_dim;
    int cond_dim;
    int n_res;      /* len(dims), <= 8 */
    int dims[8];
    int n_blocks;
} dsg_unet_desc;

dsg_handle* dsg_create(const dsg_unet_desc* desc);
void dsg_destroy(dsg_handle* h);
const char* dsg_last_error(void);
/* sha256 of the sources this binary was built from (diffsg_amd/_lib.py refuses to load a library whose id differs from
 * the tree's: a stale binary must not be what the parity tests and the bench run). */
const char* dsg_build_id(void);

/* Parameter table in UNet1D.state_dict() order: replaces nn.Module parameter registration,
 * UNetCF.py:272-316.  dsg_bind_weights takes one device pointer per entry, in this order, and packs the weights
 * into MFMA fragment order on `stream`; call it again after the tensors change (optimizer step, load_state_dict). */
int dsg_param_count(const dsg_handle* h);
const char* dsg_param_name(const dsg_handle* h, int i);
long long dsg_param_numel(const dsg_handle* h, int i);
long long dsg_param_total(const dsg_handle* h);   /* sum of numel = length of the flat gradient bucket */
int dsg_bind_weights(dsg_handle* h, const float* const* ptrs, int n, void* stream);

/* Arithmetic of every GEMM of the denoiser:
 *   DSG_PRECISION_SPLIT_F16 (default)  float32-accurate GEMMs as hi/lo fp16 splits on the f16 matrix cores, f32 accumulate
 *                                      (22 significant bits per operand);
 *   DSG_PRECISION_F32_MFMA             exact float32 v_mfma_f32_32x32x2_f32.
 * The mode is per handle (no environment override) and applies to dsg_unet_forward, dsg_sample and dsg_train_step (forward,
 * data gradients and weight gradients). */
#define DSG_PRECISION_SPLIT_F16 0
#define DSG_PRECISION_F32_MFMA 1
int dsg_set_precision(dsg_handle* h, int mode);

/* Sharded sampling that must reproduce ONE reference call on the concatenation of all ranks' rows: the early-step renorm
 * (classifier_free_MSR.py:136-137: mean / unbiased variance over ALL B*D elements, the only cross-row coupling of the path)
 * then needs the other shards' moments.  With a hook set, on each of the (at most 4) renorm steps dsg_sample writes the
 * shard's { sum y, sum y^2, count } as float64 to stats3 (device), calls reduce(user) on the host -- the caller enqueues an
 * all-reduce(SUM) of the 3 doubles, in place, ordered on the same stream (torch.distributed does) -- and standardises with
 * the reduced moments; those steps are launched eagerly instead of from the captured graph.  reduce = NULL removes the hook
 * (default: every dsg_sample call standardises over its own rows, as the reference does per 512-row chunk, :273-279). */
int dsg_set_renorm_hook(dsg_handle* h, double* stats3, void (*reduce)(void* user), void* user);

/* fp16 range of the split path's RAW operands (Linear shortcuts, Down/Upsample, feature_proj inputs are split without
 * normalisation; LayerNorm outputs are bounded and weights are scaled at bind time).  Every kernel that splits a raw operand
 * bounds its rows by |mean| + sqrt(M2) from the LayerNorm statistics and raises a per-handle flag above 6e4 (fp16 saturates at
 * 65504: the result would be silently wrong, not inf).  dsg_range_status synchronises the device, returns the flag in
 * *exceeded and clears it; on 1 the outputs since the last query are not to be trusted: rerun with DSG_PRECISION_F32_MFMA. */
int dsg_range_status(dsg_handle* h, int* exceeded);
/* The same query behind the work already enqueued on ONE stream: synchronises `stream` only (other streams of the device keep running;
 * the default DDPM.sample() stalled every stream of the device once per call) and reads / clears the flag stream-ordered. */
int dsg_range_status_stream(dsg_handle* h, int* exceeded, void* stream);

/* Which FORM of the kernels a launch uses (same arithmetic per element, different work decomposition):
 *   coop_max_tiles          launches of at most this many 32-row tiles (both CFG passes counted) run the 64/128-wide
 *                           blocks cooperatively (one tile per workgroup, N/32 waves) and never as block+Linear pair
 *                           kernels; larger launches run one wave per tile (weight planes shared through LDS) and pairs;
 *   narrow_small_max_tiles  launches of at most this many tiles use the small-launch form of the fused narrow run.
 * Defaults 512 / 2048; a negative value restores the default, 0 forces the large-launch forms at every size (the
 * parity tests run every golden both ways).  Cached step graphs are dropped when the policy changes. */
int dsg_set_launch_policy(dsg_handle* h, int coop_max_tiles, int narrow_small_max_tiles);

/* Per-handle switches of alternative kernel FORMS (measurement and parity: every form is checked against the same goldens).
 *   DSG_OPT_NARROW_VALU8   1 (default): in large sampling launches the 8-wide bottom of the net (Downsample 16 -> 8, the 8-wide
 *                          Down / Middle / Up blocks of UNetCF.py:278-311, Upsample 8 -> 16) runs on the vector unit in exact
 *                          float32 with its skip tensors in registers (csrc/dsg_narrow8.hpp); 0: on the matrix cores like the rest
 *                          of the narrow run.  Cached step graphs are dropped when the value changes.
 *   DSG_OPT_TRAIN_TIME_BESIDE  1 (default): in training steps that use the side stream (>= 32 768 rows) the time-path backward
 *                          (TimeEmbedding and the per-block time_emb Linear, UNetCF.py:35-44) runs on the side stream beside the last
 *                          weight-gradient launch; 0: behind it on the caller's stream.  Same gradients bit for bit either way.
 *   DSG_OPT_WGRAD_NARROW_PART  0 (default) / 1: the weight gradients of the fused narrow run's blocks as one more early part on the side
 *                          stream, behind the narrow run's backward launch (measured slower at 32 768 rows: profiles/r04_train_tail_ab.txt).
 *   DSG_OPT_TILE_STEP      1 (default): launches of at most `coop_max_tiles` row tiles (dsg_set_launch_policy) run one denoiser pass as
 *                          feature_proj + ONE launch that carries every row tile through all the other operators (csrc/dsg_tile.hpp;
 *                          UNetCF.py:318-356 has no cross-row operation); 0: one launch per operator / fused run, as larger launches.
 *                          Same arithmetic per operator, bit-identical results.  Cached step graphs are dropped when the value changes.
 *   DSG_OPT_PANEL_HALF     1 (default) / 0: the persistent 128-wide kernels of large sampling launches with half-size weight panels and
 *                          4-wave workgroups, two per CU, each streaming its own panels (csrc/dsg_panel.hpp, STEPS = 2): the two waves of a
 *                          SIMD then belong to different workgroups and no longer run the block program in lock-step.  Same arithmetic and
 *                          accumulation order per element: bit-identical results; +0.3..0.9 % at 65 536 rows, 3-11 % at 12 288 - 49 152 rows.
 *   DSG_OPT_F32_PAIR       1 (default) / 0: exact-float32 path, inference: a >= 64-wide ResidualBlock and the Linear that consumes its
 *                          output (Downsample / Upsample, UNetCF.py:230-257; `final`, UNetCF.py:356) in one launch, the Linear fed from
 *                          the block's accumulators (csrc/dsg_kernels.hpp, k_resblock_lin).  Bit-identical to the two launches. */
enum { DSG_OPT_NARROW_VALU8 = 1, DSG_OPT_TRAIN_TIME_BESIDE = 2, DSG_OPT_WGRAD_NARROW_PART = 4, DSG_OPT_TILE_STEP = 8, DSG_OPT_PANEL_HALF = 16,
       DSG_OPT_F32_PAIR = 32 };
int dsg_set_option(dsg_handle* h, int option, int value);

/* Pre-size the workspace for up to `max_rows` batch rows and `max_entries` time-table rows. */
int dsg_reserve(dsg_handle* h, int max_rows, int max_entries);

/* eps[B][D] = UNet1D.forward(x[B][D], t[B], cond[B][C], cond_mask[B])   (UNetCF.py:318-356).
 * t holds the already-divided time value per row, as the reference passes it (classifier_free_MSR.py:109,126). */
int dsg_unet_forward(dsg_handle* h, const float* x, const float* t, const float* cond, const float* cond_mask,
                     float* out, int B, void* stream);

/* y0[B][D] = DDPM.sample(cond, omega)   (classifier_free_MSR.py:114-155), T reverse steps, two denoiser passes each.
 *   coef  [T][4] per step i: { betas[i]/sqrt_one_minus_alphas_cumprod[i], reciprocal_sqrt_alphas[i],
 *                              (1-alphas_cumprod[max(i-1,0)])/(1-alphas_cumprod[i]), i > 1 ? 1 : 0 }  (float32,
 *         computed by the caller from the registered buffers so that their float64->float32 casts are kept);
 *   y_T   [B][D] start state, or NULL to draw it on the device (Philox, `seed`);
 *   noise [T-2][B][D] the z of steps i = T-1 .. 2 in that order, or NULL to draw on the device;
 *   flags DSG_SAMPLE_NO_GRAPH: launch eagerly instead of replaying the captured graphs (per workspace shape: one early step and
 *         runs of 1, 2, 4, 8, 16, 32 later steps; a call replays min(T, 4) early steps and the binary decomposition of the rest --
 *         T = 20 is five graph launches; the step index is a device counter, so the graphs do not depend on T). */
#define DSG_SAMPLE_NO_GRAPH 1
/*         DSG_SAMPLE_PROFILE: eager launch with one HIP-event pair around every operator launch on `stream`
 *         (synchronises once per step); read the totals back with dsg_op_profile.  The profile walks the operator list one launch per
 *         operator (the fused narrow run and the split path's block + Linear pairs booked on their first operator): launches small enough
 *         for the one-launch-per-pass form (DSG_OPT_TILE_STEP, <= coop_max_tiles tiles) and the exact path's block + Linear pairs
 *         (DSG_OPT_F32_PAIR) are timed in their per-operator forms -- same arithmetic, not the launches the default path issues there. */
#define DSG_SAMPLE_PROFILE 2
int dsg_sample(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed,
               float omega, const float* coef, int T, float* out, int B, int flags, void* stream);

/* dsg_sample with the denoise trajectory recorded on the device (DDPM.record_denoise_path, classifier_free_MSR.py:139-141):
 * rec_y / rec_eps [T][B][D] receive y_t (after the early-step renorm) and the guided eps of every step, first step first;
 * either may be NULL.  Replaces the reference's per-step device-to-host copies. */
int dsg_sample_rec(dsg_handle* h, const float* cond, const float* y_T, const float* noise, unsigned long long seed,
                   float omega, const float* coef, int T, float* out, int B, int flags, float* rec_y, float* rec_eps,
                   void* stream);

/* The reference evaluates a test set as a sequence of INDEPENDENT sample() calls over consecutive `chunk_rows`-row slices
 * (classifier_free_MSR.py:257,273-279: 512 rows per call; CO :344-366, NU :318-336): every chunk draws its own start state and
 * noise and standardises the early steps with its OWN statistics.  dsg_sample_chunked runs ceil(B / chunk_rows) such calls in one
 * set of launches: chunk c uses the Philox stream seeds[c] (host array, one per chunk) with chunk-local element indices and a
 * segmented renorm reduction, so its rows are bit-identical to dsg_sample(cond + c * chunk_rows * C, ..., seeds[c], ...) on that
 * slice.  chunk_rows must be a multiple of 32; the last chunk may be shorter.  y_T / noise, when given, cover the whole batch
 * ([B][D], [T-2][B][D]).  Not available while a renorm hook is installed. */
int dsg_sample_chunked(dsg_handle* h, const float* cond, const float* y_T, const float* noise, const unsigned long long* seeds,
                       int chunk_rows, float omega, const float* coef, int T, float* out, int B, int flags, void* stream);

/* One training step's forward + backward: loss = DDPM.forward(y, cond) (classifier_free_MSR.py:100-112) and
 * d(loss)/d(theta) for every denoiser tensor, as `loss.backward()` produces them (classifier_free_MSR.py:223-224).
 *   y [B][D], cond [B][C] row-major;  ts [B] int32 in [0,T);  noise [B][D];  cond_mask [B] (0/1) -- the three random
 *   draws of the reference (MSR.py:101,102,107) are made by the caller so that it keeps their order and generator;
 *   sqrt_acp / sqrt_1m_acp [T] = the registered buffers sqrt_alphas_cumprod / sqrt_one_minus_alphas_cumprod;
 *   grads_flat [dsg_param_total] receives the gradients concatenated in state-dict order (the DP all-reduce bucket);
 *   loss_out  one float on the device.
 * The weights bound by the last dsg_bind_weights are used; rebind after the optimizer step. */
int dsg_train_step(dsg_handle* h, const float* y, const float* cond, const int* ts, const float* noise,
                   const float* cond_mask, const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat,
                   float* loss_out, int B, void* stream);

/* The three random draws of a training step on the device (`ts|null, noise|null, mask|null, seed`): Philox4x32-10
 * keyed by (seed, call), independent streams for  ts [B] ~ U{0..T-1}  (MSR.py:101),  noise [B][D] ~ N(0,1)  (MSR.py:102, Box-Muller)
 * and  cond_mask [B] ~ Bernoulli(keep_prob)  (MSR.py:107, keep_prob = 1 - uncond_prob).  Not torch's generator: a run that
 * must reproduce the reference's draws keeps drawing on the caller's side and passes them to dsg_train_step.
 * dsg_train_draws writes them to caller buffers (any may be NULL); dsg_train_step_seeded draws into the handle's workspace and
 * runs the step on them -- identical to dsg_train_draws followed by dsg_train_step, without the three generator launches and the
 * layout/dtype conversion passes in front of the fused step. */
int dsg_train_draws(unsigned long long seed, unsigned long long call, int T, float keep_prob, int B, int D, int* ts,
                    float* noise, float* cond_mask, void* stream);
int dsg_train_step_seeded(dsg_handle* h, const float* y, const float* cond, unsigned long long seed, unsigned long long call,
                          float keep_prob, const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat,
                          float* loss_out, int B, void* stream);

/* Measurement hook for bench.py's train leg: with the profile enabled every dsg_train_step records HIP events on `stream`
 * at its phase boundaries (no synchronisation); dsg_train_profile waits for the last profiled step and returns the times in
 * ms of { forward (+ q_sample, loss), activation backward, column sums, grouped weight-gradient launch, reduce + time path }. */
int dsg_train_profile_enable(dsg_handle* h, int on);
int dsg_train_profile(dsg_handle* h, float* ms5);

/* One Adam step (torch.optim.Adam, no amsgrad) over a flat float32 range: p, exp_avg, exp_avg_sq updated in place from g.  Element for
 * element the arithmetic of torch's FUSED kernel (ATen/native/cuda/fused_adam_utils.cuh: moment updates in double), i.e. bit-identical to
 * torch.optim.Adam(fused=True).  The reference constructs Adam with torch's default (foreach: float32 lerp / addcmul,
 * classifier_free_MSR.py:213), which can differ from the fused form in the last bit of an update; the parity tests follow the reference's
 * trajectory with a tolerance (three steps against CPU Adam).  `step` is the update's number (1 for the first).  Replaces torch's
 * multi_tensor_apply launch, which gives the 1.6 M-element flat parameter vector to 26 workgroups. */
int dsg_adam_step(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int maximize, long long step, void* stream);
/* Graph-capturable forms of the two calls whose arguments change from step to step (round 6: a whole training step -- draws, forward,
 * backward, Adam, re-pack -- replayed as ONE captured graph; diffsg_amd/train.py StepGraph).  The values a launch would otherwise bake in
 * are read from DEVICE memory and moved on by the call itself:
 *   dsg_train_step_seeded_dyn  = dsg_train_step_seeded with call = *call_dev, then *call_dev += 1
 *   dsg_adam_step_dyn          = *step_dev += 1, then dsg_adam_step with lr = *lr_dev and step = *step_dev (the count after the update)
 * Same kernels on the same operands: the results are those of the by-value calls bit for bit. */
int dsg_train_step_seeded_dyn(dsg_handle* h, const float* y, const float* cond, unsigned long long seed, unsigned long long* call_dev,
                              float keep_prob, const float* sqrt_acp, const float* sqrt_1m_acp, int T, float* grads_flat, float* loss_out,
                              int B, void* stream);
int dsg_adam_step_dyn(float* p, const float* g, float* exp_avg, float* exp_avg_sq, long long n, const double* lr_dev, double beta1, double beta2,
                      double eps, double weight_decay, int maximize, float* step_dev, void* stream);

/* avg = decay*avg + one_minus_decay*p over n floats   (ddpm_opt/ema.py:11-12). */
int dsg_ema_update(float* avg, const float* p, float decay, float one_minus_decay, long long n, void* stream);

/* ---- Solution decoders and objective evaluators.  Row-major float32 device tensors; no handle: they do
 * not depend on the denoiser.  Stream-ordered; scratch for the two global reductions comes from hipMallocAsync. ---- */
/* out[r][:] = softmax(y[r][:])   (torch.softmax(dim=1); MSR.py:147 for the first recorded states) */
int dsg_row_softmax(const float* y, float* out, long long rows, int D, void* stream);
/* custom_decoder, classifier_free_MSR.py:239-245: min-max over the WHOLE tensor, then a row softmax. */
int dsg_msr_decode(const float* y, float* out, long long rows, int D, void* stream);
/* rate[r] = sum_c log2(1 + p[r][c] * gain[r][c])   (classifier_free_MSR.py:287-288). */
int dsg_msr_rate(const float* p, const float* gain, float* rate, long long rows, int D, void* stream);
/* customized_real_decoder, classifier_free_CO.py:281-290: row softmax; rows with every entry < -10 become zero. */
int dsg_co_decode(const float* y, float* out, long long rows, int D, void* stream);
/* cost_calc, classifier_free_CO.py:255-278: X [rows][3n] = (local, transition, exec) per node, Y [rows][n] decoded shares. */
int dsg_co_cost(const float* X, const float* Y, float* cost, long long rows, int n, void* stream);
/* custom_decoder, classifier_free_NU.py:267-276: columns 0,1 min-max scaled (global over both) to width x height,
 * columns 2.. softmax * p_sum. */
int dsg_nu_decode(const float* y, float* out, long long rows, int D, float width, float height, float p_sum, void* stream);
/* rate_calc, classifier_free_NU.py:279-303: NOMA-SIC sum rate; Yd [rows][K+2] decoded, X [rows][2K] user positions; K <= 32. */
int dsg_nu_rate(const float* Yd, const float* X, float* rate, long long rows, int K, void* stream);

/* ---- Label generator of the MSR problem: SUM_RATE_GEN, utils/dataset_generate.py:280-313 ("LRH gradient
 * descent", float64 like the reference).  gs [rows][M] channel gains (the reference draws them with np.random.uniform; the
 * caller does), W total power; schemes [rows][M] and rates [rows] are written.  M <= 128.  Stream-ordered. */
int dsg_sum_rate_gen(const double* gs, double* schemes, double* rates, long long rows, int M, double W, void* stream);

/* ---- Label generator of the CO problem: the exhaustive search of CONV_CO_MINLP_GEN,
 * utils/dataset_generate.py:147-245 (float64): per sample all 2^n offloading decisions x all allocations of the server capacity
 * on the grid `choices` [nch] (np.arange(0.02, 1.02, 0.02), drawn up by the caller) that sum to 1.  params [rows][7][n] =
 * s, c, f_local, alpha, beta, r_u, cost_local per node (the caller's numpy draws and derived values, :169-184); Y [rows][2n+1]
 * receives decision | allocation | cost of the LAST delay-tolerable candidate if there is one (tolerable[r] = 1), else of the
 * FIRST cheapest one -- the reference's update rules.  n <= 7.  Stream-ordered. */
int dsg_co_minlp_search(const double* params, const double* choices, int nch, double* Y, int* tolerable, long long rows, int n,
                        double F_t, double P_t, double P_I, double theta, void* stream);

/* Measurement hooks for bench.py: the per-step operator list and a timed replay of one operator's kernel with HIP
 * events on `stream` (rows = B rows, both passes, as inside dsg_sample). */
int dsg_op_count(const dsg_handle* h);
/* Operators [lo, hi) run as ONE fused launch at inference (the narrow middle of the U-Net); lo == hi if none. */
int dsg_fused_range(const dsg_handle* h, int* lo, int* hi);
/* name: >= 64 bytes.  flops/bytes are ALGORITHMIC per batch row per reverse step (both passes; the step-invariant
 * time path and condition embeddings are computed once per call and are not counted). */
int dsg_op_info(const dsg_handle* h, int op, char* name, double* flops_per_row, double* bytes_per_row);
int dsg_time_op(dsg_handle* h, int op, int B, int iters, float* ms_avg, void* stream);
/* Summed HIP-event time (ms) and launch count of operator `op` over the last DSG_SAMPLE_PROFILE call. */
int dsg_op_profile(const dsg_handle* h, int op, double* ms_total, int* calls);

/* Box calibration for bench.py (`box`): four fixed probes on the current device, median of five 3-10 ms launches each, timed with
 * HIP events on `stream` (synchronises): out[0] = rate of a dependent v_mfma_f32_32x32x16_f16 loop on every SIMD (TFLOP/s, dense f16),
 * out[1] = the same loop with six vector instructions behind every MFMA (1e9 slots/s), out[2] = a 256 MiB device copy (GB/s, read +
 * written), out[3] = a frozen miniature of the 128-wide panel kernels' load profile (LDS-DMA weight stream, activation stream, LDS
 * reads, MFMAs and vector instructions interleaved; 1e9 MFMA slots/s) -- `out` has FOUR floats.  Not part of the reference's seam (the reference has no benchmark): it lets a reader separate a slow box from a slow tree. */
int dsg_box_calibrate(float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSG_H_ */

/* libicn -- C ABI of the MI355X-native icosahedral hex-convolution hot path.
 *
 * Drop-in boundary for the operators GenIcoNet imports from the (un-vendored) `icocnn` package:
 *   icocnn.ico_conv.IcoConvS2S      reference call sites models.py:14,25-33,46-55,104-109,165-170,269-284
 *   icocnn.ico_conv.IcoUpsampleS2S  reference call sites models.py:13,45,53
 *   icocnn.utils.ico_geometry.get_ico_faces   reference call sites losses.py:34, run.py:144,529
 * The reference has no FFI of its own (it is pure Python over PyTorch); these entry points are what a
 * ctypes binding of those modules binds (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - All activations are fp32, channels-last per chart: x[b][c5][i][j][ch] == PyTorch
 *     `torch.channels_last` storage of the (B, C, 5*2^r, 2^(r+1)) tensor the reference passes around
 *     (layout pinned by data.py:64-69).  Row-major pixel id p = (c5*n + i)*2n + j, n = 2^r.
 *   - Weights: w[Cout][Cin][7] fp32; tap order = centre, then the six hex neighbours (di,dj) =
 *     (+1,0),(0,+1),(-1,+1),(-1,0),(0,-1),(+1,-1).  bias[Cout] or NULL.
 *   - `r_in` is the subdivision level of the op's INPUT (models.py:25-29 semantics); stride 2 halves it.
 *   - corner_mode: 0 = 'zeros', 1 = 'average' (run.py:683); pole value = mean of its 5 corner pixels
 *     (losses.py:49-51).
 *   - Every device buffer (incl. workspace) is owned by the caller.  Calls are asynchronous on `stream`
 *     (a hipStream_t).  The library keeps only immutable per-(device, r, stride, corner_mode) index tables,
 *     built once under a mutex on first use (icn_prepare builds them eagerly, e.g. before graph capture).
 *   - Return 0 on success, negative on error; message via icn_last_error() (thread-local).  Nothing
 *     throws across the boundary.
 */
#ifndef ICN_H
#define ICN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICN_ABI_VERSION 7

#define ICN_CORNER_ZEROS 0
#define ICN_CORNER_AVERAGE 1

/* index codes of the host tables (see icn_table_*): >=0 pixel id, -1 nothing, -2/-3 north/south pole mean */
#define ICN_IDX_ZERO (-1)
#define ICN_IDX_POLE (-2)

#define ICN_OP_CONV_FWD 0
#define ICN_OP_CONV_BWD_DATA 1
#define ICN_OP_CONV_BWD_WEIGHT 2

int icn_abi_version(void);
const char* icn_last_error(void);

/* Build (and upload to the current device) the index tables of one conv / upsample configuration. */
int icn_prepare_conv(int r_in, int stride, int corner_mode);
int icn_prepare_upsample(int r_in, int corner_mode);

/* Bytes of caller-provided workspace an op needs (0 is a valid answer).  For the forward and data-gradient ops of the MFMA
 * path this includes ~17 MB of stream-K scratch: the persistent GEMM cuts the tiles of its last rounds in K across
 * workgroups, which park partial tiles there; the contents need not survive the call. */
size_t icn_conv_workspace_bytes(int op, int B, int Cin, int Cout, int r_in, int stride);

/* y[b,p,co] = bias[co] + sum_t sum_ci w[co,ci,t] * x[b, nbr_t(p), ci]         (IcoConvS2S.forward) */
int icn_conv_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout,
                 int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream);

/* dx = transpose of the gather, incl. the pole-mean and duplicated taps at five-valent pixels */
int icn_conv_bwd_data(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int r_in,
                      int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream);

/* dw[co,ci,t] = sum_{b,p} dy[b,p,co] * x[b,nbr_t(p),ci];  dbias[co] = sum dy (dbias may be NULL) */
int icn_conv_bwd_weight(const float* x, const float* dy, float* dw, float* dbias, int B, int Cin, int Cout,
                        int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream);

/* Pair forms: two convolutions of the SAME input with equal stride / level / corner mode -- the conv00 / conv10 branches
 * of the reference's residual blocks (models.py:25-39 BasicIcoS2SDownBlock, 43-62 BasicIcoS2SUpBlock) -- as one launch
 * per pass: forward with the output channels concatenated (one gather feeds both), bwd-data with dy0 | dy1 concatenated
 * along the GEMM's K axis (one dx; no separate gradient add), bwd-weight with the shared x gathered once.  Results are
 * those of the two single calls (bwd-data: of their sum).  icn_conv_pair_supported() says whether a shape can take this
 * path (equal output channel counts, multiples of 64, tensors below 2 GiB); otherwise use the single-conv entry points.
 * bias0 / bias1: both or neither.  Workspace: icn_conv_pair_workspace_bytes(op, ...), op as above. */
int icn_conv_pair_supported(int B, int Cin, int Cout0, int Cout1, int r_in, int stride);
size_t icn_conv_pair_workspace_bytes(int op, int B, int Cin, int Cout0, int Cout1, int r_in, int stride);
int icn_conv_pair_fwd(const float* x, const float* w0, const float* bias0, const float* w1, const float* bias1, float* y0, float* y1,
                      int B, int Cin, int Cout0, int Cout1, int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes,
                      void* stream);
int icn_conv_pair_bwd_data(const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, int B, int Cin, int Cout0,
                           int Cout1, int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes, void* stream);
int icn_conv_pair_bwd_weight(const float* x, const float* dy0, const float* dy1, float* dw0, float* dbias0, float* dw1, float* dbias1,
                             int B, int Cin, int Cout0, int Cout1, int r_in, int stride, int corner_mode, void* ws, size_t ws_bytes,
                             void* stream);

/* r_in -> r_in+1: copy at coarse sites (2i,2j+1), mean of the two edge endpoints elsewhere  (IcoUpsampleS2S) */
int icn_upsample_fwd(const float* x, float* y, int B, int C, int r_in, int corner_mode, void* stream);
int icn_upsample_bwd(const float* dy, float* dx, int B, int C, int r_in, int corner_mode, void* stream);

/* Composite  conv_stride1(upsample(x))  -- the first two operators of every decoder block of the reference (models.py:58-60:
 * conv00(upsample00(x)), conv10(upsample10(x)); also its IcoUpS2S, models.py:9-20) -- computed from the COARSE tensor x
 * (B, 10*4^r_in, Cin) in one gather-GEMM: the upsample is linear, so away from the 12 singular vertices an output pixel is
 * a fixed combination of 7 (coarse-site pixels) or 4 (edge midpoints) coarse pixels with effective weights
 * sum_t alpha[t] W_t; the 4x larger upsampled tensor never exists and 0.68 of the multiply-adds of the two operators run.
 * Same results as icn_upsample_fwd followed by icn_conv_fwd / icn_conv_pair_fwd at level r_in + 1, up to fp32 rounding
 * order.  Cout1 = 0, w1 = y1 = NULL: one convolution; else a pair sharing x (both biases or neither).  Shapes outside
 * icn_upconv_supported() (Cin % 32, Cout % 64, tensors below 2 GiB) take the two separate calls.  Forward only: the
 * backward passes are those of the separate operators (the caller recomputes the upsample from x). */
int icn_upconv_supported(int B, int Cin, int Cout0, int Cout1, int r_in);
size_t icn_upconv_workspace_bytes(int B, int Cin, int Cout0, int Cout1, int r_in);
int icn_upconv_fwd(const float* x, const float* w0, const float* bias0, const float* w1, const float* bias1, float* y0, float* y1, int B,
                   int Cin, int Cout0, int Cout1, int r_in, int corner_mode, void* ws, size_t ws_bytes, void* stream);

/* Backward of the same pair, aggregated at the COARSE level.  With U the upsample matrix and nbr_t the fine conv's gather,
 *     g_t[s] = sum_p U[nbr_t(p), s] dy[p]           (one HBM-bound pass over dy0 | dy1; 7 coarse tensors, tap-major rows)
 * both gradients become dense coarse-level contractions with a QUARTER of the fine level's multiply-adds and no gather:
 *     dx[s] = sum_t W_t^T g_t[s]        dW_t = sum_s x[s]^T g_t[s]        dbias = sum_s g_0[s]
 * -- equal to icn_conv_(pair_)bwd_data at level r_in + 1 followed by icn_upsample_bwd, and to icn_conv_(pair_)bwd_weight on
 * icn_upsample_fwd(x), up to fp32 rounding order.  dx and (dw0, dbias0, dw1, dbias1) are each optional (NULL); x is only read
 * for the weight gradients, w0 / w1 only for dx.  corner_mode 'average' only (dbias relies on upsample rows summing to one);
 * other shapes / modes: icn_upconv_bwd_supported() == 0 and the caller uses the separate operators' backward. */
int icn_upconv_bwd_supported(int B, int Cin, int Cout0, int Cout1, int r_in, int corner_mode);
size_t icn_upconv_bwd_workspace_bytes(int B, int Cin, int Cout0, int Cout1, int r_in);
int icn_upconv_bwd(const float* x, const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, float* dw0,
                   float* dbias0, float* dw1, float* dbias1, int B, int Cin, int Cout0, int Cout1, int r_in, int corner_mode, void* ws,
                   size_t ws_bytes, void* stream);
/* The same with the weight gradients on a stream of their own (ABI 5): the aggregate pass and dx run on `stream`; the weight
 * gradient (and its slab reduction) on `weight_stream`, ordered after the aggregate pass by an event, so that it can run beside
 * whatever the caller gives `stream` next.  The caller owns the join (weight_stream -> whoever reads dw*) and keeps x, dw*, ws
 * alive for weight_stream.  weight_stream == NULL or == stream: exactly icn_upconv_bwd. */
int icn_upconv_bwd_streams(const float* x, const float* dy0, const float* dy1, const float* w0, const float* w1, float* dx, float* dw0,
                           float* dbias0, float* dw1, float* dbias1, int B, int Cin, int Cout0, int Cout1, int r_in, int corner_mode,
                           void* ws, size_t ws_bytes, void* stream, void* weight_stream);

/* Fused BatchNorm + ReLU of the residual blocks (training mode; replaces the torch builtins at models.py:36-40,58-62).
 * Tensors are channels-last rows (M = B * pixels, C), C in {64, 128, 256, 512, ...: C % 4 == 0 and 256 % (C/4) == 0}.
 *   stat   [2*C]  batch mean | 1/sqrt(var + eps), written by icn_bn_stats and read by the other two
 *   ws     at least icn_bn_workspace_floats(M, C) floats
 * icn_bn_stats also updates running_mean / running_var (momentum, unbiased variance) when they are not NULL.
 * icn_bn_relu_fwd : y = relu(bn_a(a) [+ bn_b(b)])            (b == NULL: single input)
 * icn_bn_relu_bwd : da [, db], and sums[k*C + c]: k = 0 -> d(beta), 1 -> d(gamma_a), 2 -> d(gamma_b).  The ReLU mask is
 *                   recomputed from a (, b) and the affine parameters with the forward's own expression, so the saved
 *                   output y is not read (one tensor less to stream in each of the two backward passes).  The parameter
 *                   gradients are ALSO written to dbeta_a / dgamma_a / dbeta_b / dgamma_b [C] where not NULL (ABI 4): their
 *                   own tensors, which may be views into a DistributedDataParallel bucket; `sums` stays the contiguous
 *                   copy the second pass reads.                                                                     */
size_t icn_bn_workspace_floats(int M, int C);
/* icn_bn_stats2: the statistics of the two inputs of relu(bn_a(a) + bn_b(b)) in one pass (two launches instead of four);
 * same results as two icn_bn_stats calls. */
int icn_bn_stats2(const float* a, const float* b, int M, int C, float eps_a, float momentum_a, float* running_mean_a,
                  float* running_var_a, float* stat_a, float eps_b, float momentum_b, float* running_mean_b, float* running_var_b,
                  float* stat_b, float* ws, void* stream);
int icn_bn_stats(const float* x, int M, int C, float eps, float momentum, float* running_mean, float* running_var, float* stat,
                 float* ws, void* stream);
int icn_bn_relu_fwd(const float* a, const float* b, const float* stat_a, const float* stat_b, const float* gamma_a,
                    const float* beta_a, const float* gamma_b, const float* beta_b, float* y, int M, int C, void* stream);
int icn_bn_relu_bwd(const float* dy, const float* a, const float* b, const float* stat_a, const float* stat_b, const float* gamma_a,
                    const float* beta_a, const float* gamma_b, const float* beta_b, float* da, float* db, float* sums, float* ws, int M,
                    int C, float* dbeta_a, float* dgamma_a, float* dbeta_b, float* dgamma_b, void* stream);

/* Fused output head  y = tanh(x W^T + b)  (reference models.py:151-154: Conv2d(64, 3, 1x1) + Tanh).
 * x (M, Cin) channels-last rows, w (Cout, Cin), Cin in {16,32,64,128,256}, Cout <= 4; y (M, Cout).
 * icn_head_bwd: dx (may be NULL), dw (Cout, Cin), db (Cout); ws >= icn_head_workspace_floats(M, Cin) floats. */
size_t icn_head_workspace_floats(int M, int Cin);
int icn_head_fwd(const float* x, const float* w, const float* bias, float* y, int M, int Cin, int Cout, void* stream);
int icn_head_bwd(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db, float* ws,
                 int M, int Cin, int Cout, void* stream);

/* Point-to-point loss of the training step (reference losses.py:10-85 Point2Point_Loss.forward; the absent mesh helpers as
 * restated from generate.py:20-43: area-weighted vertex normals; uniform Laplacian).
 *   lap_mode           convention of the Laplacian rows (upstream's mesh.utils.compute_laplacian, generate.py:197, is absent,
 *                      so sign and normalisation are the caller's choice): ICN_LAP_MEAN_MINUS_V (default of the Python
 *                      surface) = mean(1-ring) - v; | ICN_LAP_FLIP = v - mean(1-ring); | ICN_LAP_VALENCE = times the valence,
 *                      i.e. sum(1-ring) - k v (with FLIP: k v - sum)
 *   grid   (B, P, 3)   network output, channels-last; vertices = pixels, then N / S pole = mean of their 5 corner pixels
 *                      (ico_utils.py:10-24, losses.py:47-51)
 *   target (B, 9, V)   V = P + 2: rows 0:3 positions, 3:6 normals, 6:9 Laplacians (data.py:64-69)
 *   terms  [4]         mse(v, pos) | mean(1 - cos(normal(v), nor)) | mse(lap(v), lap) | f_pos*[0] + f_nor*[1] + f_lap*[2]
 * All three terms are always evaluated (the reference reports them every iteration, losses.py:72-81).
 * icn_p2p_loss_bwd: dgrid (B, P, 3) = upstream * d terms[3] / d grid for any factors -- the auto-encoder's 1 / 0 / 0
 * (run.py:690-692; one kernel, ws may be null) as well as the VAE's 0.6 / 0.2 / 0.2 (run.py:694-696; two kernels, gather form,
 * no atomics).  Where a normalisation clamp of the normal term is active (|vertex normal| <= 1e-10 or |target normal| <= 1e-8)
 * that vertex contributes no normal-term gradient.  upstream: device scalar dLoss/dterms[3].
 * ws: icn_p2p_loss_workspace_floats / icn_p2p_loss_bwd_workspace_floats (B, r) floats.  Deterministic (fixed two-level sums). */
#define ICN_LAP_MEAN_MINUS_V 0
#define ICN_LAP_FLIP 1
#define ICN_LAP_VALENCE 2
size_t icn_p2p_loss_workspace_floats(int B, int r);
int icn_p2p_loss_fwd(const float* grid, const float* target, int B, int r, float f_pos, float f_nor, float f_lap, int lap_mode,
                     float* terms, float* ws, void* stream);
size_t icn_p2p_loss_bwd_workspace_floats(int B, int r);
int icn_p2p_loss_bwd(const float* grid, const float* target, const float* upstream, int B, int r, float f_pos, float f_nor,
                     float f_lap, int lap_mode, float* dgrid, float* ws, void* stream);

/* KL term of the VAE loss (reference losses.py:105):  out[0] = mean_b(-0.5 * mean_d(1 + logvar - mu^2 - exp(logvar)))
 * = -0.5 / n * sum over all n = B * D elements.  mu / logvar: n contiguous floats in the same element order.
 * Deterministic two-level sum; ws >= icn_kld_workspace_floats(n) floats.
 * icn_kld_bwd: dmu = upstream * mu / n,  dlogvar = upstream * 0.5 * (exp(logvar) - 1) / n   (upstream: device scalar). */
size_t icn_kld_workspace_floats(size_t n);
int icn_kld_fwd(const float* mu, const float* logvar, size_t n, float* out, float* ws, void* stream);
int icn_kld_bwd(const float* mu, const float* logvar, const float* upstream, size_t n, float* dmu, float* dlogvar, void* stream);

/* Reparameterisation of the VAE (reference models.py:89-92):  z = eps * exp(0.5 * logvar) + mu, n contiguous floats each;
 * the caller draws eps (torch.randn_like in the reference).  Backward: dmu = dz, dlogvar = dz * eps * 0.5 * exp(0.5 * logvar)
 * (dmu may alias dz's storage only if the caller no longer needs dz). */
int icn_reparam_fwd(const float* mu, const float* logvar, const float* eps, size_t n, float* z, void* stream);
int icn_reparam_bwd(const float* dz, const float* logvar, const float* eps, size_t n, float* dmu, float* dlogvar, void* stream);

/* Test-time metric of the reference (ico_utils.py:26-44 computeDistance, mode 'point2mesh'; upstream calls kaolin 0.9.1's
 * kaolin.metrics.trianglemesh.point_to_mesh_distance): for every point of (B, P, 3) the squared distance to the closest point
 * of the triangle mesh vertices (B, V, 3) / faces (F, 3) (shared by the batch; indices are NOT range-checked on the device),
 * the index of a closest face (lowest on ties) and where on it the closest point lies: 0 interior, 1 / 2 / 3 vertex 0 / 1 / 2,
 * 4 / 5 / 6 edge (0,1) / (1,2) / (2,0). */
int icn_point_to_mesh(const float* points, const float* vertices, const int32_t* faces, int B, int P, int V, int F, float* dist2,
                      int32_t* face, int32_t* kind, void* stream);

/* Adam step (reference run.py:253 `optimizer.step()` on the optimiser of run.py:446, torch.optim.Adam without amsgrad /
 * maximize) over `count` fp32 tensors, one launch per 96 tensors that share a step count (normally: one launch).  The four pointer arrays and numel / step_size / bc2_sqrt
 * are HOST arrays of length count; the pointers in them are device pointers (param, grad, exp_avg, exp_avg_sq of tensor i,
 * numel[i] contiguous elements each).  Per tensor, as torch computes them on the host in double precision:
 * step_size[i] = lr / (1 - beta1^t_i), bc2_sqrt[i] = sqrt(1 - beta2^t_i), t_i = the tensor's step count AFTER this step.
 *   g' = g + weight_decay * p;  m += (g' - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g' * g';
 *   p -= step_size * m / (sqrt(v) / bc2_sqrt + eps)
 * HBM-bound: 28 bytes per parameter. */
int icn_adam_step(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const size_t* numel, const float* step_size, const float* bc2_sqrt, double beta1, double beta2, double eps,
                  double weight_decay, void* stream);

/* The same step with its two per-step scalars read from DEVICE memory (ABI 7): scalars_dev[0] = step_size, scalars_dev[1] = bc2_sqrt,
 * shared by all `count` tensors (they have one step count).  For a training step recorded into a HIP graph (SURVEY 7 step 7,
 * the loop of run.py:240-254): kernel arguments are frozen at capture, so the host writes the step's scalars into that buffer
 * before each replay (geniconet_amd.train.Trainer, ICN_GRAPH=1).  Same arithmetic, bit-identical results. */
int icn_adam_step_dev(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const size_t* numel, const float* scalars_dev, double beta1, double beta2, double eps, double weight_decay,
                      void* stream);

/* Host-side introspection (no device needed).  Each writes at most `cap` elements and returns the element
 * count required (negative on error). */
long icn_table_conv_fwd(int r_in, int stride, int corner_mode, int32_t* out, size_t cap);      /* [7][P_out]     */
long icn_table_conv_bwd(int r_in, int stride, int corner_mode, int32_t* out, size_t cap, int* width); /* [7][E][P_in] */
long icn_table_upsample(int r_in, int corner_mode, int transpose, int32_t* idx, float* coef, size_t cap, int* width);
long icn_table_upsample_pairs(int r_in, int32_t* out, size_t cap);                            /* [2][P_fine]   */
long icn_table_faces(int r, int32_t* out, size_t cap);                                         /* [20*4^r][3]   */
/* The stream-K schedule of the persistent conv GEMM for a launch of `ntiles` tiles of `nk` K-STEPS each on `grid`
 * workgroups, no piece shorter than `ku` steps: rows {workgroup, tile, k0, k1} (step ranges) in each workgroup's walk order.
 * A row with k1 < nk is a piece the workgroup parks for the workgroup that holds the tile's last steps (k0 > 0, k1 == nk).
 * (ABI 4: ranges are K-steps; up to ABI 3 they were whole k-chunks, multiples of ku.) */
long icn_table_stream_k(int ntiles, int grid, int nk, int ku, int32_t* out, size_t cap);
/* The tile lists of a stride-2 data-gradient launch (batch B at input level r_in, tiles of bm rows x ntn column tiles, `grid`
 * workgroups = 256 x blocks per CU): [grid + 1] offsets, then each workgroup's tile ids.  Tiles of such a launch run 1 or 2
 * of the 7 taps; the lists deal every residue class's tiles (tile % 8 = workgroup % 8) so that the workgroups' K-step totals
 * are even. */
long icn_table_tile_lists(int r_in, int stride, int corner_mode, int B, int bm, int ntn, int grid, int32_t* out, size_t cap);
/* Composite table of conv_stride1(upsample(x)) over the COARSE tensor (icn_upconv_*; csrc/icn_geometry.h UpconvTable):
 * meta[7] = {P_fine, P_coarse, n_slots, E, nseg, NV (virtual taps), number of floats};
 * ints = seg[nseg][3] (rows per sample, first list position, virtual-tap mask) | pix[P_fine] (class-major list of fine
 * pixels) | code[NV][P_fine] (by list position: coarse pixel, -1 nothing, -2 - slot) | slot_idx[n_slots][E];
 * floats = alpha[NV][7] (W_eff[v] = sum_t alpha[v][t] W_t) | slot_coef[n_slots][E].  Returns the number of ints. */
long icn_table_upconv(int r_in, int corner_mode, int32_t* ints, size_t cap_ints, float* floats, size_t cap_floats, int* meta);
/* ELL matrix dy (fine) -> g (coarse, 7 taps) of the aggregated backward of the same pair (icn_upconv_bwd): row s * 7 + t
 * lists the fine pixels and coefficients of g_t[s]; [7 * P_coarse][width], -1 padded. */
long icn_table_upconv_bwd(int r_in, int corner_mode, int32_t* idx, float* coef, size_t cap, int* width);

/* Patch form of the forward table for the all-taps weight-gradient kernel (ABI 6; csrc/icn_geometry.h Wg7Table):
 * the output pixels of a sample in patches of 16 consecutive pixels; rows [npatch][U] = the UNION of a patch's gathered source
 * rows as DmaTable codes (pixel, -1 nothing, -2 - k mean of pole k), row U - 1 always -1; pos [npatch][16][8] = byte offset
 * (union row * 256) of tap t's row of pixel k in the staged union ([k][7] padding).  meta[2] = {U, npatch}, U = 64 at stride 1
 * and 112 at stride 2 (the taps of neighbouring outputs share little there); returns the number of row codes, 0 when the table
 * does not exist for this (r_in, stride): P_out % 16 != 0 or a union that does not fit. */
long icn_table_wgrad7(int r_in, int stride, int corner_mode, int32_t* rows, size_t cap_rows, uint16_t* pos, size_t cap_pos, int* meta);

/* Optional diagnostics: HIP-event timing of every MFMA kernel launch between start and stop, on the launch
 * stream.  `stop` synchronises the device and returns the number of entries written (one per kernel that ran).
 * total_flops counts what a launch multiplies: 2*7*Cin*Cout*B*P_out per pass of an ordinary convolution (= its algorithmic
 * FLOPs; the small virtual-row launch of a stride-1 bwd-data adds time, not FLOPs); the composite icn_upconv_* launches execute
 * fewer FLOPs than the two operators they replace (0.68 x forward, 0.25 x backward) and are credited with the executed ones.  The 2 * max_launches events are created once and reused.  Off by default. */
typedef struct icn_profile_entry {
    const char* kernel;
    long launches;
    double total_ms;
    double total_flops;
} icn_profile_entry;
int icn_profile_start(int max_launches);
int icn_profile_stop(icn_profile_entry* out, int cap);
/* Time only the launches of one kernel (a name as icn_profile_stop reports it); NULL or "": all MFMA kernels again.  An event
 * pair around EVERY launch costs the step about 2.5 % (a marker packet between back-to-back kernels, ~2.5 us each). */
int icn_profile_select(const char* kernel);

/* Developer routing flags (same bits as the ICN_DEBUG environment variable, which only sets the initial value):
 * 16 = convolutions on the register-staged fallback kernel, 32 = weight gradients on it, 128 = no stream-K (every tile of
 * the persistent GEMM computed whole by one workgroup), 512 = masked launches walk tiles b, b + G, ... instead of the balanced
 * tile lists, 1024 = the sparse passes of the decoder-block heads on the row-per-thread kernels, 256 = fault injection for the tests of the failure path below (every
 * stream-K finisher reports its partners lost), 2048 / 4096 = every weight gradient on the per-tap kernel / stride-2 weight gradients on
 * k_wgrad7 too, 8192 = slow stream-K partners (tests of the hand-off), 16384 = the 64 x 128 tile's launches on the eight-wave kernels
 * k_conv_dma8 / k_conv_dma_sk8 (round 5: built, bit-identical, not faster in the training step), 32768 = the decoder heads' dense
 * one-tap GEMMs on the class-major kernel k_conv_dma_sk<.., true> instead of the plain-path k_conv_dense_sk and their dense
 * weight gradient on the general k_wgrad_dma instead of k_wgrad_dense, 65536 = single convolutions on the general stream-K kernel instead of
 * k_conv_single_sk (and k_conv_b3_single_sk), 131072 / 262144 (round 6, only meaningful under ICN_ARITH_BF16X3) = weight gradients /
 * the masked stride-2 data gradients stay on the exact fp32 kernels.  Returns the previous flags. */
int icn_set_debug_flags(int flags);

/* Asynchronous failures of the CURRENT device.  Kernels cannot return an error code; the one failure they can detect --
 * a workgroup of the stream-K GEMM whose partner never parked its partial tile within about a second -- turns the tile into
 * NaNs and sets bit 0 (ICN_STATUS_STREAMK_LOST) of a per-device word in pinned host memory.  This call returns the bits set
 * since they were last cleared (>= 0) and clears them when `clear` is non-zero; -1 on error (icn_last_error).  It does not
 * synchronise: call it after the stream (or device) has been synchronised.  The reference has no counterpart (it checks
 * nothing but NaNs through torch.autograd.detect_anomaly, run.py:237); geniconet_amd.train.Trainer calls it once per step in
 * debug mode (ICN_CHECK=1) and bench.py once after the timed region. */
/* Developer instrumentation (tools/trace_conv_blocks.py): while a device buffer of n_u64 >= 8 * grid 64-bit words is
 * registered, every workgroup b of a stream-K GEMM launch (k_conv_dma_sk) writes 8 words at [8 b]: constant-clock (100 MHz)
 * timestamps of entry, first-tile tables built, ring filled, first split-phase segment, exit; the time spent waiting for
 * partners; XCC id; HW_ID.  NULL / 0 switches it off (the default; the kernel then only tests one scalar).  Not thread safe. */
int icn_debug_trace(void* device_buffer, size_t n_u64);

/* Host-only self check: runs every host-side table builder and launch planner of level r (conv stride 1 / 2, upsample,
 * composite upsample + conv tables, loss tables, workspace layouts for the model's channel counts) WITHOUT copying anything
 * to a device, so the host side of the library can be exercised under ASan / UBSan on a machine without a GPU
 * (tools/asan_host.sh).  Returns a positive element count, -1 on error. */
long icn_host_selfcheck(int r, int corner_mode);

#define ICN_STATUS_STREAMK_LOST 1
int icn_device_status(int clear);

/* Arithmetic of the channel-mixing contraction (ABI 7; the reference computes in fp32 on PyTorch, models.py / run.py have no
 * autocast -- SURVEY F1).  ICN_ARITH_F32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).  ICN_ARITH_BF16X3 (the default): every fp32 operand
 * is cut into three bf16 pieces (24 significand bits, exact) and the product is six bf16 MFMAs with fp32 accumulation -- fp32-grade
 * results (1.2 x the exact kernel's rounding error against float64, tests/test_gpu_arith.py) at 2.67 x less matrix-pipe time; used
 * by every MFMA launch of the path the LDS-DMA kernels cover (convolution forward, data gradients incl. the masked stride-2 form,
 * the decoder heads' dense GEMMs, all weight gradients); the scalar / register-staged fall-backs and the small virtual-row GEMM of a
 * stride-1 data gradient stay exact.  The process
 * default is ICN_ARITH_BF16X3; the environment overrides it (ICN_ARITH=f32|bf16x3).  icn_set_arith returns the previous mode, or -1
 * (icn_last_error). */
#define ICN_ARITH_F32 0
#define ICN_ARITH_BF16X3 1
int icn_get_arith(void);
int icn_set_arith(int mode);

/* Build switches of the loaded library (ABI 7): bits 0..15 = ICN_EXP (pricing builds that leave a feature of the convolution kernel
 * out -- THEIR RESULTS ARE WRONG BY DESIGN, only launch times are read; 0 = the product), bits 16..23 = the default wave count of
 * the 64 x 128 tile (ICN_CONV_WAVES_DEFAULT), bits 24..27 = ICN_CHAIN_PRIO.  geniconet_amd._lib refuses a library whose ICN_EXP bits
 * are non-zero unless ICN_ALLOW_EXP=1. */
unsigned icn_build_flags(void);

#ifdef __cplusplus
}
#endif
#endif /* ICN_H */

/*
 * overiva_hip.h -- C ABI of liboveriva_hip.so: the MI355X (gfx950) implementation of the
 * AuxIVA / OverIVA iteration hot path of onolab-tmu/overiva.
 *
 * The reference has no FFI: its boundary for this path is the Python call
 *     overiva(X, n_src, n_iter, proj_back, W0, model, init_eig, return_filters, callback)
 * (reference overiva.py:28-38) and auxiva_pca(X, n_src, **kwargs) (auxiva_pca.py:30).
 * The entry points below are what a ctypes binding placed inside those two functions binds;
 * each one names the reference statements it replaces.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative OIVA_ERR_* otherwise;
 *     oiva_last_error() then returns a thread-local message.
 *   - complex arrays are interleaved float32 (re, im) = numpy complex64, C order.
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device memory
 *     on the plan's device.  The caller owns every buffer it passes in; the plan owns
 *     everything it allocates; nothing returned by pointer outlives oiva_plan_destroy.
 *   - one plan = one device + one stream + one contiguous range of frequency bins.
 *     A plan is not thread-safe; distinct plans are independent.
 *   - no call synchronises with the host except the ones documented to (copies to host,
 *     oiva_plan_sync, the timing helpers).
 */
#ifndef OVERIVA_HIP_H
#define OVERIVA_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define OIVA_OK 0
#define OIVA_ERR_ARG (-1)      /* bad argument / unsupported shape */
#define OIVA_ERR_HIP (-2)      /* a HIP runtime call failed */
#define OIVA_ERR_STATE (-3)    /* call order violated (e.g. iterate before X was set) */
#define OIVA_ERR_NUMERIC (-4)  /* non-finite demixing matrix (numpy raises LinAlgError here) */

#define OIVA_MODEL_LAPLACE 0   /* overiva.py:152-153, :161-163 */
#define OIVA_MODEL_GAUSS 1     /* overiva.py:154-155, :164-167 */

/* Documented limit (the reference has none): at most 16 channels.  Every shape in BASELINE.json and in the
 * reference's overiva_sim_config.json has <= 16; oiva_plan_create rejects more with OIVA_ERR_ARG. */
#define OIVA_MAX_CHANNELS 16

/* Arithmetic of a plan (oiva_plan_set_precision).  Device data (X, Y, the W the streaming kernels read) is always
 * complex64; these bits choose where float64 is used on top of it.
 *   0                      : float32 everywhere (fp32 matrix-core chains of 8 frames folded into float64).
 *   OIVA_PREC_UPDATE_F64   : the per-bin algebra (overiva.py:181-190) in float64, W_hat carried in complex128.
 *   OIVA_PREC_COV_F64      : the weighted covariance (overiva.py:179) as float64 sums of exact float64 products (up to 8
 *                            channels: v_fma_f64 on the vector ALU; 9..16: fp64 matrix cores) -- the reference's own
 *                            arithmetic there: its float64 r_inv promotes that product to complex128 even for
 *                            complex64 input (overiva.py:127-128).
 *   OIVA_PREC_PRECISE      : both; what overiva() selects for complex128 input.
 *   OIVA_PREC_UPDATE_ROWS  : lane layout of the per-bin algebra (one lane per matrix row instead of per element). */
#define OIVA_PREC_FAST 0
#define OIVA_PREC_UPDATE_F64 1
#define OIVA_PREC_UPDATE_ROWS 2
#define OIVA_PREC_COV_F64 4
#define OIVA_PREC_PRECISE (OIVA_PREC_UPDATE_F64 | OIVA_PREC_COV_F64)

typedef struct oiva_plan oiva_plan;

/* library */
int oiva_version(void);
const char *oiva_last_error(void);
int oiva_device_count(int *n);

/*
 * Plan life cycle.  T frames, F bins OWNED BY THIS PLAN, M channels, K sources (1 <= K <= M),
 * F_total = number of bins of the whole problem (== F for a single-GPU run; > F when the bins are
 * sharded over several plans/GPUs -- only the gauss model's 1/F (overiva.py:155) depends on it).
 * stream: a hipStream_t to launch on (e.g. the framework's current stream), or NULL for a
 * plan-owned stream.
 */
int oiva_plan_create(oiva_plan **out, int device, int T, int F, int M, int K, int model, int F_total,
                     void *stream);
int oiva_plan_destroy(oiva_plan *p);

/*
 * Input  X (T, F, M) complex64  -- replaces the copy at overiva.py:132 (the transpose there is
 * not needed: kernels read the native (frames, bins, channels) order).
 * _host: copies rows of F*M complex from a host array whose consecutive frames are
 *        row_pitch_bytes apart (pass F_total*M*8 and a pointer to bin f0 to upload a bin shard
 *        of a larger array; 0 means dense).  Synchronous.
 * _dev : borrows a dense device array (no copy); it must stay valid while the plan uses it.
 */
int oiva_plan_set_x_host(oiva_plan *p, const void *X, long long row_pitch_bytes);
int oiva_plan_set_x_dev(oiva_plan *p, const void *X_dev);
/* _host_c128: as _host for a complex128 array (pitch in bytes of complex128 rows); the conversion to the device's
 * complex64 runs on the GPU, so the host never touches the data (reference overiva.py:132 copies X once too). */
int oiva_plan_set_x_host_c128(oiva_plan *p, const void *X, long long row_pitch_bytes);

/*
 * Prologue, step 1: Cx[f] = (1/T) sum_t x x^H  (overiva.py:87).  Must follow set_x.
 */
int oiva_plan_covariance(oiva_plan *p);
/* Cx as (F, M, M) complex64 (f64 = 0) or complex128 (f64 != 0) on the host (used by the host-side init_eig
 * path, overiva.py:106-109, and by auxiva_pca, auxiva_pca.py:71-75). */
int oiva_plan_get_cx(oiva_plan *p, void *Cx_host, int f64);

/*
 * Prologue, step 2: demixing matrix  (overiva.py:89-123).
 * W0_host: (F, M, K) complex64 (f64 = 0) or complex128 (f64 != 0), or NULL for the identity start
 * (overiva.py:113-114).  Builds W_hat = [W | [J; -I]] with J from the orthogonality constraint
 * (overiva.py:96-98,120-123).
 */
int oiva_plan_set_w(oiva_plan *p, const void *W0_host, int f64);

/*
 * n iterations of the loop body overiva.py:138-190 (demix -> activation r -> scale normalisation
 * -> for every source: weighted covariance V, IP1 row solve, normalisation, J update).
 * Only valid when the plan owns all bins (F == F_total).  Asynchronous.
 */
int oiva_plan_iterate(oiva_plan *p, int n);

/*
 * The same iteration cut at its one cross-bin dependency (overiva.py:152-155: r needs all bins),
 * for bin-sharded multi-GPU runs:
 *   oiva_plan_power   : partial powers of THIS plan's bins, one (T, K) float32 part per batch of 64 bins:
 *                       part[b][t,k] = sum over the batch's bins of |w_k^H x|^2.  They live in the device
 *                       buffer returned by oiva_plan_power_buffer, laid out (parts_per_rank, T, K) with the
 *                       parts this plan does not own left at zero (parts_per_rank = the largest batch
 *                       count of any rank, so that every rank contributes an equally sized message).
 *   (caller all-gathers the buffers of all ranks into parts_dev, (G * parts_per_rank, T, K) float32)
 *   oiva_plan_update  : r from the sum over parts in buffer order (rank, then batch: fixed, so every
 *                       rank gets the same bits), then the per-bin part of the iteration
 *                       (overiva.py:158-173 gamma / 1/r, :161-167 W scaling, :176-190).
 */
int oiva_plan_power(oiva_plan *p);
int oiva_plan_power_buffer(oiva_plan *p, int parts_per_rank, void **parts_dev, long long *bytes);
int oiva_plan_update(oiva_plan *p, const void *parts_dev, int nparts);

/*
 * Epilogue: Y = demix(X, W) as (T, F, K) complex64 (overiva.py:192-195), optionally scaled by
 * projection back onto channel 0 (overiva.py:197-199; also the callback payload of :142-148).
 * row_pitch_bytes as in set_x_host (0 = dense).  Synchronous.  Outputs of more than a few MB leave in slabs of frames:
 * slab k is computed while slab k - 1 crosses PCIe into a pinned ring and slab k - 2 is moved into Y_host by a small pool
 * of copy threads ($OIVA_IO_THREADS, default min(8, cores / 2); $OIVA_DEMIX_IO = legacy | ring | register, see
 * csrc/plan.hip demix_to_host).  A Y_host whose pages are already faulted in is served much faster than a fresh
 * allocation: oiva_host_prefault does that, from the pool's threads, e.g. while the iterations run.
 */
int oiva_plan_demix(oiva_plan *p, void *Y_host, long long row_pitch_bytes, int proj_back);
/* Device buffers of >= 16 MB that a destroyed plan owned (its copy of X, Y, staging) stay in a process-wide pool for the next
 * plan of the same shape -- at most $OIVA_POOL_MB (default 2048; 0: nothing is kept).  This releases them to the driver; the
 * library does so itself, and retries, before a device allocation of its own fails for want of memory. */
int oiva_pool_trim(void);
/* Bytes of output per slab of the hand-over above (0: the default, 8 MB).  Test hook. */
int oiva_plan_set_io_slab(oiva_plan *p, long long bytes);
/* Fault the pages of [ptr, ptr + bytes) in for writing, contents unchanged, using the library's copy threads.  Blocking;
 * callable from any thread (overiva() runs it on the array it will return while the queued iterations run on the GPU; beside the
 * upload of X it slowed the upload by as much as it saved: dropped, round 5). */
int oiva_host_prefault(void *ptr, long long bytes);
/* Same, but Y stays on the device: *Y_dev is the plan's own (T, F, K) complex64 buffer, valid until the next demix of
 * this plan or its destruction.  Hand it to oiva_plan_set_x_dev of another plan to chain two solves without a host
 * round trip (the PCA front-end of auxiva_pca.py:79-87).  Synchronous. */
int oiva_plan_demix_dev(oiva_plan *p, int proj_back, void **Y_dev);
/* As oiva_plan_demix, into a complex128 host array (the dtype overiva() returns for complex128 input); converted on
 * the device. */
int oiva_plan_demix_c128(oiva_plan *p, void *Y_host, long long row_pitch_bytes, int proj_back);

/*
 * Push exchange of the per-rank partial powers between the GPUs of one node -- the all-gather in front of the activation
 * (overiva.py:152-155) written for its 100 KB payload: every rank stores its part straight into every rank's gather
 * buffer (fine-grained device memory shared through hipIpcMemHandle, peer stores over xGMI) and signals a counter; a
 * rank's stream waits on its counter with a command-processor wait.  Same result layout as an all-gather (rank-major
 * parts), so oiva_plan_update() takes the buffer as is.  Epochs count the exchanges, from 1, identically on every rank.
 *   create   : allocate this rank's buffer for parts of part_bytes each
 *   export   : OIVA_XCHG_HANDLE_BYTES bytes to hand to every other rank (any host-side transport)
 *   connect  : map the other ranks' buffers; handles = world * OIVA_XCHG_HANDLE_BYTES bytes in rank order
 *   push     : on `stream`, after whatever produced part_dev: store it into slot `rank` everywhere, then signal
 *   wait     : `stream` waits until all `world` parts of this epoch are in this rank's buffer
 *   gathered : the buffer to read after wait (world parts of the padded part size, rank order)
 *   poll     : host-side check with a time-out (used to validate the transport before relying on it)
 *   force    : host store of the value a stream wait of this epoch waits for -- releases a stream whose wait would
 *              never be satisfied (a validation that failed); the exchange must not be used afterwards
 */
#define OIVA_XCHG_MAX_RANKS 16
#define OIVA_XCHG_HANDLE_BYTES 64
typedef struct oiva_xchg oiva_xchg;
int oiva_xchg_create(oiva_xchg **x, int device, int rank, int world, long long part_bytes);
int oiva_xchg_export(oiva_xchg *x, void *handle);
int oiva_xchg_connect(oiva_xchg *x, const void *handles);
int oiva_xchg_push(oiva_xchg *x, void *stream, const void *part_dev, long long part_bytes, int epoch);
int oiva_xchg_wait(oiva_xchg *x, void *stream, int epoch);
int oiva_xchg_gathered(oiva_xchg *x, int epoch, void **gathered);
int oiva_xchg_poll(oiva_xchg *x, int epoch, int timeout_ms, int *arrived);
int oiva_xchg_force(oiva_xchg *x, int epoch);
int oiva_xchg_destroy(oiva_xchg *x);

/*
 * PCA front-end of auxiva_pca (auxiva_pca.py:71-81): W := the eigenvectors of the K largest eigenvalues of the input
 * covariance, in ascending order of the eigenvalue (numpy.linalg.eigh's w[:, :, -K:]), from a Jacobi eigensolver on
 * the device (float64, one wavefront per bin); the orthogonality constraint fills J as in oiva_plan_set_w.  A following
 * demix gives new_X = X conj(w[:, :, -K:]).  The phase of each eigenvector is unspecified (as LAPACK's is a convention);
 * results after projection back do not depend on it.  evals_host: NULL or (F, M) float64, all eigenvalues ascending.
 */
int oiva_plan_set_w_pca(oiva_plan *p, double *evals_host);
/* init_eig of overiva.py:106-109: W := conj of the K principal eigenvectors of the input covariance, each with the
 * phase numpy.linalg.eig (LAPACK zgeev) gives it -- largest component real and positive.  Same eigensolver. */
int oiva_plan_set_w_eig(oiva_plan *p);
/* W (F, M, K) complex64 (f64 = 0) or complex128 (f64 != 0) -- the view returned at overiva.py:201-202.
 * Synchronous.  Returns OIVA_ERR_NUMERIC if W holds a non-finite value (W is still copied out). */
int oiva_plan_get_w(oiva_plan *p, void *W_host, int f64);

int oiva_plan_sync(oiva_plan *p);
/* A copy of the demixing state W_hat on the device, made / brought back on the plan's stream (asynchronous, ordered with the
 * iterations around it): what a caller of the in-kernel exchanges keeps to fall back on when a rank does not deliver. */
int oiva_plan_save_w(oiva_plan *p);
int oiva_plan_restore_w(oiva_plan *p);

/*
 * Measurement.  Runs n iterations bracketed by HIP events on the plan's stream and, when
 * per_kernel_ms is non-NULL, additionally brackets every kernel launch with events (eager
 * launches) and returns the summed time of each stage in per_kernel_ms[0..3]:
 * [0] demix+power pass, [1] source activation (r, gamma, 1/r), [2] weighted-covariance pass,
 * [3] per-bin update.  total_ms = wall time of the n iterations on the device.  Synchronous.
 */
int oiva_plan_iterate_timed(oiva_plan *p, int n, float *total_ms, float *per_kernel_ms);
#define OIVA_N_STAGES 4

/* Launch geometry of the weighted-covariance pass, for roofline accounting and tuning:
 * get/set the number of frame splits (0 = library default). */
int oiva_plan_get_cov_splits(oiva_plan *p, int *nsplit);
int oiva_plan_set_cov_splits(oiva_plan *p, int nsplit);
/* Which kernel takes the covariance pass of a 9..16-channel plan: 1 (default) = the vector-ALU kernels
 * that form the Hermitian half and write float64 partial sums -- float32 products: four lanes per (bin, frame) for up to 4
 * sources (csrc/kernels_cov_quad.hip; 3 and 4 sources only with the float64 per-bin algebra), 32 lanes per (bin, frame) and
 * every source in one pass for 5..16 (csrc/kernels_cov_half16.hip); OIVA_PREC_COV_F64: the float64 form of the latter for
 * 3..16 sources; a plan with 9, 11, 13 or 15 channels runs them on its own copy of X padded by one zero channel per bin
 * ((M + 1) / M of X more device memory, filled by oiva_plan_covariance) -- 0 = the planar matrix-core kernel, which
 * OIVA_PREC_COV_F64 with one or two sources always uses.  *active (may be NULL) receives whether a vector-ALU kernel is what this plan now launches.  Drops captured
 * graphs. */
int oiva_plan_set_cov_quad(oiva_plan *p, int enable, int *active);
/* same for the demix+power pass (0 = library default) */
int oiva_plan_set_pow_splits(oiva_plan *p, int nsplit);
/* 10..16 channels with 9..16 sources: the covariance pass with the sources on the fp32 matrix cores (default, csrc/kernels_cov_hmfma.hip)
 * or on the vector ALU alone (enable = 0; same bits: A/B tests). */
int oiva_plan_set_cov_hmfma(oiva_plan *p, int enable);
/* 8 channels, 2 sources + background, float32 covariance products, four frame splits (the headline shape): the weighted
 * covariance (overiva.py:179) and the per-bin update of the same bins (:181-190) as ONE launch (csrc/kernels_cov_update.hip; same
 * bits as the two launches; measured no faster, so off by default: $OIVA_COV_UPDATE=1 or enable = 1).  enable 1 / 0; -1 only
 * asks.  *active: whether this plan's iterations run it. */
int oiva_plan_set_fuse_cov_update(oiva_plan *p, int enable, int *active);
/* Replay the iteration from a captured hipGraph instead of eager launches (default off). */
int oiva_plan_use_graph(oiva_plan *p, int enable);
/* Arithmetic: an OR of OIVA_PREC_* (default OIVA_PREC_FAST).  Call it before oiva_plan_covariance so that the
 * prologue runs in the same arithmetic. */
int oiva_plan_set_precision(oiva_plan *p, int flags);

/*
 * X-resident iteration: the loop body of overiva.py:138-190 (demix + power, activation, weighted covariance, IP1
 * solve + normalisation, orthogonal-constraint update of J) fused into ONE persistent launch per oiva_plan_iterate
 * call, with the plan's slice of X held in the chip's registers + LDS for all n iterations (X leaves HBM once per
 * call instead of twice per iteration).  Applies when the slice fits on chip -- 16 bins x <= 256 frames per compute
 * unit: BASELINE configs[1], one rank's 256-bin shard of the headline shape -- with 4 or 8 channels, 1 or 2 sources
 * (K < M) and a float32 covariance pass; see csrc/resident_kernel.inc.
 *   oiva_plan_set_resident   : enable = 1 turns it on (OIVA_ERR_ARG when the shape does not qualify), 0 off.
 *                              With it on, oiva_plan_iterate is synchronous; a launch whose workgroups could not all
 *                              become resident gives up after the time-out WITHOUT having changed W, the plan falls
 *                              back to the four-launch path for that call and all later ones, and the reason is kept
 *                              (info[8]).
 *   oiva_plan_resident_info  : info[0] shape qualifies, [1] enabled, [2] bin groups, [3] frame splits, [4] frames per
 *                              split, [5] frames per lane, [6] of them in registers, [7] LDS bytes per workgroup,
 *                              [8] give-up code of the last failed launch (0: none), [9] resident launches so far,
 *                              [10] fall-backs so far, [11] the frames of X each compute unit holds, in bytes
 *   oiva_plan_resident_phases: average duration in microseconds, over the iterations of the last resident launch
 *                              (at most 256), of the phases of workgroup 0, from the 100 MHz clock read inside the
 *                              kernel: [0] demix + power, [1] the column's parts (wait + ordered sum), [2] activation,
 *                              [3] weighted covariance: accumulation, [4] its reduction over the frame phases + publication,
 *                              [5] wait for the row's partials, [6] per-bin update chain, [7] wait for the row's
 *                              demixing vectors; *n_iter = iterations covered (0: nothing recorded)
 *   oiva_plan_resident_debug : test hooks -- time-out of a wait in milliseconds (0: default 250, 2000 when sharded over processes), and the index of a
 *                              workgroup that never publishes (-1: none), which makes the launch give up;
 *                              _debug_from: the workgroup stops publishing in that iteration of a launch (0-based)
 *   oiva_plan_resident_loopback: world >= 2: ONE GPU runs the multi-GPU exchange of the kernel against itself -- per
 *                              frame split a leader workgroup gathers the rank's parts and stores the sums of all
 *                              `world` slots (its own, exact zeros for the others) into a gather buffer of the same kind
 *                              of memory the ranks of a sharded run map from each other, every workgroup adds the
 *                              slots in rank order: the leader hop and the world-slot sum of bench.py --gpus N on one
 *                              GPU (everything but the flight over xGMI), with the result of world = 1.  0 / 1: off.
 *                              Call it while resident is off; the plan must own all bins.
 */
struct oiva_xchg;
#define OIVA_RESIDENT_INFO 12
#define OIVA_RESIDENT_PHASES 8
int oiva_plan_set_resident(oiva_plan *p, int enable);
int oiva_plan_resident_info(oiva_plan *p, int *info /* OIVA_RESIDENT_INFO ints */);
int oiva_plan_resident_phases(oiva_plan *p, double *phase_us /* OIVA_RESIDENT_PHASES */, int *n_iter);
int oiva_plan_resident_debug(oiva_plan *p, int timeout_ms, int stall_block);
int oiva_plan_resident_debug_from(oiva_plan *p, int timeout_ms, int stall_block, int first_stalled_iteration);
int oiva_plan_resident_loopback(oiva_plan *p, int world);
/* Diagnostics: with enable != 0 the next resident launches (of at most 64 iterations) record the timestamps of EVERY
 * workgroup; stamps_host (NULL: just switch) receives [n_wg][n_iter][OIVA_RESIDENT_STAMPS] 100 MHz ticks of the last one. */
#define OIVA_RESIDENT_STAMPS 16
int oiva_plan_resident_trace(oiva_plan *p, int enable, unsigned long long *stamps_host, int *n_wg, int *n_iter);
/* Bins sharded over the GPUs of a node: give the plan of a shard a connected oiva_xchg (below) whose slot is
 * frame_splits * frames_per_split * K * 4 bytes; the resident kernel then exchanges the ranks' partial source powers
 * (overiva.py:152-155) itself -- one workgroup per frame split stores the rank's sums into every rank's buffer (peer
 * stores over xGMI), every workgroup adds the ranks' sums in rank order -- and oiva_plan_iterate works on a shard
 * (F < F_total).  Every rank must call oiva_plan_iterate with the same counts.  A launch that gives up returns
 * OIVA_ERR_STATE (nothing written).  x = NULL disconnects. */
int oiva_plan_resident_connect(oiva_plan *p, struct oiva_xchg *x);
/*
 * The same exchange for shards that do NOT fit on chip (2 and 4 GPUs at the headline shape), inside the four-launch
 * iteration: give the plan a connected oiva_xchg whose slot is T * K * 8 bytes; the ACTIVATION kernel then adds its rank's
 * 64-bin parts, stores the sum of every (frame, source) -- one 8-byte {value, epoch} word -- into its slot of every other
 * rank's buffer (peer stores over xGMI), polls its own buffer for the other ranks' words and adds the ranks' sums in rank
 * order (overiva.py:152-155; the same bits of r on every rank).  No collective, no stream wait, no host in the loop:
 * oiva_plan_iterate works on a shard (F < F_total) and replays captured graphs of four kernels per iteration.  Every rank
 * must run the same number of iterations.  Connecting zeroes the rank's own gather buffer and restarts the epochs: the ranks
 * must rendezvous (any barrier of the host side) between their connects and their first iteration.  A wait that gives
 * up (default 2 s) is reported by the next oiva_plan_sync / oiva_plan_get_w / oiva_plan_demix* (OIVA_ERR_STATE); the demixing
 * matrices of the iterations since are undefined: oiva_plan_save_w before, oiva_plan_restore_w after.  x = NULL disconnects
 * and clears the condition.
 *   oiva_plan_fused_loopback: world >= 2: ONE GPU runs that exchange against itself (own buffer, the other ranks' sums are
 *                             zeros: the result is the single-GPU one up to the association of the sum); 0 / 1: off.
 *   oiva_plan_fused_debug   : test hooks -- time-out of a wait in milliseconds (0: default); stall != 0: in loop-back the other
 *                             ranks' words are never stored (a rank that does not deliver)
 */
int oiva_plan_fused_connect(oiva_plan *p, struct oiva_xchg *x);
int oiva_plan_fused_loopback(oiva_plan *p, int world);
int oiva_plan_fused_debug(oiva_plan *p, int timeout_ms, int stall);
/* Number of frame splits of the resident grid (0: the library's choice).  The ranks of a sharded run must use one
 * geometry: they agree on the smallest count any of them chose (uneven shards).  Call it while resident is off. */
int oiva_plan_set_resident_splits(oiva_plan *p, int nsplit);

/*
 * Test-only stage access (per-kernel parity tests call these through the same ABI).
 * Stages run on the plan's current state; getters synchronise.
 */
int oiva_test_set_rinv(oiva_plan *p, const float *rinv_host /* (T,K) */);
int oiva_test_get_rinv(oiva_plan *p, float *rinv_host /* (T,K) */, float *wscale_host /* (K) */);
int oiva_test_run_weighted_cov(oiva_plan *p);                 /* overiva.py:179 for all K sources */
int oiva_test_get_v(oiva_plan *p, void *V_host /* (K,F,M,M) complex64 | complex128 */, int f64);
int oiva_test_run_update(oiva_plan *p);                       /* overiva.py:181-190 for s = 0..K-1 */
int oiva_test_get_what(oiva_plan *p, void *What_host /* (F,M,M) complex64 | complex128 */, int f64);
int oiva_test_set_what(oiva_plan *p, const void *What_host, int f64);
int oiva_test_run_power(oiva_plan *p, float *p_host /* (T,K) summed over this plan's bins */);
/* average duration of `reps` back-to-back launches of one stage (0 power, 1 activation, 2 covariance,
 * 3 update) on the plan's current state, HIP events on the plan's stream */
int oiva_test_time_stage(oiva_plan *p, int stage, int reps, float *avg_ms);

/*
 * OGIVE -- orthogonally constrained independent vector extraction of ONE source by gradient steps, the reference's
 * ive.py::ogive(X, n_iter, step_size, tol, update, proj_back, W0, model, init_eig, return_filters, callback)
 * (ive.py:33-256; called at overiva_sim.py:313-315).  Runs on a plan created with K = 1 after oiva_plan_covariance
 * and oiva_plan_set_w (w = column 0: identity start ive.py:129-130, or W0 / the principal eigenvector from the host):
 *   oiva_plan_ogive_begin   : Cx^-1, ||Cx||, a from w, step selection                       ive.py:100-102,136-139,173-180
 *   oiva_plan_ogive_iterate : up to n epochs of ive.py:190-246 starting at epoch index first_epoch (the switching
 *                             criterion runs when the index is a multiple of 10).  The stopping rule max ||delta|| < tol
 *                             is evaluated on the device after every epoch; once met the state is frozen and the
 *                             remaining epochs of the call are no-ops.  Synchronous: returns the number of epochs that
 *                             changed the state and whether the rule was met.
 * The result is read with oiva_plan_demix / oiva_plan_get_w as for the other algorithms.
 */
#define OIVA_OGIVE_DEMIX 0
#define OIVA_OGIVE_MIX 1
#define OIVA_OGIVE_SWITCHING 2
int oiva_plan_ogive_begin(oiva_plan *p, int update_mode, int model);
int oiva_plan_ogive_iterate(oiva_plan *p, int first_epoch, int n, double step_size, double tol, int *epochs_run,
                            int *converged, double *max_delta);

/*
 * STFT analysis / synthesis on the GPU (hipFFT): time-domain audio in and out next to the solver.
 * Replaces, in the reference's drivers, pra.transform.analysis(mics_signals.T, framesize, framesize // 2, win=win_a)
 * (overiva_oneshot.py:293-295, overiva_sim.py:206-207) and pra.transform.synthesis(Y, framesize, framesize // 2,
 * win=win_s) (overiva_oneshot.py:371-379).  Those are third-party pyroomacoustics calls whose source is absent from
 * the reference: parity unpinned; the block-processing convention restated here: every frame holds `hop` new
 * samples behind `frame - hop` old ones, zeros before the first sample, n_frames = n_samples / hop.
 *   x (n_samples, n_chan) float32, C order  ->  X (n_frames, frame/2 + 1, n_chan) complex64
 *   Y (n_frames, frame/2 + 1, k) complex64  ->  y (n_frames * hop, k) float32      (k <= n_chan)
 * win_a / win_s: analysis / synthesis windows of `frame` floats, or NULL for a rectangular window.
 * oiva_stft_analysis copies X to X_host when that is not NULL and returns, through X_dev when that is not NULL, the
 * device array holding it (valid until the next analysis/synthesis on this handle): hand it to
 * oiva_plan_set_x_dev to run the solver without a host round trip.  All calls are synchronous.
 */
typedef struct oiva_stft oiva_stft;
int oiva_stft_create(oiva_stft **out, int device, int n_samples, int n_chan, int frame, int hop, const float *win_a,
                     const float *win_s);
int oiva_stft_destroy(oiva_stft *p);
int oiva_stft_shape(oiva_stft *p, int *n_frames, int *n_freq);
int oiva_stft_analysis(oiva_stft *p, const float *x_host, void *X_host, void **X_dev);
int oiva_stft_synthesis(oiva_stft *p, const void *Y_host, int n_chan, float *y_host);

#ifdef __cplusplus
}
#endif
#endif /* OVERIVA_HIP_H */

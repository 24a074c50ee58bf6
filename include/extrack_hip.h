/* extrack_hip.h - C ABI of libextrack_hip.so: the MI355X (gfx950) implementation of ExTrack's
 * track-likelihood hot path.
 *
 * The reference (vanTeeffelenLab/ExTrack v1.6.3, pure Python/numpy) has no FFI seam for this path;
 * the functions below are what a binding for it replaces (file:line relative to the reference root):
 *
 *   extrack_upload_bucket / extrack_attach_bucket
 *       one length bucket of the {str(len): ndarray[N, len, D]} dict, as produced by
 *       extrack/readers.py:82-97 and consumed at extrack/tracking.py:1346-1367 (param_fitting) and
 *       :822-835 (predict_Bs); optional per-peak localisation errors = input_LocErr (tracking.py:1354).
 *   extrack_loglik / extrack_loglik_async
 *       sum over all buckets of Proba_Cs (extrack/tracking_0.py:440-458 -> P_Cs_inter_bound_stats,
 *       extrack/tracking.py:109-318), i.e. the body of cum_Proba_Cs after extract_params
 *       (extrack/tracking_0.py:654-700 / tracking.py:1009-1069): returns +sum(LL); the caller negates.
 *   extrack_predict
 *       P_Cs_inter_bound_stats(..., do_preds=1)[2] for one bucket (Pool_star_P_inter,
 *       extrack/tracking_0.py:460-461, driven by predict_Bs :463-563).
 *   extrack_loglik_th
 *       the same sum for the THRESHOLD-FUSION kernel that extrack.tracking calls in v1.6.3: Proba_Cs
 *       (extrack/tracking.py:769-787) -> P_Cs_inter_bound_stats_th (:427-650) + fuse_tracks_th (:652-743),
 *       evaluated in chunks as cum_Proba_Cs does (tracking.py:1043-1069, 2000 tracks per chunk).
 *   extrack_predict_th
 *       P_Cs_inter_bound_stats_th(..., do_preds=1)[2] for one bucket in chunks of nb_max tracks (predict_Bs,
 *       extrack/tracking.py:792-906).
 *   extrack_th_plan_step
 *       the merge groups fuse_tracks_th decided for one chunk and step (tracking.py:681-701) - diagnostic.
 *   extrack_p_stay_table
 *       the field-of-view survival table, extrack/tracking.py:182-191.
 *   extrack_segment_len_hist
 *       P_segment_len(...)[2] (extrack/histograms.py:26-286) summed over one bucket: the state-duration histogram that len_hist
 *       (histograms.py:294-373) accumulates over chunks of 50 tracks.
 *   extrack_refine_positions
 *       get_pos_PDF + the weighted read-out of position_refinement for one bucket (extrack/refined_localization.py:207-338).
 *   extrack_refine_pos_pdf
 *       get_pos_PDF's own return values (extrack/refined_localization.py:207-298): means / stds / log-weights of every component of the
 *       Gaussian mixture of every position, for inspection of small buckets.
 *   extrack_loglik_grad
 *       extrack_loglik AND its exact gradient in one pass.  It replaces the finite-difference loop that the reference's
 *       optimiser runs around cum_Proba_Cs (lmfit.minimize at extrack/tracking.py:1371: BFGS evaluates the objective
 *       nvar + 1 times per iteration to difference it numerically).
 *   extrack_loglik_th_grad
 *       extrack_loglik_th AND the exact gradient of that value at the evaluation's own merge plan - the objective v1.6.3's
 *       param_fitting hands to lmfit.minimize (extrack/tracking.py:1371 -> cum_Proba_Cs :991 -> P_Cs_inter_bound_stats_th :427-650);
 *       the reference differences it numerically although the grouping decisions of fuse_tracks_th (:676-701) make it only
 *       piecewise smooth.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns 0 on success or a
 * negative EXTRACK_E_* code; extrack_last_error() gives the message.  The caller owns every host
 * buffer it passes; the library owns device copies made by extrack_upload_bucket.  A context is
 * driven by one host thread at a time.  All arithmetic is IEEE fp64.
 */
#ifndef EXTRACK_HIP_H
#define EXTRACK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EXTRACK_ABI_VERSION 6

#define EXTRACK_OK 0
#define EXTRACK_E_INVALID (-1)     /* bad argument / unsupported configuration */
#define EXTRACK_E_HIP (-2)         /* HIP runtime error (message has the HIP error string) */
#define EXTRACK_E_NODEVICE (-3)    /* no usable gfx950 device */
#define EXTRACK_E_UNSUPPORTED (-4) /* valid request outside the built kernel set (e.g. S^F too large for LDS) */

typedef struct extrack_ctx extrack_ctx;

/* Model parameters after extract_params (extrack/tracking.py:913-986) plus the dataset-global scalars
 * that cum_Proba_Cs derives from the bucket list (tracking.py:1009-1010). */
typedef struct extrack_model {
    int32_t n_states;    /* S, 2..8 */
    int32_t nb_substeps; /* ns, 1..4 */
    int32_t frame_len;   /* F > ns, window of exactly enumerated states; S^F <= 2^20 sequences per track.  Up to 1024 groups
                            (S^(F-ns)) and 160 KB per track the state lives in LDS / registers; beyond that - 5 states at the reference's default
                            frame_len 6, 6 states at frame_len 5 - 6, 2 states at frame_len 11 - 15, posteriors with 7 / 8 states - in global
                            memory, one lane per track (csrc/xt_big.h: likelihood and posteriors; the gradient entry points answer
                            EXTRACK_E_UNSUPPORTED there and the caller differences the objective) */
    int32_t min_len;     /* smallest track length of the WHOLE dataset (all shards); the stay-in-FOV term starts at step max(min_len, 2) */
    int32_t max_len;     /* largest track length of the WHOLE dataset: buckets of this length get isBL = 0 */
    int32_t locerr_mode; /* 0: global locerr[]; 1: per-peak sigma uploaded with the bucket;
                            2: per-peak, sigma' = clip(sigma*slope + offset, 1e-6, inf) (tracking.py:928-930) */
    int32_t locerr_dims; /* mode 0: 1 (one value for all dims) or D (one per dim) */
    int32_t n_p_stay;    /* 0 or 1: p_stay is ONE table; n > 1: p_stay holds n tables, one per chunk (per-track time steps, see
                            extrack_set_bucket_dt) */
    double locerr[3];    /* mode 0: localisation error (std) */
    double slope, offset;
    double pBL;          /* bleaching probability per step */
    const double* ds;    /* [S] diffusion lengths sqrt(2 D dt), non-decreasing */
    const double* Fs;    /* [S] initial fractions, > 0 */
    const double* TrMat; /* [S*S] row-major per-substep transition probabilities P(i->j), > 0 */
    const double* p_stay;/* [S^ns] probability of staying in the field of view; index r has digit c
                            (r / S^c) % S with c = 0 the newest sub-state (extrack_p_stay_table) */
} extrack_model;

int extrack_abi_version(void);

/* Creates a context on HIP device `device_id` (must be gfx950).  Owns one non-blocking stream. */
int extrack_create(int device_id, extrack_ctx** out);
void extrack_destroy(extrack_ctx* ctx);
/* Message for the last error on ctx (or for a failed extrack_create when ctx == NULL). */
const char* extrack_last_error(const extrack_ctx* ctx);

/* Use an external HIP stream (hipStream_t) for all subsequent work, or NULL to go back to the
 * context's own stream. */
int extrack_set_stream(extrack_ctx* ctx, void* hip_stream);

/* Copies one length bucket to the device.  tracks: host [n][len][dims] C-order fp64.
 * sigma: NULL or host [n][len][sigma_dims] per-peak localisation errors, sigma_dims in {1, dims}.
 * Buckets may be uploaded in any order; ids are dense from 0. */
int extrack_upload_bucket(extrack_ctx* ctx, const double* tracks, int64_t n, int32_t len, int32_t dims,
                          const double* sigma, int32_t sigma_dims, int32_t* bucket_id_out);
/* Same, for buffers that already live on this device (e.g. torch tensors): no copy, caller keeps
 * them alive until extrack_clear_buckets / extrack_destroy. */
int extrack_attach_bucket(extrack_ctx* ctx, const double* d_tracks, int64_t n, int32_t len, int32_t dims,
                          const double* d_sigma, int32_t sigma_dims, int32_t* bucket_id_out);
/* Per-track time steps of one bucket (extrack/tracking.py:979-982: dt given as {len: array[n][len]}): dt host [n][len], copied
 * to the device; NULL removes them.  Only the threshold-fusion entry points use them (the reference's fixed-window kernel has no
 * such input).  With time steps set, model->ds must be the diffusion lengths for a UNIT time step (sqrt(2 D)): the diffusion
 * term added at step t is scaled by dt[track][len - t] (tracking.py:494-499, 548-551 - the reference reads its UNREVERSED array
 * at column len - current_step), and model->p_stay must hold one table per chunk, buckets in id order, chunks in row order: the
 * field-of-view table of a chunk comes from the median over the chunk's tracks of sqrt(2 D dt[track][0]) (tracking.py:507-511). */
int extrack_set_bucket_dt(extrack_ctx* ctx, int32_t bucket_id, const double* dt);
int extrack_clear_buckets(extrack_ctx* ctx);
int extrack_bucket_count(const extrack_ctx* ctx);

/* One log-likelihood evaluation over every bucket of the context.
 * total_ll (host, required): sum of per-track log-likelihoods (NOT negated).
 * per_track (host, may be NULL): per-track log-likelihoods, buckets concatenated in id order. */
int extrack_loglik(extrack_ctx* ctx, const extrack_model* model, double* total_ll, double* per_track);
/* Enqueues the evaluation on the context's stream and leaves the scalar in device memory
 * (d_total_ll, 8 bytes, device pointer) without synchronising - for a following RCCL all-reduce. */
int extrack_loglik_async(extrack_ctx* ctx, const extrack_model* model, double* d_total_ll);

/* Per-sequence log-probabilities of one bucket, the first return value of P_Cs_inter_bound_stats (extrack/tracking.py:109-318, returned
 * at :318): lp host [n][n_cols], column i = the sequence of states whose digit c is (i / n_states^c) % n_states, c = 0 the newest state
 * (get_all_Bs, tracking.py:746-757); n_cols = extrack_sequence_columns(n_states, len, nb_substeps, frame_len, isBL) with isBL =
 * (len != model->max_len).  The likelihood path never forms this matrix (it is reduced in the kernel): this entry point exists for
 * callers of the reference function that read it, on small inputs (n * n_states^(frame_len + nb_substeps) doubles cross the host). */
int64_t extrack_sequence_columns(int32_t n_states, int32_t len, int32_t nb_substeps, int32_t frame_len, int32_t isBL);
int extrack_sequence_matrix(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, double* lp, int64_t n_cols);

/* State posteriors of one bucket: preds host [n][len][S].  model->nb_substeps must be 1
 * (predict_Bs forces it, extrack/tracking.py:839). */
int extrack_predict(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, double* preds);

/* Tangent of a model along one direction theta: d(field)/d(theta) for every differentiable field of extrack_model.
 * The diffusion lengths enter as the derivative of their SQUARES (ds^2 = 2 D dt is what the recursion uses, and it keeps the
 * derivative finite at D = 0).  p_stay is a function of ds and cell_dims computed by the caller (extrack_p_stay_table), so its
 * tangent comes from the caller too. */
typedef struct extrack_model_tangent {
    double locerr[3];     /* d locerr (std), mode 0 */
    double slope, offset; /* mode 2 */
    double pBL;
    const double* ds2;    /* [S]    d (ds^2) */
    const double* Fs;     /* [S] */
    const double* TrMat;  /* [S*S] */
    const double* p_stay; /* [S^ns] */
} extrack_model_tangent;

/* One evaluation of the fixed-window log-likelihood (as extrack_loglik) together with its derivative along n_dir model
 * directions, by forward-mode differentiation inside the recursion (window fusion included: it is the exact gradient of the
 * value extrack_loglik returns, not an approximation).  total_ll: sum of per-track log-likelihoods; grad[i] = d total_ll /
 * d theta_i (host, n_dir entries).  n_dir may be 0 (then it is extrack_loglik).  Two-state models with one substep run with the
 * sequence state and its tangents in registers (csrc/xt_reg2.h), <= 8 directions per pass; 3 / 4 states by reverse mode (csrc/xt_rev.h);
 * other models with the tangents in registers + LDS exchange (csrc/xt_gradr.h) or in LDS (csrc/xt_grad.h).  Restriction: all uploaded
 * buckets must share the track dimensionality and the per-peak error layout (ONE launch group; a real dataset does) - otherwise
 * EXTRACK_E_UNSUPPORTED (extrack_loglik itself serves such mixed sets, group by group). */
int extrack_loglik_grad(extrack_ctx* ctx, const extrack_model* model, int32_t n_dir, const extrack_model_tangent* tangents,
                        double* total_ll, double* grad);
/* The same evaluation enqueued on the context's stream (extrack_set_stream) without waiting for it: d_out (DEVICE, 1 + n_dir
 * doubles) receives {sum LL, d sum LL / d theta_i} in stream order - the multi-GPU objective all-reduces that buffer on the same
 * stream (RCCL), like extrack_loglik_async does for the scalar.  The tangent arrays are consumed before the call returns. */
int extrack_loglik_grad_async(extrack_ctx* ctx, const extrack_model* model, int32_t n_dir, const extrack_model_tangent* tangents,
                              double* d_out);
/* Threshold-fusion log-likelihood (as extrack_loglik_th: same threshold / max_nb_states / chunk semantics, same value up to rounding)
 * together with its derivative along n_dir model directions AT THE FROZEN PLAN of this evaluation: the plan kernel decides the merge
 * groups of every chunk from its pilot tracks at `model`, then one forward and one backward sweep over every track (reverse mode, csrc/
 * xt_thgrad.h) return sum LL and the adjoint of every model table, contracted with the directions' tangents - the cost does not
 * depend on n_dir.  With the groups held fixed the value is a smooth function of the model and grad is its exact derivative; where a
 * parameter change flips a grouping decision the objective itself jumps (by ~1e-9 relative) and no derivative exists - finite
 * differences of the reference's optimiser sample those jumps, this gradient does not.  n_dir may be 0.  Not built: per-track time
 * steps (extrack_set_bucket_dt) and models whose n_states^(nb_substeps + 1) table adjoints exceed the LDS (EXTRACK_E_UNSUPPORTED). */
int extrack_loglik_th_grad(extrack_ctx* ctx, const extrack_model* model, double threshold, int32_t max_nb_states, int32_t chunk,
                           int32_t n_dir, const extrack_model_tangent* tangents, double* total_ll, double* grad);
/* The same, enqueued on the context's stream: d_out (DEVICE, 1 + n_dir doubles) receives {sum LL, gradient} in stream order (the plan
 * kernel's sequence counts are still read back once inside the call, as in extrack_loglik_th_async). */
int extrack_loglik_th_grad_async(extrack_ctx* ctx, const extrack_model* model, double threshold, int32_t max_nb_states, int32_t chunk,
                                 int32_t n_dir, const extrack_model_tangent* tangents, double* d_out);
/* Frozen plan: with on = 1 the threshold-fusion evaluations of this context (extrack_loglik_th[_async], extrack_loglik_th_grad[_async])
 * skip the plan kernel and follow the merge plan the LAST planning evaluation left with the buckets (same chunk size required;
 * EXTRACK_E_INVALID if there is none).  The value is then a smooth function of the model (the objective fuse_tracks_th would give if its
 * grouping decisions, /root/reference/extrack/tracking.py:676-701, did not react to the parameters) - what an optimiser should be
 * handed between two re-plannings (extrack_amd.tracking.param_fitting does: plan, minimise at that plan with the exact gradient,
 * re-plan, until the plan no longer changes).  on = 0: every evaluation decides its own plan again (the reference's semantics). */
int extrack_th_freeze_plan(extrack_ctx* ctx, int32_t on);
/* Per-sequence log-probabilities of the threshold-fusion kernel for ONE bucket taken as ONE chunk - the first return value of
 * P_Cs_inter_bound_stats_th (extrack/tracking.py:427, returned at :650) before Proba_Cs' log-sum (:780-787): lp host [n][n_cols], n_cols =
 * (state sequences alive after the last merge) x n_states^nb_substeps, column g * n_states^nb_substeps + r as in the reference; for isBL
 * buckets (len != model->max_len) WITHOUT the leaving / bleaching term, a further expansion by n_states^nb_substeps whose factors depend
 * on the model only (tracking.py:611-630; extrack_amd.tracking.P_Cs_inter_bound_stats_th adds it).  Call with lp == NULL first: *n_cols_out
 * receives n_cols (it depends on the merges, i.e. on the data).  For callers of the reference function that read the matrix, on small
 * inputs (n * n_cols doubles cross the host); the likelihood path never forms it. */
int extrack_sequence_matrix_th(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, double threshold, int32_t max_nb_states,
                               double* lp, int64_t n_cols_cap, int64_t* n_cols_out);
/* Device time (ms) of the gradient kernels of the last extrack_loglik_grad / extrack_loglik_th_grad (or _async) call (waits for them). */
int extrack_last_grad_ms(extrack_ctx* ctx, float* ms);

/* State-duration histogram of one bucket (extrack/histograms.py:26-286 P_segment_len, third return value, summed over the bucket's
 * tracks): hist host [(len - 1)][S], hist[k - 1][s] = expected number of segments of exactly k consecutive positions in state s
 * (segments as long as the whole track are not counted, histograms.py:279).  Every state sequence is followed with its full
 * history; after each position at most max_nb_states sequences survive, ranked by their probability including the next position's
 * predictive density (histograms.py:185-203; len_hist's default is 500).  model->nb_substeps must be 1; model->min_len is the
 * reference's min_l (smallest track length of the dataset), isBL = (len != model->max_len) as elsewhere.
 * Limits: len * bits_per_state <= 4096 (bits = 1 / 2 / 3 for <= 2 / 4 / 8 states), max_nb_states * n_states <= 16384, and the staged track +
 * candidate arrays + histogram must fit the 160 KiB LDS of a CU (refused with EXTRACK_E_UNSUPPORTED otherwise). */
int extrack_segment_len_hist(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, int32_t max_nb_states, double* hist);

/* Refined positions of one bucket (extrack/refined_localization.py:304-338 position_refinement -> get_pos_PDF :207 -> get_LC_Km_Ks :48):
 * mu host [n][len][dims], sigma host [n][len].  Two passes of the threshold-fusion recursion (merge decisions from the first 30
 * tracks of the bucket, fuse_tracks_th with per-track histories) - over the time-reversed tracks with model->TrMat and over the tracks
 * as they are with its transpose - record every position's surviving state sequences (mean, std, log-weight, newest state); for each
 * position the sequences of both passes that agree on its state are paired and the three Gaussians (prediction from the future,
 * localisation, prediction from the past) multiplied; mu / sigma are the probability-weighted mean / root mean variance of the pairs.
 * model: n_states, ds, TrMat, Fs (position 0 only, through a quirk of the reference), locerr[0] (one global error: locerr_mode 0,
 * locerr_dims 1), frame_len; nb_substeps must be 1; p_stay / pBL / min_len / max_len are not used.  Tracks need >= 3 positions. */
int extrack_refine_positions(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, double threshold, int32_t max_nb_states,
                             double* mu, double* sigma);

/* The mixture extrack_refine_positions reads out, as get_pos_PDF returns it (extrack/refined_localization.py:207-298: all_pos_means,
 * all_pos_stds, all_pos_weights).  counts host [len]: components of every position, in the reference's order (end positions: the
 * sequences of the pass's last record; between: for every state, sequences from the future x sequences from the past that agree on it).
 * With cum[k] = counts[0] + ... + counts[k - 1], component j of position k of track x is means[((cum[k] + j) * n + x) * dims + d],
 * stds / logw[(cum[k] + j) * n + x] (host; `capacity` = rows each array holds >= the sum of the counts).  means == NULL: counts only
 * (the sizing call).  Same model fields and limits as extrack_refine_positions; the bucket's records must fit one row block
 * (EXTRACK_REFINE_BUDGET_MB), EXTRACK_E_UNSUPPORTED otherwise - the components of a large bucket are not meant to leave the GPU. */
int extrack_refine_pos_pdf(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, double threshold, int32_t max_nb_states,
                           int32_t* counts, int64_t capacity, double* means, double* stds, double* logw);

/* Threshold-fusion log-likelihood (the kernel extrack.tracking.param_fitting / cum_Proba_Cs call in v1.6.3,
 * extrack/tracking.py:427-743).  Which state sequences are merged at a step is decided from the first 30 tracks
 * of every chunk of `chunk` consecutive tracks of a bucket (tracking.py:678-679; cum_Proba_Cs uses chunk = 2000,
 * tracking.py:1043) and applied to the whole chunk, so the value depends on `chunk` and on the track order.
 * model->frame_len is the number of most recent states whose equality forces a merge; threshold and
 * max_nb_states as in tracking.py:427 (threshold is multiplied by 1.2 at every step with more than
 * max_nb_states live sequences).  Time steps: fixed (in model->ds) or per track (extrack_set_bucket_dt).
 * total_ll / per_track as in extrack_loglik.  Limit: 32 768 expanded sequences (live sequences x n_states^nb_substeps) at any step
 * (EXTRACK_E_UNSUPPORTED beyond; 8192 for extrack_predict_th / extrack_refine_positions). */
int extrack_loglik_th(extrack_ctx* ctx, const extrack_model* model, double threshold, int32_t max_nb_states, int32_t chunk,
                      double* total_ll, double* per_track);
/* Same evaluation, the scalar left in device memory (d_total_ll, 8 bytes) for a following RCCL all-reduce on the context's
 * stream; nothing is copied back to the host at the end (counterpart of extrack_loglik_async for the threshold-fusion kernel). */
int extrack_loglik_th_async(extrack_ctx* ctx, const extrack_model* model, double threshold, int32_t max_nb_states, int32_t chunk,
                            double* d_total_ll);
/* Threshold-fusion state posteriors of one bucket (P_Cs_inter_bound_stats_th(..., do_preds=1), extrack/tracking.py:427-650,
 * driven as predict_Bs drives it, tracking.py:792-906): preds host [n][len][S].  The bucket is cut into chunks of nb_max
 * consecutive tracks (predict_Bs default: 1, i.e. every track decides its own merges); with nb_max > 30 the first 30 tracks of a
 * chunk decide the merges (tracking.py:676-691) and the others follow them with their own merge weights (tracking.py:703-741).
 * model->nb_substeps must be 1 (tracking.py:839). */
int extrack_predict_th(extrack_ctx* ctx, const extrack_model* model, int32_t bucket_id, double threshold, int32_t max_nb_states,
                       int32_t nb_max, double* preds);

/* Diagnostic: merge groups of the last extrack_loglik_th call for chunk `chunk_index` of a bucket at step t
 * (t = 1 .. len-1; merges happen for t < len-1).  n_expanded = sequences before the merge, n_groups = after
 * (0 when the step has no merge).  members[0..n_expanded) are the expanded sequence indices sorted by group,
 * gstart[0..n_groups] the group boundaries; both may be NULL; cap = capacity of both arrays in elements. */
int extrack_th_plan_step(extrack_ctx* ctx, int32_t bucket_id, int64_t chunk_index, int32_t t, int32_t* n_expanded,
                         int32_t* n_groups, uint16_t* members, uint16_t* gstart, int32_t cap);

/* Device time (ms, HIP events on the context's stream) spent in the track kernels of the last
 * extrack_loglik / extrack_loglik_async / extrack_predict call (valid after the stream is idle). */
int extrack_last_kernel_ms(extrack_ctx* ctx, float* ms);
/* Launch geometry of the last track kernel: info[0] blocks, [1] threads/block, [2] LDS bytes/block,
 * [3] tracks/block, [4] occupancy (blocks/CU), [5] compute units. */
int extrack_last_launch_info(const extrack_ctx* ctx, int32_t info[6]);

/* Host helper: p_stay table (extrack/tracking.py:182-191).  out has S^ns entries. */
int extrack_p_stay_table(const double* ds, int32_t n_states, int32_t nb_substeps, const double* cell_dims,
                         int32_t n_cell_dims, double* out);

/* ---- one process, several GPUs -------------------------------------------------------------------------------------------------------
 * The reference parallelises with multiprocessing.Pool.map over track chunks and sums the per-chunk results (extrack/tracking.py:1061-1069).
 * A host without torch.distributed (option B) drives all the GPUs of a node from one thread through these entry points:
 * every device keeps a contiguous row range of every uploaded bucket, an evaluation runs the likelihood kernel on all devices at once and ends
 * with ONE all-reduce of the scalar over RCCL / xGMI (ncclCommInitAll over the listed devices at creation, ncclAllReduce(sum, double, 1) per
 * evaluation).  use_rccl: 0 = sum the per-device totals on the host; 1 = RCCL when librccl.so can be loaded (at run time, no link-time
 * dependency) and the devices are distinct, else the host sum; 2 = RCCL or EXTRACK_E_UNSUPPORTED.  The multi-PROCESS form of the same
 * partitioning (one rank per GPU, torch.distributed) is extrack_amd/distributed.py. */
typedef struct extrack_multi extrack_multi;
int extrack_multi_create(int32_t n_devices, const int32_t* device_ids, int32_t use_rccl, extrack_multi** out);
void extrack_multi_destroy(extrack_multi* m);
const char* extrack_multi_last_error(const extrack_multi* m);
int32_t extrack_multi_device_count(const extrack_multi* m);
int32_t extrack_multi_uses_rccl(const extrack_multi* m);
/* The per-device context i (owned by m): for the single-device entry points above on one shard, e.g. extrack_predict. */
extrack_ctx* extrack_multi_context(extrack_multi* m, int32_t i);
/* One length bucket, host [n][len][dims] (+ per-peak errors): device r receives the rows [r n / w, (r + 1) n / w) (balanced to one row). */
int extrack_multi_upload_bucket(extrack_multi* m, const double* tracks, int64_t n, int32_t len, int32_t dims, const double* sigma,
                                int32_t sigma_dims);
int extrack_multi_clear_buckets(extrack_multi* m);
/* sum(LL) over all buckets on all devices (as extrack_loglik; model->min_len / max_len are the dataset-global values). */
int extrack_multi_loglik(extrack_multi* m, const extrack_model* model, double* total_ll);

#ifdef __cplusplus
}
#endif
#endif /* EXTRACK_HIP_H */

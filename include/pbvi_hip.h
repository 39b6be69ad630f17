/*
 * pbvi_hip.h -- C-ABI of the MI355X (gfx950) PBVI alpha-vector backup engine.
 *
 * Drop-in boundary for ONE path of PimLb/POMDP_PBVI_Exploration: one call of
 * PBVI_Solver.backup (reference src/pomdp.py:1447-1524) plus the two kernels its
 * callers run on the same operands (ValueFunction.prune level 2, src/mdp.py:857-866;
 * compute_change's max_v b.alpha_v, src/pomdp.py:2165).  The reference has no FFI:
 * its GPU seam is CuPy array-module dispatch (`xp = cp.get_array_module(...)`,
 * src/pomdp.py:1482) over objects moved by Model.gpu_model (src/mdp.py:533-560),
 * ValueFunction.to_gpu (src/mdp.py:782-805) and BeliefSet.to_gpu
 * (src/pomdp.py:613-634).  Each entry point below names the reference statement(s)
 * it replaces.  Plain C types only; every array is C-order exactly as NumPy lays
 * the reference's arrays out.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - `dtype` selects the element type T of every `const void*` / `void*` array:
 *     PBVI_F32 -> float, PBVI_F64 -> double.  Indices are int32 (the binding narrows
 *     NumPy int64 after a range check).
 *   - Return value: 0 on success, negative pbvi_status otherwise; the message is
 *     available from pbvi_last_error() (thread-local).  The library itself never calls abort() or
 *     exit(); a GPU memory fault, however, is raised by the HSA runtime as a process abort that no
 *     library can intercept -- the engine validates every index array it is handed (ids, keys, actions,
 *     observations) on the host before a kernel can use it for exactly that reason.
 *   - Host pointers are owned by the caller and only read/written during the call.
 *     Device memory is owned by the handle.  A handle is not thread-safe.
 *   - Every call is synchronous: results are in the output buffers on return.
 */
#ifndef PBVI_HIP_H
#define PBVI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pbvi_engine pbvi_engine_t;

enum pbvi_status {
    PBVI_OK = 0,
    PBVI_EINVAL = -1,        /* bad argument (the reference asserts; binding raises ValueError) */
    PBVI_ENOMEM = -2,        /* host or hipMalloc OOM (binding raises MemoryError, which
                                PBVI_Solver.solve catches: src/pomdp.py:2399-2401) */
    PBVI_ERUNTIME = -3,      /* HIP runtime error */
    PBVI_EUNSUPPORTED = -4   /* configuration not supported by this build */
};

enum pbvi_dtype { PBVI_F32 = 0, PBVI_F64 = 1 };

/* Transition representation used for the Gamma projection (src/pomdp.py:1485-1491). */
enum pbvi_mode {
    PBVI_SPARSE = 0,   /* reachable-state padded-ELL SpMM (what the reference always does) */
    PBVI_DENSE = 1     /* densify T[A][S][S] on device and run the projection as |A| MFMA GEMMs */
};

enum pbvi_flags {
    PBVI_BELIEF_DOMINANCE = 1   /* backup(..., belief_dominance_prune=True), src/pomdp.py:1509-1515 */
};

/* Per-call instrumentation (kernel times from HIP events on the engine's stream). */
typedef struct pbvi_stats {
    double ms_total;        /* whole device-side backup, first kernel to last */
    double ms_project;      /* Gamma projection (K1) */
    double ms_score;        /* belief x Gamma score GEMM (K2) */
    double ms_argmax;       /* argmax over alpha-vectors + near-tie detection */
    double ms_refine;       /* fp64 refinement of near-ties (f32 engines) */
    double ms_action;       /* action values + argmax (K4) */
    double ms_assemble;     /* alpha' gather-sum (K3) */
    double ms_dominance;    /* belief-dominance test (K5), 0 if not requested */
    int64_t n_pairs;        /* B*A*O (belief, action, observation) triples */
    int64_t n_dead;         /* triples with P(o|b,a) == 0 (all scores exactly 0) */
    int64_t n_refined;      /* triples whose argmax was re-decided in fp64 */
    int64_t n_refined_actions; /* beliefs whose action argmax was re-decided in fp64 */
    int64_t n_unique;       /* distinct (action, best_alpha_ind[action]) keys = alpha' rows actually assembled */
    int64_t score_flops;    /* algorithmic 2*B*S*A*O*V of the score GEMM */
    int64_t score_flops_executed; /* MFMA flops actually issued (zero tiles skipped, pad tiles included) */
    int64_t score_tiles_dense;    /* 256x256x32 tile steps a dense GEMM of the padded shape would run */
    int64_t score_tiles_run;      /* tile steps actually run */
    int64_t project_flops;          /* dense mode: algorithmic 2*A*O*V*S*S of the projection GEMMs, else 0 */
    int64_t project_flops_executed; /* dense mode: MFMA flops issued by the projection GEMMs */
    int32_t split_k;        /* max K-chunks (partial slabs) per tile pair */
    int32_t formulation;    /* which operand was projected: 1 = alpha-vectors (Gamma), 2 = beliefs (pbvi_set_formulation) */
    int64_t n_refine_candidates; /* f32 engines: near-tie candidates listed over all refined triples (a triple that moves
                                    to the GEMM path -- every alpha row re-scored -- stops listing) */
    double ms_project_gemm;  /* dense mode: the batched projection GEMM kernel alone (ms_project also holds the
                                clears and the scale/copy pass around it); 0 in sparse mode */
    int32_t screened;        /* fp64 engines: 1 when the scores came from the fp32 screen (fp32 stream-K GEMM on rounded
                                copies of the operands, near-ties re-decided from the fp64 originals), 0 = pure fp64 */
    int32_t fused_projection; /* 1 when the score GEMM generated its Gamma tiles in the operand staging (R = 1, fp32
                                scoring, alpha-side formulation): ms_project then covers only the few projected rows */
} pbvi_stats_t;

/* Library / device queries. */
int pbvi_version(void);
int pbvi_device_count(void);                 /* hipGetDeviceCount; 0 if no GPU */
const char* pbvi_last_error(void);

/*
 * Create an engine for one POMDP model on one device.  Replaces Model.gpu_model
 * (src/mdp.py:533-560): uploads and re-tiles the three tables the backup reads.
 *   reach_states [S][A][R] int32  = model.reachable_states            (src/mdp.py:194-201,296-335)
 *   rto          [S][A][O][R] T   = model.reachable_transitional_observation_table (src/pomdp.py:201-202)
 *   exp_reward   [S][A] T         = model.expected_rewards_table      (src/pomdp.py:251)
 */
int pbvi_engine_create(pbvi_engine_t** out, int device, int32_t S, int32_t A, int32_t O, int32_t R,
                       const int32_t* reach_states, const void* rto, const void* exp_reward,
                       int dtype, int mode);
void pbvi_engine_destroy(pbvi_engine_t* e);

/*
 * Device-resident alpha-vector set.  Replaces ValueFunction.to_gpu (src/mdp.py:782-805)
 * and the per-call re-stack of ValueFunction.alpha_vector_array (src/mdp.py:687-697).
 *   alpha [V][S] T, row-major.  set replaces the whole set; append adds rows at the end.
 */
int pbvi_alpha_set(pbvi_engine_t* e, const void* alpha, int64_t V);
int pbvi_alpha_append(pbvi_engine_t* e, const void* alpha, int64_t V_add);
int64_t pbvi_alpha_count(const pbvi_engine_t* e);

/*
 * Device-resident belief block.  Replaces BeliefSet.to_gpu (src/pomdp.py:613-634).
 *   beliefs [B][S] T, row-major.
 */
int pbvi_beliefs_set(pbvi_engine_t* e, const void* beliefs, int64_t B);

/*
 * Device row stores.  The reference keeps every alpha-vector / belief on the GPU as its own CuPy array and
 * re-stacks them into a matrix on demand (ValueFunction.alpha_vector_array, src/mdp.py:687-697 -- an
 * O(V*S) copy after every extend; BeliefSet.belief_array, src/pomdp.py:541-550).  Here a row is uploaded
 * once (store_append returns the id of the first appended row, ids are consecutive) and the working
 * alpha set / belief block is then SELECTED by id, in the caller's order, entirely on the device.  The
 * order matters: argmax ties go to the lowest index, and ValueFunction.extend puts new vectors first
 * (src/mdp.py:773-774), so the selection must follow the host list order, not the upload order.
 * store_reset forgets all stored rows (ids restart at 0).
 *
 * pbvi_alpha_select keeps the selected set laid out with free rows in front of it: a later selection that is k new ids
 * followed by exactly the ids of the resident one -- what `new_value_function.extend(value_function)` produces after every
 * backup of a solve loop (src/pomdp.py:1521-1522) -- gathers those k rows only; a selection at most a quarter of the
 * resident one's size (compute_change's score of the rows an expansion added, src/pomdp.py:2141-2169) is placed beside it and
 * leaves it intact.  Results never depend on which of the paths was taken (same rows in the same order).
 * pbvi_alpha_layout reports it (tests, diagnostics): returns 1 when the working set is such an extendable selection, 0
 * otherwise (uploaded with pbvi_alpha_set / _append, or a small selection); *free_rows = rows still free in front of it,
 * *layouts = how many times a selection had to be gathered afresh so far.  Either pointer may be NULL.
 */
int64_t pbvi_alpha_store_append(pbvi_engine_t* e, const void* rows /* [n][S] T */, int64_t n);
int pbvi_alpha_select(pbvi_engine_t* e, const int32_t* ids /* [V] */, int64_t V);
int pbvi_alpha_layout(pbvi_engine_t* e, int64_t* free_rows, int64_t* layouts);
int pbvi_alpha_store_reset(pbvi_engine_t* e);
int64_t pbvi_belief_store_append(pbvi_engine_t* e, const void* rows /* [n][S] T */, int64_t n);
int pbvi_beliefs_select(pbvi_engine_t* e, const int32_t* ids /* [B] */, int64_t B);
int pbvi_belief_store_reset(pbvi_engine_t* e);

/*
 * One point-based backup on the resident alpha set and belief block
 * (src/pomdp.py:1485-1515).  Results stay on the device until fetched.
 */
int pbvi_backup_run(pbvi_engine_t* e, double gamma, int flags, pbvi_stats_t* stats /* may be NULL */);

/*
 * Copy the results of the last pbvi_backup_run to caller buffers (any may be NULL).
 * Destinations may be host memory or device memory of the engine's GPU (the copy kind
 * is resolved from the address), so the multi-GPU layer can land rows in its send buffer:
 *   out_alpha      [B][S] T        alpha_vectors  (src/pomdp.py:1506)
 *   out_action     [B] int32       best_actions   (src/pomdp.py:1505)
 *   out_best_alpha [B][A][O] int32 best_alpha_ind (src/pomdp.py:1495)
 *   out_keep       [B] uint8       dominating_vectors (src/pomdp.py:1512); all 1 if the
 *                                  run did not request PBVI_BELIEF_DOMINANCE
 */
int pbvi_backup_fetch(pbvi_engine_t* e, void* out_alpha, int32_t* out_action,
                      int32_t* out_best_alpha, uint8_t* out_keep);

/*
 * Deduplicated form of the same result (K6).  Beliefs with equal (best_action, best_alpha_ind[b,
 * best_action, :]) produce byte-identical alpha' rows, so the engine assembles one row per distinct
 * key -- a subset of the duplicates ValueFunction.__init__ removes by bytes (src/mdp.py:667-669), found
 * before any row is computed or moved.  U = pbvi_backup_unique_count():
 *   out_rows  [U][S] T     row u = alpha' of the first belief (in the caller's order) with that key
 *   out_index [B] int32    alpha'[b] == out_rows[out_index[b]]
 * Destinations may be host or device memory.  pbvi_backup_fetch(out_alpha) expands this to [B][S].
 */
int64_t pbvi_backup_unique_count(const pbvi_engine_t* e);
int pbvi_backup_fetch_unique(pbvi_engine_t* e, void* out_rows, int32_t* out_index);

/*
 * Everything ValueFunction.__init__ needs of a backup (src/mdp.py:660-669 -- the `tobytes()` there is what makes
 * the reference's own backup timing include the device-to-host copy), in ONE staged transfer and one
 * synchronisation: out_rows [U][S] T, out_index [B], out_action [B], out_best_alpha [B][A][O], out_keep [B]; any
 * of them may be NULL.  Destinations in page-locked host memory (pbvi_host_alloc) are written by DMA directly;
 * pageable ones go through the engine's pinned bounce buffer.  This is the call SURVEY 8d's metric is timed on.
 */
int pbvi_backup_fetch_compact(pbvi_engine_t* e, void* out_rows, int32_t* out_index, int32_t* out_action,
                              int32_t* out_best_alpha, uint8_t* out_keep);

/*
 * pbvi_backup_run + pbvi_backup_fetch_compact in ONE call, with the rows leaving early.  The first maxima of the score
 * GEMM decide most beliefs' keys for good (the fp64 refinement re-decides near-ties only), so the distinct rows of that
 * provisional decision are assembled and written into out_rows by the device WHILE the refinement, the action stage and the
 * dedup run; the final keys are then matched against the provisional ones and only rows the refinement changed are appended.
 *   out_rows  [cap_rows][S] T   page-locked host memory (pbvi_host_alloc), cap_rows >= B; rows arrive in SLOT order
 *   out_slot  [B] int32         first U entries: row of distinct key u (order of first occurrence, as
 *                               pbvi_backup_fetch_compact lists them) is out_rows[out_slot[u]]
 *   out_index [B], out_action [B], out_best_alpha [B][A][O] (may be NULL), out_keep [B] (may be NULL): as
 *                               pbvi_backup_fetch_compact; alpha'[b] == out_rows[out_slot[out_index[b]]]
 *   *n_unique = U, *n_slots = rows written (>= U: provisional rows the refinement overturned stay behind, unreferenced)
 * Same results as the two calls (src/pomdp.py:1485-1506 + the tobytes() of src/mdp.py:667-669), bit for bit; where the
 * early path does not apply (fp64 scoring, belief-side formulation, belief-dominance test) the rows are copied at the end
 * and out_slot is the identity.
 */
int pbvi_backup_run_fetch(pbvi_engine_t* e, double gamma, int flags, pbvi_stats_t* stats, void* out_rows, int64_t cap_rows,
                          int32_t* out_slot, int32_t* out_index, int32_t* out_action, int32_t* out_best_alpha, uint8_t* out_keep,
                          int64_t* n_unique, int64_t* n_slots);

/*
 * Hashes of the U distinct rows of the last backup, out_hashes [U] uint64: sum_i bits(row, i) * (2 i + 1) mod 2^64 over
 * the row's fp32 / fp64 bit patterns.  ValueFunction.__init__ keys its dedup dictionary on `values.tobytes()`
 * (src/mdp.py:667-669) -- 120-240 KB copied and hashed per row; the host mirror keys it on this number instead (equality
 * is still decided on the bytes), computed where the rows are.
 */
int pbvi_backup_fetch_row_hashes(pbvi_engine_t* e, uint64_t* out_hashes);

/* Page-locked host memory for result buffers (hipHostMalloc / hipHostFree); NULL on failure.  The reference's
 * counterpart is CuPy's pinned-memory pool behind `cp.asnumpy` (src/mdp.py:806-831, ValueFunction.to_cpu). */
void* pbvi_host_alloc(size_t bytes);
void pbvi_host_free(void* p);

/*
 * The new alpha-vectors join the value function (ValueFunction.extend, src/mdp.py:763-779; append=True in
 * PBVI_Solver.backup, src/pomdp.py:1521-1522): append n of the last backup's distinct rows -- unique_idx[i] in
 * [0, U), in that order -- to the engine's alpha store, device to device.  Returns the store id of the first
 * appended row (ids are consecutive), or a negative error code.  The caller tags its AlphaVector objects with the
 * ids, so the next pbvi_alpha_select needs no upload of rows the engine computed itself.
 */
int64_t pbvi_backup_store_unique(pbvi_engine_t* e, const int32_t* unique_idx, int64_t n);

/*
 * Keys instead of rows (multi-GPU exchange).  The alpha' row of a belief is a function of its key
 * (a*, v*[a*, 0..O-1]) and of the replicated alpha set and model only (src/pomdp.py:1497-1506), so ranks exchange
 * keys -- (1+O) ints per distinct row -- and every rank assembles the rows it did not compute itself:
 *   pbvi_backup_fetch_unique_keys : out_keys [U][1+O] int32 of the last backup's U distinct rows (host or device)
 *   pbvi_assemble_rows            : out_rows [n][S] T (host or device) from n keys against the RESIDENT alpha set;
 *                                   byte-identical to the rows the producing rank holds.  1 <= n <= 65535.
 */
int pbvi_backup_fetch_unique_keys(pbvi_engine_t* e, int32_t* out_keys);
/* Everything a rank contributes to the exchange in one int32 buffer of 1 + 3B + B(1+O) entries (host or device):
 *   [0] U | index [B] | actions [B] | keep [B] | keys of the U distinct rows, padded with zeros to [B][1+O]. */
int pbvi_backup_fetch_exchange(pbvi_engine_t* e, int32_t* out);
/* Same with per-sized sections, per >= B (= ceil(B_total / ranks) of a sharded run): 1 + 3 per + per (1+O) entries,
 * zeros behind the B valid ones, so every rank of a ragged split sends a message of the same length. */
int pbvi_backup_fetch_exchange_padded(pbvi_engine_t* e, int64_t per, int32_t* out);
int pbvi_assemble_rows(pbvi_engine_t* e, double gamma, int64_t n, const int32_t* keys, void* out_rows);
/* pbvi_assemble_rows whose rows also join the alpha store (ValueFunction.extend on every replica of a sharded
 * backup, src/pomdp.py:1521-1522 / src/mdp.py:763-779): returns the store id of the first of the n appended rows,
 * or a negative error code.  out_rows (host or device, [n][S] T) may be NULL. */
int64_t pbvi_assemble_rows_store(pbvi_engine_t* e, double gamma, int64_t n, const int32_t* keys, void* out_rows);
/*
 * Host side of the key exchange of a belief-sharded backup (the step between the all-gather and
 * pbvi_assemble_rows_store; the reference has no counterpart: one device, Experiments/Olfactory Navigation/run_test.py:12).
 * all_meta: the `world` messages of pbvi_backup_fetch_exchange_padded in rank order, `stride` int32 apart
 * (stride >= 1 + 3 per + per key_width: a caller may carry its own trailer behind each message), HOST memory.
 * Rank r holds the beliefs [r per, min((r+1) per, n_total)).  Equal keys found by different ranks are ONE row.
 *   out_keys   [sum of the messages' counts][key_width]  the globally distinct keys in order of first occurrence
 *   out_index  [n_total]  position of each belief's key in out_keys     out_action / out_keep [n_total]
 * Returns the number of distinct keys, or a negative error code.  Pure host code: needs no engine and no device.
 */
int64_t pbvi_exchange_merge(const int32_t* all_meta, int32_t world, int64_t stride, int64_t per, int32_t key_width,
                            int64_t n_total, int32_t* out_keys, int32_t* out_index, int32_t* out_action, uint8_t* out_keep);

/*
 * Device addresses of the last run's results, for the multi-GPU layer to hand to
 * RCCL (all-gather of the new alpha rows) without a host round trip.
 *   *d_alpha [B][S] T (row stride S), *d_action [B] int32, *d_keep [B] uint8.
 */
int pbvi_backup_device_results(pbvi_engine_t* e, void** d_alpha, int32_t** d_action, uint8_t** d_keep);

/*
 * Convenience: the reference seam in one call = beliefs_set + backup_run + backup_fetch
 * (PBVI_Solver.backup(model, belief_set, value_function, ...) before ValueFunction dedup).
 */
int pbvi_backup(pbvi_engine_t* e, const void* beliefs, int64_t B, double gamma, int flags,
                void* out_alpha, int32_t* out_action, int32_t* out_best_alpha, uint8_t* out_keep,
                pbvi_stats_t* stats);

/*
 * ValueFunction.prune(level=2) on the resident alpha set (src/mdp.py:857-866):
 *   keep[i] = 1 iff no other row j has alpha[j][s] >= alpha[i][s] for every s.
 */
int pbvi_prune_dominated(pbvi_engine_t* e, uint8_t* keep /* [V] */);

/*
 * max_v b.alpha_v over the resident alpha set for the resident belief block
 * (compute_change, src/pomdp.py:2165-2166; also the |V|-limiter scan :2349-2352).
 *   out_value [B] double, out_index [B] int32 (first max), either may be NULL.
 */
int pbvi_value_max(pbvi_engine_t* e, double* out_value, int32_t* out_index);

/*
 * The same maxima for rows [0, n) of the BELIEF STORE (pbvi_belief_store_append / pbvi_belief_walk order) against the
 * working alpha set, with the store itself as the GEMM operand: nothing is gathered or sorted and the zero maps and
 * tile lists of rows scored before are kept.  compute_change (src/pomdp.py:2141-2169) scores the whole accumulated
 * belief set after every backup; that set only grows.  out_value / out_index: [n] (either may be NULL), store order.
 */
int pbvi_value_max_store(pbvi_engine_t* e, int64_t n, double* out_value, int32_t* out_index);
/*
 * max_v b.alpha_v of the beliefs of the last pbvi_backup_run against the alpha set it ran on, out_value [B] in the
 * caller's order: what compute_change (src/pomdp.py:2165) asks for next about exactly these beliefs and that set.
 * In the belief-side formulation the beliefs ride along as extra rows of the score GEMM (they fit its row padding), so
 * the values cost one row-max kernel.  fp32 engines: the GEMM's maxima (as with pbvi_set_value_max_exact(0)); f64
 * engines: the same sums pbvi_value_max returns.  PBVI_EUNSUPPORTED (-4) after an alpha-side backup
 * (pbvi_stats_t.formulation == 1).
 */
int pbvi_backup_fetch_value_max(pbvi_engine_t* e, double* out_value);
/* rows currently held by the stores (ids are 0 .. count-1) */
int64_t pbvi_belief_store_count(const pbvi_engine_t* e);
int64_t pbvi_alpha_store_count(const pbvi_engine_t* e);

/*
 * f32 engines re-score the candidates of every belief in fp64, so pbvi_value_max / pbvi_value_max_store return exact
 * maxima and first-maximum indices (np.argmax semantics).  compute_change only takes max|new - old| of the values and
 * compares it with eps * gamma / (1 - gamma) (src/pomdp.py:2167-2169, :2372): exact = 0 skips the re-scoring and
 * returns the fp32 GEMM's maxima (relative error ~1e-7, within the 1e-6 bar of fp32 engines); indices are then the
 * fp32 argmax.  Default 1.  No effect on f64 engines.
 */
int pbvi_set_value_max_exact(pbvi_engine_t* e, int exact);

/*
 * Batched belief update (Bayes step) of the resident belief block: the step that produces the beliefs the
 * backup consumes.  Replaces Belief.update (src/pomdp.py:382-421) applied to B beliefs at once (the
 * reference's own batched form lives in its simulator, src/pomdp.py:3277-3310):
 *   out[b] = normalise( sum_{s,r} b[b,s] * RTO[s, actions[b], observations[b], r]  scattered to rs[s,a,r] )
 *   actions, observations [B] int32 (caller's belief order); out_beliefs [B][S] T, host or device memory.
 * A belief whose update has zero mass (impossible observation) yields NaNs, as the reference's 0/0 does.
 */
int pbvi_belief_update(pbvi_engine_t* e, const int32_t* actions, const int32_t* observations, void* out_beliefs);

/*
 * Policy-evaluation step of the parallel simulator (Agent.run_n_simulations_parallel, src/pomdp.py:3296-3347)
 * on the resident belief block, which never leaves the device between steps:
 *   1. pbvi_value_max gives out_index[b] = first argmax_v b.alpha_v  (Agent.get_best_action, :3029); the caller
 *      maps it through ValueFunction.actions and runs its simulator (host RNG, SimulationSet.run_actions);
 *   2. pbvi_beliefs_advance replaces every belief by its Bayes update with (actions[b], observations[b])
 *      (:3306-3311) and drops the rows with keep[b] == 0 (:3326-3329, the done-filter); survivors keep the
 *      caller's order.  keep may be NULL (keep all).  *out_B (may be NULL) receives the new row count; when it
 *      is 0 no belief block is resident any more.
 * pbvi_beliefs_fetch copies the resident block out, [B][S] T in caller order (host or device memory);
 * pbvi_beliefs_count returns B (0 = none, -1 = NULL handle).
 */
int pbvi_beliefs_advance(pbvi_engine_t* e, const int32_t* actions, const int32_t* observations, const uint8_t* keep,
                         int64_t* out_B);
int pbvi_beliefs_fetch(pbvi_engine_t* e, void* out_beliefs);
int64_t pbvi_beliefs_count(const pbvi_engine_t* e);

/*
 * MDP value iteration on the device (VI_Solver.solve, src/mdp.py:1442-1525; seeds FSVI / HSVI):
 *   rows[a][s] = ER[s,a] + gamma * sum_r P[s,a,r] * v[rs[s,a,r]];   v'[s] = max_a rows[a][s]
 * repeated from v0 until max_s |v' - v| < max_change_limit (the reference's eps * gamma / (1 - gamma)) or
 * `horizon` sweeps.  Always fp64.  Not tied to an engine handle (the MDP tables are not the POMDP's).
 *   reach_states [S][A][R] int32, reach_prob [S][A][R], exp_reward [S][A], v0 [S]   (NumPy C-order)
 *   out_rows [A][S]: rows of the last sweep (the solution's alpha-vectors, before the container's dedup)
 *   out_changes [horizon] (may be NULL): max change of each sweep that ran;  out_iterations (may be NULL).
 */
int pbvi_mdp_value_iteration(int device, int32_t S, int32_t A, int32_t R, const int32_t* reach_states,
                             const double* reach_prob, const double* exp_reward, const double* v0, double gamma,
                             double max_change_limit, int32_t horizon, double* out_rows, double* out_changes,
                             int32_t* out_iterations);

/*
 * Belief walk of the FSVI-style expansions (PBVI_Solver.expand_fsvi / expand_fsvi_eg, src/pomdp.py:1895-1935; the
 * trajectory of (action, observation) pairs is simulated by the caller in the underlying MDP and does not depend
 * on the beliefs): n chained Bayes updates (Belief.update, :382-421) entirely on the device, in fp64,
 *   b_{i+1} = update(restart[i] ? b0 : b_i, actions[i], observations[i]),        b_0 = b0
 * Each b_{i+1} is appended to the belief row store (so the following backup selects it by id with no upload) and
 * copied to out_beliefs [n][S] fp64 (host), which the caller's containers need for their byte-keyed dedup.
 * restart may be NULL (never).  Returns the store id of b_1 (b_{i+1} has id + i), or a negative error code.
 */
int64_t pbvi_belief_walk(pbvi_engine_t* e, const double* b0, int64_t n, const int32_t* actions, const int32_t* observations,
                         const uint8_t* restart, double* out_beliefs);
/*
 * Dedup keys of the n beliefs of the last pbvi_belief_walk: out_keys[i] = sum_s bits(row i, s) * (2 s + 1) mod 2^64 over
 * the 64-bit patterns of the fp64 row i (the hash of pbvi_backup_fetch_row_hashes) -- the integer the host's belief containers hash on in place of the row's bytes (BeliefSet.union,
 * src/pomdp.py:585-606, keys its dictionaries on values.tobytes()) -- so the host does not pass over the rows again.
 */
int pbvi_belief_walk_keys(pbvi_engine_t* e, int64_t n, uint64_t* out_keys);
/* Optional fp64 copy of RTO ([S][A][O][R], as at creation) for pbvi_belief_walk on an f32 engine, so the fp64 belief
 * values it returns do not depend on the engine's arithmetic type (f64 engines read their own table). */
int pbvi_engine_set_rto_f64(pbvi_engine_t* e, const double* rto);

/*
 * Which operand of the score GEMM is projected through the model (sparse mode; same scores, re-associated):
 *   1 = alpha-vectors, the reference's order (Gamma[a,o,v,:], src/pomdp.py:1489-1491; GEMM [B] x [A*O*V]);
 *   2 = beliefs (bp[a,o,b,:] = gamma * sum b[s] RTO[s,a,o,r] scattered to rs[s,a,r]; GEMM [B*A*O] x [V]);
 *   0 = automatic (default): the cheaper of the two by projected rows and 256-row tile count -- the belief side
 *       when B << V, e.g. the solve loop's ~100 new beliefs against thousands of alpha-vectors -- unless Gamma and its
 *       score slabs exceed what the engine may still allocate (its cap, or the free device memory) while the projected
 *       beliefs fit: then the belief side, whatever it costs.  (The reference dies there: Sea_Robin_Real.ipynb:913.)
 */
int pbvi_set_formulation(pbvi_engine_t* e, int formulation);

/*
 * fp64 engines only (a no-op setting on fp32 ones): the fp32 SCREEN.  mode 1 (default): when the score GEMM is large
 * the scores are computed by the fp32 stream-K GEMM on fp32-rounded copies of the operands, with the tie window widened
 * by the input roundings, and every (belief, action, observation) and action whose winner is not clear is re-decided
 * from the fp64 originals -- same indices and values as the pure fp64 pipeline up to that pipeline's own summation
 * order, at about the fp32 engine's speed (the fp64 MFMA GEMM sustains a third of the fp32 one).  0: never (pure
 * fp64 arithmetic throughout, src/pomdp.py:1485-1506 literally); 2: always, whatever the size (tests).
 * PBVI_F64_SCREEN=off|auto|always in the environment sets the initial mode.
 */
int pbvi_set_f64_screen(pbvi_engine_t* e, int mode);

/*
 * fp32 scoring with one reachable state per (s, a) (every large model of the reference): 1 (default) = the score GEMM
 * generates the Gamma tiles (src/pomdp.py:1485-1491) on the way into LDS instead of reading a projected copy from HBM;
 * 0 = project first (k_project), then multiply -- the same scores bit for bit, kept for A/B measurements.
 * 2 = also with 2..7 reachable states per (s, a) (the padded-ELL SpMM inside the operand staging; bit-identical again, but
 * measured SLOWER than projecting first at |S| = 30000, R = 5 -- five shifted alpha windows per K step cost more L2 /
 * Infinity-Cache traffic than reading the projected tile once -- so it is never chosen automatically).  No effect where
 * the fused path does not apply (R > 7, successor maps without grid structure, fp64 scoring, belief-side formulation,
 * dense mode).
 */
int pbvi_set_fused_projection(pbvi_engine_t* e, int enable);

/* Tuning knob for f32 engines: relative half-width of the near-tie window that sends an
 * argmax to fp64 refinement (<= 0 restores the default derived from |S|). */
int pbvi_set_tie_window(pbvi_engine_t* e, double rel);

/* Debug aid for tests: when enabled (or PBVI_POISON is set in the environment) every fresh device
 * allocation is filled with 0xFF bytes (NaN floats, -1 indices), so a read of memory the engine never
 * wrote fails the parity tests instead of hiding behind zero-filled pages.  Returns the previous state. */
int pbvi_debug_poison(int enable);
/*
 * The MemoryError contract of src/pomdp.py:2399-2401 ("Memory full ... returning value function and history as is"):
 * a device allocation that fails makes the call return PBVI_ENOMEM (-2), which the Python seam raises as MemoryError and
 * PBVI_Solver.solve turns into the partial result.
 *   pbvi_debug_alloc_limit : cap (MiB, < 0 = none; also PBVI_ALLOC_LIMIT_MB) on the bytes of device buffers ONE engine
 *                            may hold, so that the contract can be tested deterministically; returns the previous cap.
 *   pbvi_engine_after_oom  : after a -2, release every working set, row store and scratch buffer (the model tables stay):
 *                            the engine is as freshly created and usable again.
 */
int64_t pbvi_debug_alloc_limit(int64_t mb);
int pbvi_engine_after_oom(pbvi_engine_t* e);

/* Benchmark / debug: list every K tile of every GEMM tile pair, zero or not -- the "dense backup" configuration of
 * BASELINE.json is measured this way (results are unchanged: skipped tiles only ever add +0).  Also enabled by
 * PBVI_GEMM_DENSE=1 in the environment.  Process-wide; returns the previous state. */
int pbvi_debug_gemm_dense(int enable);

/* Bytes of device memory currently held by the handle. */
int64_t pbvi_device_bytes(const pbvi_engine_t* e);

#ifdef __cplusplus
}
#endif
#endif /* PBVI_HIP_H */

// Launchers for the HBM-bound kernels of the backup (internal).
#pragma once
#include "pbvi_common.h"

namespace pbvi {

// Device view of the model tables, re-tiled at engine creation so that every kernel
// reads them coalesced along s:
//   rs  [A][R][S_pad]    int32   reachable_states          (reference layout [S][A][R])
//   rto [A][O][R][S_pad] T       RTO                       (reference layout [S][A][O][R])
//   er  [A][S_pad]       T       expected_rewards_table^T  (reference layout [S][A])
//   sup [A*O][S_pad]     uint8   1 where any_r RTO[s,a,o,r] != 0
// Pad columns (s >= S) hold rs = 0, rto = 0, er = 0, sup = 0.
template <typename T>
struct ModelView {
    int S, S_pad, A, O, R;
    const int32_t* rs;
    const T* rto;
    const T* er;
    const uint8_t* sup;
};

// Score matrix as written by the GEMM: split-K partial slabs.  f32 engines: nchunks[pair] slabs
// exist for tile pair (col>>8, row>>8) (zero-tile skipping makes the count per pair vary); f64
// engines: one dense slab (nchunks == nullptr, fixed = 1).
// Row of (observation o, action a, belief b) in the belief-side operand.  Observation-major, then belief, then action:
// the rows of one observation share tiles (a rarely-seen observation's near-empty tiles are skipped) and a 256-row
// tile holds all actions of ~256/A consecutive beliefs -- neighbours in a walk, whose supports overlap -- rather than
// a few actions of every belief of the block.
__host__ __device__ __forceinline__ int64_t push_row_index(int o, int a, int b, int A, int B) {
#ifdef PBVI_PUSH_ORDER_OAB
    return ((int64_t)o * A + a) * B + b;
#else
    return ((int64_t)o * B + b) * A + a;
#endif
}

template <typename T>
struct SlabView {
    const T* slabs;
    int64_t slab_stride;
    int ldc;
    const int* nchunks;   // [tiles_n][tiles_m] or nullptr
    int tiles_m;
    int fixed;
    // Stream-K layout (GemmPlan::c_floats): chunk 0 of every pair in the full slab, chunk z >= 1 of pair p in the 256 x 256
    // tile of block first_block[p] + z behind it; nullptr: chunk z at slabs + z * slab_stride.
    const int* first_block = nullptr;   // [tiles_n][tiles_m]
    // Belief-side formulation (beliefs projected, not alpha-vectors): rows are (observation, action, belief) =
    // ((o * A + a) * push_B + b) -- observation-major, so the rows of a rarely-seen observation (near-empty after
    // the RTO product) share 256-row tiles and those tiles are skipped -- columns are the alpha index; the
    // magnitude scores and reward dots come from side arrays.
    int push = 0;
    int push_B = 0, push_A = 1, push_O = 1;
    const double* aux_mag = nullptr;   // [B][G]  sum_s' |bp[g,b,s']| * max_v|alpha[v,s']|
    const double* aux_rd = nullptr;    // [B][A]  b . ER[:,a], accumulated in f64
    __device__ __forceinline__ T at(int64_t row, int64_t col) const {
        const T* p = slabs + row * ldc + col;
        const int64_t pair = nchunks ? (col >> 8) * tiles_m + (row >> 8) : 0;
        const int n = nchunks ? nchunks[pair] : fixed;
        T s = T(0);
        if (first_block != nullptr) {
            if (n > 0) s += p[0];
            if (n > 1) {
                const T* x = slabs + slab_stride + (int64_t)(first_block[pair] + 1) * (256 * 256) + (row & 255) * 256 + (col & 255);
                for (int z = 1; z < n; ++z, x += 256 * 256) s += *x;
            }
            return s;
        }
        for (int z = 0; z < n; ++z) s += p[(int64_t)z * slab_stride];
        return s;
    }
    __device__ __forceinline__ int64_t push_row(int b, int g) const {      // g = a * O + o
        return push_row_index(g % push_O, g / push_O, b, push_A, push_B);
    }
    // score of (belief b, group g, alpha v); V = columns per group in the alpha-side layout
    __device__ __forceinline__ T score(int b, int g, int V, int v) const {
        return push ? at(push_row(b, g), v) : at(b, (int64_t)g * V + v);
    }
    __device__ __forceinline__ double magnitude(int b, int g, int G, int V) const {
        return push ? aux_mag[(int64_t)b * G + g] : (double)at(b, (int64_t)G * V + g);
    }
};

template <typename T>
hipError_t launch_support(ModelView<T> mv, uint8_t* sup, hipStream_t st);

// K1-sparse: Gamma[(a*O+o)*V + v][s] = gamma * sum_r rto[a][o][r][s] * alpha[v][rs[a][r][s]] for v < V;
// alpha row V (the magnitude row max_v|alpha|) is projected to Gamma row A*O*V + (a*O+o), so every
// (a,o) group is exactly V rows and stays aligned to the GEMM's 256-row tiles when V % 256 == 0.
template <typename T>
hipError_t launch_project(const T* alpha, int lda, int V, ModelView<T> mv, T gamma, T* gam, int ldg,
                          const uint8_t* need /* [A*O][k_tiles] or nullptr = all */, int k_tiles, hipStream_t st,
                          const uint8_t* mat = nullptr /* [ceil(rows/256)]: 1 = write this 256-row tile of Gamma; the
                                                          others are generated inside the fused score GEMM */,
                          const int* vlist = nullptr, int n_vlist = 0 /* device list of the 4-row alpha blocks to visit */,
                          const int32_t* irr = nullptr /* [A][k_tiles]: K tiles written whatever `mat` says (the fused GEMM
                                                          for R > 1 reads the tiles it does not generate) */,
                          const int32_t* ctile = nullptr /* compact layout: 256-row tile t of Gamma is stored at tile ctile[t] */);
hipError_t launch_need_tiles(const uint8_t* nzA, int tiles_m, const uint8_t* nzB /* [AO+1][k_tiles] */, int AO, int V,
                             int k_tiles, uint8_t* need, hipStream_t st);

// dead[b][ao] = 1 iff supp(b) and supp(RTO[:,a,o,:]) are disjoint (P(o|b,a) == 0 exactly)
template <typename T>
hipError_t launch_dead(const T* bel, int ldb, int B, ModelView<T> mv,
                       const unsigned long long* nzBw /* [A*O][ceil(k_tiles/64)]: support tiles of RTO as bit words */,
                       int k_tiles, uint8_t* dead, int32_t* btl /* out: [B][k_tiles] non-zero tile lists */,
                       int32_t* btc /* out: [B] list lengths */, int* dead_count /* += dead triples, or nullptr */,
                       hipStream_t st, const uint8_t* rowflags = nullptr /* [.][k_tiles] per-row tile flags, if known */,
                       const int32_t* perm = nullptr /* belief b's row of rowflags (nullptr: b) */);

// Belief-side projection: bp[((o*A + a)*B + b)][s'] = gamma * sum_{(s,r): rs[s,a,r] = s'} b[b][s] * RTO[s,a,o,r]  (g = a*O+o), so
// that bp[g,b,:] . alpha[v,:] == b . Gamma[a,o,v,:] re-associated; accumulated in f64, rounded once to T.
// mag [B][G] (zeroed by the caller) receives sum_s' |bp| * amax[s'].  in_ptr / in_src: inverse transition lists.
template <typename T>
hipError_t launch_push_project(const T* bel, int ldb, int B, ModelView<T> mv, const int32_t* in_ptr,
                               const int32_t* in_src, double gamma, const T* amax, T* bp, int ldp, double* mag,
                               hipStream_t st, uint8_t* nzP = nullptr /* zero-tile map of the projected rows, [rows/256][ldp/32] */);
// rd[b][a] = b . ER[:,a] in f64 over the belief's non-zero tiles
template <typename T>
hipError_t launch_rdot(const T* bel, int ldb, int B, ModelView<T> mv, const int32_t* btl, const int32_t* btc, double* rd,
                       hipStream_t st);

// first-max argmax over the V columns [g*V, (g+1)*V) of each (row b, group g) of the score matrix;
// column G*V + g holds the magnitude score (b . Gamma of the max|alpha| row) that scales the tie
// window.
template <typename T>
hipError_t launch_argmax(SlabView<T> sv, int V, int G, int B, const uint8_t* dead, double tol_rel, double tol_abs,
                         const int* chain_steps /* device: K-tile steps of the longest f32 chain; used when tol_rel < 0 */,
                         int flag_all, int32_t* best_v, double* best_score, double* err, int32_t* queue, int* qcount,
                         hipStream_t st, double tol_extra = 0.0 /* added to the relative window: input rounding of a screen */);

// Work list of the refinement: entries with many near-tied candidates hand their (entry, candidate) pairs to a
// grid-wide pass instead of scoring them one block per entry (device memory; items_v == nullptr: all in-block).
struct RefineWork {
    int32_t* items_v = nullptr;          // [item_cap] candidate alpha index
    int32_t* items_slot = nullptr;       // [item_cap] slot of the entry (-1 = void item)
    double* scores = nullptr;            // [item_cap] exact score
    int32_t* slot_entry = nullptr;       // [slot_cap] queue entry (b * G + g) of the slot
    int32_t* slot_n = nullptr;           // [slot_cap] tiles in the slot's list
    int32_t* tiles = nullptr;            // [slot_cap][k_tiles] filtered tile list of the entry
    unsigned long long* emax = nullptr;  // [slot_cap] max exact score, order-preserving bit pattern (0 = none)
    int32_t* eidx = nullptr;             // [slot_cap] 0x7fffffff - (smallest candidate index attaining emax); 0 = none
    double* ib_val = nullptr;            // [slot_cap] best of the candidates the entry's own block scored
    int32_t* ib_idx = nullptr;           // [slot_cap] (0x7fffffff = none)
    int* cnt = nullptr;                  // [4] items reserved, slots reserved, entries handed to k_refine_split, (free)
    // Entries whose candidates the level-1 screen could not separate, re-decided by k_refine_split (16 blocks per entry)
    int32_t* q2_entry = nullptr;         // [q2_cap] queue entry
    int32_t* q2_n = nullptr;             // [q2_cap] surviving candidates (<= 8)
    int32_t* q2_cand = nullptr;          // [q2_cap][8] their alpha indices, ascending
    double* q2_part = nullptr;           // [q2_cap][16][8] partial exact scores per tile-list part
    int* q2_done = nullptr;              // [q2_cap] arrival counters (zero between launches)
    int q2_cap = 0;
    int item_cap = 0, slot_cap = 0;
    int defer_min = 8;                   // more candidates than this in a 256-column chunk go to the grid-wide passes
    size_t zero_bytes = 0;               // cnt, emax, eidx are one allocation starting at cnt: bytes to clear per launch
    // Slots below w_slot_cap take the GEMM path instead of items: the entry's fp64 weight row (the belief pushed
    // through (a, o), or the belief itself) against ALL alpha rows on the fp64 MFMA GEMM, then a first-max.
    int w_slot_cap = 0;
    double* W = nullptr;                 // [w_slot_cap][S_pad] weight rows
    double* Cx = nullptr;                // [w_slot_cap][V] exact scores
    uint8_t* nzW = nullptr;              // [ceil(w_slot_cap / 256)][S_pad / 32]
    int* klistW = nullptr;               // workspace of launch_gemm_nt_f64_bf32
    int* kcountW = nullptr;
    const int32_t* in_ptr = nullptr;     // inverse transition lists (engine.hip::build_inverse_lists)
    const int32_t* in_src = nullptr;
    // Level-1 screen of the per-entry pass (PROJ, fp32 engines whose Gamma rows are in HBM): the candidates' scores
    // are first re-summed in fp64 from the PROJECTED fp32 rows the GEMM read -- |Gamma_f32 - Gamma| <= l1_rel * Gamma(max|alpha|)
    // element-wise, so b . Gamma_f32 is within l1_rel * magnitude of the exact score -- and only candidates that this
    // ~100x tighter bound cannot separate are re-scored from alpha, RTO and the successor lists (R gathers per state).
    const float* gam = nullptr;          // [A*O*V + ...][ldg] projected rows, group-major (row g * V + v); nullptr = off
    int ldg = 0;
    double l1_rel = 0.0;                 // (R + 4) * 2^-24: R products, R adds, the scale, gamma's own rounding, slack
};

// fp64 re-decision of queued near-ties.  PROJ: scores are b . Gamma[a,o,v,:]; else b . alpha[v,:]
// T: type of the operands that are re-scored; TS: type of the score slabs the candidates are flagged from (TS = float,
// T = double: an fp64 engine behind its fp32 screen).
template <typename T, typename TS>
hipError_t launch_refine(bool proj, SlabView<TS> sv, int V, int G, int max_entries, const int32_t* queue,
                         const int* qcount, const T* bel, int ldb, const T* alpha, int lda, ModelView<T> mv,
                         double gamma, const int32_t* btl, const int32_t* btc /* belief tile lists or nullptr */,
                         const uint8_t* nzG /* PROJ: [G][k_tiles] support tiles of RTO per group, or nullptr */,
                         int32_t* best_v, double* best_score, double* err, int* cand_total /* += candidates, or nullptr */,
                         RefineWork work, hipStream_t st);

// The same in two parts, for callers that do not want to synchronise in between: the per-entry pass (what it deferred
// lands in counts_host[2], pinned, via the stream) and the pass over the deferred entries.
template <typename T, typename TS>
hipError_t launch_refine_scan(bool proj, SlabView<TS> sv, int V, int G, int max_entries, const int32_t* queue,
                              const int* qcount, const T* bel, int ldb, const T* alpha, int lda, ModelView<T> mv,
                              double gamma, const int32_t* btl, const int32_t* btc, const uint8_t* nzG, int32_t* best_v,
                              double* best_score, double* err, int* cand_total, RefineWork work, int* counts_host,
                              hipStream_t st);
template <typename T>
hipError_t launch_refine_deferred(bool proj, int V, int G, const T* bel, int ldb, const T* alpha, int lda, ModelView<T> mv,
                                  double gamma, int32_t* best_v, double* best_score, double* err, RefineWork work,
                                  int h_items, int h_slots, hipStream_t st);

// btl [B][k_tiles] / btc [B]: compact lists of each belief's non-zero 32-state tiles
template <typename T>
hipError_t launch_belief_tiles(const T* bel, int ldb, int B, int S, int k_tiles, int32_t* btl, int32_t* btc,
                               hipStream_t st);

// K4: val[b][a] = b.ER[:,a] + sum_o best_score[b][a][o]; action = first max; near-ties queued
// Gamma tail rows [A*O*V + A*O, +2A): ER[:,a] and |ER[:,a]|, so the score GEMM also yields b.ER[:,a]
template <typename T>
hipError_t launch_tail_rows(ModelView<T> mv, T* gam /* row 0 */, int64_t row0 /* first tail row */, int ldg, hipStream_t st,
                            const int32_t* ctile = nullptr);
template <typename T>
hipError_t launch_action(int B, ModelView<T> mv, SlabView<T> sv, int64_t rd_col0, double tol_rel, const int* chain_steps,
                         const double* best_score, const double* err, double* rdot, double* rdot_err, int32_t* action,
                         int32_t* aqueue, int* aqcount, hipStream_t st, double tol_extra = 0.0,
                         uint8_t* acand = nullptr /* [B][A] out: action inside the window of the best lower bound */);
constexpr int ACTION_SPLIT = 8;          // parts each exact dot of the action refinement is cut into (val_exact: [B][A][1+O][ACTION_SPLIT])
template <typename T>
hipError_t launch_refine_action(const T* bel, int ldb, int B, const T* alpha, int lda, ModelView<T> mv, double gamma,
                                const int32_t* btl, const int32_t* btc, const int32_t* aqueue, const int* aqcount,
                                const double* rdot, const double* rdot_err, const int32_t* best_v,
                                const double* best_score, const double* err, double* val_exact /* [B][A][1+O][ACTION_SPLIT] scratch */,
                                int32_t* action, hipStream_t st, const uint8_t* acand = nullptr /* launch_action's */);

// K3: out[u][s] = ER[s,a*] + sum_o gamma * sum_r rto[a*][o][r][s] * alpha[v*[b,a*,o]][rs[a*][r][s]], b = rows[u]
// (rows/n_rows on the device: only the unique (a*, v*) keys are assembled; nullptr = every belief)
template <typename T>
hipError_t launch_assemble(const T* alpha, int lda, ModelView<T> mv, double gamma, const int32_t* action,
                           const int32_t* best_v, const int32_t* rows, const int* n_rows, int max_rows, T* out, int ldo,
                           hipStream_t st);
// K6: key dedup in caller order (rep, uniq list, inverse index, count on the device)
hipError_t launch_dedup(int B, int A, int O, const int32_t* action, const int32_t* best_v, int32_t* rep, int32_t* uniq,
                        int32_t* inv, int32_t* slot, int* count, hipStream_t st);
template <typename T>
hipError_t launch_expand_rows(const T* uniq_rows, const int32_t* inv, T* full, int B, int S, hipStream_t st);

// K5: keep[c] = (b . alpha'[b]) > oldmax[b], c = caller index of engine row b
template <typename T>
hipError_t launch_keep(const T* bel, int ldb, const T* uniq_rows, int ldo, int B, int S, const double* oldmax,
                       const int32_t* inv, const int32_t* perm, uint8_t* keep, hipStream_t st);

// Batched Bayes step: out[b] = normalise(scatter of b[b,s]*RTO[s,a_b,o_b,r] to rs[s,a_b,r]) via inverse lists
// in_ptr [A][S+1], in_src [A][S*R] (entries s*R+r, ascending); unnorm [B][S] / mass [B] f64 scratch (mass zeroed).
// out_row (may be null = identity): result row of belief b, -1 = drop it (the simulator's done-filter).
template <typename T>
hipError_t launch_belief_update(const T* bel, int ldb, int B, ModelView<T> mv, const int32_t* in_ptr, const int32_t* in_src,
                                const int32_t* act, const int32_t* obs, const int32_t* out_row, double* unnorm,
                                double* mass, T* out, int ldo, hipStream_t st);

// one step of the belief walk: out64 [S] / out_store [S_pad] = normalised update of `base` (fp64 [S]) with (a, o);
// unnorm [S], partial [ceil(S_pad/256)] fp64 scratch; rto64: fp64 copy of RTO in mv's layout, or nullptr = use mv.rto
// one kernel per step: pushes belief i and writes belief i (normalising the previous raw result on the fly); see the kernel
template <typename T>
hipError_t launch_walk_fused(const double* plain_base, const double* prev_unnorm, const double* prev_partial, double* prev_out64,
                             T* prev_out_store, ModelView<T> mv, const double* rto64, const int32_t* in_ptr, const int32_t* in_src,
                             int a, int o, double* unnorm, double* partial, hipStream_t st);
template <typename T>
hipError_t launch_walk_finish(const double* unnorm, const double* partial, ModelView<T> mv, double* out64, T* out_store,
                              hipStream_t st);
template <typename T>
hipError_t launch_walk_step(const double* base, ModelView<T> mv, const double* rto64, const int32_t* in_ptr,
                            const int32_t* in_src, int a, int o, double* unnorm, double* partial, double* out64,
                            T* out_store, hipStream_t st);

// prune level 2: cnt[i] = #{j : alpha[j][s] >= alpha[i][s] for all s}
template <typename T>
hipError_t launch_dominated(const T* alpha, int lda, int V, int S, int* cnt, hipStream_t st);

}  // namespace pbvi

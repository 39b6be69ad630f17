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

template <typename T>
hipError_t launch_support(ModelView<T> mv, uint8_t* sup, hipStream_t st);

// K1-sparse: Gamma[(a*O+o)*V + v][s] = gamma * sum_r rto[a][o][r][s] * alpha[v][rs[a][r][s]]
template <typename T>
hipError_t launch_project(const T* alpha, int lda, int V, ModelView<T> mv, T gamma, T* gam, int ldg, hipStream_t st);

// dead[b][ao] = 1 iff supp(b) and supp(RTO[:,a,o,:]) are disjoint (P(o|b,a) == 0 exactly)
template <typename T>
hipError_t launch_dead(const T* bel, int ldb, int B, ModelView<T> mv, uint8_t* dead, hipStream_t st);

// first-max argmax over the first V columns of each (row b, group g) segment of the (split-K)
// score slabs; segments are vstride columns apart and, when vstride > V, column V of a segment
// holds the magnitude score (b . Gamma of the max|alpha| row) that scales the tie window.
template <typename T>
hipError_t launch_argmax(const T* slabs, int64_t slab_stride, int split_k, int ldc, int V, int vstride, int G, int B,
                         const uint8_t* dead, double tol_rel, double tol_abs, int flag_all,
                         int32_t* best_v, double* best_score, double* err, int32_t* queue, int* qcount,
                         hipStream_t st);

// fp64 re-decision of queued near-ties.  PROJ: scores are b . Gamma[a,o,v,:]; else b . alpha[v,:]
template <typename T>
hipError_t launch_refine(bool proj, const T* slabs, int64_t slab_stride, int split_k, int ldc, int V, int vstride,
                         int G, int max_entries, const int32_t* queue, const int* qcount, const T* bel, int ldb,
                         const T* alpha, int lda, ModelView<T> mv, double gamma, int32_t* best_v,
                         double* best_score, double* err, hipStream_t st);

// K4: val[b][a] = b.ER[:,a] + sum_o best_score[b][a][o]; action = first max; near-ties queued
template <typename T>
hipError_t launch_action(const T* bel, int ldb, int B, ModelView<T> mv, const double* best_score,
                         const double* err, double* rdot, int32_t* action, int32_t* aqueue, int* aqcount,
                         hipStream_t st);
template <typename T>
hipError_t launch_refine_action(const T* bel, int ldb, int B, const T* alpha, int lda, ModelView<T> mv,
                                double gamma, const int32_t* aqueue, const int* aqcount, const double* rdot,
                                const int32_t* best_v, double* best_score, double* err, int32_t* action,
                                hipStream_t st);

// K3: out[b][s] = ER[s,a*] + sum_o gamma * sum_r rto[a*][o][r][s] * alpha[v*[b,a*,o]][rs[a*][r][s]]
template <typename T>
hipError_t launch_assemble(const T* alpha, int lda, ModelView<T> mv, double gamma, const int32_t* action,
                           const int32_t* best_v, int B, T* out, int ldo, hipStream_t st);

// K5: keep[b] = (b . out[b]) > oldmax[b]
template <typename T>
hipError_t launch_keep(const T* bel, int ldb, const T* out, int ldo, int B, int S, const double* oldmax,
                       uint8_t* keep, hipStream_t st);

// prune level 2: cnt[i] = #{j : alpha[j][s] >= alpha[i][s] for all s}
template <typename T>
hipError_t launch_dominated(const T* alpha, int lda, int V, int S, int* cnt, hipStream_t st);

}  // namespace pbvi

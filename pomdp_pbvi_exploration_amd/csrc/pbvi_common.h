// Shared declarations for the gfx950 PBVI backup engine (internal; the public
// boundary is include/pbvi_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/pbvi_hip.h"

namespace pbvi {

void set_error(const std::string& msg);

// Score-GEMM tile geometry (gemm.hip).  Operands are padded by the engine so the
// kernels never branch on edges: rows to GEMM_BM / GEMM_BN, K to GEMM_BK.
constexpr int GEMM_BM = 256;
constexpr int GEMM_BN = 256;
constexpr int GEMM_BK = 32;

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// f32 score GEMM on MFMA with exact zero-tile skipping.
//   C[z][m][n] = sum over chunk z of the pair's K-tile list of A[m][k] * B[n][k]
// Both operands K-contiguous; M_pad/N_pad multiples of 256, K_pad a multiple of 32.  For every
// (m-tile, n-tile) pair the K tiles where both operands are non-zero are listed (klist/kcount);
// chunk z covers list entries [z*chunk_len, (z+1)*chunk_len).  nchunks[pair] tells the consumer
// how many slabs to sum (fixed order; pairs with an empty list have score exactly 0).
struct GemmPlan {
    int tiles_m, tiles_n, k_tiles, chunk_len, max_chunks, ldc;
    int64_t slab_stride;
    bool streamk;      // persistent stream-K scheduler (score GEMMs); else one block per (pair, chunk)
    int nblocks;       // stream-K grid = CUs of the device
    // Floats of C to allocate.  One block per (pair, chunk): max_chunks full slabs.  Stream-K: ONE full slab (every pair's
    // first chunk) + one 256 x 256 tile per block behind it -- a block continues at most one pair that an earlier block
    // began (the first pair of its share), so chunk z >= 1 of pair p lives in the tile of block first_block[p] + z
    // (9 full slabs were allocated for the <= nblocks continuation tiles that exist: 690 MB instead of 140 at C4).
    int64_t c_floats;
};
int set_gemm_force_dense(int enable);   // returns the previous setting (see gemm.hip)
int gemm_force_dense();
size_t streamk_workspace_ints(const GemmPlan& pl);   // ints of device workspace the stream-K plan needs
// B operand generated inside the stream-K GEMM instead of read from HBM (gemm.hip, "scheduler 2b"): row (g, v) of an
// n-tile with mat[tile] == 0 -- all of whose 256 rows belong to one group g -- is gamma * rto[g][s] * alpha[v][rs[g / O][s]]
// (one reachable state per (s, a)); tiles with mat[tile] != 0 are read from B as usual.
struct FusedB {
    const float* alpha;   // [V][lda]
    int lda;
    const int32_t* rs;    // [A][R][S_pad]
    const float* rto;     // [A*O][R][S_pad]
    int S_pad, O, V;      // groups of V rows each
    int R;                // reachable states per (s, a): 1 -> scheduler 2b, 2..7 -> scheduler 2c (gemm.hip)
    float gamma;
    const int32_t* ctile; // [tiles_n] device, or nullptr: B holds only the tiles with mat[tile] != 0, tile tn at tile ctile[tn]
    const uint8_t* mat;   // [tiles_n] device
    const int32_t* irr;   // [A][K_pad/32] device (int32: read with scalar loads): 1 = the K tile holds a 4-state chunk
                          // with non-consecutive successors (R = 1: gathered by the kernel; R > 1: that Gamma tile is
                          // projected by k_project and read from B)
};
// single_chunk: one K chunk per pair, i.e. slab 0 is the finished product (no split-K).
GemmPlan make_gemm_plan(int M_pad, int N_pad, int K_pad, bool single_chunk = false);
hipError_t launch_tile_nonzero_f32(const float* X, int ld, int rows_pad, int k_tiles, uint8_t* nz, hipStream_t stream);
// nzB: G > 0 -> per row group [G][k_tiles]; G == 0 -> per n-tile [batch][tiles_n][k_tiles]; nullptr -> dense.
// batch > 1 runs `batch` GEMMs sharing A (B, C and the tile lists advance by the batch strides).
hipError_t launch_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, const GemmPlan& pl,
                              const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int n_rows, int* klist,
                              int* kcount, int* nchunks, hipStream_t stream, int batch = 1,
                              int64_t batch_stride_b = 0, int64_t batch_stride_c = 0, int* streamk_ws = nullptr,
                              hipStream_t list_stream = nullptr, hipEvent_t list_event = nullptr,
                              hipEvent_t ev_before = nullptr, hipEvent_t ev_after = nullptr,   // around the GEMM kernel (non-stream-K)
                              const FusedB* fused = nullptr);
// fused: generate the B tiles that lie inside one row group (stream-K only); the others are read from B.
// list_stream + list_event: build the tile lists / stream-K plan on that stream (they need the zero maps only) and
// make `stream` wait for them before the GEMM kernel.
// streamk_ws layout: prefix[pairs+1], start_pair[nblocks], first_block[pairs], plan[2] = {steps per block, total steps}
// fp64 MFMA GEMM (gemm_f64.hip): C[M][N] = A[M][K_pad] * B[N][K_pad]^T, one dense result slab, zero-tile lists
// built from 32-column maps: nzA [ceil(M/256)][K_pad/32] (or nullptr), nzB [G+1][K_pad/32] per row group (or nullptr).
// klist: gemm_f64_klist_ints(M, N, K_pad/32) ints, kcount: gemm_f64_kcount_ints(...) ints of device workspace (the
// first gemm_f64_pairs(M, N) of them are the list lengths).
size_t gemm_f64_klist_ints(int M, int N, int kt32);
size_t gemm_f64_pairs(int M, int N);
int gemm_f64_bn(int N);                      // rows of B per column tile for a GEMM with N rows of B: 32, 64 or 128
size_t gemm_f64_kcount_ints(int M, int N, int kt32);
hipError_t launch_tile_nonzero_f64(const double* X, int ld, int rows, int kt32, uint8_t* nz, hipStream_t stream);
hipError_t launch_gemm_nt_f64(const double* A, int lda, int M, const double* B, int ldb, int N, double* C, int ldc,
                              int K_pad, const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int* klist,
                              int* kcount, hipStream_t stream, int split = 1 /* K parts per pair: part z is accumulated at */,
                              int64_t slab_stride = 0 /* C + z * slab_stride (workspace), then folded into C in slab order */);
int gemm_f64_split(int M, int N, int kt32);   // the split a score GEMM of this shape should use (1 once the grid fills the chip)
// same with an fp32 B operand widened exactly on the way into LDS (fp64 weights x fp32 alpha rows)
hipError_t launch_gemm_nt_f64_bf32(const double* A, int lda, int M, const float* B, int ldb, int N, double* C, int ldc,
                                   int K_pad, const uint8_t* nzA, int* klist, int* kcount, hipStream_t stream, int split = 1,
                                   int64_t slab_stride = 0);
// both operands fp32, widened exactly: fp64 scores of fp32 data (skinny value-max GEMMs of fp32 engines)
hipError_t launch_gemm_nt_f64_ff32(const float* A, int lda, int M, const float* B, int ldb, int N, double* C, int ldc, int K_pad,
                                   const uint8_t* nzA, int* klist, int* kcount, hipStream_t stream, int split = 1,
                                   int64_t slab_stride = 0);
// Any-size reference GEMM (small f64 problems, A/B checks): C[m][n], no padding requirements.
template <typename T>
hipError_t launch_gemm_nt_simple(const T* A, int lda, const T* B, int ldb, T* C, int ldc,
                                 int M, int N, int K, hipStream_t stream);


}  // namespace pbvi

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

// C[z][m][n] = sum over K-chunk z of A[m][k] * B[n][k]   (both operands K-contiguous).
// f32: hand-written MFMA kernel, split-K partial slabs (z = 0..split_k-1), fixed-order
// reduction left to the consumer.  M_pad/N_pad multiples of 256, K_pad multiple of 32.
hipError_t launch_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              int64_t slab_stride, int M_pad, int N_pad, int K_pad, int split_k,
                              hipStream_t stream);
// Any-size reference GEMM (used for f64 engines): C[m][n], no padding requirements.
template <typename T>
hipError_t launch_gemm_nt_simple(const T* A, int lda, const T* B, int ldb, T* C, int ldc,
                                 int M, int N, int K, hipStream_t stream);

// Pick the K-split so tiles * split_k fills the 256 CUs evenly.
int choose_split_k(int tiles_mn, int k_tiles);

}  // namespace pbvi

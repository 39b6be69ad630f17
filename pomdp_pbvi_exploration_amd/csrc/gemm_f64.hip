// fp64 score GEMM for gfx950: C[M x N] = A[M x K] * B[N x K]^T on v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s dense peak).
//
// f64 engines keep the reference's precision (src/pomdp.py:1494 runs in float64), so this GEMM needs no tie
// windows or refinement -- only speed.  Differences from the f32 kernel (gemm.hip), all because the f64 MFMA is
// four times slower per byte of operand: 128 x 128 x 16 tiles (64 KiB of LDS per block, two blocks per CU),
// plain global -> register -> LDS staging (LDS bandwidth is nowhere near the bound: 32 ds_read_b64 per 64 MFMAs
// of 64 cycles each), one block per tile pair with the whole K list (no split-K slabs), zero-tile lists derived
// from the same 32-state maps the f32 path uses.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "pbvi_common.h"

namespace pbvi {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int D_BM = 128, D_BN = 128, D_BK = 16;
constexpr int D_LD = D_BK + 2;          // LDS row stride in doubles: 144 B keeps the 16-row fragments off the same banks

// K lists in units of 32 columns (the granularity of the engine's zero maps):
//   nzA [ceil(M/256)][kt32]  non-zero map of A's 256-row blocks, or nullptr = dense
//   nzB [G+1][kt32]          per row group of B (rows [g*v_group, (g+1)*v_group)); rows >= G*v_group may touch any
//                            group or the extra row nzB[G]; nullptr = dense
__global__ void k_build_klists_f64(const uint8_t* __restrict__ nzA, const uint8_t* __restrict__ nzB, int G, int v_group,
                                   int n_rows, int tiles_m, int kt32, int* __restrict__ klist, int* __restrict__ kcount) {
    __shared__ int wcount[4];
    __shared__ int total;
    const int pair = blockIdx.x;
    const int tm = pair % tiles_m, tn = pair / tiles_m;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = tn * D_BN;
    int r1 = r0 + D_BN - 1;
    if (r1 >= n_rows) r1 = n_rows - 1;
    int g0 = 0, g1 = -1;
    if (nzB != nullptr && G > 0 && r0 < n_rows) {
        if (r1 >= G * v_group) {                        // magnitude / reward rows: any group, plus the extra support row
            g0 = 0;
            g1 = G;
        } else {
            g0 = r0 / v_group;
            g1 = r1 / v_group;
        }
    }
    if (tid == 0) total = 0;
    __syncthreads();
    for (int base = 0; base < kt32; base += 256) {
        const int kt = base + tid;
        int f = 0;
        if (kt < kt32 && r0 < n_rows && (nzA == nullptr || nzA[(int64_t)(tm * D_BM / 256) * kt32 + kt])) {
            if (nzB == nullptr) {
                f = 1;
            } else {
                for (int g = g0; g <= g1; ++g) f |= nzB[(int64_t)g * kt32 + kt];
            }
        }
        const unsigned long long mask = __ballot(f);
        if (lane == 0) wcount[wid] = __popcll(mask);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wid; ++w) off += wcount[w];
        if (f) klist[(int64_t)pair * kt32 + off + __popcll(mask & ((1ull << lane) - 1ull))] = kt;
        __syncthreads();
        if (tid == 0) total += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (tid == 0) kcount[pair] = total;
}

// Longest lists first: blocks are dispatched in index order, so handing out the heavy tile pairs first leaves the
// light ones to fill the tail (LPT scheduling).  One block; counting sort by list length (<= kt32), stable.
__global__ void k_order_pairs_f64(const int* __restrict__ kcount, int pairs, int kt32, int* __restrict__ hist /* [kt32+2] zeroed */,
                                  int* __restrict__ order) {
    const int tid = threadIdx.x;
    for (int p = tid; p < pairs; p += blockDim.x) atomicAdd(&hist[kt32 - kcount[p]], 1);      // bin 0 = longest
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b <= kt32; ++b) {
            const int c = hist[b];
            hist[b] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int p = tid; p < pairs; p += blockDim.x) order[atomicAdd(&hist[kt32 - kcount[p]], 1)] = p;
}

// non-zero map of a double matrix at the f32 path's granularity: nz[row block of 256][32-column tile]
__global__ void k_tile_nonzero_f64(const double* __restrict__ X, int ld, int rows, int kt32, uint8_t* __restrict__ nz) {
    const int tile = blockIdx.y;
    const int row = tile * 256 + threadIdx.x;
    for (int j = 0; j < 8; ++j) {
        const int kt = blockIdx.x * 8 + j;
        if (kt >= kt32) break;                          // block-uniform
        int f = 0;
        if (row < rows) {
            const f64x2* p = (const f64x2*)(X + (int64_t)row * ld + kt * 32);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f64x2 v = p[c];
                f |= (v[0] != 0.0) | (v[1] != 0.0);
            }
        }
        const int any = __syncthreads_or(f);
        if (threadIdx.x == 0) nz[(int64_t)tile * kt32 + kt] = any ? 1 : 0;
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f64x2 load2(const double* p) { return *(const f64x2*)p; }
__device__ __forceinline__ f64x2 load2(const float* p) {
    const f32x2 v = *(const f32x2*)p;
    return f64x2{(double)v[0], (double)v[1]};
}

// TB: element type of B (double, or float widened on the way into LDS -- exact -- for the refinement's re-scoring
// of fp32 alpha rows against fp64 weights)
template <typename TB>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_f64_mfma(const double* __restrict__ A, int lda, int M,
                                                             const TB* __restrict__ B, int ldb, int N,
                                                             double* __restrict__ C, int ldc, int tiles_m,
                                                             const int* __restrict__ klist,
                                                             const int* __restrict__ kcount, int kt32,
                                                             const int* __restrict__ order) {
    extern __shared__ double lds_f64[];                    // [2][128 * D_LD] A then [2][128 * D_LD] B: 72 KiB (dynamic:
    double (*As)[D_BM * D_LD] = reinterpret_cast<double (*)[D_BM * D_LD]>(lds_f64);                 // above the 64 KiB
    double (*Bs)[D_BN * D_LD] = reinterpret_cast<double (*)[D_BN * D_LD]>(lds_f64 + 2 * D_BM * D_LD);   // static limit)
    const int pair = order ? order[blockIdx.x] : (int)blockIdx.x;
    const int tm = pair % tiles_m, tn = pair / tiles_m;
    const int m0 = tm * D_BM, n0 = tn * D_BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;                 // 2 x 2 waves, 64 x 64 each
    const int* kl = klist + (int64_t)pair * kt32;
    const int nsteps = kcount[pair] * 2;                   // 16-column steps

    // staging: instruction c of thread t moves 16 bytes of row c*32 + t/8, column chunk t%8 -- a wave instruction
    // covers eight whole 128-byte row segments (full cache lines)
    const int srow = tid >> 3, scol = (tid & 7) * 2;
    const double* ap[4];
    const TB* bp[4];
    bool a_ok[4], b_ok[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r = c * 32 + srow;
        a_ok[c] = m0 + r < M;
        b_ok[c] = n0 + r < N;
        ap[c] = A + (int64_t)(a_ok[c] ? m0 + r : 0) * lda + scol;
        bp[c] = B + (int64_t)(b_ok[c] ? n0 + r : 0) * ldb + scol;
    }
    f64x2 ra[4], rb[4];
    auto fetch = [&](int step) {
        const int k0 = kl[step >> 1] * 32 + (step & 1) * D_BK;
        const f64x2 z = {0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            ra[c] = a_ok[c] ? *(const f64x2*)(ap[c] + k0) : z;
            rb[c] = b_ok[c] ? load2(bp[c] + k0) : z;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            *(f64x2*)(&As[buf][(c * 32 + srow) * D_LD + scol]) = ra[c];
            *(f64x2*)(&Bs[buf][(c * 32 + srow) * D_LD + scol]) = rb[c];
        }
    };

    f64x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    if (nsteps > 0) {
        fetch(0);
        stash(0);
    }
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;              // fragment row / k of this lane
    for (int step = 0; step < nsteps; ++step) {
        const int buf = step & 1;
        const bool more = step + 1 < nsteps;
        if (more) fetch(step + 1);                         // global loads fly under the MFMAs
        const double* as = &As[buf][(wm * 64 + fr) * D_LD + fk];
        const double* bs = &Bs[buf][(wn * 64 + fr) * D_LD + fk];
#pragma unroll
        for (int ks = 0; ks < D_BK / 4; ++ks) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = as[i * 16 * D_LD + ks * 4];
                b[i] = bs[i * 16 * D_LD + ks * 4];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);                          // the other buffer was last read one barrier ago
        __syncthreads();
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + (lane >> 4) + 4 * r;
                if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[i][j][r];
            }
        }
}

hipError_t launch_tile_nonzero_f64(const double* X, int ld, int rows, int kt32, uint8_t* nz, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    dim3 grid((kt32 + 7) / 8, (rows + 255) / 256);
    hipLaunchKernelGGL(k_tile_nonzero_f64, grid, dim3(256), 0, stream, X, ld, rows, kt32, nz);
    return hipGetLastError();
}

size_t gemm_f64_klist_ints(int M, int N, int kt32) {
    return (size_t)((M + D_BM - 1) / D_BM) * ((N + D_BN - 1) / D_BN) * kt32;
}
// kcount workspace: [pairs] list lengths, [pairs] dispatch order, [kt32 + 2] histogram
size_t gemm_f64_pairs(int M, int N) { return (size_t)((M + D_BM - 1) / D_BM) * ((N + D_BN - 1) / D_BN); }
size_t gemm_f64_kcount_ints(int M, int N, int kt32) { return 2 * gemm_f64_pairs(M, N) + (size_t)kt32 + 2; }

template <typename TB>
static hipError_t launch_gemm_nt_f64_t(const double* A, int lda, int M, const TB* B, int ldb, int N, double* C, int ldc,
                                       int K_pad, const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int* klist,
                                       int* kcount, hipStream_t stream) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (K_pad % 32 != 0) return hipErrorInvalidValue;
    const int kt32 = K_pad / 32;
    const int tiles_m = (M + D_BM - 1) / D_BM, tiles_n = (N + D_BN - 1) / D_BN;
    const int64_t pairs = (int64_t)tiles_m * tiles_n;
    if (pairs > 0x7fffffff) return hipErrorInvalidValue;
    const int force_dense = gemm_force_dense();          // benchmark / debug: every tile listed
    hipLaunchKernelGGL(k_build_klists_f64, dim3((unsigned)pairs), dim3(256), 0, stream, force_dense ? nullptr : nzA,
                       force_dense ? nullptr : nzB, G, v_group, N, tiles_m, kt32, klist, kcount);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    constexpr size_t lds_bytes = (size_t)2 * (D_BM + D_BN) * D_LD * sizeof(double);
    static bool attr_done = false;
    if (!attr_done) {
        e = hipFuncSetAttribute((const void*)k_gemm_nt_f64_mfma<TB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    int* order = kcount + pairs;
    int* hist = order + pairs;
    if ((e = hipMemsetAsync(hist, 0, (size_t)(kt32 + 2) * sizeof(int), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_order_pairs_f64, dim3(1), dim3(1024), 0, stream, kcount, (int)pairs, kt32, hist, order);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(k_gemm_nt_f64_mfma<TB>, dim3((unsigned)pairs), dim3(256), lds_bytes, stream, A, lda, M, B, ldb, N, C,
                       ldc, tiles_m, klist, kcount, kt32, order);
    return hipGetLastError();
}

hipError_t launch_gemm_nt_f64(const double* A, int lda, int M, const double* B, int ldb, int N, double* C, int ldc,
                              int K_pad, const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int* klist,
                              int* kcount, hipStream_t stream) {
    return launch_gemm_nt_f64_t<double>(A, lda, M, B, ldb, N, C, ldc, K_pad, nzA, nzB, G, v_group, klist, kcount, stream);
}

hipError_t launch_gemm_nt_f64_bf32(const double* A, int lda, int M, const float* B, int ldb, int N, double* C, int ldc,
                                   int K_pad, const uint8_t* nzA, int* klist, int* kcount, hipStream_t stream) {
    return launch_gemm_nt_f64_t<float>(A, lda, M, B, ldb, N, C, ldc, K_pad, nzA, nullptr, 1, N, klist, kcount, stream);
}

}  // namespace pbvi

// Score GEMM for the PBVI backup on gfx950 (CDNA4).
//
//   C[z][m][n] = sum_{k in chunk z} A[m][k] * B[n][k]
//
// In the backup A = belief block [B][S], B = Gamma [(a,o,v)][S] (reference:
// xp.tensordot(belief_array, gamma_a_o_t, (1,3)), src/pomdp.py:1495).  Both operands
// are K-contiguous, exactly the layouts the reference keeps them in, so no transpose
// copy is ever made (NumPy's tensordot makes one internally).
//
// f32 kernel: v_mfma_f32_32x32x2_f32 (exact f32 fma chains, 64 FLOP/clk/SIMD = the
// 157 TFLOP/s fp32 matrix peak).  256x256x32 tiles, 8 waves (2 per SIMD) laid out
// 2(M) x 4(N), each wave a 128x64 sub-tile = 4x2 MFMA blocks = 128 accumulator VGPRs.
// Operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into a double buffer;
// the 128-byte LDS rows are XOR-swizzled on the SOURCE address so the ds_read_b128
// fragment reads are bank-conflict free.  K is permuted inside each 8-wide group
// (lane half h takes k = 8g+4h+j for MFMA step j) so one ds_read_b128 feeds four
// MFMAs; the permutation is applied to both operands, so the sum is unchanged.
// Split-K partial slabs keep 256 CUs evenly loaded when the tile count is not a
// multiple of 256 and shorten the f32 accumulation chains; the consumer reduces the
// slabs in a fixed order (deterministic).  Workgroup ids are remapped so the M-tiles
// that share one Gamma tile run on the same XCD (one L2 fill per Gamma tile).
#include "pbvi_common.h"

namespace pbvi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE_FLOATS = 256 * GEMM_BK;          // one operand tile: 256 rows x 32 k = 32 KiB
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * 4;  // 2 buffers x (A tile + B tile) = 128 KiB

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__global__ __launch_bounds__(512) void k_gemm_nt_f32_mfma(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int64_t slab_stride, int tiles_m, int tiles_n, int k_tiles, int split_k) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;

    // XCD-aware, bijective remap: blocks b and b+8 share an XCD; give each XCD a
    // contiguous run of logical ids so its L2 sees each Gamma tile once.
    int L;
    {
        const int total = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = total >> 3, r = total & 7;
        const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        L = base + (bid >> 3);
    }
    const int tm = L % tiles_m;
    const int rest = L / tiles_m;
    const int tn = rest % tiles_n;
    const int z = rest / tiles_n;
    const int kt0 = (int)(((unsigned)z * (unsigned)k_tiles) / (unsigned)split_k);
    const int kt1 = (int)(((unsigned)(z + 1) * (unsigned)k_tiles) / (unsigned)split_k);

    const float* Ablk = A + (int64_t)tm * 256 * lda;
    const float* Bblk = B + (int64_t)tn * 256 * ldb;

    // staging: 2048 16-byte chunks per operand tile, 4 per thread.  LDS image is
    // lane-linear (chunk c at byte 16c); physical chunk p of row r holds logical
    // chunk p ^ ((r>>1)&7).
    int srow[4], scol[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 512 + tid;
        const int row = c >> 3, pc = c & 7;
        srow[it] = row;
        scol[it] = (pc ^ ((row >> 1) & 7)) * 4;
    }
    auto stage = [&](int buf, int kt) {
        float* la = lds + buf * 2 * TILE_FLOATS;
        float* lb = la + TILE_FLOATS;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int wave_chunk = it * 512 + wid * 64;        // wave-uniform LDS base (M0)
            glds16(Ablk + (int64_t)srow[it] * lda + kt * GEMM_BK + scol[it], la + wave_chunk * 4);
            glds16(Bblk + (int64_t)srow[it] * ldb + kt * GEMM_BK + scol[it], lb + wave_chunk * 4);
        }
    };

    const int i = lane & 31, h = lane >> 5;
    const int wr = wid >> 2, wc = wid & 3;
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // fragment addresses (floats) for g = 0; chunk index is XORed per g below
    int a_row[4], b_row[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) a_row[mi] = wr * 128 + mi * 32 + i;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b_row[ni] = wc * 64 + ni * 32 + i;

    auto compute = [&](int buf) {
        const float* la = lds + buf * 2 * TILE_FLOATS;
        const float* lb = la + TILE_FLOATS;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af[4], bf[2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int pc = (2 * g + h) ^ ((a_row[mi] >> 1) & 7);
                af[mi] = *(const f32x4*)(la + a_row[mi] * GEMM_BK + pc * 4);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int pc = (2 * g + h) ^ ((b_row[ni] >> 1) & 7);
                bf[ni] = *(const f32x4*)(lb + b_row[ni] * GEMM_BK + pc * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
        }
    };

    if (kt0 < kt1) {
        stage(0, kt0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            if (kt + 1 < kt1) stage(buf ^ 1, kt + 1);
            compute(buf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
    }

    // epilogue: lane holds column (lane&31); register e holds row (e&3) + 8*(e>>2) + 4*(lane>>5)
    float* Cz = C + (int64_t)z * slab_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = tn * 256 + wc * 64 + ni * 32 + i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = tm * 256 + wr * 128 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                Cz[(int64_t)row * ldc + col] = acc[mi][ni][e];
            }
        }
}

int choose_split_k(int tiles_mn, int k_tiles) {
    const int cus = 256;
    int best = 1;
    double best_eff = 0.0;
    const int smax = k_tiles < 64 ? (k_tiles < 1 ? 1 : k_tiles) : 64;
    for (int s = 1; s <= smax; ++s) {
        if (s > 1 && k_tiles / s < 4) break;                 // keep chunks long enough to pipeline
        const int64_t total = (int64_t)tiles_mn * s;
        const int64_t rounds = (total + cus - 1) / cus;
        const double eff = (double)total / (double)(rounds * cus);
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best = s;
        }
        if (eff >= 0.95) break;                               // smallest split that fills the chip
    }
    return best;
}

hipError_t launch_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                              int64_t slab_stride, int M_pad, int N_pad, int K_pad, int split_k,
                              hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)k_gemm_nt_f32_mfma,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int tiles_m = M_pad / GEMM_BM, tiles_n = N_pad / GEMM_BN, k_tiles = K_pad / GEMM_BK;
    const int64_t total = (int64_t)tiles_m * tiles_n * split_k;
    if (total <= 0 || total > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_gemm_nt_f32_mfma, dim3((unsigned)total), dim3(512), GEMM_LDS_BYTES, stream,
                       A, lda, B, ldb, C, ldc, slab_stride, tiles_m, tiles_n, k_tiles, split_k);
    return hipGetLastError();
}

// --------------------------------------------------------------------------- //
// Plain tiled GEMM for any T / any shape (f64 engines, small models).
// --------------------------------------------------------------------------- //
template <typename T>
__global__ void k_gemm_nt_simple(const T* __restrict__ A, int lda, const T* __restrict__ B, int ldb,
                                 T* __restrict__ C, int ldc, int M, int N, int K) {
    __shared__ T As[16][17];
    __shared__ T Bs[16][17];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
    T acc = 0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int ka = k0 + tx;
        const int ma = blockIdx.y * 16 + ty, nb = blockIdx.x * 16 + ty;
        As[ty][tx] = (ma < M && ka < K) ? A[(int64_t)ma * lda + ka] : T(0);
        Bs[ty][tx] = (nb < N && ka < K) ? B[(int64_t)nb * ldb + ka] : T(0);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc += As[ty][kk] * Bs[tx][kk];
        __syncthreads();
    }
    if (m < M && n < N) C[(int64_t)m * ldc + n] = acc;
}

template <typename T>
hipError_t launch_gemm_nt_simple(const T* A, int lda, const T* B, int ldb, T* C, int ldc, int M, int N,
                                 int K, hipStream_t stream) {
    if (M <= 0 || N <= 0) return hipSuccess;
    dim3 grid((N + 15) / 16, (M + 15) / 16), block(16, 16);
    hipLaunchKernelGGL(k_gemm_nt_simple<T>, grid, block, 0, stream, A, lda, B, ldb, C, ldc, M, N, K);
    return hipGetLastError();
}

template hipError_t launch_gemm_nt_simple<float>(const float*, int, const float*, int, float*, int, int, int, int, hipStream_t);
template hipError_t launch_gemm_nt_simple<double>(const double*, int, const double*, int, double*, int, int, int, int, hipStream_t);

}  // namespace pbvi

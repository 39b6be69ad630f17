// fp64 score GEMM for gfx950: C[M x N] = A[M x K] * B[N x K]^T on v_mfma_f64_16x16x4_f64 (78.6 TFLOP/s dense peak).
//
// f64 engines keep the reference's precision (src/pomdp.py:1494 runs in float64), so this GEMM needs no tie
// windows or refinement -- only speed.  Differences from the f32 kernel (gemm.hip), all because the f64 MFMA is
// four times slower per byte of operand: 128 x 128 x 16 tiles (64 KiB of LDS per block, two blocks per CU),
// plain global -> register -> LDS staging (LDS bandwidth is nowhere near the bound: 32 ds_read_b64 per 64 MFMAs
// of 64 cycles each), zero-tile lists derived from the same 32-state maps the f32 path uses.  One block per
// (tile pair, K split): a grid of fewer pairs than the chip has block slots (the solve loop's skinny GEMMs: 100 new
// beliefs x 10^4 alpha rows = 79 pairs, each walking several hundred K tiles) is split along K into `split` equal
// parts of every pair's list, one partial slab per part, summed in fixed order by the readers (SlabView::at).
#include <hip/hip_runtime.h>

#include <algorithm>
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
                                   int n_rows, int tiles_m, int kt32, int* __restrict__ klist, int* __restrict__ kcount,
                                   int split, int* __restrict__ kpart /* [pairs][split + 1] list offsets of the K parts */,
                                   int bn /* rows of B per column tile */) {
    __shared__ int wcount[4], wcount_a[4];
    __shared__ int total, total_a, n_a;
    __shared__ int part_start[33];
    const int pair = blockIdx.x;
    const int tm = pair % tiles_m, tn = pair / tiles_m;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = tn * bn;
    int r1 = r0 + bn - 1;
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
    const uint8_t* za = nzA ? nzA + (int64_t)(tm * D_BM / 256) * kt32 : nullptr;
    // K parts: boundaries from the support of the A ROW BLOCK alone (its non-zero tiles dealt evenly over the parts), so
    // every column of a row -- whichever column tile it sits in -- is summed in the same association: identical B rows
    // keep bit-identical scores and ties keep going to the lower index.
    if (tid == 0) {
        total = 0;
        total_a = 0;
        n_a = 0;
    }
    if (tid <= split) part_start[tid] = 0x7fffffff;
    __syncthreads();
    if (split > 1) {
        int c = 0;
        for (int kt = tid; kt < kt32; kt += 256) c += za ? (za[kt] != 0) : 1;
        atomicAdd(&n_a, c);
        __syncthreads();
    }
    const int na = n_a > 0 ? n_a : 1;
    for (int base = 0; base < kt32; base += 256) {
        const int kt = base + tid;
        int f = 0, fa = 0;
        if (kt < kt32) fa = za ? (za[kt] != 0) : 1;
        if (fa && r0 < n_rows) {
            if (nzB == nullptr) {
                f = 1;
            } else {
                for (int g = g0; g <= g1; ++g) f |= nzB[(int64_t)g * kt32 + kt];
            }
        }
        const unsigned long long mask = __ballot(f), mask_a = __ballot(fa);
        if (lane == 0) {
            wcount[wid] = __popcll(mask);
            wcount_a[wid] = __popcll(mask_a);
        }
        __syncthreads();
        int off = total, off_a = total_a;
        for (int w = 0; w < wid; ++w) {
            off += wcount[w];
            off_a += wcount_a[w];
        }
        if (f) {
            const int pos = off + __popcll(mask & ((1ull << lane) - 1ull));
            klist[(int64_t)pair * kt32 + pos] = kt;
            if (split > 1) {
                const int rank_a = off_a + __popcll(mask_a & ((1ull << lane) - 1ull));     // A-non-zero tiles before kt
                atomicMin(&part_start[(int)((int64_t)rank_a * split / na)], pos);
            }
        }
        __syncthreads();
        if (tid == 0) {
            total += wcount[0] + wcount[1] + wcount[2] + wcount[3];
            total_a += wcount_a[0] + wcount_a[1] + wcount_a[2] + wcount_a[3];
        }
        __syncthreads();
    }
    if (tid == 0) {
        kcount[pair] = total;
        if (split > 1) {      // empty parts start where the next non-empty one does
            int nxt = total;
            kpart[(int64_t)pair * (split + 1) + split] = total;
            for (int z = split - 1; z >= 0; --z) {
                if (part_start[z] != 0x7fffffff) nxt = part_start[z];
                kpart[(int64_t)pair * (split + 1) + z] = nxt;
            }
        }
    }
}

// Longest lists first: blocks are dispatched in index order, so handing out the heavy tile pairs first leaves the
// light ones to fill the tail (LPT scheduling).  One block; counting sort by list length (<= kt32); the order inside a
// bin is whatever the atomics give -- it only decides which of two equally long pairs is dispatched first.
__global__ void k_order_pairs_f64(const int* __restrict__ kcount, int pairs, int kt32, int* __restrict__ hist /* [kt32+2] zeroed */,
                                  int* __restrict__ order) {
    constexpr int LBINS = 4096;
    __shared__ int lhist[LBINS];
    const int tid = threadIdx.x;
    int* h = hist;
    if (kt32 + 1 <= LBINS) {                              // the histogram in LDS: the lengths cluster, and global atomics
        for (int b = tid; b <= kt32; b += blockDim.x) lhist[b] = 0;      // on a handful of addresses serialise
        h = lhist;
        __syncthreads();
    }
    for (int p = tid; p < pairs; p += blockDim.x) atomicAdd(&h[kt32 - kcount[p]], 1);      // bin 0 = longest
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b <= kt32; ++b) {
            const int c = h[b];
            h[b] = run;
            run += c;
        }
    }
    __syncthreads();
    for (int p = tid; p < pairs; p += blockDim.x) order[atomicAdd(&h[kt32 - kcount[p]], 1)] = p;
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

// TA / TB: element types of A and B (double, or float widened on the way into LDS -- exact: fp32 alpha rows against the
// refinement's fp64 weights, and both operands fp32 for the skinny value-max GEMMs of fp32 engines, whose products are
// then exact and whose sums are fp64).
// BN: rows of B per block.  128 for score GEMMs proper; 64 / 32 for the solve loop's skinny ones (compute_change scores
// the whole belief store against the few dozen alpha rows an expansion added: with a 128-wide tile three quarters of
// the MFMA work is padding and the kernel is MFMA-bound on it; with 32 columns the same launch is a streaming read of
// the store).  Waves: 2 x 2 of 64 x 64 (BN = 128), 4 x 1 of 32 x BN otherwise.
template <int BN>
struct F64Tile {
    static constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN;
    static constexpr int MI = D_BM / WM / 16, NJ = BN / WN / 16;
    static constexpr int BCH = (BN + 31) / 32;             // staging chunks of B per thread (A: 4); BN = 16, 48: the last
                                                           // chunk's upper 16 rows land in LDS padding rows, never read
    static constexpr int BROWS = BCH * 32;                 // rows of B staged (>= BN)
    static constexpr size_t lds_bytes = (size_t)2 * (D_BM + BROWS) * D_LD * sizeof(double);
    static constexpr int min_blocks = 2;                   // per CU
    static constexpr int PD = BN == 128 ? 1 : 4;           // register prefetch depth in K steps (see the kernel)
};

template <typename TA, typename TB, int BN>
__global__ __launch_bounds__(256, F64Tile<BN>::min_blocks) void k_gemm_nt_f64_mfma(
    const TA* __restrict__ A, int lda, int M, const TB* __restrict__ B, int ldb, int N, double* __restrict__ C, int ldc,
    int tiles_m, const int* __restrict__ klist, const int* __restrict__ kcount, int kt32, const int* __restrict__ order,
    int split, int64_t slab_stride, const int* __restrict__ kpart) {
    using TL = F64Tile<BN>;
    extern __shared__ double lds_f64[];                    // [2][128 * D_LD] A then [2][BN * D_LD] B: 72 KiB at BN = 128
    double (*As)[D_BM * D_LD] = reinterpret_cast<double (*)[D_BM * D_LD]>(lds_f64);       // (dynamic: above the 64 KiB
    double (*Bs)[TL::BROWS * D_LD] = reinterpret_cast<double (*)[TL::BROWS * D_LD]>(lds_f64 + 2 * D_BM * D_LD);   // static limit)
    const int pidx = (int)blockIdx.x / split, z = (int)blockIdx.x - pidx * split;
    const int pair = order ? order[pidx] : pidx;
    const int tm = pair % tiles_m, tn = pair / tiles_m;
    const int m0 = tm * D_BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / TL::WN, wn = wid % TL::WN;
    const int k_begin = split > 1 ? kpart[(int64_t)pair * (split + 1) + z] : 0;
    const int k_end = split > 1 ? kpart[(int64_t)pair * (split + 1) + z + 1] : kcount[pair];
    const int* kl = klist + (int64_t)pair * kt32 + k_begin;
    const int nsteps = (k_end - k_begin) * 2;              // 16-column steps (an empty part writes zeros)
    C += (int64_t)z * slab_stride;

    // staging: instruction c of thread t moves 16 bytes of row c*32 + t/8, column chunk t%8 -- a wave instruction
    // covers eight whole 128-byte row segments (full cache lines)
    const int srow = tid >> 3, scol = (tid & 7) * 2;
    const TA* ap[4];
    const TB* bp[TL::BCH];
    bool a_ok[4], b_ok[TL::BCH];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r = c * 32 + srow;
        a_ok[c] = m0 + r < M;
        ap[c] = A + (int64_t)(a_ok[c] ? m0 + r : 0) * lda + scol;
    }
#pragma unroll
    for (int c = 0; c < TL::BCH; ++c) {
        const int r = c * 32 + srow;
        b_ok[c] = n0 + r < N;
        bp[c] = B + (int64_t)(b_ok[c] ? n0 + r : 0) * ldb + scol;
    }
    // Register prefetch ring, PD steps deep: the loads of step s + PD are issued when step s's operands have moved into
    // LDS.  One step ahead is enough while a step's MFMAs (BN = 128: 4096 cycles per wave) outlast a global load; the
    // skinny tiles spend 1024 / 2048 cycles per step against a load latency of several thousand, and their operand is a
    // column slab of a multi-GB row store (128 rows x 128 bytes per step, one DRAM page each) -- they need several
    // steps of loads in flight to stream it.
    constexpr int PD = TL::PD;
    f64x2 ra[PD][4], rb[PD][TL::BCH];
    auto fetch = [&](int step, f64x2 (&qa)[4], f64x2 (&qb)[TL::BCH]) {
        // No predication: rows past M / N read row 0 (their pointers were clamped) and feed accumulator rows / columns that
        // are never stored, and a step past the end re-reads the last one.  A branch around a load would make the
        // compiler drain the whole ring (vmcnt(0)) before every LDS write instead of waiting for the oldest slot only.
        step = step < nsteps ? step : nsteps - 1;
        const int k0 = kl[step >> 1] * 32 + (step & 1) * D_BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) qa[c] = load2(ap[c] + k0);
#pragma unroll
        for (int c = 0; c < TL::BCH; ++c) qb[c] = load2(bp[c] + k0);
    };
    auto stash = [&](int buf, const f64x2 (&qa)[4], const f64x2 (&qb)[TL::BCH]) {
#pragma unroll
        for (int c = 0; c < 4; ++c) *(f64x2*)(&As[buf][(c * 32 + srow) * D_LD + scol]) = qa[c];
#pragma unroll
        for (int c = 0; c < TL::BCH; ++c) *(f64x2*)(&Bs[buf][(c * 32 + srow) * D_LD + scol]) = qb[c];
    };

    f64x4 acc[TL::MI][TL::NJ];
#pragma unroll
    for (int i = 0; i < TL::MI; ++i)
#pragma unroll
        for (int j = 0; j < TL::NJ; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};

    // step s lives in ring slot s % PD.  Top of step s: LDS buffer s & 1 holds step s, the ring holds steps s+1 .. s+PD.
    if (nsteps > 0) {
        fetch(0, ra[0], rb[0]);
        stash(0, ra[0], rb[0]);
#pragma unroll
        for (int u = 1; u <= PD; ++u) fetch(u, ra[u % PD], rb[u % PD]);
    }
    __syncthreads();
    const int fr = lane & 15, fk = lane >> 4;              // fragment row / k of this lane
    for (int s0 = 0; s0 < nsteps; s0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int step = s0 + u;
            if (step >= nsteps) break;
            const int buf = step & 1;
            const double* as = &As[buf][(wm * (TL::MI * 16) + fr) * D_LD + fk];
            const double* bs = &Bs[buf][(wn * (TL::NJ * 16) + fr) * D_LD + fk];
#pragma unroll
            for (int ks = 0; ks < D_BK / 4; ++ks) {
                double a[TL::MI], b[TL::NJ];
#pragma unroll
                for (int i = 0; i < TL::MI; ++i) a[i] = as[i * 16 * D_LD + ks * 4];
#pragma unroll
                for (int j = 0; j < TL::NJ; ++j) b[j] = bs[j * 16 * D_LD + ks * 4];
#pragma unroll
                for (int i = 0; i < TL::MI; ++i)
#pragma unroll
                    for (int j = 0; j < TL::NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            const int nslot = (u + 1) % PD;                // constant after unrolling
            if (step + 1 < nsteps) stash(buf ^ 1, ra[nslot], rb[nslot]);      // the other buffer was last read one barrier ago
            fetch(step + 1 + PD, ra[nslot], rb[nslot]);
            __syncthreads();
        }
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int i = 0; i < TL::MI; ++i)
#pragma unroll
        for (int j = 0; j < TL::NJ; ++j) {
            const int col = n0 + wn * (TL::NJ * 16) + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * (TL::MI * 16) + i * 16 + (lane >> 4) + 4 * r;
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

// column-tile width of a GEMM with N rows of B
int gemm_f64_bn(int N) {
    static const int forced = getenv("PBVI_F64_BN") ? atoi(getenv("PBVI_F64_BN")) : 0;      // debug / A-B only: 32, 64, 128
    if (forced == 16 || forced == 32 || forced == 48 || forced == 64 || forced == 128) return forced;
    return N <= 16 ? 16 : N <= 32 ? 32 : N <= 48 ? 48 : N <= 64 ? 64 : D_BN;
}
size_t gemm_f64_pairs(int M, int N) {
    const int bn = gemm_f64_bn(N);
    return (size_t)((M + D_BM - 1) / D_BM) * ((N + bn - 1) / bn);
}
size_t gemm_f64_klist_ints(int M, int N, int kt32) { return gemm_f64_pairs(M, N) * kt32; }
// kcount workspace: [pairs] list lengths, [pairs] dispatch order, [kt32 + 2] histogram, [pairs][33] K-part offsets
size_t gemm_f64_kcount_ints(int M, int N, int kt32) { return 35 * gemm_f64_pairs(M, N) + (size_t)kt32 + 2; }
// K split of a score GEMM: enough blocks for two per CU, parts of at least ~8 listed tiles when the lists are full
int gemm_f64_split(int M, int N, int kt32) {
    static const int forced = getenv("PBVI_F64_SPLIT") ? atoi(getenv("PBVI_F64_SPLIT")) : 0;      // debug / A-B only
    if (forced > 0) return forced;
    const int64_t pairs = (int64_t)gemm_f64_pairs(M, N);
    if (pairs <= 0 || pairs >= 384) return 1;
    // ~4 blocks per block slot of the chip (2 per CU): a grid of 1.2 x the slots would run as two rounds, the second one
    // nearly empty; with short blocks the tail is a small part of the launch
    int64_t z = (2048 + pairs - 1) / pairs;
    z = std::min<int64_t>(z, std::max(1, kt32 / 8));
    z = std::min<int64_t>(z, std::max<int64_t>(1, ((int64_t)64 << 20) / ((int64_t)M * N * 8)));      // partial slabs: <= 64 MiB
    return (int)std::min<int64_t>(z, 32);
}

// C[0] += C[1] + ... in slab order (fixed association): the K parts of a split GEMM folded into the first slab, so the
// readers see one matrix.  (Left to them, a row-per-wave reader like k_argmax walks split x N values per wave -- for a
// 100-row product that is the whole chip waiting on 100 waves.)
__global__ void k_fold_slabs(double* __restrict__ C, int ldc, int M, int N, int split, int64_t slab_stride) {
    const int64_t n = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t off = (i / N) * ldc + i % N;
        double acc = C[off];
        for (int z = 1; z < split; ++z) acc += C[off + z * slab_stride];
        C[off] = acc;
    }
}

template <typename TA, typename TB, int BN>
static hipError_t launch_f64_tile(const TA* A, int lda, int M, const TB* B, int ldb, int N, double* C, int ldc, int tiles_m,
                                  int64_t pairs, int* klist, int* kcount, int kt32, int* order, int split, int64_t slab_stride,
                                  int* kpart, hipStream_t stream) {
    static bool attr_done = false;
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute((const void*)k_gemm_nt_f64_mfma<TA, TB, BN>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)F64Tile<BN>::lds_bytes);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    hipLaunchKernelGGL((k_gemm_nt_f64_mfma<TA, TB, BN>), dim3((unsigned)(pairs * split)), dim3(256), F64Tile<BN>::lds_bytes, stream,
                       A, lda, M, B, ldb, N, C, ldc, tiles_m, klist, kcount, kt32, order, split, slab_stride, kpart);
    return hipGetLastError();
}

template <typename TA, typename TB>
static hipError_t launch_gemm_nt_f64_t(const TA* A, int lda, int M, const TB* B, int ldb, int N, double* C, int ldc,
                                       int K_pad, const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int* klist,
                                       int* kcount, hipStream_t stream, int split, int64_t slab_stride) {
    if (M <= 0 || N <= 0) return hipSuccess;
    if (split < 1 || split > 32 || (split > 1 && slab_stride < (int64_t)M * ldc)) return hipErrorInvalidValue;
    if (K_pad % 32 != 0) return hipErrorInvalidValue;
    const int kt32 = K_pad / 32;
    const int bn = gemm_f64_bn(N);
    const int tiles_m = (M + D_BM - 1) / D_BM, tiles_n = (N + bn - 1) / bn;
    const int64_t pairs = (int64_t)tiles_m * tiles_n;
    if (pairs > 0x7fffffff || pairs * split > 0x7fffffff) return hipErrorInvalidValue;
    const int force_dense = gemm_force_dense();          // benchmark / debug: every tile listed
    int* order = kcount + pairs;
    int* hist = order + pairs;
    int* kpart = hist + kt32 + 2;
    hipLaunchKernelGGL(k_build_klists_f64, dim3((unsigned)pairs), dim3(256), 0, stream, force_dense ? nullptr : nzA,
                       force_dense ? nullptr : nzB, G, v_group, N, tiles_m, kt32, klist, kcount, split, kpart, bn);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if ((e = hipMemsetAsync(hist, 0, (size_t)(kt32 + 2) * sizeof(int), stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_order_pairs_f64, dim3(1), dim3(1024), 0, stream, kcount, (int)pairs, kt32, hist, order);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    auto fold = [&](hipError_t e0) -> hipError_t {
        if (e0 != hipSuccess || split == 1) return e0;
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(k_fold_slabs, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), dim3(256), 0, stream, C, ldc, M, N,
                           split, slab_stride);
        return hipGetLastError();
    };
    if (bn == 16)
        return fold(launch_f64_tile<TA, TB, 16>(A, lda, M, B, ldb, N, C, ldc, tiles_m, pairs, klist, kcount, kt32, order, split,
                                                slab_stride, kpart, stream));
    if (bn == 48)
        return fold(launch_f64_tile<TA, TB, 48>(A, lda, M, B, ldb, N, C, ldc, tiles_m, pairs, klist, kcount, kt32, order, split, slab_stride,
                                           kpart, stream));
    if (bn == 32)
        return fold(launch_f64_tile<TA, TB, 32>(A, lda, M, B, ldb, N, C, ldc, tiles_m, pairs, klist, kcount, kt32, order, split, slab_stride,
                                       kpart, stream));
    if (bn == 64)
        return fold(launch_f64_tile<TA, TB, 64>(A, lda, M, B, ldb, N, C, ldc, tiles_m, pairs, klist, kcount, kt32, order, split, slab_stride,
                                       kpart, stream));
    return fold(launch_f64_tile<TA, TB, 128>(A, lda, M, B, ldb, N, C, ldc, tiles_m, pairs, klist, kcount, kt32, order, split, slab_stride,
                                    kpart, stream));
}

hipError_t launch_gemm_nt_f64(const double* A, int lda, int M, const double* B, int ldb, int N, double* C, int ldc,
                              int K_pad, const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int* klist,
                              int* kcount, hipStream_t stream, int split, int64_t slab_stride) {
    return launch_gemm_nt_f64_t<double, double>(A, lda, M, B, ldb, N, C, ldc, K_pad, nzA, nzB, G, v_group, klist, kcount, stream, split,
                                        slab_stride);
}

hipError_t launch_gemm_nt_f64_bf32(const double* A, int lda, int M, const float* B, int ldb, int N, double* C, int ldc,
                                   int K_pad, const uint8_t* nzA, int* klist, int* kcount, hipStream_t stream, int split,
                                   int64_t slab_stride) {
    return launch_gemm_nt_f64_t<double, float>(A, lda, M, B, ldb, N, C, ldc, K_pad, nzA, nullptr, 1, N, klist, kcount, stream, split,
                                       slab_stride);
}

hipError_t launch_gemm_nt_f64_ff32(const float* A, int lda, int M, const float* B, int ldb, int N, double* C, int ldc, int K_pad,
                                   const uint8_t* nzA, int* klist, int* kcount, hipStream_t stream, int split,
                                   int64_t slab_stride) {
    return launch_gemm_nt_f64_t<float, float>(A, lda, M, B, ldb, N, C, ldc, K_pad, nzA, nullptr, 1, N, klist, kcount, stream,
                                              split, slab_stride);
}

}  // namespace pbvi

// Ablation of the f32 tile engine (gemm.hip): which part of the K step keeps the MFMA pipe idle?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define GEMM_BK 32
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TILE_FLOATS = 256 * GEMM_BK;
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * 4;
__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
struct TileThread {
    int lane, wid, i, h, wr, wc;
    int srow[4], scol[4];
    int a_row[4], b_row[2];
    __device__ __forceinline__ void init() {
        const int tid = threadIdx.x;
        lane = tid & 63; wid = __builtin_amdgcn_readfirstlane(tid >> 6); i = lane & 31; h = lane >> 5; wr = wid >> 2; wc = wid & 3;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int c = it * 512 + tid;
            const int row = c >> 3, pc = c & 7;
            srow[it] = row;
            scol[it] = (pc ^ ((row >> 1) & 7)) * 4;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a_row[mi] = wr * 128 + mi * 32 + i;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b_row[ni] = wc * 64 + ni * 32 + i;
    }
};
__device__ __forceinline__ void tile_stage(const TileThread& t, float* lds, int buf, const float* Ablk, int lda,
                                           const float* Bblk, int ldb, int kt) {
    float* la = lds + buf * 2 * TILE_FLOATS;
    float* lb = la + TILE_FLOATS;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int wave_chunk = it * 512 + t.wid * 64;
        glds16(Ablk + (int64_t)t.srow[it] * lda + kt * GEMM_BK + t.scol[it], la + wave_chunk * 4);
        glds16(Bblk + (int64_t)t.srow[it] * ldb + kt * GEMM_BK + t.scol[it], lb + wave_chunk * 4);
    }
}
__global__ void k_fill(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (float)(x & 0xffffff) * (1.f / 16777216.f);
    }
}
template <int MODE>
__device__ __forceinline__ void tile_compute(const TileThread& t, const float* lds, int buf, f32x16 (&acc)[4][2]) {
    const float* la = lds + buf * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 af[4], bf[2];
        if (MODE == 2) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = f32x4{1.f + t.lane, 2.f, 3.f, 4.f + g};
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) bf[ni] = f32x4{1.f, 2.f + t.lane, 3.f + g, 4.f};
        } else {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int pc = (2 * g + t.h) ^ ((t.a_row[mi] >> 1) & 7);
                af[mi] = *(const f32x4*)(la + t.a_row[mi] * GEMM_BK + pc * 4);
            }
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int pc = (2 * g + t.h) ^ ((t.b_row[ni] >> 1) & 7);
                bf[ni] = *(const f32x4*)(lb + t.b_row[ni] * GEMM_BK + pc * 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }
}

__device__ __forceinline__ void load_frags(const TileThread& t, const float* la, const float* lb, int g, f32x4 (&af)[4], f32x4 (&bf)[2]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int pc = (2 * g + t.h) ^ ((t.a_row[mi] >> 1) & 7);
        af[mi] = *(const f32x4*)(la + t.a_row[mi] * GEMM_BK + pc * 4);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int pc = (2 * g + t.h) ^ ((t.b_row[ni] >> 1) & 7);
        bf[ni] = *(const f32x4*)(lb + t.b_row[ni] * GEMM_BK + pc * 4);
    }
}
__device__ __forceinline__ void mfma_block(const f32x4 (&af)[4], const f32x4 (&bf)[2], f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
}
// fragments of group g+1 are requested BEFORE the MFMAs of group g
template <int PRIO>
__device__ __forceinline__ void tile_compute_pipe(const TileThread& t, const float* lds, int buf, f32x16 (&acc)[4][2]) {
    const float* la = lds + buf * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
    f32x4 a0[4], b0[2], a1[4], b1[2];
    load_frags(t, la, lb, 0, a0, b0);
    load_frags(t, la, lb, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    if (PRIO) __builtin_amdgcn_s_setprio(1);
    mfma_block(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(t, la, lb, 2, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a1, b1, acc);
    __builtin_amdgcn_sched_barrier(0);
    load_frags(t, la, lb, 3, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a0, b0, acc);
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a1, b1, acc);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
}
// MODE 0: full   1: no global loads   2: no LDS reads   3: no barrier (wrong results, timing only)   4: no loads, no barrier

// all stage loads issued by waves 0-3 (one per SIMD); waves 4-7 go straight to the MFMAs
__device__ __forceinline__ void tile_stage_half(const TileThread& t, float* lds, int buf, const float* Ablk, int lda,
                                                const float* Bblk, int ldb, int kt) {
    float* la = lds + buf * 2 * TILE_FLOATS;
    float* lb = la + TILE_FLOATS;
    const int tid = threadIdx.x & 255;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int c = it * 256 + tid;
        const int row = c >> 3, pc = c & 7;
        const int col = (pc ^ ((row >> 1) & 7)) * 4;
        const int wave_chunk = it * 256 + t.wid * 64;
        glds16(Ablk + (int64_t)row * lda + kt * GEMM_BK + col, la + wave_chunk * 4);
        glds16(Bblk + (int64_t)row * ldb + kt * GEMM_BK + col, lb + wave_chunk * 4);
    }
}
template <int MODE>
__global__ __launch_bounds__(512) void k_run(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                             float* __restrict__ C, int k_tiles, int reps, int nrowsB, int rot) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    TileThread t;
    t.init();
    const float* Ablk = A + (int64_t)(blockIdx.x & 3) * 256 * lda;
    const float* Bblk = B + (int64_t)((blockIdx.x >> 2) % (nrowsB / 256)) * 256 * ldb;
    if (MODE == 7) {   // L2-friendly: the 32 blocks of an XCD read 4 A tiles x 8 B tiles, time-aligned
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        Ablk = A + (int64_t)(slot & 3) * 256 * lda;
        Bblk = B + (int64_t)(xcd * 8 + (slot >> 2)) * 256 * ldb;
    }
    if (MODE == 8) {   // every block of an XCD reads the same tiles
        const int xcd = blockIdx.x & 7;
        Ablk = A + (int64_t)(xcd & 3) * 256 * lda;
        Bblk = B + (int64_t)xcd * 256 * ldb;
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    for (int r = 0; r < reps; ++r) {
        const int k0 = (blockIdx.x * rot) % k_tiles;
        if (MODE == 10) { if (t.wid < 4) tile_stage_half(t, lds, 0, Ablk, lda, Bblk, ldb, k0); }
        else tile_stage(t, lds, 0, Ablk, lda, Bblk, ldb, k0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int it = 0; it < k_tiles; ++it) {
            if (MODE == 10) {
                if (it + 1 < k_tiles && t.wid < 4) { int kn = k0 + it + 1; if (kn >= k_tiles) kn -= k_tiles; tile_stage_half(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, kn); }
            } else
            if (MODE != 1 && MODE != 4)
                if (it + 1 < k_tiles) { int kn = k0 + it + 1; if (kn >= k_tiles) kn -= k_tiles; tile_stage(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, kn); }
            if (MODE == 5) tile_compute_pipe<0>(t, lds, buf, acc);
            else if (MODE == 6) tile_compute_pipe<1>(t, lds, buf, acc);
            else tile_compute<MODE == 10 ? 0 : MODE>(t, lds, buf, acc);
            if (MODE != 3 && MODE != 4) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            buf ^= 1;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[mi][ni][e];
    C[blockIdx.x * 512 + threadIdx.x] = s;
}

__device__ __forceinline__ void stage_one(const TileThread& t, float* lds, int buf, const float* Ablk, int lda,
                                          const float* Bblk, int ldb, int kt, int it) {
    float* la = lds + buf * 2 * TILE_FLOATS;
    float* lb = la + TILE_FLOATS;
    const int wave_chunk = it * 512 + t.wid * 64;
    glds16(Ablk + (int64_t)t.srow[it] * lda + kt * GEMM_BK + t.scol[it], la + wave_chunk * 4);
    glds16(Bblk + (int64_t)t.srow[it] * ldb + kt * GEMM_BK + t.scol[it], lb + wave_chunk * 4);
}
__device__ __forceinline__ void mfma_j(const f32x4 (&af)[4], const f32x4 (&bf)[2], int j, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
}
#define SB() __builtin_amdgcn_sched_barrier(0)
// One K step, software-pipelined across the step boundary.  On entry F0 holds the g=0 fragments of `buf`; on exit
// (when has_next) F0 holds the g=0 fragments of buf^1.  Stage loads of the next tile are spread between MFMA groups.
template <bool has_next>
__device__ __forceinline__ void step_pipelined(const TileThread& t, float* lds, int buf, const float* Ablk, int lda,
                                               const float* Bblk, int ldb, int kt_next,
                                               f32x4 (&a0)[4], f32x4 (&b0)[2], f32x16 (&acc)[4][2]) {
    const float* la = lds + buf * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
    f32x4 a1[4], b1[2];
    // g = 0
    mfma_j(a0, b0, 0, acc); SB();
    if (has_next) stage_one(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, kt_next, 0);
    SB();
    mfma_j(a0, b0, 1, acc); SB();
    load_frags(t, la, lb, 1, a1, b1); SB();
    mfma_j(a0, b0, 2, acc); SB();
    if (has_next) stage_one(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, kt_next, 1);
    SB();
    mfma_j(a0, b0, 3, acc); SB();
    // g = 1
    mfma_j(a1, b1, 0, acc); SB();
    if (has_next) stage_one(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, kt_next, 2);
    SB();
    mfma_j(a1, b1, 1, acc); SB();
    load_frags(t, la, lb, 2, a0, b0); SB();
    mfma_j(a1, b1, 2, acc); SB();
    if (has_next) stage_one(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, kt_next, 3);
    SB();
    mfma_j(a1, b1, 3, acc); SB();
    // g = 2
    mfma_j(a0, b0, 0, acc); SB();
    mfma_j(a0, b0, 1, acc); SB();
    load_frags(t, la, lb, 3, a1, b1); SB();
    mfma_j(a0, b0, 2, acc); SB();
    mfma_j(a0, b0, 3, acc); SB();
    // g = 3
    mfma_j(a1, b1, 0, acc); SB();
    mfma_j(a1, b1, 1, acc); SB();
    mfma_j(a1, b1, 2, acc); SB();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    SB();
    if (has_next) load_frags(t, lds + (buf ^ 1) * 2 * TILE_FLOATS, lds + (buf ^ 1) * 2 * TILE_FLOATS + TILE_FLOATS, 0, a0, b0);
    SB();
    mfma_j(a1, b1, 3, acc); SB();
}
template <int MODE>
__global__ __launch_bounds__(512) void k_run_p(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                             float* __restrict__ C, int k_tiles, int reps, int nrowsB, int rot) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    TileThread t;
    t.init();
    const float* Ablk = A + (int64_t)(blockIdx.x & 3) * 256 * lda;
    const float* Bblk = B + (int64_t)((blockIdx.x >> 2) % (nrowsB / 256)) * 256 * ldb;
    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
    for (int r = 0; r < reps; ++r) {
        const int k0 = (blockIdx.x * rot) % k_tiles;
        tile_stage(t, lds, 0, Ablk, lda, Bblk, ldb, k0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        f32x4 a0[4], b0[2];
        load_frags(t, lds, lds + TILE_FLOATS, 0, a0, b0);
        int buf = 0;
        for (int it = 0; it + 1 < k_tiles; ++it) {
            int kn = k0 + it + 1; if (kn >= k_tiles) kn -= k_tiles;
            step_pipelined<true>(t, lds, buf, Ablk, lda, Bblk, ldb, kn, a0, b0, acc);
            buf ^= 1;
        }
        step_pipelined<false>(t, lds, buf, Ablk, lda, Bblk, ldb, 0, a0, b0, acc);
    }
    float s = 0.f;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[mi][ni][e];
    C[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int MODE>
static void run(const char* name, const float* A, const float* B, float* C, int lda, int k_tiles, int reps, int nrowsB, int rot) {
    hipFuncSetAttribute((const void*)k_run<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    hipFuncSetAttribute((const void*)k_run_p<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (MODE == 9) hipLaunchKernelGGL(k_run_p<MODE>, dim3(256), dim3(512), GEMM_LDS_BYTES, 0, A, lda, B, lda, C, k_tiles, reps, nrowsB, rot);
        else hipLaunchKernelGGL(k_run<MODE>, dim3(256), dim3(512), GEMM_LDS_BYTES, 0, A, lda, B, lda, C, k_tiles, reps, nrowsB, rot);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = 256.0 * reps * k_tiles * 2.0 * 256 * 256 * 32;
        if (rep) printf("%-28s %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", name, ms, fl / ms / 1e9, fl / ms / 1e9 / 1.573);
    }
}
int main() {
    const int k_tiles = 938, lda = k_tiles * 32, nrowsB = 64 * 256;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)1024 * lda * 4);
    hipMalloc(&B, (size_t)nrowsB * lda * 4);
    hipMalloc(&C, 256 * 512 * 4);
    hipMemset(A, 0, (size_t)1024 * lda * 4);
    hipMemset(B, 0, (size_t)nrowsB * lda * 4);
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, (size_t)1024 * lda, 1u);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B, (size_t)nrowsB * lda, 2u);
        printf("-- random data --\n");
    }
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 1;
    run<0>("full aligned", A, B, C, lda, k_tiles, reps, nrowsB, 0);
    run<0>("full rotated starts", A, B, C, lda, k_tiles, reps, nrowsB, 37);
    run<1>("no global loads", A, B, C, lda, k_tiles, reps, nrowsB, 0);
    run<10>("loads by waves 0-3 only", A, B, C, lda, k_tiles, reps, nrowsB, 37);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("done: %s\n", hipGetErrorString(e));
    return 0;
}

// Score / projection GEMM for the PBVI backup on gfx950 (CDNA4).
//
//   C[z][m][n] = sum over (a chunk of) the pair's K-tile list of A[m][k] * B[n][k]
//
// In the backup A = belief block [B][S], B = Gamma [(a,o,v)][S] (reference:
// xp.tensordot(belief_array, gamma_a_o_t, (1,3)), src/pomdp.py:1495).  Both operands are
// K-contiguous, exactly the layouts the reference keeps them in, so no transpose copy is ever made
// (NumPy's tensordot makes one internally).
//
// Tile engine: v_mfma_f32_32x32x2_f32 (exact f32 fma chains, 64 FLOP/clk/SIMD = the 157 TFLOP/s fp32
// matrix peak).  256x256x32 tiles, 8 waves (2 per SIMD) laid out 2(M) x 4(N), each wave a 128x64
// sub-tile = 4x2 MFMA blocks = 128 accumulator VGPRs.  Operands go HBM/L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4) into a double buffer; the 128-byte LDS rows are XOR-swizzled on the SOURCE
// address so the ds_read_b128 fragment reads are bank-conflict free.  K is permuted inside each 8-wide
// group (lane half h takes k = 8g+4h+j for MFMA step j) so one ds_read_b128 feeds four MFMAs; the
// permutation is applied to both operands, so the sum is unchanged.
//
// Zero-tile skipping (exact): for every (m-tile, n-tile) pair only the K tiles where both operands have
// a non-zero are listed and multiplied; a skipped tile could only have added +0 products.
//
// Two schedulers over the same tile engine:
//  * k_gemm_nt_f32_mfma   -- one block per (pair, chunk of the list), optionally batched.  Used with a
//    single chunk per pair for the dense projection (slab 0 is the finished product).
//  * k_gemm_nt_f32_streamk -- persistent, one block per CU, each owning an equal contiguous share of the
//    concatenated tile lists (stream-K): per-pair list lengths differ by 1000x under zero-tile skipping,
//    so equal shares of listed tile-steps is what balances the chip.  A pair split over several blocks
//    leaves one partial slab per block; the consumer sums them in a fixed order (deterministic).
#include "pbvi_common.h"

#include <cstdlib>

namespace pbvi {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TILE_FLOATS = 256 * GEMM_BK;          // one operand tile: 256 rows x 32 k = 32 KiB
constexpr int KL_IRR_BIT = 30;                      // tile-list entries: K tile index | (needs gathers << 30), see k_build_klists
constexpr int KL_MASK = (1 << KL_IRR_BIT) - 1;
constexpr int GEMM_LDS_BYTES = 4 * TILE_FLOATS * 4;  // 2 buffers x (A tile + B tile) = 128 KiB

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// Per-thread constants of the tile engine.
struct TileThread {
    int lane, wid, i, h, wr, wc;
    int srow[4], scol[4];      // staging: row and (swizzled) first column of this thread's 4 chunks
    int a_row[4], b_row[2];    // fragment rows
    __device__ __forceinline__ void init() {
        const int tid = threadIdx.x;
        lane = tid & 63;
        wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: the LDS-DMA base (M0) of every stage load becomes an s_add
        i = lane & 31;
        h = lane >> 5;
        wr = wid >> 2;
        wc = wid & 3;
        // 2048 16-byte chunks per operand tile, 4 per thread.  LDS image is lane-linear (chunk c at byte
        // 16c); physical chunk p of row r holds logical chunk p ^ ((r>>1)&7).
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
        const int wave_chunk = it * 512 + t.wid * 64;        // wave-uniform LDS base (M0)
        glds16(Ablk + (int64_t)t.srow[it] * lda + kt * GEMM_BK + t.scol[it], la + wave_chunk * 4);
        glds16(Bblk + (int64_t)t.srow[it] * ldb + kt * GEMM_BK + t.scol[it], lb + wave_chunk * 4);
    }
}

// MFMA group g of the four 8-wide K groups of a staged tile pair (one ds_read_b128 feeds four MFMAs)
__device__ __forceinline__ void tile_compute_group(const TileThread& t, const float* lds, int buf, f32x16 (&acc)[4][2], int g) {
    const float* la = lds + buf * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
    f32x4 af[4], bf[2];
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
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
}

__device__ __forceinline__ void tile_compute(const TileThread& t, const float* lds, int buf, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) tile_compute_group(t, lds, buf, acc, g);
}

// Multiply list entries [i0, i1) of one pair into acc (double-buffered LDS-DMA pipeline).
__device__ __forceinline__ void tile_run(const TileThread& t, float* lds, const float* Ablk, int lda, const float* Bblk,
                                         int ldb, const int* __restrict__ kl, int i0, int i1, f32x16 (&acc)[4][2]) {
    tile_stage(t, lds, 0, Ablk, lda, Bblk, ldb, kl[i0] & KL_MASK);
    int k_next = (i0 + 1 < i1) ? (kl[i0 + 1] & KL_MASK) : 0;   // list entries are read one step ahead of their use
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int it = i0; it < i1; ++it) {
        int k_after = 0;
        if (it + 1 < i1) {
            tile_stage(t, lds, buf ^ 1, Ablk, lda, Bblk, ldb, k_next);
            if (it + 2 < i1) k_after = kl[it + 2] & KL_MASK;
        }
        tile_compute(t, lds, buf, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        k_next = k_after;
    }
}

__device__ __forceinline__ void tile_zero(f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
}

// lane holds column (lane&31); register e holds row (e&3) + 8*(e>>2) + 4*(lane>>5)
__device__ __forceinline__ void tile_store(const TileThread& t, float* Cz, int ldc, int tm, int tn,
                                           const f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int col = tn * 256 + t.wc * 64 + ni * 32 + t.i;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = tm * 256 + t.wr * 128 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * t.h;
                Cz[(int64_t)row * ldc + col] = acc[mi][ni][e];
            }
        }
}

// --------------------------------------------------------------------------- //
// scheduler 1: one block per (pair, chunk), batched
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(512) void k_gemm_nt_f32_mfma(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C,
    int ldc, int64_t slab_stride, int tiles_m, int tiles_n, int k_tiles, int chunk_len, int max_chunks,
    const int* __restrict__ klist, const int* __restrict__ kcount, int64_t batch_stride_b, int64_t batch_stride_c) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // Work-item map.  Blocks b and b+8 share an XCD (round-robin dispatch).  A group = the tiles_m
    // M-tiles of one (n-tile, chunk): they read the same B tile rows, so a group stays on ONE XCD (one L2
    // fill), while groups are dealt round-robin over the 8 XCDs with the chunk index slowest.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
    const int tm = slot % tiles_m;
    const int q = (slot / tiles_m) * 8 + xcd;           // group id
    const int tn = q % tiles_n;
    const int z = q / tiles_n;
    if (z >= max_chunks) return;                        // grid is padded to a multiple of 8 groups
    const int batch = blockIdx.y;
    B += batch * batch_stride_b;
    C += batch * batch_stride_c;
    const int pair = (batch * tiles_n + tn) * tiles_m + tm;
    const int cnt = kcount[pair];
    const int i0 = z * chunk_len;
    const int i1 = (i0 + chunk_len < cnt) ? i0 + chunk_len : cnt;
    if (i0 >= i1) return;                               // this chunk does not exist
    TileThread t;
    t.init();
    f32x16 acc[4][2];
    tile_zero(acc);
    tile_run(t, lds, A + (int64_t)tm * 256 * lda, lda, B + (int64_t)tn * 256 * ldb, ldb,
             klist + (int64_t)pair * k_tiles, i0, i1, acc);
    tile_store(t, C + (int64_t)z * slab_stride, ldc, tm, tn, acc);
}

// --------------------------------------------------------------------------- //
// scheduler 2: stream-K, persistent
// --------------------------------------------------------------------------- //
// Shares are equal in COST, not in tile-steps: every non-empty pair carries `ovh` extra units in front of its list
// (zeroing the accumulators, the exposed latency of its first operand tiles, the 256 KiB slab store), so a block
// whose share crosses many one-tile lists (a rarely-seen observation) is not the last one to finish.  Unit u of a
// pair maps to list entry max(0, u - ovh); a block whose part of a pair lies wholly inside the overhead units
// multiplies nothing and stores nothing.
// plan[0] = q (cost units per block, an upper bound of the tile-steps of any chain), plan[1] = T (total cost units)
__global__ void k_streamk_plan(const int* __restrict__ kcount, int pairs, int nblocks, int k_tiles, int max_split,
                               int ovh, int* __restrict__ prefix, int* __restrict__ start_pair,
                               int* __restrict__ first_block, int* __restrict__ nchunks, int* __restrict__ plan) {
    __shared__ int part[256];
    __shared__ int q_sh;
    const int tid = threadIdx.x;
    const int per = (pairs + 255) / 256;
    const int p0 = (tid * per < pairs) ? tid * per : pairs;
    const int p1 = (p0 + per < pairs) ? p0 + per : pairs;
    int sum = 0;
    for (int p = p0; p < p1; ++p) {
        const int cnt = kcount[p];
        sum += cnt > 0 ? cnt + ovh : 0;
    }
    part[tid] = sum;
    for (int i = tid; i < nblocks; i += 256) start_pair[i] = -1;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) {
            const int v = part[i];
            part[i] = run;
            run += v;
        }
        int q = (run + nblocks - 1) / nblocks;
        const int qmin = (k_tiles + ovh + max_split - 1) / max_split;   // a pair spans at most max_split+1 blocks
        if (q < qmin) q = qmin;
        if (q < 1) q = 1;
        q_sh = q;
        plan[0] = q;
        plan[1] = run;
        prefix[pairs] = run;
    }
    __syncthreads();
    const int q = q_sh;
    int run = part[tid];
    for (int p = p0; p < p1; ++p) {
        const int cnt = kcount[p];
        prefix[p] = run;
        if (cnt > 0) {
            const int cost = cnt + ovh;
            const int fb = run / q, lb = (run + cost - 1) / q;
            const int fw = (run + ovh) / q;                                // first block that reaches a list entry
            first_block[p] = fw;
            nchunks[p] = lb - fw + 1;
            for (int i = fb; i <= lb; ++i)
                if ((int64_t)i * q >= run) start_pair[i] = p;          // block i's share begins inside pair p
            run += cost;
        } else {
            first_block[p] = 0;
            nchunks[p] = 0;
        }
    }
}

__global__ __launch_bounds__(512) void k_gemm_nt_f32_streamk(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
    int64_t slab_stride, int tiles_m, int pairs, int k_tiles, const int* __restrict__ klist,
    const int* __restrict__ kcount, const int* __restrict__ prefix, const int* __restrict__ start_pair,
    const int* __restrict__ first_block, const int* __restrict__ plan, int ovh) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // XCD-contiguous logical ids: consecutive shares (which walk the same n-tile's pairs) on one XCD.
    const int nb = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qq = nb >> 3, r = nb & 7;
        const int base = (xcd < r) ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq;
        L = base + (bid >> 3);
    }
    const int q = plan[0], T = plan[1];
    int64_t pos = (int64_t)L * q;
    if (pos >= T) return;
    const int64_t end = (pos + q < T) ? pos + q : T;
    int p = start_pair[L];
    if (p < 0) return;                                   // cannot happen when pos < T; defensive
    TileThread t;
    t.init();
    while (pos < end && p < pairs) {
        const int cnt = kcount[p];
        if (cnt == 0) {                                  // block-uniform
            ++p;
            continue;
        }
        const int u_lo = (int)(pos - prefix[p]);         // cost units of this pair inside the share
        const int64_t room = end - pos;
        const int u_hi = (u_lo + room < cnt + ovh) ? (int)(u_lo + room) : cnt + ovh;
        const int lo = u_lo > ovh ? u_lo - ovh : 0;
        const int hi = u_hi > ovh ? u_hi - ovh : 0;
        if (hi > lo) {
            const int tm = p % tiles_m, tn = p / tiles_m;
            f32x16 acc[4][2];
            tile_zero(acc);
            tile_run(t, lds, A + (int64_t)tm * 256 * lda, lda, B + (int64_t)tn * 256 * ldb, ldb,
                     klist + (int64_t)p * k_tiles, lo, hi, acc);
            // first chunk of the pair: the full slab; a continuation: this block's tile behind it (GemmPlan::c_floats)
            const bool cont = L != first_block[p];
            tile_store(t, cont ? C + slab_stride + (int64_t)L * (256 * 256) : C, cont ? 256 : ldc, cont ? 0 : tm, cont ? 0 : tn, acc);
        }
        pos += u_hi - u_lo;
        ++p;
    }
}

// --------------------------------------------------------------------------- //
// scheduler 2b: stream-K with the Gamma projection FUSED into the B-operand staging (one reachable state per (s, a))
// --------------------------------------------------------------------------- //
// Gamma[(g, v)][s] = gamma * rto[g][s] * alpha[v][rs[a][s]]   (g = a*O + o; src/pomdp.py:1485-1491 with R = 1)
// is not written to HBM for n-tiles that lie inside one (action, observation) group: the block that multiplies the
// tile generates it on the way into LDS -- per thread and K step one 16-byte load each of the successor indices and
// of RTO (a thread's four 16-byte chunks of the tile sit in rows 64 apart at the same 4 states), four 16-byte alpha
// loads (a grid move maps 4 consecutive states to 4 consecutive successors; otherwise four gathers each), 16
// multiplies, four ds_write_b128 into the same swizzled image the LDS-DMA path produces.  The belief operand still
// arrives by LDS-DMA.  Same products, same rounding (rto*alpha, then *gamma, contraction off) and same summation
// order as projecting first, so the scores are bit-identical to the unfused pipeline -- without its 0.8 GB of Gamma
// writes and the 0.3 ms kernel that makes them.  Tiles flagged in `mat` (those that straddle two groups, and the tail
// tile with the magnitude / reward rows) are projected as before and take the LDS-DMA path from B.
typedef int i32x4 __attribute__((ext_vector_type(4)));
#ifndef PBVI_FUSED_EXP
#define PBVI_FUSED_EXP 0      // diagnosis builds only: 1 = no alpha loads, 2 = no LDS stores (wrong results, timing only)
#endif

// Loads of the generated operand are issued as inline assembly and waited for with explicit s_waitcnt: vector
// memory operations return in issue order, so "vmcnt(n)" means "all but the last n issued have arrived".  Left to the
// compiler, every use of a loaded value was preceded by vmcnt(0) -- which also waits for the belief tile's LDS-DMA
// loads issued a moment earlier, in the middle of the MFMA stream.
__device__ __forceinline__ void ld16(f32x4& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld16i(i32x4& dst, const int32_t* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld4(float& dst, const float* p) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}

__device__ __forceinline__ f32x4 fused_value(float gamma, f32x4 w, f32x4 av) {
#pragma clang fp contract(off)
    const f32x4 pr = w * av;                 // one rounding, then the scale: exactly k_project's arithmetic
    return gamma * pr;
}

// Multiply list entries [i0, i1) of one pair whose B tile (rows of ONE group) is generated (see above).
// rsrow / rtorow: the group's successor and RTO rows; arow0: alpha row of the tile's first row + this thread's row;
// A list entry with bit KL_IRR_BIT set: that K tile holds a 4-state chunk whose successors are not 4 consecutive states
// (gathers instead of one 16-byte load; block-uniform).
__device__ __forceinline__ void tile_run_fused(const TileThread& t, float* lds, const float* Ablk, int lda,
                                               const int32_t* __restrict__ rsrow, const float* __restrict__ rtorow,
                                               const float* __restrict__ arow0, int64_t a_step /* 64 rows of alpha */,
                                               float gamma, const int* __restrict__ kl,
                                               int i0, int i1, f32x16 (&acc)[4][2]) {
    const int tid = threadIdx.x;
    auto stage_a = [&](int buf, int entry) {
        const int kt = entry & KL_MASK;
        float* la = lds + buf * 2 * TILE_FLOATS;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            glds16(Ablk + (int64_t)t.srow[it] * lda + kt * GEMM_BK + t.scol[it], la + (it * 512 + t.wid * 64) * 4);
    };
    // scol is the same for a thread's four chunks (their rows differ by 64): one table load pair serves all four.
    // Schedule of a K step -- the MFMA groups g0..g3 of the CURRENT tile carry the latencies of the NEXT tile's operands:
    //   tables, belief DMA | g0 | wait tables; alpha loads | g1 | g2 | wait all; scale + 4 ds_write_b128 | g3 | barrier
    i32x4 idx;
    f32x4 w, av[4];
    auto tables = [&](int entry) {
        const int s = (entry & KL_MASK) * GEMM_BK + t.scol[0];
        ld16i(idx, rsrow + s);
        ld16(w, rtorow + s);
    };
    auto alphas = [&](int entry) {
        if (entry >> KL_IRR_BIT) {                        // block-uniform: the list entry carries the flag (k_build_klists)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const float* arow = arow0 + it * a_step;
                float e0, e1, e2, e3;
                ld4(e0, arow + idx[0]);
                ld4(e1, arow + idx[1]);
                ld4(e2, arow + idx[2]);
                ld4(e3, arow + idx[3]);
                // (assembled before the wait: register moves of values still in flight would be wrong, so the vector is
                // built by the same asm that waits)
                asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                             : "=&v"(av[it].x), "=&v"(av[it].y), "=&v"(av[it].z), "=&v"(av[it].w)
                             : "v"(e0), "v"(e1), "v"(e2), "v"(e3)
                             : "memory");
            }
        } else {
#if PBVI_FUSED_EXP == 1
#pragma unroll
            for (int it = 0; it < 4; ++it) av[it] = w;
#else
#pragma unroll
            for (int it = 0; it < 4; ++it) ld16(av[it], arow0 + it * a_step + idx[0]);
#endif
        }
    };
    auto store_b = [&](int buf) {
        float* lb = lds + buf * 2 * TILE_FLOATS + TILE_FLOATS;
#if PBVI_FUSED_EXP == 2
        if (gamma == 12345.f)
#endif
#pragma unroll
        for (int it = 0; it < 4; ++it) *(f32x4*)(lb + (it * 512 + tid) * 4) = fused_value(gamma, w, av[it]);
    };
    // first tile of the segment: nothing to overlap with
    tables(kl[i0]);
    stage_a(0, kl[i0]);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(idx), "+v"(w)::"memory");
    alphas(kl[i0]);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3])::"memory");
    store_b(0);
    int k_next = (i0 + 1 < i1) ? kl[i0 + 1] : 0;         // list entries are read one step ahead of their use
    __syncthreads();
    int buf = 0;
    for (int it = i0; it < i1; ++it) {
        int k_after = 0;
        const bool more = it + 1 < i1;                    // block-uniform
        if (more) {
            tables(k_next);                               // the 2 small loads first, then the 4 DMA loads of the belief tile
            stage_a(buf ^ 1, k_next);
            if (it + 2 < i1) k_after = kl[it + 2];
        }
        tile_compute_group(t, lds, buf, acc, 0);
        if (more) {
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(idx), "+v"(w)::"memory");      // the tables are here; the DMA may not be
            alphas(k_next);
        }
        tile_compute_group(t, lds, buf, acc, 1);
        tile_compute_group(t, lds, buf, acc, 2);
        if (more) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3])::"memory");
            store_b(buf ^ 1);
        }
        tile_compute_group(t, lds, buf, acc, 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        k_next = k_after;
    }
}

__global__ __launch_bounds__(512) void k_gemm_nt_f32_streamk_fused(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, FusedB fb, float* __restrict__ C, int ldc,
    int64_t slab_stride, int tiles_m, int pairs, int k_tiles, const int* __restrict__ klist, const int* __restrict__ kcount,
    const int* __restrict__ prefix, const int* __restrict__ start_pair, const int* __restrict__ first_block,
    const int* __restrict__ plan, int ovh) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nb = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qq = nb >> 3, r = nb & 7;
        const int base = (xcd < r) ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq;
        L = base + (bid >> 3);
    }
    const int q = plan[0], T = plan[1];
    int64_t pos = (int64_t)L * q;
    if (pos >= T) return;
    const int64_t end = (pos + q < T) ? pos + q : T;
    int p = start_pair[L];
    if (p < 0) return;
    TileThread t;
    t.init();
    while (pos < end && p < pairs) {
        const int cnt = kcount[p];
        if (cnt == 0) {                                  // block-uniform
            ++p;
            continue;
        }
        const int u_lo = (int)(pos - prefix[p]);
        const int64_t room = end - pos;
        const int u_hi = (u_lo + room < cnt + ovh) ? (int)(u_lo + room) : cnt + ovh;
        const int lo = u_lo > ovh ? u_lo - ovh : 0;
        const int hi = u_hi > ovh ? u_hi - ovh : 0;
        if (hi > lo) {
            const int tm = p % tiles_m, tn = p / tiles_m;
            f32x16 acc[4][2];
            tile_zero(acc);
            if (fb.mat[tn]) {                            // block-uniform: projected rows, the LDS-DMA path
                const int bt = fb.ctile != nullptr ? fb.ctile[tn] : tn;      // (compact layout: only these tiles exist)
                tile_run(t, lds, A + (int64_t)tm * 256 * lda, lda, B + (int64_t)bt * 256 * ldb, ldb,
                         klist + (int64_t)p * k_tiles, lo, hi, acc);
            } else {
                const int r0 = tn * 256;
                const int g = r0 / fb.V, v0 = r0 - g * fb.V;                 // scalar: all 256 rows belong to group g
                tile_run_fused(t, lds, A + (int64_t)tm * 256 * lda, lda, fb.rs + (int64_t)(g / fb.O) * fb.S_pad,
                               fb.rto + (int64_t)g * fb.S_pad, fb.alpha + (int64_t)(v0 + t.srow[0]) * fb.lda,
                               (int64_t)64 * fb.lda, fb.gamma,
                               klist + (int64_t)p * k_tiles, lo, hi, acc);
            }
            // first chunk of the pair: the full slab; a continuation: this block's tile behind it (GemmPlan::c_floats)
            const bool cont = L != first_block[p];
            tile_store(t, cont ? C + slab_stride + (int64_t)L * (256 * 256) : C, cont ? 256 : ldc, cont ? 0 : tm, cont ? 0 : tn, acc);
        }
        pos += u_hi - u_lo;
        ++p;
    }
}

// --------------------------------------------------------------------------- //
// scheduler 2c: stream-K with the Gamma projection fused for 2..7 reachable states per (s, a)
// --------------------------------------------------------------------------- //
// Gamma[(g, v)][s] = gamma * sum_r rto[g][r][s] * alpha[v][rs[a][r][s]]   (src/pomdp.py:1485-1491, the padded-ELL SpMM)
// With R successors a generated B tile needs R gathered alpha chunks per 16-byte output chunk: 20 16-byte loads per
// thread and K step at R = 5 against the 128 MFMAs (8192 SIMD cycles) a wave issues in that step.  What does not fit
// is registers (the tile engine leaves ~24 for the staging), so a K step is cut into 2R UNITS -- (half h of the
// thread's four rows) x (successor slot r) -- of two alpha loads each, issued at the 16 boundaries between the
// 8-MFMA sub-groups of the step and consumed two boundaries later: two units (4 loads, 16 VGPRs) in flight, one
// 2-row accumulator (8 VGPRs), the same footprint as the R = 1 kernel's staging.  The successor indices and RTO
// weights of the K tile are not held in registers at all: wave 0 copies them (R x 128 B each) into LDS by LDS-DMA
// one K step ahead, and every thread reads its 16 bytes of them right where it uses them.
// Arithmetic is k_project's, operation for operation (acc = 0; acc = acc + rto*alpha per r, contraction off; then
// gamma * acc), so the scores are bit-identical to the unfused pipeline.
// K tiles in which some 4-state chunk has non-consecutive successors for some r (grid edges; flagged per action in
// bit KL_IRR_BIT of the list entry) are NOT generated: k_project writes those Gamma tiles (8-16 % of them on the
// olfactory grids) and the step stages B by LDS-DMA like the unfused kernel.  Both kinds of step share one loop.
constexpr int FUSED_R_MAX = 7;                       // 2R + 2 <= 16 issue points per K step
constexpr int TAB_WORDS = 8 * GEMM_BK;               // one table image: [r < 8][32 states]
constexpr int GEMM_LDS_BYTES_R = GEMM_LDS_BYTES + 2 * 2 * TAB_WORDS * 4;   // + 2 buffers x (indices, weights) = 4 KiB

struct Frag {
    f32x4 af[4], bf[2];
};
__device__ __forceinline__ void frag_load(const TileThread& t, const float* lds, int buf, int g, Frag& f) {
    const float* la = lds + buf * 2 * TILE_FLOATS;
    const float* lb = la + TILE_FLOATS;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        const int pc = (2 * g + t.h) ^ ((t.a_row[mi] >> 1) & 7);
        f.af[mi] = *(const f32x4*)(la + t.a_row[mi] * GEMM_BK + pc * 4);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int pc = (2 * g + t.h) ^ ((t.b_row[ni] >> 1) & 7);
        f.bf[ni] = *(const f32x4*)(lb + t.b_row[ni] * GEMM_BK + pc * 4);
    }
}
__device__ __forceinline__ void frag_mfma(const Frag& f, int j, f32x16 (&acc)[4][2]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.af[mi][j], f.bf[ni][j], acc[mi][ni], 0, 0, 0);
}

__device__ __forceinline__ f32x4 proj_add(f32x4 acc, f32x4 w, f32x4 av) {
#pragma clang fp contract(off)
    const f32x4 pr = w * av;                 // k_project: acc = acc + w * av, one rounding per operation
    return acc + pr;
}
__device__ __forceinline__ f32x4 proj_scale(float gamma, f32x4 acc) {
#pragma clang fp contract(off)
    return gamma * acc;
}

// Multiply list entries [i0, i1) of one pair whose B rows belong to ONE group (see above).  rs_a / rto_g: the R
// successor / weight rows of the group's action / of the group, S_pad apart; arow0: alpha row of the tile's first
// row + this thread's row; Bblk: the projected Gamma rows (read for the flagged K tiles only).
template <int R>
__device__ __forceinline__ void tile_run_fused_r(const TileThread& t, float* lds, const float* Ablk, int lda,
                                                 const float* Bblk, int ldb, const int32_t* __restrict__ rs_a,
                                                 const float* __restrict__ rto_g, int S_pad,
                                                 const float* __restrict__ arow0, int64_t a_step /* 64 rows of alpha */,
                                                 float gamma, const int* __restrict__ kl, int i0, int i1,
                                                 f32x16 (&acc)[4][2]) {
    const int tid = threadIdx.x;
    int32_t* tabi = (int32_t*)(lds + 4 * TILE_FLOATS);      // [2][8][32] successor indices of the K tile
    float* tabw = (float*)(tabi + 2 * TAB_WORDS);           // [2][8][32] RTO weights
    constexpr int nunits = 2 * R;
    static_assert(R >= 2 && R <= FUSED_R_MAX, "2R + 2 issue points must fit the 16 of a K step");
    const int scol0 = t.scol[0];                            // the thread's 4 states inside the K tile (same for its 4 rows)
    auto stage_a = [&](int buf, int entry) {
        const int kt = entry & KL_MASK;
        float* la = lds + buf * 2 * TILE_FLOATS;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            glds16(Ablk + (int64_t)t.srow[it] * lda + kt * GEMM_BK + t.scol[it], la + (it * 512 + t.wid * 64) * 4);
    };
    auto stage_b = [&](int buf, int entry) {
        const int kt = entry & KL_MASK;
        float* lb = lds + buf * 2 * TILE_FLOATS + TILE_FLOATS;
#pragma unroll
        for (int it = 0; it < 4; ++it)
            glds16(Bblk + (int64_t)t.srow[it] * ldb + kt * GEMM_BK + t.scol[it], lb + (it * 512 + t.wid * 64) * 4);
    };
    auto stage_tab = [&](int tb, int entry) {               // wave 0: lane (r, c) copies chunk c of row r
        if (t.wid == 0 && t.lane < 8 * R) {
            const int64_t off = (int64_t)(t.lane >> 3) * S_pad + (entry & KL_MASK) * GEMM_BK + (t.lane & 7) * 4;
            glds16((const float*)(rs_a + off), (float*)(tabi + tb * TAB_WORDS));
            glds16(rto_g + off, tabw + tb * TAB_WORDS);
        }
    };
    f32x4 av[2][2];                                          // two units in flight
    f32x4 pacc[2];
    auto unit_issue = [&](int u, int slot, int tb) {
        const int hh = u >= R ? 1 : 0, r = u - hh * R;     // compile-time at the unrolled issue points
        const i32x4 idx = *(const i32x4*)(tabi + tb * TAB_WORDS + r * GEMM_BK + scol0);
#if PBVI_FUSED_EXP == 3      // diagnosis: every load from the row's first line (wrong results, timing only)
        const float* p0 = arow0 + (int64_t)(2 * hh) * a_step + (idx[0] & 3);
#else
        const float* p0 = arow0 + (int64_t)(2 * hh) * a_step + idx[0];
#endif
        ld16(av[slot][0], p0);
        ld16(av[slot][1], p0 + a_step);
    };
    auto unit_consume = [&](int u, int slot, int tb, int dst_buf, bool last_in_queue) {
        const int hh = u >= R ? 1 : 0, r = u - hh * R;
        const f32x4 w = *(const f32x4*)(tabw + tb * TAB_WORDS + r * GEMM_BK + scol0);
        if (last_in_queue)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(av[slot][0]), "+v"(av[slot][1])::"memory");
        else                                                  // the next unit's two loads may still be in flight
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(av[slot][0]), "+v"(av[slot][1])::"memory");
        if (r == 0) {
            pacc[0] = f32x4{0.f, 0.f, 0.f, 0.f};
            pacc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        pacc[0] = proj_add(pacc[0], w, av[slot][0]);
        pacc[1] = proj_add(pacc[1], w, av[slot][1]);
        if (r == R - 1) {
            float* lb = lds + dst_buf * 2 * TILE_FLOATS + TILE_FLOATS;
            *(f32x4*)(lb + ((2 * hh) * 512 + tid) * 4) = proj_scale(gamma, pacc[0]);
            *(f32x4*)(lb + ((2 * hh + 1) * 512 + tid) * 4) = proj_scale(gamma, pacc[1]);
        }
    };
    // first tile of the segment: nothing to overlap with
    {
        const int e0 = kl[i0];
        stage_a(0, e0);
        if (e0 >> KL_IRR_BIT) {
            stage_b(0, e0);
        } else {
            stage_tab(0, e0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int u = 0; u < nunits; ++u) {
                unit_issue(u, 0, 0);
                unit_consume(u, 0, 0, 0, true);
            }
        }
    }
    int e_next = (i0 + 1 < i1) ? kl[i0 + 1] : 0;            // list entries are read ahead of their use
    if (i0 + 1 < i1 && !(e_next >> KL_IRR_BIT)) stage_tab(1, e_next);
    int e_after = (i0 + 2 < i1) ? kl[i0 + 2] : 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int it = i0; it < i1; ++it) {
        const bool more = it + 1 < i1;                       // block-uniform, as everything that branches below
        const bool gen = more && !(e_next >> KL_IRR_BIT);
        int e_after2 = 0;
        if (more) {
            stage_a(buf ^ 1, e_next);
            if (!gen) stage_b(buf ^ 1, e_next);
            if (it + 2 < i1 && !(e_after >> KL_IRR_BIT)) stage_tab(buf, e_after);   // tile it+2's tables; tile it's are done with
            if (it + 3 < i1) e_after2 = kl[it + 3];
        }
        Frag f;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            frag_load(t, lds, buf, g, f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * g + j;                     // issue point p: unit p goes out, unit p-2 comes in
                if (gen) {
                    if (p >= 2 && p - 2 < nunits) unit_consume(p - 2, p & 1, buf ^ 1, buf ^ 1, p - 2 == nunits - 1);
                    if (p < nunits) unit_issue(p, p & 1, buf ^ 1);
                }
                frag_mfma(f, j, acc);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
        e_next = e_after;
        e_after = e_after2;
    }
}

template <int R>
__global__ __launch_bounds__(512) void k_gemm_nt_f32_streamk_fused_r(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, FusedB fb, float* __restrict__ C, int ldc,
    int64_t slab_stride, int tiles_m, int pairs, int k_tiles, const int* __restrict__ klist, const int* __restrict__ kcount,
    const int* __restrict__ prefix, const int* __restrict__ start_pair, const int* __restrict__ first_block,
    const int* __restrict__ plan, int ovh) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nb = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x, xcd = bid & 7, qq = nb >> 3, r = nb & 7;
        const int base = (xcd < r) ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq;
        L = base + (bid >> 3);
    }
    const int q = plan[0], T = plan[1];
    int64_t pos = (int64_t)L * q;
    if (pos >= T) return;
    const int64_t end = (pos + q < T) ? pos + q : T;
    int p = start_pair[L];
    if (p < 0) return;
    TileThread t;
    t.init();
    while (pos < end && p < pairs) {
        const int cnt = kcount[p];
        if (cnt == 0) {                                  // block-uniform
            ++p;
            continue;
        }
        const int u_lo = (int)(pos - prefix[p]);
        const int64_t room = end - pos;
        const int u_hi = (u_lo + room < cnt + ovh) ? (int)(u_lo + room) : cnt + ovh;
        const int lo = u_lo > ovh ? u_lo - ovh : 0;
        const int hi = u_hi > ovh ? u_hi - ovh : 0;
        if (hi > lo) {
            const int tm = p % tiles_m, tn = p / tiles_m;
            f32x16 acc[4][2];
            tile_zero(acc);
            if (fb.mat[tn]) {                            // block-uniform: projected rows, the LDS-DMA path
                tile_run(t, lds, A + (int64_t)tm * 256 * lda, lda, B + (int64_t)tn * 256 * ldb, ldb,
                         klist + (int64_t)p * k_tiles, lo, hi, acc);
            } else {
                const int r0 = tn * 256;
                const int g = r0 / fb.V, v0 = r0 - g * fb.V;                 // scalar: all 256 rows belong to group g
                tile_run_fused_r<R>(t, lds, A + (int64_t)tm * 256 * lda, lda, B + (int64_t)tn * 256 * ldb, ldb,
                                 fb.rs + (int64_t)(g / fb.O) * R * fb.S_pad, fb.rto + (int64_t)g * R * fb.S_pad,
                                 fb.S_pad, fb.alpha + (int64_t)(v0 + t.srow[0]) * fb.lda, (int64_t)64 * fb.lda, fb.gamma,
                                 klist + (int64_t)p * k_tiles, lo, hi, acc);
            }
            // first chunk of the pair: the full slab; a continuation: this block's tile behind it (GemmPlan::c_floats)
            const bool cont = L != first_block[p];
            tile_store(t, cont ? C + slab_stride + (int64_t)L * (256 * 256) : C, cont ? 256 : ldc, cont ? 0 : tm, cont ? 0 : tn, acc);
        }
        pos += u_hi - u_lo;
        ++p;
    }
}

// --------------------------------------------------------------------------- //
// Zero-tile bookkeeping
// --------------------------------------------------------------------------- //
// nz[tile][kt] = 1 iff the 256-row x 32-column block of X has a non-zero entry.
__global__ void k_tile_nonzero(const float* __restrict__ X, int ld, int k_tiles, uint8_t* __restrict__ nz) {
    const int tile = blockIdx.y;
    const int row = tile * 256 + threadIdx.x;
    const f32x4* p = (const f32x4*)(X + (int64_t)row * ld);
    for (int j = 0; j < 8; ++j) {
        const int kt = blockIdx.x * 8 + j;
        if (kt >= k_tiles) break;                       // block-uniform
        int f = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 v = p[kt * 8 + c];
            f |= (v[0] != 0.f) | (v[1] != 0.f) | (v[2] != 0.f) | (v[3] != 0.f);
        }
        const int any = __syncthreads_or(f);
        if (threadIdx.x == 0) nz[(int64_t)tile * k_tiles + kt] = any ? 1 : 0;
    }
}

// One block per (m-tile, n-tile[, batch]) pair: compact the K tiles where both operands are non-zero.
// B's zero structure: G > 0 -> per row group, rows [g*v_group, (g+1)*v_group) share nzB[g][kt] (Gamma rows of
// one (action, observation)); rows >= G*v_group (magnitude and reward rows) may touch any group or the extra
// support row nzB[G] (so nzB is [G+1][k_tiles]); G == 0 -> per n-tile flags nzB[batch][tn][kt]; nzB == nullptr
// -> dense.
// irr != nullptr (fused projection): an entry of a tile that lies inside one row group carries, in bit KL_IRR_BIT, whether
// its K tile needs gathers for that group's action (irr[a][kt], a = group / groups_per_action); consumers mask it off.
__global__ void k_build_klists(const uint8_t* __restrict__ nzA, const uint8_t* __restrict__ nzB, int G,
                               int v_group, int n_rows, int tiles_m, int k_tiles, int chunk_len, int force_dense,
                               int* __restrict__ klist, int* __restrict__ kcount, int* __restrict__ nchunks,
                               const int32_t* __restrict__ irr, int groups_per_action) {
    __shared__ int wcount[4];
    __shared__ int total;
    const int tiles_n = gridDim.x / tiles_m;
    const int batch = blockIdx.y;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int pair = (batch * tiles_n + tn) * tiles_m + tm;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r0 = tn * 256;
    int r1 = r0 + 255;
    if (r1 >= n_rows) r1 = n_rows - 1;
    int g0 = 0, g1 = -1;                                // group range touched by this n-tile
    if (nzB != nullptr && G > 0 && r0 < n_rows) {
        if (r1 >= G * v_group) {                        // tail tile: magnitude + reward rows; nzB has a G-th row
            g0 = 0;                                     // for the reward rows' support
            g1 = G;
        } else {
            g0 = r0 / v_group;
            g1 = r1 / v_group;
        }
    }
    if (tid == 0) total = 0;
    __syncthreads();
    for (int base = 0; base < k_tiles; base += 256) {
        const int kt = base + tid;
        int f = 0;
        if (force_dense) {
            f = kt < k_tiles;
        } else if (kt < k_tiles && r0 < n_rows && nzA[(int64_t)tm * k_tiles + kt]) {
            if (nzB == nullptr) {
                f = 1;
            } else if (G == 0) {
                f = nzB[((int64_t)batch * tiles_n + tn) * k_tiles + kt];
            } else {
                for (int g = g0; g <= g1; ++g) f |= nzB[(int64_t)g * k_tiles + kt];
            }
        }
        const unsigned long long mask = __ballot(f);
        if (lane == 0) wcount[wid] = __popcll(mask);
        __syncthreads();
        int off = total;
        for (int w = 0; w < wid; ++w) off += wcount[w];
        if (f) {
            int entry = kt;
            if (irr != nullptr && g0 == g1 && g1 < G && irr[(int64_t)(g0 / groups_per_action) * k_tiles + kt]) entry |= 1 << KL_IRR_BIT;
            klist[(int64_t)pair * k_tiles + off + __popcll(mask & ((1ull << lane) - 1ull))] = entry;
        }
        __syncthreads();
        if (tid == 0) total += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (tid == 0) {
        kcount[pair] = total;
        nchunks[pair] = (total + chunk_len - 1) / chunk_len;    // stream-K overwrites this in k_streamk_plan
    }
}

// Every K tile of every pair listed, zero or not (PBVI_GEMM_DENSE in the environment, or pbvi_debug_gemm_dense):
// what BASELINE's "dense backup" configuration is measured with.  Results are unchanged (zero tiles add +0).
static int g_force_dense = -1;
int gemm_force_dense() {
    if (g_force_dense < 0) g_force_dense = getenv("PBVI_GEMM_DENSE") ? (atoi(getenv("PBVI_GEMM_DENSE")) != 0) : 0;
    return g_force_dense;
}
int set_gemm_force_dense(int enable) {
    const int prev = gemm_force_dense();
    g_force_dense = enable ? 1 : 0;
    return prev;
}

int choose_chunk_len(int tiles_mn, int k_tiles) {
    if (const char* env = getenv("PBVI_GEMM_CHUNK")) {
        const int v = atoi(env);
        if (v > 0) return v < k_tiles ? v : k_tiles;
    }
    const int64_t total = (int64_t)tiles_mn * k_tiles;
    int64_t len = total / (256 * 8);
    if (len < 8) len = 8;
    if (len > 96) len = 96;
    if (len > k_tiles) len = k_tiles;
    return (int)len;
}

static int device_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

hipError_t launch_tile_nonzero_f32(const float* X, int ld, int rows_pad, int k_tiles, uint8_t* nz, hipStream_t stream) {
    dim3 grid((k_tiles + 7) / 8, rows_pad / 256);
    hipLaunchKernelGGL(k_tile_nonzero, grid, dim3(256), 0, stream, X, ld, k_tiles, nz);
    return hipGetLastError();
}

static hipError_t set_lds_attr() {
    static bool done = false;
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm_nt_f32_mfma, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       GEMM_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_gemm_nt_f32_streamk, hipFuncAttributeMaxDynamicSharedMemorySize,
                            GEMM_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute((const void*)k_gemm_nt_f32_streamk_fused, hipFuncAttributeMaxDynamicSharedMemorySize,
                            GEMM_LDS_BYTES);
    if (e != hipSuccess) return e;
#define PBVI_FUSED_R_ATTR(RR)                                                                                          \
    e = hipFuncSetAttribute((const void*)k_gemm_nt_f32_streamk_fused_r<RR>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                            GEMM_LDS_BYTES_R);                                                                          \
    if (e != hipSuccess) return e;
    PBVI_FUSED_R_ATTR(2) PBVI_FUSED_R_ATTR(3) PBVI_FUSED_R_ATTR(4) PBVI_FUSED_R_ATTR(5) PBVI_FUSED_R_ATTR(6) PBVI_FUSED_R_ATTR(7)
#undef PBVI_FUSED_R_ATTR
    done = true;
    return hipSuccess;
}

hipError_t launch_gemm_nt_f32(const float* A, int lda, const float* B, int ldb, float* C, const GemmPlan& pl,
                              const uint8_t* nzA, const uint8_t* nzB, int G, int v_group, int n_rows, int* klist,
                              int* kcount, int* nchunks, hipStream_t stream, int batch, int64_t batch_stride_b,
                              int64_t batch_stride_c, int* streamk_ws, hipStream_t list_stream, hipEvent_t list_event,
                              hipEvent_t ev_before, hipEvent_t ev_after, const FusedB* fused) {
    hipError_t e = set_lds_attr();
    if (e != hipSuccess) return e;
    if (batch < 1 || batch > 65535) return hipErrorInvalidValue;
    const int pairs = pl.tiles_m * pl.tiles_n;
    const int force_dense = gemm_force_dense();          // benchmark / debug: list every tile (a true-dense GEMM)
    // The tile lists and the stream-K plan depend on the zero maps only, not on the operands' values: with a
    // list_stream they are built beside whatever `stream` is still doing (the Gamma projection) and the GEMM waits.
    hipStream_t ls = (list_stream != nullptr && list_event != nullptr) ? list_stream : stream;
    if (fused != nullptr && !pl.streamk) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_build_klists, dim3(pairs, batch), dim3(256), 0, ls, nzA, nzB, G, v_group, n_rows,
                       pl.tiles_m, pl.k_tiles, pl.chunk_len, force_dense, klist, kcount, nchunks,
                       fused != nullptr ? fused->irr : nullptr, fused != nullptr ? fused->O : 1);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (pl.streamk) {
        if (batch != 1 || streamk_ws == nullptr) return hipErrorInvalidValue;
        int* prefix = streamk_ws;                        // [pairs+1]
        int* start_pair = prefix + pairs + 1;            // [nblocks]
        int* first_block = start_pair + pl.nblocks;      // [pairs]
        int* plan = first_block + pairs;                 // [2]
        static const int ovh = getenv("PBVI_STREAMK_OVH") ? atoi(getenv("PBVI_STREAMK_OVH")) : 1;   // per-pair fixed cost, in tile-steps
        hipLaunchKernelGGL(k_streamk_plan, dim3(1), dim3(256), 0, ls, kcount, pairs, pl.nblocks, pl.k_tiles,
                           pl.max_chunks - 1, ovh, prefix, start_pair, first_block, nchunks, plan);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        if (ls != stream) {
            if ((e = hipEventRecord(list_event, ls)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(stream, list_event, 0)) != hipSuccess) return e;
        }
        if (fused != nullptr && fused->R > 1) {
            switch (fused->R) {
#define PBVI_FUSED_R_CASE(RR)                                                                                                   \
    case RR:                                                                                                                    \
        hipLaunchKernelGGL(k_gemm_nt_f32_streamk_fused_r<RR>, dim3(pl.nblocks), dim3(512), GEMM_LDS_BYTES_R, stream, A, lda, B, \
                           ldb, *fused, C, pl.ldc, pl.slab_stride, pl.tiles_m, pairs, pl.k_tiles, klist, kcount, prefix,        \
                           start_pair, first_block, plan, ovh);                                                                 \
        break;
                PBVI_FUSED_R_CASE(2) PBVI_FUSED_R_CASE(3) PBVI_FUSED_R_CASE(4) PBVI_FUSED_R_CASE(5) PBVI_FUSED_R_CASE(6) PBVI_FUSED_R_CASE(7)
#undef PBVI_FUSED_R_CASE
                default:
                    return hipErrorInvalidValue;
            }
        } else if (fused != nullptr)
            hipLaunchKernelGGL(k_gemm_nt_f32_streamk_fused, dim3(pl.nblocks), dim3(512), GEMM_LDS_BYTES, stream, A, lda, B, ldb,
                               *fused, C, pl.ldc, pl.slab_stride, pl.tiles_m, pairs, pl.k_tiles, klist, kcount, prefix, start_pair,
                               first_block, plan, ovh);
        else
            hipLaunchKernelGGL(k_gemm_nt_f32_streamk, dim3(pl.nblocks), dim3(512), GEMM_LDS_BYTES, stream, A, lda, B, ldb, C,
                               pl.ldc, pl.slab_stride, pl.tiles_m, pairs, pl.k_tiles, klist, kcount, prefix, start_pair,
                               first_block, plan, ovh);
        return hipGetLastError();
    }
    if (ls != stream) {
        if ((e = hipEventRecord(list_event, ls)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, list_event, 0)) != hipSuccess) return e;
    }
    const int64_t groups = ((int64_t)pl.tiles_n * pl.max_chunks + 7) / 8 * 8;
    const int64_t total = groups * pl.tiles_m;
    if (total <= 0 || total > 0x7fffffff) return hipErrorInvalidValue;
    if (ev_before != nullptr && (e = hipEventRecord(ev_before, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(k_gemm_nt_f32_mfma, dim3((unsigned)total, batch), dim3(512), GEMM_LDS_BYTES, stream, A, lda, B,
                       ldb, C, pl.ldc, pl.slab_stride, pl.tiles_m, pl.tiles_n, pl.k_tiles, pl.chunk_len, pl.max_chunks,
                       klist, kcount, batch_stride_b, batch_stride_c);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_after != nullptr && (e = hipEventRecord(ev_after, stream)) != hipSuccess) return e;
    return hipSuccess;
}

GemmPlan make_gemm_plan(int M_pad, int N_pad, int K_pad, bool single_chunk) {
    GemmPlan pl;
    pl.tiles_m = M_pad / GEMM_BM;
    pl.tiles_n = N_pad / GEMM_BN;
    pl.k_tiles = K_pad / GEMM_BK;
    pl.ldc = N_pad;
    pl.slab_stride = (int64_t)M_pad * N_pad;
    static const int no_streamk = getenv("PBVI_GEMM_NO_STREAMK") ? atoi(getenv("PBVI_GEMM_NO_STREAMK")) : 0;
    pl.streamk = !single_chunk && !no_streamk;
    pl.nblocks = device_cu_count();
    if (single_chunk) {
        pl.chunk_len = pl.k_tiles;
        pl.max_chunks = 1;
    } else if (pl.streamk) {
        // a pair's list (<= k_tiles steps) is split over at most max_split+1 consecutive blocks
        const int max_split = 8;
        pl.chunk_len = pl.k_tiles;          // only used for the (overwritten) provisional nchunks
        pl.max_chunks = max_split + 1;
        pl.c_floats = pl.slab_stride + (int64_t)pl.nblocks * (256 * 256);
        return pl;
    } else {
        pl.chunk_len = choose_chunk_len(pl.tiles_m * pl.tiles_n, pl.k_tiles);
        pl.max_chunks = (pl.k_tiles + pl.chunk_len - 1) / pl.chunk_len;
    }
    pl.c_floats = (int64_t)pl.max_chunks * pl.slab_stride;
    return pl;
}

size_t streamk_workspace_ints(const GemmPlan& pl) {
    const size_t pairs = (size_t)pl.tiles_m * pl.tiles_n;
    return (pairs + 1) + (size_t)pl.nblocks + pairs + 2;
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

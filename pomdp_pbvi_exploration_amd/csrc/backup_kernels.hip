// HBM-bound kernels of the PBVI backup on gfx950: Gamma projection (padded-ELL SpMM),
// dead-triple test, wavefront argmax with near-tie detection, fp64 refinement, action
// selection, alpha' gather-sum, belief-dominance test and point-wise domination prune.
// Reference statements: src/pomdp.py:1485-1515, src/mdp.py:857-866.
//
// All kernels read the model tables in the [.][S_pad] layouts of ModelView (coalesced
// along s, 64-wide wavefronts) and reduce with wave shuffles in a fixed order, so
// results are deterministic run to run.
#include "backup_kernels.h"

#include <cstdlib>
#include <limits>

namespace pbvi {

// ------------------------------------------------------------------------- //
// helpers
// ------------------------------------------------------------------------- //
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum over the block (blockDim.x = 256); every thread gets the result.  sh: >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wid] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

// Tile list of one belief: the K tiles (32 states) that hold belief mass; list == nullptr means every tile.
struct TileList {
    const int32_t* list;
    int count;
    int first = 0;                       // list == nullptr: the tiles are first, first + 1, ...
    __device__ __forceinline__ int at(int i) const { return list ? list[i] : first + i; }
};

// Partial (this thread's share) of  sum_s b[s] * gamma * sum_r rto[a][o][r][s] * alpha_v[rs[a][r][s]]  in f64,
// over the belief's non-zero tiles only (zero tiles add exact zeros).  blockDim.x = 256 = 8 half-waves;
// half-wave q takes list entries q, q+8, ...; four entries per half-wave are in flight at once with
// unconditional loads (these dots run in latency-bound refinement kernels).
template <typename T>
__device__ __forceinline__ double proj_dot_partial(const T* __restrict__ brow, const T* __restrict__ arow,
                                                   const ModelView<T>& mv, int a, int o, double gamma, TileList tl) {
    const int32_t* __restrict__ rs = mv.rs + (int64_t)a * mv.R * mv.S_pad;
    const T* __restrict__ rto = mv.rto + (int64_t)(a * mv.O + o) * mv.R * mv.S_pad;
    const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = q; i0 < tl.count; i0 += 32) {
        int s[4];
        double w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + 8 * j;
            const bool ok = i < tl.count;
            s[j] = tl.at(ok ? i : i0) * 32 + l;               // < S_pad; pads hold rs = 0, rto = 0, b = 0
            w[j] = ok ? 1.0 : 0.0;
        }
        double g[4] = {0.0, 0.0, 0.0, 0.0};
        for (int r = 0; r < mv.R; ++r) {
            const int64_t ro = (int64_t)r * mv.S_pad;
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] += (double)rto[ro + s[j]] * (double)arow[rs[ro + s[j]]];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w[j] * ((double)brow[s[j]] * (gamma * g[j]));
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// brow must be padded to a multiple of 32 with zeros (engine belief rows are); arow valid for s < S.
template <typename T>
__device__ __forceinline__ double plain_dot_partial(const T* __restrict__ brow, const T* __restrict__ arow, int S,
                                                    TileList tl) {
    const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = q; i0 < tl.count; i0 += 32) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + 8 * j;
            const bool ok = i < tl.count;
            int s = tl.at(ok ? i : i0) * 32 + l;
            const bool in = ok && s < S;
            s = s < S ? s : 0;
            acc[j] += in ? (double)brow[s] * (double)arow[s] : 0.0;
        }
    }
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}

// The refinement's dots over a tile list, for up to 4 candidates at once with the belief (and, for exact_dots4, the model
// tables) read once.  A tile is 32 consecutive states = 128 bytes of an fp32 row: every lane takes 16 bytes of it (4 floats /
// 2 doubles), so the 256 threads cover 32 (fp32) or 16 (fp64) list entries per pass, two passes in flight.  (One state per
// lane -- 8 tiles per pass -- made these loops a chain of one memory round trip per 16 tiles: 52 us for the 825 tiles of an
// average R = 5 entry, 40 us for the 364 of an R = 1 entry, measured with wall_clock64 inside k_refine.)
// Every thread gets the 4 sums.
#ifndef PBVI_DOTS_IN_FLIGHT
#define PBVI_DOTS_IN_FLIGHT 2
#endif
constexpr int DOTS_IN_FLIGHT = PBVI_DOTS_IN_FLIGHT;       // passes in flight per thread
__device__ __forceinline__ void dots4_reduce(double (&acc)[4], double (&out)[4], double* sh /* >= 16 doubles */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = wave_sum(acc[c]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) sh[wid * 4 + c] = acc[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) out[c] = ((sh[c] + sh[4 + c]) + sh[8 + c]) + sh[12 + c];
}

// Level-1 screen: fp64 sums of  b[s] * G_c[s]  over projected fp32 rows (RefineWork::gam); fp32 engines only.
template <typename T>
__device__ __forceinline__ void gamma_dots4(const T* __restrict__ brow, const float* const (&grow)[4], TileList tl,
                                            double (&out)[4], double* sh /* >= 16 doubles */) {
    static_assert(sizeof(T) == 4, "projected rows are screened by fp32 engines only");
    typedef float F4 __attribute__((ext_vector_type(4)));
    constexpr int NF = DOTS_IN_FLIGHT;
    const int q = threadIdx.x >> 3, l = (threadIdx.x & 7) * 4;
    const F4 zero = {0.f, 0.f, 0.f, 0.f};
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = q; i0 < tl.count; i0 += 32 * NF) {
        int st[NF];
        F4 bv[NF], gv[NF][4];
#pragma unroll
        for (int h = 0; h < NF; ++h) {
            const int i = i0 + 32 * h;
            const bool ok = i < tl.count;
            st[h] = tl.at(ok ? i : i0) * 32 + l;
            bv[h] = ok ? *(const F4*)((const float*)brow + st[h]) : zero;
        }
#pragma unroll
        for (int h = 0; h < NF; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c) gv[h][c] = *(const F4*)(grow[c] + st[h]);
#pragma unroll
        for (int h = 0; h < NF; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[c] += (double)bv[h][k] * (double)gv[h][c][k];
    }
    dots4_reduce(acc, out, sh);
}

// Exact scores from the operands:  sum_s b[s] * gamma * sum_r rto[a][o][r][s] * alpha_c[rs[a][r][s]]  (PROJ) or
// sum_s b[s] * alpha_c[s], products of fp32 operands exact in fp64, fp64 sums.  Successors of consecutive states are
// usually consecutive (grid moves): then the 4 (2) alpha values of a lane are one element-aligned 16-byte load.
// Identical candidate rows give identical sums (same operations in the same order), so exact ties still go to the lower index.
template <typename T, bool PROJ>
__device__ __forceinline__ void exact_dots4(const T* __restrict__ brow, const T* const (&arow)[4], int nc /* candidates in use */,
                                            const ModelView<T>& mv, int a, int o, double gamma, TileList tl, double (&out)[4],
                                            double* sh /* >= 16 doubles */) {
    constexpr int NS = 16 / (int)sizeof(T);                 // states per lane
    constexpr int LPT = 32 / NS, TPP = 256 / LPT;           // lanes per tile, tiles per pass
    typedef T TN __attribute__((ext_vector_type(NS)));
    typedef int IN __attribute__((ext_vector_type(NS)));
    const int32_t* __restrict__ rs = mv.rs + (int64_t)a * mv.R * mv.S_pad;
    const T* __restrict__ rto = mv.rto + (int64_t)(a * mv.O + o) * mv.R * mv.S_pad;
    const int q = threadIdx.x / LPT, l = (threadIdx.x % LPT) * NS;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    constexpr int NF = DOTS_IN_FLIGHT;
    if constexpr (PROJ) {
        // Software-pipelined over (pass, successor slot): the tables of the next step (belief, successor indices, weights)
        // are requested before the alpha gathers of the current one -- one round trip per step instead of two -- and the
        // gathers of both passes of a step leave together (one branch on "every lane's successors are consecutive").
        struct Step {
            int st[NF];
            bool ok[NF];
            TN bv[NF], w[NF];
            IN idx[NF];
        };
        auto fetch = [&](int i0, int r, Step& x) {
            const int64_t ro = (int64_t)r * mv.S_pad;
#pragma unroll
            for (int h = 0; h < NF; ++h) {
                const int i = i0 + TPP * h;
                x.ok[h] = i < tl.count;
                x.st[h] = tl.at(x.ok[h] ? i : i0) * 32 + l;
                x.bv[h] = *(const TN*)(brow + x.st[h]);
                x.idx[h] = *(const IN*)(rs + ro + x.st[h]);
                x.w[h] = *(const TN*)(rto + ro + x.st[h]);
            }
        };
        int i0 = q, r = 0;
        bool live = i0 < tl.count;
        Step cur, nxt;
        if (live) fetch(i0, r, cur);
        while (live) {
            int ni0 = i0, nr = r + 1;
            if (nr == mv.R) {
                nr = 0;
                ni0 = i0 + NF * TPP;
            }
            const bool nlive = ni0 < tl.count;
            if (nlive) fetch(ni0, nr, nxt);
            bool contig = true;
#pragma unroll
            for (int h = 0; h < NF; ++h)
#pragma unroll
                for (int k = 1; k < NS; ++k) contig = contig && (cur.idx[h][k] == cur.idx[h][0] + k);
            double bw[NF][NS];
#pragma unroll
            for (int h = 0; h < NF; ++h)
#pragma unroll
                for (int k = 0; k < NS; ++k) bw[h][k] = (cur.ok[h] ? gamma * (double)cur.bv[h][k] : 0.0) * (double)cur.w[h][k];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nc) {                                // block-uniform: no gathers for unused candidate slots
                    TN av[NF];
                    if (contig) {
#pragma unroll
                        for (int h = 0; h < NF; ++h) {
                            const T* p = arow[c] + cur.idx[h][0];
#pragma unroll
                            for (int k = 0; k < NS; ++k) av[h][k] = p[k];
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < NF; ++h)
#pragma unroll
                            for (int k = 0; k < NS; ++k) av[h][k] = arow[c][cur.idx[h][k]];
                    }
#pragma unroll
                    for (int h = 0; h < NF; ++h)
#pragma unroll
                        for (int k = 0; k < NS; ++k) acc[c] += bw[h][k] * (double)av[h][k];
                }
            cur = nxt;
            i0 = ni0;
            r = nr;
            live = nlive;
        }
    } else {
        for (int i0 = q; i0 < tl.count; i0 += NF * TPP) {
            int st[NF];
            double bg[NF][NS];                               // b[s], zero for a slot that is not there
#pragma unroll
            for (int h = 0; h < NF; ++h) {
                const int i = i0 + TPP * h;
                const bool ok = i < tl.count;
                st[h] = tl.at(ok ? i : i0) * 32 + l;
                const TN bv = *(const TN*)(brow + st[h]);
#pragma unroll
                for (int k = 0; k < NS; ++k) bg[h][k] = ok ? (double)bv[k] : 0.0;
            }
#pragma unroll
            for (int h = 0; h < NF; ++h)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < nc) {
                        const TN av = *(const TN*)(arow[c] + st[h]);       // inside the row's S_pad columns; valid for s < S only
#pragma unroll
                        for (int k = 0; k < NS; ++k) acc[c] += (st[h] + k < mv.S) ? bg[h][k] * (double)av[k] : 0.0;
                    }
        }
    }
    dots4_reduce(acc, out, sh);
}

// ------------------------------------------------------------------------- //
// support mask of RTO (engine creation)
// ------------------------------------------------------------------------- //
template <typename T>
__global__ void k_support(ModelView<T> mv, uint8_t* __restrict__ sup) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int ao = blockIdx.y;
    if (s >= mv.S_pad) return;
    uint8_t f = 0;
    for (int r = 0; r < mv.R; ++r) f |= (mv.rto[((int64_t)ao * mv.R + r) * mv.S_pad + s] != T(0)) ? 1 : 0;
    sup[(int64_t)ao * mv.S_pad + s] = f;
}

template <typename T>
hipError_t launch_support(ModelView<T> mv, uint8_t* sup, hipStream_t st) {
    dim3 grid((mv.S_pad + 255) / 256, mv.A * mv.O);
    hipLaunchKernelGGL(k_support<T>, grid, dim3(256), 0, st, mv, sup);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// K1-sparse: Gamma projection (padded-ELL SpMM).  src/pomdp.py:1485-1491
// One thread per state s (coalesced), 4 alpha-vectors x 4 observations per pass so
// every gathered alpha value and every table load is reused from registers.
// ------------------------------------------------------------------------- //
template <typename T, int OB /* observations per pass: min(O, 4) */>
__global__ void k_project(const T* __restrict__ alpha, int lda, int V, ModelView<T> mv, T gamma,
                          T* __restrict__ gam, int ldg, const uint8_t* __restrict__ need, int k_tiles,
                          const uint8_t* __restrict__ mat /* per 256-row tile of Gamma: write it? (nullptr = all) */,
                          const int* __restrict__ vlist /* blocks of 4 alpha rows to visit (nullptr = blockIdx.y) */,
                          const int32_t* __restrict__ irr /* [A][k_tiles]: K tiles written whatever `mat` says (fused GEMM, R > 1:
                                                             it reads the tiles it cannot generate), or nullptr */,
                          int nx, int ny, int tile_x, int tile_y /* 1-D grid in L2-tiled order (see below); tile_x <= 0: 3-D grid */,
                          const int32_t* __restrict__ ctile /* compact layout: 256-row tile t of Gamma lives at tile ctile[t], or nullptr */) {
#pragma clang fp contract(off)   // einsum then scale: sum_r (rto*alpha), one rounding per op, as the reference
    // NS consecutive states per thread = 16 bytes of every table load and Gamma store (float: 4, double: 2; S_pad is a
    // multiple of 32), 4 alpha-vectors x up to 4 observations per pass: 16 NS-wide accumulators.  (With 4 doubles per
    // thread the accumulators alone were 128 VGPRs and the kernel spilled 166 registers to scratch: 1.9 ms instead
    // of 0.6 at C4.)
    constexpr int NS = 16 / (int)sizeof(T);
    typedef T TN __attribute__((ext_vector_type(NS)));
    typedef int IN __attribute__((ext_vector_type(NS)));
    // Block -> (state chunk x, alpha block y, action a).  With a list of alpha blocks (the fused GEMM's leftovers) the grid
    // is (x, list entry, a).  Otherwise the grid is one-dimensional and walked in an L2-TILED order: an XCD (blocks are
    // dealt round-robin over the 8 XCDs, each with its own 4 MiB L2) takes a contiguous eighth of the sequence
    //   for x-group (tile_x state chunks) / for y-group (tile_y alpha blocks) / for y / for x / for a
    // so that the alpha segments it gathers from -- the chunk's own states and the +-W rows a grid move reaches, for
    // every successor slot and every action -- and the tables of those state chunks stay in ITS L2 while they are
    // re-used: with (x, y, a) dealt out in launch order the kernel fetched 1.97 GB to write 1.45 GB at |S| = 30000,
    // R = 5 (16 passes over the alpha set instead of 1; PMC, profiles/r03_*).
    int xb, yb, a;
    if (vlist != nullptr || tile_x <= 0) {
        xb = blockIdx.x;
        yb = vlist != nullptr ? vlist[blockIdx.y] : (int)blockIdx.y;
        a = blockIdx.z;
    } else {
        const int nb = gridDim.x, bid = blockIdx.x, xcd = bid & 7, qq = nb >> 3, rr = nb & 7;
        int L = ((xcd < rr) ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (bid >> 3);
        const int A = mv.A;
        const int nxg = (nx + tile_x - 1) / tile_x, nyg = (ny + tile_y - 1) / tile_y;
        const int full_x = (nxg - 1) * tile_x * ny * A;
        int xg, xw;
        if (L < full_x) {
            xg = L / (tile_x * ny * A);
            L -= xg * (tile_x * ny * A);
            xw = tile_x;
        } else {
            xg = nxg - 1;
            L -= full_x;
            xw = nx - tile_x * (nxg - 1);
        }
        const int full_y = (nyg - 1) * tile_y * xw * A;
        int yg;
        if (L < full_y) {
            yg = L / (tile_y * xw * A);
            L -= yg * (tile_y * xw * A);
        } else {
            yg = nyg - 1;
            L -= full_y;
        }
        const int y_in = L / (xw * A);
        L -= y_in * (xw * A);
        xb = xg * tile_x + L / A;
        a = L % A;
        yb = yg * tile_y + y_in;
    }
    const int s = (xb * 256 + threadIdx.x) * NS;
    if (s >= mv.S_pad) return;
    const int v0 = yb * 4;
    const int nv = (V - v0) < 4 ? (V - v0) : 4;
    const int kt = s >> 5;                                  // GEMM K tile of these states (32 states per tile)
    const bool irr_here = irr != nullptr && irr[(int64_t)a * k_tiles + kt] != 0;
    TN zero;
#pragma unroll
    for (int j = 0; j < NS; ++j) zero[j] = T(0);
    for (int o0 = 0; o0 < mv.O; o0 += OB) {
        const int no = (mv.O - o0) < OB ? (mv.O - o0) : OB;
        // Gamma tiles the score GEMM never reads (no RTO support, or no belief mass in any row
        // block) are not computed or written: `need` is exactly the GEMM's tile-list criterion.
        bool want[OB];
#pragma unroll
        for (int oj = 0; oj < OB; ++oj) want[oj] = true;
        if (need != nullptr && v0 + nv < V) {              // the magnitude row (last of the V rows) is always written:
            bool any = false;                               // the tail tile it lives in is listed for every group's support
#pragma unroll
            for (int oj = 0; oj < OB; ++oj) {
                want[oj] = (oj < no) && need[((int64_t)a * mv.O + o0 + oj) * k_tiles + kt];
                any |= want[oj];
            }
            if (!any) continue;
        }
        if (mat != nullptr && v0 + nv < V) {               // tiles the fused GEMM generates itself are not projected
            bool any = false;
#pragma unroll
            for (int oj = 0; oj < OB; ++oj) {
                const int64_t r0 = ((int64_t)a * mv.O + o0 + oj) * (V - 1) + v0;
                want[oj] = want[oj] && (oj < no) && (irr_here || (mat[r0 >> 8] | mat[(r0 + nv - 1) >> 8]));
                any |= want[oj];
            }
            if (!any) continue;
        }
        TN acc[4][OB];
#pragma unroll
        for (int vj = 0; vj < 4; ++vj)
#pragma unroll
            for (int oj = 0; oj < OB; ++oj) acc[vj][oj] = zero;
        // Software-pipelined over the successor slots: slot r + 1's indices and weights are requested before slot r's alpha
        // gathers, and the gathers of the four alpha rows leave together behind ONE branch on "this lane's successors are
        // consecutive" (a branch per row made every row's gather its own round trip: 25 dependent rounds per block at R = 5).
        auto load_slot = [&](int r, IN& idx, TN (&w)[OB]) {
            idx = *(const IN*)(mv.rs + ((int64_t)a * mv.R + r) * mv.S_pad + s);
#pragma unroll
            for (int oj = 0; oj < OB; ++oj)
                w[oj] = (oj < no && want[oj]) ? *(const TN*)(mv.rto + (((int64_t)a * mv.O + o0 + oj) * mv.R + r) * mv.S_pad + s)
                                               : zero;
        };
        IN idx;
        TN w[OB];
        for (int r = 0; r < mv.R; ++r) {
            load_slot(r, idx, w);
            bool contig = true;
#pragma unroll
            for (int j = 1; j < NS; ++j) contig = contig && (idx[j] == idx[0] + j);
            // successors of consecutive states are usually consecutive (grid moves): one 16-byte (element-aligned) load
            // per alpha row instead of NS gathers
            TN av[4];
            if (contig) {
#pragma unroll
                for (int vj = 0; vj < 4; ++vj) {
                    const T* p = alpha + (int64_t)(v0 + (vj < nv ? vj : 0)) * lda + idx[0];
#pragma unroll
                    for (int j = 0; j < NS; ++j) av[vj][j] = p[j];
                }
            } else {
#pragma unroll
                for (int vj = 0; vj < 4; ++vj) {
                    const T* arow = alpha + (int64_t)(v0 + (vj < nv ? vj : 0)) * lda;
#pragma unroll
                    for (int j = 0; j < NS; ++j) av[vj][j] = arow[idx[j]];
                }
            }
#pragma unroll
            for (int vj = 0; vj < 4; ++vj)
#pragma unroll
                for (int oj = 0; oj < OB; ++oj) acc[vj][oj] = acc[vj][oj] + w[oj] * av[vj];
        }
#pragma unroll
        for (int vj = 0; vj < 4; ++vj)
#pragma unroll
            for (int oj = 0; oj < OB; ++oj)
                if (vj < nv && oj < no && want[oj]) {
                    const int64_t ao = (int64_t)a * mv.O + o0 + oj;
                    const int v = v0 + vj;
                    // alpha rows -> group-major rows; the magnitude row (v == V-1 of the Vt rows) -> tail
                    const int64_t row = (v < V - 1) ? ao * (V - 1) + v : (int64_t)mv.A * mv.O * (V - 1) + ao;
                    if (mat == nullptr || irr_here || mat[row >> 8]) {
                        const int64_t at = ctile != nullptr ? (int64_t)ctile[row >> 8] * 256 + (row & 255) : row;
                        __builtin_nontemporal_store(gamma * acc[vj][oj], (TN*)(gam + at * ldg + s));
                    }
                }
    }
}

template <typename T>
hipError_t launch_project(const T* alpha, int lda, int V, ModelView<T> mv, T gamma, T* gam, int ldg,
                          const uint8_t* need, int k_tiles, hipStream_t st, const uint8_t* mat, const int* vlist, int n_vlist,
                          const int32_t* irr, const int32_t* ctile) {
    if (V <= 0) return hipSuccess;
    if (vlist != nullptr && n_vlist <= 0) return hipSuccess;
    constexpr int NS = 16 / (int)sizeof(T);
    const int nx = (mv.S_pad / NS + 255) / 256, ny = (V + 3) / 4;
    // L2 tiles: 4 state chunks x 6 actions of tables (~0.5 MB each) + 32 alpha blocks x 4 chunks x 16 KiB of alpha
    static const int tile_x = getenv("PBVI_PROJ_TILE_X") ? atoi(getenv("PBVI_PROJ_TILE_X")) : 4;
    static const int tile_y = getenv("PBVI_PROJ_TILE_Y") ? atoi(getenv("PBVI_PROJ_TILE_Y")) : 32;
    const int64_t total = (int64_t)nx * ny * mv.A;
    // observations per pass = accumulator rows per alpha-vector: exactly O when O < 4 (16 fewer VGPRs per missing one)
    auto launch = [&](dim3 grid, int tx, int ty) {
        switch (mv.O < 4 ? mv.O : 4) {
            case 1: hipLaunchKernelGGL((k_project<T, 1>), grid, dim3(256), 0, st, alpha, lda, V, mv, gamma, gam, ldg, need, k_tiles, mat, vlist, irr, nx, ny, tx, ty, ctile); break;
            case 2: hipLaunchKernelGGL((k_project<T, 2>), grid, dim3(256), 0, st, alpha, lda, V, mv, gamma, gam, ldg, need, k_tiles, mat, vlist, irr, nx, ny, tx, ty, ctile); break;
            case 3: hipLaunchKernelGGL((k_project<T, 3>), grid, dim3(256), 0, st, alpha, lda, V, mv, gamma, gam, ldg, need, k_tiles, mat, vlist, irr, nx, ny, tx, ty, ctile); break;
            default: hipLaunchKernelGGL((k_project<T, 4>), grid, dim3(256), 0, st, alpha, lda, V, mv, gamma, gam, ldg, need, k_tiles, mat, vlist, irr, nx, ny, tx, ty, ctile); break;
        }
        return hipGetLastError();
    };
    if (vlist == nullptr && tile_x > 0 && tile_y > 0 && total <= 0x7fffffff) return launch(dim3((unsigned)total), tile_x, tile_y);
    dim3 grid(nx, vlist != nullptr ? n_vlist : ny, mv.A);
    if (grid.y > 65535 || grid.z > 65535) return hipErrorInvalidValue;
    return launch(grid, 0, 0);
}

// need[g][kt]: must Gamma's rows of group g be computed for K tile kt?  Yes iff some 256-row belief block
// has mass in kt AND the score GEMM can list kt for an n-tile that holds rows of g -- i.e. some group
// sharing an n-tile with g (g itself, its neighbours when V % 256 != 0, every group + the reward rows for
// the tail tile) has RTO/ER support in kt.  Where g's own support is empty the projection writes true
// zeros; tiles that are not needed are never read, so they may hold stale data.
__global__ void k_need_tiles(const uint8_t* __restrict__ nzA, int tiles_m, const uint8_t* __restrict__ nzB, int G, int V,
                             int k_tiles, uint8_t* __restrict__ need) {
    // one thread per (K tile, group): the loops over row blocks / neighbouring groups are a handful of loads each
    // (as one thread per K tile looping over all groups this was 22 us of dependent loads in front of the projection)
    const int kt = blockIdx.x * 256 + threadIdx.x;
    const int g = blockIdx.y;
    if (kt >= k_tiles) return;
    int any = 0;
    for (int m = 0; m < tiles_m; ++m) any |= nzA[(int64_t)m * k_tiles + kt];
    int f = 0;
    if (any) {
        const int64_t tail0 = (int64_t)G * V;               // first magnitude row
        const int64_t r0 = (int64_t)g * V, r1 = r0 + V - 1;
        const int64_t t0 = r0 >> 8, t1 = r1 >> 8;           // n-tiles holding rows of g
        if ((t1 << 8) + 255 >= tail0) {                     // shares a tile with the tail rows: support of anything there
            for (int x = 0; x <= G; ++x) f |= nzB[(int64_t)x * k_tiles + kt];
        } else {
            const int g0 = (int)((t0 << 8) / V), g1 = (int)(((t1 << 8) + 255) / V);
            for (int x = g0; x <= g1 && x < G; ++x) f |= nzB[(int64_t)x * k_tiles + kt];
        }
    }
    need[(int64_t)g * k_tiles + kt] = (any && f) ? 1 : 0;
}

hipError_t launch_need_tiles(const uint8_t* nzA, int tiles_m, const uint8_t* nzB, int AO, int V, int k_tiles,
                             uint8_t* need, hipStream_t st) {
    if (AO > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_need_tiles, dim3((k_tiles + 255) / 256, AO), dim3(256), 0, st, nzA, tiles_m, nzB, AO, V, k_tiles, need);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// dead triples: supp(b) disjoint from supp(RTO[:,a,o,:])  <=>  every score is exactly 0
// ------------------------------------------------------------------------- //
// Two levels: a belief's non-zero K tiles (32 states) are flagged first; element-level overlap with
// supp(RTO[:,a,o,:]) is then checked only in tiles where both the belief and the (a,o) support tile map
// nzB are non-zero -- for the "goal" observation that is one tile instead of |S| states.
template <typename T>
__global__ void k_dead(const T* __restrict__ bel, int ldb, ModelView<T> mv,
                       const unsigned long long* __restrict__ nzBw /* [A*O][ceil(k_tiles/64)] support tiles as bit words */,
                       int k_tiles, uint8_t* __restrict__ dead, int32_t* __restrict__ btl, int32_t* __restrict__ btc,
                       int* __restrict__ dead_count, const uint8_t* __restrict__ rowflags /* [.][k_tiles] or nullptr */,
                       const int32_t* __restrict__ perm /* row of rowflags = perm[b], or nullptr = b */) {
    extern __shared__ uint8_t dsm[];
    __shared__ int dead_w[4];
    uint8_t* tz = dsm;                                  // [k_tiles] belief has a non-zero in tile
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int AO = mv.A * mv.O;
    const T* brow = bel + (int64_t)b * ldb;
    int n_dead = 0;
    if (rowflags != nullptr) {                          // the block's indexing pass left them (engine.hip, k_row_flags)
        const uint8_t* f = rowflags + (int64_t)(perm ? perm[b] : b) * k_tiles;
        for (int kt = tid; kt < k_tiles; kt += 256) tz[kt] = f[kt];
    } else {
        // tile flags: 8 lanes x 4 states per tile
        for (int kt0 = 0; kt0 < k_tiles; kt0 += 32) {
            const int kt = kt0 + (tid >> 3);
            int f = 0;
            if (kt < k_tiles) {
                const int s = kt * 32 + (tid & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) f |= (s + j < mv.S && brow[s + j] != T(0)) ? 1 : 0;
            }
            f |= __shfl_xor(f, 1, 64);
            f |= __shfl_xor(f, 2, 64);
            f |= __shfl_xor(f, 4, 64);
            if (kt < k_tiles && (tid & 7) == 0) tz[kt] = (uint8_t)f;
        }
    }
    __syncthreads();
    if (wid == 0) {   // compact list of this belief's non-zero tiles, kept for the refinement dots
        int base = 0;
        for (int kt0 = 0; kt0 < k_tiles; kt0 += 64) {
            const int kt = kt0 + lane;
            const int f = (kt < k_tiles) ? tz[kt] : 0;
            const unsigned long long m = __ballot(f);
            if (f) btl[(int64_t)b * k_tiles + base + __popcll(m & ((1ull << lane) - 1ull))] = kt;
            base += __popcll(m);
        }
        if (lane == 0) btc[b] = base;
    }
    // the belief's tile flags as bit words (bit t of word w = tile 64 w + t), so that a wave sees the candidate tiles of an
    // (a, o) -- belief mass AND support, nzBw holds the supports in the same form -- after ONE load instead of one
    // dependent load per 64 tiles
    const int kw = (k_tiles + 63) >> 6;
    unsigned long long* tzw = (unsigned long long*)(dsm + ((k_tiles + 7) & ~7));
    for (int w = wid; w < kw; w += 4) {
        const int kt = w * 64 + lane;
        const unsigned long long m = __ballot(kt < k_tiles && tz[kt]);
        if (lane == 0) tzw[w] = m;
    }
    __syncthreads();
    for (int ao = wid; ao < AO; ao += 4) {                // one wave per (a,o)
        const uint8_t* sp = mv.sup + (int64_t)ao * mv.S_pad;
        int found = 0;
        for (int w0 = 0; w0 < kw && !found; w0 += 64) {
            const int wi = w0 + lane;
            const unsigned long long cw = wi < kw ? (tzw[wi] & nzBw[(int64_t)ao * kw + wi]) : 0ull;
            unsigned long long lanes = __ballot(cw != 0ull);
            while (lanes != 0ull && !found) {
                const int src = __ffsll((long long)lanes) - 1;
                lanes &= lanes - 1ull;
                unsigned long long cand = ((unsigned long long)(unsigned)__shfl((int)(cw >> 32), src, 64) << 32) |
                                          (unsigned long long)(unsigned)__shfl((int)(cw & 0xffffffffull), src, 64);
                const int kt0 = (w0 + src) * 64;
                // the wave checks the candidates' states two tiles at a time, one state per lane, and stops at the first
                // overlap -- which for a live triple is the first or second candidate
                while (cand != 0ull && !found) {
                    const int t0 = __ffsll((long long)cand) - 1;
                    cand &= cand - 1ull;
                    int t1 = -1;
                    if (cand != 0ull) {
                        t1 = __ffsll((long long)cand) - 1;
                        cand &= cand - 1ull;
                    }
                    const int t = lane < 32 ? t0 : t1;
                    int f = 0;
                    if (t >= 0) {
                        const int s = (kt0 + t) * 32 + (lane & 31);
                        f = (s < mv.S && sp[s] && brow[s] != T(0)) ? 1 : 0;
                    }
                    found = __any(f);
                }
            }
        }
        if (lane == 0) {
            dead[(int64_t)b * AO + ao] = found ? 0 : 1;
            if (!found) ++n_dead;
        }
    }
    // one atomic per block: 4830 dead triples adding to one word one by one were 63 us of the kernel's 67
    if (dead_count != nullptr) {
        if (lane == 0) dead_w[wid] = n_dead;
        __syncthreads();
        if (tid == 0) {
            const int n = dead_w[0] + dead_w[1] + dead_w[2] + dead_w[3];
            if (n > 0) atomicAdd(dead_count, n);
        }
    }
}

// Non-zero tile lists of the resident beliefs alone (value-max path, where k_dead does not run).
template <typename T>
__global__ void k_belief_tiles(const T* __restrict__ bel, int ldb, int S, int k_tiles, int32_t* __restrict__ btl,
                               int32_t* __restrict__ btc) {
    extern __shared__ uint8_t dsm[];
    uint8_t* tz = dsm;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const T* brow = bel + (int64_t)b * ldb;
    for (int kt0 = 0; kt0 < k_tiles; kt0 += 32) {
        const int kt = kt0 + (tid >> 3);
        int f = 0;
        if (kt < k_tiles) {
            const int s = kt * 32 + (tid & 7) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) f |= (s + j < S && brow[s + j] != T(0)) ? 1 : 0;
        }
        f |= __shfl_xor(f, 1, 64);
        f |= __shfl_xor(f, 2, 64);
        f |= __shfl_xor(f, 4, 64);
        if (kt < k_tiles && (tid & 7) == 0) tz[kt] = (uint8_t)f;
    }
    __syncthreads();
    if (wid == 0) {
        int base = 0;
        for (int kt0 = 0; kt0 < k_tiles; kt0 += 64) {
            const int kt = kt0 + lane;
            const int f = (kt < k_tiles) ? tz[kt] : 0;
            const unsigned long long m = __ballot(f);
            if (f) btl[(int64_t)b * k_tiles + base + __popcll(m & ((1ull << lane) - 1ull))] = kt;
            base += __popcll(m);
        }
        if (lane == 0) btc[b] = base;
    }
}

template <typename T>
hipError_t launch_belief_tiles(const T* bel, int ldb, int B, int S, int k_tiles, int32_t* btl, int32_t* btc,
                               hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_belief_tiles<T>, dim3(B), dim3(256), (size_t)k_tiles, st, bel, ldb, S, k_tiles, btl, btc);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_dead(const T* bel, int ldb, int B, ModelView<T> mv, const unsigned long long* nzBw, int k_tiles, uint8_t* dead,
                       int32_t* btl, int32_t* btc, int* dead_count, hipStream_t st, const uint8_t* rowflags,
                       const int32_t* perm) {
    if (B <= 0) return hipSuccess;
    const size_t lds = (size_t)((k_tiles + 7) & ~7) + (size_t)((k_tiles + 63) / 64) * 8;      // byte flags + bit words
    hipLaunchKernelGGL(k_dead<T>, dim3(B), dim3(256), lds, st, bel, ldb, mv, nzBw, k_tiles, dead, btl, btc,
                       dead_count, rowflags, perm);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// argmax over alpha-vectors, one wavefront per (belief, group) row segment.
// np.argmax semantics: first maximum.  src/pomdp.py:1495 (argmax part)
// ------------------------------------------------------------------------- //
template <typename T>
__global__ void k_argmax(SlabView<T> sv, int V, int G, int B, const uint8_t* __restrict__ dead, double tol_rel,
                         double tol_abs, double tol_extra, const int* __restrict__ chain_steps, int flag_all, int32_t* __restrict__ best_v, double* __restrict__ best_score,
                         double* __restrict__ err, int32_t* __restrict__ queue, int* __restrict__ qcount) {
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= (int64_t)B * G) return;
    const int b = (int)(gw / G), g = (int)(gw % G);
    if (dead != nullptr && dead[gw]) {
        if (lane == 0) {
            best_v[gw] = 0;
            best_score[gw] = 0.0;
            err[gw] = 0.0;
        }
        return;
    }
    // one pass: running maximum (first index) and runner-up value per lane, merged across the wave
    T m = -std::numeric_limits<T>::infinity(), m2 = -std::numeric_limits<T>::infinity();
    int idx = 0x7fffffff;
    // Fast path (alpha-side f32 slabs, groups aligned to 4 columns): a lane takes 4 consecutive columns per load
    // (16 bytes per slab) and looks the pair's slab count up once per 4 columns instead of once per score; columns
    // and chunks ascend per lane, so "first maximum" is unchanged.
    bool vec4 = false;
    if constexpr (sizeof(T) == 4) vec4 = sv.nchunks != nullptr && (sv.push || (V & 3) == 0) && (sv.ldc & 3) == 0;
    if (vec4) {
        typedef float F4 __attribute__((ext_vector_type(4)));
        // alpha side: row b, columns g * V + v; belief side: row (o, a, b), columns v
        const int64_t row = sv.push ? sv.push_row(b, g) : (int64_t)b;
        const int64_t col0 = sv.push ? 0 : (int64_t)g * V;
        const float* rowp = (const float*)sv.slabs + row * sv.ldc + col0;
        const int tm = (int)(row >> 8);
        // four 256-column steps at a time: their slab counts first, then their first slabs, then what further slabs
        // there are -- two rounds of independent loads per 1024 columns instead of eight dependent ones
        for (int c00 = lane * 4; c00 < V; c00 += 1024) {
            int n[4], fb[4];
            F4 acc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c0 = c00 + 256 * j;
                const int64_t pair = ((col0 + c0) >> 8) * sv.tiles_m + tm;
                n[j] = c0 < V ? sv.nchunks[pair] : 0;
                fb[j] = (c0 < V && sv.first_block != nullptr) ? sv.first_block[pair] : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const F4 zero = {0.f, 0.f, 0.f, 0.f};
                acc[j] = zero;
                if (n[j] > 0) acc[j] = acc[j] + *(const F4*)(rowp + c00 + 256 * j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (n[j] <= 1) continue;
                if (sv.first_block != nullptr) {             // continuation tiles behind the full slab (SlabView)
                    const float* x = (const float*)sv.slabs + sv.slab_stride + (int64_t)(fb[j] + 1) * (256 * 256) +
                                     (row & 255) * 256 + ((col0 + c00 + 256 * j) & 255);
                    for (int z = 1; z < n[j]; ++z, x += 256 * 256) acc[j] = acc[j] + *(const F4*)x;
                } else {
                    for (int z = 1; z < n[j]; ++z) acc[j] = acc[j] + *(const F4*)(rowp + c00 + 256 * j + (int64_t)z * sv.slab_stride);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c0 = c00 + 256 * j;
                if (c0 >= V) break;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const T sc = (T)acc[j][i];
                    const int v = c0 + i;
                    if (v >= V) break;                          // (belief side: V need not be a multiple of 4)
                    if (sc > m || idx == 0x7fffffff) {
                        m2 = (idx == 0x7fffffff) ? m2 : m;
                        m = sc;
                        idx = v;
                    } else if (sc > m2) {
                        m2 = sc;
                    }
                }
            }
        }
    } else
    for (int v0 = lane; v0 < V; v0 += 256) {               // four independent score reads in flight per lane
        T sc4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = v0 + 64 * j;
            sc4[j] = v < V ? sv.score(b, g, V, v) : T(0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int v = v0 + 64 * j;
            if (v >= V) break;
            const T sc = sc4[j];
            if (sc > m || idx == 0x7fffffff) {
                m2 = (idx == 0x7fffffff) ? m2 : m;
                m = sc;
                idx = v;
            } else if (sc > m2) {
                m2 = sc;                                    // includes sc == m (a later exact tie)
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T om = __shfl_xor(m, off, 64), om2 = __shfl_xor(m2, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        const T lo = om < m ? om : m;                       // the loser of the two maxima is a runner-up
        T n2 = m2 > om2 ? m2 : om2;
        if (oi != 0x7fffffff && idx != 0x7fffffff) n2 = lo > n2 ? lo : n2;
        if (om > m || (om == m && oi < idx)) {
            m = om;
            idx = oi;
        }
        m2 = n2;
    }
    double E = 0.0;
    int push = 0;
    if (queue != nullptr) {
        const double mag = fmax(fabs((double)m), fabs(sv.magnitude(b, g, G, V)));
        // f32 error model: 8 * 2^-24 * sqrt(longest fma chain) (+ slab sums); the chain length is the
        // stream-K share size, known only on the device
        double tr = tol_rel;
        if (chain_steps != nullptr && tol_rel < 0.0)
            tr = 8.0 * 5.9604644775390625e-08 * (sqrt((double)chain_steps[0] * 32.0) + 1.0);
        if (sv.push) tr += 2.0 * 5.9604644775390625e-08;    // bp is rounded once to f32 after its f64 accumulation
        E = (tr + tol_extra) * mag + tol_abs;               // tol_extra: input rounding of an fp32 screen of fp64 operands
        push = (flag_all || (double)m2 >= (double)m - 2.0 * E) ? 1 : 0;
    }
    if (lane == 0) {
        best_v[gw] = idx;
        best_score[gw] = (double)m;
        err[gw] = E;
        if (push) {
            const int slot = atomicAdd(qcount, 1);
            queue[slot] = (int32_t)gw;
        }
    }
}

template <typename T>
hipError_t launch_argmax(SlabView<T> sv, int V, int G, int B, const uint8_t* dead, double tol_rel, double tol_abs,
                         const int* chain_steps, int flag_all, int32_t* best_v, double* best_score, double* err,
                         int32_t* queue, int* qcount, hipStream_t st, double tol_extra) {
    const int64_t rows = (int64_t)B * G;
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_argmax<T>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, sv, V, G, B, dead, tol_rel,
                       tol_abs, tol_extra, chain_steps, flag_all, best_v, best_score, err, queue, qcount);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// fp64 re-decision of a queued (belief, group): every candidate whose f32 score is
// within the error window of the f32 maximum is re-scored exactly (f32 x f32 products
// are exact in f64) and the first maximum of the exact scores wins.
//
// Real value functions are full of near-ties (and exact ties: every alpha-vector has the same value at an
// absorbing goal state, so for the "goal" observation ALL of them tie), so this kernel sees entries with
// thousands of candidates.  Two things keep those cheap:
//   * the dot runs only over tiles where the belief AND the (a,o) support of RTO are non-zero (list built once
//     per entry in LDS) -- for a rarely-seen observation that is one or two tiles instead of the belief's
//     hundreds;
//   * candidates are scored one per wave, four at a time, with wave-level reductions only (no block barrier
//     per candidate).
// ------------------------------------------------------------------------- //
constexpr int REFINE_LIST_CAP = 2048;   // tiles kept in LDS (65536 states of support); longer lists stay global
constexpr int REFINE_SPLIT = 16;        // k_refine_split: blocks per entry (parts of its tile list)
constexpr int REFINE_SPLIT_CAND = 8;    // ... and candidates per entry it takes

// exact score of one candidate over a tile list, computed by one wave (all lanes get the sum)
template <typename T, bool PROJ>
__device__ __forceinline__ double refine_wave_dot(const T* __restrict__ brow, const T* __restrict__ arow,
                                                  const ModelView<T>& mv, int a, int o, double gamma,
                                                  const int* L, int n_tiles, int lane) {
    const int32_t* __restrict__ rs = mv.rs + (int64_t)a * mv.R * mv.S_pad;
    const T* __restrict__ rto = mv.rto + (int64_t)(a * mv.O + o) * mv.R * mv.S_pad;
    const int n_el = n_tiles * 32;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i0 = lane; i0 < n_el; i0 += 256) {
        int s[4];
        double w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = i0 + 64 * j;
            const bool ok = i < n_el;
            const int ii = ok ? i : i0;
            const int t = L ? L[ii >> 5] : (ii >> 5);
            s[j] = t * 32 + (ii & 31);                       // < S_pad; pads hold rs = 0, rto = 0, b = 0
            w[j] = ok ? 1.0 : 0.0;
        }
        if constexpr (PROJ) {
            double g[4] = {0.0, 0.0, 0.0, 0.0};
            for (int r = 0; r < mv.R; ++r) {
                const int64_t ro = (int64_t)r * mv.S_pad;
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] += (double)rto[ro + s[j]] * (double)arow[rs[ro + s[j]]];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += w[j] * ((double)brow[s[j]] * (gamma * g[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = s[j] < mv.S;                 // alpha rows are valid for s < S only
                acc[j] += (in ? w[j] : 0.0) * ((double)brow[s[j]] * (double)arow[in ? s[j] : 0]);
            }
        }
    }
    return wave_sum((acc[0] + acc[1]) + (acc[2] + acc[3]));
}

template <typename T, typename TS, bool PROJ, bool L1 /* level-1 screen over projected rows (work.gam); else candidates together */>
__global__ void k_refine(SlabView<TS> sv, int V, int G, const int32_t* __restrict__ queue,
                         const int* __restrict__ qcount,
                         const T* __restrict__ bel, int ldb, const T* __restrict__ alpha, int lda,
                         ModelView<T> mv, double gamma, const int32_t* __restrict__ btl, const int32_t* __restrict__ btc,
                         const uint8_t* __restrict__ nzG /* [G][k_tiles] support tiles per group, or nullptr */,
                         int32_t* __restrict__ best_v,
                         double* __restrict__ best_score, double* __restrict__ err, int* __restrict__ cand_total,
                         RefineWork work) {
    __shared__ int cand[256];
    __shared__ int wcount[4];
    __shared__ int ltile[REFINE_LIST_CAP];
    __shared__ int lbase, loverflow;
    __shared__ int sh_slot, sh_base;
    __shared__ double red[4];
    __shared__ double wbest[4];
    __shared__ int widx[4];
    if ((int)blockIdx.x >= *qcount) return;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int e = queue[blockIdx.x];
    const int b = e / G, g = e % G;
    const int a = PROJ ? g / mv.O : 0, o = PROJ ? g % mv.O : 0;
    const double m = best_score[e], E = err[e];
    const double thr = m - 2.0 * E;
    const T* brow = bel + (int64_t)b * ldb;
    const int k_tiles = mv.S_pad >> 5;
    // -- tile list of this entry: belief tiles, filtered by the group's support tiles
    const int32_t* src = btl ? btl + (int64_t)b * k_tiles : nullptr;
    const int n_src = btl ? btc[b] : k_tiles;
    const uint8_t* zg = (PROJ && nzG) ? nzG + (int64_t)g * k_tiles : nullptr;
    if (tid == 0) {
        lbase = 0;
        loverflow = 0;
    }
    __syncthreads();
    for (int i0 = 0; i0 < n_src; i0 += 256) {
        const int i = i0 + tid;
        int t = 0, f = 0;
        if (i < n_src) {
            t = src ? src[i] : i;
            f = zg ? (zg[t] != 0) : 1;
        }
        const unsigned long long mask = __ballot(f);
        if (lane == 0) wcount[wid] = __popcll(mask);
        __syncthreads();
        int off = lbase;
        for (int w = 0; w < wid; ++w) off += wcount[w];
        const int tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        const bool fits = lbase + tot <= REFINE_LIST_CAP;
        if (f && fits) ltile[off + __popcll(mask & ((1ull << lane) - 1ull))] = t;
        __syncthreads();
        if (tid == 0) {
            if (fits) lbase += tot;
            else loverflow = 1;
        }
        __syncthreads();
        if (loverflow) break;
    }
    // overflow: fall back to the belief's own (unfiltered) list in global memory -- zero tiles add exact zeros
    const int* L = loverflow ? src : ltile;
    const int n_tiles = loverflow ? n_src : lbase;

    double bestval = -std::numeric_limits<double>::infinity();
    int bestidx = 0x7fffffff;
    int slot = -1;                                           // >= 0 once this entry hands candidates to the work list
    if (tid == 0) sh_slot = -1;
    // exact scores of cand[0 .. ncand): first maximum into (bestval, bestidx), per wave
    auto score_exact = [&](int ncand) {
        if (ncand <= 2) {
            // the common case (a runner-up inside the window): all 256 threads on one dot at a time -- a quarter of
            // the latency of giving each candidate to a single wave
            const TileList tl{L, n_tiles};
            for (int c = 0; c < ncand; ++c) {
                const int vv = cand[c];
                const T* arow = alpha + (int64_t)vv * lda;
                const double part = PROJ ? proj_dot_partial(brow, arow, mv, a, o, gamma, tl)
                                         : plain_dot_partial(brow, arow, mv.S, tl);
                const double tot = block_sum(part, red);
                if (tot > bestval) {                         // every wave tracks the same (value, index)
                    bestval = tot;
                    bestidx = vv;
                }
            }
        } else {
            for (int c = wid; c < ncand; c += 4) {           // one candidate per wave, ascending v within a wave
                const int vv = cand[c];
                const double tot = refine_wave_dot<T, PROJ>(brow, alpha + (int64_t)vv * lda, mv, a, o, gamma, L, n_tiles, lane);
                if (tot > bestval) {
                    bestval = tot;
                    bestidx = vv;
                }
            }
        }
    };
    bool settled = false;                                    // decided before the chunk loop below (block-uniform)
    if (!loverflow) {                                        // (an overflowed list names tiles nobody projected)
        // -- collect the entry's candidates over all chunks of V (ascending v); more than L1_CAP: the chunk loop below
        constexpr int L1_CAP = 32;
        constexpr int FAST_CAP = 8;                          // candidates scored together, four per pass
        __shared__ int c1[L1_CAP];
        __shared__ double s1[L1_CAP];
        __shared__ double l1red[16];
        __shared__ int n1_sh, n2_sh;
        if (tid == 0) n1_sh = 0;
        __syncthreads();
        for (int v0 = 0; v0 < V; v0 += 256) {
            const int v = v0 + tid;
            int flag = 0;
            if (v < V) flag = ((double)sv.score(b, g, V, v) >= thr) ? 1 : 0;
            const unsigned long long mask = __ballot(flag);
            if (lane == 0) wcount[wid] = __popcll(mask);
            __syncthreads();
            int base = n1_sh;
            for (int w = 0; w < wid; ++w) base += wcount[w];
            const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
            if (flag && pos < L1_CAP) c1[pos] = v;
            __syncthreads();
            if (tid == 0) n1_sh += wcount[0] + wcount[1] + wcount[2] + wcount[3];
            __syncthreads();
        }
        const int n1 = n1_sh;
        constexpr bool l1 = L1 && PROJ && sizeof(T) == 4;   // (two instantiations: each carries one of the two paths only)
        if (n1 > 0 && n1 <= L1_CAP && (l1 || n1 <= FAST_CAP)) {
            if (tid == 0 && cand_total != nullptr) atomicAdd(cand_total, n1);
            const TileList tl{L, n_tiles};
            int n2 = n1;
            if constexpr (l1) {
                {   // level 1: fp64 sums over the projected fp32 rows the GEMM read (RefineWork::gam)
                    for (int c0 = 0; c0 < n1; c0 += 4) {
                        const float* rows[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            rows[j] = work.gam + ((int64_t)g * V + c1[c0 + j < n1 ? c0 + j : c0]) * work.ldg;
                        double d[4];
                        gamma_dots4<T>(brow, rows, tl, d, l1red);
                        if (tid == 0)
                            for (int j = 0; j < 4 && c0 + j < n1; ++j) s1[c0 + j] = d[j];
                    }
                    __syncthreads();
                    const double mag = fmax(fabs(m), fabs(sv.magnitude(b, g, G, V)));
                    const double E2 = work.l1_rel * mag;
                    if (tid == 0) {
                        double m1 = s1[0];
                        for (int c = 1; c < n1; ++c) m1 = s1[c] > m1 ? s1[c] : m1;
                        int k = 0, keep1 = 0;
                        for (int c = 0; c < n1; ++c)
                            if (s1[c] >= m1 - 2.0 * E2) {
                                keep1 = c;
                                c1[k++] = c1[c];             // survivors, still ascending (k <= c)
                            }
                        n2_sh = k;
                        if (k == 1) {                        // separated: the exact maximum is this one (within E2 of s1)
                            best_v[e] = c1[0];
                            best_score[e] = s1[keep1];
                            err[e] = E2;
                        }
                    }
                    __syncthreads();
                    n2 = n2_sh;
                    if (n2 == 1) return;
                    // What the projected rows cannot separate (duplicates on the belief's support: ~1 % of the entries) is
                    // re-scored from alpha, RTO and the successor lists -- R gathers per state over the whole list.  Inside
                    // this block that is one long chain per entry, and 30 such entries were 145 us of tail behind 3000 entries
                    // that had finished (R = 5, |S| = 30000): they go to k_refine_split, REFINE_SPLIT blocks per entry.
                    if (work.q2_entry != nullptr && n2 <= REFINE_SPLIT_CAND) {
                        if (tid == 0) {
                            const int qi = atomicAdd(&work.cnt[2], 1);
                            sh_slot = qi < work.q2_cap ? qi : -1;
                            if (qi < work.q2_cap) {
                                work.q2_entry[qi] = e;
                                work.q2_n[qi] = n2;
                                for (int c = 0; c < n2; ++c) work.q2_cand[(int64_t)qi * REFINE_SPLIT_CAND + c] = c1[c];
                            }
                        }
                        __syncthreads();
                        if (sh_slot >= 0) return;
                        if (tid == 0) sh_slot = -1;           // (no room: scored here, as before)
                        __syncthreads();
                    }
                }
            }
            if constexpr (!l1) {    // exact scores of all of them together, four per pass; first maximum
                for (int c0 = 0; c0 < n2; c0 += 4) {
                    const T* rows[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) rows[j] = alpha + (int64_t)c1[c0 + j < n2 ? c0 + j : c0] * lda;
                    double d[4];
                    exact_dots4<T, PROJ>(brow, rows, n2 - c0 < 4 ? n2 - c0 : 4, mv, a, o, gamma, tl, d, l1red);
                    if (tid == 0)
                        for (int j = 0; j < 4 && c0 + j < n2; ++j) s1[c0 + j] = d[j];
                }
                __syncthreads();
                if (tid == 0) {
                    double bv = s1[0];
                    int bi = c1[0];
                    for (int c = 1; c < n2; ++c)
                        if (s1[c] > bv) {
                            bv = s1[c];
                            bi = c1[c];
                        }
                    best_v[e] = bi;
                    best_score[e] = bv;
                    err[e] = 0.0;
                }
                return;
            }
            for (int c = tid; c < n2; c += 256) cand[c] = c1[c];
            __syncthreads();
            score_exact(n2);                                 // more than that (after level 1): one candidate per wave
            settled = true;
        }
    }
    for (int v0 = 0; v0 < V && !settled; v0 += 256) {
        const int v = v0 + tid;
        int flag = 0;
        if (v < V) flag = ((double)sv.score(b, g, V, v) >= thr) ? 1 : 0;
        const unsigned long long mask = __ballot(flag);
        __syncthreads();
        if (lane == 0) wcount[wid] = __popcll(mask);
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wid; ++w) base += wcount[w];
        const int ncand = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        if (flag) cand[base + __popcll(mask & ((1ull << lane) - 1ull))] = v;
        if (tid == 0 && cand_total != nullptr && ncand > 0) atomicAdd(cand_total, ncand);
        // Many candidates: one block scoring them four at a time is a long latency chain (tens of microseconds per
        // candidate on a wide belief); hand them to the grid-wide pass instead.
        if (work.items_v != nullptr && !loverflow && (slot >= 0 || ncand > work.defer_min)) {
            if (tid == 0) {
                int sl = slot;
                if (sl < 0) {
                    sl = atomicAdd(&work.cnt[1], 1);
                    if (sl >= work.slot_cap) sl = -1;
                }
                int ib = -1;
                if (sl >= work.w_slot_cap) {                 // item path (the GEMM path needs no items)
                    ib = atomicAdd(&work.cnt[0], ncand);
                    if (ib + ncand > work.item_cap) {        // no room: void whatever part of the range exists
                        for (int c = ib; c < work.item_cap && c < ib + ncand; ++c) work.items_slot[c] = -1;
                        ib = -1;
                    }
                }
                sh_slot = sl;
                sh_base = ib;
            }
            __syncthreads();
            const int sl = sh_slot, ib = sh_base;
            if (sl >= 0 && slot < 0) {                       // first use of the slot: publish the entry (and its tile list)
                if (sl >= work.w_slot_cap)
                    for (int i = tid; i < n_tiles; i += 256) work.tiles[(int64_t)sl * k_tiles + i] = L[i];
                if (tid == 0) {
                    work.slot_entry[sl] = e;
                    work.slot_n[sl] = n_tiles;
                }
                slot = sl;
            }
            if (sl >= 0 && sl < work.w_slot_cap) break;      // GEMM path: every alpha row gets re-scored, nothing to list
            if (sl >= 0 && ib >= 0) {
                for (int c = tid; c < ncand; c += 256) {
                    work.items_v[ib + c] = cand[c];
                    work.items_slot[ib + c] = sl;
                }
                continue;                                    // next chunk (the barrier at its top orders cand reuse)
            }
        }
        __syncthreads();
        score_exact(ncand);
    }
    // first maximum over the four waves: largest value, then smallest index
    if (lane == 0) {
        wbest[wid] = bestval;
        widx[wid] = bestidx;
    }
    __syncthreads();
    if (tid == 0) {
        double bv = wbest[0];
        int bi = widx[0];
        for (int w = 1; w < 4; ++w)
            if (wbest[w] > bv || (wbest[w] == bv && widx[w] < bi)) {
                bv = wbest[w];
                bi = widx[w];
            }
        if (slot >= 0 && slot < work.w_slot_cap) {
            // GEMM path: k_refine_slot_argmax decides from exact scores of all alpha rows
        } else if (slot >= 0) {                              // merged with the work-list results in k_refine_merge
            work.ib_val[slot] = bv;
            work.ib_idx[slot] = bi;
        } else if (bi != 0x7fffffff) {
            best_v[e] = bi;
            best_score[e] = bv;
            err[e] = 0.0;
        }
    }
}

// Exact re-decision of the entries k_refine's level-1 screen could not separate (RefineWork::q2_*): REFINE_SPLIT blocks per
// entry, block (q, p) scoring the entry's surviving candidates over part p of its tile list (exact_dots4); the block that
// arrives last adds the parts in order p = 0, 1, ... and writes the first maximum.  Identical candidate rows still give
// identical sums.  grid = (entries at a time, REFINE_SPLIT).
template <typename T, bool PROJ>
__global__ void k_refine_split(int G, const T* __restrict__ bel, int ldb, const T* __restrict__ alpha, int lda, ModelView<T> mv,
                               double gamma, const int32_t* __restrict__ btl, const int32_t* __restrict__ btc,
                               const uint8_t* __restrict__ nzG, int32_t* __restrict__ best_v, double* __restrict__ best_score,
                               double* __restrict__ err, RefineWork work) {
    __shared__ int wcount[4];
    __shared__ int ltile[REFINE_LIST_CAP];
    __shared__ int lbase, loverflow, last;
    __shared__ double red16[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, part = blockIdx.y;
    const int n_q = min(work.cnt[2], work.q2_cap);
    const int k_tiles = mv.S_pad >> 5;
    for (int q = blockIdx.x; q < n_q; q += gridDim.x) {
        const int e = work.q2_entry[q], nc_all = work.q2_n[q];
        const int b = e / G, g = e % G;
        const int a = PROJ ? g / mv.O : 0, o = PROJ ? g % mv.O : 0;
        const T* brow = bel + (int64_t)b * ldb;
        // the entry's tile list, as k_refine builds it (belief tiles filtered by the group's support tiles)
        const int32_t* src = btl ? btl + (int64_t)b * k_tiles : nullptr;
        const int n_src = btl ? btc[b] : k_tiles;
        const uint8_t* zg = (PROJ && nzG) ? nzG + (int64_t)g * k_tiles : nullptr;
        __syncthreads();                                     // (the previous entry's readers of ltile / lbase are done)
        if (tid == 0) {
            lbase = 0;
            loverflow = 0;
        }
        __syncthreads();
        for (int i0 = 0; i0 < n_src; i0 += 256) {
            const int i = i0 + tid;
            int t = 0, f = 0;
            if (i < n_src) {
                t = src ? src[i] : i;
                f = zg ? (zg[t] != 0) : 1;
            }
            const unsigned long long mask = __ballot(f);
            if (lane == 0) wcount[wid] = __popcll(mask);
            __syncthreads();
            int off = lbase;
            for (int w = 0; w < wid; ++w) off += wcount[w];
            const int tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
            const bool fits = lbase + tot <= REFINE_LIST_CAP;
            if (f && fits) ltile[off + __popcll(mask & ((1ull << lane) - 1ull))] = t;
            __syncthreads();
            if (tid == 0) {
                if (fits) lbase += tot;
                else loverflow = 1;
            }
            __syncthreads();
            if (loverflow) break;
        }
        const int* L = loverflow ? src : ltile;              // overflow: the belief's own list (zero tiles add exact zeros)
        const int n_tiles = loverflow ? n_src : lbase;
        const int lo = (int)((int64_t)n_tiles * part / REFINE_SPLIT), hi = (int)((int64_t)n_tiles * (part + 1) / REFINE_SPLIT);
        const TileList tl{L != nullptr ? L + lo : nullptr, hi - lo, lo};       // (no list at all: tiles lo, lo + 1, ...)
        double* mine = work.q2_part + ((int64_t)q * REFINE_SPLIT + part) * REFINE_SPLIT_CAND;
        for (int c0 = 0; c0 < nc_all; c0 += 4) {
            const T* rows[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                rows[j] = alpha + (int64_t)work.q2_cand[(int64_t)q * REFINE_SPLIT_CAND + (c0 + j < nc_all ? c0 + j : c0)] * lda;
            double d[4];
            exact_dots4<T, PROJ>(brow, rows, nc_all - c0 < 4 ? nc_all - c0 : 4, mv, a, o, gamma, tl, d, red16);
            if (tid == 0)
                for (int j = 0; j < 4 && c0 + j < nc_all; ++j) mine[c0 + j] = d[j];
        }
        // the last of the entry's blocks to arrive combines
        if (tid == 0) {
            __threadfence();
            const int arrived = atomicAdd(&work.q2_done[q], 1);
            last = arrived == REFINE_SPLIT - 1;
            if (last) work.q2_done[q] = 0;                   // ready for the next launch
        }
        __syncthreads();
        if (last && tid == 0) {
            __threadfence();
            const volatile double* parts = work.q2_part + (int64_t)q * REFINE_SPLIT * REFINE_SPLIT_CAND;
            double bv = 0.0;
            int bi = -1;
            for (int c = 0; c < nc_all; ++c) {               // candidates ascend: first maximum
                double sum = 0.0;
                for (int p = 0; p < REFINE_SPLIT; ++p) sum += parts[p * REFINE_SPLIT_CAND + c];
                if (bi < 0 || sum > bv) {
                    bv = sum;
                    bi = work.q2_cand[(int64_t)q * REFINE_SPLIT_CAND + c];
                }
            }
            best_v[e] = bi;
            best_score[e] = bv;
            err[e] = 0.0;
        }
    }
}

// order-preserving map of finite doubles to unsigned integers (0 is below every mapped value)
__device__ __forceinline__ unsigned long long ordered_bits(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double unordered_bits(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// Grid-wide pass over the deferred (entry, candidate) items: one wave per item.
template <typename T, bool PROJ>
__global__ void k_refine_items(int G, const T* __restrict__ bel, int ldb, const T* __restrict__ alpha, int lda,
                               ModelView<T> mv, double gamma, RefineWork work) {
    const int lane = threadIdx.x & 63;
    const int n_items = min(work.cnt[0], work.item_cap);
    const int k_tiles = mv.S_pad >> 5;
    const int waves = gridDim.x * 4;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < n_items; i += waves) {
        const int sl = work.items_slot[i];
        if (sl < 0) continue;
        const int e = work.slot_entry[sl];
        const int b = e / G, g = e % G;
        const int a = PROJ ? g / mv.O : 0, o = PROJ ? g % mv.O : 0;
        const double tot = refine_wave_dot<T, PROJ>(bel + (int64_t)b * ldb, alpha + (int64_t)work.items_v[i] * lda, mv, a, o,
                                                    gamma, work.tiles + (int64_t)sl * k_tiles, work.slot_n[sl], lane);
        if (lane == 0) {
            work.scores[i] = tot;
            if (tot == tot) atomicMax(&work.emax[sl], ordered_bits(tot));
        }
    }
}

// first maximum: among the items that attain the entry's maximum, the smallest candidate index
__global__ void k_refine_first(RefineWork work) {
    const int n_items = min(work.cnt[0], work.item_cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
        const int sl = work.items_slot[i];
        if (sl < 0) continue;
        const double sc = work.scores[i];
        if (sc == sc && ordered_bits(sc) == work.emax[sl]) atomicMax(&work.eidx[sl], 0x7fffffff - work.items_v[i]);
    }
}

__global__ void k_refine_merge(RefineWork work, int32_t* __restrict__ best_v, double* __restrict__ best_score,
                               double* __restrict__ err) {
    const int n_slots = min(work.cnt[1], work.slot_cap);
    for (int sl = work.w_slot_cap + blockIdx.x * blockDim.x + threadIdx.x; sl < n_slots; sl += gridDim.x * blockDim.x) {
        double bv = work.ib_val[sl];
        int bi = work.ib_idx[sl];
        if (work.emax[sl] != 0ull) {
            const double wv = unordered_bits(work.emax[sl]);
            const int wi = 0x7fffffff - work.eidx[sl];   // stored reversed so that zero-filled memory means "none"
            if (bi == 0x7fffffff || wv > bv || (wv == bv && wi < bi)) {
                bv = wv;
                bi = wi;
            }
        }
        if (bi != 0x7fffffff) {
            const int e = work.slot_entry[sl];
            best_v[e] = bi;
            best_score[e] = bv;
            err[e] = 0.0;
        }
    }
}

// GEMM path, step 1: the fp64 weight row of each slot's entry.  PROJ: w[s'] = gamma * sum_{(s,r)->s'} b[s] * RTO[s,a,o,r]
// (pull over the inverse lists; products of fp32 values are exact in fp64), so that w . alpha_v is the entry's exact
// score of alpha_v; plain: w = b.
template <typename T, bool PROJ>
__global__ void k_refine_weights(int G, const T* __restrict__ bel, int ldb, ModelView<T> mv, double gamma, RefineWork work) {
    const int sl = blockIdx.y, sp = blockIdx.x * 256 + threadIdx.x;
    if (sp >= mv.S_pad) return;
    const int e = work.slot_entry[sl];
    const int b = e / G, g = e % G;
    const T* brow = bel + (int64_t)b * ldb;
    double w = 0.0;
    if (sp < mv.S) {
        if constexpr (PROJ) {
            const int a = g / mv.O, o = g % mv.O;
            const int32_t* ptr = work.in_ptr + (int64_t)a * (mv.S + 1);
            const int32_t* src = work.in_src + (int64_t)a * mv.S * mv.R;
            const T* rto = mv.rto + (int64_t)(a * mv.O + o) * mv.R * mv.S_pad;
            for (int j = ptr[sp]; j < ptr[sp + 1]; ++j) {
                const int en = src[j];
                const int s = en / mv.R, r = en - s * mv.R;
                w += (double)brow[s] * (double)rto[(int64_t)r * mv.S_pad + s];
            }
            w *= gamma;
        } else {
            w = (double)brow[sp];
        }
    }
    work.W[(int64_t)sl * mv.S_pad + sp] = w;
}

// GEMM path, step 3: first maximum of the slot's exact scores over all V alpha rows
__global__ void k_refine_slot_argmax(RefineWork work, int V, int split, int64_t slab_stride, int32_t* __restrict__ best_v,
                                     double* __restrict__ best_score, double* __restrict__ err) {
    __shared__ double wv[4];
    __shared__ int wi[4];
    const int sl = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const double* row = work.Cx + (int64_t)sl * V;
    double m = -std::numeric_limits<double>::infinity();
    int idx = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        double x = row[v];
        for (int z = 1; z < split; ++z) x += row[v + (int64_t)z * slab_stride];      // K parts of the GEMM, fixed order
        if (x > m) {                                         // ascending v per thread: first maximum
            m = x;
            idx = v;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(m, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (om > m || (om == m && oi < idx)) {
            m = om;
            idx = oi;
        }
    }
    if (lane == 0) {
        wv[wid] = m;
        wi[wid] = idx;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (wv[w] > m || (wv[w] == m && wi[w] < idx)) {
                m = wv[w];
                idx = wi[w];
            }
        if (idx != 0x7fffffff) {
            const int e = work.slot_entry[sl];
            best_v[e] = idx;
            best_score[e] = m;
            err[e] = 0.0;
        }
    }
}

// Part 1: the per-entry pass.  What it deferred (work.cnt: items, slots) is copied to counts_host (pinned, 2 ints) on
// the stream; the caller synchronises -- or speculates that nothing was deferred and checks later.
template <typename T, typename TS>
hipError_t launch_refine_scan(bool proj, SlabView<TS> sv, int V, int G, int max_entries, const int32_t* queue,
                              const int* qcount, const T* bel, int ldb, const T* alpha, int lda, ModelView<T> mv,
                              double gamma, const int32_t* btl, const int32_t* btc, const uint8_t* nzG, int32_t* best_v,
                              double* best_score, double* err, int* cand_total, RefineWork work, int* counts_host,
                              hipStream_t st) {
    if (max_entries <= 0) return hipSuccess;
    hipError_t e;
    if (work.items_v != nullptr) {
        // cnt, emax and eidx live in one allocation (cnt first): one fill
        if ((e = hipMemsetAsync(work.cnt, 0, work.zero_bytes, st)) != hipSuccess) return e;
    }
    if (proj)
    {
        if (work.gam != nullptr && sizeof(T) == 4) {
            hipLaunchKernelGGL((k_refine<T, TS, true, true>), dim3(max_entries), dim3(256), 0, st, sv, V, G, queue, qcount, bel, ldb,
                               alpha, lda, mv, gamma, btl, btc, nzG, best_v, best_score, err, cand_total, work);
            if (work.q2_entry != nullptr)                    // what its level-1 screen left undecided, 16 blocks per entry
                hipLaunchKernelGGL((k_refine_split<T, true>), dim3(work.q2_cap < 256 ? work.q2_cap : 256, REFINE_SPLIT), dim3(256), 0,
                                   st, G, bel, ldb, alpha, lda, mv, gamma, btl, btc, nzG, best_v, best_score, err, work);
        } else
            hipLaunchKernelGGL((k_refine<T, TS, true, false>), dim3(max_entries), dim3(256), 0, st, sv, V, G, queue, qcount, bel, ldb,
                               alpha, lda, mv, gamma, btl, btc, nzG, best_v, best_score, err, cand_total, work);
    }
    else
        hipLaunchKernelGGL((k_refine<T, TS, false, false>), dim3(max_entries), dim3(256), 0, st, sv, V, G, queue, qcount, bel,
                           ldb, alpha, lda, mv, gamma, btl, btc, nzG, best_v, best_score, err, cand_total, work);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (work.items_v != nullptr && counts_host != nullptr)
        if ((e = hipMemcpyAsync(counts_host, work.cnt, 2 * sizeof(int), hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    return hipSuccess;
}

// Part 2: the entries the per-entry pass handed on (h_items / h_slots as read back from work.cnt).
template <typename T>
hipError_t launch_refine_deferred(bool proj, int V, int G, const T* bel, int ldb, const T* alpha, int lda, ModelView<T> mv,
                                  double gamma, int32_t* best_v, double* best_score, double* err, RefineWork work,
                                  int h_items, int h_slots, hipStream_t st) {
    hipError_t e;
    if (work.items_v == nullptr) return hipSuccess;
    const int n_items = h_items < work.item_cap ? h_items : work.item_cap;
    const int n_slots = h_slots < work.slot_cap ? h_slots : work.slot_cap;
    const int n_w = n_slots < work.w_slot_cap ? n_slots : work.w_slot_cap;
    if (n_w > 0) {   // GEMM path: weights -> tile map -> [n_w x S] x [V x S]^T in fp64 on the MFMA -> first max
        dim3 wgrid((mv.S_pad + 255) / 256, n_w);
        if (proj)
            hipLaunchKernelGGL((k_refine_weights<T, true>), wgrid, dim3(256), 0, st, G, bel, ldb, mv, gamma, work);
        else
            hipLaunchKernelGGL((k_refine_weights<T, false>), wgrid, dim3(256), 0, st, G, bel, ldb, mv, gamma, work);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = launch_tile_nonzero_f64(work.W, mv.S_pad, n_w, mv.S_pad / 32, work.nzW, st)) != hipSuccess) return e;
        // a few dozen tie-heavy entries against thousands of alpha rows is a grid of a few dozen tile pairs: split along K
        // (room for w_slot_cap rows of Cx was reserved; n_w is usually a small part of it)
        int split = gemm_f64_split(n_w, V, mv.S_pad / 32);
        if (split > work.w_slot_cap / n_w) split = work.w_slot_cap / n_w;
        if (split < 1) split = 1;
        const int64_t slab_stride = (int64_t)n_w * V;
        if constexpr (sizeof(T) == 4) {   // fp32 alpha rows widened exactly on the way into LDS
            if ((e = launch_gemm_nt_f64_bf32(work.W, mv.S_pad, n_w, (const float*)alpha, lda, V, work.Cx, V, mv.S_pad,
                                             work.nzW, work.klistW, work.kcountW, st, split, slab_stride)) != hipSuccess)
                return e;
        } else {                          // fp64 engine behind an fp32 screen: the fp64 originals
            if ((e = launch_gemm_nt_f64(work.W, mv.S_pad, n_w, (const double*)alpha, lda, V, work.Cx, V, mv.S_pad,
                                        work.nzW, nullptr, 0, 0, work.klistW, work.kcountW, st, split, slab_stride)) != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(k_refine_slot_argmax, dim3(n_w), dim3(256), 0, st, work, V, 1, slab_stride, best_v, best_score,
                           err);                     // (the launcher folded the K parts into the first slab)
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (n_items > 0) {
        const int blocks = 2048;                       // 8192 waves: every SIMD of the chip has work in flight
        if (proj)
            hipLaunchKernelGGL((k_refine_items<T, true>), dim3(blocks), dim3(256), 0, st, G, bel, ldb, alpha, lda, mv,
                               gamma, work);
        else
            hipLaunchKernelGGL((k_refine_items<T, false>), dim3(blocks), dim3(256), 0, st, G, bel, ldb, alpha, lda, mv,
                               gamma, work);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(k_refine_first, dim3(1024), dim3(256), 0, st, work);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (n_slots > n_w) hipLaunchKernelGGL(k_refine_merge, dim3(64), dim3(256), 0, st, work, best_v, best_score, err);
    return hipGetLastError();
}

template <typename T, typename TS>
hipError_t launch_refine(bool proj, SlabView<TS> sv, int V, int G, int max_entries, const int32_t* queue,
                         const int* qcount, const T* bel, int ldb, const T* alpha, int lda, ModelView<T> mv,
                         double gamma, const int32_t* btl, const int32_t* btc, const uint8_t* nzG, int32_t* best_v,
                         double* best_score, double* err, int* cand_total, RefineWork work, hipStream_t st) {
    if (max_entries <= 0) return hipSuccess;
    int h_cnt[2] = {0, 0};
    hipError_t e = launch_refine_scan<T, TS>(proj, sv, V, G, max_entries, queue, qcount, bel, ldb, alpha, lda, mv, gamma, btl,
                                             btc, nzG, best_v, best_score, err, cand_total, work, h_cnt, st);
    if (e != hipSuccess || work.items_v == nullptr) return e;
    // how much was deferred decides what is launched next (one 8-byte read-back; most launches defer nothing)
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    return launch_refine_deferred<T>(proj, V, G, bel, ldb, alpha, lda, mv, gamma, best_v, best_score, err, work, h_cnt[0],
                                     h_cnt[1], st);
}

// ------------------------------------------------------------------------- //
// K4: action values and first-max action.  src/pomdp.py:1502-1505 via the identity
//   b . alpha_a[b,a,:] = b . ER[:,a] + sum_o max_v score[b,a,o,v]
// ------------------------------------------------------------------------- //
// Tail rows of Gamma that let the score GEMM produce the reward term of the action values:
//   row j <  A : ER[:, j]         -> score column = b . ER[:,a]      (rdot)
//   row j >= A : |ER[:, j-A]|     -> score column = b . |ER[:,a]|    (scales rdot's f32 error bound)
template <typename T>
__global__ void k_tail_rows(ModelView<T> mv, T* __restrict__ gam, int64_t row0, int ldg, const int32_t* __restrict__ ctile) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int j = blockIdx.y;
    if (s >= mv.S_pad) return;
    const T e = mv.er[(int64_t)(j % mv.A) * mv.S_pad + s];
    const int64_t row = row0 + j;
    const int64_t at = ctile != nullptr ? (int64_t)ctile[row >> 8] * 256 + (row & 255) : row;      // compact layout of a fused GEMM
    gam[at * ldg + s] = (j < mv.A) ? e : (e < T(0) ? -e : e);
}

template <typename T>
hipError_t launch_tail_rows(ModelView<T> mv, T* gam, int64_t row0, int ldg, hipStream_t st, const int32_t* ctile) {
    hipLaunchKernelGGL(k_tail_rows<T>, dim3((mv.S_pad + 255) / 256, 2 * mv.A), dim3(256), 0, st, mv, gam, row0, ldg, ctile);
    return hipGetLastError();
}

// K4: val[a] = b.ER[:,a] + sum_o best_score[b][a][o]; first max; near-ties queued.  One thread per (belief, action)
// gathers the terms (one thread per belief walking its A actions was 22 us of dependent loads for 1024 beliefs), the
// belief's first thread then takes the first maximum over the A values in order.
// rdot comes from the score matrix (column rd_col0 + a, f32 engines: with error bound tol * b.|ER_a|).
template <typename T>
__global__ void k_action_select(int B, int per_block, ModelView<T> mv, SlabView<T> sv, int64_t rd_col0, double tol_rel, double tol_extra,
                                const int* __restrict__ chain_steps, const double* __restrict__ best_score,
                                const double* __restrict__ err, double* __restrict__ rdot, double* __restrict__ rdot_err,
                                int32_t* __restrict__ action, int32_t* __restrict__ aqueue, int* __restrict__ aqcount,
                                uint8_t* __restrict__ acand /* [B][A]: action within the window of the best lower bound (or nullptr) */) {
    __shared__ double sv_[256], se_[256];
    const int A = mv.A;
    const int lb = threadIdx.x / A, a = threadIdx.x - lb * A;
    const int b = blockIdx.x * per_block + lb;
    const bool live = lb < per_block && b < B;
    if (live) {
        double tr = tol_rel;
        if (chain_steps != nullptr && tol_rel < 0.0) tr = 8.0 * 5.9604644775390625e-08 * (sqrt((double)chain_steps[0] * 32.0) + 1.0);
        tr += tol_extra;
        double rd, E = 0.0;
        if (sv.push) {
            rd = sv.aux_rd[(int64_t)b * A + a];               // f64 dot: exact to 1e-16, no window needed
        } else {
            rd = (double)sv.at(b, rd_col0 + a);
            if (aqueue != nullptr) E = tr * fmax(fabs(rd), fabs((double)sv.at(b, rd_col0 + A + a)));
        }
        rdot[(int64_t)b * A + a] = rd;
        rdot_err[(int64_t)b * A + a] = E;
        double v = rd;
        for (int o = 0; o < mv.O; ++o) {
            const int64_t e = ((int64_t)b * A + a) * mv.O + o;
            v += best_score[e];
            E += err[e];
        }
        sv_[threadIdx.x] = v;
        se_[threadIdx.x] = E;
    }
    __syncthreads();
    if (!live || a != 0) return;
    int best = 0;
    double bv = -std::numeric_limits<double>::infinity(), lo = bv;
    for (int x = 0; x < A; ++x) {
        const double v = sv_[threadIdx.x + x], E = se_[threadIdx.x + x];
        if (v > bv) {
            bv = v;
            best = x;
        }
        lo = fmax(lo, v - E);
    }
    action[b] = best;
    if (aqueue != nullptr) {
        int ncand = 0, anyerr = 0;
        for (int x = 0; x < A; ++x) {
            const double v = sv_[threadIdx.x + x], E = se_[threadIdx.x + x];
            if (v + E >= lo) {
                ++ncand;
                anyerr |= (E > 0.0);
            }
            if (acand != nullptr) acand[(int64_t)b * A + x] = (v + E >= lo) ? 1 : 0;
        }
        if (ncand > 1 && anyerr) aqueue[atomicAdd(aqcount, 1)] = b;
    }
}

template <typename T>
hipError_t launch_action(int B, ModelView<T> mv, SlabView<T> sv, int64_t rd_col0, double tol_rel, const int* chain_steps,
                         const double* best_score, const double* err, double* rdot, double* rdot_err, int32_t* action,
                         int32_t* aqueue, int* aqcount, hipStream_t st, double tol_extra, uint8_t* acand) {
    if (B <= 0) return hipSuccess;
    if (mv.A > 256) return hipErrorInvalidValue;
    const int per_block = 256 / mv.A;
    hipLaunchKernelGGL(k_action_select<T>, dim3((B + per_block - 1) / per_block), dim3(256), 0, st, B, per_block, mv, sv, rd_col0,
                       tol_rel, tol_extra, chain_steps, best_score, err, rdot, rdot_err, action, aqueue, aqcount, acand);
    return hipGetLastError();
}

// (These dots keep the one-state-per-lane loops: the whole-solve test test_end_to_end_fsvi_at_headline_scale_matches_reference
// follows the reference through exact ties of its action values for 40 backups with THIS summation order; the 16-byte-per-lane
// dots of k_refine here -- 80 -> 25 us at R = 5 -- leave the reference's trajectory after backup 25.)
// Exact (f64) value of every candidate action of a flagged belief.  One block per (queued belief, action, term, part):
// term 0 is b . ER[:,a], term 1 + o the score of observation o -- the 1 + O dots of an action used to run one after the
// other in one block (60 us of dependent latency for two dozen beliefs) -- and each dot is cut into ACTION_SPLIT parts of
// the belief's tile list (with five successors per state one block per dot was 117 us for 24 beliefs: a chain of ~90
// dependent gather rounds); k_action_final adds parts, then terms, in a fixed order.
template <typename T>
__global__ void k_refine_action(const T* __restrict__ bel, int ldb, const T* __restrict__ alpha, int lda,
                                ModelView<T> mv, double gamma, const int32_t* __restrict__ btl,
                                const int32_t* __restrict__ btc, const int32_t* __restrict__ aqueue, const int* __restrict__ aqcount,
                                const double* __restrict__ rdot, const double* __restrict__ rdot_err,
                                const int32_t* __restrict__ best_v, const double* __restrict__ best_score,
                                const double* __restrict__ err, double* __restrict__ val_parts /* [B][A][1+O] */,
                                const uint8_t* __restrict__ acand /* k_action_select's window test per (belief, action), or nullptr */) {
    __shared__ double red[4];
    __shared__ int cand_sh;
    const int n_q = *aqcount;
    const int terms = 1 + mv.O;
    const int a = blockIdx.y / terms, term = blockIdx.y % terms, tid = threadIdx.x, part_i = blockIdx.z;
    for (int q = blockIdx.x; q < n_q; q += gridDim.x) {
        const int b = aqueue[q];
        __syncthreads();
        if (tid == 0 && acand != nullptr) {
            cand_sh = acand[(int64_t)b * mv.A + a];
        } else if (tid == 0) {   // is action a within the error window of the best lower bound?
            double lo = -std::numeric_limits<double>::infinity(), va = 0.0, Ea = 0.0;
            for (int x = 0; x < mv.A; ++x) {
                double v = rdot[(int64_t)b * mv.A + x], E = rdot_err[(int64_t)b * mv.A + x];
                for (int o = 0; o < mv.O; ++o) {
                    const int64_t e = ((int64_t)b * mv.A + x) * mv.O + o;
                    v += best_score[e];
                    E += err[e];
                }
                lo = fmax(lo, v - E);
                if (x == a) {
                    va = v;
                    Ea = E;
                }
            }
            cand_sh = (va + Ea >= lo) ? 1 : 0;
        }
        __syncthreads();
        double* part = val_parts + (((int64_t)b * mv.A + a) * terms + term) * ACTION_SPLIT + part_i;
        if (!cand_sh) {
            if (tid == 0) *part = -std::numeric_limits<double>::infinity();
            continue;
        }
        const T* brow = bel + (int64_t)b * ldb;
        const int k_tiles = mv.S_pad >> 5;
        const int n_all = btl ? btc[b] : k_tiles;
        const int per = (n_all + ACTION_SPLIT - 1) / ACTION_SPLIT;
        const int t0 = part_i * per < n_all ? part_i * per : n_all, t1 = t0 + per < n_all ? t0 + per : n_all;
        const TileList tl{btl ? btl + (int64_t)b * k_tiles + t0 : nullptr, t1 - t0, btl ? 0 : t0};
        double v;
        if (term == 0) {
            v = block_sum(plain_dot_partial(brow, mv.er + (int64_t)a * mv.S_pad, mv.S, tl), red);
        } else {
            const int o = term - 1;
            const int64_t e = ((int64_t)b * mv.A + a) * mv.O + o;
            if (err[e] > 0.0)                                       // block-uniform
                v = block_sum(proj_dot_partial(brow, alpha + (int64_t)best_v[e] * lda, mv, a, o, gamma, tl), red);
            else
                v = part_i == 0 ? best_score[e] : 0.0;              // already exact: carried by part 0
        }
        if (tid == 0) *part = v;
    }
}

__global__ void k_action_final(int A, int terms, const int32_t* __restrict__ aqueue, const int* __restrict__ aqcount,
                               const double* __restrict__ val_parts, int32_t* __restrict__ action) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= *aqcount) return;
    const int b = aqueue[i];
    int best = 0;
    double bv = -std::numeric_limits<double>::infinity();
    for (int a = 0; a < A; ++a) {
        const double* p = val_parts + ((int64_t)b * A + a) * terms * ACTION_SPLIT;
        double v = 0.0;                                         // b.ER, then the observations in order (-inf stays -inf)
        for (int j = 0; j < terms; ++j) {
            double t = p[j * ACTION_SPLIT];
            for (int z = 1; z < ACTION_SPLIT; ++z) t += p[j * ACTION_SPLIT + z];
            v = j == 0 ? t : v + t;
        }
        if (v > bv) {                                           // first maximum among the candidates
            bv = v;
            best = a;
        }
    }
    action[b] = best;
}

template <typename T>
hipError_t launch_refine_action(const T* bel, int ldb, int B, const T* alpha, int lda, ModelView<T> mv, double gamma,
                                const int32_t* btl, const int32_t* btc, const int32_t* aqueue, const int* aqcount,
                                const double* rdot, const double* rdot_err, const int32_t* best_v,
                                const double* best_score, const double* err, double* val_exact, int32_t* action,
                                hipStream_t st, const uint8_t* acand) {
    if (B <= 0) return hipSuccess;
    const int terms = 1 + mv.O;
    if ((int64_t)mv.A * terms > 65535) return hipErrorInvalidValue;
    // queued beliefs are few (two dozen of 1024 at C4): a bounded grid strides over them.  (512 x 24 x 8 blocks, of which 23 x 24 x 8
    // had work, spent 15 of the kernel's 28 us dispatching empty blocks.)
    const int gx = B < 64 ? B : 64;
    hipLaunchKernelGGL(k_refine_action<T>, dim3(gx, mv.A * terms, ACTION_SPLIT), dim3(256), 0, st, bel, ldb, alpha, lda, mv, gamma, btl, btc,
                       aqueue, aqcount, rdot, rdot_err, best_v, best_score, err, val_exact, acand);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_action_final, dim3((B + 255) / 256), dim3(256), 0, st, mv.A, terms, aqueue, aqcount, val_exact, action);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// K3: alpha' rows.  src/pomdp.py:1497-1506 restricted to the winning action:
//   out[b][s] = ER[s,a*] + ((G0 + G1) + G2 ...),  G_o = gamma * sum_r rto * alpha_{v*[b,a*,o]}[rs]
// evaluated in f64 in the reference's association, rounded once to T.
// ------------------------------------------------------------------------- //
template <typename T>
__global__ void k_assemble(const T* __restrict__ alpha, int lda, ModelView<T> mv, double gamma,
                           const int32_t* __restrict__ action, const int32_t* __restrict__ best_v,
                           const int32_t* __restrict__ rows, const int* __restrict__ n_rows, int max_rows,
                           T* __restrict__ out, int ldo) {
#pragma clang fp contract(off)
    // Output row u is the alpha' of belief rows[u] (the first belief carrying that (a*, v*) key): only
    // unique rows are materialised.  rows == nullptr: u is the belief itself.
    // The number of rows is known on the device only: a bounded grid strides over them (a grid of max_rows row
    // blocks was 120 000 blocks at C4 of which 8 000 had work -- 37 us of launch for 10 us of arithmetic).
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= mv.S) return;
    const int n = n_rows != nullptr ? *n_rows : max_rows;
    for (int u = blockIdx.y; u < n; u += gridDim.y) {
        const int b = rows ? rows[u] : u;
        const int a = action[b];
        const int32_t* rs = mv.rs + (int64_t)a * mv.R * mv.S_pad;
        double total = 0.0;
        for (int o = 0; o < mv.O; ++o) {
            const int v = best_v[((int64_t)b * mv.A + a) * mv.O + o];
            const T* arow = alpha + (int64_t)v * lda;
            const T* rto = mv.rto + (int64_t)(a * mv.O + o) * mv.R * mv.S_pad;
            double g = 0.0;
            for (int r = 0; r < mv.R; ++r)
                g = g + (double)rto[(int64_t)r * mv.S_pad + s] * (double)arow[rs[(int64_t)r * mv.S_pad + s]];
            const double go = gamma * g;
            total = (o == 0) ? go : total + go;
        }
        out[(int64_t)u * ldo + s] = (T)((double)mv.er[(int64_t)a * mv.S_pad + s] + total);
    }
}

template <typename T>
hipError_t launch_assemble(const T* alpha, int lda, ModelView<T> mv, double gamma, const int32_t* action,
                           const int32_t* best_v, const int32_t* rows, const int* n_rows, int max_rows, T* out, int ldo,
                           hipStream_t st) {
    if (max_rows <= 0) return hipSuccess;
    const int sx = (mv.S + 255) / 256;
    int gy = (8192 + sx - 1) / sx;                       // ~8k blocks keep the chip busy whatever the row count turns out to be
    if (gy > max_rows) gy = max_rows;
    if (gy > 65535) gy = 65535;
    dim3 grid(sx, gy);
    hipLaunchKernelGGL(k_assemble<T>, grid, dim3(256), 0, st, alpha, lda, mv, gamma, action, best_v, rows, n_rows, max_rows, out,
                       ldo);
    return hipGetLastError();
}

// K6: device-side dedup.  Two beliefs with the same key (a*, v*[a*, 0..O-1]) get byte-identical alpha' rows
// (same arithmetic on the same operands), so only the first belief of each key is assembled:
//   rep[c]  = first belief (caller order) with c's key
//   uniq[u] = u-th belief with rep[c] == c;  inv[c] = u such that uniq[u] == rep[c]
// This finds a subset of the reference's byte-level duplicates (src/mdp.py:667-669); rows that coincide
// under different keys are still merged by the host-side byte dedup of the ValueFunction constructor.
__global__ void k_dedup_rep(int B, int A, int O, const int32_t* __restrict__ action, const int32_t* __restrict__ best_v,
                            int32_t* __restrict__ rep) {
    // one wavefront per belief c: lanes scan the earlier beliefs d < c in parallel, lowest match wins
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= B) return;
    const int a = action[c];
    const int32_t* kc = best_v + ((int64_t)c * A + a) * O;
    int r = c;
    for (int d0 = 0; d0 < c && r == c; d0 += 256) {        // four 64-belief steps in flight (16 dependent steps were 11 us)
        int hit[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = d0 + 64 * j + lane;
            hit[j] = 0;
            if (d < c && action[d] == a) {
                const int32_t* kd = best_v + ((int64_t)d * A + a) * O;
                hit[j] = 1;
                for (int o = 0; o < O; ++o) hit[j] &= (kd[o] == kc[o]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned long long m = __ballot(hit[j]);
            if (m && r == c) r = d0 + 64 * j + __ffsll((long long)m) - 1;      // wave-uniform; lowest match wins
        }
    }
    if (lane == 0) rep[c] = r;
}

__global__ void k_dedup_compact(int B, const int32_t* __restrict__ rep, int32_t* __restrict__ uniq,
                                int32_t* __restrict__ inv, int32_t* __restrict__ slot, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;      // one block of 1024 threads
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int c0 = 0; c0 < B; c0 += 1024) {
        const int c = c0 + tid;
        const int f = (c < B && rep[c] == c) ? 1 : 0;
        const unsigned long long mask = __ballot(f);
        if (lane == 0) wsum[wid] = __popcll(mask);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wid; ++w) off += wsum[w];
        if (f) {
            const int u = off + __popcll(mask & ((1ull << lane) - 1ull));
            uniq[u] = c;
            slot[c] = u;
        }
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wsum[w];
            base_sh += t;
        }
        __syncthreads();
    }
    if (tid == 0) *count = base_sh;
    __syncthreads();
    for (int c = tid; c < B; c += 1024) inv[c] = slot[rep[c]];        // rep[c] <= c and slot[rep] was written above
}

hipError_t launch_dedup(int B, int A, int O, const int32_t* action, const int32_t* best_v, int32_t* rep, int32_t* uniq,
                        int32_t* inv, int32_t* slot, int* count, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dedup_rep, dim3((B + 3) / 4), dim3(256), 0, st, B, A, O, action, best_v, rep);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dedup_compact, dim3(1), dim3(1024), 0, st, B, rep, uniq, inv, slot, count);
    return hipGetLastError();
}

// full[c][s] = uniq_rows[inv[c]][s]   (per-belief matrix of the reference seam, built on demand)
template <typename T>
__global__ void k_expand_rows(const T* __restrict__ uniq_rows, const int32_t* __restrict__ inv, T* __restrict__ full, int S) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    full[(int64_t)blockIdx.y * S + s] = uniq_rows[(int64_t)inv[blockIdx.y] * S + s];
}

template <typename T>
hipError_t launch_expand_rows(const T* uniq_rows, const int32_t* inv, T* full, int B, int S, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_expand_rows<T>, dim3((S + 255) / 256, B), dim3(256), 0, st, uniq_rows, inv, full, S);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// K5: belief-dominance test.  src/pomdp.py:1510-1512
// ------------------------------------------------------------------------- //
template <typename T>
__global__ void k_keep(const T* __restrict__ bel, int ldb, const T* __restrict__ uniq_rows, int ldo, int S,
                       const double* __restrict__ oldmax, const int32_t* __restrict__ inv,
                       const int32_t* __restrict__ perm, uint8_t* __restrict__ keep) {
    __shared__ double red[4];
    const int b = blockIdx.x;                               // engine (possibly sorted) belief order
    const int c = perm ? perm[b] : b;                       // caller order
    const double nv = block_sum(plain_dot_partial(bel + (int64_t)b * ldb, uniq_rows + (int64_t)inv[c] * ldo, S,
                                                  TileList{nullptr, (S + 31) >> 5}), red);
    if (threadIdx.x == 0) keep[c] = (nv > oldmax[b]) ? 1 : 0;
}

template <typename T>
hipError_t launch_keep(const T* bel, int ldb, const T* uniq_rows, int ldo, int B, int S, const double* oldmax,
                       const int32_t* inv, const int32_t* perm, uint8_t* keep, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_keep<T>, dim3(B), dim3(256), 0, st, bel, ldb, uniq_rows, ldo, S, oldmax, inv, perm, keep);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// prune level 2: one wavefront per ordered pair (i, j), early exit on the first
// state where alpha[j] < alpha[i].  src/mdp.py:857-866
// ------------------------------------------------------------------------- //
template <typename T>
__global__ void k_dominated(const T* __restrict__ alpha, int lda, int V, int S, int* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x;
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= V) return;
    const T* ai = alpha + (int64_t)i * lda;
    const T* aj = alpha + (int64_t)j * lda;
    for (int s0 = 0; s0 < S; s0 += 64) {
        const int s = s0 + lane;
        const int bad = (s < S) ? !(aj[s] >= ai[s]) : 0;
        if (__any(bad)) return;
    }
    if (lane == 0) atomicAdd(&cnt[i], 1);
}

template <typename T>
hipError_t launch_dominated(const T* alpha, int lda, int V, int S, int* cnt, hipStream_t st) {
    if (V <= 0) return hipSuccess;
    dim3 grid(V, (V + 3) / 4);
    if (grid.y > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_dominated<T>, grid, dim3(256), 0, st, alpha, lda, V, S, cnt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// Batched belief update (Bayes step).  Reference: Belief.update, src/pomdp.py:405-411 (one belief at a time,
// np.bincount scatter); batched form in the simulator, src/pomdp.py:3277-3310.
//   u[b, s'] = sum over (s, r) with rs[s, a_b, r] == s' of  b[b, s] * RTO[s, a_b, o_b, r];   b'[b] = u[b] / sum(u[b])
// Pull form over inverse transition lists (CSC built at engine creation, entries in ascending (s, r) order =
// bincount's accumulation order), so the sums are deterministic and need no atomics.
// ------------------------------------------------------------------------- //
template <typename T>
__global__ void k_belief_push(const T* __restrict__ bel, int ldb, ModelView<T> mv, const int32_t* __restrict__ in_ptr,
                              const int32_t* __restrict__ in_src, const int32_t* __restrict__ act,
                              const int32_t* __restrict__ obs, const int32_t* __restrict__ out_row,
                              double* __restrict__ unnorm, double* __restrict__ mass) {
    __shared__ double red[4];
    const int b = blockIdx.y, sp = blockIdx.x * 256 + threadIdx.x;
    if (out_row && out_row[b] < 0) return;                 // dropped row (whole block leaves together)
    const int a = act[b], o = obs[b];
    double u = 0.0;
    if (sp < mv.S) {
        const int32_t* ptr = in_ptr + (int64_t)a * (mv.S + 1);
        const int32_t* src = in_src + (int64_t)a * mv.S * mv.R;
        const T* rto = mv.rto + (int64_t)(a * mv.O + o) * mv.R * mv.S_pad;
        const T* brow = bel + (int64_t)b * ldb;
        for (int j = ptr[sp]; j < ptr[sp + 1]; ++j) {
            const int e = src[j];                           // e = s * R + r
            const int s = e / mv.R, r = e - s * mv.R;
            u += (double)brow[s] * (double)rto[(int64_t)r * mv.S_pad + s];
        }
        unnorm[(int64_t)b * mv.S + sp] = u;
    }
    const double tot = block_sum(u, red);
    if (threadIdx.x == 0) atomicAdd(&mass[b], tot);        // <= S/256 adds per belief; order only affects the last bit of the norm
}

template <typename T>
__global__ void k_belief_norm(const double* __restrict__ unnorm, const double* __restrict__ mass, int S,
                              const int32_t* __restrict__ out_row, T* __restrict__ out, int ldo) {
    const int b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int row = out_row ? out_row[b] : b;              // compaction: surviving rows move up, -1 = dropped
    if (row < 0) return;
    out[(int64_t)row * ldo + s] = (T)(unnorm[(int64_t)b * S + s] / mass[b]);   // mass 0 -> NaN, as the reference's 0/0
}

template <typename T>
hipError_t launch_belief_update(const T* bel, int ldb, int B, ModelView<T> mv, const int32_t* in_ptr, const int32_t* in_src,
                                const int32_t* act, const int32_t* obs, const int32_t* out_row, double* unnorm,
                                double* mass, T* out, int ldo, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    if (B > 65535) return hipErrorInvalidValue;
    dim3 grid((mv.S + 255) / 256, B);
    hipLaunchKernelGGL(k_belief_push<T>, grid, dim3(256), 0, st, bel, ldb, mv, in_ptr, in_src, act, obs, out_row, unnorm,
                       mass);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_belief_norm<T>, grid, dim3(256), 0, st, unnorm, mass, mv.S, out_row, out, ldo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// Belief-side formulation of the score GEMM's operands (used when B * A * O rows are cheaper than A * O * V):
// project every belief through every (a, o) instead of every alpha-vector.
//   bp[((o*A + a)*B + b)][s'] = gamma * sum_{(s,r): rs[s,a,r] = s'} b[b][s] * RTO[s,a,o,r]
//   score[b,a,o,v] = bp[g,b,:] . alpha[v,:]   ( = b . Gamma[a,o,v,:] of src/pomdp.py:1489-1495, re-associated)
// Pull form over the inverse transition lists (no atomics on bp), f64 accumulation, one rounding to T.
// ------------------------------------------------------------------------- //
constexpr int PUSH_NB = 4;   // beliefs per thread of k_push_project
constexpr int PUSH_NT = 8;   // 256-state chunks per block
template <typename T>
__global__ void k_push_project(const T* __restrict__ bel, int ldb, int B, ModelView<T> mv,
                               const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src, double gamma,
                               const T* __restrict__ amax, T* __restrict__ bp, int ldp, double* __restrict__ mag,
                               uint8_t* __restrict__ nzP /* [rows / 256][ldp / 32], zeroed by the caller: set where a projected
                                                            row has a non-zero in the K tile (or nullptr) */) {
    // One thread per target state s' and PUSH_NB beliefs: the inverse list of (a, s') and the weights of its entries do not
    // depend on the belief, so they are read once for the four.  A block walks PUSH_NT chunks of 256 states and reduces the
    // magnitudes once at its end: with one chunk per block the per-thread work (one or two list entries) was a tenth of the
    // block's reduction and its atomics -- 0.89 ms for 100 beliefs at the Sea-Robin shape whether or not anything was
    // gathered or stored.  Per (belief, observation) the sum runs over the list in order, as before: an entry whose belief
    // value is zero adds an exact zero.
    __shared__ double red[PUSH_NB][16];
    const int b0 = blockIdx.y * PUSH_NB, a = blockIdx.z;
    const int nb = B - b0 < PUSH_NB ? B - b0 : PUSH_NB;
    const int G = mv.A * mv.O;
    const int32_t* ptr = in_ptr + (int64_t)a * (mv.S + 1);
    const int32_t* src = in_src + (int64_t)a * mv.S * mv.R;
    for (int o0 = 0; o0 < mv.O; o0 += 4) {
        const int no = mv.O - o0 < 4 ? mv.O - o0 : 4;
        double mg[PUSH_NB][4];
#pragma unroll
        for (int k = 0; k < PUSH_NB; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) mg[k][q] = 0.0;
        for (int c = 0; c < PUSH_NT; ++c) {
            const int sp = (blockIdx.x * PUSH_NT + c) * 256 + threadIdx.x;
            if (sp >= ldp) break;                           // (whole waves: ldp is a multiple of 32, chunks of 256)
            const int j0 = sp < mv.S ? ptr[sp] : 0, j1 = sp < mv.S ? ptr[sp + 1] : 0;
            const double am = sp < mv.S ? fabs((double)amax[sp]) : 0.0;
            double acc[PUSH_NB][4];
#pragma unroll
            for (int k = 0; k < PUSH_NB; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[k][q] = 0.0;
            for (int j = j0; j < j1; ++j) {
                const int e = src[j];                       // e = s * R + r
                const int s = mv.R == 1 ? e : e / mv.R, r = e - s * mv.R;
                double bs[PUSH_NB];
                bool any = false;
#pragma unroll
                for (int k = 0; k < PUSH_NB; ++k) {
                    bs[k] = k < nb ? (double)bel[(int64_t)(b0 + k) * ldb + s] : 0.0;
                    any = any || bs[k] != 0.0;
                }
                if (any) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (q < no) {
                            const double w = (double)mv.rto[((int64_t)(a * mv.O + o0 + q) * mv.R + r) * mv.S_pad + s];
#pragma unroll
                            for (int k = 0; k < PUSH_NB; ++k) acc[k][q] += bs[k] * w;
                        }
                }
            }
#pragma unroll
            for (int k = 0; k < PUSH_NB; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (q < no && k < nb) {                 // uniform across the block
                        const T val = (T)(gamma * acc[k][q]);
                        const int64_t row = push_row_index(o0 + q, a, b0 + k, mv.A, B);      // see SlabView
                        bp[row * ldp + sp] = val;           // pad columns get exact zeros
                        mg[k][q] += fabs((double)val) * am;
                        if (nzP != nullptr) {               // the GEMM's zero-tile map, while the values are in registers
                            const unsigned long long nzm = __ballot(val != T(0));
                            const int lane = threadIdx.x & 63;
                            if ((lane == 0 && (nzm & 0xffffffffull)) || (lane == 32 && (nzm >> 32)))
                                nzP[(row >> 8) * (ldp >> 5) + (sp >> 5)] = 1;
                        }
                    }
        }
        // magnitudes: one barrier for all beliefs and (up to four) observations of this pass
#pragma unroll
        for (int k = 0; k < PUSH_NB; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) mg[k][q] = wave_sum(mg[k][q]);
        __syncthreads();                                    // red[] of the previous pass has been read
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int k = 0; k < PUSH_NB; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[k][(threadIdx.x >> 6) * 4 + q] = mg[k][q];
        }
        __syncthreads();
        if ((int)threadIdx.x < no * nb) {
            const int k = threadIdx.x / no, q = threadIdx.x % no;
            const double part = ((red[k][q] + red[k][4 + q]) + red[k][8 + q]) + red[k][12 + q];
            if (part != 0.0) atomicAdd(&mag[(int64_t)(b0 + k) * G + a * mv.O + o0 + q], part);
        }
    }
}

template <typename T>
hipError_t launch_push_project(const T* bel, int ldb, int B, ModelView<T> mv, const int32_t* in_ptr,
                               const int32_t* in_src, double gamma, const T* amax, T* bp, int ldp, double* mag,
                               hipStream_t st, uint8_t* nzP) {
    if (B <= 0) return hipSuccess;
    if (B > 65535 || mv.A > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_push_project<T>, dim3((ldp + 256 * PUSH_NT - 1) / (256 * PUSH_NT), (B + PUSH_NB - 1) / PUSH_NB, mv.A), dim3(256), 0, st, bel, ldb, B, mv, in_ptr,
                       in_src, gamma, amax, bp, ldp, mag, nzP);
    return hipGetLastError();
}

template <typename T>
__global__ void k_rdot(const T* __restrict__ bel, int ldb, ModelView<T> mv, const int32_t* __restrict__ btl,
                       const int32_t* __restrict__ btc, double* __restrict__ rd) {
    __shared__ double red[4];
    // One block per (belief, action).  (One block per belief walking the actions was B blocks for the whole chip: 2.4 ms
    // for 100 beliefs x 16 actions at |S| = 61875 -- and since it runs on the side stream beside the persistent score GEMM,
    // whose blocks need a CU to themselves, the GEMM started on the 156 CUs it left free: 5.9 ms at 0.69 of peak.)
    const int b = blockIdx.x, a = blockIdx.y;
    const int k_tiles = mv.S_pad >> 5;
    const TileList tl{btl ? btl + (int64_t)b * k_tiles : nullptr, btl ? btc[b] : k_tiles};
    const T* brow = bel + (int64_t)b * ldb;
    const double v = block_sum(plain_dot_partial(brow, mv.er + (int64_t)a * mv.S_pad, mv.S, tl), red);
    if (threadIdx.x == 0) rd[(int64_t)b * mv.A + a] = v;
}

template <typename T>
hipError_t launch_rdot(const T* bel, int ldb, int B, ModelView<T> mv, const int32_t* btl, const int32_t* btc, double* rd,
                       hipStream_t st) {
    if (B <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_rdot<T>, dim3(B, mv.A), dim3(256), 0, st, bel, ldb, mv, btl, btc, rd);
    return hipGetLastError();
}

// ------------------------------------------------------------------------- //
// Belief walk (FSVI-style expansion, src/pomdp.py:1895-1935): n sequential Bayes updates
//   b_{i+1} = normalise( sum_{(s,r)->s'} base_i[s] * RTO[s, a_i, o_i, r] ),   base_i = restart[i] ? b0 : b_i
// in fp64 whatever the engine's T (the host mirror keeps fp64 belief values).  One step = k_walk_push (pull over
// the inverse lists in bincount's accumulation order, per-block partial masses) + k_walk_norm (every block sums
// the partials in the same fixed order, so the normaliser is deterministic; writes the fp64 row and the T row of
// the device store).  Products and sums are not fused so the un-normalised values equal NumPy's bincount.
// ------------------------------------------------------------------------- //
// TT: type of the RTO table the walk reads (an f32 engine may hold an fp64 copy for it), T: type of the store rows
template <typename TT>
__global__ void k_walk_push(const double* __restrict__ base, const TT* __restrict__ rto_all, int S, int S_pad, int O, int R,
                            const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src, int a, int o,
                            double* __restrict__ unnorm, double* __restrict__ partial) {
#pragma clang fp contract(off)
    __shared__ double red[4];
    const int sp = blockIdx.x * 256 + threadIdx.x;
    double u = 0.0;
    if (sp < S) {
        const int32_t* ptr = in_ptr + (int64_t)a * (S + 1);
        const int32_t* src = in_src + (int64_t)a * S * R;
        const TT* rto = rto_all + (int64_t)(a * O + o) * R * S_pad;
        for (int j = ptr[sp]; j < ptr[sp + 1]; ++j) {
            const int e = src[j];
            const int s = e / R, r = e - s * R;
            const double w = (double)rto[(int64_t)r * S_pad + s] * base[s];
            u = u + w;
        }
        unnorm[sp] = u;
    }
    const double tot = block_sum(u, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// Sum of n block partials, computed identically by every block that calls it (fixed tree: lane-strided loads, wave
// shuffles, then the four wave sums in order), so all blocks of a step divide by the same mass.
__device__ __forceinline__ double partials_sum(const double* __restrict__ partial, int n, double* red /* [4] shared */) {
    double m = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) m += partial[i];
    for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    const double tot = ((red[0] + red[1]) + red[2]) + red[3];
    __syncthreads();
    return tot;
}

template <typename T>
__global__ void k_walk_norm(const double* __restrict__ unnorm, const double* __restrict__ partial, int n_partial, int S,
                            int S_pad, double* __restrict__ out64, T* __restrict__ out_store) {
    __shared__ double red_m[4];
    const double mass_sh = partials_sum(partial, n_partial, red_m);
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < S) {
        const double v = unnorm[s] / mass_sh;
        out64[s] = v;
        out_store[s] = (T)v;
    } else if (s < S_pad) {
        out_store[s] = T(0);
    }
}

// One kernel per step of a chain: step i pushes belief i through (a, o) AND writes belief i itself, normalising the
// previous step's raw result on the fly (every block re-adds the previous step's block partials in the same order, so
// all blocks divide by the same mass; a gathered value prev_unnorm[s] / mass is the very double k_walk_norm would have
// stored).  Halves the dependent launches of a walk, which is what its time consists of (~6 us each).
//   plain_base : belief to push when it is already normalised (b0: first step and restarts), else nullptr
//   prev_*     : previous step's raw row / partial sums / destinations of its normalised row (all nullptr at step 0)
template <typename TT, typename T>
__global__ void k_walk_fused(const double* __restrict__ plain_base, const double* __restrict__ prev_unnorm,
                             const double* __restrict__ prev_partial, int n_partial, double* __restrict__ prev_out64,
                             T* __restrict__ prev_out_store, const TT* __restrict__ rto_all, int S, int S_pad, int O, int R,
                             const int32_t* __restrict__ in_ptr, const int32_t* __restrict__ in_src, int a, int o,
                             double* __restrict__ unnorm, double* __restrict__ partial) {
#pragma clang fp contract(off)
    __shared__ double red[4];
    const double mass = prev_unnorm != nullptr ? partials_sum(prev_partial, n_partial, red) : 1.0;
    const int sp = blockIdx.x * 256 + threadIdx.x;
    if (prev_out64 != nullptr) {                         // the previous belief, normalised (k_walk_norm's statement)
        if (sp < S) {
            const double v = prev_unnorm[sp] / mass;
            prev_out64[sp] = v;
            prev_out_store[sp] = (T)v;
        } else if (sp < S_pad) {
            prev_out_store[sp] = T(0);
        }
    }
    double u = 0.0;
    if (sp < S) {
        const int32_t* ptr = in_ptr + (int64_t)a * (S + 1);
        const int32_t* src = in_src + (int64_t)a * S * R;
        const TT* rto = rto_all + (int64_t)(a * O + o) * R * S_pad;
        for (int j = ptr[sp]; j < ptr[sp + 1]; ++j) {
            const int e = src[j];
            const int s = e / R, r = e - s * R;
            const double bs = plain_base != nullptr ? plain_base[s] : prev_unnorm[s] / mass;
            const double w = (double)rto[(int64_t)r * S_pad + s] * bs;
            u = u + w;
        }
        unnorm[sp] = u;
    }
    const double tot = block_sum(u, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

template <typename T>
hipError_t launch_walk_fused(const double* plain_base, const double* prev_unnorm, const double* prev_partial, double* prev_out64,
                             T* prev_out_store, ModelView<T> mv, const double* rto64, const int32_t* in_ptr, const int32_t* in_src,
                             int a, int o, double* unnorm, double* partial, hipStream_t st) {
    const int blocks = (mv.S_pad + 255) / 256;
    if (rto64 != nullptr)
        hipLaunchKernelGGL((k_walk_fused<double, T>), dim3(blocks), dim3(256), 0, st, plain_base, prev_unnorm, prev_partial, blocks,
                           prev_out64, prev_out_store, rto64, mv.S, mv.S_pad, mv.O, mv.R, in_ptr, in_src, a, o, unnorm, partial);
    else
        hipLaunchKernelGGL((k_walk_fused<T, T>), dim3(blocks), dim3(256), 0, st, plain_base, prev_unnorm, prev_partial, blocks,
                           prev_out64, prev_out_store, mv.rto, mv.S, mv.S_pad, mv.O, mv.R, in_ptr, in_src, a, o, unnorm, partial);
    return hipGetLastError();
}

// the last belief of a chain: only the normalisation is left
template <typename T>
hipError_t launch_walk_finish(const double* unnorm, const double* partial, ModelView<T> mv, double* out64, T* out_store,
                              hipStream_t st) {
    const int blocks = (mv.S_pad + 255) / 256;
    hipLaunchKernelGGL(k_walk_norm<T>, dim3(blocks), dim3(256), 0, st, unnorm, partial, blocks, mv.S, mv.S_pad, out64, out_store);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_walk_step(const double* base, ModelView<T> mv, const double* rto64, const int32_t* in_ptr,
                            const int32_t* in_src, int a, int o, double* unnorm, double* partial, double* out64,
                            T* out_store, hipStream_t st) {
    const int blocks = (mv.S_pad + 255) / 256;
    if (rto64 != nullptr)
        hipLaunchKernelGGL(k_walk_push<double>, dim3(blocks), dim3(256), 0, st, base, rto64, mv.S, mv.S_pad, mv.O, mv.R,
                           in_ptr, in_src, a, o, unnorm, partial);
    else
        hipLaunchKernelGGL(k_walk_push<T>, dim3(blocks), dim3(256), 0, st, base, mv.rto, mv.S, mv.S_pad, mv.O, mv.R, in_ptr,
                           in_src, a, o, unnorm, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_walk_norm<T>, dim3(blocks), dim3(256), 0, st, unnorm, partial, blocks, mv.S, mv.S_pad, out64,
                       out_store);
    return hipGetLastError();
}

// explicit instantiations
#define PBVI_INST(T)                                                                                                   \
    template hipError_t launch_support<T>(ModelView<T>, uint8_t*, hipStream_t);                                        \
    template hipError_t launch_project<T>(const T*, int, int, ModelView<T>, T, T*, int, const uint8_t*, int,           \
                                          hipStream_t, const uint8_t*, const int*, int, const int32_t*, const int32_t*); \
    template hipError_t launch_tail_rows<T>(ModelView<T>, T*, int64_t, int, hipStream_t, const int32_t*);               \
    template hipError_t launch_dead<T>(const T*, int, int, ModelView<T>, const unsigned long long*, int, uint8_t*, int32_t*, \
                                       int32_t*, int*, hipStream_t, const uint8_t*, const int32_t*);                                                   \
    template hipError_t launch_belief_tiles<T>(const T*, int, int, int, int, int32_t*, int32_t*, hipStream_t);         \
    template hipError_t launch_argmax<T>(SlabView<T>, int, int, int, const uint8_t*, double, double, const int*, int,  \
                                         int32_t*, double*, double*, int32_t*, int*, hipStream_t, double);             \
    template hipError_t launch_refine<T, T>(bool, SlabView<T>, int, int, int, const int32_t*, const int*, const T*, int, \
                                         const T*, int, ModelView<T>, double, const int32_t*, const int32_t*,          \
                                         const uint8_t*, int32_t*, double*, double*, int*, RefineWork, hipStream_t);   \
    template hipError_t launch_action<T>(int, ModelView<T>, SlabView<T>, int64_t, double, const int*, const double*,   \
                                         const double*, double*, double*, int32_t*, int32_t*, int*, hipStream_t, double, uint8_t*); \
    template hipError_t launch_refine_action<T>(const T*, int, int, const T*, int, ModelView<T>, double,               \
                                                const int32_t*, const int32_t*, const int32_t*, const int*,            \
                                                const double*, const double*, const int32_t*, const double*,           \
                                                const double*, double*, int32_t*, hipStream_t, const uint8_t*);                        \
    template hipError_t launch_assemble<T>(const T*, int, ModelView<T>, double, const int32_t*, const int32_t*,        \
                                           const int32_t*, const int*, int, T*, int, hipStream_t);                     \
    template hipError_t launch_expand_rows<T>(const T*, const int32_t*, T*, int, int, hipStream_t);                    \
    template hipError_t launch_keep<T>(const T*, int, const T*, int, int, int, const double*, const int32_t*,          \
                                       const int32_t*, uint8_t*, hipStream_t);                                         \
    template hipError_t launch_belief_update<T>(const T*, int, int, ModelView<T>, const int32_t*, const int32_t*,      \
                                                const int32_t*, const int32_t*, const int32_t*, double*, double*, T*,  \
                                                int, hipStream_t);                                                     \
    template hipError_t launch_push_project<T>(const T*, int, int, ModelView<T>, const int32_t*, const int32_t*,       \
                                               double, const T*, T*, int, double*, hipStream_t, uint8_t*);                      \
    template hipError_t launch_rdot<T>(const T*, int, int, ModelView<T>, const int32_t*, const int32_t*, double*,      \
                                       hipStream_t);                                                                   \
    template hipError_t launch_walk_step<T>(const double*, ModelView<T>, const double*, const int32_t*, const int32_t*, \
                                            int, int, double*, double*, double*, T*, hipStream_t);                     \
    template hipError_t launch_walk_fused<T>(const double*, const double*, const double*, double*, T*, ModelView<T>,      \
                                             const double*, const int32_t*, const int32_t*, int, int, double*, double*,   \
                                             hipStream_t);                                                               \
    template hipError_t launch_walk_finish<T>(const double*, const double*, ModelView<T>, double*, T*, hipStream_t);     \
    template hipError_t launch_dominated<T>(const T*, int, int, int, int*, hipStream_t);
PBVI_INST(float)
PBVI_INST(double)
#define PBVI_INST_REFINE(T, TS)                                                                                        \
    template hipError_t launch_refine_scan<T, TS>(bool, SlabView<TS>, int, int, int, const int32_t*, const int*, const T*, \
                                                  int, const T*, int, ModelView<T>, double, const int32_t*, const int32_t*, \
                                                  const uint8_t*, int32_t*, double*, double*, int*, RefineWork, int*,   \
                                                  hipStream_t);
PBVI_INST_REFINE(float, float)
PBVI_INST_REFINE(double, double)
PBVI_INST_REFINE(double, float)
template hipError_t launch_refine_deferred<float>(bool, int, int, const float*, int, const float*, int, ModelView<float>, double,
                                                  int32_t*, double*, double*, RefineWork, int, int, hipStream_t);
template hipError_t launch_refine_deferred<double>(bool, int, int, const double*, int, const double*, int, ModelView<double>,
                                                   double, int32_t*, double*, double*, RefineWork, int, int, hipStream_t);
// fp64 data re-scored behind an fp32 screen (candidates flagged from fp32 slabs)
template hipError_t launch_refine<double, float>(bool, SlabView<float>, int, int, int, const int32_t*, const int*, const double*,
                                                 int, const double*, int, ModelView<double>, double, const int32_t*,
                                                 const int32_t*, const uint8_t*, int32_t*, double*, double*, int*, RefineWork,
                                                 hipStream_t);

}  // namespace pbvi

// Engine object behind include/pbvi_hip.h: device residency of the model tables,
// the alpha-vector set and the belief block, and the kernel sequence of one backup.
//
// Device layouts (T = float | double, S_pad = S rounded up to 32, zero padded):
//   model   : see ModelView in backup_kernels.h
//   alpha   : [V_cap][S_pad]   rows 0..V-1 the set, row V = column-wise max |alpha|
//             (the magnitude row that bounds sum_s b_s |Gamma_v,s| for the tie window)
//   beliefs : [B_pad][S_pad]   B_pad = B rounded up to 256 (zero rows)
//   Gamma   : [N_pad][S_pad]   row (a*O+o)*(V+1)+v, v = V being the magnitude row
//   scores  : [split_k][B_pad][N_pad] f32 partial slabs  (f64 engines: [B][N])
//   results : out_alpha [B][S], action [B], best_v [B][A][O], keep [B]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

#include "backup_kernels.h"

namespace pbvi {

static thread_local std::string g_err;
static int g_poison = -1;   // -1: read PBVI_POISON from the environment on first use
static bool poison_enabled() {
    if (g_poison < 0) g_poison = getenv("PBVI_POISON") != nullptr ? 1 : 0;
    return g_poison == 1;
}
void set_error(const std::string& msg) { g_err = msg; }

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                    \
            (void)hipGetLastError();                                                         \
            return (_e == hipErrorOutOfMemory || _e == hipErrorMemoryAllocation) ? PBVI_ENOMEM : PBVI_ERUNTIME; \
        }                                                                                    \
    } while (0)

#define FAIL(code, msg)        \
    do {                       \
        set_error(msg);        \
        return (code);         \
    } while (0)

// Host-to-host copy of a result out of the pinned bounce buffer into the caller's (pageable) memory.  One thread moves
// ~12-15 GB/s into cold pages -- for the 24 MB of a belief walk or the 123 MB of an expanded alpha' matrix that, not the
// PCIe transfer (50 GB/s into pinned memory), was the time of the call -- so copies of a few MB and more are split over a
// small pool of worker threads that lives as long as the process (started on first use, parked on a condition variable;
// starting threads per call cost ~0.1 ms each, more than the solve loop's 6-8 MB copies gain).
class CopyPool {
public:
    static CopyPool& get() {
        static CopyPool* pool = new CopyPool();              // leaked on purpose: no destructor order to get wrong at exit
        return *pool;
    }
    void copy(void* dst, const void* src, size_t bytes) {
        constexpr size_t kMinPart = (size_t)2 << 20;
        const unsigned ways = (unsigned)std::min<size_t>(workers_.size() + 1, bytes / kMinPart);
        if (ways <= 1 || getpid() != pid_) {                 // (a forked child has the object but not its threads)
            std::memcpy(dst, src, bytes);
            return;
        }
        const size_t part = ((bytes / ways) + 4095) & ~(size_t)4095;
        unsigned posted = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (unsigned w = 1; w < ways; ++w) {
                const size_t o = (size_t)w * part;
                if (o >= bytes) break;
                tasks_.push_back(Task{static_cast<char*>(dst) + o, static_cast<const char*>(src) + o, std::min(part, bytes - o)});
                ++posted;
            }
            pending_ += posted;
        }
        cv_.notify_all();
        std::memcpy(dst, src, std::min(part, bytes));
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [&] { return pending_ == 0; });
    }

private:
    struct Task {
        char* dst;
        const char* src;
        size_t n;
    };
    CopyPool() {
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        unsigned n = getenv("PBVI_COPY_THREADS") ? (unsigned)std::max(1, atoi(getenv("PBVI_COPY_THREADS"))) : 4u;
        n = std::min(n, hw);
        for (unsigned i = 1; i < n; ++i) {
            workers_.emplace_back([this] { run(); });
            workers_.back().detach();
        }
    }
    void run() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return !tasks_.empty(); });
                t = tasks_.back();
                tasks_.pop_back();
            }
            std::memcpy(t.dst, t.src, t.n);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    const pid_t pid_ = getpid();
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<Task> tasks_;
    size_t pending_ = 0;
    std::vector<std::thread> workers_;
};
static void host_copy(void* dst, const void* src, size_t bytes) { CopyPool::get().copy(dst, src, bytes); }

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    // Deterministic out-of-memory for tests of the MemoryError contract (src/pomdp.py:2399-2401): an engine may hold at
    // most this many bytes of device buffers (pbvi_debug_alloc_limit / PBVI_ALLOC_LIMIT_MB; < 0 = no cap).
    static int64_t& limit_bytes() {
        static int64_t v = getenv("PBVI_ALLOC_LIMIT_MB") ? (int64_t)atoll(getenv("PBVI_ALLOC_LIMIT_MB")) << 20 : -1;
        return v;
    }
    static bool over_limit(int64_t held, size_t want) { return limit_bytes() >= 0 && held + (int64_t)want > limit_bytes(); }
    // Growth headroom (below) is for engines with memory to spare: under a cap only while the engine stays below 3/4 of it,
    // on the device only while four times the request is free -- an early buffer's spare half must not be what a later
    // buffer of the same call lacks.
    static bool headroom_ok(int64_t held, size_t want) {
        if (limit_bytes() >= 0) return held + (int64_t)want <= limit_bytes() / 4 * 3;
        if (want < ((size_t)256 << 20)) return true;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return fr / 4 >= want;
    }
    // grow-only; contents are NOT preserved on growth.  A buffer that grows again doubles (25 % if that fails): in a solve
    // loop the alpha set gains a few rows per backup, and re-allocating a multi-GB buffer (hipFree + hipMalloc,
    // both synchronising, hundreds of ms at 20 GB) on every call would dwarf the backup itself.
    int ensure(size_t bytes, int64_t* total) {
        if (bytes <= cap) return PBVI_OK;
        const bool regrow = p != nullptr;
        if (p) {
            (void)hipFree(p);
            *total -= (int64_t)cap;
            p = nullptr;
            cap = 0;
        }
        if (regrow) {
            for (const size_t want : {bytes * 2, bytes + bytes / 4}) {     // HBM is plentiful; re-allocations are not
                if (!headroom_ok(*total, want)) continue;
                if (hipMalloc(&p, want) == hipSuccess) {
                    cap = want;
                    *total += (int64_t)want;
                    if (poison_enabled()) {
                        if (hipMemset(p, 0xFF, want) != hipSuccess) (void)hipGetLastError();
                        (void)hipDeviceSynchronize();
                    }
                    return PBVI_OK;
                }
                (void)hipGetLastError();                 // no room for that much headroom: try less, then the exact size
                p = nullptr;
            }
        }
        if (over_limit(*total, bytes)) {
            set_error("device allocation of " + std::to_string(bytes) + " bytes refused: the engine holds " + std::to_string(*total) +
                      " bytes and its cap is " + std::to_string(limit_bytes()) + " (pbvi_debug_alloc_limit)");
            return PBVI_ENOMEM;
        }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
            set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
            return PBVI_ENOMEM;
        }
        cap = bytes;
        *total += (int64_t)bytes;
        // Debug: fill fresh allocations with 0xFF (NaN floats, -1 ints) so any read of memory the engine did
        // not write shows up in the parity tests instead of hiding behind zero-filled fresh pages.
        if (poison_enabled()) {   // null-stream memset does not order against the engine's non-blocking streams: fence it
            if (hipMemset(p, 0xFF, bytes) != hipSuccess) (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
        return PBVI_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename U>
    U* as() const { return reinterpret_cast<U*>(p); }
};

// out[s] = max_v |alpha[v][s]| (out zeroed by the caller).  Row chunks of 128 run as separate blocks and meet in an
// atomic max on the bit pattern (non-negative IEEE values order like unsigned integers): with one block per
// column strip the 5000 dependent loads of a grown alpha set made this latency-bound (0.6 ms at V = 4500).
// true iff ptr is device memory (hipMalloc'ed) rather than ordinary or pinned host memory
static bool is_device_pointer(const void* ptr) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();                         // unregistered host memory
        return false;
    }
    return attr.type == hipMemoryTypeDevice;
}
// true iff ptr is page-locked host memory the DMA engines can write directly (hipHostMalloc / pbvi_host_alloc)
static bool is_pinned_host_pointer(const void* ptr) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

// keys of the unique rows of the last backup: key[u] = (a*, v*[a*, 0..O-1]) of the u-th distinct (a*, v*) pair
__global__ void k_gather_keys(int U, int A, int O, const int32_t* __restrict__ uniq, const int32_t* __restrict__ action,
                              const int32_t* __restrict__ best_v, int32_t* __restrict__ keys) {
    const int u = blockIdx.x * 64 + threadIdx.x;
    if (u >= U) return;
    const int b = uniq[u], a = action[b];
    keys[(int64_t)u * (1 + O)] = a;
    for (int o = 0; o < O; ++o) keys[(int64_t)u * (1 + O) + 1 + o] = best_v[((int64_t)b * A + a) * O + o];
}

// everything a rank contributes to the exchange, in one int32 buffer:
//   [0] U | [1, 1+B) index | [1+B, 1+2B) action | [1+2B, 1+3B) keep | [1+3B, 1+3B+B*(1+O)) keys of the U distinct rows (rest 0)
// `per` >= B is the common block size of a sharded run (ceil(B_total / ranks)): the message has per-sized sections so
// that every rank sends the same number of integers; entries of beliefs >= B are 0.
__global__ void k_pack_exchange(int B, int per, int U, int A, int O, const int32_t* __restrict__ inv, const int32_t* __restrict__ action,
                                const uint8_t* __restrict__ keep, const int32_t* __restrict__ uniq,
                                const int32_t* __restrict__ best_v, int32_t* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0) out[0] = U;
    if (c >= per) return;
    int32_t* k = out + 1 + 3 * (int64_t)per + (int64_t)c * (1 + O);
    if (c >= B) {
        out[1 + c] = 0;
        out[1 + per + c] = 0;
        out[1 + 2 * per + c] = 0;
        for (int o = 0; o <= O; ++o) k[o] = 0;
        return;
    }
    out[1 + c] = inv[c];
    out[1 + per + c] = action[c];
    out[1 + 2 * per + c] = keep[c];
    if (c < U) {
        const int b = uniq[c], a = action[b];
        k[0] = a;
        for (int o = 0; o < O; ++o) k[1 + o] = best_v[((int64_t)b * A + a) * O + o];
    } else {
        for (int o = 0; o <= O; ++o) k[o] = 0;
    }
}

// the inverse: per-row action / best-alpha arrays in the layout k_assemble reads (only the winning action's entries)
__global__ void k_scatter_keys(int n, int A, int O, int V, const int32_t* __restrict__ keys, int32_t* __restrict__ action,
                               int32_t* __restrict__ best_v, int* __restrict__ bad) {
    const int u = blockIdx.x * 64 + threadIdx.x;
    if (u >= n) return;
    const int a = keys[(int64_t)u * (1 + O)];
    if (a < 0 || a >= A) {
        atomicAdd(bad, 1);
        action[u] = 0;
        return;
    }
    action[u] = a;
    for (int o = 0; o < O; ++o) {
        const int v = keys[(int64_t)u * (1 + O) + 1 + o];
        if (v < 0 || v >= V) atomicAdd(bad, 1);
        best_v[((int64_t)u * A + a) * O + o] = (v < 0 || v >= V) ? 0 : v;
    }
}

template <typename T>
__global__ void k_col_absmax(const T* __restrict__ alpha, int lda, int V, int S_pad, T* __restrict__ out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S_pad) return;
    const int v0 = blockIdx.y * 128, v1 = min(V, v0 + 128);
    T m[4] = {T(0), T(0), T(0), T(0)};
    for (int v = v0; v < v1; v += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const T x = (v + j < v1) ? alpha[(int64_t)(v + j) * lda + s] : T(0);
            const T ax = x < T(0) ? -x : x;
            m[j] = ax > m[j] ? ax : m[j];                    // NaN never wins, as before
        }
    }
    const T a = m[0] > m[1] ? m[0] : m[1], b = m[2] > m[3] ? m[2] : m[3];
    const T r = a > b ? a : b;
    if constexpr (sizeof(T) == 4)
        atomicMax(reinterpret_cast<unsigned int*>(out + s), __float_as_uint((float)r));
    else
        atomicMax(reinterpret_cast<unsigned long long*>(out + s), (unsigned long long)__double_as_longlong((double)r));
}

// Dense projection mode: D[ao][s][s'] = sum_{r: rs[s,a,r] = s'} RTO[s,a,o,r]  (one thread per row, no races)
template <typename T>
__global__ void k_densify(ModelView<T> mv, T* __restrict__ D, int64_t rows_pad) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int ao = blockIdx.y;
    if (s >= mv.S) return;
    const int a = ao / mv.O;
    T* row = D + ((int64_t)ao * rows_pad + s) * mv.S_pad;
    for (int r = 0; r < mv.R; ++r)
        row[mv.rs[((int64_t)a * mv.R + r) * mv.S_pad + s]] += mv.rto[((int64_t)ao * mv.R + r) * mv.S_pad + s];
}

// Gamma rows <- gamma * (alpha . D_ao^T) rows: product row v of batch ao goes to Gamma row ao*V+v.
template <typename T>
__global__ void k_scale_rows(const T* __restrict__ prod, int64_t batch_stride, int ld_prod, int V, T gamma,
                             T* __restrict__ gam, int ldg, int width) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int v = blockIdx.y, ao = blockIdx.z;
    if (s >= width) return;
    gam[((int64_t)ao * V + v) * ldg + s] = gamma * prod[(int64_t)ao * batch_stride + (int64_t)v * ld_prod + s];
}

// The A*O magnitude rows (tail of Gamma) from the ELL tables: one row each, not worth a GEMM tile row.
template <typename T>
__global__ void k_project_mag(const T* __restrict__ amax, ModelView<T> mv, T gamma, T* __restrict__ gam_tail, int ldg) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    const int ao = blockIdx.y;
    if (s >= mv.S_pad) return;
    const int a = ao / mv.O;
    T acc = T(0);
    for (int r = 0; r < mv.R; ++r)
        acc += mv.rto[((int64_t)ao * mv.R + r) * mv.S_pad + s] * amax[mv.rs[((int64_t)a * mv.R + r) * mv.S_pad + s]];
    gam_tail[(int64_t)ao * ldg + s] = gamma * acc;
}

// Indexing of a belief block (pbvi_beliefs_set / pbvi_beliefs_select), all on the device and stream-ordered -- a solve
// backs every new block up exactly once, so this is part of every backup:
//   k_row_flags  one pass over the source rows: flags[b][kt] = row b has mass in K tile kt (32 states), key[b] = its
//                first such tile
//   k_rank_sort  rows ordered by key (stable): the 256-row blocks of the score GEMM get tighter joint supports and more
//                zero tiles to skip (33.8 % -> 29.1 % of the tile steps at |S| = 30000)
//   k_gather_rows_v  the block in that order, 16 bytes per lane, straight from the store rows (no staging copy)
//   tile_or_block  the GEMM's zero map of the block from the row flags (the first rows of k_gather_rows_v's grid)
// The per-belief tile lists and the dead-triple test (backup_kernels.hip) read the same flags.
template <typename T>
__global__ void k_row_flags(const T* __restrict__ src, int ld, const int32_t* __restrict__ ids, int S_pad, int k_tiles,
                            uint8_t* __restrict__ flags, int32_t* __restrict__ key) {
    constexpr int NS = 16 / (int)sizeof(T);              // states per 16-byte chunk
    constexpr int LP = GEMM_BK / NS;                     // lanes per K tile (8 or 16)
    typedef T TN __attribute__((ext_vector_type(NS)));
    __shared__ int red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const T* row = src + (int64_t)(ids ? ids[b] : b) * ld;
    const int chunks = S_pad / NS;                       // rows are padded with zeros to S_pad (a multiple of 32)
    int first = 0x7fffffff;
    for (int c0 = 0; c0 < chunks; c0 += 1024) {          // four 16-byte loads per lane in flight
        TN v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j * 256 + tid;
            TN z;
#pragma unroll
            for (int e = 0; e < NS; ++e) z[e] = T(0);
            v[j] = c < chunks ? *(const TN*)(row + (int64_t)c * NS) : z;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j * 256 + tid;
            int f = 0;
#pragma unroll
            for (int e = 0; e < NS; ++e) f |= (v[j][e] != T(0)) ? 1 : 0;
#pragma unroll
            for (int off = 1; off < LP; off <<= 1) f |= __shfl_xor(f, off, 64);
            const int kt = c / LP;
            if (c < chunks && (tid & (LP - 1)) == 0) flags[(int64_t)b * k_tiles + kt] = (uint8_t)f;
            if (f && c < chunks && kt < first) first = kt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(first, off, 64);
        first = o < first ? o : first;
    }
    if ((tid & 63) == 0) red[tid >> 6] = first;
    __syncthreads();
    if (tid == 0) {
        int m = red[0];
        for (int w = 1; w < 4; ++w) m = red[w] < m ? red[w] : m;
        key[b] = m == 0x7fffffff ? k_tiles : m;          // an empty row sorts behind every other
    }
}

// perm[rank of row i] = i, rank = rows with a smaller (key, index): a stable sort as a rank computation -- no atomics,
// deterministic.  One wave per row (64 lanes share the B comparisons), so 1024 rows keep every CU busy for ~3 us
// (one thread per row was 23 us on four CUs).
__global__ void k_rank_sort(const int32_t* __restrict__ key, int B, int32_t* __restrict__ perm) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= B) return;
    const int ki = key[i];
    int rank = 0;
    for (int j = lane; j < B; j += 64) {
        const int kj = key[j];
        rank += (kj < ki || (kj == ki && j < i)) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off, 64);
    if (lane == 0) perm[rank] = i;
}

// dst row y = src row ids[perm[y]] (either map may be null), 16 bytes per lane; ld is a multiple of 32 elements
__device__ __forceinline__ void tile_or_block(const uint8_t* __restrict__ flags, const int32_t* __restrict__ perm, int B, int k_tiles,
                                              uint8_t* __restrict__ nz, int bx, int tile);
template <typename T>
__global__ void k_gather_rows_v(const T* __restrict__ src, T* __restrict__ dst, int ld, const int32_t* __restrict__ perm,
                                const int32_t* __restrict__ ids, const uint8_t* __restrict__ flags = nullptr, int B = 0,
                                int k_tiles = 0, uint8_t* __restrict__ nz = nullptr, int or_rows = 0) {
    // The first or_rows rows of the grid build the block's zero map (they need the order only, like the
    // gather): as a kernel of its own in front of the gather it was 14 us on the path to the score GEMM.
    if ((int)blockIdx.y < or_rows) {
        if ((int)blockIdx.x * 64 < k_tiles) tile_or_block(flags, perm, B, k_tiles, nz, blockIdx.x, blockIdx.y);
        return;
    }
    constexpr int NS = 16 / (int)sizeof(T);
    typedef T TN __attribute__((ext_vector_type(NS)));
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c * NS >= ld) return;
    const int row = blockIdx.y - or_rows;
    int r = row;
    if (perm) r = perm[r];
    if (ids) r = ids[r];
    *(TN*)(dst + (int64_t)row * ld + (int64_t)c * NS) = *(const TN*)(src + (int64_t)r * ld + (int64_t)c * NS);
}

// nz[tile][kt] = OR of flags[perm[r]][kt] over the rows r of the 256-row tile (rows >= B are padding).  Block =
// 16 row slices x 16 words of 4 K tiles: 16 rows per thread, then an OR over the slices in LDS (one thread per K tile
// looping over 256 rows was a 76 us chain of dependent byte loads in front of the GEMM's tile lists).
__device__ __forceinline__ void tile_or_block(const uint8_t* __restrict__ flags, const int32_t* __restrict__ perm, int B, int k_tiles,
                                              uint8_t* __restrict__ nz, int bx, int tile) {
    __shared__ uint32_t part[16][17];
    const int cw = threadIdx.x & 15, rs = threadIdx.x >> 4;
    const int kt0 = (bx * 16 + cw) * 4;                          // this thread's 4 K tiles
    const int r1 = (tile + 1) * 256 < B ? (tile + 1) * 256 : B;
    uint32_t f = 0;
    if (kt0 < k_tiles) {
        const bool whole = kt0 + 4 <= k_tiles && (k_tiles & 3) == 0;   // aligned 4-byte loads
#pragma unroll 4
        for (int r = tile * 256 + rs; r < r1; r += 16) {
            const uint8_t* p = flags + (int64_t)(perm ? perm[r] : r) * k_tiles + kt0;
            if (whole) {
                f |= *(const uint32_t*)p;
            } else {
                for (int e = 0; e < 4 && kt0 + e < k_tiles; ++e) f |= (uint32_t)p[e] << (8 * e);
            }
        }
    }
    part[rs][cw] = f;
    __syncthreads();
    if (rs == 0 && kt0 < k_tiles) {
        for (int x = 1; x < 16; ++x) f |= part[x][cw];
        for (int e = 0; e < 4 && kt0 + e < k_tiles; ++e) nz[(int64_t)tile * k_tiles + kt0 + e] = ((f >> (8 * e)) & 0xffu) ? 1 : 0;
    }
}

template <typename T>
__global__ void k_gather_rows(const T* __restrict__ src, T* __restrict__ dst, int ld, const int32_t* __restrict__ perm) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= ld) return;
    dst[(int64_t)blockIdx.y * ld + s] = src[(int64_t)perm[blockIdx.y] * ld + s];
}

// rows perm[y] of src (leading dimension ld_src) -> rows y of dst (ld_dst), first `cols` columns
template <typename T>
__global__ void k_gather_rows_ld(const T* __restrict__ src, int ld_src, T* __restrict__ dst, int ld_dst, int cols,
                                 const int32_t* __restrict__ perm) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= cols) return;
    dst[(int64_t)blockIdx.y * ld_dst + s] = src[(int64_t)perm[blockIdx.y] * ld_src + s];
}

// hash[r] = sum_i bits(row r, element i) * (2 i + 1)  (mod 2^64): the key the host's alpha-vector container hashes its
// rows by (mdp.py::_AlphaKey -- equality is still decided on the bytes).  Position-weighted: a solve's alpha-vectors are
// largely shifted copies of one another, which a plain sum of bit patterns cannot tell apart.
template <typename T>
__global__ void k_row_hash(const T* __restrict__ rows, int ld, int S, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long red[4];
    const T* p = rows + (int64_t)blockIdx.x * ld;
    unsigned long long acc = 0;
    for (int s = threadIdx.x; s < S; s += 256) {
        unsigned long long bits;
        if constexpr (sizeof(T) == 4)
            bits = (unsigned long long)__float_as_uint((float)p[s]);
        else
            bits = (unsigned long long)__double_as_longlong((double)p[s]);
        acc += bits * (2ull * (unsigned long long)s + 1ull);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// max_v score[row0 + b][v] for the B extra rows of the belief-side GEMM (the beliefs themselves multiplied by the alpha
// set: b . alpha_v), one wave per row; out[perm ? perm[b] : b]
template <typename T>
__global__ void k_extra_rowmax(SlabView<T> sv, int64_t row0, int B, int V, const int32_t* __restrict__ perm,
                               double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    T m = -std::numeric_limits<T>::infinity();
    for (int v = lane; v < V; v += 64) {
        const T sc = sv.at(row0 + b, v);
        m = sc > m ? sc : m;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_xor(m, off, 64);
        m = o > m ? o : m;
    }
    if (lane == 0) out[perm ? perm[b] : b] = (double)m;
}

__global__ void k_unpermute(int B, int AO, const int32_t* __restrict__ perm, const int32_t* __restrict__ action,
                            const int32_t* __restrict__ best_v, int32_t* __restrict__ action_o, int32_t* __restrict__ best_o) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int d = perm[b];
    if (tid == 0) action_o[d] = action[b];
    for (int j = tid; j < AO; j += blockDim.x) best_o[(int64_t)d * AO + j] = best_v[(int64_t)b * AO + j];
}

// ---- early rows (pbvi_backup_run_fetch) ---------------------------------------------------------------------------- //
// The small results of a run_fetch step -- slot, index and action per belief, the call's counters -- stored straight into the
// caller's page-locked arrays and the engine's pinned flag words by one kernel: five D2H copies of 4 KB or less (one of them
// into pageable stack memory) cost the end of every step ~25 us of copy-engine round trips.
__global__ void k_publish(int B, const int32_t* __restrict__ slot, const int32_t* __restrict__ index, const int32_t* __restrict__ action,
                          int32_t* __restrict__ h_slot, int32_t* __restrict__ h_index, int32_t* __restrict__ h_action,
                          const int* __restrict__ counters /* 8 */, const int* __restrict__ e_cnt /* 3 */,
                          const int* __restrict__ scr_flag /* or nullptr */, int* __restrict__ h_flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < B) {
        h_slot[i] = slot[i];
        h_index[i] = index[i];
        h_action[i] = action[i];
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) h_flag[16 + threadIdx.x] = counters[threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x >= 8 && threadIdx.x < 11) h_flag[4 + threadIdx.x - 8] = e_cnt[threadIdx.x - 8];
    if (blockIdx.x == 0 && threadIdx.x == 11 && scr_flag != nullptr) h_flag[1] = scr_flag[0];
}

// One block per distinct key u of the FINAL result: is it among the provisional keys (whose rows are on their way to the
// host already)?  Yes: slot[u] = its provisional position.  No: the row gets the next free slot behind the provisional
// ones and this block copies it there.  cnt[0] = provisional keys, cnt[1] += new rows, cnt[2] = 1 on overflow of the slots.
template <typename T>
__global__ void k_match_rows(int AO, int O, const int* __restrict__ ucount, const int32_t* __restrict__ uniq, const int32_t* __restrict__ action,
                             const int32_t* __restrict__ best, const int32_t* __restrict__ uniq_p, const int32_t* __restrict__ action_p,
                             const int32_t* __restrict__ best_p, int* __restrict__ cnt, int cap_rows, int32_t* __restrict__ slot,
                             const T* __restrict__ rows_dev, T* __restrict__ rows_host, int64_t cols) {
    __shared__ int found_sh, slot_sh;
    const int n = *ucount;
    const int np = cnt[0];
    for (int u = blockIdx.x; u < n; u += gridDim.x) {
        const int b = uniq[u], a = action[b];
        const int32_t* key = best + ((int64_t)b * AO + (int64_t)a * O);
        __syncthreads();
        if (threadIdx.x == 0) found_sh = 0x7fffffff;
        __syncthreads();
        for (int p = threadIdx.x; p < np; p += blockDim.x) {
            const int bp = uniq_p[p];
            bool eq = action_p[bp] == a;
            const int32_t* kp = best_p + ((int64_t)bp * AO + (int64_t)a * O);
            for (int o = 0; eq && o < O; ++o) eq = kp[o] == key[o];
            if (eq) atomicMin(&found_sh, p);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int sl = found_sh;
            if (sl == 0x7fffffff) {
                sl = np + atomicAdd(&cnt[1], 1);
                if (sl >= cap_rows) {
                    cnt[2] = 1;
                    sl = -1;
                }
                slot_sh = sl;
            } else {
                slot_sh = -2;                                // already on the host
            }
            slot[u] = sl;
        }
        __syncthreads();
        const int sl = slot_sh;
        if (sl >= 0) {
            const T* src = rows_dev + (int64_t)u * cols;
            T* dst = rows_host + (int64_t)sl * cols;
            for (int64_t i = threadIdx.x; i < cols; i += blockDim.x) dst[i] = src[i];
        }
    }
}

// The scoring stage (K1, K2, first-max) runs on an engine or on an fp64 engine's fp32 screen (EngineT::stage_scores).
struct ScoreIO {                      // the pipeline buffers the stage writes / reads: owned by the engine that continues
    int32_t* best_v;
    double* best_score;
    double* err;
    int32_t* queue;                   // near-tie queue (nullptr: no windows, exact arithmetic)
    int* qcount;
    const uint8_t* dead;              // exact dead-triple flags, or nullptr
    const double* prd;                // [B][A] b.ER in f64 for the belief-side formulation, or nullptr
    hipEvent_t join;                  // the side-stream work that produced dead / prd
    hipStream_t side;                 // where the GEMM's tile lists may be built, or nullptr
    hipEvent_t* ev;                   // ev[1] after the projection, ev[2] after the GEMM, ev[3] after the argmax
    bool stats;
    double tol_extra;                 // added to the relative tie window (input rounding of a screen)
};
template <typename TS>
struct ScoreStage {
    SlabView<TS> sv;
    GemmPlan plan;
    double tol_rel;
    const int* chain;
    int64_t rd_col0, extra_row0, f64_pairs;
    bool fused;                       // the GEMM generated its Gamma tiles itself (gemm.hip, scheduler 2b)
};

class EngineBase {
   public:
    virtual ~EngineBase() {}
    virtual int alpha_set(const void* alpha, int64_t V) = 0;
    virtual int alpha_append(const void* alpha, int64_t n) = 0;
    virtual int64_t alpha_count() const = 0;
    virtual int beliefs_set(const void* bel, int64_t B) = 0;
    virtual int backup_run(double gamma, int flags, pbvi_stats_t* st) = 0;
    virtual int backup_fetch(void* out_alpha, int32_t* out_action, int32_t* out_best, uint8_t* out_keep) = 0;
    virtual int device_results(void** d_alpha, int32_t** d_action, uint8_t** d_keep) = 0;
    virtual int64_t unique_count() const = 0;
    virtual int fetch_unique(void* out_rows, int32_t* out_index) = 0;
    virtual int fetch_compact(void* out_rows, int32_t* out_index, int32_t* out_action, int32_t* out_best, uint8_t* out_keep) = 0;
    virtual int run_fetch(double gamma, int flags, pbvi_stats_t* st, void* out_rows, int64_t cap_rows, int32_t* out_slot,
                          int32_t* out_index, int32_t* out_action, int32_t* out_best, uint8_t* out_keep, int64_t* n_unique,
                          int64_t* n_slots) = 0;
    virtual int fetch_unique_keys(int32_t* out_keys) = 0;
    virtual int fetch_row_hashes(uint64_t* out) = 0;
    virtual int fetch_exchange(int32_t* out, int64_t per) = 0;
    virtual int assemble_keys(double gamma, int64_t n, const int32_t* keys, void* out_rows) = 0;
    virtual int64_t assemble_keys_store(double gamma, int64_t n, const int32_t* keys, void* out_rows) = 0;
    virtual int prune_dominated(uint8_t* keep) = 0;
    virtual int value_max(double* out_value, int32_t* out_index) = 0;
    virtual int value_max_store(int64_t n, double* out_value, int32_t* out_index) = 0;
    virtual int64_t store_count(int which) const = 0;
    virtual int set_tie_window(double rel) = 0;
    virtual int set_value_max_exact(int exact) = 0;
    virtual int alpha_layout(int64_t* free_rows, int64_t* layouts) = 0;
    virtual int set_formulation(int f) = 0;
    virtual int set_screen(int mode) = 0;
    virtual int set_fused(int enable) = 0;
    virtual int64_t device_bytes() const = 0;
    virtual int64_t store_append(int which, const void* rows, int64_t n) = 0;
    virtual int64_t store_append_unique(const int32_t* unique_idx, int64_t n) = 0;
    virtual int store_select(int which, const int32_t* ids, int64_t n) = 0;
    virtual int store_reset(int which) = 0;
    virtual int after_oom() = 0;
    virtual int belief_update(const int32_t* act, const int32_t* obs, void* out) = 0;
    virtual int beliefs_advance(const int32_t* act, const int32_t* obs, const uint8_t* keep, int64_t* out_B) = 0;
    virtual int beliefs_fetch(void* out) = 0;
    virtual int set_rto_f64(const double* rto) = 0;
    virtual int64_t belief_walk(const double* b0, int64_t n, const int32_t* act, const int32_t* obs, const uint8_t* restart,
                                double* out) = 0;
    virtual int64_t beliefs_count() const = 0;
    virtual int walk_keys(int64_t n, uint64_t* out_keys) = 0;
    virtual int backup_value_max(double* out_value) = 0;
};

template <typename T>
class EngineT : public EngineBase {
   public:
    static constexpr bool kF32 = sizeof(T) == 4;

    int device_ = 0;
    hipStream_t stream_ = nullptr;
    int S_ = 0, A_ = 0, O_ = 0, R_ = 0, S_pad_ = 0, mode_ = 0;
    int64_t bytes_ = 0;
    // bytes this engine may still allocate: what its cap leaves, or what the device reports free
    int64_t room() const {
        if (DevBuf::limit_bytes() >= 0) return DevBuf::limit_bytes() - bytes_;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            (void)hipGetLastError();
            return INT64_MAX;
        }
        return (int64_t)fr;
    }
    double tie_rel_user_ = -1.0;

    DevBuf rs_, rto_, er_, sup_;
    // The working alpha set.  alpha_ is a VIEW (never allocated or released itself) of V_ data rows, the magnitude row and
    // zero rows up to the next multiple of 256 + 256, inside one of two allocations:
    //   alpha_buf_   -- the set as uploaded (alpha_set / alpha_append: view at row 0), or the PRIMARY set selected from the
    //                   alpha store, laid out with free rows in FRONT of it: the solve loop's next set is its new rows followed
    //                   by the previous set (ValueFunction.extend: new-then-old, and first-max ties go to the lower index, so
    //                   the order is part of the result) -- those rows are gathered into the free rows just before the
    //                   view, which then starts k rows earlier; nothing else moves, the magnitude row takes the maximum
    //                   with the new rows.  (Re-gathering 10^4 rows per backup was 0.1 s of a 300-expansion solve, the fp32
    //                   copy for the screen of an fp64 engine another 0.09 s.)
    //   alpha_small_ -- a selection much smaller than the primary set (compute_change scores the rows an expansion added):
    //                   gathered here so that the primary set is still there for the next backup.
    DevBuf alpha_, alpha_buf_, alpha_small_;
    std::vector<int32_t> prim_ids_;                          // store ids of the primary set, view order
    int64_t prim_off_ = 0;                                   // first data row of the primary set inside alpha_buf_
    bool prim_valid_ = false, alpha_on_primary_ = false;
    uint64_t prim_layout_ver_ = 0;                           // a fresh layout of the primary set (the screen mirrors it)
    uint64_t screen_layout_seen_ = 0;                        // fp64 engines: layout / first row the screen's copy reflects
    int64_t screen_off_seen_ = 0;
    int64_t V_ = 0;
    DevBuf bel_;
    int64_t B_ = 0, B_pad_ = 0;
    DevBuf gam_, slabs_, best_v_, best_score_, err_, dead_, queue_, counters_, rdot_, action_, aqueue_, out_, keep_;
    DevBuf bv2_, bs2_, err2_, queue2_, prune_cnt_;
    DevBuf stage_, keys_, perm_, action_res_, best_res_;   // belief reordering (f32, B > 256)
    DevBuf rep_, uniq_, inv_, slot_, out_full_;            // K6 key dedup: out_ holds the unique rows
    DevBuf store_[2], ids_;                                // device row stores: [0] alpha-vectors, [1] beliefs
    int64_t store_rows_[2] = {0, 0};
    DevBuf in_ptr_, in_src_, bu_act_, bu_obs_, bu_row_, bu_unnorm_, bu_mass_, bu_out_, walk64_, rto64_;   // batched belief update
    std::vector<int32_t> h_rs_;                                // host copy of rs [A][R][S_pad] for the lazy CSC build
    DevBuf btl_, btc_, val_exact_;                         // per-belief non-zero tile lists; exact action values
    DevBuf bp_, nzP_, pmag_, prd_;                          // belief-side formulation: projected beliefs, tile map, magnitudes, b.ER
    bool btl_valid_ = false;                                // btl_/btc_ describe the resident belief block
    bool vmax_exact_ = true;                                // f32 engines: re-score value_max's maxima in fp64
    // belief store as a GEMM operand in place (value_max_store): zero maps and tile lists kept per store row, extended
    // as rows are appended
    DevBuf snz_, sbtl_, sbtc_;
    int64_t snz_rows_ = 0, sbt_rows_ = 0;
    int64_t walk_rows_ = 0;
    DevBuf vmax_bk_;                                        // max_v b.alpha_v of the last backup's beliefs (belief-side GEMM's extra rows)
    bool have_bk_vmax_ = false;
    hipEvent_t walk_ev_[8] = {};                            // belief_walk: chain -> copy -> host hand-offs, per quarter                                 // rows the last belief_walk left in walk64_
    void* host_stage_ = nullptr;                            // pinned bounce buffer for rows going to pageable host memory
    size_t host_stage_cap_ = 0;
    DevBuf keys_tmp_, keys_act_, keys_best_, keys_rows_;    // unique-row keys out / rows from keys in (multi-GPU exchange)
    DevBuf acand_;                                           // [B][A] action inside the window (k_action_select -> k_refine_action)
    DevBuf rf_q2_, rf_q2p_, rf_q2d_;                         // k_refine_split: entries / candidates, partial scores, arrival counters
    DevBuf rf_v_, rf_slot_, rf_sc_, rf_entry_, rf_n_, rf_tiles_, rf_ibv_, rf_ibi_, rf_cnt_, rf_W_, rf_Cx_, rf_nzW_, rf_klW_, rf_kcW_;   // refinement work list
    int formulation_ = 0;                                   // 0 auto, 1 project alpha-vectors, 2 project beliefs
    int last_formulation_ = 1;
    int64_t f64_pairs_ = 0;                                 // tile pairs of the last fp64 MFMA GEMM (0: plain kernel)
    const int32_t* res_action_ = nullptr;                  // results in caller order
    const int32_t* res_best_ = nullptr;
    int64_t res_unique_ = 0;
    bool full_valid_ = false;
    std::vector<int32_t> h_perm_;
    bool h_perm_valid_ = false;                              // h_perm_ mirrors perm_ (fetched on demand, see host_perm)
    bool sorted_ = false;
    const void* gam_pad_ptr_ = nullptr;                      // Gamma buffer / row count whose pad rows are zero
    int64_t gam_pad_N_ = -1;
    DevBuf scr_flag_;                                        // fp64 engines: 1 = the screen's fp32 alpha copy holds an overflowed value
    DevBuf rowflags_;                                        // [B][k_tiles] 1 = caller's belief row b has mass in K tile kt
    uint64_t rowflags_ver_ = 0;                              // belief-block version rowflags_ describes
    uint64_t nzA_ver_ = 0;                                   // belief-block version ev_nzA_ was recorded for
    void* ids_pin_ = nullptr;                                // pinned staging of the ids of a store selection
    size_t ids_pin_cap_ = 0;
    int* h_flag_ = nullptr;                                  // pinned: see pinned_flag()
    // early rows (run_fetch): destination of the current call, buffers of the provisional decision pipeline
    void* early_rows_ = nullptr;
    int64_t early_cap_ = 0;
    bool early_used_ = false, early_dma_pending_ = false;
    struct RunFetchDst {                                     // run_fetch: the small per-belief results travel with the pipeline's
        int32_t *slot = nullptr, *index = nullptr, *action = nullptr, *best = nullptr;   // last read-back, not behind a second
        uint8_t* keep = nullptr;                                                         // synchronisation
        bool queued = false;
    } rf_dst_;
    DevBuf e_bv_, e_bs_, e_err_, e_rdot_, e_act_, e_ares_, e_bres_, e_rep_, e_uniq_, e_inv_, e_slotd_, e_cnt_, e_out_, e_slot_;
    hipEvent_t ev_early_[2] = {nullptr, nullptr};
    hipEvent_t ev_ids_ = nullptr;
    hipEvent_t ev_nzA_ = nullptr;                            // the resident block's zero map (nzA_) is complete
    DevBuf nzBw_;                                            // [A*O][ceil(k_tiles/64)] support tiles of RTO as bit words
    DevBuf nzB_, nzA_, klist_, kcount_, nchunks_, need_, skws_;   // zero-tile bookkeeping of the f32 score GEMM
    DevBuf dense_, nzD_, nzAlpha_, prod_, klistD_, kcountD_, nchunksD_;   // dense projection mode
    int64_t rows_pad_s_ = 0, dense_pairs_ = 0;
    std::vector<int> h_kcountD_;
    int* kc_pin_ = nullptr;                                   // page-locked landing area of the score GEMM's list lengths (statistics)
    size_t kc_pin_cap_ = 0;
    hipStream_t stream2_ = nullptr;                  // belief-only kernels run beside projection + GEMM
    hipStream_t stream3_ = nullptr;                  // the score GEMM's tile lists / stream-K plan: beside both (a few us of
                                                     // work that the GEMM waits for must not queue behind k_dead's 100 us)
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr, ev_lists_ = nullptr, ev_xr_[2] = {nullptr, nullptr};
    GemmPlan plan_ = {};
    hipEvent_t ev_[9] = {};
    hipEvent_t ev_pg_[2] = {};                       // around the dense projection's GEMM kernel
    bool have_result_ = false, res_sorted_ = false;
    int64_t res_B_ = 0;
    int screen_mode_ = 1;                                    // fp64 engines: 0 never screen, 1 when the GEMM is large, 2 always
    int fuse_project_ = 1;                                   // fp32: Gamma tiles generated inside the score GEMM; 0 never, 1 = where it
                                                             // is faster (R = 1), 2 = also for R = 2..7 (measured slower)
    double irr_frac_ = 0.0;                                  // share of (action, K tile) with non-consecutive successors
    DevBuf irr_;                                             // [A][k_tiles] 1 = a 4-state chunk of the K tile has non-consecutive successors
    DevBuf mat_, vlist_;                                     // [n-tiles] 1 = projected (straddles groups / tail), 0 = generated
    DevBuf ctile_;                                           // [n-tiles] compact tile index of a projected tile (-1: generated)
    std::vector<int32_t> h_ctile_;
    int n_mat_ = 0;                                          // projected tiles = tiles of Gamma that exist in memory when fused
    bool gam_pad_compact_ = false;
    std::vector<uint8_t> h_mat_;
    std::vector<int> h_vlist_;
    int64_t mat_V_ = -1;
    uint64_t dead_ver_ = 0;                                  // belief-block version dead_ / btl_ / btc_ describe
    int64_t dead_pairs_ = 0, dead_count_ = 0;
    int* rf_counts_ = nullptr;                               // [2] in host_stage_: what the refinement's per-entry pass deferred
    bool last_deferred_nothing_ = false;                     // (the first backup of an engine reads the counts mid-pipeline)
    bool owns_streams_ = true;                               // false: a screen running on its fp64 engine's streams
    EngineT<float>* screen_ = nullptr;                       // fp64 engines: the fp32 screen (see ensure_screen)
    std::vector<int32_t> h_reach_ref_;                       // fp64 engines: the tables in the reference's layout, kept
    std::vector<double> h_rto_ref_, h_er_ref_;               //   to build the screen on first use
    uint64_t alpha_ver_ = 1, bel_ver_ = 1;                   // bumped whenever alpha_ / bel_ change
    uint64_t screen_alpha_seen_ = 0, screen_bel_seen_ = 0;

    ~EngineT() override {
        (void)hipSetDevice(device_);
        if (screen_) {
            (void)hipStreamSynchronize(stream_);
            delete screen_;
            screen_ = nullptr;
        }
        DevBuf* all[] = {&rs_, &rto_, &er_, &sup_, &alpha_buf_, &alpha_small_, &bel_, &gam_, &slabs_, &best_v_, &best_score_, &err_,
                         &dead_, &queue_, &counters_, &rdot_, &action_, &aqueue_, &out_, &keep_, &bv2_, &bs2_,
                         &err2_, &queue2_, &prune_cnt_, &nzB_, &nzA_, &klist_, &kcount_, &nchunks_, &need_, &skws_, &stage_, &keys_, &perm_,
                         &action_res_, &best_res_, &rep_, &uniq_, &inv_, &slot_, &out_full_, &btl_, &btc_, &val_exact_, &store_[0], &store_[1], &ids_, &in_ptr_, &in_src_, &bu_act_, &bu_obs_,
                         &bu_unnorm_, &bu_mass_, &bu_out_, &bu_row_, &walk64_, &rto64_, &bp_, &nzP_, &pmag_, &prd_, &keys_tmp_, &keys_act_, &keys_best_, &keys_rows_, &rf_v_, &rf_slot_, &rf_sc_, &rf_entry_, &rf_n_, &rf_tiles_,
                         &snz_, &sbtl_, &sbtc_, &vmax_bk_, &rf_ibv_, &rf_ibi_, &rf_cnt_, &rf_W_, &rf_Cx_, &rf_nzW_, &rf_klW_, &rf_kcW_,
                         &dense_, &nzD_, &nzAlpha_, &prod_, &klistD_, &kcountD_, &nchunksD_, &mat_, &vlist_, &irr_, &rowflags_, &nzBw_, &scr_flag_, &ctile_, &acand_, &rf_q2_, &rf_q2p_, &rf_q2d_, &e_bv_, &e_bs_, &e_err_, &e_rdot_, &e_act_, &e_ares_, &e_bres_, &e_rep_, &e_uniq_, &e_inv_, &e_slotd_, &e_cnt_, &e_out_, &e_slot_};
        // every call is checked only to name a failure when PBVI_DEBUG is set; the thread's sticky last-error is cleared at
        // the end either way, so that a later launch check does not report a stale error of this teardown
        static const bool dbg = getenv("PBVI_DEBUG") != nullptr;
        auto chk = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && dbg) fprintf(stderr, "[pbvi] engine teardown: %s: %s\n", what, hipGetErrorString(e));
        };
        for (auto& e : walk_ev_)
            if (e) chk(hipEventDestroy(e), "hipEventDestroy(walk)");
        for (DevBuf* b : all) b->release();
        if (host_stage_) chk(hipHostFree(host_stage_), "hipHostFree(stage)");
        if (ids_pin_) chk(hipHostFree(ids_pin_), "hipHostFree(ids)");
        if (h_flag_) chk(hipHostFree(h_flag_), "hipHostFree(flag)");
        if (kc_pin_) chk(hipHostFree(kc_pin_), "hipHostFree(kcount)");
        for (hipEvent_t& ev : ev_early_)
            if (ev) chk(hipEventDestroy(ev), "hipEventDestroy(early)");
        if (ev_ids_) chk(hipEventDestroy(ev_ids_), "hipEventDestroy(ids)");
        if (ev_nzA_) chk(hipEventDestroy(ev_nzA_), "hipEventDestroy(nzA)");
        for (auto& e : ev_)
            if (e) chk(hipEventDestroy(e), "hipEventDestroy(ev)");
        for (auto& e : ev_pg_)
            if (e) chk(hipEventDestroy(e), "hipEventDestroy(ev_pg)");
        if (ev_fork_) chk(hipEventDestroy(ev_fork_), "hipEventDestroy(fork)");
        if (ev_join_) chk(hipEventDestroy(ev_join_), "hipEventDestroy(join)");
        if (ev_lists_) chk(hipEventDestroy(ev_lists_), "hipEventDestroy(lists)");
        for (auto& e : ev_xr_)
            if (e) chk(hipEventDestroy(e), "hipEventDestroy(xr)");
        if (owns_streams_) {
            if (stream3_) chk(hipStreamDestroy(stream3_), "hipStreamDestroy(3)");
            if (stream2_) chk(hipStreamDestroy(stream2_), "hipStreamDestroy(2)");
            if (stream_) chk(hipStreamDestroy(stream_), "hipStreamDestroy(1)");
        }
        chk(hipGetLastError(), "sticky error left by an earlier call");
    }

    ModelView<T> view() const {
        ModelView<T> mv;
        mv.S = S_;
        mv.S_pad = S_pad_;
        mv.A = A_;
        mv.O = O_;
        mv.R = R_;
        mv.rs = rs_.as<int32_t>();
        mv.rto = rto_.as<T>();
        mv.er = er_.as<T>();
        mv.sup = sup_.as<uint8_t>();
        return mv;
    }

    int init(int device, int S, int A, int O, int R, const int32_t* reach, const T* rto, const T* er, int mode,
             hipStream_t shared_main = nullptr, hipStream_t shared_side = nullptr, hipStream_t shared_lists = nullptr) {
        device_ = device;
        S_ = S;
        A_ = A;
        O_ = O;
        R_ = R;
        mode_ = mode;
        S_pad_ = (int)round_up(S, GEMM_BK);
        if (const char* f = getenv("PBVI_F64_SCREEN")) {      // initial setting (tests run the whole suite with the screen forced)
            const std::string v(f);
            screen_mode_ = (v == "off" || v == "0") ? 0 : (v == "always" || v == "2") ? 2 : 1;
        }
        if (const char* f = getenv("PBVI_FORMULATION")) {     // initial setting (tests run the whole suite both ways)
            const std::string v(f);
            formulation_ = (v == "alpha" || v == "1") ? 1 : (v == "belief" || v == "2") ? 2 : 0;
        }
        HIPCHK(hipSetDevice(device_));
        if (shared_main != nullptr) {   // an fp32 screen lives on its fp64 engine's streams: one pipeline, one order
            stream_ = shared_main;
            stream2_ = shared_side;
            stream3_ = shared_lists;
            owns_streams_ = false;
        } else {
        HIPCHK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        }
        if (owns_streams_) {   // The side stream gets its own priority class.  Streams of one class share a small pool of hardware
            // queues, assigned round-robin at creation: once a RCCL communicator has created its streams, two
            // normal-priority streams made afterwards can land on one queue and the side work no longer overlaps
            // the projection (measured: ms_project 0.35 -> 0.44 with an idle communicator in the process).
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            static const bool plain_side = getenv("PBVI_SIDE_STREAM_NORMAL") != nullptr;      // debug / A-B only
            if (hi < lo && !plain_side) {
                HIPCHK(hipStreamCreateWithPriority(&stream2_, hipStreamNonBlocking, hi));
                HIPCHK(hipStreamCreateWithPriority(&stream3_, hipStreamNonBlocking, hi));
            } else {
                HIPCHK(hipStreamCreateWithFlags(&stream2_, hipStreamNonBlocking));
                HIPCHK(hipStreamCreateWithFlags(&stream3_, hipStreamNonBlocking));
            }
        }
        for (auto& e : ev_) HIPCHK(hipEventCreate(&e));
        for (auto& e : ev_pg_) HIPCHK(hipEventCreate(&e));
        HIPCHK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ev_lists_, hipEventDisableTiming));
        for (auto& e : ev_xr_) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));

        // re-tile the reference's [S][A][R] / [S][A][O][R] / [S][A] tables to s-contiguous planes
        const size_t n_rs = (size_t)A * R * S_pad_, n_rto = (size_t)A * O * R * S_pad_, n_er = (size_t)A * S_pad_;
        std::vector<int32_t> h_rs;
        std::vector<T> h_rto, h_er;
        try {
            h_rs.assign(n_rs, 0);
            h_rto.assign(n_rto, T(0));
            h_er.assign(n_er, T(0));
        } catch (const std::bad_alloc&) {
            FAIL(PBVI_ENOMEM, "host allocation for table re-tiling failed");
        }
        for (int s = 0; s < S; ++s)
            for (int a = 0; a < A; ++a) {
                h_er[(size_t)a * S_pad_ + s] = er[(size_t)s * A + a];
                for (int r = 0; r < R; ++r) {
                    const int32_t t = reach[((size_t)s * A + a) * R + r];
                    if (t < 0 || t >= S) FAIL(PBVI_EINVAL, "reach_states entry out of range [0,S)");
                    h_rs[((size_t)a * R + r) * S_pad_ + s] = t;
                    for (int o = 0; o < O; ++o)
                        h_rto[(((size_t)a * O + o) * R + r) * S_pad_ + s] = rto[(((size_t)s * A + a) * O + o) * R + r];
                }
            }
        h_rs_ = h_rs;                                        // (the CSC build keeps the caller's successors)
        // Slots of weight zero for every observation -- the reference pads an (s, a) with fewer than R successors by
        // low-index states at probability 0 (src/mdp.py:308-335), and the pad states s >= S are such slots too --
        // contribute exactly 0 whichever state they name.  On the device they name the state that keeps their 4-state
        // chunk's successors consecutive, so that padded models keep the 16-byte alpha loads of the projection and the
        // fused score GEMM (gemm.hip) instead of falling to gathers.
        if (S >= 4)
            for (int a = 0; a < A; ++a)
                for (int r = 0; r < R; ++r) {
                    int32_t* row = h_rs.data() + ((size_t)a * R + r) * S_pad_;
                    for (int c = 0; c < S_pad_ / 4; ++c) {
                        bool wild[4];
                        int j0 = -1, n_wild = 0;
                        for (int j = 0; j < 4; ++j) {
                            const int s = c * 4 + j;
                            bool w = true;
                            for (int o = 0; o < O && w; ++o) w = h_rto[(((size_t)a * O + o) * R + r) * S_pad_ + s] == T(0);
                            wild[j] = w;
                            n_wild += w;
                            if (!w && j0 < 0) j0 = j;
                        }
                        if (n_wild == 0) continue;
                        const int64_t base = j0 >= 0 ? (int64_t)row[c * 4 + j0] - j0 : 0;
                        if (base < 0 || base + 3 >= S) continue;
                        bool ok = true;
                        for (int j = 0; j < 4 && ok; ++j) ok = wild[j] || row[c * 4 + j] == base + j;
                        if (!ok) continue;
                        for (int j = 0; j < 4; ++j)
                            if (wild[j]) row[c * 4 + j] = (int32_t)(base + j);
                    }
                }
        if constexpr (!kF32) {   // reference-layout copies for the fp32 screen (built on the first large backup)
            if (mode == PBVI_SPARSE && (size_t)S * A * O * R <= ((size_t)1 << 28)) {
                try {
                    h_reach_ref_.assign(reach, reach + (size_t)S * A * R);
                    h_rto_ref_.assign(rto, rto + (size_t)S * A * O * R);
                    h_er_ref_.assign(er, er + (size_t)S * A);
                } catch (const std::bad_alloc&) {
                    h_reach_ref_.clear();
                    h_rto_ref_.clear();
                    h_er_ref_.clear();
                }
            }
        }
        int rc;
        if ((rc = rs_.ensure(n_rs * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = rto_.ensure(n_rto * sizeof(T), &bytes_))) return rc;
        if ((rc = er_.ensure(n_er * sizeof(T), &bytes_))) return rc;
        if ((rc = sup_.ensure((size_t)A * O * S_pad_, &bytes_))) return rc;
        HIPCHK(hipMemcpyAsync(rs_.p, h_rs.data(), n_rs * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(rto_.p, h_rto.data(), n_rto * sizeof(T), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(er_.p, h_er.data(), n_er * sizeof(T), hipMemcpyHostToDevice, stream_));
        HIPCHK(launch_support<T>(view(), sup_.as<uint8_t>(), stream_));
        if ((rc = counters_.ensure(8 * sizeof(int), &bytes_))) return rc;
        {   // nzB[(a,o)][kt] = 1 iff RTO[:,a,o,:] has support inside K tile kt (32 states)
            const int k_tiles = S_pad_ / GEMM_BK;
            // ... plus one extra row [A*O] for the support of ER (the reward rows in Gamma's tail tile)
            std::vector<uint8_t> h_nz((size_t)(A * O + 1) * k_tiles, 0);
            for (int ao = 0; ao < A * O; ++ao)
                for (int r = 0; r < R; ++r)
                    for (int s = 0; s < S; ++s)
                        if (h_rto[((size_t)ao * R + r) * S_pad_ + s] != T(0)) h_nz[(size_t)ao * k_tiles + s / GEMM_BK] = 1;
            for (int a = 0; a < A; ++a)
                for (int s = 0; s < S; ++s)
                    if (h_er[(size_t)a * S_pad_ + s] != T(0)) h_nz[(size_t)A * O * k_tiles + s / GEMM_BK] = 1;
            if ((rc = nzB_.ensure(h_nz.size(), &bytes_))) return rc;
            HIPCHK(hipMemcpyAsync(nzB_.p, h_nz.data(), h_nz.size(), hipMemcpyHostToDevice, stream_));
            // the same supports as bit words (bit t of word w = K tile 64 w + t), for the dead-triple test
            const int kw = (k_tiles + 63) / 64;
            std::vector<unsigned long long> h_nzw((size_t)A * O * kw, 0ull);
            for (int ao = 0; ao < A * O; ++ao)
                for (int kt = 0; kt < k_tiles; ++kt)
                    if (h_nz[(size_t)ao * k_tiles + kt]) h_nzw[(size_t)ao * kw + kt / 64] |= 1ull << (kt & 63);
            if ((rc = nzBw_.ensure(h_nzw.size() * sizeof(unsigned long long), &bytes_))) return rc;
            HIPCHK(hipMemcpyAsync(nzBw_.p, h_nzw.data(), h_nzw.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream_));
            std::vector<int32_t> h_irr;
            // For the fused score GEMM (gemm.hip, schedulers 2b / 2c): K tiles that hold a 4-state chunk whose successors
            // (for some r) are not 4 consecutive states.  R = 1: the kernel gathers those itself; R = 2..7: those Gamma tiles
            // are projected and read, so fusing only pays while they are few (grid edges: ~1 in 8 on the olfactory grids;
            // on a model without grid structure every tile is one and the projection kernel stays).
            if (kF32 && R <= 7 && mode == PBVI_SPARSE) {
                h_irr.assign((size_t)A * k_tiles, 0);
                size_t n_irr = 0;
                for (int a = 0; a < A; ++a)
                    for (int r = 0; r < R; ++r)
                        for (int c = 0; c < S_pad_ / 4; ++c) {
                            const int32_t* q = h_rs.data() + ((size_t)a * R + r) * S_pad_ + (size_t)c * 4;
                            int32_t& f = h_irr[(size_t)a * k_tiles + c / 8];
                            if (!f && (q[1] != q[0] + 1 || q[2] != q[0] + 2 || q[3] != q[0] + 3)) {
                                f = 1;
                                ++n_irr;
                            }
                        }
                const double max_irr = getenv("PBVI_FUSE_MAX_IRR") ? atof(getenv("PBVI_FUSE_MAX_IRR")) : 0.30;
                irr_frac_ = (double)n_irr / (double)h_irr.size();
                if (R == 1 || irr_frac_ <= max_irr) {
                    if ((rc = irr_.ensure(h_irr.size() * sizeof(int32_t), &bytes_))) return rc;
                    HIPCHK(hipMemcpyAsync(irr_.p, h_irr.data(), h_irr.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
                }
            }
            HIPCHK(hipStreamSynchronize(stream_));
        }
        if (mode_ == PBVI_DENSE) {   // D[ao] = dense |S| x |S| transition-observation matrices
            rows_pad_s_ = round_up(S, GEMM_BN);
            const size_t n = (size_t)A * O * rows_pad_s_ * S_pad_;
            if ((rc = dense_.ensure(n * sizeof(T), &bytes_))) return rc;
            HIPCHK(hipMemsetAsync(dense_.p, 0, n * sizeof(T), stream_));
            hipLaunchKernelGGL(k_densify<T>, dim3((S + 255) / 256, A * O), dim3(256), 0, stream_, view(), dense_.as<T>(),
                               rows_pad_s_);
            HIPCHK(hipGetLastError());
            if (kF32) {
                const int k_tiles = S_pad_ / GEMM_BK;
                const size_t per = (size_t)(rows_pad_s_ / GEMM_BN) * k_tiles;
                if ((rc = nzD_.ensure((size_t)A * O * per, &bytes_))) return rc;
                for (int ao = 0; ao < A * O; ++ao)
                    HIPCHK(launch_tile_nonzero_f32((const float*)dense_.p + (size_t)ao * rows_pad_s_ * S_pad_, S_pad_,
                                                   (int)rows_pad_s_, k_tiles, nzD_.as<uint8_t>() + ao * per, stream_));
            }
            HIPCHK(hipStreamSynchronize(stream_));
        }
        HIPCHK(hipStreamSynchronize(stream_));
        return PBVI_OK;
    }

    // ---- alpha set ------------------------------------------------------- //
    size_t alpha_rows_cap(int64_t V) const { return (size_t)round_up(V + 1, GEMM_BN); }

    void alpha_view(const DevBuf& own, int64_t row_off) {
        const size_t skip = (size_t)row_off * S_pad_ * sizeof(T);
        alpha_.p = static_cast<char*>(own.p) + skip;
        alpha_.cap = own.cap - skip;
    }

    // magnitude row (row V_ of the view) := column-wise max |alpha| of the view's rows
    int refresh_magnitude_row() {
        T* base = alpha_.as<T>();
        HIPCHK(hipMemsetAsync(base + (size_t)V_ * S_pad_, 0, (size_t)S_pad_ * sizeof(T), stream_));
        return merge_magnitude_row(base, V_);
    }
    // ... := max(itself, column-wise max |.| of `n` rows at `rows`)
    int merge_magnitude_row(const T* rows, int64_t n) {
        if (n <= 0) return PBVI_OK;
        hipLaunchKernelGGL(k_col_absmax<T>, dim3((S_pad_ + 255) / 256, (unsigned)((n + 127) / 128)), dim3(256), 0, stream_,
                           rows, S_pad_, (int)n, S_pad_, alpha_.as<T>() + (size_t)V_ * S_pad_);
        HIPCHK(hipGetLastError());
        return PBVI_OK;
    }

    int alpha_set(const void* alpha, int64_t V) override {
        if (V <= 0 || alpha == nullptr) FAIL(PBVI_EINVAL, "alpha_set: need V > 0 and a non-null array");
        HIPCHK(hipSetDevice(device_));
        const size_t rows = alpha_rows_cap(V);
        int rc = alpha_buf_.ensure(rows * S_pad_ * sizeof(T), &bytes_);
        if (rc) return rc;
        alpha_view(alpha_buf_, 0);
        prim_valid_ = alpha_on_primary_ = false;
        HIPCHK(hipMemsetAsync(alpha_.p, 0, alpha_.cap, stream_));
        HIPCHK(hipMemcpy2DAsync(alpha_.p, (size_t)S_pad_ * sizeof(T), alpha, (size_t)S_ * sizeof(T),
                                (size_t)S_ * sizeof(T), (size_t)V, hipMemcpyHostToDevice, stream_));
        V_ = V;
        ++alpha_ver_;
        rc = refresh_magnitude_row();
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(stream_));
        have_result_ = false;
        return PBVI_OK;
    }

    int alpha_append(const void* alpha, int64_t n) override {
        if (n < 0 || (n > 0 && alpha == nullptr)) FAIL(PBVI_EINVAL, "alpha_append: bad arguments");
        if (n == 0) return PBVI_OK;
        HIPCHK(hipSetDevice(device_));
        const int64_t Vn = V_ + n;
        const size_t need = alpha_rows_cap(Vn) * S_pad_ * sizeof(T);
        const bool plain = alpha_.p == alpha_buf_.p;          // the view starts the allocation (no rows kept free in front)
        if (need > alpha_.cap || !plain) {   // grow (or move to a plain layout), preserving the resident rows
            DevBuf nb;
            int rc = nb.ensure(std::max(need, plain ? alpha_buf_.cap * 2 : need), &bytes_);
            if (rc) return rc;
            HIPCHK(hipMemsetAsync(nb.p, 0, nb.cap, stream_));
            if (V_ > 0)
                HIPCHK(hipMemcpyAsync(nb.p, alpha_.p, (size_t)V_ * S_pad_ * sizeof(T), hipMemcpyDeviceToDevice, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            bytes_ -= (int64_t)alpha_buf_.cap;
            alpha_buf_.release();
            alpha_buf_ = nb;
            alpha_view(alpha_buf_, 0);
        } else {
            // clear the old magnitude row's pad columns / content before it becomes a data row
            HIPCHK(hipMemsetAsync(alpha_.as<T>() + (size_t)V_ * S_pad_, 0, (size_t)S_pad_ * sizeof(T), stream_));
        }
        prim_valid_ = alpha_on_primary_ = false;
        HIPCHK(hipMemcpy2DAsync(alpha_.as<T>() + (size_t)V_ * S_pad_, (size_t)S_pad_ * sizeof(T), alpha,
                                (size_t)S_ * sizeof(T), (size_t)S_ * sizeof(T), (size_t)n, hipMemcpyHostToDevice, stream_));
        V_ = Vn;
        ++alpha_ver_;
        int rc = refresh_magnitude_row();
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(stream_));
        have_result_ = false;
        return PBVI_OK;
    }

    int64_t alpha_count() const override { return V_; }

    // Belief block from rows on the device -- `src` [.][S_pad], row b of the block = src row (src_ids ? src_ids[b] : b) --
    // reordered, padded to the GEMM's 256-row blocks, with its zero-tile map.  Stream-ordered, no host synchronisation:
    // the host's copy of the order (h_perm_) is fetched when a host-side consumer asks for it (host_perm).
    int beliefs_finish(int64_t B, const T* src, const int32_t* src_ids) {
        const int64_t Bp = round_up(B, GEMM_BM);
        const int k_tiles = S_pad_ / GEMM_BK;
        int rc = bel_.ensure((size_t)Bp * S_pad_ * sizeof(T), &bytes_);
        if (rc) return rc;
        if ((rc = rowflags_.ensure((size_t)B * k_tiles, &bytes_))) return rc;
        if ((rc = keys_.ensure((size_t)B * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = perm_.ensure((size_t)B * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = nzA_.ensure((size_t)(Bp / GEMM_BM) * k_tiles, &bytes_))) return rc;
        if (Bp > B) HIPCHK(hipMemsetAsync(bel_.as<T>() + (size_t)B * S_pad_, 0, (size_t)(Bp - B) * S_pad_ * sizeof(T), stream_));
        static const bool no_sort = getenv("PBVI_NO_BELIEF_SORT") != nullptr;     // debug / A-B only
        sorted_ = B > (kF32 ? GEMM_BM : 128) && !no_sort;     // more than one row block of the score GEMM
        hipLaunchKernelGGL(k_row_flags<T>, dim3((unsigned)B), dim3(256), 0, stream_, src, S_pad_, src_ids, S_pad_, k_tiles,
                           rowflags_.as<uint8_t>(), keys_.as<int32_t>());
        HIPCHK(hipGetLastError());
        const int32_t* perm = nullptr;
        if (sorted_) {
            hipLaunchKernelGGL(k_rank_sort, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, stream_, keys_.as<int32_t>(), (int)B,
                               perm_.as<int32_t>());
            HIPCHK(hipGetLastError());
            perm = perm_.as<int32_t>();
        }
        // the zero map (the first rows of the grid) and the rows in that order, one launch
        constexpr int NS = 16 / (int)sizeof(T);
        const int or_rows = (int)(Bp / GEMM_BM);
        hipLaunchKernelGGL(k_gather_rows_v<T>, dim3((S_pad_ / NS + 255) / 256, (unsigned)(B + or_rows)), dim3(256), 0, stream_, src,
                           bel_.as<T>(), S_pad_, perm, src_ids, rowflags_.as<uint8_t>(), (int)B, k_tiles, nzA_.as<uint8_t>(), or_rows);
        HIPCHK(hipGetLastError());
        if (!ev_nzA_) HIPCHK(hipEventCreateWithFlags(&ev_nzA_, hipEventDisableTiming));
        HIPCHK(hipEventRecord(ev_nzA_, stream_));
        B_ = B;
        B_pad_ = Bp;
        ++bel_ver_;
        rowflags_ver_ = bel_ver_;
        nzA_ver_ = bel_ver_;
        h_perm_valid_ = false;
        have_result_ = false;
        btl_valid_ = false;
        return PBVI_OK;
    }

    // one page-locked int for flags read back with the results (a stack variable would be written by the copy after an
    // early error return had released it)
    int* pinned_flag() {
        if (!h_flag_ && hipHostMalloc((void**)&h_flag_, 128, hipHostMallocDefault) != hipSuccess) {   // 32 ints: [16..23] = run_fetch's counters
            (void)hipGetLastError();
            h_flag_ = nullptr;
        }
        return h_flag_;
    }

    // h_perm_[i] = caller's index of engine row i (sorted blocks), for the host-side consumers of the order
    int host_perm() {
        if (!sorted_ || h_perm_valid_) return PBVI_OK;
        h_perm_.resize((size_t)B_);
        HIPCHK(hipMemcpyAsync(h_perm_.data(), perm_.p, (size_t)B_ * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        h_perm_valid_ = true;
        return PBVI_OK;
    }

    int beliefs_set(const void* bel, int64_t B) override {
        if (B <= 0 || bel == nullptr) FAIL(PBVI_EINVAL, "beliefs_set: need B > 0 and a non-null array");
        if (B > 65535) FAIL(PBVI_EUNSUPPORTED, "beliefs_set: at most 65535 beliefs per block");
        HIPCHK(hipSetDevice(device_));
        int rc = stage_.ensure((size_t)B * S_pad_ * sizeof(T), &bytes_);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(stage_.p, 0, (size_t)B * S_pad_ * sizeof(T), stream_));
        HIPCHK(hipMemcpy2DAsync(stage_.p, (size_t)S_pad_ * sizeof(T), bel, (size_t)S_ * sizeof(T), (size_t)S_ * sizeof(T),
                                (size_t)B, hipMemcpyHostToDevice, stream_));
        if ((rc = beliefs_finish(B, stage_.as<T>(), nullptr))) return rc;
        HIPCHK(hipStreamSynchronize(stream_));               // the caller's array is free again
        return PBVI_OK;
    }

    // ---- batched belief update ---------------------------------------------- //
    int build_inverse_lists() {
        if (in_ptr_.p) return PBVI_OK;
        // CSC of the padded-ELL transition structure per action: for every landing state s' the (s, r) pairs with
        // rs[s,a,r] == s', in ascending s*R+r order (the order np.bincount accumulates in, src/pomdp.py:406).
        std::vector<int32_t> ptr((size_t)A_ * (S_ + 1), 0), src((size_t)A_ * S_ * R_);
        for (int a = 0; a < A_; ++a) {
            int32_t* p = ptr.data() + (size_t)a * (S_ + 1);
            for (int s = 0; s < S_; ++s)
                for (int r = 0; r < R_; ++r) ++p[h_rs_[((size_t)a * R_ + r) * S_pad_ + s] + 1];
            for (int s = 0; s < S_; ++s) p[s + 1] += p[s];
            std::vector<int32_t> fill(p, p + S_);
            int32_t* q = src.data() + (size_t)a * S_ * R_;
            for (int s = 0; s < S_; ++s)
                for (int r = 0; r < R_; ++r) q[fill[h_rs_[((size_t)a * R_ + r) * S_pad_ + s]]++] = s * R_ + r;
        }
        int rc;
        if ((rc = in_ptr_.ensure(ptr.size() * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = in_src_.ensure(src.size() * sizeof(int32_t), &bytes_))) return rc;
        HIPCHK(hipMemcpy(in_ptr_.p, ptr.data(), ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(in_src_.p, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        return PBVI_OK;
    }

    // out[b] = Bayes update of the resident belief b with (act[b], obs[b]); out: [B][S] T, host or device
    int belief_update(const int32_t* act, const int32_t* obs, void* out) override {
        if (B_ <= 0) FAIL(PBVI_EINVAL, "belief_update: no belief block resident");
        if (!act || !obs || !out) FAIL(PBVI_EINVAL, "belief_update: NULL argument");
        if ((int64_t)S_ * R_ > 0x7fffffff) FAIL(PBVI_EUNSUPPORTED, "belief_update: S*R exceeds int32");
        for (int64_t b = 0; b < B_; ++b)
            if (act[b] < 0 || act[b] >= A_ || obs[b] < 0 || obs[b] >= O_) FAIL(PBVI_EINVAL, "belief_update: action / observation out of range");
        HIPCHK(hipSetDevice(device_));
        int rc = build_inverse_lists();
        if (rc) return rc;
        if ((rc = bu_act_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = bu_obs_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = bu_unnorm_.ensure((size_t)B_ * S_ * sizeof(double), &bytes_))) return rc;
        if ((rc = bu_mass_.ensure((size_t)B_ * sizeof(double), &bytes_))) return rc;
        if ((rc = bu_out_.ensure((size_t)B_ * S_ * sizeof(T), &bytes_))) return rc;
        // act/obs arrive in the caller's belief order; the resident block may be sorted
        if ((rc = host_perm())) return rc;
        std::vector<int32_t> ha((size_t)B_), ho((size_t)B_);
        for (int64_t i = 0; i < B_; ++i) {
            const int64_t c = sorted_ ? h_perm_[(size_t)i] : i;
            ha[(size_t)i] = act[c];
            ho[(size_t)i] = obs[c];
        }
        HIPCHK(hipMemcpyAsync(bu_act_.p, ha.data(), (size_t)B_ * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(bu_obs_.p, ho.data(), (size_t)B_ * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemsetAsync(bu_mass_.p, 0, (size_t)B_ * sizeof(double), stream_));
        // results go to rows in caller order: kernel row i writes out row perm[i] via a row-pointer trick is not
        // needed -- compute in engine order into stage, then gather back
        HIPCHK(launch_belief_update<T>(bel_.as<T>(), S_pad_, (int)B_, view(), in_ptr_.as<int32_t>(), in_src_.as<int32_t>(),
                                       bu_act_.as<int32_t>(), bu_obs_.as<int32_t>(), nullptr, bu_unnorm_.as<double>(),
                                       bu_mass_.as<double>(), bu_out_.as<T>(), S_, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (!sorted_) {
            HIPCHK(hipMemcpy(out, bu_out_.p, (size_t)B_ * S_ * sizeof(T), hipMemcpyDefault));
        } else {   // row i of the engine order is the caller's row perm[i]
            for (int64_t i = 0; i < B_; ++i)
                HIPCHK(hipMemcpyAsync((char*)out + (size_t)h_perm_[(size_t)i] * S_ * sizeof(T),
                                      bu_out_.as<T>() + (size_t)i * S_, (size_t)S_ * sizeof(T), hipMemcpyDefault, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
        }
        return PBVI_OK;
    }

    // Simulator step on the resident block (src/pomdp.py:3305-3311 update, :3326-3329 done-filter): every belief is
    // updated with its own (a, o); rows with keep[b] == 0 are dropped and the survivors become the resident block,
    // in the caller's order.  The beliefs never leave the device.
    int beliefs_advance(const int32_t* act, const int32_t* obs, const uint8_t* keep, int64_t* out_B) override {
        if (B_ <= 0) FAIL(PBVI_EINVAL, "beliefs_advance: no belief block resident");
        if (!act || !obs) FAIL(PBVI_EINVAL, "beliefs_advance: NULL argument");
        if ((int64_t)S_ * R_ > 0x7fffffff) FAIL(PBVI_EUNSUPPORTED, "beliefs_advance: S*R exceeds int32");
        for (int64_t b = 0; b < B_; ++b)
            if (act[b] < 0 || act[b] >= A_ || obs[b] < 0 || obs[b] >= O_) FAIL(PBVI_EINVAL, "beliefs_advance: action / observation out of range");
        HIPCHK(hipSetDevice(device_));
        int rc = build_inverse_lists();
        if (rc) return rc;
        std::vector<int32_t> dst((size_t)B_);           // caller row -> surviving row
        int64_t nb = 0;
        for (int64_t c = 0; c < B_; ++c) dst[(size_t)c] = (!keep || keep[c]) ? (int32_t)nb++ : -1;
        if (out_B) *out_B = nb;
        if (nb == 0) {                                   // every simulation finished
            B_ = 0;
            B_pad_ = 0;
            sorted_ = false;
            have_result_ = false;
            return PBVI_OK;
        }
        if ((rc = bu_act_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = bu_obs_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = bu_row_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = bu_unnorm_.ensure((size_t)B_ * S_ * sizeof(double), &bytes_))) return rc;
        if ((rc = bu_mass_.ensure((size_t)B_ * sizeof(double), &bytes_))) return rc;
        if ((rc = stage_.ensure((size_t)nb * S_pad_ * sizeof(T), &bytes_))) return rc;
        if ((rc = host_perm())) return rc;
        std::vector<int32_t> ha((size_t)B_), ho((size_t)B_), hr((size_t)B_);
        for (int64_t i = 0; i < B_; ++i) {               // engine row i holds the caller's row c
            const int64_t c = sorted_ ? h_perm_[(size_t)i] : i;
            ha[(size_t)i] = act[c];
            ho[(size_t)i] = obs[c];
            hr[(size_t)i] = dst[(size_t)c];
        }
        HIPCHK(hipMemcpyAsync(bu_act_.p, ha.data(), (size_t)B_ * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(bu_obs_.p, ho.data(), (size_t)B_ * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemcpyAsync(bu_row_.p, hr.data(), (size_t)B_ * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        HIPCHK(hipMemsetAsync(bu_mass_.p, 0, (size_t)B_ * sizeof(double), stream_));
        HIPCHK(hipMemsetAsync(stage_.p, 0, (size_t)nb * S_pad_ * sizeof(T), stream_));   // pad columns stay zero
        HIPCHK(launch_belief_update<T>(bel_.as<T>(), S_pad_, (int)B_, view(), in_ptr_.as<int32_t>(), in_src_.as<int32_t>(),
                                       bu_act_.as<int32_t>(), bu_obs_.as<int32_t>(), bu_row_.as<int32_t>(),
                                       bu_unnorm_.as<double>(), bu_mass_.as<double>(), stage_.as<T>(), S_pad_, stream_));
        HIPCHK(hipStreamSynchronize(stream_));           // ha/ho/hr are pageable host vectors
        return beliefs_finish(nb, stage_.as<T>(), nullptr);
    }

    // resident belief block back to the host (or a device buffer), caller order: [B][S] T
    int beliefs_fetch(void* out) override {
        if (B_ <= 0) FAIL(PBVI_EINVAL, "beliefs_fetch: no belief block resident");
        if (!out) FAIL(PBVI_EINVAL, "beliefs_fetch: NULL argument");
        HIPCHK(hipSetDevice(device_));
        if (!sorted_) {
            HIPCHK(hipMemcpy2DAsync(out, (size_t)S_ * sizeof(T), bel_.p, (size_t)S_pad_ * sizeof(T), (size_t)S_ * sizeof(T),
                                    (size_t)B_, hipMemcpyDefault, stream_));
        } else {
            int rc = host_perm();
            if (rc) return rc;
            for (int64_t i = 0; i < B_; ++i)
                HIPCHK(hipMemcpyAsync((char*)out + (size_t)h_perm_[(size_t)i] * S_ * sizeof(T),
                                      bel_.as<T>() + (size_t)i * S_pad_, (size_t)S_ * sizeof(T), hipMemcpyDefault, stream_));
        }
        HIPCHK(hipStreamSynchronize(stream_));
        return PBVI_OK;
    }

    int64_t beliefs_count() const override { return B_; }

    // ---- device row stores ------------------------------------------------ //
    int64_t store_append(int which, const void* rows, int64_t n) override {
        if (which < 0 || which > 1 || n < 0 || (n > 0 && rows == nullptr)) FAIL(PBVI_EINVAL, "store_append: bad arguments");
        if (hipSetDevice(device_) != hipSuccess) FAIL(PBVI_ERUNTIME, "hipSetDevice failed");
        DevBuf& st = store_[which];
        const int64_t have = store_rows_[which];
        const size_t need = (size_t)(have + n) * S_pad_ * sizeof(T);
        if (need > st.cap) {   // grow geometrically, keep the stored rows
            DevBuf nb;
            int rc = nb.ensure(std::max(need, st.cap * 2), &bytes_);
            if (rc) return rc;
            if (have > 0 && hipMemcpyAsync(nb.p, st.p, (size_t)have * S_pad_ * sizeof(T), hipMemcpyDeviceToDevice, stream_) != hipSuccess)
                FAIL(PBVI_ERUNTIME, "store_append: device copy failed");
            if (hipStreamSynchronize(stream_) != hipSuccess) FAIL(PBVI_ERUNTIME, "store_append: sync failed");
            bytes_ -= (int64_t)st.cap;
            st.release();
            st = nb;
        }
        if (n > 0) {
            T* dst = st.as<T>() + (size_t)have * S_pad_;
            if (hipMemsetAsync(dst, 0, (size_t)n * S_pad_ * sizeof(T), stream_) != hipSuccess ||
                hipMemcpy2DAsync(dst, (size_t)S_pad_ * sizeof(T), rows, (size_t)S_ * sizeof(T), (size_t)S_ * sizeof(T),
                                 (size_t)n, hipMemcpyHostToDevice, stream_) != hipSuccess ||
                hipStreamSynchronize(stream_) != hipSuccess)
                FAIL(PBVI_ERUNTIME, "store_append: upload failed");
        }
        store_rows_[which] = have + n;
        return have;
    }

    // distinct alpha' rows of the last backup -> alpha store, device to device (ids are consecutive from the return)
    int64_t store_append_unique(const int32_t* unique_idx, int64_t n) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_store_unique: no backup result resident");
        if (n <= 0 || n > 65535 || unique_idx == nullptr) FAIL(PBVI_EINVAL, "backup_store_unique: bad arguments (1 <= n <= 65535)");
        for (int64_t i = 0; i < n; ++i)
            if (unique_idx[i] < 0 || unique_idx[i] >= res_unique_) FAIL(PBVI_EINVAL, "backup_store_unique: row index out of range");
        HIPCHK(hipSetDevice(device_));
        T* dst = nullptr;
        int rc = store_reserve(0, n, &dst);
        if (rc) return rc;
        if ((rc = ids_.ensure((size_t)n * sizeof(int32_t), &bytes_))) return rc;
        HIPCHK(hipMemcpyAsync(ids_.p, unique_idx, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        if (S_pad_ > S_) HIPCHK(hipMemsetAsync(dst, 0, (size_t)n * S_pad_ * sizeof(T), stream_));   // pad columns stay zero
        hipLaunchKernelGGL(k_gather_rows_ld<T>, dim3((S_ + 255) / 256, (unsigned)n), dim3(256), 0, stream_, out_.as<T>(), S_,
                           dst, S_pad_, S_, ids_.as<int32_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream_));            // unique_idx is the caller's host memory
        const int64_t first = store_rows_[0];
        store_rows_[0] = first + n;
        return first;
    }

    // make room for n more rows in a store, keeping what is there; returns the destination of the new rows
    int store_reserve(int which, int64_t n, T** dst) {
        DevBuf& st = store_[which];
        const int64_t have = store_rows_[which];
        const size_t need = (size_t)(have + n) * S_pad_ * sizeof(T);
        if (need > st.cap) {
            DevBuf nb;
            int rc = nb.ensure(std::max(need, st.cap * 2), &bytes_);
            if (rc) return rc;
            if (have > 0) HIPCHK(hipMemcpyAsync(nb.p, st.p, (size_t)have * S_pad_ * sizeof(T), hipMemcpyDeviceToDevice, stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            bytes_ -= (int64_t)st.cap;
            st.release();
            st = nb;
        }
        *dst = st.as<T>() + (size_t)have * S_pad_;
        return PBVI_OK;
    }

    // fp64 copy of RTO (reference layout [S][A][O][R]) for the belief walk of an f32 engine: the host mirror keeps
    // fp64 belief values, and they should not depend on the engine's arithmetic type
    int set_rto_f64(const double* rto) override {
        if (!rto) FAIL(PBVI_EINVAL, "set_rto_f64: NULL table");
        HIPCHK(hipSetDevice(device_));
        std::vector<double> h((size_t)A_ * O_ * R_ * S_pad_, 0.0);
        for (int s = 0; s < S_; ++s)
            for (int a = 0; a < A_; ++a)
                for (int o = 0; o < O_; ++o)
                    for (int r = 0; r < R_; ++r)
                        h[(((size_t)a * O_ + o) * R_ + r) * S_pad_ + s] = rto[(((size_t)s * A_ + a) * O_ + o) * R_ + r];
        int rc = rto64_.ensure(h.size() * sizeof(double), &bytes_);
        if (rc) return rc;
        HIPCHK(hipMemcpy(rto64_.p, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        return PBVI_OK;
    }

    // FSVI-style walk: n chained Bayes updates on the device (fp64), every new belief appended to the belief store
    // and copied to `out` [n][S] fp64.  Returns the store id of the first new belief.
    int64_t belief_walk(const double* b0, int64_t n, const int32_t* act, const int32_t* obs, const uint8_t* restart,
                        double* out) override {
        if (!b0 || n <= 0 || !act || !obs || !out) FAIL(PBVI_EINVAL, "belief_walk: bad arguments");
        if ((int64_t)S_ * R_ > 0x7fffffff) FAIL(PBVI_EUNSUPPORTED, "belief_walk: S*R exceeds int32");
        for (int64_t i = 0; i < n; ++i)
            if (act[i] < 0 || act[i] >= A_ || obs[i] < 0 || obs[i] >= O_) FAIL(PBVI_EINVAL, "belief_walk: action / observation out of range");
        HIPCHK(hipSetDevice(device_));
        int rc = build_inverse_lists();
        if (rc) return rc;
        const int blocks = (S_pad_ + 255) / 256;
        if ((rc = walk64_.ensure((size_t)(n + 1) * S_ * sizeof(double), &bytes_))) return rc;      // row 0 = b0
        if ((rc = bu_unnorm_.ensure((size_t)2 * S_ * sizeof(double), &bytes_))) return rc;          // ping-pong: raw row of step i
        if ((rc = bu_mass_.ensure((size_t)2 * blocks * sizeof(double), &bytes_))) return rc;      // and its block partials
        T* dst = nullptr;
        if ((rc = store_reserve(1, n, &dst))) return rc;
        double* rows = walk64_.as<double>();
        HIPCHK(hipMemcpyAsync(rows, b0, (size_t)S_ * sizeof(double), hipMemcpyHostToDevice, stream_));
        const ModelView<T> mv = view();
        // The fp64 rows go back to the host while the chain is still running: in quarters, each copied by the side
        // stream into the pinned buffer as soon as its last step is done, and moved on to the caller's memory by this
        // thread while the device works on the next quarter (the 24 MB of a 99-step walk at S = 30000 took 1.4 ms
        // after a 1.2 ms chain; now most of it hides behind the chain).
        const bool to_host = !is_device_pointer(out);
        constexpr int kChunks = 4;
        int64_t chunk_end[kChunks];
        int n_chunks = 0;
        if (to_host && n >= 2 * kChunks) {
            if ((rc = out_begin())) return rc;                                   // (nothing of an earlier call is staged)
            if ((rc = stage_reserve((size_t)n * S_ * sizeof(double)))) return rc;
            for (auto& e : walk_ev_)
                if (!e) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            n_chunks = kChunks;
            for (int c = 0; c < kChunks; ++c) chunk_end[c] = n * (c + 1) / kChunks;
        }
        // step i: one kernel pushes belief i (b0 at the start and after a restart, else the previous step's raw row
        // normalised on the fly) and writes belief i (rows64 row i, store row i - 1); the last belief is finished by a
        // normalisation of its own.  A quarter of the rows is complete once the kernel after its last step is queued.
        static const bool two_kernels = getenv("PBVI_WALK_TWO_KERNELS") != nullptr;      // debug / A-B only: the unfused chain
        auto queue_copy = [&](int c) -> int {
            const int64_t r0 = c ? chunk_end[c - 1] : 0, r1 = chunk_end[c];
            HIPCHK(hipEventRecord(walk_ev_[2 * c], stream_));
            HIPCHK(hipStreamWaitEvent(stream2_, walk_ev_[2 * c], 0));
            HIPCHK(hipMemcpyAsync((char*)host_stage_ + (size_t)r0 * S_ * sizeof(double), rows + (size_t)(r0 + 1) * S_,
                                  (size_t)(r1 - r0) * S_ * sizeof(double), hipMemcpyDeviceToHost, stream2_));
            HIPCHK(hipEventRecord(walk_ev_[2 * c + 1], stream2_));
            return PBVI_OK;
        };
        int c_next = 0;
        double* un[2] = {bu_unnorm_.as<double>(), bu_unnorm_.as<double>() + S_};
        double* pa[2] = {bu_mass_.as<double>(), bu_mass_.as<double>() + blocks};
        for (int64_t i = 0; i < n; ++i) {
            if (two_kernels) {
                const double* base = (restart && restart[i]) ? rows : rows + (size_t)i * S_;       // row i = b_i (row 0 = b0)
                HIPCHK(launch_walk_step<T>(base, mv, rto64_.as<double>(), in_ptr_.as<int32_t>(), in_src_.as<int32_t>(), act[i], obs[i],
                                           un[0], pa[0], rows + (size_t)(i + 1) * S_, dst + (size_t)i * S_pad_, stream_));
                if (c_next < n_chunks && i + 1 == chunk_end[c_next]) {
                    if ((rc = queue_copy(c_next))) return rc;
                    ++c_next;
                }
                continue;
            }
            const bool from_b0 = i == 0 || (restart && restart[i]);
            const double* prev_un = i > 0 ? un[(i - 1) & 1] : nullptr;
            HIPCHK(launch_walk_fused<T>(from_b0 ? rows : nullptr, prev_un, i > 0 ? pa[(i - 1) & 1] : nullptr,
                                        i > 0 ? rows + (size_t)i * S_ : nullptr, i > 0 ? dst + (size_t)(i - 1) * S_pad_ : nullptr, mv,
                                        rto64_.as<double>(), in_ptr_.as<int32_t>(), in_src_.as<int32_t>(), act[i], obs[i], un[i & 1],
                                        pa[i & 1], stream_));
            if (c_next < n_chunks && i == chunk_end[c_next]) {         // beliefs 1 .. i are written
                if ((rc = queue_copy(c_next))) return rc;
                ++c_next;
            }
        }
        if (!two_kernels) {
            HIPCHK(launch_walk_finish<T>(un[(n - 1) & 1], pa[(n - 1) & 1], mv, rows + (size_t)n * S_, dst + (size_t)(n - 1) * S_pad_, stream_));
            while (c_next < n_chunks) {
                if ((rc = queue_copy(c_next))) return rc;
                ++c_next;
            }
        }
        if (n_chunks > 0) {
            for (int c = 0; c < n_chunks; ++c) {
                const int64_t r0 = c ? chunk_end[c - 1] : 0;
                HIPCHK(hipEventSynchronize(walk_ev_[2 * c + 1]));
                host_copy((char*)out + (size_t)r0 * S_ * sizeof(double), (const char*)host_stage_ + (size_t)r0 * S_ * sizeof(double),
                            (size_t)(chunk_end[c] - r0) * S_ * sizeof(double));
            }
            HIPCHK(hipStreamSynchronize(stream_));
        } else {
            if ((rc = out_begin())) return rc;
            if ((rc = out_add(out, rows + S_, (size_t)n * S_ * sizeof(double)))) return rc;
            if ((rc = out_finish())) return rc;
        }
        const int64_t first = store_rows_[1];
        store_rows_[1] = first + n;
        walk_rows_ = n;
        return first;
    }

    // max_v b.alpha_v of the last backup's beliefs against its alpha set, caller order (belief-side formulation only)
    int backup_value_max(double* out_value) override {
        if (!have_result_ || !have_bk_vmax_)
            FAIL(PBVI_EUNSUPPORTED, "backup_fetch_value_max: the last backup did not run in the belief-side formulation");
        if (!out_value) FAIL(PBVI_EINVAL, "backup_fetch_value_max: NULL destination");
        HIPCHK(hipSetDevice(device_));
        int rc;
        if ((rc = out_begin())) return rc;
        if ((rc = out_add(out_value, vmax_bk_.p, (size_t)res_B_ * sizeof(double)))) return rc;
        return out_finish();
    }

    // position-weighted bit-pattern hashes (k_row_hash) of the fp64 rows of the last walk (rows 1..n of walk64_), for the
    // host's dedup keys
    int walk_keys(int64_t n, uint64_t* out_keys) override {
        if (n <= 0 || n != walk_rows_ || !out_keys) FAIL(PBVI_EINVAL, "belief_walk_keys: n must be the length of the last walk");
        HIPCHK(hipSetDevice(device_));
        int rc = keys_tmp_.ensure((size_t)n * sizeof(uint64_t), &bytes_);
        if (rc) return rc;
        hipLaunchKernelGGL(k_row_hash<double>, dim3((unsigned)n), dim3(256), 0, stream_, walk64_.as<double>() + S_, S_, S_,
                           keys_tmp_.as<unsigned long long>());
        HIPCHK(hipGetLastError());
        if ((rc = out_begin())) return rc;
        if ((rc = out_add(out_keys, keys_tmp_.p, (size_t)n * sizeof(uint64_t)))) return rc;
        return out_finish();
    }

    // After a call returned PBVI_ENOMEM: back to the state of a fresh engine (model tables kept, every working set, row
    // store and scratch buffer released), so that the caller -- PBVI_Solver.solve turns the error into "return the
    // partial result" like src/pomdp.py:2399-2401 -- can go on using the engine with whatever it uploads next.
    int after_oom() override {
        HIPCHK(hipSetDevice(device_));
        (void)hipStreamSynchronize(stream_);
        if (stream2_) (void)hipStreamSynchronize(stream2_);
        if (stream3_) (void)hipStreamSynchronize(stream3_);
        DevBuf* drop[] = {&alpha_buf_, &alpha_small_, &bel_, &gam_, &slabs_, &best_v_, &best_score_, &err_, &dead_, &queue_, &rdot_,
                          &action_, &aqueue_, &out_, &keep_, &bv2_, &bs2_, &err2_, &queue2_, &prune_cnt_, &nzA_, &klist_, &kcount_,
                          &nchunks_, &need_, &skws_, &stage_, &keys_, &perm_, &action_res_, &best_res_, &rep_, &uniq_, &inv_, &slot_,
                          &out_full_, &btl_, &btc_, &val_exact_, &store_[0], &store_[1], &ids_, &bu_act_, &bu_obs_, &bu_unnorm_,
                          &bu_mass_, &bu_out_, &bu_row_, &walk64_, &bp_, &nzP_, &pmag_, &prd_, &keys_tmp_, &keys_act_, &keys_best_,
                          &keys_rows_, &rf_v_, &rf_slot_, &rf_sc_, &rf_entry_, &rf_n_, &rf_tiles_, &snz_, &sbtl_, &sbtc_, &vmax_bk_,
                          &rf_ibv_, &rf_ibi_, &rf_cnt_, &rf_W_, &rf_Cx_, &rf_nzW_, &rf_klW_, &rf_kcW_, &nzAlpha_, &prod_, &klistD_,
                          &kcountD_, &nchunksD_, &mat_, &vlist_, &rowflags_, &ctile_, &acand_, &rf_q2_, &rf_q2p_, &rf_q2d_, &e_bv_, &e_bs_, &e_err_, &e_rdot_, &e_act_, &e_ares_, &e_bres_, &e_rep_, &e_uniq_, &e_inv_, &e_slotd_, &e_out_, &e_slot_};
        for (DevBuf* b : drop) {
            bytes_ -= (int64_t)b->cap;
            b->release();
        }
        alpha_ = DevBuf{};                                   // a view into alpha_buf_ / alpha_small_
        V_ = 0;
        B_ = B_pad_ = 0;
        prim_valid_ = alpha_on_primary_ = false;
        prim_ids_.clear();
        prim_off_ = 0;
        store_rows_[0] = store_rows_[1] = 0;
        snz_rows_ = sbt_rows_ = 0;
        walk_rows_ = 0;
        have_result_ = have_bk_vmax_ = false;
        btl_valid_ = false;
        sorted_ = false;
        h_perm_valid_ = false;
        mat_V_ = -1;
        gam_pad_ptr_ = nullptr;
        last_deferred_nothing_ = false;
        ++alpha_ver_;
        ++bel_ver_;
        if (screen_) {
            screen_alpha_seen_ = screen_bel_seen_ = 0;
            screen_layout_seen_ = 0;
            return screen_->after_oom();
        }
        return PBVI_OK;
    }

    int store_reset(int which) override {
        if (which < 0 || which > 1) FAIL(PBVI_EINVAL, "store_reset: bad store");
        store_rows_[which] = 0;
        if (which == 1) snz_rows_ = sbt_rows_ = 0;
        if (which == 0) {                                    // the ids of the primary working set name rows that are gone
            prim_valid_ = false;
            prim_ids_.clear();
        }
        return PBVI_OK;
    }

    int store_select(int which, const int32_t* ids, int64_t n) override {
        if (which < 0 || which > 1 || n <= 0 || ids == nullptr) FAIL(PBVI_EINVAL, "store_select: bad arguments");
        if (n > 65535 && which == 1) FAIL(PBVI_EUNSUPPORTED, "at most 65535 beliefs per block");
        for (int64_t i = 0; i < n; ++i)
            if (ids[i] < 0 || ids[i] >= store_rows_[which]) FAIL(PBVI_EINVAL, "store_select: id out of range");
        HIPCHK(hipSetDevice(device_));
        int rc = ids_.ensure((size_t)n * sizeof(int32_t), &bytes_);
        if (rc) return rc;
        if (which == 1) {   // not synchronised below: the ids travel through a pinned buffer of the engine's own
            const size_t bytes = (size_t)n * sizeof(int32_t);
            if (ids_pin_cap_ < bytes) {
                HIPCHK(hipStreamSynchronize(stream_));
                if (ids_pin_) (void)hipHostFree(ids_pin_);
                ids_pin_ = nullptr;
                ids_pin_cap_ = 0;
                if (hipHostMalloc(&ids_pin_, std::max<size_t>(bytes, 65536), hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    FAIL(PBVI_ENOMEM, "store_select: pinned staging for the ids");
                }
                ids_pin_cap_ = std::max<size_t>(bytes, 65536);
            } else if (ev_ids_) {
                HIPCHK(hipEventSynchronize(ev_ids_));        // the previous selection's copy has left the buffer
            }
            std::memcpy(ids_pin_, ids, bytes);
            HIPCHK(hipMemcpyAsync(ids_.p, ids_pin_, bytes, hipMemcpyHostToDevice, stream_));
            if (!ev_ids_) HIPCHK(hipEventCreateWithFlags(&ev_ids_, hipEventDisableTiming));
            HIPCHK(hipEventRecord(ev_ids_, stream_));
        } else {
            HIPCHK(hipMemcpyAsync(ids_.p, ids, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
        }
        if (which == 0) {   // working alpha set := stored rows in the caller's order
            static const bool no_prepend = getenv("PBVI_NO_ALPHA_PREPEND") != nullptr;      // debug / A-B only
            auto gather = [&](T* dst, int64_t first, int64_t cnt) -> int {             // ids [first, first + cnt) -> dst rows
                for (int64_t r0 = 0; r0 < cnt; r0 += 65535) {
                    const unsigned c = (unsigned)std::min<int64_t>(65535, cnt - r0);
                    hipLaunchKernelGGL(k_gather_rows<T>, dim3((S_pad_ + 255) / 256, c), dim3(256), 0, stream_, store_[0].as<T>(),
                                       dst + (size_t)r0 * S_pad_, S_pad_, ids_.as<int32_t>() + first + r0);
                    HIPCHK(hipGetLastError());
                }
                return PBVI_OK;
            };
            const int64_t pv = (int64_t)prim_ids_.size();
            // (1) the primary set with k new rows in front of it (k = 0: the primary set again, after a small selection)
            if (!no_prepend && prim_valid_ && n >= pv && n - pv <= prim_off_ &&
                std::equal(prim_ids_.begin(), prim_ids_.end(), ids + (n - pv))) {
                const int64_t k = n - pv;
                prim_off_ -= k;
                alpha_view(alpha_buf_, prim_off_);
                V_ = n;                                           // the magnitude row stays where it is: row n of the new view
                if (k > 0) {
                    if ((rc = gather(alpha_.as<T>(), 0, k))) return rc;
                    if ((rc = merge_magnitude_row(alpha_.as<T>(), k))) return rc;
                    prim_ids_.insert(prim_ids_.begin(), ids, ids + k);
                }
                alpha_on_primary_ = true;
                ++alpha_ver_;
                HIPCHK(hipStreamSynchronize(stream_));
                have_result_ = false;
                return PBVI_OK;
            }
            // (2) a selection much smaller than the primary set: beside it
            if (!no_prepend && prim_valid_ && n <= 4096 && 4 * n <= pv) {
                const size_t rows = alpha_rows_cap(n);
                if ((rc = alpha_small_.ensure(rows * S_pad_ * sizeof(T), &bytes_))) return rc;
                alpha_view(alpha_small_, 0);
                alpha_on_primary_ = false;
                HIPCHK(hipMemsetAsync(alpha_.as<T>() + (size_t)n * S_pad_, 0, (rows - (size_t)n) * S_pad_ * sizeof(T), stream_));
                if ((rc = gather(alpha_.as<T>(), 0, n))) return rc;
                V_ = n;
                ++alpha_ver_;
                if ((rc = refresh_magnitude_row())) return rc;
                HIPCHK(hipStreamSynchronize(stream_));
                have_result_ = false;
                return PBVI_OK;
            }
            // (3) a new primary set: free rows in front (half its size again, for the sets that will extend it), the data,
            // the magnitude row, and 256 + padding zero rows behind (the view's 256-row padding reaches further back as the
            // view starts earlier)
            const int64_t front = no_prepend ? 0 : std::max<int64_t>(512, n / 2);
            const size_t rows = (size_t)front + alpha_rows_cap(n) + (no_prepend ? 0 : GEMM_BN);
            if ((rc = alpha_buf_.ensure(rows * S_pad_ * sizeof(T), &bytes_))) return rc;
            prim_off_ = front;
            alpha_view(alpha_buf_, prim_off_);
            HIPCHK(hipMemsetAsync(alpha_.as<T>() + (size_t)n * S_pad_, 0, (rows - (size_t)front - (size_t)n) * S_pad_ * sizeof(T), stream_));
            if ((rc = gather(alpha_.as<T>(), 0, n))) return rc;
            V_ = n;
            prim_ids_.assign(ids, ids + n);
            prim_valid_ = !no_prepend;
            alpha_on_primary_ = !no_prepend;
            ++prim_layout_ver_;
            ++alpha_ver_;
            if ((rc = refresh_magnitude_row())) return rc;
            HIPCHK(hipStreamSynchronize(stream_));
            have_result_ = false;
            return PBVI_OK;
        }
        // the block is gathered straight from the store rows, in sorted order (no staging copy); nothing here waits for the
        // device: the backup that follows is enqueued behind it
        return beliefs_finish(n, store_[1].as<T>(), ids_.as<int32_t>());
    }

    // device pointer to the stream-K share size (K-tile steps of the longest f32 chain) or nullptr
    const int* chain_steps() const {
        if (!kF32 || !plan_.streamk) return nullptr;
        const size_t pairs = (size_t)plan_.tiles_m * plan_.tiles_n;
        return skws_.as<int>() + (pairs + 1) + plan_.nblocks + pairs;
    }
    double tie_window(int k_chunk) const {
        if (!kF32) return 0.0;
        if (tie_rel_user_ > 0.0) return tie_rel_user_;
        if (plan_.streamk) return -1.0;            // computed on the device from the stream-K share size
        const double u = 5.9604644775390625e-08;   // 2^-24
        return 8.0 * u * std::sqrt((double)std::max(k_chunk, 1)) + 8.0 * u;
    }

    // score GEMM: C = beliefs[B_pad][S_pad] . Y[rows_y][S_pad]^T -> slabs_.  Y's zero structure:
    // G row groups of v_group rows with support nzB (nullptr = dense Y).
    // scores of X rows (default: the resident belief block) against the rows of Y
    int score_gemm(const T* Y, int64_t rows_y, const uint8_t* nzB, int G, int v_group, SlabView<T>* sv,
                   const T* X = nullptr, int64_t x_rows = 0, const uint8_t* nzX = nullptr, hipStream_t list_stream = nullptr,
                   const FusedB* fused = nullptr);

    int project_dense(double gamma);   // K1-dense: Gamma = gamma * alpha . D_ao^T as A*O (batched) GEMMs
    int backup_run(double gamma, int flags, pbvi_stats_t* st) override;

    // ---- the scoring stage (K1, K2, first-max), runnable on this engine or on an fp64 engine's fp32 screen ---- //
    bool choose_push(int64_t N) const;
    int stage_scores(double gamma, bool use_push, const ScoreIO& io, ScoreStage<T>* out);
    template <typename TS>
    int run_pipeline(EngineT<TS>& scorer, double gamma, int flags, pbvi_stats_t* st);
    int ensure_screen();
    int sync_screen();

    // per-belief alpha' matrix [B][S] (the reference seam) expanded from the unique rows on first use
    int ensure_full() {
        if (full_valid_) return PBVI_OK;
        int rc = out_full_.ensure((size_t)res_B_ * S_ * sizeof(T), &bytes_);
        if (rc) return rc;
        HIPCHK(launch_expand_rows<T>(out_.as<T>(), inv_.as<int32_t>(), out_full_.as<T>(), (int)res_B_, S_, stream_));
        full_valid_ = true;
        return PBVI_OK;
    }

    int backup_fetch(void* out_alpha, int32_t* out_action, int32_t* out_best, uint8_t* out_keep) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_fetch: no backup result resident (call pbvi_backup_run first)");
        HIPCHK(hipSetDevice(device_));
        const size_t B = (size_t)res_B_;
        int rc;
        if ((rc = out_begin())) return rc;
        if (out_alpha) {
            if ((rc = ensure_full())) return rc;
            if ((rc = out_add(out_alpha, out_full_.p, B * S_ * sizeof(T)))) return rc;
        }
        if (out_action && (rc = out_add(out_action, res_action_, B * sizeof(int32_t)))) return rc;
        if (out_best && (rc = out_add(out_best, res_best_, B * A_ * O_ * sizeof(int32_t)))) return rc;
        if (out_keep && (rc = out_add(out_keep, keep_.p, B))) return rc;
        return out_finish();
    }

    int64_t unique_count() const override { return have_result_ ? res_unique_ : -1; }

    int fetch_unique(void* out_rows, int32_t* out_index) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_fetch_unique: no backup result resident");
        HIPCHK(hipSetDevice(device_));
        int rc;
        if ((rc = out_begin())) return rc;
        if (out_rows && (rc = out_add(out_rows, out_.p, (size_t)res_unique_ * S_ * sizeof(T)))) return rc;
        if (out_index && (rc = out_add(out_index, inv_.p, (size_t)res_B_ * sizeof(int32_t)))) return rc;
        return out_finish();
    }

    // everything a caller of backup() needs, in one staged transfer and one synchronisation: the U distinct rows, the
    // per-belief index into them, actions, best_alpha_ind and the keep mask (any destination may be NULL)
    // pbvi_backup_run + pbvi_backup_fetch_compact in one call; the rows go to page-locked `out_rows` in SLOT order
    // (out_slot[u] = row of distinct key u) and most of them leave while the refinement is still running (run_pipeline).
    int run_fetch(double gamma, int flags, pbvi_stats_t* st, void* out_rows, int64_t cap_rows, int32_t* out_slot, int32_t* out_index,
                  int32_t* out_action, int32_t* out_best, uint8_t* out_keep, int64_t* n_unique, int64_t* n_slots) override {
        if (!out_rows || !out_slot || !out_index || !out_action) FAIL(PBVI_EINVAL, "backup_run_fetch: NULL destination");
        if (cap_rows < B_) FAIL(PBVI_EINVAL, "backup_run_fetch: out_rows must have room for one row per belief");
        if (!is_pinned_host_pointer(out_rows)) FAIL(PBVI_EINVAL, "backup_run_fetch: out_rows must be page-locked host memory (pbvi_host_alloc)");
        early_rows_ = out_rows;
        early_cap_ = cap_rows;
        rf_dst_ = RunFetchDst{out_slot, out_index, out_action, out_best, out_keep, false};
        int rc = backup_run(gamma, flags, st);
        early_rows_ = nullptr;
        const bool queued = rf_dst_.queued;
        rf_dst_ = RunFetchDst{};
        if (rc) return rc;
        const size_t B = (size_t)res_B_;
        int64_t used = res_unique_;
        const bool early_ok = early_used_ && h_flag_ && !h_flag_[6];
        if (early_ok) used = (int64_t)h_flag_[4] + h_flag_[5];
        if (queued) {                                        // the small arrays came with the pipeline's last read-back
            if ((rc = out_flush())) return rc;               // (host-side copies of staged items; the stream is idle)
            if (!early_ok) {                                 // the slots overflowed: rows in the plain order
                if ((rc = out_begin())) return rc;
                if ((rc = out_add(out_rows, out_.p, (size_t)res_unique_ * S_ * sizeof(T)))) return rc;
                if ((rc = out_finish())) return rc;
            }
        } else {                                             // the early path did not apply: everything now
            if ((rc = out_begin())) return rc;
            if ((rc = out_add(out_rows, out_.p, (size_t)res_unique_ * S_ * sizeof(T)))) return rc;
            if ((rc = out_add(out_index, inv_.p, B * sizeof(int32_t)))) return rc;
            if ((rc = out_add(out_action, res_action_, B * sizeof(int32_t)))) return rc;
            if (out_best && (rc = out_add(out_best, res_best_, B * A_ * O_ * sizeof(int32_t)))) return rc;
            if (out_keep && (rc = out_add(out_keep, keep_.p, B))) return rc;
            if ((rc = out_finish())) return rc;
        }
        if (!early_ok)
            for (int64_t u = 0; u < res_unique_; ++u) out_slot[u] = (int32_t)u;
        if (n_unique) *n_unique = res_unique_;
        if (n_slots) *n_slots = used;
        early_used_ = false;
        return PBVI_OK;
    }

    int fetch_compact(void* out_rows, int32_t* out_index, int32_t* out_action, int32_t* out_best, uint8_t* out_keep) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_fetch_compact: no backup result resident");
        HIPCHK(hipSetDevice(device_));
        const size_t B = (size_t)res_B_;
        int rc;
        if ((rc = out_begin())) return rc;
        if (out_rows && (rc = out_add(out_rows, out_.p, (size_t)res_unique_ * S_ * sizeof(T)))) return rc;
        if (out_index && (rc = out_add(out_index, inv_.p, B * sizeof(int32_t)))) return rc;
        if (out_action && (rc = out_add(out_action, res_action_, B * sizeof(int32_t)))) return rc;
        if (out_best && (rc = out_add(out_best, res_best_, B * A_ * O_ * sizeof(int32_t)))) return rc;
        if (out_keep && (rc = out_add(out_keep, keep_.p, B))) return rc;
        return out_finish();
    }

    // ---- device -> host results ------------------------------------------------------------------------------------ //
    // Results bound for ordinary (pageable) host memory go through one long-lived pinned buffer and a CPU memcpy:
    // the DMA never targets memory the driver has to register first.  Device destinations are copied directly.
    // (The caller should not map fresh host memory per call either: see HostArena in engine.py.)
    struct OutItem {
        void* dst;
        size_t off, bytes;
    };
    std::vector<OutItem> out_items_;
    size_t out_used_ = 0;

    int out_begin() {
        out_items_.clear();
        out_used_ = 0;
        return PBVI_OK;
    }
    int out_flush() {
        HIPCHK(hipStreamSynchronize(stream_));
        for (const OutItem& it : out_items_) host_copy(it.dst, (const char*)host_stage_ + it.off, it.bytes);
        out_items_.clear();
        out_used_ = 0;
        return PBVI_OK;
    }
    int out_add(void* dst, const void* src_dev, size_t bytes) {
        if (bytes == 0) return PBVI_OK;
        if (is_device_pointer(dst)) {
            HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToDevice, stream_));
            return PBVI_OK;
        }
        if (is_pinned_host_pointer(dst)) {                   // caller's page-locked buffer: DMA straight into it
            HIPCHK(hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDeviceToHost, stream_));
            return PBVI_OK;
        }
        const size_t need = (out_used_ + 255) / 256 * 256 + bytes;
        if (need > host_stage_cap_) {
            int rc = out_flush();                            // nothing staged may be lost when the buffer moves
            if (rc) return rc;
            if (bytes > host_stage_cap_) {
                if (host_stage_) (void)hipHostFree(host_stage_);
                host_stage_ = nullptr;
                host_stage_cap_ = 0;
                const size_t want = std::max<size_t>(bytes * 2, (size_t)64 << 20);
                if (hipHostMalloc(&host_stage_, want, hipHostMallocDefault) != hipSuccess) {
                    (void)hipGetLastError();
                    host_stage_ = nullptr;
                    FAIL(PBVI_ENOMEM, "pinned staging allocation failed");
                }
                host_stage_cap_ = want;
            }
        }
        const size_t off = (out_used_ + 255) / 256 * 256;
        HIPCHK(hipMemcpyAsync((char*)host_stage_ + off, src_dev, bytes, hipMemcpyDeviceToHost, stream_));
        out_items_.push_back({dst, off, bytes});
        out_used_ = off + bytes;
        return PBVI_OK;
    }
    int out_finish() { return out_flush(); }
    // pinned buffer of at least `bytes` with nothing staged in it
    int stage_reserve(size_t bytes) {
        int rc = out_flush();
        if (rc) return rc;
        if (bytes > host_stage_cap_) {
            if (host_stage_) (void)hipHostFree(host_stage_);
            host_stage_ = nullptr;
            host_stage_cap_ = 0;
            const size_t want = std::max<size_t>(bytes * 2, (size_t)64 << 20);
            if (hipHostMalloc(&host_stage_, want, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                host_stage_ = nullptr;
                FAIL(PBVI_ENOMEM, "pinned staging allocation failed");
            }
            host_stage_cap_ = want;
        }
        return PBVI_OK;
    }

    // position-weighted bit-pattern hashes of the U distinct rows of the last backup (see k_row_hash)
    int fetch_row_hashes(uint64_t* out) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_fetch_row_hashes: no backup result resident");
        if (!out) FAIL(PBVI_EINVAL, "backup_fetch_row_hashes: NULL destination");
        if (res_unique_ <= 0) return PBVI_OK;
        HIPCHK(hipSetDevice(device_));
        int rc = keys_tmp_.ensure((size_t)res_unique_ * sizeof(uint64_t), &bytes_);
        if (rc) return rc;
        hipLaunchKernelGGL(k_row_hash<T>, dim3((unsigned)res_unique_), dim3(256), 0, stream_, out_.as<T>(), S_, S_,
                           keys_tmp_.as<unsigned long long>());
        HIPCHK(hipGetLastError());
        if ((rc = out_begin())) return rc;
        if ((rc = out_add(out, keys_tmp_.p, (size_t)res_unique_ * sizeof(uint64_t)))) return rc;
        return out_finish();
    }

    // (a*, v*[a*,:]) of every unique row of the last backup: [U][1+O] int32, host or device destination
    int fetch_unique_keys(int32_t* out_keys) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_fetch_unique_keys: no backup result resident");
        if (!out_keys) FAIL(PBVI_EINVAL, "backup_fetch_unique_keys: NULL destination");
        if (res_unique_ <= 0) return PBVI_OK;
        HIPCHK(hipSetDevice(device_));
        int rc = keys_tmp_.ensure((size_t)res_unique_ * (1 + O_) * sizeof(int32_t), &bytes_);
        if (rc) return rc;
        hipLaunchKernelGGL(k_gather_keys, dim3((unsigned)((res_unique_ + 63) / 64)), dim3(64), 0, stream_, (int)res_unique_, A_, O_,
                           uniq_.as<int32_t>(), res_action_, res_best_, keys_tmp_.as<int32_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(out_keys, keys_tmp_.p, (size_t)res_unique_ * (1 + O_) * sizeof(int32_t), hipMemcpyDefault, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        return PBVI_OK;
    }

    // one buffer with everything this rank contributes to the multi-GPU exchange (layout: k_pack_exchange)
    int fetch_exchange(int32_t* out, int64_t per) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "backup_fetch_exchange: no backup result resident");
        if (!out) FAIL(PBVI_EINVAL, "backup_fetch_exchange: NULL destination");
        if (per < res_B_ || per > 0x7fffffff) FAIL(PBVI_EINVAL, "backup_fetch_exchange: per must be >= the number of beliefs of the last backup");
        HIPCHK(hipSetDevice(device_));
        const size_t n = 1 + 3 * (size_t)per + (size_t)per * (1 + O_);
        int32_t* dst = out;
        const bool direct = is_device_pointer(out);
        if (!direct) {
            int rc = keys_tmp_.ensure(n * sizeof(int32_t), &bytes_);
            if (rc) return rc;
            dst = keys_tmp_.as<int32_t>();
        }
        hipLaunchKernelGGL(k_pack_exchange, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, stream_, (int)res_B_, (int)per,
                           (int)res_unique_, A_, O_, inv_.as<int32_t>(), res_action_, keep_.as<uint8_t>(), uniq_.as<int32_t>(),
                           res_best_, dst);
        HIPCHK(hipGetLastError());
        if (!direct) HIPCHK(hipMemcpyAsync(out, dst, n * sizeof(int32_t), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        return PBVI_OK;
    }

    // alpha' rows from keys against the RESIDENT alpha set (K3 of src/pomdp.py:1497-1506 for given winners): what a
    // rank needs to rebuild the rows other ranks found, since the alpha set and the model are replicated.
    int assemble_keys(double gamma, int64_t n, const int32_t* keys, void* out_rows) override {
        if (V_ <= 0) FAIL(PBVI_EINVAL, "assemble_keys: no alpha set resident");
        if (n <= 0 || n > 65535 || !keys || !out_rows) FAIL(PBVI_EINVAL, "assemble_keys: bad arguments (1 <= n <= 65535)");
        HIPCHK(hipSetDevice(device_));
        int rc;
        if ((rc = keys_tmp_.ensure((size_t)n * (1 + O_) * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = keys_act_.ensure((size_t)(n + 1) * sizeof(int32_t), &bytes_))) return rc;      // [n] actions + 1 error counter
        if ((rc = keys_best_.ensure((size_t)n * A_ * O_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = keys_rows_.ensure((size_t)n * S_ * sizeof(T), &bytes_))) return rc;
        int* bad = keys_act_.as<int>() + n;
        HIPCHK(hipMemcpyAsync(keys_tmp_.p, keys, (size_t)n * (1 + O_) * sizeof(int32_t), hipMemcpyDefault, stream_));
        HIPCHK(hipMemsetAsync(bad, 0, sizeof(int), stream_));
        hipLaunchKernelGGL(k_scatter_keys, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream_, (int)n, A_, O_, (int)V_,
                           keys_tmp_.as<int32_t>(), keys_act_.as<int32_t>(), keys_best_.as<int32_t>(), bad);
        HIPCHK(hipGetLastError());
        HIPCHK(launch_assemble<T>(alpha_.as<T>(), S_pad_, view(), gamma, keys_act_.as<int32_t>(), keys_best_.as<int32_t>(), nullptr,
                                  nullptr, (int)n, keys_rows_.as<T>(), S_, stream_));
        int* h_bad = pinned_flag();
        if (!h_bad) FAIL(PBVI_ENOMEM, "assemble_keys: pinned flag");
        *h_bad = 0;
        HIPCHK(hipMemcpyAsync(h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipMemcpyAsync(out_rows, keys_rows_.p, (size_t)n * S_ * sizeof(T), hipMemcpyDefault, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (*h_bad) FAIL(PBVI_EINVAL, "assemble_keys: action or alpha index out of range");
        return PBVI_OK;
    }

    // Same, and the rows also join the alpha store (ids consecutive from the return value): the sharded backup's
    // "every replica appends the same rows" step, device to device.  out_rows may be NULL.
    int64_t assemble_keys_store(double gamma, int64_t n, const int32_t* keys, void* out_rows) override {
        if (V_ <= 0) FAIL(PBVI_EINVAL, "assemble_rows_store: no alpha set resident");
        if (n <= 0 || n > 65535 || !keys) FAIL(PBVI_EINVAL, "assemble_rows_store: bad arguments (1 <= n <= 65535)");
        HIPCHK(hipSetDevice(device_));
        int rc;
        if ((rc = keys_tmp_.ensure((size_t)n * (1 + O_) * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = keys_act_.ensure((size_t)(n + 1) * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = keys_best_.ensure((size_t)n * A_ * O_ * sizeof(int32_t), &bytes_))) return rc;
        T* dst = nullptr;
        if ((rc = store_reserve(0, n, &dst))) return rc;
        int* bad = keys_act_.as<int>() + n;
        HIPCHK(hipMemcpyAsync(keys_tmp_.p, keys, (size_t)n * (1 + O_) * sizeof(int32_t), hipMemcpyDefault, stream_));
        HIPCHK(hipMemsetAsync(bad, 0, sizeof(int), stream_));
        hipLaunchKernelGGL(k_scatter_keys, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream_, (int)n, A_, O_, (int)V_,
                           keys_tmp_.as<int32_t>(), keys_act_.as<int32_t>(), keys_best_.as<int32_t>(), bad);
        HIPCHK(hipGetLastError());
        if (S_pad_ > S_) HIPCHK(hipMemsetAsync(dst, 0, (size_t)n * S_pad_ * sizeof(T), stream_));   // pad columns stay zero
        HIPCHK(launch_assemble<T>(alpha_.as<T>(), S_pad_, view(), gamma, keys_act_.as<int32_t>(), keys_best_.as<int32_t>(), nullptr,
                                  nullptr, (int)n, dst, S_pad_, stream_));
        int* h_bad = pinned_flag();
        if (!h_bad) FAIL(PBVI_ENOMEM, "assemble_rows_store: pinned flag");
        *h_bad = 0;
        HIPCHK(hipMemcpyAsync(h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, stream_));
        if (out_rows) {
            if (is_device_pointer(out_rows) || is_pinned_host_pointer(out_rows)) {
                HIPCHK(hipMemcpy2DAsync(out_rows, (size_t)S_ * sizeof(T), dst, (size_t)S_pad_ * sizeof(T), (size_t)S_ * sizeof(T),
                                        (size_t)n, hipMemcpyDefault, stream_));
            } else {   // pageable destination: compact on the device, then through the pinned bounce buffer
                if ((rc = keys_rows_.ensure((size_t)n * S_ * sizeof(T), &bytes_))) return rc;
                HIPCHK(hipMemcpy2DAsync(keys_rows_.p, (size_t)S_ * sizeof(T), dst, (size_t)S_pad_ * sizeof(T), (size_t)S_ * sizeof(T),
                                        (size_t)n, hipMemcpyDeviceToDevice, stream_));
                if ((rc = out_begin())) return rc;
                if ((rc = out_add(out_rows, keys_rows_.p, (size_t)n * S_ * sizeof(T)))) return rc;
                if ((rc = out_finish())) return rc;
            }
        }
        HIPCHK(hipStreamSynchronize(stream_));
        if (*h_bad) FAIL(PBVI_EINVAL, "assemble_rows_store: action or alpha index out of range");
        const int64_t first = store_rows_[0];
        store_rows_[0] = first + n;
        return first;
    }

    int device_results(void** d_alpha, int32_t** d_action, uint8_t** d_keep) override {
        if (!have_result_) FAIL(PBVI_EINVAL, "no backup result resident");
        if (d_alpha) {
            int rc = ensure_full();
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(stream_));
            *d_alpha = out_full_.p;
        }
        if (d_action) *d_action = const_cast<int32_t*>(res_action_);
        if (d_keep) *d_keep = keep_.as<uint8_t>();
        return PBVI_OK;
    }

    int prune_dominated(uint8_t* keep) override {
        if (V_ <= 0) FAIL(PBVI_EINVAL, "prune_dominated: no alpha set resident");
        if (!keep) FAIL(PBVI_EINVAL, "prune_dominated: keep is NULL");
        HIPCHK(hipSetDevice(device_));
        int rc = prune_cnt_.ensure((size_t)V_ * sizeof(int), &bytes_);
        if (rc) return rc;
        HIPCHK(hipMemsetAsync(prune_cnt_.p, 0, (size_t)V_ * sizeof(int), stream_));
        HIPCHK(launch_dominated<T>(alpha_.as<T>(), S_pad_, (int)V_, S_, prune_cnt_.as<int>(), stream_));
        std::vector<int> cnt((size_t)V_);
        HIPCHK(hipMemcpyAsync(cnt.data(), prune_cnt_.p, (size_t)V_ * sizeof(int), hipMemcpyDeviceToHost, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        for (int64_t i = 0; i < V_; ++i) keep[i] = (cnt[(size_t)i] == 1) ? 1 : 0;
        return PBVI_OK;
    }

    // exact (f64-refined) max_v b.alpha_v of the resident blocks into bs2_/bv2_
    int value_max_device();

    int value_max(double* out_value, int32_t* out_index) override {
        if (V_ <= 0 || B_ <= 0) FAIL(PBVI_EINVAL, "value_max: need a resident alpha set and belief block");
        HIPCHK(hipSetDevice(device_));
        int rc = value_max_device();
        if (rc) return rc;
        std::vector<double> tv;
        std::vector<int32_t> ti;
        double* dv = out_value;
        int32_t* di = out_index;
        if (sorted_) {                      // device results are in sorted belief order
            if ((rc = host_perm())) return rc;
            tv.resize((size_t)B_);
            ti.resize((size_t)B_);
            dv = tv.data();
            di = ti.data();
        }
        if ((rc = out_begin())) return rc;
        if (out_value && (rc = out_add(dv, bs2_.p, (size_t)B_ * sizeof(double)))) return rc;
        if (out_index && (rc = out_add(di, bv2_.p, (size_t)B_ * sizeof(int32_t)))) return rc;
        if ((rc = out_finish())) return rc;
        if (sorted_)
            for (int64_t i = 0; i < B_; ++i) {
                if (out_value) out_value[h_perm_[(size_t)i]] = tv[(size_t)i];
                if (out_index) out_index[h_perm_[(size_t)i]] = ti[(size_t)i];
            }
        return PBVI_OK;
    }

    int64_t store_count(int which) const override { return (which == 0 || which == 1) ? store_rows_[which] : -1; }

    // grow a buffer to `need` bytes keeping its first `used` bytes
    int grow_keep(DevBuf& b, size_t need, size_t used) {
        if (need <= b.cap) return PBVI_OK;
        DevBuf nb;
        int rc = nb.ensure(std::max(need, b.cap * 2), &bytes_);
        if (rc) return rc;
        if (used > 0) HIPCHK(hipMemcpyAsync(nb.p, b.p, used, hipMemcpyDeviceToDevice, stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        bytes_ -= (int64_t)b.cap;
        b.release();
        b = nb;
        return PBVI_OK;
    }

    // max_v b.alpha_v (exact, like value_max) for rows [0, n) of the BELIEF STORE against the working alpha set, with the
    // store itself as the GEMM operand: no gather, no sort, and the zero maps / tile lists of rows seen before are reused.
    // compute_change (src/pomdp.py:2141-2169) scores the whole accumulated belief set after every backup; that set only
    // grows, and its rows already sit in the store in arrival order (a walk's successive beliefs overlap heavily, so
    // 256-row blocks of the store have tight joint support without sorting).
    int value_max_store(int64_t n, double* out_value, int32_t* out_index) override {
        if (V_ <= 0) FAIL(PBVI_EINVAL, "value_max_store: no alpha set resident");
        if (n <= 0 || n > store_rows_[1] || (!out_value && !out_index)) FAIL(PBVI_EINVAL, "value_max_store: bad arguments");
        HIPCHK(hipSetDevice(device_));
        int rc;
        const int k_tiles = S_pad_ / GEMM_BK;
        const int64_t have = store_rows_[1], have_pad = round_up(have, GEMM_BM);
        // rows [have, have_pad) are read by the last row block: make them exist and be zero
        if ((size_t)have_pad * S_pad_ * sizeof(T) > store_[1].cap) {
            T* unused = nullptr;
            if ((rc = store_reserve(1, have_pad - have, &unused))) return rc;
        }
        T* base = store_[1].as<T>();
        if (have_pad > have)
            HIPCHK(hipMemsetAsync(base + (size_t)have * S_pad_, 0, (size_t)(have_pad - have) * S_pad_ * sizeof(T), stream_));
        // zero maps: complete blocks computed earlier stay; the block that was partial then and everything after is (re)done
        if ((rc = grow_keep(snz_, (size_t)(have_pad / GEMM_BM) * k_tiles, (size_t)(snz_rows_ / GEMM_BM) * k_tiles))) return rc;
        {
            const int64_t r0 = snz_rows_ / GEMM_BM * GEMM_BM;
            if (have_pad > r0) {
                uint8_t* nz = snz_.as<uint8_t>() + (size_t)(r0 / GEMM_BM) * k_tiles;
                if constexpr (kF32)
                    HIPCHK(launch_tile_nonzero_f32((const float*)(base + (size_t)r0 * S_pad_), S_pad_, (int)(have_pad - r0), k_tiles, nz, stream_));
                else
                    HIPCHK(launch_tile_nonzero_f64((const double*)(base + (size_t)r0 * S_pad_), S_pad_, (int)(have_pad - r0), k_tiles, nz, stream_));
            }
            snz_rows_ = have;
        }
        if (kF32 && !vmax_skinny()) {   // per-row tile lists for the f64 refinement
            if ((rc = grow_keep(sbtl_, (size_t)have * k_tiles * sizeof(int32_t), (size_t)sbt_rows_ * k_tiles * sizeof(int32_t)))) return rc;
            if ((rc = grow_keep(sbtc_, (size_t)have * sizeof(int32_t), (size_t)sbt_rows_ * sizeof(int32_t)))) return rc;
            for (int64_t r0 = sbt_rows_; r0 < have; r0 += 32768) {
                const int cnt = (int)std::min<int64_t>(32768, have - r0);
                HIPCHK(launch_belief_tiles<T>(base + (size_t)r0 * S_pad_, S_pad_, cnt, S_, k_tiles,
                                              sbtl_.as<int32_t>() + (size_t)r0 * k_tiles, sbtc_.as<int32_t>() + r0, stream_));
            }
            sbt_rows_ = have;
        }
        // run value_max_device on windows of the store: the block members point into the store for the duration
        struct Saved {
            DevBuf bel, nzA, btl, btc;
            int64_t B, B_pad;
            bool sorted, btl_valid, have_result;
        } sv{bel_, nzA_, btl_, btc_, B_, B_pad_, sorted_, btl_valid_, have_result_};
        auto restore = [&]() {
            bel_ = sv.bel; nzA_ = sv.nzA; btl_ = sv.btl; btc_ = sv.btc;
            B_ = sv.B; B_pad_ = sv.B_pad; sorted_ = sv.sorted; btl_valid_ = sv.btl_valid; have_result_ = sv.have_result;
        };
        const size_t huge = ~(size_t)0 >> 1;              // aliased buffers must never look too small to ensure()
        const int64_t window = 32768;
        rc = PBVI_OK;
        for (int64_t r0 = 0; r0 < n && rc == PBVI_OK; r0 += window) {
            const int64_t cnt = std::min<int64_t>(window, n - r0);
            bel_.p = base + (size_t)r0 * S_pad_;                       bel_.cap = huge;
            nzA_.p = snz_.as<uint8_t>() + (size_t)(r0 / GEMM_BM) * k_tiles;   nzA_.cap = huge;
            if (kF32 && !vmax_skinny()) {
                btl_.p = sbtl_.as<int32_t>() + (size_t)r0 * k_tiles;   btl_.cap = huge;
                btc_.p = sbtc_.as<int32_t>() + r0;                     btc_.cap = huge;
            }
            B_ = cnt;
            B_pad_ = round_up(cnt, GEMM_BM);
            sorted_ = false;
            btl_valid_ = kF32 && !vmax_skinny();
            rc = value_max_device();
            if (rc == PBVI_OK) rc = out_begin();
            if (rc == PBVI_OK && out_value) rc = out_add(out_value + r0, bs2_.p, (size_t)cnt * sizeof(double));
            if (rc == PBVI_OK && out_index) rc = out_add(out_index + r0, bv2_.p, (size_t)cnt * sizeof(int32_t));
            if (rc == PBVI_OK) rc = out_finish();
        }
        if (rc != PBVI_OK) (void)hipStreamSynchronize(stream_);
        restore();
        return rc;
    }

    // Work list of the f64 refinement (see RefineWork): room for every candidate of up to 16M (entry, alpha) pairs
    // and the tile lists of up to 65536 entries; what does not fit is scored by the entry's own block.
    int refine_work(int64_t max_entries, int64_t V, RefineWork* w) {
        static const bool off = getenv("PBVI_REFINE_INBLOCK") != nullptr;     // debug / A-B only
        *w = RefineWork{};
        if (off) return PBVI_OK;
        const int k_tiles = S_pad_ / GEMM_BK;
        // capacities move in powers of two: the alpha set grows by a few rows per backup in a solve loop and a
        // reallocation (hipFree + hipMalloc, both synchronising) per call would cost more than the refinement
        auto pow2 = [](int64_t x, int64_t lo, int64_t hi) {
            int64_t p = lo;
            while (p < x && p < hi) p <<= 1;
            return p;
        };
        int64_t items = pow2(max_entries * V, (int64_t)1 << 16, (int64_t)1 << 24);
        int64_t slots = pow2(max_entries, 1024, 65536);
        if (const char* c = getenv("PBVI_REFINE_ITEM_CAP")) items = std::max<int64_t>(1, atoll(c));   // tests: force the
        if (const char* c = getenv("PBVI_REFINE_SLOT_CAP")) slots = std::max<int64_t>(1, atoll(c));   // overflow paths
        if (items <= 0 || slots <= 0) return PBVI_OK;
        int rc;
        if ((rc = rf_v_.ensure((size_t)items * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = rf_slot_.ensure((size_t)items * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = rf_sc_.ensure((size_t)items * sizeof(double), &bytes_))) return rc;
        if ((rc = rf_entry_.ensure((size_t)slots * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = rf_n_.ensure((size_t)slots * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = rf_tiles_.ensure((size_t)slots * k_tiles * sizeof(int32_t), &bytes_))) return rc;
        const size_t zero_bytes = 16 + (size_t)slots * (sizeof(unsigned long long) + sizeof(int32_t));
        if ((rc = rf_cnt_.ensure(zero_bytes, &bytes_))) return rc;      // [cnt: 16 B][emax][eidx]
        if ((rc = rf_ibv_.ensure((size_t)slots * sizeof(double), &bytes_))) return rc;
        if ((rc = rf_ibi_.ensure((size_t)slots * sizeof(int32_t), &bytes_))) return rc;
        w->items_v = rf_v_.as<int32_t>();
        w->items_slot = rf_slot_.as<int32_t>();
        w->scores = rf_sc_.as<double>();
        w->slot_entry = rf_entry_.as<int32_t>();
        w->slot_n = rf_n_.as<int32_t>();
        w->tiles = rf_tiles_.as<int32_t>();
        w->emax = reinterpret_cast<unsigned long long*>(rf_cnt_.as<char>() + 16);
        w->eidx = reinterpret_cast<int32_t*>(rf_cnt_.as<char>() + 16 + (size_t)slots * sizeof(unsigned long long));
        w->zero_bytes = zero_bytes;
        w->ib_val = rf_ibv_.as<double>();
        w->ib_idx = rf_ibi_.as<int32_t>();
        w->cnt = rf_cnt_.as<int>();
        {   // hand-over to k_refine_split (entries the level-1 screen leaves undecided; ~1 % of the queue)
            int64_t q2 = std::min<int64_t>(max_entries, 16384);
            if (const char* c = getenv("PBVI_REFINE_Q2_CAP")) q2 = std::max<int64_t>(1, std::min<int64_t>(q2, atoll(c)));   // tests: force the hand-over's overflow path
            const size_t need = (size_t)q2 * (2 + 8) * sizeof(int32_t);
            if ((rc = rf_q2_.ensure(need, &bytes_))) return rc;
            if ((rc = rf_q2p_.ensure((size_t)q2 * 16 * 8 * sizeof(double), &bytes_))) return rc;
            if (rf_q2d_.cap < (size_t)q2 * sizeof(int)) {
                if ((rc = rf_q2d_.ensure((size_t)q2 * sizeof(int), &bytes_))) return rc;
                HIPCHK(hipMemsetAsync(rf_q2d_.p, 0, rf_q2d_.cap, stream_));      // arrival counters: zero between launches
            }
            w->q2_entry = rf_q2_.as<int32_t>();
            w->q2_n = rf_q2_.as<int32_t>() + q2;
            w->q2_cand = rf_q2_.as<int32_t>() + 2 * q2;
            w->q2_part = rf_q2p_.as<double>();
            w->q2_done = rf_q2d_.as<int>();
            w->q2_cap = (int)q2;
        }
        w->item_cap = (int)items;
        w->slot_cap = (int)slots;
        // GEMM path of the tie-heavy entries: fp64 weight rows for as many slots as ~2 GiB holds
        int64_t w_slots = std::min<int64_t>(std::min<int64_t>(slots, 65535),                       // grid.y of the weights kernel
                                            ((int64_t)2 << 30) / ((int64_t)S_pad_ * sizeof(double)));
        if (const char* c = getenv("PBVI_REFINE_DEFER_MIN")) w->defer_min = atoi(c);                 // A/B: -1 = every candidate to the grid-wide pass
        if (const char* c = getenv("PBVI_REFINE_W_SLOTS")) w_slots = std::min<int64_t>(slots, std::max<int64_t>(0, atoll(c)));   // tests
        if (w_slots > 0) {
            if ((rc = build_inverse_lists())) return rc;
            const int kt32 = S_pad_ / GEMM_BK;
            if ((rc = rf_W_.ensure((size_t)w_slots * S_pad_ * sizeof(double), &bytes_))) return rc;
            if ((rc = rf_Cx_.ensure((size_t)w_slots * V * sizeof(double), &bytes_))) return rc;
            if ((rc = rf_nzW_.ensure((size_t)((w_slots + 255) / 256) * kt32, &bytes_))) return rc;
            if ((rc = rf_klW_.ensure(gemm_f64_klist_ints((int)w_slots, (int)V, kt32) * sizeof(int), &bytes_))) return rc;
            if ((rc = rf_kcW_.ensure(gemm_f64_kcount_ints((int)w_slots, (int)V, kt32) * sizeof(int), &bytes_))) return rc;
            w->W = rf_W_.as<double>();
            w->Cx = rf_Cx_.as<double>();
            w->nzW = rf_nzW_.as<uint8_t>();
            w->klistW = rf_klW_.as<int>();
            w->kcountW = rf_kcW_.as<int>();
            w->in_ptr = in_ptr_.as<int32_t>();
            w->in_src = in_src_.as<int32_t>();
        }
        w->w_slot_cap = (int)w_slots;
        return PBVI_OK;
    }

    // fp32 engines: value-max GEMMs against at most 64 alpha rows take the skinny fp64-accumulating tile (value_max_device)
    bool vmax_skinny() const {
        static const bool off = getenv("PBVI_NO_SKINNY") != nullptr;      // debug / A-B only
        return kF32 && !off && V_ > 0 && V_ <= 64 && mode_ == PBVI_SPARSE;
    }
    // fp64 engines: MFMA GEMM unless the problem is a handful of tiles (the plain kernel is as good there)
    static bool f64_uses_mfma(int64_t m_rows, int64_t n_rows) {
        static const bool simple = getenv("PBVI_F64_SIMPLE") != nullptr;      // debug / A-B only
        return !simple && m_rows * n_rows >= 64 * 64;
    }

    int set_fused(int enable) override {
        fuse_project_ = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
        if (screen_) screen_->fuse_project_ = fuse_project_;
        return PBVI_OK;
    }
    int set_screen(int mode) override {
        if (mode < 0 || mode > 2) FAIL(PBVI_EINVAL, "set_f64_screen: 0 = never, 1 = automatic, 2 = always");
        screen_mode_ = mode;
        return PBVI_OK;
    }
    int set_formulation(int f) override {
        if (f < 0 || f > 2) FAIL(PBVI_EINVAL, "set_formulation: 0 = auto, 1 = project alpha-vectors, 2 = project beliefs");
        formulation_ = f;
        return PBVI_OK;
    }
    // f32 engines: value_max / value_max_store return the fp32 GEMM's maxima as they are (relative error of the order
    // of 1e-7, bounded by the tie window) instead of re-scoring every belief's candidates in fp64.  For callers that
    // only compare values against a tolerance (compute_change); argmax users keep the default.
    int set_value_max_exact(int exact) override {
        vmax_exact_ = exact != 0;
        return PBVI_OK;
    }
    int alpha_layout(int64_t* free_rows, int64_t* layouts) override {
        if (free_rows) *free_rows = alpha_on_primary_ ? prim_off_ : 0;
        if (layouts) *layouts = (int64_t)prim_layout_ver_;
        return alpha_on_primary_ ? 1 : 0;
    }

    int set_tie_window(double rel) override {
        tie_rel_user_ = rel;
        return PBVI_OK;
    }
    int64_t device_bytes() const override { return bytes_ + (screen_ ? screen_->bytes_ : 0); }
};

template <typename T>
int EngineT<T>::score_gemm(const T* Y, int64_t rows_y, const uint8_t* nzB, int G, int v_group, SlabView<T>* sv,
                           const T* X, int64_t x_rows, const uint8_t* nzX, hipStream_t list_stream, const FusedB* fused) {
    int rc;
    const int64_t m_rows = X ? x_rows : B_;
    const int64_t m_pad = X ? round_up(x_rows, GEMM_BM) : B_pad_;
    if (!X) {
        X = bel_.as<T>();
        nzX = nzA_.as<uint8_t>();
    }
    if constexpr (kF32) {
        const int64_t n_pad = round_up(rows_y, GEMM_BN);
        if (m_pad > 0x7fffffff || n_pad > 0x7fffffff) FAIL(PBVI_EUNSUPPORTED, "score_gemm: operand rows exceed int32");
        plan_ = make_gemm_plan((int)m_pad, (int)n_pad, S_pad_);
        const size_t pairs = (size_t)plan_.tiles_m * plan_.tiles_n;
        if ((rc = slabs_.ensure((size_t)plan_.c_floats * sizeof(float), &bytes_))) return rc;
        if ((rc = klist_.ensure(pairs * plan_.k_tiles * sizeof(int), &bytes_))) return rc;
        if ((rc = kcount_.ensure(pairs * sizeof(int), &bytes_))) return rc;
        if ((rc = nchunks_.ensure(pairs * sizeof(int), &bytes_))) return rc;
        if ((rc = skws_.ensure(streamk_workspace_ints(plan_) * sizeof(int), &bytes_))) return rc;
        HIPCHK(launch_gemm_nt_f32((const float*)X, S_pad_, (const float*)Y, S_pad_, slabs_.as<float>(), plan_,
                                  nzX, nzB, G, v_group, (int)rows_y, klist_.as<int>(), kcount_.as<int>(),
                                  nchunks_.as<int>(), stream_, 1, 0, 0, skws_.as<int>(), list_stream, ev_lists_, nullptr, nullptr,
                                  fused));
        sv->slabs = slabs_.as<T>();
        sv->slab_stride = plan_.slab_stride;
        sv->ldc = plan_.ldc;
        sv->nchunks = nchunks_.as<int>();
        sv->tiles_m = plan_.tiles_m;
        sv->fixed = 0;
        // (the stream-K plan's first_block array: launch_gemm_nt_f32 lays the workspace out as prefix[pairs + 1],
        // start_pair[nblocks], first_block[pairs], plan[2])
        sv->first_block = plan_.streamk ? skws_.as<int>() + (pairs + 1) + plan_.nblocks : nullptr;
    } else {
        const int kt32s = S_pad_ / GEMM_BK;
        const int split = f64_uses_mfma(m_rows, rows_y) ? gemm_f64_split((int)m_rows, (int)rows_y, kt32s) : 1;
        if ((rc = slabs_.ensure((size_t)split * m_rows * rows_y * sizeof(T), &bytes_))) return rc;
        f64_pairs_ = 0;
        if (!f64_uses_mfma(m_rows, rows_y)) {
            HIPCHK(launch_gemm_nt_simple<T>(X, S_pad_, Y, S_pad_, slabs_.as<T>(), (int)rows_y, (int)m_rows, (int)rows_y,
                                            S_, stream_));
        } else {
            const int kt32 = S_pad_ / GEMM_BK;
            if ((rc = klist_.ensure(gemm_f64_klist_ints((int)m_rows, (int)rows_y, kt32) * sizeof(int), &bytes_))) return rc;
            if ((rc = kcount_.ensure(gemm_f64_kcount_ints((int)m_rows, (int)rows_y, kt32) * sizeof(int), &bytes_))) return rc;
            HIPCHK(launch_gemm_nt_f64((const double*)X, S_pad_, (int)m_rows, (const double*)Y, S_pad_, (int)rows_y,
                                      slabs_.as<double>(), (int)rows_y, S_pad_, nzX, nzB, G, v_group, klist_.as<int>(),
                                      kcount_.as<int>(), stream_, split, m_rows * rows_y));
            f64_pairs_ = (int64_t)gemm_f64_pairs((int)m_rows, (int)rows_y);
        }
        sv->slabs = slabs_.as<T>();
        sv->slab_stride = 0;                  // the K parts were folded into the first slab by the launcher
        sv->ldc = (int)rows_y;
        sv->nchunks = nullptr;
        sv->tiles_m = 0;
        sv->fixed = 1;
    }
    return PBVI_OK;
}

template <typename T>
int EngineT<T>::project_dense(double gamma) {
    int rc;
    const int AO = A_ * O_;
    const int64_t Vt = V_, Vt_pad = round_up(Vt, GEMM_BM);   // the magnitude row is projected separately below
    int64_t ld_prod, batch_stride;
    if constexpr (kF32) {
        const int64_t n_pad = rows_pad_s_;
        GemmPlan pl = make_gemm_plan((int)Vt_pad, (int)n_pad, S_pad_, /*single_chunk=*/true);
        const size_t pairs = (size_t)pl.tiles_m * pl.tiles_n;
        ld_prod = n_pad;
        batch_stride = Vt_pad * n_pad;
        if ((rc = prod_.ensure((size_t)AO * batch_stride * sizeof(float), &bytes_))) return rc;
        if ((rc = nzAlpha_.ensure((size_t)pl.tiles_m * pl.k_tiles, &bytes_))) return rc;
        if ((rc = klistD_.ensure((size_t)AO * pairs * pl.k_tiles * sizeof(int), &bytes_))) return rc;
        if ((rc = kcountD_.ensure((size_t)AO * pairs * sizeof(int), &bytes_))) return rc;
        if ((rc = nchunksD_.ensure((size_t)AO * pairs * sizeof(int), &bytes_))) return rc;
        dense_pairs_ = (int64_t)AO * pairs;
        HIPCHK(hipMemsetAsync(prod_.p, 0, (size_t)AO * batch_stride * sizeof(float), stream_));   // pairs with empty lists
        HIPCHK(launch_tile_nonzero_f32((const float*)alpha_.p, S_pad_, (int)Vt_pad, pl.k_tiles, nzAlpha_.as<uint8_t>(), stream_));
        HIPCHK(launch_gemm_nt_f32((const float*)alpha_.p, S_pad_, (const float*)dense_.p, S_pad_, prod_.as<float>(), pl,
                                  nzAlpha_.as<uint8_t>(), nzD_.as<uint8_t>(), 0, 0, S_, klistD_.as<int>(),
                                  kcountD_.as<int>(), nchunksD_.as<int>(), stream_, AO, rows_pad_s_ * (int64_t)S_pad_,
                                  batch_stride, nullptr, nullptr, nullptr, ev_pg_[0], ev_pg_[1]));
    } else {
        ld_prod = S_;
        batch_stride = Vt * S_;
        if ((rc = prod_.ensure((size_t)AO * batch_stride * sizeof(T), &bytes_))) return rc;
        for (int ao = 0; ao < AO; ++ao)
            HIPCHK(launch_gemm_nt_simple<T>(alpha_.as<T>(), S_pad_, dense_.as<T>() + (size_t)ao * rows_pad_s_ * S_pad_, S_pad_,
                                            prod_.as<T>() + (size_t)ao * batch_stride, (int)ld_prod, (int)Vt, S_, S_, stream_));
    }
    dim3 grid((S_ + 255) / 256, (unsigned)Vt, AO);
    hipLaunchKernelGGL(k_scale_rows<T>, grid, dim3(256), 0, stream_, prod_.as<T>(), batch_stride, (int)ld_prod, (int)V_,
                       (T)gamma, gam_.as<T>(), S_pad_, S_);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(k_project_mag<T>, dim3((S_pad_ + 255) / 256, AO), dim3(256), 0, stream_,
                       alpha_.as<T>() + (size_t)V_ * S_pad_, view(), (T)gamma, gam_.as<T>() + (size_t)AO * V_ * S_pad_, S_pad_);
    HIPCHK(hipGetLastError());
    return PBVI_OK;
}

template <typename T>
int EngineT<T>::value_max_device() {
    int rc;
    const int64_t Vt = V_ + 1;   // + magnitude row
    if ((rc = bv2_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = bs2_.ensure((size_t)B_ * sizeof(double), &bytes_))) return rc;
    if ((rc = err2_.ensure((size_t)B_ * sizeof(double), &bytes_))) return rc;
    if ((rc = queue2_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    int* qc = counters_.as<int>() + 2;
    HIPCHK(hipMemsetAsync(qc, 0, sizeof(int), stream_));
    if constexpr (kF32) {
        if (vmax_skinny()) {
            // A few dozen alpha rows (compute_change: every known belief against the rows an expansion added): on the
            // 256 x 256 tile of the fp32 engine nine tenths of the MFMA work would be padding and the launch MFMA-bound on
            // it.  The fp64 tile engine's 32- / 64-column variant with both operands widened on the way into LDS streams
            // the belief block once instead -- and its scores are exact products summed in fp64, so there is no tie window
            // and nothing to re-score.
            const int kt32 = S_pad_ / GEMM_BK;
            const int split = gemm_f64_split((int)B_, (int)V_, kt32);
            const int64_t slab = B_ * V_;
            if ((rc = slabs_.ensure((size_t)split * slab * sizeof(double), &bytes_))) return rc;
            if ((rc = klist_.ensure(gemm_f64_klist_ints((int)B_, (int)V_, kt32) * sizeof(int), &bytes_))) return rc;
            if ((rc = kcount_.ensure(gemm_f64_kcount_ints((int)B_, (int)V_, kt32) * sizeof(int), &bytes_))) return rc;
            HIPCHK(launch_gemm_nt_f64_ff32((const float*)bel_.p, S_pad_, (int)B_, (const float*)alpha_.p, S_pad_, (int)V_,
                                           slabs_.as<double>(), (int)V_, S_pad_, nzA_.as<uint8_t>(), klist_.as<int>(),
                                           kcount_.as<int>(), stream_, split, slab));
            SlabView<double> svd;
            svd.slabs = slabs_.as<double>();
            svd.slab_stride = 0;              // K parts folded into the first slab by the launcher
            svd.ldc = (int)V_;
            svd.nchunks = nullptr;
            svd.tiles_m = 0;
            svd.fixed = 1;
            HIPCHK(launch_argmax<double>(svd, (int)V_, 1, (int)B_, nullptr, 0.0, 0.0, nullptr, 0, bv2_.as<int32_t>(),
                                         bs2_.as<double>(), err2_.as<double>(), nullptr, qc, stream_));
            return PBVI_OK;
        }
    }
    SlabView<T> sv;
    if ((rc = score_gemm(alpha_.as<T>(), Vt, nullptr, 1, (int)V_, &sv))) return rc;
    const int k_chunk = kF32 ? plan_.chunk_len * GEMM_BK : S_pad_;
    // every belief is re-scored exactly in f32 engines (flag_all): the comparison that
    // follows (new value > old best value) is strict and must not see GEMM rounding
    const bool rescore = kF32 && vmax_exact_;
    HIPCHK(launch_argmax<T>(sv, (int)V_, 1, (int)B_, nullptr, tie_window(k_chunk), 0.0, chain_steps(), 1, bv2_.as<int32_t>(),
                            bs2_.as<double>(), err2_.as<double>(), rescore ? queue2_.as<int32_t>() : nullptr, qc, stream_));
    if (rescore) {
        if (!btl_valid_) {   // tile lists of the resident block (the backup's k_dead builds them too)
            const int k_tiles = S_pad_ / GEMM_BK;
            if ((rc = btl_.ensure((size_t)B_ * k_tiles * sizeof(int32_t), &bytes_))) return rc;
            if ((rc = btc_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
            HIPCHK(launch_belief_tiles<T>(bel_.as<T>(), S_pad_, (int)B_, S_, k_tiles, btl_.as<int32_t>(), btc_.as<int32_t>(),
                                          stream_));
            btl_valid_ = true;
        }
        RefineWork work;
        if ((rc = refine_work(B_, V_, &work))) return rc;
        HIPCHK(launch_refine<T>(false, sv, (int)V_, 1, (int)B_, queue2_.as<int32_t>(), qc, bel_.as<T>(), S_pad_,
                                alpha_.as<T>(), S_pad_, view(), 0.0, btl_.as<int32_t>(), btc_.as<int32_t>(), nullptr,
                                bv2_.as<int32_t>(), bs2_.as<double>(), err2_.as<double>(), nullptr, work, stream_));
    }
    return PBVI_OK;
}

// fp64 -> fp32 copy of operand rows for the screen (round to nearest even, like NumPy's astype)
// *overflow (may be null) is set when a finite value leaves the fp32 range: the screen's scores would be inf / NaN
__global__ void k_narrow(const double* __restrict__ src, float* __restrict__ dst, int64_t n, int* __restrict__ overflow) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    int bad = 0;
    if (i + 3 < n) {
        const double2 a = *(const double2*)(src + i), b = *(const double2*)(src + i + 2);
        const float4 f = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
        *(float4*)(dst + i) = f;
        bad = (isinf(f.x) && !isinf(a.x)) || (isinf(f.y) && !isinf(a.y)) || (isinf(f.z) && !isinf(b.x)) || (isinf(f.w) && !isinf(b.y));
    } else {
        for (int64_t j = i; j < n; ++j) {
            const float f = (float)src[j];
            dst[j] = f;
            bad |= isinf(f) && !isinf(src[j]);
        }
    }
    if (bad && overflow != nullptr) atomicOr(overflow, 1);
}

template <typename T>
bool EngineT<T>::choose_push(int64_t N) const {
    // Which operand is projected through the model (same scores, re-associated):
    //   alpha-side (the reference's order): Gamma = A*O*(V+1)+2A rows, GEMM  [B] x [Gamma rows]
    //   belief-side: bp = B*A*O rows,                                  GEMM  [B*A*O] x [V]
    // The belief side wins when B << V (the solve loop: ~100 new beliefs against thousands of alpha-vectors):
    // fewer rows to project and far less tile padding.  Sparse mode only.
    const int AO = A_ * O_;
    if (mode_ == PBVI_DENSE || (int64_t)B_ * AO > 0x7fffffff || !(kF32 || f64_uses_mfma(B_ * AO, V_))) return false;
    const int64_t bm = kF32 ? GEMM_BM : 128;                       // GEMM tile edge and sustained rate per dtype
    const double rate = kF32 ? 130e12 : 45e12;
    // projection cost per row, measured at C4: the alpha side writes only the Gamma tiles the GEMM will read
    // (0.34 ms for 18450 rows), the belief side writes every row in full and re-reads the inverse lists per
    // belief (2.1 ms for 18432 rows)
    auto cost = [&](int64_t m_rows, int64_t n_rows, int64_t proj_rows, double proj_rate) {
        const double tiles = (double)((m_rows + bm - 1) / bm) * (double)((n_rows + bm - 1) / bm);
        return tiles * S_pad_ * (2.0 * bm * bm / rate) + (double)proj_rows * S_pad_ * sizeof(T) / proj_rate;
    };
    const double c_pull = cost(B_, N, N, 6e12), c_push = cost(B_ * AO, V_, B_ * AO, 1e12);
    if (formulation_ != 0) return formulation_ == 2;
    // Memory before speed: Gamma[A,O,V,S] is what the reference's CuPy path dies allocating (Sea_Robin_Real.ipynb:913,
    // 21.95 GB at |V| = 1386).  Where this engine would have to hold all of it (no fused generation: R > 1, fp64 scoring)
    // and it does not fit beside what the engine already holds while the projected beliefs do, the belief side is taken
    // whatever the cost model says -- its footprint grows with B, which the caller can cut (PBVI_Solver.backup halves the
    // belief block on MemoryError), not with V.
    const bool gamma_compact = kF32 && R_ == 1 && fuse_project_ >= 1 && irr_.p != nullptr;
    const int64_t row = (int64_t)S_pad_ * (int64_t)sizeof(T);
    const int64_t n_pull = kF32 ? round_up(N, GEMM_BN) : N, m_push = round_up((int64_t)B_ * AO + B_, GEMM_BM);
    const int64_t gam_new = gamma_compact ? 0 : std::max<int64_t>(0, n_pull * row - (int64_t)gam_.cap);
    if (gam_new > ((int64_t)1 << 28)) {
        // the score slabs count too
        auto slabs = [&](int64_t m, int64_t n) {
            if constexpr (kF32) return make_gemm_plan((int)round_up(m, GEMM_BM), (int)round_up(n, GEMM_BN), S_pad_).c_floats * 4;
            else return (int64_t)gemm_f64_split((int)m, (int)n, S_pad_ / GEMM_BK) * m * n * 8;
        };
        const int64_t pull = gam_new + std::max<int64_t>(0, slabs(B_, N) - (int64_t)slabs_.cap);
        const int64_t push = std::max<int64_t>(0, m_push * row - (int64_t)bp_.cap) +
                             std::max<int64_t>(0, slabs((int64_t)B_ * AO + B_, V_) - (int64_t)slabs_.cap);
        if (push < pull && pull > room()) return true;
    }
    return c_push < 0.8 * c_pull;
}

// K1 + K2 + first-max: projection, score GEMM and the argmax over alpha-vectors with near-tie detection, of THIS
// engine's resident alpha set and belief block (its element type T), into the pipeline buffers `io` names.  Run on
// the engine itself, or -- by an fp64 engine -- on its fp32 screen.
template <typename T>
int EngineT<T>::stage_scores(double gamma, bool use_push, const ScoreIO& io, ScoreStage<T>* out) {
    int rc;
    const int AO = A_ * O_;
    const int64_t Vt = V_ + 1;                 // alpha rows + magnitude row
    const int64_t N = (int64_t)AO * Vt + 2 * A_;   // Gamma rows: alpha groups, A*O magnitude rows, A reward + A |reward| rows
    const int64_t pairs = B_ * AO;
    const ModelView<T> mv = view();
    const int k_tiles = S_pad_ / GEMM_BK;
    const int64_t n_rows_alloc = kF32 ? round_up(N, GEMM_BN) : N;
    // One reachable state per (s, a) -- every large model of the reference -- on an fp32 engine: the score GEMM generates
    // the Gamma tiles that lie inside one (a, o) group itself (gemm.hip, scheduler 2b); only the tiles that straddle two
    // groups and the tail tile are projected ("mat" tiles).  Then only THOSE tiles exist in memory: tile tn of Gamma
    // lives at compact tile ctile[tn] (C4: one tile, 30 MB, instead of 2.24 GB; the reference's CuPy path died
    // allocating Gamma[A,O,V,S], Sea_Robin_Real.ipynb:913).
    bool will_fuse = false;
    if constexpr (kF32) {
        static const bool no_fuse = getenv("PBVI_NO_FUSED_PROJECT") != nullptr;     // debug / A-B only
        will_fuse = !use_push && mode_ == PBVI_SPARSE && !no_fuse && irr_.p != nullptr &&
                    (R_ == 1 ? fuse_project_ >= 1 : (R_ <= 7 && fuse_project_ >= 2));
    }
    static const bool no_compact = getenv("PBVI_NO_COMPACT_GAMMA") != nullptr;      // debug / A-B only
    const bool compact = will_fuse && R_ == 1 && !no_compact;
    if (will_fuse) {
        const int tiles_n = (int)(n_rows_alloc / GEMM_BN);
        if (mat_V_ != V_ || (int)h_mat_.size() != tiles_n) {
            h_mat_.assign((size_t)tiles_n, 0);
            h_ctile_.assign((size_t)tiles_n, -1);
            n_mat_ = 0;
            for (int tn = 0; tn < tiles_n; ++tn) {
                const int64_t r0 = (int64_t)tn * 256, r1 = r0 + 255;
                h_mat_[(size_t)tn] = (r1 >= (int64_t)AO * V_ || r0 / V_ != r1 / V_) ? 1 : 0;
                if (h_mat_[(size_t)tn]) h_ctile_[(size_t)tn] = n_mat_++;
            }
            // the 4-row blocks of alpha rows the projection still has to visit: those with a row in a
            // projected tile for some group, and the block that holds the magnitude row
            h_vlist_.clear();
            for (int64_t vb = 0; vb * 4 < Vt; ++vb) {
                bool hit = vb * 4 + 4 > V_;
                for (int ao = 0; ao < AO && !hit; ++ao)
                    for (int64_t v = vb * 4; v < vb * 4 + 4 && v < V_ && !hit; ++v) hit = h_mat_[(size_t)((ao * V_ + v) >> 8)] != 0;
                if (hit) h_vlist_.push_back((int)vb);
            }
            if ((rc = mat_.ensure((size_t)tiles_n, &bytes_))) return rc;
            if ((rc = ctile_.ensure((size_t)tiles_n * sizeof(int32_t), &bytes_))) return rc;
            if ((rc = vlist_.ensure(h_vlist_.size() * sizeof(int), &bytes_))) return rc;
            HIPCHK(hipMemcpyAsync(mat_.p, h_mat_.data(), (size_t)tiles_n, hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemcpyAsync(ctile_.p, h_ctile_.data(), (size_t)tiles_n * sizeof(int32_t), hipMemcpyHostToDevice, stream_));
            HIPCHK(hipMemcpyAsync(vlist_.p, h_vlist_.data(), h_vlist_.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
            mat_V_ = V_;
            gam_pad_ptr_ = nullptr;                          // the pad rows sit elsewhere now
        }
    }
    if (!use_push) {
        const int64_t rows_held = compact ? (int64_t)n_mat_ * GEMM_BN : n_rows_alloc;
        if ((rc = gam_.ensure((size_t)rows_held * S_pad_ * sizeof(T), &bytes_))) return rc;
        // zero the Gamma pad rows the GEMM tiles read -- once per (buffer, row count, layout): nothing writes rows >= N,
        // and the fill of up to 255 rows (27 MB, 39 us at |S| = 30000) sat in front of every backup's score GEMM
        if (n_rows_alloc > N && (gam_pad_ptr_ != gam_.p || gam_pad_N_ != N || gam_pad_compact_ != compact || mode_ == PBVI_DENSE)) {
            // (the pad rows are the end of the last tile, which is always a projected one: contiguous in either layout)
            const int64_t first = compact ? (int64_t)(n_mat_ - 1) * GEMM_BN + (N - (n_rows_alloc - GEMM_BN)) : N;
            HIPCHK(hipMemsetAsync(gam_.as<T>() + (size_t)first * S_pad_, 0, (size_t)(n_rows_alloc - N) * S_pad_ * sizeof(T), stream_));
            gam_pad_ptr_ = gam_.p;
            gam_pad_N_ = N;
            gam_pad_compact_ = compact;
        }
    }
    const int32_t* ctile = compact ? ctile_.as<int32_t>() : nullptr;
    SlabView<T> sv;
    out->extra_row0 = -1;
    out->fused = false;
    if (use_push) {
        // K1 (belief side): every belief through every (a, o); K2: [B*A*O] x [V]
        if ((rc = build_inverse_lists())) return rc;
        // B extra rows behind the projected ones: the beliefs themselves, so the same GEMM also yields b . alpha_v --
        // compute_change's max_v b.alpha_v of these beliefs against this alpha set (pbvi_backup_fetch_value_max)
        // rides in the M padding (1800 + 100 rows of 2048 in a solve loop).
        const int64_t M = (int64_t)AO * B_, Mx = M + B_, M_pad = round_up(Mx, GEMM_BM);
        if ((rc = bp_.ensure((size_t)M_pad * S_pad_ * sizeof(T), &bytes_))) return rc;
        if ((rc = pmag_.ensure((size_t)pairs * sizeof(double), &bytes_))) return rc;
        if ((rc = nzP_.ensure((size_t)(M_pad / GEMM_BM) * k_tiles, &bytes_))) return rc;
        HIPCHK(hipMemsetAsync(pmag_.p, 0, (size_t)pairs * sizeof(double), stream_));
        HIPCHK(hipMemcpyAsync(bp_.as<T>() + (size_t)M * S_pad_, bel_.p, (size_t)B_ * S_pad_ * sizeof(T), hipMemcpyDeviceToDevice, stream_));
        if (M_pad > Mx)
            HIPCHK(hipMemsetAsync(bp_.as<T>() + (size_t)Mx * S_pad_, 0, (size_t)(M_pad - Mx) * S_pad_ * sizeof(T), stream_));
        // the GEMM's zero-tile map of bp: the projection marks the tiles of its own rows while it writes them; only the row
        // tiles that also hold the belief rows / the pad are scanned afterwards (a pass over all of bp was 0.18 ms at the
        // Sea-Robin shape)
        HIPCHK(hipMemsetAsync(nzP_.p, 0, (size_t)(M_pad / GEMM_BM) * k_tiles, stream_));
        HIPCHK(launch_push_project<T>(bel_.as<T>(), S_pad_, (int)B_, mv, in_ptr_.as<int32_t>(), in_src_.as<int32_t>(), gamma,
                                      alpha_.as<T>() + (size_t)V_ * S_pad_, bp_.as<T>(), S_pad_, pmag_.as<double>(), stream_,
                                      nzP_.as<uint8_t>()));
        {
            const int64_t t0 = M / GEMM_BM;                  // first row tile with a row that is not a projected one
            const T* from = bp_.as<T>() + (size_t)t0 * GEMM_BM * S_pad_;
            uint8_t* to = nzP_.as<uint8_t>() + (size_t)t0 * k_tiles;
            if constexpr (kF32)
                HIPCHK(launch_tile_nonzero_f32((const float*)from, S_pad_, (int)(M_pad - t0 * GEMM_BM), k_tiles, to, stream_));
            else
                HIPCHK(launch_tile_nonzero_f64((const double*)from, S_pad_, (int)(M_pad - t0 * GEMM_BM), k_tiles, to, stream_));
        }
        HIPCHK(hipEventRecord(io.ev[1], stream_));
        if ((rc = score_gemm(alpha_.as<T>(), V_, nullptr, 1, (int)V_, &sv, bp_.as<T>(), Mx, nzP_.as<uint8_t>()))) return rc;
        out->extra_row0 = M;
        sv.push = 1;
        sv.push_B = (int)B_;
        sv.push_A = A_;
        sv.push_O = O_;
        sv.aux_mag = pmag_.as<double>();
        sv.aux_rd = io.prd;
    } else {
        // K1: Gamma projection of the V alpha rows and the magnitude row (only tiles the GEMM will read)
        const uint8_t* need = nullptr;
        FusedB fb{};
        const FusedB* fused = nullptr;
        // f64 engines: the 128-row tiles of their MFMA GEMM nest inside these 256-row ones, so the set is a superset;
        // the plain kernel (tiny problems) reads every Gamma element and needs them all written.
        // (a fused engine projects one or two tiles' worth of rows -- the tiles that straddle groups, the tail: all of their K
        // tiles are written, and the 12 us of k_need_tiles leave the front of every backup)
        if ((kF32 || f64_uses_mfma(B_, N)) && !(will_fuse && R_ == 1)) {
            if ((rc = need_.ensure((size_t)AO * k_tiles, &bytes_))) return rc;
            HIPCHK(launch_need_tiles(nzA_.as<uint8_t>(), (int)(B_pad_ / GEMM_BM), nzB_.as<uint8_t>(), AO, (int)V_, k_tiles,
                                     need_.as<uint8_t>(), stream_));
            need = need_.as<uint8_t>();
        }
        if (mode_ == PBVI_DENSE) {
            if (Vt > 65535) FAIL(PBVI_EUNSUPPORTED, "dense mode: at most 65534 alpha-vectors");
            // pad columns s >= S of the Gamma rows stay zero: clear them once per run (cheap) -- k_scale_rows writes s < S
            HIPCHK(hipMemsetAsync(gam_.p, 0, (size_t)N * S_pad_ * sizeof(T), stream_));
            if ((rc = project_dense(gamma))) return rc;
            if (kF32 && io.stats) {
                h_kcountD_.resize(kcountD_.cap / sizeof(int));
                HIPCHK(hipMemcpyAsync(h_kcountD_.data(), kcountD_.p, kcountD_.cap, hipMemcpyDeviceToHost, stream_));
            }
        } else {
            if constexpr (kF32) {
                if (will_fuse) {
                    fb.alpha = (const float*)alpha_.p;
                    fb.lda = S_pad_;
                    fb.rs = rs_.as<int32_t>();
                    fb.rto = (const float*)rto_.p;
                    fb.S_pad = S_pad_;
                    fb.O = O_;
                    fb.V = (int)V_;
                    fb.R = R_;
                    fb.gamma = (float)gamma;
                    fb.mat = mat_.as<uint8_t>();
                    fb.irr = irr_.as<int32_t>();
                    fb.ctile = ctile;
                    fused = &fb;
                }
            }
            HIPCHK(launch_project<T>(alpha_.as<T>(), S_pad_, (int)Vt, mv, (T)gamma, gam_.as<T>(), S_pad_, need, k_tiles, stream_,
                                     fused ? mat_.as<uint8_t>() : nullptr, fused && R_ == 1 ? vlist_.as<int>() : nullptr,
                                     fused && R_ == 1 ? (int)h_vlist_.size() : 0,
                                     fused && R_ > 1 ? irr_.as<int32_t>() : nullptr, ctile));
        }
        HIPCHK(launch_tail_rows<T>(mv, gam_.as<T>(), (int64_t)AO * Vt, S_pad_, stream_, ctile));
        HIPCHK(hipEventRecord(io.ev[1], stream_));
        // K2: scores
        // (its tile lists and stream-K plan are built on the side stream, beside the projection)
        if ((rc = score_gemm(gam_.as<T>(), N, nzB_.as<uint8_t>(), AO, (int)V_, &sv, nullptr, 0, nullptr, io.side, fused))) return rc;
        out->fused = fused != nullptr;
    }
    HIPCHK(hipEventRecord(io.ev[2], stream_));
    HIPCHK(hipStreamWaitEvent(stream_, io.join, 0));       // dead flags + rdot ready
    out->plan = plan_;    // value_max_device (K5) re-plans; keep this GEMM's for the stats
    const int k_chunk = kF32 ? plan_.chunk_len * GEMM_BK : S_pad_;
    out->tol_rel = tie_window(k_chunk);
    out->chain = chain_steps();
    out->rd_col0 = (int64_t)AO * Vt;
    out->f64_pairs = f64_pairs_;
    HIPCHK(launch_argmax<T>(sv, (int)V_, AO, (int)B_, io.dead, out->tol_rel, 0.0, out->chain, 0, io.best_v, io.best_score,
                            io.err, io.queue, io.qcount, stream_, io.tol_extra));
    HIPCHK(hipEventRecord(io.ev[3], stream_));
    out->sv = sv;
    return PBVI_OK;
}

// fp64 engines: an fp32 twin of the model on the same device and streams, used as a SCREEN.  The fp64 MFMA GEMM runs at
// its instruction's ceiling (47.6 TFLOP/s measured) -- a third of the fp32 stream-K GEMM -- and decides
// nothing the fp32 one cannot decide except near-ties.  So the scores are computed in fp32 on rounded copies of the
// operands, with the tie window widened by the input roundings, and every (belief, action, observation) whose winner
// is not clear is re-scored from the fp64 originals (the refinement fp32 engines already have, instantiated for
// double).  Indices and values are those of the pure fp64 path up to its own summation-order noise.
template <typename T>
int EngineT<T>::ensure_screen() {
    if constexpr (kF32) {
        return PBVI_OK;
    } else {
        if (screen_) return PBVI_OK;
        if (h_rto_ref_.empty()) FAIL(PBVI_ERUNTIME, "screen: host tables were not kept");
        std::vector<float> rto32(h_rto_ref_.begin(), h_rto_ref_.end()), er32(h_er_ref_.begin(), h_er_ref_.end());
        auto* e = new (std::nothrow) EngineT<float>();
        if (!e) FAIL(PBVI_ENOMEM, "screen: host allocation failed");
        const int rc = e->init(device_, S_, A_, O_, R_, h_reach_ref_.data(), rto32.data(), er32.data(), PBVI_SPARSE, stream_, stream2_, stream3_);
        if (rc != PBVI_OK) {
            delete e;
            return rc;
        }
        e->formulation_ = formulation_;
        e->fuse_project_ = fuse_project_;
        screen_ = e;
        return PBVI_OK;
    }
}

template <typename T>
int EngineT<T>::sync_screen() {
    if constexpr (kF32) {
        return PBVI_OK;
    } else {
        EngineT<float>* sc = screen_;
        int rc;
        if (screen_alpha_seen_ != alpha_ver_) {
            if (!scr_flag_.p) {
                if ((rc = scr_flag_.ensure(sizeof(int), &bytes_))) return rc;
                HIPCHK(hipMemsetAsync(scr_flag_.p, 0, sizeof(int), stream_));
            }
            auto narrow = [&](const double* src, float* dst, int64_t rows) -> int {
                const int64_t n = rows * S_pad_;
                if (n <= 0) return PBVI_OK;
                hipLaunchKernelGGL(k_narrow, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, stream_, src, dst, n,
                                   scr_flag_.as<int>());
                HIPCHK(hipGetLastError());
                return PBVI_OK;
            };
            // The screen's copy mirrors this engine's layout row for row (same free rows in front of a primary set), so a
            // set that grew at the front costs the fp32 copy its new rows only.
            const int64_t off = alpha_on_primary_ ? prim_off_ : 0;
            const size_t rows_total = alpha_on_primary_ ? alpha_buf_.cap / ((size_t)S_pad_ * sizeof(double)) : alpha_rows_cap(V_);
            if (alpha_on_primary_ && screen_layout_seen_ == prim_layout_ver_ && screen_off_seen_ >= off &&
                sc->alpha_buf_.cap >= rows_total * S_pad_ * sizeof(float)) {
                const int64_t k = screen_off_seen_ - off;
                sc->alpha_view(sc->alpha_buf_, off);
                sc->V_ = V_;
                if ((rc = narrow(alpha_.as<double>(), sc->alpha_.template as<float>(), k))) return rc;
                if ((rc = sc->merge_magnitude_row(sc->alpha_.template as<float>(), k))) return rc;
            } else {
                if ((rc = sc->alpha_buf_.ensure(rows_total * S_pad_ * sizeof(float), &sc->bytes_))) return rc;
                sc->alpha_view(sc->alpha_buf_, off);
                sc->V_ = V_;
                HIPCHK(hipMemsetAsync(scr_flag_.p, 0, sizeof(int), stream_));        // a fresh copy: no row has overflowed yet
                if ((rc = narrow(alpha_.as<double>(), sc->alpha_.template as<float>(), V_))) return rc;
                HIPCHK(hipMemsetAsync(sc->alpha_.template as<float>() + (size_t)V_ * S_pad_, 0,
                                      (rows_total - (size_t)off - (size_t)V_) * S_pad_ * sizeof(float), stream_));
                if ((rc = sc->refresh_magnitude_row())) return rc;
            }
            screen_layout_seen_ = alpha_on_primary_ ? prim_layout_ver_ : 0;      // 0: no primary layout is mirrored
            screen_off_seen_ = off;
            sc->prim_valid_ = sc->alpha_on_primary_ = false;                     // (the screen never selects by itself)
            sc->have_result_ = false;
            screen_alpha_seen_ = alpha_ver_;
        }
        if (screen_bel_seen_ != bel_ver_) {
            const int k_tiles = S_pad_ / GEMM_BK;
            if ((rc = sc->bel_.ensure((size_t)B_pad_ * S_pad_ * sizeof(float), &sc->bytes_))) return rc;
            if ((rc = sc->nzA_.ensure((size_t)(B_pad_ / GEMM_BM) * k_tiles, &sc->bytes_))) return rc;
            const int64_t n = (int64_t)B_pad_ * S_pad_;      // same row order as this engine's block (pad rows are zero)
            hipLaunchKernelGGL(k_narrow, dim3((unsigned)((n / 4 + 255) / 256 + 1)), dim3(256), 0, stream_, bel_.as<double>(),
                               sc->bel_.template as<float>(), n, (int*)nullptr);
            HIPCHK(hipGetLastError());
            HIPCHK(launch_tile_nonzero_f32(sc->bel_.template as<float>(), S_pad_, (int)B_pad_, k_tiles, sc->nzA_.template as<uint8_t>(), stream_));
            sc->B_ = B_;
            sc->B_pad_ = B_pad_;
            sc->sorted_ = false;
            sc->btl_valid_ = false;
            sc->have_result_ = false;
            screen_bel_seen_ = bel_ver_;
        }
        sc->formulation_ = formulation_;
        sc->tie_rel_user_ = tie_rel_user_;
        return PBVI_OK;
    }
}

template <typename T>
int EngineT<T>::backup_run(double gamma, int flags, pbvi_stats_t* st) {
    if (V_ <= 0) FAIL(PBVI_EINVAL, "backup_run: no alpha set resident (call pbvi_alpha_set)");
    if (B_ <= 0) FAIL(PBVI_EINVAL, "backup_run: no belief block resident (call pbvi_beliefs_set)");
    HIPCHK(hipSetDevice(device_));
    if constexpr (!kF32) {
        const int64_t N = (int64_t)A_ * O_ * (V_ + 1) + 2 * A_;
        // worth it once the score GEMM is more than a few tiles (tiger, the 4x3 grid, S = 600 models stay pure fp64)
        const double gemm_flops = 2.0 * (double)round_up(B_, 128) * (double)N * (double)S_pad_;   // ~0.4 ms of fp64 MFMA
        const bool want = screen_mode_ == 2 || (screen_mode_ == 1 && f64_uses_mfma(B_, N) && gemm_flops >= 2e10);
        if (want && mode_ == PBVI_SPARSE && !h_rto_ref_.empty()) {
            int rc = ensure_screen();
            if (rc) return rc;
            if ((rc = sync_screen())) return rc;
            if (int* f = pinned_flag()) f[1] = 0;
            if ((rc = run_pipeline<float>(*screen_, gamma, flags, st))) return rc;
            // |alpha| beyond FLT_MAX became inf in the screen's copy (NaN scores follow): the fp64 pipeline decides alone
            if (h_flag_ && h_flag_[1]) return run_pipeline<T>(*this, gamma, flags, st);
            return PBVI_OK;
        }
    }
    return run_pipeline<T>(*this, gamma, flags, st);
}

// One backup: `scorer` (this engine, or its fp32 screen) produces scores and first maxima; this engine, which owns
// the operands in their original precision, decides near-ties, actions and assembles the rows.
template <typename T>
template <typename TS>
int EngineT<T>::run_pipeline(EngineT<TS>& scorer, double gamma, int flags, pbvi_stats_t* st) {
    constexpr bool windows = sizeof(TS) == 4;              // fp32 scores: tie windows + fp64 re-decision
    constexpr bool screened = sizeof(TS) != sizeof(T);
    int rc;
    const int AO = A_ * O_;
    const int64_t Vt = V_ + 1;                 // alpha rows + magnitude row
    const int64_t N = (int64_t)AO * Vt + 2 * A_;
    const int64_t pairs = B_ * AO;
    if (N > 0x7fffffff || pairs > 0x7fffffff) FAIL(PBVI_EUNSUPPORTED, "backup_run: A*O*(V+1) or B*A*O exceeds int32");
    const ModelView<T> mv = view();
    const bool use_push = scorer.choose_push(N);
    last_formulation_ = use_push ? 2 : 1;
    if ((rc = best_v_.ensure((size_t)pairs * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = best_score_.ensure((size_t)pairs * sizeof(double), &bytes_))) return rc;
    if ((rc = err_.ensure((size_t)pairs * sizeof(double), &bytes_))) return rc;
    if ((rc = queue_.ensure((size_t)pairs * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = dead_.ensure((size_t)pairs, &bytes_))) return rc;
    if ((rc = rdot_.ensure((size_t)B_ * A_ * 2 * sizeof(double), &bytes_))) return rc;
    if ((rc = action_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = aqueue_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = acand_.ensure((size_t)B_ * A_, &bytes_))) return rc;
    if ((rc = out_.ensure((size_t)B_ * S_ * sizeof(T), &bytes_))) return rc;
    if ((rc = keep_.ensure((size_t)B_, &bytes_))) return rc;
    int* qcount = counters_.as<int>();
    int* aqcount = counters_.as<int>() + 1;
    if (windows && (rc = stage_reserve(256))) return rc;      // pinned room for the refinement's counts, while the stream is idle
    HIPCHK(hipMemsetAsync(counters_.p, 0, 8 * sizeof(int), stream_));

    HIPCHK(hipEventRecord(ev_[0], stream_));
    // Belief-only work (dead triples, b.ER: ~0.1 ms each) on the side stream, beside the projection.
    // (Beside the persistent stream-K GEMM they starve: k_rdot took 3.1 ms there instead of 0.1.)
    static const bool no_side = getenv("PBVI_NO_SIDE_STREAM") != nullptr;      // debug / A-B only
    hipStream_t side = no_side ? stream_ : stream2_;
    const int k_tiles = S_pad_ / GEMM_BK;
    HIPCHK(hipEventRecord(ev_fork_, stream_));
    HIPCHK(hipStreamWaitEvent(side, ev_fork_, 0));
    static const bool lists_on_side = getenv("PBVI_LISTS_ON_SIDE") != nullptr;      // debug / A-B only: lists behind k_dead
    hipStream_t lists = no_side ? nullptr : (lists_on_side ? side : stream3_);
    // (the tile lists need the block's zero map only: where the block was indexed just now they do not wait for its gather)
    if (lists != nullptr && lists != side)
        HIPCHK(hipStreamWaitEvent(lists, (!screened && ev_nzA_ != nullptr && nzA_ver_ == bel_ver_) ? ev_nzA_ : ev_fork_, 0));
    if (windows) {   // exact, from the supports of the ORIGINAL operands (an fp32 copy may have flushed tiny values to zero)
        if ((rc = btl_.ensure((size_t)B_ * k_tiles * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = btc_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
        if ((rc = val_exact_.ensure((size_t)B_ * A_ * (1 + O_) * ACTION_SPLIT * sizeof(double), &bytes_))) return rc;
        // Dead triples and the per-belief tile lists are a function of the belief block and the model alone: computed on
        // the first backup of a block, kept for the following ones (a solve backs a block up once; update_passes > 1,
        // the belief-dominance loops of the notebooks and the benchmark back the same block up again and again).  Like
        // the block's sort order and zero-tile map they index the INPUT; nothing of a backup's result is kept.
        static const bool no_cache = getenv("PBVI_NO_DEAD_CACHE") != nullptr;       // debug / A-B only
        if (dead_ver_ != bel_ver_ || !btl_valid_ || dead_pairs_ != pairs || no_cache) {
            const bool have_flags = rowflags_ver_ == bel_ver_ && rowflags_.p != nullptr;   // left by the block's indexing pass
            HIPCHK(launch_dead<T>(bel_.as<T>(), S_pad_, (int)B_, mv, nzBw_.as<unsigned long long>(), k_tiles, dead_.as<uint8_t>(),
                                  btl_.as<int32_t>(), btc_.as<int32_t>(), counters_.as<int>() + 5, side,
                                  have_flags ? rowflags_.as<uint8_t>() : nullptr,
                                  have_flags && sorted_ ? perm_.as<int32_t>() : nullptr));
            btl_valid_ = true;
            dead_ver_ = bel_ver_;
            dead_pairs_ = pairs;
            dead_count_ = -1;                                // read back with this call's counters
        }
    }
    if (use_push) {   // b . ER[:,a] in f64 (the alpha-side gets it from Gamma's reward rows)
        if ((rc = prd_.ensure((size_t)B_ * A_ * sizeof(double), &bytes_))) return rc;
        HIPCHK(launch_rdot<T>(bel_.as<T>(), S_pad_, (int)B_, mv, windows ? btl_.as<int32_t>() : nullptr,
                              windows ? btc_.as<int32_t>() : nullptr, prd_.as<double>(), side));
    }
    // (run_fetch's provisional pipeline: its counters are cleared here, beside the projection, not in front of its kernels)
    const bool early_wanted = windows && early_rows_ != nullptr && !(flags & PBVI_BELIEF_DOMINANCE) && early_cap_ >= B_ && !no_side;
    if (early_wanted) {
        if ((rc = e_cnt_.ensure(4 * sizeof(int), &bytes_))) return rc;
        HIPCHK(hipMemsetAsync(e_cnt_.p, 0, 4 * sizeof(int), side));
    }
    HIPCHK(hipEventRecord(ev_join_, side));

    ScoreIO io;
    io.best_v = best_v_.as<int32_t>();
    io.best_score = best_score_.as<double>();
    io.err = err_.as<double>();
    io.queue = windows ? queue_.as<int32_t>() : nullptr;
    io.qcount = qcount;
    io.dead = windows ? dead_.as<uint8_t>() : nullptr;
    io.prd = use_push ? prd_.as<double>() : nullptr;
    io.join = ev_join_;
    io.side = lists;
    io.ev = ev_;
    io.stats = st != nullptr;
    // a screen multiplies operands rounded to fp32 (alpha, belief, RTO: 2^-24 relative each) and gamma in fp32
    io.tol_extra = screened ? 4.0 * 5.9604644775390625e-08 : 0.0;
    ScoreStage<TS> sc;
    if ((rc = scorer.stage_scores(gamma, use_push, io, &sc))) return rc;
    const SlabView<TS>& sv = sc.sv;
    const GemmPlan plan = sc.plan;
    bool xr_pending = false;
    if (use_push && !screened) {   // max_v b.alpha_v of these beliefs from the GEMM's extra rows (exact engines only)
        // beside the refinement, on the side stream: a wave per belief row over V scores is 0.1 ms that nothing in this
        // call waits for (the slabs stay as they are until the next GEMM; K5 is not run with this formulation's extras
        // pending -- it joins first, below)
        if ((rc = vmax_bk_.ensure((size_t)B_ * sizeof(double), &bytes_))) return rc;
        hipStream_t xs = no_side ? stream_ : stream2_;
        if (xs != stream_) {
            HIPCHK(hipEventRecord(ev_xr_[0], stream_));
            HIPCHK(hipStreamWaitEvent(xs, ev_xr_[0], 0));
        }
        hipLaunchKernelGGL(k_extra_rowmax<TS>, dim3((unsigned)((B_ + 3) / 4)), dim3(256), 0, xs, sv, sc.extra_row0, (int)B_,
                           (int)V_, sorted_ ? perm_.as<int32_t>() : nullptr, vmax_bk_.as<double>());
        HIPCHK(hipGetLastError());
        if (xs != stream_) {
            HIPCHK(hipEventRecord(ev_xr_[1], xs));
            xr_pending = true;
        }
    }
    // List lengths of this GEMM, for the statistics (K5 rebuilds the lists for its own GEMM later): into PAGE-LOCKED memory.
    // (The copy used to land in a std::vector: a device-to-pageable copy is staged by the runtime and holds the host until
    // the GEMM and the argmax in front of it have finished, so nothing behind the argmax was enqueued before then.)
    const int* h_kcount = nullptr;
    size_t n_kcount = 0;
    const int64_t f64_pairs = sc.f64_pairs;
    if (st && (windows || f64_pairs > 0)) {
        n_kcount = windows ? (size_t)plan.tiles_m * plan.tiles_n : (size_t)f64_pairs;
        if (n_kcount > kc_pin_cap_) {
            if (kc_pin_) (void)hipHostFree(kc_pin_);         // (no copy into it is pending: every call ends synchronised)
            kc_pin_ = nullptr;
            kc_pin_cap_ = 0;
            const size_t want = std::max<size_t>(n_kcount * 2, 4096);
            if (hipHostMalloc((void**)&kc_pin_, want * sizeof(int), hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                kc_pin_ = nullptr;
                FAIL(PBVI_ENOMEM, "pinned buffer for the list lengths");
            }
            kc_pin_cap_ = want;
        }
        h_kcount = kc_pin_;                                  // (the copy itself is enqueued behind the refinement, below)
    }
    // fp64 re-decision of near-ties.  The per-entry pass hands entries with many tied candidates on to grid-wide
    // passes whose launch needs the host to know how many there are -- a read-back in the middle of the pipeline
    // (~30 us of idle device).  A value function either has such ties (absorbing goals, unexplored regions: every backup
    // of a solve loop) or it has not (the synthetic sets): after a backup that deferred nothing the next one SPECULATES
    // that it will not either -- the later stages are enqueued at once, the counts are read at the end, and if there was
    // deferred work after all it runs then and the later stages are repeated.
    // Early rows (pbvi_backup_run_fetch): the first maxima and the action values they give decide MOST beliefs' keys for
    // good -- the refinement re-decides near-ties only -- so the distinct rows of that provisional decision are assembled
    // and written to the caller's page-locked buffer on the side stream NOW, under the refinement, the action stage and
    // the dedup (the 8 MB of rows used to cross PCIe after all of them: 0.17 ms of a 3 ms backup).  Afterwards the final
    // keys are matched against the provisional ones (k_match_rows): rows already on the host keep their slot, the few
    // that the refinement changed are appended.  Snapshots, because the refinement rewrites best_v / best_score in place.
    early_used_ = false;
    if (early_wanted) {
        int rc2;
        if ((rc2 = e_bv_.ensure((size_t)pairs * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_rdot_.ensure((size_t)B_ * A_ * 2 * sizeof(double), &bytes_))) return rc2;
        if ((rc2 = e_act_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_ares_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_bres_.ensure((size_t)pairs * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_rep_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_uniq_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_inv_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_slotd_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_slot_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
        if ((rc2 = e_cnt_.ensure(4 * sizeof(int), &bytes_))) return rc2;
        if ((rc2 = e_out_.ensure((size_t)B_ * S_ * sizeof(T), &bytes_))) return rc2;
        for (hipEvent_t& ev : ev_early_)
            if (!ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        // The provisional decision itself is 40 us of small kernels: they run HERE, on the main stream in front of the
        // refinement (which rewrites best_v / best_score in place), and only the copy -- 8 MB over PCIe, ~150 us, few blocks
        // -- goes to the side stream.  (Snapshots of the three arrays + the whole provisional pipeline beside the
        // refinement cost it 0.13 ms: a net loss.)
        HIPCHK(launch_action<TS>((int)B_, scorer.view(), sv, sc.rd_col0, sc.tol_rel, sc.chain, best_score_.as<double>(), err_.as<double>(),
                                 e_rdot_.as<double>(), e_rdot_.as<double>() + (size_t)B_ * A_, e_act_.as<int32_t>(), nullptr, nullptr, stream_,
                                 io.tol_extra));
        const int32_t* pa = e_act_.as<int32_t>();
        const int32_t* pb = e_bv_.as<int32_t>();
        if (sorted_) {
            hipLaunchKernelGGL(k_unpermute, dim3((unsigned)B_), dim3(64), 0, stream_, (int)B_, AO, perm_.as<int32_t>(), e_act_.as<int32_t>(),
                               best_v_.as<int32_t>(), e_ares_.as<int32_t>(), e_bres_.as<int32_t>());
            HIPCHK(hipGetLastError());
            pa = e_ares_.as<int32_t>();
            pb = e_bres_.as<int32_t>();
        } else {                                             // (one row block: the keys as they are now)
            HIPCHK(hipMemcpyAsync(e_bv_.p, best_v_.p, (size_t)pairs * sizeof(int32_t), hipMemcpyDeviceToDevice, stream_));
        }
        HIPCHK(launch_dedup((int)B_, A_, O_, pa, pb, e_rep_.as<int32_t>(), e_uniq_.as<int32_t>(), e_inv_.as<int32_t>(),
                            e_slotd_.as<int32_t>(), e_cnt_.as<int>(), stream_));
        HIPCHK(launch_assemble<T>(alpha_.as<T>(), S_pad_, mv, gamma, pa, pb, e_uniq_.as<int32_t>(), e_cnt_.as<int>(), (int)B_,
                                  e_out_.as<T>(), S_, stream_));
        HIPCHK(hipEventRecord(ev_early_[0], stream_));
        hipStream_t es = stream2_;
        HIPCHK(hipStreamWaitEvent(es, ev_early_[0], 0));
        // The copy is a plain hipMemcpyAsync on the side stream, sized by the host: it waits (below, once the refinement is
        // enqueued) for this 4-byte count.  A copy KERNEL storing to the page-locked buffer needs no host and reaches the
        // link's rate (54.7 GB/s, profiles/microbench/pcie_store.hip) -- but beside it k_refine took 0.24 ms instead of 0.14
        // whatever its grid (8 .. 1024 blocks): the runtime's copy disturbs the refinement far less (0.15 ms).
        early_dma_pending_ = true;
        int* f = pinned_flag();
        if (!f) FAIL(PBVI_ENOMEM, "run_fetch: pinned flag");
        HIPCHK(hipMemcpyAsync(f + 8, e_cnt_.p, sizeof(int), hipMemcpyDeviceToHost, es));
        HIPCHK(hipEventRecord(ev_early_[0], es));
        early_used_ = true;
    }
    RefineWork work;
    bool speculate = false;
    if (windows) {
        if ((rc = refine_work(pairs, V_, &work))) return rc;
        if constexpr (kF32 && !screened) {
            // Gamma is in HBM as the GEMM read it (projection kernel, not the fused GEMM): the per-entry pass first
            // separates candidates by fp64 sums over those fp32 rows (backup_kernels.h, RefineWork::gam)
            static const bool no_l1 = getenv("PBVI_NO_L1_SCREEN") != nullptr;      // debug / A-B only
            if (!no_l1 && !use_push && !sc.fused && mode_ == PBVI_SPARSE) {
                work.gam = (const float*)gam_.p;
                work.ldg = S_pad_;
                work.l1_rel = (double)(R_ + 4) * 5.9604644775390625e-08;
            }
        }
        // the counts land in the engine's pinned bounce buffer (reserved at the top of the call; idle during a backup:
        // results are staged through it only by the fetch calls that follow)
        rf_counts_ = reinterpret_cast<int*>(host_stage_);
        rf_counts_[0] = rf_counts_[1] = 0;
        static const bool no_spec = getenv("PBVI_NO_SPECULATION") != nullptr;      // debug / A-B only
        // (not with the belief-dominance test: its own value-max refinement re-uses the work-list buffers, so the
        // deferred entries of this one have to be finished first)
        speculate = !no_spec && last_deferred_nothing_ && work.items_v != nullptr && !(flags & PBVI_BELIEF_DOMINANCE);
        HIPCHK((launch_refine_scan<T, TS>(true, sv, (int)V_, AO, (int)pairs, queue_.as<int32_t>(), qcount, bel_.as<T>(), S_pad_,
                                          alpha_.as<T>(), S_pad_, mv, gamma, btl_.as<int32_t>(), btc_.as<int32_t>(),
                                          nzB_.as<uint8_t>(), best_v_.as<int32_t>(), best_score_.as<double>(), err_.as<double>(),
                                          counters_.as<int>() + 4, work, rf_counts_, stream_)));
        if (!speculate && work.items_v != nullptr) {
            HIPCHK(hipStreamSynchronize(stream_));
            last_deferred_nothing_ = rf_counts_[0] == 0 && rf_counts_[1] == 0;
            HIPCHK(launch_refine_deferred<T>(true, (int)V_, AO, bel_.as<T>(), S_pad_, alpha_.as<T>(), S_pad_, mv, gamma,
                                             best_v_.as<int32_t>(), best_score_.as<double>(), err_.as<double>(), work,
                                             rf_counts_[0], rf_counts_[1], stream_));
        }
    }
    if (early_used_ && early_dma_pending_) {   // the refinement is enqueued: now the host can wait for the provisional count
        HIPCHK(hipEventSynchronize(ev_early_[0]));
        const int np = h_flag_[8];
        if (np > 0)
            HIPCHK(hipMemcpyAsync(early_rows_, e_out_.p, (size_t)np * S_ * sizeof(T), hipMemcpyDeviceToHost, stream2_));
        HIPCHK(hipEventRecord(ev_early_[1], stream2_));
        early_dma_pending_ = false;
    }
    if (h_kcount != nullptr)   // statistics: not in front of the provisional decision and the refinement (K5 re-uses kcount_ later)
        HIPCHK(hipMemcpyAsync(kc_pin_, scorer.kcount_.p, n_kcount * sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHK(hipEventRecord(ev_[4], stream_));
    const int32_t* perm = sorted_ ? perm_.as<int32_t>() : nullptr;
    int* ucount = counters_.as<int>() + 3;
    int h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // all counters in the one read-back that precedes the final sync
    auto later_stages = [&]() -> int {
        bool publish = false;
        // K4: action
        double* rdot_err = rdot_.as<double>() + (size_t)B_ * A_;
        HIPCHK(launch_action<TS>((int)B_, scorer.view(), sv, sc.rd_col0, sc.tol_rel, sc.chain, best_score_.as<double>(),
                                 err_.as<double>(), rdot_.as<double>(), rdot_err, action_.as<int32_t>(),
                                 windows ? aqueue_.as<int32_t>() : nullptr, aqcount, stream_, io.tol_extra,
                                 windows ? acand_.as<uint8_t>() : nullptr));
        if (windows)
            HIPCHK(launch_refine_action<T>(bel_.as<T>(), S_pad_, (int)B_, alpha_.as<T>(), S_pad_, mv, gamma,
                                           btl_.as<int32_t>(), btc_.as<int32_t>(),
                                           aqueue_.as<int32_t>(), aqcount, rdot_.as<double>(), rdot_err, best_v_.as<int32_t>(),
                                           best_score_.as<double>(), err_.as<double>(), val_exact_.as<double>(),
                                           action_.as<int32_t>(), stream_, acand_.as<uint8_t>()));
        HIPCHK(hipEventRecord(ev_[5], stream_));
        // results to the caller's belief order, then K6: dedup by (a*, v*) key
        if (sorted_) {
            int rc2;
            if ((rc2 = action_res_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc2;
            if ((rc2 = best_res_.ensure((size_t)pairs * sizeof(int32_t), &bytes_))) return rc2;
            hipLaunchKernelGGL(k_unpermute, dim3((unsigned)B_), dim3(64), 0, stream_, (int)B_, AO, perm, action_.as<int32_t>(),
                               best_v_.as<int32_t>(), action_res_.as<int32_t>(), best_res_.as<int32_t>());
            HIPCHK(hipGetLastError());
            res_action_ = action_res_.as<int32_t>();
            res_best_ = best_res_.as<int32_t>();
        } else {
            res_action_ = action_.as<int32_t>();
            res_best_ = best_v_.as<int32_t>();
        }
        HIPCHK(launch_dedup((int)B_, A_, O_, res_action_, res_best_, rep_.as<int32_t>(), uniq_.as<int32_t>(),
                            inv_.as<int32_t>(), slot_.as<int32_t>(), ucount, stream_));
        // K3: alpha' rows of the unique keys only
        HIPCHK(launch_assemble<T>(alpha_.as<T>(), S_pad_, mv, gamma, res_action_, res_best_, uniq_.as<int32_t>(), ucount,
                                  (int)B_, out_.as<T>(), S_, stream_));
        if (early_used_) {   // final keys against the provisional ones whose rows are on their way (or there) already
            HIPCHK(hipStreamWaitEvent(stream_, ev_early_[1], 0));          // (the provisional count and rows are complete)
            HIPCHK(hipMemsetAsync(e_cnt_.as<int>() + 1, 0, 2 * sizeof(int), stream_));
            const int32_t* pa = sorted_ ? e_ares_.as<int32_t>() : e_act_.as<int32_t>();
            const int32_t* pb = sorted_ ? e_bres_.as<int32_t>() : e_bv_.as<int32_t>();
            hipLaunchKernelGGL(k_match_rows<T>, dim3((unsigned)std::min<int64_t>(B_, 2048)), dim3(256), 0, stream_, AO, O_, ucount,
                               uniq_.as<int32_t>(), res_action_, res_best_, e_uniq_.as<int32_t>(), pa, pb, e_cnt_.as<int>(),
                               (int)early_cap_, e_slot_.as<int32_t>(), out_.as<T>(), (T*)early_rows_, (int64_t)S_);
            HIPCHK(hipGetLastError());
            int* f = pinned_flag();
            if (!f) FAIL(PBVI_ENOMEM, "run_fetch: pinned flag");
            publish = rf_dst_.slot != nullptr && !(flags & PBVI_BELIEF_DOMINANCE) && rf_dst_.best == nullptr && rf_dst_.keep == nullptr &&
                      is_pinned_host_pointer(rf_dst_.slot) && is_pinned_host_pointer(rf_dst_.index) &&
                      is_pinned_host_pointer(rf_dst_.action);
            if (!publish) HIPCHK(hipMemcpyAsync(f + 4, e_cnt_.p, 3 * sizeof(int), hipMemcpyDeviceToHost, stream_));
        }
        HIPCHK(hipEventRecord(ev_[6], stream_));
        if (xr_pending) {                            // the extra rows' maxima read the slabs K5's GEMM is about to overwrite
            HIPCHK(hipStreamWaitEvent(stream_, ev_xr_[1], 0));
            xr_pending = false;
        }
        // K5: belief dominance (keep is written in caller order)
        if (flags & PBVI_BELIEF_DOMINANCE) {
            int rc2;
            if ((rc2 = value_max_device())) return rc2;
            HIPCHK(launch_keep<T>(bel_.as<T>(), S_pad_, out_.as<T>(), S_, (int)B_, S_, bs2_.as<double>(), inv_.as<int32_t>(),
                                  perm, keep_.as<uint8_t>(), stream_));
        } else {
            HIPCHK(hipMemsetAsync(keep_.p, 1, (size_t)B_, stream_));
        }
        rf_dst_.queued = false;
        if (publish) {                                       // run_fetch into page-locked arrays: one kernel stores everything small
            int* f = pinned_flag();
            hipLaunchKernelGGL(k_publish, dim3((unsigned)((B_ + 255) / 256)), dim3(256), 0, stream_, (int)B_, e_slot_.as<int32_t>(),
                               inv_.as<int32_t>(), res_action_, (int32_t*)rf_dst_.slot, (int32_t*)rf_dst_.index, (int32_t*)rf_dst_.action,
                               counters_.as<int>(), e_cnt_.as<int>(), (screened && scr_flag_.p) ? scr_flag_.as<int>() : nullptr, f);
            HIPCHK(hipGetLastError());
            rf_dst_.queued = true;
            HIPCHK(hipEventRecord(ev_[7], stream_));
            HIPCHK(hipStreamSynchronize(stream_));
            std::memcpy(h_cnt, f + 16, sizeof(h_cnt));
            return PBVI_OK;
        }
        if (early_used_ && rf_dst_.slot != nullptr) {        // run_fetch: slots, index, actions (best, keep) with this read-back
            int rc2;
            if ((rc2 = out_begin())) return rc2;
            out_used_ = 256;                                 // (the first bytes of the staging buffer hold the refinement's counts)
            if ((rc2 = out_add(rf_dst_.slot, e_slot_.p, (size_t)B_ * sizeof(int32_t)))) return rc2;
            if ((rc2 = out_add(rf_dst_.index, inv_.p, (size_t)B_ * sizeof(int32_t)))) return rc2;
            if ((rc2 = out_add(rf_dst_.action, res_action_, (size_t)B_ * sizeof(int32_t)))) return rc2;
            if (rf_dst_.best && (rc2 = out_add(rf_dst_.best, res_best_, (size_t)pairs * sizeof(int32_t)))) return rc2;
            if (rf_dst_.keep && (rc2 = out_add(rf_dst_.keep, keep_.p, (size_t)B_))) return rc2;
            rf_dst_.queued = true;
        }
        int* fc = pinned_flag();                             // (page-locked words: a copy into the stack array is staged by the runtime)
        HIPCHK(hipMemcpyAsync(fc ? (void*)(fc + 16) : (void*)h_cnt, counters_.p, sizeof(h_cnt), hipMemcpyDeviceToHost, stream_));
        if constexpr (screened) {   // did an alpha value leave the fp32 range when the screen's copy was made?
            int* f = pinned_flag();
            if (f && scr_flag_.p) HIPCHK(hipMemcpyAsync(f + 1, scr_flag_.p, sizeof(int), hipMemcpyDeviceToHost, stream_));
        }
        HIPCHK(hipEventRecord(ev_[7], stream_));
        HIPCHK(hipStreamSynchronize(stream_));
        if (fc) std::memcpy(h_cnt, fc + 16, sizeof(h_cnt));
        return PBVI_OK;
    };
    if ((rc = rep_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = uniq_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = inv_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = slot_.ensure((size_t)B_ * sizeof(int32_t), &bytes_))) return rc;
    if ((rc = later_stages())) return rc;
    if (speculate) {
        last_deferred_nothing_ = rf_counts_[0] == 0 && rf_counts_[1] == 0;
        if (!last_deferred_nothing_) {   // there was deferred work after all: do it, then the later stages again
            HIPCHK(launch_refine_deferred<T>(true, (int)V_, AO, bel_.as<T>(), S_pad_, alpha_.as<T>(), S_pad_, mv, gamma,
                                             best_v_.as<int32_t>(), best_score_.as<double>(), err_.as<double>(), work,
                                             rf_counts_[0], rf_counts_[1], stream_));
            HIPCHK(hipMemsetAsync(counters_.as<int>() + 1, 0, sizeof(int), stream_));       // aqcount
            HIPCHK(hipMemsetAsync(counters_.as<int>() + 3, 0, sizeof(int), stream_));       // ucount (value_max's counter [2] is reset there)
            if ((rc = later_stages())) return rc;
        }
    }
    if (windows && dead_count_ < 0) dead_count_ = h_cnt[5];
    full_valid_ = false;
    res_sorted_ = sorted_;
    have_result_ = true;
    have_bk_vmax_ = use_push && !screened;
    res_B_ = B_;
    const int h_ucount = h_cnt[3];
    res_unique_ = h_ucount;

    if (st) {
        std::memset(st, 0, sizeof(*st));
        auto el = [&](int i, int j) {
            float t = 0.f;
            (void)hipEventElapsedTime(&t, ev_[i], ev_[j]);
            return (double)t;
        };
        st->ms_project = el(0, 1);
        st->ms_score = el(1, 2);
        st->ms_argmax = el(2, 3);
        st->ms_refine = el(3, 4);
        st->ms_action = el(4, 5);
        st->ms_assemble = el(5, 6);
        st->ms_dominance = el(6, 7);
        st->ms_total = el(0, 7);
        st->n_pairs = pairs;
        const int* h = h_cnt;
        st->n_refine_candidates = h[4];
        st->n_refined = h[0];
        st->n_refined_actions = h[1];
        st->n_unique = h_ucount;
        st->formulation = last_formulation_;
        st->n_dead = windows ? dead_count_ : 0;              // counted by k_dead (when the block was first backed up)
        if (mode_ == PBVI_DENSE) {
            if (kF32) {
                float t = 0.f;
                (void)hipEventElapsedTime(&t, ev_pg_[0], ev_pg_[1]);
                st->ms_project_gemm = (double)t;
            }
            st->project_flops = 2LL * AO * V_ * (int64_t)S_ * S_;
            st->project_flops_executed = st->project_flops;
            if (kF32) {
                int64_t kt_sum = 0;
                for (int64_t i = 0; i < dense_pairs_ && i < (int64_t)h_kcountD_.size(); ++i) kt_sum += h_kcountD_[(size_t)i];
                st->project_flops_executed = kt_sum * 2LL * GEMM_BM * GEMM_BN * GEMM_BK;
            }
        }
        st->score_flops = 2 * B_ * (int64_t)S_ * AO * V_;
        if (windows) {
            int64_t kt_sum = 0;
            for (size_t i = 0; i < n_kcount; ++i) kt_sum += h_kcount[i];
            st->score_flops_executed = kt_sum * 2LL * GEMM_BM * GEMM_BN * GEMM_BK;
            st->score_tiles_dense = (int64_t)plan.tiles_m * plan.tiles_n * plan.k_tiles;
            st->score_tiles_run = kt_sum;
            st->split_k = plan.max_chunks;
        } else if (f64_pairs > 0) {   // fp64 MFMA path: 128 x 128 tiles, lists in 32-column steps
            int64_t kt_sum = 0;
            for (size_t i = 0; i < n_kcount; ++i) kt_sum += h_kcount[i];
            st->score_flops_executed = kt_sum * 2LL * 128 * gemm_f64_bn((int)std::min<int64_t>(use_push ? V_ : N, 0x7fffffff)) * 32;
            st->score_tiles_dense = f64_pairs * (S_pad_ / GEMM_BK);
            st->score_tiles_run = kt_sum;
            st->split_k = 1;
        } else {
            st->score_flops_executed = st->score_flops;
            st->split_k = 1;
        }
        st->screened = screened ? 1 : 0;
        st->fused_projection = sc.fused ? 1 : 0;
    }
    return PBVI_OK;
}

// --------------------------------------------------------------------------- //
// MDP value iteration (VI_Solver.solve, src/mdp.py:1485-1510).  One thread per state, tables re-tiled [A][R][S] so
// every load is coalesced; HBM/latency-bound (A*R gathers per state per sweep).  Products and sums are kept
// un-fused (fp contract off) so R = 1 reproduces NumPy's  ER + gamma * (P * v)  bit for bit.
// Sweep `it` leaves at once when sweep it-1 already met the change limit, so the host can enqueue sweeps in
// batches without syncing; the rows / values of the converged sweep stay in place.
// --------------------------------------------------------------------------- //
__global__ void k_vi_sweep(int S, int A, int R, const int32_t* __restrict__ rs, const double* __restrict__ prob,
                           const double* __restrict__ er, double gamma, double limit, int it,
                           const double* __restrict__ v_in, double* __restrict__ v_out, double* __restrict__ rows,
                           double* __restrict__ changes) {
#pragma clang fp contract(off)   // no FMA fusion: every product and sum rounds like NumPy's separate ufuncs
    if (it > 0 && changes[it - 1] < limit) return;
    __shared__ double red[4];
    const int s = blockIdx.x * 256 + threadIdx.x;
    double diff = 0.0;
    if (s < S) {
        double best = 0.0;
        for (int a = 0; a < A; ++a) {
            double acc = 0.0;
            for (int r = 0; r < R; ++r) {
                const int64_t k = ((int64_t)a * R + r) * S + s;
                const double t = prob[k] * v_in[rs[k]];
                acc = r == 0 ? t : acc + t;
            }
            const double scaled = gamma * acc;
            const double row = er[(int64_t)a * S + s] + scaled;
            rows[(int64_t)a * S + s] = row;
            best = (a == 0 || row > best) ? row : best;
        }
        v_out[s] = best;
        diff = fabs(best - v_in[s]);
    }
    // block max, then one atomic per block (non-negative doubles order like their bit patterns)
    for (int off = 32; off > 0; off >>= 1) diff = fmax(diff, __shfl_xor(diff, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = diff;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double m = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        atomicMax(reinterpret_cast<unsigned long long*>(&changes[it]), (unsigned long long)__double_as_longlong(m));
    }
}

static int mdp_value_iteration(int device, int S, int A, int R, const int32_t* rs, const double* prob, const double* er,
                               const double* v0, double gamma, double limit, int horizon, double* out_rows,
                               double* out_changes, int32_t* out_iterations) {
    if (S <= 0 || A <= 0 || R <= 0 || horizon < 0 || !rs || !prob || !er || !v0 || !out_rows)
        FAIL(PBVI_EINVAL, "mdp_value_iteration: bad arguments");
    if ((int64_t)S * A * R > 0x7fffffff) FAIL(PBVI_EUNSUPPORTED, "mdp_value_iteration: S*A*R exceeds int32");
    const size_t n = (size_t)S * A * R;
    std::vector<int32_t> h_rs(n);
    std::vector<double> h_p(n), h_er((size_t)S * A);
    for (int s = 0; s < S; ++s)
        for (int a = 0; a < A; ++a) {
            h_er[(size_t)a * S + s] = er[(size_t)s * A + a];
            for (int r = 0; r < R; ++r) {
                const size_t src = ((size_t)s * A + a) * R + r, dst = ((size_t)a * R + r) * S + s;
                if (rs[src] < 0 || rs[src] >= S) FAIL(PBVI_EINVAL, "mdp_value_iteration: reachable state out of range");
                h_rs[dst] = rs[src];
                h_p[dst] = prob[src];
            }
        }
    HIPCHK(hipSetDevice(device));
    int64_t bytes = 0;
    DevBuf d_rs, d_p, d_er, d_v[2], d_rows, d_ch;
    struct Guard {
        DevBuf* b[7];
        hipStream_t st = nullptr;
        ~Guard() {
            for (DevBuf* x : b) x->release();
            if (st) (void)hipStreamDestroy(st);
        }
    } guard{{&d_rs, &d_p, &d_er, &d_v[0], &d_v[1], &d_rows, &d_ch}};
    int rc;
    if ((rc = d_rs.ensure(n * sizeof(int32_t), &bytes))) return rc;
    if ((rc = d_p.ensure(n * sizeof(double), &bytes))) return rc;
    if ((rc = d_er.ensure((size_t)S * A * sizeof(double), &bytes))) return rc;
    if ((rc = d_v[0].ensure((size_t)S * sizeof(double), &bytes))) return rc;
    if ((rc = d_v[1].ensure((size_t)S * sizeof(double), &bytes))) return rc;
    if ((rc = d_rows.ensure((size_t)S * A * sizeof(double), &bytes))) return rc;
    if ((rc = d_ch.ensure((size_t)std::max(horizon, 1) * sizeof(double), &bytes))) return rc;
    HIPCHK(hipStreamCreateWithFlags(&guard.st, hipStreamNonBlocking));
    hipStream_t st = guard.st;
    HIPCHK(hipMemcpyAsync(d_rs.p, h_rs.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_p.p, h_p.data(), n * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_er.p, h_er.data(), (size_t)S * A * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(d_v[0].p, v0, (size_t)S * sizeof(double), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(d_ch.p, 0, (size_t)std::max(horizon, 1) * sizeof(double), st));
    HIPCHK(hipMemsetAsync(d_rows.p, 0, (size_t)S * A * sizeof(double), st));
    std::vector<double> ch((size_t)std::max(horizon, 1), 0.0);
    int done = 0;              // sweeps that ran
    bool converged = false;
    const int batch = 64;
    for (int it0 = 0; it0 < horizon && !converged; it0 += batch) {
        const int it1 = std::min(horizon, it0 + batch);
        for (int it = it0; it < it1; ++it) {
            hipLaunchKernelGGL(k_vi_sweep, dim3((S + 255) / 256), dim3(256), 0, st, S, A, R, d_rs.as<int32_t>(),
                               d_p.as<double>(), d_er.as<double>(), gamma, limit, it, d_v[it & 1].as<double>(),
                               d_v[(it + 1) & 1].as<double>(), d_rows.as<double>(), d_ch.as<double>());
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(ch.data() + it0, d_ch.as<double>() + it0, (size_t)(it1 - it0) * sizeof(double),
                              hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        for (int it = it0; it < it1; ++it) {
            done = it + 1;
            if (ch[(size_t)it] < limit) {
                converged = true;
                break;
            }
        }
    }
    HIPCHK(hipMemcpyAsync(out_rows, d_rows.p, (size_t)S * A * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (out_changes)
        for (int it = 0; it < done; ++it) out_changes[it] = ch[(size_t)it];
    if (out_iterations) *out_iterations = done;
    return PBVI_OK;
}

}  // namespace pbvi

// --------------------------------------------------------------------------- //
// C-ABI
// --------------------------------------------------------------------------- //
struct pbvi_engine {
    pbvi::EngineBase* impl;
};

extern "C" {

int pbvi_version(void) { return 100; }

int pbvi_debug_poison(int enable) {
    const int prev = pbvi::poison_enabled() ? 1 : 0;
    pbvi::g_poison = enable ? 1 : 0;
    return prev;
}
int64_t pbvi_debug_alloc_limit(int64_t mb) {
    const int64_t prev = pbvi::DevBuf::limit_bytes();
    pbvi::DevBuf::limit_bytes() = mb < 0 ? -1 : mb << 20;
    return prev < 0 ? -1 : prev >> 20;
}


int pbvi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

const char* pbvi_last_error(void) { return pbvi::g_err.c_str(); }

int pbvi_engine_create(pbvi_engine_t** out, int device, int32_t S, int32_t A, int32_t O, int32_t R,
                       const int32_t* reach_states, const void* rto, const void* exp_reward, int dtype, int mode) {
    using namespace pbvi;
    if (!out) FAIL(PBVI_EINVAL, "engine_create: out is NULL");
    *out = nullptr;
    if (S <= 0 || A <= 0 || O <= 0 || R <= 0) FAIL(PBVI_EINVAL, "engine_create: S, A, O, R must be positive");
    if (!reach_states || !rto || !exp_reward) FAIL(PBVI_EINVAL, "engine_create: NULL table");
    if (dtype != PBVI_F32 && dtype != PBVI_F64) FAIL(PBVI_EINVAL, "engine_create: dtype must be PBVI_F32 or PBVI_F64");
    if (mode != PBVI_SPARSE && mode != PBVI_DENSE) FAIL(PBVI_EINVAL, "engine_create: unknown mode");
    int ndev = pbvi_device_count();
    if (ndev <= 0) FAIL(PBVI_ERUNTIME, "engine_create: no HIP device visible");
    if (device < 0 || device >= ndev) FAIL(PBVI_EINVAL, "engine_create: device index out of range");
    EngineBase* impl = nullptr;
    int rc;
    if (dtype == PBVI_F32) {
        auto* e = new (std::nothrow) EngineT<float>();
        if (!e) FAIL(PBVI_ENOMEM, "engine_create: host allocation failed");
        rc = e->init(device, S, A, O, R, reach_states, (const float*)rto, (const float*)exp_reward, mode);
        impl = e;
    } else {
        auto* e = new (std::nothrow) EngineT<double>();
        if (!e) FAIL(PBVI_ENOMEM, "engine_create: host allocation failed");
        rc = e->init(device, S, A, O, R, reach_states, (const double*)rto, (const double*)exp_reward, mode);
        impl = e;
    }
    if (rc != PBVI_OK) {
        delete impl;
        return rc;
    }
    pbvi_engine* h = new (std::nothrow) pbvi_engine{impl};
    if (!h) {
        delete impl;
        FAIL(PBVI_ENOMEM, "engine_create: host allocation failed");
    }
    *out = h;
    return PBVI_OK;
}

void pbvi_engine_destroy(pbvi_engine_t* e) {
    if (!e) return;
    delete e->impl;
    delete e;
}

#define NEED(e)                                            \
    do {                                                   \
        if (!(e) || !(e)->impl) {                          \
            pbvi::set_error("NULL engine handle");         \
            return PBVI_EINVAL;                            \
        }                                                  \
    } while (0)

int pbvi_alpha_set(pbvi_engine_t* e, const void* alpha, int64_t V) {
    NEED(e);
    return e->impl->alpha_set(alpha, V);
}
int pbvi_alpha_append(pbvi_engine_t* e, const void* alpha, int64_t V_add) {
    NEED(e);
    return e->impl->alpha_append(alpha, V_add);
}
int64_t pbvi_alpha_count(const pbvi_engine_t* e) { return (e && e->impl) ? e->impl->alpha_count() : -1; }
int pbvi_beliefs_set(pbvi_engine_t* e, const void* beliefs, int64_t B) {
    NEED(e);
    return e->impl->beliefs_set(beliefs, B);
}
int pbvi_backup_run(pbvi_engine_t* e, double gamma, int flags, pbvi_stats_t* stats) {
    NEED(e);
    return e->impl->backup_run(gamma, flags, stats);
}
int pbvi_backup_fetch(pbvi_engine_t* e, void* out_alpha, int32_t* out_action, int32_t* out_best_alpha, uint8_t* out_keep) {
    NEED(e);
    return e->impl->backup_fetch(out_alpha, out_action, out_best_alpha, out_keep);
}
int64_t pbvi_backup_store_unique(pbvi_engine_t* e, const int32_t* unique_idx, int64_t n) {
    NEED(e);
    return e->impl->store_append_unique(unique_idx, n);
}

int pbvi_backup_fetch_row_hashes(pbvi_engine_t* e, uint64_t* out_hashes) {
    NEED(e);
    return e->impl->fetch_row_hashes(out_hashes);
}
int pbvi_backup_fetch_unique_keys(pbvi_engine_t* e, int32_t* out_keys) {
    NEED(e);
    return e->impl->fetch_unique_keys(out_keys);
}
int pbvi_backup_fetch_exchange(pbvi_engine_t* e, int32_t* out) {
    NEED(e);
    return e->impl->fetch_exchange(out, e->impl->beliefs_count());
}
int pbvi_backup_fetch_exchange_padded(pbvi_engine_t* e, int64_t per, int32_t* out) {
    NEED(e);
    return e->impl->fetch_exchange(out, per);
}
int64_t pbvi_assemble_rows_store(pbvi_engine_t* e, double gamma, int64_t n, const int32_t* keys, void* out_rows) {
    NEED(e);
    return e->impl->assemble_keys_store(gamma, n, keys, out_rows);
}
// Host side of the key exchange (no engine, no device): the ranks' messages -> globally distinct keys in order of first
// occurrence + per-belief positions.  A few hundred keys and <= 65535 beliefs per rank: one pass with a small
// open-addressing table (the NumPy version -- np.unique over key rows, five fancy-index passes -- took 0.36 ms for eight
// messages of 1024 beliefs; this is a few microseconds).
int64_t pbvi_exchange_merge(const int32_t* all_meta, int32_t world, int64_t stride, int64_t per, int32_t key_width,
                            int64_t n_total, int32_t* out_keys, int32_t* out_index, int32_t* out_action, uint8_t* out_keep) {
    if (!all_meta || world <= 0 || per <= 0 || key_width <= 0 || key_width > 64 || n_total < 0 || !out_keys || !out_index ||
        !out_action || !out_keep) {
        pbvi::set_error("exchange_merge: bad arguments");
        return PBVI_EINVAL;
    }
    const int64_t n_meta = 1 + 3 * per + per * (int64_t)key_width;
    if (stride < n_meta || n_total > (int64_t)world * per || (int64_t)world * per > 0x7fffffff) {
        pbvi::set_error("exchange_merge: message stride / belief count do not fit the block size");
        return PBVI_EINVAL;
    }
    int64_t total = 0;
    for (int r = 0; r < world; ++r) {
        const int64_t c = all_meta[(int64_t)r * stride];
        if (c < 0 || c > per) {
            pbvi::set_error("exchange_merge: corrupt message (unique-row count out of range)");
            return PBVI_EINVAL;
        }
        total += c;
    }
    size_t cap = 16;
    while (cap < (size_t)total * 2 + 1) cap <<= 1;
    std::vector<int32_t> table, pos;
    try {
        table.assign(cap, -1);                               // slot -> position in out_keys
        pos.assign((size_t)std::max<int64_t>(total, 1), 0);  // (rank, u) -> position, ranks concatenated
    } catch (const std::bad_alloc&) {
        pbvi::set_error("exchange_merge: host allocation failed");
        return PBVI_ENOMEM;
    }
    int64_t n_keys = 0, base = 0;
    for (int r = 0; r < world; ++r) {
        const int32_t* msg = all_meta + (int64_t)r * stride;
        const int32_t* keys = msg + 1 + 3 * per;
        const int64_t c = msg[0];
        for (int64_t u = 0; u < c; ++u) {
            const int32_t* k = keys + u * key_width;
            uint64_t h = 1469598103934665603ull;
            for (int j = 0; j < key_width; ++j) h = (h ^ (uint32_t)k[j]) * 1099511628211ull;
            size_t slot = (size_t)(h ^ (h >> 29)) & (cap - 1);
            for (;;) {
                const int32_t at = table[slot];
                if (at < 0) {                                // first occurrence: the key gets the next position
                    std::memcpy(out_keys + n_keys * key_width, k, (size_t)key_width * sizeof(int32_t));
                    table[slot] = (int32_t)n_keys;
                    pos[(size_t)(base + u)] = (int32_t)n_keys++;
                    break;
                }
                if (std::memcmp(out_keys + (int64_t)at * key_width, k, (size_t)key_width * sizeof(int32_t)) == 0) {
                    pos[(size_t)(base + u)] = at;            // the same row found by another rank (or twice by this one)
                    break;
                }
                slot = (slot + 1) & (cap - 1);
            }
        }
        // this rank's beliefs, global belief order: rank r holds [r * per, min((r + 1) * per, n_total))
        const int64_t lo = std::min<int64_t>((int64_t)r * per, n_total), hi = std::min<int64_t>(lo + per, n_total);
        for (int64_t j = 0; j < hi - lo; ++j) {
            const int32_t u = msg[1 + j];
            if (u < 0 || u >= c) {
                pbvi::set_error("exchange_merge: corrupt message (row index out of range)");
                return PBVI_EINVAL;
            }
            out_index[lo + j] = pos[(size_t)(base + u)];
            out_action[lo + j] = msg[1 + per + j];
            out_keep[lo + j] = msg[1 + 2 * per + j] ? 1 : 0;
        }
        base += c;
    }
    return n_keys;
}
int pbvi_assemble_rows(pbvi_engine_t* e, double gamma, int64_t n, const int32_t* keys, void* out_rows) {
    NEED(e);
    return e->impl->assemble_keys(gamma, n, keys, out_rows);
}
int pbvi_engine_after_oom(pbvi_engine_t* e) {
    NEED(e);
    return e->impl->after_oom();
}
int pbvi_backup_device_results(pbvi_engine_t* e, void** d_alpha, int32_t** d_action, uint8_t** d_keep) {
    NEED(e);
    return e->impl->device_results(d_alpha, d_action, d_keep);
}
int64_t pbvi_backup_unique_count(const pbvi_engine_t* e) { return (e && e->impl) ? e->impl->unique_count() : -1; }
int pbvi_backup_fetch_unique(pbvi_engine_t* e, void* out_rows, int32_t* out_index) {
    NEED(e);
    return e->impl->fetch_unique(out_rows, out_index);
}
int pbvi_backup_run_fetch(pbvi_engine_t* e, double gamma, int flags, pbvi_stats_t* stats, void* out_rows, int64_t cap_rows,
                          int32_t* out_slot, int32_t* out_index, int32_t* out_action, int32_t* out_best_alpha, uint8_t* out_keep,
                          int64_t* n_unique, int64_t* n_slots) {
    NEED(e);
    return e->impl->run_fetch(gamma, flags, stats, out_rows, cap_rows, out_slot, out_index, out_action, out_best_alpha, out_keep,
                              n_unique, n_slots);
}
int pbvi_backup_fetch_compact(pbvi_engine_t* e, void* out_rows, int32_t* out_index, int32_t* out_action,
                              int32_t* out_best_alpha, uint8_t* out_keep) {
    NEED(e);
    return e->impl->fetch_compact(out_rows, out_index, out_action, out_best_alpha, out_keep);
}
void* pbvi_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0) return nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        pbvi::set_error("pbvi_host_alloc: hipHostMalloc of " + std::to_string(bytes) + " bytes failed");
        return nullptr;
    }
    return p;
}
void pbvi_host_free(void* p) {
    if (p && hipHostFree(p) != hipSuccess) (void)hipGetLastError();
}
int pbvi_debug_gemm_dense(int enable) { return pbvi::set_gemm_force_dense(enable); }
int pbvi_backup(pbvi_engine_t* e, const void* beliefs, int64_t B, double gamma, int flags, void* out_alpha,
                int32_t* out_action, int32_t* out_best_alpha, uint8_t* out_keep, pbvi_stats_t* stats) {
    NEED(e);
    int rc = e->impl->beliefs_set(beliefs, B);
    if (rc) return rc;
    rc = e->impl->backup_run(gamma, flags, stats);
    if (rc) return rc;
    return e->impl->backup_fetch(out_alpha, out_action, out_best_alpha, out_keep);
}
int pbvi_prune_dominated(pbvi_engine_t* e, uint8_t* keep) {
    NEED(e);
    return e->impl->prune_dominated(keep);
}
int pbvi_value_max(pbvi_engine_t* e, double* out_value, int32_t* out_index) {
    NEED(e);
    return e->impl->value_max(out_value, out_index);
}
int64_t pbvi_alpha_store_append(pbvi_engine_t* e, const void* rows, int64_t n) {
    NEED(e);
    return e->impl->store_append(0, rows, n);
}
int pbvi_value_max_store(pbvi_engine_t* e, int64_t n, double* out_value, int32_t* out_index) {
    NEED(e);
    return e->impl->value_max_store(n, out_value, out_index);
}
int64_t pbvi_belief_store_count(const pbvi_engine_t* e) { return (e && e->impl) ? e->impl->store_count(1) : -1; }
int64_t pbvi_alpha_store_count(const pbvi_engine_t* e) { return (e && e->impl) ? e->impl->store_count(0) : -1; }
int pbvi_alpha_select(pbvi_engine_t* e, const int32_t* ids, int64_t V) {
    NEED(e);
    return e->impl->store_select(0, ids, V);
}
int pbvi_alpha_store_reset(pbvi_engine_t* e) {
    NEED(e);
    return e->impl->store_reset(0);
}
int64_t pbvi_belief_store_append(pbvi_engine_t* e, const void* rows, int64_t n) {
    NEED(e);
    return e->impl->store_append(1, rows, n);
}
int pbvi_beliefs_select(pbvi_engine_t* e, const int32_t* ids, int64_t B) {
    NEED(e);
    return e->impl->store_select(1, ids, B);
}
int pbvi_belief_store_reset(pbvi_engine_t* e) {
    NEED(e);
    return e->impl->store_reset(1);
}
int pbvi_beliefs_advance(pbvi_engine_t* e, const int32_t* actions, const int32_t* observations, const uint8_t* keep,
                         int64_t* out_B) {
    NEED(e);
    return e->impl->beliefs_advance(actions, observations, keep, out_B);
}

int pbvi_backup_fetch_value_max(pbvi_engine_t* e, double* out_value) {
    NEED(e);
    return e->impl->backup_value_max(out_value);
}
int pbvi_belief_walk_keys(pbvi_engine_t* e, int64_t n, uint64_t* out_keys) {
    NEED(e);
    return e->impl->walk_keys(n, out_keys);
}
int64_t pbvi_belief_walk(pbvi_engine_t* e, const double* b0, int64_t n, const int32_t* actions, const int32_t* observations,
                         const uint8_t* restart, double* out_beliefs) {
    NEED(e);
    return e->impl->belief_walk(b0, n, actions, observations, restart, out_beliefs);
}

int pbvi_engine_set_rto_f64(pbvi_engine_t* e, const double* rto) {
    NEED(e);
    return e->impl->set_rto_f64(rto);
}

int pbvi_beliefs_fetch(pbvi_engine_t* e, void* out_beliefs) {
    NEED(e);
    return e->impl->beliefs_fetch(out_beliefs);
}

int64_t pbvi_beliefs_count(const pbvi_engine_t* e) { return (e && e->impl) ? e->impl->beliefs_count() : -1; }

int pbvi_mdp_value_iteration(int device, int32_t S, int32_t A, int32_t R, const int32_t* reach_states,
                             const double* reach_prob, const double* exp_reward, const double* v0, double gamma,
                             double max_change_limit, int32_t horizon, double* out_rows, double* out_changes,
                             int32_t* out_iterations) {
    return pbvi::mdp_value_iteration(device, S, A, R, reach_states, reach_prob, exp_reward, v0, gamma, max_change_limit,
                                     horizon, out_rows, out_changes, out_iterations);
}

int pbvi_belief_update(pbvi_engine_t* e, const int32_t* actions, const int32_t* observations, void* out_beliefs) {
    NEED(e);
    return e->impl->belief_update(actions, observations, out_beliefs);
}
int pbvi_set_formulation(pbvi_engine_t* e, int formulation) {
    NEED(e);
    return e->impl->set_formulation(formulation);
}
int pbvi_set_fused_projection(pbvi_engine_t* e, int enable) {
    NEED(e);
    return e->impl->set_fused(enable);
}
int pbvi_set_f64_screen(pbvi_engine_t* e, int mode) {
    NEED(e);
    return e->impl->set_screen(mode);
}
int pbvi_set_value_max_exact(pbvi_engine_t* e, int exact) {
    NEED(e);
    return e->impl->set_value_max_exact(exact);
}
int pbvi_alpha_layout(pbvi_engine_t* e, int64_t* free_rows, int64_t* layouts) {
    NEED(e);
    return e->impl->alpha_layout(free_rows, layouts);
}
int pbvi_set_tie_window(pbvi_engine_t* e, double rel) {
    NEED(e);
    return e->impl->set_tie_window(rel);
}
int64_t pbvi_device_bytes(const pbvi_engine_t* e) { return (e && e->impl) ? e->impl->device_bytes() : -1; }

}  // extern "C"

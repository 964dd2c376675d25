// Peer-to-peer ("mesh") gradient exchange for the data-parallel step (SURVEY section 8 e: "a hand-written peer-to-peer
// kernel over hipIpcMemHandle-mapped buffers"; the reference has no counterpart: src/mlp_df_wgan_train.py:98-100 picks
// one device).
//
// An MI355X node is a full mesh: every GPU has its own xGMI link to each of the 7 others (7 x ~153 GB/s), so the
// bandwidth-optimal exchange is not a ring (per-link bound: 2 (W-1)/W n bytes through ONE link) but "every rank owns 1/W of
// the buffer": rank r PULLS its shard of the gradients from all W ranks over W-1 links at once, sums them in rank order
// (one owner per element: the result is what every rank gets, bit for bit), and PUSHES the result back over the same
// links -- (W-1)/W n bytes each way per rank, spread over all links.  Every rank's exchange region is mapped into every
// other rank's address space once at set-up (hipIpcGetMemHandle / hipIpcOpenMemHandle), so the exchange is ONE plain kernel
// launch on the step's stream: no host action, no second stream, capturable in the step's HIP graph like any other launch.
//
//   (A) start barrier   workgroup 0 raises START[r] in every rank's flag block; every workgroup waits until START[p] of all
//                       ranks p reached this collective's number in ITS OWN flag block (peers write, the owner polls
//                       locally).  A rank raises START only when its stream has reached the launch, i.e. when its
//                       contribution is final and nothing earlier on its stream still reads what the peers will overwrite.
//   (B) reduce + push   float4 grid-stride over the rank's shard: W loads (one per rank, all issued before the first add),
//                       sum in rank order, then either the sum itself (all-reduce) or -- fused optimizer form -- RMSprop
//                       on the owner's parameter / square-average element and the UPDATED PARAMETER, stored to all W ranks.
//                       An element is read and written by the same thread only, so the exchange is in place.
//   (C) end barrier     every thread releases its stores at system scope; the rank's last workgroup (a device-scope ticket)
//                       raises DONE[r] everywhere; every workgroup waits for DONE[p] of all ranks before it exits, so the
//                       next launch on the stream sees every peer's pushes, and no peer still reads this rank's region.
//
// Flags are monotonically increasing collective numbers (no reset, no ABA), one per 128-byte line, in an uncached
// allocation of their own; all flag traffic uses system-scope release stores / acquire loads.  Every wait is BOUNDED (20 s
// of the 100 MHz wall clock by default, wfd_mesh_set_timeout_ms): a peer that never arrives is counted in the error word and the kernel ends -- no wave can
// spin for ever.  The collective number lives in device memory (not in a kernel argument), so a captured graph replays it.
#include "common.h"
#include <cstring>

namespace {

constexpr int MAXW = 8;
constexpr int FLAG_WORDS = 32;                                   // one flag per 128-byte line
constexpr long FLAG_BLOCK_BYTES = 2L * MAXW * FLAG_WORDS * 4;    // START[8] then DONE[8]
constexpr unsigned long long DEFAULT_SPIN_TICKS = 2000000000ull;  // 20 s at 100 MHz (wfd_mesh_set_timeout_ms)
constexpr int MESH_THREADS = 1024;                               // large workgroups: the fences are per workgroup
constexpr int MESH_MAX_BLOCKS = 256;                             // one per CU: all resident at once (they wait for peers);
                                                                 // wfd_mesh_set_max_blocks lowers it for ranks sharing a GPU

struct MeshDev {                 // by-value kernel argument
    float* data[MAXW];           // rank p's exchange region as mapped into THIS process
    unsigned* flags[MAXW];       // rank p's flag block as mapped into this process
    unsigned* local;             // this rank's private words: [0] collectives completed, [1] workgroup ticket, [2] time-outs
    unsigned long long spin_ticks;   // bound of every wait, in ticks of the 100 MHz wall clock
    int rank, world;
};

struct Mesh {
    MeshDev d;
    int device;
    long region_floats;
    int max_blocks;
};

__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// polls are relaxed (the flag block is uncached: every load goes to memory); the acquire that orders the data accesses behind
// a successful wait is ONE system-scope fence after the wait, not a cache invalidate per poll
__device__ __forceinline__ unsigned ld_sys(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// until *f has reached collective number e (wrap-safe), at most `limit` ticks
__device__ __forceinline__ bool wait_flag(const unsigned* f, unsigned e, unsigned long long limit) {
    const unsigned long long t0 = wall_clock64();
    while ((int)(ld_sys(f) - e) < 0) {
        if (wall_clock64() - t0 > limit) return false;
        __builtin_amdgcn_s_sleep(4);
    }
    return true;
}

struct MeshOpt { long poff; float* sq; float lr, alpha, eps, gscale; unsigned long long* state; unsigned long long inc0, inc1; };

// U: elements (float4) per thread in flight.  The grid is capped (every workgroup must be resident: they wait for peers), so
// a thread walks its part in several steps, each a full memory round trip; U independent elements per step keep U * W (+ 2)
// 16-byte loads in flight per thread: U = 4, 2, 1 for W = 1, 2..4, 5..8.
template <int W, bool OPT, int U>
__global__ __launch_bounds__(MESH_THREADS) void mesh_exchange_kernel(MeshDev m, long goff, long n4, MeshOpt o) {
    __shared__ int s_last;
    const int tid = threadIdx.x, r = m.rank;
    const unsigned e = __hip_atomic_load(m.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    bool ok = true;
    if constexpr (OPT) {
        if (o.state && blockIdx.x == 0 && tid == 0) { o.state[0] += o.inc0; o.state[1] += o.inc1; }
    }
    // (A)
    if (blockIdx.x == 0 && tid < W) st_sys(m.flags[tid] + r * FLAG_WORDS, e);
    if (tid < W) ok = wait_flag(m.flags[r] + tid * FLAG_WORDS, e, m.spin_ticks);
    // ONE wave per workgroup acquires at system scope (a cache invalidate; the waves of a workgroup share the CU's caches and
    // the barrier orders the others behind it): a fence per wave made this phase, not the data, the cost of the launch
    if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    // (B)
    const long per = (n4 + W - 1) / W;
    const long a = (long)r * per, b = a + per < n4 ? a + per : n4;
    const float4* src[W];
    float4* dst[W];
#pragma unroll
    for (int p = 0; p < W; ++p) {
        src[p] = reinterpret_cast<const float4*>(m.data[p] + goff);
        dst[p] = reinterpret_cast<float4*>(m.data[p] + (OPT ? o.poff : goff));
    }
    const long stride = (long)gridDim.x * MESH_THREADS;
    for (long i0 = a + (long)blockIdx.x * MESH_THREADS + tid; i0 < b; i0 += U * stride) {
        float4 v[U][W], pv[U], sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i < b) {
#pragma unroll
                for (int p = 0; p < W; ++p) {
                    // (fused optimizer form: the gradients are read once -- they stream past the caches)
                    if constexpr (OPT) v[u][p] = nt_load4(reinterpret_cast<const float*>(src[p] + i));
                    else v[u][p] = src[p][i];
                }
                if constexpr (OPT) { pv[u] = dst[r][i]; sv[u] = nt_load4(o.sq + 4 * i); }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = i0 + u * stride;
            if (i < b) {
                float4 acc = v[u][0];
#pragma unroll
                for (int p = 1; p < W; ++p) { acc.x += v[u][p].x; acc.y += v[u][p].y; acc.z += v[u][p].z; acc.w += v[u][p].w; }
                if constexpr (OPT) {
                    rmsprop_update(pv[u].x, sv[u].x, acc.x * o.gscale, o.lr, o.alpha, o.eps);
                    rmsprop_update(pv[u].y, sv[u].y, acc.y * o.gscale, o.lr, o.alpha, o.eps);
                    rmsprop_update(pv[u].z, sv[u].z, acc.z * o.gscale, o.lr, o.alpha, o.eps);
                    rmsprop_update(pv[u].w, sv[u].w, acc.w * o.gscale, o.lr, o.alpha, o.eps);
                    nt_store4(o.sq + 4 * i, sv[u]);               // touched once per step: streams past the caches
                    acc = pv[u];
                }
#pragma unroll
                for (int p = 0; p < W; ++p) dst[p][i] = acc;
            }
        }
    }
    // (C)  the barrier orders every thread's stores before thread 0, which releases them at system scope for the whole
    // workgroup (ONE cache write-back per workgroup: a release per wave cost ~0.1 ms per launch) and takes the ticket
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        const unsigned t = __hip_atomic_fetch_add(m.local + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last) {                // every workgroup of this rank has released its part
        if (tid < W) st_sys(m.flags[tid] + (MAXW + r) * FLAG_WORDS, e);
        if (tid == 0) {
            __hip_atomic_store(m.local + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(m.local, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < W) {
        ok = wait_flag(m.flags[r] + (MAXW + tid) * FLAG_WORDS, e, m.spin_ticks) && ok;
        if (!ok) __hip_atomic_fetch_add(m.local + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (the launch ends here: the next launch on the stream starts with its own acquire and then sees every peer's stores)
}

// Small exchanges of the exact (global-batch) mode -- BatchNorm statistics, BatchNorm backward sums, the selected-row count:
// a few thousand floats, latency-bound, so ONE-shot: after the start barrier every rank reads the vector of ALL ranks and
// forms the result itself, in rank order (identical bits everywhere); the end barrier then tells every rank that its
// source has been read, and only then does it overwrite it with the result (in place).  No stores to peers at all.
//   MODE 0  x[j] <- sum_p x_p[j]
//   MODE 1  (mean[j], M2[j]) of equal-sized row shards -> the pair over all rows (Chan et al.'s update, what
//           parallel.Comm.combine_bn_stats_ computes from an all-gather): mean = sum_p mean_p / W,
//           M2 = sum_p M2_p + rows * sum_p (mean_p - mean)^2
template <int W, int MODE>
__global__ __launch_bounds__(MESH_THREADS) void mesh_small_kernel(MeshDev m, long off0, long off1, int n, float rows) {
    __shared__ int s_last;
    const int tid = threadIdx.x, r = m.rank;
    const unsigned e = __hip_atomic_load(m.local, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    bool ok = true;
    if (blockIdx.x == 0 && tid < W) st_sys(m.flags[tid] + r * FLAG_WORDS, e);
    if (tid < W) ok = wait_flag(m.flags[r] + tid * FLAG_WORDS, e, m.spin_ticks);
    if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    __syncthreads();
    const int j = blockIdx.x * MESH_THREADS + tid;
    float r0 = 0.f, r1 = 0.f;
    if (j < n) {
#pragma clang fp contract(off)
        float a[W], b[W];
#pragma unroll
        for (int p = 0; p < W; ++p) {
            a[p] = m.data[p][off0 + j];
            if constexpr (MODE == 1) b[p] = m.data[p][off1 + j];
        }
        r0 = a[0];
#pragma unroll
        for (int p = 1; p < W; ++p) r0 += a[p];
        if constexpr (MODE == 1) {
            const float mu = r0 / (float)W;
            float sm = b[0], sd = (a[0] - mu) * (a[0] - mu);
#pragma unroll
            for (int p = 1; p < W; ++p) { sm += b[p]; sd += (a[p] - mu) * (a[p] - mu); }
            r0 = mu;
            r1 = sm + rows * sd;
        }
    }
    asm volatile("" ::"v"(r0), "v"(r1));          // the loads have landed before this workgroup reports "read"
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(m.local + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last) {
        if (tid < W) st_sys(m.flags[tid] + (MAXW + r) * FLAG_WORDS, e);
        if (tid == 0) {
            __hip_atomic_store(m.local + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(m.local, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < W) {
        ok = wait_flag(m.flags[r] + (MAXW + tid) * FLAG_WORDS, e, m.spin_ticks) && ok;
        if (!ok) __hip_atomic_fetch_add(m.local + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                              // every rank has read this rank's source: now it may be overwritten
    if (j < n) {
        m.data[r][off0 + j] = r0;
        if constexpr (MODE == 1) m.data[r][off1 + j] = r1;
    }
}

template <int MODE>
int launch_small(const Mesh* ms, long off0, long off1, int n, float rows, hipStream_t st) {
    const int blocks = (n + MESH_THREADS - 1) / MESH_THREADS;
    if (blocks > ms->max_blocks) return WFD_ERR_ARG;          // (all workgroups must be resident: this form is for small vectors)
    const dim3 g((unsigned)blocks), b(MESH_THREADS);
    switch (ms->d.world) {
#define WFD_MESH_CASE(w) case w: hipLaunchKernelGGL((mesh_small_kernel<w, MODE>), g, b, 0, st, ms->d, off0, off1, n, rows); break;
        WFD_MESH_CASE(1) WFD_MESH_CASE(2) WFD_MESH_CASE(3) WFD_MESH_CASE(4)
        WFD_MESH_CASE(5) WFD_MESH_CASE(6) WFD_MESH_CASE(7) WFD_MESH_CASE(8)
#undef WFD_MESH_CASE
        default: return WFD_ERR_ARG;
    }
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

template <bool OPT>
int launch_mesh(const Mesh* ms, long goff, long n, const MeshOpt& o, hipStream_t st) {
    const long n4 = n / 4;
    const int W = ms->d.world;
    const long per = (n4 + W - 1) / W;
    const int U = W == 1 ? 4 : W <= 4 ? 2 : 1;
    long blocks = (per + (long)MESH_THREADS * U - 1) / ((long)MESH_THREADS * U);
    if (blocks > ms->max_blocks) blocks = ms->max_blocks;
    if (blocks < 1) blocks = 1;
    const dim3 g((unsigned)blocks), b(MESH_THREADS);
    switch (W) {
#define WFD_MESH_CASE(w, u) case w: hipLaunchKernelGGL((mesh_exchange_kernel<w, OPT, u>), g, b, 0, st, ms->d, goff, n4, o); break;
        WFD_MESH_CASE(1, 4) WFD_MESH_CASE(2, 2) WFD_MESH_CASE(3, 2) WFD_MESH_CASE(4, 2)
        WFD_MESH_CASE(5, 1) WFD_MESH_CASE(6, 1) WFD_MESH_CASE(7, 1) WFD_MESH_CASE(8, 1)
#undef WFD_MESH_CASE
        default: return WFD_ERR_ARG;
    }
    (void)U;
    WFD_CHECK_LAUNCH();
    return WFD_OK;
}

int check_range(const Mesh* ms, long off, long n) {
    if (!ms || off < 0 || n <= 0 || (off & 3) || (n & 3) || off + n > ms->region_floats) return WFD_ERR_ARG;
    return WFD_OK;
}

int on_mesh_device(const Mesh* ms, void* stream) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return WFD_ERR_LAUNCH;
    if (cur != ms->device) return WFD_ERR_ARG;
    if (stream) {
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) != hipSuccess) return WFD_ERR_LAUNCH;
        if ((int)sdev != cur) return WFD_ERR_ARG;
    }
    return WFD_OK;
}

}  // namespace

extern "C" {

int wfd_mesh_flag_bytes(void) { return (int)FLAG_BLOCK_BYTES; }

int wfd_mesh_alloc(long bytes, int kind, void** ptr, void* ipc_handle) {
    if (bytes <= 0 || !ptr || !ipc_handle || (kind != 0 && kind != 1)) return WFD_ERR_ARG;
    void* p = nullptr;
    hipError_t e;
    if (kind == 1) {      // flag block: uncached, so that a peer's store is what the owner's next poll reads
        e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
        if (e != hipSuccess) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained); }
    } else {
        e = hipMalloc(&p, (size_t)bytes);
    }
    if (e != hipSuccess || !p) { (void)hipGetLastError(); return WFD_ERR_WORKSPACE; }
    if (hipMemset(p, 0, (size_t)bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
        hipIpcGetMemHandle(reinterpret_cast<hipIpcMemHandle_t*>(ipc_handle), p) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return WFD_ERR_LAUNCH;
    }
    *ptr = p;
    return WFD_OK;
}

int wfd_mesh_open(const void* ipc_handle, void** ptr) {
    if (!ipc_handle || !ptr) return WFD_ERR_ARG;
    hipIpcMemHandle_t h;
    std::memcpy(&h, ipc_handle, sizeof(h));
    void* p = nullptr;
    if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) { (void)hipGetLastError(); return WFD_ERR_LAUNCH; }
    *ptr = p;
    return WFD_OK;
}

int wfd_mesh_close(void* ptr) {
    if (!ptr) return WFD_ERR_ARG;
    if (hipIpcCloseMemHandle(ptr) != hipSuccess) { (void)hipGetLastError(); return WFD_ERR_LAUNCH; }
    return WFD_OK;
}

int wfd_mesh_free(void* ptr) {
    if (!ptr) return WFD_ERR_ARG;
    if (hipFree(ptr) != hipSuccess) { (void)hipGetLastError(); return WFD_ERR_LAUNCH; }
    return WFD_OK;
}

int wfd_mesh_create(int rank, int world, void* const* data, void* const* flags, long region_floats, void** mesh) {
    if (world < 1 || world > MAXW || rank < 0 || rank >= world || !data || !flags || region_floats <= 0 || !mesh) return WFD_ERR_ARG;
    Mesh* ms = new Mesh{};
    for (int p = 0; p < world; ++p) {
        if (!data[p] || !flags[p] || ((uintptr_t)data[p] & 15)) { delete ms; return WFD_ERR_ARG; }
        ms->d.data[p] = static_cast<float*>(data[p]);
        ms->d.flags[p] = static_cast<unsigned*>(flags[p]);
    }
    ms->d.rank = rank;
    ms->d.world = world;
    ms->d.spin_ticks = DEFAULT_SPIN_TICKS;
    ms->region_floats = region_floats;
    ms->max_blocks = MESH_MAX_BLOCKS / 2;       // measured at world 1: 64 / 128 / 256 workgroups -> 2892 / 2924 / 2849 it/s
    if (hipGetDevice(&ms->device) != hipSuccess || hipMalloc((void**)&ms->d.local, 256) != hipSuccess ||
        hipMemset(ms->d.local, 0, 256) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        (void)hipGetLastError();
        delete ms;
        return WFD_ERR_WORKSPACE;
    }
    *mesh = ms;
    return WFD_OK;
}

int wfd_mesh_set_max_blocks(void* mesh, int max_blocks) {
    if (!mesh || max_blocks < 1 || max_blocks > MESH_MAX_BLOCKS) return WFD_ERR_ARG;
    static_cast<Mesh*>(mesh)->max_blocks = max_blocks;
    return WFD_OK;
}

int wfd_mesh_set_timeout_ms(void* mesh, long ms) {
    if (!mesh || ms < 1 || ms > 600000) return WFD_ERR_ARG;
    static_cast<Mesh*>(mesh)->d.spin_ticks = (unsigned long long)ms * 100000ull;
    return WFD_OK;
}

int wfd_mesh_destroy(void* mesh) {
    if (!mesh) return WFD_ERR_ARG;
    Mesh* ms = static_cast<Mesh*>(mesh);
    (void)hipFree(ms->d.local);
    delete ms;
    return WFD_OK;
}

int wfd_mesh_allreduce_f32(void* mesh, long off, long n, void* stream) {
    const Mesh* ms = static_cast<const Mesh*>(mesh);
    int rc = check_range(ms, off, n);
    if (rc == WFD_OK) rc = on_mesh_device(ms, stream);
    if (rc != WFD_OK) return rc;
    return launch_mesh<false>(ms, off, n, MeshOpt{}, (hipStream_t)stream);
}

int wfd_mesh_reduce_rmsprop(void* mesh, long grad_off, long param_off, float* sq, long n, float lr, float alpha, float eps,
                            float gscale, unsigned long long* state, unsigned long long inc0, unsigned long long inc1,
                            void* stream) {
    const Mesh* ms = static_cast<const Mesh*>(mesh);
    int rc = check_range(ms, grad_off, n);
    if (rc == WFD_OK) rc = check_range(ms, param_off, n);
    if (rc == WFD_OK) rc = on_mesh_device(ms, stream);
    if (rc != WFD_OK) return rc;
    if (!sq || ((uintptr_t)sq & 15)) return WFD_ERR_ARG;
    // the two ranges must not overlap: a parameter store of one thread would race a gradient load of another
    if (grad_off < param_off + n && param_off < grad_off + n) return WFD_ERR_ARG;
    return launch_mesh<true>(ms, grad_off, n, MeshOpt{param_off, sq, lr, alpha, eps, gscale, state, inc0, inc1}, (hipStream_t)stream);
}

int wfd_mesh_allreduce_small(void* mesh, long off, int n, void* stream) {
    const Mesh* ms = static_cast<const Mesh*>(mesh);
    if (!ms || off < 0 || n <= 0 || off + n > ms->region_floats) return WFD_ERR_ARG;
    const int rc = on_mesh_device(ms, stream);
    if (rc != WFD_OK) return rc;
    return launch_small<0>(ms, off, 0, n, 0.f, (hipStream_t)stream);
}

int wfd_mesh_bn_combine(void* mesh, long mean_off, long m2_off, int n, float rows_per_rank, void* stream) {
    const Mesh* ms = static_cast<const Mesh*>(mesh);
    if (!ms || mean_off < 0 || m2_off < 0 || n <= 0 || mean_off + n > ms->region_floats || m2_off + n > ms->region_floats)
        return WFD_ERR_ARG;
    if (mean_off < m2_off + n && m2_off < mean_off + n) return WFD_ERR_ARG;
    const int rc = on_mesh_device(ms, stream);
    if (rc != WFD_OK) return rc;
    return launch_small<1>(ms, mean_off, m2_off, n, rows_per_rank, (hipStream_t)stream);
}

int wfd_mesh_status(void* mesh, long* out3) {
    if (!mesh || !out3) return WFD_ERR_ARG;
    const Mesh* ms = static_cast<const Mesh*>(mesh);
    unsigned w[3] = {0, 0, 0};
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(w, ms->d.local, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return WFD_ERR_LAUNCH;
    }
    out3[0] = w[0]; out3[1] = w[1]; out3[2] = w[2];
    return WFD_OK;
}

}  // extern "C"

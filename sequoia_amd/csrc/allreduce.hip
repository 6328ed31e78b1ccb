// e: tensor-parallel all-reduce(sum) of the row-parallel projections' partial outputs over peer-mapped buffers -- the
// reference's counterpart is the host-offload path this build replaces (Engine/offload_engine.py:388-451); the exchange
// itself has no reference line (2 all-reduces per layer of [q, hidden] fp16, 2.1 MB for the 129-node tree).
//
// xGMI is point-to-point (7 links per GPU): a ring all-reduce crosses 2 (W - 1) hops in sequence and each hop pays a
// flag round trip, so a 2 MB message is latency-bound on it.  Here every rank talks to every peer at once, two phases
// ("two-shot"), each over all links in parallel:
//   phase 1 (reduce-scatter): rank r stores chunk p of its input straight into peer p's receive area (slot r);
//   phase 2 (all-gather):     rank p sums the W copies of chunk p -- fp32, in RANK ORDER 0..W-1, one rounding to fp16 --
//                             and stores the result into every peer's result area; each element is reduced by exactly
//                             one rank, so all ranks end up with bit-identical rows (the replicated draft / sampler /
//                             verifier of the tensor-parallel loop rely on that);
//   phase 3: every rank copies the W - 1 foreign chunks from its result area into the caller's tensor.
// A chunk is cut into `blocks` sub-ranges; block b of every rank owns sub-range b in all three phases, so the only
// synchronisation is per (peer, block): one 4-byte flag per phase, carrying the call's epoch (monotonic, kept in device
// memory: the launch has no step-dependent argument and replays from a hipGraph).  The workspace is allocated uncached
// (MTYPE UC: remote stores and local reads bypass the XCD L2s) and exported through hipIpc; every spin is bounded and
// reports through a status word instead of hanging the GPU.
#include "common.h"
#include <stdlib.h>
#include <string.h>

#define AR_MAX_WORLD 8
#define AR_MAX_BLOCKS 64
#define AR_THREADS 512
#define AR_FLAG_STRIDE 16                 // uint32 per flag slot: one 64-byte line each
#define AR_SPIN_LIMIT (1u << 22)          // x (poll + s_sleep) ~ a few seconds; SEQUOIA_AR_SPIN_LIMIT overrides (tests)

// workspace layout (bytes): [header 4 KB: +0 status bits, +8 address of the external fault word, +256 / +512 per-block epochs]
// [flags1][flags2][area A: W slots][area B: n elements]
// [area C: the all-gather's [rows][W v] image]
struct ArLayout {
    size_t flags1, flags2, area_a, area_b, area_c, total, chunk_cap;
};
__host__ __device__ static inline ArLayout ar_layout(int world, size_t max_elems, size_t gather_elems = 0) {
    ArLayout L;
    const size_t flag_bytes = (size_t)AR_MAX_WORLD * AR_MAX_BLOCKS * AR_FLAG_STRIDE * 4;
    const size_t chunk = ((max_elems + world - 1) / world + 7) / 8 * 8;          // elements per chunk, 16-byte multiple
    L.chunk_cap = chunk;
    L.flags1 = 4096;
    L.flags2 = L.flags1 + flag_bytes;
    L.area_a = L.flags2 + flag_bytes;
    L.area_b = L.area_a + (size_t)world * chunk * 2;
    L.area_c = L.area_b + (size_t)world * chunk * 2;                 // the all-gather's image (its own area: see below)
    L.total = L.area_c + ((gather_elems + 7) / 8 * 8) * 2;
    return L;
}

struct ArParams {
    const float* slab;               // optional input: fp32 split-K partials [splits][n]; the rank's rows are h(sum_s slab[s])
    int splits;
    size_t split_stride;
    half_t* data;                    // [n] in / out (local); out only when slab != null
    char* ws[AR_MAX_WORLD];          // workspace of every rank as mapped in THIS process (ws[rank] = own)
    size_t n, max_elems;
    int rank, world, blocks;
    uint32_t spin_limit;
};

// A bounded spin that ran out: the collective continues on whatever its areas hold, so the result is WRONG from here on.
// The bit goes into the workspace's status word and -- when the owner registered one (sq_ar_set_fault_word) -- into a
// word in pinned host memory that the host loop reads for free at every step (Tree/_native_tree.py::collect_step /
// verify raise on it): a tensor-parallel job fails loudly at the step after the timeout instead of decoding on stale
// partial sums (the status word used to be read by the self-check and bench.py only).
// The external word takes a PLAIN system-scope store of the bits (any non-zero value raises on the host; a read-modify-write
// to pinned host memory would need PCIe atomics, which not every platform routes -- ADVICE r04).
__device__ __forceinline__ void ar_fault(char* mine, uint32_t bits) {
    atomicOr((uint32_t*)mine, bits);
    uint32_t* ext = *(uint32_t* const*)(mine + 8);
    if (ext) __hip_atomic_store(ext, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ void ar_flag_store(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// `mine`: this rank's workspace; once its status word is non-zero (an earlier wait of this job already gave up: a dead or
// stalled peer) every later wait looks ONCE instead of spinning through its full bound again -- the rest of the step's
// collectives (2 waits x 2 all-reduces x N layers) then cost microseconds, and the host's per-step fault check raises the
// informative XgmiCollectiveTimeout before any generic time-out does.
__device__ __forceinline__ bool ar_flag_wait(const uint32_t* p, uint32_t want, uint32_t spin_limit, const char* mine) {
    if (__hip_atomic_load((const uint32_t*)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) spin_limit = 1u;
    for (uint32_t it = 0; it < spin_limit; ++it) {
        // relaxed polls, ONE acquire after the hit (acquire loads in the loop cost 2-3x per hop); epochs only grow: a
        // peer that is already one call ahead has, by construction, finished this one
        if ((int32_t)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) >= 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            return true;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}

// this rank's 8 input elements at vector g: the caller's fp16 rows, or the fp16 rounding of the split-K partial sums (summed
// in split order: exactly the rows sq_add_rmsnorm_slabs_f16 would have materialised first)
__device__ __forceinline__ half8 ar_input(const ArParams& P, size_t g) {
    if (!P.slab) return *(const half8*)(P.data + g * 8);
    const float* sp = P.slab + g * 8;
    const float* const spp[1] = {sp};
    floatx4 av[1], bv[1];
    slab_sum8<1>(spp, P.splits, P.split_stride, av, bv);
    const floatx4 a = av[0], b = bv[0];
    half8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = (half_t)a[j]; o[4 + j] = (half_t)b[j]; }
    return o;
}

__global__ void __launch_bounds__(AR_THREADS) allreduce_two_shot_kernel(const ArParams P) {
    const int b = blockIdx.x, tid = threadIdx.x, W = P.world, R = P.rank;
    const ArLayout L = ar_layout(W, P.max_elems);
    char* mine = P.ws[R];
    uint32_t* epochs = (uint32_t*)(mine + 256);               // [blocks]
    __shared__ uint32_t s_epoch;
    if (tid == 0) s_epoch = epochs[b] + 1;
    __syncthreads();
    const uint32_t epoch = s_epoch;

    const size_t chunk = ((P.n + W - 1) / W + 7) / 8 * 8;     // elements (multiple of 8); the last chunk may be short
    const size_t vec_per_chunk = chunk / 8;
    const size_t v0 = vec_per_chunk * b / P.blocks, v1 = vec_per_chunk * (b + 1) / P.blocks;   // this block's 16-byte vectors
    const size_t n_vec = P.n / 8;                             // n is a multiple of 8 (checked by the host entry)

    // ---- phase 1: my copy of chunk p -> peer p's area A, slot R ---------------------------------------------------
    for (int d = 1; d < W; ++d) {
        const int p = (R + d) % W;                            // staggered: at any moment every link carries one stream
        half_t* dst = (half_t*)(P.ws[p] + L.area_a) + (size_t)R * L.chunk_cap;
        for (size_t v = v0 + tid; v < v1; v += AR_THREADS) {
            const size_t g = (size_t)p * vec_per_chunk + v;
            if (g < n_vec) *(half8*)(dst + v * 8) = ar_input(P, g);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < W && tid != R)
        ar_flag_store((uint32_t*)(P.ws[tid] + L.flags1) + ((size_t)R * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch);

    // ---- phase 2: reduce my chunk (rank order, fp32), publish it to every peer's area B ----------------------------------
    if (tid < W && tid != R) {
        if (!ar_flag_wait((const uint32_t*)(mine + L.flags1) + ((size_t)tid * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch, P.spin_limit, mine)) {
            ar_fault(mine, 1u);
        }
    }
    __syncthreads();
    {
        const half_t* a = (const half_t*)(mine + L.area_a);
        for (size_t v = v0 + tid; v < v1; v += AR_THREADS) {
            const size_t g = (size_t)R * vec_per_chunk + v;
            if (g >= n_vec) continue;
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
            for (int p = 0; p < W; ++p) {
                const half8 x = p == R ? ar_input(P, g)
                                       : __builtin_nontemporal_load((const half8*)(a + (size_t)p * L.chunk_cap + v * 8));
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)x[j];
            }
            half8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
            *(half8*)(P.data + g * 8) = o;
            for (int d = 1; d < W; ++d) {
                const int p = (R + d) % W;
                *(half8*)((half_t*)(P.ws[p] + L.area_b) + (size_t)R * L.chunk_cap + v * 8) = o;
            }
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < W && tid != R)
        ar_flag_store((uint32_t*)(P.ws[tid] + L.flags2) + ((size_t)R * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch);

    // ---- phase 3: the other ranks' reduced chunks -> the caller's tensor ----------------------------------------------------
    if (tid < W && tid != R) {
        if (!ar_flag_wait((const uint32_t*)(mine + L.flags2) + ((size_t)tid * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch, P.spin_limit, mine)) {
            ar_fault(mine, 2u);
        }
    }
    __syncthreads();
    {
        const half_t* bsrc = (const half_t*)(mine + L.area_b);
        for (int d = 1; d < W; ++d) {
            const int p = (R + d) % W;
            for (size_t v = v0 + tid; v < v1; v += AR_THREADS) {
                const size_t g = (size_t)p * vec_per_chunk + v;
                if (g < n_vec) *(half8*)(P.data + g * 8) = __builtin_nontemporal_load((const half8*)(bsrc + (size_t)p * L.chunk_cap + v * 8));
            }
        }
    }
    __syncthreads();
    if (tid == 0) epochs[b] = epoch;           // the next call (or graph replay) of this block uses epoch + 1
}

extern "C" size_t sq_ar_workspace_bytes(int world, size_t max_elems, size_t max_gather_elems) {
    if (world < 1 || world > AR_MAX_WORLD || max_elems == 0) return 0;
    return ar_layout(world, max_elems, max_gather_elems).total;
}

extern "C" int sq_ar_alloc(void** ptr, size_t bytes) {
    if (!ptr || bytes == 0) return SQ_EINVAL;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); return SQ_EUNSUPPORTED; }
    if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return SQ_ELAUNCH; }
    *ptr = p;
    return SQ_OK;
}

// Fault word: a uint32 in pinned (fine-grained, device-visible) host memory that every timeout ORs its bit into.
// host_word = the DEVICE-visible address of that word (0 clears it).  Stored in the workspace header at byte 8.
extern "C" int sq_ar_set_fault_word(void* own_ws, void* host_word) {
    if (!own_ws) return SQ_EINVAL;
    const uint64_t v = (uint64_t)(uintptr_t)host_word;
    if (hipMemcpy((char*)own_ws + 8, &v, sizeof(v), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); return SQ_ELAUNCH; }
    return SQ_OK;
}

// Workspaces in fine-grained HOST memory shared between processes (SEQUOIA_AR_WS=host; a test rig, not a fast path): a POSIX
// shared-memory object mapped by every rank and registered with the HIP runtime (hipHostRegisterMapped), so that every
// payload store, flag store and flag poll of the protocol leaves the device over PCIe -- no XCD L2 and no local HBM can
// make a missing system-scope fence or a cached flag read look correct the way two ranks on ONE GPU's memory can.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
extern "C" int sq_ar_shared_host_open(const char* name, size_t bytes, int create, void** host_ptr, void** dev_ptr) {
    if (!name || !host_ptr || !dev_ptr || bytes == 0) return SQ_EINVAL;
    const int fd = shm_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR, 0600);
    if (fd < 0) return SQ_EUNSUPPORTED;
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return SQ_EUNSUPPORTED; }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (create) shm_unlink(name); return SQ_EUNSUPPORTED; }
    if (create) memset(p, 0, bytes);
    void* d = nullptr;
    if (hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess ||
        hipHostGetDevicePointer(&d, p, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        munmap(p, bytes);
        if (create) shm_unlink(name);
        return SQ_EUNSUPPORTED;
    }
    *host_ptr = p; *dev_ptr = d;
    return SQ_OK;
}

extern "C" int sq_ar_shared_host_close(const char* name_to_unlink, void* host_ptr, size_t bytes) {
    int rc = SQ_OK;
    if (host_ptr) {
        if (hipHostUnregister(host_ptr) != hipSuccess) { (void)hipGetLastError(); rc = SQ_EINVAL; }
        munmap(host_ptr, bytes);
    }
    if (name_to_unlink) shm_unlink(name_to_unlink);
    return rc;
}

extern "C" int sq_ar_free(void* ptr) {
    return (ptr && hipFree(ptr) == hipSuccess) ? SQ_OK : SQ_EINVAL;
}

extern "C" int sq_ar_ipc_export(void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as 64 bytes");
    if (!ptr || !handle64) return SQ_EINVAL;
    if (hipIpcGetMemHandle((hipIpcMemHandle_t*)handle64, ptr) != hipSuccess) { (void)hipGetLastError(); return SQ_EUNSUPPORTED; }
    return SQ_OK;
}

extern "C" int sq_ar_ipc_open(const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return SQ_EINVAL;
    hipIpcMemHandle_t h;
    __builtin_memcpy(&h, handle64, sizeof(h));
    if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return SQ_EUNSUPPORTED; }
    return SQ_OK;
}

extern "C" int sq_ar_ipc_close(void* ptr) {
    return (ptr && hipIpcCloseMemHandle(ptr) == hipSuccess) ? SQ_OK : SQ_EINVAL;
}

extern "C" int sq_ar_status(const void* own_ws, int* status) {
    if (!own_ws || !status) return SQ_EINVAL;
    return hipMemcpy(status, own_ws, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess ? SQ_OK : SQ_ELAUNCH;
}

static int ar_launch(const float* slab, int splits, void* data, size_t n, int rank, int world, void* const* ws, size_t max_elems,
                     int blocks, void* stream);

extern "C" int sq_allreduce_sum_f16(void* data, size_t n, int rank, int world, void* const* ws, size_t max_elems, int blocks,
                                    void* stream) {
    return ar_launch(nullptr, 0, data, n, rank, world, ws, max_elems, blocks, stream);
}

extern "C" int sq_allreduce_sum_slabs_f16(const void* slab, int splits, void* out, size_t n, int rank, int world, void* const* ws,
                                          size_t max_elems, int blocks, void* stream) {
    if (!slab || splits < 1 || ((uintptr_t)slab & 15)) return SQ_EINVAL;
    return ar_launch((const float*)slab, splits, out, n, rank, world, ws, max_elems, blocks, stream);
}

static int ar_launch(const float* slab, int splits, void* data, size_t n, int rank, int world, void* const* ws, size_t max_elems,
                     int blocks, void* stream) {
    if (!data || !ws || world < 1 || world > AR_MAX_WORLD || rank < 0 || rank >= world || n == 0) return SQ_EINVAL;
    if (n > max_elems || (n & 7) || ((uintptr_t)data & 15)) return SQ_EUNSUPPORTED;
    ArParams P;
    P.slab = slab; P.splits = splits; P.split_stride = n;
    P.data = (half_t*)data; P.n = n; P.max_elems = max_elems; P.rank = rank; P.world = world;
    for (int i = 0; i < AR_MAX_WORLD; ++i) P.ws[i] = i < world ? (char*)ws[i] : nullptr;
    for (int i = 0; i < world; ++i)
        if (!P.ws[i]) return SQ_EINVAL;
    if (world == 1) return slab ? SQ_EUNSUPPORTED : SQ_OK;
    if (blocks <= 0) {
        // one block per 4 KB of a chunk keeps a block's three phases short (the flags are per block) without starving the
        // links: 2.1 MB over 8 ranks = 264 KB chunks -> 64 blocks; tiny messages take one
        const size_t chunk_bytes = (n + world - 1) / world * 2;
        blocks = (int)((chunk_bytes + 4095) / 4096);
    }
    P.blocks = blocks < 1 ? 1 : (blocks > AR_MAX_BLOCKS ? AR_MAX_BLOCKS : blocks);
    static uint32_t spin_limit = 0;
    if (!spin_limit) {
        const char* e = getenv("SEQUOIA_AR_SPIN_LIMIT");
        const long v = e ? atol(e) : 0;
        spin_limit = v > 0 ? (uint32_t)v : AR_SPIN_LIMIT;
    }
    P.spin_limit = spin_limit;
    hipLaunchKernelGGL(allreduce_two_shot_kernel, dim3(P.blocks), dim3(AR_THREADS), 0, (hipStream_t)stream, P);
    return sq_check_launch();
}

// ---- all-gather of the vocabulary-parallel logits ------------------------------------------------------------------------------
// Rank r holds out[:, r v : (r + 1) v] of the [rows][W v] logits (column-parallel lm_head).  One exchange: every rank
// stores its [rows][v] slice into every peer's area C at the FINAL layout position (row i, columns r v ..), raises a
// per-(peer, block) "written" flag, waits for its peers' flags and copies the foreign column blocks from its own area C
// into the caller's tensor; then it raises a "read" flag -- the next all-gather waits for it before it overwrites that
// peer's image (area C is the all-gather's own, so the all-reduces in between never touch it).  Rows are cut over the blocks.
struct AgParams {
    const half_t* slice;             // [rows][v] local
    half_t* out;                     // [rows][W v] local
    char* ws[AR_MAX_WORLD];
    size_t max_elems, gather_elems;
    int rows, v, rank, world, blocks;
    uint32_t spin_limit;
};

__global__ void __launch_bounds__(AR_THREADS) allgather_cols_kernel(const AgParams P) {
    const int b = blockIdx.x, tid = threadIdx.x, W = P.world, R = P.rank;
    const ArLayout L = ar_layout(W, P.max_elems, P.gather_elems);
    char* mine = P.ws[R];
    uint32_t* epochs = (uint32_t*)(mine + 512);               // [blocks] (the all-reduce keeps its own at + 256)
    __shared__ uint32_t s_epoch;
    if (tid == 0) s_epoch = epochs[b] + 1;
    __syncthreads();
    const uint32_t epoch = s_epoch;
    // rows are dealt to blocks by a map that does not depend on the row count of the call: row i belongs to block i % 64
    // (blocks = min(rows, 64): stride `blocks` from b).  The "read" handshake below is per (peer, block); with a map that
    // moved with `rows`, two gathers of different heights could have a block overwrite rows of a peer's image that ANOTHER
    // block of that peer was still reading from the previous gather.
    const int vec = P.v / 8;                                  // 16-byte vectors per slice row
    const size_t ld = (size_t)W * P.v;
    // the peers have read the previous image out of their area C (word + 12 of the flag line = "read" epoch)
    if (tid < W && tid != R) {
        if (!ar_flag_wait((const uint32_t*)(mine + L.flags2) + ((size_t)tid * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE + 12, epoch - 1, P.spin_limit, mine))
            ar_fault(mine, 8u);
    }
    __syncthreads();
    // my slice -> my own output and every peer's area C (final layout)
    for (int i = b; i < P.rows; i += P.blocks) {
        for (int c = tid; c < vec; c += AR_THREADS) {
            const half8 x = *(const half8*)(P.slice + (size_t)i * P.v + c * 8);
            const size_t off = (size_t)i * ld + (size_t)R * P.v + c * 8;
            *(half8*)(P.out + off) = x;
            for (int d = 1; d < W; ++d) {
                const int p = (R + d) % W;
                *(half8*)((half_t*)(P.ws[p] + L.area_c) + off) = x;
            }
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < W && tid != R)
        ar_flag_store((uint32_t*)(P.ws[tid] + L.flags2) + ((size_t)R * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE + 8, epoch);
    if (tid < W && tid != R) {
        if (!ar_flag_wait((const uint32_t*)(mine + L.flags2) + ((size_t)tid * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE + 8, epoch, P.spin_limit, mine))
            ar_fault(mine, 4u);
    }
    __syncthreads();
    const half_t* img = (const half_t*)(mine + L.area_c);
    for (int i = b; i < P.rows; i += P.blocks) {
        for (int d = 1; d < W; ++d) {
            const int p = (R + d) % W;
            for (int c = tid; c < vec; c += AR_THREADS) {
                const size_t off = (size_t)i * ld + (size_t)p * P.v + c * 8;
                *(half8*)(P.out + off) = __builtin_nontemporal_load((const half8*)(img + off));
            }
        }
    }
    __syncthreads();
    if (tid < W && tid != R)          // "read": peer tid may overwrite block b's rows of my image in its next all-gather
        ar_flag_store((uint32_t*)(P.ws[tid] + L.flags2) + ((size_t)R * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE + 12, epoch);
    if (tid == 0) epochs[b] = epoch;
}

extern "C" int sq_allgather_cols_f16(const void* slice, void* out, int rows, int v, int rank, int world, void* const* ws,
                                     size_t max_elems, size_t max_gather_elems, void* stream) {
    if (!slice || !out || !ws || rows <= 0 || v <= 0 || world < 1 || world > AR_MAX_WORLD || rank < 0 || rank >= world) return SQ_EINVAL;
    if ((v & 7) || ((uintptr_t)slice & 15) || ((uintptr_t)out & 15) || (size_t)rows * world * v > max_gather_elems)
        return SQ_EUNSUPPORTED;
    AgParams P;
    P.slice = (const half_t*)slice; P.out = (half_t*)out; P.rows = rows; P.v = v; P.rank = rank; P.world = world;
    P.max_elems = max_elems; P.gather_elems = max_gather_elems;
    for (int i = 0; i < AR_MAX_WORLD; ++i) P.ws[i] = i < world ? (char*)ws[i] : nullptr;
    for (int i = 0; i < world; ++i)
        if (!P.ws[i]) return SQ_EINVAL;
    if (world == 1) {
        if (slice != out && hipMemcpyAsync(out, slice, (size_t)rows * v * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess)
            return SQ_ELAUNCH;
        return SQ_OK;
    }
    P.blocks = rows < AR_MAX_BLOCKS ? rows : AR_MAX_BLOCKS;
    static uint32_t spin_limit = 0;
    if (!spin_limit) {
        const char* e = getenv("SEQUOIA_AR_SPIN_LIMIT");
        const long vv = e ? atol(e) : 0;
        spin_limit = vv > 0 ? (uint32_t)vv : AR_SPIN_LIMIT;
    }
    P.spin_limit = spin_limit;
    hipLaunchKernelGGL(allgather_cols_kernel, dim3(P.blocks), dim3(AR_THREADS), 0, (hipStream_t)stream, P);
    return sq_check_launch();
}

// ---- all-reduce + residual add + RMSNorm in one launch -----------------------------------------------------------------------------
// The tensor-parallel decoder layer continues, after each all-reduce, with x <- x + sum (the skip connection) and
// h <- RMSNorm(x) * g for the next projection (Engine/Llama_modules.py:282-288,341-346) -- a row-wise pass that needs whole
// rows.  Here the two-shot all-reduce is cut along ROWS (rank p reduces rows [p rpc, (p + 1) rpc), rpc = ceil(rows / W);
// block b of every rank owns the rows b, b + B, ... of every chunk in all phases), so that after the all-gather phase a block
// holds complete rows and finishes them itself: residual add, sum of squares, normalisation, fragment-major (or row-major)
// image of the next projection's operand.  Per layer two launches (sq_add_rmsnorm_*) fewer.  The arithmetic is that of
// rmsnorm_kernel<true, .> (fused_ops.hip) to the last bit -- same fp16 roundings, and the sum of squares is accumulated in
// that kernel's order (256 lanes x strided chunks, 4 wave sums combined by wave_sum_f32_dpp) -- so the fused and the
// three-launch form of a tensor-parallel forward agree bit for bit (tests/test_xgmi_allreduce_gpu.py).
struct ArnParams {
    const float* slab;               // input: fp32 split-K partials [splits][rows][hidden], or
    int splits;
    size_t split_stride;
    const half_t* in_rows;           //        fp16 rows [rows][hidden]
    half_t* x;                       // residual stream [rows][hidden], updated in place
    const half_t* g;                 // norm weight [hidden]
    half_t* out;                     // normalised rows: fragment-major image (frag_mtp > 0) or row-major
    char* ws[AR_MAX_WORLD];
    size_t max_elems;
    int rows, hidden, frag_mtp, rank, world, blocks, rpc;
    float eps;
    uint32_t spin_limit;
};

__device__ __forceinline__ half8 arn_input(const ArnParams& P, size_t e) {       // e: element index, multiple of 8
    if (!P.slab) return *(const half8*)(P.in_rows + e);
    const float* sp = P.slab + e;
    const float* const spp[1] = {sp};
    floatx4 av[1], bv[1];
    slab_sum8<1>(spp, P.splits, P.split_stride, av, bv);
    const floatx4 a = av[0], b = bv[0];
    half8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = (half_t)a[j]; o[4 + j] = (half_t)b[j]; }
    return o;
}

__global__ void __launch_bounds__(AR_THREADS) allreduce_add_rmsnorm_kernel(const ArnParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char arn_lds[];
    half_t* s_row = (half_t*)arn_lds;                          // [hidden] the summed row x + reduced
    __shared__ float s_f[4];
    const int b = blockIdx.x, tid = threadIdx.x, W = P.world, R = P.rank;
    const ArLayout L = ar_layout(W, P.max_elems);
    char* mine = P.ws[R];
    uint32_t* epochs = (uint32_t*)(mine + 256);
    __shared__ uint32_t s_epoch;
    if (tid == 0) s_epoch = epochs[b] + 1;
    __syncthreads();
    const uint32_t epoch = s_epoch;
    const int vec = P.hidden >> 3, rpc = P.rpc;

    // ---- phase 1: my partial of the rows rank p reduces -> peer p's area A, slot R ---------------------------------------
    for (int d = 1; d < W; ++d) {
        const int p = (R + d) % W;
        half_t* dst = (half_t*)(P.ws[p] + L.area_a) + (size_t)R * L.chunk_cap;
        for (int lr = b; lr < rpc; lr += P.blocks) {
            const int row = p * rpc + lr;
            if (row >= P.rows) break;
            for (int v = tid; v < vec; v += AR_THREADS)
                *(half8*)(dst + (size_t)lr * P.hidden + v * 8) = arn_input(P, (size_t)row * P.hidden + v * 8);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < W && tid != R)
        ar_flag_store((uint32_t*)(P.ws[tid] + L.flags1) + ((size_t)R * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch);

    // ---- phase 2: reduce my rows (rank order, fp32, one rounding), publish them to every rank's area B (mine included) -----
    if (tid < W && tid != R) {
        if (!ar_flag_wait((const uint32_t*)(mine + L.flags1) + ((size_t)tid * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch, P.spin_limit, mine))
            ar_fault(mine, 1u);
    }
    __syncthreads();
    {
        const half_t* a = (const half_t*)(mine + L.area_a);
        for (int lr = b; lr < rpc; lr += P.blocks) {
            const int row = R * rpc + lr;
            if (row >= P.rows) break;
            for (int v = tid; v < vec; v += AR_THREADS) {
                float acc[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = 0.f;
                for (int p = 0; p < W; ++p) {
                    const half8 xin = p == R ? arn_input(P, (size_t)row * P.hidden + v * 8)
                                             : __builtin_nontemporal_load((const half8*)(a + (size_t)p * L.chunk_cap + (size_t)lr * P.hidden + v * 8));
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)xin[j];
                }
                half8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (half_t)acc[j];
                for (int p = 0; p < W; ++p)
                    *(half8*)((half_t*)(P.ws[p] + L.area_b) + (size_t)R * L.chunk_cap + (size_t)lr * P.hidden + v * 8) = o;
            }
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid < W && tid != R)
        ar_flag_store((uint32_t*)(P.ws[tid] + L.flags2) + ((size_t)R * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch);

    // ---- phase 3: every chunk's rows of this block are complete here: residual add + RMSNorm --------------------------------
    if (tid < W && tid != R) {
        if (!ar_flag_wait((const uint32_t*)(mine + L.flags2) + ((size_t)tid * AR_MAX_BLOCKS + b) * AR_FLAG_STRIDE, epoch, P.spin_limit, mine))
            ar_fault(mine, 2u);
    }
    __syncthreads();
    const half_t* bsrc = (const half_t*)(mine + L.area_b);
    const int lane = tid & 63, wave = tid >> 6;
    for (int p = 0; p < W; ++p) {
        for (int lr = b; lr < rpc; lr += P.blocks) {
            const int row = p * rpc + lr;
            if (row >= P.rows) break;
            // x <- h(reduced + x)  (fp16 add, the decoder layer's skip connection), staged in LDS
            for (int v = tid; v < vec; v += AR_THREADS) {
                const half8 o = __builtin_nontemporal_load((const half8*)(bsrc + (size_t)p * L.chunk_cap + (size_t)lr * P.hidden + v * 8));
                const half8 r = *(const half8*)(P.x + (size_t)row * P.hidden + v * 8);
                half8 s;
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] = (half_t)((float)o[j] + (float)r[j]);
                *(half8*)(P.x + (size_t)row * P.hidden + v * 8) = s;
                *(half8*)(s_row + v * 8) = s;
            }
            __syncthreads();
            // sum of squares in rmsnorm_kernel's order: 256 lanes, lane t takes chunks t, t + 256, ...; 4 wave sums, then one more
            float ss = 0.f;
            if (tid < 256) {
                for (int c = tid; c < vec; c += 256) {
                    const half8 s = *(const half8*)(s_row + c * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) ss += (float)s[j] * (float)s[j];
                }
            }
            const float wsum = wave_sum_f32_dpp(ss);
            if (wave < 4 && lane == 0) s_f[wave] = wsum;
            __syncthreads();
            float r4 = (lane < 4) ? s_f[lane] : 0.0f;
            const float tot = wave_sum_f32_dpp(r4);
            const float inv = rsqrtf(tot / (float)P.hidden + P.eps);
            for (int c = tid; c < vec; c += AR_THREADS) {
                const half8 s = *(const half8*)(s_row + c * 8);
                const half8 wv = *(const half8*)(P.g + c * 8);
                half8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const half_t nrm = (half_t)((float)s[j] * inv);
                    o[j] = (half_t)((float)wv[j] * (float)nrm);
                }
                *(half8*)(P.out + (P.frag_mtp ? frag_chunk_offset((size_t)row, c, P.frag_mtp) : (size_t)row * P.hidden + c * 8)) = o;
            }
            __syncthreads();                                   // s_row / s_f are reused by the next row
        }
    }
    if (tid == 0) epochs[b] = epoch;
}

extern "C" int sq_allreduce_add_rmsnorm_f16(const void* slab, int splits, const void* in_rows, void* x, const void* weight, void* out,
                                            int out_frag, int rows, int hidden, float eps, int rank, int world, void* const* ws,
                                            size_t max_elems, void* stream) {
    if ((!slab && !in_rows) || !x || !weight || !out || !ws || rows <= 0 || hidden <= 0) return SQ_EINVAL;
    if (world < 2 || world > AR_MAX_WORLD || rank < 0 || rank >= world) return SQ_EINVAL;
    if (slab && (splits < 1 || ((uintptr_t)slab & 15))) return SQ_EINVAL;
    const ArLayout L = ar_layout(world, max_elems);
    const int rpc = (rows + world - 1) / world;
    if ((hidden & 7) || (out_frag && (hidden & 31)) || hidden > 16384 || (size_t)rpc * hidden > L.chunk_cap ||
        ((uintptr_t)x & 15) || ((uintptr_t)out & 15) || ((uintptr_t)weight & 15) || (in_rows && ((uintptr_t)in_rows & 15)))
        return SQ_EUNSUPPORTED;
    ArnParams P;
    P.slab = (const float*)slab; P.splits = slab ? splits : 0; P.split_stride = (size_t)rows * hidden;
    P.in_rows = (const half_t*)in_rows; P.x = (half_t*)x; P.g = (const half_t*)weight; P.out = (half_t*)out;
    P.max_elems = max_elems; P.rows = rows; P.hidden = hidden; P.frag_mtp = out_frag ? (rows + 15) / 16 : 0;
    P.rank = rank; P.world = world; P.rpc = rpc; P.eps = eps;
    P.blocks = rpc < AR_MAX_BLOCKS ? rpc : AR_MAX_BLOCKS;
    for (int i = 0; i < AR_MAX_WORLD; ++i) P.ws[i] = i < world ? (char*)ws[i] : nullptr;
    for (int i = 0; i < world; ++i)
        if (!P.ws[i]) return SQ_EINVAL;
    static uint32_t spin_limit = 0;
    if (!spin_limit) {
        const char* e = getenv("SEQUOIA_AR_SPIN_LIMIT");
        const long v = e ? atol(e) : 0;
        spin_limit = v > 0 ? (uint32_t)v : AR_SPIN_LIMIT;
    }
    P.spin_limit = spin_limit;
    hipLaunchKernelGGL(allreduce_add_rmsnorm_kernel, dim3(P.blocks), dim3(AR_THREADS), (size_t)hidden * sizeof(half_t), (hipStream_t)stream, P);
    return sq_check_launch();
}

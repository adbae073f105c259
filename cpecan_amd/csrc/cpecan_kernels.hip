/*
 * cpecan_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for cPecan's banded pair-HMM
 * forward / backward / posterior path, plus the device-memory plumbing behind them.  One translation unit; the
 * device code is split over the cpk_*.inl files included below, the host side (memory, launches, timing) is here.
 *
 * What the reference does per alignment (impl/pairwiseAligner.c:756-877, getPosteriorProbsWithBanding):
 * a forward sweep over anti-diagonals with periodic partial tracebacks; each traceback runs the
 * backward recurrence from an end-state prior, refreshes the total probability every 10th diagonal
 * and emits thresholded posteriors.  The arithmetic is log-space fp64 with a piecewise-cubic logAdd
 * (:287-307) whose fold ORDER is part of the result, so the kernels reproduce it term for term.
 *
 * How it is mapped to CDNA4 (DESIGN.md sections 3-5):
 *   - sweep kernel (cpk_sweep.inl): one 64-lane wavefront per DP region, persistent, work-queue fed; lanes <-> cells
 *     of the current anti-diagonal; the previous diagonals live in LDS, position-major, with a -inf guard position so
 *     band edges need no branches; the backward recurrence is a GATHER in the reference's scatter order; forward
 *     values stream to a per-wave ring in HBM holding one traceback segment; the sequential logAdd fold of the
 *     per-diagonal total is transposed (one lane per refresh point); posteriors are thresholded and compacted with
 *     wave ballots straight into the output list order;
 *   - packed kernel (cpk_packed.inl): narrow bands, 64/GW regions per wave, the same cell functions;
 *   - team kernel (cpk_team.inl): bands of several hundred cells, four or eight waves of a workgroup per region, one
 *     barrier per diagonal, the same cell functions;
 *   - every launch of a run is one size class of regions (narrow 8/16/32, wide 128 ... LDS-sized, global) with LDS,
 *     occupancy and scratch for its own largest region (LaunchClass below);
 *   - cpk_table_gather.inl: the per-diagonal band table from the anchors, the lists' re-ordering;
 *   - cpk_post.inl: reweighting, posterior and identity scores, ordered chain, MEA chain, left shift on the device.
 * No MFMA: an fp64 stencil bounded by vector-instruction issue (HBM traffic is ~1/3 of the algorithmic figure).
 *
 * Built with -ffp-contract=off; the logAdd cubic's three FMAs are explicit (cpk_device_common.inl: what the shipped
 * build changes against the reference's operation sequence, and the EXACT=1 diagnostic build that does not).
 */
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <algorithm>
#include <atomic>
#include <utility>
#include <thread>
#include <vector>

#include "cpecan_internal.h"
#include "cpecan_band.inl"

#define NEG_INF (-__builtin_huge_val())

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

extern "C" void cpk_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *cpk_last_error(void) { return g_err; }

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            cpk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return CPECAN_EHIP;                                                                     \
        }                                                                                           \
    } while (0)

#include "cpk_device_common.inl"
#include "cpk_sweep.inl"
#include "cpk_team.inl"
#include "cpk_table_gather.inl"
#include "cpk_packed.inl"
#include "cpk_post.inl"
#include "cpk_cells.inl"

// ------------------------------------------------------------------------------------------------
// host side of the HIP TU: memory, launch, timing
// ------------------------------------------------------------------------------------------------
using KernelFn = void (*)(const KArgs);
constexpr int kMaxClasses = CPK_WIDE_CLASSES + 6;  // wide classes + three packed ones, each of which may run as a split and a whole part

// One kernel launch of a run: the regions [regionBase, regionBase + regionCount) of the device order, which share one
// size class, with LDS, occupancy and per-wave scratch sized for the largest of THEM.
struct LaunchClass {
    bool packed = false;
    int k = 0;           // class index within its kind (wide 0..3, packed 0..2)
    KernelFn fn = nullptr;
    CpkGeometry geo{};   // what the kernel reads: the scalar fields describe this class
    int waves = 0;       // workgroups of the launch
    int threads = CPK_WAVE;  // threads per workgroup: one wave, or the waves of a team
    int64_t subSlots = 0;  // scratch slots: one per wave (sweep) or one per region group of a wave (packed)
    size_t ldsBytes = 0;
    size_t ldsBytesFwd = 0;  // split, two launches: the forward launch's own LDS size (no candidate ring)
    int regionBase = 0, regionCount = 0;
    // A SPLIT class (fewer regions than wave slots): launch 1 = forward sweeps of whole regions into per-REGION rings,
    // launch 2 = one queue item per (region, traceback segment); see kModeForward / kModeTrace in cpk_sweep.inl.
    bool split = false;
    bool dense = false;  // three-state match kernels allocated for three waves per SIMD (WPS = 3)
    bool fused = false;  // split, as ONE launch (kModeFused): regions and their traceback items in one queue
    bool abs = false;    // split, with the sweeps over absolute positions (cpk_sweep.inl "Absolute-position sweeps")
    KernelFn fnTrace = nullptr;
    int wavesTrace = 0;
    int64_t itemBase = 0, itemCount = 0;  // its items in dItems
    int64_t ringTotal = 0;                // doubles of all its regions' rings (ringEl is 0 then: nothing per slot)
    int64_t ringEl = 0, candEl = 0, refEl = 0, totEl = 0, bringEl = 0, grollEl = 0;  // elements per scratch slot
    int64_t oRing = 0, oCand = 0, oRef = 0, oTot = 0, oBring = 0, oGroll = 0, oExpect = 0;  // element offsets of the class
    double slotBytes() const {
        return 8.0 * ringEl + (double)sizeof(Candidate) * candEl + 16.0 * refEl + 8.0 * totEl + 8.0 * bringEl + 8.0 * grollEl;
    }
};

// ------------------------------------------------------------------------------------------------
// Device memory and device shells are recycled: a batch of one small problem (the single-call entry points, a caller's
// loop over alignments) would otherwise spend ~40 ms in hipMalloc / hipFree / stream creation around a 1 ms kernel, and
// a pipeline of large batches cannot afford hipFree at all: it waits for the whole device, i.e. for the other batch's
// sweep kernel.  Freed blocks up to a bounded total (CPECAN_CACHE_MB, default HALF of the device's memory) wait in a
// per-device list and serve later requests of about their size; what the list will not hold goes back to the driver, and
// all of it does when an allocation fails, on cpecan_cache_trim(), or after the device has had no live batch for
// CPECAN_CACHE_IDLE_S seconds (the idle reaper below).  A block is recycled only after its batch's own streams and
// events have completed.
// ------------------------------------------------------------------------------------------------
namespace {
struct CachedBlock {
    void *ptr;
    size_t bytes;
};
struct BlockCache {
    std::vector<CachedBlock> blocks;
    size_t bytes = 0;
};
constexpr int kMaxDevices = 64;
constexpr size_t kCacheMaxBlock = (size_t)256 << 30;  // the rings of a batch's split regions are ONE block (60 GB at config B)
constexpr int kCacheMaxBlocks = 512;
// What the idle blocks of one device may hold: CPECAN_CACHE_MB, default HALF of the device's memory (the process may
// share the GPU with torch / RCCL allocations that cannot reclaim what this library hoards; cpecan_cache_trim gives
// everything back on request, and a device without live batches drops its large blocks after an idle time, below).
// The current device is `device`.
size_t cache_max_bytes(int device) {
    static const double envMb = [] {
        const char *mb = getenv("CPECAN_CACHE_MB");
        return mb ? atof(mb) : -1.0;
    }();
    if (envMb >= 0.0) return (size_t)(envMb * 1048576.0);
    static std::atomic<size_t> half[kMaxDevices];
    if (device < 0 || device >= kMaxDevices) return (size_t)16 << 30;
    size_t v = half[device].load(std::memory_order_relaxed);
    if (v == 0) {
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) {
            (void)hipGetLastError();
            totalB = (size_t)32 << 30;
        }
        v = totalB / 2 + 1;
        half[device].store(v, std::memory_order_relaxed);
    }
    return v;
}
// Idle blocks above this size go back to the driver once a device has had no live batch for cache_idle_seconds()
// (CPECAN_CACHE_KEEP_MB, default 256; negative: keep everything, as rounds 1-2 did).  Round 3 dropped them the moment
// the last live batch of a device was destroyed: a caller that runs one big batch at a time then paid hipFree +
// hipMalloc of its rings around every batch -- ~0.08 s per GB, seconds at config B (VERDICT r3 "post-trim stall").
double cache_keep_mb() {
    static const double v = [] {
        const char *mb = getenv("CPECAN_CACHE_KEEP_MB");
        return mb ? atof(mb) : 256.0;
    }();
    return v;
}
// seconds without a live batch after which a device's large idle blocks are given back (CPECAN_CACHE_IDLE_S, default
// 20; 0: at once, as round 3 did; negative: never -- only cpecan_cache_trim and failed allocations do)
double cache_idle_seconds() {
    static const double v = [] {
        const char *s = getenv("CPECAN_CACHE_IDLE_S");
        return s ? atof(s) : 20.0;
    }();
    return v;
}
std::mutex g_cacheMutex;
BlockCache g_blockCache[kMaxDevices];
std::atomic<int> g_liveShells[kMaxDevices];  // batches (device shells) created and not destroyed yet, per device
// batches of this process that have run on a device and are not destroyed yet: the consumers of such a batch (list
// gather, reweight, MEA, ...) may still have to run beside the sweep of the batch that is being planned
static std::atomic<int> g_ranAlive[kMaxDevices];

// CPECAN_TRACE_HOST=1: every hipMalloc / hipFree of 64 MB or more with its wall time, on stderr (diagnostic)
bool trace_alloc() {
    static const bool on = getenv("CPECAN_TRACE_HOST") != nullptr;
    return on;
}
hipError_t traced_malloc(void **out, size_t bytes) {
    if (!trace_alloc() || bytes < ((size_t)64 << 20)) return hipMalloc(out, bytes);
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, bytes);
    fprintf(stderr, "[cpecan] hipMalloc %.1f MB: %.3f ms%s\n", bytes / 1048576.0,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), e == hipSuccess ? "" : " FAILED");
    return e;
}
void traced_free(void *ptr, size_t bytes) {
    if (!trace_alloc() || bytes < ((size_t)64 << 20)) {
        (void)hipFree(ptr);
        return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    (void)hipFree(ptr);
    fprintf(stderr, "[cpecan] hipFree %.1f MB: %.3f ms\n", bytes / 1048576.0,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
}

size_t round_alloc(size_t bytes) {  // coarser sizes make blocks fit later requests
    const size_t g = bytes <= (64u << 10) ? 4096 : (bytes <= (16u << 20) ? (64u << 10) : (1u << 20));
    return (bytes + g - 1) / g * g;
}

// the current device is `device`
hipError_t cache_alloc(int device, void **out, size_t bytes) {
    bytes = round_alloc(bytes ? bytes : 1);
    if (device >= 0 && device < kMaxDevices) {
        std::lock_guard<std::mutex> lock(g_cacheMutex);
        BlockCache &c = g_blockCache[device];
        int best = -1;
        for (int i = 0; i < (int)c.blocks.size(); i++)
            if (c.blocks[i].bytes >= bytes && c.blocks[i].bytes <= 2 * bytes + (1u << 20) &&
                (best < 0 || c.blocks[i].bytes < c.blocks[best].bytes))
                best = i;
        if (best >= 0) {
            *out = c.blocks[best].ptr;
            c.bytes -= c.blocks[best].bytes;
            c.blocks[best] = c.blocks.back();
            c.blocks.pop_back();
            return hipSuccess;
        }
    }
    hipError_t e = traced_malloc(out, bytes);
    if (e != hipSuccess && device >= 0 && device < kMaxDevices) {  // give the cached blocks back and try once more
        std::vector<CachedBlock> drop;
        {
            std::lock_guard<std::mutex> lock(g_cacheMutex);
            drop.swap(g_blockCache[device].blocks);
            g_blockCache[device].bytes = 0;
        }
        for (const CachedBlock &b : drop) traced_free(b.ptr, b.bytes);
        (void)hipGetLastError();
        e = traced_malloc(out, bytes);
    }
    return e;
}

// the device must be idle with respect to this block (callers synchronise first)
size_t cache_bytes(int device) {
    if (device < 0 || device >= kMaxDevices) return 0;
    std::lock_guard<std::mutex> lock(g_cacheMutex);
    return g_blockCache[device].bytes;
}

void cache_free(int device, void *ptr, size_t bytes) {
    if (!ptr) return;
    bytes = round_alloc(bytes ? bytes : 1);
    if (device >= 0 && device < kMaxDevices && bytes <= kCacheMaxBlock) {
        std::lock_guard<std::mutex> lock(g_cacheMutex);
        BlockCache &c = g_blockCache[device];
        if (c.bytes + bytes <= cache_max_bytes(device) && (int)c.blocks.size() < kCacheMaxBlocks) {
            c.blocks.push_back({ptr, bytes});
            c.bytes += bytes;
            return;
        }
    }
    traced_free(ptr, bytes);
}

// Gives the idle blocks of `device` that are larger than keepBelow bytes back to the driver (hipFree waits for the
// device).  The current device is `device`.  Returns the bytes freed.
size_t cache_trim(int device, size_t keepBelow) {
    if (device < 0 || device >= kMaxDevices) return 0;
    std::vector<CachedBlock> drop;
    {
        std::lock_guard<std::mutex> lock(g_cacheMutex);
        BlockCache &c = g_blockCache[device];
        for (size_t i = 0; i < c.blocks.size();) {
            if (c.blocks[i].bytes > keepBelow) {
                drop.push_back(c.blocks[i]);
                c.bytes -= c.blocks[i].bytes;
                c.blocks[i] = c.blocks.back();
                c.blocks.pop_back();
            } else {
                i++;
            }
        }
    }
    size_t freed = 0;
    for (const CachedBlock &b : drop) {
        traced_free(b.ptr, b.bytes);
        freed += b.bytes;
    }
    return freed;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// Host blocks.  The large host arrays of a batch (symbols, anchors, region and segment tables, the result triples)
// come from here: pinned memory when a HIP device is present, recycled like the device blocks.  A realignment batch
// (BASELINE config 4: 50 000 alignments, 60 M anchors, 74 M result triples) holds ~2 GB of them; from malloc every batch
// paid ~90 ms of first-touch page faults while filling them, ~200 ms of munmap in cpecan_batch_destroy, a staging copy
// on the way up and a pageable (pin-as-you-go) copy on the way down (profiles/r02_e2e_stages_config4_before.txt).
// A block is pinned (hipHostRegister) when it is REUSED, not when it is first allocated: pinning costs ~0.2 ms per MB,
// twice the first-touch faults it replaces, and a process that runs one batch and exits (the cpecan_realign command
// line on one file) would only pay for it -- its copies go the pageable way, as before.  From its second life on a
// block is read and written by the copy engines in place.
// Blocks below 256 KB are plain malloc.  CPECAN_HOST_CACHE_MB bounds what idle blocks may hold (default 16 GiB);
// CPECAN_PINNED=0 keeps everything pageable.
// ------------------------------------------------------------------------------------------------
namespace {
struct HostBlock {
    void *ptr;
    size_t bytes;
    bool pinned;
    bool registered;  // pinned by hipHostRegister (memory of posix_memalign) rather than allocated by hipHostMalloc
};
constexpr size_t kHostPoolMin = (size_t)256 << 10;
std::mutex g_hostMutex;
std::vector<HostBlock> g_hostLive, g_hostIdle;
size_t g_hostIdleBytes = 0;
size_t host_cache_max_bytes() {
    static const size_t v = [] {
        const char *mb = getenv("CPECAN_HOST_CACHE_MB");
        return mb ? (size_t)(atof(mb) * 1048576.0) : ((size_t)16 << 30);
    }();
    return v;
}
bool host_pinning_enabled() {
    static const bool v = [] {
        const char *e = getenv("CPECAN_PINNED");
        if (e && atoi(e) == 0) return false;
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return n > 0;
    }();
    return v;
}
// Blocks that are not hipHostMalloc'ed are anonymous mappings of whole pages: page-aligned for hipHostRegister, and
// mremap grows them without a copy.
void *host_map(size_t bytes) {
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return p == MAP_FAILED ? nullptr : p;
}
void host_release(const HostBlock &b) {
    if (b.pinned && !b.registered) {
        (void)hipHostFree(b.ptr);
        return;
    }
    if (b.registered) (void)hipHostUnregister(b.ptr);
    (void)munmap(b.ptr, b.bytes);
}
}  // namespace

// mustPin: the block will be written by kernels (it has to be device-visible whatever CPECAN_PINNED says)
static void *host_alloc_impl(size_t bytes, bool mustPin) {
    if (bytes < kHostPoolMin && !mustPin) return malloc(bytes ? bytes : 1);
    bytes = (bytes + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
    {
        std::unique_lock<std::mutex> lock(g_hostMutex);
        int best = -1;
        // a block serves requests of its own size and up to a quarter less (round 4; rounds 1-3: down to half).  The wide
        // window let the arrays of a batch trade places -- a 480 MB anchor request took the idle 890 MB result block of
        // another batch, whose next result request then found nothing and mapped a fresh, unpinned block -- so a pipeline
        // of BASELINE config-4 batches kept meeting blocks in their first life (a pageable 890 MB copy: 200 ms) and their
        // second (pinning: 180 ms) long after its first batches.
        for (int i = 0; i < (int)g_hostIdle.size(); i++)
            if (g_hostIdle[i].bytes >= bytes && g_hostIdle[i].bytes <= bytes + bytes / 4 + ((size_t)8 << 20) &&
                // kernels store through the HOST pointer of a mustPin block: only hipHostMalloc'ed memory is guaranteed to
                // map at the same device address; a range pinned later by hipHostRegister is not (no hipHostGetDevicePointer
                // is ever asked for), so such blocks never serve a mustPin request
                ((g_hostIdle[i].pinned && !g_hostIdle[i].registered) || !mustPin) &&
                (best < 0 || g_hostIdle[i].bytes < g_hostIdle[best].bytes))
                best = i;
        if (best >= 0) {
            HostBlock b = g_hostIdle[best];
            g_hostIdle[best] = g_hostIdle.back();
            g_hostIdle.pop_back();
            g_hostIdleBytes -= b.bytes;
            if (!b.pinned && host_pinning_enabled()) {  // a second life: worth pinning now (its pages are resident)
                // (outside the lock: 0.2 ms per MB, and the other threads of a pipeline -- the batches' download helpers,
                // the caller packing the next batch -- allocate and free all the time; the block is in neither list meanwhile)
                lock.unlock();
                const auto t0 = std::chrono::steady_clock::now();
                const hipError_t e = hipHostRegister(b.ptr, b.bytes, hipHostRegisterPortable);
                if (e == hipSuccess) b.pinned = b.registered = true;
                else (void)hipGetLastError();
                if (trace_alloc() && b.bytes >= ((size_t)64 << 20))
                    fprintf(stderr, "[cpecan] hipHostRegister %.1f MB: %.3f ms%s\n", b.bytes / 1048576.0,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
                            e == hipSuccess ? "" : " FAILED");
                lock.lock();
            }
            g_hostLive.push_back(b);
            return b.ptr;
        }
    }
    HostBlock b{nullptr, bytes, false, false};
    if (mustPin) {
        if (hipHostMalloc(&b.ptr, bytes, hipHostMallocPortable) == hipSuccess) b.pinned = true;
        else {
            (void)hipGetLastError();
            b.ptr = nullptr;
        }
    }
    if (!b.ptr && (mustPin || (b.ptr = host_map(bytes)) == nullptr)) return nullptr;
    if (trace_alloc() && bytes >= ((size_t)64 << 20)) fprintf(stderr, "[cpecan] new host block %.1f MB%s\n", bytes / 1048576.0, mustPin ? " (pinned)" : "");
    std::lock_guard<std::mutex> lock(g_hostMutex);
    g_hostLive.push_back(b);
    return b.ptr;
}
extern "C" void *cpk_host_alloc(size_t bytes) { return host_alloc_impl(bytes, false); }

extern "C" void cpk_host_free(void *p) {
    if (!p) return;
    HostBlock b{nullptr, 0, false, false};
    {
        std::lock_guard<std::mutex> lock(g_hostMutex);
        for (size_t i = 0; i < g_hostLive.size(); i++)
            if (g_hostLive[i].ptr == p) {
                b = g_hostLive[i];
                g_hostLive[i] = g_hostLive.back();
                g_hostLive.pop_back();
                break;
            }
        if (b.ptr && g_hostIdleBytes + b.bytes <= host_cache_max_bytes() && g_hostIdle.size() < 256) {
            g_hostIdle.push_back(b);
            g_hostIdleBytes += b.bytes;
            return;
        }
    }
    if (b.ptr) host_release(b);
    else free(p);  // a small block
}

// Releases every idle host block (pinned ones included).  Returns the bytes released.
static size_t host_trim(void) {
    std::vector<HostBlock> drop;
    {
        std::lock_guard<std::mutex> lock(g_hostMutex);
        drop.swap(g_hostIdle);
        g_hostIdleBytes = 0;
    }
    size_t freed = 0;
    for (const HostBlock &b : drop) {
        host_release(b);
        freed += b.bytes;
    }
    return freed;
}

// Grows a block to newBytes keeping its first usedBytes.  A block that is not pinned grows in place where the C library
// can do that (realloc: mremap for large blocks, no copy and no new page faults for the part that exists) -- a batch that
// is filled a slice at a time grows its arrays a dozen times.
extern "C" void *cpk_host_grow(void *p, size_t usedBytes, size_t newBytes) {
    if (!p) return cpk_host_alloc(newBytes);
    bool pooled = false, pinned = false;
    size_t oldBytes = 0;
    {
        std::lock_guard<std::mutex> lock(g_hostMutex);
        for (size_t i = 0; i < g_hostLive.size(); i++)
            if (g_hostLive[i].ptr == p) {
                pooled = true;
                pinned = g_hostLive[i].pinned;
                oldBytes = g_hostLive[i].bytes;
                if (!pinned) {  // out of the books while it may move
                    g_hostLive[i] = g_hostLive.back();
                    g_hostLive.pop_back();
                }
                break;
            }
    }
    if (pooled && !pinned) {
        const size_t rounded = (newBytes + ((size_t)2 << 20) - 1) / ((size_t)2 << 20) * ((size_t)2 << 20);
        void *q = rounded <= oldBytes ? p : mremap(p, oldBytes, rounded, MREMAP_MAYMOVE);
        if (q == MAP_FAILED) q = nullptr;
        std::lock_guard<std::mutex> lock(g_hostMutex);
        g_hostLive.push_back(HostBlock{q ? q : p, q ? (rounded > oldBytes ? rounded : oldBytes) : oldBytes, false, false});
        return q;  // on failure the old block is still the caller's (and back in the books)
    }
    if (!pooled && newBytes < kHostPoolMin) return realloc(p, newBytes);
    void *q = cpk_host_alloc(newBytes);
    if (!q) return nullptr;
    memcpy(q, p, usedBytes);
    cpk_host_free(p);
    return q;
}

// is [p, p + bytes) inside a pinned block of the pool?  (then a copy needs no staging)
static bool host_is_pinned(const void *p, size_t bytes) {
    std::lock_guard<std::mutex> lock(g_hostMutex);
    for (const HostBlock &b : g_hostLive)
        if (b.pinned && (const char *)p >= (const char *)b.ptr && (const char *)p + bytes <= (const char *)b.ptr + b.bytes) return true;
    return false;
}

// Every entry point works on its batch's device and leaves the calling thread's current device as it found it: in a
// one-process-per-GPU job (torch.distributed, RCCL) the caller's allocations and collectives follow hipGetDevice().
namespace {
struct DeviceGuard {
    int prev = -1, dev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) : dev(device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    }
};
}  // namespace
#define CPK_ON_DEVICE(dev)        \
    DeviceGuard guard_(dev);      \
    HIP_TRY(guard_.err)

// ------------------------------------------------------------------------------------------------
// The idle reaper: one helper thread per process, started when a device first runs out of live batches.  A device
// that stays without a live batch for cache_idle_seconds() has its idle blocks above CPECAN_CACHE_KEEP_MB handed back
// to the driver -- so a finished alignment phase does not leave tens of GB hoarded beside torch / RCCL -- while a caller
// that runs one big batch after the other (create, run, download, destroy: zero live batches in between) keeps its
// blocks: hipFree + hipMalloc of a config-B ring block cost seconds (tests/test_gpu_parity.py:
// test_one_batch_at_a_time_does_not_stall).  The thread sleeps on a condition variable and is joined at exit, before the
// HIP runtime's own teardown (atexit handlers run in reverse order of registration; ours is registered after the first
// HIP call of the process).
// ------------------------------------------------------------------------------------------------
namespace {
struct IdleReaper {
    std::mutex m;
    std::condition_variable cv;
    std::thread th;
    bool started = false, stop = false;
    bool armed[kMaxDevices] = {};
    std::chrono::steady_clock::time_point since[kMaxDevices];
};
IdleReaper *g_reaper = nullptr;  // heap object that is never destroyed: the thread may outlive static destructors
std::once_flag g_reaperOnce;

void reaper_main() {
    IdleReaper &r = *g_reaper;
    const auto idle = std::chrono::duration_cast<std::chrono::steady_clock::duration>(
        std::chrono::duration<double>(cache_idle_seconds()));
    std::unique_lock<std::mutex> lock(r.m);
    while (!r.stop) {
        bool any = false;
        std::chrono::steady_clock::time_point next{};
        for (int dev = 0; dev < kMaxDevices; dev++)
            if (r.armed[dev] && (!any || r.since[dev] + idle < next)) {
                next = r.since[dev] + idle;
                any = true;
            }
        if (!any) {
            r.cv.wait(lock);
            continue;
        }
        if (std::chrono::steady_clock::now() < next) {
            r.cv.wait_until(lock, next);
            continue;
        }
        for (int dev = 0; dev < kMaxDevices && !r.stop; dev++) {
            if (!r.armed[dev] || std::chrono::steady_clock::now() < r.since[dev] + idle) continue;
            r.armed[dev] = false;
            if (g_liveShells[dev].load() > 0) continue;  // a batch came back in the meantime
            lock.unlock();
            if (hipSetDevice(dev) == hipSuccess)
                (void)cache_trim(dev, (size_t)(cache_keep_mb() * 1048576.0));
            else
                (void)hipGetLastError();
            lock.lock();
        }
    }
}

void reaper_stop() {
    IdleReaper *r = g_reaper;
    if (!r) return;
    {
        std::lock_guard<std::mutex> lock(r->m);
        r->stop = true;
    }
    r->cv.notify_all();
    if (r->th.joinable()) r->th.join();
}

// the last live batch of `device` has just been destroyed (the current device is `device`)
void reaper_arm(int device) {
    if (cache_keep_mb() < 0.0 || cache_idle_seconds() < 0.0) return;
    if (cache_idle_seconds() == 0.0) {  // round 3's behaviour: at once
        (void)cache_trim(device, (size_t)(cache_keep_mb() * 1048576.0));
        return;
    }
    std::call_once(g_reaperOnce, [] { g_reaper = new IdleReaper(); });
    IdleReaper &r = *g_reaper;
    {
        std::lock_guard<std::mutex> lock(r.m);
        if (r.stop) return;
        r.armed[device] = true;
        r.since[device] = std::chrono::steady_clock::now();
        if (!r.started) {
            r.started = true;
            r.th = std::thread(reaper_main);
            atexit(reaper_stop);
        }
    }
    r.cv.notify_all();
}

// a batch is being created on `device`: its idle blocks are about to be wanted again
void reaper_disarm(int device) {
    IdleReaper *r = g_reaper;
    if (!r) return;
    std::lock_guard<std::mutex> lock(r->m);
    r->armed[device] = false;
}
}  // namespace

struct CpkDevice;
static std::vector<CpkDevice *> g_shells[kMaxDevices];  // idle device shells (guarded by g_cacheMutex)

struct CpkDevice {
    int device = 0;
    int numCUs = 0;
    CpkGeometry geo{};
    KConsts kc{};
    int nLists = 1;
    int64_t nSegs = 0, nDiags = 0;
    int64_t outTriplesPerList = 0;
    int64_t dbgCells = 0, dbgDiags = 0;
    std::vector<LaunchClass> classes;  // one launch each: the wide classes (sweep kernel), then the narrow ones (packed)
    int totalWaves = 0;
    // device buffers
    CpkRegion *dRegions = nullptr;
    CpkDiag *dDiags = nullptr;
    int32_t *dDiagPos = nullptr;  // positions of the absolute-position sweeps, one word per diagonal (fixed expansions only)
    CpkSegment *dSegs = nullptr;
    uint8_t *dSymbols = nullptr;
    CpkModel *dModel = nullptr;
    double *dRing = nullptr; Candidate *dCand = nullptr; double *dForward = nullptr, *dExpect = nullptr; double *dC = nullptr, *dM = nullptr, *dTotals = nullptr, *dGroll = nullptr, *dBring = nullptr;
    // The per-region counts and per-segment offsets the sweeps write (a few integers per region, never read back on the
    // device) live in PINNED HOST memory that the kernels store to directly: the host reads them as soon as the sweep's
    // stop event has completed.  As device buffers they needed three small copies on the batch's stream, and those
    // queued behind the NEXT batch's upload in a pipeline: 18 ms per config-4 batch (profiles/r02_e2e_stages_config4.txt).
    int32_t *dCounts = nullptr, *dSegStarts = nullptr, *dSegCounts = nullptr, *dTriples = nullptr;
    void *hostCounts = nullptr;  // the block of the host pool that holds the three
    CpkItem *dItems = nullptr;
    int *dProgress = nullptr;  // fused classes: per region, segments whose forward values are complete (+ an error word)
    bool fusedRetried = false; // a fused class timed out once and runs as two launches now (cpk_device_download)
    int32_t *dCompact = nullptr; CpkChunk *dChunks = nullptr; int64_t compactCap = 0, chunkCap = 0;
    unsigned int *dQueue = nullptr;
    double *dDbgFb = nullptr, *dDbgTotals = nullptr;
    int64_t bytes = 0;
    std::vector<CachedBlock> allocs;  // every device block this batch holds, with its size (for the block cache)
    size_t compactBytes = 0, chunkBytes = 0;
    hipEvent_t evStart = nullptr, evStop = nullptr;
    hipEvent_t evA = nullptr, evB = nullptr;  // copy timing (owned by the shell: nothing to leak on an error path)
    hipEvent_t evUp0 = nullptr, evUp1 = nullptr;  // around the upload's copies; the sweep waits for evUp1 (cpk_device_run)
    bool uploadTimed = true;
    double h2dMs = 0.0;
    hipStream_t lastStream = nullptr;
    // Copies, memsets and the small kernels around the sweep (table build, list gather, consumers) run on this
    // non-blocking stream of the batch's own, never on the null stream, and the batch waits on ITS events and streams,
    // never on the device: batch k+1 is planned and uploaded and batch k-1 is gathered and downloaded while the sweep
    // kernel of batch k runs (a pipeline of batches from one host thread, or batches on several host threads).
    hipStream_t io = nullptr;
    double kernelMsAccum = 0.0;  // launches before the last one (an overflow re-run)
    // Host-to-device copies go through this pinned buffer of the shell's own.  A copy from the caller's pageable memory
    // makes the runtime pin and unpin those pages around the transfer -- GPU page-table updates that stall whatever
    // kernel is running: a batch uploaded beside another batch's sweep cost that sweep 25-30 ms
    // (profiles/r02_interference.txt).
    void *hStage = nullptr;
    size_t hStageBytes = 0;
    // every class but the first runs beside it on a stream of its own (fork / join around cpk_device_run)
    hipStream_t sideStream[kMaxClasses] = {};
    hipEvent_t sideDone[kMaxClasses] = {};
    bool ran = false;
};

extern "C" int cpk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int cpk_current_device(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    return dev;
}

static void shell_delete(CpkDevice *d) {  // the shell's device is current
    if (d->hStage) (void)hipHostFree(d->hStage);
    if (d->evStart) (void)hipEventDestroy(d->evStart);
    if (d->evStop) (void)hipEventDestroy(d->evStop);
    if (d->evA) (void)hipEventDestroy(d->evA);
    if (d->evB) (void)hipEventDestroy(d->evB);
    if (d->evUp0) (void)hipEventDestroy(d->evUp0);
    if (d->evUp1) (void)hipEventDestroy(d->evUp1);
    if (d->io) (void)hipStreamDestroy(d->io);
    for (int k = 0; k < kMaxClasses; k++) {
        if (d->sideStream[k]) (void)hipStreamDestroy(d->sideStream[k]);
        if (d->sideDone[k]) (void)hipEventDestroy(d->sideDone[k]);
    }
    delete d;
}

static int shell_init(CpkDevice *d, int device) {
    d->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    d->numCUs = prop.multiProcessorCount;
    HIP_TRY(hipEventCreate(&d->evStart));
    HIP_TRY(hipEventCreate(&d->evStop));
    HIP_TRY(hipEventCreate(&d->evA));
    HIP_TRY(hipEventCreate(&d->evB));
    HIP_TRY(hipEventCreate(&d->evUp0));
    HIP_TRY(hipEventCreate(&d->evUp1));
    {
        // The batch's own stream (copies, the table build, gather and consumers) has the high priority: its small kernels
        // then take the wave slots a draining sweep of ANOTHER batch gives back before that sweep's own queued workgroups
        // do -- the table build of batch k + 1 runs beside the tail of sweep k instead of behind it (CPECAN_IO_PRIORITY=0:
        // the default priority, as rounds 1-3).
        int prLow = 0, prHigh = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
        const char *env = getenv("CPECAN_IO_PRIORITY");
        if (env && atoi(env) == 0) prHigh = 0;
        HIP_TRY(hipStreamCreateWithPriority(&d->io, hipStreamNonBlocking, prHigh));
    }
    // The side streams of a multi-class batch are created when a batch first needs them (cpk_device_run): the runtime
    // maps streams onto a handful of hardware queues (4 by default), and ten idle streams per shell put a batch's sweep
    // into the same hardware queue as another batch's copies -- a pipeline of batches then ran its sweeps ~100 ms late
    // (profiles/r02_pipeline_trace.txt).
    return CPECAN_OK;
}

extern "C" int cpk_device_create(CpkDevice **out, int device) {
    int n = cpk_device_count();
    if (n <= 0 || device < 0 || device >= n) {
        cpk_set_error("no usable HIP device (count=%d, requested=%d): the HIP path has no CPU fallback", n, device);
        return CPECAN_ENODEVICE;
    }
    CPK_ON_DEVICE(device);
    if (device < kMaxDevices) {  // an idle shell of an earlier batch: its streams and events are ready
        std::lock_guard<std::mutex> lock(g_cacheMutex);
        if (!g_shells[device].empty()) {
            *out = g_shells[device].back();
            g_shells[device].pop_back();
            g_liveShells[device]++;
            reaper_disarm(device);
            return CPECAN_OK;
        }
    }
    CpkDevice *d = new CpkDevice();
    if (int rc = shell_init(d, device)) {
        shell_delete(d);  // whatever was created before the failure
        return rc;
    }
    if (device < kMaxDevices) {
        g_liveShells[device]++;
        reaper_disarm(device);
    }
    *out = d;
    return CPECAN_OK;
}

// Idle device blocks of `device` (-1: every device) and idle host blocks go back to the driver / the OS.  For a process
// that shares the GPU with another allocator (torch, RCCL): call it when a phase of alignment work is over.
extern "C" int64_t cpk_cache_trim(int device) {
    int64_t freed = 0;
    const int n = cpk_device_count();
    for (int dev = 0; dev < n && dev < kMaxDevices; dev++) {
        if (device >= 0 && dev != device) continue;
        {  // a device this process never used has nothing cached: do not create a context on it
            std::lock_guard<std::mutex> lock(g_cacheMutex);
            if (g_blockCache[dev].blocks.empty()) continue;
        }
        DeviceGuard guard(dev);
        if (guard.err != hipSuccess) continue;
        freed += (int64_t)cache_trim(dev, 0);
    }
    freed += (int64_t)host_trim();
    return freed;
}

// Waits for everything this batch has in flight: its sweep launches (the stop event on the caller's stream) and its own
// stream.  Nothing else ever touches the batch's blocks, so they may be recycled afterwards -- without draining the
// device, which may be busy with another batch.
static void batch_quiesce(CpkDevice *d) {
    if (d->ran) (void)hipEventSynchronize(d->evStop);
    if (d->io) (void)hipStreamSynchronize(d->io);
}

static void free_all(CpkDevice *d) {
    if (!d->allocs.empty() || d->dCompact || d->dChunks) batch_quiesce(d);  // nothing in flight uses them
    for (const CachedBlock &b : d->allocs) cache_free(d->device, b.ptr, b.bytes);
    d->allocs.clear();
    cache_free(d->device, d->dCompact, d->compactBytes);
    cache_free(d->device, d->dChunks, d->chunkBytes);
    d->compactBytes = d->chunkBytes = 0;
    d->dRegions = nullptr; d->dDiags = nullptr; d->dDiagPos = nullptr; d->dSegs = nullptr; d->dSymbols = nullptr; d->dModel = nullptr;
    d->dRing = d->dC = d->dM = d->dTotals = d->dGroll = d->dBring = nullptr;
    d->dCand = nullptr;
    d->dForward = nullptr;
    d->dExpect = nullptr;
    d->dCounts = d->dSegStarts = d->dSegCounts = d->dTriples = nullptr;
    cpk_host_free(d->hostCounts);
    d->hostCounts = nullptr;
    d->dItems = nullptr;
    d->dProgress = nullptr;
    d->dCompact = nullptr; d->dChunks = nullptr; d->compactCap = d->chunkCap = 0;
    d->dQueue = nullptr;
    d->dDbgFb = d->dDbgTotals = nullptr;
    d->bytes = 0;
}

static void ran_set(CpkDevice *d, bool ran) {
    if (d->ran != ran && d->device >= 0 && d->device < kMaxDevices) g_ranAlive[d->device] += ran ? 1 : -1;
    d->ran = ran;
}

extern "C" void cpk_device_destroy(CpkDevice *d) {
    if (!d) return;
    DeviceGuard guard(d->device);
    free_all(d);
    if (d->device >= 0 && d->device < kMaxDevices) {
        // When the last batch leaves a device its large idle blocks (a config-B batch's rings are ONE block of 60 GB) go
        // back to the driver -- not now (round 3 did, and the next big batch paid seconds of hipMalloc for it) but once
        // the device has stayed without a live batch for CPECAN_CACHE_IDLE_S: the idle reaper above.
        const bool last = --g_liveShells[d->device] <= 0;
        if (last) reaper_arm(d->device);
    }
    if (d->device < kMaxDevices) {  // keep the shell (streams, events) for the next batch on this device
        d->classes.clear();
        ran_set(d, false);
        d->lastStream = nullptr;
        std::lock_guard<std::mutex> lock(g_cacheMutex);
        if (g_shells[d->device].size() < 16) {
            g_shells[d->device].push_back(d);
            return;
        }
    }
    shell_delete(d);
}

template <typename T>
static int dev_alloc(CpkDevice *d, T **p, size_t count) {
    size_t bytes = count * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(cache_alloc(d->device, (void **)p, bytes));
    d->allocs.push_back({(void *)*p, bytes});
    d->bytes += (int64_t)bytes;
    return CPECAN_OK;
}
// gives one block of the batch back (the device is idle)
static void dev_release(CpkDevice *d, void *p) {
    for (size_t i = 0; i < d->allocs.size(); i++)
        if (d->allocs[i].ptr == p) {
            cache_free(d->device, p, d->allocs[i].bytes);
            d->bytes -= (int64_t)d->allocs[i].bytes;
            d->allocs[i] = d->allocs.back();
            d->allocs.pop_back();
            return;
        }
}

static KernelFn pick_packed_kernel(const CpkGeometry &g, int cls, bool dynamic) {  // class k: groups of 8 << k lanes
    const bool five = g.nStates == 5;
#define CPK_PICK_PACKED(E, D)                                                                                               \
    if (g.emit == (E) && dynamic == (D)) switch (cls) {                                                                     \
            case 0: return five ? cpecan_pairhmm_packed<5, 8, (E), (D)> : cpecan_pairhmm_packed<3, 8, (E), (D)>;            \
            case 1: return five ? cpecan_pairhmm_packed<5, 16, (E), (D)> : cpecan_pairhmm_packed<3, 16, (E), (D)>;          \
            case 2: return five ? cpecan_pairhmm_packed<5, 32, (E), (D)> : cpecan_pairhmm_packed<3, 32, (E), (D)>;          \
        }
    CPK_PICK_PACKED(CPECAN_EMIT_MATCH, false)
    CPK_PICK_PACKED(CPECAN_EMIT_INDEL, false)
    CPK_PICK_PACKED(CPECAN_EMIT_EXPECT, false)
    CPK_PICK_PACKED(CPECAN_EMIT_MATCH, true)  // per-anchor expansions
    CPK_PICK_PACKED(CPECAN_EMIT_INDEL, true)
    CPK_PICK_PACKED(CPECAN_EMIT_EXPECT, true)
#undef CPK_PICK_PACKED
    return nullptr;
}
// the two kernels of a split packed class (match emitter, fixed expansion; cpk_packed.inl, MODE)
static void pick_packed_split_kernels(const CpkGeometry &g, int cls, KernelFn *fwd, KernelFn *trace) {
    const bool five = g.nStates == 5;
#define CPK_PICK_PACKED_SPLIT(GW)                                                                                                \
    {                                                                                                                            \
        *fwd = five ? cpecan_pairhmm_packed<5, GW, CPECAN_EMIT_MATCH, false, kModeForward> : cpecan_pairhmm_packed<3, GW, CPECAN_EMIT_MATCH, false, kModeForward>; \
        *trace = five ? cpecan_pairhmm_packed<5, GW, CPECAN_EMIT_MATCH, false, kModeTrace> : cpecan_pairhmm_packed<3, GW, CPECAN_EMIT_MATCH, false, kModeTrace>;   \
    }
    switch (cls) {
        case 0: CPK_PICK_PACKED_SPLIT(8) break;
        case 1: CPK_PICK_PACKED_SPLIT(16) break;
        default: CPK_PICK_PACKED_SPLIT(32) break;
    }
#undef CPK_PICK_PACKED_SPLIT
}

// the two kernels of a split class (match emitter)
// Doubles of the ring a split region keeps its forward values in.  The match emitter stores the match row of every
// diagonal and every state only where the traceback reads it back: diagonal 0, the refresh diagonals of the emitting
// segment (one in CPK_REFRESH_PERIOD) and the two diagonals a forward sweep would resume from -- the table builder lays
// the diagonals end to end with exactly that many doubles each (cpk_table_gather.inl); this is the upper bound it stays
// below.  (Every state of every cell, as the per-wave rings are sized, is 204 GB for BASELINE config B; this is 62.)
static int64_t split_ring_doubles(const CpkRegion &rg, int S) {
    const int64_t N = (int64_t)rg.lX + rg.lY;
    const int64_t fullDiags = N / CPK_REFRESH_PERIOD + 3 * (int64_t)rg.nSeg + 4;  // refresh points + two resume diagonals per segment
    // + one double of padding per diagonal (match rows start and end on even doubles); an even total keeps the next region's ring aligned
    return ((int64_t)rg.cells + (N + 1) + (int64_t)(S - 1) * rg.maxWidth * fullDiags + S + 1) & ~(int64_t)1;
}
// dense: the three-state match kernels allocated for three waves per SIMD (cpk_sweep.inl, WPS)
static KernelFn pick_fused_kernel(const CpkGeometry &g, bool dense, bool abs, bool three = false) {
    const bool fast = !g.useGlobalRoll;
    if (abs && fast) {
        // three (round 4): built for three waves per SIMD, for classes whose LDS lets nine or more waves onto a CU
        if (g.nStates == 5)
            return three ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeFused, 3, true>
                         : cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeFused, CPK_SWEEP_WAVES, true>;
        return dense ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeFused, 3, true>
                     : cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeFused, CPK_SWEEP_WAVES, true>;
    }
    if (g.nStates == 5)
        return fast ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeFused> : cpecan_pairhmm_sweep<5, false, CPECAN_EMIT_MATCH, kModeFused>;
    if (dense)
        return fast ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeFused, 3> : cpecan_pairhmm_sweep<3, false, CPECAN_EMIT_MATCH, kModeFused, 3>;
    return fast ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeFused> : cpecan_pairhmm_sweep<3, false, CPECAN_EMIT_MATCH, kModeFused>;
}
static void pick_split_kernels(const CpkGeometry &g, bool dense, bool abs, KernelFn *fwd, KernelFn *trace, bool fwd3 = false,
                               bool trace3 = false) {
    const bool fast = !g.useGlobalRoll;
    if (abs && fast) {
        // fwd3: the forward launch has no candidate ring and 120 VGPRs or fewer; where its LDS lets nine or more waves
        // onto a CU it runs the build for three waves per SIMD (profiles/r03_occupancy_3_waves_per_simd.txt)
        if (g.nStates == 5) {
            *fwd = fwd3 ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeForward, 3, true>
                        : cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeForward, CPK_SWEEP_WAVES, true>;
            // trace3 (round 4): the traceback launch built for three waves per SIMD (168 VGPRs), for classes whose LDS lets
            // nine or more waves onto a CU
            *trace = trace3 ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeTrace, 3, true>
                            : cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeTrace, CPK_SWEEP_WAVES, true>;
        } else {
            *fwd = fwd3 ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeForward, 3, true>
                        : cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeForward, CPK_SWEEP_WAVES, true>;
            *trace = dense ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeTrace, 3, true>
                           : cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeTrace, CPK_SWEEP_WAVES, true>;
        }
        return;
    }
    if (g.nStates == 5) {
        *fwd = fast ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeForward> : cpecan_pairhmm_sweep<5, false, CPECAN_EMIT_MATCH, kModeForward>;
        *trace = fast ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_MATCH, kModeTrace> : cpecan_pairhmm_sweep<5, false, CPECAN_EMIT_MATCH, kModeTrace>;
    } else {
        // the forward-only kernel needs 70 VGPRs: one variant
        *fwd = fast ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeForward> : cpecan_pairhmm_sweep<3, false, CPECAN_EMIT_MATCH, kModeForward>;
        if (dense)
            *trace = fast ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeTrace, 3> : cpecan_pairhmm_sweep<3, false, CPECAN_EMIT_MATCH, kModeTrace, 3>;
        else
            *trace = fast ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeTrace> : cpecan_pairhmm_sweep<3, false, CPECAN_EMIT_MATCH, kModeTrace>;
    }
}
static KernelFn pick_dense_kernel(const CpkGeometry &g) {  // one wave per region, three-state match emitter
    return !g.useGlobalRoll ? cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_MATCH, kModeWhole, 3> : cpecan_pairhmm_sweep<3, false, CPECAN_EMIT_MATCH, kModeWhole, 3>;
}

static KernelFn pick_kernel(const CpkGeometry &g) {
    const bool fast = !g.useGlobalRoll;  // second template argument = FAST (LDS rolling buffers + LDS symbol strings)
    if (g.emit == CPECAN_EMIT_EXPECT && fast && g.expInSweep == 1)  // no diagonal wider than one 64-lane group
        return g.nStates == 5 ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_EXPECT, kModeWhole, CPK_SWEEP_WAVES, false, 1>
                              : cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_EXPECT, kModeWhole, CPK_SWEEP_WAVES, false, 1>;
    if (g.emit == CPECAN_EMIT_EXPECT && fast && g.expInSweep)
        return g.nStates == 5 ? cpecan_pairhmm_sweep<5, true, CPECAN_EMIT_EXPECT, kModeWhole, CPK_SWEEP_WAVES, false, 2>
                              : cpecan_pairhmm_sweep<3, true, CPECAN_EMIT_EXPECT, kModeWhole, CPK_SWEEP_WAVES, false, 2>;
#define CPK_PICK(E)                                                                                          \
    if (g.emit == (E)) {                                                                                     \
        if (g.nStates == 5) return fast ? cpecan_pairhmm_sweep<5, true, (E)> : cpecan_pairhmm_sweep<5, false, (E)>; \
        return fast ? cpecan_pairhmm_sweep<3, true, (E)> : cpecan_pairhmm_sweep<3, false, (E)>;              \
    }
    CPK_PICK(CPECAN_EMIT_MATCH)
    CPK_PICK(CPECAN_EMIT_INDEL)
    CPK_PICK(CPECAN_EMIT_EXPECT)
    CPK_PICK(kEmitForward)
#undef CPK_PICK
    return nullptr;
}

// Queues dst <- src (host) on the batch's stream through the shell's pinned buffer; `at` is the running offset in it.
constexpr size_t kStageMaxCopy = (size_t)64 << 20;  // larger sources are never copied through the shell's staging buffer
static int staged_h2d(CpkDevice *d, void *dst, const void *src, size_t bytes, size_t *at) {
    if (bytes == 0) return CPECAN_OK;
    // a pinned block of the host pool: the copy engine reads it where it lies.  A large block in its first life is not
    // pinned yet: the runtime copies it the pageable way (through its own staging), which only hurts beside a running
    // sweep -- and in a pipeline the blocks are in their second life.
    if (bytes > kStageMaxCopy || host_is_pinned(src, bytes)) {
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, d->io));
        return CPECAN_OK;
    }
    const size_t off = (*at + 255) / 256 * 256;
    if (off + bytes > d->hStageBytes) {
        cpk_set_error("internal: staging buffer too small");
        return CPECAN_ESTATE;
    }
    memcpy((char *)d->hStage + off, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, (char *)d->hStage + off, bytes, hipMemcpyHostToDevice, d->io));
    *at = off + bytes;
    return CPECAN_OK;
}
static int stage_reserve(CpkDevice *d, size_t bytes) {
    if (bytes <= d->hStageBytes) return CPECAN_OK;
    HIP_TRY(hipStreamSynchronize(d->io));
    if (d->hStage) (void)hipHostFree(d->hStage);
    d->hStage = nullptr;
    d->hStageBytes = 0;
    const size_t want = (bytes + (bytes >> 2) + (1u << 20)) / 4096 * 4096;
    HIP_TRY(hipHostMalloc(&d->hStage, want, hipHostMallocDefault));
    d->hStageBytes = want;
    return CPECAN_OK;
}

extern "C" int cpk_device_upload(CpkDevice *d, const CpkGeometry *geo, const CpkModel *model, CpkRegion *regions,
                                 const cpk_anchor_t *anchors, int anchorStride, int64_t nAnchors, const int32_t *runs, int64_t nRuns,
                                 int64_t nDiags, int64_t expansion, int dynamic,
                                 const CpkSegment *segs, int64_t nSegs, const uint8_t *symbols, int64_t nSymbolBytes,
                                 int64_t outTriplesPerList, int nLists, int64_t dbgCells, int64_t dbgDiags,
                                 double *h2dMs) {
    CPK_ON_DEVICE(d->device);
    free_all(d);
    d->kernelMsAccum = 0.0;
    d->fusedRetried = false;
    d->geo = *geo;
    d->kc = KConsts{model->matchContinue, model->matchFromShortX, model->matchFromShortY, model->matchFromLongX,
                    model->matchFromLongY, model->shortOpenX, model->shortOpenY, model->shortExtendX,
                    model->shortExtendY, model->shortSwitchToX, model->shortSwitchToY, model->longOpenX,
                    model->longOpenY, model->longExtendX, model->longExtendY, model->threshold};
    d->nLists = nLists;
    d->nSegs = nSegs;
    d->nDiags = nDiags;
    d->outTriplesPerList = outTriplesPerList;
    d->dbgCells = dbgCells;
    d->dbgDiags = dbgDiags;
    ran_set(d, false);
    const int S = geo->nStates;

    // ---- the launches of a run: one per size class that has regions ----
    d->classes.clear();
    // Resident single-wave workgroups per CU.  hipOccupancyMaxActiveBlocksPerMultiprocessor answers 3 for these
    // 64-thread kernels (it reports waves per SIMD), so the bound is computed from the register file and LDS
    // directly (MI355X_MICROARCH.md: 512 VGPRs per lane per SIMD in granules of 8, 4 SIMDs, 32 waves, 160 KiB LDS).
    // Over-estimating is harmless: surplus workgroups simply queue, every wave exits when the work queue is empty.
    auto wavesPerCU = [&](KernelFn fn, size_t ldsBytes, int *out) -> int {
        hipFuncAttributes attr;
        HIP_TRY(hipFuncGetAttributes(&attr, (const void *)fn));
        const int vgprAlloc = ((attr.numRegs > 0 ? attr.numRegs : 128) + 7) / 8 * 8;
        int perSimd = 512 / vgprAlloc;
        if (perSimd > 8) perSimd = 8;
        if (perSimd < 1) perSimd = 1;
        int perCU = 4 * perSimd;
        const size_t ldsTotal = ldsBytes + (size_t)attr.sharedSizeBytes;
        const int byLds = (int)((160 * 1024) / (ldsTotal ? ldsTotal : 1));
        if (byLds < perCU) perCU = byLds;
        if (perCU > 32) perCU = 32;
        if (const char *cap = getenv("CPECAN_MAX_WAVES_PER_CU")) {  // tuning/diagnostic knob
            const int c = atoi(cap);
            if (c >= 1 && c < perCU) perCU = c;
        }
        *out = perCU;
        return CPECAN_OK;
    };
    const int nCandLists = geo->emit == CPECAN_EMIT_INDEL ? 3 : 1;
    const bool expect = geo->emit == CPECAN_EMIT_EXPECT;
    int regionAt = 0;
    std::vector<LaunchClass> packedSplit, packedWhole;
    for (int k = 0; k < 3; k++) {  // narrow regions come first in the device order: the packed kernel, 64 / GW to a wave
        if (geo->nPacked[k] <= 0) continue;
        LaunchClass c;
        c.packed = true;
        c.k = k;
        c.fn = pick_packed_kernel(*geo, k, dynamic != 0);
        if (!c.fn) {
            cpk_set_error("no packed kernel for emitter %d", geo->emit);
            return CPECAN_EINVAL;
        }
        const int GW = 8 << k, G = CPK_WAVE / GW;
        c.geo = *geo;
        c.geo.ringCells = geo->pRingCells[k];
        c.geo.fbCells = geo->pFbCells[k];
        c.geo.maxRefresh = geo->pMaxRefresh[k];
        c.geo.refreshCells = (int64_t)GW * geo->pMaxRefresh[k];
        c.ldsBytes = sizeof(double) * (size_t)(kLdsCubics + kLdsEm + kLdsWeights + (expect ? kExpectCopies * 80 : 0)) +
                     (size_t)G * pack_group_bytes(S, GW);
        int perCU = 0;
        if (int rc = wavesPerCU(c.fn, c.ldsBytes, &perCU)) return rc;
        const int64_t slots = (int64_t)perCU * d->numCUs;
        c.ringEl = c.geo.ringCells * S;
        c.candEl = c.geo.fbCells;
        c.refEl = c.geo.refreshCells;
        c.totEl = c.geo.maxRefresh;
        // B of a segment's emitted cells: the expectation step's second pass, the indel emitter's list pass (cpk_packed.inl)
        c.bringEl = (expect || geo->emit == CPECAN_EMIT_INDEL) ? c.geo.fbCells * S : 0;
        if (geo->emit == CPECAN_EMIT_INDEL) c.candEl = 0;  // no candidates
        // Split (round 4, cpk_packed.inl "MODE"): forward sweeps into rings of the regions' own, then one queue item per
        // (region, traceback segment).  Worth it where ONE group's walk through the longest region -- its forward and its
        // backward steps, one after the other -- is what a launch of whole regions waits for: a realign-style batch of
        // 100-5000 bp alignments took 20 ms with 6 000 pairs and 29 with 50 000 (profiles/r04_config4_chain_bound.txt).
        // Then the LONG regions of the class -- the device order is longest first -- become a split class of their own,
        // launched first: their forward chains (half the steps) run beside the whole-region waves of the shorter ones, and
        // their tracebacks, side by side, behind.  Long: more than half the diagonals of the longest, so that the whole
        // regions' chains (2 N steps) are no longer than the longest forward chain.  CPECAN_PACKED_SPLIT=1 / 0 (tests, A/B
        // runs): every region of every packed class / none; CPECAN_PACKED_SPLIT_FROM=<diagonals>: regions longer than that.
        const int64_t base = regionAt, n = geo->nPacked[k];
        regionAt += geo->nPacked[k];
        int64_t cut = 0;  // the first `cut` regions of the class run split
        {
            int64_t stepsAll = 0, nMax = 0;
            int32_t segMax = 0;
            for (int64_t di = base; di < base + n; di++) {
                const int64_t N = (int64_t)regions[di].lX + regions[di].lY;
                segMax = regions[di].nSeg > segMax ? regions[di].nSeg : segMax;
                stepsAll += 2 * N;
                nMax = N > nMax ? N : nMax;
            }
            const char *env = getenv("CPECAN_PACKED_SPLIT"), *fromEnv = getenv("CPECAN_PACKED_SPLIT_FROM");
            const bool eligible = geo->emit == CPECAN_EMIT_MATCH && !dynamic && !geo->debug;
            // (the steps of a wave if the class's steps were dealt out evenly over every wave slot of the chip)
            const int64_t balanced = stepsAll / (G * slots) + 1;
            // ... and only for a batch that has the device to itself: with other batches of the process in flight (a
            // pipeline) their waves fill the slots a chain leaves idle, and the split form -- rings of whole regions in HBM
            // instead of a cache-resident ring per group, six launches instead of two -- costs throughput: config 4 end to end,
            // four batches in flight, 35-37 ms per batch with whole regions against 43 split (profiles/r04_config4_chain_bound.txt)
            const bool lone = !(d->device >= 0 && d->device < kMaxDevices && g_ranAlive[d->device] > 0);
            if (eligible && env && atoi(env) != 0) cut = n;
            else if (eligible && !(env && atoi(env) == 0) && (fromEnv || (lone && segMax >= 3 && 2 * nMax * 2 >= balanced * 3))) {
                const int64_t from = fromEnv ? atoll(fromEnv) : nMax / 2;
                for (int64_t di = base; di < base + n; di++)  // (ordered by cells, not by diagonals: up to the last long one)
                    if ((int64_t)regions[di].lX + regions[di].lY > from) cut = di - base + 1;
            }
        }
        for (int part = 0; part < 2; part++) {
            const int64_t pBase = part == 0 ? base : base + cut, pCount = part == 0 ? cut : n - cut;
            if (pCount <= 0) continue;
            LaunchClass cc = c;
            cc.regionBase = (int)pBase;
            cc.regionCount = (int)pCount;
            int64_t waves = (pCount + G - 1) / G;
            if (waves > slots) waves = slots;
            cc.waves = (int)waves;
            cc.subSlots = waves * G;
            if (part == 0) {
                int64_t nSegPart = 0;
                for (int64_t di = pBase; di < pBase + pCount; di++) nSegPart += regions[di].nSeg;
                cc.split = true;
                pick_packed_split_kernels(cc.geo, k, &cc.fn, &cc.fnTrace);
                int64_t wt = (nSegPart + G - 1) / G;
                if (wt > slots) wt = slots;
                cc.wavesTrace = (int)wt;
                if (wt * G > cc.subSlots) cc.subSlots = wt * G;
                cc.itemCount = nSegPart;
            }
            if (getenv("CPECAN_TRACE_HOST"))
                fprintf(stderr, "cpecan packed class %d: %d regions in groups of %d lanes, LDS %zu B, waves %d / %d, %s\n", k, cc.regionCount, GW,
                        cc.ldsBytes, cc.waves, cc.wavesTrace, cc.split ? "two launches" : "whole regions");
            (part == 0 ? packedSplit : packedWhole).push_back(cc);
        }
    }
    {
        // With other batches of the process alive (a pipeline) the packed launches of a batch SHARE the chip's wave slots in
        // proportion to what each would ask for alone, instead of each asking for all of them, and one slot per CU in eight
        // stays free: the waves are persistent, so fewer of them lose nothing, nothing waits in the hardware queues behind
        // them, and the small kernels of the batches around this one (fills, table build, gather, consumers) find a slot
        // without waiting for a class to drain.  BASELINE config 4 end to end, six batches in flight, ten runs each over four
        // calls: 2.16e10 -> 2.33e10 cells/s on average, single runs spread +-10 % either way
        // (profiles/r04_config4_chain_bound.txt).  CPECAN_PACKED_SHARE=0: as before.
        const char *shareEnv = getenv("CPECAN_PACKED_SHARE");
        if (!(shareEnv && atoi(shareEnv) == 0) && d->device >= 0 && d->device < kMaxDevices && g_ranAlive[d->device] > 0) {
            int64_t total = 0, room = 0;
            for (const LaunchClass &cc : packedSplit) total += cc.waves;
            for (const LaunchClass &cc : packedWhole) total += cc.waves;
            int perCU0 = 0;
            for (const LaunchClass &cc : packedSplit) if (!perCU0) wavesPerCU(cc.fnTrace, cc.ldsBytes, &perCU0);
            for (const LaunchClass &cc : packedWhole) if (!perCU0) wavesPerCU(cc.fn, cc.ldsBytes, &perCU0);
            room = (int64_t)perCU0 * d->numCUs - d->numCUs / 8;
            if (total > room && room > 0) {
                auto scale = [&](LaunchClass &cc) {
                    const int G = CPK_WAVE / (8 << cc.k);
                    int64_t w = (int64_t)cc.waves * room / total;
                    cc.waves = (int)(w < 1 ? 1 : w);
                    if (cc.split) {
                        int64_t wt = (int64_t)cc.wavesTrace * room / total;
                        cc.wavesTrace = (int)(wt < cc.waves ? cc.waves : wt);
                        if (cc.wavesTrace > (int)((cc.itemCount + G - 1) / G)) cc.wavesTrace = (int)((cc.itemCount + G - 1) / G);
                    }
                    const int64_t mx = cc.waves > cc.wavesTrace ? cc.waves : cc.wavesTrace;
                    cc.subSlots = mx * G;
                };
                for (LaunchClass &cc : packedSplit) scale(cc);
                for (LaunchClass &cc : packedWhole) scale(cc);
            }
        }
    }
    // the split parts first: their forward chains are the longest thing in the batch and start before anything else
    for (const LaunchClass &cc : packedSplit) d->classes.push_back(cc);
    for (const LaunchClass &cc : packedWhole) d->classes.push_back(cc);
    // The LDS of one wave of the class: tables, rolling rows, candidate stage, symbols -- by the form of its sweeps.
    // Absolute positions (cpk_sweep.inl): two arrays of S rows with a few positions of slack, a stage of 64
    // candidates, and the symbols of one traceback segment at a time instead of both whole strings.
    auto setForm = [&](LaunchClass &cc, bool abs) {
        cc.abs = abs;
        cc.geo.rollStride = cc.geo.maxWidth + (abs ? kAbsSlack : 1);
        const char *winEnv = getenv("CPECAN_ABS_WINDOWS");  // 0: the absolute-position sweeps stage whole strings (diagnostics, tests)
        cc.geo.reserved0 = (abs && winEnv && atoi(winEnv) == 0) ? 1 : 0;
        cc.geo.seqLdsBytes = (abs && !cc.geo.reserved0) ? geo->wWinLdsBytes[cc.k] : geo->wSeqLdsBytes[cc.k];
        cc.geo.rollDoubles = (int64_t)(abs ? 2 * S : 2 * S + 1) * cc.geo.rollStride;
        const size_t header = sizeof(double) * (lds_header_doubles(geo->emit) + lds_stage_doubles(geo->emit, abs));
        cc.ldsBytes = cc.geo.useGlobalRoll ? header
                                           : header + sizeof(double) * (size_t)cc.geo.rollDoubles + (size_t)((cc.geo.seqLdsBytes + 15) / 16 * 16);
    };
    for (int k = 0; k < CPK_WIDE_CLASSES; k++) {  // then the wide ones: the sweep kernel, one region per wave at a time
        if (geo->nWide[k] <= 0) continue;
        LaunchClass c;
        c.k = k;
        c.geo = *geo;
        c.geo.maxWidth = geo->wMaxWidth[k];
        c.geo.maxRefresh = geo->wMaxRefresh[k];
        c.geo.ringCells = geo->wRingCells[k];
        c.geo.fbCells = geo->wFbCells[k];
        c.geo.seqLdsBytes = geo->wSeqLdsBytes[k];
        c.geo.rollStride = c.geo.maxWidth + 1;
        // Will the class run split (its tracebacks as queue items; decided below, once the occupancy is known)?  Then, with
        // a fixed expansion and bands whose edges move one step per diagonal (CpkRegion::absOk), its sweeps index the
        // rolling rows by absolute position (cpk_sweep.inl): the rows need a few positions of slack.
        // CPECAN_ABS=0: never (A/B runs, tests of the other form).
        int64_t nSegClass = 0;
        bool absOk = geo->emit == CPECAN_EMIT_MATCH && !dynamic;
        for (int64_t di = regionAt; di < regionAt + geo->nWide[k]; di++) {
            nSegClass += regions[di].nSeg;
            absOk = absOk && regions[di].absOk;
        }
        {
            const char *env = getenv("CPECAN_SPLIT"), *absEnv = getenv("CPECAN_ABS");
            const bool splitLikely = geo->emit == CPECAN_EMIT_MATCH && (!geo->debug || env) && nSegClass > 0 &&
                                     (env ? atoi(env) != 0 : nSegClass * 4 >= (int64_t)geo->nWide[k] * 5);
            c.abs = splitLikely && absOk && !(absEnv && atoi(absEnv) == 0);
        }
        c.geo.refreshCells = (int64_t)c.geo.maxWidth * c.geo.maxRefresh;
        if (c.geo.refreshCells < 1) c.geo.refreshCells = 1;
        // LDS budget: beyond 64 KiB per wave (rolling buffers + symbol strings, in the form every class can fall back
        // to: one wave per region) the class takes the global-memory path
        const bool absWanted = c.abs;
        c.geo.useGlobalRoll = 0;
        setForm(c, false);
        c.geo.useGlobalRoll = c.ldsBytes + 16 > 64 * 1024;
        setForm(c, absWanted && !c.geo.useGlobalRoll);  // absolute positions are a form of the LDS rows
        {
            // Expectation emitter, every diagonal of the class within two 64-lane groups: the events are formed inside the
            // traceback (Sweep::tracebackExpect) from three forward diagonals kept in LDS, instead of a second pass over B
            // values parked in global memory.  CPECAN_EXP_INSWEEP=0 (tests, A/B runs): the second pass everywhere; 2: inside
            // the traceback whatever the LDS costs.
            const char *env = getenv("CPECAN_EXP_INSWEEP");
            c.geo.expInSweep = expect && !c.geo.useGlobalRoll && c.geo.maxWidth <= 2 * CPK_WAVE /* Sweep::kExpGroups */ && !(env && atoi(env) == 0);
            // 1: the build unrolled for ONE group per diagonal (classes up to 64 cells -- the host gives the expectation
            // emitter a size class of its own there, cpecan_host.c); 2: for two.  CPECAN_EXP_ONE_GROUP=0: always the latter.
            const char *oneEnv = getenv("CPECAN_EXP_ONE_GROUP");
            if (c.geo.expInSweep) c.geo.expInSweep = (c.geo.maxWidth <= CPK_WAVE && !(oneEnv && atoi(oneEnv) == 0)) ? 1 : 2;
            if (c.geo.expInSweep) {
                const size_t with = c.ldsBytes + sizeof(double) * (size_t)3 * (c.geo.maxWidth + 1) * S + sizeof(double) * kExpectWinCopies * 80 -
                                    sizeof(double) * (size_t)(lds_header_doubles(geo->emit) - lds_header_doubles(geo->emit, true));
                // ... as long as the three forward diagonals in LDS do not cost a resident wave: the kernel's registers
                // allow eight per CU (five-state: bands up to 74 cells, three-state: up to ~120; measured at 66 and 106)
                if (with <= 160 * 1024 / 8 || (env && atoi(env) >= 2)) c.ldsBytes = with;
                else c.geo.expInSweep = 0;
            }
        }
        c.fn = pick_kernel(c.geo);
        if (!c.fn) {
            cpk_set_error("no kernel for emitter %d", geo->emit);
            return CPECAN_EINVAL;
        }
        int perCU = 0;
        // Bands of several hundred cells: a team of kTeamWaves waves per region (cpk_team.inl) instead of one wave
        constexpr int kTeamWaves = 4;
        // (the team kernel stages both whole strings and keeps 3 S rows of maxWidth + 1 positions, whatever form of the
        // rows the class would take with one wave per region: NOT c.geo.seqLdsBytes / rollStride, which setForm() above
        // may have set to the symbol windows and slack of the absolute-position sweeps)
        const int teamStride = c.geo.maxWidth + 1;
        const size_t teamLds = sizeof(double) * ((size_t)team_header_doubles(expect) + (size_t)3 * S * teamStride) +
                               (size_t)((geo->wSeqLdsBytes[k] + 15) / 16 * 16);
        // A class goes to teams where one wave per region is down to three waves per CU or fewer (measured: at four per
        // CU, ~400-cell bands, the single wave still wins by 13 %; at three, ~450 cells, the team wins by 30 %), or on
        // the global-memory variant.  CPECAN_TEAM=<cells> (tests, diagnostics): from that band width instead; 0: never.
        const char *teamEnv = getenv("CPECAN_TEAM");
        int soloPerCU = 0;
        if (int rc = wavesPerCU(c.fn, c.ldsBytes, &soloPerCU)) return rc;
        const bool wanted = teamEnv ? (atoi(teamEnv) > 0 && c.geo.maxWidth >= atoi(teamEnv))
                                    : (c.geo.maxWidth > 256 && (soloPerCU <= 3 || c.geo.useGlobalRoll));
        // one workgroup per CU is all the LDS allows from ~660 cells: then eight waves share the region
        const bool big = 2 * teamLds > 160 * 1024;
        if (wanted && (geo->emit == CPECAN_EMIT_MATCH || geo->emit == CPECAN_EMIT_INDEL || expect) && !geo->debug &&
            c.geo.maxWidth <= CPK_WAVE * kTeamWaves * (big ? 2 : 1) * kTeamGroups && teamLds <= 160 * 1024) {
            if (expect)  // (round 4: the expectation emitter -- its second pass shared by the team's waves)
                c.fn = S == 5 ? (big ? cpecan_pairhmm_team<5, 2 * kTeamWaves, CPECAN_EMIT_EXPECT> : cpecan_pairhmm_team<5, kTeamWaves, CPECAN_EMIT_EXPECT>)
                              : (big ? cpecan_pairhmm_team<3, 2 * kTeamWaves, CPECAN_EMIT_EXPECT> : cpecan_pairhmm_team<3, kTeamWaves, CPECAN_EMIT_EXPECT>);
            else if (geo->emit == CPECAN_EMIT_INDEL)  // (round 4: the three lists of the indel emitter from the team as well)
                c.fn = S == 5 ? (big ? cpecan_pairhmm_team<5, 2 * kTeamWaves, CPECAN_EMIT_INDEL> : cpecan_pairhmm_team<5, kTeamWaves, CPECAN_EMIT_INDEL>)
                              : (big ? cpecan_pairhmm_team<3, 2 * kTeamWaves, CPECAN_EMIT_INDEL> : cpecan_pairhmm_team<3, kTeamWaves, CPECAN_EMIT_INDEL>);
            else
            c.fn = S == 5 ? (big ? cpecan_pairhmm_team<5, 2 * kTeamWaves> : cpecan_pairhmm_team<5, kTeamWaves>)
                          : (big ? cpecan_pairhmm_team<3, 2 * kTeamWaves> : cpecan_pairhmm_team<3, kTeamWaves>);
            c.threads = CPK_WAVE * kTeamWaves * (big ? 2 : 1);
            c.geo.useGlobalRoll = 0;
            c.abs = false;
            c.geo.reserved0 = 0;
            c.geo.expInSweep = 0;  // (the team's expectation emitter is the second pass: B of the emitted cells in `bring`)
            c.geo.rollStride = teamStride;
            c.geo.seqLdsBytes = geo->wSeqLdsBytes[k];
            c.geo.rollDoubles = (int64_t)(2 * S + 1) * c.geo.rollStride;
            c.ldsBytes = teamLds;
            c.grollEl = 0;
            hipFuncAttributes attr;
            HIP_TRY(hipFuncGetAttributes(&attr, (const void *)c.fn));
            const int vgprAlloc = ((attr.numRegs > 0 ? attr.numRegs : 128) + 7) / 8 * 8;
            int perSimd = 512 / vgprAlloc;  // a team of four puts one wave on every SIMD, a team of eight two
            if (perSimd > 8) perSimd = 8;
            if (big) perSimd /= 2;
            const int byLds = (int)((160 * 1024) / (teamLds + (size_t)attr.sharedSizeBytes));
            perCU = perSimd < byLds ? perSimd : byLds;
            if (perCU < 1) perCU = 1;
        } else {
            perCU = soloPerCU;
            // Three-state match classes with more regions than two waves per SIMD hold take the kernels allocated for
            // three (168 VGPRs, a handful of spills): 4000 pairs of 1 kb -9 to -19 %, 2500 of 2 kb -17 %, with one wave
            // per region -34 %; a class that leaves slots empty anyway loses 3-10 % to the spills and the fuller SIMDs
            // (config A, 1000 pairs: 3.68 -> 4.07 ms) and keeps the 2-wave kernels.  CPECAN_DENSE=1 / 0: always / never.
            const char *denseEnv = getenv("CPECAN_DENSE");
            if (S == 3 && geo->emit == CPECAN_EMIT_MATCH && !geo->debug &&
                (denseEnv ? atoi(denseEnv) != 0 : geo->nWide[k] >= (int64_t)soloPerCU * d->numCUs)) {
                KernelFn f3 = pick_dense_kernel(c.geo);
                int p3 = 0;
                if (int rc = wavesPerCU(f3, c.ldsBytes, &p3)) return rc;
                if (p3 > perCU) {
                    c.fn = f3;
                    c.dense = true;
                    perCU = p3;
                }
            }
        }
        if (perCU < 1) {
            cpk_set_error("kernel does not fit on a CU (LDS %zu bytes)", c.ldsBytes);
            return CPECAN_EHIP;
        }
        const int64_t n = geo->nWide[k];
        int64_t waves = (int64_t)perCU * d->numCUs;
        if (waves > n) waves = n;
        {
            // Even out the rounds: with R = ceil(regions / waves) rounds, ceil(regions / R) waves do the same work in the
            // same number of rounds with fewer waves competing per SIMD (10 000 equal pairs: 1667 waves x 6 pairs instead
            // of 1792 waves of which 1044 do 6 and 748 do 5).
            const int64_t rounds = (n + waves - 1) / waves;
            const int64_t even = (n + rounds - 1) / rounds;
            if (even >= 1 && even < waves) waves = even;
        }
        c.waves = (int)waves;
        c.subSlots = waves;
        c.regionBase = regionAt;
        c.regionCount = geo->nWide[k];
        regionAt += geo->nWide[k];
        {
            // Split the class when its regions do not fill the chip and have tracebacks to hand out: the segments of a
            // region are independent once its forward values exist.  CPECAN_SPLIT=1 / 0 (tests, diagnostics): always / never.
            int64_t maxRing = 0;
            for (int64_t di = c.regionBase; di < c.regionBase + c.regionCount; di++) {
                const int64_t rd = split_ring_doubles(regions[di], S);
                if (rd > maxRing) maxRing = rd;
            }
            const char *env = getenv("CPECAN_SPLIT");
            const int64_t slots = (int64_t)perCU * d->numCUs;
            // (debug buffers: split only where CPECAN_SPLIT asks for it -- the per-cell parity tests of the split forms)
            const bool eligible = c.threads == CPK_WAVE && geo->emit == CPECAN_EMIT_MATCH && (!geo->debug || env) && nSegClass > 0;
            // Regions with tracebacks to hand out (1.25 segments on average and more) run split whenever their rings fit:
            // measured against one wave per region on 600 to 10 000 pairs of 1-4 kb and bands of 55-124 cells per diagonal,
            // one of the two split forms won every time (tools/split_forms.py, profiles/r02_split_forms.txt).  Which one:
            // the ONE-launch form fills the partly empty last round of forward sweeps with traceback items and wins where a
            // region has many segments (2 kb and longer: -20 to -40 %); with two or three segments per region (1 kb
            // pairs, config A) its device-scope ring accesses and waiting waves cost more than that gains (+7 to +19 %) and the
            // two launches win.  The rings of whole regions are given up first when device memory is short (below).
            const bool manySegs = nSegClass * 4 >= n * 5;
            const bool wanted = env ? atoi(env) != 0 : manySegs;
            // ... and up to ~4.5 rounds of forward sweeps: beyond, every region ticket is drawn before the first item anyway,
            // the overlap is down to the seam between the two phases, and the launches' plain ring stores win against
            // the one launch's write-through ones (2 kb pairs, band 100: 5000 / 6500 / 8000 pairs -8 / -8 / -2 % for the one
            // launch, 10 000 pairs -- config B -- +1.7 %: 89.3 against 87.6 ms, and 90.2 against 87.3 ms per pipelined batch)
            // (round 4, both forms at ten waves per CU: 5000 pairs -3 % for the one launch, 7000 pairs +3 %: ~3.25 rounds)
            const bool oneLaunch = nSegClass * 2 >= n * 7 /* 3.5 segments per region and more */ && n * 4 < slots * 13;
            if (eligible && wanted) {
                c.split = true;
                // CPECAN_SPLIT=2 / 1: force the one-launch (kModeFused) / two-launch form
                // (the one-launch form addresses a region's ring with 32-bit byte offsets: Sweep::ringPut)
                c.fused = (env ? atoi(env) == 2 : oneLaunch) && maxRing < ((int64_t)1 << 28);
                if (c.fused) {
                    // an item polls this often (s_sleep between polls: seconds in all) for its region's forward values; a
                    // count that never comes is reported and the class re-run in two launches (cpk_device_download).
                    // CPECAN_FUSED_SPIN: tests force that path with a bound of a few polls.
                    const char *spinEnv = getenv("CPECAN_FUSED_SPIN");
                    c.geo.fusedSpin = spinEnv ? atoi(spinEnv) : (1 << 24);
                    c.fn = pick_fused_kernel(c.geo, c.dense, c.abs);
                    int64_t slotsF = slots;
                    {
                        const char *t3e = getenv("CPECAN_FUSED3");  // 0: never the three-waves-per-SIMD build
                        if (c.abs && S == 5 && !(t3e && atoi(t3e) == 0) && (160 * 1024) / c.ldsBytes >= 9) {
                            KernelFn f3 = pick_fused_kernel(c.geo, c.dense, c.abs, true);
                            int p3 = 0;
                            if (int rc = wavesPerCU(f3, c.ldsBytes, &p3)) return rc;
                            if (p3 > perCU) {
                                c.fn = f3;
                                slotsF = (int64_t)p3 * d->numCUs;
                            }
                        }
                    }
                    // One CU in eight keeps a wave slot (and its 19 KB of LDS) free: a launch that fills every slot to its
                    // end starves the small kernels of the batch before it -- the list consumers need a few KB of LDS --
                    // until it drains, and a pipeline two batches deep then idles between sweeps (82 ms measured).
                    // ... so the slots are left free when another batch of this process has run on the device and is still
                    // alive; a batch on its own takes them all (config B: 90.3 -> 89.5 ms).
                    const int64_t spare = (d->device >= 0 && d->device < kMaxDevices && g_ranAlive[d->device] > 0) ? d->numCUs / 8 : 0;
                    const int64_t room = slotsF - spare > 0 ? slotsF - spare : slotsF;
                    int64_t wt = room < n + nSegClass ? room : n + nSegClass;
                    c.waves = (int)wt;
                    c.subSlots = wt;
                } else {
                    pick_split_kernels(c.geo, c.dense, c.abs, &c.fn, &c.fnTrace);
                    int64_t wt = slots < nSegClass ? slots : nSegClass;
                    c.wavesTrace = (int)wt;
                    if (wt > c.subSlots) c.subSlots = wt;
                    // the forward launch: no candidate ring in its LDS, and with absolute positions few enough registers
                    // for three waves per SIMD -- more waves per CU where that LDS allows them (CPECAN_FWD3=0: never)
                    c.ldsBytesFwd = c.geo.useGlobalRoll ? c.ldsBytes : c.ldsBytes - sizeof(double) * lds_stage_doubles(geo->emit, c.abs);
                    // ... and so does the traceback launch of a five-state class since round 4 (CPECAN_TRACE3=0: never)
                    const char *t3e = getenv("CPECAN_TRACE3");
                    if (c.abs && S == 5 && !(t3e && atoi(t3e) == 0) && (160 * 1024) / c.ldsBytes >= 9) {
                        KernelFn f2 = nullptr, tr3 = nullptr;
                        pick_split_kernels(c.geo, c.dense, c.abs, &f2, &tr3, false, true);
                        int p3 = 0;
                        if (int rc = wavesPerCU(tr3, c.ldsBytes, &p3)) return rc;
                        if (p3 > perCU) {
                            c.fnTrace = tr3;
                            int64_t w3 = (int64_t)p3 * d->numCUs;
                            if (w3 > nSegClass) w3 = nSegClass;
                            c.wavesTrace = (int)w3;
                            if (w3 > c.subSlots) c.subSlots = w3;
                        }
                    }
                    const char *f3e = getenv("CPECAN_FWD3");
                    if (c.abs && !(f3e && atoi(f3e) == 0) && (160 * 1024) / c.ldsBytesFwd >= 9) {
                        KernelFn f3 = nullptr, tr = nullptr;
                        pick_split_kernels(c.geo, c.dense, c.abs, &f3, &tr, true);
                        int p3 = 0;
                        if (int rc = wavesPerCU(f3, c.ldsBytesFwd, &p3)) return rc;
                        if (p3 > perCU) {
                            c.fn = f3;
                            int64_t wf = (int64_t)p3 * d->numCUs;
                            if (wf > n) wf = n;
                            const int64_t rounds = (n + wf - 1) / wf, even = (n + rounds - 1) / rounds;
                            if (even >= 1 && even < wf) wf = even;
                            c.waves = (int)wf;  // forward waves touch no per-slot scratch: subSlots stays as it is
                        }
                    }
                }
                c.itemCount = nSegClass;
            } else if (c.abs) {
                // one wave per region: the other form of the rows (a class that went to a team of waves keeps the team's LDS)
                if (c.threads == CPK_WAVE) setForm(c, false);
                else c.abs = false;
            }
        }
        if (getenv("CPECAN_TRACE_HOST"))
            fprintf(stderr, "cpecan class %d: %d regions, widest diagonal %d, LDS %zu B (forward launch %zu B), waves %d / %d, %s%s%s%s\n", k,
                    c.regionCount, c.geo.maxWidth, c.ldsBytes, c.ldsBytesFwd, c.waves, c.wavesTrace,
                    c.split ? (c.fused ? "one launch" : "two launches") : (c.threads > CPK_WAVE ? "a team of waves per region" : "one wave per region"),
                    c.abs ? ", absolute positions" : "",
                    c.dense ? ", three waves per SIMD" : "", c.geo.expInSweep ? ", expectation events inside the traceback" : "");
        c.ringEl = c.geo.ringCells * S;
        c.candEl = c.geo.fbCells * nCandLists;
        c.refEl = c.geo.refreshCells;
        c.totEl = c.geo.maxRefresh;
        c.bringEl = expect ? (c.geo.expInSweep ? (int64_t)c.geo.maxRefresh * 96 /* Sweep::kWinDoubles */ : c.geo.fbCells * S) : 0;
        c.grollEl = (c.geo.useGlobalRoll && c.threads == CPK_WAVE) ? c.geo.rollDoubles : 0;
        d->classes.push_back(c);
    }
    if ((int)d->classes.size() > kMaxClasses || regionAt != geo->nRegions) {
        cpk_set_error("internal: the size classes do not cover the regions (%d of %d)", regionAt, geo->nRegions);
        return CPECAN_ESTATE;
    }
    // Every resident wave owns scratch sized for its class's LARGEST region (forward ring of one traceback segment,
    // candidates, refresh series).  One unanchored 3000 x 3000 region (a single segment: 360 MB of ring) in a class of
    // its own is one wave's worth; where a class still asks for more than the device has free, it keeps as many
    // waves as fit and the rest of its regions queue behind them.
    {
        double fixed = (double)sizeof(CpkRegion) * geo->nRegions + (double)(sizeof(CpkDiag) + sizeof(int32_t)) * nDiags +
                       (double)sizeof(CpkSegment) * nSegs + (double)nSymbolBytes + 24.0 * nAnchors +
                       2.0 * 12.0 * nLists * outTriplesPerList /* the triples and their compact copy */;
        size_t freeB = 0, totalB = 0;
        HIP_TRY(hipMemGetInfo(&freeB, &totalB));
        double budget = 0.9 * ((double)freeB + (double)cache_bytes(d->device));  // idle cached blocks are ours to reuse or drop
        if (const char *mb = getenv("CPECAN_MEM_BUDGET_MB")) budget = 1048576.0 * atof(mb);  // test / diagnostic knob
        // split classes keep one ring per REGION (it holds every segment of the region), nothing per slot
        auto planSplit = [&]() {
            for (LaunchClass &c : d->classes) {
                if (!c.split) continue;
                c.ringTotal = 0;
                for (int64_t di = c.regionBase; di < c.regionBase + c.regionCount; di++) c.ringTotal += split_ring_doubles(regions[di], S);
                c.ringEl = 0;
            }
        };
        auto unsplit = [&]() {  // back to one wave per region with a per-wave ring
            for (LaunchClass &c : d->classes) {
                if (!c.split) continue;
                c.split = false;
                c.fused = false;
                c.fnTrace = nullptr;
                c.itemCount = 0;
                c.ringTotal = 0;
                c.ringEl = c.geo.ringCells * S;
                if (c.packed) {
                    c.fn = pick_packed_kernel(c.geo, c.k, false);
                    c.subSlots = (int64_t)c.waves * (CPK_WAVE / (8 << c.k));
                    continue;
                }
                if (c.abs) setForm(c, false);
                c.fn = c.dense ? pick_dense_kernel(c.geo) : pick_kernel(c.geo);
                c.subSlots = c.waves;
            }
        };
        planSplit();
        // The rings of whole regions are a fixed share of the device at most (CPECAN_SPLIT_BUDGET_FRAC, default 0.45: two
        // pipelined batches fit whatever is free at this moment), so that the same batch always runs in the same form.
        double splitBudget = 0.45 * (double)totalB;
        if (const char *fr = getenv("CPECAN_SPLIT_BUDGET_FRAC")) splitBudget = atof(fr) * (double)totalB;
        double need = 0, floorNeed = 0;
        auto tally = [&]() {
            need = floorNeed = fixed;
            for (const LaunchClass &c : d->classes) {
                need += c.slotBytes() * (double)c.subSlots + 8.0 * (double)c.ringTotal;
                floorNeed += c.slotBytes() * (double)(c.subSlots / (c.waves > 0 ? c.waves : 1)) + 8.0 * (double)c.ringTotal;
            }
        };
        tally();
        if (need > budget || need > splitBudget) {  // whole-region rings are a luxury: give them up before giving up resident waves
            unsplit();
            tally();
        }
        if (floorNeed > budget) {
            cpk_set_error("out of device memory: the batch needs %.0f MB with one resident wave per size class, %.0f MB are free",
                          floorNeed / 1048576.0, budget / 1048576.0);
            return CPECAN_ENOMEM;
        }
        if (need > budget) {
            // the narrow and less wide classes first: they hold most of the regions
            double left = budget - floorNeed;
            for (LaunchClass &c : d->classes) {
                const int64_t perWave = c.subSlots / c.waves;
                const double waveBytes = c.slotBytes() * (double)perWave;
                int64_t extra = waveBytes > 0 ? (int64_t)(left / waveBytes) : c.waves - 1;
                if (extra > c.waves - 1) extra = c.waves - 1;
                if (extra < 0) extra = 0;
                left -= waveBytes * (double)extra;
                c.waves = (int)(1 + extra);
                c.subSlots = perWave * c.waves;
            }
        }
    }
    d->totalWaves = 0;
    {
        // (the rings start 64 doubles into their block and the block ends 256 doubles behind them: the streamed traceback's
        // prefetch reads up to 63 words in front of a diagonal's row and up to 191 behind it, Sweep::tracebackAbs)
        int64_t oRing = 64, oCand = 0, oRef = 0, oTot = 0, oBring = 0, oGroll = 0, oExpect = 0;
        for (LaunchClass &c : d->classes) {
            c.oRing = oRing;
            c.oCand = oCand;
            c.oRef = oRef;
            c.oTot = oTot;
            c.oBring = oBring;
            c.oGroll = oGroll;
            c.oExpect = oExpect;
            oRing += c.split ? c.ringTotal : c.subSlots * c.ringEl;
            oRing = (oRing + 1) & ~(int64_t)1;  // the rings of a split class start on 16 bytes (Sweep::ringPut)
            oCand += c.subSlots * c.candEl;
            oRef += c.subSlots * c.refEl;
            oTot += c.subSlots * c.totEl;
            oBring += c.subSlots * c.bringEl;
            oGroll += c.subSlots * c.grollEl;
            oExpect += (int64_t)c.waves * (c.threads / CPK_WAVE) * 128;  // a partial result per wave
            d->totalWaves += (c.split && c.wavesTrace > c.waves ? c.wavesTrace : c.waves) * (c.threads / CPK_WAVE);
            if (c.ldsBytes > 64 * 1024) {
                HIP_TRY(hipFuncSetAttribute((const void *)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.ldsBytes));
                if (c.fnTrace)
                    HIP_TRY(hipFuncSetAttribute((const void *)c.fnTrace, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.ldsBytes));
            }
        }
        if (int rc = dev_alloc(d, &d->dRing, (size_t)oRing + 256)) return rc;
        if (int rc = dev_alloc(d, &d->dCand, (size_t)oCand)) return rc;
        if (int rc = dev_alloc(d, &d->dC, (size_t)oRef)) return rc;
        if (int rc = dev_alloc(d, &d->dM, (size_t)oRef)) return rc;
        if (int rc = dev_alloc(d, &d->dTotals, (size_t)oTot)) return rc;
        if (oGroll > 0)
            if (int rc = dev_alloc(d, &d->dGroll, (size_t)oGroll)) return rc;
        if (oBring > 0)
            if (int rc = dev_alloc(d, &d->dBring, (size_t)oBring)) return rc;
        if (int rc = dev_alloc(d, &d->dExpect, (size_t)(oExpect > 0 ? oExpect : 128))) return rc;
    }

    if (int rc = dev_alloc(d, &d->dRegions, (size_t)geo->nRegions)) return rc;
    if (int rc = dev_alloc(d, &d->dDiags, (size_t)nDiags)) return rc;
    {
        bool anyAbs = false;
        for (const LaunchClass &c : d->classes) anyAbs = anyAbs || c.abs;
        if (anyAbs)
            if (int rc = dev_alloc(d, &d->dDiagPos, (size_t)nDiags)) return rc;
    }
    if (int rc = dev_alloc(d, &d->dSegs, (size_t)nSegs)) return rc;
    if (int rc = dev_alloc(d, &d->dSymbols, (size_t)nSymbolBytes)) return rc;
    if (int rc = dev_alloc(d, &d->dModel, 1)) return rc;
    if (int rc = dev_alloc(d, &d->dForward, (size_t)geo->nRegions)) return rc;
    {
        const size_t nC = (size_t)nLists * geo->nRegions, nS = (size_t)nLists * (nSegs ? nSegs : 1);
        const size_t words = (nC + 63) / 64 * 64 + 2 * ((nS + 63) / 64 * 64);
        const size_t bytes = sizeof(int32_t) * words;
        d->hostCounts = host_alloc_impl(bytes, true);
        if (!d->hostCounts) {
            cpk_set_error("no pinned host memory for the result counts (%zu bytes)", bytes);
            return CPECAN_ENOMEM;
        }
        memset(d->hostCounts, 0, sizeof(int32_t) * words);  // the block is idle: nothing is in flight for this batch yet
        d->dCounts = static_cast<int32_t *>(d->hostCounts);
        d->dSegStarts = d->dCounts + (nC + 63) / 64 * 64;
        d->dSegCounts = d->dSegStarts + (nS + 63) / 64 * 64;
    }
    // split classes: the regions get rings of their own (no wrap: every segment stays readable) and their tracebacks
    // become queue items, longest first
    std::vector<CpkItem> items;
    {
        for (LaunchClass &c : d->classes) {
            if (!c.split) continue;
            int64_t ringAt = 0;  // in doubles, from the class's ring pointer (dRing + oRing)
            c.itemBase = (int64_t)items.size();
            std::vector<std::pair<int64_t, CpkItem>> byCost;
            for (int64_t di = c.regionBase; di < c.regionBase + c.regionCount; di++) {
                CpkRegion &rg = regions[di];
                rg.ringCap = 0x7fffffff;  // never wraps: the ring holds every diagonal of the region
                rg.ringBase = ringAt;
                rg.split = 1;
                ringAt += split_ring_doubles(rg, S);
                for (int32_t si = 0; si < rg.nSeg; si++) {
                    const CpkSegment &sg = segs[rg.segOff + si];
                    // fused: a segment's forward values exist when the forward wave has passed its top diagonal -- items in
                    // the order in which they become ready (that diagonal), longest first among equals
                    const int64_t cost = (int64_t)(sg.dTop - sg.tbPrev) * rg.maxWidth;
                    byCost.push_back({c.fused ? ((int64_t)0x7fffffff - sg.dTop) * ((int64_t)1 << 32) + (cost >> 8) : cost, CpkItem{(int32_t)di, si}});
                }
            }
            std::stable_sort(byCost.begin(), byCost.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
            for (const auto &e : byCost) items.push_back(e.second);
            c.itemCount = (int64_t)items.size() - c.itemBase;
        }
    }
    if (int rc = dev_alloc(d, &d->dItems, items.empty() ? 1 : items.size())) return rc;
    {
        bool anyFused = false;
        for (const LaunchClass &c : d->classes) anyFused = anyFused || c.fused;
        if (anyFused)
            if (int rc = dev_alloc(d, &d->dProgress, (size_t)geo->nRegions + 1)) return rc;
    }
    if (int rc = dev_alloc(d, &d->dTriples, (size_t)nLists * outTriplesPerList * 3)) return rc;
    if (int rc = dev_alloc(d, &d->dQueue, (size_t)2 * kMaxClasses)) return rc;
    hipStream_t io = d->io;

    if (geo->debug) {
        if (int rc = dev_alloc(d, &d->dDbgFb, (size_t)dbgCells)) return rc;
        if (int rc = dev_alloc(d, &d->dDbgTotals, (size_t)dbgDiags)) return rc;
        HIP_TRY(hipMemsetAsync(d->dDbgFb, 0xff, sizeof(double) * (size_t)dbgCells, io));      // NaN pattern
        HIP_TRY(hipMemsetAsync(d->dDbgTotals, 0xff, sizeof(double) * (size_t)dbgDiags, io));
    }

    // Everything below is ordered on the batch's own stream and the call returns WITHOUT waiting for it: the sources are
    // either blocks of the host pool that the batch owns until it is destroyed, or already copied into the shell's
    // staging buffer; cpk_device_run makes the sweep wait for evUp1.  (The host goes on planning the next batch while
    // the copy engine works: 19 ms per config-4 batch.)  The anchor block stays with the batch until it is destroyed.
    size_t stageAt = 0;
    // (anchors as runs: 16 bytes per run cross the bus and cpecan_expand_runs writes the anchors the table builders read)
    const size_t anchorBytes = runs ? sizeof(int32_t) * 4 * (size_t)(nRuns > 0 ? nRuns : 1)
                                    : sizeof(cpk_anchor_t) * (size_t)anchorStride * (size_t)(nAnchors > 0 ? nAnchors : 1);
    {
        auto staged = [](const void *p, size_t bytes) { return (bytes > kStageMaxCopy || host_is_pinned(p, bytes)) ? (size_t)0 : bytes + 256; };
        if (int rc = stage_reserve(d, staged(regions, sizeof(CpkRegion) * (size_t)geo->nRegions) + staged(runs ? (const void *)runs : (const void *)anchors, anchorBytes) +
                                      staged(segs, sizeof(CpkSegment) * (size_t)nSegs) + staged(symbols, (size_t)nSymbolBytes) +
                                      sizeof(CpkModel) + sizeof(CpkItem) * items.size() + 8 * 256))
            return rc;
    }
    if (!items.empty())
        if (int rc = staged_h2d(d, d->dItems, items.data(), sizeof(CpkItem) * items.size(), &stageAt)) return rc;
    HIP_TRY(hipEventRecord(d->evUp0, io));
    if (int rc = staged_h2d(d, d->dRegions, regions, sizeof(CpkRegion) * (size_t)geo->nRegions, &stageAt)) return rc;
    {
        // anchors -> per-diagonal table, on the device (the anchors are only needed for this)
        cpk_anchor_t *dAnchors = nullptr;
        if (int rc = dev_alloc(d, &dAnchors, (size_t)anchorStride * (size_t)(nAnchors > 0 ? nAnchors : 1))) return rc;
        if (runs && nRuns > 0) {
            int32_t *dRuns = nullptr;
            if (int rc = dev_alloc(d, &dRuns, (size_t)4 * (size_t)nRuns)) return rc;
            if (int rc = staged_h2d(d, dRuns, runs, sizeof(int32_t) * 4 * (size_t)nRuns, &stageAt)) return rc;
            const int64_t blocks = (nRuns + 255) / 256;
            hipLaunchKernelGGL(cpecan_expand_runs, dim3((unsigned)(blocks < 65535 * 16 ? blocks : 65535 * 16)), dim3(256), 0, io,
                               reinterpret_cast<const int4 *>(dRuns), nRuns, dAnchors);
            HIP_TRY(hipGetLastError());
        } else if (nAnchors > 0)
            if (int rc = staged_h2d(d, dAnchors, anchors, sizeof(cpk_anchor_t) * (size_t)anchorStride * (size_t)nAnchors, &stageAt)) return rc;
        if (int rc = staged_h2d(d, d->dSegs, segs, sizeof(CpkSegment) * (size_t)nSegs, &stageAt)) return rc;  // the builder reads the schedule
        // (the symbols and the model go first: nothing but the table build is then between the last copy and the sweep)
        if (int rc = staged_h2d(d, d->dSymbols, symbols, (size_t)nSymbolBytes, &stageAt)) return rc;
        if (int rc = staged_h2d(d, d->dModel, model, sizeof(CpkModel), &stageAt)) return rc;
        // the regions of split classes: one wave per region (cpk_table_gather.inl); every other region: one thread
        // (CPECAN_TABLE_WAVE=0: one thread for all, as rounds 1-3 -- tests compare the two)
        const char *twEnv = getenv("CPECAN_TABLE_WAVE");
        const bool tableWave = !(twEnv && atoi(twEnv) == 0);
        int64_t nSplitRegions = 0;
        for (const LaunchClass &c : d->classes)
            if (c.split && tableWave) {
                hipLaunchKernelGGL(cpecan_build_diag_table_wave, dim3((unsigned)c.regionCount), dim3(64), 0, io, d->dRegions, c.regionBase,
                                   dAnchors, anchorStride, d->dSegs, geo->nStates, d->dDiags, d->dDiagPos, expansion, dynamic);
                HIP_TRY(hipGetLastError());
                nSplitRegions += c.regionCount;
            }
        if (nSplitRegions < geo->nRegions) {
            hipLaunchKernelGGL(cpecan_build_diag_table, dim3((unsigned)((geo->nRegions + 63) / 64)), dim3(64), 0, io,
                               d->dRegions, geo->nRegions, dAnchors, anchorStride, d->dSegs, geo->nStates, d->dDiags, d->dDiagPos, expansion,
                               dynamic, tableWave ? 1 : 0);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(d->evUp1, io));
    }
    d->uploadTimed = false;
    d->h2dMs = 0.0;
    if (h2dMs) *h2dMs = 0.0;  // known once the copies are done: cpk_device_download reports it
    return CPECAN_OK;
}

// duration of the upload's copies (valid once the batch has run: the sweep waited for them)
extern "C" double cpk_device_h2d_ms(CpkDevice *d) {
    if (!d->uploadTimed && d->evUp1 && hipEventSynchronize(d->evUp1) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, d->evUp0, d->evUp1) == hipSuccess) d->h2dMs = ms;
        d->uploadTimed = true;
    }
    return d->h2dMs;
}

extern "C" int cpk_device_update_regions(CpkDevice *d, const CpkRegion *regions, const CpkSegment *segs, int64_t outTriplesPerList) {
    CPK_ON_DEVICE(d->device);
    batch_quiesce(d);  // the run that overflowed has finished with the triples and the regions
    if (outTriplesPerList != d->outTriplesPerList) {
        if (d->dTriples) {
            dev_release(d, d->dTriples);
            d->dTriples = nullptr;
        }
        d->outTriplesPerList = outTriplesPerList;
        if (int rc = dev_alloc(d, &d->dTriples, (size_t)d->nLists * outTriplesPerList * 3)) return rc;
    }
    HIP_TRY(hipMemcpyAsync(d->dRegions, regions, sizeof(CpkRegion) * (size_t)d->geo.nRegions, hipMemcpyHostToDevice, d->io));
    if (segs && d->nSegs > 0)
        HIP_TRY(hipMemcpyAsync(d->dSegs, segs, sizeof(CpkSegment) * (size_t)d->nSegs, hipMemcpyHostToDevice, d->io));
    HIP_TRY(hipStreamSynchronize(d->io));
    return CPECAN_OK;
}

// Runs the batch once more on the stream of its last run (after an output overflow); the time of the launches so far
// is kept so that the batch's kernel time covers all of them.
extern "C" int cpk_device_rerun(CpkDevice *d) {
    CPK_ON_DEVICE(d->device);
    if (d->ran) {
        HIP_TRY(hipEventSynchronize(d->evStop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, d->evStart, d->evStop));
        d->kernelMsAccum += ms;
    }
    return cpk_device_run(d, (void *)d->lastStream);
}

extern "C" int cpk_device_run(CpkDevice *d, void *stream) {
    CPK_ON_DEVICE(d->device);
    hipStream_t st = (hipStream_t)stream;
    KArgs a{};
    a.kc = d->kc;
    a.regions = d->dRegions;
    a.diags = d->dDiags;
    a.dpos = d->dDiagPos;
    a.segs = d->dSegs;
    a.symbols = d->dSymbols;
    a.model = d->dModel;
    a.geo = d->geo;
#ifdef CPK_DIAGNOSTICS  // tools/ab_build.sh <tag> -DCPK_DIAGNOSTICS: phase bisection for timing runs; never in the shipped library
    if (const char *skip = getenv("CPECAN_DEBUG_SKIP")) a.geo.debug |= (atoi(skip) & 6);
#endif
    a.ring = d->dRing;
    a.cand = d->dCand;
    a.cbuf = d->dC;
    a.mbuf = d->dM;
    a.totals = d->dTotals;
    a.groll = d->dGroll;
    a.bring = d->dBring;
    a.outCounts = d->dCounts;
    a.segStarts = d->dSegStarts;
    a.segCounts = d->dSegCounts;
    a.items = d->dItems;
    a.progress = d->dProgress;
    a.itemCount = 0;
    a.triples = d->dTriples;
    a.outTriplesPerList = d->outTriplesPerList;
    a.nSegsTotal = d->nSegs;
    a.queue = d->dQueue;
    a.forwardOut = d->dForward;
    a.expectOut = d->dExpect;
    a.dbgFb = d->dDbgFb;
    a.dbgTotals = d->dDbgTotals;
    HIP_TRY(hipStreamWaitEvent(st, d->evUp1, 0));  // the upload's copies and the table build (the batch's own stream)
    HIP_TRY(hipMemsetAsync(d->dQueue, 0, 2 * kMaxClasses * sizeof(unsigned int), st));
    if (d->dProgress) HIP_TRY(hipMemsetAsync(d->dProgress, 0, sizeof(int) * ((size_t)d->geo.nRegions + 1), st));
    if (d->geo.emit == CPECAN_EMIT_EXPECT)
        HIP_TRY(hipMemsetAsync(d->dExpect, 0, sizeof(double) * 128 * (size_t)(d->totalWaves > 0 ? d->totalWaves : 1), st));
    HIP_TRY(hipEventRecord(d->evStart, st));
    // one launch per size class, side by side: the last class (the widest regions, usually the bulk of the work) on the
    // caller's stream, the others on streams of their own, joined below
    const int nClasses = (int)d->classes.size();
    for (int i = 0; i < nClasses; i++) {
        const LaunchClass &c = d->classes[i];
        KArgs p = a;
        p.geo = c.geo;
        p.geo.debug = a.geo.debug;
        p.regionBase = c.regionBase;
        p.regionCount = c.regionCount;
        p.ring = d->dRing + c.oRing;
        p.cand = d->dCand + c.oCand;
        p.cbuf = d->dC + c.oRef;
        p.mbuf = d->dM + c.oRef;
        p.totals = d->dTotals + c.oTot;
        p.groll = d->dGroll ? d->dGroll + c.oGroll : nullptr;
        p.bring = d->dBring ? d->dBring + c.oBring : nullptr;
        p.expectOut = d->dExpect + c.oExpect;
        p.queue = d->dQueue + i;
        if (c.fused) {
            p.items = d->dItems + c.itemBase;
            p.itemCount = (int32_t)c.itemCount;
        }
        const bool onCaller = i == nClasses - 1;
        if (!onCaller && !d->sideStream[i]) {
            // A priority of its own: the runtime maps the streams of one priority onto a handful of hardware queues, and a
            // pipeline of batches has more streams alive than that -- a side stream that shares its hardware queue with
            // the caller's stream runs its class BEHIND the caller's instead of beside it (BASELINE config 4, two narrow
            // classes: 45 ms per batch instead of 28 whenever that happened, profiles/r04_config4_pipeline.txt).  The
            // queues of the high priority hold nothing but these side streams.
            {
                int prLow = 0, prHigh = 0;
                HIP_TRY(hipDeviceGetStreamPriorityRange(&prLow, &prHigh));
                const char *env = getenv("CPECAN_SIDE_PRIORITY");  // 0: the default priority, as rounds 1-3 (A/B runs)
                if (env && atoi(env) == 0) prHigh = 0;
                HIP_TRY(hipStreamCreateWithPriority(&d->sideStream[i], hipStreamNonBlocking, prHigh));
            }
            HIP_TRY(hipEventCreateWithFlags(&d->sideDone[i], hipEventDisableTiming));
        }
        hipStream_t cs = onCaller ? st : d->sideStream[i];
        if (!onCaller) HIP_TRY(hipStreamWaitEvent(cs, d->evStart, 0));
        hipLaunchKernelGGL(c.fn, dim3((unsigned)c.waves), dim3((unsigned)c.threads), (c.split && !c.fused && c.ldsBytesFwd) ? c.ldsBytesFwd : c.ldsBytes, cs, p);
        HIP_TRY(hipGetLastError());
        if (c.split && !c.fused) {  // the tracebacks of the class's regions, one queue item each, behind the forward launch
            KArgs t = p;
            t.items = d->dItems + c.itemBase;
            t.regionCount = (int32_t)c.itemCount;
            t.queue = d->dQueue + kMaxClasses + i;
            hipLaunchKernelGGL(c.fnTrace, dim3((unsigned)c.wavesTrace), dim3((unsigned)c.threads), c.ldsBytes, cs, t);
            HIP_TRY(hipGetLastError());
        }
        if (!onCaller) HIP_TRY(hipEventRecord(d->sideDone[i], cs));
    }
    for (int i = 0; i + 1 < nClasses; i++)  // join: the caller's stream continues when every class is done
        HIP_TRY(hipStreamWaitEvent(st, d->sideDone[i], 0));
    HIP_TRY(hipEventRecord(d->evStop, st));
    d->lastStream = st;
    ran_set(d, true);
    return CPECAN_OK;
}

extern "C" int cpk_device_download(CpkDevice *d, int32_t *counts, int32_t *segStarts, int32_t *segCounts, double *expect,
                                   double *kernelMs, double *d2hMs) {
    CPK_ON_DEVICE(d->device);
    if (!d->ran) {
        cpk_set_error("download before run");
        return CPECAN_ESTATE;
    }
    // the batch's own stop event, not the caller's stream: that stream may already hold the next batch's launches
    const bool trace = getenv("CPECAN_TRACE_HOST") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipEventSynchronize(d->evStop));
    if (trace) fprintf(stderr, "cpecan download: waited %.1f ms for the sweep\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->evStart, d->evStop));
    if (kernelMs) *kernelMs = d->kernelMsAccum + ms;  // every launch of the batch (an overflow re-run included)
    hipStream_t io = d->io;
    HIP_TRY(hipEventRecord(d->evA, io));
    if (d->dProgress) {  // a fused launch reports a traceback whose forward values never came (it waits a bounded time)
        int err = 0;
        HIP_TRY(hipMemcpyAsync(&err, d->dProgress + d->geo.nRegions, sizeof(int), hipMemcpyDeviceToHost, io));
        HIP_TRY(hipStreamSynchronize(io));
        if (err) {
            // An item gave up waiting for its region's forward values (a bounded poll).  The rings of the class are whole:
            // run it again as two launches -- all forward sweeps, then all items -- which needs no hand-off inside a launch.
            bool any = false;
            for (LaunchClass &c : d->classes) {
                if (!c.fused) continue;
                any = true;
                c.fused = false;
                pick_split_kernels(c.geo, c.dense, c.abs, &c.fn, &c.fnTrace);
                c.wavesTrace = c.waves;
                if (c.waves > c.regionCount) c.waves = c.regionCount;
            }
            if (!any || d->fusedRetried) {
                cpk_set_error("fused launch: a traceback item waited in vain for its region's forward sweep");
                return CPECAN_EHIP;
            }
            d->fusedRetried = true;
            if (int rc = cpk_device_rerun(d)) return rc;
            return cpk_device_download(d, counts, segStarts, segCounts, expect, kernelMs, d2hMs);
        }
    }
    // written by the sweeps straight into pinned host memory; complete with the stop event
    memcpy(counts, d->dCounts, sizeof(int32_t) * (size_t)d->nLists * d->geo.nRegions);
    memcpy(segStarts, d->dSegStarts, sizeof(int32_t) * (size_t)d->nLists * d->nSegs);
    memcpy(segCounts, d->dSegCounts, sizeof(int32_t) * (size_t)d->nLists * d->nSegs);
    HIP_TRY(hipEventRecord(d->evB, io));
    HIP_TRY(hipStreamSynchronize(io));
    HIP_TRY(hipEventElapsedTime(&ms, d->evA, d->evB));
    if (d2hMs) *d2hMs = ms;
    if (expect && d->geo.emit == kEmitForward) {
        HIP_TRY(hipMemcpyAsync(expect, d->dForward, sizeof(double) * (size_t)d->geo.nRegions, hipMemcpyDeviceToHost, io));
        HIP_TRY(hipStreamSynchronize(io));
    }
    if (expect && d->geo.emit == CPECAN_EMIT_EXPECT) {
        // sum the per-wave partials (every launched wave wrote its 106 values, zeros included)
        const int nWaves = d->totalWaves;
        std::vector<double> part((size_t)nWaves * 128);
        HIP_TRY(hipMemcpyAsync(part.data(), d->dExpect, sizeof(double) * part.size(), hipMemcpyDeviceToHost, io));
        HIP_TRY(hipStreamSynchronize(io));
        for (int i = 0; i < 106; i++) expect[i] = 0.0;
        for (int w = 0; w < nWaves; w++)
            for (int i = 0; i < 106; i++) expect[i] += part[(size_t)w * 128 + i];
    }
    return CPECAN_OK;
}

extern "C" int cpk_device_gather(CpkDevice *d, const CpkChunk *chunks, int64_t nChunks, int64_t total) {
    CPK_ON_DEVICE(d->device);
    if (nChunks <= 0 || total <= 0) return CPECAN_OK;
    if (nChunks > d->chunkCap || total > d->compactCap) batch_quiesce(d);  // before blocks are recycled
    if (nChunks > d->chunkCap) {
        cache_free(d->device, d->dChunks, d->chunkBytes);
        d->dChunks = nullptr;
        d->chunkBytes = sizeof(CpkChunk) * (size_t)nChunks;
        HIP_TRY(cache_alloc(d->device, (void **)&d->dChunks, d->chunkBytes));
        d->chunkCap = nChunks;
    }
    if (total > d->compactCap) {
        cache_free(d->device, d->dCompact, d->compactBytes);
        d->dCompact = nullptr;
        d->compactBytes = sizeof(int32_t) * 3 * (size_t)total;
        HIP_TRY(cache_alloc(d->device, (void **)&d->dCompact, d->compactBytes));
        d->compactCap = total;
    }
    // the caller (cpecan_batch_download) has waited for the sweep; the chunk array is the caller's until the copy is done
    HIP_TRY(hipMemcpyAsync(d->dChunks, chunks, sizeof(CpkChunk) * (size_t)nChunks, hipMemcpyHostToDevice, d->io));
    const int64_t blocks = nChunks < 16384 ? nChunks : 16384;
    hipLaunchKernelGGL(cpecan_gather_lists, dim3((unsigned)blocks), dim3(256), 0, d->io, d->dChunks, nChunks, d->dTriples,
                       d->dCompact);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->io));
    return CPECAN_OK;
}

extern "C" int cpk_device_fetch(CpkDevice *d, int32_t *hostOut, int64_t total, double *d2hMs) {
    CPK_ON_DEVICE(d->device);
    if (total <= 0) return CPECAN_OK;
    HIP_TRY(hipEventRecord(d->evA, d->io));
    HIP_TRY(hipMemcpyAsync(hostOut, d->dCompact, sizeof(int32_t) * 3 * (size_t)total, hipMemcpyDeviceToHost, d->io));
    HIP_TRY(hipEventRecord(d->evB, d->io));
    HIP_TRY(hipStreamSynchronize(d->io));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d->evA, d->evB));
    if (d2hMs) *d2hMs += ms;
    return CPECAN_OK;
}

// The consumers on a device-resident triple buffer.  Scratch lives for the duration of the call.
namespace {
struct PostScratch {  // device blocks of one consumer stage, on the current device, used on ONE stream
    std::vector<CachedBlock> blocks;
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    explicit PostScratch(hipStream_t st) : stream(st) {
        (void)hipGetDevice(&device);
        if (!stream) {  // lists given by the host (no batch): a stream of the call's own, never the null stream
            ownStream = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
            if (!ownStream) stream = nullptr;
        }
    }
    ~PostScratch() {
        if (!blocks.empty()) (void)hipStreamSynchronize(stream);  // the stage's kernels are done with them
        for (const CachedBlock &b : blocks) cache_free(device, b.ptr, b.bytes);
        if (ownStream) (void)hipStreamDestroy(stream);
    }
    template <typename T>
    int alloc(T **out, size_t count) {
        void *p = nullptr;
        const size_t bytes = (count ? count : 1) * sizeof(T);
        if (cache_alloc(device, &p, bytes) != hipSuccess) {
            cpk_set_error("out of device memory in the list consumers");
            return CPECAN_ENOMEM;
        }
        blocks.push_back({p, bytes});
        *out = static_cast<T *>(p);
        return CPECAN_OK;
    }
};
}  // namespace

static int post_core(PostScratch &sc, int32_t *dTriples, const CpkPostJob *job) {
    const int64_t nP = job->nProblems;
    if (nP <= 0) return CPECAN_OK;
    hipStream_t st = sc.stream;
    CpkPostProblem *dProblems = nullptr;
    double *dScores = nullptr;
    int32_t *dCounts = nullptr;
    if (int rc = sc.alloc(&dProblems, (size_t)nP)) return rc;
    if (int rc = sc.alloc(&dScores, (size_t)nP * kPostScores)) return rc;
    if (int rc = sc.alloc(&dCounts, (size_t)nP * 2)) return rc;
    HIP_TRY(hipMemcpyAsync(dProblems, job->problems, sizeof(CpkPostProblem) * (size_t)nP, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(dScores, 0, sizeof(double) * (size_t)nP * kPostScores, st));
    HIP_TRY(hipMemsetAsync(dCounts, 0, sizeof(int32_t) * (size_t)nP * 2, st));
    {
        int32_t *dMass = nullptr;
        const bool rw = (job->flags & kPostReweight) != 0;
        if (int rc = sc.alloc(&dMass, rw ? (size_t)job->seqSlots : 1)) return rc;
        hipLaunchKernelGGL(cpecan_post_reweight, dim3((unsigned)nP), dim3(256), 0, st, dProblems, dTriples, dMass,
                           job->gapGamma, rw ? 1 : 0, dScores);
        HIP_TRY(hipGetLastError());
    }
    int32_t *dMea = nullptr, *dShift = nullptr;
    uint8_t *dChars = nullptr;
    const unsigned laneBlocks = (unsigned)((nP + 63) / 64);
    if (job->flags & (kPostMea | kPostLeftShift | kPostOrdered))
        if (int rc = sc.alloc(&dMea, (size_t)job->meaCap * 3)) return rc;
    if (job->chars) {
        if (int rc = sc.alloc(&dChars, (size_t)(job->nChars > 0 ? job->nChars : 1))) return rc;
        if (job->nChars > 0) HIP_TRY(hipMemcpyAsync(dChars, job->chars, (size_t)job->nChars, hipMemcpyHostToDevice, st));
    }
    if (job->flags & kPostOrdered) {
        int32_t *dSeq = nullptr, *dPrev = nullptr, *dNext = nullptr;
        double *dBest = nullptr;
        uint8_t *dChosen = nullptr;
        if (int rc = sc.alloc(&dSeq, (size_t)(job->seqSlots > 0 ? job->seqSlots : 1))) return rc;
        if (int rc = sc.alloc(&dBest, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dPrev, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dNext, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dChosen, (size_t)job->chainSlots)) return rc;
        const char *lanesEnv = getenv("CPECAN_POST_LANES");  // 1: one lane per problem (the form of rounds 1-2; tests, A/B runs)
        if (lanesEnv && atoi(lanesEnv) != 0) {
            hipLaunchKernelGGL(cpecan_post_ordered, dim3(laneBlocks), dim3(64), 0, st, dProblems, nP, dTriples, dSeq,
                               dBest, dPrev, dNext, dChosen, job->matchGamma, dMea, dCounts);
        } else {
            // one wave per problem: the pairs in column order take four more words a pair
            int32_t *dSortX = nullptr, *dSortY = nullptr;
            double *dSortW = nullptr;
            if (int rc = sc.alloc(&dSortX, (size_t)job->chainSlots)) return rc;
            if (int rc = sc.alloc(&dSortY, (size_t)job->chainSlots)) return rc;
            if (int rc = sc.alloc(&dSortW, (size_t)job->chainSlots)) return rc;
            hipLaunchKernelGGL(cpecan_post_ordered_wave, dim3((unsigned)nP), dim3(64), 0, st, dProblems, dTriples, dSeq, dBest,
                               dPrev, dNext, dSortX, dSortY, dSortW, dChosen, job->matchGamma, dMea, dCounts);
        }
        HIP_TRY(hipGetLastError());
    }
    if ((job->flags & kPostOrdered) || dChars) {
        const int fromOut = (job->flags & kPostOrdered) ? 1 : 0;
        hipLaunchKernelGGL(cpecan_post_list_scores, dim3((unsigned)nP), dim3(256), 0, st, dProblems, dTriples, dMea,
                           dCounts, fromOut, fromOut, dChars, dScores);
        HIP_TRY(hipGetLastError());
    }
    if (job->flags & kPostMea) {
        long long *dCum = nullptr;
        double *dBest = nullptr;
        int32_t *dPrev = nullptr;
        uint8_t *dRecord = nullptr;
        if (int rc = sc.alloc(&dCum, (size_t)job->seqSlots)) return rc;
        if (int rc = sc.alloc(&dBest, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dPrev, (size_t)job->chainSlots)) return rc;
        if (int rc = sc.alloc(&dRecord, (size_t)job->chainSlots)) return rc;
        const char *lanesEnv = getenv("CPECAN_POST_LANES");  // 1: one lane per problem (the form of rounds 1-2; tests, A/B runs)
        if (lanesEnv && atoi(lanesEnv) != 0) {
            hipLaunchKernelGGL(cpecan_post_mea, dim3(laneBlocks), dim3(64), 0, st, dProblems, nP, dTriples, dCum, dBest,
                               dPrev, dRecord, (float)job->gapGamma, dMea, dCounts, dScores);
        } else {
            long long *dG = nullptr, *dH = nullptr, *dT = nullptr;  // the gap masses around every pair (cpecan_post_mea_wave)
            if (int rc = sc.alloc(&dG, (size_t)job->chainSlots)) return rc;
            if (int rc = sc.alloc(&dH, (size_t)job->chainSlots)) return rc;
            if (int rc = sc.alloc(&dT, (size_t)job->chainSlots)) return rc;
            hipLaunchKernelGGL(cpecan_post_mea_wave, dim3((unsigned)nP), dim3(64), 0, st, dProblems, dTriples, dCum, dBest, dPrev,
                               dRecord, dG, dH, dT, (float)job->gapGamma, dMea, dCounts, dScores);
        }
        HIP_TRY(hipGetLastError());
    } else if (job->flags & kPostLeftShift) {
        hipLaunchKernelGGL(cpecan_post_copy_chain, dim3((unsigned)nP), dim3(256), 0, st, dProblems, dTriples, dMea,
                           dCounts);
        HIP_TRY(hipGetLastError());
    }
    if (job->flags & kPostLeftShift) {
        if (!dChars) {
            cpk_set_error("left shift needs the raw sequences");
            return CPECAN_EINVAL;
        }
        if (int rc = sc.alloc(&dShift, (size_t)job->shiftCap * 3)) return rc;
        hipLaunchKernelGGL(cpecan_post_left_shift, dim3(laneBlocks), dim3(64), 0, st, dProblems, nP, dMea, dChars,
                           dShift, dCounts);
        HIP_TRY(hipGetLastError());
    }
    if (job->scores)
        HIP_TRY(hipMemcpyAsync(job->scores, dScores, sizeof(double) * (size_t)nP * kPostScores, hipMemcpyDeviceToHost, st));
    if (job->counts) HIP_TRY(hipMemcpyAsync(job->counts, dCounts, sizeof(int32_t) * (size_t)nP * 2, hipMemcpyDeviceToHost, st));
    if (job->mea && dMea)
        HIP_TRY(hipMemcpyAsync(job->mea, dMea, sizeof(int32_t) * 3 * (size_t)job->meaCap, hipMemcpyDeviceToHost, st));
    if (job->shift && dShift)
        HIP_TRY(hipMemcpyAsync(job->shift, dShift, sizeof(int32_t) * 3 * (size_t)job->shiftCap, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return CPECAN_OK;
}

extern "C" int cpk_device_post(CpkDevice *d, const CpkPostJob *job) {
    CPK_ON_DEVICE(d->device);
    if (!d->dCompact && job->nProblems > 0) {
        // every list is empty: the consumers still need a valid base pointer
        d->compactBytes = sizeof(int32_t) * 3;
        HIP_TRY(cache_alloc(d->device, (void **)&d->dCompact, d->compactBytes));
        d->compactCap = 1;
    }
    PostScratch sc(d->io);
    return post_core(sc, d->dCompact, job);
}

extern "C" int cpk_post_lists(int device, int32_t *triples, int64_t total, const CpkPostJob *job) {
    const int nDev = cpk_device_count();
    if (nDev <= 0 || device < 0 || device >= nDev) {
        cpk_set_error("no usable HIP device (count=%d, requested=%d): the HIP path has no CPU fallback", nDev, device);
        return CPECAN_ENODEVICE;
    }
    CPK_ON_DEVICE(device);
    PostScratch sc(nullptr);
    if (!sc.stream) {
        cpk_set_error("hipStreamCreate failed in the list consumers");
        return CPECAN_EHIP;
    }
    int32_t *dTriples = nullptr;
    if (int rc = sc.alloc(&dTriples, (size_t)(total > 0 ? total : 1) * 3)) return rc;
    if (total > 0)
        HIP_TRY(hipMemcpyAsync(dTriples, triples, sizeof(int32_t) * 3 * (size_t)total, hipMemcpyHostToDevice, sc.stream));
    if (int rc = post_core(sc, dTriples, job)) return rc;
    if (total > 0) {
        HIP_TRY(hipMemcpyAsync(triples, dTriples, sizeof(int32_t) * 3 * (size_t)total, hipMemcpyDeviceToHost, sc.stream));
        HIP_TRY(hipStreamSynchronize(sc.stream));
    }
    return CPECAN_OK;
}

extern "C" int cpk_device_debug_fetch(CpkDevice *d, double *fb, int64_t cells, double *totals, int64_t diags) {
    CPK_ON_DEVICE(d->device);
    batch_quiesce(d);
    if (!d->geo.debug || !d->dDbgFb) {
        cpk_set_error("debug buffers were not enabled before upload");
        return CPECAN_ESTATE;
    }
    if (cells > d->dbgCells || diags > d->dbgDiags) {
        cpk_set_error("debug fetch larger than the debug buffers");
        return CPECAN_EINVAL;
    }
    HIP_TRY(hipMemcpy(fb, d->dDbgFb, sizeof(double) * (size_t)cells, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(totals, d->dDbgTotals, sizeof(double) * (size_t)diags, hipMemcpyDeviceToHost));
    return CPECAN_OK;
}

// The reference's unit-test primitives (cpk_cells.inl): a few hundred cells at most, one lane, blocking copies.
extern "C" int cpk_ref_cells(int device, const CpkModel *model, int mode, const CpkCellOp *ops, int64_t n, double *buf,
                             int64_t nDoubles, double total) {
    const int nDev = cpk_device_count();
    if (nDev <= 0 || device < 0 || device >= nDev) {
        cpk_set_error("no usable HIP device (count=%d, requested=%d): the HIP path has no CPU fallback", nDev, device);
        return CPECAN_ENODEVICE;
    }
    if (n <= 0 || nDoubles <= 0) return CPECAN_OK;
    CPK_ON_DEVICE(device);
    void *dOps = nullptr, *dBuf = nullptr;
    const size_t opBytes = sizeof(CpkCellOp) * (size_t)n, bufBytes = sizeof(double) * (size_t)nDoubles;
    HIP_TRY(cache_alloc(device, &dOps, opBytes));
    if (hipError_t e = cache_alloc(device, &dBuf, bufBytes); e != hipSuccess) {
        cache_free(device, dOps, opBytes);
        HIP_TRY(e);
    }
    int rc = CPECAN_OK;
    hipError_t e = hipMemcpy(dOps, ops, opBytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dBuf, buf, bufBytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(cpecan_ref_cells, dim3(1), dim3(CPK_WAVE), 0, nullptr, *model, mode, (const CpkCellOp *)dOps, (int)n,
                           (double *)dBuf, total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(buf, dBuf, bufBytes, hipMemcpyDeviceToHost);  // waits for the kernel (null stream)
    if (e != hipSuccess) {
        cpk_set_error("cpk_ref_cells: %s", hipGetErrorString(e));
        rc = CPECAN_EHIP;
    }
    cache_free(device, dOps, opBytes);
    cache_free(device, dBuf, bufBytes);
    return rc;
}

extern "C" int64_t cpk_device_bytes(const CpkDevice *d) { return d->bytes; }
extern "C" int cpk_device_form(const CpkDevice *d) {
    if (d->classes.empty()) return CPECAN_FORM_WHOLE;
    const LaunchClass &c = d->classes.back();
    return (c.split ? (c.fused ? CPECAN_FORM_FUSED : CPECAN_FORM_SPLIT) : CPECAN_FORM_WHOLE) | (c.abs ? CPECAN_FORM_ABS : 0);
}
extern "C" int cpk_device_waves(const CpkDevice *d) { return d->totalWaves; }

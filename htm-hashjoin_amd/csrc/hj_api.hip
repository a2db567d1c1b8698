// hj_api.hip -- implementation of the C ABI in include/htm_hashjoin.h.
// Host-side glue only: argument checks, device memory, stream order, HIP-event
// timing. All arithmetic on tuples happens in hj_kernels.hip / hj_prj.hip.
// There is no CPU fallback in here: every operator needs a gfx950 device.

#include "../../include/htm_hashjoin.h"
#include "hj_device.h"
#include "hj_rand.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace hj;

namespace {
enum Ev { EV_CLEAR0, EV_BUILD0, EV_BUILD1, EV_KW0, EV_KW1, EV_KC0, EV_KC1, EV_KO0, EV_KO1, EV_PROBE0, EV_PROBE1, EV_PRJ0, EV_PRJ_PART, EV_PRJ1, EV_PRJ_S0, EV_PRJ_S1, EV_COUNT };
}

struct hj_ctx {
    int device = 0;
    int nCU = 256;                // of THIS context's device (grids are sized per context, never from process statics)
    hipStream_t stream = nullptr;
    bool ownStream = false;
    hj_params params{};
    // open-addressing table
    uint64_t* table = nullptr;
    uint64_t tableCapSlots = 0;   // allocated slots incl. slack
    uint64_t tableSize = 0;       // live table (2*rSize) of the last build
    uint32_t hshift = 0;             // home-slot shift of the current table (hj_device.h); 0 unless it is a radix shard
    ShardCheck sc{0, 0, 0, 0};       // hj_set_shard_check; mask 0 = off
    uint64_t rSize = 0, sSize = 0;
    bool built = false;
    bool probeStartsAtBuildEnd = false;         // the probe was enqueued right behind the build on the library's own stream: EV_BUILD1 is its start
    bool streamAtBuildEnd = false;              // nothing has been enqueued since EV_BUILD1 (own stream only)
    // ownership build (variant 2)
    void* ownerBuf = nullptr; size_t capOwner = 0;
    void* queueBuf = nullptr; size_t capQueue = 0;
    uint32_t* queueCount = nullptr;             // device: kOwnMaxChunks words, deferred tuples per phase-A workgroup of the window build
    unsigned int* fitCount = nullptr;           // device, kSampleWords words (launch_sample_locality)
    unsigned int* hFit = nullptr;               // pinned
    unsigned long long* hPreferred = nullptr;   // pinned: Counters::preferred of the last device-side pick (0: none yet)
    unsigned long long* dPreferred = nullptr;   // the same word as the device addresses it (the sampler stores into it)
    void* boundsBuf = nullptr;                  // variant 3: per-chunk slot ranges (wave_bounds_bytes)
    // bucketised table of --algo htm (hj_htm.hip): the table itself lives in `table` (4 slots per bucket)
    bool htmBuilt = false;
    bool htmGenericChains = false;              // build_htm's second attempt: no routing, generic chain kernels
    bool htmChainsFellBack = false;             // ... and that it happened (hj_result.compactFallback bit 8)
    uint32_t htmBuckets = 0;                    // numBuckets of the last htm build
    uint64_t* htmConflicts = nullptr; size_t capHtmConflicts = 0;       // bytes
    uint32_t* htmOwnCounts = nullptr; size_t capHtmOwnCounts = 0;        // bytes: conflicts listed per chunk of the window build
    unsigned int* htmOvfCount = nullptr; uint32_t* htmOvfBase = nullptr; uint32_t* htmScan = nullptr; uint64_t capHtmBuckets = 0;
    uint64_t* htmOverflow = nullptr; uint64_t capHtmOverflow = 0;       // overflow buckets (index 0 unused)
    uint64_t htmOverflowUsed = 0;
    // streaming Zipf generator (hj_zipf_open / hj_zipf_next_dev)
    hjhost::GlibcRand* zipfRng = nullptr;
    double* zipfLut = nullptr; uint32_t* zipfAlphabet = nullptr; uint32_t zipfAlphabetSize = 0;   // device
    int* zipfRawHost[2] = {nullptr, nullptr}; int* zipfRawDev[2] = {nullptr, nullptr}; uint64_t zipfRawCap = 0;
    hipEvent_t zipfDone[2] = {nullptr, nullptr}; int zipfFlip = 0;
    uint32_t variantUsed = 1;
    // counters
    Counters* dCtr = nullptr;
    Counters* hCtr = nullptr;     // pinned
    // PRJ workspace
    PrjPlan plan{};
    uint64_t *tmpA = nullptr, *partR = nullptr, *partS = nullptr;
    void* work = nullptr;
    uint64_t capTmp = 0, capPartR = 0, capPartS = 0;
    uint32_t forceVariant = 0;                  // hj_join_dev(AUTO) has already sampled: build with this variant
    uint32_t algoUsed = 0;
    size_t capWork = 0;
    bool prjRan = false;
    bool prjOptimistic = false;   // the last radix join enqueued the histogram-free passes
    // staging for hj_run
    uint64_t *stageR = nullptr, *stageS = nullptr;
    uint64_t capStageR = 0, capStageS = 0;
    // shard helper: up to 4 inputs may sit between their histogram and their scatter
    struct ShardPlan { const uint64_t* in = nullptr; uint64_t n = 0; uint32_t nShards = 0, mode = 0; void* work = nullptr; size_t cap = 0; uint64_t stamp = 0; };
    ShardPlan shard[4];
    uint64_t shardStamp = 0;
    // timing
    hipEvent_t ev[EV_COUNT]{};
    bool evSet[EV_COUNT]{};
    double h2d_us = 0;
    std::string err;
};

namespace {

int fail(hj_ctx* c, int code, const char* what, hipError_t e = hipSuccess)
{
    if (c) {
        char buf[512];
        if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else snprintf(buf, sizeof buf, "%s", what);
        c->err = buf;
    }
    return code;
}

#define HJ_HIP(c, call)                                                     \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess)                                               \
            return fail((c), e_ == hipErrorOutOfMemory ? HJ_ERR_OOM : HJ_ERR_HIP, #call, e_); \
    } while (0)

bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }

uint32_t probe_len(const hj_params& p) { return p.probeLength ? p.probeLength : 4; }

template <typename T>
int grow(hj_ctx* c, T*& ptr, uint64_t& cap, uint64_t need)
{
    if (need <= cap) return HJ_OK;
    if (ptr) { HJ_HIP(c, hipFree(ptr)); ptr = nullptr; cap = 0; }
    void* p = nullptr;
    HJ_HIP(c, hipMalloc(&p, need * sizeof(T)));
    ptr = static_cast<T*>(p);
    cap = need;
    return HJ_OK;
}

uint32_t auto_radix_bits(uint64_t nR)
{
    // >= NUM_RADIX_BITS (prj_params.h:16) and enough that an average R partition
    // fills at most half of the LDS table; two passes of <= 8 bits
    uint32_t bits = 14;
    while (bits < 16 && (nR >> bits) > 16384) ++bits;
    return bits;
}

int record(hj_ctx* c, Ev e)
{
    HJ_HIP(c, hipEventRecord(c->ev[e], c->stream));
    c->evSet[e] = true;
    return HJ_OK;
}

double elapsed_us(hj_ctx* c, Ev a, Ev b)
{
    if (!c->evSet[a] || !c->evSet[b]) return 0.0;
    float ms = 0;
    if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) return 0.0;
    return (double)ms * 1000.0;
}

int create_common(int device, void* stream, bool own, hj_ctx** out)
{
    if (!out) return HJ_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return HJ_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return HJ_ERR_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HJ_ERR_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return HJ_ERR_NO_DEVICE;  // kernels are gfx950-only
    if (hipSetDevice(device) != hipSuccess) return HJ_ERR_HIP;
    // the kernels with more than 64 KiB of dynamic LDS need the attribute on EVERY device they run on
    if (own_set_attributes() != hipSuccess || prj_set_attributes() != hipSuccess) return HJ_ERR_HIP;
    hj_ctx* c = new (std::nothrow) hj_ctx();
    if (!c) return HJ_ERR_OOM;
    c->device = device;
    c->nCU = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (own) {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return HJ_ERR_HIP; }
        c->ownStream = true;
    } else {
        c->stream = static_cast<hipStream_t>(stream);
    }
    bool ok = hipMalloc(reinterpret_cast<void**>(&c->dCtr), sizeof(Counters)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->hCtr), sizeof(Counters)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->queueCount), kOwnMaxChunks * sizeof(uint32_t)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void**>(&c->fitCount), kSampleWords * sizeof(unsigned int)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->hFit), 8 * sizeof(unsigned int)) == hipSuccess &&
              hipHostMalloc(reinterpret_cast<void**>(&c->hPreferred), sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess &&
              hipHostGetDevicePointer(reinterpret_cast<void**>(&c->dPreferred), c->hPreferred, 0) == hipSuccess &&
              hipMalloc(&c->boundsBuf, wave_bounds_bytes(c->nCU)) == hipSuccess;
    for (int i = 0; ok && i < EV_COUNT; ++i) ok = hipEventCreate(&c->ev[i]) == hipSuccess;
    if (!ok) { hj_destroy(c); return HJ_ERR_HIP; }
    hipMemset(c->dCtr, 0, sizeof(Counters));
    memset(c->hCtr, 0, sizeof(Counters));
    *c->hPreferred = 0;
    *out = c;
    return HJ_OK;
}

}  // namespace

extern "C" {

int hj_abi_version(void) { return HJ_ABI_VERSION; }

int hj_device_count(int* count)
{
    if (!count) return HJ_ERR_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return HJ_ERR_NO_DEVICE; }
    *count = n;
    return HJ_OK;
}

int hj_create(int device, hj_ctx** out) { return create_common(device, nullptr, true, out); }

int hj_create_on_stream(int device, void* hip_stream, hj_ctx** out) { return create_common(device, hip_stream, false, out); }

static void zipf_release(hj_ctx* c)
{
    delete c->zipfRng; c->zipfRng = nullptr;
    if (c->zipfLut) hipFree(c->zipfLut);
    if (c->zipfAlphabet) hipFree(c->zipfAlphabet);
    c->zipfLut = nullptr; c->zipfAlphabet = nullptr; c->zipfAlphabetSize = 0;
    for (int i = 0; i < 2; ++i) {
        if (c->zipfRawHost[i]) hipHostFree(c->zipfRawHost[i]);
        if (c->zipfRawDev[i]) hipFree(c->zipfRawDev[i]);
        if (c->zipfDone[i]) hipEventDestroy(c->zipfDone[i]);
        c->zipfRawHost[i] = nullptr; c->zipfRawDev[i] = nullptr; c->zipfDone[i] = nullptr;
    }
    c->zipfRawCap = 0;
}

void hj_destroy(hj_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream || !c->ownStream) hipStreamSynchronize(c->stream);
    zipf_release(c);
    if (c->stream || !c->ownStream) hipStreamSynchronize(c->stream);
    void* frees[] = {c->table, c->dCtr, c->tmpA, c->partR, c->partS, c->work, c->stageR, c->stageS,
                     c->ownerBuf, c->queueBuf, c->queueCount, c->fitCount, c->boundsBuf, c->htmConflicts, c->htmOwnCounts, c->htmOvfCount,
                     c->htmOvfBase, c->htmScan, c->htmOverflow, c->shard[0].work, c->shard[1].work,
                     c->shard[2].work, c->shard[3].work};
    for (void* p : frees) if (p) hipFree(p);
    if (c->hCtr) hipHostFree(c->hCtr);
    if (c->hFit) hipHostFree(c->hFit);
    if (c->hPreferred) hipHostFree(c->hPreferred);
    for (int i = 0; i < EV_COUNT; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]);
    if (c->ownStream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char* hj_strerror(int status)
{
    switch (status) {
        case HJ_OK: return "ok";
        case HJ_ERR_INVALID: return "invalid argument";
        case HJ_ERR_NO_DEVICE: return "no gfx950 HIP device (this library has no CPU fallback)";
        case HJ_ERR_HIP: return "HIP runtime error";
        case HJ_ERR_OOM: return "out of device memory";
        case HJ_ERR_KEY_RANGE: return "tuple outside the DataGen layout (payload bits set or value 0)";
        case HJ_ERR_UNKNOWN_ALGO: return "unknown algo";
        case HJ_ERR_STATE: return "call order violated";
        default: return "unknown status";
    }
}

const char* hj_last_error(const hj_ctx* c) { return c ? c->err.c_str() : "null context"; }

int hj_synchronize(hj_ctx* c)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c) return HJ_ERR_INVALID;
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    return HJ_OK;
}

int hj_reserve(hj_ctx* c, const hj_params* params, uint64_t rSize, uint64_t sSize)
{
    if (!c || !params) return HJ_ERR_INVALID;
    if (params->algo > HJ_ALGO_AUTO) return fail(c, HJ_ERR_UNKNOWN_ALGO, "hj_reserve: algo");
    if (rSize == 0) return fail(c, HJ_ERR_INVALID, "hj_reserve: rSize == 0");
    HJ_HIP(c, hipSetDevice(c->device));
    c->params = *params;
    HJ_HIP(c, hipStreamSynchronize(c->stream));          // buffers may be replaced below; and a new workload starts without an expectation
    *c->hPreferred = 0;
    if (params->algo == HJ_ALGO_PRJ || params->algo == HJ_ALGO_AUTO) {
        if (rSize >= 0xFFFFFFFFull || sSize >= 0xFFFFFFFFull)
            return fail(c, HJ_ERR_INVALID, "hj_reserve: PRJ sizes must be < 2^32 tuples per device");
        uint32_t bits = params->radixBits ? params->radixBits : auto_radix_bits(rSize);
        if (bits < 1 || bits > 16) return fail(c, HJ_ERR_INVALID, "hj_reserve: radixBits must be in [1,16]");
        if (params->prjMode > 2) return fail(c, HJ_ERR_INVALID, "hj_reserve: prjMode must be 0, 1 or 2");
        c->plan = prj_plan(rSize, sSize, bits, params->prjMode);
        const uint64_t nmax = rSize > sSize ? rSize : sSize;
        int rc;
        // +2 tuples: the 16-byte sweeps may touch one tuple past an odd end
        if ((rc = grow(c, c->tmpA, c->capTmp, nmax + 2))) return rc;
        if ((rc = grow(c, c->partR, c->capPartR, rSize + 2))) return rc;
        if (sSize && (rc = grow(c, c->partS, c->capPartS, sSize + 2))) return rc;
        if (c->plan.workspaceBytes > c->capWork) {
            if (c->work) { HJ_HIP(c, hipFree(c->work)); c->work = nullptr; c->capWork = 0; }
            HJ_HIP(c, hipMalloc(&c->work, c->plan.workspaceBytes));
            c->capWork = c->plan.workspaceBytes;
        }
        if (params->algo == HJ_ALGO_PRJ) return HJ_OK;      // AUTO also needs the open-addressing buffers below
        // ... when the table join can take this R at all (it wants a power-of-two size, as the reference does);
        // otherwise AUTO simply is the radix join, which has no such restriction
        if (!is_pow2(rSize) || rSize > (1ull << 31)) return HJ_OK;
    }
    if (params->algo == HJ_ALGO_HTM) {
        // the bucketised table (HTMHashBuild.hpp:61-72): nextpow2(rSize/3 + 1) buckets of 32 bytes = 4 slots each;
        // any rSize (the hash is (key/3) & mask, not tied to rSize being a power of two)
        if (rSize > (1ull << 31)) return fail(c, HJ_ERR_INVALID, "hj_reserve: rSize > 2^31 per device");
        const uint64_t nb = htm_num_buckets(rSize);
        int rc = grow(c, c->table, c->tableCapSlots, 4 * nb + kTableSlack);
        if (rc) return rc;
        // rings (variant 3), workgroup window (2) or global atomics (1): buffers for the larger need of the first two
        size_t qb = wave_queue_bytes(rSize, c->nCU), cb = wave_conflict_bytes(rSize, c->nCU);
        if (own_supported(4 * nb)) {
            if (own_queue_bytes(rSize) > qb) qb = own_queue_bytes(rSize);
            if (own_conflict_bytes(rSize, c->nCU) > cb) cb = own_conflict_bytes(rSize, c->nCU);
            const size_t ob = own_owner_bytes(4 * nb), kb = own_conflict_count_bytes(rSize, c->nCU);
            if (ob > c->capOwner) {
                if (c->ownerBuf) { HJ_HIP(c, hipFree(c->ownerBuf)); c->ownerBuf = nullptr; c->capOwner = 0; }
                HJ_HIP(c, hipMalloc(&c->ownerBuf, ob)); c->capOwner = ob;
            }
            if (kb > c->capHtmOwnCounts) {
                if (c->htmOwnCounts) { HJ_HIP(c, hipFree(c->htmOwnCounts)); c->htmOwnCounts = nullptr; c->capHtmOwnCounts = 0; }
                HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmOwnCounts), kb)); c->capHtmOwnCounts = kb;
            }
        }
        if (qb > c->capQueue) {
            if (c->queueBuf) { HJ_HIP(c, hipFree(c->queueBuf)); c->queueBuf = nullptr; c->capQueue = 0; }
            HJ_HIP(c, hipMalloc(&c->queueBuf, qb)); c->capQueue = qb;
        }
        if (cb > c->capHtmConflicts) {
            if (c->htmConflicts) { HJ_HIP(c, hipFree(c->htmConflicts)); c->htmConflicts = nullptr; c->capHtmConflicts = 0; }
            HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmConflicts), cb)); c->capHtmConflicts = cb;
        }
        if (nb > c->capHtmBuckets) {
            for (void* p : {(void*)c->htmOvfCount, (void*)c->htmOvfBase, (void*)c->htmScan}) if (p) HJ_HIP(c, hipFree(p));
            c->htmOvfCount = nullptr; c->htmOvfBase = nullptr; c->htmScan = nullptr; c->capHtmBuckets = 0;
            HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmOvfCount), nb * sizeof(unsigned int)));
            HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmOvfBase), nb * sizeof(uint32_t)));
            HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmScan), scan_workspace_words(nb) * sizeof(uint32_t)));
            c->capHtmBuckets = nb;
        }
        return HJ_OK;
    }
    if (!is_pow2(rSize)) return fail(c, HJ_ERR_INVALID, "hj_reserve: rSize must be a power of two (DataGen.hpp:28, NoCCHashBuild.hpp:36)");
    if (rSize > (1ull << 31)) return fail(c, HJ_ERR_INVALID, "hj_reserve: rSize > 2^31 per device");
    int rc = grow(c, c->table, c->tableCapSlots, 2 * rSize + kTableSlack);
    if (rc) return rc;
    if (params->buildVariant > 4) return fail(c, HJ_ERR_INVALID, "hj_reserve: buildVariant must be 0, 1, 2, 3 or 4");
    if (params->buildVariant != 1 && (own_supported(2 * rSize) || wave_supported(2 * rSize))) {
        // 1/8 headroom: a radix shard may receive slightly more than its nominal share (hj_build_keys_dev)
        const size_t ob = own_owner_bytes(2 * rSize);
        size_t qb = own_queue_bytes(rSize + rSize / 8);
        if (wave_queue_bytes(rSize + rSize / 8, c->nCU) > qb) qb = wave_queue_bytes(rSize + rSize / 8, c->nCU);
        if (own_supported(2 * rSize) && ob > c->capOwner) {
            if (c->ownerBuf) { HJ_HIP(c, hipFree(c->ownerBuf)); c->ownerBuf = nullptr; c->capOwner = 0; }
            HJ_HIP(c, hipMalloc(&c->ownerBuf, ob)); c->capOwner = ob;
        }
        if (qb > c->capQueue) {
            if (c->queueBuf) { HJ_HIP(c, hipFree(c->queueBuf)); c->queueBuf = nullptr; c->capQueue = 0; }
            HJ_HIP(c, hipMalloc(&c->queueBuf, qb)); c->capQueue = qb;
        }
    }
    return HJ_OK;
}

// Locality pre-round, host-side form (hj_join_dev with HJ_ALGO_AUTO only: the choice between table join and radix join
// changes which buffers and kernels are used at all, so it is read back; hj_build_dev decides on the device, see
// build_common). 256 sample tiles of R. Answer = the build kernel worth taking:
//   3  the wavefront-private rings (hj_build_wave.hip) if at most 1/128 of the sampled tuples would fall outside
//      their ring (tight locality: the reference's default shuffle window of 16, anything up to ~100 positions);
//   2  the workgroup window (hj_build_own.hip) if it would have to defer at most 3/4 of the tuples (variant_for_sample,
//      hj_device.h: what it defers costs about what global atomics cost for every tuple);
//   1  global atomics otherwise (no locality: hj_join_dev(AUTO) then takes the radix join instead).
static int sample_variant(hj_ctx* c, const void* d, bool key32, uint64_t n, uint64_t tableSize, uint32_t hshift,
                          bool canOwn, bool canWave, uint32_t* variant, bool canCompact = false, bool htm = false)
{
    const uint32_t nSample = 256;
    HJ_HIP(c, launch_sample_locality(d, key32, n, tableSize, hshift, nSample, c->fitCount, c->stream, htm));
    HJ_HIP(c, hipMemcpyAsync(c->hFit, c->fitCount, 8 * sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    *variant = variant_for_sample(c->hFit[0], c->hFit[1], c->hFit[2], canOwn, canWave, canCompact, c->hFit[3], c->hFit[4]);
    return HJ_OK;
}

// Shared by hj_build_dev (DataGen tuples) and hj_build_keys_dev (bare keys of a radix shard).
static int build_common(hj_ctx* c, const void* d, bool key32, uint64_t n, uint32_t hshift,
                        uint64_t tableSize, uint64_t idxBase)
{
    HJ_HIP(c, hipSetDevice(c->device));
    c->rSize = n; c->sSize = 0; c->tableSize = tableSize; c->hshift = hshift;
    for (bool& b : c->evSet) b = false;
    c->prjRan = false; c->htmBuilt = false;
    HJ_HIP(c, hipMemsetAsync(c->dCtr, 0, sizeof(Counters), c->stream));
    int rc;
    if ((rc = record(c, EV_CLEAR0))) return rc;
    // which build kernel: 2, 3 and 4 need their buffers (hj_reserve) and a table of at least one window / ring.
    // 4 = the compact ring build (4-byte table, hj_build_wave.hip): what "rings" means whenever it can be tried; if it meets
    // something it cannot handle, the classic ring build (3) enqueued behind it, gated on the device, redoes the table.
    uint32_t variant = c->forceVariant ? c->forceVariant : c->params.buildVariant;
    const bool canOwn = n && own_supported(tableSize) && c->capOwner >= own_owner_bytes(tableSize) &&
                        c->capQueue >= own_queue_bytes(n);
    const bool canWave = n && wave_supported(tableSize) && c->capQueue >= wave_queue_bytes(n, c->nCU);
    const uint32_t pl = probe_len(c->params);
    const bool canCompact = canWave && wave_compact_supported(tableSize, pl);
    if (variant == 4 && !canCompact) variant = 3;
    if (variant == 3 && !canWave) variant = canOwn ? 2 : 1;
    if (variant == 2 && !canOwn) variant = 1;
    if (variant == 0 && !canOwn && !canWave) variant = 1;
    c->variantUsed = (variant == 4) ? 0 : variant;     // 0: decided on the device (4 may fall back to 3), reported from Counters::variant
    c->algoUsed = c->params.algo == HJ_ALGO_AUTO ? (uint32_t)HJ_ALGO_ATOMIC : c->params.algo;
    const unsigned long long* word = &c->dCtr->variant;
    // the dominant kernel of each LDS variant, bracketed by its own pair of events (hj_result.buildPhaseA_us)
    const KernelEvents kevWave{c->ev[EV_KW0], c->ev[EV_KW1]}, kevCompact{c->ev[EV_KC0], c->ev[EV_KC1]}, kevOwn{c->ev[EV_KO0], c->ev[EV_KO1]};
    if (variant == 0) {
        // The locality pre-round decides ON THE DEVICE (this call stays asynchronous: no read-back). Behind it the kernels
        // of the candidate variants are enqueued, each gated on the word the pre-round writes; the ones not chosen return
        // at once (~4.5 us each). Which candidates: all of them the first time; afterwards only what the context's PREVIOUS
        // pick preferred (Counters::preferred, which the sampler also stores into pinned host memory, read here without
        // waiting) -- with the classic rings behind the compact ones. The pick is taken among the enqueued variants, and
        // every LDS build is correct on any input (what does not fit its rings / window goes through its deferred phase),
        // so a workload that changes its locality class costs one slow step, never a wrong one, and the next step follows
        // it (round-2 VERDICT, launch tail: 23 -> 9 launches in the steady state).
        const uint32_t expect = (uint32_t)*reinterpret_cast<volatile unsigned long long*>(c->hPreferred);
        const uint32_t all = 2u | (canOwn ? 4u : 0u) | (canWave ? 8u : 0u) | (canCompact ? 16u : 0u);
        uint32_t allowed = all;
        if (expect == 4 && canCompact) allowed = 16u | 8u;
        else if (expect == 3 && canWave) allowed = 8u;
        else if (expect == 2 && canOwn) allowed = 4u;
        else if (expect == 1) allowed = 2u;
        const bool enqCompact = (allowed >> 4) & 1u, enqWave = (allowed >> 3) & 1u, enqOwn = (allowed >> 2) & 1u, enqGlobal = (allowed >> 1) & 1u;
        SamplePick pick;
        pick.ctr = c->dCtr; pick.hostPreferred = c->dPreferred; pick.allowedMask = allowed;
        pick.canOwn = canOwn; pick.canWave = canWave; pick.canCompact = canCompact;
        HJ_HIP(c, launch_sample_locality(d, key32, n, tableSize, hshift, 256, c->dCtr->fit, c->stream, false, pick, true));
        if ((rc = record(c, EV_BUILD0))) return rc;
        // phase A of the LDS variants, each dominant kernel inside its own pair of events, then their tails. (An event
        // record costs the stream ~5 us -- step timeline at 2^22, tools/step_timeline.sh --, so nothing is recorded that
        // hj_fetch does not read.)
        if (enqWave) {
            HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                        c->queueBuf, c->dCtr, Gate{word, 3, 4}, kWavePre, nullptr, c->stream));
            if (enqCompact)
                HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                            c->queueBuf, c->dCtr, Gate{word, 4}, kWaveMain, nullptr, c->stream, nullptr, kWaveCompact, 3, &kevCompact));
            HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                        c->queueBuf, c->dCtr, Gate{word, 3}, kWaveMain, nullptr, c->stream, nullptr, kWaveClassic, 3, &kevWave));
            c->evSet[EV_KW0] = c->evSet[EV_KW1] = true;
            c->evSet[EV_KC0] = c->evSet[EV_KC1] = enqCompact;
        }
        if (enqOwn) {
            HJ_HIP(c, launch_build_own(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->ownerBuf,
                                       c->queueBuf, c->queueCount, c->dCtr, Gate{word, 2}, 1, nullptr, c->stream, &kevOwn));
            c->evSet[EV_KO0] = c->evSet[EV_KO1] = true;
        }
        if (enqWave) {
            if (enqCompact)
                HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                            c->queueBuf, c->dCtr, Gate{word, 4}, kWaveTail, nullptr, c->stream, nullptr, kWaveCompact, 3));
            HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                        c->queueBuf, c->dCtr, Gate{word, 3}, kWaveTail, nullptr, c->stream));
        }
        if (enqOwn)
            HJ_HIP(c, launch_build_own(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->ownerBuf,
                                       c->queueBuf, c->queueCount, c->dCtr, Gate{word, 2}, 2, nullptr, c->stream));
        if (enqGlobal) {
            launch_fill_empty(c->table, tableSize + kTableSlack, Gate{word, 1}, c->stream, c->dCtr, tableSize);
            launch_build_atomic_min(d, key32, n, c->table, tableSize, hshift, pl, idxBase, c->sc, c->dCtr, Gate{word, 1}, c->stream);
        }
    } else if (variant == 4) {
        // the compact rings, asked for by the caller: the classic rings stay enqueued behind them as the gated fallback
        launch_set_variant(c->dCtr, 4, c->stream);
        if ((rc = record(c, EV_BUILD0))) return rc;
        HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                    c->queueBuf, c->dCtr, Gate{word, 4}, kWavePre | kWaveMain, nullptr, c->stream, nullptr, kWaveCompact, 3, &kevCompact));
        HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                    c->queueBuf, c->dCtr, Gate{word, 3}, kWaveMain, nullptr, c->stream, nullptr, kWaveClassic, 3, &kevWave));
        c->evSet[EV_KW0] = c->evSet[EV_KW1] = c->evSet[EV_KC0] = c->evSet[EV_KC1] = true;
        HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                    c->queueBuf, c->dCtr, Gate{word, 4}, kWaveTail, nullptr, c->stream, nullptr, kWaveCompact, 3));
        HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU, c->boundsBuf,
                                    c->queueBuf, c->dCtr, Gate{word, 3}, kWaveTail, nullptr, c->stream));
    } else if (variant == 3) {
        if ((rc = record(c, EV_BUILD0))) return rc;
        HJ_HIP(c, launch_build_wave(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU,
                                    c->boundsBuf, c->queueBuf, c->dCtr, Gate{nullptr, 0}, kWaveAll, nullptr, c->stream, nullptr, kWaveClassic, 3, &kevWave));
        c->evSet[EV_KW0] = c->evSet[EV_KW1] = true;
    } else if (variant == 2) {
        if ((rc = record(c, EV_BUILD0))) return rc;
        HJ_HIP(c, launch_build_own(d, key32, n, hshift, c->table, tableSize, pl, idxBase, c->sc, c->nCU,
                                   c->ownerBuf, c->queueBuf, c->queueCount, c->dCtr, Gate{nullptr, 0}, 3, nullptr, c->stream, &kevOwn));
        c->evSet[EV_KO0] = c->evSet[EV_KO1] = true;
    } else {
        launch_fill_empty(c->table, tableSize + kTableSlack, Gate{nullptr, 0}, c->stream);
        launch_set_full_range(tableSize, c->dCtr, Gate{nullptr, 0}, c->stream);
        HJ_HIP(c, hipGetLastError());
        if ((rc = record(c, EV_BUILD0))) return rc;
        if (n) launch_build_atomic_min(d, key32, n, c->table, tableSize, hshift, pl, idxBase, c->sc, c->dCtr, Gate{nullptr, 0}, c->stream);
    }
    HJ_HIP(c, hipGetLastError());
    if ((rc = record(c, EV_BUILD1))) return rc;
    c->built = true;
    c->streamAtBuildEnd = c->ownStream; c->probeStartsAtBuildEnd = false;
    return HJ_OK;
}

// --algo htm: the bucketised table (hj_htm.hip). Not asynchronous: the number of conflicts is read back once, to size
// the overflow area (the reference, too, builds its chains in a serial phase after the parallel build, :231-279).
static int build_htm(hj_ctx* c, const uint64_t* dR, uint64_t rSize, uint64_t idxBase)
{
    HJ_HIP(c, hipSetDevice(c->device));
    const uint32_t nb = htm_num_buckets(rSize);
    const uint64_t slots = 4ull * nb;
    if (slots + kTableSlack > c->tableCapSlots || nb > c->capHtmBuckets || c->capQueue < wave_queue_bytes(rSize, c->nCU) ||
        c->capHtmConflicts < wave_conflict_bytes(rSize, c->nCU))
        return fail(c, HJ_ERR_STATE, "hj_build_dev: hj_reserve() not called for this rSize (htm)");
    if (idxBase + rSize > 0xFFFFFFFFull) return fail(c, HJ_ERR_INVALID, "hj_build_dev: index range exceeds 2^32 - 1");
    c->rSize = rSize; c->sSize = 0; c->tableSize = slots; c->hshift = 0; c->htmBuckets = nb;
    for (bool& b : c->evSet) b = false;
    c->prjRan = false; c->built = false; c->htmBuilt = false;
    if (!c->htmGenericChains) c->htmChainsFellBack = false;
    HJ_HIP(c, hipMemsetAsync(c->dCtr, 0, sizeof(Counters), c->stream));
    int rc;
    if ((rc = record(c, EV_CLEAR0))) return rc;
    // locality pre-round with the bucketised table's own hash (bucket = key / 3 keeps the key order): the rings if they
    // will do, the workgroup window for looser locality (shuffle windows up to ~2000 positions), else global atomics
    const bool canWave = wave_supported(slots);
    const bool canOwn = own_supported(slots) && c->capOwner >= own_owner_bytes(slots) && c->capQueue >= own_queue_bytes(rSize) &&
                        c->capHtmConflicts >= own_conflict_bytes(rSize, c->nCU) && c->capHtmOwnCounts >= own_conflict_count_bytes(rSize, c->nCU);
    uint32_t variant = c->params.buildVariant > 3 ? 3 : c->params.buildVariant;
    if (variant == 0 && (canWave || canOwn) &&
        (rc = sample_variant(c, dR, false, rSize, slots, 0, canOwn, canWave, &variant, false, true))) return rc;
    if (variant == 0) variant = 1;
    if (variant == 3 && !canWave) variant = canOwn ? 2 : 1;
    if (variant == 2 && !canOwn) variant = 1;
    c->variantUsed = variant; c->algoUsed = HJ_ALGO_HTM;
    const WaveSlices sl = variant == 2 ? own_conflict_layout(rSize, c->nCU, c->htmOwnCounts) : wave_conflict_layout(rSize, c->nCU, c->boundsBuf);
    // the rings: chains in LDS (hj_htm.hip) unless an earlier attempt on this relation had to give up
    const uint32_t nParts = sl.nChunks * htm_chain_parts(sl.sliceLen);
    const bool ldsChains = variant == 3 && !c->htmGenericChains && (uint64_t)nParts + 1 <= nb && htm_chain_info_words(sl.nChunks, sl.sliceLen) <= nb;
    const KernelEvents kevW{c->ev[EV_KW0], c->ev[EV_KW1]};
    const KernelEvents kevO{c->ev[EV_KO0], c->ev[EV_KO1]};
    if (variant == 2) {
        if ((rc = record(c, EV_BUILD0))) return rc;
        HJ_HIP(c, launch_build_own(dR, false, rSize, 0, c->table, slots, 3, idxBase, ShardCheck{0, 0, 0, 0}, c->nCU, c->ownerBuf,
                                   c->queueBuf, c->queueCount, c->dCtr, Gate{nullptr, 0}, 3, nullptr, c->stream, &kevO,
                                   c->htmConflicts, c->htmOwnCounts));
        c->evSet[EV_KO0] = c->evSet[EV_KO1] = true;
    } else if (variant == 3) {
        if ((rc = record(c, EV_BUILD0))) return rc;
        HJ_HIP(c, launch_build_wave(dR, false, rSize, 0, c->table, slots, 3, idxBase, ShardCheck{0, 0, 0, 0}, c->nCU, c->boundsBuf,
                                    c->queueBuf, c->dCtr, Gate{nullptr, 0}, kWaveAll, nullptr, c->stream, c->htmConflicts, kWaveClassic, 3, &kevW,
                                    ldsChains));
        c->evSet[EV_KW0] = c->evSet[EV_KW1] = true;
        if (ldsChains) {
            // the chain phase in LDS, first half: overflow buckets per part of a slice, scanned (one word more: the total)
            HJ_HIP(c, hipMemsetAsync(c->htmOvfBase + nParts, 0, sizeof(uint32_t), c->stream));
            HJ_HIP(c, launch_htm_chain_count(c->htmConflicts, sl.counts, wave_bounds_ptr(c->nCU, c->boundsBuf), sl.nChunks, sl.sliceLen, nb,
                                             c->htmOvfBase, c->htmOvfCount, c->dCtr, c->stream));
            HJ_HIP(c, launch_exclusive_scan_u32(c->htmOvfBase, (uint64_t)nParts + 1, c->htmScan, c->stream));
            HJ_HIP(c, hipMemcpyAsync(c->hFit, c->htmOvfBase + nParts, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        }
    } else {
        launch_fill_empty(c->table, slots + kTableSlack, Gate{nullptr, 0}, c->stream);
        launch_set_full_range(slots, c->dCtr, Gate{nullptr, 0}, c->stream);
        HJ_HIP(c, hipGetLastError());
        if ((rc = record(c, EV_BUILD0))) return rc;
        HJ_HIP(c, launch_htm_build_global(dR, rSize, sl.sliceLen, sl.nChunks, c->table, slots, idxBase, c->htmConflicts,
                                          const_cast<uint32_t*>(sl.counts), c->dCtr, c->stream));
    }
    // chains -- only if some bucket overflowed (one read-back of the conflict count): count per bucket, reserve overflow
    // buckets by one scan, fill them in index order, link
    HJ_HIP(c, hipMemcpyAsync(c->hCtr, c->dCtr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    fold_counter_shards(c->hCtr);
    const uint64_t conflicts = c->hCtr->conflicts;              // >= overflow buckets needed
    c->htmOverflowUsed = conflicts;
    if (ldsChains && c->hCtr->htmChainBail) {
        // an input the LDS chain phase cannot take (hj_htm.hip): once more, unrouted, with the generic chain kernels
        c->htmGenericChains = true;
        rc = build_htm(c, dR, rSize, idxBase);
        c->htmGenericChains = false;
        c->htmChainsFellBack = true;
        return rc;
    }
    if (ldsChains) {
        const uint64_t groups = c->hFit[0];                     // overflow buckets the parts need, exactly
        if (conflicts) {
            if (groups + 1 > c->capHtmOverflow) {
                if (c->htmOverflow) { HJ_HIP(c, hipFree(c->htmOverflow)); c->htmOverflow = nullptr; c->capHtmOverflow = 0; }
                const uint64_t cap = groups + groups / 8 + 64;
                HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmOverflow), (cap + 1) * 4 * sizeof(uint64_t)));
                c->capHtmOverflow = cap + 1;
            }
            HJ_HIP(c, launch_htm_chain_fill(c->htmConflicts, sl.nChunks, sl.sliceLen, nb, c->htmOvfBase, c->htmOvfCount, c->table,
                                            c->htmOverflow, c->dCtr, c->stream));
        }
    } else if (conflicts) {
        HJ_HIP(c, launch_htm_count(c->htmConflicts, sl.counts, sl.nChunks, sl.sliceLen, nb, c->htmOvfCount, c->htmOvfBase, c->stream));
        HJ_HIP(c, launch_exclusive_scan_u32(c->htmOvfBase, nb, c->htmScan, c->stream));
        if (conflicts + 1 > c->capHtmOverflow) {
            HJ_HIP(c, hipStreamSynchronize(c->stream));
            if (c->htmOverflow) { HJ_HIP(c, hipFree(c->htmOverflow)); c->htmOverflow = nullptr; c->capHtmOverflow = 0; }
            const uint64_t cap = conflicts + conflicts / 8 + 64;
            HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->htmOverflow), (cap + 1) * 4 * sizeof(uint64_t)));
            c->capHtmOverflow = cap + 1;
        }
        HJ_HIP(c, launch_htm_chains(c->htmConflicts, sl.counts, sl.nChunks, sl.sliceLen, c->table, nb, c->htmOvfCount, c->htmOvfBase,
                                    c->htmOverflow, conflicts, c->dCtr, c->stream));
    }
    if ((rc = record(c, EV_BUILD1))) return rc;
    c->built = true; c->htmBuilt = true;
    c->streamAtBuildEnd = c->ownStream; c->probeStartsAtBuildEnd = false;
    return HJ_OK;
}

int hj_build_dev(hj_ctx* c, const uint64_t* dR, uint64_t rSize, uint64_t idxBase)
{
    if (!c || !dR) return HJ_ERR_INVALID;
    if (c->params.algo == HJ_ALGO_PRJ) return fail(c, HJ_ERR_STATE, "hj_build_dev: context is reserved for PRJ");
    if (c->params.algo == HJ_ALGO_HTM) return rSize ? build_htm(c, dR, rSize, idxBase) : HJ_ERR_INVALID;
    if (!is_pow2(rSize) || 2 * rSize + kTableSlack > c->tableCapSlots)
        return fail(c, HJ_ERR_STATE, "hj_build_dev: hj_reserve() not called for this rSize");
    // indices stay below 2^32 - 1: (index << 32 | key) of index = key = 0xFFFFFFFF would be the empty pattern
    if (idxBase + rSize > 0xFFFFFFFFull) return fail(c, HJ_ERR_INVALID, "hj_build_dev: index range exceeds 2^32 - 1");
    return build_common(c, dR, false, rSize, 0, 2 * rSize, idxBase);
}

int hj_build_keys_dev(hj_ctx* c, const uint32_t* dKeys, uint64_t n, uint32_t homeShift, uint64_t tableSize)
{
    if (!c || (!dKeys && n)) return HJ_ERR_INVALID;
    if (c->params.algo == HJ_ALGO_PRJ || c->params.algo == HJ_ALGO_HTM)
        return fail(c, HJ_ERR_STATE, "hj_build_keys_dev: context is reserved for PRJ / htm");
    if (homeShift > 6) return fail(c, HJ_ERR_INVALID, "hj_build_keys_dev: homeShift must be in [0,6]");
    if (!is_pow2(tableSize) || tableSize + kTableSlack > c->tableCapSlots)
        return fail(c, HJ_ERR_STATE, "hj_build_keys_dev: hj_reserve() not called for this table size");
    if (n > 0xFFFFFFFFull) return fail(c, HJ_ERR_INVALID, "hj_build_keys_dev: index range exceeds 2^32 - 1");
    return build_common(c, dKeys, true, n, homeShift, tableSize, 0);
}

int hj_probe_dev(hj_ctx* c, const uint64_t* dS, uint64_t sSize)
{
    if (!c || (!dS && sSize)) return HJ_ERR_INVALID;
    if (!c->built) return fail(c, HJ_ERR_STATE, "hj_probe_dev: no table (call hj_build_dev first)");
    HJ_HIP(c, hipSetDevice(c->device));
    int rc;
    // the probe's start: the build's end event if this is the very next thing on the library's own stream (one event
    // record less per step), an event of its own otherwise
    c->probeStartsAtBuildEnd = c->streamAtBuildEnd;
    c->streamAtBuildEnd = false;
    if (!c->probeStartsAtBuildEnd && (rc = record(c, EV_PROBE0))) return rc;
    if (sSize && c->htmBuilt) launch_htm_probe(dS, sSize, c->table, c->htmBuckets, c->htmOverflow, c->dCtr, c->stream);
    else if (sSize) launch_probe(dS, false, sSize, c->table, c->tableSize, c->hshift, probe_len(c->params), c->sc, c->dCtr, c->stream);
    if ((rc = record(c, EV_PROBE1))) return rc;
    HJ_HIP(c, hipGetLastError());
    c->sSize += sSize;
    return HJ_OK;
}

int hj_probe_keys_dev(hj_ctx* c, const uint32_t* dKeys, uint64_t n)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || (!dKeys && n)) return HJ_ERR_INVALID;
    if (!c->built || c->htmBuilt) return fail(c, HJ_ERR_STATE, "hj_probe_keys_dev: no open-addressing table (build first)");
    HJ_HIP(c, hipSetDevice(c->device));
    int rc;
    c->probeStartsAtBuildEnd = false;
    if ((rc = record(c, EV_PROBE0))) return rc;
    if (n) launch_probe(dKeys, true, n, c->table, c->tableSize, c->hshift, probe_len(c->params), c->sc, c->dCtr, c->stream);
    if ((rc = record(c, EV_PROBE1))) return rc;
    HJ_HIP(c, hipGetLastError());
    c->sSize += n;
    return HJ_OK;
}

int hj_prj_join_dev(hj_ctx* c, const uint64_t* dR, uint64_t rSize, const uint64_t* dS, uint64_t sSize)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || !dR || rSize == 0) return HJ_ERR_INVALID;
    if (sSize == 0) dS = nullptr;                    // an empty S is no S (the join kernel clamps its loads to nS - 1)
    if (c->params.algo != HJ_ALGO_PRJ && c->params.algo != HJ_ALGO_AUTO)
        return fail(c, HJ_ERR_STATE, "hj_prj_join_dev: context not reserved for PRJ");
    const uint64_t nmax = rSize > sSize ? rSize : sSize;
    if (nmax + 2 > c->capTmp || rSize + 2 > c->capPartR || (dS && sSize + 2 > c->capPartS))
        return fail(c, HJ_ERR_STATE, "hj_prj_join_dev: hj_reserve() not called for these sizes");
    HJ_HIP(c, hipSetDevice(c->device));
    // the plan depends on the sizes (chunking); re-plan with the reserved bit count
    const PrjPlan pl = prj_plan(rSize, dS ? sSize : 0, c->plan.radixBits, c->params.prjMode);
    if (pl.workspaceBytes > c->capWork) return fail(c, HJ_ERR_STATE, "hj_prj_join_dev: workspace too small");
    for (bool& b : c->evSet) b = false;
    c->built = false;
    c->rSize = rSize; c->sSize = dS ? sSize : 0; c->tableSize = 0;
    HJ_HIP(c, hipMemsetAsync(c->dCtr, 0, sizeof(Counters), c->stream));
    int rc;
    if ((rc = record(c, EV_PRJ0))) return rc;
    PrjBuffers buf{c->tmpA, c->partR, c->partS, c->work};
    HJ_HIP(c, launch_prj(pl, buf, dR, rSize, dS, dS ? sSize : 0, c->nCU, c->dCtr, c->ev[EV_PRJ_PART], c->ev[EV_PRJ_S0],
                         c->ev[EV_PRJ_S1], c->stream));
    c->evSet[EV_PRJ_PART] = c->evSet[EV_PRJ_S0] = c->evSet[EV_PRJ_S1] = true;
    if ((rc = record(c, EV_PRJ1))) return rc;
    HJ_HIP(c, hipGetLastError());
    c->prjRan = true;
    c->prjOptimistic = pl.optimistic;
    c->algoUsed = HJ_ALGO_PRJ;
    return HJ_OK;
}

int hj_join_dev(hj_ctx* c, const uint64_t* dR, uint64_t rSize, const uint64_t* dS, uint64_t sSize)
{
    if (!c || !dR || rSize == 0) return HJ_ERR_INVALID;
    if (!dS) sSize = 0;
    if (sSize == 0) dS = nullptr;
    int rc;
    bool prj = c->params.algo == HJ_ALGO_PRJ;
    uint32_t force = 0;
    if (c->params.algo == HJ_ALGO_AUTO) {
        // the same question the build asks itself for buildVariant 0, asked once here: with locality the
        // LDS-window build + linear probe wins, without it both of them turn into random HBM accesses
        // and two radix passes are cheaper
        HJ_HIP(c, hipSetDevice(c->device));
        if (!is_pow2(rSize) || rSize > (1ull << 31)) return hj_prj_join_dev(c, dR, rSize, dS, sSize);   // see hj_reserve
        if (2 * rSize + kTableSlack > c->tableCapSlots)
            return fail(c, HJ_ERR_STATE, "hj_join_dev: hj_reserve() not called for this rSize");
        const bool canOwn = own_supported(2 * rSize) && c->capOwner >= own_owner_bytes(2 * rSize) &&
                            c->capQueue >= own_queue_bytes(rSize);
        const bool canWave = wave_supported(2 * rSize) && c->capQueue >= wave_queue_bytes(rSize, c->nCU);
        uint32_t v = 1;
        const bool canCompact = canWave && wave_compact_supported(2 * rSize, probe_len(c->params));
        if ((canOwn || canWave) && c->params.buildVariant != 1 &&
            (rc = sample_variant(c, dR, false, rSize, 2 * rSize, 0, canOwn, canWave, &v, canCompact))) return rc;
        // no locality: both table phases would be random HBM accesses. Loose locality (variant 2) pays for every tuple
        // that leaves its window with global atomics: at 2^27, local_shuffle W=2^11 (3.8 % deferred) the table join
        // takes 1.09 ms against the radix join's 1.81 ms, at W=2^12 (36 % deferred) 2.79 against 1.83 (round 3, deferred
        // queue sliced per workgroup; 2.06 / 1.89 at W=2^11 before): the radix join from 1/8 of the sample outside the window.
        prj = v == 1 || (v == 2 && (uint64_t)c->hFit[0] * 8u > c->hFit[1]);
        force = v;
    }
    if (prj) return hj_prj_join_dev(c, dR, rSize, dS, sSize);
    c->forceVariant = force;
    rc = hj_build_dev(c, dR, rSize, 0);
    c->forceVariant = 0;
    if (rc) return rc;
    return hj_probe_dev(c, dS, sSize);
}

int hj_checksums_dev(hj_ctx* c)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c) return HJ_ERR_INVALID;
    if (!c->built) return fail(c, HJ_ERR_STATE, "hj_checksums_dev: no table");
    HJ_HIP(c, hipSetDevice(c->device));
    // zero the two sums so the call is idempotent
    HJ_HIP(c, hipMemsetAsync(&c->dCtr->tableSumHalf, 0, 2 * sizeof(unsigned long long), c->stream));
    if (c->htmBuilt) {
        HJ_HIP(c, hipMemsetAsync(&c->dCtr->htmOverflowSum, 0, sizeof(unsigned long long), c->stream));
        launch_htm_sums(c->table, c->htmBuckets, c->htmOverflow, c->dCtr, c->stream);
        HJ_HIP(c, hipGetLastError());
        return HJ_OK;
    }
    launch_table_sums(c->table, c->tableSize, c->tableSize / 2, c->dCtr, c->stream);
    HJ_HIP(c, hipGetLastError());
    return HJ_OK;
}

int hj_fetch_result(hj_ctx* c, hj_result* out)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || !out) return HJ_ERR_INVALID;
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipMemcpyAsync(c->hCtr, c->dCtr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    fold_counter_shards(c->hCtr);
    memset(out, 0, sizeof(*out));
    const Counters& k = *c->hCtr;
    out->rSize = c->rSize; out->sSize = c->sSize; out->tableSize = c->tableSize;
    out->inputSum = k.inputSum;
    if (c->prjRan) {
        out->totalMatches = k.prjMatches;
        out->prjChecksum = k.prjChecksum;
        out->prjPartitions = 1ull << c->plan.radixBits;
        out->radixBits = c->plan.radixBits;
        out->partition_us = elapsed_us(c, EV_PRJ0, EV_PRJ_PART);
        out->join_us = elapsed_us(c, EV_PRJ_PART, EV_PRJ1);
        out->total_us = elapsed_us(c, EV_PRJ0, EV_PRJ1);
        out->prjScatterPass1R_us = elapsed_us(c, EV_PRJ_S0, EV_PRJ_S1);
        out->prjPath = !c->prjOptimistic ? 0u : (k.prjFallback ? 2u : 1u);
    } else {
        out->conflicts = k.conflicts;
        out->conflictSum = k.conflictSum;
        out->totalMatches = k.matches;
        out->tableSumHalf = k.tableSumHalf;
        out->tableSumFull = k.tableSumFull;
        out->outputSum = (c->params.algo == HJ_ALGO_NOCC ? k.tableSumHalf : k.tableSumFull) + k.conflictSum;
        if (c->htmBuilt) {
            // every conflict sits in an overflow bucket of its own bucket's chain: tuples in buckets + tuples in
            // chains = input (HTMHashBuild.hpp:452 adds conflictSum on top of the chains, counting them twice)
            out->htmBuckets = c->htmBuckets; out->htmOverflowBuckets = k.htmOverflowBuckets; out->htmOverflowSum = k.htmOverflowSum;
            out->outputSum = k.tableSumFull + k.htmOverflowSum;
        }
        out->buildVariant = c->variantUsed ? c->variantUsed : (uint32_t)k.variant;   // 0: the device chose
        out->compactFallback = k.compactFail | ((c->htmBuilt && c->htmChainsFellBack) ? 0x100ull : 0ull);
        out->buildDeferred = k.deferred;
        // the dominant build kernel ALONE: the launch of the LDS build that ran is bracketed by its own pair of events (the
        // launches of the variants the device did not pick return at once: microseconds; the largest bracket is the kernel)
        out->buildPhaseA_us = 0.0;
        static const Ev kBrackets[3][2] = {{EV_KW0, EV_KW1}, {EV_KC0, EV_KC1}, {EV_KO0, EV_KO1}};
        for (const auto& pr : kBrackets) {
            const double us = elapsed_us(c, pr[0], pr[1]);
            if (us > out->buildPhaseA_us) out->buildPhaseA_us = us;
        }
        out->clear_us = elapsed_us(c, EV_CLEAR0, EV_BUILD0);
        out->build_us = elapsed_us(c, EV_BUILD0, EV_BUILD1);
        out->probe_us = elapsed_us(c, c->probeStartsAtBuildEnd ? EV_BUILD1 : EV_PROBE0, EV_PROBE1);
        // the reference's timed region is build+probe, table zeroing excluded
        // (NoCCHashBuild.hpp:24-34,83); clear_us is reported beside it
        out->total_us = out->build_us + out->probe_us;
    }
    out->h2d_us = c->h2d_us;
    out->algoUsed = c->algoUsed;
    out->foreignTuples = k.foreign;
    if (k.badKeys) return fail(c, HJ_ERR_KEY_RANGE, "input holds tuples with payload bits set or value 0");
    return HJ_OK;
}

int hj_export_table(hj_ctx* c, uint64_t* host_table, uint64_t tableSize)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || !host_table) return HJ_ERR_INVALID;
    if (!c->built || c->htmBuilt || tableSize != c->tableSize) return fail(c, HJ_ERR_STATE, "hj_export_table: no open-addressing table of that size");
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    Counters k;
    HJ_HIP(c, hipMemcpy(&k, c->dCtr, sizeof(k), hipMemcpyDeviceToHost));
    // only [validLo, validHiEx + 512) holds defined values (hj_device.h); the rest is empty by definition
    const uint64_t lo = k.validLo, hi = k.validHiEx + 512 < tableSize ? k.validHiEx + 512 : tableSize;
    if (k.tableFormat == kFormatKeys4) {
        // compact device format (4-byte keys, 0xFFFFFFFF = empty) -> reference format (key, 0 = empty): the keys land in
        // the upper half of the caller's buffer and are widened from the front (the write position never passes the read one)
        uint32_t* keys = reinterpret_cast<uint32_t*>(host_table) + tableSize;
        HJ_HIP(c, hipMemcpy(keys, c->table, tableSize * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < tableSize; ++i) {
            const uint32_t v = keys[i];
            host_table[i] = (i < lo || i >= hi || v == 0xFFFFFFFFu) ? 0 : v;
        }
        return HJ_OK;
    }
    HJ_HIP(c, hipMemcpy(host_table, c->table, tableSize * sizeof(uint64_t), hipMemcpyDeviceToHost));
    // device format (index << 32 | key, all ones = empty) -> reference format (key, 0 = empty)
    for (uint64_t i = 0; i < tableSize; ++i)
        host_table[i] = (i < lo || i >= hi || host_table[i] == kEmpty) ? 0 : (uint32_t)host_table[i];
    return HJ_OK;
}

int hj_export_buckets(hj_ctx* c, void* host_buckets, uint64_t numBuckets, void* host_overflows, uint64_t overflowCap,
                      uint64_t* nOverflow)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || !host_buckets) return HJ_ERR_INVALID;
    if (!c->htmBuilt || numBuckets != c->htmBuckets) return fail(c, HJ_ERR_STATE, "hj_export_buckets: no htm table of that many buckets");
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipMemcpyAsync(c->hCtr, c->dCtr, sizeof(Counters), hipMemcpyDeviceToHost, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    fold_counter_shards(c->hCtr);
    const uint64_t used = c->hCtr->htmOverflowBuckets;
    if (nOverflow) *nOverflow = used;
    if (used && (!host_overflows || overflowCap < used + 1)) return fail(c, HJ_ERR_INVALID, "hj_export_buckets: overflow buffer too small");
    // device format (index << 32 | key, all ones = empty; slot 3 = next << 32 | count, or all ones = "no chain") ->
    // struct Bucket {tuples[3], count, nextIndex}; buckets [lo, hi) are the ones the build defined, the rest are empty
    auto convert = [](uint64_t* b, uint64_t n, uint64_t lo, uint64_t hi) {
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t* p = b + 4 * i;
            if (i < lo || i >= hi) { p[0] = p[1] = p[2] = p[3] = 0; continue; }
            uint32_t count = 0;
            for (int j = 0; j < 3; ++j) { count += p[j] != kEmpty; p[j] = (p[j] == kEmpty) ? 0 : (uint32_t)p[j]; }
            const uint32_t next = p[3] == kEmpty ? 0u : (uint32_t)(p[3] >> 32);
            p[3] = (uint64_t)count | ((uint64_t)next << 32);     // little-endian {uint32 count; uint32 nextIndex}
        }
    };
    const uint64_t defLo = c->hCtr->validLo >> 2;
    uint64_t defHi = (c->hCtr->validHiEx + 512) >> 2;
    defHi = defHi < numBuckets ? defHi : numBuckets;
    HJ_HIP(c, hipMemcpy(host_buckets, c->table, numBuckets * 32, hipMemcpyDeviceToHost));
    convert(static_cast<uint64_t*>(host_buckets), numBuckets, defLo, defHi);
    if (host_overflows) {
        memset(host_overflows, 0, 32);                           // index 0 is unused (as in the reference)
        if (used) {
            HJ_HIP(c, hipMemcpy(static_cast<char*>(host_overflows) + 32, c->htmOverflow + 4, used * 32, hipMemcpyDeviceToHost));
            convert(static_cast<uint64_t*>(host_overflows) + 4, used, 0, used);
        }
    }
    return HJ_OK;
}

int hj_run(hj_ctx* c, const hj_params* params, const uint64_t* relR, uint64_t rSize,
           const uint64_t* relS, uint64_t sSize, hj_result* out)
{
    if (!c || !params || !relR || !out) return HJ_ERR_INVALID;
    if (!relS) sSize = 0;
    int rc;
    if ((rc = hj_reserve(c, params, rSize, sSize))) return rc;
    if ((rc = grow(c, c->stageR, c->capStageR, rSize + 2))) return rc;
    if (sSize && (rc = grow(c, c->stageS, c->capStageS, sSize + 2))) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    HJ_HIP(c, hipMemcpyAsync(c->stageR, relR, rSize * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    if (sSize) HJ_HIP(c, hipMemcpyAsync(c->stageS, relS, sSize * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    c->h2d_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if ((rc = hj_join_dev(c, c->stageR, rSize, sSize ? c->stageS : nullptr, sSize))) return rc;
    if (!c->prjRan && (rc = hj_checksums_dev(c))) return rc;
    return hj_fetch_result(c, out);
}

int hj_prj_fragment_info(uint64_t rSize, uint64_t sSize, uint32_t radixBits, uint32_t prjMode, uint64_t out[13])
{
    if (!out || radixBits > 16 || prjMode > 2) return HJ_ERR_INVALID;
    const uint32_t bits = radixBits ? radixBits : auto_radix_bits(rSize);
    const PrjPlan pl = prj_plan(rSize, sSize, bits, prjMode);
    out[0] = pl.optimistic ? 1 : 0;
    const PrjFrag* g[2] = {&pl.fragR, &pl.fragS};
    for (int k = 0; k < 2; ++k) {
        uint64_t* o = out + 1 + 5 * k;
        o[0] = g[k]->C1; o[1] = g[k]->cap1; o[2] = g[k]->chunkLen1; o[3] = g[k]->C2; o[4] = g[k]->cap2;
    }
    out[11] = pl.bits1; out[12] = pl.bits2;
    return HJ_OK;
}

int hj_prj_workspace_info(uint64_t rSize, uint64_t sSize, uint32_t radixBits, uint64_t out[4])
{
    if (!out || radixBits > 16) return HJ_ERR_INVALID;
    const uint32_t bits = radixBits ? radixBits : auto_radix_bits(rSize);
    const PrjPlan pl = prj_plan(rSize, sSize, bits);
    out[0] = pl.workspaceBytes; out[1] = pl.histEntries;
    out[2] = prj_hist_entries_needed(rSize, bits); out[3] = prj_hist_entries_needed(sSize, bits);
    return HJ_OK;
}

// ---- streaming Zipf generator ---------------------------------------------------
int hj_zipf_open(hj_ctx* c, uint64_t alphabetSize, double theta, unsigned seed)
{
    if (!c) return HJ_ERR_INVALID;
    if (alphabetSize == 0 || alphabetSize > 0xFFFFFFFFull || !(theta >= 0.0)) return fail(c, HJ_ERR_INVALID, "hj_zipf_open: alphabet in [1, 2^32), theta >= 0");
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    zipf_release(c);
    c->zipfRng = new (std::nothrow) hjhost::GlibcRand(seed);
    if (!c->zipfRng) return HJ_ERR_OOM;
    std::vector<uint32_t> alphabet;
    std::vector<double> lut;
    hjhost::zipf_tables(*c->zipfRng, (uint32_t)alphabetSize, theta, alphabet, lut);       // consumes alphabetSize - 1 draws
    HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->zipfLut), alphabetSize * sizeof(double)));
    HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->zipfAlphabet), alphabetSize * sizeof(uint32_t)));
    HJ_HIP(c, hipMemcpy(c->zipfLut, lut.data(), alphabetSize * sizeof(double), hipMemcpyHostToDevice));
    HJ_HIP(c, hipMemcpy(c->zipfAlphabet, alphabet.data(), alphabetSize * sizeof(uint32_t), hipMemcpyHostToDevice));
    c->zipfAlphabetSize = (uint32_t)alphabetSize;
    for (int i = 0; i < 2; ++i) HJ_HIP(c, hipEventCreateWithFlags(&c->zipfDone[i], hipEventDisableTiming));
    return HJ_OK;
}

int hj_zipf_next_dev(hj_ctx* c, uint64_t n, uint64_t* dOut)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || (!dOut && n)) return HJ_ERR_INVALID;
    if (!c->zipfRng) return fail(c, HJ_ERR_STATE, "hj_zipf_next_dev: hj_zipf_open() first");
    HJ_HIP(c, hipSetDevice(c->device));
    // pieces of at most 2^26 draws through two pinned buffers: the host draws piece k + 1 of the serial rand() stream
    // while the device still copies and searches piece k
    const uint64_t piece = 1ull << 26;
    if (c->zipfRawCap == 0) {
        for (int i = 0; i < 2; ++i) {
            HJ_HIP(c, hipHostMalloc(reinterpret_cast<void**>(&c->zipfRawHost[i]), piece * sizeof(int)));
            HJ_HIP(c, hipMalloc(reinterpret_cast<void**>(&c->zipfRawDev[i]), piece * sizeof(int)));
        }
        c->zipfRawCap = piece;
    }
    for (uint64_t off = 0; off < n; off += piece) {
        const uint64_t m = n - off < piece ? n - off : piece;
        const int b = c->zipfFlip;
        c->zipfFlip ^= 1;
        HJ_HIP(c, hipEventSynchronize(c->zipfDone[b]));          // the previous use of this buffer pair has been consumed
        int* h = c->zipfRawHost[b];
        for (uint64_t i = 0; i < m; ++i) h[i] = c->zipfRng->next();
        HJ_HIP(c, hipMemcpyAsync(c->zipfRawDev[b], h, m * sizeof(int), hipMemcpyHostToDevice, c->stream));
        launch_zipf_lookup(c->zipfRawDev[b], m, c->zipfLut, c->zipfAlphabet, c->zipfAlphabetSize, dOut + off, c->stream);
        HJ_HIP(c, hipGetLastError());
        HJ_HIP(c, hipEventRecord(c->zipfDone[b], c->stream));
    }
    return HJ_OK;
}

int hj_zipf_close(hj_ctx* c)
{
    if (!c) return HJ_ERR_INVALID;
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    zipf_release(c);
    return HJ_OK;
}

// ---- shard helpers -----------------------------------------------------------
int hj_set_shard_check(hj_ctx* c, uint32_t nShards, uint32_t mode, uint32_t shardId)
{
    if (!c) return HJ_ERR_INVALID;
    if (nShards == 0) { c->sc = ShardCheck{0, 0, 0, 0}; return HJ_OK; }
    if (!is_pow2(nShards) || nShards > 64 || shardId >= nShards || (mode & 0xFFu) > 31 || (mode >> 9) != 0)
        return fail(c, HJ_ERR_INVALID, "hj_set_shard_check: nShards a power of two <= 64, shardId < nShards, mode as for hj_shard_histogram_dev");
    c->sc = ShardCheck{nShards - 1, mode & 0xFFu, (mode >> 8) & 1u, shardId};
    return HJ_OK;
}

static int shard_check(hj_ctx* c, const char* who, uint64_t n, uint32_t nShards, uint32_t mode)
{
    if (!is_pow2(nShards) || nShards > 64) return fail(c, HJ_ERR_INVALID, who);
    if ((mode & 0xFFu) > 31 || (mode >> 9) != 0)
        return fail(c, HJ_ERR_INVALID, "shard helpers: mode = digit position (0..31), optionally | HJ_SHARD_ONE_BASED");
    if (n >= 0xFFFFFFFFull) return fail(c, HJ_ERR_INVALID, "shard helpers: n must be < 2^32");
    return HJ_OK;
}

int hj_shard_histogram_dev(hj_ctx* c, const uint64_t* dIn, uint64_t n, uint32_t nShards, uint32_t mode,
                           uint64_t* dCounts)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || (!dIn && n) || !dCounts) return HJ_ERR_INVALID;
    int rc = shard_check(c, "hj_shard_histogram_dev: nShards must be a power of two <= 64", n, nShards, mode);
    if (rc) return rc;
    HJ_HIP(c, hipSetDevice(c->device));
    // reuse the slot of the same input, else the least recently used one
    hj_ctx::ShardPlan* slot = &c->shard[0];
    for (auto& sp : c->shard) if (sp.in == dIn && sp.n == n) { slot = &sp; break; } else if (sp.stamp < slot->stamp) slot = &sp;
    const size_t need = shard_work_bytes(n, nShards);
    if (need > slot->cap) {
        HJ_HIP(c, hipStreamSynchronize(c->stream));
        if (slot->work) { HJ_HIP(c, hipFree(slot->work)); slot->work = nullptr; slot->cap = 0; }
        HJ_HIP(c, hipMalloc(&slot->work, need));
        slot->cap = need;
    }
    slot->in = dIn; slot->n = n; slot->nShards = nShards; slot->mode = mode; slot->stamp = ++c->shardStamp;
    HJ_HIP(c, launch_shard_hist(dIn, n, nShards, mode, slot->work, reinterpret_cast<unsigned long long*>(dCounts), c->stream));
    return HJ_OK;
}

int hj_shard_scatter_dev(hj_ctx* c, const uint64_t* dIn, uint64_t n, uint32_t nShards, uint32_t mode,
                         const uint64_t* dCounts, uint32_t* dOutKeys)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || (!dIn && n) || !dCounts || (!dOutKeys && n)) return HJ_ERR_INVALID;
    int rc = shard_check(c, "hj_shard_scatter_dev: nShards must be a power of two <= 64", n, nShards, mode);
    if (rc) return rc;
    hj_ctx::ShardPlan* slot = nullptr;
    for (auto& sp : c->shard) if (sp.in == dIn && sp.n == n && sp.nShards == nShards && sp.mode == mode && sp.work) slot = &sp;
    if (!slot) return fail(c, HJ_ERR_STATE, "hj_shard_scatter_dev: call hj_shard_histogram_dev on this input (same nShards and mode) first");
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, launch_shard_scatter_ordered(dIn, n, nShards, mode, slot->work, dOutKeys, c->stream));
    slot->in = nullptr;   // consumed
    return HJ_OK;
}

// ---- raw device memory ---------------------------------------------------------
int hj_dev_alloc(hj_ctx* c, uint64_t bytes, void** dptr)
{
    if (!c || !dptr) return HJ_ERR_INVALID;
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipMalloc(dptr, bytes ? bytes : 16));
    return HJ_OK;
}

int hj_dev_free(hj_ctx* c, void* dptr)
{
    if (!c) return HJ_ERR_INVALID;
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    if (dptr) HJ_HIP(c, hipFree(dptr));
    return HJ_OK;
}

int hj_copy_h2d(hj_ctx* c, void* dst_dev, const void* src_host, uint64_t bytes)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || !dst_dev || !src_host) return HJ_ERR_INVALID;
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    return HJ_OK;
}

int hj_copy_d2h(hj_ctx* c, void* dst_host, const void* src_dev, uint64_t bytes)
{
    if (c) c->streamAtBuildEnd = false;
    if (!c || !dst_host || !src_dev) return HJ_ERR_INVALID;
    HJ_HIP(c, hipSetDevice(c->device));
    HJ_HIP(c, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HJ_HIP(c, hipStreamSynchronize(c->stream));
    return HJ_OK;
}

}  // extern "C"

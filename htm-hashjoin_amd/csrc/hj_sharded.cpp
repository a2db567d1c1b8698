// hj_sharded.cpp -- the radix-sharded join of one node behind the C ABI: libhtmjoin_sharded.so.
//
// New design (the reference is single-process shared memory; SURVEY.md 8b asks for "one host thread drives all GPUs
// (or one thread per GPU inside the lib)" and 8e for the exchange): ONE process, one host thread per GPU, RCCL linked
// directly. Rank g = devices[g] holds the g-th contiguous piece of R and of S in its own HBM. Per join, every rank's
// thread runs the steps htm-hashjoin_amd/sharded.py runs per process (that file is the reference for the layout; its
// gloo tests pin the semantics):
//
//   histogram  destination of a tuple = a radix digit of its key (HASH_BIT_MODULO, mc/src/parallel_radix_join.c:59):
//              hj_shard_histogram_dev on R and S
//   counts     the G x G count matrix is plain host memory here (the threads share an address space): a barrier, no
//              collective
//   split      hj_shard_scatter_dev: stable, tuples in, bare 32-bit keys out, grouped by destination
//   exchange   one grouped ncclSend / ncclRecv per relation and rank (every pair directly: all xGMI links at once), on the
//              rank's own stream behind its split -- R's exchange overlaps the split of S on the same stream order, S's
//              the build. The receiver lays the pieces out in source-rank order, so position = global input order
//   local join hj_build_keys_dev (index = position, home slot = (key >> log2 G) & mask under the low-bit split) +
//              hj_probe_keys_dev on the received keys
//   totals     the per-rank counters are added on the host
//
// The per-GPU work is libhtmjoin_hip.so's (include/htm_hashjoin.h); this file is host-side only: plan arithmetic,
// threads, RCCL calls. It is a library of its own so that a single-GPU user never loads RCCL (and so that a Python
// process that has torch's RCCL loaded never sees a second one: the Python path keeps torch.distributed).
#include "../../include/htm_hashjoin.h"
#include "../../include/htm_hashjoin_sharded.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

namespace {

// all threads of a join meet here between the steps (C++17: no std::barrier)
class Barrier {
public:
    explicit Barrier(int n) : n_(n) {}
    void wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        const unsigned long gen = gen_;
        if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen != gen_; });
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    unsigned long gen_ = 0;
};

uint32_t log2u(uint32_t v) { uint32_t l = 0; while ((1u << l) < v) ++l; return l; }
bool is_pow2(uint64_t v) { return v && !(v & (v - 1)); }
uint64_t pow2ceil(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

struct hj_sharded {
    int G = 0;
    std::vector<int> devices;
    std::vector<hj_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    // per rank: split output, receive buffers, count buffers (device), grown on demand
    struct Rank {
        uint32_t *outR = nullptr, *outS = nullptr, *gotR = nullptr, *gotS = nullptr;
        uint64_t capOutR = 0, capOutS = 0, capGotR = 0, capGotS = 0;
        uint64_t* dCnt = nullptr;            // 2 * G counts
        uint64_t reservedTable = 0, reservedS = 0;
    };
    std::vector<Rank> rank;
    std::string err;
    std::mutex errMutex;
};

extern "C" {

// ---- plan arithmetic (host only; tests/test_sharded_abi.py drives it without a GPU) ------------------------------------
int hj_sharded_plan(uint32_t nRanks, const uint64_t* counts, uint64_t* sendOff, uint64_t* recvOff, uint64_t* recvTotal,
                    uint64_t* maxMessage, uint64_t* moved)
{
    if (!nRanks || !counts || !sendOff || !recvOff) return HJ_ERR_INVALID;
    const uint32_t G = nRanks;
    uint64_t mx = 0, mv = 0;
    for (uint32_t g = 0; g < G; ++g) {
        uint64_t so = 0, ro = 0;
        for (uint32_t p = 0; p < G; ++p) {
            sendOff[g * (G + 1) + p] = so; so += counts[g * G + p];          // rank g's split output: grouped by destination p
            recvOff[g * (G + 1) + p] = ro; ro += counts[p * G + g];          // rank g's receive buffer: pieces in SOURCE-rank order
            if (p != g) { mv += counts[g * G + p]; mx = counts[g * G + p] > mx ? counts[g * G + p] : mx; }
        }
        sendOff[g * (G + 1) + G] = so;
        recvOff[g * (G + 1) + G] = ro;
        if (recvTotal) recvTotal[g] = ro;
    }
    if (maxMessage) *maxMessage = mx;
    if (moved) *moved = mv;
    return HJ_OK;
}

uint32_t hj_sharded_mode(uint32_t nRanks, uint32_t split, uint64_t maxKey, uint32_t* homeShift)
{
    // low bits: digit 0, the shard bits are shifted out of the home slot; high bits (range split): the top log2 G bits of
    // (key - 1) over the key domain [1, maxKey], home slot = key & mask. A domain too small for a digit falls back to low.
    const uint32_t gbits = log2u(nRanks);
    uint32_t bits = 0;
    for (uint64_t v = maxKey ? maxKey - 1 : 0; v; v >>= 1) ++bits;
    const uint32_t digit = bits > gbits ? bits - gbits : 0;
    if (split == HJ_SPLIT_HIGH && digit > 0 && nRanks > 1) { if (homeShift) *homeShift = 0; return digit | HJ_SHARD_ONE_BASED; }
    if (homeShift) *homeShift = gbits;
    return 0;
}

int hj_sharded_create(const int* devices, int nDevices, hj_sharded** out)
{
    if (!out) return HJ_ERR_INVALID;
    *out = nullptr;
    if (!devices || nDevices <= 0 || nDevices > 64 || !is_pow2((uint64_t)nDevices)) return HJ_ERR_INVALID;
    hj_sharded* s = new (std::nothrow) hj_sharded();
    if (!s) return HJ_ERR_OOM;
    s->G = nDevices;
    s->devices.assign(devices, devices + nDevices);
    s->ctx.assign(nDevices, nullptr);
    s->stream.assign(nDevices, nullptr);
    s->comm.assign(nDevices, nullptr);
    s->rank.resize(nDevices);
    int rc = HJ_OK;
    for (int g = 0; g < nDevices && rc == HJ_OK; ++g) {
        if (hipSetDevice(devices[g]) != hipSuccess) { rc = HJ_ERR_NO_DEVICE; break; }
        if (hipStreamCreateWithFlags(&s->stream[g], hipStreamNonBlocking) != hipSuccess) { rc = HJ_ERR_HIP; break; }
        rc = hj_create_on_stream(devices[g], s->stream[g], &s->ctx[g]);
        if (rc == HJ_OK && hipMalloc(reinterpret_cast<void**>(&s->rank[g].dCnt), 2 * sizeof(uint64_t) * (size_t)nDevices) != hipSuccess) rc = HJ_ERR_OOM;
    }
    if (rc == HJ_OK && nDevices > 1 && ncclCommInitAll(s->comm.data(), nDevices, devices) != ncclSuccess) rc = HJ_ERR_HIP;
    if (rc != HJ_OK) { hj_sharded_destroy(s); return rc; }
    *out = s;
    return HJ_OK;
}

void hj_sharded_destroy(hj_sharded* s)
{
    if (!s) return;
    for (int g = 0; g < s->G; ++g) {
        hipSetDevice(s->devices[g]);
        if (s->stream[g]) hipStreamSynchronize(s->stream[g]);
        if (s->comm[g]) ncclCommDestroy(s->comm[g]);
        hj_sharded::Rank& r = s->rank[g];
        for (void* p : {(void*)r.outR, (void*)r.outS, (void*)r.gotR, (void*)r.gotS, (void*)r.dCnt}) if (p) hipFree(p);
        if (s->ctx[g]) hj_destroy(s->ctx[g]);
        if (s->stream[g]) hipStreamDestroy(s->stream[g]);
    }
    delete s;
}

const char* hj_sharded_last_error(const hj_sharded* s) { return s ? s->err.c_str() : "null handle"; }

int hj_sharded_ranks(const hj_sharded* s) { return s ? s->G : 0; }

// device memory of one rank, for hosts that never touch HIP themselves (csrc/main.cpp)
int hj_sharded_alloc(hj_sharded* s, int rank, uint64_t bytes, void** dptr)
{
    if (!s || rank < 0 || rank >= s->G || !dptr) return HJ_ERR_INVALID;
    return hj_dev_alloc(s->ctx[rank], bytes, dptr);
}
int hj_sharded_free(hj_sharded* s, int rank, void* dptr)
{
    if (!s || rank < 0 || rank >= s->G) return HJ_ERR_INVALID;
    return hj_dev_free(s->ctx[rank], dptr);
}
int hj_sharded_copy_h2d(hj_sharded* s, int rank, void* dst, const void* src, uint64_t bytes)
{
    if (!s || rank < 0 || rank >= s->G) return HJ_ERR_INVALID;
    return hj_copy_h2d(s->ctx[rank], dst, src, bytes);
}

int hj_sharded_join(hj_sharded* s, const hj_params* params, uint32_t split, uint64_t maxKey, uint64_t tableSize,
                    const uint64_t* const* dR, const uint64_t* nR, const uint64_t* const* dS, const uint64_t* nS,
                    hj_result* total, hj_sharded_stats* stats)
{
    if (!s || !params || !dR || !nR || !total) return HJ_ERR_INVALID;
    if (params->algo != HJ_ALGO_ATOMIC && params->algo != HJ_ALGO_NOCC) return HJ_ERR_UNKNOWN_ALGO;   // the open-addressing operator shards
    if (split > HJ_SPLIT_HIGH) return HJ_ERR_INVALID;
    const int G = s->G;
    uint64_t maxPiece = 0;
    for (int g = 0; g < G; ++g) maxPiece = nR[g] > maxPiece ? nR[g] : maxPiece;
    if (tableSize == 0) tableSize = 2 * pow2ceil(maxPiece ? maxPiece : 1);
    if (!is_pow2(tableSize)) return HJ_ERR_INVALID;
    uint32_t homeShift = 0;
    const uint32_t mode = hj_sharded_mode((uint32_t)G, split, maxKey, &homeShift);

    std::vector<uint64_t> cntR((size_t)G * G, 0), cntS((size_t)G * G, 0);
    std::vector<uint64_t> sOffR((size_t)G * (G + 1)), rOffR((size_t)G * (G + 1)), sOffS((size_t)G * (G + 1)), rOffS((size_t)G * (G + 1));
    std::vector<uint64_t> recvR(G, 0), recvS(G, 0);
    std::vector<hj_result> res(G);
    std::vector<int> status(G, HJ_OK);
    uint64_t maxMsgR = 0, maxMsgS = 0, movedR = 0, movedS = 0;
    Barrier bar(G);
    std::atomic<int> failed{0};

    auto fail = [&](int g, int rc, const char* what) {
        status[g] = rc;
        failed.store(1);
        std::lock_guard<std::mutex> lk(s->errMutex);
        char buf[512];
        snprintf(buf, sizeof buf, "rank %d (device %d): %s: %s", g, s->devices[g], what,
                 rc == HJ_ERR_HIP ? "HIP / RCCL error" : hj_last_error(s->ctx[g]));
        s->err = buf;
    };
    auto grow = [&](uint32_t*& p, uint64_t& cap, uint64_t need) -> bool {
        need += 4;                                                      // 16-byte sweeps may touch a few keys past the end
        if (need <= cap) return true;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&p), need * sizeof(uint32_t)) != hipSuccess) return false;
        cap = need;
        return true;
    };

    auto worker = [&](int g) {
        hj_sharded::Rank& r = s->rank[g];
        hj_ctx* c = s->ctx[g];
        hipStream_t st = s->stream[g];
        const uint64_t nr = nR[g], ns = (dS && nS) ? nS[g] : 0;
        int rc = HJ_OK;
        if (hipSetDevice(s->devices[g]) != hipSuccess) fail(g, HJ_ERR_HIP, "hipSetDevice");
        // 1. histograms -> this rank's row of the count matrices
        if (!failed.load()) {
            if ((rc = hj_shard_histogram_dev(c, dR[g], nr, (uint32_t)G, mode, r.dCnt)) != HJ_OK) fail(g, rc, "histogram of R");
            else if (ns && (rc = hj_shard_histogram_dev(c, dS[g], ns, (uint32_t)G, mode, r.dCnt + G)) != HJ_OK) fail(g, rc, "histogram of S");
            else {
                if (hipMemcpyAsync(&cntR[(size_t)g * G], r.dCnt, sizeof(uint64_t) * G, hipMemcpyDeviceToHost, st) != hipSuccess ||
                    (ns && hipMemcpyAsync(&cntS[(size_t)g * G], r.dCnt + G, sizeof(uint64_t) * G, hipMemcpyDeviceToHost, st) != hipSuccess) ||
                    hipStreamSynchronize(st) != hipSuccess)
                    fail(g, HJ_ERR_HIP, "count read-back");
            }
        }
        bar.wait();
        // 2. the plan: the same arithmetic on every thread's own copy would do; one thread computes, all read
        if (g == 0 && !failed.load()) {
            hj_sharded_plan((uint32_t)G, cntR.data(), sOffR.data(), rOffR.data(), recvR.data(), &maxMsgR, &movedR);
            hj_sharded_plan((uint32_t)G, cntS.data(), sOffS.data(), rOffS.data(), recvS.data(), &maxMsgS, &movedS);
        }
        bar.wait();
        // 3. split + exchange of R, then of S (stream order: R's transfers run while S is being split)
        for (int rel = 0; rel < 2 && !failed.load(); ++rel) {
            const bool isR = rel == 0;
            const uint64_t n = isR ? nr : ns;
            const uint64_t* in = isR ? dR[g] : (dS ? dS[g] : nullptr);
            const std::vector<uint64_t>& cnt = isR ? cntR : cntS;
            const std::vector<uint64_t>& so = isR ? sOffR : sOffS;
            const std::vector<uint64_t>& ro = isR ? rOffR : rOffS;
            uint32_t*& out = isR ? r.outR : r.outS;
            uint32_t*& got = isR ? r.gotR : r.gotS;
            const uint64_t nrecv = isR ? recvR[g] : recvS[g];
            if (!grow(out, isR ? r.capOutR : r.capOutS, n) || !grow(got, isR ? r.capGotR : r.capGotS, nrecv)) { fail(g, HJ_ERR_OOM, "exchange buffers"); break; }
            if (n && (rc = hj_shard_scatter_dev(c, in, n, (uint32_t)G, mode, r.dCnt + (isR ? 0 : G), out)) != HJ_OK) { fail(g, rc, "split"); break; }
            // my own share never leaves the GPU; every other pair exchanges directly, all inside one group
            const uint64_t own = cnt[(size_t)g * G + g];
            if (own && hipMemcpyAsync(got + ro[(size_t)g * (G + 1) + g], out + so[(size_t)g * (G + 1) + g], own * sizeof(uint32_t),
                                      hipMemcpyDeviceToDevice, st) != hipSuccess) { fail(g, HJ_ERR_HIP, "own share"); break; }
            if (G > 1) {
                bool ok = ncclGroupStart() == ncclSuccess;
                for (int off = 1; off < G && ok; ++off) {                       // staggered peer order
                    const int d = (g + off) % G, src = (g - off + G) % G;
                    const uint64_t nsend = cnt[(size_t)g * G + d], nget = cnt[(size_t)src * G + g];
                    if (nsend) ok = ok && ncclSend(out + so[(size_t)g * (G + 1) + d], nsend, ncclUint32, d, s->comm[g], st) == ncclSuccess;
                    if (nget) ok = ok && ncclRecv(got + ro[(size_t)g * (G + 1) + src], nget, ncclUint32, src, s->comm[g], st) == ncclSuccess;
                }
                ok = (ncclGroupEnd() == ncclSuccess) && ok;
                if (!ok) { fail(g, HJ_ERR_HIP, "ncclSend / ncclRecv group"); break; }
            }
        }
        // 4. local join on the received keys
        if (!failed.load()) {
            hj_params p = *params;
            p.algo = HJ_ALGO_ATOMIC;
            uint64_t rsz = tableSize / 2;
            while (rsz + rsz / 8 < recvR[g]) rsz *= 2;                          // hj_reserve keeps 1/8 headroom for uneven shards
            if (r.reservedTable != rsz || r.reservedS < recvS[g]) {
                if ((rc = hj_reserve(c, &p, rsz, recvS[g])) != HJ_OK) fail(g, rc, "hj_reserve");
                else { r.reservedTable = rsz; r.reservedS = recvS[g]; }
            }
            if (!failed.load() && (rc = hj_build_keys_dev(c, r.gotR, recvR[g], homeShift, tableSize)) != HJ_OK) fail(g, rc, "hj_build_keys_dev");
            if (!failed.load() && (rc = hj_probe_keys_dev(c, r.gotS, recvS[g])) != HJ_OK) fail(g, rc, "hj_probe_keys_dev");
            if (!failed.load() && (rc = hj_checksums_dev(c)) != HJ_OK) fail(g, rc, "hj_checksums_dev");
            if (!failed.load() && (rc = hj_fetch_result(c, &res[g])) != HJ_OK) fail(g, rc, "hj_fetch_result");
        }
        // a rank that failed before its transfers were enqueued leaves its peers' receives pending: nothing to wait for
        // here, the communicator is torn down by the caller (hj_sharded_destroy) after an error
        bar.wait();
    };

    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(worker, g);
    worker(0);
    for (auto& t : th) t.join();
    for (int g = 0; g < G; ++g) if (status[g] != HJ_OK) return status[g];

    memset(total, 0, sizeof(*total));
    for (int g = 0; g < G; ++g) {
        const hj_result& r = res[g];
        total->rSize += nR[g]; total->sSize += (dS && nS) ? nS[g] : 0;
        total->conflicts += r.conflicts; total->totalMatches += r.totalMatches; total->inputSum += r.inputSum;
        total->tableSumHalf += r.tableSumHalf; total->tableSumFull += r.tableSumFull; total->conflictSum += r.conflictSum;
        total->buildDeferred += r.buildDeferred; total->foreignTuples += r.foreignTuples;
        total->build_us = r.build_us > total->build_us ? r.build_us : total->build_us;          // slowest rank
        total->probe_us = r.probe_us > total->probe_us ? r.probe_us : total->probe_us;
        total->compactFallback |= r.compactFallback;
    }
    total->tableSize = tableSize * (uint64_t)G;
    total->outputSum = (params->algo == HJ_ALGO_NOCC ? total->tableSumHalf : total->tableSumFull) + total->conflictSum;
    total->buildVariant = res[0].buildVariant;
    total->algoUsed = HJ_ALGO_ATOMIC;
    total->total_us = total->build_us + total->probe_us;
    if (stats) {
        stats->nRanks = (uint32_t)G; stats->mode = mode; stats->homeShift = homeShift;
        stats->keysMovedR = movedR; stats->keysMovedS = movedS;
        stats->maxMessageKeys = maxMsgR > maxMsgS ? maxMsgR : maxMsgS;
        stats->tableSizePerRank = tableSize;
    }
    return HJ_OK;
}

}  // extern "C"

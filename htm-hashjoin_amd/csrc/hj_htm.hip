// hj_htm.hip -- the bucketised table of `--algo htm` (HTMHashBuild.hpp:41-45, 54-464) on gfx950.
//
// The reference's HTM operator hashes a tuple to bucket (key / 3) & (numBuckets - 1), numBuckets =
// nextpow2(rSize / 3 + 1) (:61-62, :176); a Bucket is 32 bytes: three tuples, a count and the 1-based index of an
// overflow chain (:41-45). Groups of inserts run inside Intel TSX transactions; a tuple that finds its bucket full is
// a "conflict" (:181-183) and is chained to overflow buckets afterwards (:231-279); the probe walks bucket + chain
// (:291-305). TSX is replaced outright, not emulated:
//
//   layout   a bucket = 4 consecutive 8-byte slots = one 32-byte HBM sector: slots 0..2 hold (index << 32 | key),
//            all-ones = empty; slot 3 = (next << 32 | count) for buckets with an overflow chain, all-ones otherwise
//            (count = tuple slots in use, no chain). Overflow buckets use the same format in a second array (index 0
//            unused, as in the reference). Buckets outside the slot range the build defined are never written.
//   build    the three tuple slots are filled with the index-priority protocol of the open-addressing table
//            (hj_kernels.hip) with a probe budget of 3 and home slot = the bucket's first slot (hj_device.h,
//            home_slot_htm): whatever the scheduling, a bucket ends up with its three lowest-indexed tuples in
//            index order -- what a single thread walking R in input order stores (:177-179) -- and the tuples
//            that run out of budget are exactly that thread's conflicts. Fast path: k_build_wave<HTM>
//            (hj_build_wave.hip, LDS rings); without locality: k_htm_build_global below. Either way the
//            conflicts are collected as (index << 32 | key).
//   chains   per bucket the conflicts are counted (k_htm_count), ceil(count / 3) overflow buckets are reserved by
//            one exclusive scan (a bucket's overflow buckets are neighbours), the conflicts are inserted into
//            that region with the same priority protocol (sorted by index = the order the reference chains
//            them in, :234-236), and k_htm_link writes counts and links: the chain's head is the NEWEST
//            overflow bucket (:254-258), each links to the one created before it. Physical overflow indices
//            differ from the reference's (it numbers overflow buckets in global conflict order); bucket
//            contents and every chain's walk order are the same. The two lines of the reference's chain
//            builder that cannot mean what they say (:237 hashes relR[partition index], :249 assigns through
//            the reference into the full head bucket) are taken as intended.
//   probe    k_htm_probe: one 32-byte bucket read per S tuple, then its chain; every R tuple is stored exactly
//            once (bucket or chain), so totalMatches is the TRUE join cardinality on this path.
//
// All integer work, HBM bound; no MFMA.

#include "hj_device.h"

namespace hj {

constexpr uint32_t kHtmProbeLen = 3;                 // tuples per bucket (Bucket::tuples[3])

__device__ __forceinline__ unsigned long long htm_wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---- build without locality: global atomics, one contiguous piece of R per workgroup (conflicts go to ITS slice) ----
__global__ void __launch_bounds__(kBlock)
k_htm_build_global(const uint64_t* __restrict__ R, uint64_t n, uint32_t sliceLen, uint64_t* __restrict__ table,
                   uint64_t mask, uint64_t idxBase, uint64_t* __restrict__ conflicts, uint32_t* __restrict__ ccounts,
                   Counters* __restrict__ ctr)
{
    __shared__ unsigned int sCount;
    if (threadIdx.x == 0) sCount = 0;
    __syncthreads();
    const uint64_t b = (uint64_t)blockIdx.x * sliceLen, e = b + sliceLen < n ? b + sliceLen : n;
    unsigned long long drops = 0, dropSum = 0, inSum = 0, bad = 0;
    for (uint64_t i = b + threadIdx.x; i < e; i += kBlock) {
        const uint64_t t = R[i];
        inSum += t;
        if ((t >> 32) != 0 || t == 0) { bad += 1; continue; }
        uint64_t mine = ((idxBase + i) << 32) | t;
        uint64_t pos = home_slot_htm((uint32_t)t, mask);
        for (uint32_t budget = kHtmProbeLen;; ++pos, --budget) {
            if (budget == 0) {                                   // bucket full: HTMHashBuild.hpp:181-183
                drops += 1; dropSum += (uint32_t)mine;
                conflicts[b + atomicAdd(&sCount, 1u)] = mine;
                break;
            }
            const unsigned long long old = atomicMin(reinterpret_cast<unsigned long long*>(table + pos), (unsigned long long)mine);
            if (old == kEmpty || old == mine) break;
            if (old > mine) mine = old;                          // a later tuple sat here: it moves on instead (same home, same budget left)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) ccounts[blockIdx.x] = sCount;
    drops = htm_wave_sum(drops); dropSum = htm_wave_sum(dropSum); inSum = htm_wave_sum(inSum); bad = htm_wave_sum(bad);
    if ((threadIdx.x & 63) == 0) {
        Counters::Shard* const sh = counter_shard(ctr);
        if (drops) atomicAdd(&sh->conflicts, drops);
        if (dropSum) atomicAdd(&sh->conflictSum, dropSum);
        if (inSum) atomicAdd(&sh->inputSum, inSum);
        if (bad) atomicAdd(&sh->badKeys, bad);
    }
}

// ---- chains ---------------------------------------------------------------------------------------------------------
// conflicts of slice c: conflicts[c * sliceLen .. + ccounts[c])
__global__ void __launch_bounds__(kBlock)
k_htm_count(const uint64_t* __restrict__ conflicts, const uint32_t* __restrict__ ccounts, uint32_t nSlices, uint32_t sliceLen,
            uint32_t bucketMask, unsigned int* __restrict__ ovfCount)
{
    for (uint32_t c = blockIdx.x; c < nSlices; c += gridDim.x) {
        const uint32_t cnt = ccounts[c];
        const uint64_t* q = conflicts + (uint64_t)c * sliceLen;
        for (uint32_t i = threadIdx.x; i < cnt; i += kBlock)
            atomicAdd(&ovfCount[((uint32_t)q[i] / 3u) & bucketMask], 1u);
    }
}

// groups[b] = overflow buckets bucket b needs = ceil(conflicts of b / 3); scanned in place afterwards
__global__ void __launch_bounds__(kBlock)
k_htm_groups(const unsigned int* __restrict__ ovfCount, uint32_t numBuckets, uint32_t* __restrict__ groups)
{
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < numBuckets; b += (uint64_t)gridDim.x * kBlock)
        groups[b] = (ovfCount[b] + 2u) / 3u;
}

// tuple slot d (0-based, in index order) of the overflow region that starts at overflow bucket `first` (1-based)
__device__ __forceinline__ uint64_t htm_ovf_slot(uint32_t first, uint32_t d) { return ((uint64_t)(first + d / 3u) << 2) + d % 3u; }

__global__ void __launch_bounds__(kBlock)
k_htm_fill_overflow(const uint64_t* __restrict__ conflicts, const uint32_t* __restrict__ ccounts, uint32_t nSlices, uint32_t sliceLen,
                    uint32_t bucketMask, const unsigned int* __restrict__ ovfCount, const uint32_t* __restrict__ ovfBase,
                    uint64_t* __restrict__ overflow)
{
    for (uint32_t c = blockIdx.x; c < nSlices; c += gridDim.x) {
        const uint32_t cnt = ccounts[c];
        const uint64_t* q = conflicts + (uint64_t)c * sliceLen;
        for (uint32_t i = threadIdx.x; i < cnt; i += kBlock) {
            uint64_t mine = q[i];
            const uint32_t b = ((uint32_t)mine / 3u) & bucketMask;
            const uint32_t first = ovfBase[b] + 1u, slots = ovfCount[b];       // exactly as many slots as conflicts
            for (uint32_t d = 0; d < slots; ++d) {
                const unsigned long long old =
                    atomicMin(reinterpret_cast<unsigned long long*>(overflow + htm_ovf_slot(first, d)), (unsigned long long)mine);
                if (old == kEmpty) break;
                if (old > mine) mine = old;
            }
        }
    }
}

// Links: slot 3 = (next << 32 | count) of every primary bucket THAT HAS CONFLICTS and of its overflow buckets. Every other
// bucket keeps the all-ones word the build left there, which readers take as "no chain, count = tuple slots in use"
// (htm_meta below) -- so a build without conflicts needs no pass over the table at all, and one with conflicts reads
// 4 bytes per bucket here instead of rewriting 32.
__global__ void __launch_bounds__(kBlock)
k_htm_link(uint64_t* __restrict__ table, uint32_t numBuckets, const unsigned int* __restrict__ ovfCount,
           const uint32_t* __restrict__ ovfBase, uint64_t* __restrict__ overflow, Counters* __restrict__ ctr)
{
    unsigned long long groupsSeen = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < numBuckets; b += (uint64_t)gridDim.x * kBlock) {
        const uint32_t oc = ovfCount[b];
        if (oc == 0) continue;
        const uint32_t g = (oc + 2u) / 3u, first = ovfBase[b] + 1u;
        table[(b << 2) + 3] = ((uint64_t)(first + g - 1u) << 32) | 3u;                       // full bucket; head = the newest overflow bucket
        for (uint32_t j = 0; j < g; ++j)
            overflow[((uint64_t)(first + j) << 2) + 3] = ((uint64_t)(j ? first + j - 1u : 0u) << 32) | (j + 1 < g ? 3u : oc - 3u * (g - 1u));
        groupsSeen += g;
    }
    groupsSeen = htm_wave_sum(groupsSeen);
    if ((threadIdx.x & 63) == 0 && groupsSeen) atomicAdd(&ctr->htmOverflowBuckets, groupsSeen);
}

// ---- chains along the rings: the whole phase in LDS (round 3) ----------------------------------------------------------------
// The generic kernels above are bound by memory-side atomics (k_htm_count one per conflict, k_htm_fill_overflow a priority
// walk per conflict, 30-45 G atomics/s: 2.6 ms for the 30 M conflicts of `uniform` at 2^27) and by three passes over
// per-bucket arrays of the whole table (groups, scan, link: 1.2 ms). After the ring build (k_build_wave<HTM>) none of that is
// needed: chunk c's wavefront only ever inserts into the slot range it owns, [bounds[c], bounds[c + 1]) granules = 32
// buckets each, so the conflicts it lists are conflicts of ITS buckets, and the deferred phase files the few it finds
// under the chunk that owns their bucket (k_wave_deferred<HTM>, route). A slice of the conflict list then holds ALL
// conflicts of a contiguous bucket range, and a workgroup can do for that range in LDS what the generic kernels do in HBM:
//   count      (k_htm_chain_count, one workgroup per slice) the bucket range the slice's conflicts really span, one LDS
//              counter per bucket of it (direct index), the range cut into `parts` equal sub-ranges; per part: overflow
//              buckets needed (sum of ceil(count / 3)) -> partGroups, and the stretch of the list its conflicts lie in
//              (near-sorted keys: about a parts-th of the slice) -> info
//   (host)     exclusive scan of partGroups: where each part's overflow buckets start (a part's buckets are neighbours,
//              ordered by bucket -- as with the generic scan over buckets, only the numbering differs)
//   fill       (k_htm_chain_fill, one workgroup per part) counters of the sub-range again, exclusive scan of the groups,
//              the conflicts inserted by index priority into an LDS image of the part's overflow buckets (LDS atomicMin:
//              the same walk as k_htm_fill_overflow), and the image written out whole: 32 bytes per overflow bucket, tuple
//              slots and link word at once, plus the link word of each primary bucket with conflicts. No global atomics,
//              no per-bucket arrays, no memset of the overflow area.
// Anything that does not fit (a slice spanning more than kChainCountCap buckets: sparse keys; a part of more than kChainCap
// buckets or image slots: many duplicates; a conflict outside its slice's range) raises Counters::htmChainBail, and the
// host redoes the build without routing and chains with the generic kernels.
constexpr int kChainThreads = 512;
constexpr uint32_t kChainCountCap = 12288;           // buckets a slice's conflicts may span (k_htm_chain_count's LDS counters)
constexpr uint32_t kChainCap = 3072;                 // buckets per part = the fill's LDS counters; tuple slots of its overflow image (36 KiB: 4 workgroups per CU)
constexpr uint32_t kChainMaxParts = 16;

__device__ __forceinline__ uint32_t chain_block_exclusive_scan(uint32_t v, uint32_t* wsum /*[kChainThreads / 64]*/, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = (uint32_t)__shfl_up((int)inc, off, 64);
        if ((int)lane >= off) inc += n;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < kChainThreads / 64; ++k) { const uint32_t x = wsum[k]; if (k < (int)w) wbase += x; tot += x; }
    total = tot;
    __syncthreads();
    return wbase + inc - v;
}

// info: per slice 2 words (first bucket of the span, buckets per part), then per part 2 words (its stretch of the list)
__global__ void __launch_bounds__(kChainThreads)
k_htm_chain_count(const uint64_t* __restrict__ conflicts, const uint32_t* __restrict__ ccounts, const uint32_t* __restrict__ bounds,
                  uint32_t nSlices, uint32_t sliceLen, uint32_t parts, uint32_t numBuckets, uint32_t* __restrict__ partGroups,
                  uint32_t* __restrict__ info, Counters* __restrict__ ctr)
{
    __shared__ uint32_t cnt[kChainCountCap];
    __shared__ uint32_t sMin, sMaxInv, sStray, sBail;
    __shared__ uint32_t pGroups[kChainMaxParts], pFirst[kChainMaxParts], pLastInv[kChainMaxParts];
    const uint32_t c = blockIdx.x;
    if (threadIdx.x == 0) {
        sBail = *reinterpret_cast<volatile unsigned long long*>(&ctr->htmChainBail) != 0;
        sMin = 0xFFFFFFFFu; sMaxInv = 0xFFFFFFFFu; sStray = 0;
    }
    if (threadIdx.x < kChainMaxParts) { pGroups[threadIdx.x] = 0; pFirst[threadIdx.x] = 0xFFFFFFFFu; pLastInv[threadIdx.x] = 0xFFFFFFFFu; }
    __syncthreads();
    if (sBail) return;
    const uint32_t m = ccounts[c];
    uint32_t* const sInfo = info + 2 * (size_t)c;
    uint32_t* const pInfo = info + 2 * (size_t)nSlices + 2 * (size_t)c * parts;
    auto bail = [&]() { if (threadIdx.x == 0) atomicExch(&ctr->htmChainBail, 1ull); };
    if (m > sliceLen) { bail(); return; }
    if (m == 0) {
        if (threadIdx.x == 0) { sInfo[0] = 0; sInfo[1] = 0; }
        if (threadIdx.x < parts) { partGroups[c * parts + threadIdx.x] = 0; pInfo[2 * threadIdx.x] = 0; pInfo[2 * threadIdx.x + 1] = 0; }
        return;
    }
    const uint64_t* __restrict__ q = conflicts + (uint64_t)c * sliceLen;
    const uint32_t bucketMask = numBuckets - 1;
    // what the chunk owns (first and last chunk: also what lies outside every chunk's range), and what its conflicts span
    const uint32_t B0 = c ? bounds[c] * 32u : 0u, B1 = (c + 1 == nSlices) ? numBuckets : bounds[c + 1] * 32u;
    uint32_t lo = 0xFFFFFFFFu, hiInv = 0xFFFFFFFFu;
    bool stray = false;
    for (uint32_t i = threadIdx.x; i < m; i += kChainThreads) {
        const uint32_t b = ((uint32_t)q[i] / 3u) & bucketMask;
        stray |= (b < B0) | (b >= B1);
        lo = b < lo ? b : lo; hiInv = ~b < hiInv ? ~b : hiInv;
    }
    atomicMin(&sMin, lo); atomicMin(&sMaxInv, hiInv);
    if (stray) sStray = 1;
    __syncthreads();
    const uint32_t E0 = sMin, span = ~sMaxInv - E0 + 1u;
    const uint32_t sub = (span + parts - 1) / parts;                           // buckets per part
    if (sStray || span > kChainCountCap || sub > kChainCap) { bail(); return; }   // (a conflict filed under the wrong chunk: not this path)
    for (uint32_t i = threadIdx.x; i < span; i += kChainThreads) cnt[i] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += kChainThreads) {
        const uint32_t b = (((uint32_t)q[i] / 3u) & bucketMask) - E0;
        atomicAdd(&cnt[b], 1u);
        // the part's stretch of the list: near-sorted keys put a whole wavefront into one part -- one lane speaks for it
        // (64 lanes on one LDS address are served one after the other)
        const uint32_t p = b / sub;
        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
        const unsigned long long act = __ballot(1), same = __ballot(p == p0);
        if (same == act) {
            const uint32_t first = (uint32_t)__ffsll((long long)act) - 1u, last = 63u - (uint32_t)__clzll((long long)act);
            const uint32_t lane = threadIdx.x & 63u;
            if (lane == first) atomicMin(&pFirst[p0], i);
            if (lane == last) atomicMin(&pLastInv[p0], ~i);
        } else {
            atomicMin(&pFirst[p], i); atomicMin(&pLastInv[p], ~i);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < span; i += kChainThreads) {
        const uint32_t g = (cnt[i] + 2u) / 3u;
        if (g) atomicAdd(&pGroups[i / sub], g);
    }
    __syncthreads();
    if (threadIdx.x == 0) { sInfo[0] = E0; sInfo[1] = sub; }
    if (threadIdx.x < parts) {
        const uint32_t g = pGroups[threadIdx.x];
        if (3u * g > kChainCap) atomicExch(&ctr->htmChainBail, 1ull);
        partGroups[c * parts + threadIdx.x] = g;
        pInfo[2 * threadIdx.x] = g ? pFirst[threadIdx.x] : 0u;
        pInfo[2 * threadIdx.x + 1] = g ? ~pLastInv[threadIdx.x] + 1u : 0u;
    }
}

__global__ void __launch_bounds__(kChainThreads)
k_htm_chain_fill(const uint64_t* __restrict__ conflicts, uint32_t nSlices, uint32_t sliceLen, uint32_t parts, uint32_t numBuckets,
                 const uint32_t* __restrict__ partBase, const uint32_t* __restrict__ info, uint64_t* __restrict__ table,
                 uint64_t* __restrict__ overflow, Counters* __restrict__ ctr)
{
    __shared__ uint32_t cnt[kChainCap];              // per bucket of the part: conflicts | first group of the bucket << 16
    __shared__ unsigned long long image[kChainCap];  // the part's overflow buckets, three tuple slots each
    __shared__ uint32_t wsum[kChainThreads / 64];
    const uint32_t c = blockIdx.x / parts, p = blockIdx.x - c * parts;
    const uint32_t base = partBase[blockIdx.x], total = partBase[blockIdx.x + 1] - base;   // (one word more than parts: the grand total)
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) ctr->htmOverflowBuckets = (unsigned long long)base + total;
    if (total == 0) return;
    const uint32_t E0 = info[2 * (size_t)c], sub = info[2 * (size_t)c + 1];
    const uint32_t i0 = info[2 * (size_t)nSlices + 2 * (size_t)blockIdx.x], i1 = info[2 * (size_t)nSlices + 2 * (size_t)blockIdx.x + 1];
    const uint64_t* __restrict__ q = conflicts + (uint64_t)c * sliceLen;
    const uint32_t bucketMask = numBuckets - 1, lo = E0 + p * sub;
    for (uint32_t i = threadIdx.x; i < sub; i += kChainThreads) cnt[i] = 0;
    for (uint32_t i = threadIdx.x; i < 3u * total; i += kChainThreads) image[i] = kEmpty;
    __syncthreads();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += kChainThreads) {
        const uint32_t b = (((uint32_t)q[i] / 3u) & bucketMask) - lo;
        if (b < sub) atomicAdd(&cnt[b], 1u);
    }
    __syncthreads();
    // first group of every bucket: exclusive scan of ceil(count / 3) over the part, thread t takes `per` consecutive buckets
    const uint32_t per = (sub + kChainThreads - 1) / kChainThreads;
    const uint32_t f0 = threadIdx.x * per < sub ? threadIdx.x * per : sub, f1 = f0 + per < sub ? f0 + per : sub;
    uint32_t mine = 0;
    for (uint32_t i = f0; i < f1; ++i) mine += (cnt[i] + 2u) / 3u;
    uint32_t tot;
    uint32_t run = chain_block_exclusive_scan(mine, wsum, tot);
    for (uint32_t i = f0; i < f1; ++i) { const uint32_t k = cnt[i]; cnt[i] = k | (run << 16); run += (k + 2u) / 3u; }
    __syncthreads();
    for (uint32_t i = i0 + threadIdx.x; i < i1; i += kChainThreads) {
        unsigned long long x = q[i];
        const uint32_t b = (((uint32_t)x / 3u) & bucketMask) - lo;
        if (b >= sub) continue;
        const uint32_t w = cnt[b], k = w & 0xFFFFu, at = 3u * (w >> 16);
        for (uint32_t d = 0; d < k; ++d) {                                    // index priority, as k_htm_fill_overflow
            const unsigned long long old = atomicMin(&image[at + d], x);
            if (old == kEmpty) break;
            if (old > x) x = old;
        }
    }
    __syncthreads();
    for (uint32_t G = threadIdx.x; G < total; G += kChainThreads) {
        const unsigned long long t0 = image[3u * G], t1 = image[3u * G + 1], t2 = image[3u * G + 2];
        const uint32_t b = ((uint32_t)t0 / 3u) & bucketMask;                  // a group's first slot is never empty
        const uint32_t w = cnt[b - lo], k = w & 0xFFFFu, gb = w >> 16, g = (k + 2u) / 3u, j = G - gb;
        const uint64_t id = 1ull + base + G;                                  // overflow buckets are numbered from 1
        const unsigned long long link = ((unsigned long long)(j ? id - 1 : 0ull) << 32) | (j + 1 < g ? 3u : k - 3u * (g - 1u));
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(overflow + (id << 2));
        dst[0] = make_ulonglong2(t0, t1);
        dst[1] = make_ulonglong2(t2, link);
        if (j == 0) table[((uint64_t)b << 2) + 3] = ((unsigned long long)(id + g - 1u) << 32) | 3u;   // head = the newest overflow bucket
    }
}

// (count, next) of a bucket whose four words are a, b, c (tuple slots) and m (slot 3)
__device__ __forceinline__ uint32_t htm_next(uint64_t m) { return m == kEmpty ? 0u : (uint32_t)(m >> 32); }

// ---- probe: bucket, then its chain (HTMHashBuild.hpp:291-305) -----------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_htm_probe(const uint64_t* __restrict__ S, uint64_t n, const uint64_t* __restrict__ table, uint32_t bucketMask,
            const uint64_t* __restrict__ overflow, Counters* __restrict__ ctr)
{
    unsigned long long matches = 0;
    // buckets outside the slots the build defined were never written and hold no tuple (hj_device.h, Counters)
    const uint64_t defLo = ctr->validLo, defHi = ctr->validHiEx + 512;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const uint64_t s = S[i];
        if ((s >> 32) != 0 || s == 0) continue;                  // cannot equal any stored tuple
        const uint32_t key = (uint32_t)s;
        const uint64_t slot = (uint64_t)((key / 3u) & bucketMask) << 2;
        if (slot < defLo || slot + 3 >= defHi) continue;
        const ulonglong2* p = reinterpret_cast<const ulonglong2*>(table + slot);
        for (;;) {
            const ulonglong2 a = p[0], c = p[1];
            matches += (a.x != kEmpty && (uint32_t)a.x == key) + (a.y != kEmpty && (uint32_t)a.y == key) +
                       (c.x != kEmpty && (uint32_t)c.x == key);
            const uint32_t next = htm_next(c.y);
            if (next == 0) break;
            p = reinterpret_cast<const ulonglong2*>(overflow + ((uint64_t)next << 2));
        }
    }
    matches = htm_wave_sum(matches);
    if ((threadIdx.x & 63) == 0 && matches) atomicAdd(&counter_shard(ctr)->matches, matches);
}

// ---- checksums (:322-342): tuples in primary buckets / in overflow buckets ------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_htm_sums(const uint64_t* __restrict__ table, uint32_t numBuckets, const uint64_t* __restrict__ overflow, Counters* __restrict__ ctr)
{
    unsigned long long prim = 0, ovf = 0;
    const uint64_t defLo = ctr->validLo >> 2;
    uint64_t defHi = (ctr->validHiEx + 512) >> 2;
    defHi = defHi < numBuckets ? defHi : numBuckets;
    const uint64_t nPrim = defHi > defLo ? defHi - defLo : 0;
    const uint64_t total = nPrim + ctr->htmOverflowBuckets;                      // the overflow buckets k_htm_link linked
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (uint64_t)gridDim.x * kBlock) {
        const bool isOvf = i >= nPrim;
        const uint64_t* p = isOvf ? overflow + ((i - nPrim + 1) << 2) : table + ((defLo + i) << 2);
        unsigned long long sum = 0;
        for (uint32_t j = 0; j < 3; ++j) sum += p[j] == kEmpty ? 0u : (uint32_t)p[j];
        if (isOvf) ovf += sum; else prim += sum;
    }
    prim = htm_wave_sum(prim); ovf = htm_wave_sum(ovf);
    if ((threadIdx.x & 63) == 0) {
        if (prim) atomicAdd(&ctr->tableSumFull, prim);
        if (ovf) atomicAdd(&ctr->htmOverflowSum, ovf);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
uint32_t htm_num_buckets(uint64_t rSize)
{
    uint32_t v = (uint32_t)(rSize / 3 + 1);                      // HTMHashBuild.hpp:61
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;   // NEXT_POW_2, :29-38
    return v;
}

hipError_t launch_htm_build_global(const uint64_t* R, uint64_t n, uint32_t sliceLen, uint32_t nSlices, uint64_t* table,
                                   uint64_t tableSlots, uint64_t idxBase, uint64_t* conflicts, uint32_t* ccounts, Counters* ctr,
                                   hipStream_t s)
{
    hipLaunchKernelGGL(k_htm_build_global, dim3(nSlices), dim3(kBlock), 0, s, R, n, sliceLen, table, tableSlots - 1, idxBase,
                       conflicts, ccounts, ctr);
    return hipGetLastError();
}

hipError_t launch_htm_count(const uint64_t* conflicts, const uint32_t* ccounts, uint32_t nSlices, uint32_t sliceLen,
                            uint32_t numBuckets, unsigned int* ovfCount, uint32_t* groups, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(ovfCount, 0, (size_t)numBuckets * sizeof(unsigned int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_htm_count, dim3(nSlices < 4096 ? nSlices : 4096), dim3(kBlock), 0, s, conflicts, ccounts, nSlices, sliceLen,
                       numBuckets - 1, ovfCount);
    hipLaunchKernelGGL(k_htm_groups, dim3(2048), dim3(kBlock), 0, s, ovfCount, numBuckets, groups);
    return hipGetLastError();
}

hipError_t launch_htm_chains(const uint64_t* conflicts, const uint32_t* ccounts, uint32_t nSlices, uint32_t sliceLen,
                             uint64_t* table, uint32_t numBuckets, const unsigned int* ovfCount, const uint32_t* ovfBase,
                             uint64_t* overflow, uint64_t overflowCapBuckets, Counters* ctr, hipStream_t s)
{
    // all-ones = empty tuple slots; the link words are written by k_htm_link
    hipError_t e = hipMemsetAsync(overflow, 0xFF, (size_t)(overflowCapBuckets + 1) * 4 * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_htm_fill_overflow, dim3(nSlices < 4096 ? nSlices : 4096), dim3(kBlock), 0, s, conflicts, ccounts, nSlices,
                       sliceLen, numBuckets - 1, ovfCount, ovfBase, overflow);
    hipLaunchKernelGGL(k_htm_link, dim3(2048), dim3(kBlock), 0, s, table, numBuckets, ovfCount, ovfBase, overflow, ctr);
    return hipGetLastError();
}

uint32_t htm_chain_parts(uint32_t sliceLen)
{
    // ~4096 tuples' worth of keys per part: 1.4 k buckets on dense keys, a 1.5 k-slot image on `uniform` (half the LDS arrays)
    const uint32_t p = (sliceLen + 4223u) / 4224u;
    return p < 1 ? 1 : p > kChainMaxParts ? kChainMaxParts : p;
}
size_t htm_chain_info_words(uint32_t nSlices, uint32_t sliceLen) { return 2 * (size_t)nSlices * (1 + htm_chain_parts(sliceLen)); }

hipError_t launch_htm_chain_count(const uint64_t* conflicts, const uint32_t* ccounts, const uint32_t* bounds, uint32_t nSlices,
                                  uint32_t sliceLen, uint32_t numBuckets, uint32_t* partGroups, uint32_t* info, Counters* ctr, hipStream_t s)
{
    hipLaunchKernelGGL(k_htm_chain_count, dim3(nSlices), dim3(kChainThreads), 0, s, conflicts, ccounts, bounds, nSlices, sliceLen,
                       htm_chain_parts(sliceLen), numBuckets, partGroups, info, ctr);
    return hipGetLastError();
}

hipError_t launch_htm_chain_fill(const uint64_t* conflicts, uint32_t nSlices, uint32_t sliceLen, uint32_t numBuckets, const uint32_t* partBase,
                                 const uint32_t* info, uint64_t* table, uint64_t* overflow, Counters* ctr, hipStream_t s)
{
    const uint32_t parts = htm_chain_parts(sliceLen);
    hipLaunchKernelGGL(k_htm_chain_fill, dim3(nSlices * parts), dim3(kChainThreads), 0, s, conflicts, nSlices, sliceLen, parts, numBuckets,
                       partBase, info, table, overflow, ctr);
    return hipGetLastError();
}

void launch_htm_probe(const uint64_t* S, uint64_t n, const uint64_t* table, uint32_t numBuckets, const uint64_t* overflow,
                      Counters* ctr, hipStream_t s)
{
    uint64_t blocks = (n + kBlock - 1) / kBlock;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_htm_probe, dim3((unsigned)blocks), dim3(kBlock), 0, s, S, n, table, numBuckets - 1, overflow, ctr);
}

void launch_htm_sums(const uint64_t* table, uint32_t numBuckets, const uint64_t* overflow, Counters* ctr, hipStream_t s)
{
    hipLaunchKernelGGL(k_htm_sums, dim3(2048), dim3(kBlock), 0, s, table, numBuckets, overflow, ctr);
}

}  // namespace hj

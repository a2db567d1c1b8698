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

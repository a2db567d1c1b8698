// hj_build_own.hip -- the locality-exploiting build: slot-block ownership + LDS window.
//
// Why: on MI355X a returning 64-bit global atomic is one memory-side request per
// lane (measured with k_build_atomic_min: 17 G atomics/s, 0.28 TB/s of
// algorithmic bytes, 14x slower than the clear and probe kernels beside it). The
// reference's whole point (README.md:6, figs/perf.png) is that join inputs often
// have locality and that a build which exploits it beats partitioning; its
// HTM path does so with Intel TSX transactions over groups of inserts
// (HTMHashBuild.hpp:157-215), aborting and retrying serially on conflict
// (:219-238). This kernel is the MI355X counterpart, TSX replaced outright:
//
//   phase A  k_build_own: the input is cut into contiguous chunks (like the
//            reference's numPartitions chunks, NoCCHashBuild.hpp:37). A workgroup
//            walks its chunk tile by tile and keeps a sliding window of the table
//            (kWinSlots slots, 64 KiB) in LDS. The table is divided into blocks of
//            kBlkSlots slots; a workgroup may only touch blocks it OWNS, and it
//            claims a block (one atomicCAS on a small owner table) when the block
//            enters its window. Inserts into owned blocks run the index-priority
//            protocol of hj_kernels.hip on LDS (ds_min_rtn_u64). When the window
//            slides, finished blocks go to HBM as whole 4 KiB runs of plain
//            16-byte stores. A tuple whose probe walk reaches a block that is not
//            owned (lost claim at a chunk seam, key far from the window, spill
//            past the window end) is "aborted": it is appended, with the slot it
//            had reached, to a deferred queue.
//   clear    k_clear_unowned: blocks nobody claimed are filled with the empty
//            pattern (owned blocks were written whole in phase A, so the table is
//            written exactly once).
//   phase B  k_build_deferred: the deferred tuples finish their probe walk with
//            the global atomicMin protocol (the "serial retry" of the reference,
//            except that it is parallel and order independent).
//
// The index-priority protocol is confluent (see hj_kernels.hip): any interleaving
// of "insert tuple (idx,key) from slot p with its remaining budget" operations
// reaches the same final table, the one sequential insertion in input order
// produces. Phase A applies a subset of the operations on disjoint, exclusively
// owned blocks; phase B applies the rest. Which workgroup wins a claim changes
// only how many tuples are deferred, never the result.

#include "hj_device.h"

namespace hj {

constexpr int kOwnThreads = 512;                 // 8 wavefronts
constexpr int kOwnVecPerThread = 2;              // 16-byte loads per thread per tile
constexpr int kOwnTile = kOwnThreads * kOwnVecPerThread * 2;   // 2048 tuples
constexpr uint32_t kBlkShift = 9;
constexpr uint32_t kBlkSlots = 1u << kBlkShift;  // 512 slots = 4 KiB
constexpr uint32_t kWinBlocks = 16;
constexpr uint32_t kWinSlots = kBlkSlots * kWinBlocks;   // 8192 slots = 64 KiB
constexpr uint32_t kBackBlocks = 4;              // window keeps this much room behind a tile's lowest key
constexpr int kPerThread = kOwnVecPerThread * 2;

struct DeferredEntry { uint64_t pos; uint64_t packed; };

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// owner[blk]: 0 = free, otherwise (workgroup id + 1).
__global__ void __launch_bounds__(kOwnThreads)
k_build_own(const uint64_t* __restrict__ R, uint64_t n, uint64_t chunkLen,
            uint64_t* __restrict__ table, uint64_t mask, uint32_t probeLen, uint64_t idxBase,
            unsigned int* __restrict__ owner, DeferredEntry* __restrict__ queue,
            unsigned long long* __restrict__ queueCount, Counters* __restrict__ ctr)
{
    extern __shared__ uint64_t win[];            // kWinSlots slots, ring indexed by (slot & (kWinSlots-1))
    __shared__ unsigned int owned[kWinBlocks];   // per ring block: 1 = claimed by this workgroup
    __shared__ unsigned int sTileMin;
    __shared__ unsigned int sDefCount;
    __shared__ unsigned long long sDefBase;

    const uint64_t cb = (uint64_t)blockIdx.x * chunkLen;
    if (cb >= n) return;
    const uint64_t ce = (cb + chunkLen < n) ? cb + chunkLen : n;
    const uint32_t numBlocks = (uint32_t)((mask + 1) >> kBlkShift);   // table blocks (tableSize >= kBlkSlots, host-checked)
    const uint32_t blkMask = numBlocks - 1;
    const uint32_t me = blockIdx.x + 1;

    for (uint32_t i = threadIdx.x; i < kWinSlots; i += kOwnThreads) win[i] = kEmpty;
    if (threadIdx.x < kWinBlocks) owned[threadIdx.x] = 0;
    if (threadIdx.x == 0) { sTileMin = 0xFFFFFFFFu; sDefCount = 0; }
    __syncthreads();

    // window = table blocks [wb, wb + kWinBlocks); haveWin = false until the first tile
    uint32_t wb = 0;
    bool haveWin = false;
    unsigned long long drops = 0, dropSum = 0, inSum = 0, bad = 0, deferred = 0;

    // element e sits at R + e; a 16-byte load needs (R + e) 16-byte aligned, i.e. (e + a0) even
    const long long a0 = (reinterpret_cast<uintptr_t>(R) & 8) ? 1 : 0;
    const long long lcb = (long long)cb, lce = (long long)ce;
    const long long tb0 = lcb - ((lcb + a0) & 1);   // may be cb-1 (even -1): lanes mask elements outside [cb, ce)

    // loads one tile into registers; raw[] holds 2 tuples per 16-byte vector
    auto load_tile = [&](long long tb, uint64_t (&raw)[kPerThread]) {
#pragma unroll
        for (int k = 0; k < kOwnVecPerThread; ++k) {
            const long long e = tb + 2 * ((long long)k * kOwnThreads + threadIdx.x);
            const bool inx = e >= lcb && e < lce, iny = e + 1 >= lcb && e + 1 < lce;
            uint64_t x = 0, y = 0;
            if (inx && iny) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2*>(R + e);
                x = t.x; y = t.y;
            } else {
                if (inx) x = R[e];
                if (iny) y = R[e + 1];
            }
            raw[2 * k] = x; raw[2 * k + 1] = y;
        }
    };

    uint64_t nxt[kPerThread];
    load_tile(tb0, nxt);
    for (long long tb = tb0; tb < lce; tb += kOwnTile) {
        // ---- take the prefetched tile, start loading the next one ----
        uint64_t key[kPerThread];
        uint64_t gidx[kPerThread];
        bool live[kPerThread];
        uint32_t myMin = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < kOwnVecPerThread; ++k) {
            const long long e = tb + 2 * ((long long)k * kOwnThreads + threadIdx.x);
            key[2 * k] = nxt[2 * k]; key[2 * k + 1] = nxt[2 * k + 1];
            gidx[2 * k] = idxBase + (uint64_t)e; gidx[2 * k + 1] = idxBase + (uint64_t)(e + 1);
            live[2 * k] = e >= lcb && e < lce; live[2 * k + 1] = e + 1 >= lcb && e + 1 < lce;
        }
        if (tb + kOwnTile < lce) load_tile(tb + kOwnTile, nxt);
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            if (!live[j]) continue;
            inSum += key[j];
            if ((key[j] >> 32) != 0 || key[j] == 0) { bad += 1; live[j] = false; continue; }
            const uint32_t hb = (uint32_t)((key[j] & mask) >> kBlkShift);
            myMin = hb < myMin ? hb : myMin;
        }
        myMin = wave_min_u32(myMin);
        if ((threadIdx.x & 63) == 0 && myMin != 0xFFFFFFFFu) atomicMin(&sTileMin, myMin);
        __syncthreads();
        const uint32_t tmin = sTileMin;          // 0xFFFFFFFF if the tile holds no valid tuple
        __syncthreads();
        if (threadIdx.x == 0) sTileMin = 0xFFFFFFFFu;

        // ---- slide the window ----
        if (tmin != 0xFFFFFFFFu) {
            uint32_t nb = tmin > kBackBlocks ? tmin - kBackBlocks : 0;
            if (nb + kWinBlocks > numBlocks) nb = numBlocks > kWinBlocks ? numBlocks - kWinBlocks : 0;
            if (!haveWin) {
                // first window: claim all its blocks
                wb = nb;
                if (threadIdx.x < kWinBlocks && wb + threadIdx.x < numBlocks) {
                    const uint32_t blk = wb + threadIdx.x;
                    owned[blk & (kWinBlocks - 1)] = (atomicCAS(&owner[blk], 0u, me) == 0u) ? 1u : 0u;
                }
                haveWin = true;
                __syncthreads();
            } else if (nb > wb) {
                // retire blocks [wb, min(nb, wb+K)): owned ones go to HBM whole, then reset
                const uint32_t nRetire = (nb - wb) < kWinBlocks ? (nb - wb) : kWinBlocks;
                for (uint32_t r = 0; r < nRetire; ++r) {
                    const uint32_t blk = wb + r, ring = blk & (kWinBlocks - 1);
                    if (owned[ring]) {
                        ulonglong2* dst = reinterpret_cast<ulonglong2*>(table + ((uint64_t)blk << kBlkShift));
                        ulonglong2* src = reinterpret_cast<ulonglong2*>(win + ((uint64_t)ring << kBlkShift));
                        for (uint32_t v = threadIdx.x; v < kBlkSlots / 2; v += kOwnThreads) {
                            dst[v] = src[v];
                            src[v] = make_ulonglong2(kEmpty, kEmpty);
                        }
                    }
                }
                __syncthreads();
                // claim the blocks that enter: [max(wb+K, nb), nb+K)
                const uint32_t enter0 = (wb + kWinBlocks > nb) ? wb + kWinBlocks : nb;
                if (threadIdx.x < kWinBlocks) {
                    const uint32_t blk = enter0 + threadIdx.x;
                    if (blk < nb + kWinBlocks && blk < numBlocks)
                        owned[blk & (kWinBlocks - 1)] = (atomicCAS(&owner[blk], 0u, me) == 0u) ? 1u : 0u;
                }
                wb = nb;
                __syncthreads();
            }
        }

        // ---- insert (index priority on LDS) ----
        uint64_t dpos[kPerThread], dval[kPerThread];
        int nd = 0;
#pragma unroll
        for (int j = 0; j < kPerThread; ++j) {
            if (!live[j]) continue;
            uint64_t mine = (gidx[j] << 32) | key[j];
            uint64_t pos = key[j] & mask;
            uint32_t budget = probeLen;
            for (;;) {
                if (budget == 0) { drops += 1; dropSum += (uint32_t)mine; break; }   // NoCCHashBuild.hpp:57-58
                const uint32_t blk = (uint32_t)(pos >> kBlkShift);
                const bool mineBlk = haveWin && blk >= wb && blk < wb + kWinBlocks && owned[blk & (kWinBlocks - 1)];
                if (!mineBlk) { dpos[nd] = pos; dval[nd] = mine; ++nd; break; }      // abort -> deferred queue
                const unsigned long long old =
                    atomicMin(reinterpret_cast<unsigned long long*>(&win[pos & (kWinSlots - 1)]), (unsigned long long)mine);
                if (old == kEmpty || old == mine) break;
                if (old > mine) {
                    mine = old;
                    const uint64_t home = (uint32_t)old & mask;
                    budget = probeLen - ((uint32_t)((pos - home) & mask) + 1);
                } else {
                    budget -= 1;
                }
                pos = (pos + 1) & mask;
            }
        }
        (void)blkMask;

        // ---- flush this tile's aborted tuples to the deferred queue ----
        unsigned int myOff = 0;
        if (nd) myOff = atomicAdd(&sDefCount, (unsigned int)nd);
        __syncthreads();
        const unsigned int tileDef = sDefCount;
        if (tileDef) {
            if (threadIdx.x == 0) sDefBase = atomicAdd(queueCount, (unsigned long long)tileDef);
            __syncthreads();
            const unsigned long long base = sDefBase;
            for (int d = 0; d < nd; ++d) {
                queue[base + myOff + d].pos = dpos[d];
                queue[base + myOff + d].packed = dval[d];
            }
            deferred += nd;
            __syncthreads();
            if (threadIdx.x == 0) sDefCount = 0;
        }
        // (the next tile's first barrier orders the reset)
    }

    // ---- retire what is left of the window ----
    __syncthreads();
    if (haveWin) {
        for (uint32_t r = 0; r < kWinBlocks; ++r) {
            const uint32_t blk = wb + r, ring = blk & (kWinBlocks - 1);
            if (blk < numBlocks && owned[ring]) {
                ulonglong2* dst = reinterpret_cast<ulonglong2*>(table + ((uint64_t)blk << kBlkShift));
                const ulonglong2* src = reinterpret_cast<const ulonglong2*>(win + ((uint64_t)ring << kBlkShift));
                for (uint32_t v = threadIdx.x; v < kBlkSlots / 2; v += kOwnThreads) dst[v] = src[v];
            }
        }
    }
    // counters: one atomic per wavefront
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        drops += __shfl_down(drops, off, 64);
        dropSum += __shfl_down(dropSum, off, 64);
        inSum += __shfl_down(inSum, off, 64);
        bad += __shfl_down(bad, off, 64);
        deferred += __shfl_down(deferred, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (drops) atomicAdd(&ctr->conflicts, drops);
        if (dropSum) atomicAdd(&ctr->conflictSum, dropSum);
        if (inSum) atomicAdd(&ctr->inputSum, inSum);
        if (bad) atomicAdd(&ctr->badKeys, bad);
        if (deferred) atomicAdd(&ctr->spare[0], deferred);
    }
}

// Blocks nobody claimed (and the slack past the table end) get the empty pattern.
__global__ void __launch_bounds__(kBlock)
k_clear_unowned(uint64_t* __restrict__ table, const unsigned int* __restrict__ owner, uint32_t numBlocks,
                uint64_t tableSize)
{
    const ulonglong2 e = make_ulonglong2(kEmpty, kEmpty);
    // one wavefront per block: 64 lanes x 16 B x 4 = 4 KiB
    const uint32_t wavesPerGrid = gridDim.x * (kBlock / 64);
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    for (uint32_t blk = wave; blk < numBlocks; blk += wavesPerGrid) {
        if (owner[blk] != 0) continue;
        ulonglong2* dst = reinterpret_cast<ulonglong2*>(table + ((uint64_t)blk << kBlkShift));
#pragma unroll
        for (uint32_t v = 0; v < kBlkSlots / 2 / 64; ++v) dst[v * 64 + lane] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x < kTableSlack) table[tableSize + threadIdx.x] = kEmpty;
}

// Phase B: finish the probe walk of every deferred tuple with global atomics.
__global__ void __launch_bounds__(kBlock)
k_build_deferred(const DeferredEntry* __restrict__ queue, const unsigned long long* __restrict__ queueCount,
                 uint64_t* __restrict__ table, uint64_t mask, uint32_t probeLen, Counters* __restrict__ ctr)
{
    const unsigned long long nq = *queueCount;
    unsigned long long drops = 0, dropSum = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * kBlock + threadIdx.x; i < nq;
         i += (unsigned long long)gridDim.x * kBlock) {
        uint64_t mine = queue[i].packed;
        uint64_t pos = queue[i].pos;
        const uint64_t home0 = (uint32_t)mine & mask;
        uint32_t budget = probeLen - (uint32_t)((pos - home0) & mask);
        for (;;) {
            if (budget == 0) { drops += 1; dropSum += (uint32_t)mine; break; }
            const unsigned long long old =
                atomicMin(reinterpret_cast<unsigned long long*>(table + pos), (unsigned long long)mine);
            if (old == kEmpty || old == mine) break;
            if (old > mine) {
                mine = old;
                const uint64_t home = (uint32_t)old & mask;
                budget = probeLen - ((uint32_t)((pos - home) & mask) + 1);
            } else {
                budget -= 1;
            }
            pos = (pos + 1) & mask;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        drops += __shfl_down(drops, off, 64);
        dropSum += __shfl_down(dropSum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (drops) atomicAdd(&ctr->conflicts, drops);
        if (dropSum) atomicAdd(&ctr->conflictSum, dropSum);
    }
}

// Locality probe (the reference samples a prefix to decide whether to switch to the
// radix join, HTMHashBuild.hpp:100-154): over nSample tiles spread across R, count
// the tiles whose home-slot span fits the LDS window.
__global__ void __launch_bounds__(kBlock)
k_sample_locality(const uint64_t* __restrict__ R, uint64_t n, uint64_t mask, uint32_t nSample,
                  unsigned int* __restrict__ fitCount)
{
    __shared__ unsigned long long sMin, sMax;
    const uint64_t tiles = (n + kOwnTile - 1) / kOwnTile;
    for (uint32_t s = blockIdx.x; s < nSample; s += gridDim.x) {
        const uint64_t tile = (tiles * s) / nSample;
        const uint64_t b = tile * kOwnTile, e = (b + kOwnTile < n) ? b + kOwnTile : n;
        if (threadIdx.x == 0) { sMin = ~0ull; sMax = 0; }
        __syncthreads();
        unsigned long long lo = ~0ull, hi = 0;
        for (uint64_t i = b + threadIdx.x; i < e; i += kBlock) {
            const unsigned long long h = R[i] & mask;
            lo = h < lo ? h : lo; hi = h > hi ? h : hi;
        }
        if (lo != ~0ull) { atomicMin(&sMin, lo); atomicMax(&sMax, hi); }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long span = sMax >= sMin ? sMax - sMin : 0;
            if (span + 4 <= (unsigned long long)(kWinSlots - kBackBlocks * kBlkSlots)) atomicAdd(fitCount, 1u);
        }
        __syncthreads();
    }
}

// ---- host side ---------------------------------------------------------------
size_t own_queue_bytes(uint64_t rSize) { return (rSize + 64) * sizeof(DeferredEntry); }
size_t own_owner_bytes(uint64_t tableSize) { return ((tableSize >> kBlkShift) + 1) * sizeof(unsigned int); }
bool own_supported(uint64_t tableSize) { return tableSize >= (uint64_t)kWinSlots; }

void launch_sample_locality(const uint64_t* R, uint64_t n, uint64_t tableSize, uint32_t nSample,
                            unsigned int* fitCount, hipStream_t s)
{
    (void)hipMemsetAsync(fitCount, 0, sizeof(unsigned int), s);
    hipLaunchKernelGGL(k_sample_locality, dim3(nSample < 256 ? nSample : 256), dim3(kBlock), 0, s,
                       R, n, tableSize - 1, nSample, fitCount);
}

void launch_build_own(const uint64_t* R, uint64_t n, uint64_t* table, uint64_t tableSize, uint32_t probeLen,
                      uint64_t idxBase, void* ownerBuf, void* queueBuf, unsigned long long* queueCount,
                      Counters* ctr, hipEvent_t evPhaseA, hipStream_t s)
{
    static bool attrSet = false;
    if (!attrSet) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_build_own),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kWinSlots * sizeof(uint64_t));
        attrSet = true;
    }
    const uint32_t numBlocks = (uint32_t)(tableSize >> kBlkShift);
    (void)hipMemsetAsync(ownerBuf, 0, own_owner_bytes(tableSize), s);
    (void)hipMemsetAsync(queueCount, 0, sizeof(unsigned long long), s);
    // chunks: ~1024 workgroups (2 resident per CU x 256 CUs x 2 rounds), whole tiles
    uint64_t chunkLen = (n + 1023) / 1024;
    chunkLen = (chunkLen + kOwnTile - 1) / kOwnTile * kOwnTile;
    if (chunkLen < (uint64_t)kOwnTile * 4) chunkLen = (uint64_t)kOwnTile * 4;
    const unsigned grid = (unsigned)((n + chunkLen - 1) / chunkLen);
    hipLaunchKernelGGL(k_build_own, dim3(grid), dim3(kOwnThreads), kWinSlots * sizeof(uint64_t), s,
                       R, n, chunkLen, table, tableSize - 1, probeLen, idxBase,
                       static_cast<unsigned int*>(ownerBuf), static_cast<DeferredEntry*>(queueBuf), queueCount, ctr);
    if (evPhaseA) (void)hipEventRecord(evPhaseA, s);
    hipLaunchKernelGGL(k_clear_unowned, dim3(2048), dim3(kBlock), 0, s, table,
                       static_cast<const unsigned int*>(ownerBuf), numBlocks, tableSize);
    hipLaunchKernelGGL(k_build_deferred, dim3(1024), dim3(kBlock), 0, s,
                       static_cast<const DeferredEntry*>(queueBuf), queueCount, table, tableSize - 1, probeLen, ctr);
}

}  // namespace hj

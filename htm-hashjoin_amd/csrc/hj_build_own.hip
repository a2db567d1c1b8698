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
//            claims a block (one atomicCAS on a small owner table) lazily, the first
//            time a tile needs it. Inserts into owned blocks run the index-priority
//            protocol of hj_kernels.hip on LDS (ds_min_rtn_u64). When the window
//            slides, finished blocks go to HBM as whole 4 KiB runs of plain
//            16-byte stores. A tuple whose probe walk reaches a block that is not
//            owned (lost claim at a chunk seam, key far from the window, spill
//            past the window end) is "aborted": it is appended, with the slot it
//            had reached, to a deferred queue.
//   clear    k_finalize_range + k_clear_unowned: the block range any tuple can reach
//            becomes the table's "valid range"; inside it, blocks nobody claimed are
//            filled with the empty pattern (owned blocks were written whole in phase
//            A, so every reachable slot is written exactly once); outside it the
//            table is neither written nor, later, probed.
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

#include <type_traits>

namespace hj {

constexpr int kOwnThreads = 512;                 // 8 wavefronts
constexpr int kOwnTile = 4096;                    // tuples per SAMPLE tile and the unit chunk lengths are rounded to (the build's own tiles: k_build_own, PER)
constexpr uint32_t kBlkShift = 9;
constexpr uint32_t kBlkSlots = 1u << kBlkShift;  // 512 slots = 4 KiB
constexpr uint32_t kWinBlocks = 16;
constexpr uint32_t kWinSlots = kBlkSlots * kWinBlocks;   // 8192 slots = 64 KiB
constexpr uint32_t kBackBlocks = 2;              // window keeps this much room behind a tile's lowest key

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return wave_umin(v); }      // DPP steps (hj_device.h), not LDS permutes

__device__ __forceinline__ uint64_t pack64(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }

#ifndef HJ_OWN_PRIO
#define HJ_OWN_PRIO 0
#endif
#ifndef HJ_OWN_PRIO_SHIFT
#define HJ_OWN_PRIO_SHIFT 10
#endif
#ifndef HJ_OWN_MAX_ROUNDS
#define HJ_OWN_MAX_ROUNDS 1                      // rounds of workgroups for large relations (see launch_build_own)
#endif
#ifndef HJ_OWN_MIN_CHUNK
#define HJ_OWN_MIN_CHUNK 524288                  // tuples per chunk below which no further round is added
#endif
#ifndef HJ_DRAIN_INPLACE
#define HJ_DRAIN_INPLACE 1
#endif
#ifndef HJ_LOOK_FIRST
#define HJ_LOOK_FIRST 1
#endif
#ifndef HJ_DRAIN_AT
#define HJ_DRAIN_AT 64
#endif
constexpr uint32_t kDrainAt = HJ_DRAIN_AT;            // run retry rounds once this many entries wait (<= 64)
constexpr int kQCap = 128;                        // per-wavefront retry queue entries (LDS)

// owner[blk]: 0 = free, otherwise (workgroup id + 1).
// KEY32 = false: R holds DataGen tuples (value = key); KEY32 = true: bare 32-bit keys (what the multi-GPU exchange
// delivers). Either way index = idxBase + position and home slot = (key >> hshift) & mask (hj_device.h).
//
// rocprof showed the first versions VALU-issue bound (120 VALU + 90 SALU instructions per 64
// tuples on unique keys, 5x that on duplicate-heavy `uniform`, LDS <10 % busy), so the insert is
// split in two:
//   fast step   64 consecutive tuples (lanes in input order) each try their home slot once with an
//               LDS atomicMin. Unique, in-window, owned -> done in ~10 instructions.
//   retry queue whatever did not finish (slot taken, displaced a later tuple, block not owned) is
//               pushed -- compacted with a ballot -- to a small per-wavefront LDS queue as
//               (slot, value). Full rounds (ownership test, look-before-leap over the probe window,
//               atomicMin) run on 64 queue entries at a time, so every round is dense regardless of
//               how long individual probe/displacement chains get.
// All per-tuple arithmetic is 32-bit: key = low word, slot numbers < 2^32, value = {key, index}.
// HTM = true: the bucketised table of --algo htm (hj_device.h, home_slot_htm: every tuple of a bucket has the bucket's
// first slot as its home, probeLen = 3); a tuple that runs out of budget is one of the reference's conflicts
// (HTMHashBuild.hpp:181-183) and is appended, as (index << 32 | key), to this workgroup's slice of htmConflicts
// (confSlice entries per workgroup; ccounts[workgroup] = how many) for the chain phase (hj_htm.hip).
template <bool KEY32, bool CHECK = false, bool HTM = false>
__global__ void __launch_bounds__(kOwnThreads, 4)
k_build_own(const void* __restrict__ Rv, uint64_t n, uint64_t chunkLen,
            uint64_t* __restrict__ table, uint64_t mask, uint32_t hshift, uint32_t probeLen, uint64_t idxBase, ShardCheck sc,
            unsigned int* __restrict__ owner, DeferredEntry* __restrict__ queue,
            uint32_t* __restrict__ deferCounts, Counters* __restrict__ ctr, Gate gate,
            uint64_t* __restrict__ htmConflicts, uint32_t* __restrict__ ccounts, uint32_t confSlice)
{
    if (gate_closed(gate)) return;
    extern __shared__ __align__(16) uint64_t win[];   // kWinSlots slots, ring indexed by (slot & (kWinSlots-1))
    __shared__ unsigned int owned[kWinBlocks];   // per ring block: 0 = not tried yet, 1 = claimed by this workgroup, 2 = someone else's
    __shared__ unsigned int need[kWinBlocks];    // per ring block: wanted by the current tile (count, or 0x10000 = unconditional)
    __shared__ unsigned int sTileMin;
    __shared__ unsigned int sConf;               // HTM: conflicts this workgroup has listed so far
    __shared__ unsigned int sDefer;              // tuples this workgroup has deferred so far (entries of its queue slice)
    __shared__ uint32_t qPos[kOwnThreads / 64][kQCap], qLo[kOwnThreads / 64][kQCap], qHi[kOwnThreads / 64][kQCap];

    const uint64_t cb = (uint64_t)blockIdx.x * chunkLen;
    if (cb >= n) return;
    const uint64_t ce = (cb + chunkLen < n) ? cb + chunkLen : n;
    const uint32_t clen = (uint32_t)(ce - cb);                        // chunk length (< 2^32)
    const uint32_t mask32 = (uint32_t)mask;                           // tableSize <= 2^32 slots
    const uint32_t numBlocks = (uint32_t)((mask + 1) >> kBlkShift);   // tableSize >= kWinSlots, host-checked
    const uint32_t me = blockIdx.x + 1;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    using Elem = typename std::conditional<KEY32, uint32_t, uint64_t>::type;
    const Elem* __restrict__ Rc = static_cast<const Elem*>(Rv) + cb;   // this chunk
    const uint32_t idx0 = (uint32_t)(idxBase + cb);                    // index of the chunk's first tuple (fits 32 bits, host-checked)
    uint32_t* const myQPos = qPos[wave]; uint32_t* const myQLo = qLo[wave]; uint32_t* const myQHi = qHi[wave];

    for (uint32_t i = threadIdx.x; i < kWinSlots; i += kOwnThreads) win[i] = kEmpty;
    if (threadIdx.x < kWinBlocks) { owned[threadIdx.x] = 0; need[threadIdx.x] = 0; }
    if (threadIdx.x == 0) { sTileMin = 0xFFFFFFFFu; sConf = 0; sDefer = 0; }
    __syncthreads();
    // The deferred queue is sliced like R: this workgroup's entries go to queue[cb ..), at most one per tuple of its
    // chunk, counted in LDS. (One global counter for all workgroups -- a returning atomic on ONE address per retry round
    // that defers anything -- was what loose locality paid for: 3.5 ms of a 4.4 ms phase A on the bucketised table at
    // 2^27, W = 1024, for 4.8 % of the tuples deferred.)
    DeferredEntry* const myQueue = queue + cb;
    uint64_t* const myConflicts = HTM ? htmConflicts + (uint64_t)blockIdx.x * confSlice : nullptr;
    (void)myConflicts;

    uint32_t wb = 0;                 // window = table blocks [wb, wb + kWinBlocks)
    bool haveWin = false;            // false until the first tile with a valid tuple
    unsigned long long dropSum = 0, inSum = 0;
    uint32_t drops = 0, bad = 0, deferred = 0, foreign = 0;
    uint32_t usedLo = 0xFFFFFFFFu, usedHi1 = 0;   // blocks this thread claimed or deferred into
    uint32_t ownedMask = 0;          // bit r: ring block r is mine (refreshed per tile)
    uint32_t qCount = 0;             // entries in this wavefront's retry queue (wave-uniform)

    // the work of one round on the entry (pos, mlo, mhi) a lane holds; returns "not finished" and leaves the entry's
    // next state in place
    auto round_body = [&](uint32_t& pos, uint32_t& mlo, uint32_t& mhi, const bool has) -> bool {
        const uint32_t key = mlo;
        uint32_t budget = probeLen - ((pos - home32<HTM>(key, hshift, mask32)) & mask32);
        const uint32_t blk = pos >> kBlkShift;
        const bool ownOk = (blk - wb < kWinBlocks) & (((ownedMask >> (blk & (kWinBlocks - 1))) & 1u) != 0);
        const bool drop0 = has & (budget == 0);                           // NoCCHashBuild.hpp:57-58
        const bool toDefer = has & !drop0 & !ownOk;                        // abort -> global deferred queue
        const bool work = has & !drop0 & ownOk;
        const uint64_t mine = pack64(mhi, mlo);
        // look before leaping: read the next 4 slots when they sit in this (owned) block -- otherwise
        // read the block's first 4 slots and ignore them. Slot values only decrease, so a slot seen
        // below `mine` stays below it.
        const bool inBlk = HJ_LOOK_FIRST && ((pos & (kBlkSlots - 1)) <= kBlkSlots - 4);
        const uint32_t rd = work ? (inBlk ? pos : (pos & ~(kBlkSlots - 1))) : 0u;
        const uint64_t* w = &win[rd & (kWinSlots - 1)];
        const uint64_t v0 = w[0], v1 = w[1], v2 = w[2], v3 = w[3];
        const bool c0 = v0 < mine, c1 = c0 & (v1 < mine), c2 = c1 & (v2 < mine), c3 = c2 & (v3 < mine);
        uint32_t skip = (uint32_t)c0 + (uint32_t)c1 + (uint32_t)c2 + (uint32_t)c3;
        skip = (work & inBlk) ? (skip < budget ? skip : budget) : 0u;
        pos = (pos + skip) & mask32; budget -= skip;
        const bool drop1 = work & (budget == 0);
        const bool recheck = work & !drop1 & (skip == 4);                  // may have left the block: next round
        const bool doAtomic = work & !drop1 & !recheck;
        unsigned long long old = kEmpty;
        if (doAtomic)
            old = atomicMin(reinterpret_cast<unsigned long long*>(&win[pos & (kWinSlots - 1)]), (unsigned long long)mine);
        const bool fail = doAtomic & (old != kEmpty) & (old != mine);
        const bool disp = fail & (old > mine);                             // displaced a later tuple: carry it on
        mlo = disp ? (uint32_t)old : mlo; mhi = disp ? (uint32_t)(old >> 32) : mhi;
        const bool dropped = drop0 | drop1;
        drops += dropped ? 1u : 0u; dropSum += dropped ? (unsigned long long)key : 0ull;
        if constexpr (HTM) {       // the bucket is full: one of the reference's conflicts, listed for the chain phase
            const unsigned long long cm = __ballot(dropped);
            if (cm) {
                unsigned int base = 0;
                if (lane == 0) base = atomicAdd(&sConf, (unsigned int)__popcll(cm));
                base = (unsigned int)__shfl((int)base, 0, 64);
                if (dropped) myConflicts[base + lane_rank(cm)] = mine;
            }
        }
        // deferred tuples leave for the global queue (one returning atomic per round that has any)
        const unsigned long long dm = __ballot(toDefer);
        if (dm) {
            unsigned int base = 0;
            if (lane == 0) base = atomicAdd(&sDefer, (unsigned int)__popcll(dm));
            base = (unsigned int)__shfl((int)base, 0, 64);
            if (toDefer) {
                const uint32_t at = base + lane_rank(dm);
                myQueue[at].pos = pos; myQueue[at].packed = mine;
                deferred += 1;
                const uint32_t db = pos >> kBlkShift;
                usedLo = db < usedLo ? db : usedLo; usedHi1 = db + 1 > usedHi1 ? db + 1 : usedHi1;
            }
        }
        pos = fail ? ((pos + 1) & mask32) : pos;
        return recheck | fail;
    };
    // One dense retry round over up to 64 queue entries (popped from the tail). Unfinished entries
    // are pushed back. Terminal events: placed, dropped (budget exhausted), deferred (block not owned).
    // (written with flags and selects rather than nested branches: the first version of this kernel
    //  spent as many SALU instructions on exec-mask bookkeeping as VALU instructions on tuples)
    auto retry_round = [&]() {
        const uint32_t take = qCount < 64u ? qCount : 64u;
        qCount -= take;
        const bool has = lane < take;
        // lanes >= take read stale-but-in-bounds queue entries (qCount + lane < kQCap) and ignore them
        uint32_t pos = myQPos[qCount + lane], mlo = myQLo[qCount + lane], mhi = myQHi[qCount + lane];
        const bool again = round_body(pos, mlo, mhi, has);
        const unsigned long long am = __ballot(again);
        if (again) {
            const uint32_t at = qCount + lane_rank(am);
            myQPos[at] = pos; myQLo[at] = mlo; myQHi[at] = mhi;
        }
        qCount += (uint32_t)__popcll(am);
    };
    // End-of-tile drain: dense rounds while more than a wavefront's worth is queued, then the last <= 64 entries
    // stay in registers until they are done (no queue round trip between the sparse rounds: 852 -> 828 us on
    // `uniform` at 2^27; doing the same inside the dense rounds while >= 40/24/12 lanes stay busy was slower)
    auto drain = [&]() {
#if HJ_DRAIN_INPLACE
        while (qCount > 64u) retry_round();
        bool act = lane < qCount;
        uint32_t pos = myQPos[lane], mlo = myQLo[lane], mhi = myQHi[lane];
        qCount = 0;
        while (__ballot(act)) act = round_body(pos, mlo, mhi, act);
#else
        while (qCount) retry_round();
#endif
    };

    // Tile geometry: 3072 tuples, six per thread (round 3; 4096 before). A shorter tile's home slots span less of the 16-block
    // window, so the window holds wider shuffle windows: at 2^27 the open-addressing build takes 549 instead of 743 us at
    // W = 2^11 and 2.16 instead of 2.48 ms at 2^12, the same 511 us at W = 2^10 and 2 % more at 2^8 (more tiles = more window
    // slides and barriers per tuple; tiles of 2048: +5 % there, little more gained beyond). For the bucketised table, whose
    // keys spread 4/3 as wide (four slots per three keys), it is what makes W = 2^10 fit at all: 4.8 % of the tuples
    // deferred with tiles of 4096, 0.25 % with 3072 (build 1.58 -> 1.03 ms).
#ifndef HJ_OWN_PER
#define HJ_OWN_PER 6
#endif
    constexpr int PER = HJ_OWN_PER;
    constexpr int TILE = kOwnThreads * PER, SPAN = 64 * PER;
    // tile t covers chunk offsets [t*TILE, ...); this thread's tuple j sits at offset
    // t*TILE + wave*SPAN + 64 j + lane
    const uint32_t tOff = wave * SPAN + lane;
    uint64_t nxt[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const uint32_t o = tOff + 64 * j;
        nxt[j] = o < clen ? Rc[o] : 0;
    }

    for (uint32_t tb = 0; tb < clen; tb += TILE) {
        // ---- take the prefetched tile, start loading the next one ----
        uint32_t klo[PER], khi[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) { klo[j] = (uint32_t)nxt[j]; khi[j] = (uint32_t)(nxt[j] >> 32); }
#if HJ_OWN_PRIO
        {   // the CU's two workgroup slots take turns at the higher issue priority (hj_build_wave.hip: oldest-first
            // arbitration lets the workgroup a CU received first finish well before the second)
            const uint32_t slot = (blockIdx.x / ((gridDim.x + 1u) / 2u)) & 1u;
            if ((((uint32_t)(wall_clock64() >> HJ_OWN_PRIO_SHIFT)) + slot) & 1u) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        const bool full = tb + TILE <= clen;                      // wave-uniform
        const bool firstTile = tb == 0, lastTile = tb + TILE >= clen;
        if (!lastTile) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t o = tb + TILE + tOff + 64 * j;
                nxt[j] = o < clen ? Rc[o] : 0;
            }
        }
        uint32_t liveMask = 0;
        uint32_t myMin = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const bool in = full | (tb + tOff + 64 * j < clen);
            const bool okKey = (khi[j] == 0) & (klo[j] != 0);
            const bool ok = in & okKey;
            inSum += in ? (unsigned long long)pack64(khi[j], klo[j]) : 0ull;
            bad += (in & !okKey) ? 1u : 0u;
            if constexpr (CHECK) foreign += (in & is_foreign(klo[j], sc)) ? 1u : 0u;   // shard check: its own instance
            liveMask |= ok ? (1u << j) : 0u;
            const uint32_t hb = home32<HTM>(klo[j], hshift, mask32) >> kBlkShift;
            myMin = (ok & (hb < myMin)) ? hb : myMin;
        }
        myMin = wave_min_u32(myMin);
        if (lane == 0 && myMin != 0xFFFFFFFFu) atomicMin(&sTileMin, myMin);
        __syncthreads();
        const uint32_t tmin = sTileMin;          // 0xFFFFFFFF if the tile holds no valid tuple
        __syncthreads();
        if (threadIdx.x == 0) sTileMin = 0xFFFFFFFFu;

        // ---- slide the window ----
        if (tmin != 0xFFFFFFFFu) {
            uint32_t nb = tmin > kBackBlocks ? tmin - kBackBlocks : 0;
            if (nb + kWinBlocks > numBlocks) nb = numBlocks - kWinBlocks;
            if (!haveWin) {
                wb = nb;             // first window; blocks are claimed lazily below
                haveWin = true;
            } else if (nb > wb) {
                // retire blocks [wb, min(nb, wb+K)): owned ones go to HBM whole (empties included), then their
                // LDS copy is reset (ablation, round 1: this phase is 200 of 550 us at 2^27). ownedMask still describes
                // the window of the previous tile, i.e. exactly the blocks that leave.
                const uint32_t nRetire = (nb - wb) < kWinBlocks ? (nb - wb) : kWinBlocks;
                constexpr uint32_t kVecPerBlk = kBlkSlots / 2;                      // 256 x 16 B
                constexpr uint32_t kRetireIter = kWinBlocks * kVecPerBlk / kOwnThreads;   // 8
                // two blocks per pass: threads 0..255 take block r, 256..511 block r+1, one vector each
                // (more vectors in flight per thread spill: the kernel sits at the 128-VGPR cap)
                (void)kRetireIter;
                const uint32_t half = threadIdx.x / kVecPerBlk, v = threadIdx.x % kVecPerBlk;
                for (uint32_t r0 = 0; r0 < nRetire; r0 += kOwnThreads / kVecPerBlk) {
                    const uint32_t r = r0 + half;
                    const uint32_t blk = wb + r, ring = blk & (kWinBlocks - 1);
                    if (r < nRetire && ((ownedMask >> ring) & 1u)) {
                        ulonglong2* src = reinterpret_cast<ulonglong2*>(win + ((uint64_t)ring << kBlkShift)) + v;
                        const ulonglong2 t = *src;
                        reinterpret_cast<ulonglong2*>(table + ((uint64_t)blk << kBlkShift))[v] = t;
                        *src = make_ulonglong2(kEmpty, kEmpty);
                    }
                }
                __syncthreads();
                // the ring positions just vacated now stand for the blocks that enter: not tried yet
                if (threadIdx.x < nRetire) owned[(wb + threadIdx.x) & (kWinBlocks - 1)] = 0;
                wb = nb;
            }
        }

        // ---- claim, lazily, the blocks this tile needs ----
        // need[ring]: wanted blocks (home block; also the next block when a probe window straddles
        // the block end). In a chunk's first and last tile the home blocks are COUNTED and a block is
        // claimed only if it holds at least 1/4 of what the tile's fullest block holds: stragglers
        // across a chunk seam must not take a whole block from the neighbour chunk that fills it
        // (they are deferred instead). Elsewhere any touched block is claimed.
        const bool seamTile = firstTile || lastTile;
        if (haveWin) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const uint32_t home = home32<HTM>(klo[j], hshift, mask32);
                const uint32_t hb = home >> kBlkShift;
                const bool lv = (liveMask >> j) & 1u;
                uint32_t r0 = (lv & (hb - wb < kWinBlocks)) ? (hb & (kWinBlocks - 1)) : 0xFFu;
                const uint32_t eb = ((home + probeLen - 1) & mask32) >> kBlkShift;
                if (lv & (eb != hb) & (eb - wb < kWinBlocks)) need[eb & (kWinBlocks - 1)] = 0x10000u;   // straddle: always wanted
                if (seamTile) {
                    // wave-aggregated counting: near-sorted input puts a wavefront in 1-2 blocks
                    for (;;) {
                        const unsigned long long act = __ballot(r0 != 0xFFu);
                        if (!act) break;
                        const uint32_t lead = __shfl(r0, __ffsll((long long)act) - 1, 64);
                        const unsigned long long same = __ballot(r0 == lead);
                        if (lane == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&need[lead], (unsigned int)__popcll(same));
                        if (r0 == lead) r0 = 0xFFu;
                    }
                } else if (r0 != 0xFFu) {
                    need[r0] = 0x10000u;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t t = threadIdx.x;
            const uint32_t c = t < kWinBlocks ? need[t] : 0;
            uint32_t mx = c < 0x10000u ? c : 0;      // fullest COUNTED block (flags are not counts)
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(mx, off, 64); mx = o > mx ? o : mx; }
            if (t < kWinBlocks) {
                need[t] = 0;
                if (c && (c >= 0x10000u || c * 4 >= mx) && owned[t] == 0) {
                    const uint32_t blk = wb + ((t - wb) & (kWinBlocks - 1));   // ring position -> block in [wb, wb+K)
                    owned[t] = (blk < numBlocks && atomicCAS(&owner[blk], 0u, me) == 0u) ? 1u : 2u;
                    if (owned[t] == 1u) { usedLo = blk < usedLo ? blk : usedLo; usedHi1 = blk + 1 > usedHi1 ? blk + 1 : usedHi1; }
                }
            }
        }
        __syncthreads();
        // one LDS read per wavefront: bit r = ring block r is mine
        ownedMask = (uint32_t)__ballot(lane < kWinBlocks && owned[lane & (kWinBlocks - 1)] == 1u && haveWin);

        // ---- insert: fast step per 64 consecutive tuples, everything else through the retry queue ----
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            while (qCount >= kDrainAt) retry_round();                   // keep room for one full step
            const bool lv = (liveMask >> j) & 1u;
            uint32_t mlo = klo[j];
            uint32_t mhi = idx0 + tb + tOff + 64 * j;
            uint32_t pos = home32<HTM>(klo[j], hshift, mask32);
            const uint32_t blk = pos >> kBlkShift;
            const bool own = lv & (blk - wb < kWinBlocks) & (((ownedMask >> (blk & (kWinBlocks - 1))) & 1u) != 0);
            const uint64_t mine = pack64(mhi, mlo);
            unsigned long long old = kEmpty;
            if (own)
                old = atomicMin(reinterpret_cast<unsigned long long*>(&win[pos & (kWinSlots - 1)]), (unsigned long long)mine);
            const bool fail = own & (old != kEmpty);                    // slot was taken
            const bool disp = fail & (old > mine);                      // ... by a later tuple: it moves on instead
            mlo = disp ? (uint32_t)old : mlo; mhi = disp ? (uint32_t)(old >> 32) : mhi;
            pos = fail ? ((pos + 1) & mask32) : pos;
            const bool again = fail | (lv & !own);                      // not owned: the retry round defers it
            const unsigned long long am = __ballot(again);
            if (am) {
                if (again) {
                    const uint32_t at = qCount + lane_rank(am);
                    myQPos[at] = pos; myQLo[at] = mlo; myQHi[at] = mhi;
                }
                qCount += (uint32_t)__popcll(am);
            }
        }
        drain();                          // before the window may slide
    }

    // ---- retire what is left of the window ----
    __syncthreads();
    if (haveWin) {
        for (uint32_t r = 0; r < kWinBlocks; ++r) {
            const uint32_t blk = wb + r, ring = blk & (kWinBlocks - 1);
            if (blk < numBlocks && owned[ring] == 1) {
                ulonglong2* dst = reinterpret_cast<ulonglong2*>(table + ((uint64_t)blk << kBlkShift));
                const ulonglong2* src = reinterpret_cast<const ulonglong2*>(win + ((uint64_t)ring << kBlkShift));
                for (uint32_t v = threadIdx.x; v < kBlkSlots / 2; v += kOwnThreads) dst[v] = src[v];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        deferCounts[blockIdx.x] = sDefer;
        if constexpr (HTM) ccounts[blockIdx.x] = sConf;
    }
    // counters: one atomic per wavefront
    unsigned long long c0 = drops, c3 = bad | ((unsigned long long)foreign << 32), c4 = deferred;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off, 64);
        dropSum += __shfl_down(dropSum, off, 64);
        inSum += __shfl_down(inSum, off, 64);
        c3 += __shfl_down(c3, off, 64);
        c4 += __shfl_down(c4, off, 64);
    }
    // block range touched: min via max of the complement (counters start at 0)
    uint32_t loInv = ~usedLo, hi1 = usedHi1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = __shfl_down(loInv, off, 64), b = __shfl_down(hi1, off, 64);
        loInv = a > loInv ? a : loInv; hi1 = b > hi1 ? b : hi1;
    }
    if (lane == 0) {
        Counters::Shard* const sh = counter_shard(ctr);
        if (c0) atomicAdd(&sh->conflicts, c0);
        if (dropSum) atomicAdd(&sh->conflictSum, dropSum);
        if (inSum) atomicAdd(&sh->inputSum, inSum);
        if (c3 & 0xFFFFFFFFull) atomicAdd(&sh->badKeys, c3 & 0xFFFFFFFFull);
        if (c3 >> 32) atomicAdd(&sh->foreign, c3 >> 32);
        if (c4) atomicAdd(&sh->deferred, c4);
        if (hi1) { atomicMax(&sh->usedLoInv, (unsigned long long)loInv); atomicMax(&sh->usedHi1, (unsigned long long)hi1); }
    }
}

// After phase A: the valid slot range. Stored tuples end up in blocks [lo, hi+1] (a probe walk spills
// at most probeLen-1 slots past a home slot), probes may read one block further, so blocks [lo, hi+2]
// are given defined contents and home slots in [lo*512, (hi+2)*512) are probed. If that reaches the
// table end (probe walks wrap there) the whole table is made valid.
__global__ void k_finalize_range(Counters* __restrict__ ctr, uint32_t numBlocks, uint64_t tableSize, Gate gate)
{
    if (blockIdx.x != 0 || threadIdx.x >= 64 || gate_closed(gate)) return;
    // the two maxima: what was written directly + the 64 shards (hj_device.h, Counters), one shard per lane
    static_assert(Counters::kShards == 64, "one shard per lane of the single wavefront this kernel runs as");
    unsigned long long usedLoInvAll = ctr->shard[threadIdx.x & 63].usedLoInv, usedHi1All = ctr->shard[threadIdx.x & 63].usedHi1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(usedLoInvAll, off, 64), b = __shfl_xor(usedHi1All, off, 64);
        usedLoInvAll = a > usedLoInvAll ? a : usedLoInvAll; usedHi1All = b > usedHi1All ? b : usedHi1All;
    }
    usedLoInvAll = ctr->usedLoInv > usedLoInvAll ? ctr->usedLoInv : usedLoInvAll;
    usedHi1All = ctr->usedHi1 > usedHi1All ? ctr->usedHi1 : usedHi1All;
    if (threadIdx.x != 0) return;
    const unsigned long long hi1 = usedHi1All;
    if (hi1 == 0) { ctr->validLo = 0; ctr->validHiEx = 0; return; }          // nothing inserted anywhere
    const unsigned long long lo = (unsigned long long)(uint32_t)~(uint32_t)usedLoInvAll;
    const unsigned long long hiEx = hi1 + 1;                                   // blocks [lo, hi+1] probed
    if (hiEx + 1 >= numBlocks) { ctr->validLo = 0; ctr->validHiEx = tableSize; }
    else { ctr->validLo = lo << kBlkShift; ctr->validHiEx = hiEx << kBlkShift; }
}

// Blocks of the valid range nobody claimed (and the slack past the table end) get the empty pattern.
// Blocks outside [validLo, validHiEx + 512) are never read, so they are not written either.
__global__ void __launch_bounds__(kBlock)
k_clear_unowned(uint64_t* __restrict__ table, const unsigned int* __restrict__ owner, const Counters* __restrict__ ctr,
                uint32_t numBlocks, uint64_t tableSize, Gate gate)
{
    if (gate_closed(gate)) return;
    const ulonglong2 e = make_ulonglong2(kEmpty, kEmpty);
    const uint32_t b0 = (uint32_t)(ctr->validLo >> kBlkShift);
    uint32_t b1 = (uint32_t)(ctr->validHiEx >> kBlkShift) + 1;     // exclusive; one block past the probed range
    if (b1 > numBlocks) b1 = numBlocks;
    // a wavefront looks at 64 owner words at a time (one per lane) and then fills the unclaimed blocks among
    // them, each as 64 lanes x 16 B x 4 = 4 KiB (one dependent owner load per block made this kernel
    // latency bound: 100 us at 2^30 for 8 MB of owner words)
    const uint32_t wavesPerGrid = gridDim.x * (kBlock / 64);
    const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t base = (uint64_t)b0 + (uint64_t)wave * 64; base < b1; base += (uint64_t)wavesPerGrid * 64) {
        const uint64_t mine = base + lane;
        unsigned long long m = __ballot(mine < b1 && owner[mine] == 0);
        while (m) {
            const uint32_t j = (uint32_t)__ffsll((long long)m) - 1;
            m &= m - 1;
            ulonglong2* dst = reinterpret_cast<ulonglong2*>(table + ((base + j) << kBlkShift));
#pragma unroll
            for (uint32_t v = 0; v < kBlkSlots / 2 / 64; ++v) dst[v * 64 + lane] = e;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < kTableSlack) table[tableSize + threadIdx.x] = kEmpty;
}

// Phase B: finish the probe walk of every deferred tuple with global atomics. HTM: the tuples that run out of budget
// here are conflicts too; they go to the list's LAST slice (one slot per tuple of the relation: everything may be
// deferred), reserved with one atomic per wavefront.
template <bool HTM>
__global__ void __launch_bounds__(kBlock)
k_build_deferred(const DeferredEntry* __restrict__ queueAll, const uint32_t* __restrict__ deferCounts, uint32_t sliceLen, uint32_t parts,
                 uint64_t* __restrict__ table, uint64_t mask, uint32_t hshift, uint32_t probeLen,
                 Counters* __restrict__ ctr, Gate gate, uint64_t* __restrict__ lastSlice, uint32_t* __restrict__ lastCount)
{
    if (gate_closed(gate)) return;
    // slice c = blockIdx.x / parts (the entries workgroup c of phase A deferred), `parts` workgroups share it
    const uint32_t c = blockIdx.x / parts, part = blockIdx.x - c * parts;
    const DeferredEntry* __restrict__ queue = queueAll + (uint64_t)c * sliceLen;
    const unsigned long long nq = deferCounts[c];
    unsigned long long drops = 0, dropSum = 0;
    const unsigned long long nqUp = (nq + 63ull) & ~63ull;                  // whole wavefronts stay in the loop (ballots below)
    for (unsigned long long i = (unsigned long long)part * kBlock + threadIdx.x; i < nqUp;
         i += (unsigned long long)parts * kBlock) {
        const bool has = i < nq;
        uint64_t mine = has ? queue[i].packed : 0;
        uint64_t pos = has ? queue[i].pos : 0;
        const uint64_t home0 = home32<HTM>((uint32_t)mine, hshift, (uint32_t)mask);
        uint32_t budget = probeLen - (uint32_t)((pos - home0) & mask);
        bool dropped = false;
        for (; has;) {
            if (budget == 0) { drops += 1; dropSum += (uint32_t)mine; dropped = true; break; }
            const unsigned long long old =
                atomicMin(reinterpret_cast<unsigned long long*>(table + pos), (unsigned long long)mine);
            if (old == kEmpty || old == mine) break;
            if (old > mine) {
                mine = old;
                const uint64_t home = home32<HTM>((uint32_t)old, hshift, (uint32_t)mask);
                budget = probeLen - ((uint32_t)((pos - home) & mask) + 1);
            } else {
                budget -= 1;
            }
            pos = (pos + 1) & mask;
        }
        if constexpr (HTM) {
            const unsigned long long cm = __ballot(dropped);
            if (cm) {
                const uint32_t lane = threadIdx.x & 63;
                uint32_t base = 0;
                if (lane == (uint32_t)__ffsll((long long)cm) - 1u) base = atomicAdd(lastCount, (uint32_t)__popcll(cm));
                base = (uint32_t)__shfl((int)base, __ffsll((long long)cm) - 1, 64);
                if (dropped) lastSlice[base + lane_rank(cm)] = mine;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        drops += __shfl_down(drops, off, 64);
        dropSum += __shfl_down(dropSum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (drops) atomicAdd(&counter_shard(ctr)->conflicts, drops);
        if (dropSum) atomicAdd(&counter_shard(ctr)->conflictSum, dropSum);
    }
}

// Locality probe (the reference samples a prefix to decide whether to switch to the
// radix join, HTMHashBuild.hpp:100-154): over nSample tiles spread across R, count the
// tuples that would fall outside the LDS window k_build_own would place for their tile
// (window base = the tile's lowest home block - kBackBlocks) and therefore be deferred, and the
// same for the 8 KiB ring k_build_wave would place for each of the tile's eight wavefront tiles.
// out[0] = tuples outside variant 2's window, out[1] = tuples looked at, out[2] = outside variant 3's ring.
template <bool KEY32, bool HTM = false>
__global__ void __launch_bounds__(kBlock)
k_sample_locality(const void* __restrict__ Rv, uint64_t n, uint64_t mask, uint32_t hshift, uint32_t nSample,
                  unsigned int* __restrict__ out, SamplePick pick)
{
    using Elem = typename std::conditional<KEY32, uint32_t, uint64_t>::type;
    const Elem* __restrict__ R = static_cast<const Elem*>(Rv);
    __shared__ unsigned int sMinBlk, sOutside, sOutsideWave, sDup, sFar, sLast;
    __shared__ unsigned int seen[(2 * kOwnTile) / 32];               // one bit per home slot of a tile's neighbourhood
    const uint64_t tiles = (n + kOwnTile - 1) / kOwnTile;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int tot[5] = {0, 0, 0, 0, 0};                             // thread 0: this workgroup's five counts
    for (uint32_t s = blockIdx.x; s < nSample; s += gridDim.x) {
        const uint64_t tile = (tiles * s) / nSample;
        const uint64_t b = tile * kOwnTile, e = (b + kOwnTile < n) ? b + kOwnTile : n;
        if (threadIdx.x == 0) { sMinBlk = 0xFFFFFFFFu; sOutside = 0; sOutsideWave = 0; sDup = 0; sFar = 0; }
        for (uint32_t i = threadIdx.x; i < (2 * kOwnTile) / 32; i += kBlock) seen[i] = 0;
        __syncthreads();
        // The tile is read ONCE, sixteen independent loads per thread, and everything below works on the home slots in
        // registers (three dependent sweeps over the tile -- min, window test, ring test -- were 33 us of latency per
        // build whatever the size: most of a small relation's launch tail). Thread (wave w, lane l) holds, for k = 0, 1 and
        // r = 0..7, tuple b + (4 k + w) * 512 + 64 r + l: wavefront w's two ring tiles of 512 consecutive tuples (what
        // k_build_wave's wavefronts take), row r of 64 tuples in lane order.
        static_assert(kOwnTile == 2 * (kBlock / 64) * (int)kWvTileTuples && kWvTileTuples == 8 * 64, "sample layout: 2 ring tiles of 8 rows per wavefront");
        uint32_t h[16];
        uint32_t okBits = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint64_t i = b + (uint64_t)((q >> 3) * (kBlock / 64) + wave) * kWvTileTuples + (uint32_t)(q & 7) * 64u + lane;
            const bool ok = i < e;
            h[q] = (uint32_t)R[ok ? i : b];
            okBits |= (ok ? 1u : 0u) << q;
        }
        uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            h[q] = home32<HTM>(h[q], hshift, (uint32_t)mask);
            const uint32_t hb = ((okBits >> q) & 1u) ? h[q] >> kBlkShift : 0xFFFFFFFFu;
            lo = hb < lo ? hb : lo;
        }
        lo = wave_min_u32(lo);
        if (lane == 0 && lo != 0xFFFFFFFFu) atomicMin(&sMinBlk, lo);
        __syncthreads();
        const uint32_t wbase = sMinBlk > kBackBlocks ? sMinBlk - kBackBlocks : 0;
        uint32_t outside = 0, dup = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (!((okBits >> q) & 1u)) continue;
            const uint32_t hb = h[q] >> kBlkShift;
            outside += (hb - wbase >= kWinBlocks) ? 1u : 0u;
            // tuples whose home slot another tuple of the tile has too (out[3]): with tight locality a tile's home slots lie
            // within about its own length, so one bit per slot of twice that tells -- the share of duplicate keys, which
            // decides between the two ring builds (the compact table's forced rounds only pay where retry rounds are few)
            const uint32_t d = (h[q] - (sMinBlk << kBlkShift)) & (2 * kOwnTile - 1);
            dup += (atomicOr(&seen[d >> 5], 1u << (d & 31)) >> (d & 31)) & 1u;
        }
        // variant 3: wavefront tiles of kWvTileTuples consecutive tuples; the ring of kWvRingGran granules ends just
        // above the tile's highest home granule but never starts above its lowest (k_build_wave's rule)
        uint32_t outsideWave = 0, farRows = 0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint64_t sb = b + (uint64_t)(k * (kBlock / 64) + wave) * kWvTileTuples;
            if (sb >= e) continue;                                       // wave-uniform
            const bool whole = sb + kWvTileTuples <= e;
            uint32_t glo = 0xFFFFFFFFu, ghiInv = 0xFFFFFFFFu;
            // out[4]: rows of 64 consecutive tuples whose highest home slot lies above the lowest one of the row TWO rows
            // later -- disorder that reaches further than 64 positions, which is as far as the compact ring build's seam
            // zones reach (hj_build_wave.hip); the classic rings defer such stragglers, the compact ones would have to give up
            uint32_t prevMax2 = 0, prevMax1 = 0;                       // highest home slot of the rows two / one before (wave-uniform)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int q = k * 8 + r;
                if ((okBits >> q) & 1u) {
                    const uint32_t g = h[q] >> kWvGranShift;
                    glo = g < glo ? g : glo; ghiInv = ~g < ghiInv ? ~g : ghiInv;
                }
                if (whole) {                                             // whole rows only (all 64 lanes take part in the reductions)
                    const uint32_t rowMin = wave_min_u32(h[q]), rowMax = ~wave_min_u32(~h[q]);
                    farRows += (r >= 2 && prevMax2 > rowMin && lane == 0) ? 1u : 0u;
                    prevMax2 = prevMax1; prevMax1 = rowMax;
                }
            }
            glo = wave_min_u32(glo);
            const uint32_t top = ~wave_min_u32(ghiInv) + 1;
            uint32_t gbase = top > kWvRingGran ? top - kWvRingGran : 0;
            gbase = gbase < glo ? gbase : glo;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int q = k * 8 + r;
                const uint32_t g = h[q] >> kWvGranShift;
                outsideWave += (((okBits >> q) & 1u) && g - gbase >= kWvRingGran) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            outside += __shfl_down(outside, off, 64);
            outsideWave += __shfl_down(outsideWave, off, 64);
            dup += __shfl_down(dup, off, 64);
        }
        if (lane == 0 && farRows) atomicAdd(&sFar, farRows);
        if (lane == 0 && outside) atomicAdd(&sOutside, outside);
        if (lane == 0 && outsideWave) atomicAdd(&sOutsideWave, outsideWave);
        if (lane == 0 && dup) atomicAdd(&sDup, dup);
        __syncthreads();
        if (threadIdx.x == 0) { tot[0] += sOutside; tot[1] += (unsigned int)(e - b); tot[2] += sOutsideWave; tot[3] += sDup; tot[4] += sFar; }
        __syncthreads();
    }
    // Totals without atomics on shared words (256 workgroups x 5 atomic adds on five addresses were served one after the
    // other: half of this kernel's 25 us): every workgroup stores its five counts in a slot of its own, takes a ticket
    // (out[7]), and the workgroup that finishes last adds the slots up -- and, for build_common's pre-round, picks on the
    // device: it writes the variant word every build kernel enqueued behind this launch is gated on (no second launch),
    // and what the sample PREFERS (whether or not its kernels were enqueued this time) goes straight into pinned host
    // memory, where the next step's enqueue reads it without waiting: a store, not a copy in the stream.
    unsigned int* const slots = out + 8;
    if (threadIdx.x == 0) {
        for (int i = 0; i < 5; ++i) slots[8 * blockIdx.x + i] = tot[i];
        __threadfence();
        sLast = atomicAdd(&out[7], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!sLast) return;
    __threadfence();
    unsigned int f[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        unsigned int v = 0;
        for (uint32_t bk = threadIdx.x; bk < gridDim.x; bk += kBlock)
            v += __hip_atomic_load(&slots[8 * bk + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        f[i] = v;
    }
    if (threadIdx.x == 0) { sOutside = 0; sOutsideWave = 0; sDup = 0; sFar = 0; sMinBlk = 0; }
    __syncthreads();
    if (lane == 0) { atomicAdd(&sOutside, f[0]); atomicAdd(&sMinBlk, f[1]); atomicAdd(&sOutsideWave, f[2]); atomicAdd(&sDup, f[3]); atomicAdd(&sFar, f[4]); }
    __syncthreads();
    if (threadIdx.x == 0) {
        f[0] = sOutside; f[1] = sMinBlk; f[2] = sOutsideWave; f[3] = sDup; f[4] = sFar;
        for (int i = 0; i < 5; ++i) out[i] = f[i];
        if (pick.ctr) {
            const uint32_t pref = variant_for_sample(f[0], f[1], f[2], pick.canOwn, pick.canWave, pick.canCompact, f[3], f[4]);
            pick.ctr->preferred = pref;
            pick.ctr->variant = variant_among_allowed(pref, pick.allowedMask);
            if (pick.hostPreferred) __hip_atomic_store(pick.hostPreferred, (unsigned long long)pref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- host side ---------------------------------------------------------------
size_t own_queue_bytes(uint64_t rSize) { return (rSize + 64) * sizeof(DeferredEntry); }
size_t own_owner_bytes(uint64_t tableSize) { return ((tableSize >> kBlkShift) + 1) * sizeof(unsigned int); }
bool own_supported(uint64_t tableSize) { return tableSize >= (uint64_t)kWinSlots; }

hipError_t launch_sample_locality(const void* R, bool key32, uint64_t n, uint64_t tableSize, uint32_t hshift, uint32_t nSample,
                                  unsigned int* fitCount, hipStream_t s, bool htm, SamplePick pick, bool zeroed)
{
    if (!zeroed) {      // the totals and the ticket (build_common keeps the words inside Counters: its one memset has cleared them already)
        const hipError_t e = hipMemsetAsync(fitCount, 0, 8 * sizeof(unsigned int), s);
        if (e != hipSuccess) return e;
    }
    if (htm)        // the bucketised table's own hash ((key / 3) << 2: the keys spread 4/3 as wide as in the open-addressing table)
        hipLaunchKernelGGL((k_sample_locality<false, true>), dim3(nSample < 256 ? nSample : 256), dim3(kBlock), 0, s,
                           R, n, tableSize - 1, hshift, nSample, fitCount, pick);
    else if (key32)
        hipLaunchKernelGGL(k_sample_locality<true>, dim3(nSample < 256 ? nSample : 256), dim3(kBlock), 0, s,
                           R, n, tableSize - 1, hshift, nSample, fitCount, pick);
    else
        hipLaunchKernelGGL(k_sample_locality<false>, dim3(nSample < 256 ? nSample : 256), dim3(kBlock), 0, s,
                           R, n, tableSize - 1, hshift, nSample, fitCount, pick);
    return hipGetLastError();
}

hipError_t own_set_attributes()
{
    // hipFuncAttributeMaxDynamicSharedMemorySize is per device: hj_create calls this with its device current
    const void* ks[] = {reinterpret_cast<const void*>(k_build_own<false, false>), reinterpret_cast<const void*>(k_build_own<true, false>),
                        reinterpret_cast<const void*>(k_build_own<false, true>), reinterpret_cast<const void*>(k_build_own<true, true>),
                        reinterpret_cast<const void*>(k_build_own<false, false, true>)};
    for (const void* k : ks) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kWinSlots * sizeof(uint64_t));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// chunk geometry of the workgroup-window build: (chunks, tuples per chunk)
static void own_geometry(uint64_t n, int nCU, uint64_t* nChunksOut, uint64_t* chunkLenOut)
{
    // one chunk per resident workgroup (2 per CU: 76 KiB LDS each): a single wave of workgroups, no tail,
    // and the fewest chunk seams (measured: 512 chunks beat 768/1024/2048/4096 on MI355X)
    const int resident = 2 * (nCU > 0 ? nCU : 256);
    uint64_t rounds = n / ((uint64_t)resident * HJ_OWN_MIN_CHUNK);
    rounds = rounds < 1 ? 1 : rounds > HJ_OWN_MAX_ROUNDS ? HJ_OWN_MAX_ROUNDS : rounds;
    const uint64_t nChunks = (uint64_t)resident * rounds;
    uint64_t chunkLen = (n + nChunks - 1) / nChunks;
    chunkLen = (chunkLen + kOwnTile - 1) / kOwnTile * kOwnTile;
    if (chunkLen < (uint64_t)kOwnTile * 4) chunkLen = (uint64_t)kOwnTile * 4;
    *chunkLenOut = chunkLen;
    *nChunksOut = (n + chunkLen - 1) / chunkLen;
}
// htm: the conflict list of the window build = one slice per chunk + a last slice for the deferred phase's conflicts
WaveSlices own_conflict_layout(uint64_t n, int nCU, void* countsBuf)
{
    uint64_t nChunks, chunkLen;
    own_geometry(n, nCU, &nChunks, &chunkLen);
    return WaveSlices{(uint32_t)nChunks + 1, (uint32_t)chunkLen, static_cast<const uint32_t*>(countsBuf)};
}
size_t own_conflict_bytes(uint64_t n, int nCU)
{
    uint64_t nChunks, chunkLen;
    own_geometry(n, nCU, &nChunks, &chunkLen);
    return (size_t)(nChunks * chunkLen + n + 64) * sizeof(uint64_t);
}
size_t own_conflict_count_bytes(uint64_t n, int nCU)
{
    uint64_t nChunks, chunkLen;
    own_geometry(n, nCU, &nChunks, &chunkLen);
    return (size_t)(nChunks + 2) * sizeof(uint32_t);
}

hipError_t launch_build_own(const void* R, bool key32, uint64_t n, uint32_t hshift, uint64_t* table,
                            uint64_t tableSize, uint32_t probeLen, uint64_t idxBase, ShardCheck sc, int nCU, void* ownerBuf,
                            void* queueBuf, uint32_t* deferCounts, Counters* ctr, Gate gate, int parts,
                            hipEvent_t evPhaseA, hipStream_t s, const KernelEvents* kev, uint64_t* htmConflicts, uint32_t* htmCounts)
{
    const bool htm = htmConflicts != nullptr;
    if (htm && (key32 || probeLen != 3 || sc.mask || hshift)) return hipErrorInvalidValue;
    const uint32_t numBlocks = (uint32_t)(tableSize >> kBlkShift);
    uint64_t nChunks, chunkLen;
    own_geometry(n, nCU, &nChunks, &chunkLen);
    hipError_t e;
    if (parts & 1) {
    if ((e = hipMemsetAsync(ownerBuf, 0, own_owner_bytes(tableSize), s)) != hipSuccess) return e;
    if (nChunks > kOwnMaxChunks) return hipErrorInvalidValue;
    if (htm && (e = hipMemsetAsync(htmCounts, 0, own_conflict_count_bytes(n, nCU), s)) != hipSuccess) return e;
    const unsigned grid = (unsigned)nChunks;
    if (kev && (e = hipEventRecord(kev->before, s)) != hipSuccess) return e;
#define HJ_OWN_LAUNCH(K32, CHK, HTM)                                                                                 \
    hipLaunchKernelGGL((k_build_own<K32, CHK, HTM>), dim3(grid), dim3(kOwnThreads), kWinSlots * sizeof(uint64_t), s,  \
                       R, n, chunkLen, table, tableSize - 1, hshift, probeLen, idxBase, sc,                            \
                       static_cast<unsigned int*>(ownerBuf), static_cast<DeferredEntry*>(queueBuf), deferCounts, ctr, gate, \
                       htmConflicts, htmCounts, (uint32_t)chunkLen)
    if (htm) HJ_OWN_LAUNCH(false, false, true);
    else if (sc.mask) { if (key32) HJ_OWN_LAUNCH(true, true, false); else HJ_OWN_LAUNCH(false, true, false); }   // the instances that count foreign tuples
    else { if (key32) HJ_OWN_LAUNCH(true, false, false); else HJ_OWN_LAUNCH(false, false, false); }
#undef HJ_OWN_LAUNCH
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (kev && (e = hipEventRecord(kev->after, s)) != hipSuccess) return e;
    if (evPhaseA && (e = hipEventRecord(evPhaseA, s)) != hipSuccess) return e;
    }
    if (!(parts & 2)) return hipSuccess;
    hipLaunchKernelGGL(k_finalize_range, dim3(1), dim3(64), 0, s, ctr, numBlocks, tableSize, gate);
    hipLaunchKernelGGL(k_clear_unowned, dim3(2048), dim3(kBlock), 0, s, table,
                       static_cast<const unsigned int*>(ownerBuf), ctr, numBlocks, tableSize, gate);
    // phase B: `parts` workgroups per slice (about 4096 in all: the walks are chains of dependent global atomics)
    const uint32_t defParts = nChunks >= 4096 ? 1u : (uint32_t)(4096 / nChunks);
    const dim3 gDef((unsigned)(nChunks * defParts));
    if (htm)
        hipLaunchKernelGGL(k_build_deferred<true>, gDef, dim3(kBlock), 0, s, static_cast<const DeferredEntry*>(queueBuf), deferCounts,
                           (uint32_t)chunkLen, defParts, table, tableSize - 1, hshift, probeLen, ctr, gate, htmConflicts + nChunks * chunkLen,
                           htmCounts + nChunks);
    else
        hipLaunchKernelGGL(k_build_deferred<false>, gDef, dim3(kBlock), 0, s, static_cast<const DeferredEntry*>(queueBuf), deferCounts,
                           (uint32_t)chunkLen, defParts, table, tableSize - 1, hshift, probeLen, ctr, gate, nullptr, nullptr);
    return hipGetLastError();
}

}  // namespace hj

// hj_device.h -- shared declarations between the HIP kernels (hj_kernels.hip,
// hj_prj.hip) and the C-ABI implementation (hj_api.hip). gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hj {

// A table slot holds (inputIndex << 32 | key32). All-ones is "empty" so that a
// 64-bit atomicMin implements index priority (smaller input index wins).
constexpr uint64_t kEmpty = ~0ull;
// Slots past tableSize that always stay empty: the probe walk does not wrap
// (NoCCHashBuild.hpp:74-75 does curSlot++ without & tableMask).
constexpr uint32_t kTableSlack = 16;

// Home slot of a key: (key >> hshift) & (tableSize - 1). hshift = 0 is the reference's hash (NoCCHashBuild.hpp:41);
// a radix shard of a multi-GPU join holds only keys with the same low log2(shards) bits, which therefore carry no
// information inside the shard and are shifted out of the slot number (hshift = log2(shards)). The slot still stores
// the whole key.
__host__ __device__ inline uint64_t home_slot(uint32_t key, uint32_t hshift, uint64_t mask) { return (uint64_t)(key >> hshift) & mask; }

// Home slot in the bucketised table of --algo htm (HTMHashBuild.hpp:41-45,176): a bucket is 4 consecutive 8-byte
// slots (three tuples + one word of count / overflow link = 32 bytes, one HBM sector), bucket = (key / 3) &
// (numBuckets - 1); mask = 4 * numBuckets - 1. The three tuple slots are filled by the same index-priority protocol as
// the open-addressing table with a probe budget of 3: every tuple of a bucket has the same home slot, so the bucket ends
// up holding the three lowest-indexed tuples in index order and the rest run out of budget = the reference's conflicts.
__host__ __device__ inline uint64_t home_slot_htm(uint32_t key, uint64_t mask) { return ((uint64_t)(key / 3u) << 2) & mask; }
template <bool HTM>
__host__ __device__ inline uint32_t home32(uint32_t key, uint32_t hshift, uint32_t mask32)
{
    if constexpr (HTM) return ((key / 3u) << 2) & mask32;
    else return (key >> hshift) & mask32;
}

// A lane's rank among the set lanes of a 64-bit lane mask (a ballot, or a per-lane mask of peers): the number of set bits
// below the lane. v_mbcnt_lo + v_mbcnt_hi -- two instructions, the mask may stay in scalar registers -- instead of two
// ANDs with a "lanes below me" mask and two popcounts.
__device__ __forceinline__ uint32_t lane_rank(unsigned long long m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
#else
    (void)m; return 0u;         // host pass of hipcc: parsed, never called
#endif
}

// Optional shard-membership check riding on build and probe (hj_set_shard_check): a tuple is "foreign" when its
// destination digit ((key - bias) >> shift) & mask differs from id. mask = 0 (and id = 0) switches it off at no
// cost in branches: every tuple's digit is then 0.
struct ShardCheck { uint32_t mask, shift, bias, id; };
__host__ __device__ inline bool is_foreign(uint32_t key, const ShardCheck& sc) { return (((key - sc.bias) >> sc.shift) & sc.mask) != sc.id; }

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kWave = 64;

// words launch_sample_locality works on: 8 (five totals, [7] = the ticket) + 8 per workgroup of the sampler (its own counts)
constexpr int kSampleBlocks = 256;
constexpr int kSampleWords = 8 + 8 * kSampleBlocks;

// Device-resident counters, zeroed at the start of a build. One cache line
// apart is not needed: each is touched once per wavefront at kernel end.
struct Counters {
    unsigned long long conflicts;
    unsigned long long conflictSum;
    unsigned long long inputSum;
    unsigned long long matches;
    unsigned long long tableSumHalf;
    unsigned long long tableSumFull;
    unsigned long long badKeys;      // tuples with payload bits set or value 0
    unsigned long long prjMatches;
    unsigned long long prjChecksum;
    unsigned long long prjOverflowParts; // partitions joined in several LDS blocks
    unsigned long long deferred;         // variant 2: tuples finished by the global-atomic phase
    // Variant 2 touches only the table blocks some tuple can reach; everything else is neither cleared
    // nor read. usedLoInv / usedHi1 collect (max of ~block) and (max of block+1) over claimed blocks
    // and deferred targets; k_finalize_range turns them into the valid SLOT range
    // [validLo, validHiEx): a home slot outside it cannot match anything (k_probe skips it), slots in
    // [validLo, validHiEx + 512) hold defined values.
    unsigned long long usedLoInv, usedHi1;
    unsigned long long validLo, validHiEx;
    unsigned long long foreign;      // tuples of the build / probe inputs that fail the shard check (ShardCheck)
    // Variant 3 (hj_build_wave.hip): the stretch of the table its wavefronts own and write whole, [ownLo, ownHiEx)
    unsigned long long ownLo, ownHiEx;
    // Build kernel the device-side locality pre-round picked (hj_params.buildVariant 0): written by the sampler's last workgroup (k_sample_locality),
    // read through the Gate of every build kernel enqueued behind it, reported as hj_result.buildVariant
    unsigned long long variant;
    // what the pre-round would have picked had every variant been enqueued (k_sample_locality). hj_build_dev only enqueues
    // the kernels of the variant the PREVIOUS build of the context preferred (+ the classic rings behind the compact
    // ones, + global atomics: always correct); the pick is taken among those, and this word tells the host what to
    // enqueue next time (the kernel stores it into pinned host memory as well, read without waiting at the next build)
    unsigned long long preferred;
    // --algo htm (hj_htm.hip): overflow buckets linked, sum of the tuples they hold
    unsigned long long htmOverflowBuckets, htmOverflowSum;
    // raised by the LDS chain phase (k_htm_chain_lds) or by the routing of the deferred phase's conflicts when an input
    // does not fit them: the host then redoes the build without routing and chains with the generic kernels
    unsigned long long htmChainBail;
    // PRJ, histogram-free partitioning (hj_prj.hip): set to 1 by the scatter kernel that finds a fragment too small;
    // the rest of that path then returns at once and the exact (histogram) path, gated on this word, runs instead
    unsigned long long prjFallback;
    // Open-addressing table formats (k_build_wave<COMPACT>, hj_build_wave.hip). tableFormat says what the table buffer
    // holds after a build: kFormatSlots8 = one 8-byte slot (index << 32 | key) per table slot, all ones = empty (every
    // build but the compact one); kFormatKeys4 = one 4-byte KEY per table slot, 0xFFFFFFFF = empty (the index words only
    // order the inserts; the compact build settles every order inside its LDS rings and never writes them out). k_probe,
    // k_table_sums and hj_export_table read the word. compactFail: raised by the compact build when it meets something
    // only the classic builds can handle; k_wave_decide then resets the counters and hands over to the classic build.
    unsigned long long tableFormat, compactFail;
    // the locality pre-round's sample of hj_build_dev (k_sample_locality: launch_sample_locality's eight words, [7] = its
    // ticket) -- inside this struct so that the one memset at the start of a build clears them too
    unsigned int fit[kSampleWords];
    // The sums every wavefront contributes to at the END of a kernel (above: conflicts, conflictSum, inputSum, matches,
    // badKeys, prjMatches, prjChecksum, deferred, foreign, and the two maxima usedLoInv / usedHi1) are collected in 64
    // shards, each on a 128-byte line of its own, picked by wavefront number. Thousands of wavefronts finish together, and
    // their atomics on ONE address are served one after the other: measured 95 us at the end of k_probe (8192 wavefronts)
    // and 67 us at the end of k_build_wave whatever the size -- most of those kernels at 2^22 tuples, 17 % / 4 % at 2^27,
    // and 70 us of the deferred phase at 2^30. The fields above hold the totals only after fold_counter_shards() (host,
    // after the copy back); on the device nothing reads the sums, and the finalize kernels fold the two maxima themselves.
    struct alignas(128) Shard {
        unsigned long long conflicts, conflictSum, inputSum, matches, badKeys, prjMatches, prjChecksum, deferred, foreign;
        unsigned long long usedLoInv, usedHi1;
    };
    static constexpr int kShards = 64;
    Shard shard[kShards];
};

// minimum over the wavefront, result wave-uniform: four DPP steps inside each row of 16, then the four rows
__device__ __forceinline__ uint32_t wave_umin(uint32_t v)
{
    auto step = [](uint32_t x, const int ctrl) {
        uint32_t o;
        switch (ctrl) {     // the control word must be an immediate
            case 0: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
            case 1: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
            case 2: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, true); break;  // row_half_mirror
            default: o = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, true); break; // row_mirror
        }
        return o < x ? o : x;
    };
    v = step(v, 0); v = step(v, 1); v = step(v, 2); v = step(v, 3);
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// the shard of the calling wavefront
__device__ inline Counters::Shard* counter_shard(Counters* ctr)
{
    return &ctr->shard[(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (Counters::kShards - 1)];
}
// host, on a copy of the counters: totals = what was added directly + the shards
inline void fold_counter_shards(Counters* h)
{
    for (int i = 0; i < Counters::kShards; ++i) {
        Counters::Shard& s = h->shard[i];
        h->conflicts += s.conflicts; h->conflictSum += s.conflictSum; h->inputSum += s.inputSum; h->matches += s.matches;
        h->badKeys += s.badKeys; h->prjMatches += s.prjMatches; h->prjChecksum += s.prjChecksum; h->deferred += s.deferred;
        h->foreign += s.foreign;
        h->usedLoInv = s.usedLoInv > h->usedLoInv ? s.usedLoInv : h->usedLoInv;
        h->usedHi1 = s.usedHi1 > h->usedHi1 ? s.usedHi1 : h->usedHi1;
        s = Counters::Shard{};
    }
}

// Device-side choice between the build variants (hj_build_dev must stay asynchronous: no host read-back). The host
// enqueues the kernels of EVERY candidate variant; each looks at the word the pre-round wrote and returns at once
// unless it is the chosen one. word = nullptr: no gate (the variant was fixed on the host).
// alt: a second value that opens the gate (the pre-pass shared by the compact and the classic ring build)
struct Gate { const unsigned long long* word; unsigned long long want; unsigned long long alt = ~0ull; };
__device__ inline bool gate_closed(const Gate& g)
{
    if (g.word == nullptr) return false;
    const unsigned long long v = *g.word;
    return v != g.want && v != g.alt;
}
constexpr unsigned long long kFormatSlots8 = 0, kFormatKeys4 = 1;

// A tuple that left its LDS window (variants 2 and 3): the slot it had reached and (index << 32 | key)
struct DeferredEntry { uint64_t pos; uint64_t packed; };
// HIP events recorded right before and right after ONE kernel launch (the dominant build kernel of a variant): its device
// time for the roofline, without the pre-pass and the gated-off launches of the other variants around it
struct KernelEvents { hipEvent_t before, after; };

// ---- launch wrappers (defined in hj_kernels.hip) ---------------------------
// Inputs come in two element formats: 8-byte DataGen tuples (key32 = false; value = key, payload bits must be 0)
// or bare 32-bit keys (key32 = true; what the multi-GPU exchange delivers). Index of element i = idxBase + i.
// fullRange != nullptr: also marks the whole table valid (variant 1 clears and may touch all of it): one launch less
void launch_fill_empty(uint64_t* table, uint64_t nSlots, Gate gate, hipStream_t s, Counters* fullRange = nullptr, uint64_t tableSize = 0);
void launch_build_atomic_min(const void* R, bool key32, uint64_t n, uint64_t* table, uint64_t tableSize, uint32_t hshift,
                             uint32_t probeLen, uint64_t idxBase, ShardCheck sc, Counters* ctr, Gate gate, hipStream_t s);
// the probe and the checksums read Counters::tableFormat on the device: either table format, one launch
void launch_probe(const void* S, bool key32, uint64_t n, const uint64_t* table, uint64_t tableSize, uint32_t hshift,
                  uint32_t probeLen, ShardCheck sc, Counters* ctr, hipStream_t s);
void launch_table_sums(const uint64_t* table, uint64_t tableSize, uint64_t halfSlots, Counters* ctr, hipStream_t s);
void launch_zipf_lookup(const int* raw, uint64_t n, const double* lut, const uint32_t* alphabet, uint32_t alphabetSize,
                        uint64_t* out, hipStream_t s);
// Marks the whole table valid (variant 1 clears and may touch all of it).
void launch_set_full_range(uint64_t tableSize, Counters* ctr, Gate gate, hipStream_t s);
// multi-GPU destination split (defined in hj_prj.hip: one order-preserving radix pass, tuples in, keys out);
// destination = (key >> digitShift) & (nShards - 1)
size_t shard_work_bytes(uint64_t n, uint32_t nShards);
hipError_t launch_shard_hist(const uint64_t* in, uint64_t n, uint32_t nShards, uint32_t digitShift, void* work,
                             unsigned long long* counts, hipStream_t s);
hipError_t launch_shard_scatter_ordered(const uint64_t* in, uint64_t n, uint32_t nShards, uint32_t digitShift, void* work,
                                        uint32_t* outKeys, hipStream_t s);

// ---- ownership build (defined in hj_build_own.hip) ---------------------------
size_t own_queue_bytes(uint64_t rSize);
size_t own_owner_bytes(uint64_t tableSize);
bool   own_supported(uint64_t tableSize);
// fitCount[0] = sampled tuples outside variant 2's window, [1] = tuples sampled, [2] = outside variant 3's ring,
// [3] = sampled tuples that share their home slot with another tuple of their tile (duplicate keys),
// [4] = sampled rows of 64 tuples with disorder beyond 64 positions (8 words in all)
// pick.ctr != nullptr: the workgroup of the sample that finishes last decides on the device -- ctr->preferred = the variant
// the sample asks for (variant_for_sample: the thresholds the host applies), ctr->variant = the best one among the
// variants whose kernels are enqueued behind the sample (allowedMask: bit v set = variant v is; bit 1, global atomics,
// always is), *hostPreferred (pinned host memory, may be null) = preferred, stored by the kernel itself.
struct SamplePick {
    Counters* ctr = nullptr;
    unsigned long long* hostPreferred = nullptr;
    uint32_t allowedMask = 0x1E;
    bool canOwn = false, canWave = false, canCompact = false;
};
// zeroed: fitCount's eight words are zero already (no memset of its own)
hipError_t launch_sample_locality(const void* R, bool key32, uint64_t n, uint64_t tableSize, uint32_t hshift, uint32_t nSample,
                                  unsigned int* fitCount, hipStream_t s, bool htm = false,    // htm: the bucketised table's hash
                                  SamplePick pick = SamplePick{}, bool zeroed = false);
// the best enqueued variant for a preferred one: itself if enqueued, else the next looser build that is, else (nothing
// looser is enqueued: a context whose relation lost its locality since the last build) the tightest LDS build that is --
// rings and window are correct on any input (what falls outside goes through their deferred phases: global atomics,
// slow for that one step); the compact rings only with the classic ones behind them
__host__ __device__ inline uint32_t variant_among_allowed(uint32_t preferred, uint32_t allowedMask)
{
    for (uint32_t v = preferred; v >= 1; --v)
        if ((allowedMask >> v) & 1u) {
            if (v == 3 && preferred == 2) continue;      // loose locality: global atomics before the rings, if they are there
            return v;
        }
    for (uint32_t v = preferred + 1; v <= 3; ++v)
        if ((allowedMask >> v) & 1u) return v;
    return ((allowedMask >> 3) & 1u) ? 3u : 1u;
}
// variant worth taking for a sample (outside variant 2's window, tuples seen, outside variant 3's ring)
// dup = sampled tuples that share their home slot with another tuple of their tile: rings with few duplicate keys take the
// compact table (4: 2.6 against 3.4 ms build at 2^30 on unique keys, and a 4-byte probe), rings with many keep the classic
// one (3): on `uniform` (37 % of the tuples repeat a key) the build is bound by the vector work of its retry rounds, the
// compact build adds forced rounds to it (4.0 against 3.6 ms) and the whole step comes out even.
__host__ __device__ inline uint32_t variant_for_sample(uint64_t outOwn, uint64_t seen, uint64_t outWave, bool canOwn, bool canWave,
                                                      bool canCompact = false, uint64_t dup = 0, uint64_t farRows = 0)
{
    // farRows: sampled rows of 64 tuples that reach above the row two further on (disorder beyond 64 positions: more than
    // the compact build's seam zones cover -- it would start, give up and hand over to the classic rings)
    if (canWave && outWave * 128 <= seen) return (canCompact && dup * 8 <= seen && farRows == 0) ? 4 : 3;
    // the workgroup window while it can take at least a quarter of the tuples itself: what it defers is finished by global
    // atomics at about their own pace (2^27, local_shuffle, build + probe: W = 2^12 defers 36 % -> 2.8 ms against 5.8 ms
    // for the global-atomic build; 2^13: 71 % -> 4.7 / 5.8; 2^14: 88 % -> 5.6 / 5.9; 2^16: 97 % -> 6.6 / 5.9 -- since the
    // deferred queue is sliced per workgroup; with one global queue counter the window lost from 8 % on)
    if (canOwn && outOwn * 4 <= seen * 3) return 2;
    return 1;
}
hipError_t own_set_attributes();          // per device, at hj_create
// phase A (LDS window) -> clear of unowned blocks -> phase B (deferred tuples).
// Writes every table slot exactly once: no separate launch_fill_empty needed.
hipError_t launch_build_own(const void* R, bool key32, uint64_t n, uint32_t hshift, uint64_t* table,
                            uint64_t tableSize, uint32_t probeLen, uint64_t idxBase, ShardCheck sc, int nCU, void* ownerBuf,
                            void* queueBuf, uint32_t* deferCounts, Counters* ctr, Gate gate, int parts,
                            hipEvent_t evPhaseA, hipStream_t s, const KernelEvents* kev = nullptr,
                            uint64_t* htmConflicts = nullptr, uint32_t* htmCounts = nullptr);   // parts: 1 = phase A (up to evPhaseA), 2 = the rest, 3 = both
// htmConflicts != nullptr: the bucketised table of --algo htm through the workgroup window (tuples only, probeLen 3): the
// tuples that find their bucket full are listed per chunk, plus one last slice for the deferred phase's (own_conflict_layout)

// deferCounts: kOwnMaxChunks words (the deferred queue is sliced by phase-A workgroup; entries per slice)
constexpr uint32_t kOwnMaxChunks = 8192;

// ---- wavefront-private build (defined in hj_build_wave.hip) ------------------
// geometry the locality sampler (k_sample_locality) needs to predict what k_build_wave would defer
constexpr uint32_t kWvGranShift = 7;      // retire granule: 128 slots = 1 KiB
constexpr uint32_t kWvRingGran = 8;       // ring = 8 granules = 1024 slots = 8 KiB per wavefront
constexpr uint32_t kWvTileTuples = 512;   // tuples per wavefront tile
size_t wave_lds_bytes();
bool   wave_supported(uint64_t tableSize);
size_t wave_bounds_bytes(int nCU);
size_t wave_queue_bytes(uint64_t n, int nCU);   // deferred queue: one slice per chunk
// bounds pre-pass -> k_build_wave -> valid range + edge fill -> phase B. queueBuf: own_queue_bytes(n), used as one
// slice per chunk (a wavefront's deferred tuples go to ITS slice: no atomics in the kernel).
// parts: kWavePre = the seam / bounds pre-pass, kWaveMain = the build kernel (then evPhaseA), kWaveTail = valid range,
// edge fill and the deferred phase. mode kWaveCompact: the compact build (4-byte table, no deferred phase) -- its main
// part is k_build_wave<COMPACT> + the seam check + k_wave_decide, which on failure resets the counters and sets
// Counters::variant = fallbackVariant so that the classic build enqueued behind it (gated on that word) redoes the table;
// its tail is the edge fill alone. The pre-pass is the same for both modes (gate it with alt).
constexpr int kWavePre = 1, kWaveMain = 2, kWaveTail = 4, kWaveAll = 7;
constexpr int kWaveClassic = 0, kWaveCompact = 1;
hipError_t launch_build_wave(const void* R, bool key32, uint64_t n, uint32_t hshift, uint64_t* table, uint64_t tableSize,
                             uint32_t probeLen, uint64_t idxBase, ShardCheck sc, int nCU, void* boundsBuf, void* queueBuf,
                             Counters* ctr, Gate gate, int parts, hipEvent_t evPhaseA, hipStream_t s,
                             uint64_t* htmConflicts = nullptr, int mode = kWaveClassic, uint32_t fallbackVariant = 3,
                             const KernelEvents* kev = nullptr, bool htmRoute = false);
// htmRoute: the deferred phase files its conflicts under the chunk that owns their bucket (the LDS chain phase needs that)
const uint32_t* wave_bounds_ptr(int nCU, const void* boundsBuf);     // chunk c owns granules [bounds[c], bounds[c + 1])
bool wave_compact_supported(uint64_t tableSize, uint32_t probeLen);
void launch_set_variant(Counters* ctr, uint32_t v, hipStream_t s);
// htmConflicts != nullptr: the bucketised table of --algo htm (home_slot_htm, probeLen must be 3, tuples only); every
// tuple that runs out of budget is appended as (index << 32 | key) to its chunk's slice of htmConflicts (slices and
// their counts as wave_conflict_layout describes)
struct WaveSlices { uint32_t nChunks, sliceLen; const uint32_t* counts; };
WaveSlices own_conflict_layout(uint64_t n, int nCU, void* countsBuf);
size_t own_conflict_bytes(uint64_t n, int nCU);
size_t own_conflict_count_bytes(uint64_t n, int nCU);
WaveSlices wave_conflict_layout(uint64_t n, int nCU, void* boundsBuf);
size_t wave_conflict_bytes(uint64_t n, int nCU);

// ---- bucketised table of --algo htm (defined in hj_htm.hip) -----------------
uint32_t htm_num_buckets(uint64_t rSize);         // nextpow2(rSize / 3 + 1), HTMHashBuild.hpp:61-62
hipError_t launch_htm_build_global(const uint64_t* R, uint64_t n, uint32_t sliceLen, uint32_t nSlices, uint64_t* table,
                                   uint64_t tableSlots, uint64_t idxBase, uint64_t* conflicts, uint32_t* ccounts, Counters* ctr,
                                   hipStream_t s);
// per-bucket conflict counts -> ovfCount, overflow buckets needed per bucket -> groups (to be scanned in place)
hipError_t launch_htm_count(const uint64_t* conflicts, const uint32_t* ccounts, uint32_t nSlices, uint32_t sliceLen,
                            uint32_t numBuckets, unsigned int* ovfCount, uint32_t* groups, hipStream_t s);
hipError_t launch_htm_chains(const uint64_t* conflicts, const uint32_t* ccounts, uint32_t nSlices, uint32_t sliceLen,
                             uint64_t* table, uint32_t numBuckets, const unsigned int* ovfCount, const uint32_t* ovfBase,
                             uint64_t* overflow, uint64_t overflowCapBuckets, Counters* ctr, hipStream_t s);
// the chain phase in LDS, after the ring build with routed conflicts (hj_htm.hip): count -> partGroups[nSlices * parts] (overflow
// buckets per part; scan it, one word more for the total) and info (htm_chain_info_words words); fill builds the chains
uint32_t htm_chain_parts(uint32_t sliceLen);
size_t htm_chain_info_words(uint32_t nSlices, uint32_t sliceLen);
hipError_t launch_htm_chain_count(const uint64_t* conflicts, const uint32_t* ccounts, const uint32_t* bounds, uint32_t nSlices,
                                  uint32_t sliceLen, uint32_t numBuckets, uint32_t* partGroups, uint32_t* info, Counters* ctr, hipStream_t s);
hipError_t launch_htm_chain_fill(const uint64_t* conflicts, uint32_t nSlices, uint32_t sliceLen, uint32_t numBuckets, const uint32_t* partBase,
                                 const uint32_t* info, uint64_t* table, uint64_t* overflow, Counters* ctr, hipStream_t s);
void launch_htm_probe(const uint64_t* S, uint64_t n, const uint64_t* table, uint32_t numBuckets, const uint64_t* overflow,
                      Counters* ctr, hipStream_t s);
void launch_htm_sums(const uint64_t* table, uint32_t numBuckets, const uint64_t* overflow, Counters* ctr, hipStream_t s);
// exclusive scan of a uint32 array in place (defined in hj_prj.hip); sums: ceil(n / 4096) + 1 words of workspace
size_t scan_workspace_words(uint64_t n);
hipError_t launch_exclusive_scan_u32(uint32_t* data, uint64_t n, uint32_t* sums, hipStream_t s);

// ---- PRJ (defined in hj_prj.hip) -------------------------------------------
// Fragment geometry of the histogram-free partitioning of ONE relation (hj_prj.hip, "histogram-free partitioning"):
// pass 1 cuts the relation into C1 chunks and writes bin b of chunk c to the fragment (b * C1 + c) of cap1 key slots;
// pass 2 cuts every pass-1 partition (C1 fragments) into C2 chunks and writes to fragments of cap2 slots; a final
// partition is C2 fragments. C = 0: the relation takes the exact path only.
struct PrjFrag {
    uint32_t C1, cap1, chunkLen1;
    uint32_t C2, cap2, log2C2;
};
struct PrjPlan {
    uint32_t radixBits;   // total
    uint32_t bits1, bits2;
    bool     optimistic;               // try the histogram-free path first (both relations qualify)
    PrjFrag  fragR, fragS;
    uint64_t cnt1Entries, cnt2EntriesR, cnt2EntriesS;   // fragment counters in the workspace
    uint64_t maxChunks1, maxChunks2;   // chunk descriptors per pass (upper bounds over both relations)
    uint64_t histEntries, scanBlocks;  // histogram / block-sum entries: the larger need of R's and S's layouts in
                                       // either pass (the chunk length, hence the chunk count, is NOT monotone in
                                       // the relation size: the smaller relation can need the larger histogram)
    size_t   workspaceBytes;           // everything below, excluding tuple buffers
};
// Sizes the workspace for (nR, nS): each relation is laid out with its own chunk length (run_pass), so every
// region is the maximum over the two relations' layouts.
// mode (hj_params.prjMode): 0 = histogram-free path for large relations, 1 = exact path only, 2 = histogram-free path
// at any size it can be laid out for (tests)
PrjPlan prj_plan(uint64_t nR, uint64_t nS, uint32_t radixBits, uint32_t mode = 0);
// histogram entries the passes over ONE relation of n tuples write (what run_pass memsets and scans)
uint64_t prj_hist_entries_needed(uint64_t n, uint32_t radixBits);
struct PrjBuffers {
    uint64_t* tmpA;      // max(nR,nS) tuples
    uint64_t* partR;     // nR tuples (final partitioned R)
    uint64_t* partS;     // nS tuples
    void*     work;      // plan.workspaceBytes
};
// Enqueues partition(R), partition(S) and the per-partition LDS join.
// evPartDone (may be null) is recorded between partitioning and join.
hipError_t launch_prj(const PrjPlan& plan, const PrjBuffers& buf,
                      const uint64_t* R, uint64_t nR, const uint64_t* S, uint64_t nS, int nCU,
                      Counters* ctr, hipEvent_t evPartDone, hipEvent_t evScatter0, hipEvent_t evScatter1, hipStream_t s);
// evScatter0/1 (may be null): recorded around the pass-1 scatter of R, PRJ's dominant kernel
hipError_t prj_set_attributes();          // per device, at hj_create

}  // namespace hj

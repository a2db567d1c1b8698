// hj_build_wave.hip -- build variant 3: wavefront-private LDS windows over statically owned slot ranges.
//
// What it replaces: HOT LOOP 1 of the reference (NoCCHashBuild.hpp:37-62 / AtomicHashBuild.hpp:37-67, and the
// TSX group insert of HTMHashBuild.hpp:157-238) for inputs whose locality is TIGHT -- the reference's own
// default, `--shuffleRange 16`, and anything up to a few hundred positions of disorder. Variant 2
// (hj_build_own.hip) keeps a 64 KiB window per 512-thread workgroup, claims table blocks with global atomics and
// synchronises its eight wavefronts five times per tile; rocprof showed it neither HBM nor issue bound but
// waiting (51 % of the wave-cycles parked at barriers and on dependent LDS round trips, profiles/r01_*). Here
// nothing is shared between wavefronts, so nothing has to be waited for:
//
//   * R is cut into contiguous chunks, one per wavefront: as many as there are resident wavefronts (16 per CU), and
//     for large relations up to eight times as many (wave_chunk_len: later workgroups take over from the ones that
//     finish, which balances the wavefronts' uneven progress). The chunk's slot range
//     [bounds[c], bounds[c+1]) is fixed BEFORE the build by a pre-pass that looks at the first 64 tuples of
//     every chunk (k_wave_bounds): start = lowest home slot among them, made monotone by a prefix maximum.
//     A wavefront owns its range outright -- no claims, no owner table, no atomics on HBM.
//   * The wavefront walks its chunk in tiles of 64 x PER tuples with a ring of 2^WINLOG slots (8 KiB) of ITS
//     range in LDS. Inserts run the index-priority protocol of hj_kernels.hip with LDS atomics: the PER
//     home-slot attempts of a tile are issued back to back (independent round trips), whatever fails goes
//     through a compacted per-wavefront retry queue in dense rounds (as in variant 2).
//   * When the tile's lowest home slot moves on, the ring's tail leaves for HBM in 1 KiB granules (one
//     16-byte store per lane), empties included: every slot of the range is written exactly once, in order,
//     by its owner. At the end of the chunk the rest of the range is written the same way.
//   * A tuple that cannot be handled inside the range and the ring -- a straggler across a chunk seam, a key
//     far from its neighbours, a probe walk that leaves the range -- is "aborted" into the global deferred
//     queue with the slot it reached, and k_build_deferred (hj_build_own.hip) finishes it with global atomics
//     after this kernel: the protocol is confluent, so WHICH tuples take that road never changes the table.
//     On inputs without locality nearly every tuple takes it (correct, slow); hj_api.hip only picks this
//     variant when a sample of R says the ring will do.
//
// All integer work on 8-byte tuples (or bare 32-bit keys); HBM traffic = R read once + every reachable slot
// written once = the algorithmic 16 bytes per tuple of SURVEY.md 8d. No MFMA.

#include "hj_device.h"

#include <type_traits>

namespace hj {

#ifndef HJ_WV_THREADS
#define HJ_WV_THREADS 256
#endif
constexpr int kWvThreads = HJ_WV_THREADS;           // 4 wavefronts per workgroup; they never synchronise
constexpr int kWvWaves = kWvThreads / 64;
constexpr uint32_t kGranShift = kWvGranShift;       // retire granule: 128 slots = 1 KiB = 64 lanes x 16 bytes
constexpr uint32_t kGranSlots = 1u << kGranShift;
constexpr uint32_t kNone = 0xFFFFFFFFu;

#ifndef HJ_WV_PER
#define HJ_WV_PER 8
#endif
#ifndef HJ_WV_WINLOG
#define HJ_WV_WINLOG 10
#endif
#ifndef HJ_WV_QCAP
#define HJ_WV_QCAP 128
#endif
#ifndef HJ_WV_PF
#define HJ_WV_PF 1
#endif
#ifndef HJ_WV_WPE
#define HJ_WV_WPE 4                                 // wavefronts per SIMD the registers must allow (16 per CU: what the rings' LDS allows)
#endif
#ifndef HJ_WV_WAVES_PER_CU
#define HJ_WV_WAVES_PER_CU 16                       // resident wavefronts per CU (what the rings' LDS allows)
#endif
#ifndef HJ_WV_MAX_ROUNDS
#define HJ_WV_MAX_ROUNDS 8                          // chunks = resident wavefronts x rounds (wave_chunk_len below)
#endif
#ifndef HJ_WV_PRIO
#define HJ_WV_PRIO 2                                // 2: issue priorities rotate in time among a CU's workgroup slots (0: off, 1: static, inverse to age)
#endif
#ifndef HJ_WV_PRIO_SHIFT
#define HJ_WV_PRIO_SHIFT 10                         // rotation period = 2^shift ticks of the 100 MHz clock (10.24 us)
#endif
#ifndef HJ_WV_SKEW
#define HJ_WV_SKEW 0                                // extra tiles per chunk: chunk starts off the power-of-two stride
#endif
#ifndef HJ_WV_MIN_CHUNK
#define HJ_WV_MIN_CHUNK 32768                       // tuples: a second round of workgroups only while chunks stay this long
#endif
#ifndef HJ_WV_HICMP
#define HJ_WV_HICMP 1                               // 1: slot values are compared by their index words (32-bit compares, half the look's LDS bytes)
#endif
#ifndef HJ_WV_ALLIN
#define HJ_WV_ALLIN 1                               // 1: tiles wholly inside ring and range skip the per-tuple ring test
#endif
#ifndef HJ_WV_REQUEUE_FRONT
#define HJ_WV_REQUEUE_FRONT 1                       // 1: unfinished retry entries return to the queue's front (0: to its tail)
#endif
#ifndef HJ_WV_CARRY
#define HJ_WV_CARRY 1                               // 1: leave < 64 retry entries queued across tiles
#endif
constexpr int kWvPer = HJ_WV_PER;                   // tuples per lane per tile
constexpr int kWvTile = 64 * kWvPer;
constexpr int kWvPf = HJ_WV_PF;                     // tiles of R in flight per wavefront (register prefetch depth)
constexpr uint32_t kWvWinLog = HJ_WV_WINLOG;
constexpr uint32_t kWvWin = 1u << kWvWinLog;        // ring slots per wavefront
constexpr uint32_t kWvGran = kWvWin >> kGranShift;  // ring granules
constexpr uint32_t kWvQCap = HJ_WV_QCAP;            // retry queue entries per wavefront
#ifndef HJ_WV_ROUNDAT
#define HJ_WV_ROUNDAT 64
#endif
constexpr uint32_t kWvRoundAt = HJ_WV_ROUNDAT;      // a retry round runs once this many entries wait (<= 64)
#ifndef HJ_WV_EXPERIMENT
static_assert(kWvGran == kWvRingGran && kWvTile == (int)kWvTileTuples, "hj_device.h describes this geometry to the sampler");
#endif
static_assert(kWvQCap >= 128 && (kWvQCap & (kWvQCap - 1)) == 0, "FIFO ring: a power of two that takes one full step on top of < 64 waiting entries");
constexpr size_t kWvLdsBytes = (size_t)kWvWaves * (kWvWin * sizeof(uint64_t) + 3 * kWvQCap * sizeof(uint32_t));

__device__ __forceinline__ uint64_t wv_pack(uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; }
// The lane mask of a condition. HIP's __ballot(int) compares an INTEGER with zero: a condition that already is a lane mask
// (every compare produces one) is first turned into 0 / 1 per lane and compared again -- two vector instructions per
// ballot, ~50 per tile in a kernel whose retry rounds are bound by vector issue. This form takes the mask as it is.
__device__ __forceinline__ unsigned long long wv_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }


// ---- pre-pass: where every chunk starts and which slot range it owns -----------------------------------------------
// Chunk c nominally starts at position p = c * chunkLen. The seam is moved to where the data crosses a granule
// boundary: with m = the lowest home slot among the 64 tuples from p, the chunk starts at the first position q in
// [p, p + kWvLook) whose home slot lies in a later granule than m's, and its slot range starts at that granule --
// so (on near-sorted input) no tuple before q belongs above the boundary, and the few tuples after q that still
// belong below it are taken by the previous wavefront, which reads kWvOverlap positions past its own end. Without
// this every seam cost ~70 deferred tuples on `uniform` (half a granule's worth), each a cascade of global atomics.
// starts[c] = q (or p if no crossing shows up: many tuples per granule), raw[c] = the granule (kNone: no valid tuple).
constexpr uint32_t kWvLook = 256;
#ifndef HJ_WV_OVERLAP
#define HJ_WV_OVERLAP 64
#endif
constexpr uint32_t kWvOverlap = HJ_WV_OVERLAP;
// COMPACT build: positions before its seam a wavefront also reads (shadow zone), so that shadow zone + head zone are
// exactly its first tile; and the crossers one seam may let in before the build gives up on the compact table
constexpr uint32_t kWvShadow = 64 * HJ_WV_PER - HJ_WV_OVERLAP;
constexpr uint32_t kWvTail = HJ_WV_OVERLAP;      // the last positions of a chunk whose next-range tuples the NEXT wavefront inserts
constexpr uint32_t kWvPredCap = 64;
template <bool KEY32, bool HTM>
__global__ void __launch_bounds__(kBlock)
k_wave_seams(const void* __restrict__ Rv, uint64_t n, uint32_t chunkLen, uint32_t nChunks, uint64_t mask,
             uint32_t hshift, uint32_t* __restrict__ starts, uint32_t* __restrict__ raw, Gate gate)
{
    if (gate_closed(gate)) return;
    using Elem = typename std::conditional<KEY32, uint32_t, uint64_t>::type;
    const Elem* __restrict__ R = static_cast<const Elem*>(Rv);
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    if (c > nChunks) return;
    if (c == nChunks) { if (lane == 0) starts[c] = (uint32_t)n; return; }
    const uint64_t p = (uint64_t)c * chunkLen;
    // The look proceeds in steps of 128 tuples (two loads per lane) and stops at the first crossing: on the reference's
    // inputs one or two steps do (a granule is 128 slots), and the pre-pass reads 1-2 KiB per seam instead of 4 KiB
    // (32768 seams at 2^30: the full look was 134 MB of reads, 35 us).
    auto load_home = [&](uint64_t i) -> uint32_t {
        if (i >= n) return kNone;
        const uint64_t t = R[i];
        return ((t >> 32) == 0 && t != 0) ? home32<HTM>((uint32_t)t, hshift, (uint32_t)mask) : kNone;
    };
    uint32_t h0 = load_home(p + lane), h1 = load_home(p + 64 + lane);
    const uint32_t m = wave_umin(h0);
    uint32_t start = (uint32_t)p, g = m == kNone ? kNone : m >> kGranShift;
    if (c > 0 && p + kWvLook > n) {
        // A last chunk shorter than the look (a radix shard's key count is no multiple of the chunk length): too few tuples
        // to place a seam among -- without a crossing the seam would sit at p, in the middle of a granule's worth of tuples
        // that then belong to the other side in bulk. The previous chunk takes these tuples too (its slice has room for
        // kWvLook more), and this chunk starts at the relation's end, owning the table from the granule after the highest
        // home slot among them: no tuples, only its part of the table (and, in the compact build, its shadow granule).
        uint32_t mx = 0;                                              // highest home slot among this lane's valid tuples
        if (h0 != kNone) mx = h0;
        if (h1 != kNone && h1 > mx) mx = h1;
        for (uint32_t k = 2; k < kWvLook / 64; ++k) { const uint32_t h = load_home(p + 64 * k + lane); if (h != kNone && h > mx) mx = h; }
        const uint32_t hi = ~wave_umin(~mx);
        start = (uint32_t)n;
        g = m == kNone ? kNone : (hi >> kGranShift) + 1u;
    } else if (c > 0 && m != kNone) {
        const uint32_t edge = ((m >> kGranShift) + 1) << kGranShift;          // first slot of the next granule (0 on wrap: no crossing)
        for (uint32_t k = 0; k < kWvLook / 64 && edge != 0; k += 2) {
            if (k) { h0 = load_home(p + 64 * k + lane); h1 = load_home(p + 64 * (k + 1) + lane); }
            const unsigned long long hit0 = wv_ballot(h0 != kNone && h0 >= edge), hit1 = wv_ballot(h1 != kNone && h1 >= edge);
            if (hit0 | hit1) {
                start = hit0 ? (uint32_t)(p + 64 * k + (uint32_t)__ffsll((long long)hit0) - 1)
                             : (uint32_t)(p + 64 * (k + 1) + (uint32_t)__ffsll((long long)hit1) - 1);
                g = edge >> kGranShift;
                break;
            }
        }
    }
    if (lane == 0) { starts[c] = start; raw[c] = g; }
}

// bounds[c] = max over chunks <= c of raw (chunks without a valid sample inherit; chunks before the first valid sample
// take the first one's: nothing below it is owned), bounds[nChunks] = the table's end. Every workgroup takes kBlock
// consecutive chunks and finds the maximum over all chunks before its own by itself (coalesced reads of an array of at most
// 128 KiB that sits in L2) -- no second kernel, no look-back chain; the single-workgroup version of round 2 took 60 us
// for 32768 chunks.
__global__ void __launch_bounds__(kBlock)
k_wave_bounds_scan(const uint32_t* __restrict__ raw, uint32_t nChunks, uint32_t numGran, uint32_t* __restrict__ bounds, Gate gate)
{
    if (gate_closed(gate)) return;
    __shared__ uint32_t wmax[kBlock / 64], wscan[kBlock / 64];
    const uint32_t t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t base = blockIdx.x * kBlock, c = base + t;
    // maximum over the valid samples of every chunk before this workgroup's (kNone = no sample = contributes nothing)
    uint32_t before = 0;
    bool any = false;
    {   // 16-byte loads, four in flight per thread: the loop is a chain of L2 round trips otherwise (34 us for 32768 chunks)
        const uint4* raw4 = reinterpret_cast<const uint4*>(raw);          // raw is the start of a hipMalloc'ed buffer; base % 256 == 0
        const uint32_t nv = base >> 2;
        auto take = [&](uint32_t v) { before = (v != kNone && v > before) ? v : before; any |= v != kNone; };
        uint32_t k = t;
        for (; k + 3 * kBlock < nv; k += 4 * kBlock) {
            const uint4 a = raw4[k], b = raw4[k + kBlock], c4 = raw4[k + 2 * kBlock], d = raw4[k + 3 * kBlock];
            take(a.x); take(a.y); take(a.z); take(a.w); take(b.x); take(b.y); take(b.z); take(b.w);
            take(c4.x); take(c4.y); take(c4.z); take(c4.w); take(d.x); take(d.y); take(d.z); take(d.w);
        }
        for (; k < nv; k += kBlock) { const uint4 a = raw4[k]; take(a.x); take(a.y); take(a.z); take(a.w); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_xor(before, off, 64); before = o > before ? o : before; }
    const unsigned long long anyW = wv_ballot(any);
    if (lane == 0) { wmax[w] = before; wscan[w] = anyW ? 1u : 0u; }
    __syncthreads();
    uint32_t run = 0; bool have = false;
    for (int k = 0; k < kBlock / 64; ++k) { run = wmax[k] > run ? wmax[k] : run; have |= wscan[k] != 0; }
    __syncthreads();
    // inclusive prefix maximum inside the workgroup
    const uint32_t v = c < nChunks ? raw[c] : kNone;
    uint32_t inc = v == kNone ? 0u : v;
    const unsigned long long validMask = wv_ballot(v != kNone);
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(inc, off, 64); if ((int)lane >= off) inc = o > inc ? o : inc; }
    if (lane == 63) wmax[w] = inc;
    if (lane == 0) wscan[w] = validMask ? 1u : 0u;
    __syncthreads();
    bool haveHere = have || (validMask & ((2ull << lane) - 1ull)) != 0;
    for (uint32_t k = 0; k < w; ++k) { run = wmax[k] > run ? wmax[k] : run; haveHere |= wscan[k] != 0; }
    run = inc > run ? inc : run;
    if (c < nChunks) {
        if (!haveHere) {
            // no valid sample up to this chunk: the first valid one after it opens the first range (a relation that
            // starts with invalid tuples: rare, and the walk is short)
            run = 0;
            for (uint32_t k = c + 1; k < nChunks; ++k) { const uint32_t x = raw[k]; if (x != kNone) { run = x; break; } }
        }
        bounds[c] = run < numGran ? run : numGran;
    }
    if (blockIdx.x == 0 && t == 0) bounds[nChunks] = numGran;
}

#ifdef HJ_WV_CLOCKS
// Development builds only (tools/mk_variant.sh ... "-DHJ_WV_CLOCKS", tools/wave_clocks.py): when every chunk's wavefront
// started and ended, in ticks of the 100 MHz wall clock. Not in the product library.
__device__ uint32_t g_wvClk[2 * 65536];
extern "C" int hj_debug_wave_clocks(uint32_t* out, uint32_t nChunks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wvClk), (size_t)nChunks * 2 * sizeof(uint32_t));
}
#endif

// ---- the build ------------------------------------------------------------------------------------------------
// COMPACT (the fast path of the open-addressing table, hj_device.h "table formats"): the ring still holds (index << 32 | key),
// but what leaves for HBM is the KEY WORD alone -- 4 bytes per slot instead of 8, and the probe reads 4 bytes per slot too.
// The index words exist only to order the inserts, and inside a wavefront's ring they have done that by the time a granule
// retires; nothing after this kernel may therefore need them: there is no deferred phase. Every way a tuple used to get
// deferred is either handled here or raises Counters::compactFail, after which the classic 8-byte build (gated on the word
// k_wave_decide rewrites) redoes the whole table:
//   * walks that cross the seam into the next wavefront's range ("crossers", ~1.6 per seam on `uniform`): wavefront c
//     ALSO inserts, into a shadow granule below its range that it never writes out, the tuples whose home slot lies in
//     the last 128 slots of wavefront c-1's range (it reads kWvShadow positions before its seam for them), and so sees
//     for itself which of them walk across. Influence only flows towards higher slots, so c-1's own range never depends
//     on c's. Whether the shadow saw the truth is CHECKED, not assumed: c-1 lists the tuples that really left its range
//     at the seam, c lists the ones it let in, k_wave_validate compares the two lists seam by seam (equal lists = one
//     consistent run of the protocol over both ranges = the unique fixed point).
//   * retry entries the ring is about to leave behind get their rounds before it moves (forced rounds);
//   * anything else (a key far from its neighbours, a walk that wraps around the table end, key 0xFFFFFFFF = the compact
//     empty pattern) raises the flag.
template <bool KEY32, bool CHECK, bool HTM, bool COMPACT>
#ifndef HJ_WV_WPE_COMPACT
#define HJ_WV_WPE_COMPACT 4                         // COMPACT: wavefronts per SIMD the registers must allow (115 VGPRs once the chunk state is scalar)
#endif
__global__ void __launch_bounds__(kWvThreads, COMPACT ? HJ_WV_WPE_COMPACT : HJ_WV_WPE)
k_build_wave(const void* __restrict__ Rv, uint64_t n, uint32_t sliceLen, uint32_t nChunks, const uint32_t* __restrict__ starts,
             const uint32_t* __restrict__ bounds, uint64_t* __restrict__ table, uint64_t mask, uint32_t hshift,
             uint32_t probeLen, uint64_t idxBase, ShardCheck sc, DeferredEntry* __restrict__ queue,
             uint32_t* __restrict__ dcounts, Counters* __restrict__ ctr, Gate gate, uint64_t* __restrict__ htmConflicts,
             uint32_t* __restrict__ ccounts, uint32_t residentWG, uint32_t* __restrict__ pcounts)
{
    static_assert(!(COMPACT && HTM), "the bucketised table keeps its own layout");
    if (gate_closed(gate)) return;
    extern __shared__ __align__(16) uint64_t lds[];
    // the wavefront's number inside the workgroup is the same in all its lanes, but the compiler only knows that when told:
    // everything derived from the chunk number (its seams, its slot range, the ring's position, every test against them) then
    // lives in scalar registers and is computed by the scalar unit -- the vector unit is what this kernel runs out of
    const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t c = blockIdx.x * kWvWaves + wave;
    if (c >= nChunks) return;                                            // whole wavefronts leave; nobody waits for them
    if constexpr (COMPACT) {
        // some wavefront has met an input the compact table cannot take: the classic build will redo everything,
        // whatever is written from here on is never looked at (k_wave_decide reads the flag before anything else)
        if (*reinterpret_cast<volatile unsigned long long*>(&ctr->compactFail)) return;
    }
#ifdef HJ_WV_CLOCKS
    if (lane == 0 && c < 65536) g_wvClk[2 * c] = (uint32_t)wall_clock64();
#endif
    uint64_t* const win = lds + wave * kWvWin;                           // ring: slot s lives at win[s & (kWvWin - 1)]
    uint32_t* const myQPos = reinterpret_cast<uint32_t*>(lds + kWvWaves * kWvWin) + wave * 3 * kWvQCap;
    uint32_t* const myQLo = myQPos + kWvQCap;
    uint32_t* const myQHi = myQLo + kWvQCap;

#if HJ_WV_PRIO
    // workgroups per quarter of the first, resident round (residentWG = what this device holds at once, from the launch)
    const uint32_t prioDiv = ((gridDim.x <= residentWG ? gridDim.x : residentWG) + 3u) / 4u;
#endif
    const bool lastChunk = c + 1 == nChunks;
    const uint64_t cb0 = starts[c];
    const uint32_t clen = starts[c + 1] - starts[c];                     // the chunk proper: its tuples are counted here
    // COMPACT: the kWvShadow positions before the seam are read too (shadow zone, see above); every offset below is
    // relative to the first position READ, cb = cb0 - S, and the chunk proper is [S, S + clen)
    const uint32_t S = (COMPACT && c) ? (cb0 < kWvShadow ? (uint32_t)cb0 : kWvShadow) : 0u;
    const uint64_t cb = cb0 - S;
    const uint32_t cend = S + clen;
    // COMPACT: tuples of the NEXT range that sit before the seam ("early arrivals": keys a little ahead of their neighbours)
    // are left to the next wavefront, which reads those positions anyway (its shadow zone). Where they can sit is known
    // from the NOMINAL seam alone, which both sides know: the pre-pass moved the real seam forward from it by less than
    // kWvLook positions, to the first position of the next range -- so early arrivals lie before the nominal seam, a shuffle
    // window's length at most; the zone starts kWvTail positions before it. [tailFrom, cend): my tail zone;
    // [claimFrom, S): the previous chunk's tail zone as I see it.
    const uint32_t chunkLenNom = sliceLen - kWvLook - kWvOverlap;
    const uint32_t tailFrom = (COMPACT && !lastChunk) ? (uint32_t)((uint64_t)(c + 1) * chunkLenNom - cb) - kWvTail : 0xFFFFFFFFu;
    const uint32_t claimFrom = (COMPACT && c) ? (uint32_t)((uint64_t)c * chunkLenNom - cb) - kWvTail : 0u;
    static_assert(kWvShadow >= kWvLook + kWvTail, "the shadow zone reaches back past the previous chunk's tail zone");
    (void)tailFrom; (void)claimFrom;
    // ... and kWvOverlap positions of the next chunk are read too: its stragglers below the seam are inserted here
    const uint32_t plen = (uint32_t)((cb + cend + kWvOverlap < n ? cb + cend + kWvOverlap : n) - cb);
    const uint32_t mask32 = (uint32_t)mask;
    const uint32_t numGran = (uint32_t)((mask + 1) >> kGranShift);
    if constexpr (!KEY32) hshift = 0u;            // 8-byte tuples always hash with shift 0 (launch_build_wave checks): one shift less per home slot
    using Elem = typename std::conditional<KEY32, uint32_t, uint64_t>::type;
    const Elem* __restrict__ Rc = static_cast<const Elem*>(Rv) + cb;
    const uint32_t idx0 = (uint32_t)(idxBase + cb);
    // this wavefront's slot range, in granules: [loG, limG). The last chunk's range is open up to the table's end;
    // how far it got is published below (ownHiEx).
    const uint32_t loG = bounds[c];
    const uint32_t limG = lastChunk ? numGran : bounds[c + 1];
    const uint32_t loSlot = c ? loG << kGranShift : 0u;                 // head zone: tuples below it are the previous wavefront's
    const uint32_t limSlot = limG << kGranShift;                        // overlap zone: only tuples below it are mine (0 = 2^32: table end)
    // COMPACT: the shadow granule [shadowLo, loSlot) of the previous wavefront's range is simulated here, never written
    bool shadowOn = COMPACT && S != 0 && loG != 0;  // wave-uniform: the shadow granule is still in the ring (the first tile and its rounds)
    uint32_t pCount = 0;                         // COMPACT: crossers let in at the seam so far (wave-uniform)
    // COMPACT: something only the classic build can handle raises Counters::compactFail at once (a rare path: no register is
    // kept for it); the per-tuple tests of classify() collect in one lane mask that is looked at once per tile
    auto raise = [&](bool cond, unsigned long long why) {
        if (wv_ballot(cond) && lane == 0) atomicOr(&ctr->compactFail, why);
    };
    bool ffSeen = false;                         // a key equal to the compact empty pattern
    (void)pCount; (void)ffSeen;

    for (uint32_t i = lane; i < kWvWin / 2; i += 64) reinterpret_cast<ulonglong2*>(win)[i] = make_ulonglong2(kEmpty, kEmpty);

    uint32_t winLoG = shadowOn ? loG - 1 : loG;  // ring = granules [winLoG, winLoG + kWvGran), wave-uniform
    uint32_t qCount = 0, qHead = 0;              // retry queue (FIFO ring): entries and position of the oldest, wave-uniform
    uint32_t rounds = 0;                         // retry rounds so far (wave-uniform)
#ifdef HJ_WV_STATS
    uint32_t forcedRounds = 0;                   // development builds: rounds forced by the ring's movement; reported through `deferred`
#endif
    unsigned long long dropSum = 0, inSum = 0;
    uint32_t drops = 0, bad = 0, foreign = 0;
    uint32_t dCount = 0;                         // tuples deferred so far (wave-uniform)
    DeferredEntry* const myDeferred = queue + (uint64_t)c * sliceLen;      // <= clen + kWvOverlap <= sliceLen entries
    uint32_t cCount = 0;                         // HTM: conflicts recorded so far (wave-uniform)
    uint64_t* const myConflicts = HTM ? htmConflicts + (uint64_t)c * sliceLen : nullptr;
    (void)cCount; (void)myConflicts;
    uint32_t usedLo = kNone, usedHi1 = 0;        // 512-slot blocks this lane deferred into (Counters::usedLoInv / usedHi1)

    // the ring's tail up to granule `target` leaves for HBM (empties included) and its LDS copy is reset
    auto advance = [&](uint32_t target) {
        while (winLoG < target) {
            ulonglong2* src = reinterpret_cast<ulonglong2*>(win + ((winLoG & (kWvGran - 1)) << kGranShift)) + lane;
            const ulonglong2 t = *src;
#if defined(HJ_WV_ABL_NOSTORE)
            if (t.x == 0x1234567ull) table[lane] = t.y;        // ablation (development builds): the retire stores never happen
#else
            if constexpr (COMPACT) {
                // the key words alone leave: 512 bytes per granule (the empty pattern's low word is the compact empty
                // pattern); the shadow granule belongs to the previous wavefront and is not written
                if (winLoG >= loG) {
                    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                    u2 vv; vv.x = (uint32_t)t.x; vv.y = (uint32_t)t.y;
                    __builtin_nontemporal_store(vv, reinterpret_cast<u2*>(reinterpret_cast<uint32_t*>(table) + ((uint64_t)winLoG << kGranShift)) + lane);
                }
            } else {
                // written once and not read again by this kernel: nontemporal stores (-1.5 % kernel time at 2^30, and
                // the probe that follows runs 1 % faster; nontemporal LOADS of R were slower)
                typedef unsigned long long v2 __attribute__((ext_vector_type(2)));
                v2 vv; vv.x = t.x; vv.y = t.y;
                __builtin_nontemporal_store(vv, reinterpret_cast<v2*>(table + ((uint64_t)winLoG << kGranShift)) + lane);
            }
#endif
            *src = make_ulonglong2(kEmpty, kEmpty);
            ++winLoG;
        }
    };
    // may this wavefront touch slot pos right now: inside the ring and inside the owned range
    auto in_ring = [&](uint32_t pos) -> bool {
        const uint32_t g = pos >> kGranShift;
        return (g - winLoG < kWvGran) & (g < limG);
    };

    // One round on the entry (pos, mlo, mhi) a lane holds (hj_build_own.hip's round, with range + ring standing in
    // for block ownership): terminal events are placed / dropped (budget exhausted, NoCCHashBuild.hpp:57-58) /
    // deferred; returns "not finished" with the entry's next state in place.
    auto round_body = [&](uint32_t& pos, uint32_t& mlo, uint32_t& mhi, const bool has) -> bool {
        const uint32_t key = mlo;
        const uint32_t homeOfKey = home32<HTM>(key, hshift, mask32);
        uint32_t budget = probeLen - ((pos - homeOfKey) & mask32);
        if constexpr (COMPACT) {
            // a tuple of the shadow granule arriving at my first slot: a crosser I let in (each passes pos == loSlot
            // exactly once: walks advance slot by slot, and the look never skips across a granule boundary)
            if (shadowOn) {
                const bool cross = has & (pos == loSlot) & (homeOfKey < loSlot) & (budget != 0);
                const unsigned long long xm = wv_ballot(cross);
                if (xm) {
                    const uint32_t at = pCount + lane_rank(xm);
                    // the list lives at the end of this chunk's slice of the deferred queue (which the compact build otherwise
                    // uses for the handful of crossers that leave the range): kWvPredCap 8-byte entries
                    uint64_t* const myPred = reinterpret_cast<uint64_t*>(queue + (uint64_t)(c + 1) * sliceLen) - kWvPredCap;
                    if (cross & (at < kWvPredCap)) myPred[at] = wv_pack(mhi, mlo);
                    raise(cross & (at >= kWvPredCap), 2ull);
                    pCount += (uint32_t)__popcll(xm);
                }
            }
        }
        const bool ownOk = in_ring(pos);
        const bool drop0 = has & (budget == 0);
        const bool toDefer = has & !drop0 & !ownOk;
        const bool work = has & !drop0 & ownOk;
        const uint64_t mine = wv_pack(mhi, mlo);
        // look before leaping: the next 4 slots, when they sit in the same granule (contiguous in the ring and owned
        // together); slot values only decrease, so a slot seen below `mine` stays below it
        const bool inGran = (pos & (kGranSlots - 1)) <= kGranSlots - 4;
        const uint32_t rd = (work & inGran) ? pos : (winLoG << kGranShift);
#if HJ_WV_HICMP
        // slot values are (index << 32 | key) with one index per tuple, the empty pattern has the highest index word: order
        // and equality of two values are those of their INDEX words. The look reads those alone (half the LDS bytes,
        // two ds_read2_b32) and every compare is a 32-bit one.
        const uint32_t* w = reinterpret_cast<const uint32_t*>(&win[rd & (kWvWin - 1)]) + 1;
        const uint32_t myIdx = mhi;
        const bool c0 = w[0] < myIdx, c1 = c0 & (w[2] < myIdx), c2 = c1 & (w[4] < myIdx), c3 = c2 & (w[6] < myIdx);
#else
        const uint64_t* w = &win[rd & (kWvWin - 1)];
        const uint64_t v0 = w[0], v1 = w[1], v2 = w[2], v3 = w[3];
        const bool c0 = v0 < mine, c1 = c0 & (v1 < mine), c2 = c1 & (v2 < mine), c3 = c2 & (v3 < mine);
#endif
        uint32_t skip = (uint32_t)c0 + (uint32_t)c1 + (uint32_t)c2 + (uint32_t)c3;
        skip = (work & inGran) ? (skip < budget ? skip : budget) : 0u;
        pos = (pos + skip) & mask32; budget -= skip;
        const bool drop1 = work & (budget == 0);
        const bool recheck = work & !drop1 & (skip == 4);                  // may have left the granule: next round
        const bool doAtomic = work & !drop1 & !recheck;
        unsigned long long old = kEmpty;
        if (doAtomic)
            old = atomicMin(reinterpret_cast<unsigned long long*>(&win[pos & (kWvWin - 1)]), (unsigned long long)mine);
#if HJ_WV_HICMP
        const uint32_t oldIdx = (uint32_t)(old >> 32);
        const bool fail = doAtomic & (oldIdx != 0xFFFFFFFFu) & (oldIdx != myIdx);
        const bool disp = fail & (oldIdx > myIdx);                         // displaced a later tuple: carry it on
#else
        const bool fail = doAtomic & (old != kEmpty) & (old != mine);
        const bool disp = fail & (old > mine);                             // displaced a later tuple: carry it on
#endif
        mlo = disp ? (uint32_t)old : mlo; mhi = disp ? (uint32_t)(old >> 32) : mhi;
        bool dropped = drop0 | drop1;
        if constexpr (COMPACT) {
            // a shadow tuple that runs out of budget BELOW my range is the previous wavefront's conflict, not mine; one that
            // tried a slot of my range first (its last try is home + probeLen - 1) came in as a crosser and is mine
            if (shadowOn) dropped = dropped & ((homeOfKey >= loSlot) | (homeOfKey + (probeLen - 1u) >= loSlot));
        }
        drops += dropped ? 1u : 0u; dropSum += dropped ? (unsigned long long)key : 0ull;
        if constexpr (HTM) {       // the bucket is full: the tuple is one of the reference's conflicts (HTMHashBuild.hpp:181-183)
            const unsigned long long cm = wv_ballot(dropped);
            if (cm) {
                if (dropped) myConflicts[cCount + lane_rank(cm)] = mine;
                cCount += (uint32_t)__popcll(cm);
            }
        }
        // deferred tuples go to this wavefront's OWN slice of the deferred queue, [cb, cb + clen): a tuple leaves at
        // most once, so the slice cannot overflow, and no atomic is needed to place it (a returning global atomic
        // per round with a straggler stalled the wavefront for microseconds)
        const unsigned long long dm = wv_ballot(toDefer);
        if (dm) {
            if constexpr (COMPACT) {
                // no deferred phase: a walk that leaves my range at the seam is listed for the check against what the
                // next wavefront let in (it has inserted the tuple itself); everything else only the classic build can do
                const bool crosser = toDefer & (pos == limSlot) & (homeOfKey < limSlot) & !lastChunk;
                const unsigned long long cmk = wv_ballot(crosser);
                raise(toDefer & !crosser, 1ull);
                if (cmk) {
                    if (crosser) myDeferred[dCount + lane_rank(cmk)].packed = mine;
                    dCount += (uint32_t)__popcll(cmk);
                }
            } else {
            if (toDefer) {
                DeferredEntry* q = myDeferred + dCount + lane_rank(dm);
                q->pos = pos; q->packed = mine;
                const uint32_t db = pos >> 9;
                usedLo = db < usedLo ? db : usedLo; usedHi1 = db + 1 > usedHi1 ? db + 1 : usedHi1;
            }
            dCount += (uint32_t)__popcll(dm);
            }
        }
        pos = fail ? ((pos + 1) & mask32) : pos;
        return recheck | fail;
    };
    // The retry queue is a FIFO ring (oldest entries first): a round takes the up to 64 oldest entries, whatever is
    // not finished goes back to the tail. With HJ_WV_CARRY the queue is NOT drained at the end of a tile: fewer than
    // 64 entries wait for the next tile's failures, so that every round is dense (draining a tile to completion cost
    // ~10 sparse rounds per tile, more instructions than all the dense work together -- PMC, profiles/r02_*).
    auto retry_round = [&]() {
        const uint32_t take = qCount < 64u ? qCount : 64u;
        const bool has = lane < take;
        // lanes >= take read stale-but-in-bounds entries and ignore them
        const uint32_t at0 = (qHead + lane) & (kWvQCap - 1);
        uint32_t pos = myQPos[at0], mlo = myQLo[at0], mhi = myQHi[at0];
        qHead = (qHead + take) & (kWvQCap - 1);
        qCount -= take;
        const bool again = round_body(pos, mlo, mhi, has);
        const unsigned long long am = wv_ballot(again);
#if HJ_WV_REQUEUE_FRONT
        // what did not finish goes back to the queue's FRONT: an entry then gets its (at most probeLength) rounds one after
        // the other and is done while its slots are still far from the ring's tail, instead of waiting behind a tile's
        // worth of newer entries (COMPACT has no deferred phase: an entry the ring is about to leave behind costs a
        // forced round, below)
        const uint32_t back = (uint32_t)__popcll(am);
        qHead = (qHead - back) & (kWvQCap - 1);
        if (again) {
            const uint32_t at = (qHead + lane_rank(am)) & (kWvQCap - 1);
            myQPos[at] = pos; myQLo[at] = mlo; myQHi[at] = mhi;
        }
        qCount += back;
#else
        if (again) {
            const uint32_t at = (qHead + qCount + lane_rank(am)) & (kWvQCap - 1);
            myQPos[at] = pos; myQLo[at] = mlo; myQHi[at] = mhi;
        }
        qCount += (uint32_t)__popcll(am);
#endif
        ++rounds;
    };
    // to completion: dense rounds while more than a wavefront's worth is queued, then the last <= 64 entries stay in
    // registers until they are done
    auto drain = [&]() {
        while (qCount > 64u) retry_round();
        bool act = lane < qCount;
        const uint32_t at0 = (qHead + lane) & (kWvQCap - 1);
        uint32_t pos = myQPos[at0], mlo = myQLo[at0], mhi = myQHi[at0];
        qCount = 0;
        while (wv_ballot(act)) act = round_body(pos, mlo, mhi, act);
    };

    // tile t covers chunk offsets [t * kWvTile, ...); lane's tuple j sits at offset t * kWvTile + 64 j + lane
    // register prefetch, kWvPf tiles deep: the loads are issued unconditionally (lanes past the chunk's end are
    // masked, not branched around), so the compiler can count them and wait for one tile's loads only
    Elem nxt[kWvPf][kWvPer];
    auto issue = [&](Elem (&buf)[kWvPer], uint32_t at) {
#pragma unroll
        for (int j = 0; j < kWvPer; ++j) {
            const uint32_t o = at + lane + 64 * j;
            buf[j] = o < plen ? Rc[o] : (Elem)0;
        }
    };
#pragma unroll
    for (int p = 0; p < kWvPf; ++p) issue(nxt[p], (uint32_t)p * kWvTile);

    // COMPACT: lowest / highest home slot of the tile about to start, taken from its keys at the end of the tile before it
    // (forced rounds, below); only for full tiles
    uint32_t tminF = kNone, tmaxF = 0;
    (void)tminF; (void)tmaxF;
    static_assert(!COMPACT || kWvPf == 1, "the forced rounds look one tile ahead");
    for (uint32_t tb0 = 0; tb0 < plen; tb0 += kWvPf * kWvTile) {
#pragma unroll
      for (int p = 0; p < kWvPf; ++p) {
        const uint32_t tb = tb0 + (uint32_t)p * kWvTile;
        if (tb >= plen) break;                                        // wave-uniform
        uint32_t klo[kWvPer], khi[kWvPer];
#pragma unroll
        for (int j = 0; j < kWvPer; ++j) {
            klo[j] = (uint32_t)nxt[p][j];
            khi[j] = KEY32 ? 0u : (uint32_t)((uint64_t)nxt[p][j] >> 32);
        }
#if HJ_WV_PRIO
        {   // The CU arbitrates oldest wavefront first, and the four workgroups of a CU are dispatched one after the other:
            // measured at 2^27 (tools/wave_clocks.py, every wavefront's start and end), the wavefronts of a CU's first
            // workgroup took 349 us for their chunk, those of the second 380, the third 421, the fourth 470 -- the kernel
            // lasted 497 us and ran half empty for its last fifth. So every wavefront sets its issue priority per tile
            // from the wall clock: the four workgroup slots of a CU take turns at the top (10 us each). With it: 389 / 398 /
            // 398 / 400 us, kernel 446 us. (From 2^28 tuples on the later rounds of workgroups do the balancing; no gain or
            // loss there. An input without retry rounds is memory bound and hardly reacts: the arbitration that is
            // unfair to its younger wavefronts is the memory system's.)
            const uint32_t slot = (blockIdx.x / prioDiv) & 3u;
#if HJ_WV_PRIO == 1
            const uint32_t pr = slot;
            if (tb == 0) {
#else
            const uint32_t pr = ((uint32_t)(wall_clock64() >> HJ_WV_PRIO_SHIFT) + slot) & 3u;
            {
#endif
                if (pr == 0) __builtin_amdgcn_s_setprio(0);
                else if (pr == 1) __builtin_amdgcn_s_setprio(1);
                else if (pr == 2) __builtin_amdgcn_s_setprio(2);
                else __builtin_amdgcn_s_setprio(3);
            }
        }
#endif
        // wave-uniform: no shadow / head zone, no overlap zone (COMPACT: and no tail zone)
        if constexpr (COMPACT) {
            // somebody has raised the flag: nothing written from here on is ever looked at (every 8th tile: one scalar load)
            if ((tb & (8u * kWvTile - 1u)) == 7u * kWvTile && *reinterpret_cast<volatile unsigned long long*>(&ctr->compactFail)) return;
        }
        const bool full = (tb + kWvTile <= (cend < tailFrom ? cend : tailFrom)) & (tb >= S + kWvOverlap);
        issue(nxt[p], tb + kWvPf * kWvTile);
        const uint32_t roundsAtTileStart = rounds;
        (void)roundsAtTileStart;
        uint32_t myMin = kNone, myMaxInv = kNone;                     // max kept as min of the complement
        uint32_t home[kWvPer];
        // per row: "a tuple this wavefront inserts" / "... and whose home slot it may touch now". Kept as one boolean per
        // row -- a lane mask in a scalar register pair, free to produce and to branch on -- not as bits of a per-lane word
        // (two VALU instructions per tuple to pack, two to unpack, in a kernel whose retry rounds are VALU bound)
        bool live[kWvPer], own[kWvPer];
        // FULL tiles (every position is this chunk's own, no seam zone) skip the zone tests: they are all but the
        // first and the last one or two tiles of a chunk
        uint32_t badTile = 0;                                         // wave-uniform: invalid tuples of a full tile
        // COMPACT, first and last tiles of a chunk: which of the tile's tuples are inserted here (bit j) and counted here
        // (bit 8 + j). The zone tests are many, and unrolled over the tile's eight tuples per lane their lane masks took ~40
        // scalar registers more than the kernel has (spilled to vector lanes all over the hot path); so this cold path
        // takes the tuples one at a time in a rolled loop, re-reading the keys (cache hits).
        uint32_t zb = 0;
        (void)zb;
        if constexpr (COMPACT) {
            if (!full) {
#pragma unroll 1
                for (int j = 0; j < kWvPer; ++j) {
                    const uint32_t o = tb + lane + 64 * j;
                    const Elem t = o < plen ? Rc[o] : (Elem)0;
                    const uint32_t key = (uint32_t)t;
                    const bool okKey = (KEY32 || (uint32_t)((uint64_t)t >> 32) == 0) & (key != 0);
                    const uint32_t h = home32<HTM>(key, hshift, mask32);
                    const bool in = (o >= S) & (o < cend);
                    // inserted here: my tuples, except head-zone stragglers of the previous range; plus the next chunk's
                    // stragglers of MY range in the overlap zone ...
                    bool mine = o < cend ? !((o < S + kWvOverlap) & (h < loSlot)) : ((o < plen) & !lastChunk & (h < limSlot));
                    // ... shadow zone and head zone: tuples of the previous range's last granule are simulated here as well
                    // (it lists the ones that really leave it; k_wave_validate compares)
                    const bool shadowTuple = (o < S + kWvOverlap) & (h - loSlot + kGranSlots < kGranSlots) & (loSlot != 0);
                    // ... and a tuple of MY range that sits before the seam ("early arrival": a key a little ahead of its
                    // neighbours) is mine to insert; the previous wavefront leaves it (its tail zone, below)
                    if (o < S) mine = shadowTuple | ((o >= claimFrom) & (h >= loSlot)); else mine = mine | shadowTuple;
                    // my tail zone: tuples of the next range are the next wavefront's
                    if ((o >= tailFrom) & (o < cend) & (h >= limSlot)) mine = false;
                    // a tuple of the chunk proper, past the head zone, that belongs below my range: nobody else will insert
                    // it (the previous wavefront reads kWvOverlap positions past its end, no more)
                    const bool below = okKey & (o >= S + kWvOverlap) & (o < cend) & (h < loSlot);
                    raise(below, 8ull);
                    mine = mine & !below;
                    zb |= ((mine ? 1u : 0u) << j) | ((in ? 1u : 0u) << (8 + j));
                }
            }
        }
        auto classify = [&](auto fullTag) {
            constexpr bool FULL = decltype(fullTag)::value;
#pragma unroll
            for (int j = 0; j < kWvPer; ++j) {
                const uint32_t o = tb + lane + 64 * j;
                bool in, mineHere;
                if constexpr (COMPACT && !FULL) {
                    in = ((zb >> (8 + j)) & 1u) != 0; mineHere = ((zb >> j) & 1u) != 0;
                } else {
                    in = FULL || ((o >= S) & (o < cend));                         // counted here
                }
                bool okKey = (khi[j] == 0) & (klo[j] != 0);
                home[j] = home32<HTM>(klo[j], hshift, mask32);
                if constexpr (!(COMPACT && !FULL)) {
                    // inserted here: my tuples, except head-zone stragglers of the previous range; plus the next chunk's
                    // stragglers of MY range in the overlap zone
                    mineHere = FULL || (o < cend ? !((o < S + kWvOverlap) & (home[j] < loSlot))
                                                 : ((o < plen) & !lastChunk & (home[j] < limSlot)));
                }
                if constexpr (COMPACT) {
                    // the compact empty pattern cannot be a key: the classic build takes such an input. Its home slot is the
                    // table's last one, so a full tile (whose highest home slot is already known, tmaxF) only looks when
                    // that slot occurs at all -- one vector instruction per tuple saved in every other tile
                    if (!FULL || tmaxF == mask32) ffSeen |= in & (klo[j] == 0xFFFFFFFFu);
                }
                const bool ok = mineHere & okKey;
                inSum += in ? (unsigned long long)wv_pack(khi[j], klo[j]) : 0ull;
                if constexpr (FULL) badTile += 64u - (uint32_t)__popcll(wv_ballot(okKey));     // scalar: no per-lane count
                else bad += (in & !okKey) ? 1u : 0u;
                if constexpr (CHECK) foreign += (in & is_foreign(klo[j], sc)) ? 1u : 0u;
                live[j] = ok;
                if constexpr (FULL && COMPACT) {
                    // the bounds of a full tile were taken at its top (forced rounds)
                } else if constexpr (FULL && HJ_WV_ALLIN) {
                    // full tiles take the bounds over every tuple, valid or not: an invalid key (the build fails with
                    // HJ_ERR_KEY_RANGE anyway) can only make the ring move less, and every access stays guarded
                    myMin = home[j] < myMin ? home[j] : myMin;
                    myMaxInv = ~home[j] < myMaxInv ? ~home[j] : myMaxInv;
                } else {
                    myMin = (ok & (home[j] < myMin)) ? home[j] : myMin;
                    myMaxInv = (ok & (~home[j] < myMaxInv)) ? ~home[j] : myMaxInv;
                }
            }
        };
        if (full) classify(std::true_type{}); else classify(std::false_type{});
        bad += lane == 0 ? badTile : 0u;
        uint32_t tmin, tmax = 0;                                      // tmin == kNone: the tile holds no valid tuple
        if (COMPACT && full) { tmin = tminF; tmax = tmaxF; }          // taken from the keys at the end of the tile before
        else { tmin = wave_umin(myMin); if (tmin != kNone) tmax = ~wave_umin(myMaxInv); }
        bool allIn = false;                                           // wave-uniform: every home slot of the tile lies in ring and range
        if (tmin != kNone) {
            // The ring moves only as far as it must for the tile's highest home slot (+ a probe walk) to fit, so it
            // keeps as much history as it can: retry entries carried over from the previous tile are still inside.
            // It never moves past the tile's lowest home slot (an outlier key is deferred, not the tile).
            const uint32_t top = (uint32_t)(((uint64_t)tmax + probeLen + 8u) >> kGranShift) + 1u;
            uint32_t target = top > kWvGran ? top - kWvGran : 0u;
            const uint32_t gmin = tmin >> kGranShift;
            target = target < gmin ? target : gmin;
            target = target < limG ? target : limG;
            advance(target);
            if (HJ_WV_ALLIN) {
                const uint32_t gmax = tmax >> kGranShift;
                allIn = full & (gmin - winLoG < kWvGran) & (gmax - winLoG < kWvGran) & (gmax < limG);
            }
        }

        // ---- the PER home-slot attempts of the tile, issued together (independent LDS round trips) ----
        unsigned long long oldv[kWvPer];
        // (when the tile's lowest and highest home slot are inside ring and range -- nearly every full tile -- the
        // per-tuple test is skipped: the retry rounds leave the VALU little to spare, hj DESIGN 4.2)
        auto attempts = [&](auto allTag) {
            constexpr bool ALL = decltype(allTag)::value;
#pragma unroll
            for (int j = 0; j < kWvPer; ++j) {
                own[j] = live[j] & (ALL || in_ring(home[j]));
                oldv[j] = kEmpty;
                if (own[j])
                    oldv[j] = atomicMin(reinterpret_cast<unsigned long long*>(&win[home[j] & (kWvWin - 1)]),
                                        (unsigned long long)wv_pack(idx0 + tb + lane + 64 * j, klo[j]));
            }
        };
        if (allIn) attempts(std::true_type{}); else attempts(std::false_type{});
        // ---- whatever did not finish goes to the retry queue, compacted ----
#pragma unroll
        for (int j = 0; j < kWvPer; ++j) {
            while (qCount >= kWvRoundAt) retry_round();                // dense rounds; leaves room for one full step
            const bool lv = live[j], ow = own[j];
            uint32_t mlo = klo[j], mhi = idx0 + tb + lane + 64 * j;
#if HJ_WV_HICMP
            const uint32_t oldIdx = (uint32_t)(oldv[j] >> 32);
            const bool fail = ow & (oldIdx != 0xFFFFFFFFu);            // the slot was taken
            const bool disp = fail & (oldIdx > mhi);                   // ... by a later tuple: it moves on instead
#else
            const uint64_t mine = wv_pack(mhi, mlo);
            const bool fail = ow & (oldv[j] != kEmpty);                // the slot was taken
            const bool disp = fail & (oldv[j] > mine);                 // ... by a later tuple: it moves on instead
#endif
            mlo = disp ? (uint32_t)oldv[j] : mlo; mhi = disp ? (uint32_t)(oldv[j] >> 32) : mhi;
            const uint32_t pos = fail ? ((home[j] + 1) & mask32) : home[j];
            const bool again = fail | (lv & !ow);                      // outside ring or range: the retry round defers it
            const unsigned long long am = wv_ballot(again);
            if (am) {
                if (again) {
                    const uint32_t at = (qHead + qCount + lane_rank(am)) & (kWvQCap - 1);
                    myQPos[at] = pos; myQLo[at] = mlo; myQHi[at] = mhi;
                }
                qCount += (uint32_t)__popcll(am);
            }
        }
#if HJ_WV_CARRY
        if constexpr (COMPACT) {
            // Forced rounds. A queued entry whose slot the ring is about to leave behind would have to be deferred, and
            // there is no deferred phase: it gets its rounds now, at the end of the tile, where nothing of the tile is
            // in registers any more and the one retry loop of this place serves (fewer than 64 entries wait then: one
            // lane each). How far the ring will move is known from the NEXT tile's keys, which the prefetch has already
            // brought: a full tile's bounds go over every tuple, valid or not (rule of the ring above), and stay in two
            // scalar registers for its classification; before a chunk's first and last tiles everything queued simply
            // finishes. The first tile (shadow zone + head zone) ends the shadow phase the same way: once the queue is
            // empty the shadow granule is dropped (it is never written), so that from here on a tuple that belongs below
            // my range cannot land in it unnoticed -- it falls out of the ring and raises the flag.
            const uint32_t tbN = tb + kWvTile;
            uint32_t tslot = 0xFFFFFFFFu;
            const bool endShadow = shadowOn;
            if ((tbN + kWvTile <= (cend < tailFrom ? cend : tailFrom)) & (tbN >= S + kWvOverlap)) {
                uint32_t mn = kNone, mxInv = kNone;
#pragma unroll
                for (int j = 0; j < kWvPer; ++j) {
                    const uint32_t h = home32<HTM>((uint32_t)nxt[p][j], hshift, mask32);
                    mn = h < mn ? h : mn; mxInv = ~h < mxInv ? ~h : mxInv;
                }
                tminF = wave_umin(mn); tmaxF = ~wave_umin(mxInv);
                const uint32_t top = (uint32_t)(((uint64_t)tmaxF + probeLen + 8u) >> kGranShift) + 1u;
                uint32_t target = top > kWvGran ? top - kWvGran : 0u;
                const uint32_t gmin = tminF >> kGranShift;
                target = target < gmin ? target : gmin;
                target = target < limG ? target : limG;
                if (!endShadow) tslot = target > winLoG ? target << kGranShift : 0u;
            }
            for (;;) {
                bool need = qCount >= kWvRoundAt;
                if (!need && qCount && tslot) {
                    const uint32_t qp = myQPos[(qHead + lane) & (kWvQCap - 1)];
                    need = wv_ballot((lane < qCount) & (qp < tslot)) != 0;
#ifdef HJ_WV_STATS
                    forcedRounds += need ? 1u : 0u;
#endif
                }
                if (!need) break;
                retry_round();
            }
            if (endShadow) {
                shadowOn = false;
                while (winLoG < loG) {
                    reinterpret_cast<ulonglong2*>(win + ((winLoG & (kWvGran - 1)) << kGranShift))[lane] = make_ulonglong2(kEmpty, kEmpty);
                    ++winLoG;
                }
            }
        } else {
        while (qCount >= kWvRoundAt) retry_round();
        // an entry may wait for company for one tile, not longer (the ring moves on): no round during this tile -> one now
        if (qCount && rounds == roundsAtTileStart) retry_round();
        }
#else
        if (qCount) drain();                                           // before the ring may move on
#endif
      }
    }
    if (qCount) drain();

    // ---- the rest of the range: what is left in the ring, then empties up to the next chunk's range ----
    uint32_t endG = limG;
    if (lastChunk) endG = winLoG + kWvGran < numGran ? winLoG + kWvGran : numGran;
    advance(endG);
    if (lane == 0) {
        if (c == 0) ctr->ownLo = (unsigned long long)loG << kGranShift;
        if (lastChunk) ctr->ownHiEx = (unsigned long long)winLoG << kGranShift;
        dcounts[c] = dCount;
        if constexpr (HTM) ccounts[c] = cCount;
        if constexpr (COMPACT) pcounts[c] = pCount;
    }
    if constexpr (COMPACT) {
        raise(ffSeen, 4ull);
        dCount = 0;                              // crossers are not deferred tuples: the next wavefront has inserted them
    }

#ifdef HJ_WV_CLOCKS
    if (lane == 0 && c < 65536) g_wvClk[2 * c + 1] = (uint32_t)wall_clock64();
#endif
#if HJ_WV_PRIO
    __builtin_amdgcn_s_setprio(0);               // the rotation's priority is not carried into the counter atomics
#endif
    // counters: one atomic per wavefront
    unsigned long long c0 = drops, c3 = bad | ((unsigned long long)foreign << 32);
#ifdef HJ_WV_STATS
    const unsigned long long c4 = (unsigned long long)rounds | ((unsigned long long)forcedRounds << 32);
#else
    const unsigned long long c4 = dCount;
#endif
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        c0 += __shfl_down(c0, off, 64);
        dropSum += __shfl_down(dropSum, off, 64);
        inSum += __shfl_down(inSum, off, 64);
        c3 += __shfl_down(c3, off, 64);
    }
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

// Phase B for the sliced deferred queue: chunk c's entries are queue[c * chunkLen .. + dcounts[c]). Same walk as
// k_build_deferred (hj_build_own.hip): the probe walk of every deferred tuple finished with global atomics. The walks
// are chains of dependent memory-side atomics, and what bounds the phase is their THROUGHPUT, not its shape: 947 k entries
// at 2^30 `uniform` take 101-107 us whether a workgroup takes a slice (round 2), a wavefront does (now), or the entries are
// numbered through and dealt out to all lanes (round 3, binary search in a prefix of the counts: 104 us) -- 1.8 returning
// 64-bit atomics per entry at the 17 G/s the global-atomic build reaches too.
template <bool HTM>
__global__ void __launch_bounds__(kBlock)
k_wave_deferred(const DeferredEntry* __restrict__ queue, const uint32_t* __restrict__ dcounts, uint32_t nChunks,
                uint32_t chunkLen, uint64_t* __restrict__ table, uint64_t mask, uint32_t hshift, uint32_t probeLen,
                Counters* __restrict__ ctr, Gate gate, uint64_t* __restrict__ htmConflicts, uint32_t* __restrict__ ccounts,
                const uint32_t* __restrict__ routeBounds)
{
    if (gate_closed(gate)) return;
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long drops = 0, dropSum = 0;
    // a wavefront per slice (~220 entries per slice on `uniform` with one round of chunks, ~29 in each of 32768 slices at 2^30)
    for (uint32_t c = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); c < nChunks; c += gridDim.x * (kBlock / 64)) {
        const uint32_t cnt = dcounts[c];
        const DeferredEntry* q = queue + (uint64_t)c * chunkLen;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool has = i < cnt;
            uint64_t mine = has ? q[i].packed : 0ull;
            uint64_t pos = has ? q[i].pos : 0ull;
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
                // the slice's conflict list is appended to by four wavefronts now: one atomic per wavefront reserves the places
                // (ccounts[c] holds what k_build_wave recorded; the list's order is free, the chain phase sorts by index)
                if (routeBounds) {
                    // routed (hj_htm.hip, the chain phase in LDS): the conflict is filed under the chunk that OWNS its bucket's
                    // granule -- the last chunk whose first granule is not above it --, so that a slice of the list holds all
                    // conflicts of its chunk's range. A slice that cannot take it: the host redoes the build unrouted.
                    if (dropped) {
                        const uint32_t gran = (uint32_t)(home32<true>((uint32_t)mine, hshift, (uint32_t)mask) >> kGranShift);
                        uint32_t a = 0, b = nChunks;                                   // bounds[a] <= gran (or a = 0), answer in [a, b)
                        while (b - a > 1) { const uint32_t mid = (a + b) >> 1; if (routeBounds[mid] <= gran) a = mid; else b = mid; }
                        const uint32_t at = atomicAdd(&ccounts[a], 1u);
                        if (at < chunkLen) htmConflicts[(uint64_t)a * chunkLen + at] = mine;
                        else atomicExch(&ctr->htmChainBail, 1ull);
                    }
                    continue;
                }
                const unsigned long long cm = wv_ballot(dropped);
                if (cm) {
                    uint32_t base = 0;
                    if (lane == (uint32_t)__ffsll((long long)cm) - 1u) base = atomicAdd(&ccounts[c], (uint32_t)__popcll(cm));
                    base = (uint32_t)__shfl((int)base, __ffsll((long long)cm) - 1, 64);
                    if (dropped) htmConflicts[(uint64_t)c * chunkLen + base + lane_rank(cm)] = mine;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        drops += __shfl_down(drops, off, 64);
        dropSum += __shfl_down(dropSum, off, 64);
    }
    if (lane == 0) {
        if (drops) atomicAdd(&counter_shard(ctr)->conflicts, drops);
        if (dropSum) atomicAdd(&counter_shard(ctr)->conflictSum, dropSum);
    }
}

// After k_build_wave: the valid slot range (hj_device.h, Counters) = the owned stretch [ownLo, ownHiEx) joined
// with the blocks deferred tuples start from (+1: a probe walk spills at most probeLen - 1 slots). If it reaches
// the table's end (walks wrap there) the whole table is made valid. One wavefront. (Round 3 tried this fold inside
// k_wave_fill_edges, every workgroup for itself, to save the launch: 4 us at 2^22 -- and 55 us at 2^30, with 128 or
// with 1024 workgroups, against 4.4 + 4.9 us for the two launches: taken back.)
__global__ void k_wave_finalize_range(Counters* __restrict__ ctr, uint64_t tableSize, Gate gate)
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
    unsigned long long lo = ctr->ownLo, hiEx = ctr->ownHiEx;
    const unsigned long long hi1 = usedHi1All;
    if (hi1) {
        const unsigned long long dlo = (unsigned long long)(uint32_t)~(uint32_t)usedLoInvAll << 9, dhi = (hi1 + 1) << 9;
        lo = dlo < lo ? dlo : lo; hiEx = dhi > hiEx ? dhi : hiEx;
    }
    if (hiEx + 512 >= tableSize) { lo = 0; hiEx = tableSize; }
    ctr->validLo = lo; ctr->validHiEx = hiEx;
}

// Slots of the valid range (+512 slots of defined contents past it, + the slack past the table) that no wavefront
// owned: [validLo, ownLo) and [ownHiEx, validHiEx + 512).
__global__ void __launch_bounds__(kBlock)
k_wave_fill_edges(uint64_t* __restrict__ table, const Counters* __restrict__ ctr, uint64_t tableSize, Gate gate)
{
    if (gate_closed(gate)) return;
    const ulonglong2 e = make_ulonglong2(kEmpty, kEmpty);
    ulonglong2* t2 = reinterpret_cast<ulonglong2*>(table);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock, t0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t hi = ctr->validHiEx + 512;
    hi = hi < tableSize ? hi : tableSize;
    const uint64_t a0 = ctr->validLo >> 1, a1 = ctr->ownLo >> 1;          // all bounds are even (granules / blocks)
    for (uint64_t v = a0 + t0; v < a1; v += stride) t2[v] = e;
    const uint64_t b0 = ctr->ownHiEx >> 1, b1 = hi >> 1;
    for (uint64_t v = b0 + t0; v < b1; v += stride) t2[v] = e;
    if (blockIdx.x == 0 && threadIdx.x < kTableSlack) table[tableSize + threadIdx.x] = kEmpty;
}


// ---- compact table: the check and the decision (see k_build_wave<COMPACT>) ----------------------------------------
// Seam c (between chunks c-1 and c): the tuples that left chunk c-1's range at its upper end must be exactly the ones
// chunk c let in from its shadow granule. One thread per seam; the lists hold a handful of entries.
__global__ void __launch_bounds__(kBlock)
k_wave_validate(const DeferredEntry* __restrict__ queue, const uint32_t* __restrict__ dcounts, const uint32_t* __restrict__ pcounts,
                uint32_t nChunks, uint32_t sliceLen, Counters* __restrict__ ctr, Gate gate)
{
    if (gate_closed(gate)) return;
    if (*reinterpret_cast<volatile unsigned long long*>(&ctr->compactFail)) return;    // counts of skipped chunks are not defined
    const uint32_t c = blockIdx.x * kBlock + threadIdx.x + 1;
    if (c >= nChunks) return;
    const uint32_t nOut = dcounts[c - 1], nIn = pcounts[c];
    bool ok = nOut == nIn && nIn <= kWvPredCap;
    if (ok && nIn) {
        const DeferredEntry* out = queue + (uint64_t)(c - 1) * sliceLen;
        const uint64_t* in = reinterpret_cast<const uint64_t*>(queue + (uint64_t)(c + 1) * sliceLen) - kWvPredCap;
        for (uint32_t i = 0; i < nOut && ok; ++i) {
            const uint64_t v = out[i].packed;
            bool found = false;
            for (uint32_t k = 0; k < nIn; ++k) found |= in[k] == v;
            ok = found;                            // index words are unique: equal counts + every element found = equal sets
        }
    }
    if (!ok) atomicOr(&ctr->compactFail, 16ull);
}

// One wavefront. All seams check out and nobody raised the flag: the table is in the compact format, its valid range is
// what the wavefronts owned. Otherwise: as if the compact build had never run -- counters back to zero, and the variant
// word set to the classic build that is enqueued behind this kernel, gated on it.
__global__ void k_wave_decide(Counters* __restrict__ ctr, uint64_t tableSize, uint32_t fallbackVariant, Gate gate)
{
    if (blockIdx.x != 0 || threadIdx.x >= 64 || gate_closed(gate)) return;
    static_assert(Counters::kShards == 64, "one shard per lane");
    const bool fail = ctr->compactFail != 0;
    if (fail) {
        ctr->shard[threadIdx.x] = Counters::Shard{};
        if (threadIdx.x == 0) {
            ctr->conflicts = 0; ctr->conflictSum = 0; ctr->inputSum = 0; ctr->badKeys = 0; ctr->deferred = 0; ctr->foreign = 0;
            ctr->usedLoInv = 0; ctr->usedHi1 = 0; ctr->ownLo = 0; ctr->ownHiEx = 0;
            ctr->tableFormat = kFormatSlots8;
            ctr->variant = fallbackVariant;
        }
        return;
    }
    if (threadIdx.x != 0) return;
    unsigned long long lo = ctr->ownLo, hiEx = ctr->ownHiEx;
    if (hiEx + 512 >= tableSize) { lo = 0; hiEx = tableSize; }
    ctr->validLo = lo; ctr->validHiEx = hiEx;
    ctr->tableFormat = kFormatKeys4;
}

// compact counterpart of k_wave_fill_edges: 4-byte empties over [validLo, ownLo) and [ownHiEx, validHiEx + 512) + slack
__global__ void __launch_bounds__(kBlock)
k_wave_fill_edges_keys(uint32_t* __restrict__ keys, const Counters* __restrict__ ctr, uint64_t tableSize, Gate gate)
{
    if (gate_closed(gate)) return;
    const uint4 e = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    uint4* t4 = reinterpret_cast<uint4*>(keys);
    const uint64_t stride = (uint64_t)gridDim.x * kBlock, t0 = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    uint64_t hi = ctr->validHiEx + 512;
    hi = hi < tableSize ? hi : tableSize;
    const uint64_t a0 = ctr->validLo >> 2, a1 = ctr->ownLo >> 2;          // all bounds are multiples of 128 (granules) or 512
    for (uint64_t v = a0 + t0; v < a1; v += stride) t4[v] = e;
    const uint64_t b0 = ctr->ownHiEx >> 2, b1 = hi >> 2;
    for (uint64_t v = b0 + t0; v < b1; v += stride) t4[v] = e;
    if (blockIdx.x == 0 && threadIdx.x < kTableSlack) keys[tableSize + threadIdx.x] = 0xFFFFFFFFu;
}

__global__ void k_set_variant(Counters* __restrict__ ctr, uint32_t v)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->variant = v;
}
void launch_set_variant(Counters* ctr, uint32_t v, hipStream_t s) { hipLaunchKernelGGL(k_set_variant, dim3(1), dim3(64), 0, s, ctr, v); }

// ---- host side ----------------------------------------------------------------------------------------------------
size_t wave_lds_bytes() { return kWvLdsBytes; }
bool wave_supported(uint64_t tableSize) { return tableSize >= (uint64_t)kWvWin; }
uint32_t wave_max_chunks(int nCU) { return (uint32_t)HJ_WV_WAVES_PER_CU * (uint32_t)HJ_WV_MAX_ROUNDS * (uint32_t)nCU; }
size_t wave_bounds_bytes(int nCU) { return (6 * (size_t)wave_max_chunks(nCU) + 4) * sizeof(uint32_t); }   // raw, bounds (+1), starts (+1), dcounts, ccounts, pcounts
bool wave_compact_supported(uint64_t tableSize, uint32_t probeLen)
{
    // the shadow granule must cover every slot a walk across the seam can start from
    return wave_supported(tableSize) && probeLen >= 1 && probeLen <= 32;
}
static uint64_t wave_chunk_len(uint64_t n, int nCU)
{
    // One chunk per resident wavefront is a single round of workgroups -- and the kernel then lasts as long as its
    // SLOWEST wavefront: the streams of 4096 wavefronts do not advance at the same rate (memory channels, neighbours on the
    // CU), and with a static split nobody takes over from a wavefront that is done. Measured: the kernel's time is
    // 97 us + 1.81 ms per 2^29 tuples from 2^27 to 2^30 -- a size-independent ~100 us of waiting for stragglers. So large
    // relations are cut into several rounds' worth of chunks and the hardware's workgroup dispatcher does the balancing
    // (a workgroup that finishes makes room for the next four chunks): 2^30 tuples in 4 x 4096 chunks 3.68 -> 3.49 ms,
    // in 8 x 4096 chunks 3.46 ms.
    // Chunks stay >= HJ_WV_MIN_CHUNK tuples, though: every chunk pays for its ring (fill, final flush) and for two seam
    // tiles, and at 2^27 two rounds of 16384-tuple chunks were 2 % SLOWER than one round, four rounds 6 % (32768 against
    // 65536 as the minimum: the same at 2^28, -2 % at 2^29, -1..3 % at 2^30; 16 rounds instead of 8: slower).
    const uint32_t resident = (uint32_t)HJ_WV_WAVES_PER_CU * (uint32_t)nCU;
    uint64_t rounds = n / ((uint64_t)resident * HJ_WV_MIN_CHUNK);
    rounds = rounds < 1 ? 1 : rounds > HJ_WV_MAX_ROUNDS ? HJ_WV_MAX_ROUNDS : rounds;
    const uint64_t chunks = (uint64_t)resident * rounds;
    uint64_t chunkLen = (n + chunks - 1) / chunks;
    chunkLen = (chunkLen + kWvTile - 1) / kWvTile * kWvTile + (uint64_t)kWvTile * HJ_WV_SKEW;
    return chunkLen < (uint64_t)kWvTile * 4 ? (uint64_t)kWvTile * 4 : chunkLen;
}
static uint64_t wave_slice_len(uint64_t chunkLen) { return chunkLen + kWvLook + kWvOverlap; }
size_t wave_queue_bytes(uint64_t n, int nCU)
{
    const uint64_t chunkLen = wave_chunk_len(n, nCU);
    return (size_t)(((n + chunkLen - 1) / chunkLen) * wave_slice_len(chunkLen) + 64) * sizeof(DeferredEntry);
}

WaveSlices wave_conflict_layout(uint64_t n, int nCU, void* boundsBuf)
{
    const uint32_t maxChunks = wave_max_chunks(nCU);
    const uint64_t chunkLen = wave_chunk_len(n, nCU);
    return WaveSlices{(uint32_t)((n + chunkLen - 1) / chunkLen), (uint32_t)wave_slice_len(chunkLen),
                      static_cast<const uint32_t*>(boundsBuf) + 4 * (size_t)maxChunks + 2};
}
const uint32_t* wave_bounds_ptr(int nCU, const void* boundsBuf) { return static_cast<const uint32_t*>(boundsBuf) + wave_max_chunks(nCU); }
size_t wave_conflict_bytes(uint64_t n, int nCU) { return wave_queue_bytes(n, nCU) / sizeof(DeferredEntry) * sizeof(uint64_t); }

hipError_t launch_build_wave(const void* R, bool key32, uint64_t n, uint32_t hshift, uint64_t* table, uint64_t tableSize,
                             uint32_t probeLen, uint64_t idxBase, ShardCheck sc, int nCU, void* boundsBuf, void* queueBuf,
                             Counters* ctr, Gate gate, int parts, hipEvent_t evPhaseA, hipStream_t s, uint64_t* htmConflicts,
                             int mode, uint32_t fallbackVariant, const KernelEvents* kev, bool htmRoute)
{
    const bool htm = htmConflicts != nullptr;
    const bool compact = mode == kWaveCompact;
    if (htm && (key32 || probeLen != 3 || sc.mask || compact)) return hipErrorInvalidValue;
    if (!key32 && hshift) return hipErrorInvalidValue;         // the kernel's tuple instances assume it
    const uint32_t maxChunks = wave_max_chunks(nCU);
    const uint64_t chunkLen = wave_chunk_len(n, nCU);
    static_assert(kWvTile * 4 > (int)(kWvLook + kWvOverlap), "a seam may move by less than the shortest chunk");
    static_assert(kWvShadow + kWvOverlap == (uint32_t)kWvTile, "shadow zone + head zone = the first tile of a compact chunk");
    const uint32_t sliceLen = (uint32_t)wave_slice_len(chunkLen);
    const uint32_t nChunks = (uint32_t)((n + chunkLen - 1) / chunkLen);
    uint32_t* const raw = static_cast<uint32_t*>(boundsBuf);
    uint32_t* const bounds = raw + maxChunks;                 // nChunks + 1 entries
    uint32_t* const starts = bounds + maxChunks + 1;          // nChunks + 1 entries
    uint32_t* const dcounts = starts + maxChunks + 1;
    uint32_t* const ccounts = dcounts + maxChunks;            // == wave_conflict_layout(...).counts
    uint32_t* const pcounts = ccounts + maxChunks;
    const uint32_t numGran = (uint32_t)(tableSize >> kGranShift);
    // workgroups a device of nCU compute units holds at once (the rotation of issue priorities goes by it)
    const uint32_t residentWG = (uint32_t)HJ_WV_WAVES_PER_CU * (uint32_t)(nCU > 0 ? nCU : 256) / (uint32_t)kWvWaves;
    hipError_t e;
    const dim3 gRaw((nChunks + 1 + kBlock / 64 - 1) / (kBlock / 64)), gMain((nChunks + kWvWaves - 1) / kWvWaves);
    if (parts & kWavePre) {
        if (htm) hipLaunchKernelGGL((k_wave_seams<false, true>), gRaw, dim3(kBlock), 0, s, R, n, (uint32_t)chunkLen, nChunks, tableSize - 1, hshift, starts, raw, gate);
        else if (key32) hipLaunchKernelGGL((k_wave_seams<true, false>), gRaw, dim3(kBlock), 0, s, R, n, (uint32_t)chunkLen, nChunks, tableSize - 1, hshift, starts, raw, gate);
        else hipLaunchKernelGGL((k_wave_seams<false, false>), gRaw, dim3(kBlock), 0, s, R, n, (uint32_t)chunkLen, nChunks, tableSize - 1, hshift, starts, raw, gate);
        hipLaunchKernelGGL(k_wave_bounds_scan, dim3((nChunks + kBlock - 1) / kBlock), dim3(kBlock), 0, s, raw, nChunks, numGran, bounds, gate);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (parts & kWaveMain) {
        if (kev && (e = hipEventRecord(kev->before, s)) != hipSuccess) return e;
#define HJ_WV_LAUNCH(K32, CHK, HTM, CMP)                                                                             \
    hipLaunchKernelGGL((k_build_wave<K32, CHK, HTM, CMP>), gMain, dim3(kWvThreads), kWvLdsBytes, s, R, n, sliceLen, \
                       nChunks, starts, bounds, table, tableSize - 1, hshift, probeLen, idxBase, sc,                         \
                       static_cast<DeferredEntry*>(queueBuf), dcounts, ctr, gate, htmConflicts, ccounts, residentWG, pcounts)
        if (htm) HJ_WV_LAUNCH(false, false, true, false);
        else if (compact) {
            if (sc.mask) { if (key32) HJ_WV_LAUNCH(true, true, false, true); else HJ_WV_LAUNCH(false, true, false, true); }
            else { if (key32) HJ_WV_LAUNCH(true, false, false, true); else HJ_WV_LAUNCH(false, false, false, true); }
        }
        else if (sc.mask) { if (key32) HJ_WV_LAUNCH(true, true, false, false); else HJ_WV_LAUNCH(false, true, false, false); }
        else { if (key32) HJ_WV_LAUNCH(true, false, false, false); else HJ_WV_LAUNCH(false, false, false, false); }
#undef HJ_WV_LAUNCH
        if (kev && (e = hipEventRecord(kev->after, s)) != hipSuccess) return e;
        if (compact) {
            // the seams check out or the classic build takes over (Counters::variant := fallbackVariant)
            hipLaunchKernelGGL(k_wave_validate, dim3((nChunks + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                               static_cast<const DeferredEntry*>(queueBuf), dcounts, pcounts, nChunks, sliceLen, ctr, gate);
            hipLaunchKernelGGL(k_wave_decide, dim3(1), dim3(64), 0, s, ctr, tableSize, fallbackVariant, gate);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if (evPhaseA && (e = hipEventRecord(evPhaseA, s)) != hipSuccess) return e;
    }
    if (!(parts & kWaveTail)) return hipSuccess;
    if (compact) {
        hipLaunchKernelGGL(k_wave_fill_edges_keys, dim3(512), dim3(kBlock), 0, s, reinterpret_cast<uint32_t*>(table), ctr, tableSize, gate);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_wave_finalize_range, dim3(1), dim3(64), 0, s, ctr, tableSize, gate);
    hipLaunchKernelGGL(k_wave_fill_edges, dim3(2048), dim3(kBlock), 0, s, table, ctr, tableSize, gate);
    const dim3 gDef((nChunks + kBlock / 64 - 1) / (kBlock / 64));   // one wavefront per slice
    if (htm) hipLaunchKernelGGL(k_wave_deferred<true>, gDef, dim3(kBlock), 0, s, static_cast<const DeferredEntry*>(queueBuf), dcounts,
                                nChunks, sliceLen, table, tableSize - 1, hshift, probeLen, ctr, gate, htmConflicts, ccounts,
                                htmRoute ? bounds : nullptr);
    else hipLaunchKernelGGL(k_wave_deferred<false>, gDef, dim3(kBlock), 0, s, static_cast<const DeferredEntry*>(queueBuf), dcounts,
                            nChunks, sliceLen, table, tableSize - 1, hshift, probeLen, ctr, gate, nullptr, nullptr, nullptr);
    return hipGetLastError();
}

}  // namespace hj

// hj_prj.hip -- radix-partitioned join (PRJ path) for gfx950 (MI355X).
//
// What this replaces (paths relative to anilshanbhag/HTM-HashJoin, mc/src/):
//   k_radix_hist     the histogram loops of parallel_radix_partition (:586-589) and
//                    radix_cluster (:418-421)
//   k_scan_*         the prefix / cursor computation (:592-617, :423-430)
//   k_radix_scatter  the scatter loops (:622-626, :433-437), here staged through LDS so
//                    every partition's tuples leave the CU as contiguous runs
//   k_prj_join       bucket_chaining_join (:231-283) with the probe loop the fork
//                    commented out (:259-276) restored; the table lives in LDS
//   k_radix_scatter_frag  the same two passes WITHOUT the histogram loops (the default for large
//                    relations): private fragments per chunk and bin, capacity checked, the exact
//                    kernels above enqueued behind as a gated fallback (see "histogram-free partitioning")
// Orchestration (prj_thread :808-1122: 6 pthread barriers, 2 task queues) becomes
// stream order between kernels.
//
// Partition function: HASH_BIT_MODULO(key, MASK, R) = (key & MASK) >> R on the low
// 32 bits of the tuple (tuple_t.key), pass 1 on bits [0, bits1), pass 2 on
// [bits1, bits1+bits2), bits1 = radixBits/2 as in prj_thread :814-816. The final
// partition id is (pass-1 bin << bits2) | pass-2 bin, the same order the
// reference's task queues visit them in.
//
// Layout: a pass works on "segments" (pass 1: the whole relation; pass 2: each
// pass-1 partition). Segments are cut into chunks of chunkLen tuples; a workgroup
// owns one chunk. Histogram entries are stored [segment][bin][chunk] so that one
// exclusive scan over the whole array yields every chunk's write cursor.

#include "hj_device.h"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace hj {

constexpr int kTile = 8192;                       // largest scatter tile (elements); chunk lengths are multiples of it
constexpr int kMaxFan = 256;                      // <= 8 radix bits per pass
constexpr uint32_t kJoinSlots = 32768;            // LDS table slots (uint32) = 128 KiB
constexpr uint32_t kJoinBlockTuples = 24576;      // R tuples per LDS build (load <= 0.75)
constexpr int kJoinThreads = 1024;
constexpr uint32_t kEmpty32 = 0xFFFFFFFFu;
constexpr Gate kNoGate{nullptr, 0ull};

struct PassParams {
    const uint32_t* segOff;     // [nSeg + 1] tuple offsets of the input segments
    uint32_t nSeg;
    uint32_t chunkLen;          // multiple of kTile
    const uint32_t* chunkBase;  // [nSeg + 1] exclusive prefix of chunks per segment
    uint32_t shift, fan;        // bin = ((key - bias) >> shift) & (fan - 1)
    uint32_t bias;              // 0, or 1 for the range split of 1-based DataGen keys (shard split only)
    uint32_t zeroBad;           // shard split only: a tuple with payload bits set counts and travels as key 0, which the
                                // receiving build reports (HJ_ERR_KEY_RANGE); PRJ reads the key word alone, as mc does
};

// chunkBase[s] = sum_{t<s} ceil(len_t / chunkLen). nSeg <= 256: one thread.
__global__ void k_chunk_base(const uint32_t* __restrict__ segOff, uint32_t nSeg, uint32_t chunkLen,
                             uint32_t* __restrict__ chunkBase)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t acc = 0;
        for (uint32_t s = 0; s < nSeg; ++s) {
            chunkBase[s] = acc;
            const uint32_t len = segOff[s + 1] - segOff[s];
            acc += (len + chunkLen - 1) / chunkLen;
        }
        chunkBase[nSeg] = acc;
    }
}

__global__ void k_init_seg(uint32_t* seg, uint32_t n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { seg[0] = 0; seg[1] = n; }
}

// Finds the segment of chunk c (largest s with chunkBase[s] <= c). Wave-uniform.
__device__ __forceinline__ uint32_t find_segment(const uint32_t* __restrict__ chunkBase, uint32_t nSeg, uint32_t c)
{
    uint32_t lo = 0, hi = nSeg;  // invariant: chunkBase[lo] <= c < chunkBase[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (chunkBase[mid] <= c) lo = mid; else hi = mid;
    }
    return lo;
}

struct ChunkRange { uint32_t seg, local, nLocal, begin, end; };

__device__ __forceinline__ ChunkRange chunk_range(const PassParams& p, uint32_t c)
{
    ChunkRange r;
    r.seg = find_segment(p.chunkBase, p.nSeg, c);
    r.local = c - p.chunkBase[r.seg];
    r.nLocal = p.chunkBase[r.seg + 1] - p.chunkBase[r.seg];
    const uint32_t sb = p.segOff[r.seg], se = p.segOff[r.seg + 1];
    r.begin = sb + r.local * p.chunkLen;
    const uint64_t e = (uint64_t)r.begin + p.chunkLen;
    r.end = e < se ? (uint32_t)e : se;
    return r;
}

// hist index of (segment, bin, local chunk)
__device__ __forceinline__ uint64_t hist_index(const PassParams& p, const ChunkRange& r, uint32_t bin)
{
    return (uint64_t)p.chunkBase[r.seg] * p.fan + (uint64_t)bin * r.nLocal + r.local;
}

// ---------------------------------------------------------------------------
// element formats
//   IN32 / OUT32 = false: 8-byte DataGen tuples (value = key, tuple_t{key,payload});
//   = true: bare 32-bit keys. The join consumes keys only (bucket_chaining_join never reads
//   the payload, :247-256 and the commented probe :264-275), so pass 1 of the PRJ path writes
//   the key word alone and everything after it moves 4 bytes per tuple instead of 8.
//   Inputs are read as 16-byte vectors (2 tuples or 4 keys).
// ---------------------------------------------------------------------------
template <bool IN32> struct Fmt { static constexpr uint32_t EPV = IN32 ? 4u : 2u; };   // elements per vector

template <bool IN32, int E>
__device__ __forceinline__ uint32_t vec_key(const uint4& t)
{
    if constexpr (IN32) return E == 0 ? t.x : E == 1 ? t.y : E == 2 ? t.z : t.w;
    else return E == 0 ? t.x : t.z;
}
template <int E>
__device__ __forceinline__ uint64_t vec_tuple(const uint4& t)      // 8-byte formats only
{
    return E == 0 ? (((uint64_t)t.y << 32) | t.x) : (((uint64_t)t.w << 32) | t.z);
}

// ---------------------------------------------------------------------------
// histogram
// ---------------------------------------------------------------------------
template <bool IN32>
__global__ void __launch_bounds__(kBlock)
k_radix_hist(const void* __restrict__ in, PassParams p, uint32_t* __restrict__ hist, Gate gate)
{
    constexpr uint32_t EPV = Fmt<IN32>::EPV;
    __shared__ unsigned int h[kMaxFan];
    if (gate_closed(gate)) return;
    const uint32_t c = blockIdx.x;
    if (c >= p.chunkBase[p.nSeg]) return;
    const ChunkRange r = chunk_range(p, c);
    if (threadIdx.x < kMaxFan) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t fmask = p.fan - 1;
    // 16-byte aligned sweep: start at the vector that holds element `begin`
    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    if (p.fan <= 16) {
        // few bins (the multi-GPU split: fan = number of shards): 64 lanes adding to a handful of LDS words
        // serialise on the address (measured: 2.4 TB/s at fan 1 against 5.8 TB/s at fan 256), so count with
        // ballots into wave-uniform registers instead and add once per wavefront at the end
        uint32_t wc[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) wc[b] = 0;
        bool live = false;
        auto tally = [&](uint32_t key, uint32_t at) {
            const bool ok = live && at >= r.begin && at < r.end;
            const uint32_t bin = ((key - p.bias) >> p.shift) & fmask;
#pragma unroll
            for (int b = 0; b < 16; ++b)
                if (b < (int)p.fan) wc[b] += (uint32_t)__popcll(__ballot(ok && bin == (uint32_t)b));
        };
        const uint32_t vEnd = (uint32_t)(((uint64_t)r.end + EPV - 1) / EPV);
        // whole wavefronts iterate together (ballots need every lane): round the trip count up
        // four 16-byte loads in flight per lane before the first ballot (one load per trip left the kernel at 4.0 TB/s:
        // a wavefront had 1 KiB outstanding)
        constexpr int U = 4;
        for (uint32_t v0 = r.begin / EPV; v0 < vEnd; v0 += U * kBlock) {
            uint4 t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t v = v0 + (uint32_t)u * kBlock + threadIdx.x;
                t[u] = in4[v < vEnd ? v : vEnd - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t v = v0 + (uint32_t)u * kBlock + threadIdx.x;
                live = v < vEnd;
                const uint32_t i = v * EPV;
                if constexpr (IN32) {
                    tally(t[u].x, i); tally(t[u].y, i + 1); tally(t[u].z, i + 2); tally(t[u].w, i + 3);
                } else {
                    tally((p.zeroBad && t[u].y) ? 0u : t[u].x, i);
                    tally((p.zeroBad && t[u].w) ? 0u : t[u].z, i + 1);
                }
            }
        }
        if ((threadIdx.x & 63) == 0) {
#pragma unroll
            for (int b = 0; b < 16; ++b)
                if (b < (int)p.fan && wc[b]) atomicAdd(&h[b], wc[b]);
        }
        __syncthreads();
        if (threadIdx.x < p.fan) hist[hist_index(p, r, threadIdx.x)] = h[threadIdx.x];
        return;
    }
    for (uint32_t v = r.begin / EPV + threadIdx.x; (uint64_t)v * EPV < r.end; v += kBlock) {
        const uint4 t = in4[v];
        const uint32_t i = v * EPV;
        auto one = [&](uint32_t key, uint32_t at) {
            if (at >= r.begin && at < r.end) atomicAdd(&h[((key - p.bias) >> p.shift) & fmask], 1u);
        };
        if constexpr (IN32) {
            one(t.x, i); one(t.y, i + 1); one(t.z, i + 2); one(t.w, i + 3);
        } else {
            one((p.zeroBad && t.y) ? 0u : t.x, i);
            one((p.zeroBad && t.w) ? 0u : t.z, i + 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < p.fan) hist[hist_index(p, r, threadIdx.x)] = h[threadIdx.x];
}

// ---------------------------------------------------------------------------
// exclusive scan of a uint32 array (3 kernels: block scan, scan of block sums, add)
// ---------------------------------------------------------------------------
constexpr int kScanPerThread = 16;
constexpr int kScanTile = kBlock * kScanPerThread;  // 4096

template <int NT = kBlock>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wsum /*[NT/64]*/, uint32_t& total)
{
    // inclusive scan inside the wavefront
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t n = __shfl_up(inc, off, 64);
        if (lane >= off) inc += n;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t wbase = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; ++k) {
        const uint32_t s = wsum[k];
        if (k < w) wbase += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return wbase + inc - v;
}

__global__ void __launch_bounds__(kBlock)
k_scan_blocks(uint32_t* __restrict__ data, uint64_t n, uint32_t* __restrict__ blockSums, Gate gate)
{
    __shared__ uint32_t wsum[kBlock / 64];
    if (gate_closed(gate)) return;
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPerThread;
    uint32_t v[kScanPerThread], local = 0;
    // a thread's 16 values are 64 contiguous, 64-byte aligned bytes: four 16-byte loads / stores when they all exist
    // (one dword per instruction made every wave-instruction touch 64 different lines: 0.79 ms for the 2^26 bucket
    // counts of the bucketised table at 2^27)
    const bool whole = base + kScanPerThread <= n;                   // (hipMalloc'ed arrays: data is 256-byte aligned)
    if (whole) {
        const uint4* p4 = reinterpret_cast<const uint4*>(data + base);
#pragma unroll
        for (int q = 0; q < kScanPerThread / 4; ++q) {
            const uint4 t = p4[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) v[k] = (base + k < n) ? data[base + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) local += v[k];
    uint32_t total;
    uint32_t ex = block_exclusive_scan(local, wsum, total);
    if (whole) {
        uint4* p4 = reinterpret_cast<uint4*>(data + base);
#pragma unroll
        for (int q = 0; q < kScanPerThread / 4; ++q) {
            uint4 t;
            t.x = ex; ex += v[4 * q]; t.y = ex; ex += v[4 * q + 1]; t.z = ex; ex += v[4 * q + 2]; t.w = ex; ex += v[4 * q + 3];
            p4[q] = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < kScanPerThread; ++k) {
            if (base + k < n) data[base + k] = ex;
            ex += v[k];
        }
    }
    if (threadIdx.x == 0) blockSums[blockIdx.x] = total;
}

__global__ void __launch_bounds__(kBlock)
k_scan_sums(uint32_t* __restrict__ sums, uint32_t m, Gate gate)
{
    __shared__ uint32_t wsum[kBlock / 64];
    if (gate_closed(gate)) return;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < m; base += kBlock) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < m ? sums[i] : 0;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan(v, wsum, total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
}

__global__ void __launch_bounds__(kBlock)
k_scan_add(uint32_t* __restrict__ data, uint64_t n, const uint32_t* __restrict__ blockSums, Gate gate)
{
    if (gate_closed(gate)) return;
    const uint32_t add = blockSums[blockIdx.x];
    const uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)threadIdx.x * kScanPerThread;
    if (base + kScanPerThread <= n) {
        uint4* p4 = reinterpret_cast<uint4*>(data + base);
#pragma unroll
        for (int q = 0; q < kScanPerThread / 4; ++q) {
            uint4 t = p4[q];
            t.x += add; t.y += add; t.z += add; t.w += add;
            p4[q] = t;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k)
        if (base + k < n) data[base + k] += add;
}

// New segment offsets after a pass: segOut[s*fan + bin] = start of that bin.
__global__ void __launch_bounds__(kBlock)
k_seg_offsets(PassParams p, const uint32_t* __restrict__ scanned, uint32_t nTotal, uint32_t* __restrict__ segOut)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t total = p.nSeg * p.fan;
    if (i == total) segOut[i] = nTotal;
    if (i >= total) return;
    const uint32_t s = i / p.fan, bin = i - s * p.fan;
    const uint32_t nLocal = p.chunkBase[s + 1] - p.chunkBase[s];
    // an empty segment has no histogram entries: all its bins start where it starts
    segOut[i] = nLocal ? scanned[(uint64_t)p.chunkBase[s] * p.fan + (uint64_t)bin * nLocal] : p.segOff[s];
}

// ---------------------------------------------------------------------------
// scatter: tile-local counting sort in LDS, then contiguous runs to HBM
//   NT threads, TV 16-byte loads per thread and tile -> tile = NT * TV * EPV elements.
//   Instances: pass 1 of PRJ   tuples -> keys, 512 x 8 x 2 = 8192 per tile (128-byte runs at fan 256)
//              pass 2 of PRJ   keys -> keys,   512 x 4 x 4 = 8192 per tile
//              shard scatter   the pass-1 instance with fan-out = number of shards
// ---------------------------------------------------------------------------
template <bool IN32, bool OUT32, int NT, int TV, bool PF = true, int WPE = 1>
__global__ void __launch_bounds__(NT, WPE)
k_radix_scatter(const void* __restrict__ in, void* __restrict__ outv, PassParams p,
                const uint32_t* __restrict__ scanned, Gate gate)
{
    static_assert(!(IN32 && !OUT32), "keys cannot become tuples again");
    constexpr uint32_t EPV = Fmt<IN32>::EPV;
    constexpr int E = (int)EPV * TV;                 // elements per thread per tile
    constexpr uint32_t TILE = (uint32_t)NT * E;
    static_assert(!PF || E == 16, "vmcnt immediate below assumes 16 stores per thread per full tile");
    static_assert(TILE <= 65536 && E % 2 == 0, "ranks are kept in 16 bits, two to a register");
    using OutT = typename std::conditional<OUT32, uint32_t, uint64_t>::type;
    OutT* __restrict__ out = static_cast<OutT*>(outv);

    // Staged position q lives at stage[q + (q >> kPadShift)]: one pad element per 32 banks' worth. Without it,
    // equal-sized bins (dense keys: every bin of a tile holds exactly TILE/fan elements) start at the same
    // bank and lanes, whose ranks advance in step, collide (PMC: 80 % of the LDS cycles were bank conflicts).
    constexpr uint32_t kPadShift = OUT32 ? 5 : 4;
    constexpr uint32_t kDump = TILE + (TILE >> kPadShift);    // slot for elements outside the chunk
    __shared__ OutT stage[kDump + 1];
    __shared__ unsigned int tileCnt[kMaxFan];    // per bin: elements of this tile
    __shared__ unsigned int tileOff[kMaxFan];    // exclusive scan of tileCnt
    __shared__ unsigned int delta[kMaxFan];      // write cursor of the bin - tileOff: output index = delta[bin] + staged position
    __shared__ unsigned int cursor[kMaxFan];
    __shared__ unsigned int sValid;

    if (gate_closed(gate)) return;
    const uint32_t c = blockIdx.x;
    if (c >= p.chunkBase[p.nSeg]) return;
    const ChunkRange r = chunk_range(p, c);
    const uint32_t fmask = p.fan - 1;
    const uint32_t len = r.end - r.begin;
    if (threadIdx.x < kMaxFan) {
        tileCnt[threadIdx.x] = 0;
        cursor[threadIdx.x] = threadIdx.x < p.fan ? scanned[hist_index(p, r, threadIdx.x)] : 0;
    }
    __syncthreads();

    const uint4* in4 = reinterpret_cast<const uint4*>(in);
    const uint32_t b0 = r.begin & ~(EPV - 1);
    // Register prefetch of the next tile (one unconditional, clamped load path; waits placed by hand: see
    // hj_build_own.hip). vmcnt counts loads and stores together in issue order: after a full tile every
    // thread has exactly E stores younger than the prefetch loads, so vmcnt(E) waits for the loads only
    // and leaves the stores in flight.
    const uint32_t lastVec = (r.end - 1) / EPV;
    uint4 nxt[TV];
    auto issue = [&](uint64_t tb) {
#pragma unroll
        for (int k = 0; k < TV; ++k) {
            const uint64_t v = tb / EPV + (uint64_t)k * NT + threadIdx.x;
            nxt[k] = in4[v < lastVec ? v : lastVec];
        }
    };
    if (PF) issue(b0);
    bool prevFull = false;
    // Three barriers per tile. The first version of this loop had eight, and a range-check branch around
    // every LDS atomic, which made each of the 16 returning atomics of a thread its own round trip
    // (s_waitcnt lgkmcnt(0) inside every branch): 3.6 ms per 2^30-tuple pass. Now every lane ranks every
    // element unconditionally (one outside the chunk adds 0 and is staged to the dump slot).
    for (uint64_t tb = b0; tb < r.end; tb += TILE) {
        uint4 cur[TV];
        if (PF) {
            if (prevFull) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < TV; ++k) cur[k] = nxt[k];
            __builtin_amdgcn_sched_barrier(0);
            issue(tb + TILE);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            issue(tb);
#pragma unroll
            for (int k = 0; k < TV; ++k) cur[k] = nxt[k];
        }
        // ---- rank: position of every element inside its (tile, bin) ----
        OutT tv[E];
        uint32_t br[E];      // bin << 16 | rank
        uint32_t okMask = 0;
        // offset from the chunk's first element; wraps below zero (= huge) for the elements before `begin`
        const uint32_t rel0 = (uint32_t)tb - r.begin + EPV * threadIdx.x;
        auto one = [&](int slot, uint32_t key, uint64_t tuple, uint32_t rel) {
            const uint32_t bin = (key >> p.shift) & fmask;
            const bool ok = rel < len;
            if constexpr (OUT32) tv[slot] = key;
            else tv[slot] = tuple;
            okMask |= ok ? (1u << slot) : 0u;
            br[slot] = (bin << 16) | atomicAdd(&tileCnt[bin], ok ? 1u : 0u);
        };
#pragma unroll
        for (int k = 0; k < TV; ++k) {
            const uint32_t rel = rel0 + EPV * (uint32_t)k * NT;
            const uint4 t = cur[k];
            if constexpr (IN32) {
                one(4 * k, t.x, 0, rel); one(4 * k + 1, t.y, 0, rel + 1); one(4 * k + 2, t.z, 0, rel + 2); one(4 * k + 3, t.w, 0, rel + 3);
            } else {
                one(2 * k, t.x, vec_tuple<0>(t), rel); one(2 * k + 1, t.z, vec_tuple<1>(t), rel + 1);
            }
        }
        __syncthreads();
        // ---- one wavefront: exclusive scan of the 256 counters (4 per lane), cursors, reset ----
        if (threadIdx.x < 64) {
            const uint32_t b4 = 4 * threadIdx.x;
            uint32_t cnt[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cnt[j] = tileCnt[b4 + j]; tileCnt[b4 + j] = 0; sum += cnt[j]; }
            uint32_t inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t n = __shfl_up(inc, off, 64);
                if ((int)threadIdx.x >= off) inc += n;
            }
            uint32_t ex = inc - sum;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t cu = cursor[b4 + j];
                tileOff[b4 + j] = ex;
                delta[b4 + j] = cu - ex;
                cursor[b4 + j] = cu + cnt[j];
                ex += cnt[j];
            }
            if (threadIdx.x == 63) sValid = inc;
        }
        __syncthreads();
        // ---- stage: tile-local counting sort ----
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const uint32_t at = tileOff[br[k] >> 16] + (br[k] & 0xFFFFu);
            stage[((okMask >> k) & 1u) ? at + (at >> kPadShift) : kDump] = tv[k];
        }
        __syncthreads();
        // ---- contiguous runs to HBM ----
        const uint32_t valid = sValid;
        auto emit = [&](uint32_t q) {
            const OutT t = stage[q + (q >> kPadShift)];
            const uint32_t bin = ((uint32_t)t >> p.shift) & fmask;
            out[delta[bin] + q] = t;
        };
        if (valid == TILE) {
#pragma unroll
            for (int k = 0; k < E; ++k) emit((uint32_t)k * NT + threadIdx.x);
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const uint32_t q = (uint32_t)k * NT + threadIdx.x;
                if (q < valid) emit(q);
            }
        }
        prevFull = (valid == TILE);
        // (the next tile's atomics touch only tileCnt, reset above; its scan step, which rewrites
        //  tileOff/delta, runs after a barrier every wavefront reaches only when it has finished this phase)
    }
}

// ---------------------------------------------------------------------------
// histogram-free partitioning
//   The exact passes above read everything twice: once to count, once to move (parallel_radix_partition :586-626 does
//   the same). The counting exists only to make every partition one dense run. Here a pass does not count: chunk c of
//   segment s owns, for every bin b, a private FRAGMENT of `outCap` key slots at ((s * fan + b) * C + c) * outCap and
//   fills it from the front; what it wrote is recorded per fragment (outCnt). A partition is then C fragments with
//   holes between them, which the next pass (reading positions, masking the holes by the counts) and the join (one
//   fragment at a time) walk without ever compacting. outCap = mean + 7 sigma of a Poisson count (host: frag_cap), so
//   DataGen-shaped inputs -- dense or uniformly drawn keys, whose low bits are uniform in every contiguous piece of
//   the relation -- fit; a fragment that would overflow sets Counters::prjFallback, every kernel of this path
//   returns at once when it sees the word set, and the exact passes, gated on the same word, redo the join.
//   No atomics, no scan, no second read: 12 B (pass 1) and 8 B (pass 2) per tuple instead of 20 and 12.
// ---------------------------------------------------------------------------
constexpr uint32_t kMaxInFrags = 1024;     // pass-1 fragments one pass-2 chunk may span (C1 <= 1024)
// workgroup geometry of the two passes: threads, elements per thread and tile, waves per SIMD the registers must allow
#ifndef HJ_FRAG1_NT
#define HJ_FRAG1_NT 1024
#define HJ_FRAG1_E 32
#define HJ_FRAG1_WPE 4
#endif
#ifndef HJ_FRAG1_LANECOL
#define HJ_FRAG1_LANECOL 1                    // pass 1 ranks into per-(bin, lane column) counters (k_radix_scatter_frag)
#endif
#ifndef HJ_FRAG2_NT
#define HJ_FRAG2_NT 512
#define HJ_FRAG2_E 16
#define HJ_FRAG2_WPE 4
#endif

struct FragPass {
    uint32_t C;              // chunks per input segment = fragments per output partition
    uint32_t shift, fan;
    uint32_t outCap;         // key slots per output fragment (multiple of 32: whole 128-byte lines)
    uint32_t* outCnt;        // [nSeg * fan * C] keys written per output fragment
    // pass 1 (tuples in): one segment, the dense relation
    uint32_t n, chunkLen;
    // pass 2 (keys in): segment s = the inFrags * C fragments of pass-1 bin s, chunk c of it = fragments [c * inFrags, ..)
    uint32_t inFrags, inCap;
    const uint32_t* inCnt;
};

// LANECOL (pass 1, tiles of 32768 keys): the ranking counters are kept per (bin, lane & 15) -- 16 columns per bin, so
// the bank of a lane's counter is ((bin & 1) << 4 | lane & 15): whatever bins the 32 lanes of a group hold, at most two of
// them meet on a bank, which a 4-cycle LDS atomic hides (MI355X_MICROARCH.md, LDS). Ranking 64 shuffled keys into one
// counter per bin put 3-5 lanes on the busiest bank and serialised equal bins (r03_prj_pmc_sq.txt: 244 M conflict cycles
// of 416 M LDS-active on the shuffled R against 104 M of 276 M on the sorted S). One word per counter: the count of the
// current tile in the low half (<= 2048: 64 threads x 32 keys share a column), the counter's first staged position --
// the exclusive scan over (bin, column), written by the scan that also resets the count -- in the high half (< 65536).
// The scan is done by all threads (4 counters each) instead of one wavefront over 256 bins.
template <bool IN32, int NT, int E = 16, int WPE = 4, bool LANECOL = false>         // E = elements per thread and tile
__global__ void __launch_bounds__(NT, WPE)
k_radix_scatter_frag(const void* __restrict__ in, uint32_t* __restrict__ out, FragPass p, Counters* __restrict__ ctr)
{
    constexpr uint32_t TILE = (uint32_t)NT * E;
    static_assert(TILE <= 65536 && E % 2 == 0, "ranks are kept in 16 bits, two to a register");
    static_assert(!LANECOL || (NT == 4 * kMaxFan && TILE < 65536), "LANECOL: four counters per thread, positions in 16 bits");
    constexpr uint32_t kPadShift = 5;                         // see k_radix_scatter
    constexpr uint32_t kDump = TILE + (TILE >> kPadShift);
    __shared__ uint32_t stage[kDump + 1];
    __shared__ unsigned int tileCnt[LANECOL ? 1 : kMaxFan], tileOff[LANECOL ? 1 : kMaxFan], delta[kMaxFan], cursor[kMaxFan];
    __shared__ uint4 colCnt[LANECOL ? kMaxFan * 4 : 1];       // LANECOL: [bin][16 columns], as uint4 for the scan
    __shared__ unsigned int sValid, sAbort;
    __shared__ uint32_t prefix[IN32 ? kMaxInFrags + 1 : 1];   // pass 2: keys of this chunk before input fragment i
    __shared__ uint32_t wsum[NT / 64];

    // The fallback word is set by OTHER workgroups of this very launch (below): every wavefront reading it for itself
    // could split a workgroup -- some wavefronts gone, the rest scanning with stale wsum[] entries, and pass 2 would then
    // read through garbage prefixes. One thread reads, the workgroup decides together (round-2 ADVICE).
    if (threadIdx.x == 0) sAbort = *reinterpret_cast<volatile unsigned long long*>(&ctr->prjFallback) != 0;
    __syncthreads();
    if (sAbort) return;
    const uint32_t c = blockIdx.x;
    const uint32_t seg = c / p.C, local = c - seg * p.C;
    const uint32_t fmask = p.fan - 1;
    // first slot of this chunk's fragment of bin b
    auto frag_start = [&](uint32_t b) { return ((seg * p.fan + b) * p.C + local) * p.outCap; };

    // ---- what this chunk reads: `total` elements, element d at ... ----
    uint32_t total;                                            // elements of the chunk
    uint32_t begin = 0;                                        // pass 1: first tuple
    const uint32_t firstFrag = (seg * p.C + local) * p.inFrags;   // pass 2: first input fragment
    if constexpr (!IN32) {
        begin = local * p.chunkLen;
        const uint32_t end = begin + p.chunkLen < p.n ? begin + p.chunkLen : p.n;   // host: n <= 2^31
        total = begin < end ? end - begin : 0u;
    } else {
        // The holes between the input fragments are skipped, not masked: tiles are full (their runs whole 128-byte
        // lines on dense keys), and nothing is read that is not a key. prefix[] = exclusive scan of the counts.
        uint32_t carry = 0;
        for (uint32_t base = 0; base < p.inFrags; base += NT) {
            const uint32_t i = base + threadIdx.x;
            const uint32_t v = i < p.inFrags ? p.inCnt[firstFrag + i] : 0u;
            uint32_t sum;
            const uint32_t ex = block_exclusive_scan<NT>(v, wsum, sum);
            if (i < p.inFrags) prefix[i] = carry + ex;
            carry += sum;
        }
        if (threadIdx.x == 0) prefix[p.inFrags] = carry;
        total = carry;
        // counts that cannot be (more keys than the input fragments hold): never index with them
        if (total > p.inFrags * p.inCap) {
            if (threadIdx.x == 0) atomicExch(&ctr->prjFallback, 1ull);
            return;                                             // workgroup-uniform: carry is the same in every thread
        }
    }
    if (total == 0) {                                           // the last chunks of a short relation
        if (threadIdx.x < p.fan) p.outCnt[(seg * p.fan + threadIdx.x) * p.C + local] = 0;
        return;
    }
    if (threadIdx.x < kMaxFan) {
        if constexpr (!LANECOL) tileCnt[threadIdx.x] = 0;
        cursor[threadIdx.x] = threadIdx.x < p.fan ? frag_start(threadIdx.x) : 0;
    }
    if constexpr (LANECOL) colCnt[threadIdx.x] = uint4{0u, 0u, 0u, 0u};
    if (threadIdx.x == 0) sAbort = 0;
    __syncthreads();

    const uint32_t* in1 = reinterpret_cast<const uint32_t*>(in);
    uint32_t fA = 0;                                            // pass 2: input fragment the tile starts in
    uint32_t tileNo = 0;
    for (uint32_t T = 0; T < total; T += TILE, ++tileNo) {
        // the thread index is made opaque per tile: everything derived from it (16 positions, LDS addresses, offsets) is
        // one add away, and hoisted out of the loop it only fills registers until they spill
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        unsigned int* const colWord = reinterpret_cast<unsigned int*>(colCnt) + (tid & 15u);   // LANECOL: + 16 * bin
        uint32_t tv[E], okMask = 0;
        if constexpr (!IN32) {
            // one register per element instead of two (the payload word is never looked at, as in mc: the partition
            // function and the join read tuple_t.key only); lanes read every second word of contiguous lines
            // (a chunk is at most 2^21 tuples -- frag_geometry --, so byte offsets from the chunk's base fit 32 bits:
            // scalar base + one offset register per load)
            const char* const base = static_cast<const char*>(in) + 8ull * (begin + T);
            if (T + TILE <= total) {                            // a full tile: scalar base per load, one shared offset register
#pragma unroll
                for (int j = 0; j < E; ++j)
                    tv[j] = *reinterpret_cast<const uint32_t*>(base + (size_t)j * NT * 8 + (tid << 3));
                okMask = E == 32 ? 0xFFFFFFFFu : (1u << (E & 31)) - 1u;
            } else {
                const uint32_t left = total - T;                // >= 1
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const uint32_t d = (uint32_t)j * NT + tid;
                    const bool ok = d < left;
                    tv[j] = *reinterpret_cast<const uint32_t*>(base + ((ok ? d : left - 1) << 3));
                    okMask |= (ok ? 1u : 0u) << j;
                }
            }
        } else {
            // element d of the chunk lives in input fragment f (prefix[f] <= d < prefix[f+1]) at slot d - prefix[f].
            // A thread's elements d = T + j * NT + thread grow with j, so it walks the fragments once per tile: the
            // current fragment's end and address offset stay in registers, and a step to the next fragment (a tile of
            // 32768 keys meets ~8 fragments of 4096) costs two LDS reads.
            while (fA + 1 < p.inFrags && prefix[fA + 1] <= T) ++fA;
            uint32_t f = fA;
            uint32_t bound = f + 1 < p.inFrags ? prefix[f + 1] : 0xFFFFFFFFu;
            uint32_t off = (firstFrag + f) * p.inCap - prefix[f];
#pragma unroll
            for (int j = 0; j < E; ++j) {
                const uint32_t d = T + (uint32_t)j * NT + tid;
                const bool ok = d < total;
                while (ok && d >= bound) {
                    ++f;
                    bound = f + 1 < p.inFrags ? prefix[f + 1] : 0xFFFFFFFFu;
                    off = (firstFrag + f) * p.inCap - prefix[f];
                }
                tv[j] = in1[ok ? d + off : firstFrag * p.inCap];
                okMask |= (ok ? 1u : 0u) << j;
            }
        }
        // ---- rank ----
        uint32_t rk[E / 2];                                     // two 16-bit ranks per register; the bin is recomputed from the key
#pragma unroll
        for (int j = 0; j < E; ++j) {
            const uint32_t bin = (tv[j] >> p.shift) & fmask;
            uint32_t r;
            if constexpr (LANECOL) r = atomicAdd(&colWord[bin << 4], (okMask >> j) & 1u) & 0xFFFFu;
            else r = atomicAdd(&tileCnt[bin], (okMask >> j) & 1u);
            if (j & 1) rk[j / 2] |= r << 16; else rk[j / 2] = r;
        }
        __syncthreads();
        if constexpr (LANECOL) {
            // ---- all threads: scan of the 4096 counters (thread t: bin t / 4, columns 4 (t % 4) ..), cursors, capacity check ----
            const uint4 w = colCnt[tid];
            const uint32_t c0 = w.x & 0xFFFFu, c1 = w.y & 0xFFFFu, c2 = w.z & 0xFFFFu, c3 = w.w & 0xFFFFu;
            const uint32_t mine = c0 + c1 + c2 + c3;
            uint32_t tot;
            const uint32_t ex = block_exclusive_scan<NT>(mine, wsum, tot);
            colCnt[tid] = uint4{ex << 16, (ex + c0) << 16, (ex + c0 + c1) << 16, (ex + c0 + c1 + c2) << 16};
            const uint32_t endOfBin = (uint32_t)__shfl_down((int)(ex + mine), 3, 64);     // the four threads of a bin share a wavefront
            if ((tid & 3u) == 0) {
                const uint32_t b = tid >> 2, cnt = endOfBin - ex;
                const uint32_t cu = cursor[b];
                delta[b] = cu - ex;
                cursor[b] = cu + cnt;
                if (b < p.fan && cu + cnt - frag_start(b) > p.outCap) { sAbort = 1; atomicExch(&ctr->prjFallback, 1ull); }
            }
            if (tid == 0) {
                sValid = tot;
                // someone else's overflow: looked at every 16th tile
                if ((tileNo & 15u) == 15u && *reinterpret_cast<volatile unsigned long long*>(&ctr->prjFallback) != 0) sAbort = 1;
            }
        } else
        // ---- one wavefront: scan of the counters, cursors, capacity check ----
        if (tid < 64) {
            const uint32_t b4 = 4 * tid;
            uint32_t cnt[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cnt[j] = tileCnt[b4 + j]; tileCnt[b4 + j] = 0; sum += cnt[j]; }
            uint32_t inc = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t n = __shfl_up(inc, off, 64);
                if ((int)tid >= off) inc += n;
            }
            uint32_t ex = inc - sum;
            bool over = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t cu = cursor[b4 + j];
                tileOff[b4 + j] = ex;
                delta[b4 + j] = cu - ex;
                cursor[b4 + j] = cu + cnt[j];
                over |= (b4 + j < p.fan) && (cu + cnt[j] - frag_start(b4 + j) > p.outCap);
                ex += cnt[j];
            }
            // someone else's overflow: looked at every 16th tile (one wave-uniform load of a word nobody writes otherwise)
            bool stop = __any(over);
            if (!stop && (tileNo & 15u) == 15u) stop = *reinterpret_cast<volatile unsigned long long*>(&ctr->prjFallback) != 0;
            if (tid == 63) { sValid = inc; if (stop) sAbort = 1; }
            if (tid == 0 && __any(over)) atomicExch(&ctr->prjFallback, 1ull);
        }
        __syncthreads();
        if (sAbort) return;                                     // nothing of this path is used any more
        // ---- stage ----
#pragma unroll
        for (int k = 0; k < E; ++k) {
            const uint32_t binK = (tv[k] >> p.shift) & fmask;
            const uint32_t at = (LANECOL ? colWord[binK << 4] >> 16 : tileOff[binK]) + ((rk[k / 2] >> (16 * (k & 1))) & 0xFFFFu);
            stage[((okMask >> k) & 1u) ? at + (at >> kPadShift) : kDump] = tv[k];
        }
        __syncthreads();
        // ---- contiguous runs to HBM ----
        const uint32_t valid = sValid;
        // staged position q = k * NT + thread lives at q + (q >> 5) = (thread + (thread >> 5)) + k * (NT + NT / 32): one
        // address register and an immediate per k
        static_assert(NT % 32 == 0, "the pad of a staged position splits into a thread part and a k part");
        const uint32_t* const myStage = stage + (tid + (tid >> kPadShift));
        auto emit = [&](int k) {
            const uint32_t t = myStage[k * (NT + (NT >> kPadShift))];
            out[delta[(t >> p.shift) & fmask] + (uint32_t)k * NT + tid] = t;
        };
        if (valid == TILE) {
#pragma unroll
            for (int k = 0; k < E; ++k) emit(k);
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k)
                if ((uint32_t)k * NT + tid < valid) emit(k);
        }
    }
    __syncthreads();
    if (threadIdx.x < p.fan) p.outCnt[(seg * p.fan + threadIdx.x) * p.C + local] = cursor[threadIdx.x] - frag_start(threadIdx.x);
}

// ---------------------------------------------------------------------------
// per-partition join in LDS
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t next_pow2_u32(uint32_t v)
{
    // NEXT_POW_2, parallel_radix_join.c:66-76
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    return v;
}

// Slot of key-remainder k in the LDS table. NOT the identity: inside a partition the remainders of
// dense keys are consecutive integers, which under linear probing form one solid run that every
// probe would have to walk to its end (measured: 2.4 s instead of milliseconds at 2^30). A
// Fibonacci multiplicative hash scatters them; the reference avoids the issue with chained buckets
// (bucket_chaining_join :247-252), whose idx = k & (N-1) is still what prjChecksum sums.
__device__ __forceinline__ uint32_t join_hash(uint32_t k)
{
    return (k * 0x9E3779B1u) >> (32 - 15);   // kJoinSlots = 2^15
}
static_assert(kJoinSlots == (1u << 15), "join_hash assumes 2^15 slots");

constexpr int kJoinPre = 16;   // tuples per thread and relation prefetched in registers (16 x 1024 = 16384)

// One persistent workgroup per CU walks the partitions pid = blockIdx.x, += gridDim.x. The LDS table takes
// 128 KiB, so only one workgroup fits a CU and nothing else could hide the HBM latency of a partition's
// 2 x 128 KiB: while partition p is built and probed out of registers, the first 16384 R and S tuples of
// the next partition are already in flight into the other register set (explicit vmcnt wait at the top,
// one unconditional clamped load path -- see hj_build_own.hip for why). Measured at 2^30: 13.4 ms -> see
// profiles/.
// DIRECT (radixBits >= 16): the part of a key that tells the keys of a partition apart, k = key >> radixBits, has at
// most 16 bits, so the 128 KiB of LDS hold one 16-bit counter for EVERY possible k: build = one counter increment,
// probe = one read, no hashing and no probe walks (the hash-table version spends 26 VALU instructions per tuple on
// them, PMC). Counters cannot overflow while |R partition| <= 65535; larger ones take the hash-table path.
// A relation's final partitions as the join sees them. Exact passes: partition pid = part[off[pid] .. off[pid+1]), one
// dense run (cnt == nullptr, log2C = 0). Histogram-free passes: 2^log2C fragments of `cap` slots from pid << log2C,
// fragment f holding cnt[(pid << log2C) + f] keys at its front.
struct PartView {
    const uint32_t* part;
    const uint32_t* off;
    const uint32_t* cnt;
    uint32_t log2C, cap;
    uint32_t total;          // slots of `part` that may be read (loads are clamped to total - 1)
};
struct PartGeom { uint32_t log2C, cap, total; };

// The six arrays are separate __restrict__ parameters, not members of the views: only then does the compiler read the
// offsets and counts (wave-uniform addresses) with scalar loads. As plain struct members they became vector loads
// whose s_waitcnt vmcnt(0) also waited for the register prefetch of the next partition (join 1.6 -> 2.5 ms).
template <bool DIRECT>
__global__ void __launch_bounds__(kJoinThreads)
k_prj_join(const uint32_t* __restrict__ partR, const uint32_t* __restrict__ offR, const uint32_t* __restrict__ cntR, PartGeom gR,
           const uint32_t* __restrict__ partS, const uint32_t* __restrict__ offS, const uint32_t* __restrict__ cntS, PartGeom gS,
           uint32_t radixBits, uint32_t nParts, Counters* __restrict__ ctr, Gate gate)
{
    extern __shared__ uint32_t tab[];  // kJoinSlots
    if (gate_closed(gate)) return;
    const PartView R{partR, offR, cntR, gR.log2C, gR.cap, gR.total}, S{partS, offS, cntS, gS.log2C, gS.cap, gS.total};
    unsigned long long matches = 0, checksum = 0;
    uint32_t overflowParts = 0;
    const bool haveS = S.part != nullptr;

    // slot j of the 16-deep register prefetch covers element ((j & (spf - 1)) * 1024 + thread) of fragment j >> spfShift
    // (one dense run: 16 slots of the one fragment; 4 fragments: 4 slots each). A fragment longer than spf * 1024
    // is finished by a loop over the rest.
    const uint32_t spfShiftR = 4u - R.log2C, spfShiftS = 4u - S.log2C;
    auto base_of = [](const PartView& v, uint32_t pid) { return v.cnt ? (pid << v.log2C) * v.cap : v.off[pid]; };
    auto cnt_of = [](const PartView& v, uint32_t pid, uint32_t f) {
        return v.cnt ? v.cnt[(pid << v.log2C) + f] : v.off[pid + 1] - v.off[pid];
    };
    auto slot_elem = [](uint32_t j, uint32_t spfShift) { return (j & ((1u << spfShift) - 1u)) * kJoinThreads + threadIdx.x; };

    // Register pipeline: S(p) is loaded while R(p) is built, R(p+1) while S(p) is probed; each buffer is
    // refilled only after its last use, so no copy of in-flight registers is ever needed.
    uint32_t bufR[kJoinPre], bufS[kJoinPre];   // partitions hold bare keys (see element formats above)
    auto load_R = [&](uint32_t pid) {   // clamped: every lane always loads a valid address; validity decided at use
        const uint32_t rb0 = base_of(R, pid < nParts ? pid : nParts - 1);
#pragma unroll
        for (int j = 0; j < kJoinPre; ++j) {
            const uint32_t o = rb0 + ((uint32_t)j >> spfShiftR) * R.cap + slot_elem(j, spfShiftR);
            bufR[j] = R.part[o < R.total ? o : R.total - 1];
        }
    };
    auto load_S = [&](uint32_t pid) {
        const uint32_t sb0 = base_of(S, pid);
#pragma unroll
        for (int j = 0; j < kJoinPre; ++j) {
            const uint32_t o = sb0 + ((uint32_t)j >> spfShiftS) * S.cap + slot_elem(j, spfShiftS);
            bufS[j] = S.part[o < S.total ? o : S.total - 1];
        }
    };
    // What a partition holds, read BEFORE any LDS work of the partition: the counts come through scalar loads, and
    // waiting for one (lgkmcnt) also waits for every LDS atomic in flight -- a count looked up between two atomics
    // serialises them (measured: join 1.6 -> 2.5 ms). mask bit j = prefetch slot j holds a key; tail = some fragment
    // is longer than the prefetch covers.
    struct Shape { uint32_t n, mask; bool tail; };
    auto shape_of = [&](const PartView& v, uint32_t pid, uint32_t spfShift) {
        Shape sh{0u, 0u, false};
        const uint32_t spf = 1u << spfShift;                       // prefetch slots per fragment
        for (uint32_t fr = 0; fr < (1u << v.log2C); ++fr) {        // one scalar load per fragment
            const uint32_t n = cnt_of(v, pid, fr);
            sh.n += n;
            sh.tail |= n > (kJoinThreads << spfShift);
            // this thread's slots of the fragment hold elements thread, thread + 1024, ...: the first `mine` are keys
            uint32_t mine = n > threadIdx.x ? (n - threadIdx.x + kJoinThreads - 1) / kJoinThreads : 0u;
            mine = mine < spf ? mine : spf;
            sh.mask |= ((1u << mine) - 1u) << (fr << spfShift);
        }
        return sh;
    };
    // f(key) for every key of partition pid: the prefetched slots out of `buf`, the rest of long fragments from memory
    auto for_each = [&](const PartView& v, uint32_t pid, uint32_t spfShift, const Shape& sh, const uint32_t (&buf)[kJoinPre], auto&& f) {
#pragma unroll
        for (int j = 0; j < kJoinPre; ++j)
            if ((sh.mask >> j) & 1u) f(buf[j]);
        if (sh.tail) {
            const uint32_t b = base_of(v, pid);
            for (uint32_t fr = 0; fr < (1u << v.log2C); ++fr) {
                const uint32_t n = cnt_of(v, pid, fr);
                for (uint32_t i = (kJoinThreads << spfShift) + threadIdx.x; i < n; i += kJoinThreads) f(v.part[b + fr * v.cap + i]);
            }
        }
    };
    load_R(blockIdx.x);

    for (uint32_t pid = blockIdx.x; pid < nParts; pid += gridDim.x) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // R(pid) has landed
        __builtin_amdgcn_sched_barrier(0);
        if (haveS) load_S(pid);                                // in flight while R is built
        __builtin_amdgcn_sched_barrier(0);

        const Shape shR = shape_of(R, pid, spfShiftR);
        const Shape shS = haveS ? shape_of(S, pid, spfShiftS) : Shape{0u, 0u, false};
        const uint32_t nR = shR.n;
        if (nR == 0) {          // serial_radix_partition :531 queues only non-empty R parts (wave-uniform)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_R(pid + gridDim.x);
            continue;
        }
        const uint32_t idxMask = next_pow2_u32(nR) - 1;  // bucket idx mask, :242-245

        if (DIRECT && nR <= 65535u) {
            // ---- direct-addressed counters: tab[k >> 1] holds the counts of k = 2i (low half) and 2i + 1 (high half) ----
            for (uint32_t i = threadIdx.x; i < kJoinSlots; i += kJoinThreads) tab[i] = 0u;
            __syncthreads();
            for_each(R, pid, spfShiftR, shR, bufR, [&](uint32_t key) {
                const uint32_t k = key >> radixBits;
                checksum += k & idxMask;                          // :249,256
                atomicAdd(&tab[k >> 1], 1u << (16u * (k & 1u)));
            });
            __syncthreads();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // S(pid) has landed; bufR is free
            __builtin_amdgcn_sched_barrier(0);
            load_R(pid + gridDim.x);                           // in flight while S is probed
            __builtin_amdgcn_sched_barrier(0);
            if (haveS) {
                uint32_t m32 = 0;                              // <= 65535 per probe, a few dozen probes per thread
#pragma unroll
                for (int j = 0; j < kJoinPre; ++j)
                    if ((shS.mask >> j) & 1u) {
                        const uint32_t k = bufS[j] >> radixBits;
                        m32 += (tab[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;      // :268-271, all equal keys at once
                    }
                matches += m32;
                if (shS.tail) {
                    const uint32_t sb = base_of(S, pid);
                    for (uint32_t fr = 0; fr < (1u << S.log2C); ++fr) {
                        const uint32_t n = cnt_of(S, pid, fr);
                        for (uint32_t i = (kJoinThreads << spfShiftS) + threadIdx.x; i < n; i += kJoinThreads) {
                            const uint32_t k = S.part[sb + fr * S.cap + i] >> radixBits;
                            matches += (tab[k >> 1] >> (16u * (k & 1u))) & 0xFFFFu;
                        }
                    }
                }
            }
            __syncthreads();
        } else if (nR <= kJoinBlockTuples) {
            // ---- the common case: the whole R partition fits one LDS table ----
            for (uint32_t i = threadIdx.x; i < kJoinSlots; i += kJoinThreads) tab[i] = kEmpty32;
            __syncthreads();
            for_each(R, pid, spfShiftR, shR, bufR, [&](uint32_t key) {
                const uint32_t k = key >> radixBits;           // distinguishes keys inside a partition
                checksum += k & idxMask;                          // :249,256
                uint32_t h = join_hash(k);
                while (atomicCAS(&tab[h], kEmpty32, k) != kEmpty32) h = (h + 1) & (kJoinSlots - 1);
            });
            __syncthreads();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // S(pid) has landed; bufR is free
            __builtin_amdgcn_sched_barrier(0);
            load_R(pid + gridDim.x);                           // in flight while S is probed
            __builtin_amdgcn_sched_barrier(0);
            if (haveS)
                for_each(S, pid, spfShiftS, shS, bufS, [&](uint32_t key) {
                    const uint32_t k = key >> radixBits;
                    uint32_t h = join_hash(k);
                    for (;;) {
                        const uint32_t v = tab[h];
                        if (v == kEmpty32) break;
                        matches += (v == k);                          // :268-271
                        h = (h + 1) & (kJoinSlots - 1);
                    }
                });
            __syncthreads();
        } else {
            // ---- oversized R partition (skew): several LDS builds, S probed against each. Exact layout only: the
            // histogram-free passes are not planned for partitions that could outgrow one table (prj_plan)
            overflowParts += 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            load_R(pid + gridDim.x);
            const uint32_t rb = R.off[pid], re = R.off[pid + 1];
            const uint32_t sb = haveS ? S.off[pid] : 0, se = haveS ? S.off[pid + 1] : 0;
            for (uint32_t blk = rb; blk < re; blk += kJoinBlockTuples) {
                const uint32_t bend = (re - blk > kJoinBlockTuples) ? blk + kJoinBlockTuples : re;
                for (uint32_t i = threadIdx.x; i < kJoinSlots; i += kJoinThreads) tab[i] = kEmpty32;
                __syncthreads();
                for (uint32_t i = blk + threadIdx.x; i < bend; i += kJoinThreads) {
                    const uint32_t k = R.part[i] >> radixBits;
                    checksum += k & idxMask;
                    uint32_t h = join_hash(k);
                    while (atomicCAS(&tab[h], kEmpty32, k) != kEmpty32) h = (h + 1) & (kJoinSlots - 1);
                }
                __syncthreads();
                for (uint32_t i = sb + threadIdx.x; i < se; i += kJoinThreads) {
                    const uint32_t k = S.part[i] >> radixBits;
                    uint32_t h = join_hash(k);
                    for (;;) {
                        const uint32_t v = tab[h];
                        if (v == kEmpty32) break;
                        matches += (v == k);
                        h = (h + 1) & (kJoinSlots - 1);
                    }
                }
                __syncthreads();
            }
        }
    }
    // one atomic per wavefront
    for (int off = 32; off > 0; off >>= 1) {
        matches += __shfl_down(matches, off, 64);
        checksum += __shfl_down(checksum, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (matches) atomicAdd(&counter_shard(ctr)->prjMatches, matches);
        if (checksum) atomicAdd(&counter_shard(ctr)->prjChecksum, checksum);
    }
    if (threadIdx.x == 0 && overflowParts) atomicAdd(&ctr->prjOverflowParts, (unsigned long long)overflowParts);
}

// ---------------------------------------------------------------------------
// host-side planning and launch
// ---------------------------------------------------------------------------
static uint32_t pick_chunk_len(uint64_t n)
{
    // aim for ~4096 chunks (>= 2 per CU-slot), whole tiles, 4Ki..64Ki tuples
    uint64_t len = n / 4096;
    len = (len + kTile - 1) / kTile * kTile;
    if (len < (uint64_t)kTile) len = kTile;
    if (len > 65536) len = 65536;
    return (uint32_t)len;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct PassLayout { uint32_t chunkLen; uint64_t maxChunks; uint64_t histEntries; uint64_t scanBlocks; };

static PassLayout pass_layout(uint64_t n, uint32_t nSeg, uint32_t fan)
{
    PassLayout l;
    l.chunkLen = pick_chunk_len(n);
    l.maxChunks = n / l.chunkLen + nSeg + 1;
    l.histEntries = l.maxChunks * fan;
    l.scanBlocks = (l.histEntries + kScanTile - 1) / kScanTile;
    return l;
}

// Slots of a fragment that is to hold a Poisson-distributed number of keys of the given mean: mean + 7 sigma (one
// fragment in 10^12 is larger; a run has ~10^6 of them) + a little for tiny means, in whole 16-byte vectors.
static uint32_t frag_cap(double mean)
{
    // whole 128-byte lines: a tile leaves runs of 128 bytes per bin, and with fragments that start inside a line every one
    // of them straddles two (measured at 2^30: 14.8 ms of partitioning with 4580-slot fragments, 12.5 ms with 4576)
    const double c = mean + 7.0 * std::sqrt(mean) + 32.0;
    return ((uint32_t)c + 31u) & ~31u;
}

constexpr uint64_t kFragMinTuples = 1ull << 25;      // below this the exact passes take well under a millisecond
constexpr double kFragMinMean = 256.0;               // smaller fragments are mostly slack

// Fragment geometry of one relation, or C1 = 0 when it has to take the exact passes.
static PrjFrag frag_geometry(uint64_t n, uint32_t radixBits, uint32_t bits1, uint32_t bits2, uint32_t mode)
{
    PrjFrag g{};
    if (n == 0 || bits2 == 0 || mode == 1 || n > (1ull << 31)) return g;
    if (mode == 0 && n < kFragMinTuples) return g;
    const uint32_t F1 = 1u << bits1, F2 = 1u << bits2;
    // pass 1: as many chunks as two rounds of workgroups (2 per CU), fewer while the fragments would get short
    uint32_t C1 = kMaxInFrags;      // 512 .. 4096 measured at 2^30: 11.8 - 11.9 ms throughout
    while (C1 > 256 && (double)n / C1 / F1 < 2048.0) C1 >>= 1;
    while (C1 > 1 && (double)n / C1 / F1 < kFragMinMean) C1 >>= 1;
    const uint64_t chunkLen1 = ((n + C1 - 1) / C1 + kTile - 1) / kTile * kTile;
    const double mean1 = (double)chunkLen1 / F1;
    if (mean1 < kFragMinMean) return g;
    const uint32_t cap1 = frag_cap(mean1);
    // pass 2: ~1024 chunks in all, each a whole number of pass-1 fragments; the join wants <= 16 fragments per partition
    uint32_t C2 = 1024 / F1 ? 1024 / F1 : 1;
    if (C2 > 16) C2 = 16;
    if (C2 > C1) C2 = C1;
    auto mean2_of = [&](uint32_t c2) { return (double)(C1 / c2) * mean1 / F2; };
    while (C2 > 1 && mean2_of(C2) < kFragMinMean) C2 >>= 1;
    const double mean2 = mean2_of(C2);
    if (mean2 < kFragMinMean) return g;
    const uint32_t cap2 = frag_cap(mean2);
    // the buffers hold 8 bytes per tuple (hj_reserve), the positions are 32-bit, and a partition must fit one LDS table
    const uint64_t slots1 = (uint64_t)F1 * C1 * cap1, slots2 = (uint64_t)F1 * F2 * C2 * cap2;
    if (slots1 > 2 * n || slots2 > 2 * n || slots1 >= (1ull << 32) || slots2 >= (1ull << 32)) return g;
    if ((uint64_t)C2 * cap2 > (radixBits >= 16 ? 65535u : kJoinBlockTuples)) return g;
    g.C1 = C1; g.cap1 = cap1; g.chunkLen1 = (uint32_t)chunkLen1;
    g.C2 = C2; g.cap2 = cap2;
    while ((1u << g.log2C2) < C2) ++g.log2C2;
    return g;
}

PrjPlan prj_plan(uint64_t nR, uint64_t nS, uint32_t radixBits, uint32_t mode)
{
    PrjPlan pl{};
    pl.radixBits = radixBits;
    if (radixBits <= 8) { pl.bits1 = radixBits; pl.bits2 = 0; }
    else { pl.bits1 = radixBits / 2; pl.bits2 = radixBits - pl.bits1; }  // prj_thread :814-816
    const uint32_t F1 = 1u << pl.bits1, F2 = 1u << pl.bits2;
    pl.fragR = frag_geometry(nR, radixBits, pl.bits1, pl.bits2, mode);
    pl.fragS = frag_geometry(nS, radixBits, pl.bits1, pl.bits2, mode);
    pl.optimistic = pl.fragR.C1 != 0 && (nS == 0 || pl.fragS.C1 != 0);
    if (pl.optimistic) {
        pl.cnt1Entries = (uint64_t)F1 * (pl.fragR.C1 > pl.fragS.C1 ? pl.fragR.C1 : pl.fragS.C1);
        pl.cnt2EntriesR = (uint64_t)F1 * F2 * pl.fragR.C2;
        pl.cnt2EntriesS = (uint64_t)F1 * F2 * pl.fragS.C2;
    }
    // run_pass lays every relation out with pick_chunk_len of ITS size, and the chunk count is not monotone in
    // the size (chunkLen doubles at the 8192 -> 16384 step: |R| = 40e6 has fewer chunks than |S| = 2^25), so
    // each region takes the larger of the two relations' needs, pass by pass
    for (const uint64_t n : {nR, nS}) {
        if (n == 0) continue;
        const PassLayout l1 = pass_layout(n, 1, F1);
        const PassLayout l2 = pass_layout(n, F1, F2);
        if (l1.maxChunks > pl.maxChunks1) pl.maxChunks1 = l1.maxChunks;
        if (l2.maxChunks > pl.maxChunks2) pl.maxChunks2 = l2.maxChunks;
        for (const PassLayout& l : {l1, l2}) {
            if (l.histEntries > pl.histEntries) pl.histEntries = l.histEntries;
            if (l.scanBlocks > pl.scanBlocks) pl.scanBlocks = l.scanBlocks;
        }
    }
    const uint64_t P = (uint64_t)F1 * F2;
    size_t bytes = 0;
    bytes += align_up(sizeof(uint32_t) * 2, 256);               // seg0
    bytes += align_up(sizeof(uint32_t) * (F1 + 1), 256);        // seg1 (after pass 1)
    bytes += align_up(sizeof(uint32_t) * (F1 + 2), 256);        // chunkBase
    bytes += align_up(sizeof(uint32_t) * pl.histEntries, 256);  // hist / scanned
    bytes += align_up(sizeof(uint32_t) * (pl.scanBlocks + 1), 256);   // block sums
    bytes += 2 * align_up(sizeof(uint32_t) * (P + 1), 256);     // final offsets R, S
    bytes += align_up(sizeof(uint32_t) * pl.cnt1Entries, 256);  // fragment counters: pass 1 (R, then S), pass 2 of R, of S
    bytes += align_up(sizeof(uint32_t) * pl.cnt2EntriesR, 256) + align_up(sizeof(uint32_t) * pl.cnt2EntriesS, 256);
    pl.workspaceBytes = bytes;
    return pl;
}

uint64_t prj_hist_entries_needed(uint64_t n, uint32_t radixBits)
{
    const PrjPlan shape = prj_plan(0, 0, radixBits);            // bit split only
    const uint32_t F1 = 1u << shape.bits1, F2 = 1u << shape.bits2;
    if (n == 0) return 0;
    const uint64_t a = pass_layout(n, 1, F1).histEntries, b = pass_layout(n, F1, F2).histEntries;
    return a > b ? a : b;
}

namespace {
struct Work {
    uint32_t *seg0, *seg1, *chunkBase, *hist, *sums, *offR, *offS, *cnt1, *cnt2R, *cnt2S;
};
Work carve(const PrjPlan& pl, void* base)
{
    const uint32_t F1 = 1u << pl.bits1, F2 = 1u << pl.bits2;
    const uint64_t P = (uint64_t)F1 * F2;
    char* p = static_cast<char*>(base);
    Work w;
    w.seg0 = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * 2, 256);
    w.seg1 = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * (F1 + 1), 256);
    w.chunkBase = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * (F1 + 2), 256);
    w.hist = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * pl.histEntries, 256);
    w.sums = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * (pl.scanBlocks + 1), 256);
    w.offR = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * (P + 1), 256);
    w.offS = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * (P + 1), 256);
    w.cnt1 = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * pl.cnt1Entries, 256);
    w.cnt2R = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * pl.cnt2EntriesR, 256);
    w.cnt2S = reinterpret_cast<uint32_t*>(p);
    return w;
}

// One radix pass: in -> out, segments segIn[nSeg+1] -> segOut[nSeg*fan+1].
// in32: the input already holds bare keys (pass 2); the output always does.
hipError_t run_pass(const void* in, bool in32, uint32_t* out, uint64_t n, const uint32_t* segIn, uint32_t nSeg,
                    uint32_t shift, uint32_t bits, uint32_t* segOut, const Work& w, Gate gate, hipStream_t s,
                    hipEvent_t evScatter0 = nullptr, hipEvent_t evScatter1 = nullptr)
{
    const uint32_t fan = 1u << bits;
    const PassLayout l = pass_layout(n, nSeg, fan);
    hipLaunchKernelGGL(k_chunk_base, dim3(1), dim3(64), 0, s, segIn, nSeg, l.chunkLen, w.chunkBase);
    PassParams p{segIn, nSeg, l.chunkLen, w.chunkBase, shift, fan, 0u, 0u};
    // entries past the live chunks must be zero for the scan to be a prefix of live data only
    const hipError_t e = hipMemsetAsync(w.hist, 0, sizeof(uint32_t) * l.histEntries, s);
    if (e != hipSuccess) return e;
    if (in32) hipLaunchKernelGGL(k_radix_hist<true>, dim3((unsigned)l.maxChunks), dim3(kBlock), 0, s, in, p, w.hist, gate);
    else hipLaunchKernelGGL(k_radix_hist<false>, dim3((unsigned)l.maxChunks), dim3(kBlock), 0, s, in, p, w.hist, gate);
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)l.scanBlocks), dim3(kBlock), 0, s, w.hist, l.histEntries, w.sums, gate);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, s, w.sums, (uint32_t)l.scanBlocks, gate);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)l.scanBlocks), dim3(kBlock), 0, s, w.hist, l.histEntries, w.sums, gate);
    const uint32_t nOut = nSeg * fan + 1;
    hipLaunchKernelGGL(k_seg_offsets, dim3((nOut + kBlock - 1) / kBlock), dim3(kBlock), 0, s, p, w.hist, (uint32_t)n, segOut);
    // instance choice measured at 2^30 (tools/prj_variants.sh, profiles/r01_prj_variants.txt): 512 threads x 16
    // elements, no register prefetch, <= 128 VGPRs (2 workgroups per CU); more, smaller workgroups and the
    // prefetching instance were slower or equal
    if (evScatter0) (void)hipEventRecord(evScatter0, s);
    if (in32) hipLaunchKernelGGL((k_radix_scatter<true, true, 512, 4, false, 4>), dim3((unsigned)l.maxChunks), dim3(512), 0, s,
                                 in, static_cast<void*>(out), p, w.hist, gate);
    else hipLaunchKernelGGL((k_radix_scatter<false, true, 512, 8, false, 4>), dim3((unsigned)l.maxChunks), dim3(512), 0, s,
                            in, static_cast<void*>(out), p, w.hist, gate);
    if (evScatter1) (void)hipEventRecord(evScatter1, s);
    return hipGetLastError();
}

hipError_t partition_relation(const PrjPlan& pl, const Work& w, const uint64_t* in, uint64_t n,
                              uint32_t* tmp, uint32_t* out, uint32_t* finalOff, Gate gate, hipStream_t s,
                              hipEvent_t evS0 = nullptr, hipEvent_t evS1 = nullptr)
{
    hipLaunchKernelGGL(k_init_seg, dim3(1), dim3(64), 0, s, w.seg0, (uint32_t)n);
    if (pl.bits2 == 0) return run_pass(in, false, out, n, w.seg0, 1, 0, pl.bits1, finalOff, w, gate, s, evS0, evS1);
    const hipError_t e = run_pass(in, false, tmp, n, w.seg0, 1, 0, pl.bits1, w.seg1, w, gate, s, evS0, evS1);   // pass 1, R = 0: tuples -> keys
    if (e != hipSuccess) return e;
    return run_pass(tmp, true, out, n, w.seg1, 1u << pl.bits1, pl.bits1, pl.bits2, finalOff, w, gate, s);    // pass 2, R = bits1
}

// The two histogram-free passes over one relation: tuples -> fragments of tmp -> fragments of out, counts in cnt2.
hipError_t partition_relation_frag(const PrjPlan& pl, const PrjFrag& g, const Work& w, const uint64_t* in, uint64_t n,
                                   uint32_t* tmp, uint32_t* out, uint32_t* cnt2, Counters* ctr, hipStream_t s,
                                   hipEvent_t evS0 = nullptr, hipEvent_t evS1 = nullptr)
{
    const uint32_t F1 = 1u << pl.bits1, F2 = 1u << pl.bits2;
    const FragPass p1{g.C1, 0u, F1, g.cap1, w.cnt1, (uint32_t)n, g.chunkLen1, 0u, 0u, nullptr};
    if (evS0) (void)hipEventRecord(evS0, s);
    hipLaunchKernelGGL((k_radix_scatter_frag<false, HJ_FRAG1_NT, HJ_FRAG1_E, HJ_FRAG1_WPE, HJ_FRAG1_LANECOL != 0>), dim3(g.C1), dim3(HJ_FRAG1_NT), 0, s, static_cast<const void*>(in), tmp, p1, ctr);
    if (evS1) (void)hipEventRecord(evS1, s);
    const FragPass p2{g.C2, pl.bits1, F2, g.cap2, cnt2, 0u, 0u, g.C1 / g.C2, g.cap1, w.cnt1};
    hipLaunchKernelGGL((k_radix_scatter_frag<true, HJ_FRAG2_NT, HJ_FRAG2_E, HJ_FRAG2_WPE>), dim3(F1 * g.C2), dim3(HJ_FRAG2_NT), 0, s, static_cast<const void*>(tmp), out, p2, ctr);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_prj(const PrjPlan& pl, const PrjBuffers& buf, const uint64_t* R, uint64_t nR,
                      const uint64_t* S, uint64_t nS, int nCU, Counters* ctr, hipEvent_t evPartDone, hipEvent_t evScatter0,
                      hipEvent_t evScatter1, hipStream_t s)
{
    const Work w = carve(pl, buf.work);
    uint32_t* const tmp = reinterpret_cast<uint32_t*>(buf.tmpA);
    uint32_t* const partR = reinterpret_cast<uint32_t*>(buf.partR);
    uint32_t* const partS = reinterpret_cast<uint32_t*>(buf.partS);
    hipError_t e;
    // Histogram-free passes first (Counters::prjFallback is 0: the caller zeroed the counters); the exact passes are
    // enqueued behind them and return at once unless a fragment overflowed.
    Gate exact = kNoGate;
    if (pl.optimistic) {
        if ((e = partition_relation_frag(pl, pl.fragR, w, R, nR, tmp, partR, w.cnt2R, ctr, s, evScatter0, evScatter1)) != hipSuccess) return e;
        if (S && (e = partition_relation_frag(pl, pl.fragS, w, S, nS, tmp, partS, w.cnt2S, ctr, s)) != hipSuccess) return e;
        exact = Gate{&ctr->prjFallback, 1ull};
        evScatter0 = evScatter1 = nullptr;
    }
    if ((e = partition_relation(pl, w, R, nR, tmp, partR, w.offR, exact, s, evScatter0, evScatter1)) != hipSuccess) return e;
    if (S && (e = partition_relation(pl, w, S, nS, tmp, partS, w.offS, exact, s)) != hipSuccess) return e;
    if (evPartDone && (e = hipEventRecord(evPartDone, s)) != hipSuccess) return e;
    const uint32_t P = 1u << pl.radixBits;
#ifndef HJ_JOIN_ROUNDS
#define HJ_JOIN_ROUNDS 1
#endif
    const unsigned want = (unsigned)nCU * HJ_JOIN_ROUNDS;          // one persistent workgroup per CU (x rounds: development flag)
    const unsigned grid = P < want ? P : want;
    static_assert(kJoinSlots * 2 == 65536, "two 16-bit counters per LDS word cover every 16-bit key remainder");
    const PartView none{nullptr, nullptr, nullptr, 0u, 0u, 1u};
    auto join = [&](const PartView& vr, const PartView& vs, Gate gate) {
        if (pl.radixBits >= 16)
            hipLaunchKernelGGL(k_prj_join<true>, dim3(grid), dim3(kJoinThreads), kJoinSlots * sizeof(uint32_t), s,
                               vr.part, vr.off, vr.cnt, PartGeom{vr.log2C, vr.cap, vr.total},
                               vs.part, vs.off, vs.cnt, PartGeom{vs.log2C, vs.cap, vs.total}, pl.radixBits, P, ctr, gate);
        else
            hipLaunchKernelGGL(k_prj_join<false>, dim3(grid), dim3(kJoinThreads), kJoinSlots * sizeof(uint32_t), s,
                               vr.part, vr.off, vr.cnt, PartGeom{vr.log2C, vr.cap, vr.total},
                               vs.part, vs.off, vs.cnt, PartGeom{vs.log2C, vs.cap, vs.total}, pl.radixBits, P, ctr, gate);
    };
    if (pl.optimistic) {
        const PartView vr{partR, nullptr, w.cnt2R, pl.fragR.log2C2, pl.fragR.cap2, P * pl.fragR.C2 * pl.fragR.cap2};
        const PartView vs{partS, nullptr, w.cnt2S, pl.fragS.log2C2, pl.fragS.cap2, P * pl.fragS.C2 * pl.fragS.cap2};
        join(vr, S ? vs : none, Gate{&ctr->prjFallback, 0ull});
    }
    const PartView er{partR, w.offR, nullptr, 0u, 0u, (uint32_t)nR};
    const PartView es{partS, w.offS, nullptr, 0u, 0u, (uint32_t)nS};
    join(er, S ? es : none, exact);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// multi-GPU destination split (hj_shard_histogram_dev / hj_shard_scatter_dev): one radix pass on
// the low log2(nShards) key bits, tuples in, bare keys out, and STABLE: inside a destination the keys
// keep their input order exactly. The join's semantics are "insert in input order"; the receiving rank
// takes a key's position in its receive buffer as that order, so the scatter may not permute anything
// (the PRJ scatter above ranks with LDS atomics and is only stable tile by tile -- the radix join does
// not care). It also keeps a near-sorted input near-sorted, so the receiver's locality pre-round still
// picks the LDS-window build.
//
// Ranking without atomics: lane l of wavefront w holds, for k = 0..7, the tile's element k*1024 + 64 w + l,
// so (k, w, l) is input order. For one k a wavefront finds, per lane, the lanes with the same
// destination (log2(fan) ballots) -- its rank among them is a popcount of the lower lanes, their
// number goes to cnt[bin][k][w]. An exclusive scan of cnt in exactly that order gives every
// (bin, k, w) group its place in the tile's staged order.
// ---------------------------------------------------------------------------
constexpr int kStabThreads = 1024;
constexpr int kStabPer = 8;                                    // tuples per thread per tile
constexpr int kStabTile = kStabThreads * kStabPer;             // 8192
constexpr int kStabGroups = kStabPer * (kStabThreads / 64);    // (k, wavefront) groups per tile = 128
constexpr int kStabMaxFan = 64;

__global__ void __launch_bounds__(kStabThreads, 8)
k_shard_scatter_stable(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, PassParams p,
                       const uint32_t* __restrict__ scanned)
{
    extern __shared__ uint32_t stab[];              // cnt[fan][kStabGroups], then stage[kStabTile + kStabTile/32 + 1]
    uint32_t* const cnt = stab;
    const uint32_t nCnt = p.fan * kStabGroups;
    uint32_t* const stage = stab + nCnt;
    __shared__ unsigned int delta[kStabMaxFan];     // write cursor of the bin - its offset in the staged tile
    __shared__ unsigned int cursor[kStabMaxFan];
    __shared__ uint32_t wsum[kStabThreads / 64];
    constexpr uint32_t kDump = kStabTile + (kStabTile >> 5);

    const uint32_t c = blockIdx.x;
    if (c >= p.chunkBase[p.nSeg]) return;
    const ChunkRange r = chunk_range(p, c);
    const uint32_t fmask = p.fan - 1;
    const uint32_t len = r.end - r.begin;
    const uint32_t wave = threadIdx.x >> 6;
    int bits = 0;
    while ((1u << bits) < p.fan) ++bits;
    if (threadIdx.x < p.fan) cursor[threadIdx.x] = scanned[hist_index(p, r, threadIdx.x)];
    const uint32_t perThread = nCnt / kStabThreads;               // fan * 128 / 1024 = fan / 8 (fan >= 8) ...
    const uint32_t items = perThread ? perThread : 1;             // ... or 1 with some threads idle (fan < 8)
    __syncthreads();

    for (uint32_t tb = 0; tb < len; tb += kStabTile) {
        for (uint32_t i = threadIdx.x; i < nCnt; i += kStabThreads) cnt[i] = 0;
        uint32_t key[kStabPer], rk4[kStabPer / 4];    // ranks are < 64: four to a register
        uint32_t okMask = 0;
#pragma unroll
        for (int k = 0; k < kStabPer / 4; ++k) rk4[k] = 0;
#pragma unroll
        for (int k = 0; k < kStabPer; ++k) {
            const uint32_t rel = tb + (uint32_t)k * kStabThreads + threadIdx.x;
            const bool ok = rel < len;
            const uint64_t t = ok ? in[(uint64_t)r.begin + rel] : 0ull;
            key[k] = (t >> 32) ? 0u : (uint32_t)t;                 // payload bits set: see PassParams::zeroBad
            okMask |= ok ? (1u << k) : 0u;
        }
        __syncthreads();                                          // cnt is zero
#pragma unroll
        for (int k = 0; k < kStabPer; ++k) {
            const bool ok = (okMask >> k) & 1u;
            const uint32_t bin = ((key[k] - p.bias) >> p.shift) & fmask;
            unsigned long long peers = __ballot(ok);
            for (int b = 0; b < bits; ++b) {
                const bool bit = (bin >> b) & 1u;
                const unsigned long long m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            const uint32_t rk = lane_rank(peers);
            rk4[k / 4] |= rk << (8 * (k % 4));
            if (ok && rk == 0) cnt[(bin * kStabPer + k) * (kStabThreads / 64) + wave] = (uint32_t)__popcll(peers);
        }
        __syncthreads();
        {   // exclusive scan of cnt[0..nCnt) in place, `items` consecutive entries per thread
            const uint32_t base = threadIdx.x * items;
            uint32_t local = 0;
            for (uint32_t i = 0; i < items; ++i) local += (base + i < nCnt) ? cnt[base + i] : 0u;
            uint32_t total;
            uint32_t ex = block_exclusive_scan<kStabThreads>(local, wsum, total);
            for (uint32_t i = 0; i < items; ++i) {
                if (base + i < nCnt) { const uint32_t v = cnt[base + i]; cnt[base + i] = ex; ex += v; }
            }
        }
        __syncthreads();
        if (threadIdx.x < p.fan) {
            const uint32_t off = cnt[threadIdx.x * kStabGroups];                      // first group of the bin
            const uint32_t end = threadIdx.x + 1 < p.fan ? cnt[(threadIdx.x + 1) * kStabGroups]
                                                         : (len - tb < (uint32_t)kStabTile ? len - tb : (uint32_t)kStabTile);
            const uint32_t cu = cursor[threadIdx.x];
            delta[threadIdx.x] = cu - off;
            cursor[threadIdx.x] = cu + (end - off);
        }
#pragma unroll
        for (int k = 0; k < kStabPer; ++k) {
            const uint32_t bin = ((key[k] - p.bias) >> p.shift) & fmask;
            const uint32_t at = cnt[(bin * kStabPer + k) * (kStabThreads / 64) + wave] + ((rk4[k / 4] >> (8 * (k % 4))) & 0xFFu);
            stage[((okMask >> k) & 1u) ? at + (at >> 5) : kDump] = key[k];
        }
        __syncthreads();
        const uint32_t valid = len - tb < (uint32_t)kStabTile ? len - tb : (uint32_t)kStabTile;
#pragma unroll 4
        for (int k = 0; k < kStabPer; ++k) {
            const uint32_t q = (uint32_t)k * kStabThreads + threadIdx.x;
            if (q < valid) {
                const uint32_t t = stage[q + (q >> 5)];
                out[delta[((t - p.bias) >> p.shift) & fmask] + q] = t;
            }
        }
        __syncthreads();                                          // stage, cnt and delta are reused by the next tile
    }
}


// The same split for fan-outs <= 16 (every multi-GPU node there is), one WAVEFRONT per chunk and no workgroup barrier:
// the kernel above synchronises its 16 wavefronts seven times per tile (3.0 TB/s of its 12 bytes per tuple). Here a
// wavefront walks its chunk in tiles of 512 tuples (lane l holds tile elements 64 k + l, k = 0..7: (k, l) is input
// order), ranks them bin by bin with ballots -- fan x 8 ballots per tile, the running count is the tile offset of the
// next bin -- stages the tile in its own 2 KiB of LDS grouped by destination and writes the runs out through cursors it
// keeps in registers (start = the chunk's scanned histogram entries). Same output, element for element.
constexpr int kSwThreads = 256;
constexpr int kSwPer = 8;
constexpr int kSwTile = 64 * kSwPer;             // 512 tuples per wavefront tile
constexpr int kSwMaxFan = 16;

__global__ void __launch_bounds__(kSwThreads)
k_shard_scatter_wave(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, PassParams p,
                     const uint32_t* __restrict__ scanned)
{
    __shared__ uint32_t stage[kSwThreads / 64][kSwTile];
    __shared__ uint32_t delta[kSwThreads / 64][kSwMaxFan];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * (kSwThreads / 64) + wave;
    if (c >= p.chunkBase[p.nSeg]) return;                          // whole wavefronts leave; nobody waits for them
    const ChunkRange r = chunk_range(p, c);
    const uint32_t fan = p.fan, fmask = fan - 1;
    const uint32_t len = r.end - r.begin;
    uint32_t cur[kSwMaxFan];                                       // write cursor per destination (wave-uniform)
#pragma unroll
    for (int b = 0; b < kSwMaxFan; ++b)
        cur[b] = (uint32_t)b < fan ? (uint32_t)__builtin_amdgcn_readfirstlane((int)scanned[hist_index(p, r, (uint32_t)b)]) : 0u;
    const uint64_t* __restrict__ src = in + r.begin;
    uint64_t nxt[kSwPer];
#pragma unroll
    for (int k = 0; k < kSwPer; ++k) { const uint32_t rel = 64u * k + lane; nxt[k] = rel < len ? src[rel] : 0ull; }
    for (uint32_t tb = 0; tb < len; tb += kSwTile) {
        uint32_t key[kSwPer], bin[kSwPer], pos[kSwPer];
        uint32_t okMask = 0;
#pragma unroll
        for (int k = 0; k < kSwPer; ++k) {
            const uint64_t t = nxt[k];
            key[k] = (t >> 32) ? 0u : (uint32_t)t;                 // payload bits set: travels as key 0 (PassParams::zeroBad)
            const bool ok = tb + 64u * k + lane < len;
            bin[k] = ok ? ((key[k] - p.bias) >> p.shift) & fmask : 0xFFFFFFFFu;   // past the chunk's end: no destination's
            okMask |= ok ? (1u << k) : 0u;
            pos[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < kSwPer; ++k) { const uint32_t rel = tb + kSwTile + 64u * k + lane; nxt[k] = rel < len ? src[rel] : 0ull; }
        uint32_t off = 0;                                          // staged position where the next destination's run starts
#pragma unroll
        for (int b = 0; b < kSwMaxFan; ++b) {
            if ((uint32_t)b < fan) {                               // wave-uniform
                uint32_t cnt = 0;
#pragma unroll
                for (int k = 0; k < kSwPer; ++k) {
                    const bool mine = bin[k] == (uint32_t)b;
                    const unsigned long long m = __ballot(mine);
                    pos[k] = mine ? off + cnt + lane_rank(m) : pos[k];
                    cnt += (uint32_t)__popcll(m);
                }
                if (lane == 0) delta[wave][b] = cur[b] - off;      // output index of staged position q = delta[bin] + q
                cur[b] += cnt;
                off += cnt;
            }
        }
#pragma unroll
        for (int k = 0; k < kSwPer; ++k)
            if ((okMask >> k) & 1u) stage[wave][pos[k]] = key[k];
        // (LDS operations of one wavefront complete in issue order: the reads below see the stores above)
        const uint32_t valid = len - tb < (uint32_t)kSwTile ? len - tb : (uint32_t)kSwTile;
#pragma unroll
        for (int k = 0; k < kSwPer; ++k) {
            const uint32_t q = 64u * k + lane;
            if (q < valid) {
                const uint32_t t = stage[wave][q];
                out[delta[wave][((t - p.bias) >> p.shift) & fmask] + q] = t;
            }
        }
    }
}

namespace {
struct ShardWork { uint32_t *seg0, *segOut, *chunkBase, *hist, *sums; };
ShardWork shard_carve(void* base, uint64_t n, uint32_t fan)
{
    const PassLayout l = pass_layout(n, 1, fan);
    char* p = static_cast<char*>(base);
    ShardWork w;
    w.seg0 = reinterpret_cast<uint32_t*>(p); p += 256;
    w.segOut = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * (fan + 1), 256);
    w.chunkBase = reinterpret_cast<uint32_t*>(p); p += 256;
    w.hist = reinterpret_cast<uint32_t*>(p); p += align_up(sizeof(uint32_t) * l.histEntries, 256);
    w.sums = reinterpret_cast<uint32_t*>(p);
    return w;
}
__global__ void k_shard_counts(const uint32_t* __restrict__ segOut, uint32_t fan, unsigned long long* __restrict__ counts)
{
    if (threadIdx.x < fan) counts[threadIdx.x] = segOut[threadIdx.x + 1] - segOut[threadIdx.x];
}
}  // namespace

size_t shard_work_bytes(uint64_t n, uint32_t nShards)
{
    const PassLayout l = pass_layout(n, 1, nShards);
    return 256 + align_up(sizeof(uint32_t) * (nShards + 1), 256) + 256 +
           align_up(sizeof(uint32_t) * l.histEntries, 256) + align_up(sizeof(uint32_t) * (l.scanBlocks + 1), 256);
}

hipError_t launch_shard_hist(const uint64_t* in, uint64_t n, uint32_t nShards, uint32_t digitShift, void* work,
                             unsigned long long* counts, hipStream_t s)
{
    const ShardWork w = shard_carve(work, n, nShards);
    const PassLayout l = pass_layout(n, 1, nShards);
    hipLaunchKernelGGL(k_init_seg, dim3(1), dim3(64), 0, s, w.seg0, (uint32_t)n);
    hipLaunchKernelGGL(k_chunk_base, dim3(1), dim3(64), 0, s, w.seg0, 1u, l.chunkLen, w.chunkBase);
    PassParams p{w.seg0, 1u, l.chunkLen, w.chunkBase, digitShift & 0xFFu, nShards, (digitShift >> 8) & 1u, 1u};
    const hipError_t e = hipMemsetAsync(w.hist, 0, sizeof(uint32_t) * l.histEntries, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_radix_hist<false>, dim3((unsigned)l.maxChunks), dim3(kBlock), 0, s, static_cast<const void*>(in), p, w.hist, kNoGate);
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)l.scanBlocks), dim3(kBlock), 0, s, w.hist, l.histEntries, w.sums, kNoGate);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, s, w.sums, (uint32_t)l.scanBlocks, kNoGate);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)l.scanBlocks), dim3(kBlock), 0, s, w.hist, l.histEntries, w.sums, kNoGate);
    hipLaunchKernelGGL(k_seg_offsets, dim3(1), dim3(kBlock), 0, s, p, w.hist, (uint32_t)n, w.segOut);
    hipLaunchKernelGGL(k_shard_counts, dim3(1), dim3(64), 0, s, w.segOut, nShards, counts);
    return hipGetLastError();
}

// tuples in, bare keys out, stable: the exchange moves 4 bytes per tuple and no index
hipError_t launch_shard_scatter_ordered(const uint64_t* in, uint64_t n, uint32_t nShards, uint32_t digitShift, void* work,
                                        uint32_t* outKeys, hipStream_t s)
{
    const ShardWork w = shard_carve(work, n, nShards);
    const PassLayout l = pass_layout(n, 1, nShards);
    PassParams p{w.seg0, 1u, l.chunkLen, w.chunkBase, digitShift & 0xFFu, nShards, (digitShift >> 8) & 1u, 1u};
    const size_t lds = sizeof(uint32_t) * ((size_t)nShards * kStabGroups + kStabTile + (kStabTile >> 5) + 1);   // <= 65.1 KiB
    if (nShards <= (uint32_t)kSwMaxFan)
        hipLaunchKernelGGL(k_shard_scatter_wave, dim3((unsigned)((l.maxChunks + kSwThreads / 64 - 1) / (kSwThreads / 64))),
                           dim3(kSwThreads), 0, s, in, outKeys, p, w.hist);
    else
        hipLaunchKernelGGL(k_shard_scatter_stable, dim3((unsigned)l.maxChunks), dim3(kStabThreads), lds, s,
                           in, outKeys, p, w.hist);
    return hipGetLastError();
}

size_t scan_workspace_words(uint64_t n) { return (size_t)((n + kScanTile - 1) / kScanTile) + 1; }

hipError_t launch_exclusive_scan_u32(uint32_t* data, uint64_t n, uint32_t* sums, hipStream_t s)
{
    const uint64_t blocks = (n + kScanTile - 1) / kScanTile;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)blocks), dim3(kBlock), 0, s, data, n, sums, kNoGate);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kBlock), 0, s, sums, (uint32_t)blocks, kNoGate);
    hipLaunchKernelGGL(k_scan_add, dim3((unsigned)blocks), dim3(kBlock), 0, s, data, n, sums, kNoGate);
    return hipGetLastError();
}

hipError_t prj_set_attributes()
{
    // per device (hj_create calls this with its device current): the LDS join table and the stable split's staging
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_prj_join<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kJoinSlots * sizeof(uint32_t))) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_prj_join<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kJoinSlots * sizeof(uint32_t))) != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(k_shard_scatter_stable),
                               hipFuncAttributeMaxDynamicSharedMemorySize,
                               sizeof(uint32_t) * (kStabMaxFan * kStabGroups + kStabTile + (kStabTile >> 5) + 1));
}

}  // namespace hj

// hj_kernels.hip -- open-addressing build + probe kernels for gfx950 (MI355X).
//
// What these replace (reference paths relative to anilshanbhag/HTM-HashJoin):
//   k_build_atomic_min  HOT LOOP 1, NoCCHashBuild.hpp:37-62 / AtomicHashBuild.hpp:37-67
//                       (and the TSX group insert of HTMHashBuild.hpp:157-215, replaced outright)
//   k_probe             HOT LOOP 2, NoCCHashBuild.hpp:66-80 / AtomicHashBuild.hpp:71-85
//   k_table_sums        the untimed reductions, NoCCHashBuild.hpp:94-101 / AtomicHashBuild.hpp:100-107
//
// Semantics: the reference's parallel build is order dependent on duplicate
// keys (racy store / first-come CAS). The well-defined result is the one a
// single thread produces walking R in input order; these kernels reproduce
// exactly that result under any scheduling:
//   a slot holds (inputIndex << 32 | key), empty = all ones, and a tuple claims
//   a slot with a 64-bit atomicMin. If the previous holder had a larger index it
//   is displaced and re-inserted from the next slot with the budget it has left
//   (its displacement from its home slot is recoverable from slot and key);
//   if it had a smaller index the newcomer moves on. Slot values only ever
//   decrease, so the fixed point is unique: every tuple ends in the first slot
//   of its probe window that no smaller-indexed tuple finally occupies, which
//   is where sequential insertion puts it; tuples that run out of budget are
//   the sequential run's conflicts.
//
// All integer work, HBM/atomic bound; no MFMA.

#include "hj_device.h"

#include <type_traits>

namespace hj {

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;  // valid in lane 0
}

// Adds each wavefront's partial sums to the global counters, one atomic per
// wavefront and counter (skipped when the wavefront has nothing to add).
__device__ __forceinline__ void flush_counter(unsigned long long* dst, unsigned long long v)
{
    v = wave_sum(v);
    if ((threadIdx.x & (kWave - 1)) == 0 && v != 0) atomicAdd(dst, v);
}

static inline unsigned grid_for(uint64_t items, unsigned perBlock)
{
    uint64_t blocks = (items + perBlock - 1) / perBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256u * 8u) blocks = 256u * 8u;  // 256 CUs x 8 blocks, grid-stride the rest
    return (unsigned)blocks;
}

// ---------------------------------------------------------------------------
// table clear: 16-byte stores of the empty pattern
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) k_fill_empty(uint64_t* __restrict__ table, uint64_t nSlots, Gate gate,
                                                       Counters* __restrict__ fullRange, uint64_t tableSize)
{
    if (gate_closed(gate)) return;
    if (fullRange && blockIdx.x == 0 && threadIdx.x == 0) { fullRange->validLo = 0; fullRange->validHiEx = tableSize; }
    // table is hipMalloc'ed (256-B aligned); nSlots is even by construction
    ulonglong2* t2 = reinterpret_cast<ulonglong2*>(table);
    const uint64_t nv = nSlots >> 1;
    const ulonglong2 e = make_ulonglong2(kEmpty, kEmpty);
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (uint64_t)gridDim.x * kBlock)
        t2[v] = e;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (nSlots & 1)) table[nSlots - 1] = kEmpty;
}

void launch_fill_empty(uint64_t* table, uint64_t nSlots, Gate gate, hipStream_t s, Counters* fullRange, uint64_t tableSize)
{
    hipLaunchKernelGGL(k_fill_empty, dim3(grid_for(nSlots / 2, kBlock * 4)), dim3(kBlock), 0, s, table, nSlots, gate, fullRange, tableSize);
}

// ---------------------------------------------------------------------------
// build
// ---------------------------------------------------------------------------
// Inserts `mine` = (index << 32 | key) starting at slot `pos` with `budget` probes left. The home slot
// of a key is (key >> hshift) & homeMask, homeMask = tableSize - 1 (hj_device.h).
__device__ __forceinline__ void insert_priority(uint64_t* __restrict__ table, uint64_t homeMask,
                                                uint32_t hshift, uint32_t probeLen,
                                                uint64_t mine, uint64_t pos, uint32_t budget,
                                                unsigned long long& drops, unsigned long long& dropSum)
{
    for (;;) {
        if (budget == 0) {  // NoCCHashBuild.hpp:57-58
            drops += 1;
            dropSum += (uint32_t)mine;
            return;
        }
        const unsigned long long old =
            atomicMin(reinterpret_cast<unsigned long long*>(table + pos), (unsigned long long)mine);
        if (old == kEmpty || old == mine) return;  // claimed an empty slot
        if (old > mine) {
            // displaced a later tuple: carry it on from the next slot with the
            // budget it has left there
            mine = old;
            const uint64_t home = home_slot((uint32_t)old, hshift, homeMask);
            const uint32_t disp = (uint32_t)((pos - home) & homeMask);
            budget = probeLen - (disp + 1);
        } else {
            budget -= 1;  // occupied by an earlier tuple: NoCCHashBuild.hpp:50-53
        }
        pos = (pos + 1) & homeMask;
    }
}

// t = the element zero-extended to 64 bits (a tuple as it is, or a bare key)
__device__ __forceinline__ void build_one(uint64_t t, uint64_t idx, uint64_t* __restrict__ table,
                                          uint64_t mask, uint32_t hshift, uint32_t probeLen, const ShardCheck& sc,
                                          unsigned long long& drops, unsigned long long& dropSum,
                                          unsigned long long& inSum, unsigned long long& bad)
{
    inSum += t;
    bad += is_foreign((uint32_t)t, sc) ? (1ull << 32) : 0ull;      // foreign tuples ride in the high half of `bad`
    if ((t >> 32) != 0 || t == 0) { bad += 1; return; }
    const uint64_t mine = (idx << 32) | t;
    insert_priority(table, mask, hshift, probeLen, mine, home_slot((uint32_t)t, hshift, mask), probeLen, drops, dropSum);
}

// KEY32 = false: R holds 8-byte tuples (16-byte loads over the aligned body, head/tail element by one
// thread); KEY32 = true: bare keys, one 4-byte load per lane (this kernel is bound by its atomics).
template <bool KEY32>
__global__ void __launch_bounds__(kBlock)
k_build_atomic_min(const void* __restrict__ Rv, uint64_t n, uint64_t* __restrict__ table,
                   uint64_t mask, uint32_t hshift, uint32_t probeLen, uint64_t idxBase, ShardCheck sc, Counters* __restrict__ ctr,
                   Gate gate)
{
    if (gate_closed(gate)) return;
    unsigned long long drops = 0, dropSum = 0, inSum = 0, bad = 0;
    if constexpr (KEY32) {
        const uint32_t* K = static_cast<const uint32_t*>(Rv);
        for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock)
            build_one(K[i], idxBase + i, table, mask, hshift, probeLen, sc, drops, dropSum, inSum, bad);
    } else {
        const uint64_t* R = static_cast<const uint64_t*>(Rv);
        const uint64_t head = (n > 0 && (reinterpret_cast<uintptr_t>(R) & 8)) ? 1 : 0;
        const ulonglong2* R2 = reinterpret_cast<const ulonglong2*>(R + head);
        const uint64_t nv = (n - head) >> 1;
        for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (uint64_t)gridDim.x * kBlock) {
            const ulonglong2 t = R2[v];
            const uint64_t i = head + 2 * v;
            build_one(t.x, idxBase + i, table, mask, hshift, probeLen, sc, drops, dropSum, inSum, bad);
            build_one(t.y, idxBase + i + 1, table, mask, hshift, probeLen, sc, drops, dropSum, inSum, bad);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (head) build_one(R[0], idxBase, table, mask, hshift, probeLen, sc, drops, dropSum, inSum, bad);
            const uint64_t tail = head + 2 * nv;
            if (tail < n) build_one(R[tail], idxBase + tail, table, mask, hshift, probeLen, sc, drops, dropSum, inSum, bad);
        }
    }
    Counters::Shard* const sh = counter_shard(ctr);
    flush_counter(&sh->conflicts, drops);
    flush_counter(&sh->conflictSum, dropSum);
    flush_counter(&sh->inputSum, inSum);
    flush_counter(&sh->badKeys, bad & 0xFFFFFFFFull);
    flush_counter(&sh->foreign, bad >> 32);
}

void launch_build_atomic_min(const void* R, bool key32, uint64_t n, uint64_t* table, uint64_t tableSize, uint32_t hshift,
                             uint32_t probeLen, uint64_t idxBase, ShardCheck sc, Counters* ctr, Gate gate, hipStream_t s)
{
    if (key32)
        hipLaunchKernelGGL(k_build_atomic_min<true>, dim3(grid_for(n + 1, kBlock)), dim3(kBlock), 0, s,
                           R, n, table, tableSize - 1, hshift, probeLen, idxBase, sc, ctr, gate);
    else
        hipLaunchKernelGGL(k_build_atomic_min<false>, dim3(grid_for(n / 2 + 1, kBlock)), dim3(kBlock), 0, s,
                           R, n, table, tableSize - 1, hshift, probeLen, idxBase, sc, ctr, gate);
}

// ---------------------------------------------------------------------------
// probe
// ---------------------------------------------------------------------------
// sk = the S element zero-extended to 64 bits
__device__ __forceinline__ uint32_t probe_one(uint64_t sk, const uint64_t* __restrict__ table,
                                              uint64_t mask, uint32_t hshift, uint32_t probeLen,
                                              uint64_t validLo, uint64_t validHiEx)
{
    // a tuple with payload bits set can match nothing; outside the valid range no stored tuple can
    // have this home slot (hj_device.h, Counters)
    const uint32_t key = (uint32_t)sk;
    const uint64_t home = home_slot(key, hshift, mask);
    if ((sk >> 32) != 0 || home < validLo || home >= validHiEx) return 0;
    // NoCCHashBuild.hpp:70-79: walk at most probeLen consecutive slots from the
    // home slot, stop at the first empty one, count slots equal to the tuple.
    const uint64_t* p = table + home;
    uint32_t m = 0;
    if (probeLen == 4) {
        const uint64_t a = p[0], b = p[1], c = p[2], d = p[3];  // slack slots make this safe
        const bool ea = a != kEmpty, eb = ea && b != kEmpty, ec = eb && c != kEmpty, ed = ec && d != kEmpty;
        m += (ea && (uint32_t)a == key);
        m += (eb && (uint32_t)b == key);
        m += (ec && (uint32_t)c == key);
        m += (ed && (uint32_t)d == key);
    } else {
        for (uint32_t j = 0; j < probeLen; ++j) {
            const uint64_t v = p[j];
            if (v == kEmpty) break;
            m += ((uint32_t)v == key);
        }
    }
    return m;
}

// The same walk over a compact table (Counters::tableFormat == kFormatKeys4): one 4-byte key per slot, 0xFFFFFFFF = empty
__device__ __forceinline__ uint32_t probe_one_keys(uint64_t sk, const uint32_t* __restrict__ keys,
                                                   uint64_t mask, uint32_t hshift, uint32_t probeLen,
                                                   uint64_t validLo, uint64_t validHiEx)
{
    const uint32_t key = (uint32_t)sk;
    const uint64_t home = home_slot(key, hshift, mask);
    if ((sk >> 32) != 0 || home < validLo || home >= validHiEx) return 0;
    const uint32_t* p = keys + home;
    constexpr uint32_t e = 0xFFFFFFFFu;
    uint32_t m = 0;
    if (probeLen == 4) {
        const uint32_t a = p[0], b = p[1], c = p[2], d = p[3];    // slack slots make this safe
        const bool ea = a != e, eb = ea && b != e, ec = eb && c != e, ed = ec && d != e;
        m += (ea && a == key);
        m += (eb && b == key);
        m += (ec && c == key);
        m += (ed && d == key);
    } else {
        for (uint32_t j = 0; j < probeLen; ++j) {
            const uint32_t v = p[j];
            if (v == e) break;
            m += (v == key);
        }
    }
    return m;
}

// Window of one S element, loaded unconditionally (two 16-byte loads; the slack slots past the table end make
// that safe): an element that cannot match reads the window at `dummy` instead and counts nothing. No branch sits
// between the loads of a lane's elements, so all of them are in flight together.
struct Window { uint64_t a, b, c, d; uint32_t key; bool ok; };

__device__ __forceinline__ Window load_window(uint32_t key, const uint64_t* __restrict__ table, uint64_t mask,
                                              uint32_t hshift, uint64_t validLo, uint64_t validHiEx, uint64_t dummy)
{
    Window w;
    w.key = key;
    const uint64_t home = home_slot(key, hshift, mask);
    w.ok = home >= validLo && home < validHiEx;
    const uint64_t* p = table + (w.ok ? home : dummy);
    w.a = p[0]; w.b = p[1]; w.c = p[2]; w.d = p[3];
    return w;
}

struct WindowK { uint32_t a, b, c, d; uint32_t key; bool ok; };
__device__ __forceinline__ WindowK load_window_keys(uint32_t key, const uint32_t* __restrict__ keys, uint64_t mask,
                                                    uint32_t hshift, uint64_t validLo, uint64_t validHiEx, uint64_t dummy)
{
    WindowK w;
    w.key = key;
    const uint64_t home = home_slot(key, hshift, mask);
    w.ok = home >= validLo && home < validHiEx;
    const uint32_t* p = keys + (w.ok ? home : dummy);
    w.a = p[0]; w.b = p[1]; w.c = p[2]; w.d = p[3];
    return w;
}
__device__ __forceinline__ uint32_t count_window_keys(const WindowK& w)
{
    constexpr uint32_t e = 0xFFFFFFFFu;
    const bool ea = w.a != e, eb = ea && w.b != e, ec = eb && w.c != e, ed = ec && w.d != e;
    const uint32_t m = (uint32_t)(ea && w.a == w.key) + (uint32_t)(eb && w.b == w.key) + (uint32_t)(ec && w.c == w.key) +
                       (uint32_t)(ed && w.d == w.key);
    return w.ok ? m : 0u;
}

__device__ __forceinline__ uint32_t count_window(const Window& w)
{
    // NoCCHashBuild.hpp:70-79 with probeLength 4: stop at the first empty slot, count slots equal to the key
    const bool ea = w.a != kEmpty, eb = ea && w.b != kEmpty, ec = eb && w.c != kEmpty, ed = ec && w.d != kEmpty;
    const uint32_t m = (uint32_t)(ea && (uint32_t)w.a == w.key) + (uint32_t)(eb && (uint32_t)w.b == w.key) +
                       (uint32_t)(ec && (uint32_t)w.c == w.key) + (uint32_t)(ed && (uint32_t)w.d == w.key);
    return w.ok ? m : 0u;
}

// 16-byte loads over the aligned body (2 tuples or 4 keys per lane), the few elements before and after
// it by one thread. Tuples: two windows one after the other -- that version already runs at the chip's
// read-stream rate (6.1-6.35 TB/s), and batching its loads made it slower. Keys: four windows per lane,
// which one after the other meant five dependent round trips per iteration (4.2 TB/s); their eight loads are
// issued together.
template <bool KEY32, bool COMPACT>
__device__ __forceinline__ void probe_body(const void* __restrict__ Sv, uint64_t n, const uint64_t* __restrict__ table, uint64_t mask,
                                           uint32_t hshift, uint32_t probeLen, ShardCheck sc, Counters* __restrict__ ctr)
{
    using Elem = typename std::conditional<KEY32, uint32_t, uint64_t>::type;
    const uint32_t* const keys = reinterpret_cast<const uint32_t*>(table);      // COMPACT: the table buffer holds 4-byte keys
    (void)keys;
    constexpr uint64_t EPV = 16 / sizeof(Elem);
    unsigned long long matches = 0;
    uint32_t foreign = 0;                                          // shard check (off: always 0)
    const uint64_t validLo = ctr->validLo, validHiEx = ctr->validHiEx;
    const uint64_t dummy = validLo < mask ? validLo : 0;          // any in-table slot; this one is in cache
    const Elem* S = static_cast<const Elem*>(Sv);
    uint64_t head = ((16 - (reinterpret_cast<uintptr_t>(S) & 15)) & 15) / sizeof(Elem);
    if (head > n) head = n;
    const uint4* S4 = reinterpret_cast<const uint4*>(S + head);
    const uint64_t nv = (n - head) / EPV;
    for (uint64_t v = (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (uint64_t)gridDim.x * kBlock) {
        // S is read once: a nontemporal load leaves the caches to the table lines neighbouring lanes share (-2 % per step at 2^30)
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 tt = __builtin_nontemporal_load(reinterpret_cast<const u4*>(S4) + v);
        const uint4 t = make_uint4(tt.x, tt.y, tt.z, tt.w);
        if constexpr (KEY32) {
            foreign += (uint32_t)is_foreign(t.x, sc) + (uint32_t)is_foreign(t.y, sc) + (uint32_t)is_foreign(t.z, sc) + (uint32_t)is_foreign(t.w, sc);
            if constexpr (COMPACT) {
                if (probeLen == 4) {
                    const WindowK w0 = load_window_keys(t.x, keys, mask, hshift, validLo, validHiEx, dummy);
                    const WindowK w1 = load_window_keys(t.y, keys, mask, hshift, validLo, validHiEx, dummy);
                    const WindowK w2 = load_window_keys(t.z, keys, mask, hshift, validLo, validHiEx, dummy);
                    const WindowK w3 = load_window_keys(t.w, keys, mask, hshift, validLo, validHiEx, dummy);
                    matches += count_window_keys(w0) + count_window_keys(w1) + count_window_keys(w2) + count_window_keys(w3);
                } else {
                    matches += probe_one_keys(t.x, keys, mask, hshift, probeLen, validLo, validHiEx);
                    matches += probe_one_keys(t.y, keys, mask, hshift, probeLen, validLo, validHiEx);
                    matches += probe_one_keys(t.z, keys, mask, hshift, probeLen, validLo, validHiEx);
                    matches += probe_one_keys(t.w, keys, mask, hshift, probeLen, validLo, validHiEx);
                }
            } else if (probeLen == 4) {
                const Window w0 = load_window(t.x, table, mask, hshift, validLo, validHiEx, dummy);
                const Window w1 = load_window(t.y, table, mask, hshift, validLo, validHiEx, dummy);
                const Window w2 = load_window(t.z, table, mask, hshift, validLo, validHiEx, dummy);
                const Window w3 = load_window(t.w, table, mask, hshift, validLo, validHiEx, dummy);
                matches += count_window(w0) + count_window(w1) + count_window(w2) + count_window(w3);
            } else {
                matches += probe_one(t.x, table, mask, hshift, probeLen, validLo, validHiEx);
                matches += probe_one(t.y, table, mask, hshift, probeLen, validLo, validHiEx);
                matches += probe_one(t.z, table, mask, hshift, probeLen, validLo, validHiEx);
                matches += probe_one(t.w, table, mask, hshift, probeLen, validLo, validHiEx);
            }
        } else {
            foreign += (uint32_t)is_foreign(t.x, sc) + (uint32_t)is_foreign(t.z, sc);
            if constexpr (COMPACT) {
                matches += probe_one_keys(((uint64_t)t.y << 32) | t.x, keys, mask, hshift, probeLen, validLo, validHiEx);
                matches += probe_one_keys(((uint64_t)t.w << 32) | t.z, keys, mask, hshift, probeLen, validLo, validHiEx);
            } else {
                matches += probe_one(((uint64_t)t.y << 32) | t.x, table, mask, hshift, probeLen, validLo, validHiEx);
                matches += probe_one(((uint64_t)t.w << 32) | t.z, table, mask, hshift, probeLen, validLo, validHiEx);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        auto one = [&](uint64_t sk) {
            return COMPACT ? probe_one_keys(sk, keys, mask, hshift, probeLen, validLo, validHiEx)
                           : probe_one(sk, table, mask, hshift, probeLen, validLo, validHiEx);
        };
        for (uint64_t i = 0; i < head; ++i) { matches += one(S[i]); foreign += is_foreign((uint32_t)S[i], sc); }
        for (uint64_t i = head + nv * EPV; i < n; ++i) { matches += one(S[i]); foreign += is_foreign((uint32_t)S[i], sc); }
    }
    flush_counter(&counter_shard(ctr)->matches, matches);
    flush_counter(&counter_shard(ctr)->foreign, foreign);
}

// One launch for either table format: the build decides the format on the device (Counters::tableFormat), the probe is
// enqueued behind it without a host round trip and branches once, wave-uniformly.
template <bool KEY32>
__global__ void __launch_bounds__(kBlock)
k_probe(const void* __restrict__ Sv, uint64_t n, const uint64_t* __restrict__ table, uint64_t mask,
        uint32_t hshift, uint32_t probeLen, ShardCheck sc, Counters* __restrict__ ctr)
{
    if (ctr->tableFormat == kFormatKeys4) probe_body<KEY32, true>(Sv, n, table, mask, hshift, probeLen, sc, ctr);
    else probe_body<KEY32, false>(Sv, n, table, mask, hshift, probeLen, sc, ctr);
}

void launch_probe(const void* S, bool key32, uint64_t n, const uint64_t* table, uint64_t tableSize, uint32_t hshift,
                  uint32_t probeLen, ShardCheck sc, Counters* ctr, hipStream_t s)
{
    if (key32)
        hipLaunchKernelGGL(k_probe<true>, dim3(grid_for(n / 4 + 1, kBlock)), dim3(kBlock), 0, s,
                           S, n, table, tableSize - 1, hshift, probeLen, sc, ctr);
    else
        hipLaunchKernelGGL(k_probe<false>, dim3(grid_for(n / 2 + 1, kBlock)), dim3(kBlock), 0, s,
                           S, n, table, tableSize - 1, hshift, probeLen, sc, ctr);
}

// ---------------------------------------------------------------------------
// table checksums
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_table_sums(const uint64_t* __restrict__ table, uint64_t tableSize, uint64_t halfSlots, Counters* __restrict__ ctr)
{
    unsigned long long half = 0, full = 0;
    const ulonglong2* t2 = reinterpret_cast<const ulonglong2*>(table);
    // only [validLo, validHiEx + 512) holds defined values; both bounds are even
    const uint64_t lo = ctr->validLo;
    uint64_t hi = ctr->validHiEx + 512;
    if (hi > tableSize) hi = tableSize;
    if (ctr->tableFormat == kFormatKeys4) {               // compact table: 4-byte keys, two per 8 bytes
        const uint2* k2 = reinterpret_cast<const uint2*>(table);
        for (uint64_t v = (lo >> 1) + (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < (hi >> 1); v += (uint64_t)gridDim.x * kBlock) {
            const uint2 t = k2[v];
            const uint64_t a = t.x == 0xFFFFFFFFu ? 0 : t.x, b = t.y == 0xFFFFFFFFu ? 0 : t.y;
            full += a + b;
            if (2 * v < halfSlots) half += a;
            if (2 * v + 1 < halfSlots) half += b;
        }
        flush_counter(&ctr->tableSumHalf, half);
        flush_counter(&ctr->tableSumFull, full);
        return;
    }
    const uint64_t nv = hi >> 1;
    for (uint64_t v = (lo >> 1) + (uint64_t)blockIdx.x * kBlock + threadIdx.x; v < nv; v += (uint64_t)gridDim.x * kBlock) {
        const ulonglong2 t = t2[v];
        const uint64_t a = (t.x == kEmpty) ? 0 : (uint32_t)t.x;
        const uint64_t b = (t.y == kEmpty) ? 0 : (uint32_t)t.y;
        full += a + b;
        if (2 * v < halfSlots) half += a;
        if (2 * v + 1 < halfSlots) half += b;
    }
    flush_counter(&ctr->tableSumHalf, half);
    flush_counter(&ctr->tableSumFull, full);
}

void launch_table_sums(const uint64_t* table, uint64_t tableSize, uint64_t halfSlots, Counters* ctr, hipStream_t s)
{
    hipLaunchKernelGGL(k_table_sums, dim3(grid_for(tableSize / 2, kBlock * 4)), dim3(kBlock), 0, s,
                       table, tableSize, halfSlots, ctr);
}

// ---------------------------------------------------------------------------
// Zipf draws on the device (hj_zipf_next_dev): the binary search of gen_zipf (mc/src/genzipf.c:118-151) for n raw
// rand() values against the cumulative table; the host only produces the serial rand() stream.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
k_zipf_lookup(const int* __restrict__ raw, uint64_t n, const double* __restrict__ lut, const uint32_t* __restrict__ alphabet,
              uint32_t alphabetSize, uint64_t* __restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const double r = ((double)raw[i]) / 2147483647.0;      // (double) rand() / RAND_MAX
        uint32_t pos = 0;
        if (!(lut[0] >= r)) {
            uint32_t left = 0, right = alphabetSize - 1;
            while (right - left > 1) {
                const uint32_t m = (left + right) / 2;
                if (lut[m] < r) left = m; else right = m;
            }
            pos = right;
        }
        out[i] = alphabet[pos];
    }
}

void launch_zipf_lookup(const int* raw, uint64_t n, const double* lut, const uint32_t* alphabet, uint32_t alphabetSize,
                        uint64_t* out, hipStream_t s)
{
    hipLaunchKernelGGL(k_zipf_lookup, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, s, raw, n, lut, alphabet, alphabetSize, out);
}

__global__ void k_set_full_range(uint64_t tableSize, Counters* __restrict__ ctr, Gate gate)
{
    if (gate_closed(gate)) return;
    if (threadIdx.x == 0 && blockIdx.x == 0) { ctr->validLo = 0; ctr->validHiEx = tableSize; }
}

void launch_set_full_range(uint64_t tableSize, Counters* ctr, Gate gate, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_full_range, dim3(1), dim3(64), 0, s, tableSize, ctr, gate);
}

}  // namespace hj

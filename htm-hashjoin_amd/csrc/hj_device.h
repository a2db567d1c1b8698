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

// Radix-sharded input (multi-GPU): all keys of a shard share their low `strip` bits (= the shard number),
// so a slot stores key' = key >> strip in its low (32 - strip) bits and gives the freed bits to the
// index: slot = (globalIndex << (32 - strip)) | key'. strip = 0 is the single-GPU format (idx << 32 | key).
__host__ __device__ inline uint32_t key_mask(uint32_t strip) { return 0xFFFFFFFFu >> strip; }
__host__ __device__ inline uint32_t slot_key(uint64_t slot, uint32_t strip) { return (uint32_t)slot & key_mask(strip); }
__host__ __device__ inline uint64_t full_key(uint32_t keyPrime, uint32_t strip, uint32_t shard) { return ((uint64_t)keyPrime << strip) | shard; }

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kWave = 64;

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
    unsigned long long spare[1];
};

// ---- launch wrappers (defined in hj_kernels.hip) ---------------------------
void launch_fill_empty(uint64_t* table, uint64_t nSlots, hipStream_t s);
void launch_build_atomic_min(const uint64_t* R, uint64_t n, uint64_t* table,
                             uint64_t tableSize, uint32_t probeLen, uint64_t idxBase,
                             Counters* ctr, hipStream_t s);
void launch_build_packed(const uint64_t* packed, uint64_t n, uint64_t* table, uint64_t tableSize,
                         uint32_t strip, uint32_t shard, uint32_t probeLen, Counters* ctr, hipStream_t s);
void launch_probe(const uint64_t* S, uint64_t n, const uint64_t* table, uint64_t tableSize,
                  uint32_t strip, uint32_t probeLen, Counters* ctr, hipStream_t s);
void launch_table_sums(const uint64_t* table, uint64_t tableSize, uint64_t halfSlots, uint32_t strip, uint32_t shard,
                       Counters* ctr, hipStream_t s);
// Marks the whole table valid (variant 1 clears and may touch all of it).
void launch_set_full_range(uint64_t tableSize, Counters* ctr, hipStream_t s);
// multi-GPU destination split (defined in hj_prj.hip: one order-preserving radix pass)
size_t shard_work_bytes(uint64_t n, uint32_t nShards);
void launch_shard_hist(const uint64_t* in, uint64_t n, uint32_t nShards, void* work, unsigned long long* counts,
                       hipStream_t s);
void launch_shard_scatter_ordered(const uint64_t* in, uint64_t n, uint32_t nShards, void* work, uint64_t packIdxBase,
                                  uint32_t strip, uint64_t* out, hipStream_t s);

// ---- ownership build (defined in hj_build_own.hip) ---------------------------
size_t own_queue_bytes(uint64_t rSize);
size_t own_owner_bytes(uint64_t tableSize);
bool   own_supported(uint64_t tableSize);
void launch_sample_locality(const uint64_t* R, uint64_t n, uint64_t tableSize, uint32_t strip, uint32_t nSample,
                            unsigned int* fitCount, hipStream_t s);
// phase A (LDS window) -> clear of unowned blocks -> phase B (deferred tuples).
// Writes every table slot exactly once: no separate launch_fill_empty needed.
void launch_build_own(const uint64_t* R, uint64_t n, bool packed, uint32_t strip, uint32_t shard, uint64_t* table,
                      uint64_t tableSize, uint32_t probeLen, uint64_t idxBase, void* ownerBuf, void* queueBuf,
                      unsigned long long* queueCount, Counters* ctr, hipEvent_t evPhaseA, hipStream_t s);

// ---- PRJ (defined in hj_prj.hip) -------------------------------------------
struct PrjPlan {
    uint32_t radixBits;   // total
    uint32_t bits1, bits2;
    uint64_t maxChunks1, maxChunks2;   // chunk descriptors per pass (upper bounds)
    size_t   workspaceBytes;           // everything below, excluding tuple buffers
};
// Sizes the workspace for (nR, nS).
PrjPlan prj_plan(uint64_t nR, uint64_t nS, uint32_t radixBits);
struct PrjBuffers {
    uint64_t* tmpA;      // max(nR,nS) tuples
    uint64_t* partR;     // nR tuples (final partitioned R)
    uint64_t* partS;     // nS tuples
    void*     work;      // plan.workspaceBytes
};
// Enqueues partition(R), partition(S) and the per-partition LDS join.
// evPartDone (may be null) is recorded between partitioning and join.
void launch_prj(const PrjPlan& plan, const PrjBuffers& buf,
                const uint64_t* R, uint64_t nR, const uint64_t* S, uint64_t nS,
                Counters* ctr, hipEvent_t evPartDone, hipStream_t s);

}  // namespace hj

/* htm_hashjoin_sharded.h -- the radix-sharded join of one node behind the C ABI (libhtmjoin_sharded.so).
 *
 * Replaces nothing in the reference (it is single-process shared memory: NoCCHashBuild.hpp:37 is its only
 * parallelism); this is the multi-GPU entry SURVEY.md 8b asks for -- `hj_params{..., num_gpus, radix_bits}`, "one
 * host thread drives all GPUs (or one thread per GPU inside the lib)" -- and 8e lays out: shard by a radix digit of
 * the key (HASH_BIT_MODULO, mc/src/parallel_radix_join.c:59), ONE exchange per relation (grouped ncclSend / ncclRecv,
 * every pair directly over xGMI), local open-addressing build + probe per GPU, counters added up. One process, one
 * host thread per GPU inside the library, RCCL linked directly. The per-GPU kernels are libhtmjoin_hip.so's
 * (htm_hashjoin.h: hj_shard_histogram_dev / hj_shard_scatter_dev / hj_build_keys_dev / hj_probe_keys_dev); the
 * Python host (htm-hashjoin_amd/sharded.py, one process per GPU over torch.distributed) runs the same steps and its
 * tests pin the semantics: shard g's table holds the tuples whose digit is g, inserted in GLOBAL input order with the
 * reference's probe budget.
 *
 * Only 32-bit keys travel (a DataGen tuple is its key, DataGen.hpp:29), 4 bytes per tuple; rank g must hold the g-th
 * contiguous piece of each relation, so that position in the receive buffer is global input order. */
#ifndef HTM_HASHJOIN_SHARDED_H
#define HTM_HASHJOIN_SHARDED_H

#include "htm_hashjoin.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hj_sharded hj_sharded;

/* which key bits pick the destination GPU */
enum { HJ_SPLIT_LOW = 0,   /* key & (G-1): balanced for any keys; (G-1)/G of a relation held as key ranges moves      */
       HJ_SPLIT_HIGH = 1   /* top log2 G bits of (key-1) over [1, maxKey]: a range split, such pieces mostly stay put  */ };

typedef struct {
    uint32_t nRanks;
    uint32_t mode;             /* digit position | HJ_SHARD_ONE_BASED, as hj_shard_histogram_dev takes it               */
    uint32_t homeShift;        /* home slot = (key >> homeShift) & (tableSizePerRank - 1)                               */
    uint32_t reserved;
    uint64_t keysMovedR, keysMovedS;   /* keys that crossed GPUs (all ranks): x 4 bytes over xGMI                       */
    uint64_t maxMessageKeys;           /* largest single rank-to-rank message                                           */
    uint64_t tableSizePerRank;
} hj_sharded_stats;

/* nDevices a power of two <= 64; device g becomes rank g. Creates one context (htm_hashjoin.h) and one stream per
 * device and, for nDevices > 1, the RCCL communicators (ncclCommInitAll). */
int  hj_sharded_create(const int *devices, int nDevices, hj_sharded **out);
void hj_sharded_destroy(hj_sharded *s);
const char *hj_sharded_last_error(const hj_sharded *s);
int  hj_sharded_ranks(const hj_sharded *s);
/* device memory of rank `rank` (for hosts that never call HIP themselves, like csrc/main.cpp) */
int  hj_sharded_alloc(hj_sharded *s, int rank, uint64_t bytes, void **dptr);
int  hj_sharded_free(hj_sharded *s, int rank, void *dptr);
int  hj_sharded_copy_h2d(hj_sharded *s, int rank, void *dst_dev, const void *src_host, uint64_t bytes);

/* One build + probe over the pieces dR[g] (nR[g] tuples, device memory of rank g) and dS[g] (may be NULL: build only).
 * params: algo HJ_ALGO_ATOMIC (or NOCC: same table, the nocc outputSum), probeLength, buildVariant as for hj_run.
 * split / maxKey: see above (maxKey = upper bound of the keys, DataGen: the relation size; only HJ_SPLIT_HIGH reads it).
 * tableSize: slots per rank, a power of two; 0 = 2 * nextpow2(largest piece of R).
 * total: the counters of all ranks added up (build_us / probe_us: the slowest rank's); stats may be NULL.
 * Blocks until every rank has finished. Returns HJ_OK or the first failing rank's status (hj_sharded_last_error). */
int hj_sharded_join(hj_sharded *s, const hj_params *params, uint32_t split, uint64_t maxKey, uint64_t tableSize,
                    const uint64_t *const *dR, const uint64_t *nR, const uint64_t *const *dS, const uint64_t *nS,
                    hj_result *total, hj_sharded_stats *stats);

/* Host-side plan arithmetic, exported for tests (no GPU needed).
 * counts[g * nRanks + p] = keys rank g sends to rank p. sendOff / recvOff: nRanks rows of nRanks + 1 offsets -- where
 * rank g's split output for destination p starts, and where the piece from SOURCE rank p starts in rank g's receive
 * buffer (pieces in source-rank order: position = global input order). */
int hj_sharded_plan(uint32_t nRanks, const uint64_t *counts, uint64_t *sendOff, uint64_t *recvOff, uint64_t *recvTotal,
                    uint64_t *maxMessage, uint64_t *moved);
/* mode for hj_shard_histogram_dev / hj_shard_scatter_dev and the home shift of the local tables */
uint32_t hj_sharded_mode(uint32_t nRanks, uint32_t split, uint64_t maxKey, uint32_t *homeShift);

#ifdef __cplusplus
}
#endif
#endif

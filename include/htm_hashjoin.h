/*
 * htm_hashjoin.h -- C ABI of libhtmjoin_hip.so, the MI355X (gfx950) hash-join
 * build+probe engine.
 *
 * This is the drop-in boundary for the hot path of anilshanbhag/HTM-HashJoin.
 * The reference has no FFI layer: its operator interface is a set of C++ free
 * functions picked by a string compare in main (main.cpp:99-108 with probe,
 * :115-122 build only), plus mc's function-pointer table
 * (mc/src/main.c:262-301).  Each entry point below names the reference
 * interface it replaces.  Plain pointers and sizes only; no HIP, torch or C++
 * types cross this boundary.  INTEGRATION.md shows the reference-side binding.
 *
 * Tuple layout (A0): one uint64_t per tuple whose value is the key; seen as
 * little-endian {uint32 key; uint32 payload=0} it is mc's tuple_t
 * (include/DataGen.hpp:29, mc/src/types.h:34-37).
 *
 * Error behaviour: the reference returns void and exits on allocation failure
 * (HTMHashBuild.hpp:66-70, mc MALLOC_CHECK).  This library never exits: every
 * call returns HJ_OK (0) or a negative hj_status; hj_last_error() gives text.
 * There is NO CPU fallback: without a usable gfx950 device hj_create fails
 * with HJ_ERR_NO_DEVICE.
 */
#ifndef HTM_HASHJOIN_H
#define HTM_HASHJOIN_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HJ_ABI_VERSION 4

typedef enum {
    HJ_OK                  = 0,
    HJ_ERR_INVALID         = -1,  /* bad argument / unsupported size           */
    HJ_ERR_NO_DEVICE       = -2,  /* no HIP device / not gfx950                */
    HJ_ERR_HIP             = -3,  /* a HIP runtime call failed                 */
    HJ_ERR_OOM             = -4,  /* device allocation failed                  */
    HJ_ERR_KEY_RANGE       = -5,  /* a tuple has non-zero payload bits or is 0 */
    HJ_ERR_UNKNOWN_ALGO    = -6,
    HJ_ERR_STATE           = -7   /* call order violated (e.g. probe w/o build)*/
} hj_status;

/* Which reference operator the call stands in for. NOCC and ATOMIC run the same
 * order-deterministic open-addressing kernels (the racy store and the CAS loop
 * replaced outright) and differ only in which checksum quirk outputSum follows
 * (see hj_result). HTM builds the reference's OTHER table: 32-byte buckets of
 * three tuples, bucket = (key / 3) & (numBuckets - 1), numBuckets =
 * nextpow2(rSize / 3 + 1), tuples beyond the third ("conflicts") chained into
 * overflow buckets, probe = bucket + chain (HTMHashBuild.hpp:41-45, 61-62, 176-183,
 * 231-279, 291-305); Intel TSX is replaced outright by the index-priority fill,
 * so a bucket holds its three lowest-indexed tuples whatever the scheduling. */
typedef enum {
    HJ_ALGO_NOCC   = 0,  /* NoCCHashBuild   (NoCCHashBuild.hpp:13-151)   */
    HJ_ALGO_ATOMIC = 1,  /* AtomicHashBuild (AtomicHashBuild.hpp:14-157) */
    HJ_ALGO_HTM    = 2,  /* HTMHashBuild    (HTMHashBuild.hpp:54-464): bucketised table + overflow chains;
                            rSize need not be a power of two; totalMatches = true join cardinality */
    HJ_ALGO_PRJ    = 3,  /* mc PRO          (mc/src/parallel_radix_join.c:1305) */
    HJ_ALGO_AUTO   = 4   /* the reference's adaptive idea (README.md:6, the sampling pre-round of
                            HTMHashBuild.hpp:98-154 and its 0.4 % / 2 % thresholds :209-210): sample R for
                            locality; with locality run the no-partition path (HJ_ALGO_ATOMIC, LDS-window
                            build), otherwise the radix join (HJ_ALGO_PRJ) -- also when rSize is not a power of
                            two, which the table join does not take. hj_result.algoUsed says which;
                            totalMatches agrees between the two whenever R has unique keys */
} hj_algo;

/* Mirrors the trailing arguments of the reference signatures
 * (NoCCHashBuild.hpp:13-19; HTMHashBuild.hpp:54-60) and mc's compile-time
 * NUM_RADIX_BITS (mc/src/prj_params.h:16). Zero means "reference default". */
typedef struct {
    uint32_t algo;            /* hj_algo                                        */
    uint32_t scaleOutput;     /* accepted, unused: tableSize is 2*rSize
                                 (main.cpp:59-60, NoCCHashBuild.hpp:20)         */
    uint32_t numPartitions;   /* default 64 (main.cpp:84); informational        */
    uint32_t probeLength;     /* default 4  (main.cpp:80)                       */
    uint32_t transactionSize; /* default 16 (main.cpp:82); echoed for htm       */
    uint32_t radixBits;       /* PRJ only. 0 = auto (>= 14 so that every
                                 R partition fits one LDS table)                */
    uint32_t buildVariant;    /* 0 = auto (samples R for locality, like the pre-round of
                                 HTMHashBuild.hpp:100-154); 1 = global atomicMin kernel;
                                 2 = block-ownership + workgroup LDS-window kernel (locality
                                 up to a shuffle window of ~2000 positions);
                                 3 = wavefront-private LDS rings over statically owned slot
                                 ranges (tight locality, the reference's default
                                 --shuffleRange 16);
                                 4 = the same rings writing a COMPACT table (4-byte keys: the
                                 index words that order the inserts never leave the LDS; no
                                 deferred phase; if the input needs one, 3 redoes the table,
                                 decided on the device) -- what 0 picks where 3 would do;
                                 all give the same table (hj_export_table) and counters    */
    uint32_t prjMode;         /* PRJ: 0 = partition large relations without histograms (fragments
                                 sized for uniform low key bits, checked; the exact passes of
                                 parallel_radix_join.c:586-626 run instead when one overflows);
                                 1 = exact passes only; 2 = as 0 at any size that can be laid out   */
    uint32_t reserved[4];
} hj_params;

/* Everything the reference prints in its JSON line (NoCCHashBuild.hpp:127-146,
 * AtomicHashBuild.hpp:133-152) plus device timings. */
typedef struct {
    uint64_t rSize, sSize, tableSize;
    uint64_t conflicts;       /* tuples that exhausted probeLength ("conflicts") */
    uint64_t totalMatches;
    uint64_t inputSum;        /* sum of R                                        */
    uint64_t tableSumHalf;    /* sum of table[0..rSize)                          */
    uint64_t tableSumFull;    /* sum of table[0..tableSize)                      */
    uint64_t conflictSum;     /* sum of dropped keys                             */
    uint64_t outputSum;       /* nocc:  tableSumHalf + conflictSum (the
                                 NoCCHashBuild.hpp:94 quirk, kept for log parity);
                                 atomic/htm: tableSumFull + conflictSum           */
    uint64_t prjChecksum;     /* PRJ: sum of bucket idx == mc PRO "Results"
                                 (parallel_radix_join.c:256) when radixBits=14    */
    uint64_t prjPartitions;   /* PRJ: number of final partitions                  */
    uint32_t radixBits;       /* PRJ: bits actually used                          */
    uint32_t buildVariant;    /* kernel actually used (4 only if the compact build held) */
    /* device time of the last call of each phase, from HIP events on the
     * context's stream, in microseconds */
    double clear_us, build_us, probe_us, partition_us, join_us, total_us;
    double h2d_us;            /* hj_run only: host->device copies (reported
                                 separately, never part of total_us)              */
    uint64_t buildDeferred;   /* buildVariant 2: tuples that left the LDS window and
                                 were finished by the global-atomic phase            */
    double   buildPhaseA_us;  /* buildVariant 2: device time of k_build_own alone
                                 (build_us also covers k_clear_unowned and
                                 k_build_deferred)                                   */
    uint32_t algoUsed;        /* hj_algo that produced this result (differs from
                                 hj_params.algo only for HJ_ALGO_AUTO)               */
    uint32_t prjPath;         /* PRJ: 0 = exact (histogram) passes; 1 = histogram-free passes;
                                 2 = histogram-free passes overflowed, exact passes redid the join */
    uint64_t foreignTuples;   /* hj_set_shard_check: build + probe tuples whose
                                 destination is another shard (0 when the check is off) */
    double   prjScatterPass1R_us; /* PRJ: device time of the pass-1 scatter of R alone (the
                                 dominant kernel: 8 B read + 4 B written per tuple)      */
    /* HJ_ALGO_HTM: conflicts = the reference's conflictCount (tuples that found their
     * bucket full, HTMHashBuild.hpp:181-183, :225-228), tableSumFull = sum of the tuples in
     * primary buckets, outputSum = tableSumFull + htmOverflowSum (== inputSum) */
    uint64_t htmBuckets;          /* numBuckets                                           */
    uint64_t htmOverflowBuckets;  /* overflow buckets linked into chains (:231-279)       */
    uint64_t htmOverflowSum;      /* sum of the tuples they hold (== conflictSum)         */
    uint64_t compactFallback;     /* open addressing: 0 = the compact ring build (buildVariant 4) held or was not
                                     tried; else why it handed over to the classic build -- bit 0: a tuple outside
                                     ring and range (no locality there), bit 1: more than 64 walks across one seam,
                                     bit 2: key 0xFFFFFFFF, bit 3: a tuple far below its chunk's range, bit 4: a
                                     seam's two sides disagree (the shadow granule missed a tuple).
                                     HJ_ALGO_HTM: bit 8 = the chain phase could not run in LDS behind the ring build (key
                                     range of a chunk or its conflicts too large for the LDS image) and the generic
                                     chain kernels redid it                                                            */
} hj_result;

typedef struct hj_ctx hj_ctx;

/* ---- context ------------------------------------------------------------- */
int  hj_abi_version(void);
int  hj_device_count(int *count);
/* Binds to `device`, checks it is gfx950, creates a private stream. */
int  hj_create(int device, hj_ctx **out);
/* Same, but launches on the caller's stream (a hipStream_t passed as void*;
 * NULL = the default stream). Lets a host that owns streams (PyTorch) time and
 * order the kernels itself. */
int  hj_create_on_stream(int device, void *hip_stream, hj_ctx **out);
void hj_destroy(hj_ctx *ctx);
const char *hj_strerror(int status);
const char *hj_last_error(const hj_ctx *ctx);
int  hj_synchronize(hj_ctx *ctx);

/* ---- one-shot operator, host buffers ------------------------------------- */
/* Replaces NoCCHashBuild/AtomicHashBuild/HTMHashBuild(relR, rSize, relS, sSize,
 * ...) as called at main.cpp:99-104, and mc's PRO(relR, relS, nthreads)
 * (mc/src/main.c:292-301) for HJ_ALGO_PRJ. relS may be NULL / sSize 0 for the
 * build-only variant (main.cpp:115-120, ENABLE_PROBE 0). Inputs are read-only
 * and stay host-owned; device memory is owned by ctx. Like the reference
 * (NoCCHashBuild.hpp:24-34) allocation, host->device copies and checksums are
 * outside total_us. */
int hj_run(hj_ctx *ctx, const hj_params *params,
           const uint64_t *relR, uint64_t rSize,
           const uint64_t *relS, uint64_t sSize, hj_result *out);

/* ---- split operator, device-resident buffers ------------------------------ */
/* Allocate the table / partition workspace for these sizes (the `new[]` block
 * of NoCCHashBuild.hpp:24-31). Idempotent; grows only. */
int hj_reserve(hj_ctx *ctx, const hj_params *params, uint64_t rSize, uint64_t sSize);
/* HOT LOOP 1 (NoCCHashBuild.hpp:37-62 / AtomicHashBuild.hpp:37-67): clears the
 * table and inserts dR[0..rSize). Asynchronous on the context's stream, also
 * with buildVariant 0: the locality pre-round's decision is taken and acted on
 * by the device (no read-back), hj_result.buildVariant reports it afterwards.
 * idxBase = global index of dR[0] (0 unless R is a shard of a larger input). */
int hj_build_dev(hj_ctx *ctx, const uint64_t *dR, uint64_t rSize, uint64_t idxBase);
/* HOT LOOP 2 (NoCCHashBuild.hpp:66-80): probes dS[0..sSize) against the table
 * of the last hj_build_dev and accumulates totalMatches. Asynchronous. */
int hj_probe_dev(hj_ctx *ctx, const uint64_t *dS, uint64_t sSize);
/* PRJ (parallel_radix_join.c:808-1122): radix-partitions dR and dS and joins
 * each partition pair in LDS. Asynchronous. dS may be NULL (fork behaviour:
 * R-side only, checksum only). */
int hj_prj_join_dev(hj_ctx *ctx, const uint64_t *dR, uint64_t rSize,
                    const uint64_t *dS, uint64_t sSize);
/* Build + probe of dR x dS by whatever hj_reserve was given: HJ_ALGO_NOCC/ATOMIC/HTM = hj_build_dev(idxBase 0) then
 * hj_probe_dev; HJ_ALGO_PRJ = hj_prj_join_dev; HJ_ALGO_AUTO = one locality pre-round over dR (k_sample_locality, one
 * small device->host read-back) and then one of the two. Asynchronous apart from that read-back. */
int hj_join_dev(hj_ctx *ctx, const uint64_t *dR, uint64_t rSize,
                const uint64_t *dS, uint64_t sSize);
/* The untimed reductions of NoCCHashBuild.hpp:85-113 (table sums). Async. */
int hj_checksums_dev(hj_ctx *ctx);
/* Waits for the stream and returns counters + timings of the calls above. */
int hj_fetch_result(hj_ctx *ctx, hj_result *out);
/* Copies the open-addressing table to host in the reference's format
 * (tableSize slots, value = key, 0 = empty). */
int hj_export_table(hj_ctx *ctx, uint64_t *host_table, uint64_t tableSize);
/* HJ_ALGO_HTM: copies the bucket table to host as the reference's `struct Bucket
 * {uint64_t tuples[3]; uint32_t count; uint32_t nextIndex;}` (HTMHashBuild.hpp:41-45,
 * 32 bytes): host_buckets[numBuckets] and host_overflows[0 .. *nOverflow] (index 0
 * unused, nextIndex is 1-based, 0 = end of chain; overflowCap >= *nOverflow + 1
 * entries; pass NULL / 0 to learn *nOverflow first). A chain's head is its newest
 * overflow bucket, as in the reference; the physical overflow indices are this
 * library's (a bucket's overflow buckets are neighbours), not the reference's
 * creation order. */
int hj_export_buckets(hj_ctx *ctx, void *host_buckets, uint64_t numBuckets,
                      void *host_overflows, uint64_t overflowCap, uint64_t *nOverflow);

/* ---- multi-GPU sharding helpers (new design) ---------------- */
/* dest(key) = ((key - b) >> d) & (nShards-1), nShards a power of two <= 64, with d = mode & 0xFF the bit position
 * of the radix digit and b = 1 if mode has HJ_SHARD_ONE_BASED set, else 0: HASH_BIT_MODULO(key, MASK, R) of
 * parallel_radix_join.c:59 with R = d. d = 0 takes the low key bits: balanced for any key distribution, but on a
 * relation whose pieces are contiguous key ranges (the reference's near-sorted inputs, held piecewise) all but
 * 1/nShards of the tuples change GPU. d = (bits of the key domain) - log2(nShards) takes the HIGH bits, i.e. a
 * range split: such pieces mostly stay where they are and only what does not fit a rank's range moves
 * (HJ_SHARD_ONE_BASED makes the ranges of DataGen's keys 1..N come out even). Both relations of a join must use
 * the same mode. */
#define HJ_SHARD_ONE_BASED 0x100u
/* Counts tuples per destination into dCounts[nShards] (device, uint64) and keeps
 * the per-chunk write cursors for the scatter of the same input. Async. */
int hj_shard_histogram_dev(hj_ctx *ctx, const uint64_t *dIn, uint64_t n,
                           uint32_t nShards, uint32_t mode, uint64_t *dCounts);
/* Scatter of the n tuples of dIn, grouped by destination, into dOutKeys as bare
 * 32-bit keys (a DataGen tuple is its key, DataGen.hpp:29: the exchange then
 * moves 4 bytes per tuple instead of 8). Must follow hj_shard_histogram_dev on
 * the same (dIn, n). The input order is preserved inside every destination
 * (scan-based cursors, no atomics across workgroups). That is what carries the
 * reference's insertion order through the exchange: when rank g holds the g-th
 * contiguous piece of the relation and the receiver lays the pieces out in rank
 * order, position in the received buffer IS global input order, so no index
 * has to travel. Async. */
int hj_shard_scatter_dev(hj_ctx *ctx, const uint64_t *dIn, uint64_t n,
                         uint32_t nShards, uint32_t mode,
                         const uint64_t *dCounts, uint32_t *dOutKeys);
/* hj_build_dev / hj_probe_dev for bare keys (the received side of the exchange):
 * tuple i of dKeys has index i; table of tableSize slots (a power of two,
 * reserved via hj_reserve(rSize = tableSize/2)); home slot of a key =
 * (key >> homeShift) & (tableSize-1) with homeShift = log2(nShards): inside a
 * shard the low key bits are the same for every key and would leave all but
 * every nShards-th slot unused. */
int hj_build_keys_dev(hj_ctx *ctx, const uint32_t *dKeys, uint64_t n,
                      uint32_t homeShift, uint64_t tableSize);
int hj_probe_keys_dev(hj_ctx *ctx, const uint32_t *dKeys, uint64_t n);
/* Optimistic joining in place: under a range split the pieces a rank holds are often already its shards. While a
 * check is set (nShards > 0), every later build and probe on this context also counts, at no extra pass over the
 * data, the tuples whose destination under (nShards, mode) is NOT shardId -> hj_result.foreignTuples. A caller
 * joins its pieces in place with hj_build_dev / hj_probe_dev and takes the result if the count is 0 on every
 * rank, else redoes the step with split + exchange. nShards = 0 switches the check off. */
int hj_set_shard_check(hj_ctx *ctx, uint32_t nShards, uint32_t mode, uint32_t shardId);

/* Host-only arithmetic (no device needed): how hj_reserve sizes the PRJ histogram workspace for (rSize, sSize,
 * radixBits; 0 = auto). out[0] = workspace bytes, out[1] = histogram entries planned, out[2] / out[3] = entries the
 * passes over R / over S write. Each relation is chunked by its own size (mc's per-thread slices,
 * parallel_radix_join.c:586-617, become per-chunk histograms), so out[1] >= max(out[2], out[3]) must hold. */
int hj_prj_workspace_info(uint64_t rSize, uint64_t sSize, uint32_t radixBits, uint64_t out[4]);
/* Host-only arithmetic: the fragment geometry of the histogram-free passes for (rSize, sSize, radixBits, prjMode).
 * out[0] = 1 when those passes would be enqueued (both relations qualify), else 0 (exact passes only);
 * out[1..5] = R's chunks of pass 1, slots per pass-1 fragment, tuples per pass-1 chunk, chunks of pass 2 per pass-1
 * partition (= fragments per final partition), slots per pass-2 fragment; out[6..10] = the same for S; out[11], out[12] =
 * radix bits of pass 1 and pass 2. A relation with out[1] (out[6]) == 0 does not qualify. */
int hj_prj_fragment_info(uint64_t rSize, uint64_t sSize, uint32_t radixBits, uint32_t prjMode, uint64_t out[13]);

/* ---- device memory for hosts without a HIP runtime of their own ----------- */
int hj_dev_alloc(hj_ctx *ctx, uint64_t bytes, void **dptr);
int hj_dev_free(hj_ctx *ctx, void *dptr);
int hj_copy_h2d(hj_ctx *ctx, void *dst_dev, const void *src_host, uint64_t bytes);
int hj_copy_d2h(hj_ctx *ctx, void *dst_host, const void *src_dev, uint64_t bytes);

/* ---- input layer ---------------------------------------------------------- */
/* generate_data(dist, n, distinct, window) of include/DataGen.hpp:26-122 into a
 * caller-owned host buffer: same glibc rand() stream after srand(0), same
 * distributions ("uniform","random","sorted","shuffle","local_shuffle").
 * "zipf" (an empty stub in the reference, :72-77) draws keys over [1,distinct]
 * with the LUT method of mc/src/genzipf.c:60-158, theta = zipfTheta.
 * Returns HJ_ERR_INVALID for an unknown distribution (the reference exits). */
int hj_generate_data(const char *dist, uint64_t n, uint64_t distinct, int window,
                     double zipfTheta, uint64_t *out);

/* Streaming Zipf generator for probe sides too large to generate and copy in one piece (BASELINE config 5:
 * |S| = 4 * 10^9 draws over 2^28 keys). hj_zipf_open builds gen_zipf's alphabet permutation and cumulative table
 * (mc/src/genzipf.c:28-93) after srand(seed) and keeps them on the device; every hj_zipf_next_dev(n) continues the SAME
 * rand() stream where the last call stopped and writes the next n draws to dOut as 8-byte tuples: the host produces
 * only the serial rand() values, the binary search of genzipf.c:118-151 runs on the GPU. The concatenation of the
 * slices equals hj_generate_relation("zipf", total, alphabet, 0, theta, seed, ...) element for element (seed 0:
 * hj_generate_data("zipf")). Asynchronous on the context's stream apart from the host's own drawing. */
int hj_zipf_open(hj_ctx *ctx, uint64_t alphabetSize, double theta, unsigned seed);
int hj_zipf_next_dev(hj_ctx *ctx, uint64_t n, uint64_t *dOut);
int hj_zipf_close(hj_ctx *ctx);

/* The relation generators of the reference's mc/ comparison code (mc/src/generator.c), serial forms, same glibc
 * rand() stream after srand(seed) (seed_generator, :56-61; mc/src/main.c:337-338 seeds R with 12345, S with 54321),
 * written as 8-byte tuples {key, payload = 0}:
 *   "pk"          create_relation_pk (:241-261): keys 1..n, Knuth shuffle (:83-93)
 *   "pk_lshuffle" create_relation_pk_lshuffle (:263-284): keys 1..n, lshuffle(window) (:96-110)
 *   "fk"          create_relation_fk (:408-445): foreign keys into 1..maxid: n / maxid shuffled copies of 1..maxid,
 *                 then a shuffled 1..(n % maxid)  -- workload A/B of mc/src/main.c:171-215
 *   "nonunique"   create_relation_nonunique (:494-509): key = RAND_RANGE(maxid), 0 .. maxid-1. Key 0 is the table
 *                 paths' empty marker (HJ_ERR_KEY_RANGE there); HJ_ALGO_PRJ takes it
 *   "zipf"        create_relation_zipf (:521-538): gen_zipf(n, maxid, theta) of mc/src/genzipf.c
 * Returns HJ_ERR_INVALID for an unknown kind or arguments the reference would divide by zero on. */
int hj_generate_relation(const char *kind, uint64_t n, uint64_t maxid, int window, double theta,
                         unsigned seed, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* HTM_HASHJOIN_H */

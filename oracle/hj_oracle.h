/*
 * hj_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's hash-join build+probe hot path
 * (anilshanbhag/HTM-HashJoin).  It exists only so that tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() can check the HIP engine
 * against the reference's semantics.  Nothing under htm-hashjoin_amd/ may
 * include, link or call it.
 *
 * Parity status: PINNED (HTM bucket path: pinned on unique keys only, see below).
 *   - nocc/atomic path: pinned by the reference's own committed run logs
 *     (experiments/new_backup/probe_log*, AtomicsVsHTMVsNoCC_log*,
 *     experiments/overflow_log1) -- see tests/golden/reference_logs.json --
 *     and by SURVEY.md Appendix B/C values.  The reference's main.cpp path
 *     itself is NOT buildable here (it needs Intel TBB headers, which this
 *     image lacks), so no oracle/_ref binary exists for it.
 *   - HTM bucketised path (orc_htm_build_probe_seq): HTMHashBuild.hpp needs TBB and
 *     RTM and cannot be built here. Pinned by the `htm` lines of the reference's logs
 *     (probe_log*, AtomicsVsHTMVsNoCC_log*: conflictCount 0, totalMatches = rSize,
 *     inputSum = outputSum = N(N+1)/2 on unique keys). On duplicate keys the
 *     reference's own numbers are run dependent (TSX aborts reorder the inserts;
 *     experiments/overflow_log1 vs _log2 come from an older code version and differ
 *     line by line): conflictCount is order independent and restated exactly, the
 *     chain contents follow sequential input order -- PARITY UNPINNED there.
 *   - PRJ path: pinned by oracle/_ref/mchashjoins, compiled from the
 *     reference's mc/src/ sources where they lie (oracle/Makefile), and by
 *     experiments/new_backup/motivation_log1:8 (Results = 549688705024).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#ifndef HJ_ORACLE_H
#define HJ_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- DataGen (include/DataGen.hpp:26-122) -------------------------------- */

/* Fills out[0..n) exactly as generate_data(dist, n, distinct, window) does,
 * calling libc srand(0)/rand() in the same order.  dist is one of
 * "uniform","random","sorted","shuffle","local_shuffle".
 * Returns 0, or -1 for an unknown distribution (the reference exits(1)). */
int orc_generate_data(const char *dist, uint64_t n, uint64_t distinct,
                      int window, uint64_t *out);

/* Zipf probe-side generator (extension; the reference's DataGen "zipf" branch
 * is an empty stub, DataGen.hpp:72-77).  Follows mc/src/genzipf.c:60-158:
 * random alphabet permutation + cumulative LUT + binary search, driven by
 * libc rand() after srand(seed). */
int orc_generate_zipf(uint64_t n, uint32_t alphabet, double theta,
                      unsigned seed, uint64_t *out);

/* mc's relation generators (mc/src/generator.c), serial forms, driven by libc rand() after srand(seed) exactly as
 * seed_generator does (:56-61); mc seeds R with 12345 and S with 54321 (mc/src/main.c:337-338). Values are written
 * as 8-byte tuples {key, payload = 0}. kind:
 *   "pk"          create_relation_pk (:241-261): keys 1..n, knuth_shuffle (:83-93, j = RAND_RANGE(i), :20)
 *   "pk_lshuffle" create_relation_pk_lshuffle (:263-284): keys 1..n, lshuffle(window) (:96-110)
 *   "fk"          create_relation_fk (:408-445): n / maxid blocks of 1..maxid, each knuth-shuffled, then a block of
 *                 1..(n % maxid)
 *   "nonunique"   create_relation_nonunique (:494-509): key = RAND_RANGE(maxid), i.e. 0 .. maxid-1 (0 INCLUDED)
 *   "zipf"        create_relation_zipf (:521-538) = gen_zipf(n, maxid, theta)
 * Returns -1 for an unknown kind. */
int orc_generate_relation(const char *kind, uint64_t n, uint64_t maxid, int window, double theta,
                          unsigned seed, uint64_t *out);

/* ---- nocc / atomic build + probe (sequential order) ---------------------- */

typedef struct {
    uint64_t rSize, sSize, tableSize;
    uint64_t conflicts;     /* tuples that exhausted the probe budget         */
    uint64_t totalMatches;  /* probe matches (NoCCHashBuild.hpp:66-80)        */
    uint64_t inputSum;      /* sum of R                 (:85-92)              */
    uint64_t tableSumHalf;  /* sum output[0..rSize)     (:94-101, nocc quirk) */
    uint64_t tableSumFull;  /* sum output[0..tableSize) (Atomic :100-107)     */
    uint64_t conflictSum;   /* sum of dropped keys      (:103-113)            */
    uint64_t outputSumNocc;   /* tableSumHalf + conflictSum  (:145)           */
    uint64_t outputSumAtomic; /* tableSumFull + conflictSum  (intended value
                                 of AtomicHashBuild.hpp:151; the reference's
                                 own conflictSum indexes out of bounds when
                                 conflicts>0, :111-114, so only the
                                 conflicts==0 value is pinned)               */
    double   build_us, probe_us;
} orc_result;

/* Sequential-order restatement of NoCCHashBuild.hpp:37-81 (== AtomicHashBuild
 * .hpp:37-86 when run by one thread: the CAS never fails).  tableSize =
 * 2*rSize, identity hash key&(tableSize-1), linear probing with probeLength
 * budget, dropped tuples counted as conflicts.  S may be NULL (build only).
 * If table_out != NULL it receives the tableSize final slots (0 = empty).
 * The table is allocated with 4 zero slots of slack so that the reference's
 * unmasked probe walk (curSlot++ without & tableMask, :74-75) reads "empty"
 * where the reference would read out of bounds. */
int orc_build_probe_seq(const uint64_t *R, uint64_t rSize,
                        const uint64_t *S, uint64_t sSize,
                        uint32_t probeLength, orc_result *res,
                        uint64_t *table_out);

/* The same loop with the table size given explicitly (a power of two) and the home slot
 * taken as (key >> homeShift) & (tableSize-1). Used for the radix-sharded multi-GPU
 * semantics, where shard g builds a table of 2 * (total |R| / shards) slots over the
 * tuples whose low log2(shards) key bits equal g; those bits carry no information inside
 * a shard and are left out of the slot number (homeShift = log2(shards)). */
int orc_build_probe_seq_ts(const uint64_t *R, uint64_t rSize,
                           const uint64_t *S, uint64_t sSize,
                           uint32_t probeLength, uint64_t tableSize, uint32_t homeShift,
                           orc_result *res, uint64_t *table_out);

/* Threaded port used ONLY as the timed CPU baseline ("port"): numPartitions
 * contiguous chunks pulled by nthreads pthreads, exactly the chunking of
 * parallel_for(blocked_range(0,rSize,rSize/numPartitions)).  atomic=0 -> the
 * racy plain-store loop (nocc), atomic=1 -> relaxed load + CAS incl. the
 * "failed CAS costs budget without advancing" quirk (AtomicHashBuild.hpp:
 * 50-54).  On duplicate keys its counts are order dependent, as the
 * reference's are. */
int orc_build_probe_mt(const uint64_t *R, uint64_t rSize,
                       const uint64_t *S, uint64_t sSize,
                       uint32_t probeLength, uint32_t numPartitions,
                       int nthreads, int atomic, orc_result *res);
/* the same with the table first touched by the worker threads (parallelTouch != 0) instead of by the calling thread */
int orc_build_probe_mt_ex(const uint64_t *R, uint64_t rSize, const uint64_t *S, uint64_t sSize, uint32_t probeLength,
                          uint32_t numPartitions, int nthreads, int atomic, int parallelTouch, orc_result *res);

/* ---- HTM bucketised table (HTMHashBuild.hpp) ------------------------------ */

/* Bucket, HTMHashBuild.hpp:41-45: three tuples, their count, and the 1-based index of
 * the first overflow bucket of the chain (0 = none). 32 bytes. */
typedef struct {
    uint64_t tuples[3];
    uint32_t count;
    uint32_t nextIndex;
} orc_bucket;

typedef struct {
    uint64_t rSize, sSize, numBuckets;
    uint64_t conflictCount;   /* tuples that found their bucket full (:181-183, :225-228)              */
    uint64_t conflictSum;     /* sum of those tuples (:369-380)                                         */
    uint64_t overflowBuckets; /* overflow buckets allocated (curCounter - 1, :232)                      */
    uint64_t totalMatches;    /* probe over bucket + chain (:291-305, the BUILD_OVERFLOW_TABLE branch)  */
    uint64_t inputSum;        /* :312-320                                                               */
    uint64_t bucketSum;       /* sum of the tuples in the primary buckets                               */
    uint64_t overflowSum;     /* sum of the tuples in overflow buckets (== conflictSum: every conflict
                                 is chained to its own bucket)                                          */
    uint64_t outputSum;       /* bucketSum + overflowSum: the checksum the field exists for
                                 (== inputSum). NOT what :452 prints when conflicts > 0, see below      */
    uint64_t outputSumAsWritten; /* `sum + failedTransactionSum + conflictSum` evaluated over the code
                                 exactly AS WRITTEN, bugs included (see orc_htm_build_probe_seq);
                                 equals inputSum when conflictCount == 0, which is all the
                                 reference's logs pin                                                   */
} orc_htm_result;

/* Sequential-order restatement of HTMHashBuild (HTMHashBuild.hpp:54-464) with every
 * transaction committing (TSX is replaced, not emulated: an aborted group is retried
 * serially by the reference, :219-238, which on one thread is just the same inserts later):
 *   numBuckets = nextpow2(rSize/3 + 1) (:61-62); slot = (key/3) & (numBuckets-1) (:176);
 *   bucket not full -> tuples[count++] = key, else the tuple is a conflict of its input
 *   partition (:177-183), partitions = rSize/numPartitions consecutive tuples (:63);
 *   then the overflow chains (:231-279) and the probe that walks bucket + chain (:291-305).
 * Two lines of the chain builder cannot mean what they say and are restated as INTENDED
 * (the as-written behaviour is evaluated separately for outputSumAsWritten):
 *   :245 `slot = (relR[i]/3) & tableMask` hashes relR[PARTITION INDEX], not the conflict
 *        being chained (conflicts[j]): as written every conflict of partition i hangs off
 *        the bucket of the i-th tuple of R. Intended: the conflict's own bucket.
 *   :257 `curBucket = overflows[curCounter]` assigns THROUGH the reference into the full
 *        head bucket, wiping its three tuples, instead of starting a new head. Intended:
 *        a new head bucket linked to the old one.
 * buckets_out (numBuckets) / overflows_out (rSize + 1, index 0 unused) may be NULL. */
int orc_htm_build_probe_seq(const uint64_t *R, uint64_t rSize, const uint64_t *S, uint64_t sSize,
                            uint32_t numPartitions, orc_htm_result *res,
                            orc_bucket *buckets_out, orc_bucket *overflows_out);

/* ---- PRJ (mc/src/parallel_radix_join.c) ---------------------------------- */

typedef struct {
    uint64_t matches;    /* join cardinality: the probe loop the fork commented
                            out (parallel_radix_join.c:259-276), restored     */
    uint64_t checksum;   /* sum of bucket idx over all R tuples (:249-256):
                            what the fork's PRO prints as "Results"           */
    uint64_t partitions; /* non-empty R partitions joined                     */
    double   part_us, join_us;
} orc_prj_result;

/* Two-pass radix partition of R and S on the low radix_bits key bits
 * (pass 1: bits [0, radix_bits/2), pass 2: the rest -- prj_thread :814-816,
 * radix_cluster :402-440) followed by bucket_chaining_join (:231-283) per
 * partition pair.  S may be NULL (then matches = 0, as in the fork). */
int orc_prj_join(const uint64_t *R, uint64_t nR, const uint64_t *S,
                 uint64_t nS, uint32_t radix_bits, orc_prj_result *res);

/* Reference-free cross-check: exact join cardinality sum_k cntR(k)*cntS(k)
 * by sorting both sides. */
uint64_t orc_true_cardinality(const uint64_t *R, uint64_t nR,
                              const uint64_t *S, uint64_t nS);

#ifdef __cplusplus
}
#endif
#endif

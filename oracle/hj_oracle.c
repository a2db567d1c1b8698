/*
 * hj_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See hj_oracle.h for scope, pinning status and the import rules.
 *
 * Plain C (gnu99) + pthreads.  Integer arithmetic only, except the Zipf LUT.
 */
#define _GNU_SOURCE
#include "hj_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

static double now_us(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec * 1e6 + tv.tv_usec;
}

/* ------------------------------------------------------------------------ */
/* sorting: any correct sort of plain integers equals std::sort's output     */
/* (DataGen.hpp:43,60).  LSD radix sort, 16 bits per pass.                   */
/* ------------------------------------------------------------------------ */
static void sort_u64(uint64_t *a, uint64_t n)
{
    if (n < 2) return;
    uint64_t *tmp = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *cnt = (uint64_t *)malloc(65536 * sizeof(uint64_t));
    uint64_t ormask = 0;
    for (uint64_t i = 0; i < n; i++) ormask |= a[i];
    uint64_t *src = a, *dst = tmp;
    for (int shift = 0; shift < 64; shift += 16) {
        if (((ormask >> shift) & 0xFFFF) == 0) continue; /* digit all zero */
        memset(cnt, 0, 65536 * sizeof(uint64_t));
        for (uint64_t i = 0; i < n; i++) cnt[(src[i] >> shift) & 0xFFFF]++;
        uint64_t sum = 0;
        for (int d = 0; d < 65536; d++) { uint64_t c = cnt[d]; cnt[d] = sum; sum += c; }
        for (uint64_t i = 0; i < n; i++) dst[cnt[(src[i] >> shift) & 0xFFFF]++] = src[i];
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(uint64_t));
    free(cnt);
    free(tmp);
}

/* DataGen.hpp:44-54 (also :61-71, :97/107-115): the window shuffle. */
static void window_shuffle(uint64_t *input, uint64_t n, int window)
{
    unsigned char *shuffled = (unsigned char *)calloc(n ? n : 1, 1);
    for (uint64_t i = 0; i + 1 < n; i++) {
        if (!shuffled[i]) {
            /* rand() % min(local_shuffle_range, (int)(size_in_tuples - i)) */
            int rem = (int)(n - i);
            int m = window < rem ? window : rem;
            int swap = rand() % m;
            uint64_t temp = input[i];
            input[i] = input[i + swap];
            input[i + swap] = temp;
            shuffled[i + swap] = 1;
        }
    }
    free(shuffled);
}

int orc_generate_data(const char *dist, uint64_t n, uint64_t distinct,
                      int window, uint64_t *out)
{
    srand(0);                                   /* DataGen.hpp:27 */
    uint32_t mod_mask = (uint32_t)(distinct - 1); /* :28 */
    if (strcmp(dist, "uniform") == 0) {         /* :30-54 */
        for (uint64_t i = 0; i < n; i++) out[i] = ((uint32_t)rand() & mod_mask) + 1;
        sort_u64(out, n);
        window_shuffle(out, n, window);
    } else if (strcmp(dist, "random") == 0) {   /* :55-71 */
        for (uint64_t i = 0; i < n; i++) {
            out[i] = (uint64_t)rand();
            while (out[i] == 0) out[i] = (uint64_t)rand();
        }
        sort_u64(out, n);
        window_shuffle(out, n, window);
    } else if (strcmp(dist, "sorted") == 0) {   /* :78-85 */
        for (uint64_t i = 0; i < n; i++) out[i] = i + 1;
    } else if (strcmp(dist, "shuffle") == 0) {  /* :86-95 */
        for (uint64_t i = 0; i < n; i++) out[i] = i + 1;
        /* std::random_shuffle(first,last) of libstdc++ (bits/stl_algo.h):
         * for i in [1,n): j = rand() % (i+1); if (i != j) swap(a[i], a[j]) */
        for (uint64_t i = 1; i < n; i++) {
            uint64_t j = (uint64_t)rand() % (i + 1);
            if (i != j) { uint64_t t = out[i]; out[i] = out[j]; out[j] = t; }
        }
    } else if (strcmp(dist, "local_shuffle") == 0) { /* :96-115 */
        for (uint64_t i = 0; i < n; i++) out[i] = i + 1;
        window_shuffle(out, n, window);
    } else {
        return -1;                              /* :116-119 exits(1) */
    }
    return 0;
}

int orc_generate_zipf(uint64_t n, uint32_t alphabet_size, double theta,
                      unsigned seed, uint64_t *out)
{
    if (alphabet_size == 0) return -1;
    srand(seed);
    /* gen_alphabet, genzipf.c:28-53 */
    uint32_t *alphabet = (uint32_t *)malloc((size_t)alphabet_size * sizeof(uint32_t));
    for (uint32_t i = 0; i < alphabet_size; i++) alphabet[i] = i + 1;
    for (uint32_t i = alphabet_size - 1; i > 0; i--) {
        unsigned int k = (unsigned int)((unsigned long)i * rand() / RAND_MAX);
        uint32_t tmp = alphabet[i];
        alphabet[i] = alphabet[k];
        alphabet[k] = tmp;
    }
    /* gen_zipf_lut, genzipf.c:60-93 */
    double *lut = (double *)malloc((size_t)alphabet_size * sizeof(double));
    double scaling = 0.0;
    for (uint32_t i = 1; i <= alphabet_size; i++) scaling += 1.0 / pow((double)i, theta);
    double sum = 0.0;
    for (uint32_t i = 1; i <= alphabet_size; i++) {
        sum += 1.0 / pow((double)i, theta);
        lut[i - 1] = sum / scaling;
    }
    /* gen_zipf, genzipf.c:118-151 */
    for (uint64_t i = 0; i < n; i++) {
        double r = ((double)rand()) / RAND_MAX;
        unsigned int left = 0, right = alphabet_size - 1, m, pos;
        if (lut[0] >= r) {
            pos = 0;
        } else {
            while (right - left > 1) {
                m = (left + right) / 2;
                if (lut[m] < r) left = m; else right = m;
            }
            pos = right;
        }
        out[i] = alphabet[pos];
    }
    free(lut);
    free(alphabet);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* nocc / atomic, sequential order                                           */
/* ------------------------------------------------------------------------ */
#define ORC_SLACK 4

int orc_build_probe_seq(const uint64_t *R, uint64_t rSize,
                        const uint64_t *S, uint64_t sSize,
                        uint32_t probeLength, orc_result *res,
                        uint64_t *table_out)
{
    /* NoCCHashBuild.hpp:20: tableSize = rSize*2 */
    return orc_build_probe_seq_ts(R, rSize, S, sSize, probeLength, rSize * 2, 0, res, table_out);
}

int orc_build_probe_seq_ts(const uint64_t *R, uint64_t rSize,
                           const uint64_t *S, uint64_t sSize,
                           uint32_t probeLength, uint64_t tableSize, uint32_t homeShift,
                           orc_result *res, uint64_t *table_out)
{
    memset(res, 0, sizeof(*res));
    if (tableSize == 0 || (tableSize & (tableSize - 1))) return -1;
    uint64_t tableMask = tableSize - 1;         /* :36 */
    uint64_t *output = (uint64_t *)calloc(tableSize + ORC_SLACK, sizeof(uint64_t));
    if (!output) return -1;
    uint64_t conflicts = 0, conflictSum = 0;

    double t0 = now_us();
    for (uint64_t i = 0; i < rSize; i++) {      /* :43-59 */
        uint64_t curSlot = (R[i] >> homeShift) & tableMask;
        uint32_t probeBudget = probeLength;
        while (probeBudget != 0) {
            if (output[curSlot] == 0) {
                output[curSlot] = R[i];
                break;
            } else {
                curSlot += 1;
                curSlot &= tableMask;
                probeBudget--;
            }
        }
        if (probeBudget == 0) { conflicts++; conflictSum += R[i]; } /* :57-58 */
    }
    double t1 = now_us();

    uint64_t matches = 0;
    if (S) {
        for (uint64_t i = 0; i < sSize; i++) {  /* :70-79 */
            uint64_t curSlot = (S[i] >> homeShift) & tableMask;
            uint32_t probeBudget = probeLength;
            while (probeBudget-- && output[curSlot] != 0) {
                if (output[curSlot] == S[i]) { matches++; curSlot++; }
                else if (output[curSlot] != 0) curSlot++;
                else break;
            }
        }
    }
    double t2 = now_us();

    uint64_t inputSum = 0, half = 0, full = 0;
    for (uint64_t i = 0; i < rSize; i++) inputSum += R[i];     /* :85-92  */
    for (uint64_t i = 0; i < tableSize / 2; i++) half += output[i]; /* :94-101 (tableSize/2 == rSize there) */
    for (uint64_t i = 0; i < tableSize; i++) full += output[i];

    res->rSize = rSize; res->sSize = S ? sSize : 0; res->tableSize = tableSize;
    res->conflicts = conflicts; res->totalMatches = matches;
    res->inputSum = inputSum; res->tableSumHalf = half; res->tableSumFull = full;
    res->conflictSum = conflictSum;
    res->outputSumNocc = half + conflictSum;
    res->outputSumAtomic = full + conflictSum;
    res->build_us = t1 - t0; res->probe_us = t2 - t1;
    if (table_out) memcpy(table_out, output, tableSize * sizeof(uint64_t));
    free(output);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* threaded port (timed CPU baseline only)                                   */
/* ------------------------------------------------------------------------ */
typedef struct {
    const uint64_t *R, *S;
    uint64_t rSize, sSize, tableMask;
    uint64_t *output;
    uint32_t probeLength, numPartitions;
    int atomic;
    volatile uint32_t next_chunk;
    uint64_t *chunkConflicts, *chunkConflictSum, *chunkMatches;
    int phase; /* 0 build, 1 probe */
} mt_shared;

static void mt_build_chunk(mt_shared *sh, uint32_t c)
{
    uint64_t psize = sh->rSize / sh->numPartitions;
    uint64_t b = (uint64_t)c * psize;
    uint64_t e = (c + 1 == sh->numPartitions) ? sh->rSize : b + psize;
    uint64_t *output = sh->output;
    uint64_t mask = sh->tableMask, nconf = 0, csum = 0;
    for (uint64_t i = b; i < e; i++) {
        uint64_t cur = sh->R[i] & mask;
        uint32_t budget = sh->probeLength;
        if (!sh->atomic) {                      /* NoCCHashBuild.hpp:43-56 */
            while (budget != 0) {
                if (output[cur] == 0) { output[cur] = sh->R[i]; break; }
                cur = (cur + 1) & mask; budget--;
            }
        } else {                                /* AtomicHashBuild.hpp:44-60 */
            while (budget != 0) {
                uint64_t prev = __atomic_load_n(&output[cur], __ATOMIC_RELAXED);
                if (prev == 0) {
                    uint64_t zero = 0;
                    if (__atomic_compare_exchange_n(&output[cur], &zero, sh->R[i], 0,
                                                    __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST))
                        break;
                    budget--;                   /* :54 quirk: no advance */
                } else {
                    cur = (cur + 1) & mask; budget--;
                }
            }
        }
        if (budget == 0) { nconf++; csum += sh->R[i]; }
    }
    sh->chunkConflicts[c] = nconf;
    sh->chunkConflictSum[c] = csum;
}

static void mt_probe_chunk(mt_shared *sh, uint32_t c)
{
    uint64_t psize = sh->sSize / sh->numPartitions;
    uint64_t b = (uint64_t)c * psize;
    uint64_t e = (c + 1 == sh->numPartitions) ? sh->sSize : b + psize;
    const uint64_t *output = sh->output;
    uint64_t mask = sh->tableMask, matches = 0;
    for (uint64_t i = b; i < e; i++) {          /* NoCCHashBuild.hpp:70-79 */
        uint64_t cur = sh->S[i] & mask;
        uint32_t budget = sh->probeLength;
        while (budget-- && output[cur] != 0) {
            if (output[cur] == sh->S[i]) { matches++; cur++; }
            else if (output[cur] != 0) cur++;
            else break;
        }
    }
    sh->chunkMatches[c] = matches;
}

static void *mt_worker(void *p)
{
    mt_shared *sh = (mt_shared *)p;
    for (;;) {
        uint32_t c = __atomic_fetch_add(&sh->next_chunk, 1, __ATOMIC_RELAXED);
        if (c >= sh->numPartitions) break;
        if (sh->phase == 0) mt_build_chunk(sh, c); else mt_probe_chunk(sh, c);
    }
    return NULL;
}

static void mt_run(mt_shared *sh, int nthreads)
{
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    sh->next_chunk = 0;
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, mt_worker, sh);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
}

/* first touch of the table by the worker threads themselves, piece t, t + T, ... of 4096 by thread t: on a multi-socket
 * host the table's pages then spread over the memory of every socket the threads run on, instead of all sitting on the
 * one node of the thread that called calloc (measured on the GPU box's host: profiles/r03_summary.md, "CPU baseline") */
typedef struct { uint64_t *p; uint64_t n; int t, T; } touch_arg;
static void *touch_worker(void *a_)
{
    touch_arg *a = (touch_arg *)a_;
    const uint64_t pieces = 4096, len = (a->n + pieces - 1) / pieces;
    for (uint64_t k = (uint64_t)a->t; k < pieces; k += (uint64_t)a->T) {
        volatile uint64_t *q = a->p;
        const uint64_t b = k * len, e = b + len < a->n ? b + len : a->n;
        for (uint64_t i = b; i < e; i += 512) q[i] = 0;
    }
    return NULL;
}

int orc_build_probe_mt_ex(const uint64_t *R, uint64_t rSize, const uint64_t *S, uint64_t sSize, uint32_t probeLength,
                          uint32_t numPartitions, int nthreads, int atomic, int parallelTouch, orc_result *res);

int orc_build_probe_mt(const uint64_t *R, uint64_t rSize,
                       const uint64_t *S, uint64_t sSize,
                       uint32_t probeLength, uint32_t numPartitions,
                       int nthreads, int atomic, orc_result *res)
{
    return orc_build_probe_mt_ex(R, rSize, S, sSize, probeLength, numPartitions, nthreads, atomic, 0, res);
}

int orc_build_probe_mt_ex(const uint64_t *R, uint64_t rSize,
                          const uint64_t *S, uint64_t sSize,
                          uint32_t probeLength, uint32_t numPartitions,
                          int nthreads, int atomic, int parallelTouch, orc_result *res)
{
    memset(res, 0, sizeof(*res));
    if (numPartitions == 0 || nthreads <= 0) return -1;
    mt_shared sh;
    memset(&sh, 0, sizeof(sh));
    uint64_t tableSize = rSize * 2;
    sh.R = R; sh.S = S; sh.rSize = rSize; sh.sSize = sSize;
    sh.tableMask = tableSize - 1;
    sh.probeLength = probeLength; sh.numPartitions = numPartitions;
    sh.atomic = atomic;
    sh.output = (uint64_t *)calloc(tableSize + ORC_SLACK, sizeof(uint64_t));
    sh.chunkConflicts = (uint64_t *)calloc(numPartitions, sizeof(uint64_t));
    sh.chunkConflictSum = (uint64_t *)calloc(numPartitions, sizeof(uint64_t));
    sh.chunkMatches = (uint64_t *)calloc(numPartitions, sizeof(uint64_t));
    if (!sh.output) return -1;
    /* touch every page of the table so that page faults stay outside the timed region, as
     * new uint64_t[tableSize]{} does in the reference (:24). A plain memset after calloc is
     * elided by the compiler (calloc memory is known to be zero), which left the faults --
     * and the kernel's serialisation of them -- inside the timed build. */
    if (parallelTouch) {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        touch_arg *ta = (touch_arg *)malloc(sizeof(touch_arg) * (size_t)nthreads);
        for (int t = 0; t < nthreads; t++) {
            ta[t].p = sh.output; ta[t].n = tableSize + ORC_SLACK; ta[t].t = t; ta[t].T = nthreads;
            pthread_create(&th[t], NULL, touch_worker, &ta[t]);
        }
        for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
        free(th); free(ta);
    }
    {
        volatile uint64_t *touch = sh.output;
        for (uint64_t i = 0; i < tableSize + ORC_SLACK; i += 512) touch[i] = 0;
        touch[tableSize + ORC_SLACK - 1] = 0;
    }

    double t0 = now_us();
    sh.phase = 0; mt_run(&sh, nthreads);
    double t1 = now_us();
    if (S) { sh.phase = 1; mt_run(&sh, nthreads); }
    double t2 = now_us();

    uint64_t inputSum = 0, half = 0, full = 0;
    for (uint64_t i = 0; i < rSize; i++) inputSum += R[i];
    for (uint64_t i = 0; i < rSize; i++) half += sh.output[i];
    for (uint64_t i = 0; i < tableSize; i++) full += sh.output[i];
    for (uint32_t c = 0; c < numPartitions; c++) {
        res->conflicts += sh.chunkConflicts[c];
        res->conflictSum += sh.chunkConflictSum[c];
        res->totalMatches += sh.chunkMatches[c];
    }
    res->rSize = rSize; res->sSize = S ? sSize : 0; res->tableSize = tableSize;
    res->inputSum = inputSum; res->tableSumHalf = half; res->tableSumFull = full;
    res->outputSumNocc = half + res->conflictSum;
    res->outputSumAtomic = full + res->conflictSum;
    res->build_us = t1 - t0; res->probe_us = t2 - t1;
    free(sh.output); free(sh.chunkConflicts); free(sh.chunkConflictSum); free(sh.chunkMatches);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* mc/src/generator.c                                                          */
/* ------------------------------------------------------------------------ */
#define ORC_RAND_RANGE(N) ((double)rand() / ((double)RAND_MAX + 1) * (N))     /* generator.c:20 */

static void orc_knuth_shuffle(uint64_t *t, uint64_t n)                       /* :83-93 */
{
    for (int64_t i = (int64_t)n - 1; i > 0; i--) {
        int32_t j = (int32_t)ORC_RAND_RANGE(i);
        uint64_t tmp = t[i]; t[i] = t[j]; t[j] = tmp;
    }
}

static void orc_random_unique_gen(uint64_t *t, uint64_t n)                   /* :125-136 */
{
    for (uint64_t i = 0; i < n; i++) t[i] = i + 1;
    orc_knuth_shuffle(t, n);
}

int orc_generate_relation(const char *kind, uint64_t n, uint64_t maxid, int window, double theta,
                          unsigned seed, uint64_t *out)
{
    if (strcmp(kind, "zipf") == 0) return orc_generate_zipf(n, (uint32_t)maxid, theta, seed, out);   /* :521-538 */
    srand(seed);                                                             /* seed_generator, :56-61 */
    if (strcmp(kind, "pk") == 0) {
        orc_random_unique_gen(out, n);
    } else if (strcmp(kind, "pk_lshuffle") == 0) {                           /* :138-151, lshuffle :96-110 */
        if (window <= 0) return -1;
        for (uint64_t i = 0; i < n; i++) out[i] = i + 1;
        for (uint64_t i = 0; i < n; i++) {
            int32_t runway = (int32_t)(n - i);
            int32_t mod = runway > window ? window : runway;
            int32_t swap = rand() % mod;
            uint64_t j = i + (uint64_t)swap;
            uint64_t tmp = out[i]; out[i] = out[j]; out[j] = tmp;
        }
    } else if (strcmp(kind, "fk") == 0) {                                    /* :408-445 */
        if (maxid == 0) return -1;
        uint64_t iters = n / maxid, rem = n % maxid;
        for (uint64_t i = 0; i < iters; i++) orc_random_unique_gen(out + maxid * i, maxid);
        if (rem > 0) orc_random_unique_gen(out + maxid * iters, rem);
    } else if (strcmp(kind, "nonunique") == 0) {                             /* :494-509, random_gen :230-238 */
        for (uint64_t i = 0; i < n; i++) out[i] = (uint64_t)(int32_t)ORC_RAND_RANGE((int32_t)maxid);
    } else {
        return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* HTM bucketised table, sequential order (HTMHashBuild.hpp)                  */
/* ------------------------------------------------------------------------ */
static uint32_t next_pow2_u32(uint32_t v)
{   /* NEXT_POW_2, HTMHashBuild.hpp:29-38 */
    v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v++;
    return v;
}

/* the chain builder and the checksum walk exactly as written (:231-279, :322-342), for outputSumAsWritten.
 * Works on private copies: it mutates buckets through its `Bucket&` assignments. */
static uint64_t htm_output_sum_as_written(const uint64_t *R, uint64_t rSize, const orc_bucket *b0, uint32_t numBuckets,
                                          const uint64_t *conflicts, const uint32_t *conflictCounts,
                                          uint32_t numPartitions, uint32_t inputPartitionSize, uint64_t conflictCount,
                                          uint64_t conflictSum)
{
    uint32_t tableMask = numBuckets - 1;
    orc_bucket *buckets = (orc_bucket *)malloc((size_t)numBuckets * sizeof(orc_bucket));
    orc_bucket *overflows = (orc_bucket *)calloc((size_t)conflictCount + 2, sizeof(orc_bucket));
    if (!buckets || !overflows) { free(buckets); free(overflows); return 0; }
    memcpy(buckets, b0, (size_t)numBuckets * sizeof(orc_bucket));
    int curCounter = 1;
    for (uint32_t i = 0; i < numPartitions; i++) {                         /* :234 */
        uint32_t conflictPartitionStart = inputPartitionSize * i;
        for (uint32_t j = conflictPartitionStart; j < conflictPartitionStart + conflictCounts[i]; j++) {
            uint32_t slot = (uint32_t)(R[i < rSize ? i : 0] / 3) & tableMask;   /* :237 relR[i], i = partition index */
            if (buckets[slot].count == 3) {
                int nextIndex = (int)buckets[slot].nextIndex;
                if (nextIndex == 0) {
                    buckets[slot].nextIndex = (uint32_t)curCounter;
                    overflows[curCounter].count = 1;
                    overflows[curCounter].tuples[0] = conflicts[j];
                    curCounter += 1;
                } else {
                    orc_bucket *curBucket = &overflows[nextIndex];          /* Bucket& curBucket */
                    if (curBucket->count == 3) {
                        *curBucket = overflows[curCounter];                 /* :249 assigns through the reference */
                        curBucket->nextIndex = buckets[slot].nextIndex;
                        buckets[slot].nextIndex = (uint32_t)curCounter;
                        curCounter += 1;
                        curBucket->count = 1;
                        curBucket->tuples[0] = conflicts[j];
                    } else {
                        curBucket->tuples[curBucket->count] = conflicts[j];
                        curBucket->count += 1;
                    }
                }
            } else {
                buckets[slot].tuples[buckets[slot].count] = conflicts[j];
                buckets[slot].count++;
            }
        }
    }
    /* :322-342: `Bucket& curBucket = buckets[i]; ... curBucket = overflows[curBucket.nextIndex]` copies the overflow
     * bucket over buckets[i] while walking; a chain that reaches itself would not end, so the walk is capped */
    uint64_t sum = 0;
    for (uint32_t i = 0; i < numBuckets; i++) {
        orc_bucket *curBucket = &buckets[i];
        for (uint64_t guard = 0; guard <= conflictCount + 1; guard++) {
            for (uint32_t j = 0; j < curBucket->count; j++) sum += curBucket->tuples[j];
            if (curBucket->nextIndex == 0) break;
            *curBucket = overflows[curBucket->nextIndex];
        }
    }
    free(buckets); free(overflows);
    return sum + 0 /* failedTransactionSum with TM_RETRY, :366 */ + conflictSum;   /* :452 */
}

int orc_htm_build_probe_seq(const uint64_t *R, uint64_t rSize, const uint64_t *S, uint64_t sSize,
                            uint32_t numPartitions, orc_htm_result *res,
                            orc_bucket *buckets_out, orc_bucket *overflows_out)
{
    memset(res, 0, sizeof(*res));
    if (rSize == 0 || rSize > 0xFFFFFFFFull || numPartitions == 0) return -1;
    uint32_t numBuckets = (uint32_t)(rSize / 3 + 1);                     /* :61 */
    numBuckets = next_pow2_u32(numBuckets);                              /* :62 */
    uint32_t tableMask = numBuckets - 1;                                 /* :96 */
    uint32_t inputPartitionSize = (uint32_t)(rSize / numPartitions);     /* :63 */
    if (inputPartitionSize == 0) { inputPartitionSize = (uint32_t)rSize; numPartitions = 1; }
    orc_bucket *buckets = (orc_bucket *)calloc(numBuckets, sizeof(orc_bucket));        /* :65-72 */
    uint64_t *conflicts = (uint64_t *)calloc(rSize, sizeof(uint64_t));                 /* :79 */
    uint32_t *conflictCounts = (uint32_t *)calloc(numPartitions + 1, sizeof(uint32_t)); /* :80 */
    if (!buckets || !conflicts || !conflictCounts) { free(buckets); free(conflicts); free(conflictCounts); return -1; }

    /* :157-215 with every transaction committing; blocked_range(0, rSize, inputPartitionSize) cuts R into chunks of
     * inputPartitionSize tuples (a last shorter one if rSize is not a multiple); chunk p's conflicts are stored from
     * conflicts[inputPartitionSize * p] */
    uint64_t conflictCount = 0, conflictSum = 0;
    for (uint64_t begin = 0, p = 0; begin < rSize; begin += inputPartitionSize, p++) {
        uint64_t end = begin + inputPartitionSize < rSize ? begin + inputPartitionSize : rSize;
        uint32_t pid = p < numPartitions ? (uint32_t)p : numPartitions - 1;   /* a ragged tail joins the last partition's list */
        for (uint64_t i = begin; i < end; i++) {
            uint32_t slot = (uint32_t)(R[i] / 3) & tableMask;             /* :176 */
            if (buckets[slot].count != 3) {
                buckets[slot].tuples[buckets[slot].count++] = R[i];       /* :178 */
            } else {
                conflicts[(uint64_t)inputPartitionSize * pid + conflictCounts[pid]++] = R[i];   /* :180 */
                conflictCount++; conflictSum += R[i];
            }
        }
    }
    uint64_t asWritten = htm_output_sum_as_written(R, rSize, buckets, numBuckets, conflicts, conflictCounts, numPartitions,
                                                   inputPartitionSize, conflictCount, conflictSum);

    /* :231-279 as intended: each conflict is chained to ITS OWN bucket; a full head gets a new head in front of it */
    orc_bucket *overflows = (orc_bucket *)calloc((size_t)conflictCount + 2, sizeof(orc_bucket));
    if (!overflows) { free(buckets); free(conflicts); free(conflictCounts); return -1; }
    uint32_t curCounter = 1;
    for (uint32_t i = 0; i < numPartitions; i++) {
        uint64_t start = (uint64_t)inputPartitionSize * i;
        for (uint64_t j = start; j < start + conflictCounts[i]; j++) {
            uint32_t slot = (uint32_t)(conflicts[j] / 3) & tableMask;
            if (buckets[slot].count == 3) {
                uint32_t nextIndex = buckets[slot].nextIndex;
                if (nextIndex == 0 || overflows[nextIndex].count == 3) {
                    overflows[curCounter].nextIndex = nextIndex;          /* new head, linked to the old one */
                    overflows[curCounter].count = 1;
                    overflows[curCounter].tuples[0] = conflicts[j];
                    buckets[slot].nextIndex = curCounter;
                    curCounter += 1;
                } else {
                    overflows[nextIndex].tuples[overflows[nextIndex].count++] = conflicts[j];
                }
            } else {
                buckets[slot].tuples[buckets[slot].count++] = conflicts[j];
            }
        }
    }

    /* probe, :291-305 (the BUILD_OVERFLOW_TABLE branch): bucket, then its chain */
    uint64_t matches = 0;
    if (S) {
        for (uint64_t i = 0; i < sSize; i++) {
            const orc_bucket *cur = &buckets[(uint32_t)(S[i] / 3) & tableMask];
            for (;;) {
                for (uint32_t j = 0; j < cur->count; j++) if (cur->tuples[j] == S[i]) matches++;
                if (cur->nextIndex == 0) break;
                cur = &overflows[cur->nextIndex];
            }
        }
    }
    uint64_t inputSum = 0, bucketSum = 0, overflowSum = 0;
    for (uint64_t i = 0; i < rSize; i++) inputSum += R[i];               /* :312-320 */
    for (uint32_t i = 0; i < numBuckets; i++)
        for (uint32_t j = 0; j < buckets[i].count; j++) bucketSum += buckets[i].tuples[j];
    for (uint32_t i = 1; i < curCounter; i++)
        for (uint32_t j = 0; j < overflows[i].count; j++) overflowSum += overflows[i].tuples[j];

    res->rSize = rSize; res->sSize = S ? sSize : 0; res->numBuckets = numBuckets;
    res->conflictCount = conflictCount; res->conflictSum = conflictSum; res->overflowBuckets = curCounter - 1;
    res->totalMatches = matches; res->inputSum = inputSum; res->bucketSum = bucketSum; res->overflowSum = overflowSum;
    res->outputSum = bucketSum + overflowSum; res->outputSumAsWritten = asWritten;
    if (buckets_out) memcpy(buckets_out, buckets, (size_t)numBuckets * sizeof(orc_bucket));
    if (overflows_out) memcpy(overflows_out, overflows, (size_t)(conflictCount + 1) * sizeof(orc_bucket));
    free(buckets); free(conflicts); free(conflictCounts); free(overflows);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* PRJ                                                                       */
/* ------------------------------------------------------------------------ */
#define HASH_BIT_MODULO(K, MASK, NBITS) (((K) & (MASK)) >> (NBITS)) /* :59 */

static uint32_t next_pow_2(uint32_t v)          /* :66-76 NEXT_POW_2 */
{
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    v++;
    return v;
}

/* radix_cluster (:402-440) without the SMALL_PADDING_TUPLES gaps: the gaps
 * only move partitions apart in memory, they carry no tuples. hist[fanOut+1]
 * receives the exclusive prefix (partition starts). */
static void radix_cluster(uint64_t *out, const uint64_t *in, uint64_t n,
                          uint64_t *hist, int R, int D)
{
    uint32_t fanOut = 1u << D;
    uint64_t M = (uint64_t)(fanOut - 1) << R;
    uint64_t *dst = (uint64_t *)calloc(fanOut + 1, sizeof(uint64_t));
    memset(hist, 0, (fanOut + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) hist[HASH_BIT_MODULO((uint32_t)in[i], M, R)]++;
    uint64_t offset = 0;
    for (uint32_t i = 0; i < fanOut; i++) { dst[i] = offset; offset += hist[i]; }
    dst[fanOut] = offset;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t idx = (uint32_t)HASH_BIT_MODULO((uint32_t)in[i], M, R);
        out[dst[idx]++] = in[i];
    }
    /* turn counts into starts */
    offset = 0;
    for (uint32_t i = 0; i < fanOut; i++) { uint64_t c = hist[i]; hist[i] = offset; offset += c; }
    hist[fanOut] = offset;
    free(dst);
}

/* bucket_chaining_join (:231-283) with the probe loop (:259-276) restored.
 * Only the 32-bit key half of the tuple takes part (tuple_t.key, types.h:34). */
static void bucket_chaining_join(const uint64_t *Rt, uint32_t numR,
                                 const uint64_t *St, uint32_t numS,
                                 uint32_t radix_bits,
                                 uint64_t *matches, uint64_t *checksum)
{
    uint32_t N = next_pow_2(numR);
    const uint32_t MASK = (N - 1) << radix_bits;
    int *next = (int *)malloc(sizeof(int) * (numR ? numR : 1));
    int *bucket = (int *)calloc(N ? N : 1, sizeof(int));
    uint64_t cs = 0, m = 0;
    for (uint32_t i = 0; i < numR;) {
        uint32_t idx = HASH_BIT_MODULO((uint32_t)Rt[i], MASK, radix_bits);
        next[i] = bucket[idx];
        bucket[idx] = ++i;
        cs += idx;                              /* :256 */
    }
    for (uint32_t i = 0; i < numS; i++) {
        uint32_t idx = HASH_BIT_MODULO((uint32_t)St[i], MASK, radix_bits);
        for (int hit = bucket[idx]; hit > 0; hit = next[hit - 1])
            if ((uint32_t)St[i] == (uint32_t)Rt[hit - 1]) m++;
    }
    free(bucket); free(next);
    *matches += m; *checksum += cs;
}

/* two-pass partition of one relation; returns malloc'ed partitioned copy and
 * fills starts[2^radix_bits + 1] in (pass-1 bin major, pass-2 bin minor) order,
 * i.e. final partition id = (low bits << D2) | next bits. */
static uint64_t *two_pass_partition(const uint64_t *in, uint64_t n,
                                    uint32_t radix_bits, uint64_t *starts)
{
    int D1 = (int)(radix_bits / 2);             /* :814  NUM_RADIX_BITS/NUM_PASSES */
    int D2 = (int)radix_bits - D1;              /* :816 */
    uint32_t F1 = 1u << D1, F2 = 1u << D2;
    uint64_t *tmp = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint64_t *out = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    uint64_t *h1 = (uint64_t *)malloc((F1 + 1) * sizeof(uint64_t));
    uint64_t *h2 = (uint64_t *)malloc((F2 + 1) * sizeof(uint64_t));
    radix_cluster(tmp, in, n, h1, 0, D1);       /* pass 1: R=0 (:872) */
    for (uint32_t p = 0; p < F1; p++) {         /* pass 2: R=D1, D=D2 (:948) */
        uint64_t b = h1[p], len = h1[p + 1] - h1[p];
        radix_cluster(out + b, tmp + b, len, h2, D1, D2);
        for (uint32_t q = 0; q < F2; q++) starts[(uint64_t)p * F2 + q] = b + h2[q];
    }
    starts[(uint64_t)F1 * F2] = n;
    free(h1); free(h2); free(tmp);
    return out;
}

int orc_prj_join(const uint64_t *R, uint64_t nR, const uint64_t *S,
                 uint64_t nS, uint32_t radix_bits, orc_prj_result *res)
{
    memset(res, 0, sizeof(*res));
    if (radix_bits < 2 || radix_bits > 24) return -1;
    uint64_t P = 1ull << radix_bits;
    uint64_t *sr = (uint64_t *)malloc((P + 1) * sizeof(uint64_t));
    uint64_t *ss = (uint64_t *)malloc((P + 1) * sizeof(uint64_t));
    double t0 = now_us();
    uint64_t *pr = two_pass_partition(R, nR, radix_bits, sr);
    uint64_t *ps = NULL;
    if (S) ps = two_pass_partition(S, nS, radix_bits, ss);
    double t1 = now_us();
    for (uint64_t p = 0; p < P; p++) {
        uint64_t lr = sr[p + 1] - sr[p];
        if (lr == 0) continue;                  /* :531 only non-empty R parts */
        uint64_t ls = S ? ss[p + 1] - ss[p] : 0;
        bucket_chaining_join(pr + sr[p], (uint32_t)lr, S ? ps + ss[p] : NULL,
                             (uint32_t)ls, radix_bits, &res->matches, &res->checksum);
        res->partitions++;
    }
    double t2 = now_us();
    res->part_us = t1 - t0; res->join_us = t2 - t1;
    free(pr); free(ps); free(sr); free(ss);
    return 0;
}

uint64_t orc_true_cardinality(const uint64_t *R, uint64_t nR,
                              const uint64_t *S, uint64_t nS)
{
    uint64_t *a = (uint64_t *)malloc((nR ? nR : 1) * sizeof(uint64_t));
    uint64_t *b = (uint64_t *)malloc((nS ? nS : 1) * sizeof(uint64_t));
    memcpy(a, R, nR * sizeof(uint64_t));
    memcpy(b, S, nS * sizeof(uint64_t));
    sort_u64(a, nR); sort_u64(b, nS);
    uint64_t i = 0, j = 0, total = 0;
    while (i < nR && j < nS) {
        if (a[i] < b[j]) i++;
        else if (a[i] > b[j]) j++;
        else {
            uint64_t k = a[i], ci = 0, cj = 0;
            while (i < nR && a[i] == k) { i++; ci++; }
            while (j < nS && b[j] == k) { j++; cj++; }
            total += ci * cj;
        }
    }
    free(a); free(b);
    return total;
}

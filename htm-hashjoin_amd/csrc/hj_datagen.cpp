// hj_datagen.cpp -- the input layer: hj_generate_data(), a from-scratch equivalent
// of generate_data() in the reference's include/DataGen.hpp:26-122.
//
// The reference draws from libc rand() after srand(0), so its inputs (and the
// inputSum values in its logs) are a function of glibc's generator: hj_rand.h
// restates that generator (GlibcRand); tests/test_datagen.py checks it against
// libc rand() itself.

#include "../../include/htm_hashjoin.h"
#include "hj_rand.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

using hjhost::GlibcRand;
using hjhost::kRandMax;

// Parallel LSD radix sort of values < 2^32 (11 bits x 3 passes). Any correct sort
// gives std::sort's result on plain integers (DataGen.hpp:43,60).
void sort_keys(uint64_t* a, uint64_t n)
{
    if (n < 2) return;
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    if (n < (1u << 16)) nt = 1;
    std::vector<uint64_t> tmp(n);
    uint64_t* src = a;
    uint64_t* dst = tmp.data();
    constexpr int kBits = 11, kFan = 1 << kBits;
    uint64_t ormask = 0;
    for (uint64_t i = 0; i < n; ++i) ormask |= a[i];
    if (ormask >> 33) { std::sort(a, a + n); return; }  // not DataGen-shaped input
    std::vector<uint64_t> hist((size_t)nt * kFan);
    for (int shift = 0; shift < 33; shift += kBits) {
        if (((ormask >> shift) & (kFan - 1)) == 0) continue;
        std::fill(hist.begin(), hist.end(), 0);
        auto range = [&](unsigned t, uint64_t& b, uint64_t& e) { b = n * t / nt; e = n * (t + 1) / nt; };
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                uint64_t b, e; range(t, b, e);
                uint64_t* h = &hist[(size_t)t * kFan];
                for (uint64_t i = b; i < e; ++i) h[(src[i] >> shift) & (kFan - 1)]++;
            });
        for (auto& x : th) x.join();
        th.clear();
        uint64_t sum = 0;  // digit-major, thread-minor prefix keeps the sort stable
        for (int d = 0; d < kFan; ++d)
            for (unsigned t = 0; t < nt; ++t) { uint64_t c = hist[(size_t)t * kFan + d]; hist[(size_t)t * kFan + d] = sum; sum += c; }
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                uint64_t b, e; range(t, b, e);
                uint64_t* h = &hist[(size_t)t * kFan];
                for (uint64_t i = b; i < e; ++i) dst[h[(src[i] >> shift) & (kFan - 1)]++] = src[i];
            });
        for (auto& x : th) x.join();
        std::swap(src, dst);
    }
    if (src != a) memcpy(a, src, n * sizeof(uint64_t));
}

// DataGen.hpp:44-54 / :61-71 / :97,107-115
void window_shuffle(GlibcRand& rng, uint64_t* input, uint64_t n, int window)
{
    std::vector<unsigned char> shuffled(n ? n : 1, 0);
    for (uint64_t i = 0; i + 1 < n; ++i) {
        if (!shuffled[i]) {
            const int rem = (int)(n - i);
            const int swap = rng.next() % std::min(window, rem);
            std::swap(input[i], input[i + swap]);
            shuffled[i + swap] = 1;
        }
    }
}

void iota1(uint64_t* out, uint64_t n)  // DataGen.hpp:79-85
{
    for (uint64_t i = 0; i < n; ++i) out[i] = i + 1;
}


// gen_zipf, mc/src/genzipf.c:95-158: random alphabet permutation (:28-53), cumulative LUT (:60-93), one rand() and one
// binary search per draw (:118-151)
int gen_zipf(GlibcRand& rng, uint64_t n, uint64_t distinct, double zipfTheta, uint64_t* out)
{
    if (distinct == 0 || distinct > 0xFFFFFFFFull || !(zipfTheta >= 0.0)) return HJ_ERR_INVALID;
    const uint32_t asz = (uint32_t)distinct;
    std::vector<uint32_t> alphabet;
    std::vector<double> lut;
    hjhost::zipf_tables(rng, asz, zipfTheta, alphabet, lut);
    for (uint64_t i = 0; i < n; ++i)
        out[i] = alphabet[hjhost::zipf_position(lut.data(), asz, ((double)rng.next()) / kRandMax)];
    return HJ_OK;
}

// RAND_RANGE(N), mc/src/generator.c:20
inline double rand_range(GlibcRand& rng, double n) { return (double)rng.next() / ((double)kRandMax + 1) * n; }

void knuth_shuffle(GlibcRand& rng, uint64_t* t, uint64_t n)      // generator.c:83-93
{
    for (int64_t i = (int64_t)n - 1; i > 0; --i) {
        const int32_t j = (int32_t)rand_range(rng, (double)i);
        std::swap(t[i], t[j]);
    }
}

void random_unique_gen(GlibcRand& rng, uint64_t* t, uint64_t n)  // generator.c:125-136
{
    iota1(t, n);
    knuth_shuffle(rng, t, n);
}

}  // namespace

extern "C" int hj_generate_data(const char* dist, uint64_t n, uint64_t distinct, int window,
                                double zipfTheta, uint64_t* out)
{
    if (!dist || (!out && n)) return HJ_ERR_INVALID;
    GlibcRand rng(0);  // srand(0), DataGen.hpp:27
    const uint32_t mod_mask = (uint32_t)(distinct - 1);
    if (strcmp(dist, "uniform") == 0) {
        if (window <= 0) return HJ_ERR_INVALID;
        for (uint64_t i = 0; i < n; ++i) out[i] = ((uint32_t)rng.next() & mod_mask) + 1;
        sort_keys(out, n);
        window_shuffle(rng, out, n, window);
    } else if (strcmp(dist, "random") == 0) {
        if (window <= 0) return HJ_ERR_INVALID;
        for (uint64_t i = 0; i < n; ++i) {
            out[i] = (uint64_t)rng.next();
            while (out[i] == 0) out[i] = (uint64_t)rng.next();
        }
        sort_keys(out, n);
        window_shuffle(rng, out, n, window);
    } else if (strcmp(dist, "sorted") == 0) {
        iota1(out, n);
    } else if (strcmp(dist, "shuffle") == 0) {
        iota1(out, n);
        // std::random_shuffle (libstdc++): j = rand() % (i + 1), swap(a[i], a[j])
        for (uint64_t i = 1; i < n; ++i) {
            const uint64_t j = (uint64_t)rng.next() % (i + 1);
            if (i != j) std::swap(out[i], out[j]);
        }
    } else if (strcmp(dist, "local_shuffle") == 0) {
        if (window <= 0) return HJ_ERR_INVALID;
        iota1(out, n);
        window_shuffle(rng, out, n, window);
    } else if (strcmp(dist, "zipf") == 0) {
        // mc/src/genzipf.c:28-151, keys over [1, distinct]; the reference DataGen
        // branch (:72-77) is an empty stub, so the seed is ours: srand(0) like the rest
        return gen_zipf(rng, n, distinct, zipfTheta, out);
    } else {
        return HJ_ERR_INVALID;  // DataGen.hpp:116-119 prints "Unknown distribution" and exits
    }
    return HJ_OK;
}

extern "C" int hj_generate_relation(const char* kind, uint64_t n, uint64_t maxid, int window, double theta,
                                    unsigned seed, uint64_t* out)
{
    if (!kind || (!out && n)) return HJ_ERR_INVALID;
    GlibcRand rng(seed);                                         // seed_generator, generator.c:56-61
    if (strcmp(kind, "pk") == 0) {
        random_unique_gen(rng, out, n);
    } else if (strcmp(kind, "pk_lshuffle") == 0) {               // lshuffle, generator.c:96-110
        if (window <= 0) return HJ_ERR_INVALID;
        iota1(out, n);
        for (uint64_t i = 0; i < n; ++i) {
            const int64_t runway = (int64_t)(n - i);
            const int mod = runway > window ? window : (int)runway;
            std::swap(out[i], out[i + (uint64_t)(rng.next() % mod)]);
        }
    } else if (strcmp(kind, "fk") == 0) {                        // create_relation_fk, generator.c:408-445
        if (maxid == 0) return HJ_ERR_INVALID;
        const uint64_t iters = n / maxid, rem = n % maxid;
        for (uint64_t i = 0; i < iters; ++i) random_unique_gen(rng, out + maxid * i, maxid);
        if (rem) random_unique_gen(rng, out + maxid * iters, rem);
    } else if (strcmp(kind, "nonunique") == 0) {                 // random_gen, generator.c:230-238
        if (maxid == 0 || maxid > 0x7FFFFFFFull) return HJ_ERR_INVALID;
        for (uint64_t i = 0; i < n; ++i) out[i] = (uint64_t)(int32_t)rand_range(rng, (double)(int32_t)maxid);
    } else if (strcmp(kind, "zipf") == 0) {
        return gen_zipf(rng, n, maxid, theta, out);
    } else {
        return HJ_ERR_INVALID;
    }
    return HJ_OK;
}

// hj_rand.h -- glibc's rand() stream and the Zipf tables built from it, shared by the host generators
// (hj_datagen.cpp) and the streaming device generator (hj_api.hip: hj_zipf_open / hj_zipf_next_dev). Host code only.
#pragma once

#include <cmath>
#include <cstdint>
#include <thread>
#include <utility>
#include <vector>

namespace hjhost {

// glibc stdlib/random_r.c, TYPE_3: the additive feedback r[i] = r[i-3] + r[i-31] over 31 words, seeded by the Lehmer
// LCG 16807 mod 2^31-1, first 310 outputs discarded, result = r >> 1. The reference draws from libc rand() after
// srand(seed) (DataGen.hpp:27, mc/src/generator.c:56-61); restating the generator keeps the inputs bit-identical
// without libc's locked global state. tests/test_datagen.py checks it against libc rand() itself.
class GlibcRand {
  public:
    explicit GlibcRand(unsigned seed) { reseed(seed); }
    void reseed(unsigned seed)
    {
        if (seed == 0) seed = 1;
        int32_t word = (int32_t)seed;
        st_[0] = (uint32_t)word;
        for (int i = 1; i < 31; ++i) {
            const long hi = word / 127773, lo = word % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            word = (int32_t)w;
            st_[i] = (uint32_t)word;
        }
        f_ = 3; r_ = 0;
        for (int k = 0; k < 310; ++k) (void)next();
    }
    inline int next()
    {
        const uint32_t v = (st_[f_] += st_[r_]);
        if (++f_ >= 31) f_ = 0;
        if (++r_ >= 31) r_ = 0;
        return (int)(v >> 1);
    }

  private:
    uint32_t st_[31];
    int f_, r_;
};

constexpr int kRandMax = 2147483647;

// gen_alphabet + gen_zipf_lut of mc/src/genzipf.c:28-93: a random permutation of 1..alphabet (consumes alphabet - 1
// draws of rng, in the reference's order) and the cumulative distribution lut[i] = sum_{j<=i} j^-theta / sum_all.
// The 2 * alphabet pow() calls of the reference are the expensive part (20 s at 2^28): the terms are computed once,
// in parallel; both running sums are then taken serially in index order, so every partial sum -- and with it every
// lut entry -- is bit-identical to the reference's.
inline void zipf_tables(GlibcRand& rng, uint32_t alphabetSize, double theta, std::vector<uint32_t>& alphabet,
                        std::vector<double>& lut)
{
    alphabet.resize(alphabetSize);
    for (uint32_t i = 0; i < alphabetSize; ++i) alphabet[i] = i + 1;
    for (uint32_t i = alphabetSize - 1; i > 0; --i) {
        const unsigned k = (unsigned)((unsigned long)i * (unsigned long)rng.next() / kRandMax);
        std::swap(alphabet[i], alphabet[k]);
    }
    lut.resize(alphabetSize);
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 32 ? 32 : nt);
    if (alphabetSize < (1u << 16)) nt = 1;
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            const uint64_t b = (uint64_t)alphabetSize * t / nt, e = (uint64_t)alphabetSize * (t + 1) / nt;
            for (uint64_t i = b; i < e; ++i) lut[i] = 1.0 / std::pow((double)(i + 1), theta);
        });
    for (auto& x : th) x.join();
    double scaling = 0.0, sum = 0.0;
    for (uint32_t i = 0; i < alphabetSize; ++i) scaling += lut[i];
    for (uint32_t i = 0; i < alphabetSize; ++i) { sum += lut[i]; lut[i] = sum / scaling; }
}

// one draw of gen_zipf (genzipf.c:118-151): position of r in the cumulative table
inline uint32_t zipf_position(const double* lut, uint32_t alphabetSize, double r)
{
    if (lut[0] >= r) return 0;
    unsigned left = 0, right = alphabetSize - 1;
    while (right - left > 1) {
        const unsigned m = (left + right) / 2;
        if (lut[m] < r) left = m; else right = m;
    }
    return right;
}

}  // namespace hjhost

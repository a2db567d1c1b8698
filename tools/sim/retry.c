// Wave-level simulation of k_build_own's insert path (fast step + retry queue with look-before-leap), to count retry
// entries and rounds under variations of the fast step. One wavefront after the other (no cross-wave races), tile
// drain every 8 steps like the kernel. usage: retry <file of u64 tuples> <n> <scheme>
//   scheme 0: fast step = one atomicMin at the home slot (the kernel as it is)
//   scheme 1: a lane starts at home + (number of earlier lanes of the step with the same home), drops at once if that is >= 4
//   scheme 2 / 3: the same, but only lanes whose home lies in [base, base + 64) / [base, base + 128) take part,
//                 base = home of lane 0 - 16 / - 48 (what a per-wavefront LDS bitmask table of that size can resolve)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define EMPTY (~0ull)
static uint64_t *tab, mask; static const int PL = 4;
typedef struct { uint64_t mine; uint64_t pos; } ent_t;
static ent_t q[1 << 16]; static int qn;
static uint64_t drops, entries, rounds, fastFails, atomics;
static uint64_t home_of(uint64_t v) { return (uint32_t)v & mask; }
static void attempt(uint64_t mine, uint64_t pos)   // one atomicMin; pushes the follow-up, if any
{
    atomics++;
    uint64_t old = tab[pos];
    if (mine < old) tab[pos] = mine;
    if (old == EMPTY || old == mine) return;
    ent_t e; e.mine = old > mine ? old : mine; e.pos = (pos + 1) & mask;
    q[qn++] = e;
}
static void retry_round(void)
{
    int take = qn < 64 ? qn : 64; qn -= take; rounds++; entries += take;
    ent_t cur[64]; memcpy(cur, q + qn, take * sizeof(ent_t));
    // phase 1: all lanes look at the state at the start of the round
    uint64_t tgt[64]; int st[64];    // st: 0 atomic, 1 dropped
    for (int l = 0; l < take; l++) {
        uint64_t pos = cur[l].pos; int budget = PL - (int)((pos - home_of(cur[l].mine)) & mask);
        while (budget > 0 && tab[pos] < cur[l].mine) { pos = (pos + 1) & mask; budget--; }
        st[l] = budget <= 0; tgt[l] = pos;
    }
    for (int l = 0; l < take; l++) { if (st[l]) drops++; else attempt(cur[l].mine, tgt[l]); }
}
int main(int argc, char **argv)
{
    uint64_t n = strtoull(argv[2], 0, 10); int scheme = atoi(argv[3]);
    uint64_t *R = malloc(n * 8); FILE *f = fopen(argv[1], "rb"); if (fread(R, 8, n, f) != n) return 1; fclose(f);
    mask = 2 * n - 1; tab = malloc(2 * n * 8); memset(tab, 0xFF, 2 * n * 8);
    uint64_t steps = 0;
    for (uint64_t b = 0; b + 64 <= n; b += 64, steps++) {
        while (qn >= 64) retry_round();
        for (int l = 0; l < 64; l++) {
            uint64_t i = b + l, mine = (i << 32) | R[i], pos = R[i] & mask;
            int r = 0;
            if (scheme == 1) for (int k = 0; k < l; k++) r += (R[b + k] & mask) == (R[i] & mask);
            if (scheme >= 2) {
                const uint64_t span = scheme == 2 ? 64 : 128, base = (R[b] & mask) - (scheme == 2 ? 16 : 48);
                if (((R[i] & mask) - base) < span) for (int k = 0; k < l; k++) r += (R[b + k] & mask) == (R[i] & mask);
            }
            if (r >= PL) { drops++; continue; }
            int before = qn;
            attempt(mine, (pos + r) & mask);
            fastFails += qn - before;
        }
        if ((steps & 7) == 7) while (qn) retry_round();
    }
    while (qn) retry_round();
    uint64_t occ = 0, sum = 0; for (uint64_t i = 0; i < 2 * n; i++) if (tab[i] != EMPTY) { occ++; sum += (uint32_t)tab[i]; }
    printf("scheme %d: drops %llu placed %llu tablesum %llu | fast-step failures/tuple %.3f, retry entries/tuple %.3f, rounds/step %.2f, entries/round %.1f, atomics/tuple %.2f\n",
           scheme, (unsigned long long)drops, (unsigned long long)occ, (unsigned long long)sum, (double)fastFails / n, (double)entries / n,
           (double)rounds / steps, (double)entries / rounds, (double)atomics / n);
    return 0;
}

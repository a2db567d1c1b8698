// Simulates the index-priority protocol as one 64-lane wavefront would run it, to count the
// lock-step rounds per group of tuples under different lane<->tuple assignments.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define EMPTY (~0ull)
static uint64_t *tab; static uint64_t mask; static int PL = 4;
static uint64_t drops, atomics, reads; static int PRE;
typedef struct { uint64_t mine, pos; int budget, act; } lane_t;
// one lock-step pass over `cnt` tuples held one per lane; returns rounds
static int run_step(lane_t *L, int cnt)
{
    int rounds = 0, any = 1;
    while (any) {
        any = 0;
        // phase 1: every active lane scans from the state at the start of the round
        if (PRE) for (int l = 0; l < cnt; l++) {
            if (!L[l].act) continue;
            while (L[l].budget > 0 && tab[L[l].pos] < L[l].mine) { L[l].pos = (L[l].pos + 1) & mask; L[l].budget--; reads++; }
        }
        // phase 2: the atomics, applied in lane order
        for (int l = 0; l < cnt; l++) {
            if (!L[l].act) continue;
            if (L[l].budget == 0) { drops++; L[l].act = 0; continue; }
            atomics++;
            uint64_t old = tab[L[l].pos];
            if (L[l].mine < old) tab[L[l].pos] = L[l].mine;
            if (old == EMPTY) { L[l].act = 0; continue; }
            if (old > L[l].mine) {
                L[l].mine = old;
                uint64_t home = (uint32_t)old & mask;
                L[l].budget = PL - (int)(((L[l].pos - home) & mask) + 1);
            } else L[l].budget--;
            L[l].pos = (L[l].pos + 1) & mask;
            any = 1;
        }
        rounds++;
    }
    return rounds;
}
int main(int argc, char **argv)
{
    const char *file = argv[1]; int scheme = atoi(argv[2]); uint64_t n = strtoull(argv[3], 0, 10);
    int per = argc > 4 ? atoi(argv[4]) : 4; PRE = argc > 5 ? atoi(argv[5]) : 0;     // tuples per lane per group
    uint64_t *R = malloc(n * 8); FILE *f = fopen(file, "rb"); if (fread(R, 8, n, f) != n) return 1; fclose(f);
    mask = 2 * n - 1; tab = malloc((2 * n) * 8); memset(tab, 0xFF, 2 * n * 8);
    uint64_t totalRounds = 0, groups = 0; lane_t L[64];
    const int G = 64 * per;
    for (uint64_t b = 0; b + G <= n; b += G, groups++) {
        for (int s = 0; s < per; s++) {
            for (int l = 0; l < 64; l++) {
                uint64_t i = scheme == 0 ? b + 64 * s + l      /* A: step s, lane l <-> tuple 64s + l */
                                         : b + (uint64_t)per * l + s;   /* B: lane l walks [per*l, per*l+per) */
                L[l].mine = (i << 32) | R[i]; L[l].pos = R[i] & mask; L[l].budget = PL; L[l].act = 1;
            }
            totalRounds += run_step(L, 64);
        }
    }
    uint64_t occ = 0; for (uint64_t i = 0; i < 2 * n; i++) occ += tab[i] != EMPTY;
    printf("scheme %d per %d pre %d: rounds/step = %.2f atomics/tuple = %.2f reads/tuple=%.2f drops=%llu placed=%llu\n", scheme, per, PRE,
           (double)totalRounds / (groups * per), (double)atomics / n, (double)reads / n, (unsigned long long)drops, (unsigned long long)occ);
    return 0;
}

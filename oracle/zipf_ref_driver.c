/* zipf_ref_driver.c -- TEST INFRASTRUCTURE. A main() around the reference's own gen_zipf()
 * (mc/src/genzipf.c:95-158), which oracle/Makefile compiles from the reference checkout WHERE IT
 * LIES into oracle/_ref/genzipf_ref (nothing of the reference is copied into this repository).
 * It pins the Zipf generators of the oracle (orc_generate_zipf) and of the product
 * (hj_generate_data("zipf")): tests/golden/make_zipf_ref.py runs it and stores what it prints in
 * tests/golden/zipf_ref.json.
 *
 *   genzipf_ref <stream_size> <alphabet_size> <theta> <seed> [<first>]
 * prints one JSON object: the first <first> keys, the sum of all keys and an FNV-1a hash over all
 * keys (as little-endian uint32), after srand(seed) -- mc seeds with 12345 / 54321
 * (mc/src/main.c:337-338), DataGen with 0 (DataGen.hpp:27). */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "genzipf.h" /* the reference's header: item_t = tuple_t {int32 key; int32 payload} */

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s stream_size alphabet_size theta seed [first]\n", argv[0]); return 2; }
    const unsigned int n = (unsigned int)strtoul(argv[1], NULL, 10);
    const unsigned int alphabet = (unsigned int)strtoul(argv[2], NULL, 10);
    const double theta = atof(argv[3]);
    const unsigned int seed = (unsigned int)strtoul(argv[4], NULL, 10);
    const unsigned int first = argc > 5 ? (unsigned int)strtoul(argv[5], NULL, 10) : 32;
    item_t *out = (item_t *)calloc(n ? n : 1, sizeof(item_t));   /* gen_zipf writes the key word only */
    if (!out) return 1;
    srand(seed);
    gen_zipf(n, alphabet, theta, &out);
    uint64_t sum = 0, h = 1469598103934665603ull;
    for (unsigned int i = 0; i < n; i++) {
        const uint32_t k = (uint32_t)out[i].key;
        sum += k;
        for (int b = 0; b < 4; b++) { h ^= (k >> (8 * b)) & 0xFFu; h *= 1099511628211ull; }
    }
    printf("{\"stream_size\": %u, \"alphabet_size\": %u, \"theta\": %.17g, \"seed\": %u, \"first\": [", n, alphabet, theta, seed);
    for (unsigned int i = 0; i < first && i < n; i++) printf("%s%" PRIu32, i ? ", " : "", (uint32_t)out[i].key);
    printf("], \"sum\": %" PRIu64 ", \"fnv1a64\": %" PRIu64 "}\n", sum, h);
    free(out);
    return 0;
}

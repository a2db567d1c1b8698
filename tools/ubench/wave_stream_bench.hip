// Microbenchmark: what the memory system gives the k_build_wave geometry (hj_build_wave.hip) -- one chunk per
// wavefront, 16 wavefronts per CU, a tile of 512 tuples loaded as 8 x 8 bytes per lane and prefetched in registers,
// the tile passed through the wavefront's LDS ring and retired as 1 KiB runs of 16-byte stores -- against the same
// instruction mix with grid-strided tiles (a plain copy), wider loads, deeper prefetch, store flavours, and a retire
// loop whose trip count the compiler cannot see (then every wait for a prefetched tile is s_waitcnt vmcnt(0) and also
// waits for the stores issued after the prefetch: vmcnt counts loads and stores together, in order).
//   hipcc -O3 --offload-arch=gfx950 -o wave_stream_bench wave_stream_bench.hip && ./wave_stream_bench [log2n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

constexpr int kWaves = 4;           // per workgroup; they never synchronise
constexpr int kTile = 512;          // tuples per wavefront and tile
typedef unsigned long long v2 __attribute__((ext_vector_type(2)));

// MODE 0: wavefront c owns chunk c (static), MODE 1: tiles strided over all wavefronts (copy order)
// LOADW 8 / 16 bytes per lane and load, DEPTH tiles in flight, NT nontemporal stores, VAR retire trip count opaque,
// WR 0 = read only
template <int MODE, int LOADW, int DEPTH, int NT, int VAR, int WR>
__global__ void __launch_bounds__(kWaves * 64, 4)
k(const uint64_t* __restrict__ R, uint64_t* __restrict__ T, uint64_t n, uint32_t chunkLen, uint32_t nChunks, int nst,
  unsigned long long* sink)
{
    extern __shared__ __align__(16) uint64_t lds[];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t c = blockIdx.x * kWaves + wave;
    if (c >= nChunks) return;
    uint64_t* const win = lds + wave * 1024;                   // 8 KiB ring per wavefront, as the product kernel
    const uint32_t nWaves = nChunks;
    const uint32_t tiles = MODE == 0 ? chunkLen / kTile : (uint32_t)((n / kTile - c + nWaves - 1) / nWaves);
    auto tile_base = [&](uint32_t t) -> uint64_t {
        return MODE == 0 ? (uint64_t)c * chunkLen + (uint64_t)t * kTile : ((uint64_t)t * nWaves + c) * kTile;
    };
    uint64_t buf[DEPTH][8];
    auto issue = [&](uint64_t (&b)[8], uint32_t t) {
        const bool ok = t < tiles;
        const uint64_t* p = R + (ok ? tile_base(t) : tile_base(0));
        if (LOADW == 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) b[j] = p[lane + 64 * j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2 x = *reinterpret_cast<const v2*>(p + 2 * lane + 128 * j);
                b[2 * j] = x.x; b[2 * j + 1] = x.y;
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(buf[d], (uint32_t)d);
    unsigned long long acc = 0;
    for (uint32_t t0 = 0; t0 < tiles; t0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint32_t t = t0 + d;
            if (t >= tiles) break;
            uint64_t cur[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) cur[j] = buf[d][j];
            issue(buf[d], t + DEPTH);
            if (WR) {
                // through the ring, then out in 1 KiB runs
                uint64_t* half = win + (t & 1) * 512;
                if (LOADW == 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) half[lane + 64 * j] = cur[j];
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v2 x; x.x = cur[2 * j]; x.y = cur[2 * j + 1]; *reinterpret_cast<v2*>(half + 2 * lane + 128 * j) = x; }
                }
                uint64_t* out = T + tile_base(t);
                const int trips = VAR ? nst : 4;
                for (int g = 0; g < trips; ++g) {
                    const v2 x = *reinterpret_cast<const v2*>(half + 128 * g + 2 * lane);
                    if (NT) __builtin_nontemporal_store(x, reinterpret_cast<v2*>(out + 128 * g) + lane);
                    else *(reinterpret_cast<v2*>(out + 128 * g) + lane) = x;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += cur[j];
            }
        }
    }
    if (acc == 42) sink[0] = acc;
}

static uint64_t* gR; static uint64_t* gT; static unsigned long long* gSink;


// ---- references: a plain grid-strided 16-byte copy, a fill and a read, at the same sizes --------------------------
template <int NT, int UNROLL>
__global__ void __launch_bounds__(256) k_copy(const v2* __restrict__ in, v2* __restrict__ out, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256 * UNROLL;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 * UNROLL + threadIdx.x; i < nvec; i += stride) {
        v2 x[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = in[i + 256 * u];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) __builtin_nontemporal_store(x[u], out + i + 256 * u); else out[i + 256 * u] = x[u];
        }
    }
}
template <int NT>
__global__ void __launch_bounds__(256) k_fill(v2* __restrict__ out, uint64_t nvec)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    v2 x; x.x = ~0ull; x.y = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
// every wavefront fills its own contiguous chunk (the build kernel's write pattern alone)
template <int NT>
__global__ void __launch_bounds__(256) k_fill_chunks(v2* __restrict__ out, uint64_t vecPerChunk)
{
    const uint32_t lane = threadIdx.x & 63;
    v2* o = out + ((uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * vecPerChunk;
    v2 x; x.x = ~0ull; x.y = ~0ull;
    for (uint64_t i = lane; i < vecPerChunk; i += 64) {
        if (NT) __builtin_nontemporal_store(x, o + i); else o[i] = x;
    }
}
__global__ void __launch_bounds__(256) k_read(const v2* __restrict__ in, uint64_t nvec, unsigned long long* sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) { const v2 x = in[i]; acc += x.x + x.y; }
    if (acc == 42) sink[0] = acc;
}

template <typename F>
static void timeit(const char* what, double bytes, F launch)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("%-60s : %8.1f us  %.2f TB/s\n", what, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

static void references(uint64_t n)
{
    const uint64_t nvec = n / 2;
    const double b1 = n * 8.0;
    for (int blocks : {2048, 8192, 65536}) {
        char name[128];
        snprintf(name, sizeof name, "plain copy 16 B, %d blocks", blocks);
        timeit(name, 2 * b1, [&] { k_copy<0, 1><<<blocks, 256>>>((const v2*)gR, (v2*)gT, nvec); });
        snprintf(name, sizeof name, "plain copy 16 B nt, %d blocks", blocks);
        timeit(name, 2 * b1, [&] { k_copy<1, 1><<<blocks, 256>>>((const v2*)gR, (v2*)gT, nvec); });
        snprintf(name, sizeof name, "plain copy 16 B x4 unrolled nt, %d blocks", blocks);
        timeit(name, 2 * b1, [&] { k_copy<1, 4><<<blocks, 256>>>((const v2*)gR, (v2*)gT, nvec); });
        snprintf(name, sizeof name, "fill 16 B, %d blocks", blocks);
        timeit(name, b1, [&] { k_fill<0><<<blocks, 256>>>((v2*)gT, nvec); });
        snprintf(name, sizeof name, "fill 16 B nt, %d blocks", blocks);
        timeit(name, b1, [&] { k_fill<1><<<blocks, 256>>>((v2*)gT, nvec); });
        snprintf(name, sizeof name, "read 16 B, %d blocks", blocks);
        timeit(name, b1, [&] { k_read<<<blocks, 256>>>((const v2*)gR, nvec, gSink); });
    }
    for (int chunks : {4096, 32768}) {
        char name[128];
        snprintf(name, sizeof name, "fill, one contiguous chunk per wavefront, %d chunks", chunks);
        timeit(name, b1, [&] { k_fill_chunks<0><<<chunks / 4, 256>>>((v2*)gT, nvec / chunks); });
        snprintf(name, sizeof name, "fill nt, one contiguous chunk per wavefront, %d chunks", chunks);
        timeit(name, b1, [&] { k_fill_chunks<1><<<chunks / 4, 256>>>((v2*)gT, nvec / chunks); });
    }
    timeit("hipMemcpyAsync device to device", 2 * b1, [&] { hipMemcpyAsync(gT, gR, n * 8, hipMemcpyDeviceToDevice, 0); });
    timeit("hipMemsetAsync", b1, [&] { hipMemsetAsync(gT, 0xFF, n * 8, 0); });
}

template <int MODE, int LOADW, int DEPTH, int NT, int VAR, int WR>
void run(const char* what, uint64_t n, int rounds)
{
    const uint32_t resident = 16 * 256;
    uint32_t nChunks = resident * rounds;
    uint32_t chunkLen = (uint32_t)(n / nChunks);
    if (MODE == 1) { nChunks = resident; chunkLen = 0; }
    const size_t ldsBytes = 40 << 10;
    auto fn = k<MODE, LOADW, DEPTH, NT, VAR, WR>;
    hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid((nChunks + kWaves - 1) / kWaves);
    fn<<<grid, kWaves * 64, ldsBytes>>>(gR, gT, n, chunkLen, nChunks, 4, gSink);
    hipEventRecord(a);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) fn<<<grid, kWaves * 64, ldsBytes>>>(gR, gT, n, chunkLen, nChunks, 4, gSink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double bytes = n * 8.0 * (WR ? 2 : 1);
    printf("%-34s mode=%d loadw=%2d depth=%d nt=%d var=%d wr=%d rounds=%d : %8.1f us  %.2f TB/s\n", what, MODE, LOADW, DEPTH, NT, VAR, WR,
           rounds, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char** argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30;
    const uint64_t n = 1ull << log2n;
    if (hipMalloc(&gR, n * 8) != hipSuccess || hipMalloc(&gT, n * 8) != hipSuccess || hipMalloc(&gSink, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(gR, 1, n * 8); hipMemset(gT, 0, n * 8);
    printf("n = 2^%d tuples: %.2f GB read, %.2f GB written per launch\n", log2n, n * 8e-9, n * 8e-9);
    const int rmax = log2n >= 30 ? 8 : log2n >= 28 ? 2 : 1;
    if (argc > 2) { references(n); return 0; }
    // the product kernel's shape: static chunks, 8-byte loads, one tile prefetched, nt stores, opaque retire loop
    run<0, 8, 1, 1, 1, 1>("product shape, 1 round", n, 1);
    run<0, 8, 1, 1, 1, 1>("product shape, rounds", n, rmax);
    run<0, 8, 1, 1, 0, 1>("counted retire loop", n, rmax);
    run<0, 8, 2, 1, 1, 1>("2 tiles in flight", n, rmax);
    run<0, 8, 2, 1, 0, 1>("2 tiles in flight, counted", n, rmax);
    run<0, 8, 3, 1, 0, 1>("3 tiles in flight, counted", n, rmax);
    run<0, 16, 1, 1, 1, 1>("16-byte loads", n, rmax);
    run<0, 16, 2, 1, 0, 1>("16-byte loads, 2 tiles, counted", n, rmax);
    run<0, 8, 1, 0, 1, 1>("plain stores", n, rmax);
    run<1, 8, 1, 1, 1, 1>("strided tiles (copy order)", n, 1);
    run<1, 8, 2, 1, 0, 1>("strided tiles, 2 tiles, counted", n, 1);
    run<1, 16, 2, 1, 0, 1>("strided, 16-byte, 2 tiles, counted", n, 1);
    run<0, 8, 1, 1, 1, 0>("read only, static", n, rmax);
    run<0, 8, 2, 1, 0, 0>("read only, static, 2 tiles", n, rmax);
    run<1, 8, 1, 1, 1, 0>("read only, strided", n, 1);
    run<1, 16, 2, 1, 0, 0>("read only, strided, 16-byte, 2", n, 1);
    return 0;
}

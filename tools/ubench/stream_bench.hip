// Microbenchmark: read-stream rate of the k_build_own geometry (512 workgroups x 512 threads, 64 KiB LDS each
// = 2 workgroups per CU, one tile prefetched in registers) vs load width and tiles in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int WIDTH, int DEPTH, int BARRIERS>   // bytes per load (8 or 16), tiles in flight, barriers per tile
__global__ void __launch_bounds__(512) k(const uint64_t* __restrict__ R, uint64_t n, uint64_t chunkLen, unsigned long long* out)
{
    extern __shared__ uint64_t lds[];
    constexpr int PER = 8;                              // tuples per thread per tile
    constexpr int TILE = 512 * PER;
    const uint64_t cb = (uint64_t)blockIdx.x * chunkLen;
    if (cb >= n) return;
    const uint32_t clen = (uint32_t)((cb + chunkLen < n ? cb + chunkLen : n) - cb);
    const uint64_t* Rc = R + cb;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long acc = 0;
    uint64_t buf[DEPTH][PER];
    auto issue = [&](uint32_t tb, uint64_t (&b)[PER]) {
        if (WIDTH == 8) {
#pragma unroll
            for (int j = 0; j < PER; ++j) { uint32_t o = tb + wave * 512 + 64 * j + lane; b[j] = Rc[o < clen ? o : clen - 1]; }
        } else {
#pragma unroll
            for (int j = 0; j < PER / 2; ++j) {
                uint32_t o = tb + wave * 512 + 128 * j + 2 * lane; if (o + 1 >= clen) o = clen - 2;
                ulonglong2 t = *reinterpret_cast<const ulonglong2*>(Rc + o); b[2 * j] = t.x; b[2 * j + 1] = t.y;
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) issue(d * TILE, buf[d]);
    for (uint32_t tb = 0; tb < clen; tb += DEPTH * TILE) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int j = 0; j < PER; ++j) acc += buf[d][j];
            issue(tb + (DEPTH + d) * TILE, buf[d]);
#pragma unroll
            for (int b = 0; b < BARRIERS; ++b) { lds[threadIdx.x] = acc; __syncthreads(); acc += lds[(threadIdx.x + 64) & 511]; }
        }
    }
    if (acc == 42) out[0] = acc;
}

template <int WIDTH, int DEPTH, int BARRIERS>
void run(const uint64_t* R, uint64_t n, unsigned long long* out, int nChunks, size_t ldsBytes)
{
    uint64_t chunkLen = (n + nChunks - 1) / nChunks;
    hipFuncSetAttribute((const void*)k<WIDTH, DEPTH, BARRIERS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<WIDTH, DEPTH, BARRIERS><<<nChunks, 512, ldsBytes>>>(R, n, chunkLen, out);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k<WIDTH, DEPTH, BARRIERS><<<nChunks, 512, ldsBytes>>>(R, n, chunkLen, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("width=%2d depth=%d barriers=%d chunks=%4d lds=%3zuK : %.1f us  %.2f TB/s\n", WIDTH, DEPTH, BARRIERS, nChunks, ldsBytes >> 10,
           ms * 1e3, n * 8.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    const uint64_t n = 1ull << 27;
    uint64_t* R; unsigned long long* out;
    hipMalloc(&R, n * 8); hipMalloc(&out, 64); hipMemset(R, 1, n * 8);
    for (size_t lds : {(size_t)76 << 10, (size_t)40 << 10, (size_t)16 << 10}) {
        int wgPerCu = lds > (60 << 10) ? 2 : lds > (30 << 10) ? 3 : 4;
        int chunks = 256 * wgPerCu;
        run<8, 1, 0>(R, n, out, chunks, lds);
        run<8, 1, 5>(R, n, out, chunks, lds);
        run<16, 1, 0>(R, n, out, chunks, lds);
        run<8, 2, 0>(R, n, out, chunks, lds);
        run<8, 2, 5>(R, n, out, chunks, lds);
        run<16, 2, 5>(R, n, out, chunks, lds);
    }
    return 0;
}

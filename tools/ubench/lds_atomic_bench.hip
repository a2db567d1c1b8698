// Microbenchmark: LDS atomic throughput on gfx950 as a function of active lanes,
// operand width and address pattern. One workgroup of W waves per CU, every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <typename T, int MODE>
__global__ void k(unsigned long long* out, int iters, int activeLanes, int waves)
{
    __shared__ T lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (T)~0ull;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T acc = 0;
    long long t0 = clock64();
    if (lane < activeLanes) {
        // MODE 0: distinct consecutive addresses; 1: all lanes same address; 2: pairs share an address; 3: stride 2
        int idx = wave * 64 + (MODE == 0 ? lane : MODE == 1 ? 0 : MODE == 2 ? (lane >> 1) : 2 * lane);
        T v = (T)(((unsigned long long)threadIdx.x << 20) + 1000000);
        for (int i = 0; i < iters; ++i) {
            T old = atomicMin(&lds[(idx + i * 512) & 8191], v);
            acc += old;          // keeps the returned value live (dependent chain on acc only)
            v -= 1;
        }
    }
    long long t1 = clock64();
    if (acc == 12345) out[1] = (unsigned long long)acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (unsigned long long)(t1 - t0);
}

template <typename T, int MODE>
void run(const char* name, int waves, int active, unsigned long long* d)
{
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<T, MODE><<<256, waves * 64>>>(d, iters, active, waves);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<T, MODE><<<256, waves * 64>>>(d, iters, active, waves);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    double instr = (double)iters * waves;             // wave-instructions per CU
    printf("%-28s waves/CU=%2d active=%2d  %.1f ns/wave-instr/CU  (%.0f cycles@2.1GHz)  clk=%llu\n", name, waves, active,
           ms * 1e6 / instr, ms * 1e6 / instr * 2.1, h[0]);
}

int main()
{
    unsigned long long* d; hipMalloc(&d, 64);
    for (int waves : {4, 8, 16}) {
        run<unsigned long long, 0>("u64 min rtn distinct", waves, 64, d);
        run<unsigned long long, 0>("u64 min rtn distinct", waves, 16, d);
        run<unsigned long long, 0>("u64 min rtn distinct", waves, 4, d);
        run<unsigned long long, 1>("u64 min rtn same-addr", waves, 64, d);
        run<unsigned long long, 2>("u64 min rtn pairs", waves, 64, d);
        run<unsigned int, 0>("u32 min rtn distinct", waves, 64, d);
        run<unsigned int, 0>("u32 min rtn distinct", waves, 16, d);
        run<unsigned int, 1>("u32 min rtn same-addr", waves, 64, d);
        run<unsigned int, 2>("u32 min rtn pairs", waves, 64, d);
    }
    return 0;
}

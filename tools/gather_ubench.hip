// Micro-benchmark: what does a random 4-byte gather cost at the L2 <-> fabric interface of MI355X, and what does
// rocprofv3's FETCH_SIZE report for it?  (The sparse-label sweep kernel gathers one n_kw entry per allowed topic
// per site; MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced reads.)
//   hipcc --offload-arch=gfx950 -O3 tools/gather_ubench.hip -o /tmp/gub
//   /tmp/gub                      -> rate of random line touches for several footprints / touches per line
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/gub one <footprint MiB> <ints per touch: 1 | 4 | 16, or 2 = pair>
// Every lane reads `W` consecutive ints at a pseudo-random, W*4-byte aligned place of a buffer of the given footprint;
// N touches per launch.  Known byte counts: distinct 128-byte lines touched ~ N (footprint >> N*128 is not required:
// with a footprint far above the 256 MiB Infinity Cache nearly every touch is a DRAM line fill).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int W, int UNROLL>
__global__ void __launch_bounds__(256) gather(const int *buf, uint64_t n_slots, int iters, int *sink, uint64_t seed)
{
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    int acc = 0;
    for (int it = 0; it < iters; it += UNROLL) {
        int v[UNROLL][W];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t slot = mix(seed + tid * 0x9E3779B97F4A7C15ull + (uint64_t)(it + u)) % n_slots;
            const int *p = buf + slot * W;
#pragma unroll
            for (int w = 0; w < W; ++w) v[u][w] = p[w];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int w = 0; w < W; ++w) acc += v[u][w];
    }
    if (acc == 0x7fffffff) *sink = acc;
}

// two 4-byte reads per touch, 64 bytes apart inside one 128-byte aligned line: if the L2 fills whole 128-byte lines the
// second read is free (same FETCH_SIZE and time as one read); if it fills 64-byte halves, both double
template <int UNROLL>
__global__ void __launch_bounds__(256) gather_pair(const int *buf, uint64_t n_lines, int iters, int *sink, uint64_t seed)
{
    const uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    int acc = 0;
    for (int it = 0; it < iters; it += UNROLL) {
        int v[UNROLL][2];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const uint64_t line = mix(seed + tid * 0x9E3779B97F4A7C15ull + (uint64_t)(it + u)) % n_lines;
            const int *p = buf + line * 32;
            v[u][0] = p[0]; v[u][1] = p[16];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u][0] + v[u][1];
    }
    if (acc == 0x7fffffff) *sink = acc;
}

static double run_pair(const int *buf, size_t bytes, int *sink, int iters, int blocks, int reps)
{
    const uint64_t n_lines = bytes / 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather_pair<8>), dim3(blocks), dim3(256), 0, 0, buf, n_lines, iters, sink, 1ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((gather_pair<8>), dim3(blocks), dim3(256), 0, 0, buf, n_lines, iters, sink, 2ull + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int W>
static double run(const int *buf, size_t bytes, int *sink, int iters, int blocks, int reps)
{
    const uint64_t n_slots = bytes / (4 * W);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((gather<W, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_slots, iters, sink, 1ull);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((gather<W, 8>), dim3(blocks), dim3(256), 0, 0, buf, n_slots, iters, sink, 2ull + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t max_bytes = 8ull << 30;
    int *buf, *sink;
    hipMalloc(&buf, max_bytes);
    hipMalloc(&sink, 4);
    hipMemset(buf, 0, max_bytes);
    const int blocks = 256 * 16, iters = 256;                  // 16 workgroups per CU, 256 touches per lane
    const double touches = (double)blocks * 256 * iters;
    if (argc >= 4 && !strcmp(argv[1], "one")) {
        const size_t bytes = (size_t)atol(argv[2]) << 20;
        const int W = atoi(argv[3]);
        double ms = W == 2 ? run_pair(buf, bytes, sink, iters, blocks, 3)
                  : W == 1 ? run<1>(buf, bytes, sink, iters, blocks, 3) : W == 4 ? run<4>(buf, bytes, sink, iters, blocks, 3)
                                                                                 : run<16>(buf, bytes, sink, iters, blocks, 3);
        printf("{\"footprint_MiB\": %zu, \"ints_per_touch\": %d, \"touches_per_launch\": %.0f, \"ms\": %.4f, \"launches\": 4}\n",
               bytes >> 20, W, touches, ms);
        return 0;
    }
    printf("%-14s %-6s %10s %14s %16s %16s\n", "footprint", "ints", "ms", "G touches/s", "TB/s at 64 B", "TB/s at 128 B");
    const size_t sizes[] = {16ull << 20, 128ull << 20, 205ull << 20, 1ull << 30, 8ull << 30};
    for (size_t bytes : sizes) {
        for (int W : {1, 4, 16}) {
            double ms = W == 1 ? run<1>(buf, bytes, sink, iters, blocks, 3) : W == 4 ? run<4>(buf, bytes, sink, iters, blocks, 3)
                                                                                     : run<16>(buf, bytes, sink, iters, blocks, 3);
            const double rate = touches / (ms * 1e-3);
            printf("%-10zu MiB %-6d %10.3f %14.2f %16.2f %16.2f\n", bytes >> 20, W, ms, rate / 1e9, rate * 64 / 1e12,
                   rate * 128 / 1e12);
        }
    }
    for (size_t bytes : {205ull << 20, 8ull << 30}) {
        const double ms = run_pair(buf, bytes, sink, iters, blocks, 3);
        printf("%-10zu MiB %-6s %10.3f %14.2f   (two reads 64 B apart in one 128-byte line per touch)\n", bytes >> 20, "pair", ms,
               touches / (ms * 1e-3) / 1e9);
    }
    return 0;
}

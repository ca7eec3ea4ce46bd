// Micro-benchmark: throughput and drain latency of no-return int32 global atomics in the pattern of the sweep
// kernel's commit (commit_site: ACTIVE lanes per wave each hit a random cell of a [V][KP] delta matrix), with
// FILL VALU instructions between two commits and one s_waitcnt vmcnt(0) per step, as in the site loop.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_ubench.hip -o /tmp/aub && /tmp/aub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// MODE 0: atomics   MODE 1: plain 4-byte stores to the same addresses   MODE 2: no memory operation
template <int MODE>
__global__ void __launch_bounds__(256, 3) commit_like(int32_t *buf, uint32_t cells, int steps, int stride, int fill,
                                                      long long *cycles)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const bool active = (lane % stride) == 0;
    uint32_t h = mix(blockIdx.x * 256u + tid + 1u);
    float acc = (float)tid;
    long long waited = 0;
    for (int s = 0; s < steps; ++s) {
        h = mix(h + s);
        const uint32_t a = h % cells, b = mix(h) % cells;
        if (active) {
            if (MODE == 0) { atomicAdd(buf + a, -1); atomicAdd(buf + b, 1); }
            if (MODE == 1) { buf[a] = s; buf[b] = s; }
        }
        for (int i = 0; i < fill; ++i) asm volatile("v_fmac_f32 %0, %0, %0" : "+v"(acc));
        const long long t0 = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        waited += clock64() - t0;
    }
    if (acc == 1.2345f) buf[0] = 1;
    if (tid == 0) atomicAdd((unsigned long long *)cycles, (unsigned long long)waited);
}

int main()
{
    const uint32_t cells = 50000u * 128u;                  // synth1's n_kw_delta
    int32_t *buf; long long *cyc;
    hipMalloc(&buf, (size_t)cells * 4); hipMemset(buf, 0, (size_t)cells * 4);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 3, steps = 2000;
    printf("mode   lanes fill  ms      Gops/s   wait cycles/step/wave\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int stride : {8, 32, 1})
            for (int fill : {0, 300, 1200}) {
                hipMemset(cyc, 0, 8);
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(commit_like<0>, dim3(blocks), dim3(256), 0, 0, buf, cells, steps, stride, fill, cyc);
                    if (mode == 1) hipLaunchKernelGGL(commit_like<1>, dim3(blocks), dim3(256), 0, 0, buf, cells, steps, stride, fill, cyc);
                    if (mode == 2) hipLaunchKernelGGL(commit_like<2>, dim3(blocks), dim3(256), 0, 0, buf, cells, steps, stride, fill, cyc);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
                const double ops = (double)blocks * 4 * (64 / stride) * 2.0 * steps;
                printf("%-6s %5d %4d  %7.3f %7.2f  %9.1f\n", mode == 0 ? "atomic" : mode == 1 ? "store" : "none", 64 / stride, fill,
                       ms, ops / ms * 1e-6, (double)c / 2 / ((double)blocks * steps));
            }
    return 0;
}

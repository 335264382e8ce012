// Dependent-load latency probe: one wave follows a random cycle through a buffer of 32-byte slots (the size of a tree
// node), every lane its own chain.  Prints ns per dependent load for working sets from 256 KiB to 4 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>

__global__ void chase(const uint4* buf, uint32_t start_stride, uint32_t steps, uint32_t* out, unsigned long long* ticks)
{
    uint32_t i = threadIdx.x * start_stride;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (uint32_t k = 0; k < steps; ++k) i = buf[2 * (size_t)i].x;
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x + blockIdx.x * blockDim.x] = i;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    for (size_t mib : {1, 2, 8, 32, 64, 128, 512, 2048}) {
        const size_t slots = mib * 1024 * 1024 / 32;
        std::vector<uint32_t> perm(slots);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 rng(1234);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint4> host(2 * slots);
        for (size_t k = 0; k < slots; ++k) host[2 * (size_t)perm[k]].x = perm[(k + 1) % slots];   // one cycle through all slots
        uint4* d; uint32_t* out; unsigned long long* ticks;
        hipMalloc((void**)&d, host.size() * sizeof(uint4));
        hipMalloc((void**)&out, 64 * 1024 * sizeof(uint32_t));
        hipMalloc((void**)&ticks, 1024 * sizeof(unsigned long long));
        hipMemcpy(d, host.data(), host.size() * sizeof(uint4), hipMemcpyHostToDevice);
        for (int lanes : {1, 64}) {
            for (int blocks : {1, 256}) {
                const uint32_t steps = 4096;
                chase<<<blocks, lanes>>>(d, (uint32_t)(slots / 64 / 2), 256, out, ticks);        // warm
                chase<<<blocks, lanes>>>(d, (uint32_t)(slots / 64), steps, out, ticks);
                unsigned long long t[1024];
                hipMemcpy(t, ticks, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                double sum = 0; for (int b = 0; b < blocks; ++b) sum += (double)t[b];
                printf("%5zu MiB  lanes %2d  waves %3d : %.0f ns per dependent load (100 MHz ticks)\n", mib, lanes, blocks,
                       sum / blocks * 10.0 / steps);
            }
        }
        hipFree(d); hipFree(out); hipFree(ticks);
    }
    return 0;
}

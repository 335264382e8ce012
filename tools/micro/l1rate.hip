// L1 line-lookup rate probe: every lane reads 96 bytes (three 32-byte tree nodes) at a pseudo-random 32-byte-aligned place
// of a table, trip after trip, with all waves of the chip resident.  Three ways to issue the same reads:
//   mode 0: six 16-byte loads per lane at its own place (every instruction touches one line per lane);
//   mode 1: lanes 2k and 2k+1 share the work: in each instruction they read the two adjacent 16-byte halves of one 32-byte
//           node, first three instructions for lane 2k's place, the next three for lane 2k+1's (every instruction touches one
//           line per PAIR of lanes);
//   mode 2: four lanes read four adjacent 16-byte pieces (64 bytes) -- 128 bytes per place, eight instructions.
// Prints bytes of wanted data per second; if the vector L1 looks up one line per cycle whatever the lanes want from it,
// mode 1 moves the same data about twice as fast as mode 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(1024) void probe(const uint4* table, uint32_t mask, uint32_t trips, uint32_t* out)
{
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < trips; ++k) {
        x = x * 1664525u + 1013904223u;
        const uint32_t place = (x >> 8) & mask;                  // in units of 32 bytes
        if (MODE == 0) {
            const uint4* a = table + 2 * (size_t)place;
            const uint4 q0 = a[0], q1 = a[1], q2 = a[2], q3 = a[3], q4 = a[4], q5 = a[5];
            acc += q0.x ^ q1.y ^ q2.z ^ q3.w ^ q4.x ^ q5.y;
        } else if (MODE == 1) {
            const uint32_t other = __builtin_amdgcn_update_dpp(0u, place, 0xB1, 0xF, 0xF, false);   // the partner's place
            const uint32_t pa = (lane & 1u) ? other : place, pb = (lane & 1u) ? place : other;
            const uint4* a = table + 2 * (size_t)pa + (lane & 1u);
            const uint4* b = table + 2 * (size_t)pb + (lane & 1u);
            const uint4 q0 = a[0], q1 = a[2], q2 = a[4], q3 = b[0], q4 = b[2], q5 = b[4];
            acc += q0.x ^ q1.y ^ q2.z ^ q3.w ^ q4.x ^ q5.y;
        } else {
            uint32_t p4[4];
            p4[0] = __builtin_amdgcn_update_dpp(0u, place, 0x00, 0xF, 0xF, false);
            p4[1] = __builtin_amdgcn_update_dpp(0u, place, 0x55, 0xF, 0xF, false);
            p4[2] = __builtin_amdgcn_update_dpp(0u, place, 0xAA, 0xF, 0xF, false);
            p4[3] = __builtin_amdgcn_update_dpp(0u, place, 0xFF, 0xF, 0xF, false);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4* a = table + 2 * (size_t)p4[j] + (lane & 3u);
                const uint4 q0 = a[0], q1 = a[4];
                acc += q0.x ^ q1.y;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (size_t mib : {4, 64, 1024}) {
        const size_t slots = mib * 1024 * 1024 / 32;
        uint4* d; uint32_t* out;
        if (hipMalloc(&d, slots * 32 + 256) != hipSuccess || hipMalloc(&out, (size_t)cus * 1024 * 4) != hipSuccess) return 1;
        hipMemset(d, 1, slots * 32 + 256);
        const uint32_t mask = (uint32_t)slots - 8u - 1u;         // (a power of two minus 1 minus room for the 128-byte reads)
        const uint32_t trips = 4096;
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) probe<0><<<cus, 1024>>>(d, mask, trips, out);
                if (mode == 1) probe<1><<<cus, 1024>>>(d, mask, trips, out);
                if (mode == 2) probe<2><<<cus, 1024>>>(d, mask, trips, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            const double places = (double)cus * 1024 * trips;
            const double bytes = places * (mode == 2 ? 128.0 : 96.0);
            printf("table %4zu MiB mode %d: %.3f ms, %.2f G places/s, %.2f TB/s of wanted bytes, %.2f wave-trips per CU per us\n", mib, mode, ms,
                   places / ms / 1e6, bytes / ms / 1e9, places / 64 / cus / ms / 1e3);
        }
        hipFree(d); hipFree(out);
    }
    return 0;
}

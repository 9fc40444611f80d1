// read_bw.hip - how fast can MI355X stream a 20 GB buffer through its CUs read-only?
// Build: hipcc --offload-arch=gfx950 -O3 -o read_bw read_bw.hip ;  run: ./read_bw
// Variants: loads per thread in flight (U), threads per workgroup, workgroups per CU, plain /
// non-temporal loads, contiguous slabs / interleaved tiles.  Prints GB/s per variant (best of 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool SLAB>
__global__ void rd(const u32x4* __restrict__ p, long long n16, unsigned* out) {
    const long long T = (long long)gridDim.x * blockDim.x;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    if (SLAB) {
        const long long per = (n16 + gridDim.x - 1) / gridDim.x;
        const long long lo = per * blockIdx.x, hi = lo + per < n16 ? lo + per : n16;
        for (long long i = lo + threadIdx.x; i < hi; i += (long long)blockDim.x * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long j = i + (long long)u * blockDim.x;
                v[u] = j < hi ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    } else {
        for (long long i = t; i < n16; i += T * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long j = i + (long long)u * T;
                v[u] = j < n16 ? (NT ? __builtin_nontemporal_load(p + j) : p[j]) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int U, bool NT, bool SLAB>
void run(const u32x4* p, long long n16, unsigned* out, int block, int wg_per_cu, const char* name) {
    int cus = 256;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) == hipSuccess) cus = pr.multiProcessorCount;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<U, NT, SLAB>), dim3(cus * wg_per_cu), dim3(block), 0, 0, p, n16, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && ms < best) best = ms;
    }
    printf("%-26s U=%d block=%4d wg/cu=%2d  %7.3f ms  %7.1f GB/s\n", name, U, block, wg_per_cu, best,
           n16 * 16.0 / (best * 1e-3) / 1e9);
    fflush(stdout);
}

int main() {
    const long long bytes = 20000000000LL, n16 = bytes / 16;
    u32x4* p;
    unsigned* out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    hipMemset(p, 1, bytes);
    hipDeviceSynchronize();
    for (int wg : {1, 2, 4, 8}) {
        run<4, true, false>(p, n16, out, 1024, wg, "nt interleaved");
        run<8, true, false>(p, n16, out, 512, wg, "nt interleaved");
        run<4, false, false>(p, n16, out, 1024, wg, "plain interleaved");
        run<4, true, true>(p, n16, out, 1024, wg, "nt slabs");
        run<2, true, false>(p, n16, out, 256, wg * 4, "nt interleaved");
    }
    run<8, true, false>(p, n16, out, 1024, 1, "nt interleaved");
    run<16, true, false>(p, n16, out, 256, 4, "nt interleaved");
    return 0;
}

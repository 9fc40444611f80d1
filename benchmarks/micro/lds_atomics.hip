// lds_atomics.hip - what does an LDS atomic cost on MI355X, by form and by address pattern?
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip ;  run: ./lds_atomics
// Every wave repeats the same 8 operations on 8 addresses per lane that are fixed before the loop
// (no address arithmetic inside it), one workgroup per CU.  Printed: shader cycles per
// wave-instruction per CU (s_memtime of the slowest workgroup; the CU's waves share the LDS pipe).
//   patterns: lin      lane l -> element l + 64 j          (conflict-free, all distinct)
//             rnd(W)   uniform random element below W      (random bank conflicts, few equal)
//             same(m)  groups of m adjacent lanes share an element, groups on distinct banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

enum Op { ADD_U32, ADD_RTN_U32, ADD_U64, ADD_F64, ADD_RTN_F64, MAX_U32, WRITE_B32, READ_B32, ADD_F32, OPS };
static const char* op_name[OPS] = {"ds_add_u32", "ds_add_rtn_u32", "ds_add_u64", "ds_add_f64", "ds_add_rtn_f64",
                                   "ds_max_u32", "ds_write_b32", "ds_read_b32", "ds_add_f32"};

template <int OP>
__global__ __launch_bounds__(1024) void k(const unsigned* __restrict__ idx, int iters, unsigned long long* cyc,
                                         double* sink) {
    extern __shared__ unsigned long long lds[];  // 128 KiB
    unsigned* l32 = (unsigned*)lds;
    double* lf = (double*)lds;
    float* lf32 = (float*)lds;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0;
    unsigned a[8];
    for (int j = 0; j < 8; ++j) a[j] = idx[((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8 + j];
    __syncthreads();
    double accd = 0;
    unsigned accu = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == ADD_U32) __hip_atomic_fetch_add(&l32[a[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == ADD_RTN_U32) accu += __hip_atomic_fetch_add(&l32[a[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == ADD_U64) __hip_atomic_fetch_add(&lds[a[j]], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == ADD_F64) __hip_atomic_fetch_add(&lf[a[j]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == ADD_RTN_F64) accd += __hip_atomic_fetch_add(&lf[a[j]], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == MAX_U32) __hip_atomic_fetch_max(&l32[a[j]], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OP == WRITE_B32) asm volatile("ds_write_b32 %0, %1" ::"v"(a[j] * 4u), "v"((unsigned)it) : "memory");
            if (OP == READ_B32) {
                unsigned r;
                asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(4)" : "=v"(r) : "v"(a[j] * 4u) : "memory");
                accu += r;
            }
            if (OP == ADD_F32) __hip_atomic_fetch_add(&lf32[a[j]], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (accd == 12345.678 || accu == 0x12345678u) sink[0] = accd + accu;
}

typedef void (*kern_t)(const unsigned*, int, unsigned long long*, double*);
static kern_t kerns[OPS] = {k<ADD_U32>, k<ADD_RTN_U32>, k<ADD_U64>, k<ADD_F64>, k<ADD_RTN_F64>,
                            k<MAX_U32>, k<WRITE_B32>, k<READ_B32>, k<ADD_F32>};

static unsigned long long rng = 88172645463325252ULL;
static unsigned rnd() {
    rng ^= rng << 13;
    rng ^= rng >> 7;
    rng ^= rng << 17;
    return (unsigned)(rng >> 16);
}

int main() {
    int cus = 256;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, 0) == hipSuccess) cus = pr.multiProcessorCount;
    const int iters = 2000;
    unsigned* d_idx;
    unsigned long long* d_cyc;
    double* d_sink;
    const size_t nidx = (size_t)cus * 1024 * 8;
    if (hipMalloc(&d_idx, nidx * 4) != hipSuccess || hipMalloc(&d_cyc, cus * 8) != hipSuccess ||
        hipMalloc(&d_sink, 64) != hipSuccess)
        return 1;
    std::vector<unsigned> h(nidx);
    std::vector<unsigned long long> hc(cus);
    for (int o = 0; o < OPS; ++o) hipFuncSetAttribute((const void*)kerns[o], hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    struct Pat {
        const char* name;
        int kind, arg;
    };
    const Pat pats[] = {{"lin", 0, 0},          {"rnd(1024)", 1, 1024}, {"rnd(8192)", 1, 8192}, {"rnd(16384)", 1, 16384},
                        {"same(2)", 2, 2},      {"same(4)", 2, 4},      {"same(8)", 2, 8},      {"same(64)", 2, 64},
                        {"rnd(64)", 1, 64}};
    for (int block : {1024, 256}) {
        for (const Pat& p : pats) {
            for (size_t t = 0; t < (size_t)cus * 1024; ++t) {
                const unsigned lane = t & 63;
                for (int j = 0; j < 8; ++j) {
                    unsigned e;
                    if (p.kind == 0) e = lane + 64 * j + 512 * ((t >> 6) & 15);
                    else if (p.kind == 1) e = rnd() % (unsigned)p.arg;
                    else e = lane / p.arg + 64 * j + 512 * ((t >> 6) & 15);
                    h[t * 8 + j] = e;
                }
            }
            hipMemcpy(d_idx, h.data(), nidx * 4, hipMemcpyHostToDevice);
            printf("block %4d  %-11s", block, p.name);
            for (int o = 0; o < OPS; ++o) {
                double best = 1e30;
                for (int r = 0; r < 3; ++r) {
                    hipLaunchKernelGGL(kerns[o], dim3(cus), dim3(block), 131072, 0, d_idx, iters, d_cyc, d_sink);
                    if (hipDeviceSynchronize() != hipSuccess) {
                        printf("launch failed\n");
                        return 1;
                    }
                    hipMemcpy(hc.data(), d_cyc, cus * 8, hipMemcpyDeviceToHost);
                    unsigned long long mx = 0;
                    for (int c = 0; c < cus; ++c) mx = hc[c] > mx ? hc[c] : mx;
                    const double per = (double)mx / ((double)iters * 8 * (block / 64));
                    if (per < best) best = per;
                }
                printf("  %s %6.1f", op_name[o] + 3, best);
            }
            printf("\n");
            fflush(stdout);
        }
    }
    return 0;
}

// Micro-benchmark: how fast does one SIMD retire v_mfma_f32_16x16x32_f16 in the register pattern of
// scan_kernel's 8 x 2 shape (two accumulators used alternately, 48 distinct B fragments, A fragments
// changing every second MFMA), with one or two waves per SIMD?  Prints cycles per MFMA per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NB>
__global__ void k(const half8* in, float* out, int iters, long long* cyc) {
    half8 b[NB];
    half8 a[4];
#pragma unroll
    for (int i = 0; i < NB; ++i) b[i] = in[(threadIdx.x + i * 64) & 1023];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = in[(threadIdx.x * 3 + i) & 1023];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 24; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int idx = (s * 2 + j);
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                             : "+v"(acc[idx % NACC]) : "v"(a[s & 3]), "v"(b[idx % NB]));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int NB>
void run(const char* tag, int threads, const half8* in, float* out, long long* cyc) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, NB>), dim3(256), dim3(threads), 0, 0, in, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NB>), dim3(256), dim3(threads), 0, 0, in, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0;
    for (int i = 0; i < 256; ++i) c += h[i];
    c /= 256;
    const double waves_per_simd = threads / 256.0;
    const double mfma_per_simd = 48.0 * iters * waves_per_simd;
    printf("%-40s threads %d: %.3f ms, %.1f shader cycles per MFMA per SIMD, clock %.2f GHz, %.0f TFLOP/s\n", tag, threads,
           ms, c / mfma_per_simd, c / (ms * 1e-3) / 1e9, 2.0 * 16 * 16 * 32 * mfma_per_simd * 1024 / (ms * 1e-3) / 1e12);
}

int main() {
    half8* in;
    float* out;
    long long* cyc;
    hipMalloc(&in, 1024 * sizeof(half8));
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    _Float16 h[8192];
    srand(1);
    for (int i = 0; i < 8192; ++i) h[i] = (_Float16)((rand() % 2001 - 1000) / 1000.0f);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 48>("2 acc alternating, 48 B frags", 256, in, out, cyc);
        run<2, 48>("2 acc alternating, 48 B frags", 512, in, out, cyc);
        run<4, 48>("4 acc, 48 B frags", 256, in, out, cyc);
        run<4, 48>("4 acc, 48 B frags", 512, in, out, cyc);
        run<1, 48>("1 acc, 48 B frags", 256, in, out, cyc);
        run<1, 48>("1 acc, 48 B frags", 512, in, out, cyc);
        run<2, 4>("2 acc, 4 B frags", 512, in, out, cyc);
        run<16, 16>("16 acc, 16 B frags", 512, in, out, cyc);
    }
    return 0;
}

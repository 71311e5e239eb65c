// Probe: does VALU work hide under v_mfma_f32_32x32x16_bf16 on gfx950, (a) inside ONE wave's instruction stream, (b) across the waves of
// a SIMD?  One 512- or 1024-thread workgroup per CU; every "multiplier" wave issues MF MFMAs per iteration, VALU fillers are placed
// either between its own MFMAs (mode 1) or in separate "mover" waves (mode 2).  Prints cycles per iteration (s_memtime) and TF/s.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/coissue_probe tools/coissue_probe.hip && tools/coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned split_step(float a, float b) {   // the f32x3 split's instruction mix: cvt_pk, shifts, subs
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
}

// MODE 0: MFMA only (8 waves).  MODE 1: FILL VALU instructions after every MFMA in the same wave (8 waves).
// MODE 2: 8 MFMA waves + 8 waves that do the same total VALU work (16 waves).
template <int MODE, int FILL>
__global__ __launch_bounds__(MODE == 2 ? 1024 : 512) void probe(float* out, int iters, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float x = threadIdx.x * 0.37f, y = threadIdx.x * 0.11f + 1.f;
    unsigned sink = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 2 && wave >= 8) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 24 * FILL / 3; ++m) {       // 3 VALU per step: cvt_pk, shift, sub
                const unsigned h = split_step(x, y);
                const float xh = __uint_as_float(h << 16);
                x = x - xh + 1.0001f;
                sink ^= h;
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
                if (MODE == 1) {
#pragma unroll
                    for (int f = 0; f < FILL / 3; ++f) {
                        const unsigned h = split_step(x, y);
                        const float xh = __uint_as_float(h << 16);
                        x = x - xh + 1.0001f;
                        sink ^= h;
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, FILL, 0);   // FILL VALU
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = x + (float)sink;
    for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE, int FILL>
static void run(const char* name) {
    const int blocks = 256, iters = 2000, nt = MODE == 2 ? 1024 : 512;
    float* o; unsigned long long* clk;
    CK(hipMalloc(&o, blocks * 1024 * 4)); CK(hipMalloc(&clk, blocks * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((probe<MODE, FILL>), dim3(blocks), dim3(nt), 0, 0, o, iters, clk);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<MODE, FILL>), dim3(blocks), dim3(nt), 0, 0, o, iters, clk);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double fl = (double)blocks * 8 * iters * 24 * 32768.0;
    printf("%-58s %7.3f ms  %7.1f TF/s(bf16)  %6.0f cycles/iteration (24 MFMAs per wave, 2 MFMA waves per SIMD: 1536 = MFMA-bound)\n", name, ms, fl / ms / 1e9, (double)c / iters);
    CK(hipFree(o)); CK(hipFree(clk));
}

int main() {
    run<0, 0>("MFMA only, 8 waves");
    run<1, 3>("same wave: 3 VALU after every MFMA");
    run<1, 6>("same wave: 6 VALU after every MFMA");
    run<1, 9>("same wave: 9 VALU after every MFMA");
    run<2, 3>("mover waves: 3 VALU per MFMA, in 8 other waves");
    run<2, 6>("mover waves: 6 VALU per MFMA, in 8 other waves");
    run<2, 9>("mover waves: 9 VALU per MFMA, in 8 other waves");
    return 0;
}

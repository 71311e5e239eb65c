// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of the f16x2 GEMM (round-5 review, Weak #3 / Next #3a):
//   k_stream_vgpr     every byte of a 512 MB buffer once, global_load_dwordx4 (16 bytes per lane into registers)
//   k_stream_lds      the same bytes once, global_load_lds_dwordx4 (16 bytes per lane, global -> LDS, the GEMM's request type)
//   k_xcd_reread_lds  a 10 MB buffer read ONCE BY EVERY XCD (the 32 workgroups of an XCD cover it once: blockIdx.x & 7 = XCD), by
//                     global_load_lds_dwordx4 - the A operand of the wide GEMM launch, which every XCD's L2 fetches for itself.
//                     80 MB requested at the fabric, 10 MB unique: does FETCH_SIZE count the 7 re-reads (Infinity-Cache hits)?
//   k_xcd_reread_lds  again with a 200 MB sweep of another buffer in between (the step's weights) - not needed: each launch is cold
// The byte counts are printed; run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- ./tools/fetch_calib
// and compare Counter_Value (KiB) per kernel with the printed figures (tools/fetch_calib.py does).
//   hipcc -O2 --offload-arch=gfx950 -o tools/fetch_calib tools/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// 256 threads; workgroup w reads bytes [w * per, (w + 1) * per) in 4 KB rounds (one 16-byte load per lane and round)
__global__ __launch_bounds__(256) void k_stream_vgpr(const float4* __restrict__ src, long long per16, float* __restrict__ sink) {
    const float4* p = src + (long long)blockIdx.x * per16 + threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long long i = 0; i < per16; i += 256) {
        const float4 v = p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;      // (never true: keeps the loads)
}

__global__ __launch_bounds__(256) void k_stream_lds(const float4* __restrict__ src, long long per16, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) float4 lds[4 * 256];         // four 4 KB landing buffers
    const float4* p = src + (long long)blockIdx.x * per16 + threadIdx.x;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)lds)
                        + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 1024u;
    int b = 0;
    for (long long i = 0; i < per16; i += 256) {
        glds16(p + i, lds0 + (unsigned)b * 4096u);
        b = (b + 1) & 3;
        if (b == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x].x == 12345.678f) sink[0] = lds[threadIdx.x].y;
}

// workgroup b: XCD b & 7, slice b >> 3 of `nslice` slices of the buffer: every XCD reads the whole buffer once
__global__ __launch_bounds__(256) void k_xcd_reread_lds(const float4* __restrict__ src, long long per16, float* __restrict__ sink) {
    __shared__ __attribute__((aligned(1024))) float4 lds[4 * 256];
    const float4* p = src + (long long)(blockIdx.x >> 3) * per16 + threadIdx.x;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)lds)
                        + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 1024u;
    int b = 0;
    for (long long i = 0; i < per16; i += 256) {
        glds16(p + i, lds0 + (unsigned)b * 4096u);
        b = (b + 1) & 3;
        if (b == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lds[threadIdx.x].x == 12345.678f) sink[0] = lds[threadIdx.x].y;
}

__global__ void k_fill(float* p, long long n, float v) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
    const long long BIG = 512LL << 20, SMALL = 10LL << 20;      // bytes
    float *big, *big2, *small_, *sink;
    CK(hipMalloc(&big, BIG)); CK(hipMalloc(&big2, BIG)); CK(hipMalloc(&small_, SMALL)); CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, big, BIG / 4, 1.f);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, big2, BIG / 4, 2.f);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, small_, SMALL / 4, 3.f);
    CK(hipDeviceSynchronize());
    const int WG = 2048;
    const long long per16 = BIG / 16 / WG;                      // float4 per workgroup (16384 = 64 rounds of 256)
    // each measured launch reads a buffer the previous launch did NOT touch (512 MB of other data went through the 256 MB Infinity Cache)
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream_vgpr, dim3(WG), dim3(256), 0, 0, reinterpret_cast<const float4*>(big), per16, sink);
        hipLaunchKernelGGL(k_stream_lds, dim3(WG), dim3(256), 0, 0, reinterpret_cast<const float4*>(big2), per16, sink);
    }
    CK(hipDeviceSynchronize());
    printf("k_stream_vgpr   : %lld bytes read once (global_load_dwordx4)\n", BIG);
    printf("k_stream_lds    : %lld bytes read once (global_load_lds_dwordx4)\n", BIG);
    const long long per16s = SMALL / 16 / 32;                   // 32 slices (one per workgroup of an XCD): 20480 float4 = 80 rounds
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_stream_vgpr, dim3(WG), dim3(256), 0, 0, reinterpret_cast<const float4*>(big), per16, sink);     // flush the caches
        hipLaunchKernelGGL(k_xcd_reread_lds, dim3(256), dim3(256), 0, 0, reinterpret_cast<const float4*>(small_), per16s, sink);
    }
    CK(hipDeviceSynchronize());
    printf("k_xcd_reread_lds: %lld unique bytes, read once by each of 8 XCDs = %lld bytes requested at the fabric\n", SMALL, SMALL * 8);
    return 0;
}

// Where does one global_load_lds_dwordx4 of a wave land?  One 64-lane workgroup, 160 KB of LDS filled with 0xEE, lane l sends 16 bytes
// (four words l * 4 + {0,1,2,3} + 0x1000) with M0 = base; the whole LDS is copied out and the host reports the landing place of every lane's
// piece.  (Is the LDS address M0 + 16 * lane?  Does M0 reach beyond 64 KB / 128 KB?)   hipcc -O2 --offload-arch=gfx950 -o tools/lds_dma_probe tools/lds_dma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(64) void probe(const unsigned* src, unsigned* out, unsigned base) {
    __shared__ __attribute__((aligned(1024))) unsigned lds[40960];
    for (int i = threadIdx.x; i < 40960; i += 64) lds[i] = 0xEEEEEEEEu;
    __syncthreads();
    const unsigned* g = src + threadIdx.x * 4;
    const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)lds) + base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(lds0) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 40960; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 0x1000 + i;
    unsigned *src, *out;
    CK(hipMalloc(&src, 1024)); CK(hipMalloc(&out, 163840));
    CK(hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice));
    const unsigned bases[] = {0, 1024, 61440, 65536, 98304, 130048, 131072, 147456, 162816};
    for (unsigned b : bases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, b);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> o(40960);
        CK(hipMemcpy(o.data(), out, 163840, hipMemcpyDeviceToHost));
        int found = 0, first = -1, linear = 1;
        for (int i = 0; i < 40960; ++i) if (o[i] != 0xEEEEEEEEu) { if (first < 0) first = i; ++found; }
        // expected: words [base/4, base/4 + 256) = 0x1000 .. 0x10ff in order
        for (int i = 0; i < 256; ++i) if (b / 4 + i >= 40960 || o[b / 4 + i] != 0x1000u + i) linear = 0;
        printf("M0 = LDS base + %6u: %3d words changed, first at byte %6d, lane l at base + 16 l in order: %s", b, found, first * 4, linear ? "yes" : "NO");
        if (!linear && first >= 0) { printf("  (first words:"); for (int i = 0; i < 8; ++i) printf(" %x", o[first + i]); printf(")"); }
        printf("\n");
    }
    return 0;
}

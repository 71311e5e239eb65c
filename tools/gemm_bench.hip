// Standalone check + timing of the grouped fp32-MFMA GEMM (vsr-guided-cic_amd/csrc/gemm_f32.h) on the shapes
// of one beam-5 / greedy decoder timestep.  Build & run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/gemm_bench tools/gemm_bench.hip && /tmp/gemm_bench [M] [tile] [units]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// The product headers under csrc/ have one code path each and no diagnostic hooks; the round-2 copies with the -D hooks (GEMM_ABLATE /
// GEMM_STAMP / GEMM_PHASES / X3_EXP / B16_ABLATE ...) that produced the ablation numbers of DESIGN.md section 4 are in the history:
// `git show 4789752:tools/ablate/gemm_f32.h` etc.  tools/x3_ablate.py patches scratch copies of the shipped headers instead.
#define GEMM_ABLATE 0
#include "../vsr-guided-cic_amd/csrc/gemm_f32.h"
#include "../vsr-guided-cic_amd/csrc/gemm_bf16.h"
#include "../vsr-guided-cic_amd/csrc/gemm_x3.h"
#include "../vsr-guided-cic_amd/csrc/gemm_x3s.h"
#include "../vsr-guided-cic_amd/csrc/gemm_h2.h"
#include "../vsr-guided-cic_amd/csrc/gemm_h2a.h"
#include "../vsr-guided-cic_amd/csrc/gemm_b16a.h"

using namespace vsr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// bf16 twins of the weight buffers (variant 1664 = gemm_nt_bf16w_kernel): same element layout, 2 bytes per element
struct Twin { const float* f; size_t n; uint16_t* b; uint32_t* h2; int slot; };
static std::vector<Twin> g_twins;
static bool g_bf16 = false, g_a16 = false;     // g_a16: variant 1665 = the bf16 kernel reading bf16 images of A (GemmSeg::A16)
static bool g_h2 = false;                      // variants 5200 (wide) / 5300 (streaming): the f16x2 kernels of gemm_h2.h (fp16-pair images of W, scale exponents)
static int* g_exps = nullptr;                  // device table of scale exponents, one slot per dev_rand buffer
static unsigned* g_bounds = nullptr;
static int g_nslot = 0;
constexpr int MAX_SLOTS = 4096;
// f32x3 kernel (gemm_x3.h): variant "3300 <tm><tn>" with <tm><tn> = 22 (128 x 256 tile; also "1") or 21 (128 x 128)
static bool is_x3(int tm) { return tm == 3300; }
static bool g_h2a = false;                     // variant 5400: the all-DMA wide kernel (gemm_h2a.h): A operands are images too
static bool is_h2(int tm) { return tm == 5200 || tm == 5300 || tm == 5400; }
static void set_variant_globals(int tm) {
    g_bf16 = tm == 1664 || tm == 1665 || tm == 1666;      // 1666: the all-DMA bf16 kernel (gemm_b16a.h), bf16 images of both operands
    g_a16 = tm == 1665 || tm == 1666;
    g_h2 = is_h2(tm);
    g_h2a = tm == 5400;
}

static float* dev_rand(size_t n, unsigned seed, float amp = 1.f) {
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = amp * (((s >> 8) & 0xFFFF) / 65536.0f - 0.5f); }
    float* d;
    CK(hipMalloc(&d, (n + 8) * sizeof(float)));
    CK(hipMemset(d, 0, (n + 8) * sizeof(float)));
    CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    g_twins.push_back(Twin{d, n, nullptr, nullptr, -1});
    return d;
}

// f16x2: scale-exponent slot of a buffer (from its max |x| on the device; H2_LOOSE=<bits> loosens every bound by that many bits)
static int slot_of(const float* p) {
    if (!g_exps) { CK(hipMalloc(&g_exps, MAX_SLOTS * sizeof(int))); CK(hipMalloc(&g_bounds, MAX_SLOTS * sizeof(unsigned))); }
    for (Twin& t : g_twins)
        if (p >= t.f && p < t.f + t.n) {
            if (t.slot < 0) {
                if (g_nslot >= MAX_SLOTS) g_nslot = 0;          // (fuzz: slots are recycled together with the buffers)
                t.slot = g_nslot++;
                CK(hipMemset(g_bounds + t.slot, 0, sizeof(unsigned)));
                hipLaunchKernelGGL(vsr::k_absmax, dim3(256), dim3(256), 0, 0, t.f, (long long)t.n, g_bounds + t.slot);
                const int loose = getenv("H2_LOOSE") ? atoi(getenv("H2_LOOSE")) : 0;
                int hidx[3] = {t.slot, -1, t.slot};
                int* didx; CK(hipMalloc(&didx, sizeof(hidx))); CK(hipMemcpy(didx, hidx, sizeof(hidx), hipMemcpyHostToDevice));
                hipLaunchKernelGGL(vsr::k_h2_exps, dim3(1), dim3(64), 0, 0, g_bounds, didx, didx + 1, didx + 2, 1, g_exps);
                CK(hipDeviceSynchronize());
                CK(hipFree(didx));
                if (loose) { int e; CK(hipMemcpy(&e, g_exps + t.slot, 4, hipMemcpyDeviceToHost)); e -= loose; CK(hipMemcpy(g_exps + t.slot, &e, 4, hipMemcpyHostToDevice)); }
            }
            return t.slot;
        }
    printf("no buffer for %p\n", (const void*)p); exit(1);
}
static const float* h2_image_of(const float* W) {
    for (Twin& t : g_twins)
        if (W >= t.f && W < t.f + t.n) {
            if (!t.h2) {
                const int slot = slot_of(t.f);
                const size_t n8 = (t.n + 7) & ~size_t(7);
                CK(hipMalloc(&t.h2, n8 * 4));
                hipLaunchKernelGGL(vsr::k_f32_to_h2, dim3((unsigned)((n8 / 8 + 255) / 256)), dim3(256), 0, 0, t.f, t.h2, (long long)n8, g_exps, slot);
                CK(hipDeviceSynchronize());
            }
            return reinterpret_cast<const float*>(t.h2) + (W - t.f);
        }
    printf("no image for %p\n", (const void*)W); exit(1);
}

static const float* twin_of(const float* W) {
    for (Twin& t : g_twins)
        if (W >= t.f && W < t.f + t.n) {
            if (!t.b) {
                CK(hipMalloc(&t.b, t.n * 2));
                hipLaunchKernelGGL(vsr::k_f32_to_bf16, dim3((unsigned)((t.n / 8 + 255) / 256 + 1)), dim3(256), 0, 0, t.f, t.b, (long long)t.n);
                CK(hipDeviceSynchronize());
            }
            return reinterpret_cast<const float*>(t.b + (W - t.f));
        }
    printf("no twin for %p\n", (const void*)W); exit(1);
}

struct Builder {
    GemmArgs a;
    int slots, min_iters, tm, tn;
    Builder(int slots_, int min_iters_, int tm_ = 1, int tn_ = 1) : slots(slots_), min_iters(min_iters_), tm(tm_), tn(tn_) { memset(&a, 0, sizeof(a)); }
    GemmProb& prob(int M, int N, float* C, int ldc) { GemmProb& p = a.p[a.nprob++]; p.M = M; p.N = N; p.C = C; p.ldc = ldc; return p; }
    static void seg(GemmProb& p, const float* A, int lda, const int* idx, const float* W, int ldw, int K) {
        GemmSeg& s = p.seg[p.nseg++]; s.A = g_h2a ? h2_image_of(A) : A; s.lda = lda; s.a_idx = idx; s.W = g_bf16 ? twin_of(W) : g_h2 ? h2_image_of(W) : W; s.ldw = ldw; s.K = K;
        s.A16 = g_a16 ? reinterpret_cast<const uint16_t*>(twin_of(A)) : nullptr;
        s.exp_idx = 0;
        if (g_h2) s.exp_idx = slot_of(W) | (slot_of(A) << 16);
    }
    int finish() {
        int bm = tm >= 12 ? 128 : 64 * tm, bn = tm >= 12 ? 128 : 64 * tn;
        if (tm == 112) { bm = 64; bn = 128; }
        if (tm == 121 || tm == 221 || tm == 321) { bm = 128; bn = 64; }
        if (tm == 322) { bm = 128; bn = 128; }
        if (tm == 211) { bm = 64; bn = 64; }
        if (tm == 2242) { bm = 256; bn = 128; }
        if (tm == 2142) { bm = 256; bn = 64; }
        if (tm == 2224) { bm = 128; bn = 256; }
        if (is_x3(tm)) { bm = 128; bn = tn == 21 ? 128 : 256; }       // f32x3 kernel
        if (tm == 3400) {                                                            // weight-streaming f32x3 kernel (<= 128 rows): "3400 <MT or 0 = by M>", k-aligned pieces only
            const int mi = getenv("GEMM_PLAN_ALIGNED") ? atoi(getenv("GEMM_PLAN_ALIGNED")) : 8;
            int ns = gemm_plan_aligned(a, slots, mi, 128, x3s_bn(getenv("X3S_NS") ? atoi(getenv("X3S_NS")) : 2), X3_BK);
            if (!ns) { printf("x3s: the 64-column blocks of this launch exceed %d slots\n", slots); exit(1); }
            for (int i = 0; i < a.nprob; ++i) a.p[i].slab_stride = (long long)a.p[i].M * a.p[i].ldc;
            return ns;
        }
        if (tm == 5200 || tm == 5400) { bm = 128; bn = tn == 21 ? 128 : 256; a.exps = g_exps; }    // f16x2 wide kernel (H2_NW=3|4 stages of the weight ring)
        if (tm == 5300) {                                                            // f16x2 streaming kernel (<= 128 rows): "5300 <MT or 0 = by M>", H2S_NS=1|2 strips per wave
            a.exps = g_exps;
            const int mi = getenv("GEMM_PLAN_ALIGNED") ? atoi(getenv("GEMM_PLAN_ALIGNED")) : 8;
            int ns = gemm_plan_aligned(a, slots, mi, 128, h2s_bn(getenv("H2S_NS") ? atoi(getenv("H2S_NS")) : 1), H2_BK);
            if (!ns) { printf("h2s: the column blocks of this launch exceed %d slots\n", slots); exit(1); }
            for (int i = 0; i < a.nprob; ++i) a.p[i].slab_stride = (long long)a.p[i].M * a.p[i].ldc;
            return ns;
        }
        if (tm > 1600 && tm <= 1608) { bm = 16 * (tm - 1600); bn = 64 * tn; }      // rows-16 kernel: tm = 1600 + TM, tn = TN
        if (tm == 1664 || tm == 1665 || tm == 1666) { bm = 128; bn = tn == 21 ? 128 : 256; }       // bf16 throughput kernel (bf16 W twins; 1665: bf16 A images too); tn 21: 128 x 128 tile
        int ns = 0;
        const bool b16 = tm == 1664 || tm == 1665 || tm == 1666;
        if ((b16 || is_x3(tm) || tm == 5200 || tm == 5400) && getenv("GEMM_PLAN_ALIGNED")) ns = gemm_plan_aligned(a, slots, atoi(getenv("GEMM_PLAN_ALIGNED")), bm, bn, b16 ? B16_BK : GEMM_BK);
        if (!ns) ns = gemm_plan(a, slots, min_iters, bm, bn, b16 ? B16_BK : GEMM_BK);
        for (int i = 0; i < a.nprob; ++i) a.p[i].slab_stride = (long long)a.p[i].M * a.p[i].ldc;
        return ns;
    }
    double flops() const { return gemm_flops(a); }
    void launch(hipStream_t st) {
        dim3 g(gemm_grid(a)), b(256);
#define R16(TM_) else if (tm == 1600 + TM_ && tn == 2) hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<TM_, 2>), g, dim3(512), 0, st, a); \
                 else if (tm == 1600 + TM_ && tn == 4) hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<TM_, 4>), g, dim3(512), 0, st, a);
        if (tm == 3400) {
            int maxM = 0; for (int i = 0; i < a.nprob; ++i) maxM = a.p[i].M > maxM ? a.p[i].M : maxM;
            const int mt = tn > 1 && tn <= 8 ? tn : (maxM + 15) / 16;
            const bool ns1 = getenv("X3S_NS") && atoi(getenv("X3S_NS")) == 1;
            switch (mt) {
                case 1: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<1, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<1, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                case 2: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<2, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<2, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                case 3: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<3, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<3, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                case 4: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<4, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<4, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                case 5: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<5, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<5, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                case 6: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<6, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<6, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                case 7: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<7, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<7, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
                default: if (ns1) hipLaunchKernelGGL((gemm_nt_x3s_kernel<8, 1>), g, dim3(X3S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_x3s_kernel<8, 2>), g, dim3(X3S_THREADS), 0, st, a); break;
            }
        }
        else if (tm == 5300) {
            int maxM = 0; for (int i = 0; i < a.nprob; ++i) maxM = a.p[i].M > maxM ? a.p[i].M : maxM;
            const int mt = tn > 1 && tn <= 8 ? tn : (maxM + 15) / 16;
            const bool ns2 = getenv("H2S_NS") && atoi(getenv("H2S_NS")) == 2;
#define H2S(MT_) case MT_: if (ns2) hipLaunchKernelGGL((gemm_nt_h2s_kernel<MT_, 2>), g, dim3(H2S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_h2s_kernel<MT_, 1>), g, dim3(H2S_THREADS), 0, st, a); break;
            switch (mt) { H2S(1) H2S(2) H2S(3) H2S(4) H2S(5) H2S(6) H2S(7) default: if (ns2) hipLaunchKernelGGL((gemm_nt_h2s_kernel<8, 2>), g, dim3(H2S_THREADS), 0, st, a); else hipLaunchKernelGGL((gemm_nt_h2s_kernel<8, 1>), g, dim3(H2S_THREADS), 0, st, a); break; }
#undef H2S
        }
        else if (tm == 5400) {
            const int nw = getenv("H2_NW") ? atoi(getenv("H2_NW")) : 3;         // ring stages of the 128 x 128 tile (the 128 x 256 tile has room for three)
            if (tn == 21 && nw == 4) hipLaunchKernelGGL((gemm_nt_h2a_kernel<2, 1, 4>), g, dim3(H2_THREADS), 0, st, a);
            else if (tn == 21) hipLaunchKernelGGL((gemm_nt_h2a_kernel<2, 1, 3>), g, dim3(H2_THREADS), 0, st, a);
            else hipLaunchKernelGGL((gemm_nt_h2a_kernel<2, 2, 3>), g, dim3(H2_THREADS), 0, st, a);
        }
        else if (tm == 5200) {
            const int nw = getenv("H2_NW") ? atoi(getenv("H2_NW")) : 3;
            if (tn == 21 && nw == 4) hipLaunchKernelGGL((gemm_nt_h2_kernel<2, 1, 4>), g, dim3(H2_THREADS), 0, st, a);
            else if (tn == 21) hipLaunchKernelGGL((gemm_nt_h2_kernel<2, 1, 3>), g, dim3(H2_THREADS), 0, st, a);
            else if (nw == 4) hipLaunchKernelGGL((gemm_nt_h2_kernel<2, 2, 4>), g, dim3(H2_THREADS), 0, st, a);
            else hipLaunchKernelGGL((gemm_nt_h2_kernel<2, 2, 3>), g, dim3(H2_THREADS), 0, st, a);
        }
        else if (is_x3(tm) && tn == 21) hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 1>), g, dim3(X3_THREADS), 0, st, a);
        else if (is_x3(tm)) hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2>), g, dim3(X3_THREADS), 0, st, a);
        else if (tm == 1666 && tn == 21) hipLaunchKernelGGL((gemm_nt_b16a_kernel<2, 1>), g, dim3(B16_THREADS), 0, st, a);
        else if (tm == 1666) hipLaunchKernelGGL((gemm_nt_b16a_kernel<2, 2>), g, dim3(B16_THREADS), 0, st, a);
        else if (tm == 1664 && tn == 21) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<false, 1>), g, dim3(B16_THREADS), 0, st, a);
        else if (tm == 1665 && tn == 21) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<true, 1>), g, dim3(B16_THREADS), 0, st, a);
        else if (tm == 1664) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<false, 2>), g, dim3(B16_THREADS), 0, st, a);
        else if (tm == 1665) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<true, 2>), g, dim3(B16_THREADS), 0, st, a);
        else if (tm == 2 && tn == 2) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 2>), g, b, 0, st, a);
        R16(1) R16(2) R16(3) R16(4) R16(5) R16(6) R16(7) R16(8)
        else if (tm == 112) hipLaunchKernelGGL((gemm_nt_f32_kernel<1, 2, 2, 2>), g, b, 0, st, a);
        else if (tm == 2242) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 2, 4, 2>), g, dim3(512), 0, st, a);
        else if (tm == 2142) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 1, 4, 2>), g, dim3(512), 0, st, a);
        else if (tm == 2224) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 2, 2, 4>), g, dim3(512), 0, st, a);
        else if (tm == 121) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 1, 2, 2>), g, b, 0, st, a);
        else if (tm == 12) hipLaunchKernelGGL((gemm_nt_f32_kernel<1, 2, 4, 2>), g, dim3(512), 0, st, a);
        else if (tm == 21) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 1, 2, 4>), g, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((gemm_nt_f32_kernel<1, 1>), g, b, 0, st, a);
    }
};

static double time_it(Builder& b, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) b.launch(0);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) b.launch(0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

// register-only MFMA loop: what the matrix pipe sustains on this box (and the clock it holds: guide 'DVFS give-back' 6)
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters, unsigned long long* clk) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-4f + 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// one dependent accumulator chain per wave, operands re-read from LDS each 16 MFMAs (the GEMM's inner loop shape)
template <int LDSREAD>
__global__ __launch_bounds__(256) void mfma_chain(float* out, int iters) {
    __shared__ float sm[128 * 36];
    for (int i = threadIdx.x; i < 128 * 36; i += 256) sm[i] = i * 1e-4f;
    __syncthreads();
    f32x16 a0 = {0};
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5, wave = threadIdx.x >> 6;
    const float* ab = sm + ((wave >> 1) * 32 + r) * 36 + 4 * hh;
    const float* bb = sm + (64 + (wave & 1) * 32 + r) * 36 + 4 * hh;
    float4 av[4], bv[4];
    for (int kk = 0; kk < 4; ++kk) { av[kk] = *(const float4*)(ab + kk * 8); bv[kk] = *(const float4*)(bb + kk * 8); }
    for (int i = 0; i < iters; ++i) {
        if (LDSREAD) {
            int off = 0;
            asm volatile("" : "+v"(off));          // opaque: forces the reads to be re-issued every iteration
            for (int kk = 0; kk < 4; ++kk) { av[kk] = *(const float4*)(ab + off + kk * 8); bv[kk] = *(const float4*)(bb + off + kk * 8); }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk].x, bv[kk].x, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk].y, bv[kk].y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk].z, bv[kk].z, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk].w, bv[kk].w, a0, 0, 0, 0);
        }
        if (LDSREAD == 2) __syncthreads();
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += a0[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// L2 -> register feed rate of one workgroup per CU with the GEMM's tile access pattern and nothing else: every thread keeps
// DEPTH x 8 16-byte loads in flight (A rows of 256 B at stride lda, W rows of 128 B at stride ldw), working set L2/MALL resident
template <int DEPTH>
__global__ __launch_bounds__(512) void feed_kernel(const float* __restrict__ A, int lda, int rowsA, const uint16_t* __restrict__ W, int ldw, int rowsW,
                                                   int K, int iters, float* out) {
    const int tid = threadIdx.x;
    const int arow = tid >> 4, ak = (tid & 15) * 4, brow = tid >> 3, bk = (tid & 7) * 8;
    const int m0 = (blockIdx.x % 4) * 128, n0 = ((blockIdx.x / 4) * 256) % (rowsW - 256);
    float4 acc = make_float4(0, 0, 0, 0);
    int k = 0;
    for (int it = 0; it < iters; it += DEPTH) {
        float4 ra[DEPTH][4]; uint4 rb[DEPTH][4];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[d][i] = *reinterpret_cast<const float4*>(A + (long long)((m0 + arow + 32 * i) % rowsA) * lda + k + ak);
                rb[d][i] = *reinterpret_cast<const uint4*>(W + (long long)(n0 + brow + 64 * i) * ldw + k + bk);
            }
            k += 64; if (k + 64 > K) k = 0;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) { acc.x += ra[d][i].x + __uint_as_float(rb[d][i].x); acc.y += ra[d][i].y; acc.z += ra[d][i].z + __uint_as_float(rb[d][i].w); acc.w += ra[d][i].w; }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[blockIdx.x * 512 + tid] = acc.x;
}


// ---- fuzz: random ragged grouped launches (1-3 problems, 1-3 k segments each, row gathers, K tails, odd leading dimensions of C,
// both work decompositions) of the variant under test against an fp64 host reference.  `tools/gemm_bench fuzz <tm> <tn> [cases] [seed]`
static int fuzz(int tm, int tn, int cases, unsigned seed) {
    set_variant_globals(tm);
    const bool wide = g_bf16 || is_x3(tm) || is_h2(tm);            // the 16-wave kernels and the f16x2 images: K multiples of 8
    unsigned st = seed * 747796405u + 2891336453u;
    auto rnd = [&](int lo, int hi) { st = st * 1664525u + 1013904223u; return lo + (int)((st >> 8) % (unsigned)(hi - lo + 1)); };
    int bad = 0;
    for (int cs = 0; cs < cases; ++cs) {
        const int slots = (tm == 3400 || tm == 5300) ? (int[]){64, 256, 384, 768}[rnd(0, 3)] : (int[]){8, 64, 256, 256}[rnd(0, 3)];
        const int aligned = (tm == 3400 || tm == 5300) ? 1 : rnd(0, 1);      // (the weight-streaming kernel only has k-aligned pieces)
        if (aligned) setenv("GEMM_PLAN_ALIGNED", rnd(0, 1) ? "8" : "2", 1); else unsetenv("GEMM_PLAN_ALIGNED");
        Builder b(slots, rnd(1, 8), tm, tn);
        const int nprob = rnd(1, 3);
        struct Host { int M, N, ldc, coff; std::vector<std::vector<float>> A, W; std::vector<int> K, lda, ldw, woff; std::vector<std::vector<int>> idx; float* C; double amp2k; };
        std::vector<Host> H(nprob);
        std::vector<void*> to_free;
        for (int p = 0; p < nprob; ++p) {
            Host& h = H[p];
            h.M = rnd(0, 5) == 0 ? rnd(1, 20) : rnd(1, (tm == 3400 || tm == 5300) ? 128 : 600);
            h.N = rnd(0, 5) == 0 ? rnd(1, 40) : rnd(8, 700);
            h.coff = rnd(0, 1) ? 4 * rnd(0, 3) : rnd(0, 5);
            h.ldc = h.N + h.coff + (rnd(0, 1) ? 4 * rnd(0, 4) : rnd(0, 7));
            if (rnd(0, 2)) { h.ldc = (h.ldc + 3) & ~3; h.coff &= ~3; }
            CK(hipMalloc(&h.C, (size_t)8 * h.M * h.ldc * sizeof(float) + 64)); to_free.push_back(h.C);
            CK(hipMemset(h.C, 0, (size_t)8 * h.M * h.ldc * sizeof(float) + 64));
            GemmProb& gp = b.prob(h.M, h.N, h.C + h.coff, h.ldc);
            const int nseg = rnd(1, 3);
            double amp2k = 0;
            for (int sg = 0; sg < nseg; ++sg) {
                const int q = wide ? 8 : 4;
                const int K = q * rnd(1, rnd(0, 3) ? 520 / q : 1600 / q);
                const int rowsA = rnd(0, 1) ? h.M : h.M + rnd(1, 50);
                const int lda = (g_a16 || g_h2a) ? K + 8 * rnd(0, 2) : K + 4 * rnd(0, 3), woff = q * rnd(0, 4), ldw = woff + K + q * rnd(0, 5);
                // f16x2: the segments of a problem get operands of different magnitudes (different scale exponents: the accumulator units
                // differ from segment to segment - gemm_h2a.h rescales, gemm_h2.h scales A per segment)
                const float amp = g_h2 ? ldexpf(1.f, rnd(-6, 6)) : 1.f;
                amp2k += (double)amp * amp * K;
                float* A = dev_rand((size_t)rowsA * lda, seed * 131 + cs * 17 + p * 5 + sg, amp); to_free.push_back(A);
                float* W = dev_rand((size_t)h.N * ldw + 64, seed * 137 + cs * 19 + p * 7 + sg); to_free.push_back(W);
                std::vector<int> idx;
                int* didx = nullptr;
                if (rowsA != h.M || rnd(0, 1)) {
                    idx.resize(h.M);
                    for (int i = 0; i < h.M; ++i) idx[i] = rnd(0, rowsA - 1);
                    CK(hipMalloc(&didx, h.M * sizeof(int))); to_free.push_back(didx);
                    CK(hipMemcpy(didx, idx.data(), h.M * sizeof(int), hipMemcpyHostToDevice));
                }
                Builder::seg(gp, A, lda, didx, W + woff, ldw, K);
                std::vector<float> hA((size_t)rowsA * lda), hW((size_t)h.N * ldw + 64);
                CK(hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
                h.A.push_back(std::move(hA)); h.W.push_back(std::move(hW)); h.K.push_back(K); h.lda.push_back(lda); h.ldw.push_back(ldw); h.woff.push_back(woff);
                h.idx.push_back(idx);
            }
            h.amp2k = amp2k;
        }
        const int ns = b.finish();
        const bool tight = rnd(0, 1);                       // per-problem slab counts (gemm_tight_slabs): only those slabs are summed below
        if (tight) for (int p = 0; p < nprob; ++p) b.a.p[p].nslab = gemm_tight_slabs(b.a, p);
        b.launch(0);
        CK(hipDeviceSynchronize());
        double worst = 0, scale = 0;
        for (int p = 0; p < nprob; ++p) {
            Host& h = H[p];
            std::vector<float> hC((size_t)ns * h.M * h.ldc);
            CK(hipMemcpy(hC.data(), h.C, hC.size() * 4, hipMemcpyDeviceToHost));
            int ktot = 0; for (int k : h.K) ktot += k;
            for (int i = 0; i < h.M; ++i)
                for (int j = 0; j < h.N; ++j) {
                    double ref = 0;
                    for (size_t sg = 0; sg < h.K.size(); ++sg) {
                        const int row = h.idx[sg].empty() ? i : h.idx[sg][i];
                        const float* a = h.A[sg].data() + (size_t)row * h.lda[sg];
                        const float* w = h.W[sg].data() + (size_t)j * h.ldw[sg] + h.woff[sg];
                        for (int k = 0; k < h.K[sg]; ++k) ref += (double)a[k] * w[k];
                    }
                    double got = 0;
                    for (int q = 0; q < b.a.p[p].nslab; ++q) got += hC[(size_t)q * h.M * h.ldc + (size_t)i * h.ldc + h.coff + j];
                    worst = fmax(worst, fabs(got - ref));
                }
            scale = fmax(scale, sqrt(g_h2 ? h.amp2k : (double)ktot) * 0.083);      // |a| |w| ~ U(-0.5, 0.5) (x the segment's amplitude): products ~ 1/12 rms
            // the columns of the C window outside [coff, coff + N) must stay untouched (zero)
            for (int i = 0; i < h.M; ++i)
                for (int j = 0; j < h.ldc; ++j)
                    if (j < h.coff || j >= h.coff + h.N)
                        for (int q = 0; q < ns; ++q)
                            if (hC[(size_t)q * h.M * h.ldc + (size_t)i * h.ldc + j] != 0.f) { worst = 1e9; }
        }
        const double tol = (g_bf16 ? 2e-2 : 2e-5) * fmax(1.0, scale);
        const bool ok = worst < tol;
        bad += !ok;
        printf("fuzz %3d: %d problems (M %d N %d ...), slots %d, %s, G %d nslab %d: max |err| %.3g (tol %.3g) %s\n", cs, nprob, H[0].M, H[0].N, slots,
               b.a.aligned ? "k-aligned" : "stream-K", b.a.G, ns, worst, tol, ok ? "OK" : "FAIL");
        for (void* q : to_free) CK(hipFree(q));
        for (Twin& t : g_twins) { if (t.b) CK(hipFree(t.b)); if (t.h2) CK(hipFree(t.h2)); }
        g_twins.clear();
        g_nslot = 0;
    }
    printf("fuzz: %d of %d cases failed\n", bad, cases);
    return bad ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "chain")) {
        float* o; CK(hipMalloc(&o, 4096 * 256 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode = 0; mode < 3; ++mode)
            for (int mult = 1; mult <= 4; mult *= 2) {
                const int blocks = 256 * mult, iters = argc > 2 ? atoi(argv[2]) : 2000;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    if (mode == 0) hipLaunchKernelGGL(mfma_chain<0>, dim3(blocks), dim3(256), 0, 0, o, iters);
                    else if (mode == 1) hipLaunchKernelGGL(mfma_chain<1>, dim3(blocks), dim3(256), 0, 0, o, iters);
                    else hipLaunchKernelGGL(mfma_chain<2>, dim3(blocks), dim3(256), 0, 0, o, iters);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("chain mode %d (0 regs, 1 +ds_read, 2 +barrier) %d waves/SIMD: %.1f TF/s\n", mode, mult, (double)blocks * 4 * iters * 16 * 4096.0 / ms / 1e9);
            }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "fuzz"))
        return fuzz(argc > 2 ? atoi(argv[2]) : 1664, argc > 3 ? atoi(argv[3]) : 1, argc > 4 ? atoi(argv[4]) : 40, argc > 5 ? atoi(argv[5]) : 1);
    if (argc > 1 && !strcmp(argv[1], "feed")) {
        const int M = 512, K = 3000, N = 6144, lda = 3000, ldw = 4048;
        float* A = dev_rand((size_t)M * lda, 1); float* Wf = dev_rand((size_t)N * ldw / 2 + 64, 2);
        float* o; CK(hipMalloc(&o, 4096 * 512 * 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int blocks = 64; blocks <= 1024; blocks *= 2)
            for (int depth = 1; depth <= 4; depth *= 2) {
                const int iters = 400;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipEventRecord(e0, 0));
                    if (depth == 1) hipLaunchKernelGGL(feed_kernel<1>, dim3(blocks), dim3(512), 0, 0, A, lda, M, (const uint16_t*)Wf, ldw, N, K, iters, o);
                    else if (depth == 2) hipLaunchKernelGGL(feed_kernel<2>, dim3(blocks), dim3(512), 0, 0, A, lda, M, (const uint16_t*)Wf, ldw, N, K, iters, o);
                    else hipLaunchKernelGGL(feed_kernel<4>, dim3(blocks), dim3(512), 0, 0, A, lda, M, (const uint16_t*)Wf, ldw, N, K, iters, o);
                    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double bytes = (double)blocks * iters * 65536.0;
                printf("feed: %4d workgroups x 512 threads, %d tiles in flight: %.3f ms, %.2f TB/s aggregate, %.1f GB/s per workgroup, %.2f us per 64 KB tile\n",
                       blocks, depth, ms, bytes / ms / 1e9, 65536.0 * iters / ms / 1e6, ms * 1e3 / iters);
            }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "peak")) {
        const int blocks = 256 * (argc > 2 ? atoi(argv[2]) : 1), iters = 20000;
        float* o; unsigned long long* clk; CK(hipMalloc(&o, blocks * 256 * 4)); CK(hipMalloc(&clk, blocks * 16));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, o, iters, clk);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mfma_peak, dim3(blocks), dim3(256), 0, 0, o, iters, clk);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * 2); CK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
        double fl = (double)blocks * 4 * iters * 4 * 4096.0;
        printf("mfma_peak: %d blocks, %.3f ms, %.1f TF/s, in-kernel clock %.0f MHz\n", blocks, ms, fl / ms / 1e9, (double)h[0] / h[1] * 100.0);
        return 0;
    }
    const int M = argc > 1 ? atoi(argv[1]) : 500;
    const int slots = argc > 2 ? atoi(argv[2]) : 1024, min_iters = argc > 3 ? atoi(argv[3]) : 8;
    const int tm = argc > 4 ? atoi(argv[4]) : 1, tn = argc > 5 ? atoi(argv[5]) : 1;
    set_variant_globals(tm);
    // GEMM_ALIGNED=1: hidden sizes rounded to 1024 so that every row stride is a multiple of 128 B (cache-line aligned rows)
    const bool aligned = getenv("GEMM_ALIGNED") != nullptr;
    const int H = aligned ? 1024 : 1000, E = H, D = 2048, A = 512, V = 10000, in1 = H + D + E, in2 = H + D;

    // ---- correctness on a ragged problem: gather index, 3 segments with K tails, split-K, column window
    {
        const int m = 77, n = 150, k1 = 72, k2 = 40, k3 = g_h2a ? 104 : 100, ldw = 264, nrowsA = 200;     // (ldw a multiple of 8: the f16x2 image groups)
        float* A1 = dev_rand((size_t)nrowsA * k1, 1); float* A2 = dev_rand((size_t)nrowsA * k2, 2); float* A3 = dev_rand((size_t)m * k3, 3);
        float* W = dev_rand((size_t)n * ldw, 4);
        std::vector<int> hidx(m); for (int i = 0; i < m; ++i) hidx[i] = (i * 37 + 11) % nrowsA;
        int* idx; CK(hipMalloc(&idx, m * sizeof(int))); CK(hipMemcpy(idx, hidx.data(), m * sizeof(int), hipMemcpyHostToDevice));
        float* C; CK(hipMalloc(&C, (size_t)8 * m * 160 * sizeof(float))); CK(hipMemset(C, 0, (size_t)8 * m * 160 * sizeof(float)));
        Builder b(slots, min_iters, tm, tn);
        GemmProb& p = b.prob(m, n, C + 5, 160);
        Builder::seg(p, A1, k1, idx, W, ldw, k1); Builder::seg(p, A2, k2, idx, W + k1, ldw, k2); Builder::seg(p, A3, k3, nullptr, W + k1 + k2, ldw, k3);
        int ns = b.finish();
        b.launch(0); CK(hipDeviceSynchronize());
        std::vector<float> hC((size_t)ns * m * 160), hA1((size_t)nrowsA * k1), hA2((size_t)nrowsA * k2), hA3((size_t)m * k3), hW((size_t)n * ldw);
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hA1.data(), A1, hA1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hA2.data(), A2, hA2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hA3.data(), A3, hA3.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
        double maxerr = 0;
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
            double ref = 0;
            for (int k = 0; k < k1; ++k) ref += (double)hA1[(size_t)hidx[i] * k1 + k] * hW[(size_t)j * ldw + k];
            for (int k = 0; k < k2; ++k) ref += (double)hA2[(size_t)hidx[i] * k2 + k] * hW[(size_t)j * ldw + k1 + k];
            for (int k = 0; k < k3; ++k) ref += (double)hA3[(size_t)i * k3 + k] * hW[(size_t)j * ldw + k1 + k2 + k];
            double got = 0;
            for (int s = 0; s < ns; ++s) got += hC[(size_t)s * m * 160 + (size_t)i * 160 + 5 + j];
            maxerr = fmax(maxerr, fabs(got - ref));
        }
        const double tol = g_bf16 ? 5e-2 : 1e-4;
        printf("correctness (G %d, nslab %d): max |err| = %.3g %s\n", b.a.G, ns, maxerr, maxerr < tol ? "OK" : "FAIL");
        if (maxerr >= tol && !GEMM_ABLATE && !getenv("GEMM_NOCHECK")) return 1;
    }

    // ---- accuracy on a decoder-like product (M=64, N=256, K=1000) against fp64: this variant vs the fp32-MFMA kernel
    {
        const int m = 64, n = 256, kk = 1000;
        float* A1 = dev_rand((size_t)m * kk, 41); float* W = dev_rand((size_t)n * kk, 42);
        float* Cx; CK(hipMalloc(&Cx, (size_t)8 * m * n * sizeof(float)));
        std::vector<float> hA((size_t)m * kk), hW((size_t)n * kk);
        CK(hipMemcpy(hA.data(), A1, hA.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> ref((size_t)m * n);
        for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < kk; ++k) s += (double)hA[(size_t)i * kk + k] * hW[(size_t)j * kk + k]; ref[(size_t)i * n + j] = s; }
        for (int variant = 0; variant < 2; ++variant) {
            set_variant_globals(variant ? tm : 1);
            Builder b(slots, min_iters, variant ? tm : 1, variant ? tn : 1);
            GemmProb& p = b.prob(m, n, Cx, n); Builder::seg(p, A1, kk, nullptr, W, kk, kk);
            int ns = b.finish(); b.launch(0); CK(hipDeviceSynchronize());
            std::vector<float> hC((size_t)ns * m * n); CK(hipMemcpy(hC.data(), Cx, hC.size() * 4, hipMemcpyDeviceToHost));
            double maxe = 0, sume = 0;
            for (size_t i = 0; i < (size_t)m * n; ++i) { double g = 0; for (int q = 0; q < ns; ++q) g += hC[(size_t)q * m * n + i]; double e = fabs(g - ref[i]); maxe = fmax(maxe, e); sume += e * e; }
            printf("accuracy K=1000 (%s): max |err| %.3e rms %.3e (|C| ~ %.2f)\n", variant ? "this variant" : "fp32 MFMA 64x64", maxe, sqrt(sume / (m * n)), 2.6);
        }
    }

    set_variant_globals(tm);
    // ---- timing on the decoder shapes
    float* h2 = dev_rand((size_t)M * H, 5); float* h1 = dev_rand((size_t)M * H, 6); float* att = dev_rand((size_t)M * D, 7);
    float* emb = dev_rand((size_t)V * E, 8);
    float* Wih1 = dev_rand((size_t)4 * H * in1, 9); float* Whh1 = dev_rand((size_t)4 * H * H, 10);
    float* Wis = dev_rand((size_t)H * in1, 11); float* Whs = dev_rand((size_t)H * H, 12); float* Wig = dev_rand((size_t)H * in1, 13);
    float* Wih2 = dev_rand((size_t)4 * H * in2, 14); float* Whh2 = dev_rand((size_t)4 * H * H, 15);
    float* Wout = dev_rand((size_t)V * H, 16); float* Wsfc = dev_rand((size_t)D * H, 17); float* Wa = dev_rand((size_t)A * H, 18);
    std::vector<int> hw(M), hp(M);
    for (int i = 0; i < M; ++i) { hw[i] = (i * 7919 + 13) % V; hp[i] = (i / 5) * 5 + (i * 3) % 5; if (hp[i] >= M) hp[i] = i; }
    int *widx, *pidx; CK(hipMalloc(&widx, M * 4)); CK(hipMalloc(&pidx, M * 4));
    CK(hipMemcpy(widx, hw.data(), M * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(pidx, hp.data(), M * 4, hipMemcpyHostToDevice));
    float* C; CK(hipMalloc(&C, (size_t)8 * M * V * sizeof(float)));
    double tot_ms = 0, tot_fl = 0;
    {
        Builder b(slots, min_iters, tm, tn);
        const float* Wi[3] = {Wih1, Wis, Wig}; const float* Wh[3] = {Whh1, Whs, nullptr}; const int N[3] = {4 * H, H, H}, off[3] = {0, 4 * H, 5 * H};
        for (int i = 0; i < 3; ++i) {
            GemmProb& p = b.prob(M, N[i], C + off[i], 6 * H);
            Builder::seg(p, h2, H, pidx, Wi[i], in1, H); Builder::seg(p, emb, E, widx, Wi[i] + H + D, in1, E);
            if (Wh[i]) Builder::seg(p, h1, H, pidx, Wh[i], H, H);
        }
        int ns = b.finish(); double ms = time_it(b, 20);
        printf("S1  nslab %d G %5d  %8.1f us  %6.1f TF/s\n", ns, b.a.G, ms * 1e3, b.flops() / ms / 1e9); tot_ms += ms; tot_fl += b.flops();
    }
    {
        Builder b(slots, min_iters, tm, tn);
        GemmProb& p0 = b.prob(M, H, C, H + A); Builder::seg(p0, h1, H, nullptr, Whs, H, H);
        GemmProb& p1 = b.prob(M, A, C + H, H + A); Builder::seg(p1, h1, H, nullptr, Wa, H, H);
        GemmProb& p2 = b.prob(M, D, C + 8 * M * (H + A), D + A); Builder::seg(p2, h2, H, nullptr, Wsfc, H, H);
        GemmProb& p3 = b.prob(M, A, C + 8 * M * (H + A) + D, D + A); Builder::seg(p3, h2, H, nullptr, Wa, H, H);
        int ns = b.finish(); double ms = time_it(b, 20);
        printf("S2  nslab %d G %5d  %8.1f us  %6.1f TF/s\n", ns, b.a.G, ms * 1e3, b.flops() / ms / 1e9); tot_ms += ms; tot_fl += b.flops();
    }
    {
        Builder b(slots, min_iters, tm, tn);
        GemmProb& p0 = b.prob(M, 4 * H, C, 4 * H);
        Builder::seg(p0, h1, H, nullptr, Wih2, in2, H); Builder::seg(p0, att, D, nullptr, Wih2 + H, in2, D); Builder::seg(p0, h2, H, pidx, Whh2, H, H);
        GemmProb& p1 = b.prob(M, A, C + 8 * M * 4 * H, A); Builder::seg(p1, h1, H, nullptr, Wa, H, H);
        int ns = b.finish(); double ms = time_it(b, 20);
        printf("S5  nslab %d G %5d  %8.1f us  %6.1f TF/s\n", ns, b.a.G, ms * 1e3, b.flops() / ms / 1e9); tot_ms += ms; tot_fl += b.flops();
    }
    {
        Builder b(slots, min_iters, tm, tn);
        GemmProb& p0 = b.prob(M, V, C, V); Builder::seg(p0, h2, H, nullptr, Wout, H, H);
        int ns = b.finish(); double ms = time_it(b, 20);
        printf("S6  nslab %d G %5d  %8.1f us  %6.1f TF/s\n", ns, b.a.G, ms * 1e3, b.flops() / ms / 1e9); tot_ms += ms; tot_fl += b.flops();
    }
    if (getenv("GEMM_REPACK")) {
        // round 6: what would other compositions of the step's two wide launches cost?  (all-DMA kernel: run as `GEMM_REPACK=1 tools/gemm_bench 500 256 4 5400 1`)
        // S5 runs a k-aligned plan on 200 of 256 CUs (64 LSTM2 tiles x 3 pieces + 8 att_ga tiles), S6 stream-K ranges over vocabulary + next LSTM1 sums.
        auto run = [&](const char* name, bool aligned, auto fill) {
            if (aligned) setenv("GEMM_PLAN_ALIGNED", "4", 1); else unsetenv("GEMM_PLAN_ALIGNED");
            Builder b(slots, min_iters, tm, tn);
            fill(b);
            int ns = b.finish(); double ms = time_it(b, 20);
            int T = 0;
            if (b.a.aligned) for (int i = 0; i < b.a.nprob; ++i) T = std::max(T, (b.a.p[i].ktiles + b.a.p[i].split - 1) / b.a.p[i].split);
            else T = (b.a.total_iters + b.a.G - 1) / b.a.G;
            printf("%-58s %s nslab %d G %4d  k-tiles per workgroup %3d  %7.1f us  %6.1f TF/s\n", name, b.a.aligned ? "aligned " : "stream-K", ns, b.a.G, T, ms * 1e3, b.flops() / ms / 1e9);
            return ms;
        };
        auto lstm2 = [&](Builder& b) { GemmProb& p0 = b.prob(M, 4 * H, C, 4 * H);
            Builder::seg(p0, h1, H, nullptr, Wih2, in2, H); Builder::seg(p0, att, D, nullptr, Wih2 + H, in2, D); Builder::seg(p0, h2, H, pidx, Whh2, H, H); };
        auto ga = [&](Builder& b) { GemmProb& p1 = b.prob(M, A, C + 8 * (size_t)M * 4 * H, A); Builder::seg(p1, h1, H, nullptr, Wa, H, H); };
        auto vocab = [&](Builder& b) { GemmProb& p0 = b.prob(M, V, C, V); Builder::seg(p0, h2, H, nullptr, Wout, H, H); };
        float* C2 = C + 2 * (size_t)M * V;
        const float* Wi[3] = {Wih1, Wis, Wig}; const float* Wh[3] = {Whh1, Whs, nullptr}; const int N3[3] = {4 * H, H, H}, off3[3] = {0, 4 * H, 5 * H};
        auto l1 = [&](Builder& b, int i, bool with_h2, bool with_h1) { GemmProb& p = b.prob(M, N3[i], C2 + off3[i], 6 * H);
            if (with_h2) Builder::seg(p, h2, H, nullptr, Wi[i], in1, H);
            if (with_h1 && Wh[i]) Builder::seg(p, h1, H, nullptr, Wh[i], H, H); };
        const double a5 = run("A5 now: LSTM2 | att_ga", true, [&](Builder& b) { lstm2(b); ga(b); });
        const double a6 = run("A6 now: vocab | gates(h2+h1) | is(h2+h1) | ig(h2)", false, [&](Builder& b) { vocab(b); l1(b, 0, true, true); l1(b, 1, true, true); l1(b, 2, true, false); });
        const double b5 = run("B5: LSTM2 | gates-h1", true, [&](Builder& b) { lstm2(b); l1(b, 0, false, true); });
        const double b6 = run("B6: vocab | gates-h2 | is(h2+h1) | ig(h2)   [att_ga omitted]", false, [&](Builder& b) { vocab(b); l1(b, 0, true, false); l1(b, 1, true, true); l1(b, 2, true, false); });
        const double b6a = run("B6a: the same, k-aligned", true, [&](Builder& b) { vocab(b); l1(b, 0, true, false); l1(b, 1, true, true); l1(b, 2, true, false); });
        const double c6 = run("C6: vocab | gates-h2 | is-h2 | ig-h2 (uniform K = H)", true, [&](Builder& b) { vocab(b); l1(b, 0, true, false); l1(b, 1, true, false); l1(b, 2, true, false); });
        const double c5 = run("C5: LSTM2 | gates-h1 | hs-h1", true, [&](Builder& b) { lstm2(b); l1(b, 0, false, true); l1(b, 1, false, true); });
        const double c5s = run("C5s: the same, stream-K", false, [&](Builder& b) { lstm2(b); l1(b, 0, false, true); l1(b, 1, false, true); });
        const double d5 = run("D5: LSTM2 | gates-h1 | hs-h1 | att_ga, stream-K", false, [&](Builder& b) { lstm2(b); l1(b, 0, false, true); l1(b, 1, false, true); ga(b); });
        printf("now A5 + A6 = %.1f us;  B5 + B6 = %.1f us;  B5 + B6a = %.1f us;  C5 + C6 = %.1f us;  C5s + C6 = %.1f us;  D5 + C6 = %.1f us\n",
               (a5 + a6) * 1e3, (b5 + b6) * 1e3, (b5 + b6a) * 1e3, (c5 + c6) * 1e3, (c5s + c6) * 1e3, (d5 + c6) * 1e3);
        return 0;
    }
    if (getenv("GEMM_LONG")) {
        const int KL = 16000, NL_ = 4096;
        float* Al = dev_rand((size_t)M * KL, 31); float* Wl = dev_rand((size_t)NL_ * KL, 32);
        Builder b(slots, min_iters, tm, tn);
        GemmProb& p0 = b.prob(M, NL_, C, NL_); Builder::seg(p0, Al, KL, nullptr, Wl, KL, KL);
        int ns = b.finish(); double ms = time_it(b, 10);
        printf("LONG K=%d N=%d nslab %d G %5d  %8.1f us  %6.1f TF/s\n", KL, NL_, ns, b.a.G, ms * 1e3, b.flops() / ms / 1e9);
    }
    printf("step GEMMs M=%d slots %d min_iters %d tile %dx%d: %.1f us, %.1f TF/s\n", M, slots, min_iters, 64 * tm, 64 * tn, tot_ms * 1e3, tot_fl / tot_ms / 1e9);
    return 0;
}

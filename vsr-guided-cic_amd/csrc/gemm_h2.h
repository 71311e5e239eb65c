// "f16x2": fp32-accurate "NT" GEMM on the fp16 matrix cores with THREE MFMAs per product and weights that need no split at all.
//
//   C_p[m][n] = sum over segments g, k:  A_pg[row_g(m)][k] * W_pg[n][k]        fp32 operands, fp32 accumulation, S partial slabs
//
// Every fp32 operand element x of a tensor with a known bound |x| <= b is scaled by a power of two, s = 2^e with b s in (2^14, 2^15]
// (exact), and written as two fp16 terms
//     hi = f16(x s),   lo = f16(x s - hi)            (round to nearest; x s - hi is exact in fp32)
// so that x s = hi + lo up to 2^-22 |x s| (and up to 2^-25 absolute where lo is a subnormal).  A product a.w is accumulated in
// fp32 as  lo.hi + hi.lo + hi.hi  (three v_mfma_f32_*_f16, smallest first); the dropped lo.lo is <= 2^-22 of the product.  The
// accumulator is in units of 2^S, S = e_a + e_w, the same for every segment of a problem (the A scale of a segment is chosen as
// S - e_w: A is split inside the kernel, so its scale is free below the bound); the epilogue multiplies by 2^-S (exact).
// Measured against fp64 next to the exact fma chain and the six-MFMA bf16 split (tools/gemm_bench, tests/test_gpu_h2.py): the error is
// the accumulation's, not the split's - three accumulator roundings per 16 k's instead of six (bf16x3) or sixteen (fma chain).
//
// WEIGHTS come as an fp16-pair IMAGE with the byte geometry of the fp32 matrix itself (4 bytes per element, refreshed per weight
// version: vsr_refresh_h2_weights): the 8 elements [n][8 g .. 8 g + 7] occupy the 32 bytes the fp32 values would, as
// [hi x 8 | lo x 8].  So a weight window is addressed exactly like the fp32 window, a 16-byte chunk is a ready MFMA operand
// fragment, and nothing is converted in the kernels: the movers of the wide kernel only MOVE weights (the round-3 kernels spent
// 5 vector instructions per MFMA on the split), and the streaming kernel puts them from global memory straight into the matrix
// core.  Only A (activations: a third of the wide tile's rows, a few rows of the streaming kernel's) is split in the kernel.
//
// Two kernels, both with the work decompositions of gemm_f32.h (stream-K ranges / k-aligned pieces, slab outputs):
//   gemm_nt_h2_kernel<TM, TN, NW>  16 waves (8 multiply, 8 move), 128 x 256 / 128 x 128 x 32 tiles, W by LDS-DMA into a ring of NW stages
//   gemm_nt_h2s_kernel<MT, NS>  4 waves, <= 128 rows: W global -> register -> MFMA B operand, A staged through LDS: gemm_x3s.h's
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_x3.h"
#include "gemm_x3s.h"

namespace vsr {

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

constexpr int H2_BK = 32;
constexpr int H2_ROW = 32;                                // fp16 elements per LDS row (64 bytes, unpadded, XOR-swizzled chunks)
constexpr int H2_THREADS = 1024;
constexpr int H2_TOP = 15;                                // a tensor with bound b gets the exponent e = H2_TOP - ceil(log2 b): b 2^e in (2^14, 2^15]

__device__ __forceinline__ float h2_pow2(int e) { return __int_as_float((127 + e) << 23); }      // -126 <= e <= 127

// (a, b) scaled by sc -> packed fp16 pairs (hi, lo)
__device__ __forceinline__ void split_h2(float a, float b, float sc, uint32_t& hi, uint32_t& lo) {
    const f32x2_t xs = {a * sc, b * sc};
    const f16x2_t h = __builtin_convertvector(xs, f16x2_t);
    const f32x2_t r = {xs.x - (float)h.x, xs.y - (float)h.y};          // exact
    const f16x2_t l = __builtin_convertvector(r, f16x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, l);
}

// ---- images and bounds --------------------------------------------------------------------------------------------------
// max |x| over n floats into *out (bit pattern of a non-negative float: unsigned order = float order; *out zeroed by the caller)
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            for (long long j = i; j < n; ++j) m = fmaxf(m, fabsf(x[j]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (!(m == m)) m = __int_as_float(0x7f800000);       // a NaN anywhere: treated as an infinite bound (exponent clamps, NaN stays NaN)
        if (__float_as_uint(m) > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, __float_as_uint(m));      // (the slot only grows: skip the same-address atomic when there is nothing to add)
    }
}

// max over rows of (sum_k |W[n][k]| + |bias[n]|): a bound on |W x + bias| for |x| <= 1 (the sentinel vector, step :155)
constexpr int L1MAX_ROWS = 16;                            // rows per workgroup (four per wave): ONE atomic per workgroup - 2 048 same-address atomics were 25 of this kernel's 30 us
__global__ __launch_bounds__(256) void k_row_l1_max(const float* __restrict__ W, const float* __restrict__ bias, int N, int K, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (K & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
    float best = 0.f;
    bool nan = false;
    for (int q = 0; q < L1MAX_ROWS / 4; ++q) {
        const int n = blockIdx.x * L1MAX_ROWS + q * 4 + wave;
        float s = 0.f;
        if (n < N) {
            const float* w = W + (long long)n * K;
            if (vec) {
                // 16 bytes per lane, every load of a 1 024-element span issued before the first use
                float4 v[4];
                for (int k0 = lane * 4; k0 < K; k0 += 1024) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = k0 + 256 * u < K ? *reinterpret_cast<const float4*>(w + k0 + 256 * u) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 4; ++u) s += (fabsf(v[u].x) + fabsf(v[u].y)) + (fabsf(v[u].z) + fabsf(v[u].w));
                }
            } else {
                for (int k = lane; k < K; k += 64) s += fabsf(w[k]);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (n < N) {
            s += bias ? fabsf(bias[n]) : 0.f;
            nan = nan || !(s == s);
            best = fmaxf(best, s);
        }
    }
    if (nan) best = __int_as_float(0x7f800000);
    __shared__ float wm[4];
    if (lane == 0) wm[wave] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (__float_as_uint(m) > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, __float_as_uint(m));      // (the slot only grows: skip the same-address atomic when there is nothing to add)
    }
}

// exponent of a bound: e = H2_TOP - ceil(log2 b), clamped (b = 0: any scale will do)
__device__ __forceinline__ int h2_exp_of(float b) {
    if (!(b > 0.f)) return 0;
    int e2;
    const float m = frexpf(b, &e2);                        // b = m 2^e2, m in [0.5, 1)
    const int c = (m == 0.5f) ? e2 - 1 : e2;               // ceil(log2 b)
    const int e = H2_TOP - c;
    return e < -100 ? -100 : (e > 100 ? 100 : e);
}
// exps[dst[i]] = exponent of max(bounds[a[i]], bounds[b[i]])   (b[i] < 0: one bound)
__global__ void k_h2_exps(const unsigned* __restrict__ bounds, const int* __restrict__ a, const int* __restrict__ b, const int* __restrict__ dst,
                          int n, int* __restrict__ exps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = __uint_as_float(bounds[a[i]]);
    if (b[i] >= 0) v = fmaxf(v, __uint_as_float(bounds[b[i]]));
    exps[dst[i]] = h2_exp_of(v);
}

// fp32 matrix -> fp16-pair image of the same byte geometry: elements [8 g, 8 g + 8) -> [hi x 8 | lo x 8]; n a multiple of 8
__global__ __launch_bounds__(256) void k_f32_to_h2(const float* __restrict__ src, uint32_t* __restrict__ dst, long long n, const int* __restrict__ exps, int slot) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i + 8 > n) return;
    const float sc = h2_pow2(exps[slot]);
    const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
    uint4 hi, lo;
    split_h2(a.x, a.y, sc, hi.x, lo.x); split_h2(a.z, a.w, sc, hi.y, lo.y);
    split_h2(b.x, b.y, sc, hi.z, lo.z); split_h2(b.z, b.w, sc, hi.w, lo.w);
    *reinterpret_cast<uint4*>(dst + i) = hi;
    *reinterpret_cast<uint4*>(dst + i + 4) = lo;
}

// the refresh after an optimizer step as TWO launches over all tensors instead of fifteen pairs (a training step pays for it: fifteen
// k_absmax + fourteen k_f32_to_h2 launches were 0.36 ms of a 10 ms XE step).  Block -> tensor through a compare chain over static
// indices (a run-time index into a by-value argument struct makes hipcc copy the struct to scratch).
constexpr int H2_MT = 16;
struct H2Multi {
    const float* src[H2_MT];
    long long n[H2_MT];         // elements (images: rounded up to 8)
    long long dst_off[H2_MT];   // image offset in floats from the image base (conversion only)
    int blk[H2_MT + 1];         // first block of tensor i; blk[nt] = grid size
    int slot[H2_MT];            // bounds / exponent slot of tensor i
    int nt;
};
__global__ __launch_bounds__(256) void k_absmax_multi(const H2Multi t, unsigned* __restrict__ bounds) {
    const float* x = t.src[0];
    long long n = t.n[0];
    int b0 = 0, b1 = t.blk[1], slot = t.slot[0];
#pragma unroll
    for (int i = 1; i < H2_MT; ++i)
        if (i < t.nt && (int)blockIdx.x >= t.blk[i]) { x = t.src[i]; n = t.n[i]; b0 = t.blk[i]; b1 = t.blk[i + 1]; slot = t.slot[i]; }
    float m = 0.f;
    const long long stride = (long long)(b1 - b0) * 256 * 4;
    long long i = ((long long)(blockIdx.x - b0) * 256 + threadIdx.x) * 4;
    // four independent 16-byte loads per round (a refresh runs once per training step: round 6)
    for (; i + 3 * stride + 3 < n; i += 4 * stride) {
        const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + i + stride);
        const float4 c = *reinterpret_cast<const float4*>(x + i + 2 * stride), d = *reinterpret_cast<const float4*>(x + i + 3 * stride);
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
        m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))), fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)))));
    }
    for (; i < n; i += stride) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(x + i);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
            for (long long j = i; j < n; ++j) m = fmaxf(m, fabsf(x[j]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (!(m == m)) m = __int_as_float(0x7f800000);
        if (__float_as_uint(m) > __atomic_load_n(bounds + slot, __ATOMIC_RELAXED)) atomicMax(bounds + slot, __float_as_uint(m));
    }
}
__global__ __launch_bounds__(256) void k_f32_to_h2_multi(const H2Multi t, uint32_t* __restrict__ img, const int* __restrict__ exps) {
    const float* x = t.src[0];
    long long n = t.n[0], off = t.dst_off[0];
    int b0 = 0, slot = t.slot[0];
#pragma unroll
    for (int i = 1; i < H2_MT; ++i)
        if (i < t.nt && (int)blockIdx.x >= t.blk[i]) { x = t.src[i]; n = t.n[i]; off = t.dst_off[i]; b0 = t.blk[i]; slot = t.slot[i]; }
    const long long i = ((long long)(blockIdx.x - b0) * 256 + threadIdx.x) * 8;
    if (i + 8 > n) return;
    const float sc = h2_pow2(exps[slot]);
    const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(x + i + 4);
    uint4 hi, lo;
    split_h2(a.x, a.y, sc, hi.x, lo.x); split_h2(a.z, a.w, sc, hi.y, lo.y);
    split_h2(b.x, b.y, sc, hi.z, lo.z); split_h2(b.z, b.w, sc, hi.w, lo.w);
    *reinterpret_cast<uint4*>(img + off + i) = hi;
    *reinterpret_cast<uint4*>(img + off + i + 4) = lo;
}

constexpr int H2_DYN0 = 256;        // first "dynamic" slot of the exponent table (see h2_prob_exp)
// S of a problem: the common accumulator exponent = min over its segments of (weight exponent + exponent of A's bound)
__device__ __forceinline__ int h2_prob_exp(const GemmArgs& args, const GemmProb& P) {
    int S = 1 << 20;
#pragma unroll
    for (int sg = 0; sg < 3; ++sg)
        if (sg < P.nseg) {
            // A slots from H2_DYN0 up hold the BOUND itself (bit pattern of a non-negative float, folded in by the kernel that wrote the
            // operand - the gradient operands of the training pass, whose range is only known once they exist)
            const int ia = P.seg[sg].exp_idx >> 16, va = args.exps[ia];
            const int e = args.exps[P.seg[sg].exp_idx & 0xffff] + (ia >= H2_DYN0 ? h2_exp_of(__int_as_float(va)) : va);
            S = e < S ? e : S;
        }
    return S < -120 ? -120 : (S > 120 ? 120 : S);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Wide kernel.  16 waves: waves 0-7 MULTIPLY (WM x WN waves, TM x TN 32x32 tiles each, three MFMAs per product), waves 8-15 MOVE.
// WEIGHTS go global -> LDS directly (global_load_lds_dwordx4: no landing registers, no ds_write, nothing of W in the movers' dependent chain)
// into a ring of NW stages, NW - 1 k-tiles ahead; only A (a third of the tile's bytes) passes through registers to be scaled and split.
// The first version of this kernel staged W through registers like gemm_x3.h; its phase stamps (tools/h2_stamp.py on that version,
// profiles/r04_d_h2_wide_plane_pad_and_phase_stamps.txt) showed a k-tile of ~3 300 cycles of which the multipliers work 765 and the movers'
// chain - wait for the loads 930 | split + LDS stores 880 | issue 320 | barrier skew ~1 100 - is the rest; this version is 2 % faster end to
// end (profiles/r04_g_h2_dma_ab.txt): the k loop is paced by the ~14.5 bytes per clock a CU takes in at the ~1.4 GHz it holds under this
// load (48 KB per k-tile), not by latency - deeper rings (NW = 4) and more tiles in flight change nothing.
//   * W stage: BN rows of 128 bytes = 8 chunks of 16 bytes, logical chunk c = 4 plane + k-group; position p of row R holds chunk
//     p ^ ((R >> 1) & 7) (conflict-free for the ds_read_b128 lane groups at a 128-byte row stride).  One DMA instruction of a wave writes
//     1 KB = 8 whole rows; lane l sends the image bytes of (row l >> 3, chunk (l & 7) ^ swizzle): every 128-byte line is read whole.
//   * A: two fp16 planes of 128 rows x 64 bytes per buffer, chunk c of row r at c ^ ((r >> 2) & 3), double buffered, register-staged two
//     k-tiles ahead (asynchronous loads with hand-counted waits: gemm_bf16.h's rules).
//   * per k-tile j the movers: [issue W(j + NW - 1) into the stage the barrier has just freed, A(j + 2) into the register set stored last
//     k-tile] [s_waitcnt vmcnt: A(j + 1) and every older request - W(j + 1) among them - have landed] [A(j + 1) -> split -> its buffer]
//     [barrier].  K tails read a 16-byte block of zeros instead of the image.  The two instances of the period (register sets swapped)
//     ALTERNATE STATICALLY: selected by a run-time test, the asynchronously loaded sets flow through a control-flow merge and hipcc
//     moves them before they have landed (wrong sums and wild addresses: found the hard way, twice).
// LDS: 32 KB of A + NW x 32 KB (BN = 256) / NW x 16 KB (BN = 128): 128 / 80 KB at NW = 3.  One barrier per k-tile; LDS-staged 16-byte
// epilogue stores scaled by 2^-S.
__device__ float g_h2_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// one LDS-DMA request (tools/gemm_dma_variant.h's): every lane sends its 16 bytes at gsrc to LDS byte address lds_dst + 16 * lane
__device__ __forceinline__ void h2_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int TM, int TN, int NW = 3>
__global__ __launch_bounds__(H2_THREADS)
void gemm_nt_h2_kernel(const GemmArgs args) {
    constexpr int WM = 4 / TM, WN = 8 / WM;
    constexpr int BM = 128, BN = 32 * TN * WN, BK = H2_BK;
    static_assert(32 * TM * WM == BM, "tile shape");
    constexpr int APLANE = BM * H2_ROW;                   // fp16 elements per A plane
    constexpr int ABUF = 2 * APLANE;                      // hi | lo
    constexpr int WSTAGE = BN * 64;                       // fp16 elements per W stage (BN rows x 128 bytes)
    constexpr int L = 2 + BN / 64;                        // vector-memory requests per mover thread and k-tile: 2 A loads + BN / 64 DMAs
    static_assert((2 * ABUF + NW * WSTAGE) * 2 <= 163840, "LDS");
    __shared__ __attribute__((aligned(1024))) uint16_t smem[2 * ABUF + NW * WSTAGE];
    uint16_t* const sW = smem + 2 * ABUF;

    const int G = args.G;
    const int g = gemm_wg_of_block(args);
    if (g >= G) return;
    const GemmRange rg = gemm_range(args, g);
    const int it0 = rg.it0, it1 = rg.it1;
    if (it0 >= it1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const bool mover = wave >= 8;
    const int r = lane & 31, hh = lane >> 5;

    // accumulator exponent of every problem of the launch, once, packed as four signed bytes of ONE register (static indices into the
    // argument struct: a run-time problem index made hipcc copy the whole struct to scratch; and a 4-element array of them went to
    // scratch itself, whose loads share vmcnt with the asynchronous tile loads)
    unsigned pS = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) pS |= (unsigned)((i < args.nprob ? h2_prob_exp(args, args.p[i]) : 0) & 0xff) << (8 * i);
    auto exp_of_prob = [&](int p) __attribute__((always_inline)) { return (int)(pS << (24 - 8 * p)) >> 24; };

    int c_prob = 0, c_tile = 0, c_left = 0, c_piece = 0;
    bool c_last = false;
    auto decode = [&](int it) __attribute__((always_inline)) {
        if (args.aligned) {
            c_prob = rg.prob; c_tile = rg.tile; c_piece = rg.piece;
            c_left = it1 - it;
            c_last = rg.piece == rg.split - 1;
            return it - (args.p[rg.prob].it_begin + rg.tile * args.p[rg.prob].ktiles);
        }
        int p = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (i < args.nprob && it >= args.p[i].it_begin) p = i;
        const GemmProb& P = args.p[p];
        const int local = it - P.it_begin;
        c_prob = p;
        c_tile = local / P.ktiles;
        const int kt = local - c_tile * P.ktiles;
        const int tile_base = it - kt;
        const int g_first = (int)((((long long)tile_base + 1) * G - 1) / args.total_iters);
        c_piece = g - g_first;
        const int rem = P.ktiles - kt;
        c_left = rem < it1 - it ? rem : it1 - it;
        c_last = (c_left == rem);
        return kt;
    };

    // Epilogue of one tile piece, staged band by band in the W stage the multipliers have just finished with (32 rows x BN floats = one
    // stage exactly).  Every request in flight is waited for first: the stage may be the target of nothing then, and stores share vmcnt.
    constexpr int ST_LD = BN;
    static_assert(32 * ST_LD * 4 <= WSTAGE * 2, "staging band must fit one W stage");
    f32x16 acc[TM][TN];
    auto flush = [&](auto MULT, float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const float unscale = h2_pow2(-exp_of_prob(c_prob));
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? P.nslab - 1 - c_piece : 0;
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;
        constexpr int RPP = H2_THREADS / TPR;
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
        const int wm = wave / WN, wn = wave % WN;
        wait_loads<0>();
        __syncthreads();                                   // ... for every wave's requests: nothing lands in `stage` from here on
#pragma unroll
        for (int band = 0; band < BM / 32; ++band) {
            if (m0 + band * 32 >= P.M) break;
            if constexpr (decltype(MULT)::value) {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
                    if (wm * TM + ti == band) {
#pragma unroll
                        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                stage[((e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e] * unscale;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < (32 + RPP - 1) / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;
                const int m = m0 + band * 32 + sr;
                if (sr < 32 && m < P.M && n < P.N) {
                    const float4 v = *reinterpret_cast<const float4*>(stage + sr * ST_LD + c4);
                    float* dst = C + (long long)m * P.ldc + n;
                    if (vec_ok && n + 3 < P.N) {
                        *reinterpret_cast<float4*>(dst) = v;
                        for (int x = 1; x <= extra; ++x)
                            *reinterpret_cast<float4*>(dst + (long long)x * P.slab_stride) = make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        const float vv[4] = {v.x, v.y, v.z, v.w};
                        for (int q = 0; q < 4; ++q)
                            if (n + q < P.N) {
                                dst[q] = vv[q];
                                for (int x = 1; x <= extra; ++x) dst[(long long)x * P.slab_stride + q] = 0.f;
                            }
                    }
                }
            }
            __syncthreads();
        }
        wait_loads<0>();
    };
    // k-tile number since it0 (its A buffer and A register set are j & 1, its W stage j % NW); end of a k-tile for BOTH kinds of waves.
    // A flush drains every wave's request queue first and uses the stage just multiplied from: the tiles already prefetched stay where
    // they are (the other stages, the A registers), only the queue is empty afterwards - the counted waits then return at once.
    int j = 0, it = it0, ws = 0;                           // ws = j % NW
    auto end_of_ktile = [&](auto MULT) __attribute__((always_inline)) {
        ++it;
        const bool piece_done = --c_left == 0;
        __syncthreads();
        if (piece_done) {
            flush(MULT, reinterpret_cast<float*>(sW + ws * WSTAGE));
            if (it < it1) decode(it);
        }
        ++j;
        ws = ws + 1 == NW ? 0 : ws + 1;
        return piece_done;
    };

    const int kt0 = decode(it0);

    if (mover) {
        // ================================================================================================ movers
        const int ptid = tid - 512, mw = wave - 8;
        const int lrow = ptid >> 3, lk = (ptid & 7) * 4;   // A: 8 lanes x 16 bytes cover a row's k-tile, 64 rows per pass
        constexpr int LA = BM / 64, LB = BN / 64;          // A loads / W DMAs per thread and k-tile
        // W DMA: this lane's place in a 1 KB block = (row lane >> 3, position lane & 7) -> logical chunk -> image bytes
        const int wr8 = lane >> 3;
        const int wc = (lane & 7) ^ ((4 * (mw & 1) + (lane >> 4)) & 7);      // (R >> 1) & 7 for R = 8 (mw + 8 i) + (lane >> 3): the same for every i
        const int wg8 = 8 * (wc & 3);                      // first k of the chunk's group inside the k-tile
        const int wboff = (wc & 3) * 8 + (wc >> 2) * 4;    // its offset in the row's k-tile, in floats (32 bytes per group, the lo half 16 bytes in)
        const unsigned lds_w0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)sW)
                              + (unsigned)__builtin_amdgcn_readfirstlane(mw) * 1024u;
        // TWO cursors walk the k-tiles of the range: W runs NW - 1 k-tiles ahead, A two.  (Scalars only; the pointers a role derives
        // from its cursor are recomputed when the cursor enters a segment.)
        struct Cursor {
            int prob, tile, tile_left, seg, seg_left, k, K;
            bool fresh;
            __device__ __forceinline__ void open(const GemmArgs& a, int prob_, int tile_, int kt) {
                prob = prob_; tile = tile_;
                const GemmProb& P = a.p[prob_];
                tile_left = P.ktiles - kt;
                int sg = 0;
                while (sg < P.nseg - 1 && kt >= (P.seg[sg].K + H2_BK - 1) / H2_BK) { kt -= (P.seg[sg].K + H2_BK - 1) / H2_BK; ++sg; }
                seg = sg; K = P.seg[sg].K; k = kt * H2_BK; seg_left = (K + H2_BK - 1) / H2_BK - kt; fresh = true;
            }
            __device__ __forceinline__ void settle(const GemmArgs& a) {      // stand on a k-tile: cross segment / tile / problem boundaries
                if (tile_left == 0) {
                    if (tile + 1 < a.p[prob].tiles_m * a.p[prob].tiles_n) open(a, prob, tile + 1, 0);
                    else open(a, prob + 1, 0, 0);
                } else if (seg_left == 0) {
                    ++seg; K = a.p[prob].seg[seg].K; k = 0; seg_left = (K + H2_BK - 1) / H2_BK; fresh = true;
                }
            }
            __device__ __forceinline__ void next() { k += H2_BK; --seg_left; --tile_left; }
        };
        Cursor cw, ca;
        const float* pa[LA];
        const float* pbW[LB];
        float a_sc = 1.f;
        auto issue_w = [&](int stage) __attribute__((always_inline)) {       // DMA the W cursor's k-tile into `stage`
            cw.settle(args);
            if (cw.fresh) {
                const GemmProb& P = args.p[__builtin_amdgcn_readfirstlane(cw.prob)];
                const GemmSeg& S = P.seg[__builtin_amdgcn_readfirstlane(cw.seg)];
                const int n0 = (cw.tile / P.tiles_m) * BN;
#pragma unroll
                for (int i = 0; i < LB; ++i) {
                    int n = n0 + 8 * (mw + 8 * i) + wr8;
                    n = n < P.N ? n : P.N - 1;
                    pbW[i] = S.W + (long long)n * S.ldw + wboff;       // the image has the fp32 matrix's byte geometry
                }
                cw.fresh = false;
            }
            const bool in = cw.k + wg8 < cw.K;             // K is a multiple of 8: a group is inside or outside as a whole
            const unsigned base = lds_w0 + (unsigned)__builtin_amdgcn_readfirstlane(stage) * (unsigned)(WSTAGE * 2);
#pragma unroll
            for (int i = 0; i < LB; ++i) h2_glds16(in ? (const void*)(pbW[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(8 * i) * 1024u);
            cw.next();
        };
        f32x4_t ra[2][LA];
        bool stla[2] = {false, false};
        float ssc[2] = {1.f, 1.f};
        auto issue_a = [&](auto S_) __attribute__((always_inline)) {         // the A cursor's k-tile into register set S_
            constexpr int s_ = decltype(S_)::value;
            ca.settle(args);
            if (ca.fresh) {
                const GemmProb& P = args.p[__builtin_amdgcn_readfirstlane(ca.prob)];
                const GemmSeg& S = P.seg[__builtin_amdgcn_readfirstlane(ca.seg)];
                const int m0 = (ca.tile % P.tiles_m) * BM;
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    int m = m0 + lrow + 64 * i;
                    m = m < P.M ? m : P.M - 1;
                    const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
                    pa[i] = S.A + row * S.lda + lk;
                }
                a_sc = h2_pow2(exp_of_prob(ca.prob) - args.exps[S.exp_idx & 0xffff]);      // <= the exponent of A's bound: no overflow
                ca.fresh = false;
            }
            const bool tail = !(ca.k + lk < ca.K);
            const int ko = tail ? 0 : ca.k;
#pragma unroll
            for (int i = 0; i < LA; ++i) async_load16(ra[s_][i], pa[i] + ko);
            stla[s_] = tail; ssc[s_] = a_sc;
            ca.next();
        };
        auto store_a = [&](auto S_, int b) __attribute__((always_inline)) {
            constexpr int s_ = decltype(S_)::value;
            uint16_t* buf = smem + b * ABUF;
            const float sc = ssc[s_];
#pragma unroll
            for (int i = 0; i < LA; ++i) landed(ra[s_][i]);
#pragma unroll
            for (int i = 0; i < LA; ++i) {
                const int R = lrow + 64 * i;
                f32x4_t v = ra[s_][i];
                if (stla[s_]) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
                uint32_t h0, l0, h1, l1;
                split_h2(v.x, v.y, sc, h0, l0);
                split_h2(v.z, v.w, sc, h1, l1);
                const int pos = R * H2_ROW + 8 * ((lk >> 3) ^ ((R >> 2) & 3)) + (lk & 4);
                *reinterpret_cast<uint2*>(buf + pos) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(buf + APLANE + pos) = make_uint2(l0, l1);
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // prologue: W(0 .. NW - 2), A(0), A(1); A(0) -> buffer 0
        cw.open(args, c_prob, c_tile, kt0);
        ca = cw;
        {
            int st = 0;
            for (int n = 0; n < NW - 1 && it0 + n < it1; ++n) { issue_w(st); ++st; }
        }
        issue_a(S0{});
        if (it0 + 1 < it1) { issue_a(S1{}); wait_loads<LA>(); } else wait_loads<0>();      // A(0) and every W request before it
        store_a(S0{}, 0);
        __syncthreads();                                   // k-tile 0 is ready
        // k-tile j: issue W(j + NW - 1) into the stage freed by the last barrier and A(j + 2) into the set stored last k-tile; wait for
        // A(j + 1) - everything older, W(j + 1) included, has then landed; A(j + 1) -> split -> buffer (j + 1) & 1; barrier
        auto period = [&](auto SN /* register set of k-tile j + 1 */, auto SP /* ... of k-tiles j and j + 2 */) __attribute__((always_inline)) {
            const bool more_w = it + NW - 1 < it1, more_a = it + 2 < it1;
            int st = ws + NW - 1;
            st = st >= NW ? st - NW : st;
            if (more_w) issue_w(st);
            if (more_a) issue_a(SP);
            if (it + 1 < it1) {
                if (more_w && more_a) wait_loads<LB + LA>();
                else if (more_w) wait_loads<LB>();
                else if (more_a) wait_loads<LA>();
                else wait_loads<0>();
                store_a(SN, (j + 1) & 1);
            }
            end_of_ktile(std::false_type{});
        };
        // (static alternation: see the header comment)
        while (it < it1) {
            period(S1{}, S0{});                            // j even
            if (it < it1) period(S0{}, S1{});              // j odd
        }
    } else {
        // ================================================================================================ multipliers
        const int wm = wave / WN, wn = wave % WN;
        const int swz = (r >> 2) & 3;                      // A planes: rows 32 t + r of every subtile share it
        const int wsw = (r >> 1) & 7;                      // W stage: ((32 t + r) >> 1) & 7
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
        };
        zero_acc();
        __syncthreads();
        while (it < it1) {
            const uint16_t* a_row = smem + (j & 1) * ABUF + (wm * (32 * TM) + r) * H2_ROW;
            const uint16_t* b_row = sW + ws * WSTAGE + (wn * (32 * TN) + r) * 64;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int ch = 8 * ((2 * kk + hh) ^ swz);
                const int wh = 8 * ((2 * kk + hh) ^ wsw), wl = 8 * ((4 + 2 * kk + hh) ^ wsw);
                f16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * H2_ROW + ch);
                    al[i] = *reinterpret_cast<const f16x8_t*>(a_row + APLANE + i * 32 * H2_ROW + ch);
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wh);
                    bl[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wl);
                }
#define H2D_TERM(X, Y)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
        _Pragma("unroll") for (int jj = 0; jj < TN; ++jj)                                              \
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[i], Y[jj], acc[i][jj], 0, 0, 0);
                H2D_TERM(al, bh)
                H2D_TERM(ah, bl)
                H2D_TERM(ah, bh)
#undef H2D_TERM
            }
            if (end_of_ktile(std::true_type{})) zero_acc();
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Streaming kernel for launches of at most 128 rows (k-aligned pieces only).  Workgroup = 4 waves = 64 NS columns of W over one k
// piece; lane (c = lane & 15, q = lane >> 4) of a wave loads the 32 image bytes of W[n][k + 8 q .. + 7] for each of its NS strips of 16
// weight rows - two 16-byte loads that ARE the hi and lo B operands of v_mfma_f32_16x16x32_f16 (a wave instruction reads whole
// 128-byte lines of 16 rows) - PF k-tiles ahead, with loads hipcc counts exactly (gemm_x3s.h's rules: no conditional loads in the
// steady state, a raw s_barrier).  A (16 MT rows x 32 k per k-tile, shared by the four waves) is scaled, split and staged through LDS
// as two fp16 planes, double buffered, one barrier per k-tile.  3 MT NS MFMAs per k-tile and wave, term-major over the accumulators.
constexpr int H2S_THREADS = 256;
constexpr int h2s_bn(int NS) { return 4 * 16 * NS; }
constexpr int H2S_PF = 3;

template <int MT, int NS>
__global__ __launch_bounds__(H2S_THREADS)
void gemm_nt_h2s_kernel(const GemmArgs args) {
    constexpr int BK = H2_BK, ROWS = 16 * MT, BN = h2s_bn(NS);
    constexpr int PLANE = ROWS * H2_ROW;
    constexpr int NQ = (ROWS * 8 + H2S_THREADS - 1) / H2S_THREADS;
    constexpr int LASTQ = ROWS * 8 - H2S_THREADS * (NQ - 1);
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * 2 * PLANE];

    const int G = args.G;
    const int g = gemm_wg_of_block(args);
    if (g >= G) return;
    const GemmRange rg = gemm_range(args, g);
    const int it0 = rg.it0, it1 = rg.it1;
    if (it0 >= it1) return;
    const GemmProb& P = args.p[__builtin_amdgcn_readfirstlane(rg.prob)];
    const int n0 = rg.tile * BN;
    const int kt0 = it0 - (P.it_begin + rg.tile * P.ktiles), nkt = it1 - it0;
    const int pS = h2_prob_exp(args, P);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;

    const float* cW = nullptr;
    const float* cA = nullptr;
    int oW[NS], oA[NQ];
    int cK = 0, ck = 0, cseg = -1, cleft = 0;
    float csc = 1.f;
    auto open_segment = [&](int sg, int first_tile) __attribute__((always_inline)) {
        const GemmSeg& S = P.seg[__builtin_amdgcn_readfirstlane(sg)];
        cseg = sg;
        cK = S.K;
        ck = first_tile * BK;
        cleft = (S.K + BK - 1) / BK - first_tile;
        cW = S.W;
        cA = S.A;
        csc = h2_pow2(pS - args.exps[S.exp_idx & 0xffff]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            int n = n0 + 16 * (NS * wave + s) + c;
            n = n < P.N ? n : P.N - 1;
            oW[s] = n * S.ldw + 8 * q;
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qi = tid + H2S_THREADS * i;
            int m = qi >> 3;
            m = m < P.M ? m : P.M - 1;
            const int row = S.a_idx ? S.a_idx[m] : m;
            oA[i] = row * S.lda + (qi & 7) * 4;
        }
    };
    {
        int kt = kt0, sg = 0;
        while (sg < P.nseg - 1 && kt >= (P.seg[sg].K + BK - 1) / BK) { kt -= (P.seg[sg].K + BK - 1) / BK; ++sg; }
        open_segment(sg, kt);
    }
    f32x4_t wq[H2S_PF][NS][2];                              // [0]: hi x 8, [1]: lo x 8 of this lane's k group
    f32x4_t aq[H2S_PF][NQ];
    int rem[H2S_PF];
    float asc[H2S_PF];
    int issued = 0;
    auto load_next = [&](int slot) __attribute__((always_inline)) {
        if (issued < nkt && cleft == 0) open_segment(cseg + 1, 0);
        const int k = issued < nkt ? ck : ck - BK;
        rem[slot] = cK - k;
        asc[slot] = csc;
        const bool z = !(k + 8 * q < cK);                   // K is a multiple of 8: a group is in or out as a whole
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float* src = cW + oW[s] + (z ? -8 * q : k);
            wq[slot][s][0] = *reinterpret_cast<const f32x4_t*>(src);
            wq[slot][s][1] = *reinterpret_cast<const f32x4_t*>(src + 4);
        }
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int lk = ((tid + H2S_THREADS * i) & 7) * 4;
            aq[slot][i] = *reinterpret_cast<const f32x4_t*>(cA + oA[i] + (k + lk < cK ? k : -lk));
        }
        if (issued < nkt) { ck += BK; --cleft; ++issued; }
    };
    auto store_a = [&](int slot, int b) __attribute__((always_inline)) {
        uint16_t* buf = smem + b * 2 * PLANE;
        const float sc = asc[slot];
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int qi = tid + H2S_THREADS * i;
            const int R = qi >> 3, lk = (qi & 7) * 4;
            f32x4_t v = aq[slot][i];
            if (!(lk < rem[slot])) v = f32x4_t{0.f, 0.f, 0.f, 0.f};
            uint32_t h0, l0, h1, l1;
            split_h2(v.x, v.y, sc, h0, l0);
            split_h2(v.z, v.w, sc, h1, l1);
            const int pos = R * H2_ROW + 8 * ((lk >> 3) ^ ((R >> 2) & 3)) + (lk & 4);
            if (i + 1 < NQ || LASTQ == H2S_THREADS || tid < LASTQ) {
                *reinterpret_cast<uint2*>(buf + pos) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(buf + PLANE + pos) = make_uint2(l0, l1);
            }
        }
    };

    f32x4_acc acc[NS][MT];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[s][i] = f32x4_acc{0.f, 0.f, 0.f, 0.f};

    auto multiply = [&](int slot, int b) __attribute__((always_inline)) {
        f16x8_t bh[NS], bl[NS];
        const bool z = !(8 * q < rem[slot]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            f32x4_t w0 = wq[slot][s][0], w1 = wq[slot][s][1];
            if (z) { w0 = f32x4_t{0.f, 0.f, 0.f, 0.f}; w1 = w0; }
            bh[s] = __builtin_bit_cast(f16x8_t, w0);
            bl[s] = __builtin_bit_cast(f16x8_t, w1);
        }
        const uint16_t* base = smem + b * 2 * PLANE;
        constexpr int MH = (MT + 1) / 2;
#define H2S_TERM(AF, BF)                                                                                              \
    _Pragma("unroll") for (int s = 0; s < NS; ++s)                                                                  \
        _Pragma("unroll") for (int i = 0; i < MH; ++i)                                                               \
            if (i0 + i < MT) acc[s][i0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(AF[i], BF[s], acc[s][i0 + i], 0, 0, 0);
#pragma unroll
        for (int i0 = 0; i0 < MT; i0 += MH) {
            f16x8_t ah[MH], al[MH];
#pragma unroll
            for (int i = 0; i < MH; ++i) {
                const int R = 16 * ((i0 + i < MT) ? i0 + i : 0) + c;
                const uint16_t* p = base + R * H2_ROW + 8 * (q ^ ((R >> 2) & 3));
                ah[i] = *reinterpret_cast<const f16x8_t*>(p);
                al[i] = *reinterpret_cast<const f16x8_t*>(p + PLANE);
            }
            H2S_TERM(al, bh) H2S_TERM(ah, bl) H2S_TERM(ah, bh)
        }
#undef H2S_TERM
    };

    auto lds_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

#pragma unroll
    for (int u = 0; u < H2S_PF; ++u) load_next(u);
    store_a(0, 0);
    __syncthreads();

    int j0 = 0;
    for (; j0 + H2S_PF <= nkt; j0 += H2S_PF) {
#pragma unroll
        for (int u = 0; u < H2S_PF; ++u) {
            multiply(u, (j0 + u) & 1);
            store_a((u + 1) % H2S_PF, (j0 + u + 1) & 1);
            load_next(u);
            lds_barrier();
        }
    }
#pragma unroll
    for (int u = 0; u < H2S_PF - 1; ++u) {
        const int j = j0 + u;
        if (j < nkt) {
            multiply(u, j & 1);
            if (j + 1 < nkt) store_a((u + 1) % H2S_PF, (j + 1) & 1);
            lds_barrier();
        }
    }

    {
        const float unscale = h2_pow2(-pS);
        float* C = P.C + (long long)rg.piece * P.slab_stride;
        const int extra = rg.piece == rg.split - 1 ? P.nslab - rg.split : 0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = n0 + 16 * (NS * wave + s) + c;
            if (n < P.N) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int m = 16 * i + 4 * q + e;
                        if (m < P.M) {
                            float* dst = C + (long long)m * P.ldc + n;
                            *dst = acc[s][i][e] * unscale;
                            for (int x = 1; x <= extra; ++x) dst[(long long)x * P.slab_stride] = 0.f;
                        }
                    }
            }
        }
    }
}

}  // namespace vsr

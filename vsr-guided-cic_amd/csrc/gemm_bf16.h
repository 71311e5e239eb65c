// bf16 THROUGHPUT mode of the grouped stream-K "NT" GEMM (BASELINE configs[3] names bf16).  Never the parity
// path: fp32 (gemm_f32.h) stays the default and the mode every token / 1e-4 claim is made in.
//
//   C_p[m][n] = sum over segments g, k:  bf16(A_pg[row_g(m)][k]) * W16_pg[n][k]       fp32 accumulate, S partial slabs
//
// Operands: A (activations, gradients) stays fp32 in memory and is rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) on its way into LDS; W is a bf16 COPY in memory (weights: refreshed by vsr_refresh_bf16_weights after
// every optimizer step from the fp32 master weights; transposed weights / transposed activations of the backward pass:
// written as bf16 by the transposing kernels).  Matrix instruction: v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).
//
// At that rate the k loop is bound by the bytes a CU can pull from L2 into LDS (measured ~22 GB/s per CU in the fp32 kernel,
// DESIGN.md section 4), so the tile is chosen for flop per loaded byte, not for the matrix pipe: 128 (M) x 256 (N) x 32 with
// 8 waves (2 x 4, 64 x 64 per wave) loads 16 KB of fp32 A + 16 KB of bf16 W per 2.1 Mflop = 65 flop / byte, 3x the fp32
// 128x64 tile.  LDS rows are 32 bf16 + 8 pad = 80 bytes: the ds_read_b128 lane groups of the 32x32x16 operand map
// (lane r = l & 31 reads row r, k = 8 (l >> 5) .. +7) hit 64 distinct banks.  Stream-K decomposition, slab outputs and
// cursor are those of gemm_nt_f32_kernel; the epilogue stores accumulator registers directly (128-byte runs per half wave).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gemm_f32.h"

namespace vsr {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

constexpr int B16_ROW = 72;     // bf16 elements per LDS row: 64 + 8 pad = 144 bytes = 36 dwords (the fp32 kernel's conflict-free row stride)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// fp32 -> bf16 copy (weights after an optimizer step), n multiple of 2 handled with a scalar tail
__global__ void k_f32_to_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, long long n) {
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 7 < n) {
        const float4 a = *reinterpret_cast<const float4*>(src + i), b = *reinterpret_cast<const float4*>(src + i + 4);
        uint4 o;
        o.x = pack_bf16(a.x, a.y); o.y = pack_bf16(a.z, a.w); o.z = pack_bf16(b.x, b.y); o.w = pack_bf16(b.z, b.w);
        *reinterpret_cast<uint4*>(dst + i) = o;
    } else {
        for (long long j = i; j < n; ++j) dst[j] = (uint16_t)(pack_bf16(src[j], 0.f) & 0xffffu);
    }
}

constexpr int B16_BK = 64;                                // k-tile: 64 bf16 = one 128-byte line per W row, two lines per fp32 A row

template <int WM, int WN, int TM, int TN>      // waves WM x WN, wave tile (32 TM) x (32 TN)
__global__ __launch_bounds__(64 * WM * WN)
void gemm_nt_bf16w_kernel(const GemmArgs args) {
    constexpr int NT = 64 * WM * WN;
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    constexpr int BK = B16_BK;
    constexpr int RPA = NT / 16, RPB = NT / 8;            // rows per load pass: 16 lanes x float4 / 8 lanes x 16 B cover a row's k-tile
    constexpr int LA = BM / RPA, LB = BN / RPB;
    static_assert(BM % RPA == 0 && BN % RPB == 0, "tile shape");
    constexpr int BUF = (BM + BN) * B16_ROW;              // bf16 elements per k buffer
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];
    auto sA = [&](int buf) { return smem + buf * BUF; };
    auto sB = [&](int buf) { return smem + buf * BUF + BM * B16_ROW; };

    const int G = args.G;
    const int g = (blockIdx.x & 7) * ((G + 7) >> 3) + (blockIdx.x >> 3);
    if (g >= G) return;
    const int it0 = gemm_range_begin(g, args.total_iters, G);
    const int it1 = gemm_range_begin(g + 1, args.total_iters, G);
    if (it0 >= it1) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, hh = lane >> 5;
    const int arow = tid >> 4, ak = (tid & 15) * 4;       // A: 16 lanes per row, 4 fp32 each (full 128-byte lines per wave instruction)
    const int brow = tid >> 3, bk = (tid & 7) * 8;        // W: 8 lanes per row, 8 bf16 each

    // ------------------------------------------------------------------ load cursor (runs TWO iterations ahead of the MFMAs)
    float4 ra[LA];
    uint4 rb[LB];
    const float* pa[LA];
    const uint16_t* pb[LB];
    int l_prob = 0, l_tile = 0, l_tile_left = 0;
    int l_seg = 0, l_seg_left = 0, l_k = 0, l_K = 0;
    auto open_segment = [&](int s, int first_tile) __attribute__((always_inline)) {
        const GemmProb& P = args.p[l_prob];
        const GemmSeg& S = P.seg[s];
        const int m0 = (l_tile % P.tiles_m) * BM, n0 = (l_tile / P.tiles_m) * BN;
        l_seg = s;
        l_K = S.K;
        l_k = first_tile * BK;
        l_seg_left = (S.K + BK - 1) / BK - first_tile;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            int m = m0 + arow + RPA * i;
            m = m < P.M ? m : P.M - 1;
            const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
            pa[i] = S.A + row * S.lda + ak;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            int n = n0 + brow + RPB * i;
            n = n < P.N ? n : P.N - 1;
            pb[i] = reinterpret_cast<const uint16_t*>(S.W) + (long long)n * S.ldw + bk;
        }
    };
    auto open_tile = [&](int prob, int tile, int kt) __attribute__((always_inline)) {
        l_prob = prob;
        l_tile = tile;
        const GemmProb& P = args.p[prob];
        l_tile_left = P.ktiles - kt;
        int s = 0;
        while (s < P.nseg - 1 && kt >= (P.seg[s].K + BK - 1) / BK) { kt -= (P.seg[s].K + BK - 1) / BK; ++s; }
        open_segment(s, kt);
    };
    int ka = 0, kb = 0;                                    // k offsets of the tile being loaded (0 in the K tail: a valid address)
    bool ta = false, tb = false;                           // ... and whether this thread's piece lies in the tail (zeros are stored)
    auto advance = [&]() __attribute__((always_inline)) {
        if (l_tile_left == 0) {
            if (l_tile + 1 < args.p[l_prob].tiles_m * args.p[l_prob].tiles_n) open_tile(l_prob, l_tile + 1, 0);
            else open_tile(l_prob + 1, 0, 0);
        } else if (l_seg_left == 0) {
            open_segment(l_seg + 1, 0);
        }
        ta = !(l_k + ak < l_K);                            // K is a multiple of 8: a 4-float piece is inside or outside as a whole
        tb = !(l_k + bk < l_K);
        ka = ta ? 0 : l_k;
        kb = tb ? 0 : l_k;
        l_k += BK;
        --l_seg_left;
        --l_tile_left;
    };
    auto load_a = [&](int i) __attribute__((always_inline)) { ra[i] = *reinterpret_cast<const float4*>(pa[i] + ka); };
    auto load_b = [&](int i) __attribute__((always_inline)) { rb[i] = *reinterpret_cast<const uint4*>(pb[i] + kb); };
    bool sta = false, stb = false;                         // tail flags of the tile held in the registers
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            uint2 o;
            o.x = pack_bf16(ra[i].x, ra[i].y); o.y = pack_bf16(ra[i].z, ra[i].w);
            if (sta) o = make_uint2(0u, 0u);
            *reinterpret_cast<uint2*>(sA(buf) + (arow + RPA * i) * B16_ROW + ak) = o;
        }
#pragma unroll
        for (int i = 0; i < LB; ++i)
            *reinterpret_cast<uint4*>(sB(buf) + (brow + RPB * i) * B16_ROW + bk) = stb ? make_uint4(0u, 0u, 0u, 0u) : rb[i];
    };

    // ------------------------------------------------------------------ compute-side tile bookkeeping
    int c_prob = 0, c_tile = 0, c_left = 0, c_piece = 0;
    bool c_last = false;
    auto decode = [&](int it) __attribute__((always_inline)) {
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

    // Epilogue of one tile piece.  Accumulator element e of subtile (ti, tj) is row (e & 3) + 8 (e >> 2) + 4 hh, column r (the
    // C/D layout is dtype independent).  Bands of 32 tile rows go through the idle k buffer so that they leave as 16-byte row
    // stores (dword stores from the registers cost the launch 40 % more: 60-80 MB of slab output per launch at M = 500).
    constexpr int ST_LD = BN + 4;
    static_assert(32 * ST_LD * 4 <= BUF * 2, "staging band must fit one k buffer");
    auto flush = [&](const f32x16 (&acc)[TM][TN], float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? args.nslab - 1 - c_piece : 0;     // unused slabs of a finished tile: zeros
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;                        // threads per staged row
        constexpr int RPP = NT / TPR;                      // rows per store pass
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
#pragma unroll
        for (int band = 0; band < BM / 32; ++band) {
#pragma unroll
            for (int ti = 0; ti < TM; ++ti)
                if (wm * TM + ti == band) {
#pragma unroll
                    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            stage[((e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e];
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 32 / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;
                const int m = m0 + band * 32 + sr;
                if (m < P.M && n < P.N) {
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
    };

    // Pipeline: tile i is multiplied from LDS while tile i + 1 sits in registers and the loads of tile i + 2 are in flight.
    //   iteration i:  [registers (tile i+1) -> bf16 -> the other LDS buffer]  [loads of tile i+2 between the MFMAs of tile i]  [barrier]
    // A load therefore has a whole iteration to land, and its issue shares the iteration with the MFMAs instead of preceding them.
    {
        const int kt = decode(it0);
        open_tile(c_prob, c_tile, kt);
    }
    advance();
#pragma unroll
    for (int i = 0; i < LA; ++i) load_a(i);
#pragma unroll
    for (int i = 0; i < LB; ++i) load_b(i);
    sta = ta; stb = tb;
    store_tile(0);
    if (it0 + 1 < it1) {
        advance();
#pragma unroll
        for (int i = 0; i < LA; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < LB; ++i) load_b(i);
        sta = ta; stb = tb;
    }
    __syncthreads();
    int cur = 0;
    for (int it = it0; it < it1;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int n_it = c_left;
        for (int j_it = 0; j_it < n_it; ++j_it, ++it) {
            const bool more = it + 1 < it1, more2 = it + 2 < it1;
            if (more) store_tile(cur ^ 1);                 // tile it + 1 (loaded one iteration ago): nobody reads that buffer now
            if (more2) advance();
            const uint16_t* a_base = sA(cur) + (wm * (32 * TM) + r) * B16_ROW + 8 * hh;
            const uint16_t* b_base = sB(cur) + (wn * (32 * TN) + r) * B16_ROW + 8 * hh;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                if (more2) {                               // this k-step's share of the next-but-one tile's loads
#pragma unroll
                    for (int i = 0; i < LA; ++i)
                        if (i * (BK / 16) / LA == kk) load_a(i);
#pragma unroll
                    for (int i = 0; i < LB; ++i)
                        if (i * (BK / 16) / LB == kk) load_b(i);
                }
                bf16x8_t av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8_t*>(a_base + i * 32 * B16_ROW + kk * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const bf16x8_t*>(b_base + j * 32 * B16_ROW + kk * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more2) { sta = ta; stb = tb; }
            if (more) {
                __syncthreads();
                cur ^= 1;
            }
        }
        flush(acc, reinterpret_cast<float*>(sA(cur ^ 1)));
        if (it < it1) decode(it);
    }
}

}  // namespace vsr

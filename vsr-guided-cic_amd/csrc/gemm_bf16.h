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
// At that rate the k loop is bound by the bytes a CU can pull from L2 into LDS (~52 GB/s per CU at best, tools/gemm_bench feed), so
// the tile is chosen for flop per loaded byte, not for the matrix pipe: 128 (M) x 256 (N) x 64 (K) loads 32 KB of fp32 A + 32 KB
// of bf16 W per 4.2 Mflop = 65 flop / byte, 3x the fp32 128x64 tile.  LDS rows are 64 bf16 + 8 pad = 144 bytes (36 dwords, the
// fp32 kernel's conflict-free stride for the ds_read_b128 lane groups of the 32x32x16 operand map: lane r = l & 31 reads row r,
// k = 8 (l >> 5) .. +7).  16 waves per workgroup - 8 multiply, 8 move data - and asynchronous loads with hand-written waits
// (both explained at the kernel); two work decompositions: stream-K ranges (gemm_plan) or one k-aligned piece of one tile per
// workgroup (gemm_plan_aligned, used whenever the tiles fit the CUs); slab outputs; LDS-staged 16-byte epilogue stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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

// Asynchronous 16-byte global loads.  The k loop keeps TWO tiles in flight per thread (two register sets).  Written with plain
// C++ loads the compiler's waitcnt pass collapses that queue: it merges the "loads issued / not issued" paths of the loop and
// then waits with vmcnt(<= 7) - i.e. for everything but the youngest quarter tile - before every LDS refill (measured: one, two
// or three register sets, wave-specialised or not, all 2.7 us per k-tile).  So the loads are issued with inline asm, which the
// pass does not track, and the waits are written by hand: vmcnt(N) = "all but the N youngest loads have landed" (loads return in
// order).  Rules that keep this safe: every asm-loaded register is re-defined ("+v") right behind the wait, so no use can move
// above it; the compiler's own VMEM operations (row gathers in open_segment, epilogue stores) only ever ADD waits; the epilogue
// drains the queue (vmcnt(0)) before and after its stores, because stores share the counter.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void async_load16(f32x4_t& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void async_load16(u32x4_t& d, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
template <int N> __device__ __forceinline__ void wait_loads() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void landed(f32x4_t& d) { asm volatile("" : "+v"(d)); }
__device__ __forceinline__ void landed(u32x4_t& d) { asm volatile("" : "+v"(d)); }

// 16 waves: waves 0-7 MULTIPLY (2 x 4, 64 x 64 per wave, two per SIMD), waves 8-15 MOVE data (global -> registers -> bf16 -> LDS).
// Why two kinds of waves: a CU pulls at most ~52 GB/s out of L2 (tools/gemm_bench feed: 1.26 us per 64 KB tile whatever the
// queue depth, and only with all of its waves issuing loads), and a wave whose load cannot enter the memory pipeline stalls
// at ISSUE - with one kind of wave the 1.26 us of feed, the 0.5 us LDS refill and the 1.1 us of ds_read + MFMA therefore add up
// (2.7 us per k-tile, measured with one, two and three tiles in flight alike).  The movers take the stall, the multipliers keep
// the matrix pipe busy; they meet at ONE barrier per k-tile, which then costs max(feed + refill, multiply).
// A16K: every segment's A operand comes with a bf16 image (GemmSeg::A16, written by A's producer): the movers load 8 bf16 per
// lane and store them as they are - half the A bytes, no conversion.  Otherwise A is fp32 and converted on the way into LDS.
constexpr int B16_THREADS = 1024;
// TN = 2: workgroup tile 128 x 256; TN = 1: 128 x 128, for launches whose rows fit one m-tile (the recurrent GEMMs of the training
// pass at batch 100, greedy decoding): the number of tiles is then the number of n-tiles (as in gemm_x3.h, round 3)
template <bool A16K, int TN = 2>
__global__ __launch_bounds__(B16_THREADS)
void gemm_nt_bf16w_kernel(const GemmArgs args) {
    constexpr int WN = 4, TM = 2;                          // multipliers: 2 x 4 waves, wave tile 64 x 32 TN
    constexpr int BM = 128, BN = 32 * TN * WN, BK = B16_BK;
    constexpr int LA = BM / 32, LB = BN / 64;              // movers: 512 threads cover 32 A rows / 64 W rows per pass
    constexpr int BUF = (BM + BN) * B16_ROW;               // bf16 elements per k buffer
    __shared__ __attribute__((aligned(16))) uint16_t smem[2 * BUF];
    auto sA = [&](int buf) { return smem + buf * BUF; };
    auto sB = [&](int buf) { return smem + buf * BUF + BM * B16_ROW; };

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

    // ------------------------------------------------------------------ tile bookkeeping (both kinds: the epilogue is shared)
    int c_prob = 0, c_tile = 0, c_left = 0, c_piece = 0;
    bool c_last = false;
    auto decode = [&](int it) __attribute__((always_inline)) {
        if (args.aligned) {                                // one piece of one tile: nothing to search
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

    // Epilogue of one tile piece.  Accumulator element e of subtile (ti, tj) is row (e & 3) + 8 (e >> 2) + 4 hh, column r.  Bands
    // of 32 tile rows are staged in a k buffer by the four multiplier waves that own them and leave as 16-byte row stores issued
    // by all 1024 threads (dword stores from the registers cost the launch 40 % more: 60-80 MB of slab output per launch).
    constexpr int ST_LD = BN + 4;
    static_assert(32 * ST_LD * 4 <= BUF * 2, "staging band must fit one k buffer");
    f32x16 acc[TM][TN];
    auto flush = [&](auto MULT, float* stage) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const int m0 = (c_tile % P.tiles_m) * BM, n0 = (c_tile / P.tiles_m) * BN;
        float* C = P.C + (long long)c_piece * P.slab_stride;
        const int extra = c_last ? P.nslab - 1 - c_piece : 0;     // unused slabs of a finished tile: zeros
        const bool vec_ok = ((P.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.C) & 15) == 0) && ((P.slab_stride & 3) == 0);
        constexpr int TPR = BN / 4;                        // 64 threads per staged row
        constexpr int RPP = B16_THREADS / TPR;             // 16 / 32 rows per store pass
        const int c4 = (tid % TPR) * 4;
        const int n = n0 + c4;
        const int wm = wave / WN, wn = wave % WN;          // (multipliers)
        wait_loads<0>();                                   // stores share vmcnt with the asynchronous loads: start from an empty queue
#pragma unroll
        for (int band = 0; band < BM / 32; ++band) {
            if constexpr (decltype(MULT)::value) {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
                    if (wm * TM + ti == band) {
#pragma unroll
                        for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                            for (int e = 0; e < 16; ++e)
                                stage[((e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e];
                    }
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
        wait_loads<0>();                                   // ... and leave it empty (the hand-written waits count loads only)
    };
    // End of a k-tile for BOTH kinds of waves: the barrier, then - when the tile piece is complete - the epilogue, staged in the
    // buffer the multipliers have just finished with (the movers wrote the OTHER one during this k-tile and touch this one only
    // after the next barrier).
    int cur = 0, it = it0;
    auto end_of_ktile = [&](auto MULT) __attribute__((always_inline)) {
        ++it;
        const bool piece_done = --c_left == 0;
        __syncthreads();
        if (piece_done) {
            flush(MULT, reinterpret_cast<float*>(sA(cur)));
            if (it < it1) decode(it);
        }
        cur ^= 1;
        return piece_done;
    };

    const int kt0 = decode(it0);

    if (mover) {
        // ================================================================================================ movers
        const int ptid = tid - 512;
        const int arow = ptid >> 4, ak = (ptid & 15) * 4;  // A: 16 lanes per row, 4 fp32 each (whole 128-byte lines per wave instruction)
        const int brow = ptid >> 3, bk = (ptid & 7) * 8;   // W: 8 lanes per row, 8 bf16 each
        // A pieces per thread and tile: fp32 - LA loads of 4 floats (16 lanes per row, converted on the way into LDS); bf16 image -
        // LA / 2 loads of 8 bf16 (8 lanes per row, stored as they are)
        constexpr int NA = A16K ? LA / 2 : LA;
        const int arow16 = ptid >> 3, ak16 = (ptid & 7) * 8;
        f32x4_t ra[2][NA];                                 // tile j (counted from it0) lives in register set j & 1
        u32x4_t rb[2][LB];
        bool sta[2] = {false, false}, stb[2] = {false, false};
        const float* pa[NA];
        const uint16_t* pa16[NA];
        const uint16_t* pb[LB];
        int l_prob = 0, l_tile = 0, l_tile_left = 0;
        int l_seg = 0, l_seg_left = 0, l_k = 0, l_K = 0;
        auto open_segment = [&](int sg, int first_tile) __attribute__((always_inline)) {
            const GemmProb& P = args.p[l_prob];
            const GemmSeg& S = P.seg[sg];
            const int m0 = (l_tile % P.tiles_m) * BM, n0 = (l_tile / P.tiles_m) * BN;
            l_seg = sg;
            l_K = S.K;
            l_k = first_tile * BK;
            l_seg_left = (S.K + BK - 1) / BK - first_tile;
            if constexpr (A16K) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int m = m0 + arow16 + 64 * i;
                    m = m < P.M ? m : P.M - 1;
                    const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
                    pa16[i] = S.A16 + row * S.lda + ak16;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    int m = m0 + arow + 32 * i;
                    m = m < P.M ? m : P.M - 1;
                    const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
                    pa[i] = S.A + row * S.lda + ak;
                }
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                int n = n0 + brow + 64 * i;
                n = n < P.N ? n : P.N - 1;
                pb[i] = reinterpret_cast<const uint16_t*>(S.W) + (long long)n * S.ldw + bk;
            }
        };
        auto open_tile = [&](int prob, int tile, int kt) __attribute__((always_inline)) {
            l_prob = prob;
            l_tile = tile;
            const GemmProb& P = args.p[prob];
            l_tile_left = P.ktiles - kt;
            int sg = 0;
            while (sg < P.nseg - 1 && kt >= (P.seg[sg].K + BK - 1) / BK) { kt -= (P.seg[sg].K + BK - 1) / BK; ++sg; }
            open_segment(sg, kt);
        };
        int ka = 0, kb = 0;                                // k offsets of the tile being loaded (0 in the K tail: a valid address)
        bool ta = false, tb = false;                       // ... and whether this thread's piece lies in the tail (zeros are stored)
        auto advance = [&]() __attribute__((always_inline)) {
            if (l_tile_left == 0) {
                if (l_tile + 1 < args.p[l_prob].tiles_m * args.p[l_prob].tiles_n) open_tile(l_prob, l_tile + 1, 0);
                else open_tile(l_prob + 1, 0, 0);
            } else if (l_seg_left == 0) {
                open_segment(l_seg + 1, 0);
            }
            ta = !(l_k + (A16K ? ak16 : ak) < l_K);        // K is a multiple of 8: a piece is inside or outside as a whole
            tb = !(l_k + bk < l_K);
            ka = ta ? 0 : l_k;
            kb = tb ? 0 : l_k;
            l_k += BK;
            --l_seg_left;
            --l_tile_left;
        };
        // issue the loads of the next tile of the range into set S (asynchronous: nothing waits here)
        auto issue = [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            advance();
            if constexpr (A16K) {
#pragma unroll
                for (int i = 0; i < NA; ++i) async_load16(ra[s][i], pa16[i] + ka);
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) async_load16(ra[s][i], pa[i] + ka);
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) async_load16(rb[s][i], pb[i] + kb);
            sta[s] = ta; stb[s] = tb;
        };
        // set S has landed once at most the loads of the other (younger) set are still in flight
        auto landed_set = [&](auto S, bool other_in_flight) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            if (other_in_flight) wait_loads<NA + LB>(); else wait_loads<0>();
#pragma unroll
            for (int i = 0; i < NA; ++i) landed(ra[s][i]);
#pragma unroll
            for (int i = 0; i < LB; ++i) landed(rb[s][i]);
        };
        auto store_tile = [&](auto S, int buf) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            if constexpr (A16K) {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    uint4 o = make_uint4(__float_as_uint(ra[s][i].x), __float_as_uint(ra[s][i].y), __float_as_uint(ra[s][i].z), __float_as_uint(ra[s][i].w));
                    if (sta[s]) o = make_uint4(0u, 0u, 0u, 0u);
                    *reinterpret_cast<uint4*>(sA(buf) + (arow16 + 64 * i) * B16_ROW + ak16) = o;
                }
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    uint2 o;
                    o.x = pack_bf16(ra[s][i].x, ra[s][i].y); o.y = pack_bf16(ra[s][i].z, ra[s][i].w);
                    if (sta[s]) o = make_uint2(0u, 0u);
                    *reinterpret_cast<uint2*>(sA(buf) + (arow + 32 * i) * B16_ROW + ak) = o;
                }
            }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                uint4 o = make_uint4(rb[s][i].x, rb[s][i].y, rb[s][i].z, rb[s][i].w);
                if (stb[s]) o = make_uint4(0u, 0u, 0u, 0u);
                *reinterpret_cast<uint4*>(sB(buf) + (brow + 64 * i) * B16_ROW + bk) = o;
            }
        };
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        // k-tile j:  [wait for tile j+1 (issued one k-tile ago; tile j+2 may stay in flight)] [tile j+1 -> bf16 -> the other buffer]
        //            [issue tile j+3 into the set just emptied] [barrier]
        open_tile(c_prob, c_tile, kt0);
        issue(S0{});                                       // tile 0
        landed_set(S0{}, false);
        store_tile(S0{}, 0);
        if (it0 + 1 < it1) issue(S1{});                    // tile 1
        if (it0 + 2 < it1) issue(S0{});                    // tile 2
        __syncthreads();                                   // buffer 0 is ready
        auto step = [&](auto S) __attribute__((always_inline)) {
            if (it + 1 < it1) {
                landed_set(S, it + 2 < it1);
                store_tile(S, cur ^ 1);
                if (it + 3 < it1) issue(S);
            }
            end_of_ktile(std::false_type{});
        };
        while (it < it1) {
            step(S1{});
            if (it < it1) step(S0{});
        }
    } else {
        // ================================================================================================ multipliers
        const int wm = wave / WN, wn = wave % WN;
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        };
        zero_acc();
        __syncthreads();                                   // buffer 0 is ready
        while (it < it1) {
            const uint16_t* a_base = sA(cur) + (wm * (32 * TM) + r) * B16_ROW + 8 * hh;
            const uint16_t* b_base = sB(cur) + (wn * (32 * TN) + r) * B16_ROW + 8 * hh;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                bf16x8_t av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const bf16x8_t*>(a_base + i * 32 * B16_ROW + kk * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const bf16x8_t*>(b_base + j * 32 * B16_ROW + kk * 16);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
            if (end_of_ktile(std::true_type{})) zero_acc();
        }
    }
}

}  // namespace vsr

// All-DMA variant of the wide f16x2 kernel (gemm_h2.h): BOTH operands arrive as fp16-pair images and go global -> LDS by
// global_load_lds_dwordx4 into one ring of NW stages (BN weight rows + 128 A rows of 128 bytes each, the same chunk swizzle for both);
// the movers do no arithmetic and own no landing registers, so their loop is [issue k-tile j + NW - 1] [counted wait: k-tile j + 1 has
// landed] [barrier].  The A images carry their CLASS exponent (the producing kernel scales by it), so the segments of one problem
// multiply in different units: a mover leaves the product exponent of every k-tile beside its stage and the multipliers move their
// accumulators to the new unit at a segment boundary (a power of two: exact).
#pragma once
#include "gemm_h2.h"

namespace vsr {

template <int TM, int TN, int NW = 3>
__global__ __launch_bounds__(H2_THREADS)
void gemm_nt_h2a_kernel(const GemmArgs args) {
    constexpr int WM = 4 / TM, WN = 8 / WM;
    constexpr int BM = 128, BN = 32 * TN * WN, BK = H2_BK;
    static_assert(32 * TM * WM == BM, "tile shape");
    constexpr int WSTAGE = BN * 64;                       // fp16 elements of the W rows of a stage (BN rows x 128 bytes)
    constexpr int ASTAGE = BM * 64;                       // ... of its A rows
    constexpr int STG = WSTAGE + ASTAGE;
    static_assert(NW * STG * 2 + 64 <= 163840, "LDS");
    __shared__ __attribute__((aligned(1024))) uint16_t smem[NW * STG + 32];
    uint16_t* const sW = smem;
    int* const s_exp = reinterpret_cast<int*>(smem + NW * STG);      // product exponent of the k-tile in stage s (written by a mover, NW - 1 periods ahead)

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

    int cur_s = 0;                                         // multipliers: exponent of the products accumulated so far (2^cur_s units)
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
    auto flush = [&](auto MULT, float* stage, bool range_done) __attribute__((always_inline)) {
        const GemmProb& P = args.p[c_prob];
        const float unscale = h2_pow2(-cur_s);
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
        if (args.aligned || range_done) {
            // k-aligned plan: this piece is the workgroup's only one, nothing is prefetched behind it, the whole ring is free: stage all 128
            // rows at once - two barriers instead of eight, every thread's stores back to back (the flush is 3 of a skinny launch's 23 us)
            // (round 6: the same for the LAST piece of a stream-K range - the movers have issued nothing behind it either)
            static_assert(BM * ST_LD * 4 <= NW * STG * 2, "whole-tile staging must fit the ring");
            static_assert(BM % RPP == 0, "the whole-tile flush walks BM / RPP row groups with no remainder handling");
            float* const all = reinterpret_cast<float*>(sW);
            if constexpr (decltype(MULT)::value) {
#pragma unroll
                for (int ti = 0; ti < TM; ++ti)
#pragma unroll
                    for (int tj = 0; tj < TN; ++tj)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            all[(32 * (wm * TM + ti) + (e & 3) + 8 * (e >> 2) + 4 * hh) * ST_LD + wn * (32 * TN) + tj * 32 + r] = acc[ti][tj][e] * unscale;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < BM / RPP; ++i) {
                const int sr = tid / TPR + RPP * i;
                const int m = m0 + sr;
                if (m < P.M && n < P.N) {
                    const float4 v = *reinterpret_cast<const float4*>(all + sr * ST_LD + c4);
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
            wait_loads<0>();
            return;
        }
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
            flush(MULT, reinterpret_cast<float*>(sW + ws * STG), it >= it1);
            if (it < it1) decode(it);
        }
        ++j;
        ws = ws + 1 == NW ? 0 : ws + 1;
        return piece_done;
    };

    const int kt0 = decode(it0);

    if (mover) {
        // ================================================================================================ movers: DMA only
        const int mw = wave - 8;
        constexpr int LA = BM / 64, LB = BN / 64;          // A / W DMAs per thread and k-tile
        // this lane's place in a 1 KB block = (row lane >> 3, position lane & 7) -> logical chunk -> image bytes
        const int wr8 = lane >> 3;
        const int wc = (lane & 7) ^ ((4 * (mw & 1) + (lane >> 4)) & 7);      // (R >> 1) & 7 for R = 8 (mw + 8 i) + (lane >> 3): the same for every i
        const int wg8 = 8 * (wc & 3);                      // first k of the chunk's group inside the k-tile
        const int wboff = (wc & 3) * 8 + (wc >> 2) * 4;    // its offset in the row's k-tile, in floats (32 bytes per group, the lo half 16 bytes in)
        const unsigned lds_0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(size_t)(__attribute__((address_space(3))) void*)sW)
                             + (unsigned)__builtin_amdgcn_readfirstlane(mw) * 1024u;
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
            __device__ __forceinline__ void settle(const GemmArgs& a) {
                if (tile_left == 0) {
                    if (tile + 1 < a.p[prob].tiles_m * a.p[prob].tiles_n) open(a, prob, tile + 1, 0);
                    else open(a, prob + 1, 0, 0);
                } else if (seg_left == 0) {
                    ++seg; K = a.p[prob].seg[seg].K; k = 0; seg_left = (K + H2_BK - 1) / H2_BK; fresh = true;
                }
            }
            __device__ __forceinline__ void next() { k += H2_BK; --seg_left; --tile_left; }
        };
        Cursor cw;
        const float* pa[LA];
        const float* pbW[LB];
        int seg_exp = 0;
        auto issue = [&](int stage) __attribute__((always_inline)) {         // DMA the cursor's k-tile (W rows, then A rows) into `stage`
            cw.settle(args);
            if (cw.fresh) {
                const GemmProb& P = args.p[__builtin_amdgcn_readfirstlane(cw.prob)];
                const GemmSeg& S = P.seg[__builtin_amdgcn_readfirstlane(cw.seg)];
                const int n0 = (cw.tile / P.tiles_m) * BN, m0 = (cw.tile % P.tiles_m) * BM;
#pragma unroll
                for (int i = 0; i < LB; ++i) {
                    int n = n0 + 8 * (mw + 8 * i) + wr8;
                    n = n < P.N ? n : P.N - 1;
                    pbW[i] = S.W + (long long)n * S.ldw + wboff;       // the images have the fp32 matrices' byte geometry
                }
#pragma unroll
                for (int i = 0; i < LA; ++i) {
                    int m = m0 + 8 * (mw + 8 * i) + wr8;
                    m = m < P.M ? m : P.M - 1;
                    const long long row = S.a_idx ? (long long)S.a_idx[m] : (long long)m;
                    pa[i] = S.A + row * S.lda + wboff;
                }
                const int ia = S.exp_idx >> 16, va = args.exps[ia];           // (a dynamic slot holds the bound itself: gemm_h2.h h2_prob_exp)
                seg_exp = args.exps[S.exp_idx & 0xffff] + (ia >= H2_DYN0 ? h2_exp_of(__int_as_float(va)) : va);
                seg_exp = seg_exp < -120 ? -120 : (seg_exp > 120 ? 120 : seg_exp);      // (as h2_prob_exp: the unscale factor 2^-seg_exp must stay a normal float)
                cw.fresh = false;
            }
            const bool in = cw.k + wg8 < cw.K;             // K is a multiple of 8: a group is inside or outside as a whole
            const unsigned base = lds_0 + (unsigned)__builtin_amdgcn_readfirstlane(stage) * (unsigned)(STG * 2);
#pragma unroll
            for (int i = 0; i < LB; ++i) h2_glds16(in ? (const void*)(pbW[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(8 * i) * 1024u);
#pragma unroll
            for (int i = 0; i < LA; ++i) h2_glds16(in ? (const void*)(pa[i] + cw.k) : (const void*)g_h2_zero16, base + (unsigned)(WSTAGE * 2) + (unsigned)(8 * i) * 1024u);
            if (tid == 512) s_exp[stage] = seg_exp;
            cw.next();
        };
        // prologue: k-tiles 0 .. NW - 2
        cw.open(args, c_prob, c_tile, kt0);
        {
            int st = 0;
            for (int n = 0; n < NW - 1 && it0 + n < it1; ++n) { issue(st); ++st; }
        }
        // (a full prologue of NW - 1 k-tiles: k-tile 0 has landed once at most the NW - 2 younger ones are in flight; a shorter one: drain)
        if (NW >= 3 && it0 + NW - 2 < it1) wait_loads<(NW - 2) * (LA + LB)>(); else wait_loads<0>();
        __syncthreads();                                   // k-tile 0 is ready
        // k-tile j: issue k-tile j + NW - 1 into the stage the last barrier freed; wait until k-tile j + 1 has landed (at most the NW - 2
        // k-tiles behind it in flight); barrier.  (NW = 4 fits the 128 x 128 tile and was measured: no gain - profiles/r05_e_*; at M = 100 the
        // launch is 23 us of which the k loop without any DMA is 19.5 and with idle multipliers 18.9 - profiles/r05_d_*: fixed cost and MFMA time)
        static_assert(NW == 3 || NW == 4, "the counted waits below are written for rings of three / four");
        while (it < it1) {
            int st = ws + NW - 1;
            st = st >= NW ? st - NW : st;
            const bool more = it + NW - 1 < it1;
            if (more) { issue(st); wait_loads<(NW - 2) * (LA + LB)>(); } else wait_loads<0>();
            end_of_ktile(std::false_type{});
        }
    } else {
        // ================================================================================================ multipliers
        const int wm = wave / WN, wn = wave % WN;
        const int wsw = (r >> 1) & 7;                      // both kinds of rows: ((32 t + r) >> 1) & 7
        auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][jj][e] = 0.f;
        };
        zero_acc();
        bool first = true;
        __syncthreads();
        while (it < it1) {
            const uint16_t* b_row = sW + ws * STG + (wn * (32 * TN) + r) * 64;
            const uint16_t* a_row = sW + ws * STG + WSTAGE + (wm * (32 * TM) + r) * 64;
            // segments of one problem multiply operands of different scales: the accumulators move to the new unit (a power of two: exact)
            const int sj = __builtin_amdgcn_readfirstlane(s_exp[ws]);
            if (sj != cur_s) {
                if (!first) {
                    int dlt = sj - cur_s;
                    dlt = dlt < -126 ? -126 : (dlt > 127 ? 127 : dlt);
                    const float f = h2_pow2(dlt);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int jj = 0; jj < TN; ++jj)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[i][jj][e] *= f;
                }
                cur_s = sj;
            }
            first = false;
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int wh = 8 * ((2 * kk + hh) ^ wsw), wl = 8 * ((4 + 2 * kk + hh) ^ wsw);
                f16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * 64 + wh);
                    al[i] = *reinterpret_cast<const f16x8_t*>(a_row + i * 32 * 64 + wl);
                }
#pragma unroll
                for (int jj = 0; jj < TN; ++jj) {
                    bh[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wh);
                    bl[jj] = *reinterpret_cast<const f16x8_t*>(b_row + jj * 32 * 64 + wl);
                }
#define H2A_TERM(X, Y)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                     \
        _Pragma("unroll") for (int jj = 0; jj < TN; ++jj)                                              \
            acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(X[i], Y[jj], acc[i][jj], 0, 0, 0);
                H2A_TERM(al, bh)
                H2A_TERM(ah, bl)
                H2A_TERM(ah, bh)
#undef H2A_TERM
            }
            if (end_of_ktile(std::true_type{})) { zero_acc(); first = true; }
        }
    }
}

}  // namespace vsr

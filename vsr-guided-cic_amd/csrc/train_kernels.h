// Backward (BPTT) kernels of the decoder step: the hand-written counterpart of what autograd derives from
// /root/reference/models/controllable_captioning.py:117-190 when coco_scripts/train.py:112 calls loss.backward().
// Every matrix product of the backward pass is run by the same fp32-MFMA "NT" GEMM (gemm_f32.h) on transposed
// copies (weights once per step, activations / pre-activation gradients once per call); the kernels here are the
// pointwise / reduction parts.  Notation: d<x> is dLoss/d<x>.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace vsr {

// Transposes, 64 x 64 tiles through LDS, 16 bytes per lane on both sides (256-byte row pieces; the 32 x 32 / 4-byte version
// moved 1.5 TB/s).  GATHER: row r of the input is row list[r]; BF16: ONLY the bf16 image (round-to-nearest-even) is written, to
// out16 - the bf16 mode's W operands (gemm_bf16.h); the fp32 buffer `out` then only lends its address (the twin is looked up by
// it) and a launch that names it but cannot take the bf16 kernel is refused (GemmBuilder::launch).  Unaligned shapes (ld or a
// base not a multiple of 4 floats) take scalar accesses.
__device__ __forceinline__ uint16_t to_bf16_bits(float x) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {x, 0.f};
    return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2)) & 0xffffu);
}
// f16x2 flavour (gemm_h2.h): the GEMMs of the backward pass scale their A operands - gradients, whose range is only known once they
// exist - by a bound the PRODUCING kernel folds into a slot of the exponent table: max |x| over the workgroup, one atomicMax per
// workgroup (bit patterns of non-negative floats order like integers).  bm == nullptr: flavour off.  Every thread of the workgroup
// must reach the call (threads outside the range bring 0).
template <int NV>
__device__ __forceinline__ void block_absmax_to(const float (&m_in)[NV], int* const (&bm)[NV]) {
    __shared__ float wm[NV][16];
    float m[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        m[v] = m_in[v];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[v] = fmaxf(m[v], __shfl_xor(m[v], o, 64));
        if ((threadIdx.x & 63) == 0) wm[v][threadIdx.x >> 6] = m[v];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        float r = 0.f;
        const int nw = (blockDim.x + 63) >> 6;
        for (int i = 0; i < nw; ++i) r = fmaxf(r, wm[threadIdx.x][i]);
        // the slot only ever grows: a workgroup whose maximum does not exceed what the slot already holds (a relaxed, possibly STALE read -
        // stale means smaller, so the test errs on the side of the atomic) has nothing to add.  Same-address atomics serialise at ~12 ns each
        // on this chip (2 048 of them were 25 of k_row_l1_max's 30 us): with ~400 workgroups per launch that was 4-5 us of every backward
        // pointwise kernel's tail (round 6).
        if (r > 0.f && __float_as_int(r) > __atomic_load_n(bm[threadIdx.x], __ATOMIC_RELAXED)) atomicMax(bm[threadIdx.x], __float_as_int(r));
    }
}
__device__ __forceinline__ void block_absmax_to(float m, int* bm) {
    const float mm[1] = {m};
    int* const bb[1] = {bm};
    block_absmax_to<1>(mm, bb);
}

// IMG = 2: ONLY the f16x2 flavour's fp16-pair image (kernels.h: img_store, scaled by 2^exps[slot]) is written, to out16 (2-byte units of a
// 4-byte-per-element buffer) - the W operands of the backward pass's GEMMs when they run on the f16x2 kernels (train.inc.h: transpose()).
template <bool GATHER, int IMG>
__device__ __forceinline__ void transpose_tile(const float* __restrict__ in, long long ld_in, const int* __restrict__ list, int R, int C,
                                               float* __restrict__ out, long long ld_out, uint16_t* __restrict__ out16,
                                               const int* __restrict__ exps, int slot, const int* __restrict__ rlimit, int bx, int by) {
    constexpr bool BF16 = IMG != 0;            // (either image kind: the fp32 buffer is not written)
    // rlimit (device): input rows from *rlimit up are taken as ZERO rows - the row list of a vsr_prepare*() under a caller's row bound is
    // padded to the bound with copies of its first entry (k_pad_row_list), which the weight-gradient reduction must not see
    const int Rl = rlimit ? min(R, *rlimit) : R;
    float isc = 0.f;
    if constexpr (IMG == 2) {                   // (slots from H2_DYN0 up hold the measured BOUND of a gradient operand, not its exponent: gemm_h2.h)
        const int ev = exps[slot];
        isc = h2_pow2(slot >= H2_DYN0 ? h2_exp_of(__int_as_float(ev)) : ev);
    }
    __shared__ float t[64][65];
    const int c0 = bx * 64, r0 = by * 64;
    const int q = threadIdx.x & 15, p = threadIdx.x >> 4;              // 16 lanes x 4 floats cover 64 columns; 16 rows per pass
    const bool vin = ((ld_in & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0);
    const bool vout = ((ld_out & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                      (!BF16 || (reinterpret_cast<uintptr_t>(out16) & 7) == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + p + 16 * i, c = c0 + 4 * q;
        if (r >= Rl && r < R) {
            t[p + 16 * i][4 * q] = 0.f; t[p + 16 * i][4 * q + 1] = 0.f; t[p + 16 * i][4 * q + 2] = 0.f; t[p + 16 * i][4 * q + 3] = 0.f;
        } else if (r < R) {
            const float* src = in + (long long)(GATHER ? list[r] : r) * ld_in;
            if (vin && c + 3 < C) {
                const float4 v = *reinterpret_cast<const float4*>(src + c);
                t[p + 16 * i][4 * q] = v.x; t[p + 16 * i][4 * q + 1] = v.y; t[p + 16 * i][4 * q + 2] = v.z; t[p + 16 * i][4 * q + 3] = v.w;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (c + e < C) t[p + 16 * i][4 * q + e] = src[c + e];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + p + 16 * i, r = r0 + 4 * q;                 // output row c, output columns r .. r + 3
        if (c < C) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = t[4 * q + e][p + 16 * i];
            float* dst = out + (long long)c * ld_out + r;
            if (vout && r + 3 < R) {
                if (BF16) {
                    img_store4(out16, (long long)c * ld_out + r, make_float4(v[0], v[1], v[2], v[3]), isc);
                } else {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                // (columns R .. ld_out - 1 are the K padding of the GEMM that reads this operand - fewer than 8: zeros, written here
                // instead of a memset of the whole transposed region per step)
                for (int e = 0; e < 4; ++e)
                    if (r + e < R) {
                        if (BF16) img_store(out16, (long long)c * ld_out + r + e, v[e], isc);
                        else dst[e] = v[e];
                    } else if (r + e < ld_out) {
                        if (BF16) img_store(out16, (long long)c * ld_out + r + e, 0.f, isc);
                        else dst[e] = 0.f;
                    }
            }
        }
    }
}

template <bool GATHER, int IMG>
__global__ __launch_bounds__(256) void k_transpose_t(const float* __restrict__ in, long long ld_in, const int* __restrict__ list, int R, int C,
                                                     float* __restrict__ out, long long ld_out, uint16_t* __restrict__ out16,
                                                     const int* __restrict__ exps = nullptr, int slot = 0, const int* __restrict__ rlimit = nullptr) {
    transpose_tile<GATHER, IMG>(in, ld_in, list, R, C, out, ld_out, out16, exps, slot, rlimit, blockIdx.x, blockIdx.y);
}

// up to TR_MT transposes of one image kind in ONE launch (round 6: the 13 weight transposes at the head of a backward pass and the 9 + 7
// activation / gradient transposes of its weight-gradient phase were a 5-10 us launch each)
constexpr int TR_MT = 16;
struct TransMulti {
    const float* in[TR_MT]; float* out[TR_MT]; uint16_t* out16[TR_MT];
    long long ld_in[TR_MT], ld_out[TR_MT];
    int R[TR_MT], C[TR_MT], slot[TR_MT], blk[TR_MT + 1];
    int nt;
};
template <int IMG>
__global__ __launch_bounds__(256) void k_transpose_multi(const TransMulti m, const int* __restrict__ exps) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < TR_MT; ++k)
        if (k < m.nt && (int)blockIdx.x >= m.blk[k]) i = k;
    const int local = (int)blockIdx.x - m.blk[i], tc = (m.C[i] + 63) / 64;
    transpose_tile<false, IMG>(m.in[i], m.ld_in[i], nullptr, m.R[i], m.C[i], m.out[i], m.ld_out[i], m.out16[i], exps, m.slot[i], nullptr, local % tc, local / tc);
}

// dst (rows, w) window with leading dimension ldd  =  sum of nslab compact (rows, w) slabs
__global__ void k_slab_reduce_2d(const float* __restrict__ slabs, int nslab, long long stride, int rows, int w,
                                 float* __restrict__ dst, long long ldd) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * w) return;
    const int r = (int)(i / w), c = (int)(i % w);
    float s = slab_sum(slabs + i, nslab, stride);
    dst[(long long)r * ldd + c] = s;
}

// column sums: out[c] = sum_r X[r][c]   (bias gradients).  Two deterministic stages: grid (C/64, NCH) blocks each sum
// a chunk of rows into part[chunk][c], then k_colsum_finish adds the NCH partials in order.
constexpr int COLSUM_CHUNKS = 32;
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ X, long long ld, int R, int C, float* __restrict__ part) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int per = (R + COLSUM_CHUNKS - 1) / COLSUM_CHUNKS;
    const int r0 = blockIdx.y * per, r1 = min(R, r0 + per);
    float s = 0.f;
    if (c < C)
        for (int r = r0 + q; r < r1; r += 4) s += X[(long long)r * ld + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < C)
        part[(long long)blockIdx.y * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// columns [c0, c0 + C) of a partial table with Cs columns per chunk; out2 (optional): a second copy (the two bias vectors of an LSTM cell /
// a gate pair receive the same gradient: lstm_cell_N.bias_ih / bias_hh, W1_is.bias / W1_hs.bias, W1_ig.bias / W1_hg.bias)
__global__ void k_colsum_finish(const float* __restrict__ part, int Cs, int c0, int C, float* __restrict__ out, float* __restrict__ out2 = nullptr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int k = 0; k < COLSUM_CHUNKS; ++k) s += part[(long long)k * Cs + c0 + c];
    out[c] = s;
    if (out2) out2[c] = s;
}

// several buffers zeroed by ONE launch (a hipMemsetAsync each was a 5-6 us launch of its own: 14 per training step)
constexpr int ZERO_MT = 8;
struct ZeroMulti { void* p[ZERO_MT]; long long n16[ZERO_MT]; int blk[ZERO_MT + 1]; int nt; };       // n16: 16-byte units
__global__ __launch_bounds__(256) void k_zero_multi(const ZeroMulti z) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < ZERO_MT; ++k)
        if (k < z.nt && (int)blockIdx.x >= z.blk[k]) i = k;
    float4* p = reinterpret_cast<float4*>(z.p[i]);
    const long long n = z.n16[i], stride = (long long)(z.blk[i + 1] - z.blk[i]) * 256;
    for (long long j = (long long)((int)blockIdx.x - z.blk[i]) * 256 + threadIdx.x; j < n; j += stride) p[j] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// gather of embedding rows for all steps: x_all[(t*B+b)] = embed[word_in[b][t]]
__global__ void k_gather_rows(const float* __restrict__ table, const int* __restrict__ idx, int rows, int E, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * E) return;
    const int r = (int)(i / E), e = (int)(i % E);
    out[i] = table[(long long)idx[r] * E + e];
}
// dEmbed[w] = sum of dx[r] over the rows r with idx[r] == w, IN ASCENDING r: a deterministic segmented sum (float atomics
// gave run-to-run different bits whenever a word repeats inside a batch).  One workgroup per row r; it only works when r is
// the FIRST occurrence of its word (every wave scans the id list 64 entries at a time and leaves as soon as it sees an earlier
// one), then adds the later occurrences in order.  The table gradient is zero-filled beforehand for the untouched rows.
__global__ __launch_bounds__(256) void k_embed_grad_rows(const float* __restrict__ dx, const int* __restrict__ idx, int rows, int E,
                                                         float* __restrict__ table_grad) {
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int w = idx[r];
    constexpr int CMAX = 8;                         // columns per thread: E <= 2048 in one pass, more in further passes
    for (int e0 = 0; e0 < E; e0 += 256 * CMAX) {
        float acc[CMAX];
#pragma unroll
        for (int q = 0; q < CMAX; ++q) acc[q] = 0.f;
        for (int c0 = 0; c0 < rows; c0 += 64) {
            const int j = c0 + lane;
            const bool hit = j < rows && idx[j] == w;
            unsigned long long m = __ballot(hit);
            if (c0 < r) {                            // an earlier row owns this word (wave-uniform exit)
                const unsigned long long before = (r - c0 >= 64) ? ~0ull : ((1ull << (r - c0)) - 1ull);
                if (m & before) return;
            }
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float* src = dx + (long long)(c0 + b) * E + e0;
#pragma unroll
                for (int q = 0; q < CMAX; ++q) {
                    const int e = tid + 256 * q;
                    if (e0 + e < E) acc[q] += src[e];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < CMAX; ++q) {
            const int e = e0 + tid + 256 * q;
            if (e < E) table_grad[(long long)w * E + e] = acc[q];
        }
    }
}

// Index-list regions (data/field.py:44-61 as indices): dP_bank[row] = sum of the entry gradients dP[e] over the slot
// entries e of the row's image that name bank row `row`, IN ASCENDING e (deterministic).  One workgroup per bank row, one float4
// column group per thread (A <= 512 per pass); the image's entry list (L R ints) is scanned 64 entries at a time.
__global__ __launch_bounds__(128) void k_dP_to_bank(const float* __restrict__ dP, const int* __restrict__ ridx, int LR, int Rb, int A,
                                                    float* __restrict__ dP_bank) {
    const int row = blockIdx.x, img = row / Rb, tid = threadIdx.x, lane = tid & 63;
    const int* e_idx = ridx + (long long)img * LR;
    const float* src = dP + (long long)img * LR * A;
    for (int a0 = 0; a0 < A; a0 += 512) {
        const int a = a0 + tid * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c0 = 0; c0 < LR; c0 += 64) {
            const int j = c0 + lane;
            unsigned long long m = __ballot(j < LR && e_idx[j] == row);
            while (m) {
                const int b = __ffsll((long long)m) - 1;
                m &= m - 1;
                if (a < A) {
                    const float4 v = *reinterpret_cast<const float4*>(src + (long long)(c0 + b) * A + a);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        if (a < A) *reinterpret_cast<float4*>(dP_bank + (long long)row * A + a) = acc;
    }
}

// ---------------------------------------------------------------------------------------------- forward saves
// LSTM1 + gates, training flavour: also stores the post-activation gates (B, 6H) = [i f g o s_gate .]
__global__ void k_lstm1_train(const float* __restrict__ pre, int nsplit, long long stride, const float* __restrict__ vproj,
                              const float* __restrict__ xproj /* (M, 6H) embedding part of this step, projected for all steps at once */,
                              const float* __restrict__ c1_old, int M, int H, float* __restrict__ h1n, float* __restrict__ c1n,
                              float* __restrict__ s_t, float* __restrict__ gpre, float* __restrict__ gates,
                              int nblk /* column blocks (of 6) that the recurrent GEMM wrote */,
                              uint16_t* __restrict__ h1n16 = nullptr, uint16_t* __restrict__ s_t16 = nullptr /* optional images (img_store) */, float isc = 0.f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * H) return;
    const int row = (int)(i / H), j = (int)(i % H);
    const long long base = (long long)row * 6 * H + j;
    float q[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const float s = g < nblk ? slab_sum(pre + base + (long long)g * H, nsplit, stride) : 0.f;
        q[g] = s + xproj[base + (long long)g * H] + vproj[base + (long long)g * H];
    }
    const float ig = sigmoidf_(q[0]), fg = sigmoidf_(q[1]), gg = tanhf(q[2]), og = sigmoidf_(q[3]), sg = sigmoidf_(q[4]);
    const float c = fg * c1_old[i] + ig * gg;
    const float tc = tanhf(c);
    const float h1v = og * tc, stv = sg * tc;
    h1n[i] = h1v;
    c1n[i] = c;
    s_t[i] = stv;
    gpre[i] = q[5];
    if (h1n16) { img_store(h1n16, i, h1v, isc); img_store(s_t16, i, stv, isc); }
    gates[base] = ig; gates[base + H] = fg; gates[base + 2LL * H] = gg; gates[base + 3LL * H] = og; gates[base + 4LL * H] = sg;
}

// after the S5 GEMM, one launch: blocks [0, gblocks = M) finish att_ga(g_t) from its slabs (saved for the backward pass) and
// write the step's gate log-probs, one workgroup per row; the other blocks are LSTM2 with its post-activation gates saved
__global__ __launch_bounds__(256) void k_fwd_tail(const GateLogitArgs gl, int gblocks,
                                                  const float* __restrict__ pre, int nsplit, long long stride, const float* __restrict__ b_ih,
                                                  const float* __restrict__ b_hh, const float* __restrict__ vproj2, const float* __restrict__ c2_old,
                                                  int M, int H, float* __restrict__ h2n, float* __restrict__ c2n, float* __restrict__ gates,
                                                  uint16_t* __restrict__ h2n16 = nullptr /* optional image (img_store) */, float isc = 0.f) {
    if ((int)blockIdx.x < gblocks) {                      // one workgroup per row: every slab of a column in flight at once (a wave
        __shared__ float red[4];                          // per row walked the slabs in dependent rounds: 20 us at 8 slabs)
        gatelogit_block<256>(gl, blockIdx.x, red);
        return;
    }
    const long long i = (long long)(blockIdx.x - gblocks) * 256 + threadIdx.x;
    if (i >= (long long)M * H) return;
    const int row = (int)(i / H), j = (int)(i % H);
    const long long base = (long long)row * 4 * H + j;
    float q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = slab_sum(pre + base + (long long)g * H, nsplit, stride);
        s += b_ih[g * H + j] + b_hh[g * H + j];
        if (vproj2) s += vproj2[base + (long long)g * H];
        q[g] = s;
    }
    const float ig = sigmoidf_(q[0]), fg = sigmoidf_(q[1]), gg = tanhf(q[2]), og = sigmoidf_(q[3]);
    const float c = fg * c2_old[i] + ig * gg;
    const float h2v = og * tanhf(c);
    h2n[i] = h2v;
    c2n[i] = c;
    gates[base] = ig; gates[base + H] = fg; gates[base + 2LL * H] = gg; gates[base + 3LL * H] = og;
    if (h2n16) img_store(h2n16, i, h2v, isc);
}

// ---------------------------------------------------------------------------------------------- backward
// Pointwise formulas used by the fused step kernels further down:
//   gate log-softmax + z_g (:184-188): d[z_g, zsum] = dlg - exp(lg) * (dlg0 + dlg1);  z_g = w_g . tanh(ga + hA);
//       dga = dz_g * w_g * (1 - th^2) (also the first contribution to dhA), dw_g partial per row = dz_g * th
//   LSTM cell (both cells): dtc = dh * o (+ gradient reaching tanh(c) from the s_t / g_t paths, LSTM1 only);
//       dc = dc_next + dtc (1 - tanh^2 c);  dpre = [dc g i(1-i), dc c_prev f(1-f), dc i (1-g^2), dh tanh(c) o(1-o)];  dc_prev = dc f
//   shift gate (:181-182): g_t = gg tanh(c1):  dq = dg_t tanh(c1) gg (1-gg),  dtc = dg_t gg
//   sentinel gate (:151-154): s_t = sg tanh(c1):  ds_pre = ds_t tanh(c1) sg (1-sg),  dtc += ds_t sg

// dalpha[row][j] = datt[row] . regions_j  (j = 0: sentinel), one WAVE per (row, j): B x (R+1) independent dot products
// over D, so that a training batch of 100 rows still fills the chip (one workgroup per row left 60 % of the CUs idle).
__global__ __launch_bounds__(256) void k_dalpha(const float* __restrict__ datt, const float* __restrict__ sent,
                                                const float* __restrict__ X, const float* __restrict__ rmask,
                                                const int* __restrict__ ridx /* index-list regions: bank row of every slot entry, or null */,
                                                const int* __restrict__ slot, int M, int L, int R, int D, float* __restrict__ dalpha) {
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= M * (R + 1)) return;
    const int row = item / (R + 1), j = item % (R + 1), lane = threadIdx.x & 63;
    const long long sl = (long long)row * L + slot[row];
    float s = 0.f;
    if (j == 0 || rmask[sl * R + j - 1] != 0.f) {
        const float* g = datt + (long long)row * D;
        const float* src = (j == 0) ? sent + (long long)row * D : X + (ridx ? (long long)ridx[sl * R + j - 1] : sl * R + j - 1) * D;
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
        int d = lane * 4;
        for (; d + 768 < D; d += 1024) {
            const float4 b0 = *reinterpret_cast<const float4*>(src + d), b1 = *reinterpret_cast<const float4*>(src + d + 256);
            const float4 b2 = *reinterpret_cast<const float4*>(src + d + 512), b3 = *reinterpret_cast<const float4*>(src + d + 768);
            const float4 a0 = *reinterpret_cast<const float4*>(g + d), a1 = *reinterpret_cast<const float4*>(g + d + 256);
            const float4 a2 = *reinterpret_cast<const float4*>(g + d + 512), a3 = *reinterpret_cast<const float4*>(g + d + 768);
            p0 += a0.x * b0.x + a0.y * b0.y + a0.z * b0.z + a0.w * b0.w;
            p1 += a1.x * b1.x + a1.y * b1.y + a1.z * b1.z + a1.w * b1.w;
            p2 += a2.x * b2.x + a2.y * b2.y + a2.z * b2.z + a2.w * b2.w;
            p3 += a3.x * b3.x + a3.y * b3.y + a3.z * b3.z + a3.w * b3.w;
        }
        for (; d < D; d += 256) {
            const float4 a = *reinterpret_cast<const float4*>(g + d);
            const float4 b = *reinterpret_cast<const float4*>(src + d);
            p0 += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
        s = wave_sum((p0 + p1) + (p2 + p3));
    }
    if (lane == 0) dalpha[item] = s;
}

// attention backward (:158-171, :187), one 256-thread workgroup per row.
//   att = a0 * sent + sum_r a_r X_r ;  a = (softmax(z) * m) / sum(softmax(z) * m) ;  zsum = sum_r m_r z_r
//   z_r = w_a . tanh(P_r + hA) ;  z_0 = w_s . tanh(sa + hA)
// in : datt (M,D), dzsum (M), alpha (M,R+1), saved hA, sa, sent; P, X, rmask of the row's (image, slot)
// out: dsent (M,D) = a0 * datt; dsa (M,A); dhA (M,A) += ; dP[(image,slot)] (R,A) += ; per-row partials of dw_a, dw_s
template <int NT>
__global__ __launch_bounds__(NT) void k_attend_bwd(const float* __restrict__ datt, const float* __restrict__ dalpha_in,
                                                    const float* __restrict__ dzsum,
                                                    const float* __restrict__ alpha, const float* __restrict__ hA,
                                                    const float* __restrict__ sa, const float* __restrict__ sent,
                                                    const float* __restrict__ P, const float* __restrict__ X,
                                                    const float* __restrict__ rmask, const int* __restrict__ ridx, const int* __restrict__ slot, int fixed_slot,
                                                    int M, int L, int R, int A, int D, const float* __restrict__ w_a,
                                                    const float* __restrict__ w_s, float* __restrict__ dsent, float* __restrict__ dsa,
                                                    float* __restrict__ dhA, float* __restrict__ dP, float* __restrict__ dwa_rows,
                                                    float* __restrict__ dws_rows, int* bm_dhA, int* bm_dsent, int* bm_dsa,
                                                    int nparts = 1 /* workgroups per row (round 6, launches of <= 128 rows): part p owns columns [p A / nparts, ..) of the
                                                                      A-wide outputs and [p D / nparts, ..) of dsent; the NT threads of a part are A / nparts columns x
                                                                      RG row groups (region rows r = rg, rg + RG, ..): with 256 columns and two groups a thread's chain
                                                                      is 18 region rows whose loads are all in flight at once instead of five rounds of eight */) {
    extern __shared__ float sm[];
    float* da = sm;                 // R+1: dalpha, then dz
    float* red = sm + R + 1;        // 8
    float* part_s = red + 8;        // nparts > 1: (RG - 1) x Ap x 2 partial sums (dh, dwa) of the row groups behind the first
    const int item = xcd_item(M * nparts);
    if (item < 0) return;           // (whole workgroups)
    const int row = item / nparts, part = item - row * nparts;
    float mx[3] = {0.f, 0.f, 0.f};  // max |dhA|, |dsent|, |dsa| of this row (f16x2 bounds)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = slot ? slot[row] : fixed_slot;
    const long long sl = (long long)row * L + k;
    const float* g = datt + (long long)row * D;
    const float* al = alpha + (long long)row * (R + 1);
    const float* mk = rmask + sl * R;
    // dalpha comes from k_dalpha (one wave per (row, j))
    for (int j = tid; j < R + 1; j += NT) da[j] = dalpha_in[(long long)row * (R + 1) + j];
    // dsent = alpha_0 * datt   (this part's share of the D columns)
    const float a0 = al[0];
    const int Dp = D / nparts;
    for (int d = part * Dp + tid * 4; d < (part + 1) * Dp; d += 4 * NT) {
        const float4 a = *reinterpret_cast<const float4*>(g + d);
        const float4 o = make_float4(a0 * a.x, a0 * a.y, a0 * a.z, a0 * a.w);
        *reinterpret_cast<float4*>(dsent + (long long)row * D + d) = o;
        mx[1] = fmaxf(fmaxf(mx[1], fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    __syncthreads();
    if (wave == 0) {
        // alpha = q / Q, q = s * m.  Recover s from alpha is not possible for masked entries (alpha = 0), but their
        // softmax value only matters through sum_k ds_k s_k with ds_k = dq_k m_k = 0: masked entries drop out, and for
        // unmasked ones s_j = alpha_j * Q with Q = sum of unmasked s.  Everything below is therefore in terms of alpha and Q:
        //   dq_j = (dalpha_j - sum_k dalpha_k alpha_k) / Q ;  dz_j = s_j (dq_j m_j - sum_k dq_k m_k s_k)
        //        = alpha_j Q [ (dalpha_j - S)/Q  - sum_k (dalpha_k - S)/Q * alpha_k Q ... ]   with S = sum dalpha alpha
        // sum_k (dalpha_k - S) alpha_k = S - S = 0  =>  dz_j = alpha_j (dalpha_j - S)   for unmasked j, 0 for masked j.
        float S = 0.f;
        for (int j = lane; j < R + 1; j += 64) S += da[j] * al[j];
        S = wave_sum(S);
        const float dzs = dzsum[row];
        for (int j = lane; j < R + 1; j += 64) {
            float dz = al[j] * (da[j] - S);
            if (j > 0) dz += mk[j - 1] * dzs;             // the shift logit sums the raw scores of the valid regions
            da[j] = dz;
        }
    }
    __syncthreads();
    // du = dz * w * (1 - tanh^2(P + hA)); dP rows, dhA, dsa and the per-row partials of dw_a / dw_s
    const float* Pk = P + sl * R * A;
    float* dPk = dP + sl * R * A;
    if (nparts == 1) {
        for (int a = tid; a < A; a += NT) {
            const float h = hA[(long long)row * A + a];
            const float wa = w_a[a];
            float dh = 0.f, dwa = 0.f;
            // eight region rows per round: their P and dP loads are all in flight before the first tanh (the row loop was one
            // dependent load -> tanh -> read-modify-write per row: 36 serial L2 round trips)
            for (int r0 = 0; r0 < R; r0 += 8) {
                float pv[8], dpv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = r0 + q;
                    const bool live = r < R && da[r + 1] != 0.f;            // uniform over the workgroup
                    pv[q] = live ? (ridx ? P[(long long)ridx[sl * R + r] * A + a] : Pk[(long long)r * A + a]) : 0.f;
                    dpv[q] = live ? dPk[(long long)r * A + a] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = r0 + q;
                    if (r < R) {
                        const float dz = da[r + 1];
                        if (dz != 0.f) {
                            const float th = tanhf(pv[q] + h);
                            const float du = dz * wa * (1.f - th * th);
                            dPk[(long long)r * A + a] = dpv[q] + du;           // a row visits its slots one step at a time: no race
                            dh += du;
                            dwa += dz * th;
                        }
                    }
                }
            }
            const float ths = tanhf(sa[(long long)row * A + a] + h);
            const float dus = da[0] * w_s[a] * (1.f - ths * ths);
            dsa[(long long)row * A + a] = dus;
            const float dhn = dhA[(long long)row * A + a] + (dh + dus);
            dhA[(long long)row * A + a] = dhn;
            mx[0] = fmaxf(mx[0], fabsf(dhn)); mx[2] = fmaxf(mx[2], fabsf(dus));
            dwa_rows[(long long)row * A + a] = dwa;
            dws_rows[(long long)row * A + a] = da[0] * ths;
        }
    } else {
        // this part's Ap columns x RG row groups (NT = Ap RG: run_step checks)
        const int Ap = A / nparts, RG = NT / Ap;
        const int al_ = tid % Ap, rg = tid / Ap;
        const int a = part * Ap + al_;
        const float h = hA[(long long)row * A + a];
        const float wa = w_a[a];
        float dh = 0.f, dwa = 0.f;
        constexpr int CH = 18;                       // region rows in flight per thread: one round covers R = 36 with two row groups
        for (int r0 = rg; r0 < R; r0 += CH * RG) {
            float pv[CH], dpv[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int r = r0 + q * RG;
                const bool live = r < R && da[r + 1] != 0.f;
                pv[q] = live ? (ridx ? P[(long long)ridx[sl * R + r] * A + a] : Pk[(long long)r * A + a]) : 0.f;
                dpv[q] = live ? dPk[(long long)r * A + a] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int r = r0 + q * RG;
                if (r < R) {
                    const float dz = da[r + 1];
                    if (dz != 0.f) {
                        const float th = tanhf(pv[q] + h);
                        const float du = dz * wa * (1.f - th * th);
                        dPk[(long long)r * A + a] = dpv[q] + du;
                        dh += du;
                        dwa += dz * th;
                    }
                }
            }
        }
        // the row groups' partial sums are added in group order by the first group's thread (fixed order: bitwise repeatable)
        if (rg > 0) { part_s[((rg - 1) * Ap + al_) * 2] = dh; part_s[((rg - 1) * Ap + al_) * 2 + 1] = dwa; }
        __syncthreads();
        if (rg == 0) {
            for (int q = 1; q < RG; ++q) { dh += part_s[((q - 1) * Ap + al_) * 2]; dwa += part_s[((q - 1) * Ap + al_) * 2 + 1]; }
            const float ths = tanhf(sa[(long long)row * A + a] + h);
            const float dus = da[0] * w_s[a] * (1.f - ths * ths);
            dsa[(long long)row * A + a] = dus;
            const float dhn = dhA[(long long)row * A + a] + (dh + dus);
            dhA[(long long)row * A + a] = dhn;
            mx[0] = fmaxf(mx[0], fabsf(dhn)); mx[2] = fmaxf(mx[2], fabsf(dus));
            dwa_rows[(long long)row * A + a] = dwa;
            dws_rows[(long long)row * A + a] = da[0] * ths;
        }
    }
    if (bm_dhA) {
        int* const bb[3] = {bm_dhA, bm_dsent, bm_dsa};
        block_absmax_to<3>(mx, bb);
    }
}

// ---------------------------------------------------------------------------------------------- fused step kernels
// One backward step used to be 11 pointwise launches of ~5 us around its 3 GEMMs; the three kernels below do the same
// arithmetic in the same order (bit-identical results) in 3 launches + k_dalpha + k_attend_bwd.
//
// k_bwd_head: (a) blocks [0, gblocks): gate log-softmax backward, one wave per row (the former k_gatelogit_bwd);
//             (b) the remaining blocks, one thread per (row, j): finish the carries of the LATER step's GEMM 3 straight from its
//                 slabs (dh1 carry -> dh1_c; LSTM1-input part of the dh2 carry added to the hh part in dh2_c), add the
//                 vocabulary part of dh2 and run the LSTM2 pointwise backward.
struct BwdHeadArgs {
    const float* lg; const float* dlg; long long lg_stride; const float* ga; const float* hA; const float* w_g; int A;
    float* dga; float* dhA; float* dzsum; float* dwg_rows; int gblocks;
    const float* s_h1; const float* s_h2; int nslab3; long long stride3;    // GEMM 3 slabs of the later step (nslab3 = 0: none yet)
    float* dh1_c; const float* dh2_c; const float* dh2_voc;
    const float* dc_next; const float* gates2; const float* c2; const float* c2_prev;
    int M, H; float* dpre2; float* dc_prev;
    int* bm_dga; int* bm_dpre2;       // f16x2 bound slots of dga / dpre2 (null: flavour off)
};
__global__ __launch_bounds__(256) void k_bwd_head(const BwdHeadArgs q) {
    if ((int)blockIdx.x < q.gblocks) {
        // one WORKGROUP per row (round 6; a wave per row walked the A columns in eight dependent rounds - load, tanh, three stores - while
        // the elementwise blocks behind it had long finished: ~6 of this kernel's 13.5 us at batch 100)
        const int row = blockIdx.x;
        float mx = 0.f;
        if (row < q.M) {
            const int A = q.A;
            const float l0 = q.lg[row * q.lg_stride], l1 = q.lg[row * q.lg_stride + 1];
            const float g0 = q.dlg[row * q.lg_stride], g1 = q.dlg[row * q.lg_stride + 1];
            const float tot = g0 + g1;
            const float dzg = g0 - expf(l0) * tot, dzs = g1 - expf(l1) * tot;
            for (int a = threadIdx.x; a < A; a += 256) {
                const float th = tanhf(q.ga[(long long)row * A + a] + q.hA[(long long)row * A + a]);
                const float du = dzg * q.w_g[a] * (1.f - th * th);
                q.dga[(long long)row * A + a] = du;
                q.dhA[(long long)row * A + a] = du;               // first writer of dhA for this step
                q.dwg_rows[(long long)row * A + a] = dzg * th;    // summed over rows later (k_colsum)
                mx = fmaxf(mx, fabsf(du));
            }
            if (threadIdx.x == 0) q.dzsum[row] = dzs;
        }
        if (q.bm_dga) block_absmax_to(mx, q.bm_dga);
        return;
    }
    const int H = q.H;
    const long long i = (long long)(blockIdx.x - q.gblocks) * 256 + threadIdx.x;
    float mx = 0.f;
    if (i < (long long)q.M * H) {
        const int row = (int)(i / H), j = (int)(i % H);
        float carry = q.dh2_c[i];
        if (q.nslab3 > 0) {
            q.dh1_c[i] = slab_sum(q.s_h1 + i, q.nslab3, q.stride3);
            if (q.s_h2) carry = slab_sum(q.s_h2 + i, q.nslab3, q.stride3, carry);
        }
        const float dhv = q.dh2_voc[i] + carry + 0.f;
        const float* g = q.gates2 + (long long)row * 4 * H + j;
        const float ig = g[0], fg = g[H], gg = g[2LL * H], og = g[3LL * H];
        const float tc = tanhf(q.c2[i]);
        const float dtc = dhv * og;
        const float dc = q.dc_next[i] + dtc * (1.f - tc * tc);
        float* d = q.dpre2 + (long long)row * 4 * H + j;
        const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * (q.c2_prev ? q.c2_prev[i] : 0.f) * fg * (1.f - fg);
        const float d2 = dc * ig * (1.f - gg * gg), d3 = dhv * tc * og * (1.f - og);
        d[0] = d0; d[H] = d1; d[2LL * H] = d2; d[3LL * H] = d3;
        q.dc_prev[i] = dc * fg;
        mx = fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3)));
    }
    if (q.bm_dpre2) block_absmax_to(mx, q.bm_dpre2);
}

// k_bwd_mid (after GEMM 1; grid.y = part): 0: datt = sum of the slabs' columns [H, H+D);  1: dg_t = sum of the att_ga slabs,
// straight into the shift-gate backward (dq, dtc; the former k_gate2_bwd);  2: dh_tot = dh1_c + slabs' columns [0, H) and the
// hh part of the new dh2 carry.
struct BwdMidArgs {
    const float* C0; const float* C1; const float* C2; int nslab; long long st0, st1, st2;
    int M, H, D;
    float* datt; const float* dh1_c; float* dh_tot; float* dh2_c;
    const float* gates1; const float* c1; float* dq; float* dtc;          // gates1 (M,6H); dq window of ld 6H
    int* bm_dq;                                                           // f16x2 bound slot of dq (null: flavour off)
};
__global__ __launch_bounds__(256) void k_bwd_mid(const BwdMidArgs q) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const int H = q.H, D = q.D;
    if (blockIdx.y == 0) {
        if (i >= (long long)q.M * D) return;
        const int r = (int)(i / D), c = (int)(i % D);
        q.datt[i] = slab_sum(q.C0 + (long long)r * (H + D) + H + c, q.nslab, q.st0);
        return;
    }
    const bool live = i < (long long)q.M * H;
    const int r = (int)(i / H), c = (int)(i % H);
    if (blockIdx.y == 1) {
        float mx = 0.f;
        if (live) {
            const float d = slab_sum(q.C2 + i, q.nslab, q.st2);
            const float tc = tanhf(q.c1[i]);
            const float gg = q.gates1[(long long)r * 6 * H + 5LL * H + c];
            const float o = d * tc * gg * (1.f - gg);
            q.dq[(long long)r * 6 * H + c] = o;
            q.dtc[i] = d * gg;
            mx = fabsf(o);
        }
        if (q.bm_dq) block_absmax_to(mx, q.bm_dq);
    } else if (live) {
        q.dh_tot[i] = slab_sum(q.C0 + (long long)r * (H + D) + c, q.nslab, q.st0, q.dh1_c[i]);
        q.dh2_c[i] = slab_sum(q.C1 + i, q.nslab, q.st1);
    }
}

// k_bwd_tail (after GEMM 2): dh1 += slabs of [dq | dhA] . [W1_hg ; att_ha], ds_t = slabs of [dsent | dsa] . [s_fc ; att_sa], the
// sentinel-gate backward and the LSTM1 pointwise backward (the former k_slab_cols_multi + k_sgate_bwd + k_lstm_bwd).
__global__ __launch_bounds__(256) void k_bwd_tail(const float* __restrict__ Ca, const float* __restrict__ Cb, int nslab, long long st,
                                                  const float* __restrict__ dh_tot, const float* __restrict__ dtc_in,
                                                  const float* __restrict__ dc_next, const float* __restrict__ gates1,
                                                  const float* __restrict__ c1, const float* __restrict__ c1_prev, int M, int H,
                                                  float* __restrict__ dpre1, float* __restrict__ dc_prev, int* bm_dpre1) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float mx = 0.f;
    if (i < (long long)M * H) {
        const int row = (int)(i / H), j = (int)(i % H);
        const float dhv = slab_sum(Ca + i, nslab, st, dh_tot[i]);
        const float ds = slab_sum(Cb + i, nslab, st);
        const float* g = gates1 + (long long)row * 6 * H + j;
        const float ig = g[0], fg = g[H], gg = g[2LL * H], og = g[3LL * H], sg = g[4LL * H];
        const float tc = tanhf(c1[i]);
        float* d = dpre1 + (long long)row * 6 * H + j;
        const float d4 = ds * tc * sg * (1.f - sg);
        d[4LL * H] = d4;
        const float extra = dtc_in[i] + ds * sg;
        float dtc = dhv * og;
        dtc += extra;
        const float dc = dc_next[i] + dtc * (1.f - tc * tc);
        const float d0 = dc * gg * ig * (1.f - ig), d1 = dc * (c1_prev ? c1_prev[i] : 0.f) * fg * (1.f - fg);
        const float d2 = dc * ig * (1.f - gg * gg), d3 = dhv * tc * og * (1.f - og);
        d[0] = d0; d[H] = d1; d[2LL * H] = d2; d[3LL * H] = d3;
        dc_prev[i] = dc * fg;
        mx = fmaxf(fmaxf(fmaxf(fabsf(d0), fabsf(d1)), fmaxf(fabsf(d2), fabsf(d3))), fabsf(d4));
    }
    if (bm_dpre1) block_absmax_to(mx, bm_dpre1);      // (rows [0, 5H) of dpre1; the gate block [5H, 6H) is k_bwd_mid's dq)
}

// (B, T) int64 captions / slot traces -> (T, B) int32 step-major copies, plus the (b, t)-ordered row list of the saved
// states (row (t + 1) * B + b) that the batched vocabulary projection gathers through.  slots == null: slot = step.
__global__ void k_train_indices(const int64_t* __restrict__ word_in, const int64_t* __restrict__ slots, int T, int B, int V, int L,
                                int* __restrict__ word32, int* __restrict__ slot32, int* __restrict__ rows_bt, int* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * B) return;
    const int tt = i / B, b = i - tt * B;
    long long w = word_in[(long long)b * T + tt];
    long long k = slots ? slots[(long long)b * T + tt] : tt;
    if (w < 0 || w >= V) { atomicAdd(bad, 1); w = w < 0 ? 0 : V - 1; }      // counted, clamped: never an out-of-bounds gather
    if (k < 0 || k >= L) { atomicAdd(bad, 1); k = k < 0 ? 0 : L - 1; }
    word32[i] = (int)w;
    slot32[i] = (int)k;
    rows_bt[b * T + tt] = (tt + 1) * B + b;
}

}  // namespace vsr

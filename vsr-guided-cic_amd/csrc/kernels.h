// Pointwise / reduction kernels of the decoder step (everything that is not a GEMM).
// Reference equations: /root/reference/models/controllable_captioning.py:117-190 (step), :192-297 (step_v);
// loops: /root/reference/models/CaptioningModel.py:38-76 (greedy, sampling), :116-294 (beam search).
// All arithmetic is fp32 with accurate expf/tanhf/logf (no fast-math), wave64 shuffles for reductions.  One exception,
// measured and bounded: the ~19 000 tanh per row-step of the attention scores use tanhf() below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsr {

constexpr int KMAX = 8;  // VSR_MAX_BEAM

// sum of up to 8 slabs at one offset, in slab order, with every load issued before the first add (the slab count is a
// run-time number: a plain loop is a chain of dependent L2 round trips)
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, int nslab, long long stride, float init = 0.f) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < nslab ? p[k * stride] : 0.f;
    float s = init;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (k < nslab) s += v[k];
    return s;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// bf16 image (round-to-nearest-even) of a value / of four consecutive values: the bf16 GEMM mode lets the producers of its A
// operands write this image next to the fp32 value, so that the GEMM loads half the bytes and converts nothing
__device__ __forceinline__ uint16_t bf16_bits(float x) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {x, 0.f};
    return (uint16_t)(__builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2)) & 0xffffu);
}
__device__ __forceinline__ uint2 bf16_bits4(float4 v) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, b2)), __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, b2)));
}

// Images of A operands for the GEMM kernels that take them (GemmSeg::A16), written by the producers next to the fp32 values.
// isc == 0: bf16, 2 bytes per element (bf16 mode).  isc > 0: the f16x2 flavour's fp16 pairs of x * isc in the fp32 matrix's byte geometry
// (gemm_h2a.h: both operands go global -> LDS by DMA, nothing is converted in the GEMM): elements [8 g, 8 g + 8) -> [hi x 8 | lo x 8],
// `img` then addresses 2-byte units of a 4-byte-per-element buffer.  idx = row * ld + column with ld a multiple of 8.
__device__ __forceinline__ void img_store(uint16_t* __restrict__ img, long long idx, float v, float isc) {
    if (isc == 0.f) { img[idx] = bf16_bits(v); return; }
    const float x = v * isc;
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)(x - (float)hi);
    uint16_t* g = img + ((idx >> 3) << 4) + (idx & 7);
    g[0] = __builtin_bit_cast(uint16_t, hi);
    g[8] = __builtin_bit_cast(uint16_t, lo);
}
__device__ __forceinline__ void img_store4(uint16_t* __restrict__ img, long long idx /* a multiple of 4 */, float4 v, float isc) {
    if (isc == 0.f) { *reinterpret_cast<uint2*>(img + idx) = bf16_bits4(v); return; }
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 x = {v.x * isc, v.y * isc, v.z * isc, v.w * isc};
    const h4 hi = __builtin_convertvector(x, h4);
    const f4 rr = x - __builtin_convertvector(hi, f4);
    const h4 lo = __builtin_convertvector(rr, h4);
    uint16_t* g = img + ((idx >> 3) << 4) + (idx & 7);
    *reinterpret_cast<uint2*>(g) = __builtin_bit_cast(uint2, hi);
    *reinterpret_cast<uint2*>(g + 8) = __builtin_bit_cast(uint2, lo);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-contiguous dealing of n work items over a grid of 8*ceil(n/8) blocks: rows of one image (adjacent
// items) land on the same XCD and share its L2.  Returns -1 for the padding blocks.
__device__ __forceinline__ int xcd_item(int n) {
    const int chunk = (n + 7) >> 3;
    const int it = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    return it < n ? it : -1;
}

// ---------------------------------------------------------------------------------------------- prepare
// pooled descriptor vbar[b] = sum_r det[b,r,:] / #(rows with non-zero sum)           (step :126-128)
// dmask = k_rowmask over the (n_img * R0) detection rows; one block per (decoder row, 1024-column chunk), rows summed in
// order r = 0..R0-1.  row_img (optional): decoder row b pools the detections of image row_img[b].
__global__ __launch_bounds__(256) void k_pool(const float* __restrict__ det, const int* __restrict__ row_img,
                                              const float* __restrict__ dmask, int R0, int D, float* __restrict__ vbar) {
    const int b = blockIdx.x, d = (blockIdx.y * 256 + threadIdx.x) * 4;
    const int img = row_img ? row_img[b] : b;
    const float* X = det + (long long)img * R0 * D;
    const float* mk = dmask + (long long)img * R0;
    float n = 0.f;
    for (int r = 0; r < R0; ++r) n += mk[r];
    if (d >= D) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < R0; ++r) {
        const float4 v = *reinterpret_cast<const float4*>(X + (long long)r * D + d);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(vbar + (long long)b * D + d) = make_float4(s.x / n, s.y / n, s.z / n, s.w / n);
}

// region-row masks m[row] = (sum_d regions[row,:] != 0), one wave per row                 (step :159)
// blockmax (optional): max |x| over the block's four rows, one float per block (the f16x2 GEMM flavour scales the region / detection
// operands by a bound measured here, in the pass that reads them anyway; k_max_reduce folds the per-block values)
__global__ __launch_bounds__(256) void k_rowmask(const float* __restrict__ X, long long rows, int D, float* __restrict__ mask,
                                                 float* __restrict__ blockmax = nullptr) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float s = 0.f, mx = 0.f;
    if (row < rows) {
        for (int d = lane * 4; d < D; d += 256) {
            float4 v = *reinterpret_cast<const float4*>(X + row * D + d);
            s += (v.x + v.y) + (v.z + v.w);
            if (blockmax) mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        s = wave_sum(s);
        if (lane == 0) mask[row] = (s != 0.f) ? 1.f : 0.f;
    }
    if (blockmax) {                                       // (block-uniform)
        __shared__ float wm[4];
        mx = wave_max(mx);
        if (lane == 0) wm[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    }
}
// *out = max(v[0 .. n)) as the bit pattern of a non-negative float (a NaN counts as +inf); one block
__global__ __launch_bounds__(1024) void k_max_reduce(const float* __restrict__ v, long long n, unsigned* __restrict__ out) {
    __shared__ float wm[16];
    float m = 0.f;
    bool nan = false;
    for (long long i = threadIdx.x; i < n; i += 1024) { const float x = v[i]; nan |= !(x == x); m = fmaxf(m, x); }
    if (nan) m = __int_as_float(0x7f800000);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, wm[w]);
        *out = __float_as_uint(m);
    }
}

// list of the non-padding region rows (ascending) and their count: att_va(0) = 0 (no bias), so the hoisted region
// projection only has to run over these rows.  (k_compact_count / k_compact_scan / k_compact_write below.)
// Index-list region format: slot entry (b, l, r) names row slot_idx[b,l,r] of image row_img[b]'s feature
// bank (-1 = padding).  ridx = absolute bank row (or -1), rmask = the reference's row mask of the dense tensor the list
// stands for (an all-zero bank row is masked exactly as its dense copy would be).  bad counts out-of-range indices.
__global__ void k_index_rows(const int* __restrict__ slot_idx, const int* __restrict__ row_img, const float* __restrict__ bmask,
                             int B, int LR, int Rb, int n_img, int* __restrict__ ridx, float* __restrict__ rmask,
                             int* __restrict__ bad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)B * LR) return;
    const int b = (int)(i / LR);
    const int img = row_img ? row_img[b] : b;
    const int id = slot_idx[i];
    int row = -1;
    if (id >= Rb || id < -1 || img < 0 || img >= n_img) atomicAdd(bad, 1);
    else if (id >= 0) row = img * Rb + id;
    ridx[i] = row;
    rmask[i] = row >= 0 ? bmask[row] : 0.f;
}

// Slot re-ordering of the eval loop on index lists (eval_coco.py:222-241): one block per caption.
//   recons[j] = slots[rank[j]] for j < len(rank) (rank padded with -1), empty otherwise      (:222-229 perm_matrix . slots)
//   empty slots (no row with a non-zero bank row) are dropped, order kept                     (:230)
//   the last kept slot is replicated to the end                                               (:232-234)
//   verbs[j] = verbs_in[rank[j]] or -1 where the permutation has no row j; NOT compacted      (:237-238)
__global__ __launch_bounds__(64) void k_reorder_slots(const int* __restrict__ slot_in, const int* __restrict__ rank,
                                                      const float* __restrict__ verbs_in, const float* __restrict__ bmask,
                                                      const int* __restrict__ row_img, int L, int R, int Rb,
                                                      int* __restrict__ slot_out, float* __restrict__ verbs_out) {
    extern __shared__ int sh[];
    int* src = sh;                 // L: source slot of output position j after compaction, -1 = none
    const int n = blockIdx.x, tid = threadIdx.x;
    const int img = row_img ? row_img[n] : n;
    const int* S = slot_in + (long long)n * L * R;
    if (tid == 0) {
        int kept = 0;
        for (int j = 0; j < L; ++j) {
            const int rk = rank[(long long)n * L + j];
            bool live = false;
            if (rk >= 0 && rk < L)
                for (int r = 0; r < R && !live; ++r) {
                    const int id = S[rk * R + r];
                    live = id >= 0 && id < Rb && (!bmask || bmask[(long long)img * Rb + id] != 0.f);
                }
            if (live) src[kept++] = rk;
        }
        for (int j = kept; j < L; ++j) src[j] = kept > 0 ? src[kept - 1] : -1;
    }
    __syncthreads();
    for (int i = tid; i < L * R; i += 64) {
        const int j = i / R, r = i - j * R;
        slot_out[(long long)n * L * R + i] = src[j] >= 0 ? S[src[j] * R + r] : -1;
    }
    if (verbs_in)
        for (int j = tid; j < L; j += 64) {
            const int rk = rank[(long long)n * L + j];
            verbs_out[(long long)n * L + j] = (rk >= 0 && rk < L) ? verbs_in[(long long)n * L + rk] : -1.f;
        }
}

// Three small launches (a single 1024-thread block walking its rows serially took 42-100 us): per-256-row counts, an exclusive scan of the (<= 4096 x 1024) block counts, ordered writes.
__global__ __launch_bounds__(256) void k_compact_count(const float* __restrict__ mask, int rows, int* __restrict__ bcount) {
    __shared__ int wsum[4];
    const int r = blockIdx.x * 256 + threadIdx.x;
    const bool v = r < rows && mask[r] != 0.f;
    const int n = __popcll(__ballot(v));
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(1024) void k_compact_scan(int* __restrict__ bcount, int nblocks, int* __restrict__ total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = tid * per, b1 = min(nblocks, b0 + per);
    int n = 0;
    for (int b = b0; b < b1; ++b) n += bcount[b];
    part[tid] = n;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - n;                       // exclusive prefix of this thread's blocks
    for (int b = b0; b < b1; ++b) { const int c = bcount[b]; bcount[b] = run; run += c; }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(256) void k_compact_write(const float* __restrict__ mask, int rows, const int* __restrict__ boffset,
                                                       int* __restrict__ vlist) {
    __shared__ int wsum[4];
    const int r = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool v = r < rows && mask[r] != 0.f;
    const unsigned long long bal = __ballot(v);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int base = boffset[blockIdx.x];
    for (int w = 0; w < wave; ++w) base += wsum[w];
    if (v) vlist[base + __popcll(bal & ((1ull << lane) - 1ull))] = r;
}

// Caller-supplied bound on the non-padding rows (vsr_set_valid_rows_bound): the list is padded up to `bound` entries with its first row
// (the projection GEMM is sized by the bound, the scatter below stops at the real count, which never leaves the device).
// counts: [0] rows found, [1] bad slot indices of the index-list format, [2] the bad-id counter, [3] rows beyond the bound (reported by vsr_bad_ids)
__global__ void k_pad_row_list(int* __restrict__ vlist, int* __restrict__ counts, int bound) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = counts[0];
    if (i == 0) {
        if (n > bound) counts[3] = n - bound;
        if (counts[1] != 0) atomicAdd(counts + 2, counts[1]);
    }
    if (i >= n && i < bound) vlist[i] = n > 0 ? vlist[0] : 0;
}

// rows vlist[bound .. n) of P = 0: the non-padding rows a caller's bound left without a projection (n = counts[0] on the device)
__global__ void k_zero_rows_beyond(const int* __restrict__ vlist, const int* __restrict__ counts, int bound, int A, float* __restrict__ P) {
    const int n = counts[0];
    for (int i = bound + (int)blockIdx.x; i < n; i += (int)gridDim.x) {
        float* row = P + (long long)vlist[i] * A;
        for (int a = threadIdx.x; a < A; a += blockDim.x) row[a] = 0.f;
    }
}

// P[vlist[m]] = sum of the slabs' row m   (scatter of the compact projection back to the dense row index)
// rows_dev (optional): the number of real rows lives on the device (the launch covers `rows` = the caller's bound)
__global__ void k_slab_reduce_scatter(const float* __restrict__ slabs, int nslab, long long stride, int rows, int A,
                                      const int* __restrict__ vlist, float* __restrict__ P, const int* __restrict__ rows_dev = nullptr) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (rows_dev && *rows_dev < rows) rows = *rows_dev;
    if (i >= (long long)rows * A) return;
    const int m = (int)(i / A), a = (int)(i % A);
    float s = slab_sum(slabs + i, nslab, stride);
    P[(long long)vlist[m] * A + a] = s;
}

// hoisted image part of the LSTM1 / gate pre-activations: sum the split-K slabs and fold in all biases.
// n in [0,4H): b_ih + b_hh of lstm_cell_1;  [4H,5H): W1_is.bias + W1_hs.bias;  [5H,6H): W1_ig.bias + W1_hg.bias
__global__ void k_vproj_finish(const float* __restrict__ slabs, int nsplit, long long stride, int B, int H,
                               const float* b_ih, const float* b_hh, const float* b_is, const float* b_hs,
                               const float* b_ig, const float* b_hg, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = 6 * H;
    if (i >= (long long)B * N) return;
    const int n = (int)(i % N);
    float s = slab_sum(slabs + i, nsplit, stride);
    float bias;
    if (n < 4 * H) bias = b_ih[n] + b_hh[n];
    else if (n < 5 * H) bias = b_is[n - 4 * H] + b_hs[n - 4 * H];
    else bias = b_ig[n - 5 * H] + b_hg[n - 5 * H];
    out[i] = s + bias;
}

__global__ void k_slab_reduce(const float* __restrict__ slabs, int nsplit, long long stride, long long n, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = slab_sum(slabs + i, nsplit, stride);
    out[i] = s;
}

// decode state at t = 0: slot pointer 0, previous word = bos (init_state :109-115; the h/c states are one memset)
__global__ void k_init_rows(int* __restrict__ slot, int* __restrict__ word, int bos, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { slot[i] = 0; word[i] = bos; }
}

// f16x2 flavour: a state handed to vsr_step by the caller is an A operand of class "unit" (|x| < 2 at the fixed scale 2^15: what
// sigmoid x tanh produces); elements outside that range would overflow fp16 in the GEMMs' in-kernel split.  They are COUNTED into the
// input-contract counter (vsr_bad_ids) - the f32x3 / f32 flavours accept any state.
__global__ void k_count_outside_unit(const float* __restrict__ x, long long n, int* __restrict__ count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool bad = i < n && !(fabsf(x[i]) < 1.9990234375f);          // (NaN counts too)
    const unsigned long long b = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}

__global__ void k_fill_i32(int* p, int v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// int64 ids of the caller -> int32, range-checked: an id outside [0, hi) is counted in *bad and clamped, so that a
// padding id of -1 or an id >= V never indexes the embedding / projection tables out of bounds (nn.Embedding raises
// there; here the count is reported by vsr_bad_ids()).
__global__ void k_i64_to_i32(const int64_t* src, long long src_stride, int* dst, int n, int hi, int* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long v = src[(long long)i * src_stride];
    if (v < 0 || v >= hi) {
        atomicAdd(bad, 1);
        v = v < 0 ? 0 : hi - 1;
    }
    dst[i] = (int)v;
}
__global__ void k_i32_to_i64(const int* src, int64_t* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------------------- step
// LSTM1 + sentinel gate + image part of the shift gate                                (step :151-154, :181)
// pre: (nsplit, M, 6H) raw GEMM sums of [h2 | x | h1_old]; vproj: (B, 6H) hoisted vbar part + biases.
// xproj (optional): (V, 6H) cached projection of every embedding row (decode cache), gathered by word[row];
// nblk: number of leading gate blocks (of 6) that the GEMM produced (the rest only has hoisted terms).
// the cell itself: q = [i, f, g, o, s-gate, g-gate image part] pre-activations of one (row, unit)
__device__ __forceinline__ void lstm1_point(const float (&q)[6], float c_old, long long i, float* __restrict__ h1n, float* __restrict__ c1n,
                                            float* __restrict__ s_t, float* __restrict__ gpre, uint16_t* __restrict__ h1n16,
                                            uint16_t* __restrict__ s_t16, float isc) {
    const float c = sigmoidf_(q[1]) * c_old + sigmoidf_(q[0]) * tanhf(q[2]);
    const float tc = tanhf(c);
    const float h1v = sigmoidf_(q[3]) * tc, stv = sigmoidf_(q[4]) * tc;
    h1n[i] = h1v;
    c1n[i] = c;
    s_t[i] = stv;
    gpre[i] = q[5];
    if (h1n16) { img_store(h1n16, i, h1v, isc); img_store(s_t16, i, stv, isc); }
}

__global__ void k_lstm1(const float* __restrict__ pre, int nsplit, long long stride, const float* __restrict__ vproj,
                        int rpi, const int* __restrict__ parent, const float* __restrict__ c1_old, int M, int H,
                        float* __restrict__ h1n, float* __restrict__ c1n, float* __restrict__ s_t, float* __restrict__ gpre,
                        const float* __restrict__ xproj, const int* __restrict__ word, int nblk, int pre_by_parent,
                        uint16_t* __restrict__ h1n16, uint16_t* __restrict__ s_t16 /* optional images (img_store) */, float isc = 0.f,
                        int skip5 = 0 /* leading slabs that gate block 5 does not have (round 6: the h1 part of the sums comes from the S5 launch, and the shift gate has none) */) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * H) return;
    const int row = (int)(i / H), j = (int)(i % H);
    const int prow = parent ? parent[row] : row;
    // pre_by_parent: the sums were produced one step early over the PRE-selection rows (merged with the previous
    // step's vocabulary GEMM), so this row reads the sums of the hypothesis it descends from
    const long long base = (long long)(pre_by_parent ? prow : row) * 6 * H + j;
    const float* vp = vproj + (long long)(row / rpi) * 6 * H + j;
    const float* xp = xproj ? xproj + (long long)word[row] * 6 * H + j : nullptr;
    float q[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int sk = g == 5 ? skip5 : 0;
        float s = g < nblk ? slab_sum(pre + base + (long long)g * H + sk * stride, nsplit - sk, stride) : 0.f;
        if (xp) s += xp[(long long)g * H];
        q[g] = s + vp[(long long)g * H];
    }
    lstm1_point(q, c1_old[(long long)prow * H + j], i, h1n, c1n, s_t, gpre, h1n16, s_t16, isc);
}

// reduce the slabs of h1 -> [W1_hg | att_ha] and s_t -> [s_fc | att_sa]; finish the shift-gate vector
// g_t = sigmoid(gpre + W1_hg h1_new) * tanh(c1_new)                                     (step :155, :181-182)
__global__ void k_gate2(const float* __restrict__ c2a, const float* __restrict__ c2b, int nsplit, long long stride_a,
                        long long stride_b, const float* __restrict__ gpre, const float* __restrict__ c1n,
                        const float* __restrict__ b_sfc, int M, int H, int A, int D, float* __restrict__ g_t,
                        float* __restrict__ hA, float* __restrict__ sent, float* __restrict__ sa,
                        float* __restrict__ gates6 /* optional (M,6H): column block 5 receives the shift gate */,
                        uint16_t* __restrict__ g_t16 = nullptr /* optional image of g_t (img_store) */, float isc = 0.f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = H + A + D + A;
    if (i >= (long long)M * W) return;
    const int row = (int)(i / W), c = (int)(i % W);
    if (c < H + A) {
        float s = slab_sum(c2a + (long long)row * (H + A) + c, nsplit, stride_a);
        if (c < H) {
            const long long o = (long long)row * H + c;
            const float gg = sigmoidf_(gpre[o] + s);
            const float gv = gg * tanhf(c1n[o]);
            g_t[o] = gv;
            if (g_t16) img_store(g_t16, o, gv, isc);
            if (gates6) gates6[(long long)row * 6 * H + 5LL * H + c] = gg;
        } else {
            hA[(long long)row * A + (c - H)] = s;
        }
    } else {
        const int cc = c - (H + A);
        float s = slab_sum(c2b + (long long)row * (D + A) + cc, nsplit, stride_b);
        if (cc < D) sent[(long long)row * D + cc] = s + b_sfc[cc];
        else sa[(long long)row * A + (cc - D)] = s;
    }
}

// adaptive attention over [sentinel ; regions of the current slot]                    (step :158-171, :187)
// one workgroup per row (512 threads at D >= 2048: one float4 column group per thread, half the dependent load
// rounds of the weighted sum; results do not depend on the size); regions / projections are indexed by (image, slot), never copied.
//   z_det[r] = w_a . tanh(P[img,slot,r,:] + hA)      z_sent = w_s . tanh(sa + hA)
//   alpha    = softmax([z_sent ; z_det]) * mask ; alpha /= sum(alpha)
//   att      = alpha_0 * sentinel + sum_r alpha_r * regions[img,slot,r,:]
//   zsum     = sum_r mask_r * z_det[r]   (raw logits: the "shift" logit of the gate)
// When g2.c2a is set the kernel first does the row's share of k_gate2 itself (sums of the S2 slabs -> g_t to global for
// the next GEMM, hA to LDS and global, s_a and the sentinel to LDS only): one launch and one 8 KB round trip per row less.
struct Gate2Args {
    const float* c2a; const float* c2b; int nsplit; long long stride_a, stride_b;
    const float* gpre; const float* c1n; const float* b_sfc; int H;
    float* g_t; float* hA_out;
    uint16_t* g_t16 = nullptr;      // optional image of g_t (img_store)
    float isc = 0.f;                // ... its kind / scale; att16 of k_attend: bf16 when 0, else fp16 pairs scaled by 2^*att_exp
};

// (Round 4 measured the split the round-3 review asked for - this kernel stopping behind the softmax, a second kernel forming the weighted
// sums per IMAGE so that hypotheses on one slot share the region rows they read: bit-identical, and SLOWER end to end, 297.8 k against
// 345.5 k tokens/s beam-5 in one run (profiles/r04_d_attention_split_ab.txt): the scores kernel alone takes 22.7 of the fused kernel's 31 us,
// the weighted sums were never its long pole, and the second launch adds its own dependent round trips.  Not kept.)
template <int NT>
__global__ __launch_bounds__(NT) void k_attend(const Gate2Args g2, const float* __restrict__ hA, const float* __restrict__ sa,
                                                const float* __restrict__ sent, const float* __restrict__ P,
                                                const float* __restrict__ regions, const float* __restrict__ rmask,
                                                const int* __restrict__ ridx,
                                                const int* __restrict__ slot, int fixed_slot, int rpi, int M, int L,
                                                int R, int A, int D, const float* __restrict__ w_a,
                                                const float* __restrict__ w_s, float* __restrict__ att,
                                                float* __restrict__ zsum, float* __restrict__ alpha_out,
                                                uint16_t* __restrict__ att16 = nullptr /* optional image of att */,
                                                const int* __restrict__ att_exp = nullptr,
                                                int nparts = 1 /* workgroups per row (round 6): every part redoes the row's slab sums, scores and softmax
                                                                  (same arithmetic, same values) and forms 1 / nparts of the D columns of the weighted sum -
                                                                  a launch of <= 128 rows left half the CUs idle while every row waited for ONE CU's ingest */) {
    extern __shared__ float sm[];
    float* hA_s = sm;             // A
    float* sa_s = hA_s + A;       // A   (fused gate2 only)
    float* sent_s = sa_s + A;     // D   (fused gate2 only)
    float* z_s = sent_s + D;      // R + 1  (then alpha)
    float* red = z_s + R + 1;     // 8
    int* ri_s = reinterpret_cast<int*>(red + 8);   // R: row of P / regions behind slot entry r (dense: its own row)
    const int item = xcd_item(M * nparts);
    if (item < 0) return;
    const int row = item / nparts, part = item - row * nparts;
    const bool first = part == 0;                  // the part that stores what is not split (g_t, hA, alpha, zsum)
    constexpr int NW = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int img = row / rpi;
    const int k = slot ? slot[row] : fixed_slot;
    const long long sl = (long long)img * L + k;
    const bool fused = g2.c2a != nullptr;
    if (fused) {
        // Slab sums of the S2 GEMM for this row.  FOUR columns per thread and pass with every slab load (and the g_t operands)
        // issued before the first use: one column per loop iteration made each iteration's loads wait for the previous
        // iteration's global store (8 dependent L2 round trips per row: 10 of the kernel's 34 us).
        const int H = g2.H;
        if (((H | A | D) & 3) == 0) {
            // 16-byte path: four consecutive columns per thread, every slab load of a pass issued before the first use
            for (int c = tid * 4; c < H + A; c += NT * 4) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < g2.nsplit) v[k] = *reinterpret_cast<const float4*>(g2.c2a + (long long)k * g2.stride_a + (long long)row * (H + A) + c);
                float4 gp = make_float4(0, 0, 0, 0), cn = gp;
                if (c < H) {
                    gp = *reinterpret_cast<const float4*>(g2.gpre + (long long)row * H + c);
                    cn = *reinterpret_cast<const float4*>(g2.c1n + (long long)row * H + c);
                }
                float4 s = v[0];
#pragma unroll
                for (int k = 1; k < 8; ++k)
                    if (k < g2.nsplit) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
                if (c < H) {
                    float4 o;
                    o.x = sigmoidf_(gp.x + s.x) * tanhf(cn.x); o.y = sigmoidf_(gp.y + s.y) * tanhf(cn.y);
                    o.z = sigmoidf_(gp.z + s.z) * tanhf(cn.z); o.w = sigmoidf_(gp.w + s.w) * tanhf(cn.w);
                    if (first) {
                        *reinterpret_cast<float4*>(g2.g_t + (long long)row * H + c) = o;
                        if (g2.g_t16) img_store4(g2.g_t16, (long long)row * H + c, o, g2.isc);
                    }
                } else {
                    *reinterpret_cast<float4*>(hA_s + (c - H)) = s;
                    if (first) *reinterpret_cast<float4*>(g2.hA_out + (long long)row * A + (c - H)) = s;
                }
            }
            for (int cc = tid * 4; cc < D + A; cc += NT * 4) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < g2.nsplit) v[k] = *reinterpret_cast<const float4*>(g2.c2b + (long long)k * g2.stride_b + (long long)row * (D + A) + cc);
                float4 s = v[0];
#pragma unroll
                for (int k = 1; k < 8; ++k)
                    if (k < g2.nsplit) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
                if (cc < D) {
                    const float4 bs = *reinterpret_cast<const float4*>(g2.b_sfc + cc);
                    s.x += bs.x; s.y += bs.y; s.z += bs.z; s.w += bs.w;
                    *reinterpret_cast<float4*>(sent_s + cc) = s;
                } else {
                    *reinterpret_cast<float4*>(sa_s + (cc - D)) = s;
                }
            }
        } else {
            for (int c = tid; c < H + A; c += NT) {
                const float s = slab_sum(g2.c2a + (long long)row * (H + A) + c, g2.nsplit, g2.stride_a);
                if (c < H) {
                    const float gv = sigmoidf_(g2.gpre[(long long)row * H + c] + s) * tanhf(g2.c1n[(long long)row * H + c]);
                    if (first) {
                        g2.g_t[(long long)row * H + c] = gv;
                        if (g2.g_t16) img_store(g2.g_t16, (long long)row * H + c, gv, g2.isc);
                    }
                } else {
                    hA_s[c - H] = s;
                    if (first) g2.hA_out[(long long)row * A + (c - H)] = s;
                }
            }
            for (int cc = tid; cc < D + A; cc += NT) {
                const float s = slab_sum(g2.c2b + (long long)row * (D + A) + cc, g2.nsplit, g2.stride_b);
                if (cc < D) sent_s[cc] = s + g2.b_sfc[cc];
                else sa_s[cc - D] = s;
            }
        }
    } else {
        for (int a = tid; a < A; a += NT) hA_s[a] = hA[(long long)row * A + a];
    }
    const float* sa_row = fused ? sa_s : sa + (long long)row * A;
    const float* srow = fused ? sent_s : sent + (long long)row * D;
    for (int r = tid; r < R; r += NT) {
        const int e = ridx ? ridx[sl * R + r] : (int)(sl * R + r);
        ri_s[r] = e < 0 ? 0 : e;                   // padding entries are masked and never dereferenced
    }
    __syncthreads();

    // scores: wave w takes rows w, w+4, ... of [regions ; sentinel]; four rows per pass so that their projection
    // loads are all in flight before the first tanh (one L2 round trip per pass instead of one per row)
    const float* mk_row = rmask + sl * R;
    constexpr int QN = NT == 512 ? 5 : 4;        // rows per wave and pass: 8 waves x 5 cover the 37 score rows of R = 36 at once
    for (int r0 = wave; r0 < R + 1; r0 += QN * NW) {
        float sc[QN];
#pragma unroll
        for (int q = 0; q < QN; ++q) sc[q] = 0.f;
        for (int a = lane * 4; a < A; a += 256) {
            float4 p[QN];
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const int r = r0 + NW * q;
                const float* src = (r < R) ? P + (long long)ri_s[r] * A : sa_row;
                // padding rows were never projected (att_va(0) = 0): their P entry is not defined, use the exact zero
                const bool live = r < R + 1 && (r >= R || mk_row[r] != 0.f);
                p[q] = live ? *reinterpret_cast<const float4*>(src + a) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            const float4 h = *reinterpret_cast<const float4*>(hA_s + a);
            const float4 wa = *reinterpret_cast<const float4*>(w_a + a);
            const float4 ws = *reinterpret_cast<const float4*>(w_s + a);
#pragma unroll
            for (int q = 0; q < QN; ++q) {
                const int r = r0 + NW * q;
                // a padding row's score never reaches the output (alpha = mask * softmax = 0, and the shift logit sums
                // mask * z): its 512 tanh are skipped (wave-uniform: a wave owns whole rows); its z stays 0
                const bool live = r < R + 1 && (r >= R || mk_row[r] != 0.f);
                if (!live) continue;
                const float4 w = (r < R) ? wa : ws;
                sc[q] += w.x * tanhf(p[q].x + h.x);
                sc[q] += w.y * tanhf(p[q].y + h.y);
                sc[q] += w.z * tanhf(p[q].z + h.z);
                sc[q] += w.w * tanhf(p[q].w + h.w);
            }
        }
#pragma unroll
        for (int q = 0; q < QN; ++q) {
            const int r = r0 + NW * q;
            const float v = wave_sum(sc[q]);
            if (lane == 0 && r < R + 1) z_s[(r < R) ? r + 1 : 0] = v;
        }
    }
    // sentinel row-sum for its mask
    float ss = 0.f;
    for (int d = tid * 4; d < D; d += 4 * NT) {
        const float4 v = *reinterpret_cast<const float4*>(srow + d);
        ss += (v.x + v.y) + (v.z + v.w);
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();

    if (wave == 0) {
        float ssum = 0.f;
        for (int w = 0; w < NW; ++w) ssum += red[w];
        const float m0 = (ssum != 0.f) ? 1.f : 0.f;
        const float* mk = rmask + sl * R;
        // R + 1 <= 64 handled by one pass per 64 entries
        float mx = -INFINITY;
        for (int j = lane; j < R + 1; j += 64) mx = fmaxf(mx, z_s[j]);
        mx = wave_max(mx);
        float se = 0.f, zs = 0.f;
        for (int j = lane; j < R + 1; j += 64) se += expf(z_s[j] - mx);
        se = wave_sum(se);
        float s2 = 0.f;
        for (int j = lane; j < R + 1; j += 64) {
            const float m = (j == 0) ? m0 : mk[j - 1];
            const float z = z_s[j];
            if (j > 0) zs += m * z;
            s2 += (expf(z - mx) / se) * m;
        }
        s2 = wave_sum(s2);
        zs = wave_sum(zs);
        for (int j = lane; j < R + 1; j += 64) {
            const float m = (j == 0) ? m0 : mk[j - 1];
            const float al = ((expf(z_s[j] - mx) / se) * m) / s2;
            z_s[j] = al;
            if (alpha_out && first) alpha_out[(long long)row * (R + 1) + j] = al;
        }
        if (lane == 0 && first) zsum[row] = zs;
    }
    __syncthreads();

    // weighted sum.  Rows with alpha == 0 (zero padding) are skipped: no HBM read for them.  Four region rows are in
    // flight per thread (independent loads) so that the 8 KB rows stream instead of paying one L2/HBM latency each.
    const float a0 = z_s[0];
    const float att_isc = att_exp ? __int_as_float((127 + *att_exp) << 23) : 0.f;      // 2^exponent of the attended vector's bound class
    const int Dp = D / nparts;                      // (nparts > 1 only when D is a multiple of 4 nparts: run_step)
    for (int d = part * Dp + tid * 4; d < (part + 1) * Dp; d += 4 * NT) {
        const float4 s = *reinterpret_cast<const float4*>(srow + d);
        float4 acc = make_float4(a0 * s.x, a0 * s.y, a0 * s.z, a0 * s.w);
        int r = 0;
        constexpr int CH = 18;                     // region rows in flight per thread (two passes cover R = 36)
        for (; r + CH <= R; r += CH) {
            float al[CH];
            bool any = false;
#pragma unroll
            for (int q = 0; q < CH; ++q) { al[q] = z_s[r + 1 + q]; any |= al[q] != 0.f; }
            if (!any) continue;
            float4 x[CH];
#pragma unroll
            for (int q = 0; q < CH; ++q)
                x[q] = al[q] != 0.f ? *reinterpret_cast<const float4*>(regions + (long long)ri_s[r + q] * D + d) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                acc.x += al[q] * x[q].x; acc.y += al[q] * x[q].y; acc.z += al[q] * x[q].z; acc.w += al[q] * x[q].w;
            }
        }
        for (; r < R; ++r) {
            const float al = z_s[r + 1];
            if (al != 0.f) {
                const float4 x = *reinterpret_cast<const float4*>(regions + (long long)ri_s[r] * D + d);
                acc.x += al * x.x; acc.y += al * x.y; acc.z += al * x.z; acc.w += al * x.w;
            }
        }
        *reinterpret_cast<float4*>(att + (long long)row * D + d) = acc;
        if (att16) img_store4(att16, (long long)row * D + d, acc, att_isc);
    }
}

// LSTM2 pointwise                                                                     (step :176-177)
__global__ void k_lstm2(const float* __restrict__ pre, int nsplit, long long stride, const float* __restrict__ b_ih,
                        const float* __restrict__ b_hh, const float* __restrict__ vproj2, int rpi,
                        const int* __restrict__ parent, const float* __restrict__ c2_old, int M, int H,
                        float* __restrict__ h2n, float* __restrict__ c2n, uint16_t* __restrict__ h2n16 = nullptr /* optional image (img_store) */,
                        float isc = 0.f) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * H) return;
    const int row = (int)(i / H), j = (int)(i % H);
    const long long base = (long long)row * 4 * H + j;
    float q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s = slab_sum(pre + base + (long long)g * H, nsplit, stride);
        s += b_ih[g * H + j] + b_hh[g * H + j];
        if (vproj2) s += vproj2[(long long)(row / rpi) * 4 * H + g * H + j];
        q[g] = s;
    }
    const int prow = parent ? parent[row] : row;
    const float c = sigmoidf_(q[1]) * c2_old[(long long)prow * H + j] + sigmoidf_(q[0]) * tanhf(q[2]);
    const float h2v = sigmoidf_(q[3]) * tanhf(c);
    h2n[i] = h2v;
    c2n[i] = c;
    if (h2n16) img_store(h2n16, i, h2v, isc);
}

// shift-gate log-probabilities: z_g = w_g . tanh(att_ga g_t + hA); gate = log_softmax([z_g, zsum])   (:184-188)
// verb-forced rows get [-1e3, 0] (step_v :271, :295).  one wave per row.
struct GateLogitArgs {
    const float* ga; int nsplit; long long stride;       // (nsplit, M, A) raw sums of att_ga(g_t)
    const float* hA; const float* w_g; const float* zsum; const float* verbs; const int* slot;
    int rpi, L, M, A;
    float* lg; long long lg_stride;
    float* ga_out = nullptr;                             // optional (M, A): the slab sums (the training forward saves them)
};

// one row by a whole workgroup (the vocabulary kernel's blocks do their row's gate logits on the side; k_fwd_tail of the
// training forward gives every row a workgroup): one column per thread, every slab of it in flight at once, wave sums
// combined in wave order
template <int NT>
__device__ __forceinline__ void gatelogit_block(const GateLogitArgs& g, int row, float* red /* NT / 64 floats of LDS */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s = 0.f;
    for (int a = tid; a < g.A; a += NT) {
        const float x = slab_sum(g.ga + (long long)row * g.A + a, g.nsplit, g.stride);
        if (g.ga_out) g.ga_out[(long long)row * g.A + a] = x;
        s += g.w_g[a] * tanhf(x + g.hA[(long long)row * g.A + a]);
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        float a = 0.f;
        for (int w = 0; w < NT / 64; ++w) a += red[w];
        const float b = g.zsum[row];
        const float mx = fmaxf(a, b);
        const float lse = mx + logf(expf(a - mx) + expf(b - mx));
        float l0 = a - lse, l1 = b - lse;
        if (g.verbs) {
            const float v = g.verbs[(long long)(row / g.rpi) * g.L + g.slot[row]];
            if (v != -1.f) { l0 = -1e3f; l1 = 0.f; }
        }
        g.lg[(long long)row * g.lg_stride] = l0;
        g.lg[(long long)row * g.lg_stride + 1] = l1;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------- vocab rows
struct Philox {
    // Philox4x32-10 (Salmon et al., SC'11)
    static __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static __device__ __forceinline__ void gen(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
        uint32_t c[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int i = 0; i < 10; ++i) { round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
    static __device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
};

struct TopEntry { float v; int i; };
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

enum VocabMode { VM_TOPK = 0, VM_SAMPLE = 1, VM_FORCED = 2, VM_FULL = 3 };

// Per row: logits = sum of slabs + bias; log-sum-exp; then by mode
//   VM_TOPK   K best (log-prob, id) pairs (K = 1: greedy arg-max)          (CaptioningModel.py:47, :152)
//   VM_SAMPLE Gumbel-max draw from Categorical(logits) + its log-prob      (:66-70)
//   VM_FORCED log-prob of a given id (sampling replay)
//   VM_FULL   the whole log_softmax row is written to full_out             (step :178, forward :34)
// Verb-forced rows (step_v :268-293) emit one word with log-prob 0 and -1e6 elsewhere.
template <int K, int NT>
__global__ __launch_bounds__(NT) void k_vocab(const float* __restrict__ logits, int nsplit, long long stride,
                                               const float* __restrict__ bias, int M, int V, int mode,
                                               float* __restrict__ top_v, int* __restrict__ top_i,
                                               float* __restrict__ full_out, long long full_stride,
                                               const int* __restrict__ forced, uint64_t seed, uint32_t t,
                                               const float* __restrict__ verbs, const int* __restrict__ slot, int rpi,
                                               int L, int gt, const int* __restrict__ vt_ptr,
                                               const int* __restrict__ vt_ids, int n_verbs, int lds_row,
                                               const GateLogitArgs gate, int* __restrict__ bad) {
    constexpr int NW = NT / 64;
    __shared__ float sv[NT * K];
    __shared__ int si[NT * K];
    __shared__ float red[2 * NW];
    __shared__ int redi[NW];
    __shared__ int pick_s;
    extern __shared__ float lrow[];          // V floats when the launch passes dynamic LDS: the combined row is
    const bool use_lds = lds_row != 0;       // summed from the slabs ONCE and the later passes read it from LDS
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = logits + (long long)row * V;
    const int V4 = (V + 3) & ~3;
    const bool vec = ((V & 3) == 0);
    // Slab sums for VU float4 groups per thread are loaded together (every slab of every group in flight at once).
    constexpr int VU = 5, KU = NT == 512 ? 2 : 4;      // 512 threads: <= 128 VGPRs so that two rows share a CU
    auto load_parts = [&](int vb, float4 (&part)[VU][KU + 1]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int v0 = vb + 4 * NT * u;
            const bool in = v0 < V4;
            part[u][0] = in ? *reinterpret_cast<const float4*>(bias + v0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < KU; ++k)
                part[u][k + 1] = (in && k < nsplit) ? *reinterpret_cast<const float4*>(src + k * stride + v0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // the gate logits of this row (independent of the vocabulary work; nothing reads them before the selection kernel).  (Round 5: requesting
    // the row's first round of loads BEFORE this block - so that its two barriers run under their round trip - keeps 60 more registers
    // live across it: 155 VGPRs instead of 125, one row per CU instead of two.  Not kept.)
    if (gate.M > 0) gatelogit_block<NT>(gate, row, red);

    // ---- verb forcing: thread 0 resolves the forced word, everybody takes the short path
    int verb = -1;
    if (verbs) {
        const float vf = verbs[(long long)(row / rpi) * L + slot[row]];
        verb = (vf != -1.f) ? (int)vf : -1;
    }
    if (verb != -1) {
        if (tid == 0) {
            int pick = 0;
            if (gt) {
                pick = verb;
                if (verb < 0 || verb >= V) { atomicAdd(bad, 1); pick = 0; }     // the reference's out[i, verb] = 0 would raise
            } else if (verb >= 0 && verb < n_verbs && vt_ptr[verb + 1] > vt_ptr[verb]) {
                float best = -1e6f;
                pick = -1;
                // the reference compares log-probs; logits differ by the row constant lse, so compare
                // logit - lse > -1e6 <=> always true for finite logits: first strict maximum wins
                for (int q = vt_ptr[verb]; q < vt_ptr[verb + 1]; ++q) {
                    const int id = vt_ids[q];
                    float x = bias[id];
                    for (int k = 0; k < nsplit; ++k) x += src[k * stride + id];
                    if (pick < 0 || x > best) { best = x; pick = id; }
                }
            }
            pick_s = pick;
        }
        __syncthreads();
        const int pick = pick_s;
        if (mode == VM_FULL) {
            for (int v = tid; v < V; v += NT) full_out[(long long)row * full_stride + v] = (v == pick) ? 0.f : -1e6f;
        } else if (mode == VM_TOPK) {
            if (tid < K) {   // the forced word first, then the lowest other ids at -1e6 (ties are arbitrary in the reference)
                int id = (tid == 0) ? pick : ((tid - 1 < pick) ? tid - 1 : tid);
                top_v[(long long)row * K + tid] = (tid == 0) ? 0.f : -1e6f;
                top_i[(long long)row * K + tid] = id;
            }
        } else if (tid == 0) {
            int id = (mode == VM_FORCED) ? forced[row] : pick;   // Categorical over {0, -1e6...} is a point mass
            top_v[row] = (id == pick) ? 0.f : -1e6f;
            top_i[row] = id;
        }
        return;
    }

    // ---- pass 1: combined row (slabs + bias) -> LDS, row max, and every thread's single best key (logit, or
    // Gumbel-perturbed logit when sampling).  No per-thread top-K list: with 64 lanes some lane inserts at almost every
    // element, so a wave paid the whole insertion chain per element (5 800 VALU instructions per wave at K = 5).
    float bkey = -INFINITY;
    int bidx = 0x7fffffff;
    float mx = -INFINITY;
    for (int vb = tid * 4; vb < V4; vb += 4 * NT * VU) {
        float xs_all[VU][4];
        if (vec) {
            float4 part[VU][KU + 1];
            load_parts(vb, part);
#pragma unroll
            for (int u = 0; u < VU; ++u) {
                const int v0 = vb + 4 * NT * u;
                float4 a = part[u][0];
#pragma unroll
                for (int k = 0; k < KU; ++k)
                    if (k < nsplit) { a.x += part[u][k + 1].x; a.y += part[u][k + 1].y; a.z += part[u][k + 1].z; a.w += part[u][k + 1].w; }
                if (v0 < V4)
                    for (int k = KU; k < nsplit; ++k) {        // deeper splits than KU: the tail of the slab list, in order
                        const float4 s4 = *reinterpret_cast<const float4*>(src + k * stride + v0);
                        a.x += s4.x; a.y += s4.y; a.z += s4.z; a.w += s4.w;
                    }
                xs_all[u][0] = a.x; xs_all[u][1] = a.y; xs_all[u][2] = a.z; xs_all[u][3] = a.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < VU; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int v = vb + 4 * NT * u + e;
                    float x = 0.f;
                    if (v < V) {
                        x = bias[v];
                        for (int k = 0; k < nsplit; ++k) x += src[k * stride + v];
                    }
                    xs_all[u][e] = x;
                }
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int v0 = vb + 4 * NT * u;
            if (v0 >= V4) break;
            uint32_t rnd[4] = {0, 0, 0, 0};
            if (mode == VM_SAMPLE) Philox::gen(seed, (uint32_t)(v0 >> 2), (uint32_t)row, t, 0u, rnd);
            if (use_lds && vec) *reinterpret_cast<float4*>(lrow + v0) = make_float4(xs_all[u][0], xs_all[u][1], xs_all[u][2], xs_all[u][3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = v0 + e;
                if (v < V) {
                    const float x = xs_all[u][e];
                    if (use_lds && !vec) lrow[v] = x;
                    mx = fmaxf(mx, x);
                    float key = x;
                    if (mode == VM_SAMPLE) key = x - logf(-logf(Philox::u01(rnd[e])));
                    if (better(key, v, bkey, bidx)) { bkey = key; bidx = v; }
                }
            }
        }
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();

    auto getx = [&](int v) {
        if (use_lds) return lrow[v];
        float x = bias[v];
        for (int k = 0; k < nsplit; ++k) x += src[k * stride + v];
        return x;
    };
    // block arg-max of one (value, index) pair per thread; every thread returns the winner
    auto block_best = [&](float bv, int bi, float& gv, int& gi) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();                                   // red / redi of the previous round have been read
        if (lane == 0) { red[wave] = bv; redi[wave] = bi; }
        __syncthreads();
        gv = red[0];
        gi = redi[0];
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (better(red[w], redi[w], gv, gi)) { gv = red[w]; gi = redi[w]; }
    };

    // ---- threshold for the top-K: the K-th best of the NT thread maxima.  K distinct elements are >= it, so every one
    // of the row's K best is too; pass 2 collects the (few) elements that reach it.
    float tau = -INFINITY;
    if (mode == VM_TOPK && K > 1 && NW >= K) {
        // at least K waves: the K-th largest WAVE maximum is such a bound already (K distinct elements reach it), and the
        // wave maxima are in `red` from the row-max reduction: no selection rounds at all
        float wm[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) wm[w] = red[w];
#pragma unroll
        for (int q = 0; q < K; ++q) {
            float best = wm[0];
            int at = 0;
#pragma unroll
            for (int w = 1; w < NW; ++w)
                if (wm[w] > best) { best = wm[w]; at = w; }
            tau = best;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (w == at) wm[w] = -INFINITY;
        }
    } else if (mode == VM_TOPK && K > 1) {
        float cv = bkey;
        int ci = bidx;
        for (int round = 0; round < K; ++round) {
            float gv; int gi;
            block_best(cv, ci, gv, gi);
            if (ci == gi) { cv = -INFINITY; ci = 0x7fffffff; }      // taken (indices are unique; the -inf filler never matches a real one twice)
            tau = gv;
        }
    }

    // ---- pass 2: sum of exp, and the candidates for the top-K
    __shared__ int ncand;
    if (tid == 0) ncand = 0;
    __syncthreads();
    float se = 0.f;
    const bool collect = (mode == VM_TOPK && K > 1);
    auto visit = [&](float x, int v) {
        se += expf(x - mx);
        if (collect && x >= tau) {
            const int pos = atomicAdd(&ncand, 1);
            if (pos < NT * K) { sv[pos] = x; si[pos] = v; }
        }
    };
    if (use_lds && vec) {
        for (int v0 = tid * 4; v0 < V; v0 += 4 * NT) {
            const float4 x4 = *reinterpret_cast<const float4*>(lrow + v0);
            visit(x4.x, v0); visit(x4.y, v0 + 1); visit(x4.z, v0 + 2); visit(x4.w, v0 + 3);
        }
    } else {
        for (int v = tid; v < V; v += NT) visit(getx(v), v);
    }
    se = wave_sum(se);
    __syncthreads();
    if (lane == 0) red[NW + wave] = se;
    __syncthreads();
    float se_all = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) se_all += red[NW + w];
    const float lse = mx + logf(se_all);
    const int nc = ncand;

    if (mode == VM_FULL) {
        for (int v = tid; v < V; v += NT) full_out[(long long)row * full_stride + v] = getx(v) - lse;
        return;
    }
    if (mode == VM_FORCED) {
        if (tid == 0) {
            const int id = forced[row];
            top_v[row] = getx(id) - lse;
            top_i[row] = id;
        }
        return;
    }
    if (mode == VM_SAMPLE || K == 1) {                     // one winner: the best thread maximum
        float gv; int gi;
        block_best(bkey, bidx, gv, gi);
        if (tid == 0) {
            top_v[row] = (mode == VM_SAMPLE ? getx(gi) : gv) - lse;
            top_i[row] = gi;
        }
        return;
    }
    if (nc <= 64) {
        // ---- the usual case, a handful of candidates: wave 0 alone picks the K best, one candidate per lane
        if (wave != 0) return;
        float cv = lane < nc ? sv[lane] : -INFINITY;
        int ci = lane < nc ? si[lane] : 0x7fffffff;
        for (int round = 0; round < K; ++round) {
            float bv = cv;
            int bi = ci;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            }
            if (ci == bi) { cv = -INFINITY; ci = 0x7fffffff; }
            if (lane == 0) {
                top_v[(long long)row * K + round] = bv - lse;
                top_i[(long long)row * K + round] = bi;
            }
        }
        return;
    }
    if (nc <= NT * K) {
        // ---- K rounds of block arg-max over the candidate list (<= K entries per thread, normally <= 1 in total per thread)
        float lv[K];
        int li[K];
#pragma unroll
        for (int q = 0; q < K; ++q) {
            const int pos = tid + NT * q;
            lv[q] = pos < nc ? sv[pos] : -INFINITY;
            li[q] = pos < nc ? si[pos] : 0x7fffffff;
        }
        for (int round = 0; round < K; ++round) {
            float bv = lv[0];
            int bi = li[0];
#pragma unroll
            for (int q = 1; q < K; ++q)
                if (better(lv[q], li[q], bv, bi)) { bv = lv[q]; bi = li[q]; }
            float gv; int gi;
            block_best(bv, bi, gv, gi);
#pragma unroll
            for (int q = 0; q < K; ++q)
                if (li[q] == gi) { lv[q] = -INFINITY; li[q] = 0x7fffffff; }
            if (tid == 0) {
                top_v[(long long)row * K + round] = gv - lse;
                top_i[(long long)row * K + round] = gi;
            }
        }
        return;
    }
    // ---- more than NT K elements tie with the threshold (constant rows): K rounds of arg-max over the whole row,
    // each skipping the ids already emitted
    __shared__ int taken[K];
    for (int round = 0; round < K; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int v = tid; v < V; v += NT) {
            bool skip = false;
            for (int q = 0; q < round; ++q) skip |= taken[q] == v;
            const float x = getx(v);
            if (!skip && better(x, v, bv, bi)) { bv = x; bi = v; }
        }
        float gv; int gi;
        block_best(bv, bi, gv, gi);
        if (tid == 0) {
            taken[round] = gi;
            top_v[(long long)row * K + round] = gv - lse;
            top_i[(long long)row * K + round] = gi;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------- selection
// greedy / sampling / replay: next word and gate per row, advance the slot pointer     (:47, :66-70, step :135-140)
struct SelSimpleArgs {
    int mode; const float* top_v; const int* top_i; const float* lg; const int* forced_gate; uint64_t seed; uint32_t t;
    const int* slot; int L, M, T;
    int* word_next; int* gate_next; int* slot_next; int64_t* words; int64_t* gates; float* lp_w; float* lp_g;
};
// one row's selection; `write`: this thread stores the row's outputs.  Returns the word.
__device__ __forceinline__ int select_simple_row(const SelSimpleArgs& a, int row, bool write) {
    const float l0 = a.lg[row * 2], l1 = a.lg[row * 2 + 1];
    int g;
    if (a.mode == VM_TOPK) g = (l1 > l0) ? 1 : 0;                       // torch.max: first maximum on ties
    else if (a.mode == VM_FORCED) g = a.forced_gate[row];
    else {
        uint32_t rnd[4];
        Philox::gen(a.seed, 0xFFFFFFFFu, (uint32_t)row, a.t, 1u, rnd);
        g = (Philox::u01(rnd[0]) < expf(l0)) ? 0 : 1;
    }
    const int w = a.top_i[row];
    if (write) {
        a.word_next[row] = w;
        a.gate_next[row] = g;
        int k = a.slot[row] + g;
        a.slot_next[row] = k < 0 ? 0 : (k > a.L - 1 ? a.L - 1 : k);
        a.words[(long long)row * a.T + a.t] = w;
        a.gates[(long long)row * a.T + a.t] = g;
        if (a.lp_w) a.lp_w[(long long)row * a.T + a.t] = a.top_v[row];
        if (a.lp_g) a.lp_g[(long long)row * a.T + a.t] = g ? l1 : l0;
    }
    return w;
}
__global__ void k_select_simple(const SelSimpleArgs a) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.M) return;
    select_simple_row(a, row, true);
}

// k_select_simple of step t - 1 inside k_lstm1 of step t (round 5): the selection is per row and a handful of operations, so every
// thread of the LSTM1 kernel redoes its row's (the one with unit 0 stores it) instead of a 6 us launch of its own in front of it.
// Same arithmetic as the two kernels it replaces (select_simple_row, lstm1_point).  Greedy / sampling / replay; rows are their own parents.
__global__ void k_select_simple_lstm1(const SelSimpleArgs sel, const float* __restrict__ pre, int nsplit, long long stride,
                                      const float* __restrict__ vproj, const float* __restrict__ c1_old, int M, int H,
                                      float* __restrict__ h1n, float* __restrict__ c1n, float* __restrict__ s_t, float* __restrict__ gpre,
                                      const float* __restrict__ xproj, int nblk, uint16_t* __restrict__ h1n16, uint16_t* __restrict__ s_t16,
                                      float isc, int skip5) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * H) return;
    const int row = (int)(i / H), j = (int)(i % H);
    const long long base = (long long)row * 6 * H + j;
    const float* vp = vproj + (long long)row * 6 * H + j;
    // the slab sums do not depend on the selection: their loads are in flight while it is made
    float q[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int sk = g == 5 ? skip5 : 0;
        q[g] = g < nblk ? slab_sum(pre + base + (long long)g * H + sk * stride, nsplit - sk, stride) : 0.f;
    }
    const int w = select_simple_row(sel, row, j == 0);
    const float* xp = xproj + (long long)w * 6 * H + j;
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        float s = q[g];
        s += xp[(long long)g * H];
        q[g] = s + vp[(long long)g * H];
    }
    lstm1_point(q, c1_old[(long long)row * H + j], i, h1n, c1n, s_t, gpre, h1n16, s_t16, isc);
}

// joint (word x gate) beam selection, one wave per image                       (CaptioningModel.py:136-180)
// candidates: cb beams x K best words x 2 gates; score = seq + (lw + lg) in that association.
struct SelBeamArgs {
    int t, cb, beam, L; int64_t eos_w, eos_g;
    const float* top_v; const int* top_i; const float* lg; const int* slot; const int* word_prev; const int* gate_prev;
    const float* seq_in; float* seq_out; const float* mask_in; float* mask_out;
    int* word_next; int* gate_next; int* slot_next; int* parent_row;
    int* hist_parent; int* hist_word; int* hist_gate; float* hist_lpw; float* hist_lpg; int B;
};
// one wave selects the `beam` survivors of image b.  write: store them (one caller per image does); sel_s (optional, LDS):
// [q] = the survivor's parent beam j, [KMAX + q] = its word - for a caller that goes on with the survivors itself.
template <int K>
__device__ __forceinline__ void select_beam_wave(const SelBeamArgs& a, int b, int lane, bool write, int* sel_s) {
    const int t = a.t, cb = a.cb, beam = a.beam, L = a.L, B = a.B;
    const int64_t eos_w = a.eos_w, eos_g = a.eos_g;
    const float* __restrict__ top_v = a.top_v; const int* __restrict__ top_i = a.top_i; const float* __restrict__ lg = a.lg;
    const int* __restrict__ slot = a.slot; const int* __restrict__ word_prev = a.word_prev; const int* __restrict__ gate_prev = a.gate_prev;
    const float* __restrict__ seq_in = a.seq_in; const float* __restrict__ mask_in = a.mask_in;
    float* __restrict__ seq_out = a.seq_out; float* __restrict__ mask_out = a.mask_out;
    int* __restrict__ word_next = a.word_next; int* __restrict__ gate_next = a.gate_next; int* __restrict__ slot_next = a.slot_next;
    int* __restrict__ parent_row = a.parent_row; int* __restrict__ hist_parent = a.hist_parent; int* __restrict__ hist_word = a.hist_word;
    int* __restrict__ hist_gate = a.hist_gate; float* __restrict__ hist_lpw = a.hist_lpw; float* __restrict__ hist_lpg = a.hist_lpg;
    const int ncand = cb * K * 2;
    // each lane owns up to 2 candidates (ncand <= 128) and loads everything a winner will have to write with them: the selection
    // rounds below are shuffles only (a winner that went back to memory for its log-probs paid up to K dependent round trips)
    float cv[2], cmw[2], cmg[2], clw[2], clg[2];
    int cj[2], cw[2], cg[2], ck[2];
    long long cflat[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = lane + 64 * u;
        cv[u] = -INFINITY; cj[u] = 0; cw[u] = 0; cg[u] = 0; ck[u] = 0; cflat[u] = 0x7fffffffffffffffLL;
        cmw[u] = 1.f; cmg[u] = 1.f; clw[u] = 0.f; clg[u] = 0.f;
        if (c < ncand) {
            const int j = c / (2 * K), i = (c / 2) % K, g = c & 1;
            const int row = b * cb + j;
            // stream masks of the CURRENT beams (updated with the outputs selected at t-1)
            float mw = 1.f, mg = 1.f;
            if (t > 0) {
                mw = mask_in[(b * beam + j) * 2] * ((word_prev[row] != eos_w) ? 1.f : 0.f);
                mg = mask_in[(b * beam + j) * 2 + 1] * ((gate_prev[row] != eos_g) ? 1.f : 0.f);
            }
            const float seq = (t > 0) ? seq_in[b * beam + j] : 0.f;
            const float alive = fminf(fmaxf(mw + mg, 0.f), 1.f);
            const int wi = top_i[(long long)row * K + i];
            const float lw = top_v[(long long)row * K + i], lgv = lg[row * 2 + g];
            int w;
            float val;
            if (alive != 0.f) {
                w = wi;
                val = seq + (lw + lgv);
                clw[u] = lw;              // (ids are distinct within a row's list: the selection's word log-prob is this entry's)
            } else {            // frozen hypothesis: word 0 keeps the old score, everything else -999; its word log-prob reads 0
                w = i;
                val = (i == 0) ? seq : -999.f;
            }
            const int k = slot[row] + g;
            ck[u] = k < 0 ? 0 : (k > L - 1 ? L - 1 : k);
            cv[u] = val; cj[u] = j; cw[u] = w; cg[u] = g; cmw[u] = mw; cmg[u] = mg; clg[u] = lgv;
            cflat[u] = ((long long)j * 0x40000000LL + w) * 2 + g;
        }
    }
    for (int q = 0; q < beam; ++q) {
        // wave arg-max with deterministic tie-break on the flat (beam, word, gate) index
        int u_best = 0;
        if (cv[1] > cv[0] || (cv[1] == cv[0] && cflat[1] < cflat[0])) u_best = 1;
        float bv = cv[u_best];
        long long bf = cflat[u_best];
        int bl = lane * 2 + u_best;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const long long of = __shfl_xor(bf, o, 64);
            const int ol = __shfl_xor(bl, o, 64);
            if (ov > bv || (ov == bv && of < bf)) { bv = ov; bf = of; bl = ol; }
        }
        if ((bl >> 1) == lane) {
            const int u = bl & 1;
            const int j = cj[u], w = cw[u], g = cg[u];
            const int orow = b * beam + q, prow = b * cb + j;
            if (sel_s) { sel_s[q] = j; sel_s[KMAX + q] = w; }
            if (write) {
                seq_out[orow] = bv;
                word_next[orow] = w;
                gate_next[orow] = g;
                parent_row[orow] = prow;
                slot_next[orow] = ck[u];
                mask_out[orow * 2] = cmw[u];
                mask_out[orow * 2 + 1] = cmg[u];
                const long long hrow = (long long)t * B * beam + orow;
                hist_parent[hrow] = j;
                hist_word[hrow] = w;
                hist_gate[hrow] = g;
                // returned per-slot log-probs: log-prob of the selection, zeroed once its stream saw EOS
                hist_lpw[hrow] = clw[u] * cmw[u];
                hist_lpg[hrow] = clg[u] * cmg[u];
            }
            cv[u] = -INFINITY;
            cflat[u] = 0x7fffffffffffffffLL;
        }
    }
}
template <int K>
__global__ __launch_bounds__(64) void k_select_beam(const SelBeamArgs a) { select_beam_wave<K>(a, blockIdx.x, threadIdx.x, true, nullptr); }

// k_select_beam of step t - 1 inside k_lstm1 of step t (round 5).  One workgroup per (image, slice of SL_UB = 64 hidden units) of
// (beam + 1) waves: the last wave makes the image's selection (every slice's workgroup makes it - it is one wave's work - slice 0 stores
// it) WHILE wave p < cb adds the LSTM1 / gate slabs of the image's PARENT row p for the slice's units (the sums do not depend on the
// selection; a child row reads its parent's) into LDS; then wave q is CHILD row q: its parent's sums from LDS, its word's cached
// embedding projection, the image's hoisted terms - the expressions and their order are k_lstm1's (slab_sum, + xproj, + vproj,
// lstm1_point), and every (row, unit) element is one thread's work as there.  (A first version gave a thread all five rows of a unit:
// five dependent rounds of loads instead of one - slower than the two launches it replaced, profiles/r05_f_*.)
constexpr int SL_UB = 64;
template <int K>
__global__ __launch_bounds__((K + 1) * 64) void k_select_lstm1(const SelBeamArgs sel, const float* __restrict__ pre, int nsplit, long long stride,
                                                               const float* __restrict__ vproj, const float* __restrict__ c1_old, int H, int nslice,
                                                               float* __restrict__ h1n, float* __restrict__ c1n, float* __restrict__ s_t,
                                                               float* __restrict__ gpre, const float* __restrict__ xproj, int nblk,
                                                               uint16_t* __restrict__ h1n16, uint16_t* __restrict__ s_t16, float isc, int skip5) {
    __shared__ float sums[K * 7 * SL_UB];                  // per parent: six gate sums and its old cell state
    __shared__ int sel_s[2 * KMAX];
    const int b = blockIdx.x / nslice, slice = blockIdx.x % nslice;
    const int wave = threadIdx.x >> 6, u = threadIdx.x & 63, cb = sel.cb;
    const int j = slice * SL_UB + u;
    float vp[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (wave == K) {
        select_beam_wave<K>(sel, b, u, slice == 0, sel_s);
    } else if (j < H) {
        // everything that does not depend on the selection is requested before the barrier: the parents' sums and cell states (into LDS:
        // a child reads its parent's), the image's hoisted terms (the same for every row of the image)
#pragma unroll
        for (int g = 0; g < 6; ++g) vp[g] = vproj[(long long)b * 6 * H + (long long)g * H + j];
        if (wave < cb) {
            const long long base = (long long)(b * cb + wave) * 6 * H + j;
            const float co = c1_old[(long long)(b * cb + wave) * H + j];
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                const int sk = g == 5 ? skip5 : 0;
                sums[(wave * 7 + g) * SL_UB + u] = g < nblk ? slab_sum(pre + base + (long long)g * H + sk * stride, nsplit - sk, stride) : 0.f;
            }
            sums[(wave * 7 + 6) * SL_UB + u] = co;
        }
    }
    __syncthreads();
    if (wave == K || j >= H) return;
    const int pl = sel_s[wave], w = sel_s[KMAX + wave];
    const int row = b * K + wave;
    const float* xp = xproj + (long long)w * 6 * H + j;
    float q[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        float s = sums[(pl * 7 + g) * SL_UB + u];
        s += xp[(long long)g * H];
        q[g] = s + vp[g];
    }
    lstm1_point(q, sums[(pl * 7 + 6) * SL_UB + u], (long long)row * H + j, h1n, c1n, s_t, gpre, h1n16, s_t16, isc);
}

// final ordering by sequence log-prob + back-tracking through the parent pointers      (:182-194)
// One wave per image.  The image's whole history (T x beam parents / words / gates, strided by B * beam in global memory)
// is fetched with independent loads into LDS first: the walk itself is then T dependent LDS reads instead of T dependent
// global round trips per thread (a thread-per-image walk took 31 us for 100 images).
__global__ __launch_bounds__(64) void k_backtrack(int T, int B, int beam, int out_size, const float* __restrict__ seq,
                                                  const int* __restrict__ hist_parent, const int* __restrict__ hist_word,
                                                  const int* __restrict__ hist_gate, const float* __restrict__ hist_lpw,
                                                  const float* __restrict__ hist_lpg, int64_t* __restrict__ words, int64_t* __restrict__ gates,
                                                  float* __restrict__ lp_w, float* __restrict__ lp_g, float* __restrict__ scores) {
    extern __shared__ int bt[];                  // [3][T][beam]: parent, word, gate of this image
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = T * beam;
    int* par = bt;
    int* wrd = bt + n;
    int* gat = bt + 2 * n;
    for (int i = lane; i < n; i += 64) {
        const int t = i / beam, q = i - t * beam;
        const long long hrow = (long long)t * B * beam + b * beam + q;
        par[i] = hist_parent[hrow];
        wrd[i] = hist_word[hrow];
        gat[i] = hist_gate[hrow];
    }
    __shared__ int order[KMAX];
    if (lane == 0) {
        for (int q = 0; q < beam; ++q) order[q] = q;
        for (int a = 1; a < beam; ++a) {          // stable insertion sort, descending
            const int oa = order[a];
            const float va = seq[b * beam + oa];
            int p = a - 1;
            while (p >= 0 && seq[b * beam + order[p]] < va) { order[p + 1] = order[p]; --p; }
            order[p + 1] = oa;
        }
    }
    __syncthreads();
    for (int i = lane; i < out_size * T; i += 64) {   // per-slot log-probs follow the final SLOT, not the ancestry
        const int o = i / T, t = i - o * T;
        const long long hrow = (long long)t * B * beam + b * beam + order[o];
        const long long dst = ((long long)b * out_size + o) * T + t;
        if (lp_w) lp_w[dst] = hist_lpw[hrow];
        if (lp_g) lp_g[dst] = hist_lpg[hrow];
    }
    if (lane < out_size) {
        int q = order[lane];
        const long long dst = ((long long)b * out_size + lane) * T;
        if (scores) scores[b * out_size + lane] = seq[b * beam + q];
        for (int t = T - 1; t >= 0; --t) {
            words[dst + t] = wrd[t * beam + q];
            gates[dst + t] = gat[t * beam + q];
            q = par[t * beam + q];
        }
    }
}

}  // namespace vsr

// Kernels of the two ordering models that run before the decoder in the eval loop:
//   S-SSP  /root/reference/models/sort_model.py:105-183 (generate, mode 'not-normal'), sort_modules.py:25-135,
//          transformer_modules.py:18-147 (attention), :182-215 (embedding x sqrt(512)), :302-345 (feed-forward, encoder layer)
//   R-SSP  /root/reference/models/sinkhorn_network.py:30-51 (MLP, 20 Sinkhorn iterations) and the assignment of
//          coco_scripts/eval_coco.py:185-189 (munkres on max - value of the transposed matrix)
// All sequences / items of a loader batch are processed together: the matrix products run on the stream-K fp32-MFMA GEMM
// (rows = sequences x positions), everything here is the pointwise / tiny-attention / selection part.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace vsr {

constexpr int SSP_H = 512, SSP_HEADS = 8, SSP_HD = 64, SSP_FF = 2048, SSP_LEN = 10, SSP_ROLES = 26;

// x[s, j, :] = sqrt(512) * (table[tok[s * ld_tok + j]] (+ vtable[verb[s] % 10000]))     (transformer_modules.py:193-203,
// sort_modules.py:52: v_embed(verb) + sr_embed(roles); sort_modules.py:125: embed_layer(tokens))
__global__ __launch_bounds__(128) void k_ssp_embed(const int* __restrict__ tok, int ld_tok, int len, const float* __restrict__ table,
                                                   const int64_t* __restrict__ verbs, const float* __restrict__ vtable, int n_verbs,
                                                   int S, float* __restrict__ out, int* __restrict__ bad) {
    const int row = blockIdx.x;                       // s * len + j
    const int s = row / len, j = row - s * len;
    int t = tok[s * ld_tok + j];
    if (t < 0 || t >= SSP_ROLES) { if (threadIdx.x == 0) atomicAdd(bad, 1); t = 0; }
    const float sc = 22.627416997969522f;             // sqrt(512)
    const float4 a = *reinterpret_cast<const float4*>(table + (long long)t * SSP_H + threadIdx.x * 4);
    float4 o = make_float4(a.x * sc, a.y * sc, a.z * sc, a.w * sc);
    if (verbs) {
        long long v = verbs[s] % 10000;                // sort_model.py:108
        if (v < 0 || v >= n_verbs) { if (threadIdx.x == 0) atomicAdd(bad, 1); v = 0; }
        const float4 b = *reinterpret_cast<const float4*>(vtable + v * SSP_H + threadIdx.x * 4);
        o = make_float4(b.x * sc + a.x * sc, b.y * sc + a.y * sc, b.z * sc + a.z * sc, b.w * sc + a.w * sc);
    }
    *reinterpret_cast<float4*>(out + (long long)row * SSP_H + threadIdx.x * 4) = o;
}

// nn.LayerNorm(512), eps 1e-5, biased variance: one wave per row
__global__ __launch_bounds__(256) void k_layernorm512(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                      int rows, float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + (long long)row * SSP_H;
    float4 v[2];
    v[0] = *reinterpret_cast<const float4*>(xr + lane * 4);
    v[1] = *reinterpret_cast<const float4*>(xr + 256 + lane * 4);
    float s = (v[0].x + v[0].y) + (v[0].z + v[0].w) + (v[1].x + v[1].y) + (v[1].z + v[1].w);
    const float mean = wave_sum(s) * (1.0f / SSP_H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / SSP_H) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = i * 256 + lane * 4;
        const float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(b + c);
        float4 o;
        o.x = (v[i].x - mean) * rstd * ww.x + bb.x; o.y = (v[i].y - mean) * rstd * ww.y + bb.y;
        o.z = (v[i].z - mean) * rstd * ww.z + bb.z; o.w = (v[i].w - mean) * rstd * ww.w + bb.w;
        *reinterpret_cast<float4*>(out + (long long)row * SSP_H + c) = o;
    }
}

// out[m][n] = act(sum of slabs + bias[n]) (+ residual[m][n]);  act: 0 none, 1 relu, 2 tanh
__global__ void k_linear_finish(const float* __restrict__ slabs, int nslab, long long stride, int M, int N, const float* __restrict__ bias,
                                int act, const float* __restrict__ residual, long long ldr, float* __restrict__ out, long long ldo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float s = slab_sum(slabs + i, nslab, stride) + (bias ? bias[n] : 0.f);
    if (act == 1) s = fmaxf(s, 0.f);
    else if (act == 2) s = tanhf(s);
    if (residual) s += residual[(long long)m * ldr + n];
    out[(long long)m * ldo + n] = s;
}

// multi-head attention over short sequences: one wave per (sequence, head), lane = channel of the 64-wide head.
//   logits[i][j] = q_i . k_j / 8, masked entries -1e3 (transformer_modules.py:36-53), softmax over ALL Tk keys, ctx = weights . v
// mask_tok (optional): decoder self-attention, key j visible to query i iff j <= i and tok[s][j] != 0 (sort_modules.py:121-128);
// a query with no visible key gets the uniform softmax of Tk equal logits, as in the reference.
__global__ __launch_bounds__(64) void k_ssp_mha(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v, int Tq, int Tk,
                                                const int* __restrict__ mask_tok, int ld_tok, float* __restrict__ ctx) {
    const int s = blockIdx.x, hd = blockIdx.y, lane = threadIdx.x;
    const long long col = (long long)hd * SSP_HD + lane;
    float kk[SSP_LEN + 1], vv[SSP_LEN + 1];
#pragma unroll
    for (int j = 0; j < SSP_LEN + 1; ++j) {
        kk[j] = j < Tk ? k[((long long)s * Tk + j) * SSP_H + col] : 0.f;
        vv[j] = j < Tk ? v[((long long)s * Tk + j) * SSP_H + col] : 0.f;
    }
    for (int i = 0; i < Tq; ++i) {
        const float qi = q[((long long)s * Tq + i) * SSP_H + col];
        float lg[SSP_LEN + 1];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < SSP_LEN + 1; ++j) {
            if (j < Tk) {
                float d = wave_sum(qi * kk[j]) * 0.125f;
                if (mask_tok && !(j <= i && mask_tok[s * ld_tok + j] != 0)) d = -1e3f;
                lg[j] = d;
                mx = fmaxf(mx, d);
            }
        }
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < SSP_LEN + 1; ++j)
            if (j < Tk) { lg[j] = expf(lg[j] - mx); se += lg[j]; }
        float o = 0.f;
#pragma unroll
        for (int j = 0; j < SSP_LEN + 1; ++j)
            if (j < Tk) o += (lg[j] / se) * vv[j];
        ctx[((long long)s * Tq + i) * SSP_H + col] = o;
    }
}

// one step of the greedy "pick from the remaining roles" decode (sort_model.py:146-175): one wave per sequence.
//   logits (S, 26) = expander(state of the last position); among the roles still remaining (in their input order) the one
//   with the largest log-prob (first maximum) is emitted, removed, and becomes the next input token.
__global__ __launch_bounds__(64) void k_ssp_select(const float* __restrict__ logits, const int* __restrict__ roles, int* __restrict__ remain,
                                                   int t, int S, int* __restrict__ tokens /* (S, 11) */, int* __restrict__ pred, float* __restrict__ logp) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const float x = lane < SSP_ROLES ? logits[s * SSP_ROLES + lane] : -INFINITY;
    const float mx = wave_max(x);
    const float se = wave_sum(lane < SSP_ROLES ? expf(x - mx) : 0.f);
    const float lse = mx + logf(se);
    float best = -INFINITY;
    int at = -1;
    if (lane == 0) {
        for (int j = 0; j < SSP_LEN; ++j)
            if (remain[s * SSP_LEN + j]) {
                const float lp = logits[s * SSP_ROLES + roles[s * SSP_LEN + j]] - lse;
                if (lp > best) { best = lp; at = j; }              // strict: first maximum
            }
        int tok = 0;
        if (at >= 0) {
            tok = roles[s * SSP_LEN + at];
            remain[s * SSP_LEN + at] = 0;
            pred[s * SSP_LEN + t] = tok;
            logp[s * SSP_LEN + t] = best;
        }
        tokens[s * (SSP_LEN + 1) + t + 1] = tok;
    }
}

// A role id outside [0, SSP_ROLES) is not "remaining" (k_ssp_select indexes the 26 logits with it) and is counted in `bad`;
// the reference raises IndexError in its embedding for such an id (sort_model.py:108), the host wrapper does the same.
__global__ void k_ssp_init(const int* __restrict__ roles, int S, int* __restrict__ remain, int* __restrict__ tokens, int* __restrict__ pred,
                           float* __restrict__ logp, int* __restrict__ bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < S * SSP_LEN) {
        const int r = roles[i];
        const bool ok = r >= 0 && r < SSP_ROLES;
        if (!ok) atomicAdd(bad, 1);
        remain[i] = ok && r != 0;
        pred[i] = 0;
        logp[i] = 0.f;
    }
    if (i < S * (SSP_LEN + 1)) tokens[i] = 0;
}

// gather the last position's rows: out[s] = x[s * T + T - 1]
__global__ __launch_bounds__(128) void k_ssp_last(const float* __restrict__ x, int T, float* __restrict__ out) {
    const int s = blockIdx.x;
    *reinterpret_cast<float4*>(out + (long long)s * SSP_H + threadIdx.x * 4) =
        *reinterpret_cast<const float4*>(x + ((long long)s * T + T - 1) * SSP_H + threadIdx.x * 4);
}

// ---------------------------------------------------------------------------------------------- Sinkhorn + assignment
// cat[r] = [t1 (128) | v2 (128) | pos (4)] from the two MLP branches and the raw position columns of the input row
__global__ void k_sh_cat(const float* __restrict__ t1, const float* __restrict__ v2, const float* __restrict__ seq, int rows, float* __restrict__ cat) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)rows * 260) return;
    const int r = (int)(i / 260), c = (int)(i % 260);
    cat[i] = c < 128 ? t1[(long long)r * 128 + c] : c < 256 ? v2[(long long)r * 128 + c - 128] : seq[(long long)r * 2352 + 2348 + c - 256];
}

// One wave per item: x = exp(tanh(fc) / tau) (N x N, N <= 16), n_iters x (column-normalise, row-normalise) with the reference's
// eps 10e-8 (sinkhorn_network.py:30-37), then the assignment of eval_coco.py:185-189 on mx = x^T: columns chosen so that
// sum(max(mx) - mx[row][col]) is minimal (Kuhn-Munkres with potentials, O(N^3), fp64, lane 0).  assign[row] = column.
__global__ __launch_bounds__(64) void k_sinkhorn_assign(const float* __restrict__ fc /* (Q, N, N): tanh already applied */, int N, int n_iters,
                                                        float tau, float* __restrict__ tr, int* __restrict__ assign) {
    __shared__ float x[16][17];
    __shared__ double cost[16][16];
    const int qi = blockIdx.x, lane = threadIdx.x;
    const float* f = fc + (long long)qi * N * N;
    for (int i = lane; i < N * N; i += 64) x[i / N][i % N] = expf(f[i] / tau);
    __syncthreads();
    for (int it = 0; it < n_iters; ++it) {
        if (lane < N) {                                  // x / (eps + sum over rows): lane = column
            float s = 0.f;
            for (int r = 0; r < N; ++r) s += x[r][lane];
            s += 10e-8f;
            for (int r = 0; r < N; ++r) x[r][lane] = x[r][lane] / s;
        }
        __syncthreads();
        if (lane < N) {                                  // x / (eps + sum over columns): lane = row
            float s = 0.f;
            for (int c = 0; c < N; ++c) s += x[lane][c];
            s += 10e-8f;
            for (int c = 0; c < N; ++c) x[lane][c] = x[lane][c] / s;
        }
        __syncthreads();
    }
    if (tr)
        for (int i = lane; i < N * N; i += 64) tr[(long long)qi * N * N + i] = x[i / N][i % N];
    if (lane == 0) {
        double mxv = -1e300;
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < N; ++c) mxv = fmax(mxv, (double)x[r][c]);
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < N; ++c) cost[r][c] = mxv - (double)x[c][r];      // mx = x^T
        // Hungarian algorithm (potentials u, v; p[j] = row matched to column j), 1-based as in the classic formulation
        double u[17], v[17], minv[17];
        int p[17], way[17];
        bool used[17];
        for (int i = 0; i <= N; ++i) { u[i] = 0; v[i] = 0; p[i] = 0; way[i] = 0; }
        for (int i = 1; i <= N; ++i) {
            p[0] = i;
            int j0 = 0;
            for (int j = 0; j <= N; ++j) { minv[j] = 1e300; used[j] = false; }
            do {
                used[j0] = true;
                const int i0 = p[j0];
                double delta = 1e300;
                int j1 = 0;
                for (int j = 1; j <= N; ++j)
                    if (!used[j]) {
                        const double cur = cost[i0 - 1][j - 1] - u[i0] - v[j];
                        if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                        if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                    }
                for (int j = 0; j <= N; ++j)
                    if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                    else minv[j] -= delta;
                j0 = j1;
            } while (p[j0] != 0);
            do {
                const int j1 = way[j0];
                p[j0] = p[j1];
                j0 = j1;
            } while (j0);
        }
        for (int j = 1; j <= N; ++j) assign[(long long)qi * N + p[j] - 1] = j - 1;
    }
}

}  // namespace vsr

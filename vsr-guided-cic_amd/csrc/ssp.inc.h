// C ABI of the two ordering models that feed the decoder in the eval loop; included at the end of vsrcap.hip.
// Reference: coco_scripts/eval_coco.py:127-221 calls S_SSP.generate (batch size 1) once per (caption, verb) and
// SinkhornNet + munkres once per repeated role, each with host round trips; here ALL sequences / items of a loader batch go
// through one call each, and the results (role orders, assignments) stay on the device until the host glue
// (vsrcap/evalbatch.py: rank_captions) turns them into the (N, L) rank tensor vsr_reorder_slots consumes.
#include "ssp_kernels.h"

struct vsr_ssp {
    vsr_handle cfg;                  // GEMM launch configuration only (stream-K slots, tile choice); fp32
    vsr_ssp_weights w;
    vsr_sinkhorn_weights sw;
    bool has_ssp = false, has_sh = false;
};

extern "C" int vsr_ssp_create(vsr_ssp** out) {
    if (!out) return fail("vsr_ssp_create: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("vsr_ssp_create: no HIP device");
    vsr_ssp* e = new vsr_ssp();
    e->cfg.x3_on = false;            // the ordering models stay on the exact fp32 chain (their fixtures pin integer-truncated log-probs)
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
        e->cfg.gemm_slots = prop.multiProcessorCount * 4;
        e->cfg.gemm_slots_small = prop.multiProcessorCount * 3;
        e->cfg.gemm_slots_r16 = prop.multiProcessorCount;
    }
    *out = e;
    return 0;
}
extern "C" void vsr_ssp_destroy(vsr_ssp* e) { delete e; }

extern "C" int vsr_ssp_bind(vsr_ssp* e, const vsr_ssp_weights* w, const vsr_sinkhorn_weights* sw) {
    if (!e) return fail("vsr_ssp_bind: null handle");
    if (w) {
        if (!w->sr_embed || !w->v_embed || !w->fc_w || !w->exp_w || w->n_verbs <= 0) return fail("vsr_ssp_bind: incomplete S-SSP weights");
        e->w = *w;
        e->has_ssp = true;
    }
    if (sw) {
        if (!sw->W1_txt_w || !sw->W_fc_w || sw->N < 2 || sw->N > 16 || sw->n_iters < 0 || !(sw->tau > 0.f)) return fail("vsr_ssp_bind: bad Sinkhorn weights (2 <= N <= 16)");
        e->sw = *sw;
        e->has_sh = true;
    }
    return 0;
}

struct SspWs {
    float *x, *y, *q, *k, *v, *ctx, *x1, *ff, *prior, *pk[3], *pv[3], *last, *logits, *scratch;
    int *roles32, *remain, *tokens, *bad;
    size_t scratch_floats;
};
static size_t carve_ssp(int S, char* base, SspWs& w) {
    const size_t R = (size_t)S * (SSP_LEN + 1), H = SSP_H;
    Bump b{base};
    w.x = b.take<float>(R * H); w.y = b.take<float>(R * H); w.q = b.take<float>(R * H); w.k = b.take<float>(R * H);
    w.v = b.take<float>(R * H); w.ctx = b.take<float>(R * H); w.x1 = b.take<float>(R * H); w.ff = b.take<float>(R * SSP_FF);
    w.prior = b.take<float>((size_t)S * SSP_LEN * H);
    for (int l = 0; l < 3; ++l) { w.pk[l] = b.take<float>((size_t)S * SSP_LEN * H); w.pv[l] = b.take<float>((size_t)S * SSP_LEN * H); }
    w.last = b.take<float>((size_t)S * H); w.logits = b.take<float>((size_t)S * SSP_ROLES);
    w.roles32 = b.take<int>((size_t)S * SSP_LEN); w.remain = b.take<int>((size_t)S * SSP_LEN); w.tokens = b.take<int>(R); w.bad = b.take<int>(4);
    w.scratch_floats = R * SSP_FF * 8;
    w.scratch = b.take<float>(w.scratch_floats);
    return (b.off + 255) & ~size_t(255);
}
extern "C" size_t vsr_ssp_workspace_bytes(int32_t S) {
    if (S <= 0) return 0;
    SspWs w;
    return carve_ssp(S, nullptr, w);
}

// out (M, N) = act(A (M, K) . W (N, K)^T + bias) (+ residual); up to three products of the same A in one launch
struct LinSpec { const float* W; const float* bias; int N; float* out; int act; const float* residual; };
static int linear_n(vsr_ssp* e, SspWs& ws, hipStream_t s, int M, int K, const float* A, int lda, const LinSpec* sp, int n) {
    GemmBuilder g;
    size_t off = 0;
    int ns = 0;
    for (int i = 0; i < n; ++i) {
        GemmProb& p = g.prob(M, sp[i].N, nullptr, sp[i].N);
        GemmBuilder::seg(p, A, lda, nullptr, sp[i].W, K, K);
    }
    ns = g.finish(&e->cfg);
    for (int i = 0; i < n; ++i) {
        g.a.p[i].C = ws.scratch + off;
        g.a.p[i].slab_stride = (long long)M * sp[i].N;
        off += (size_t)M * sp[i].N * ns;
    }
    if (off > ws.scratch_floats) return fail("ssp: GEMM scratch too small");
    if (g.launch(s, &e->cfg)) return fail("ssp: gemm launch failed");
    for (int i = 0; i < n; ++i) {
        const long long tot = (long long)M * sp[i].N;
        hipLaunchKernelGGL(k_linear_finish, dim3(cdiv(tot, 256)), dim3(256), 0, s, g.a.p[i].C, ns, tot, M, sp[i].N, sp[i].bias, sp[i].act,
                           sp[i].residual, (long long)sp[i].N, sp[i].out, (long long)sp[i].N);
    }
    return 0;
}
static int linear1(vsr_ssp* e, SspWs& ws, hipStream_t s, int M, int N, int K, const float* A, int lda, const float* W, const float* bias, int act,
                   const float* residual, float* out) {
    LinSpec sp{W, bias, N, out, act, residual};
    return linear_n(e, ws, s, M, K, A, lda, &sp, 1);
}
static void layernorm(hipStream_t s, const float* x, const float* w, const float* b, int rows, float* out) {
    hipLaunchKernelGGL(k_layernorm512, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, w, b, rows, out);
}

// S_SSP.generate(mode='not-normal') (sort_model.py:105-183) for S sequences at once.
//   verbs (S) int64 (taken % 10000 as in :108), roles (S, 10) int32 role ids, 0 = padding (the reference's verb_det_seqs_sr)
//   pred (S, 10) int32: roles in generated order, 0 beyond; logp (S, 10) fp32: log-prob of each pick (the reference returns these
//   truncated to integers - its buffer inherits the integer dtype of the role tensor, :121 - and its callers ignore them)
extern "C" int vsr_ssp_generate(vsr_ssp* e, const int64_t* verbs, const int32_t* roles, int32_t S, int32_t* pred, float* logp, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (!e || !e->has_ssp) return fail("vsr_ssp_generate: S-SSP weights not bound");
    if (!verbs || !roles || !pred || !logp || !workspace || S <= 0) return fail("vsr_ssp_generate: bad arguments");
    SspWs ws;
    if (carve_ssp(S, reinterpret_cast<char*>(workspace), ws) > workspace_bytes) return fail("vsr_ssp_generate: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const vsr_ssp_weights& w = e->w;
    const int H = SSP_H, L = SSP_LEN;
    HIPCHK(hipMemsetAsync(ws.bad, 0, 4 * sizeof(int), s));
    hipLaunchKernelGGL(k_ssp_init, dim3(cdiv(S * (L + 1), 256)), dim3(256), 0, s, roles, S, ws.remain, ws.tokens, pred, logp, ws.bad);
    // ---- encoder (sort_modules.py:50-62): embeddings -> fc_feat -> 3 pre-LN layers -> LN
    int R = S * L;
    hipLaunchKernelGGL(k_ssp_embed, dim3(R), dim3(128), 0, s, roles, L, L, w.sr_embed, verbs, w.v_embed, w.n_verbs, S, ws.y, ws.bad);
    if (linear1(e, ws, s, R, H, H, ws.y, H, w.fc_w, w.fc_b, 0, nullptr, ws.x)) return 1;
    for (int l = 0; l < 3; ++l) {
        const vsr_ssp_layer& ly = w.enc[l];
        layernorm(s, ws.x, ly.ln1_w, ly.ln1_b, R, ws.y);
        LinSpec qkv[3] = {{ly.Wq, ly.bq, H, ws.q, 0, nullptr}, {ly.Wk, ly.bk, H, ws.k, 0, nullptr}, {ly.Wv, ly.bv, H, ws.v, 0, nullptr}};
        if (linear_n(e, ws, s, R, H, ws.y, H, qkv, 3)) return 1;
        hipLaunchKernelGGL(k_ssp_mha, dim3(S, SSP_HEADS), dim3(64), 0, s, ws.q, ws.k, ws.v, L, L, (const int*)nullptr, 0, ws.ctx);
        if (linear1(e, ws, s, R, H, H, ws.ctx, H, ly.Wo, ly.bo, 0, ws.x, ws.x1)) return 1;
        layernorm(s, ws.x1, ly.ln2_w, ly.ln2_b, R, ws.y);
        if (linear1(e, ws, s, R, SSP_FF, H, ws.y, H, ly.W1, ly.b1, 1, nullptr, ws.ff)) return 1;
        if (linear1(e, ws, s, R, H, SSP_FF, ws.ff, SSP_FF, ly.W2, ly.b2, 0, ws.x1, ws.x)) return 1;
    }
    layernorm(s, ws.x, w.enc_ln_w, w.enc_ln_b, R, ws.prior);
    // keys / values of the cross attention: prior states through each decoder layer's (self-)attention K / V projections,
    // constant over the decode steps (sort_modules.py:88 re-uses self.attention for the cross attention)
    for (int l = 0; l < 3; ++l) {
        LinSpec kv[2] = {{w.dec[l].Wk, w.dec[l].bk, H, ws.pk[l], 0, nullptr}, {w.dec[l].Wv, w.dec[l].bv, H, ws.pv[l], 0, nullptr}};
        if (linear_n(e, ws, s, R, H, ws.prior, H, kv, 2)) return 1;
    }
    LAUNCHCHK();
    // ---- decoder: step t re-runs the stack on [bos, picks 0..t-1] as the reference does (:154-160) and picks among the remaining roles
    for (int t = 0; t < L; ++t) {
        const int T = t + 1;
        R = S * T;
        hipLaunchKernelGGL(k_ssp_embed, dim3(R), dim3(128), 0, s, ws.tokens, L + 1, T, w.sr_embed, (const int64_t*)nullptr, (const float*)nullptr, 0, S,
                           ws.x, ws.bad);
        for (int l = 0; l < 3; ++l) {
            const vsr_ssp_layer& ly = w.dec[l];
            layernorm(s, ws.x, ly.ln1_w, ly.ln1_b, R, ws.y);
            LinSpec qkv[3] = {{ly.Wq, ly.bq, H, ws.q, 0, nullptr}, {ly.Wk, ly.bk, H, ws.k, 0, nullptr}, {ly.Wv, ly.bv, H, ws.v, 0, nullptr}};
            if (linear_n(e, ws, s, R, H, ws.y, H, qkv, 3)) return 1;
            hipLaunchKernelGGL(k_ssp_mha, dim3(S, SSP_HEADS), dim3(64), 0, s, ws.q, ws.k, ws.v, T, T, ws.tokens, L + 1, ws.ctx);
            if (linear1(e, ws, s, R, H, H, ws.ctx, H, ly.Wo, ly.bo, 0, ws.x, ws.x1)) return 1;            // h1 = attn + x
            layernorm(s, ws.x1, ly.ln2_w, ly.ln2_b, R, ws.y);
            if (linear1(e, ws, s, R, H, H, ws.y, H, ly.Wq, ly.bq, 0, nullptr, ws.q)) return 1;
            hipLaunchKernelGGL(k_ssp_mha, dim3(S, SSP_HEADS), dim3(64), 0, s, ws.q, ws.pk[l], ws.pv[l], T, L, (const int*)nullptr, 0, ws.ctx);
            if (linear1(e, ws, s, R, H, H, ws.ctx, H, ly.Wo, ly.bo, 0, ws.x1, ws.x)) return 1;            // h2 = cross + h1   (in ws.x)
            layernorm(s, ws.x, ly.ln3_w, ly.ln3_b, R, ws.y);
            if (linear1(e, ws, s, R, SSP_FF, H, ws.y, H, ly.W1, ly.b1, 1, nullptr, ws.ff)) return 1;
            if (linear1(e, ws, s, R, H, SSP_FF, ws.ff, SSP_FF, ly.W2, ly.b2, 0, ws.x, ws.x1)) return 1;    // h3 = ff + h2      (in ws.x1)
            std::swap(ws.x, ws.x1);
        }
        hipLaunchKernelGGL(k_ssp_last, dim3(S), dim3(128), 0, s, ws.x, T, ws.last);
        layernorm(s, ws.last, w.dec_ln_w, w.dec_ln_b, S, ws.y);
        if (linear1(e, ws, s, S, SSP_ROLES, H, ws.y, H, w.exp_w, w.exp_b, 0, nullptr, ws.logits)) return 1;
        hipLaunchKernelGGL(k_ssp_select, dim3(S), dim3(64), 0, s, ws.logits, roles, ws.remain, t, S, ws.tokens, pred, logp);
        LAUNCHCHK();
    }
    return 0;
}

// SinkhornNet.forward (sinkhorn_network.py:39-51) + the assignment of eval_coco.py:185-189 for Q items at once.
//   seq (Q, N, 2352) fp32 rows [300 | 2048 | 4]; tr (Q, N, N) fp32 or NULL: the doubly-normalised matrix; assign (Q, N) int32:
//   assign[q][i] = column chosen for row i of tr[q]^T (the munkres result "(i, assign)" of :187).
extern "C" size_t vsr_sinkhorn_workspace_bytes(int32_t Q, int32_t N) {
    if (Q <= 0 || N <= 0) return 0;
    const size_t R = (size_t)Q * N;
    return (R * (128 + 512 + 128 + 260 + 256 + 16) + R * 512 * 8 + 1024) * sizeof(float);
}
extern "C" int vsr_sinkhorn_assign(vsr_ssp* e, const float* seq, int32_t Q, float* tr, int32_t* assign, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!e || !e->has_sh) return fail("vsr_sinkhorn_assign: Sinkhorn weights not bound");
    if (!seq || !assign || !workspace || Q <= 0) return fail("vsr_sinkhorn_assign: bad arguments");
    const vsr_sinkhorn_weights& w = e->sw;
    const int N = w.N, R = Q * N;
    if (workspace_bytes < vsr_sinkhorn_workspace_bytes(Q, N)) return fail("vsr_sinkhorn_assign: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Bump b{reinterpret_cast<char*>(workspace)};
    float* t1 = b.take<float>((size_t)R * 128);
    float* v1 = b.take<float>((size_t)R * 512);
    float* v2 = b.take<float>((size_t)R * 128);
    float* cat = b.take<float>((size_t)R * 260);
    float* f1 = b.take<float>((size_t)R * 256);
    float* fc = b.take<float>((size_t)R * 16);
    SspWs ws{};
    ws.scratch_floats = (size_t)R * 512 * 8;
    ws.scratch = b.take<float>(ws.scratch_floats);
    if (linear1(e, ws, s, R, 128, 300, seq, 2352, w.W1_txt_w, w.W1_txt_b, 1, nullptr, t1)) return 1;
    if (linear1(e, ws, s, R, 512, 2048, seq + 300, 2352, w.W1_vis_w, w.W1_vis_b, 1, nullptr, v1)) return 1;
    if (linear1(e, ws, s, R, 128, 512, v1, 512, w.W2_vis_w, w.W2_vis_b, 1, nullptr, v2)) return 1;
    hipLaunchKernelGGL(k_sh_cat, dim3(cdiv((long long)R * 260, 256)), dim3(256), 0, s, t1, v2, seq, R, cat);
    if (linear1(e, ws, s, R, 256, 260, cat, 260, w.W_fc_pos_w, w.W_fc_pos_b, 1, nullptr, f1)) return 1;
    if (linear1(e, ws, s, R, N, 256, f1, 256, w.W_fc_w, w.W_fc_b, 2, nullptr, fc)) return 1;
    hipLaunchKernelGGL(k_sinkhorn_assign, dim3(Q), dim3(64), 0, s, fc, N, w.n_iters, w.tau, tr, assign);
    LAUNCHCHK();
    return 0;
}

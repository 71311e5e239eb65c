// Training path of libvsrcap.so: teacher-forced (or sample-replayed) forward that saves activations, and the
// hand-written BPTT backward.  Included at the end of vsrcap.hip.
//
// Reference: coco_scripts/train.py:103-113 (XE: model(...) -> NLL losses -> loss.backward()) and :151-178 (SCST:
// sample_rl log-probs -> loss.backward()); the graph autograd differentiates there is step :117-190 unrolled T times.
//
// Backward structure (B rows, T steps, rows of the "all-steps" matrices ordered (t, b)):
//   phase 0  dlogits for all steps (log_softmax backward) and dh2_vocab = dlogits . W_out        one big GEMM
//   phase A  t = T-1 .. 0: pointwise backward kernels + 3 grouped GEMMs per step against TRANSPOSED weights
//            (data gradients only): dpre2_t, dpre1_t, d[hg|hA]_t, d[sent|sa]_t, dga_t, dP rows, state carries
//   phase B  all weight gradients as big GEMMs with the reduction over the T*B rows:
//            dW[n][k] = sum_rows dY^T[n][row] * X^T[k][row]   (both operands transposed once per call)
// so every matrix product of the backward pass runs on the same stream-K fp32-MFMA kernel as the forward pass.

constexpr int VSR_GRAD_BUCKETS = 5;
struct TrainCtx {                      // (plain data + two vectors: copied into SavedForwards)
    bool valid = false;
    const char* tws_lo = nullptr;      // the training workspace this forward was carved into
    size_t tws_bytes = 0;
    int* h2_stash = nullptr;           // f16x2: the per-batch exponents / bounds of the handle's table as they were for THIS forward (8 ints)
    long long generation = 0;          // bumped by every vsr_train_forward: identifies the saved forward a backward belongs to
    int B = 0, T = 0, TB = 0, TBp = 0, Bp = 0, RLp = 0;
    const float* logp_w = nullptr;     // caller's (B,T,V) output of the forward, needed by the backward
    const float* logp_g = nullptr;     // (B,T,2)
    float *h1s, *c1s, *h2s, *c2s;      // (T+1, B, H): slot 0 = zeros
    float *gates1, *gates2, *s_ts, *g_ts, *hAs, *sas, *gas, *sents, *atts, *alphas, *x_all;
    int *word32, *slot32, *rows_bt;
    float *dlogits, *dh2_voc, *dpre1, *dpre2, *dhA_all, *dsent_all, *dsa_all, *dga_all, *dwa_rows, *dws_rows, *dwg_rows, *dP;
    float* dP_bank = nullptr;        // index-list regions: dP summed over the slot entries that name each bank row
    float *dalpha;
    float *datt, *dtc, *dh_tot, *dzsum, *dh1_c, *dh2_c, *dc1_c[2], *dc2_c[2], *dpre1sum, *dpre2sum, *dx_all;
    float *wT_ih1, *wT_is, *wT_ig, *wT_hh1, *wT_hs, *wT_ih2, *wT_hh2, *wT_hg, *wT_ha, *wT_sfc, *wT_sa, *wT_ga, *wT_out;
    float *tX_h2prev, *tX_x, *tX_h1prev, *tX_h1, *tX_att, *tX_st, *tX_gt, *tX_h2, *tX_vbar, *tX_reg;
    float *tY_dpre1, *tY_dpre2, *tY_dlogits, *tY_dhA, *tY_dsent, *tY_dsa, *tY_dga, *tY_dpre1sum, *tY_dpre2sum, *tY_dP;
    float* xproj_all;            // (T B, 6H): embedding rows through the x columns of the LSTM1 / gate input weights
    float* scratch;
    size_t scratch_floats = 0;
    // bf16 mode: every transposed operand that a GEMM takes as W (transposed weights, transposed activations of the
    // weight-gradient products) has a bf16 twin the transposing kernels write instead of the fp32 buffer
    struct Twin { const float* f; size_t n; uint16_t* b; };
    std::vector<Twin> twins;
    // f16x2 flavour: the same operands have fp16-pair images (gemm_h2.h) in h2img, written by the transposing kernel INSTEAD of the
    // fp32 buffer (transpose()); slot = exponent slot of the values (the weight's own, or the class of the activation)
    struct Img { const float* f; size_t n; size_t off; int slot; };
    std::vector<Img> imgs;
    float* h2img = nullptr;
    // ... and the forward pass's A operands are written as fp16-pair images by their producers (kernels.h img_store; gemm_h2a.h takes them):
    // h1 / h2 of every step (T + 1 slots like the fp32 saves), s_t / g_t / the attended vector of the current step
    uint16_t *h1s16 = nullptr, *h2s16 = nullptr, *s_t16 = nullptr, *g_t16 = nullptr, *att16 = nullptr;
};

static inline size_t up4(size_t x) { return (x + 7) & ~size_t(7); }   // K paddings: multiples of 8 (16-byte bf16 chunks; fp32 needs 4)

static size_t carve_train(const vsr_handle* h, TrainCtx& t, char* base) {
    const vsr_dims& d = h->d;
    const Ctx& c = h->c;
    const size_t B = t.B, T = t.T, H = d.rnn_size, A = d.att_size, D = d.det_feat_size, E = d.input_encoding_size, V = d.vocab_size;
    const size_t in1 = (d.h2_first_lstm ? H : 0) + D + E, in2 = H + D + (d.img_second_lstm ? D : 0);
    const size_t TB = T * B, TBp = up4(TB), Bp = up4(B), RL = (size_t)c.B * c.L * c.R, R1 = c.R + 1;
    // rows att_va's weight gradient runs over: the slot entries (dense regions) or the feature-bank rows (index lists)
    const size_t PR = c.Rb > 0 ? (size_t)c.n_img * c.Rb : RL, RLp = up4(std::max(RL, PR));
    t.TB = (int)TB; t.TBp = (int)TBp; t.Bp = (int)Bp; t.RLp = (int)RLp;
    Bump b{base};
    t.h1s = b.take<float>((T + 1) * B * H); t.c1s = b.take<float>((T + 1) * B * H);
    t.h2s = b.take<float>((T + 1) * B * H); t.c2s = b.take<float>((T + 1) * B * H);
    t.gates1 = b.take<float>(TB * 6 * H); t.gates2 = b.take<float>(TB * 4 * H);
    t.s_ts = b.take<float>(TB * H); t.g_ts = b.take<float>(TB * H);
    t.hAs = b.take<float>(TB * A); t.sas = b.take<float>(TB * A); t.gas = b.take<float>(TB * A);
    t.sents = b.take<float>(TB * D); t.atts = b.take<float>(TB * D); t.alphas = b.take<float>(TB * R1);
    t.x_all = b.take<float>(TB * E);
    t.word32 = b.take<int>(TB); t.slot32 = b.take<int>(TB); t.rows_bt = b.take<int>(TB);
    t.h2_stash = b.take<int>(8);
    t.dlogits = b.take<float>(TB * up4(V)); t.dh2_voc = b.take<float>(TB * H);
    t.dpre1 = b.take<float>(TB * 6 * H); t.dpre2 = b.take<float>(TB * 4 * H);
    t.dhA_all = b.take<float>(TB * A); t.dsent_all = b.take<float>(TB * D); t.dsa_all = b.take<float>(TB * A); t.dga_all = b.take<float>(TB * A);
    t.dwa_rows = b.take<float>(TB * A); t.dws_rows = b.take<float>(TB * A); t.dwg_rows = b.take<float>(TB * A);
    t.dP = b.take<float>(RL * A);
    t.dP_bank = c.Rb > 0 ? b.take<float>(PR * A) : nullptr;
    t.datt = b.take<float>(B * D); t.dtc = b.take<float>(B * H);
    t.dh_tot = b.take<float>(B * H); t.dzsum = b.take<float>(B); t.dalpha = b.take<float>(B * R1);
    t.dh1_c = b.take<float>(B * H); t.dh2_c = b.take<float>(B * H);
    for (int i = 0; i < 2; ++i) { t.dc1_c[i] = b.take<float>(B * H); t.dc2_c[i] = b.take<float>(B * H); }
    t.dpre1sum = b.take<float>(B * 6 * H); t.dpre2sum = b.take<float>(B * 4 * H); t.dx_all = b.take<float>(TB * E); t.xproj_all = b.take<float>(TB * 6 * H);
    t.wT_ih1 = b.take<float>(in1 * 4 * H); t.wT_is = b.take<float>(in1 * H); t.wT_ig = b.take<float>(in1 * H);
    t.wT_hh1 = b.take<float>(H * 4 * H); t.wT_hs = b.take<float>(H * H);
    t.wT_ih2 = b.take<float>(in2 * 4 * H); t.wT_hh2 = b.take<float>(H * 4 * H);
    t.wT_hg = b.take<float>(H * H); t.wT_ha = b.take<float>(H * A); t.wT_sfc = b.take<float>(H * D);
    t.wT_sa = b.take<float>(H * A); t.wT_ga = b.take<float>(H * A); t.wT_out = b.take<float>(H * up4(V));
    t.twins.clear();
    auto twin = [&](const float* f, size_t n) { t.twins.push_back(TrainCtx::Twin{f, n, b.take<uint16_t>(n)}); };
    twin(t.wT_ih1, in1 * 4 * H); twin(t.wT_is, in1 * H); twin(t.wT_ig, in1 * H); twin(t.wT_hh1, H * 4 * H); twin(t.wT_hs, H * H);
    twin(t.wT_ih2, in2 * 4 * H); twin(t.wT_hh2, H * 4 * H); twin(t.wT_hg, H * H); twin(t.wT_ha, H * A); twin(t.wT_sfc, H * D);
    twin(t.wT_sa, H * A); twin(t.wT_ga, H * A); twin(t.wT_out, H * up4(V));
    b.off = (b.off + 255) & ~size_t(255);
    t.tX_h2prev = b.take<float>(H * TBp); t.tX_x = b.take<float>(E * TBp); t.tX_h1prev = b.take<float>(H * TBp);
    t.tX_h1 = b.take<float>(H * TBp); t.tX_att = b.take<float>(D * TBp); t.tX_st = b.take<float>(H * TBp);
    t.tX_gt = b.take<float>(H * TBp); t.tX_h2 = b.take<float>(H * TBp); t.tX_vbar = b.take<float>(D * Bp);
    t.tX_reg = b.take<float>(D * RLp);
    t.tY_dpre1 = b.take<float>(6 * H * TBp); t.tY_dpre2 = b.take<float>(4 * H * TBp); t.tY_dlogits = b.take<float>(V * TBp);
    t.tY_dhA = b.take<float>(A * TBp); t.tY_dsent = b.take<float>(D * TBp); t.tY_dsa = b.take<float>(A * TBp); t.tY_dga = b.take<float>(A * TBp);
    t.tY_dpre1sum = b.take<float>(6 * H * Bp); t.tY_dpre2sum = b.take<float>(4 * H * Bp); t.tY_dP = b.take<float>(A * RLp);
    twin(t.tX_h2prev, H * TBp); twin(t.tX_x, E * TBp); twin(t.tX_h1prev, H * TBp); twin(t.tX_h1, H * TBp); twin(t.tX_att, D * TBp);
    twin(t.tX_st, H * TBp); twin(t.tX_gt, H * TBp); twin(t.tX_h2, H * TBp); twin(t.tX_vbar, D * Bp); twin(t.tX_reg, D * RLp);
    t.imgs.clear(); t.h2img = nullptr;
    if (h->h2_on && !h->bf16_on) {
        size_t off = 0;
        auto img = [&](const float* f, size_t n, int slot) { t.imgs.push_back(TrainCtx::Img{f, n, off, slot}); off += up4(n); };
        // (slots 0..13: field order of b16_weight_list - the transposed matrix has the bound of the matrix)
        const vsr_weights& w = h->w;
        img(t.wT_ih1, in1 * 4 * H, h->h2_slot_of(w.lstm1_weight_ih)); img(t.wT_is, in1 * H, h->h2_slot_of(w.W1_is_weight));
        img(t.wT_ig, in1 * H, h->h2_slot_of(w.W1_ig_weight)); img(t.wT_hh1, H * 4 * H, h->h2_slot_of(w.lstm1_weight_hh));
        img(t.wT_hs, H * H, h->h2_slot_of(w.W1_hs_weight)); img(t.wT_ih2, in2 * 4 * H, h->h2_slot_of(w.lstm2_weight_ih));
        img(t.wT_hh2, H * 4 * H, h->h2_slot_of(w.lstm2_weight_hh)); img(t.wT_hg, H * H, h->h2_slot_of(w.W1_hg_weight));
        img(t.wT_ha, H * A, h->h2_slot_of(w.att_ha_weight)); img(t.wT_sfc, H * D, h->h2_slot_of(w.s_fc_weight));
        img(t.wT_sa, H * A, h->h2_slot_of(w.att_sa_weight)); img(t.wT_ga, H * A, h->h2_slot_of(w.att_ga_weight));
        img(t.wT_out, H * up4(V), h->h2_slot_of(w.out_fc_weight));
        img(t.tX_h2prev, H * TBp, H2A_UNIT); img(t.tX_x, E * TBp, H2A_EMBED); img(t.tX_h1prev, H * TBp, H2A_UNIT); img(t.tX_h1, H * TBp, H2A_UNIT);
        img(t.tX_att, D * TBp, H2A_ATT); img(t.tX_st, H * TBp, H2A_UNIT); img(t.tX_gt, H * TBp, H2A_UNIT); img(t.tX_h2, H * TBp, H2A_UNIT);
        img(t.tX_vbar, D * Bp, H2A_DET); img(t.tX_reg, D * RLp, H2A_REGION);
        b.off = (b.off + 255) & ~size_t(255);
        t.h2img = b.take<float>(off);
        b.off = (b.off + 255) & ~size_t(255);
        t.h1s16 = b.take<uint16_t>(2 * (T + 1) * B * H); t.h2s16 = b.take<uint16_t>(2 * (T + 1) * B * H);
        t.s_t16 = b.take<uint16_t>(2 * B * H); t.g_t16 = b.take<uint16_t>(2 * B * H); t.att16 = b.take<uint16_t>(2 * B * D);
    } else {
        t.h1s16 = t.h2s16 = t.s_t16 = t.g_t16 = t.att16 = nullptr;
    }
    // GEMM slab scratch: 8 slabs of the largest product of the training path
    size_t big = std::max({4 * H * in1, V * H, 4 * H * in2, A * D, TB * H, TB * E, D * H, B * (in2 + 2 * H), H * in1, TB * V, TB * 6 * H});
    t.scratch_floats = big * 8;
    t.scratch = b.take<float>(t.scratch_floats);
    return (b.off + 255) & ~size_t(255);
}

struct SegSpec { const float* A; int lda; const float* W; int ldw; int K; int a_cls = H2A_NONE; const uint16_t* A16 = nullptr; };

// whole-pass bounds from the per-step ones (block 63: max over the steps; block 62 slots 2, 3: the sums over t of dpre1 / dpre2 rows)
__global__ void k_h2_dyn_fold(int* __restrict__ dyn, int T) {
    const int j = threadIdx.x;
    if (j >= 8) return;
    int m = 0;
    for (int tt = 0; tt < T; ++tt) m = max(m, dyn[tt * 8 + j]);
    dyn[63 * 8 + j] = m;
    if (j == 0) dyn[62 * 8 + 3] = __float_as_int(__int_as_float(m) * (float)T);
    if (j == 6) {
        int m2 = 0;
        for (int tt = 0; tt < T; ++tt) m2 = max(m2, dyn[tt * 8 + 2]);
        dyn[62 * 8 + 2] = __float_as_int(fmaxf(__int_as_float(m), __int_as_float(m2)) * (float)T);
        dyn[62 * 8 + 4] = max(m, m2);                                       // all six column blocks of dpre1 (its transpose is one image)
    }
}

// one problem, several K segments -> dst window (ldd), through the slab scratch
static int gemm_to(vsr_handle* h, TrainCtx& t, hipStream_t s, int M, int N, const SegSpec* segs, int nseg, float* dst, long long ldd) {
    GemmBuilder g;
    GemmProb& p = g.prob(M, N, t.scratch, N);
    for (int i = 0; i < nseg; ++i) GemmBuilder::seg(p, segs[i].A, segs[i].lda, nullptr, segs[i].W, segs[i].ldw, segs[i].K, segs[i].A16, segs[i].a_cls);
    for (int i = 0; i < nseg; ++i) g.a_image_only = g.a_image_only || (segs[i].A16 && !h->bf16_on);
    const int ns = g.finish(h);
    for (int i = 0; i < nseg; ++i)
        if (segs[i].A16 && !h->bf16_on && g.big != 37) return fail("training gemm: an A operand exists only as an fp16-pair image but the launch did not take the all-DMA kernel");
    const long long stride = (long long)M * N;
    if (ns == 1) {                          // every tile is produced by one workgroup: it writes the destination window itself
        g.a.p[0].C = dst; g.a.p[0].ldc = (int)ldd; g.a.p[0].slab_stride = 0;
        if (g.launch(s, h)) return fail("training gemm launch failed");
        return 0;
    }
    if ((size_t)stride * ns > t.scratch_floats) return fail("training scratch too small (%lld x %d)", stride, ns);
    g.a.p[0].slab_stride = stride;
    if (g.launch(s, h)) return fail("training gemm launch failed");
    hipLaunchKernelGGL(k_slab_reduce_2d, dim3(cdiv(stride, 256)), dim3(256), 0, s, t.scratch, ns, stride, M, N, dst, ldd);
    return 0;
}
static int gemm_to1(vsr_handle* h, TrainCtx& t, hipStream_t s, int M, int N, int K, const float* A, int lda, const float* W, int ldw,
                    float* dst, long long ldd, int a_cls = H2A_NONE, bool a_is_image = false) {
    // a_is_image: A is a transposed gradient that transpose() wrote as an fp16-pair image IN PLACE of the fp32 values (f16x2 flavour)
    SegSpec sg{A, lda, W, ldw, K, a_cls, a_is_image ? reinterpret_cast<const uint16_t*>(A) : nullptr};
    return gemm_to(h, t, s, M, N, &sg, 1, dst, ldd);
}
// Up to four INDEPENDENT products (one K segment each) in ONE launch (round 6): the weight-gradient GEMMs of a bucket.  Alone, a
// 4000 x 1000 gradient is 128 tiles of 63 k-tiles - cut in halves to fill the chip, i.e. two slabs and a reducing launch - and a
// 1000 x 1000 one 32 tiles cut in eight; two or three of them together are ~256 whole tiles: no slabs, no reduce, one launch.
struct GProb { int M, N, K; const float* A; int lda; const float* W; int ldw; float* dst; long long ldd; int a_cls; bool a_img; };
static int gemm_group(vsr_handle* h, TrainCtx& t, hipStream_t s, const GProb* P, int n) {
    if (n <= 0) return 0;
    GemmBuilder g;
    for (int i = 0; i < n; ++i) {
        GemmProb& p = g.prob(P[i].M, P[i].N, t.scratch, P[i].N);
        GemmBuilder::seg(p, P[i].A, P[i].lda, nullptr, P[i].W, P[i].ldw, P[i].K, P[i].a_img ? reinterpret_cast<const uint16_t*>(P[i].A) : nullptr, P[i].a_cls);
        g.a_image_only = g.a_image_only || (P[i].a_img && !h->bf16_on);
    }
    const int ns = g.finish(h);
    for (int i = 0; i < n; ++i)
        if (P[i].a_img && !h->bf16_on && g.big != 37) return fail("training gemm group: an A operand exists only as an fp16-pair image but the launch did not take the all-DMA kernel");
    if (ns == 1) {                          // every tile is produced by one workgroup: written in place
        for (int i = 0; i < n; ++i) { g.a.p[i].C = P[i].dst; g.a.p[i].ldc = (int)P[i].ldd; g.a.p[i].slab_stride = 0; }
        if (g.launch(s, h)) return fail("training gemm group launch failed");
        return 0;
    }
    size_t off[4], tot = 0;
    for (int i = 0; i < n; ++i) {
        g.a.p[i].nslab = gemm_tight_slabs(g.a, i);
        off[i] = tot;
        tot += (size_t)P[i].M * P[i].N * g.a.p[i].nslab;
    }
    if (tot > t.scratch_floats) return fail("training scratch too small for a gemm group (%zu floats)", tot);
    for (int i = 0; i < n; ++i) { g.a.p[i].C = t.scratch + off[i]; g.a.p[i].slab_stride = (long long)P[i].M * P[i].N; }
    if (g.launch(s, h)) return fail("training gemm group launch failed");
    for (int i = 0; i < n; ++i) {
        const long long stride = (long long)P[i].M * P[i].N;
        hipLaunchKernelGGL(k_slab_reduce_2d, dim3(cdiv(stride, 256)), dim3(256), 0, s, t.scratch + off[i], g.a.p[i].nslab, stride, P[i].M, P[i].N, P[i].dst, P[i].ldd);
    }
    return 0;
}
// deterministic two-stage column sum through the (idle) slab scratch
static void colsum(TrainCtx& t, hipStream_t s, const float* X, long long ld, int R, int C, float* out, float* out2 = nullptr) {
    hipLaunchKernelGGL(k_colsum, dim3(cdiv(C, 64), COLSUM_CHUNKS), dim3(256), 0, s, X, ld, R, C, t.scratch);
    hipLaunchKernelGGL(k_colsum_finish, dim3(cdiv(C, 256)), dim3(256), 0, s, t.scratch, C, 0, C, out, out2);
}
// up to ZERO_MT buffers zeroed by one launch (sizes in bytes, multiples of 16; pointers 16-byte aligned)
struct ZeroList {
    ZeroMulti z; int blocks = 0;
    ZeroList() { memset(&z, 0, sizeof(z)); }
    void add(void* p, size_t bytes) {
        if (!p || bytes == 0) return;
        z.p[z.nt] = p; z.n16[z.nt] = (long long)(bytes / 16); z.blk[z.nt] = blocks;
        blocks += (int)std::max<long long>(1, std::min<long long>(2048, (long long)(bytes / 16 + 1023) / 1024));
        z.blk[++z.nt] = blocks;
    }
    int launch(hipStream_t s) {
        if (z.nt == 0) return 0;
        hipLaunchKernelGGL(k_zero_multi, dim3(blocks), dim3(256), 0, s, z);
        return hipGetLastError() == hipSuccess ? 0 : 1;
    }
};
// out = in^T (rows optionally gathered through `list`).  A buffer that a GEMM takes as W receives ONLY its image - the registered bf16
// twin in the bf16 mode, the fp16-pair image of the f16x2 flavour when the backward pass runs on those kernels (h2img) - and the fp32
// buffer `out` then only lends its address
// self_slot >= 0: `out` itself receives the image (an A operand: a transposed gradient, scaled by the bound in that slot of the table)
// (batch: transposes without a row list may be COLLECTED and launched together - TransBatch::flush - in the order they were added)
struct TransBatch {
    TransMulti m; int kind = -1, blocks = 0; const int* exps = nullptr;
    TransBatch() { memset(&m, 0, sizeof(m)); }
    void flush(hipStream_t s) {
        if (m.nt == 0) return;
        if (kind == 2) hipLaunchKernelGGL((k_transpose_multi<2>), dim3(blocks), dim3(256), 0, s, m, exps);
        else if (kind == 1) hipLaunchKernelGGL((k_transpose_multi<1>), dim3(blocks), dim3(256), 0, s, m, exps);
        else hipLaunchKernelGGL((k_transpose_multi<0>), dim3(blocks), dim3(256), 0, s, m, exps);
        memset(&m, 0, sizeof(m)); kind = -1; blocks = 0;
    }
    void add(hipStream_t s, int k, const int* ex, const float* in, long long ld_in, int R, int C, float* out, long long ld_out, uint16_t* img, int slot) {
        if (m.nt == TR_MT || (m.nt > 0 && k != kind)) flush(s);
        kind = k; exps = ex;
        const int i = m.nt++;
        m.in[i] = in; m.ld_in[i] = ld_in; m.R[i] = R; m.C[i] = C; m.out[i] = out; m.ld_out[i] = ld_out; m.out16[i] = img; m.slot[i] = slot;
        m.blk[i] = blocks;
        blocks += cdiv(C, 64) * cdiv(R, 64);
        m.blk[i + 1] = blocks;
    }
};
static void transpose(vsr_handle* h, hipStream_t s, const float* in, long long ld_in, int R, int C, float* out, long long ld_out,
                      const int* list = nullptr, bool h2img = false, int self_slot = -1, const int* rlimit = nullptr, TransBatch* tb = nullptr) {
    uint16_t* tw = h->bf16_on ? const_cast<uint16_t*>(h->map16(out)) : nullptr;
    const dim3 grid(cdiv(C, 64), cdiv(R, 64)), block(256);
    const H2Range* r2 = (h2img && !tw) ? h->map_h2(out) : nullptr;
    if (tb && !list) {
        if (h2img && !tw && !r2 && self_slot >= 0 && ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 31) == 0)
            tb->add(s, 2, h->h2_exps, in, ld_in, R, C, out, ld_out, reinterpret_cast<uint16_t*>(out), self_slot);
        else if (r2) tb->add(s, 2, h->h2_exps, in, ld_in, R, C, out, ld_out, reinterpret_cast<uint16_t*>(const_cast<float*>(r2->img + (out - r2->lo))), r2->slot);
        else if (tw) tb->add(s, 1, nullptr, in, ld_in, R, C, out, ld_out, tw, 0);
        else tb->add(s, 0, nullptr, in, ld_in, R, C, out, ld_out, nullptr, 0);
        return;
    }
    if (h2img && !tw && !r2 && self_slot >= 0 && ld_out % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 31) == 0) {
        uint16_t* img = reinterpret_cast<uint16_t*>(out);
        if (list) hipLaunchKernelGGL((k_transpose_t<true, 2>), grid, block, 0, s, in, ld_in, list, R, C, out, ld_out, img, h->h2_exps, self_slot, rlimit);
        else hipLaunchKernelGGL((k_transpose_t<false, 2>), grid, block, 0, s, in, ld_in, (const int*)nullptr, R, C, out, ld_out, img, h->h2_exps, self_slot);
    } else if (r2) {
        uint16_t* img = reinterpret_cast<uint16_t*>(const_cast<float*>(r2->img + (out - r2->lo)));
        if (list) hipLaunchKernelGGL((k_transpose_t<true, 2>), grid, block, 0, s, in, ld_in, list, R, C, out, ld_out, img, h->h2_exps, r2->slot, rlimit);
        else hipLaunchKernelGGL((k_transpose_t<false, 2>), grid, block, 0, s, in, ld_in, (const int*)nullptr, R, C, out, ld_out, img, h->h2_exps, r2->slot);
    } else if (list && tw) hipLaunchKernelGGL((k_transpose_t<true, 1>), grid, block, 0, s, in, ld_in, list, R, C, out, ld_out, tw, (const int*)nullptr, 0, rlimit);
    else if (list) hipLaunchKernelGGL((k_transpose_t<true, 0>), grid, block, 0, s, in, ld_in, list, R, C, out, ld_out, (uint16_t*)nullptr, (const int*)nullptr, 0, rlimit);
    else if (tw) hipLaunchKernelGGL((k_transpose_t<false, 1>), grid, block, 0, s, in, ld_in, (const int*)nullptr, R, C, out, ld_out, tw);
    else hipLaunchKernelGGL((k_transpose_t<false, 0>), grid, block, 0, s, in, ld_in, (const int*)nullptr, R, C, out, ld_out, (uint16_t*)nullptr);
}

// ---------------------------------------------------------------------------------------------- more than one live forward
// The reference runs under eager autograd: two forwards and then (l1 + l2).backward(), or a decode between a forward and its backward,
// just work (CaptioningModel.py:22-36).  Here a forward's saved state is the pair (Ctx, TrainCtx) of pointers into TWO caller buffers:
// the vsr_prepare*() workspace and the training workspace.  Every vsr_train_forward files a copy of that pair under its generation; a
// later vsr_prepare*() / vsr_train_forward drops the copies whose buffers it is about to overwrite (address overlap) and keeps the
// rest.  A caller that gives its second forward OTHER buffers can therefore still differentiate the first: vsr_train_select() makes a
// filed forward the handle's current one again (pointers, image registrations, the per-batch rows of the f16x2 exponent table).
struct SavedForward { Ctx c; TrainCtx t; const char* ws_lo; size_t ws_bytes; };
struct SavedForwards { std::vector<SavedForward> v; };
static SavedForwards* new_saved_forwards() { return new SavedForwards(); }
static void free_saved_forwards(SavedForwards* s) { delete s; }
static void drop_saved_forwards(SavedForwards* s) { if (s) s->v.clear(); }
static bool ranges_overlap(const char* a, size_t na, const char* b, size_t nb) { return a && b && a < b + nb && b < a + na; }
static void drop_saved_forwards_in(SavedForwards* s, const void* lo, size_t bytes) {
    if (!s) return;
    const char* p = reinterpret_cast<const char*>(lo);
    for (size_t i = s->v.size(); i-- > 0;)
        if (ranges_overlap(p, bytes, s->v[i].ws_lo, s->v[i].ws_bytes) || ranges_overlap(p, bytes, s->v[i].t.tws_lo, s->v[i].t.tws_bytes))
            s->v.erase(s->v.begin() + i);
}
// the images of a training workspace's transposed operands: the GEMM builder finds them by address
static void register_train_images(vsr_handle* h, const TrainCtx& t) {
    h->b16.resize(h->b16_weights);
    for (const TrainCtx::Twin& tw : t.twins) h->b16.push_back(Bf16Range{tw.f, tw.f + tw.n, tw.b});
    h->h2t.clear();
    h->h2t_only = false;
    for (const TrainCtx::Img& im : t.imgs) h->h2t.push_back(H2Range{im.f, im.f + im.n, t.h2img + im.off, im.slot});
}
// rows of the f16x2 exponent / bound tables that vsr_prepare*() fills per batch (H2A_REGION, H2A_DET, H2A_ATT): dir 0 = table -> stash
__global__ void k_h2_stash(int* __restrict__ exps, unsigned* __restrict__ bounds, int* __restrict__ stash, int dir) {
    const int i = threadIdx.x;
    if (i >= 3) return;
    if (dir == 0) { stash[i] = exps[H2A_REGION + i]; stash[4 + i] = (int)bounds[H2A_REGION + i]; }
    else { exps[H2A_REGION + i] = stash[i]; bounds[H2A_REGION + i] = (unsigned)stash[4 + i]; }
}
static_assert(H2A_DET == H2A_REGION + 1 && H2A_ATT == H2A_REGION + 2, "k_h2_stash walks three consecutive slots");

extern "C" int vsr_train_select(vsr_handle* h, int64_t generation, void* stream) {
    if (!h || !h->tc) return fail("vsr_train_select: null handle");
    if (generation <= 0) return fail("vsr_train_select: generation %lld", (long long)generation);
    if (h->tc->valid && h->tc->generation == generation && h->prepared) return 0;
    for (const SavedForward& sf : h->saved->v)
        if (sf.t.generation == generation) {
            h->c = sf.c;
            *h->tc = sf.t;
            h->tc->valid = true;
            h->ws_lo = sf.ws_lo; h->ws_bytes = sf.ws_bytes;
            h->prepared = true;
            register_train_images(h, *h->tc);
            if (h->h2_on && h->tc->h2_stash) hipLaunchKernelGGL(k_h2_stash, dim3(1), dim3(64), 0, (hipStream_t)stream, h->h2_exps, h->h2_bounds, h->tc->h2_stash, 1);
            LAUNCHCHK();
            return 0;
        }
    return fail("vsr_train_select: the forward pass of generation %lld is gone (a later vsr_prepare*() / vsr_train_forward was given its "
                "workspaces, or the GEMM flavour / the weight binding changed)", (long long)generation);
}

extern "C" size_t vsr_train_workspace_bytes(const vsr_handle* h, int32_t B, int32_t T) {
    if (!h || !h->prepared || B != h->c.B || T <= 0) return 0;
    TrainCtx t;
    t.B = B; t.T = T;
    return carve_train(h, t, nullptr);
}

// ---------------------------------------------------------------------------------------------- forward with saves
extern "C" int vsr_train_forward(vsr_handle* h, const int64_t* word_in, const int64_t* slots, int32_t T, float* logp_words,
                                 float* logp_gates, void* train_ws, size_t train_ws_bytes, void* stream) {
    if (check_ready(h, "vsr_train_forward")) return 1;
    if (!word_in || !logp_words || !logp_gates || !train_ws) return fail("vsr_train_forward: null tensor");
    Ctx& c = h->c;
    if (c.beam != 1 && c.Mmax != c.B) return fail("vsr_train_forward: prepare() must be called with beam = 1");
    // index-list regions (vsr_prepare_indexed) train too, with one decoder row per image (the XE / SCST batches of train.py: every
    // sample brings its own detections): the backward pass sums dP over the entries of a row that name the same bank row
    if (c.ridx && !c.rows_are_images) return fail("vsr_train_forward: index-list regions with a row -> image map (row_img) are a decode-side format; training needs one row per image (row_img = NULL)");
    if (!slots && (T < 1 || T > c.L)) return fail("vsr_train_forward: without a slot trace step t reads slot t: need 1 <= T <= L (T %d, L %d)", T, c.L);
    hipStream_t s = (hipStream_t)stream;
    const vsr_dims& d = h->d;
    const vsr_weights& w = h->w;
    const int B = c.B, H = d.rnn_size, A = d.att_size, D = d.det_feat_size, E = d.input_encoding_size, V = d.vocab_size;
    const int in1 = (d.h2_first_lstm ? H : 0) + D + E, xoff = (d.h2_first_lstm ? H : 0) + D, in2 = H + D + (d.img_second_lstm ? D : 0);
    TrainCtx& t = *h->tc;
    t.valid = false;
    t.B = B; t.T = T;
    const size_t need = carve_train(h, t, reinterpret_cast<char*>(train_ws));
    if (need > train_ws_bytes) return fail("vsr_train_forward: training workspace too small (%zu < %zu)", train_ws_bytes, need);
    drop_saved_forwards_in(h->saved, train_ws, need);       // forwards filed in THIS buffer are overwritten now
    t.tws_lo = reinterpret_cast<const char*>(train_ws); t.tws_bytes = need;
    register_train_images(h, t);                           // the bf16 twins / fp16-pair images of this workspace's transposed operands
    const int TB = T * B;
    const size_t BH = (size_t)B * H;
    // f16x2 flavour: fp16-pair images of the A operands (all-DMA kernel); BH elements of 4 bytes per state slot
    const bool im = h->h2_on && !h->bf16_on && h->x3_on && h->h2_aimg && t.h1s16 && (B * H) % 8 == 0;
    const float isc = im ? 32768.f : 0.f;                  // (2^15: the exponent of the unit-bounded class)
    const int* att_exp = im ? h->h2_exps + H2A_ATT : nullptr;
    {   // the zero states of step 0 (and their images): one launch
        ZeroList zl;
        zl.add(t.h1s, BH * sizeof(float)); zl.add(t.c1s, BH * sizeof(float)); zl.add(t.h2s, BH * sizeof(float)); zl.add(t.c2s, BH * sizeof(float));
        if (im) { zl.add(t.h1s16, BH * 2 * sizeof(uint16_t)); zl.add(t.h2s16, BH * 2 * sizeof(uint16_t)); }
        if (zl.launch(s)) return fail("vsr_train_forward: zero launch failed");
    }
    hipLaunchKernelGGL(k_train_indices, dim3(cdiv(TB, 256)), dim3(256), 0, s, word_in, slots, T, B, V, c.L, t.word32, t.slot32, t.rows_bt, c.nvalid_dev + 2);
    hipLaunchKernelGGL(k_gather_rows, dim3(cdiv((long long)TB * E, 256)), dim3(256), 0, s, w.embed_weight, t.word32, TB, E, t.x_all);
    LAUNCHCHK();
    {   // the x part of every step's LSTM1 / gate pre-activations does not depend on the recurrence: one GEMM with M = T B
        GemmBuilder g;
        const float* Wih[3] = {w.lstm1_weight_ih, w.W1_is_weight, w.W1_ig_weight};
        const int Nn[3] = {4 * H, H, H}, off[3] = {0, 4 * H, 5 * H};
        for (int i = 0; i < 3; ++i) {
            GemmProb& p = g.prob(TB, Nn[i], t.scratch + off[i], 6 * H);
            // (with images: the rows are gathered from the embedding table's image by the GEMM itself)
            if (im) GemmBuilder::seg(p, w.embed_weight, E, t.word32, Wih[i] + xoff, in1, E, nullptr, H2A_EMBED);
            else GemmBuilder::seg(p, t.x_all, E, nullptr, Wih[i] + xoff, in1, E, nullptr, H2A_EMBED);
        }
        const int ns = g.finish(h);
        const long long stride = (long long)TB * 6 * H;
        if ((size_t)stride * ns > t.scratch_floats) return fail("training scratch too small for the x projection (%lld x %d)", stride, ns);
        for (int i = 0; i < 3; ++i) g.a.p[i].slab_stride = stride;
        if (g.launch(s, h)) return fail("train x-projection gemm launch failed");
        hipLaunchKernelGGL(k_slab_reduce, dim3(cdiv(stride, 256)), dim3(256), 0, s, t.scratch, ns, stride, stride, t.xproj_all);
        LAUNCHCHK();
    }

    for (int tt = 0; tt < T; ++tt) {
        const float *h1o = t.h1s + (size_t)tt * BH, *c1o = t.c1s + (size_t)tt * BH, *h2o = t.h2s + (size_t)tt * BH, *c2o = t.c2s + (size_t)tt * BH;
        float *h1n = t.h1s + (size_t)(tt + 1) * BH, *c1n = t.c1s + (size_t)(tt + 1) * BH, *h2n = t.h2s + (size_t)(tt + 1) * BH, *c2n = t.c2s + (size_t)(tt + 1) * BH;
        float* g1 = t.gates1 + (size_t)tt * B * 6 * H;
        float* g2 = t.gates2 + (size_t)tt * B * 4 * H;
        float *s_t = t.s_ts + (size_t)tt * BH, *g_t = t.g_ts + (size_t)tt * BH;
        float *hA = t.hAs + (size_t)tt * B * A, *sa = t.sas + (size_t)tt * B * A, *ga = t.gas + (size_t)tt * B * A;
        float *sent = t.sents + (size_t)tt * B * D, *att = t.atts + (size_t)tt * B * D, *alpha = t.alphas + (size_t)tt * B * (c.R + 1);
        const int* slot = t.slot32 + (size_t)tt * B;
        const uint16_t *h1o16 = im ? t.h1s16 + (size_t)tt * BH * 2 : nullptr, *h2o16 = im ? t.h2s16 + (size_t)tt * BH * 2 : nullptr;
        uint16_t *h1n16 = im ? t.h1s16 + (size_t)(tt + 1) * BH * 2 : nullptr, *h2n16 = im ? t.h2s16 + (size_t)(tt + 1) * BH * 2 : nullptr;
        uint16_t *s_t16 = im ? t.s_t16 : nullptr, *g_t16 = im ? t.g_t16 : nullptr, *att16 = im ? t.att16 : nullptr;
        {   // S1: the recurrent parts only (h2, h1 of the previous step); nothing to multiply at step 0
            int ns = 0;
            const long long stride = (long long)B * 6 * H;
            if (tt > 0) {
                GemmBuilder g;
                const float* Wih[3] = {w.lstm1_weight_ih, w.W1_is_weight, w.W1_ig_weight};
                const float* Whh[3] = {w.lstm1_weight_hh, w.W1_hs_weight, nullptr};
                const int Nn[3] = {4 * H, H, H}, off[3] = {0, 4 * H, 5 * H};
                for (int i = 0; i < 3; ++i) {
                    if (!d.h2_first_lstm && !Whh[i]) continue;
                    GemmProb& p = g.prob(B, Nn[i], c.scratch + off[i], 6 * H);
                    if (d.h2_first_lstm) GemmBuilder::seg(p, h2o, H, nullptr, Wih[i], in1, H, h2o16, H2A_UNIT);
                    if (Whh[i]) GemmBuilder::seg(p, h1o, H, nullptr, Whh[i], H, H, h1o16, H2A_UNIT);
                }
                ns = g.finish(h);
                for (int i = 0; i < g.a.nprob; ++i) g.a.p[i].slab_stride = stride;
                if (g.launch(s, h)) return fail("train S1 gemm launch failed");
            }
            hipLaunchKernelGGL(k_lstm1_train, dim3(cdiv((long long)B * H, 256)), dim3(256), 0, s, c.scratch, ns, stride, c.vproj,
                               t.xproj_all + (size_t)tt * B * 6 * H, c1o, B, H, h1n, c1n, s_t, c.gpre, g1,
                               d.h2_first_lstm ? 6 : 5 /* without h2 in the input the shift-gate block has no recurrent part */, h1n16, s_t16, isc);
        }
        {   // S2
            GemmBuilder g;
            GemmProb& p0 = g.prob(B, H, c.scratch, H + A);
            GemmBuilder::seg(p0, h1n, H, nullptr, w.W1_hg_weight, H, H, h1n16, H2A_UNIT);
            GemmProb& p1 = g.prob(B, A, c.scratch + H, H + A);
            GemmBuilder::seg(p1, h1n, H, nullptr, w.att_ha_weight, H, H, h1n16, H2A_UNIT);
            GemmProb& p2 = g.prob(B, D, nullptr, D + A);
            GemmBuilder::seg(p2, s_t, H, nullptr, w.s_fc_weight, H, H, s_t16, H2A_UNIT);
            GemmProb& p3 = g.prob(B, A, nullptr, D + A);
            GemmBuilder::seg(p3, s_t, H, nullptr, w.att_sa_weight, H, H, s_t16, H2A_UNIT);
            const int ns = g.finish(h);
            const long long stride_a = (long long)B * (H + A), stride_b = (long long)B * (D + A);
            float* c2b = c.scratch + stride_a * ns;
            g.a.p[0].slab_stride = g.a.p[1].slab_stride = stride_a;
            g.a.p[2].C = c2b; g.a.p[3].C = c2b + D;
            g.a.p[2].slab_stride = g.a.p[3].slab_stride = stride_b;
            if (g.launch(s, h)) return fail("train S2 gemm launch failed");
            // (Round 5: k_gate2's work inside the attention kernel's row blocks, as in decoding, with the sentinel / s_a / the shift gate stored
            // on the way for the backward pass: measured at no gain - 10.99-11.06 k against 10.85-11.10 k samples/s, profiles/r05_h_* - and not kept.)
            const long long n = (long long)B * (H + A + D + A);
            hipLaunchKernelGGL(k_gate2, dim3(cdiv(n, 256)), dim3(256), 0, s, c.scratch, c2b, ns, stride_a, stride_b, c.gpre, c1n, w.s_fc_bias,
                               B, H, A, D, g_t, hA, sent, sa, g1, g_t16, isc);
        }
        {
            const size_t smem = (size_t)(2 * A + D + c.R + 1 + 8 + c.R) * sizeof(float);
            const int np = (h->attend_parts > 1 && B * h->attend_parts <= h->attend_limit && D % (4 * h->attend_parts) == 0) ? h->attend_parts : 1;      // (as in run_step)
            if (D >= 2048) hipLaunchKernelGGL(k_attend<512>, dim3(cdiv(B * np, 8) * 8), dim3(512), smem, s, Gate2Args{}, hA, sa, sent, c.P, c.regions, c.rmask, c.ridx, slot, 0, 1, B, c.L,
                               c.R, A, D, w.att_a_weight, w.att_s_weight, att, c.zsum, alpha, att16, att_exp, np);
            else hipLaunchKernelGGL(k_attend<256>, dim3(cdiv(B * np, 8) * 8), dim3(256), smem, s, Gate2Args{}, hA, sa, sent, c.P, c.regions, c.rmask, c.ridx, slot, 0, 1, B, c.L,
                               c.R, A, D, w.att_a_weight, w.att_s_weight, att, c.zsum, alpha, att16, att_exp, np);
        }
        {   // S5
            GemmBuilder g;
            GemmProb& p0 = g.prob(B, 4 * H, c.scratch, 4 * H);
            GemmBuilder::seg(p0, h1n, H, nullptr, w.lstm2_weight_ih, in2, H, h1n16, H2A_UNIT);
            GemmBuilder::seg(p0, att, D, nullptr, w.lstm2_weight_ih + H, in2, D, att16, H2A_ATT);
            if (tt > 0) GemmBuilder::seg(p0, h2o, H, nullptr, w.lstm2_weight_hh, H, H, h2o16, H2A_UNIT);
            GemmProb& p1 = g.prob(B, A, nullptr, A);
            GemmBuilder::seg(p1, g_t, H, nullptr, w.att_ga_weight, H, H, g_t16, H2A_UNIT);
            const int ns = g.finish(h);
            const long long stride = (long long)B * 4 * H, stride_g = (long long)B * A;
            float* gas = c.scratch + stride * ns;
            g.a.p[0].slab_stride = stride;
            g.a.p[1].C = gas; g.a.p[1].slab_stride = stride_g;
            if (g.launch(s, h)) return fail("train S5 gemm launch failed");
            GateLogitArgs gl{gas, ns, stride_g, hA, w.att_g_weight, c.zsum, nullptr, slot, 1, c.L, B, A,
                             logp_gates + (size_t)tt * 2, (long long)T * 2};
            gl.ga_out = ga;
            const int gblocks = B;
            hipLaunchKernelGGL(k_fwd_tail, dim3(gblocks + cdiv((long long)B * H, 256)), dim3(256), 0, s, gl, gblocks, c.scratch, ns, stride,
                               w.lstm2_bias_ih, w.lstm2_bias_hh, d.img_second_lstm ? c.vproj2 : nullptr, c2o, B, H, h2n, c2n, g2, h2n16, isc);
        }
        LAUNCHCHK();
    }
    {   // S6 for all steps at once: logits = h2[(b, t)] . out_fc^T (M = T B rows gathered in (b, t) order, so that the
        // (B, T, V) log-prob tensor is written row by row), then one log_softmax launch
        GemmBuilder g;
        GemmProb& p0 = g.prob(TB, V, t.scratch, V);
        GemmBuilder::seg(p0, t.h2s, H, t.rows_bt, w.out_fc_weight, H, H, im ? t.h2s16 : nullptr, H2A_UNIT);
        const int ns = g.finish(h);
        const long long stride = (long long)TB * V;
        if ((size_t)stride * ns > t.scratch_floats) return fail("training scratch too small for the vocabulary projection (%lld x %d)", stride, ns);
        g.a.p[0].slab_stride = stride;
        if (g.launch(s, h)) return fail("train S6 gemm launch failed");
        const int lds_row = V <= VOCAB_LDS_MAX ? 1 : 0;
        const size_t vsm = lds_row ? (size_t)V * sizeof(float) : 0;
        const GateLogitArgs no_gate{nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0, 0, nullptr, 0};
#define TRAIN_VOCAB_ARGS t.scratch, ns, stride, w.out_fc_bias, TB, V, (int)VM_FULL, c.top_v, c.top_i, logp_words, (long long)V, (const int*)nullptr, \
                         (uint64_t)0, (uint32_t)0, (const float*)nullptr, (const int*)nullptr, 1, c.L, 0, h->vt_ptr, h->vt_ids, h->n_verbs, lds_row, no_gate, c.nvalid_dev + 2
        if (V >= 4096) hipLaunchKernelGGL((k_vocab<1, 512>), dim3(TB), dim3(512), vsm, s, TRAIN_VOCAB_ARGS);
        else hipLaunchKernelGGL((k_vocab<1, 256>), dim3(TB), dim3(256), vsm, s, TRAIN_VOCAB_ARGS);
#undef TRAIN_VOCAB_ARGS
        LAUNCHCHK();
    }
    t.logp_w = logp_words;
    t.logp_g = logp_gates;
    t.valid = true;
    t.generation = ++h->gen_counter;
    if (h->h2_on) hipLaunchKernelGGL(k_h2_stash, dim3(1), dim3(64), 0, s, h->h2_exps, h->h2_bounds, t.h2_stash, 0);
    LAUNCHCHK();
    h->saved->v.push_back(SavedForward{c, t, h->ws_lo, h->ws_bytes});
    return 0;
}

// Identity of the saved forward pass (0 = none).  The training workspace holds ONE forward at a time: a caller that keeps
// several autograd nodes alive compares the value it got after its forward with the current one before differentiating.
extern "C" int64_t vsr_train_generation(const vsr_handle* h) { return (h && h->tc && h->tc->valid) ? (int64_t)h->tc->generation : 0; }

// dlogits in (t, b) row order from the (B, T, V) tensors
__global__ __launch_bounds__(256) void k_dlogits_tb(const float* __restrict__ logp, const float* __restrict__ dlogp, int B, int T, int V,
                                                    int Vp, float* __restrict__ dlogits, int* bm) {
    __shared__ float red[4];
    const int tt = blockIdx.x / B, b = blockIdx.x % B;
    const long long src = ((long long)b * T + tt) * V, dst = (long long)blockIdx.x * Vp;
    const int tid = threadIdx.x;
    // 16 bytes per lane where the rows allow it (V a multiple of 4: every row of the (B, T, V) tensors starts 16-byte aligned)
    const bool vec = (V & 3) == 0 && (Vp & 3) == 0 && ((reinterpret_cast<uintptr_t>(logp) | reinterpret_cast<uintptr_t>(dlogp) | reinterpret_cast<uintptr_t>(dlogits)) & 15) == 0;
    float s = 0.f;
    if (vec) {
        for (int v = tid * 4; v < V; v += 1024) {
            const float4 g = *reinterpret_cast<const float4*>(dlogp + src + v);
            s += (g.x + g.y) + (g.z + g.w);
        }
    } else {
        for (int v = tid; v < V; v += 256) s += dlogp[src + v];
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    float mx = 0.f;
    if (vec) {
        for (int v = tid * 4; v < Vp; v += 1024) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v < V) {
                const float4 g = *reinterpret_cast<const float4*>(dlogp + src + v), l = *reinterpret_cast<const float4*>(logp + src + v);
                o = make_float4(g.x - expf(l.x) * tot, g.y - expf(l.y) * tot, g.z - expf(l.z) * tot, g.w - expf(l.w) * tot);
            }
            *reinterpret_cast<float4*>(dlogits + dst + v) = o;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
    } else {
        for (int v = tid; v < Vp; v += 256) {
            const float o = v < V ? dlogp[src + v] - expf(logp[src + v]) * tot : 0.f;
            dlogits[dst + v] = o;
            mx = fmaxf(mx, fabsf(o));
        }
    }
    if (bm) block_absmax_to(mx, bm);
}

__global__ void k_sum_over_t(const float* __restrict__ X, int T, long long per_t, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_t) return;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += X[t * per_t + i];
    out[i] = s;
}

// ---------------------------------------------------------------------------------------------- backward
extern "C" int vsr_train_backward(vsr_handle* h, const float* grad_logp_words, const float* grad_logp_gates, const vsr_weights* grads,
                                  void* stream) {
    if (check_ready(h, "vsr_train_backward")) return 1;
    TrainCtx& t = *h->tc;
    if (!t.valid) return fail("vsr_train_backward: no saved forward (call vsr_train_forward first)");
    if (!grad_logp_words || !grad_logp_gates || !grads) return fail("vsr_train_backward: null tensor");
    hipStream_t s = (hipStream_t)stream;
    const vsr_dims& d = h->d;
    const vsr_weights& w = h->w;
    Ctx& c = h->c;
    const int B = t.B, T = t.T, TB = t.TB, TBp = t.TBp, Bp = t.Bp, RLp = t.RLp;
    const int H = d.rnn_size, A = d.att_size, D = d.det_feat_size, E = d.input_encoding_size, V = d.vocab_size;
    const int in1 = (d.h2_first_lstm ? H : 0) + D + E, voff = d.h2_first_lstm ? H : 0, xoff = voff + D, in2 = H + D + (d.img_second_lstm ? D : 0);
    const int RL = c.B * c.L * c.R, R1 = c.R + 1;
    const size_t BH = (size_t)B * H;
    float* const* G = reinterpret_cast<float* const*>(grads);     // same field order as vsr_weights
    enum { g_embed, g_Wis, g_bis, g_Whs, g_bhs, g_Wva, g_Wha, g_wa, g_Wsa, g_ws, g_Wih1, g_Whh1, g_bih1, g_bhh1, g_Wih2, g_Whh2, g_bih2,
           g_bhh2, g_Wout, g_bout, g_Wsfc, g_bsfc, g_Wig, g_big, g_Whg, g_bhg, g_Wga, g_wg };
    for (int i = 0; i < 28; ++i)
        if (!G[i]) return fail("vsr_train_backward: gradient pointer %d is null", i);

    // f16x2 flavour: the W operands of the backward GEMMs - transposed weights here, transposed activations in phase B - are written as
    // fp16-pair images by the transposing kernel itself; the A operands are gradients: their producers fold max |x| into "dynamic" slots
    // of the exponent table (block tt of 8 slots for step tt, blocks 62 / 63 for the whole-pass operands)
    // (the same conditions GemmBuilder::finish() routes a launch to the f16x2 kernels by: a backward pass that wrote only images for
    // launches that then fall through to the fp32-operand kernels would read unwritten buffers - finish() refuses such a launch too)
    const bool h2b = h->h2_on && h->x3_on && h->gemm_tile == 0 && !h->bf16_on && t.h2img && T <= H2_NDYN / 8 - 2;
    h->h2t_only = h2b;
    int* dyn = h2b ? h->h2_exps + H2_DYN0 : nullptr;
    enum { DY_dpre2, DY_dga, DY_dq, DY_dhA, DY_dsent, DY_dsa, DY_dpre1 };               // per-step block
    enum { DW_dlogits = 62 * 8, DW_dP, DW_dpre1sum, DW_dpre2sum, DW_dpre1all, DW_step = 63 * 8 };      // whole-pass slots (DW_step + DY_x: max over the steps)
    auto dslot = [&](int i) { return h2b ? H2_DYN0 + i : (int)H2A_NONE; };
    auto dptr = [&](int i) { return h2b ? dyn + i : (int*)nullptr; };
    if (h2b) HIPCHK(hipMemsetAsync(dyn, 0, H2_NDYN * sizeof(int), s));
    // ---- transposed weights (the optimizer may have changed them since the last call): one batched launch (TransBatch)
    TransBatch tbw;
    transpose(h, s, w.lstm1_weight_ih, in1, 4 * H, in1, t.wT_ih1, 4 * H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.W1_is_weight, in1, H, in1, t.wT_is, H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.W1_ig_weight, in1, H, in1, t.wT_ig, H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.lstm1_weight_hh, H, 4 * H, H, t.wT_hh1, 4 * H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.W1_hs_weight, H, H, H, t.wT_hs, H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.lstm2_weight_ih, in2, 4 * H, in2, t.wT_ih2, 4 * H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.lstm2_weight_hh, H, 4 * H, H, t.wT_hh2, 4 * H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.W1_hg_weight, H, H, H, t.wT_hg, H, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.att_ha_weight, H, A, H, t.wT_ha, A, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.s_fc_weight, H, D, H, t.wT_sfc, D, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.att_sa_weight, H, A, H, t.wT_sa, A, nullptr, h2b, -1, nullptr, &tbw);
    transpose(h, s, w.att_ga_weight, H, A, H, t.wT_ga, A, nullptr, h2b, -1, nullptr, &tbw);
    const int Vp = (int)up4(V);        // K of the dh2_vocab GEMM must be a multiple of 8: zero-padded columns
    if (Vp != V) {
        HIPCHK(hipMemsetAsync(t.wT_out, 0, (size_t)H * Vp * sizeof(float), s));
        if (uint16_t* tw = h->bf16_on ? const_cast<uint16_t*>(h->map16(t.wT_out)) : nullptr) HIPCHK(hipMemsetAsync(tw, 0, (size_t)H * Vp * sizeof(uint16_t), s));
        if (const H2Range* r2 = h2b ? h->map_h2(t.wT_out) : nullptr) HIPCHK(hipMemsetAsync(const_cast<float*>(r2->img), 0, (size_t)H * Vp * sizeof(float), s));
    }
    transpose(h, s, w.out_fc_weight, H, V, H, t.wT_out, Vp, nullptr, h2b, -1, nullptr, &tbw);
    tbw.flush(s);
    {   // dP, the carries of the last step: one launch
        ZeroList zl;
        zl.add(t.dP, (size_t)RL * A * sizeof(float));
        zl.add(t.dh1_c, BH * sizeof(float)); zl.add(t.dh2_c, BH * sizeof(float)); zl.add(t.dc1_c[0], BH * sizeof(float)); zl.add(t.dc2_c[0], BH * sizeof(float));
        if (zl.launch(s)) return fail("vsr_train_backward: zero launch failed");
    }
    const int NV = c.nvalid, NVp = (int)up4((size_t)NV);     // non-padding region rows: the only ones att_va saw
    // under a caller's row bound (vsr_set_valid_rows_bound) NV is the BOUND and the list's tail [n, NV) repeats its first entry: the two
    // gathers below read those rows as zeros (k_transpose_t's rlimit = the device-side count), so att_va's gradient sums over n rows
    const int* nv_dev = c.bounded ? c.nvalid_dev : nullptr;
    // (the K padding of the transposed operands - TBp, Bp, NVp columns - is zero-filled by the transposing kernel itself)

    // ---- phase 0: dlogits (t,b) and the vocabulary part of dh2 for every step
    hipLaunchKernelGGL(k_dlogits_tb, dim3(TB), dim3(256), 0, s, t.logp_w, grad_logp_words, B, T, V, Vp, t.dlogits, dptr(DW_dlogits));
    if (gemm_to1(h, t, s, TB, H, Vp, t.dlogits, Vp, t.wT_out, Vp, t.dh2_voc, H, dslot(DW_dlogits))) return 1;
    LAUNCHCHK();

    // ---- phase A: reverse time.  Per step: k_bwd_head, GEMM 1, k_bwd_mid, k_dalpha, k_attend_bwd, GEMM 2, k_bwd_tail, GEMM 3
    int cb = 0;
    int ns3 = 0;                              // slabs of the later step's GEMM 3 waiting in t.scratch (consumed by k_bwd_head)
    const long long st3 = (long long)B * H;
    for (int tt = T - 1; tt >= 0; --tt, cb ^= 1) {
        const float *c1 = t.c1s + (size_t)(tt + 1) * BH, *c1p = t.c1s + (size_t)tt * BH;
        const float *c2 = t.c2s + (size_t)(tt + 1) * BH, *c2p = t.c2s + (size_t)tt * BH;
        const float* g1 = t.gates1 + (size_t)tt * B * 6 * H;
        const float* g2 = t.gates2 + (size_t)tt * B * 4 * H;
        const float *hA = t.hAs + (size_t)tt * B * A, *sa = t.sas + (size_t)tt * B * A, *ga = t.gas + (size_t)tt * B * A;
        const float *sent = t.sents + (size_t)tt * B * D, *alpha = t.alphas + (size_t)tt * B * R1;
        const int* slot = t.slot32 + (size_t)tt * B;
        float* dpre1 = t.dpre1 + (size_t)tt * B * 6 * H;
        float* dpre2 = t.dpre2 + (size_t)tt * B * 4 * H;
        float *dhA = t.dhA_all + (size_t)tt * B * A, *dsent = t.dsent_all + (size_t)tt * B * D, *dsa = t.dsa_all + (size_t)tt * B * A;
        float* dga = t.dga_all + (size_t)tt * B * A;
        float *dwa = t.dwa_rows + (size_t)tt * B * A, *dws = t.dws_rows + (size_t)tt * B * A, *dwg = t.dwg_rows + (size_t)tt * B * A;

        {   // gate log-probs -> dga, dhA (first writer), dzsum;  carries of step tt + 1 + vocabulary part -> LSTM2 backward
            BwdHeadArgs q;
            q.lg = t.logp_g + (size_t)tt * 2; q.dlg = grad_logp_gates + (size_t)tt * 2; q.lg_stride = (long long)T * 2;
            q.ga = ga; q.hA = hA; q.w_g = w.att_g_weight; q.A = A;
            q.dga = dga; q.dhA = dhA; q.dzsum = t.dzsum; q.dwg_rows = dwg; q.gblocks = B;
            q.s_h1 = t.scratch; q.s_h2 = d.h2_first_lstm ? t.scratch + st3 * ns3 : nullptr; q.nslab3 = ns3; q.stride3 = st3;
            q.dh1_c = t.dh1_c; q.dh2_c = t.dh2_c; q.dh2_voc = t.dh2_voc + (size_t)tt * BH;
            q.dc_next = t.dc2_c[cb]; q.gates2 = g2; q.c2 = c2; q.c2_prev = tt > 0 ? c2p : nullptr;
            q.M = B; q.H = H; q.dpre2 = dpre2; q.dc_prev = t.dc2_c[cb ^ 1];
            q.bm_dga = dptr(tt * 8 + DY_dga); q.bm_dpre2 = dptr(tt * 8 + DY_dpre2);
            hipLaunchKernelGGL(k_bwd_head, dim3(q.gblocks + cdiv((long long)BH, 256)), dim3(256), 0, s, q);
        }
        {   // grouped GEMM 1: dpre2 -> [dh1_a | datt (| dvbar)] , dh2 carry (hh part);  dga -> dg_t
            GemmBuilder g;
            GemmProb& p0 = g.prob(B, H + D, nullptr, H + D);
            GemmBuilder::seg(p0, dpre2, 4 * H, nullptr, t.wT_ih2, 4 * H, 4 * H, nullptr, dslot(tt * 8 + DY_dpre2));
            GemmProb& p1 = g.prob(B, H, nullptr, H);
            GemmBuilder::seg(p1, dpre2, 4 * H, nullptr, t.wT_hh2, 4 * H, 4 * H, nullptr, dslot(tt * 8 + DY_dpre2));
            GemmProb& p2 = g.prob(B, H, nullptr, H);
            GemmBuilder::seg(p2, dga, A, nullptr, t.wT_ga, A, A, nullptr, dslot(tt * 8 + DY_dga));
            const int ns1 = g.finish(h);
            BwdMidArgs q;
            q.nslab = ns1; q.st0 = (long long)B * (H + D); q.st1 = (long long)B * H; q.st2 = (long long)B * H;
            float *C0 = t.scratch, *C1 = C0 + q.st0 * ns1, *C2 = C1 + q.st1 * ns1;
            g.a.p[0].C = C0; g.a.p[0].slab_stride = q.st0;
            g.a.p[1].C = C1; g.a.p[1].slab_stride = q.st1;
            g.a.p[2].C = C2; g.a.p[2].slab_stride = q.st2;
            if (g.launch(s, h)) return fail("bwd gemm 1 launch failed");
            // datt; dg_t -> shift-gate backward (dq into dpre1[:, 5H:6H], dtc); dh1 so far (= dh1_a + carry, into dh_tot) and the
            // hh part of the new dh2 carry (the LSTM1-input part is added by the next k_bwd_head from GEMM 3's slabs)
            q.C0 = C0; q.C1 = C1; q.C2 = C2; q.M = B; q.H = H; q.D = D;
            q.datt = t.datt; q.dh1_c = t.dh1_c; q.dh_tot = t.dh_tot; q.dh2_c = t.dh2_c;
            q.gates1 = g1; q.c1 = c1; q.dq = dpre1 + 5 * H; q.dtc = t.dtc; q.bm_dq = dptr(tt * 8 + DY_dq);
            const int wmax = D > H ? D : H;
            hipLaunchKernelGGL(k_bwd_mid, dim3(cdiv((long long)B * wmax, 256), 3), dim3(256), 0, s, q);
        }
        // attention
        {
            hipLaunchKernelGGL(k_dalpha, dim3(cdiv((long long)B * R1, 4)), dim3(256), 0, s, t.datt, sent, c.regions, c.rmask, c.ridx, slot, B, c.L, c.R, D,
                               t.dalpha);
            // two workgroups per row in launches of <= 128 rows (as k_attend): half the A columns each, its 512 threads = 256 columns x 2 row groups
            const int NTb = A >= 512 ? 512 : 256;
            int np = (h->attend_parts > 1 && B * h->attend_parts <= h->attend_limit && A % h->attend_parts == 0 && D % (4 * h->attend_parts) == 0) ? h->attend_parts : 1;
            if (np > 1 && (NTb % (A / np) != 0 || NTb / (A / np) < 1 || (A / np) > NTb)) np = 1;
            const int RG = np > 1 ? NTb / (A / np) : 1;
            const size_t smem = (size_t)(R1 + 8 + (np > 1 ? (RG - 1) * (A / np) * 2 : 0)) * sizeof(float);
            if (A >= 512) hipLaunchKernelGGL(k_attend_bwd<512>, dim3(cdiv(B * np, 8) * 8), dim3(512), smem, s, t.datt, t.dalpha, t.dzsum, alpha, hA, sa, sent, c.P, c.regions, c.rmask, c.ridx,
                               slot, 0, B, c.L, c.R, A, D, w.att_a_weight, w.att_s_weight, dsent, dsa, dhA, t.dP, dwa, dws,
                               dptr(tt * 8 + DY_dhA), dptr(tt * 8 + DY_dsent), dptr(tt * 8 + DY_dsa), np);
            else hipLaunchKernelGGL(k_attend_bwd<256>, dim3(cdiv(B * np, 8) * 8), dim3(256), smem, s, t.datt, t.dalpha, t.dzsum, alpha, hA, sa, sent, c.P, c.regions, c.rmask, c.ridx,
                               slot, 0, B, c.L, c.R, A, D, w.att_a_weight, w.att_s_weight, dsent, dsa, dhA, t.dP, dwa, dws,
                               dptr(tt * 8 + DY_dhA), dptr(tt * 8 + DY_dsent), dptr(tt * 8 + DY_dsa), np);
        }
        // grouped GEMM 2: [dq | dhA] -> dh1_b ; [dsent | dsa] -> ds_t;  then the sentinel gate and LSTM1 pointwise backward
        {
            GemmBuilder g;
            GemmProb& p0 = g.prob(B, H, nullptr, H);
            GemmBuilder::seg(p0, dpre1 + 5 * H, 6 * H, nullptr, t.wT_hg, H, H, nullptr, dslot(tt * 8 + DY_dq));
            GemmBuilder::seg(p0, dhA, A, nullptr, t.wT_ha, A, A, nullptr, dslot(tt * 8 + DY_dhA));
            GemmProb& p1 = g.prob(B, H, nullptr, H);
            GemmBuilder::seg(p1, dsent, D, nullptr, t.wT_sfc, D, D, nullptr, dslot(tt * 8 + DY_dsent));
            GemmBuilder::seg(p1, dsa, A, nullptr, t.wT_sa, A, A, nullptr, dslot(tt * 8 + DY_dsa));
            const int ns = g.finish(h);
            const long long st = (long long)B * H;
            float *Ca = t.scratch, *Cb = Ca + st * ns;
            g.a.p[0].C = Ca; g.a.p[0].slab_stride = st;
            g.a.p[1].C = Cb; g.a.p[1].slab_stride = st;
            if (g.launch(s, h)) return fail("bwd gemm 2 launch failed");
            hipLaunchKernelGGL(k_bwd_tail, dim3(cdiv((long long)BH, 256)), dim3(256), 0, s, Ca, Cb, ns, st, t.dh_tot, t.dtc, t.dc1_c[cb], g1, c1,
                               tt > 0 ? c1p : (const float*)nullptr, B, H, dpre1, t.dc1_c[cb ^ 1], dptr(tt * 8 + DY_dpre1));
        }
        // grouped GEMM 3: dpre1 -> dh1 carry, dh2 carry (LSTM1 input part); its slabs stay in t.scratch for the next k_bwd_head
        ns3 = 0;
        if (tt > 0) {
            GemmBuilder g;
            GemmProb& p1 = g.prob(B, H, nullptr, H);
            const int s1 = dslot(tt * 8 + DY_dpre1), sq = dslot(tt * 8 + DY_dq);
            GemmBuilder::seg(p1, dpre1, 6 * H, nullptr, t.wT_hh1, 4 * H, 4 * H, nullptr, s1);
            GemmBuilder::seg(p1, dpre1 + 4 * H, 6 * H, nullptr, t.wT_hs, H, H, nullptr, s1);
            if (d.h2_first_lstm) {
                GemmProb& p0 = g.prob(B, H, nullptr, H);
                GemmBuilder::seg(p0, dpre1, 6 * H, nullptr, t.wT_ih1, 4 * H, 4 * H, nullptr, s1);          // rows 0..H-1 of W_ih1^T = h2 columns
                GemmBuilder::seg(p0, dpre1 + 4 * H, 6 * H, nullptr, t.wT_is, H, H, nullptr, s1);
                GemmBuilder::seg(p0, dpre1 + 5 * H, 6 * H, nullptr, t.wT_ig, H, H, nullptr, sq);
            }
            ns3 = g.finish(h);
            g.a.p[0].C = t.scratch; g.a.p[0].slab_stride = st3;
            if (d.h2_first_lstm) { g.a.p[1].C = t.scratch + st3 * ns3; g.a.p[1].slab_stride = st3; }
            if (g.launch(s, h)) return fail("bwd gemm 3 launch failed");
        }
        LAUNCHCHK();
    }

    // ---- phase B: weight gradients, reduction over all T*B rows
    const float* h1prev = t.h1s;              // rows (t,b): state entering step t
    const float* h1cur = t.h1s + BH;
    const float* h2prev = t.h2s;
    const float* h2cur = t.h2s + BH;
    TransBatch tbx, tby;
    transpose(h, s, h2prev, H, TB, H, t.tX_h2prev, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, t.x_all, E, TB, E, t.tX_x, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, h1prev, H, TB, H, t.tX_h1prev, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, h1cur, H, TB, H, t.tX_h1, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, t.atts, D, TB, D, t.tX_att, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, t.s_ts, H, TB, H, t.tX_st, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, t.g_ts, H, TB, H, t.tX_gt, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, h2cur, H, TB, H, t.tX_h2, TBp, nullptr, h2b, -1, nullptr, &tbx);
    transpose(h, s, c.vbar, D, B, D, t.tX_vbar, Bp, nullptr, h2b, -1, nullptr, &tbx);
    tbx.flush(s);
    if (NV > 0) transpose(h, s, c.regions, (long long)D, NV, D, t.tX_reg, (long long)NVp, c.vlist, h2b, -1, nv_dev);
    if (h2b) hipLaunchKernelGGL(k_h2_dyn_fold, dim3(1), dim3(64), 0, s, dyn, T);
    const int sP1 = dslot(DW_step + DY_dpre1), sQ = dslot(DW_step + DY_dq), sP2 = dslot(DW_step + DY_dpre2);
    // the transposed gradients are the A operands of the weight-gradient GEMMs: with the whole-pass bounds known (k_h2_dyn_fold) they are
    // written as fp16-pair images IN PLACE of the fp32 values and those GEMMs take the all-DMA kernel (tyi: the buffer holds an image)
    const bool tyi = h2b && h->h2_aimg && TBp % 8 == 0 && (reinterpret_cast<uintptr_t>(t.tY_dpre1) & 255) == 0;     // (every buffer of the workspace shares that alignment)
    const int sY1 = dslot(DW_dpre1all);
    transpose(h, s, t.dpre1, 6 * H, TB, 6 * H, t.tY_dpre1, TBp, nullptr, tyi, sY1, nullptr, &tby);
    transpose(h, s, t.dpre2, 4 * H, TB, 4 * H, t.tY_dpre2, TBp, nullptr, tyi, sP2, nullptr, &tby);
    transpose(h, s, t.dlogits, Vp, TB, V, t.tY_dlogits, TBp, nullptr, tyi, dslot(DW_dlogits), nullptr, &tby);
    transpose(h, s, t.dhA_all, A, TB, A, t.tY_dhA, TBp, nullptr, tyi, dslot(DW_step + DY_dhA), nullptr, &tby);
    transpose(h, s, t.dsent_all, D, TB, D, t.tY_dsent, TBp, nullptr, tyi, dslot(DW_step + DY_dsent), nullptr, &tby);
    transpose(h, s, t.dsa_all, A, TB, A, t.tY_dsa, TBp, nullptr, tyi, dslot(DW_step + DY_dsa), nullptr, &tby);
    transpose(h, s, t.dga_all, A, TB, A, t.tY_dga, TBp, nullptr, tyi, dslot(DW_step + DY_dga), nullptr, &tby);
    tby.flush(s);
    const float* dP_rows = t.dP;
    if (c.ridx) {
        // index lists: several slot entries of an image name the same bank row; att_va's gradient runs over bank rows, so the
        // entry gradients are first summed per bank row, in ascending entry order (deterministic, like the embedding gradient)
        hipLaunchKernelGGL(k_dP_to_bank, dim3(c.n_img * c.Rb), dim3(128), 0, s, t.dP, c.ridx, c.L * c.R, c.Rb, A, t.dP_bank);
        dP_rows = t.dP_bank;
    }
    if (h2b && NV > 0) {
        const long long ndp = (long long)(c.ridx ? (size_t)c.n_img * c.Rb : (size_t)RL) * A;
        hipLaunchKernelGGL(k_absmax, dim3((unsigned)std::min<long long>(1024, cdiv(ndp, 1024))), dim3(256), 0, s, dP_rows, ndp, reinterpret_cast<unsigned*>(dyn + DW_dP));
    }
    const bool tyiP = tyi && NVp % 8 == 0;
    if (NV > 0) transpose(h, s, dP_rows, (long long)A, NV, A, t.tY_dP, (long long)NVp, c.vlist, tyiP, dslot(DW_dP), nv_dev);
    hipLaunchKernelGGL(k_sum_over_t, dim3(cdiv((long long)B * 6 * H, 256)), dim3(256), 0, s, t.dpre1, T, (long long)B * 6 * H, t.dpre1sum);
    transpose(h, s, t.dpre1sum, 6 * H, B, 6 * H, t.tY_dpre1sum, Bp);
    LAUNCHCHK();

    // Gradients are finished bucket by bucket, largest first, and an event is recorded after each bucket
    // (vsr_train_bucket_map / vsr_train_wait_bucket): a data-parallel caller starts the all-reduce of a finished bucket on
    // a side stream while the remaining weight-gradient GEMMs run.  The last bucket is the smallest (14 MB).
    if (!h->bucket_ev[0])
        for (int i = 0; i < VSR_GRAD_BUCKETS; ++i) HIPCHK(hipEventCreateWithFlags(&h->bucket_ev[i], hipEventDisableTiming));
    // ---- bucket 0: lstm_cell_1.weight_ih / W1_is / W1_ig: row blocks [0,4H), [4H,5H), [5H,6H) of dpre1^T against [h2_prev | vbar | x]
    {
        float* Gw[3] = {G[g_Wih1], G[g_Wis], G[g_Wig]};
        const int r0[3] = {0, 4 * H, 5 * H}, nr[3] = {4 * H, H, H};
        auto dy = [&](int i) { return t.tY_dpre1 + (size_t)r0[i] * TBp; };
        auto sa = [&](int i) { return tyi ? sY1 : (i == 2 ? sQ : sP1); };
        auto h2part = [&](int i) { return GProb{nr[i], H, TBp, dy(i), TBp, t.tX_h2prev, TBp, Gw[i], in1, sa(i), tyi}; };
        auto xpart = [&](int i) { return GProb{nr[i], E, TBp, dy(i), TBp, t.tX_x, TBp, Gw[i] + xoff, in1, sa(i), tyi}; };
        auto vpart = [&](int i) { return GProb{nr[i], D, Bp, t.tY_dpre1sum + (size_t)r0[i] * Bp, Bp, t.tX_vbar, Bp, Gw[i] + voff, in1, dslot(DW_dpre1sum), false}; };
        // (grouped launches, round 6: the two 4H-row products are 256 whole tiles together; the four H-row ones share a launch; the three
        // short-K image-constant parts share one)
        if (d.h2_first_lstm) {
            const GProb a[2] = {h2part(0), xpart(0)};
            if (gemm_group(h, t, s, a, 2)) return 1;
            const GProb b[4] = {h2part(1), xpart(1), h2part(2), xpart(2)};
            if (gemm_group(h, t, s, b, 4)) return 1;
        } else {
            const GProb a[3] = {xpart(0), xpart(1), xpart(2)};
            if (gemm_group(h, t, s, a, 3)) return 1;
        }
        const GProb c3[3] = {vpart(0), vpart(1), vpart(2)};
        if (gemm_group(h, t, s, c3, 3)) return 1;
    }
    HIPCHK(hipEventRecord(h->bucket_ev[0], s));
    // ---- bucket 1: lstm_cell_2
    {   // (the attended-vector part is 256 whole tiles on its own; the h1 part and lstm_cell_2.weight_hh are 256 together)
        const GProb a[2] = {GProb{4 * H, H, TBp, t.tY_dpre2, TBp, t.tX_h1, TBp, G[g_Wih2], in2, sP2, tyi},
                            GProb{4 * H, H, TBp, t.tY_dpre2, TBp, t.tX_h2prev, TBp, G[g_Whh2], H, sP2, tyi}};
        if (gemm_group(h, t, s, a, 2)) return 1;
    }
    if (gemm_to1(h, t, s, 4 * H, D, TBp, t.tY_dpre2, TBp, t.tX_att, TBp, G[g_Wih2] + H, in2, sP2, tyi)) return 1;
    if (d.img_second_lstm) {
        hipLaunchKernelGGL(k_sum_over_t, dim3(cdiv((long long)B * 4 * H, 256)), dim3(256), 0, s, t.dpre2, T, (long long)B * 4 * H, t.dpre2sum);
        transpose(h, s, t.dpre2sum, 4 * H, B, 4 * H, t.tY_dpre2sum, Bp);
        if (gemm_to1(h, t, s, 4 * H, D, Bp, t.tY_dpre2sum, Bp, t.tX_vbar, Bp, G[g_Wih2] + H + D, in2, dslot(DW_dpre2sum))) return 1;
    }
    colsum(t, s, t.dpre2, (long long)4 * H, TB, 4 * H, G[g_bih2], G[g_bhh2]);
    HIPCHK(hipEventRecord(h->bucket_ev[1], s));
    // ---- bucket 2: out_fc and the embedding
    if (gemm_to1(h, t, s, V, H, TBp, t.tY_dlogits, TBp, t.tX_h2, TBp, G[g_Wout], H, dslot(DW_dlogits), tyi)) return 1;
    colsum(t, s, t.dlogits, (long long)Vp, TB, V, G[g_bout]);
    {   // embedding: dx = dpre1 . [W_ih1 ; W_is ; W_ig][:, x columns], summed onto the rows that were looked up (ordered, no atomics)
        SegSpec sg[3] = {{t.dpre1, 6 * H, t.wT_ih1 + (size_t)xoff * 4 * H, 4 * H, 4 * H, sP1},
                         {t.dpre1 + 4 * H, 6 * H, t.wT_is + (size_t)xoff * H, H, H, sP1},
                         {t.dpre1 + 5 * H, 6 * H, t.wT_ig + (size_t)xoff * H, H, H, sQ}};
        if (gemm_to(h, t, s, TB, E, sg, 3, t.dx_all, E)) return 1;
        HIPCHK(hipMemsetAsync(G[g_embed], 0, (size_t)V * E * sizeof(float), s));
        hipLaunchKernelGGL(k_embed_grad_rows, dim3(TB), dim3(256), 0, s, t.dx_all, t.word32, TB, E, G[g_embed]);
    }
    HIPCHK(hipEventRecord(h->bucket_ev[2], s));
    // ---- bucket 3: the recurrent LSTM1 / sentinel-gate weights, all LSTM1 / gate biases, s_fc, att_va
    {   // lstm_cell_1.weight_hh, W1_hs and s_fc: 224 whole tiles in one launch
        const GProb a[3] = {GProb{4 * H, H, TBp, t.tY_dpre1, TBp, t.tX_h1prev, TBp, G[g_Whh1], H, tyi ? sY1 : sP1, tyi},
                            GProb{H, H, TBp, t.tY_dpre1 + (size_t)4 * H * TBp, TBp, t.tX_h1prev, TBp, G[g_Whs], H, tyi ? sY1 : sP1, tyi},
                            GProb{D, H, TBp, t.tY_dsent, TBp, t.tX_st, TBp, G[g_Wsfc], H, dslot(DW_step + DY_dsent), tyi}};
        if (gemm_group(h, t, s, a, 3)) return 1;
    }
    // the six LSTM1 / gate bias gradients are column sums of ONE matrix (dpre1, 6H columns): one partial-sum launch, three finishing launches
    // that write both members of a pair (same per-column arithmetic as three separate sums: bit-identical)
    hipLaunchKernelGGL(k_colsum, dim3(cdiv(6 * H, 64), COLSUM_CHUNKS), dim3(256), 0, s, t.dpre1, (long long)6 * H, TB, 6 * H, t.scratch);
    hipLaunchKernelGGL(k_colsum_finish, dim3(cdiv(4 * H, 256)), dim3(256), 0, s, t.scratch, 6 * H, 0, 4 * H, G[g_bih1], G[g_bhh1]);
    hipLaunchKernelGGL(k_colsum_finish, dim3(cdiv(H, 256)), dim3(256), 0, s, t.scratch, 6 * H, 4 * H, H, G[g_bis], G[g_bhs]);
    hipLaunchKernelGGL(k_colsum_finish, dim3(cdiv(H, 256)), dim3(256), 0, s, t.scratch, 6 * H, 5 * H, H, G[g_big], G[g_bhg]);
    colsum(t, s, t.dsent_all, (long long)D, TB, D, G[g_bsfc]);
    // att_va: dP^T (A, NV) x regions^T (D, NV) over the non-padding rows
    if (NV > 0) {
        if (gemm_to1(h, t, s, A, D, NVp, t.tY_dP, NVp, t.tX_reg, NVp, G[g_Wva], D, dslot(DW_dP), tyiP)) return 1;
    } else {
        HIPCHK(hipMemsetAsync(G[g_Wva], 0, (size_t)A * D * sizeof(float), s));
    }
    HIPCHK(hipEventRecord(h->bucket_ev[3], s));
    // ---- bucket 4 (the tail, 3.5 M floats): W1_hg, att_ha, att_sa, att_ga and the three score vectors
    {
        const GProb a[4] = {GProb{H, H, TBp, t.tY_dpre1 + (size_t)5 * H * TBp, TBp, t.tX_h1, TBp, G[g_Whg], H, tyi ? sY1 : sQ, tyi},
                            GProb{A, H, TBp, t.tY_dhA, TBp, t.tX_h1, TBp, G[g_Wha], H, dslot(DW_step + DY_dhA), tyi},
                            GProb{A, H, TBp, t.tY_dsa, TBp, t.tX_st, TBp, G[g_Wsa], H, dslot(DW_step + DY_dsa), tyi},
                            GProb{A, H, TBp, t.tY_dga, TBp, t.tX_gt, TBp, G[g_Wga], H, dslot(DW_step + DY_dga), tyi}};
        if (gemm_group(h, t, s, a, 4)) return 1;
    }
    colsum(t, s, t.dwa_rows, (long long)A, TB, A, G[g_wa]);
    colsum(t, s, t.dws_rows, (long long)A, TB, A, G[g_ws]);
    colsum(t, s, t.dwg_rows, (long long)A, TB, A, G[g_wg]);
    HIPCHK(hipEventRecord(h->bucket_ev[4], s));
    h->buckets_recorded = true;
    LAUNCHCHK();
    return 0;
}

// Completion order of the 28 gradients inside vsr_train_backward: bucket_of[i] (field order of vsr_weights) in
// [0, VSR_GRAD_BUCKETS); bucket b is complete when its event has fired.  Static (does not depend on the shapes).
extern "C" int vsr_train_bucket_map(int32_t* bucket_of, int32_t* n_buckets) {
    if (!bucket_of || !n_buckets) return fail("vsr_train_bucket_map: null argument");
    enum { g_embed, g_Wis, g_bis, g_Whs, g_bhs, g_Wva, g_Wha, g_wa, g_Wsa, g_ws, g_Wih1, g_Whh1, g_bih1, g_bhh1, g_Wih2, g_Whh2, g_bih2,
           g_bhh2, g_Wout, g_bout, g_Wsfc, g_bsfc, g_Wig, g_big, g_Whg, g_bhg, g_Wga, g_wg };
    const int b0[] = {g_Wih1, g_Wis, g_Wig}, b1[] = {g_Wih2, g_Whh2, g_bih2, g_bhh2}, b2[] = {g_Wout, g_bout, g_embed},
              b3[] = {g_Whh1, g_Whs, g_bih1, g_bhh1, g_bis, g_bhs, g_big, g_bhg, g_Wsfc, g_bsfc, g_Wva},
              b4[] = {g_Whg, g_Wha, g_Wsa, g_Wga, g_wa, g_ws, g_wg};
    for (int i = 0; i < 28; ++i) bucket_of[i] = -1;
    for (int i : b0) bucket_of[i] = 0;
    for (int i : b1) bucket_of[i] = 1;
    for (int i : b2) bucket_of[i] = 2;
    for (int i : b3) bucket_of[i] = 3;
    for (int i : b4) bucket_of[i] = 4;
    for (int i = 0; i < 28; ++i)
        if (bucket_of[i] < 0) return fail("vsr_train_bucket_map: gradient %d has no bucket", i);
    *n_buckets = VSR_GRAD_BUCKETS;
    return 0;
}

// Make `stream` wait (on the device; the host does not block) until bucket `bucket` of the last vsr_train_backward is complete.
extern "C" int vsr_train_wait_bucket(vsr_handle* h, int32_t bucket, void* stream) {
    if (!h || !h->tc) return fail("vsr_train_wait_bucket: null handle");
    if (bucket < 0 || bucket >= VSR_GRAD_BUCKETS) return fail("vsr_train_wait_bucket: bucket %d not in [0, %d)", bucket, VSR_GRAD_BUCKETS);
    if (!h->buckets_recorded) return fail("vsr_train_wait_bucket: no backward pass has run on this handle");
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, h->bucket_ev[bucket], 0));
    return 0;
}

// ---------------------------------------------------------------------------------------------- test hook
// copies an internal training buffer (by name) into a caller tensor: lets the parity tests localise a gradient
// mismatch to one stage instead of only seeing the 28 end results.
extern "C" int vsr_debug_copy(vsr_handle* h, const char* name, float* dst, size_t n_floats, void* stream) {
    if (!h || !name || !dst) return fail("vsr_debug_copy: null argument");
    TrainCtx& t = *h->tc;
    struct Ent { const char* n; const float* p; };
    const Ent tab[] = {{"dh2_voc", t.dh2_voc}, {"dpre1", t.dpre1}, {"dpre2", t.dpre2}, {"gates1", t.gates1}, {"gates2", t.gates2},
                       {"c1s", t.c1s}, {"c2s", t.c2s}, {"h1s", t.h1s}, {"h2s", t.h2s}, {"dlogits", t.dlogits}, {"alphas", t.alphas},
                       {"dhA", t.dhA_all}, {"dsent", t.dsent_all}, {"dsa", t.dsa_all}, {"dga", t.dga_all}, {"dP", t.dP},
                       {"atts", t.atts}, {"sents", t.sents}, {"wT_out", t.wT_out}, {"dx_all", t.dx_all}};
    for (const Ent& e : tab)
        if (!strcmp(e.n, name)) {
            HIPCHK(hipMemcpyAsync(dst, e.p, n_floats * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
            return 0;
        }
    return fail("vsr_debug_copy: unknown buffer '%s'", name);
}

// libvsrcap.so - C ABI + per-timestep orchestration of the VSR captioning decoder on gfx950.
// Entry points are declared in include/vsrcap.h; reference behaviour cited there and in kernels.h.
//
// One decoder timestep = 4 grouped fp32-MFMA GEMM launches + 5 small kernels, all on the caller's stream:
//   S1  [h2 | x | h1_old] -> LSTM1 gates (4H) | sentinel gate (H) | shift-gate image part (H)      gemm
//       k_lstm1                                                                                    pointwise
//   S2  h1_new -> [W1_hg | att_ha] ,  s_t -> [s_fc | att_sa]                                       gemm (4 problems)
//       k_attend (shift-gate vector and the S2 slab sums fused in)                                 pointwise / HBM-bound
//   S5  [h1_new | att | h2_old] -> LSTM2 gates (4H) ,  g_t -> att_ga                               gemm (2 problems)
//       k_lstm2
//   S6  h2_new -> vocabulary logits (V)                                                            gemm
//       k_vocab (log-sum-exp + arg-max / top-k / Gumbel sample / full row; the step's gate logits on the side), k_select_*
// (the training forward, train.inc.h, keeps k_gate2 separate and saves what the backward pass needs)
// The image-constant work (pooled descriptor and its projection, att_va(regions), row masks) is hoisted
// into vsr_prepare().  Beams never copy statics: rows index their image (row / beam) and their parent.
#include "../../include/vsrcap.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_f32.h"
#include "gemm_bf16.h"
#include "gemm_x3.h"
#include "gemm_x3s.h"
#include "gemm_h2.h"
#include "gemm_h2a.h"
#include "gemm_b16a.h"
#include "kernels.h"
#include "train_kernels.h"

using namespace vsr;

static thread_local char g_err[512] = "";
static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail("%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define LAUNCHCHK() HIPCHK(hipGetLastError())

struct Ctx {
    int B = 0, R0 = 0, L = 0, R = 0, beam = 1, Mmax = 0;
    int n_img = 0, Rb = 0;       // index-list region format: images and rows per image of the feature bank (0 = dense regions)
    const float* det = nullptr;
    const float* regions = nullptr;   // dense (B, L, R, D) region tensor, or the (n_img, Rb, D) feature bank
    const int* ridx = nullptr;   // (B, L, R) absolute bank row per slot entry (index-list format) or null
    bool rows_are_images = false;   // index-list format without a row -> image map: decoder row b owns bank rows [b Rb, (b + 1) Rb) (training needs this)
    int* ridx_buf = nullptr;
    float* bmask = nullptr;      // row masks of the rows att_va runs over (dense: = rmask)
    float* dmask = nullptr;      // row masks of the detection rows (pooled descriptor)
    float *vbar, *vproj, *vproj2, *P, *rmask;
    int *vlist, *nvalid_dev;     // non-padding region rows (ascending) and their number
    int* bcount;                 // per-256-row counts / offsets of the compaction
    int nvalid = 0;
    bool bounded = false;        // nvalid is the caller's row bound (vsr_set_valid_rows_bound), the real count lives in nvalid_dev[0]
    float* st[2][4];   // h1, c1, h2, c2 double-buffered
    int *slot[2], *word[2], *gate[2], *parent;
    float *s_t, *gpre, *g_t, *hA, *sa, *sent, *att, *zsum, *lg, *top_v;
    float* ga_slabs;             // att_ga(g_t) partial sums: kept apart from `scratch`, the vocabulary GEMM overwrites that first
    int* top_i;
    float *seq[2], *mask[2];
    int *hist_parent, *hist_word, *hist_gate;
    float *hist_lpw, *hist_lpg;
    int *cap32, *forced_w32, *forced_g32;
    int64_t* tmp_i64;
    float* scratch;
    size_t scratch_floats = 0;
    // bf16 GEMM mode: bf16 images of the GEMMs' A operands, written by their producers next to the fp32 values (h1, h2 of both
    // state buffers, s_t, g_t, att); st16_ok[i]: the images of state buffer i match its fp32 content
    uint16_t* st16[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    uint16_t *s_t16 = nullptr, *g_t16 = nullptr, *att16 = nullptr;
    bool st16_ok[2] = {false, false};
    float* pre1 = nullptr;       // LSTM1/gate sums of the NEXT step, produced early (merged with the vocabulary GEMM)
    int pre1_ns = 0, pre1_nblk = 0, pre1_skip5 = 0;
    long long pre1_stride = 0;
};

struct TrainCtx;
static TrainCtx* new_train_ctx();
static void free_train_ctx(TrainCtx*);
static void invalidate_train_ctx(TrainCtx*);
struct SavedForwards;                 // the training forwards whose workspaces are still intact (train.inc.h)
static SavedForwards* new_saved_forwards();
static void free_saved_forwards(SavedForwards*);
static void drop_saved_forwards(SavedForwards*);                                            // all of them (a change of GEMM flavour / weights binding)
static void drop_saved_forwards_in(SavedForwards*, const void* lo, size_t bytes);           // those whose workspaces overlap [lo, lo + bytes)

struct Bf16Range { const float* lo; const float* hi; const uint16_t* b; };   // fp32 matrix [lo, hi) has a bf16 copy at b
struct H2Range { const float* lo; const float* hi; const float* img; int slot; };   // ... an fp16-pair image (gemm_h2.h) at img, scale exponent in slot

// f16x2 flavour: slots of the scale-exponent table (device ints at the head of the image buffer; a twin table of float bounds next to it).
// 0..13: the 14 weight matrices; then the bounds of the A operands a GEMM segment can name (GemmBuilder::seg's a_cls)
enum H2Slot { H2A_NONE = -1, H2A_EMBED = 14, H2A_UNIT = 15, H2A_REGION = 16, H2A_DET = 17, H2A_ATT = 18, H2B_SENT = 19, H2_NSLOT = 32 };

struct vsr_handle {
    TrainCtx* tc = nullptr;
    SavedForwards* saved = nullptr;
    hipEvent_t bucket_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // recorded by vsr_train_backward after each gradient bucket
    bool buckets_recorded = false;
    long long gen_counter = 0;        // generations of the training forwards: strictly increasing per handle
    const char* ws_lo = nullptr;      // the workspace the current vsr_prepare*() carved (h->c points into it)
    size_t ws_bytes = 0;
    // bf16 throughput mode (gemm_bf16.h): off unless vsr_refresh_bf16_weights() has been given a buffer
    bool bf16_on = false;
    // f16x2 flavour (gemm_h2.h): on once vsr_refresh_h2_weights() has been given a buffer, and only together with x3_on (a launch
    // that does not qualify - an operand without an image / a bound, sizes that are not multiples of 8 - takes the f32x3 kernels)
    bool h2_on = false;
    std::vector<H2Range> h2;
    int* h2_exps = nullptr;           // device: H2_NSLOT scale exponents ...
    unsigned* h2_bounds = nullptr;    // ... and the bounds they come from (bit patterns of non-negative floats)
    // streaming kernel: launches of at most h2s_max rows (VSR_H2S_MAX / _SLOTS / _MIN / _NS).  Measured end to end in one run
    // (profiles/r04_e_h2s_routing.txt): 80 - greedy (M = 100, then on the 128 x 128 tile) 660 k tokens/s against 632 k at 128, the 13-image shard
    // (M = 65) 2.61 ms either way; 48 - greedy 668 k, the shard 2.75 ms
    int h2s_max = 80, h2s_slots = 512, h2s_min = 8, h2s_ns = 1;
    int h2_aligned_min = 4;      // shortest k-aligned piece of the f16x2 kernels, in 32-wide k-tiles (VSR_H2_ALIGNED_MIN) ...
    int h2_aligned_min_small = 8;   // ... and in launches whose rows fit ONE m-tile (<= 128 rows: greedy decoding, the per-step GEMMs of training; VSR_H2_ALIGNED_MIN_SMALL).
                                    // Round 6: 8 instead of 4 there - pieces of 4 k-tiles cost more in their flush than in their k loop: XE +2.4 %, greedy +1.1 %
                                    // (profiles/r06_t_*).  For the wide launches 8 was REJECTED by the flip-rate fixture (one caption of 1 024 flipped in the default flavour).
    // the producers of the decoder's A operands (h1, h2, s_t, g_t, the attended vector) write fp16-pair images next to the fp32 values
    // and launches whose A operands all have one take the all-DMA kernel (gemm_h2a.h); VSR_H2_AIMG=0: in-kernel split of fp32 A only
    double aligned_eff_min = 0.75;    // wide launches: k-aligned pieces when they keep at least this share of the CUs busy, stream-K ranges otherwise (VSR_ALIGNED_EFF, percent)
    bool h2_aimg = true;
    // the selection of step t inside the LSTM1 kernel of step t + 1 (kernels.h: k_select_lstm1, k_select_simple_lstm1); VSR_FUSE_SELECT=0: a launch of its own
    int fuse_select = 3;              // bit 0: greedy / sampling / replay (k_select_simple_lstm1), bit 1: beam search (k_select_lstm1)
    bool b16_dma = true;              // bf16 mode: launches whose A operands all have bf16 images take the all-DMA kernel (VSR_B16_DMA=0: register-staged)
    int h2a_max_small = 128;          // launches of at most this many rows (and more than h2s_max) : 128 x 128 tiles of the all-DMA kernel
    std::vector<H2Range> h2t;         // the training pass's transposed operands (vsr_train_forward registers the images of its workspace)
    const H2Range* map_h2(const float* p, bool with_train = true) const {
        for (const H2Range& r : h2)
            if (p >= r.lo && p < r.hi) return &r;
        if (with_train)
            for (const H2Range& r : h2t)
                if (p >= r.lo && p < r.hi) return &r;
        return nullptr;
    }
    // A operands are looked up among the images registered at refresh only (the embedding table): the training workspace's
    // ranges (h2t) describe W operands and may outlive the memory they were registered for
    const H2Range* map_h2_a(const float* p) const { return map_h2(p, false); }
    bool h2t_only = false;            // the running backward pass writes ONLY the images of its transposed operands (train.inc.h: h2b)
    bool is_h2_train_image(const float* p) const {      // p lies in a transposed operand of the training pass that exists ONLY as an fp16-pair image
        if (!h2t_only) return false;
        for (const H2Range& r : h2t)
            if (p >= r.lo && p < r.hi) return true;
        return false;
    }
    int h2_slot_of(const float* p) const { const H2Range* r = map_h2(p); return r ? r->slot : 0; }
    bool x3_on = true;                // launches of >= gemm_x3_min_rows rows: fp32 products through three bf16 terms per operand (gemm_f32x3.h); fp32 operands, no copies.  vsr_set_gemm_mode(h, 0): exact fma chain everywhere
    std::vector<Bf16Range> b16;        // weights (refresh) + the training pass's transposed operands (carve_train)
    size_t b16_weights = 0;            // entries of b16 that belong to the weights
    int gemm_slots_bf16 = 256;         // ONE 16-wave workgroup per CU (108 KB of LDS: two 128+256-row x 64-k bf16 buffers; 147 KB for f32x3)
    const uint16_t* map16(const float* p) const {
        for (const Bf16Range& r : b16)
            if (p >= r.lo && p < r.hi) return r.b + (p - r.lo);
        return nullptr;
    }
    bool is_train_twin(const float* p) const {       // p lies in a transposed operand of the training pass (bf16 image only)
        for (size_t i = b16_weights; i < b16.size(); ++i)
            if (p >= b16[i].lo && p < b16[i].hi) return true;
        return false;
    }
    vsr_dims d;
    vsr_weights w;
    bool bound = false, prepared = false;
    const int* vt_ptr = nullptr;
    const int* vt_ids = nullptr;
    int n_verbs = 0;
    int gemm_slots = 1024;       // resident 64x64 GEMM workgroups to fill: 256 CUs x 4 (36.9 KB LDS each)
    int gemm_slots_small = 768;  // 64x64 tiles (M <= 192): 3 per CU measured best (greedy 473 k vs 461 k tokens/s at 4 per CU)
    int gemm_min_iters = 8;
    int gemm_x3_min_rows = 193;  // f32x3 flavour: launches of at least this many rows take the 128 x 256 tile (VSR_X3_MIN_ROWS)
    int x3_skinny = 1;           // ... launches of r16_max < rows <= 128 the 128 x 128 tile (one m-tile holds every row; VSR_X3_SKINNY=0: exact kernels)
    // k-aligned pieces (gemm_plan_aligned) or stream-K ranges.  VSR_X3_ALIGNED=<wide><skinny> as two digits; wide: 0 never, 1 whenever the
    // tiles fit the CUs, 2 (default) per launch by its efficiency (GemmBuilder::finish) and always from 1024 rows up.  Measured end to
    // end: beam-5 (M = 500) 265.7 k tokens/s with stream-K ranges everywhere against 256.5 k with aligned pieces everywhere; XE step
    // (its wide launches have 2000 rows) 9.52 k against 9.40 k samples/s; greedy (M = 100) 572 k with aligned pieces against 550 k
    int x3_aligned_wide = 2, x3_aligned_skinny = 1;
    int x3_aligned_min = 4;      // shortest k-aligned piece of the f32x3 kernels, in 32-wide k-tiles
    // f32x3 launches of at most x3s_max rows: the weight-streaming kernel (gemm_x3s.h) when its k-aligned plan exists.  Measured over
    // the four step GEMMs (tools/gemm_bench, one 16-column strip per wave, two workgroups per CU): M = 13: 53 us against 65 (rows-16
    // kernel); M = 32: 61 against 76; M = 65: 99 against 107 (128 x 128 tile); M = 100: 123 against 112 - so up to 80 rows.
    // VSR_X3S_MAX=0 turns it off.
    int x3s_max = 80, x3s_slots = 512, x3s_min = 8;
    int gemm_slots_r16 = 256;    // rows-16 kernel: ONE 8-wave workgroup per CU (two waves per SIMD)
    // Problems with at most this many rows take the rows-16 kernel (VSR_GEMM_R16_MAX=0 disables it).  Measured end to end on
    // one MI355X: at M = 100 it is level with the 64x64 kernel inside a GEMM (61.5 vs 60.6 TF/s) but its tiles are cut into
    // 7-8 stream-K pieces instead of 4-6, and the consumers' extra slab reads cost more than its 11 %-instead-of-28 %
    // padding saves (greedy 459 k vs 481 k tokens/s, XE step 6.8 k vs 7.4 k samples/s).  Below 64 rows (a data-parallel
    // shard of 12-13 images and its 65 beam rows, small eval batches) the 64-row tiles are mostly padding and the rows-16 kernel wins
    // (M = 13: 19.5 vs 13.2 TF/s over the four step GEMMs; beam-5 over a 13-image shard, M = 65: 3.48 vs 3.81 ms per call).
    int gemm_r16_max = 40;
    int bf16_p_fp32 = 1;         // bf16 mode: the hoisted att_va(regions) GEMM of vsr_prepare*() stays fp32-equivalent (VSR_BF16_P_FP32=0: bf16 like the rest)
    int bf16_a16 = 1;            // bf16 mode: the decode step's producers write bf16 images of the GEMM A operands (VSR_BF16_A16=0: off)
    int gemm_aligned = 1;        // 128 x 256 kernels: k-aligned pieces (gemm_plan_aligned) when the tiles fit the CUs; VSR_GEMM_ALIGNED=0: stream-K always
    int gemm_aligned_min = 8;    // shortest piece, in 64-wide k-tiles (VSR_GEMM_ALIGNED_MIN)
    const float* xproj = nullptr;     // decode cache: (V, 6H) projection of the embedding table, valid for the bound weights
    long long rows_bound = 0;    // vsr_set_valid_rows_bound: > 0 = the caller's upper bound on the non-padding region rows; vsr_prepare*() then never waits for the host
    int attend_parts = 2, attend_limit = 256;        // workgroups per row of k_attend in launches of <= attend_limit / parts rows (VSR_ATTEND_PARTS, VSR_ATTEND_LIMIT)
    int split_pre1 = 1;          // the h1 part of the next step's LSTM1 sums in the S5 launch, the h2 part with the vocabulary (run_step; VSR_SPLIT_PRE1=0: all of it with the vocabulary, as in rounds 2-5)
    int xcd_groups = 0;          // VSR_XCD_GROUPS=1: k-aligned plans deal whole m-groups of tiles to an XCD (gemm_plan_aligned).  Measured: 2 % less fabric traffic on the wide kernel, 1.3 % SLOWER end to end (profiles/r06_d_xcd_group_dealing_ab.txt): off
    int gemm_tile = 0;           // 0 = by M; VSR_GEMM_TILE=64 | 12864 | 128 forces 64x64 / 128x64 / 128x128
    Ctx c;
    // measurement
    bool profiling = false;
    hipEvent_t ev_count = nullptr;   // prepare(): the row count has reached the host
    int* host_back = nullptr;        // pinned landing place of that read-back (2 ints)
    std::vector<hipEvent_t> ev;      // pool, pairs (start, stop)
    size_t ev_used = 0;
    double prof_flops = 0, prof_bytes = 0;
    int prof_every = 1;              // time every prof_every-th GEMM launch (1 = all of them)
    long long prof_seen = 0;         // GEMM launches since vsr_profile_begin*
};

// ---------------------------------------------------------------------------------------------- workspace
struct Bump {
    char* base;
    size_t off = 0;
    template <typename T>
    T* take(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

static size_t carve(const vsr_handle* h, Ctx& c, char* base) {
    const vsr_dims& d = h->d;
    const size_t B = c.B, H = d.rnn_size, A = d.att_size, D = d.det_feat_size, V = d.vocab_size, T = d.seq_len;
    const size_t M = c.Mmax, rows = B * c.L * c.R;
    const size_t prows = c.Rb > 0 ? (size_t)c.n_img * c.Rb : rows;      // rows the hoisted att_va projection covers
    Bump b{base};
    c.vbar = b.take<float>(B * D);
    c.vproj = b.take<float>(B * 6 * H);
    c.vproj2 = b.take<float>(B * 4 * H);
    c.P = b.take<float>(prows * A);
    c.rmask = b.take<float>(rows);
    if (c.Rb > 0) {
        c.bmask = b.take<float>(prows);
        c.ridx_buf = b.take<int>(rows);
    } else {
        c.bmask = c.rmask;
        c.ridx_buf = nullptr;
    }
    c.dmask = b.take<float>((size_t)(c.Rb > 0 ? c.n_img : B) * c.R0);
    c.vlist = b.take<int>(prows);
    c.bcount = b.take<int>((prows + 255) / 256 + 1);
    c.nvalid_dev = b.take<int>(4);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) c.st[i][j] = b.take<float>(M * H);
    for (int i = 0; i < 2; ++i) {
        c.slot[i] = b.take<int>(M);
        c.word[i] = b.take<int>(M);
        c.gate[i] = b.take<int>(M);
        c.seq[i] = b.take<float>(M);
        c.mask[i] = b.take<float>(M * 2);
    }
    c.parent = b.take<int>(M);
    c.s_t = b.take<float>(M * H);
    c.gpre = b.take<float>(M * H);
    c.g_t = b.take<float>(M * H);
    c.hA = b.take<float>(M * A);
    c.sa = b.take<float>(M * A);
    c.sent = b.take<float>(M * D);
    c.att = b.take<float>(M * D);
    c.zsum = b.take<float>(M);
    c.lg = b.take<float>(M * 2);
    c.ga_slabs = b.take<float>(M * A * 8);
    c.top_v = b.take<float>(M * KMAX);
    c.top_i = b.take<int>(M * KMAX);
    c.hist_parent = b.take<int>(T * M);
    c.hist_word = b.take<int>(T * M);
    c.hist_gate = b.take<int>(T * M);
    c.hist_lpw = b.take<float>(T * M);
    c.hist_lpg = b.take<float>(T * M);
    c.cap32 = b.take<int>(T * M);
    c.forced_w32 = b.take<int>(T * M);
    c.forced_g32 = b.take<int>(T * M);
    c.tmp_i64 = b.take<int64_t>(T * M);
    // split-K slab scratch: the widest stage, 8 slabs
    size_t widest = std::max({6 * H, (H + A) + (D + A), 4 * H + A, V});
    size_t stage = std::max(M * widest, B * 6 * H);
    c.scratch_floats = std::max(stage * 8, std::max(rows, prows) * A * 8);   // att_va slabs of prepare(): over the bank rows when indexed
    c.scratch = b.take<float>(c.scratch_floats);
    c.pre1 = b.take<float>(M * 6 * H * 8);
    b.off = (b.off + 15) & ~size_t(15);
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) c.st16[i][j] = b.take<uint16_t>(2 * ((M * H + 7) & ~size_t(7)));      // (bf16: 2 bytes per element; fp16 pairs: 4)
    c.s_t16 = b.take<uint16_t>(2 * ((M * H + 7) & ~size_t(7)));
    c.g_t16 = b.take<uint16_t>(2 * ((M * H + 7) & ~size_t(7)));
    c.att16 = b.take<uint16_t>(2 * ((M * D + 7) & ~size_t(7)));
    return (b.off + 255) & ~size_t(255);
}

// ---------------------------------------------------------------------------------------------- GEMM launch
struct GemmBuilder {
    GemmArgs a;
    GemmBuilder() { memset(&a, 0, sizeof(a)); }
    GemmProb& prob(int M, int N, float* C, int ldc) {
        GemmProb& p = a.p[a.nprob++];
        p.M = M; p.N = N; p.C = C; p.ldc = ldc; p.nseg = 0;
        return p;
    }
    // a_cls: which bound the A operand obeys (H2Slot; the f16x2 kernels scale A by it); H2A_NONE: the launch cannot take them
    static void seg(GemmProb& p, const float* A, int lda, const int* idx, const float* W, int ldw, int K, const uint16_t* A16 = nullptr, int a_cls = H2A_NONE) {
        if (K <= 0) return;
        GemmSeg& s = p.seg[p.nseg++];
        s.A = A; s.lda = lda; s.a_idx = idx; s.W = W; s.ldw = ldw; s.K = K; s.A16 = A16;
        s.exp_idx = a_cls;               // (finish() turns it into (a slot << 16) | w slot when the launch takes the f16x2 kernels)
    }
    int big = 0;       // 2: 128x128 workgroup tiles, 1: 128x64, 0: 64x64 (32x32x2 MFMA); 16: rows-16 kernel (16x16x4 MFMA), r16_tm tiles
    int r16_tm = 0;
    int x3_tn = 2;       // f32x3 and bf16 kernels: workgroup tile 128 x 256 (2) or 128 x 128 (1)
    int x3s_mt = 0;      // weight-streaming f32x3 kernel (big = 34): 16-row tiles of A
    bool a16_all = false;   // bf16 kernel: every segment's A operand has a bf16 image (GemmSeg::A16)
    bool keep_fp32 = false; // bf16 mode: this launch stays fp32-equivalent (f32x3 kernels): the hoisted att_va(regions) projection, whose
                            // outputs are summed RAW over up to 36 rows into the shift logit (step :187) - bf16 rounding adds up coherently there
    bool a_image_only = false;   // f16x2 flavour: an A operand exists ONLY as an fp16-pair image (GemmSeg::A16; the training pass's transposed gradients): the launch must take the all-DMA kernel
    bool stale_h2 = false;  // f16x2 flavour: a W operand exists only as an fp16-pair image (a transposed operand of the training pass) but the launch does not take an f16x2 kernel
    bool stale_w = false;   // bf16 mode: a W operand exists only as a bf16 image but the launch does not qualify for the bf16 kernel
    // stream-K plan: returns the slab count; the caller then sets every problem's C / slab_stride
    int finish(const vsr_handle* h) {
        int maxM = 0;
        for (int i = 0; i < a.nprob; ++i) maxM = std::max(maxM, a.p[i].M);
        if (h->bf16_on && !(keep_fp32 && h->bf16_p_fp32)) {
            // bf16 mode: every W operand of the launch must have a bf16 copy (and 16-byte-aligned 8-element chunks);
            // a launch that does not qualify runs on the fp32 kernel
            bool ok = true;
            for (int i = 0; i < a.nprob && ok; ++i)
                for (int sg = 0; sg < a.p[i].nseg && ok; ++sg) {
                    const GemmSeg& S = a.p[i].seg[sg];
                    const uint16_t* w16 = h->map16(S.W);
                    ok = w16 && (S.K % 8 == 0) && (S.ldw % 8 == 0) && (S.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(w16) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(S.A) & 15) == 0);
                }
            if (!ok) {
                // the transposing kernels of the training pass write ONLY the bf16 image of such an operand: the fp32 kernel
                // would read a stale buffer.  (Does not happen for sizes the mode accepts: every K / leading dimension is a
                // multiple of 8.)
                for (int i = 0; i < a.nprob; ++i)
                    for (int sg = 0; sg < a.p[i].nseg; ++sg) stale_w = stale_w || h->is_train_twin(a.p[i].seg[sg].W);
            }
            if (ok) {
                a16_all = true;
                for (int i = 0; i < a.nprob; ++i)
                    for (int sg = 0; sg < a.p[i].nseg; ++sg) {
                        GemmSeg& S = a.p[i].seg[sg];
                        S.W = reinterpret_cast<const float*>(h->map16(S.W));
                        a16_all = a16_all && S.A16 && (S.lda % 8 == 0) && ((reinterpret_cast<uintptr_t>(S.A16) & 15) == 0);
                    }
                big = (a16_all && h->b16_dma) ? 38 : 32;        // 38: both operands are images: the all-DMA kernel (gemm_b16a.h)
                // launches whose rows fit one m-tile: 128 x 128 tiles (twice the tiles, half the k pieces per tile), as for f32x3
                x3_tn = (h->x3_skinny && maxM <= 128) ? 1 : 2;
                const int BN = x3_tn == 1 ? 128 : 256;
                if (h->gemm_aligned)
                    if (const int ns = gemm_plan_aligned(a, h->gemm_slots_bf16, x3_tn == 1 ? 2 : h->gemm_aligned_min, 128, BN, B16_BK)) return ns;
                return gemm_plan(a, h->gemm_slots_bf16, 4, 128, BN, B16_BK);
            }
        }
        big = h->gemm_tile == 128 ? 2 : h->gemm_tile == 12864 ? 1 : h->gemm_tile == 64 ? 0 : (maxM >= 1024 ? 2 : maxM > 192 ? 1 : 0);
        if (h->h2_on && h->x3_on && h->gemm_tile == 0) {
            // f16x2 (gemm_h2.h): every W operand has an fp16-pair image (window starts and leading dimensions in whole 8-element
            // groups), every A operand a bound class
            bool ok = true;
            for (int i = 0; i < a.nprob && ok; ++i)
                for (int sg = 0; sg < a.p[i].nseg && ok; ++sg) {
                    const GemmSeg& S = a.p[i].seg[sg];
                    const H2Range* r = h->map_h2(S.W);
                    ok = r && S.exp_idx >= 0 && (S.K % 8 == 0) && (S.ldw % 8 == 0) && (S.lda % 4 == 0) && ((S.W - r->lo) % 8 == 0) &&
                         ((reinterpret_cast<uintptr_t>(S.A) & 15) == 0);
                }
            if (ok) {
                GemmArgs ah = a;
                for (int i = 0; i < ah.nprob; ++i)
                    for (int sg = 0; sg < ah.p[i].nseg; ++sg) {
                        GemmSeg& S = ah.p[i].seg[sg];
                        const H2Range* r = h->map_h2(S.W);
                        S.exp_idx = (S.exp_idx << 16) | r->slot;
                        S.W = r->img + (S.W - r->lo);
                    }
                ah.exps = h->h2_exps;
                // all-DMA kernel (gemm_h2a.h): every A operand has an fp16-pair image too - written by its producer (GemmSeg::A16 in this
                // flavour) or a registered one (the embedding table)
                bool aimg = h->h2_aimg;
                for (int i = 0; i < ah.nprob && aimg; ++i)
                    for (int sg = 0; sg < ah.p[i].nseg && aimg; ++sg) {
                        const GemmSeg& S = ah.p[i].seg[sg];
                        const H2Range* ra = S.A16 ? nullptr : h->map_h2_a(S.A);
                        aimg = (S.lda % 8 == 0) && (S.A16 ? (reinterpret_cast<uintptr_t>(S.A16) & 31) == 0
                                                          : (ra && ra->slot == (S.exp_idx >> 16) && (S.A - ra->lo) % 8 == 0));
                    }
                auto with_a_images = [&](GemmArgs& g) {
                    for (int i = 0; i < g.nprob; ++i)
                        for (int sg = 0; sg < g.p[i].nseg; ++sg) {
                            GemmSeg& S = g.p[i].seg[sg];
                            if (S.A16) S.A = reinterpret_cast<const float*>(S.A16);
                            else { const H2Range* ra = h->map_h2_a(S.A); S.A = ra->img + (S.A - ra->lo); }
                        }
                };
                const int slots = h->gemm_slots_bf16;
                if (maxM <= h->h2s_max && maxM <= 128 && !(a_image_only && aimg)) {
                    GemmArgs as = ah;
                    if (const int ns = gemm_plan_aligned(as, h->h2s_slots, h->h2s_min, 128, h2s_bn(h->h2s_ns), H2_BK)) { a = as; big = 36; x3s_mt = (maxM + 15) / 16; return ns; }
                }
                big = aimg ? 37 : 35;
                a = ah;
                if (aimg) with_a_images(a);
                auto aligned_eff = [&](GemmArgs& g) {
                    int T = 1;
                    for (int i = 0; i < g.nprob; ++i) T = std::max(T, (g.p[i].ktiles + g.p[i].split - 1) / g.p[i].split);
                    return (double)g.total_iters / ((double)slots * T);
                };
                if (maxM <= 128) {
                    x3_tn = 1;
                    if (const int ns = gemm_plan_aligned(a, slots, h->h2_aligned_min_small, 128, 128, H2_BK)) return ns;
                    return gemm_plan(a, slots, 4, 128, 128, H2_BK);
                }
                if (h->x3_aligned_wide != 0) {             // (the planner of the f32x3 wide kernel, below)
                    const bool force = h->x3_aligned_wide == 1 || maxM >= 1024;
                    GemmArgs a22 = a, a21 = a;
                    const int ns22 = gemm_plan_aligned(a22, slots, h->h2_aligned_min, 128, 256, H2_BK);
                    int tiles22 = 0;
                    for (int i = 0; i < a.nprob; ++i) tiles22 += ((a.p[i].M + 127) / 128) * ((a.p[i].N + 255) / 256);
                    if (tiles22 <= 64 && maxM < 1024) {
                        const int ns21 = gemm_plan_aligned(a21, slots, h->h2_aligned_min, 128, 128, H2_BK);
                        if (ns21 && aligned_eff(a21) >= 0.95 && (!ns22 || ns21 < ns22)) { a = a21; x3_tn = 1; return ns21; }
                    }
                    if (ns22 && (force || aligned_eff(a22) >= h->aligned_eff_min)) { a = a22; x3_tn = 2; return ns22; }
                }
                x3_tn = 2;
                return gemm_plan(a, slots, 4, 128, 256, H2_BK);
            }
        }
        // from here on the launch reads fp32 operands: the transposing kernels of an f16x2 backward pass wrote ONLY the images of theirs
        if (h->h2_on && !h->bf16_on)
            for (int i = 0; i < a.nprob; ++i)
                for (int sg = 0; sg < a.p[i].nseg; ++sg) stale_h2 = stale_h2 || h->is_h2_train_image(a.p[i].seg[sg].W);
        if ((h->x3_on || (keep_fp32 && h->bf16_on && h->bf16_p_fp32)) && h->gemm_tile == 0) {
            // f32x3 (gemm_x3.h): 128 x 256 tiles from 193 rows up; 128 x 128 tiles for launches whose rows fit ONE m-tile (greedy
            // decoding, sampling, the per-step GEMMs of the training pass at batch 100, a shard of a strong-scaled decode): the
            // number of tiles is then the number of n-tiles, which 256-wide tiles would have to cut into ~10 k pieces each.
            // Measured over the four step GEMMs (tools/gemm_bench): M = 100: 112 us against 160 us for the exact 64 x 64 kernel;
            // M = 65: 107 against 120 (rows-16) / 154; M = 13: 100 against 64 for the rows-16 kernel, which keeps the shortest launches.
            bool ok = true;
            for (int i = 0; i < a.nprob && ok; ++i)
                for (int sg = 0; sg < a.p[i].nseg && ok; ++sg) {
                    const GemmSeg& S = a.p[i].seg[sg];
                    ok = (S.K % 4 == 0) && (S.ldw % 4 == 0) && (S.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(S.W) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(S.A) & 15) == 0);
                }
            const bool wide = ok && maxM >= h->gemm_x3_min_rows;
            const bool stream = ok && !wide && h->x3_skinny && maxM <= h->x3s_max;
            if (stream) {
                // weight-streaming kernel: 64-column blocks x k-aligned pieces, two workgroups per CU
                GemmArgs as = a;
                if (const int ns = gemm_plan_aligned(as, h->x3s_slots, h->x3s_min, 128, x3s_bn(1), X3_BK)) { a = as; big = 34; x3s_mt = (maxM + 15) / 16; return ns; }
            }
            const bool skinny = ok && !wide && h->x3_skinny && maxM <= 128 && maxM > h->gemm_r16_max;
            if (wide || skinny) {
                big = 33;
                const int slots = h->gemm_slots_bf16;
                // efficiency of a k-aligned plan: work units over (slots x longest piece); 1 = every CU busy for the whole launch
                auto aligned_eff = [&](GemmArgs& g) {
                    int T = 1;
                    for (int i = 0; i < g.nprob; ++i) T = std::max(T, (g.p[i].ktiles + g.p[i].split - 1) / g.p[i].split);
                    return (double)g.total_iters / ((double)slots * T);
                };
                if (skinny) {
                    x3_tn = 1;
                    if (h->x3_aligned_skinny)
                        if (const int ns = gemm_plan_aligned(a, slots, h->x3_aligned_min, 128, 128, X3_BK)) return ns;
                    return gemm_plan(a, slots, 4, 128, 128, X3_BK);
                }
                // Wide launches: stream-K ranges keep every CU busy but cut a tile into 3-5 pieces (slabs every consumer has to add);
                // k-aligned pieces share their k-windows in L2 and write exactly `split` slabs, but leave CUs idle when tiles x split
                // does not fill the chip.  Measured on the beam-5 step shapes (tools/gemm_bench, M = 500): S2 (64 tiles of K = 1000)
                // 42 us / 5 slabs with stream-K ranges, 35 us / 2 slabs with k-aligned halves of 128 x 128 tiles; S5 125 us / 5 slabs
                // vs 124 us / 3 slabs (efficiency 0.76); S1 120 vs 137 us (0.74); the vocabulary GEMM 82 vs 88 us (0.63).
                if (h->x3_aligned_wide != 0) {
                    const bool force = h->x3_aligned_wide == 1 || maxM >= 1024;
                    GemmArgs a22 = a, a21 = a;
                    const int ns22 = gemm_plan_aligned(a22, slots, h->x3_aligned_min, 128, 256, X3_BK);
                    int tiles22 = 0;
                    for (int i = 0; i < a.nprob; ++i) tiles22 += ((a.p[i].M + 127) / 128) * ((a.p[i].N + 255) / 256);
                    if (tiles22 <= 64 && maxM < 1024) {              // a small launch: halves of narrow tiles fill the chip with fewer slabs
                        const int ns21 = gemm_plan_aligned(a21, slots, h->x3_aligned_min, 128, 128, X3_BK);
                        if (ns21 && aligned_eff(a21) >= 0.95 && (!ns22 || ns21 < ns22)) { a = a21; x3_tn = 1; return ns21; }
                    }
                    if (ns22 && (force || aligned_eff(a22) >= h->aligned_eff_min)) { a = a22; x3_tn = 2; return ns22; }
                }
                x3_tn = 2;
                return gemm_plan(a, slots, 4, 128, 256, X3_BK);
            }
        }
        if (h->gemm_tile == 0 && maxM <= h->gemm_r16_max) {
            // short problems: every row of an m-tile in one workgroup, rows in units of 16 (M = 100 -> 112, not 128)
            const int tiles = (maxM + 127) / 128;
            r16_tm = (((maxM + tiles - 1) / tiles) + 15) / 16;
            big = 16;
            return gemm_plan(a, h->gemm_slots_r16, 4, 16 * r16_tm, 128);
        }
        // resident workgroups per CU: 4 at 36.9 KB LDS (64x64), 2 at 55.3 KB (128x64) or 73.7 KB (128x128).
        // 128x128 for M >= 1024 (weight-gradient GEMMs: one tile per workgroup, 130 TF/s at long K);
        // 128x64 is the default for tall problems: as fast as 128x128 in the GEMM itself (91.8 vs 93.7 TF/s) but its
        // tiles are cut into ~3 stream-K pieces instead of ~5, so every consumer kernel reads 40 % fewer slab bytes.
        return gemm_plan(a, big ? h->gemm_slots / 2 : h->gemm_slots_small, h->gemm_min_iters, big ? 128 : 64, big == 2 ? 128 : 64);
    }
    int launch(hipStream_t s, vsr_handle* h);
};

int GemmBuilder::launch(hipStream_t s, vsr_handle* h) {
    if (stale_h2) return fail("f16x2 flavour: a GEMM launch names a transposed operand of the training pass that only exists as an fp16-pair image but cannot take an f16x2 kernel (gemm mode / tile override changed since vsr_train_forward, or K / leading dimensions not multiples of 8)");
    if (stale_w) return fail("bf16 mode: a GEMM launch names a transposed operand that only exists as a bf16 image but cannot take the bf16 kernel (K / leading dimensions must be multiples of 8)");
    if (!h->xcd_groups) a.xcd_chunk = 0;              // VSR_XCD_GROUPS=0: ceil(G / 8) workgroups per XCD whatever the m-groups (A/B)
    dim3 grid(gemm_grid(a)), block((big == 32 || big == 38) ? B16_THREADS : (big == 33 || big == 35 || big == 37) ? X3_THREADS : big == 16 ? 512 : 256);       // (big == 34 / 36: 256 = X3S_THREADS = H2S_THREADS)
    const bool prof = h->profiling && (h->prof_seen++ % h->prof_every) == 0 && h->ev_used + 2 <= h->ev.size();
    if (prof) (void)hipEventRecord(h->ev[h->ev_used], s);
    if (big == 36) {
#define H2S_CASE(MT_) case MT_: if (h->h2s_ns == 2) hipLaunchKernelGGL((gemm_nt_h2s_kernel<MT_, 2>), grid, block, 0, s, a); else hipLaunchKernelGGL((gemm_nt_h2s_kernel<MT_, 1>), grid, block, 0, s, a); break;
        switch (x3s_mt) {
            H2S_CASE(1) H2S_CASE(2) H2S_CASE(3) H2S_CASE(4) H2S_CASE(5) H2S_CASE(6) H2S_CASE(7)
            default: if (h->h2s_ns == 2) hipLaunchKernelGGL((gemm_nt_h2s_kernel<8, 2>), grid, block, 0, s, a); else hipLaunchKernelGGL((gemm_nt_h2s_kernel<8, 1>), grid, block, 0, s, a); break;
        }
#undef H2S_CASE
    } else if (big == 38 && x3_tn == 2) hipLaunchKernelGGL((gemm_nt_b16a_kernel<2, 2>), grid, block, 0, s, a);
    else if (big == 38) hipLaunchKernelGGL((gemm_nt_b16a_kernel<2, 1>), grid, block, 0, s, a);
    else if (big == 37 && x3_tn == 2) hipLaunchKernelGGL((gemm_nt_h2a_kernel<2, 2>), grid, block, 0, s, a);
    else if (big == 37) hipLaunchKernelGGL((gemm_nt_h2a_kernel<2, 1>), grid, block, 0, s, a);       // (a ring of four stages fits this tile and changes nothing: tools/gemm_bench H2_NW=4, profiles/r05_e_*)
    else if (big == 35 && x3_tn == 2) hipLaunchKernelGGL((gemm_nt_h2_kernel<2, 2>), grid, block, 0, s, a);
    else if (big == 35) hipLaunchKernelGGL((gemm_nt_h2_kernel<2, 1>), grid, block, 0, s, a);
    else if (big == 34) {
        switch (x3s_mt) {
            case 1: hipLaunchKernelGGL((gemm_nt_x3s_kernel<1, 1>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((gemm_nt_x3s_kernel<2, 1>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((gemm_nt_x3s_kernel<3, 1>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((gemm_nt_x3s_kernel<4, 1>), grid, block, 0, s, a); break;
            case 5: hipLaunchKernelGGL((gemm_nt_x3s_kernel<5, 1>), grid, block, 0, s, a); break;
            case 6: hipLaunchKernelGGL((gemm_nt_x3s_kernel<6, 1>), grid, block, 0, s, a); break;
            case 7: hipLaunchKernelGGL((gemm_nt_x3s_kernel<7, 1>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((gemm_nt_x3s_kernel<8, 1>), grid, block, 0, s, a); break;
        }
    } else if (big == 33 && x3_tn == 2) hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 2>), grid, block, 0, s, a);
    else if (big == 33) hipLaunchKernelGGL((gemm_nt_x3_kernel<2, 1>), grid, block, 0, s, a);
    else if (big == 32 && a16_all && x3_tn == 1) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<true, 1>), grid, block, 0, s, a);
    else if (big == 32 && x3_tn == 1) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<false, 1>), grid, block, 0, s, a);
    else if (big == 32 && a16_all) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<true, 2>), grid, block, 0, s, a);
    else if (big == 32) hipLaunchKernelGGL((gemm_nt_bf16w_kernel<false, 2>), grid, block, 0, s, a);
    else if (big == 16) {
        switch (r16_tm) {
            case 1: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<1, 2>), grid, block, 0, s, a); break;
            case 2: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<2, 2>), grid, block, 0, s, a); break;
            case 3: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<3, 2>), grid, block, 0, s, a); break;
            case 4: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<4, 2>), grid, block, 0, s, a); break;
            case 5: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<5, 2>), grid, block, 0, s, a); break;
            case 6: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<6, 2>), grid, block, 0, s, a); break;
            case 7: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<7, 2>), grid, block, 0, s, a); break;
            default: hipLaunchKernelGGL((gemm_nt_f32_r16_kernel<8, 2>), grid, block, 0, s, a); break;
        }
    } else if (big == 2) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 2>), grid, block, 0, s, a);
    else if (big == 1) hipLaunchKernelGGL((gemm_nt_f32_kernel<2, 1>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((gemm_nt_f32_kernel<1, 1>), grid, block, 0, s, a);
    if (prof) {
        (void)hipEventRecord(h->ev[h->ev_used + 1], s);
        h->ev_used += 2;
        h->prof_flops += gemm_flops(a);
        h->prof_bytes += gemm_bytes(a, big == 32 ? 2 : 4);
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

constexpr int VOCAB_LDS_MAX = 12288;    // 48 KB of dynamic LDS for the combined row
static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------- API: lifetime
extern "C" int vsr_abi_version(void) { return VSR_ABI_VERSION; }
extern "C" const char* vsr_last_error(void) { return g_err; }

extern "C" int vsr_create(const vsr_dims* dims, vsr_handle** out) {
    if (!dims || !out) return fail("vsr_create: null argument");
    const vsr_dims& d = *dims;
    if (d.seq_len <= 0 || d.vocab_size <= 0) return fail("vsr_create: seq_len and vocab_size must be positive");
    if (d.det_feat_size % 4 || d.input_encoding_size % 4 || d.rnn_size % 4 || d.att_size % 4)
        return fail("vsr_create: det_feat_size, input_encoding_size, rnn_size and att_size must be multiples of 4 "
                    "(16-byte vector loads); got %d %d %d %d", d.det_feat_size, d.input_encoding_size, d.rnn_size, d.att_size);
    if (d.bos_idx < 0 || d.bos_idx >= d.vocab_size) return fail("vsr_create: bos_idx out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("vsr_create: no HIP device");
    vsr_handle* h = new vsr_handle();
    h->d = d;
    h->tc = new_train_ctx();
    h->saved = new_saved_forwards();
    hipDeviceProp_t prop;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) {
        h->gemm_slots = prop.multiProcessorCount * 4;
        h->gemm_slots_small = prop.multiProcessorCount * 3;
        h->gemm_slots_r16 = prop.multiProcessorCount;
        h->gemm_slots_bf16 = prop.multiProcessorCount;
        h->x3s_slots = prop.multiProcessorCount * 2;
        h->h2s_slots = prop.multiProcessorCount * 2;
    }
    if (const char* e = getenv("VSR_X3_MIN_ROWS")) h->gemm_x3_min_rows = atoi(e);
    if (const char* e = getenv("VSR_X3_SKINNY")) h->x3_skinny = atoi(e);
    if (const char* e = getenv("VSR_X3S_MAX")) h->x3s_max = std::min(atoi(e), 128);     // (the streaming kernels hold every row in ONE m-tile)
    if (const char* e = getenv("VSR_H2S_MAX")) h->h2s_max = std::min(atoi(e), 128);
    if (const char* e = getenv("VSR_H2S_SLOTS")) h->h2s_slots = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_H2S_MIN")) h->h2s_min = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_H2S_NS")) h->h2s_ns = atoi(e) == 2 ? 2 : 1;
    if (const char* e = getenv("VSR_H2_ALIGNED_MIN")) h->h2_aligned_min = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_H2_ALIGNED_MIN_SMALL")) h->h2_aligned_min_small = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_H2_AIMG")) h->h2_aimg = atoi(e) != 0;
    if (const char* e = getenv("VSR_FUSE_SELECT")) h->fuse_select = atoi(e);
    if (const char* e = getenv("VSR_B16_DMA")) h->b16_dma = atoi(e) != 0;
    if (const char* e = getenv("VSR_ALIGNED_EFF")) h->aligned_eff_min = atoi(e) / 100.0;
    if (const char* e = getenv("VSR_X3S_SLOTS")) h->x3s_slots = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_X3S_MIN")) h->x3s_min = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_X3_ALIGNED")) { h->x3_aligned_wide = atoi(e) / 10; h->x3_aligned_skinny = atoi(e) % 10; }
    if (const char* e = getenv("VSR_X3_ALIGNED_MIN")) h->x3_aligned_min = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_SLOTS_BF16")) h->gemm_slots_bf16 = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_SLOTS_R16")) h->gemm_slots_r16 = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_R16_MAX")) h->gemm_r16_max = atoi(e);
    if (const char* e = getenv("VSR_GEMM_ALIGNED")) h->gemm_aligned = atoi(e);
    if (const char* e = getenv("VSR_BF16_A16")) h->bf16_a16 = atoi(e);
    if (const char* e = getenv("VSR_BF16_P_FP32")) h->bf16_p_fp32 = atoi(e);
    if (const char* e = getenv("VSR_GEMM_ALIGNED_MIN")) h->gemm_aligned_min = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_SLOTS")) h->gemm_slots = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_SLOTS_SMALL")) h->gemm_slots_small = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_TILE")) h->gemm_tile = atoi(e);
    if (const char* e = getenv("VSR_XCD_GROUPS")) h->xcd_groups = atoi(e);
    if (const char* e = getenv("VSR_SPLIT_PRE1")) h->split_pre1 = atoi(e);
    if (const char* e = getenv("VSR_ATTEND_PARTS")) h->attend_parts = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_ATTEND_LIMIT")) h->attend_limit = std::max(1, atoi(e));
    if (const char* e = getenv("VSR_GEMM_MIN_ITERS")) h->gemm_min_iters = std::max(1, atoi(e));
    *out = h;
    return 0;
}

extern "C" void vsr_destroy(vsr_handle* h) {
    if (!h) return;
    for (hipEvent_t e : h->ev) (void)hipEventDestroy(e);
    if (h->ev_count) (void)hipEventDestroy(h->ev_count);
    if (h->host_back) (void)hipHostFree(h->host_back);
    free_train_ctx(h->tc);
    free_saved_forwards(h->saved);
    for (hipEvent_t e : h->bucket_ev) if (e) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int vsr_profile_begin_sampled(vsr_handle* h, int32_t every);
extern "C" int vsr_profile_begin(vsr_handle* h) { return vsr_profile_begin_sampled(h, 1); }

extern "C" int64_t vsr_profile_seen(const vsr_handle* h) { return h ? (int64_t)h->prof_seen : 0; }
extern "C" double vsr_profile_bytes(const vsr_handle* h) { return h ? h->prof_bytes : 0.0; }

extern "C" int vsr_profile_begin_sampled(vsr_handle* h, int32_t every) {
    if (!h) return fail("vsr_profile_begin: null handle");
    if (every < 1) return fail("vsr_profile_begin_sampled: every must be >= 1");
    h->prof_every = every;
    h->prof_seen = 0;
    const size_t want = 2 * 4096;
    while (h->ev.size() < want) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->ev.push_back(e);
    }
    h->ev_used = 0;
    h->prof_flops = 0;
    h->prof_bytes = 0;
    h->profiling = true;
    return 0;
}

extern "C" int vsr_profile_end(vsr_handle* h, void* stream, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops) {
    if (!h || !h->profiling) return fail("vsr_profile_end: profiling not active");
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    double ms = 0;
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float t = 0;
        HIPCHK(hipEventElapsedTime(&t, h->ev[i], h->ev[i + 1]));
        ms += t;
    }
    if (gemm_ms) *gemm_ms = ms;
    if (gemm_launches) *gemm_launches = (int64_t)(h->ev_used / 2);
    if (gemm_flops) *gemm_flops = h->prof_flops;
    h->profiling = false;
    return 0;
}

extern "C" int vsr_bind_weights(vsr_handle* h, const vsr_weights* w) {
    if (!h || !w) return fail("vsr_bind_weights: null argument");
    const float* const* p = reinterpret_cast<const float* const*>(w);
    for (size_t i = 0; i < sizeof(vsr_weights) / sizeof(float*); ++i)
        if (!p[i]) return fail("vsr_bind_weights: weight pointer %zu is null", i);
    h->w = *w;
    h->bound = true;
    invalidate_train_ctx(h->tc); drop_saved_forwards(h->saved);     // saved forwards were taken with the old tensors
    h->xproj = nullptr;               // a cache built for other weight pointers is void
    if (h->bf16_on) {                 // ... and so are the bf16 copies: back to fp32 until vsr_refresh_bf16_weights is called again
        h->bf16_on = false;
        h->b16.erase(h->b16.begin(), h->b16.begin() + h->b16_weights);
        h->b16_weights = 0;
    }
    if (h->h2_on) {                   // ... and the fp16-pair images (vsr_refresh_h2_weights)
        h->h2_on = false;
        h->h2.clear(); h->h2t.clear();
        h->prepared = false;
    }
    return 0;
}

// ---- decode cache: xproj[v] = [W_ih1 ; W1_is ; W1_ig][:, x columns] . embed[v]  for every vocabulary row.
// Weight-only work hoisted out of the time loop AND out of the call: 6 of the 47.9 M MAC per row-step.
static const int XPROJ_CHUNK = 512;
extern "C" size_t vsr_decode_cache_floats(const vsr_handle* h) {
    if (!h) return 0;
    const size_t V = h->d.vocab_size, H = h->d.rnn_size;
    return V * 6 * H + (size_t)XPROJ_CHUNK * 6 * H * 8 + 64;
}
extern "C" int vsr_build_decode_cache(vsr_handle* h, float* buf, size_t n_floats, void* stream) {
    if (!h || !h->bound) return fail("vsr_build_decode_cache: weights not bound");
    if (!buf) { h->xproj = nullptr; return 0; }
    if (n_floats < vsr_decode_cache_floats(h)) return fail("vsr_build_decode_cache: buffer too small");
    hipStream_t s = (hipStream_t)stream;
    const vsr_dims& d = h->d;
    const vsr_weights& w = h->w;
    const int V = d.vocab_size, H = d.rnn_size, E = d.input_encoding_size, D = d.det_feat_size;
    const int in1 = (d.h2_first_lstm ? H : 0) + D + E, xoff = (d.h2_first_lstm ? H : 0) + D;
    float* slabs = buf + (size_t)V * 6 * H;
    for (int v0 = 0; v0 < V; v0 += XPROJ_CHUNK) {
        const int m = std::min(XPROJ_CHUNK, V - v0);
        GemmBuilder g;
        const float* Wih[3] = {w.lstm1_weight_ih, w.W1_is_weight, w.W1_ig_weight};
        const int Nn[3] = {4 * H, H, H}, off[3] = {0, 4 * H, 5 * H};
        for (int i = 0; i < 3; ++i) {
            GemmProb& p = g.prob(m, Nn[i], slabs + off[i], 6 * H);
            GemmBuilder::seg(p, w.embed_weight + (size_t)v0 * E, E, nullptr, Wih[i] + xoff, in1, E, nullptr, H2A_EMBED);
        }
        const int ns = g.finish(h);
        const long long stride = (long long)m * 6 * H;
        for (int i = 0; i < 3; ++i) g.a.p[i].slab_stride = stride;
        if (g.launch(s, h)) return fail("decode cache gemm launch failed");
        hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)((stride + 255) / 256)), dim3(256), 0, s, slabs, ns, stride, stride, buf + (size_t)v0 * 6 * H);
    }
    LAUNCHCHK();
    h->xproj = buf;
    return 0;
}

// ---- bf16 throughput mode: bf16 copies of the 14 weight matrices the GEMMs multiply by (fp32 stays the master copy)
static const int B16_NW = 14;
static void b16_weight_list(const vsr_dims& d, const vsr_weights& w, const float* (&ptr)[B16_NW], size_t (&n)[B16_NW]) {
    const size_t H = d.rnn_size, A = d.att_size, D = d.det_feat_size, E = d.input_encoding_size, V = d.vocab_size;
    const size_t in1 = (d.h2_first_lstm ? H : 0) + D + E, in2 = H + D + (d.img_second_lstm ? D : 0);
    const float* p[B16_NW] = {w.W1_is_weight, w.W1_hs_weight, w.att_va_weight, w.att_ha_weight, w.att_sa_weight, w.lstm1_weight_ih,
                              w.lstm1_weight_hh, w.lstm2_weight_ih, w.lstm2_weight_hh, w.out_fc_weight, w.s_fc_weight, w.W1_ig_weight,
                              w.W1_hg_weight, w.att_ga_weight};
    const size_t c[B16_NW] = {H * in1, H * H, A * D, A * H, A * H, 4 * H * in1, 4 * H * H, 4 * H * in2, 4 * H * H, V * H, D * H, H * in1, H * H, A * H};
    for (int i = 0; i < B16_NW; ++i) { ptr[i] = p[i]; n[i] = c[i]; }
}
extern "C" size_t vsr_bf16_weight_bytes(const vsr_handle* h) {
    if (!h) return 0;
    const float* p[B16_NW]; size_t n[B16_NW];
    vsr_weights none;                          // only the sizes are needed (the weights may be unbound)
    memset(&none, 0, sizeof(none));
    b16_weight_list(h->d, none, p, n);
    size_t tot = 0;
    for (int i = 0; i < B16_NW; ++i) tot += ((n[i] + 7) & ~size_t(7)) * sizeof(uint16_t);
    return tot + 256;
}
extern "C" int vsr_refresh_bf16_weights(vsr_handle* h, void* buffer, size_t bytes, void* stream) {
    if (!h) return fail("vsr_refresh_bf16_weights: null handle");
    if (!buffer) {                                          // back to the fp32 parity mode
        h->bf16_on = false;
        h->b16.erase(h->b16.begin(), h->b16.begin() + h->b16_weights);     // the weight copies only: the training workspace's twins stay registered
        h->b16_weights = 0;
        h->xproj = nullptr;                                 // a decode cache built in the other precision is void
        invalidate_train_ctx(h->tc); drop_saved_forwards(h->saved);      // a saved forward of the other precision cannot be differentiated in this one
        return 0;
    }
    if (!h->bound) return fail("vsr_refresh_bf16_weights: weights not bound");
    const vsr_dims& d = h->d;
    if (d.det_feat_size % 8 || d.input_encoding_size % 8 || d.rnn_size % 8 || d.att_size % 8)
        return fail("vsr_refresh_bf16_weights: the bf16 mode needs det_feat_size, input_encoding_size, rnn_size and att_size to be "
                    "multiples of 8 (16-byte bf16 chunks); got %d %d %d %d", d.det_feat_size, d.input_encoding_size, d.rnn_size, d.att_size);
    if (bytes < vsr_bf16_weight_bytes(h)) return fail("vsr_refresh_bf16_weights: buffer too small");
    if (reinterpret_cast<uintptr_t>(buffer) & 15) return fail("vsr_refresh_bf16_weights: buffer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float* p[B16_NW]; size_t n[B16_NW];
    b16_weight_list(h->d, h->w, p, n);
    std::vector<Bf16Range> keep(h->b16.begin() + h->b16_weights, h->b16.end());   // training operands registered by carve_train
    h->b16.resize(0);
    uint16_t* out = reinterpret_cast<uint16_t*>(buffer);
    for (int i = 0; i < B16_NW; ++i) {
        hipLaunchKernelGGL(k_f32_to_bf16, dim3(cdiv((long long)n[i], 8 * 256)), dim3(256), 0, s, p[i], out, (long long)n[i]);
        h->b16.push_back(Bf16Range{p[i], p[i] + n[i], out});
        out += (n[i] + 7) & ~size_t(7);
    }
    h->b16_weights = h->b16.size();
    h->b16.insert(h->b16.end(), keep.begin(), keep.end());
    LAUNCHCHK();
    if (!h->bf16_on) {
        h->xproj = nullptr;
        invalidate_train_ctx(h->tc); drop_saved_forwards(h->saved);      // (as above: the GEMM precision of a saved forward and its backward must match)
    }
    h->bf16_on = true;
    return 0;
}

// ---- f16x2 flavour (gemm_h2.h): fp16-pair images of the 14 weight matrices the GEMMs multiply by, their power-of-two scales, and the
// bounds of the A operands that depend on the weights only (embedding rows, the sentinel vector).  fp32 stays the master copy.
static const size_t H2_HEAD = 4096;      // bytes: exponent table | bound table | index scratch | from byte 1024: H2_NDYN dynamic slots (ints H2_DYN0 ..)
constexpr int H2_NDYN = 512;             // 64 blocks of 8: one block per timestep of the backward pass (bounds of its gradient operands), the last two for the whole-pass operands
static_assert(H2_DYN0 * 4 + H2_NDYN * 4 <= 4096, "dynamic slots inside the head");
extern "C" size_t vsr_h2_weight_bytes(const vsr_handle* h) {
    if (!h) return 0;
    const float* p[B16_NW]; size_t n[B16_NW];
    vsr_weights none;
    memset(&none, 0, sizeof(none));
    b16_weight_list(h->d, none, p, n);
    size_t tot = H2_HEAD;
    for (int i = 0; i < B16_NW; ++i) tot += ((n[i] + 7) & ~size_t(7)) * sizeof(float);
    tot += (((size_t)h->d.vocab_size * h->d.input_encoding_size + 7) & ~size_t(7)) * sizeof(float);      // the embedding table (an A operand: gemm_h2a.h)
    return tot + 256;
}
__global__ void k_h2_head_init(int* exps, unsigned* bounds, int* idx) {
    const int i = threadIdx.x;
    if (i < H2_NSLOT) {
        exps[i] = 0;
        bounds[i] = i == H2A_UNIT ? __float_as_uint(1.f) : 0u;
        idx[i] = i; idx[H2_NSLOT + i] = -1; idx[2 * H2_NSLOT + i] = i;       // k_h2_exps: slot i from bound i alone
    }
}
// exponents of the operands vsr_prepare*() measures: region rows (their max); the pooled descriptor (sum of <= R0 detection rows over a
// count >= 1, step :126-128: R0 x the detections' max - loose bounds cost nothing, fp16 subnormals are honoured); the attended vector
// (a convex combination of the sentinel and region rows, step :167-171): max(region max, sentinel bound)
__global__ void k_h2_prepare_exps(const unsigned* __restrict__ bounds, int R0, int* __restrict__ exps) {
    if (threadIdx.x != 0) return;
    const float r = __uint_as_float(bounds[H2A_REGION]), dt = __uint_as_float(bounds[H2A_DET]), sn = __uint_as_float(bounds[H2B_SENT]);
    exps[H2A_REGION] = h2_exp_of(r);
    exps[H2A_DET] = h2_exp_of(dt * (float)R0);
    exps[H2A_ATT] = h2_exp_of(fmaxf(r, sn));
}
extern "C" int vsr_refresh_h2_weights(vsr_handle* h, void* buffer, size_t bytes, void* stream) {
    if (!h) return fail("vsr_refresh_h2_weights: null handle");
    if (!buffer) {                                          // back to f32x3 for every launch
        if (h->h2_on) { h->xproj = nullptr; invalidate_train_ctx(h->tc); drop_saved_forwards(h->saved); h->prepared = false; }
        h->h2_on = false;
        h->h2.clear(); h->h2t.clear();
        return 0;
    }
    if (!h->bound) return fail("vsr_refresh_h2_weights: weights not bound");
    const vsr_dims& d = h->d;
    if (d.det_feat_size % 8 || d.input_encoding_size % 8 || d.rnn_size % 8 || d.att_size % 8)
        return fail("vsr_refresh_h2_weights: the f16x2 flavour needs det_feat_size, input_encoding_size, rnn_size and att_size to be "
                    "multiples of 8 (whole 8-element groups of the fp16-pair images); got %d %d %d %d", d.det_feat_size, d.input_encoding_size, d.rnn_size, d.att_size);
    if (bytes < vsr_h2_weight_bytes(h)) return fail("vsr_refresh_h2_weights: buffer too small");
    if (reinterpret_cast<uintptr_t>(buffer) & 255) return fail("vsr_refresh_h2_weights: buffer must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const float* p[B16_NW]; size_t n[B16_NW];
    b16_weight_list(h->d, h->w, p, n);
    char* base = reinterpret_cast<char*>(buffer);
    int* exps = reinterpret_cast<int*>(base);
    unsigned* bounds = reinterpret_cast<unsigned*>(base + 128);
    int* idx = reinterpret_cast<int*>(base + 256);          // 3 x H2_NSLOT ints of index lists for k_h2_exps
    hipLaunchKernelGGL(k_h2_head_init, dim3(1), dim3(64), 0, s, exps, bounds, idx);
    // bounds of the 14 matrices and of the embedding rows in one launch, the images in another (a training step refreshes them)
    static_assert(B16_NW + 1 <= H2_MT, "multi-tensor table");
    H2Multi mt;
    memset(&mt, 0, sizeof(mt));
    const long long ne = (long long)d.vocab_size * d.input_encoding_size;
    int blocks = 0;
    for (int i = 0; i <= B16_NW; ++i) {
        mt.src[i] = i < B16_NW ? p[i] : h->w.embed_weight;
        mt.n[i] = i < B16_NW ? (long long)n[i] : ne;
        mt.slot[i] = i < B16_NW ? i : (int)H2A_EMBED;
        mt.blk[i] = blocks;
        blocks += (int)std::max<long long>(1, std::min<long long>(256, cdiv(mt.n[i], 8192)));
    }
    mt.blk[B16_NW + 1] = blocks;
    mt.nt = B16_NW + 1;
    hipLaunchKernelGGL(k_absmax_multi, dim3(blocks), dim3(256), 0, s, mt, bounds);
    // |sentinel| = |s_fc s_t + b| <= max_d (sum_j |W_dj| + |b_d|) since |s_t| < 1                                   (step :155)
    hipLaunchKernelGGL(k_row_l1_max, dim3(cdiv(d.det_feat_size, L1MAX_ROWS)), dim3(256), 0, s, h->w.s_fc_weight, h->w.s_fc_bias, d.det_feat_size, d.rnn_size, bounds + H2B_SENT);
    hipLaunchKernelGGL(k_h2_exps, dim3(1), dim3(64), 0, s, bounds, idx, idx + H2_NSLOT, idx + 2 * H2_NSLOT, (int)H2A_REGION, exps);   // slots 0 .. 15
    h->h2.clear();
    float* out = reinterpret_cast<float*>(base + H2_HEAD);
    H2Multi mc;
    memset(&mc, 0, sizeof(mc));
    long long off = 0;
    blocks = 0;
    for (int i = 0; i < B16_NW; ++i) {
        const size_t n8 = (n[i] + 7) & ~size_t(7);
        mc.src[i] = p[i]; mc.n[i] = (long long)n8; mc.dst_off[i] = off; mc.slot[i] = i; mc.blk[i] = blocks;
        blocks += cdiv((long long)n8, 8 * 256);
        h->h2.push_back(H2Range{p[i], p[i] + n[i], out + off, i});
        off += (long long)n8;
    }
    {   // the embedding table: an A operand (gathered rows) of the all-DMA kernel, scaled by its own bound class
        const size_t n8 = ((size_t)ne + 7) & ~size_t(7);
        mc.src[B16_NW] = h->w.embed_weight; mc.n[B16_NW] = (long long)n8; mc.dst_off[B16_NW] = off; mc.slot[B16_NW] = H2A_EMBED; mc.blk[B16_NW] = blocks;
        blocks += cdiv((long long)n8, 8 * 256);
        h->h2.push_back(H2Range{h->w.embed_weight, h->w.embed_weight + ne, out + off, H2A_EMBED});
        off += (long long)n8;
    }
    mc.blk[B16_NW + 1] = blocks;
    mc.nt = B16_NW + 1;
    hipLaunchKernelGGL(k_f32_to_h2_multi, dim3(blocks), dim3(256), 0, s, mc, reinterpret_cast<uint32_t*>(out), exps);
    LAUNCHCHK();
    h->h2_exps = exps; h->h2_bounds = bounds;
    if (!h->h2_on) { h->xproj = nullptr; invalidate_train_ctx(h->tc); drop_saved_forwards(h->saved); }
    h->prepared = false;              // the bounds of the region / detection operands are measured by vsr_prepare*()
    h->h2_on = true;
    return 0;
}

// fp32 GEMM flavour: 0 = exact k-ordered fma chain (v_mfma_f32_32x32x2_f32) for every launch,
// 1 (default since round 3) = "f32x3" for launches of more than 192 rows: each fp32 operand split into three bf16 terms, six bf16
// MFMAs per product, fp32 accumulation (gemm_f32x3.h).  Every reference fixture is checked in both (tests/conftest.py).
extern "C" int vsr_set_gemm_mode(vsr_handle* h, int32_t mode) {
    if (!h) return fail("vsr_set_gemm_mode: null handle");
    if (mode != 0 && mode != 1) return fail("vsr_set_gemm_mode: mode %d not in {0, 1}", mode);
    if (h->x3_on != (mode == 1)) {
        h->xproj = nullptr;                                // the decode cache is rebuilt in the new flavour,
        invalidate_train_ctx(h->tc); drop_saved_forwards(h->saved);     // a saved forward of the other flavour is not differentiated in this one,
        h->prepared = false;                               // and the hoisted projections are redone: call vsr_prepare*() after a switch
    }
    h->x3_on = mode == 1;
    return 0;
}

extern "C" int vsr_set_valid_rows_bound(vsr_handle* h, int64_t max_valid_rows) {
    if (!h) return fail("vsr_set_valid_rows_bound: null handle");
    if (max_valid_rows < 0) return fail("vsr_set_valid_rows_bound: negative bound");
    h->rows_bound = max_valid_rows;
    return 0;
}

extern "C" int vsr_set_verb_table(vsr_handle* h, const int32_t* row_ptr, const int32_t* vocab_ids, int32_t n_verbs) {
    if (!h) return fail("vsr_set_verb_table: null handle");
    h->vt_ptr = row_ptr;
    h->vt_ids = vocab_ids;
    h->n_verbs = row_ptr ? n_verbs : 0;
    return 0;
}

extern "C" size_t vsr_workspace_bytes(const vsr_handle* h, int32_t B, int32_t R0, int32_t L, int32_t R, int32_t beam) {
    if (!h || B <= 0 || L <= 0 || R <= 0 || beam <= 0) return 0;
    Ctx c;
    c.B = B; c.R0 = R0; c.L = L; c.R = R; c.beam = beam; c.Mmax = B * beam;
    return carve(h, c, nullptr);
}

extern "C" size_t vsr_workspace_bytes_indexed(const vsr_handle* h, int32_t B, int32_t R0, int32_t n_img, int32_t Rb, int32_t L,
                                              int32_t R, int32_t beam) {
    if (!h || B <= 0 || L <= 0 || R <= 0 || beam <= 0 || n_img <= 0 || Rb <= 0) return 0;
    Ctx c;
    c.B = B; c.R0 = R0; c.L = L; c.R = R; c.beam = beam; c.Mmax = B * beam; c.n_img = n_img; c.Rb = Rb;
    return carve(h, c, nullptr);
}

// ---------------------------------------------------------------------------------------------- prepare
// Both region formats.  Dense: regions = (B, L, R, D).  Index lists: regions = feature bank (n_img, Rb, D),
// slot_idx = (B, L, R) rows of image row_img[b]'s bank (-1 = padding), det = (n_img, R0, D).
static int prepare_impl(vsr_handle* h, const float* det, int B, int R0, const float* regions, int n_img, int Rb,
                        const int* row_img, const int* slot_idx, int L, int R, int beam, void* workspace,
                        size_t workspace_bytes, void* stream, const char* who) {
    if (!h || !h->bound) return fail("%s: weights not bound", who);
    if (!det || !regions || !workspace) return fail("%s: null tensor", who);
    if (B <= 0 || R0 <= 0 || L <= 0 || R <= 0) return fail("%s: empty batch / regions", who);
    if (beam < 1 || beam > VSR_MAX_BEAM) return fail("%s: beam size %d not in [1, %d]", who, beam, VSR_MAX_BEAM);
    if (reinterpret_cast<uintptr_t>(workspace) & 15) return fail("%s: workspace must be 16-byte aligned", who);
    const bool indexed = slot_idx != nullptr;
    if (indexed && (n_img <= 0 || Rb <= 0)) return fail("%s: empty feature bank", who);
    hipStream_t s = (hipStream_t)stream;
    const vsr_dims& d = h->d;
    const vsr_weights& w = h->w;
    const int H = d.rnn_size, A = d.att_size, D = d.det_feat_size, E = d.input_encoding_size;
    Ctx& c = h->c;
    c.B = B; c.R0 = R0; c.L = L; c.R = R; c.beam = beam; c.Mmax = B * beam;
    c.n_img = indexed ? n_img : 0; c.Rb = indexed ? Rb : 0;
    c.det = det; c.regions = regions;
    const size_t need = carve(h, c, reinterpret_cast<char*>(workspace));
    if (need > workspace_bytes) return fail("%s: workspace too small (%zu < %zu bytes)", who, workspace_bytes, need);
    c.ridx = c.ridx_buf;
    c.rows_are_images = indexed && row_img == nullptr && n_img == B;
    h->prepared = false;
    // h->c is rewritten: the CURRENT saved forward (if any) goes with it unless it lives in another workspace - forwards saved in
    // workspaces this call does not touch stay differentiable (vsr_train_select)
    invalidate_train_ctx(h->tc);
    drop_saved_forwards_in(h->saved, workspace, need);
    h->ws_lo = reinterpret_cast<const char*>(workspace); h->ws_bytes = need;

    // region-row masks, then the list of NON-PADDING rows att_va has to run over (att_va(0) = 0).  Their count has to
    // reach the host to size that launch: the one place where this library waits, and it waits for an EVENT behind the
    // copy, with the pooled descriptor and its projections queued after it, so the GPU has work while the host wakes up.
    // (The index-list format reports its out-of-range count through the same read-back.)
    const long long rows = (long long)B * L * R;                         // slot entries
    const long long prows = indexed ? (long long)n_img * Rb : rows;      // rows att_va runs over
    // f16x2 flavour: the bounds of the region / detection operands are measured in the passes that read them anyway (per-block maxima
    // into the idle GEMM scratch, folded by one block); exponents of the region rows, the pooled descriptor and the attended vector
    const bool h2b = h->h2_on;
    float* bm_regions = h2b ? c.scratch : nullptr;
    float* bm_det = h2b ? c.scratch + cdiv(prows, 4) : nullptr;
    if (h2b && (size_t)(cdiv(prows, 4) + cdiv((long long)(indexed ? n_img : B) * R0, 4)) > c.scratch_floats) return fail("%s: scratch too small for the operand bounds", who);
    hipLaunchKernelGGL(k_rowmask, dim3(cdiv(prows, 4)), dim3(256), 0, s, regions, prows, D, c.bmask, bm_regions);
    HIPCHK(hipMemsetAsync(c.nvalid_dev, 0, 4 * sizeof(int), s));      // [0] row count, [1] bad slot indices, [2] bad word / slot / verb ids
    if (indexed) {
        hipLaunchKernelGGL(k_index_rows, dim3(cdiv(rows, 256)), dim3(256), 0, s, slot_idx, row_img, c.bmask, B, L * R, Rb, n_img,
                           c.ridx_buf, c.rmask, c.nvalid_dev + 1);
    }
    {
        const int nb = (int)cdiv(prows, 256);
        hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(256), 0, s, c.bmask, (int)prows, c.bcount);
        hipLaunchKernelGGL(k_compact_scan, dim3(1), dim3(1024), 0, s, c.bcount, nb, c.nvalid_dev);
        hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(256), 0, s, c.bmask, (int)prows, c.bcount, c.vlist);
    }
    LAUNCHCHK();
    // A caller that knows an upper bound on the non-padding rows (vsr_set_valid_rows_bound: the eval script builds its region tensors on
    // the host and has the count for free) gets NO read-back: the list is padded to the bound on the device, the projection is sized by it.
    const long long bound = h->rows_bound > 0 ? std::min(h->rows_bound, prows) : 0;
    int* back = nullptr;
    if (bound > 0) {
        hipLaunchKernelGGL(k_pad_row_list, dim3(cdiv(bound, 256)), dim3(256), 0, s, c.vlist, c.nvalid_dev, (int)bound);
        LAUNCHCHK();
    } else {
        if (!h->host_back) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h->host_back), 2 * sizeof(int), hipHostMallocDefault));
        back = h->host_back;                                             // pinned: the copy does not block this thread
        HIPCHK(hipMemcpyAsync(back, c.nvalid_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        if (!h->ev_count) HIPCHK(hipEventCreateWithFlags(&h->ev_count, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->ev_count, s));
    }

    // pooled descriptor
    const long long drows = (long long)(indexed ? n_img : B) * R0;
    hipLaunchKernelGGL(k_rowmask, dim3(cdiv(drows, 4)), dim3(256), 0, s, det, drows, D, c.dmask, bm_det);
    if (h2b) {
        hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(1024), 0, s, bm_regions, (long long)cdiv(prows, 4), h->h2_bounds + H2A_REGION);
        hipLaunchKernelGGL(k_max_reduce, dim3(1), dim3(1024), 0, s, bm_det, (long long)cdiv(drows, 4), h->h2_bounds + H2A_DET);
        hipLaunchKernelGGL(k_h2_prepare_exps, dim3(1), dim3(64), 0, s, h->h2_bounds, R0, h->h2_exps);
    }
    hipLaunchKernelGGL(k_pool, dim3(B, cdiv(D, 1024)), dim3(256), 0, s, det, indexed ? row_img : nullptr, c.dmask, R0, D, c.vbar);
    LAUNCHCHK();

    // hoisted vbar projections: columns [voff, voff + D) of the LSTM1 / gate input weights
    const int in1 = (d.h2_first_lstm ? H : 0) + D + E;
    const int voff = d.h2_first_lstm ? H : 0;
    {
        GemmBuilder g;
        GemmProb& p0 = g.prob(B, 4 * H, c.scratch, 6 * H);
        GemmBuilder::seg(p0, c.vbar, D, nullptr, w.lstm1_weight_ih + voff, in1, D, nullptr, H2A_DET);
        GemmProb& p1 = g.prob(B, H, c.scratch + 4 * H, 6 * H);
        GemmBuilder::seg(p1, c.vbar, D, nullptr, w.W1_is_weight + voff, in1, D, nullptr, H2A_DET);
        GemmProb& p2 = g.prob(B, H, c.scratch + 5 * H, 6 * H);
        GemmBuilder::seg(p2, c.vbar, D, nullptr, w.W1_ig_weight + voff, in1, D, nullptr, H2A_DET);
        const int ns = g.finish(h);
        const long long stride = (long long)B * 6 * H;
        for (int i = 0; i < 3; ++i) g.a.p[i].slab_stride = stride;
        if (g.launch(s, h)) return fail("vproj gemm launch failed");
        const long long n = (long long)B * 6 * H;
        hipLaunchKernelGGL(k_vproj_finish, dim3(cdiv(n, 256)), dim3(256), 0, s, c.scratch, ns, stride, B, H,
                           w.lstm1_bias_ih, w.lstm1_bias_hh, w.W1_is_bias, w.W1_hs_bias, w.W1_ig_bias, w.W1_hg_bias, c.vproj);
        LAUNCHCHK();
    }
    if (d.img_second_lstm) {
        const int in2 = H + 2 * D;
        GemmBuilder g;
        GemmProb& p0 = g.prob(B, 4 * H, c.scratch, 4 * H);
        GemmBuilder::seg(p0, c.vbar, D, nullptr, w.lstm2_weight_ih + H + D, in2, D, nullptr, H2A_DET);
        const int ns = g.finish(h);
        const long long stride = (long long)B * 4 * H;
        g.a.p[0].slab_stride = stride;
        if (g.launch(s, h)) return fail("vproj2 gemm launch failed");
        hipLaunchKernelGGL(k_slab_reduce, dim3(cdiv(stride, 256)), dim3(256), 0, s, c.scratch, ns, stride, stride, c.vproj2);
        LAUNCHCHK();
    }
    c.bounded = bound > 0;
    if (bound > 0) {
        c.nvalid = (int)bound;         // (bad slot indices of the index-list format join the bad-id count: vsr_bad_ids)
        // a bound that is too small: the rows vlist[bound .. n) get no projection.  Their P rows are ZEROED (att_va = 0, a defined result)
        // instead of keeping whatever the workspace held, and their number is in vsr_bad_ids()'s count
        if (prows > bound)
            hipLaunchKernelGGL(k_zero_rows_beyond, dim3((unsigned)std::min<long long>(prows - bound, 2048)), dim3(128), 0, s, c.vlist, c.nvalid_dev, (int)bound, A, c.P);
    } else {
        HIPCHK(hipEventSynchronize(h->ev_count));
        c.nvalid = back[0];
        if (indexed && back[1] != 0) return fail("%s: %d slot entries index outside the feature bank [-1, %d) or name an image outside [0, %d)", who, back[1], Rb, n_img);
    }
    if (c.nvalid > 0) {
        GemmBuilder g;
        g.keep_fp32 = true;
        GemmProb& p0 = g.prob(c.nvalid, A, c.scratch, A);
        GemmBuilder::seg(p0, regions, D, c.vlist, w.att_va_weight, D, D, nullptr, H2A_REGION);
        const int ns = g.finish(h);
        const long long stride = (long long)c.nvalid * A;
        if ((size_t)stride * ns > c.scratch_floats)
            return fail("%s: att_va slabs (%lld x %d floats) exceed the workspace scratch (%zu)", who, stride, ns, c.scratch_floats);
        g.a.p[0].slab_stride = stride;
        if (g.launch(s, h)) return fail("att_va gemm launch failed");
        hipLaunchKernelGGL(k_slab_reduce_scatter, dim3(cdiv(stride, 256)), dim3(256), 0, s, c.scratch, ns, stride, c.nvalid, A, c.vlist, c.P,
                           bound > 0 ? c.nvalid_dev : (const int*)nullptr);
        LAUNCHCHK();
    }
    h->prepared = true;
    return 0;
}

extern "C" int vsr_prepare(vsr_handle* h, const float* det, int32_t B, int32_t R0, const float* regions, int32_t L,
                           int32_t R, int32_t beam, void* workspace, size_t workspace_bytes, void* stream) {
    return prepare_impl(h, det, B, R0, regions, 0, 0, nullptr, nullptr, L, R, beam, workspace, workspace_bytes, stream, "vsr_prepare");
}

extern "C" int vsr_prepare_indexed(vsr_handle* h, const float* det, int32_t n_img, int32_t R0, const float* bank, int32_t Rb,
                                   const int32_t* row_img, int32_t B, const int32_t* slot_idx, int32_t L, int32_t R,
                                   int32_t beam, void* workspace, size_t workspace_bytes, void* stream) {
    if (!slot_idx) return fail("vsr_prepare_indexed: null slot index list");
    return prepare_impl(h, det, B, R0, bank, n_img, Rb, row_img, slot_idx, L, R, beam, workspace, workspace_bytes, stream,
                        "vsr_prepare_indexed");
}

extern "C" int vsr_row_mask(const float* rows, int64_t n_rows, int32_t D, float* mask, void* stream) {
    if (!rows || !mask || n_rows <= 0 || D <= 0 || (D & 3)) return fail("vsr_row_mask: bad arguments");
    hipLaunchKernelGGL(k_rowmask, dim3(cdiv(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, rows, (long long)n_rows, D, mask);
    LAUNCHCHK();
    return 0;
}

extern "C" int vsr_reorder_slots(const int32_t* slot_idx, const int32_t* rank, const float* verbs, const float* bank_mask,
                                 const int32_t* row_img, int32_t N, int32_t L, int32_t R, int32_t Rb, int32_t* slot_out,
                                 float* verbs_out, void* stream) {
    if (!slot_idx || !rank || !slot_out || N <= 0 || L <= 0 || R <= 0 || Rb <= 0) return fail("vsr_reorder_slots: bad arguments");
    if (verbs && !verbs_out) return fail("vsr_reorder_slots: verbs given without an output");
    hipLaunchKernelGGL(k_reorder_slots, dim3(N), dim3(64), (size_t)L * sizeof(int), (hipStream_t)stream, slot_idx, rank, verbs,
                       bank_mask, row_img, L, R, Rb, slot_out, verbs_out);
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------------------------------------- one timestep
struct StepIO {
    int t, M, rpi, cur;
    const int* parent;       // state-row gather (beam parents) or null
    const int* word_prev;    // (M) int32
    const int* slot;         // (M) int32 or null -> fixed_slot
    int fixed_slot;
    int vmode, K;
    float* full_out; long long full_stride;
    const int* forced;
    uint64_t seed;
    const float* verbs; int gt;
    float* lg_out; long long lg_stride;
    float* alpha_out;
    bool s1_from_prev = false;   // this step's LSTM1 sums are already in c.pre1 (computed over the parent rows)
    // the selection of the PREVIOUS step, still to be made: it rides in this step's LSTM1 kernel (k_select_lstm1 / k_select_simple_lstm1;
    // only with s1_from_prev).  sel_images: images of the beam selection (its parents are sel_beam->cb rows per image).
    const SelBeamArgs* sel_beam = nullptr;
    const SelSimpleArgs* sel_simple = nullptr;
    bool s1_for_next = false;    // compute the next step's LSTM1 sums together with this step's vocabulary GEMM
};

static int run_step(vsr_handle* h, const StepIO& io, hipStream_t s) {
    const vsr_dims& d = h->d;
    const vsr_weights& w = h->w;
    Ctx& c = h->c;
    const int H = d.rnn_size, A = d.att_size, D = d.det_feat_size, E = d.input_encoding_size, V = d.vocab_size;
    const int M = io.M;
    const int in1 = (d.h2_first_lstm ? H : 0) + D + E;
    const int xoff = (d.h2_first_lstm ? H : 0) + D;
    const int in2 = H + D + (d.img_second_lstm ? D : 0);
    float* const* so = c.st[io.cur];          // old state
    float* const* sn = c.st[io.cur ^ 1];      // new state
    float *h1o = so[0], *c1o = so[1], *h2o = so[2], *c2o = so[3];
    float *h1n = sn[0], *c1n = sn[1], *h2n = sn[2], *c2n = sn[3];
    // bf16 images of the A operands (bf16 mode): the producers below write them, the GEMM segments name them
    // ... or fp16-pair images (f16x2 flavour, gemm_h2a.h): isc = the scale of the unit-bounded ones (2^15), the attended vector's from the table
    const bool sh2 = h->h2_on && !h->bf16_on && h->x3_on && h->h2_aimg;
    const bool sh = (h->bf16_on && h->bf16_a16) || sh2;
    const float isc = sh2 ? 32768.f : 0.f;
    const int* att_exp = sh2 ? h->h2_exps + H2A_ATT : nullptr;
    const bool sh_old = sh && c.st16_ok[io.cur];
    uint16_t *h1n16 = sh ? c.st16[io.cur ^ 1][0] : nullptr, *h2n16 = sh ? c.st16[io.cur ^ 1][1] : nullptr;
    const uint16_t *h1o16 = sh_old ? c.st16[io.cur][0] : nullptr, *h2o16 = sh_old ? c.st16[io.cur][1] : nullptr;
    uint16_t *s_t16 = sh ? c.s_t16 : nullptr, *g_t16 = sh ? c.g_t16 : nullptr, *att16 = sh ? c.att16 : nullptr;

    // ---- S1
    if (io.s1_from_prev && io.sel_beam) {
        const SelBeamArgs& sb = *io.sel_beam;
        const int nslice = cdiv(H, SL_UB);
#define SELL_LAUNCH(KK) hipLaunchKernelGGL((k_select_lstm1<KK>), dim3(sb.B * nslice), dim3((KK + 1) * 64), 0, s, sb, c.pre1, c.pre1_ns, c.pre1_stride, c.vproj, \
                                           c1o, H, nslice, h1n, c1n, c.s_t, c.gpre, h->xproj, c.pre1_nblk, h1n16, s_t16, isc, c.pre1_skip5); break;
        switch (sb.beam) {
            case 1: SELL_LAUNCH(1) case 2: SELL_LAUNCH(2) case 3: SELL_LAUNCH(3) case 4: SELL_LAUNCH(4)
            case 5: SELL_LAUNCH(5) case 6: SELL_LAUNCH(6) case 7: SELL_LAUNCH(7) default: SELL_LAUNCH(8)
        }
#undef SELL_LAUNCH
    } else if (io.s1_from_prev && io.sel_simple) {
        hipLaunchKernelGGL(k_select_simple_lstm1, dim3(cdiv((long long)M * H, 256)), dim3(256), 0, s, *io.sel_simple, c.pre1, c.pre1_ns, c.pre1_stride,
                           c.vproj, c1o, M, H, h1n, c1n, c.s_t, c.gpre, h->xproj, c.pre1_nblk, h1n16, s_t16, isc, c.pre1_skip5);
    } else if (io.s1_from_prev) {
        hipLaunchKernelGGL(k_lstm1, dim3(cdiv((long long)M * H, 256)), dim3(256), 0, s, c.pre1, c.pre1_ns, c.pre1_stride, c.vproj, io.rpi,
                           io.parent, c1o, M, H, h1n, c1n, c.s_t, c.gpre, h->xproj, io.word_prev, c.pre1_nblk, 1, h1n16, s_t16, isc, c.pre1_skip5);
    } else {
        const bool xc = h->xproj != nullptr;            // embedding part comes from the decode cache
        GemmBuilder g;
        const float* Wih[3] = {w.lstm1_weight_ih, w.W1_is_weight, w.W1_ig_weight};
        const float* Whh[3] = {w.lstm1_weight_hh, w.W1_hs_weight, nullptr};
        const int Nn[3] = {4 * H, H, H};
        const int off[3] = {0, 4 * H, 5 * H};
        int nblk = 0;
        for (int i = 0; i < 3; ++i) {
            const bool has_h2 = d.h2_first_lstm && io.t > 0, has_x = !xc, has_h1 = Whh[i] && io.t > 0;
            if (!has_h2 && !has_x && !has_h1) continue;
            GemmProb& p = g.prob(M, Nn[i], c.scratch + off[i], 6 * H);
            if (has_h2) GemmBuilder::seg(p, h2o, H, io.parent, Wih[i], in1, H, h2o16, H2A_UNIT);
            if (has_x) GemmBuilder::seg(p, w.embed_weight, E, io.word_prev, Wih[i] + xoff, in1, E, nullptr, H2A_EMBED);
            if (has_h1) GemmBuilder::seg(p, h1o, H, io.parent, Whh[i], H, H, h1o16, H2A_UNIT);
            nblk = i == 0 ? 4 : i == 1 ? 5 : 6;
        }
        int ns = 0;
        const long long stride = (long long)M * 6 * H;
        if (g.a.nprob > 0) {
            ns = g.finish(h);
            for (int i = 0; i < g.a.nprob; ++i) g.a.p[i].slab_stride = stride;
            if (g.launch(s, h)) return fail("S1 gemm launch failed");
        }
        hipLaunchKernelGGL(k_lstm1, dim3(cdiv((long long)M * H, 256)), dim3(256), 0, s, c.scratch, ns, stride, c.vproj, io.rpi,
                           io.parent, c1o, M, H, h1n, c1n, c.s_t, c.gpre, h->xproj, io.word_prev, nblk, 0, h1n16, s_t16, isc);
    }
    // ---- S2
    {
        GemmBuilder g;
        float* c2a = c.scratch;
        float* c2b_base;
        GemmProb& p0 = g.prob(M, H, c2a, H + A);
        GemmBuilder::seg(p0, h1n, H, nullptr, w.W1_hg_weight, H, H, h1n16, H2A_UNIT);
        GemmProb& p1 = g.prob(M, A, c2a + H, H + A);
        GemmBuilder::seg(p1, h1n, H, nullptr, w.att_ha_weight, H, H, h1n16, H2A_UNIT);
        GemmProb& p2 = g.prob(M, D, nullptr, D + A);
        GemmBuilder::seg(p2, c.s_t, H, nullptr, w.s_fc_weight, H, H, s_t16, H2A_UNIT);
        GemmProb& p3 = g.prob(M, A, nullptr, D + A);
        GemmBuilder::seg(p3, c.s_t, H, nullptr, w.att_sa_weight, H, H, s_t16, H2A_UNIT);
        const int ns = g.finish(h);
        const long long stride_a = (long long)M * (H + A), stride_b = (long long)M * (D + A);
        c2b_base = c2a + stride_a * ns;
        g.a.p[0].slab_stride = g.a.p[1].slab_stride = stride_a;
        g.a.p[2].C = c2b_base; g.a.p[3].C = c2b_base + D;
        g.a.p[2].slab_stride = g.a.p[3].slab_stride = stride_b;
        if (g.launch(s, h)) return fail("S2 gemm launch failed");
        // k_gate2's work (g_t, hA, s_a, sentinel from the S2 slabs) is done by the attention kernel's row blocks themselves
        const Gate2Args g2{c2a, c2b_base, ns, stride_a, stride_b, c.gpre, c1n, w.s_fc_bias, H, c.g_t, c.hA, g_t16, isc};
        const size_t smem = (size_t)(2 * A + D + c.R + 1 + 8 + c.R) * sizeof(float);
        // launches of <= 128 rows: two workgroups per row (each forms half of the attended vector's columns): every row's ~370 KB then come in
        // through two CUs' ingest instead of one while the other half of the chip idles (k_attend, nparts; VSR_ATTEND_PARTS=1: off)
        const int np = (h->attend_parts > 1 && M * h->attend_parts <= h->attend_limit && D % (4 * h->attend_parts) == 0) ? h->attend_parts : 1;
        if (D >= 2048) hipLaunchKernelGGL(k_attend<512>, dim3(cdiv(M * np, 8) * 8), dim3(512), smem, s, g2, c.hA, c.sa, c.sent, c.P, c.regions, c.rmask, c.ridx, io.slot,
                           io.fixed_slot, io.rpi, M, c.L, c.R, A, D, w.att_a_weight, w.att_s_weight, c.att, c.zsum, io.alpha_out, att16, att_exp, np);
        else hipLaunchKernelGGL(k_attend<256>, dim3(cdiv(M * np, 8) * 8), dim3(256), smem, s, g2, c.hA, c.sa, c.sent, c.P, c.regions, c.rmask, c.ridx, io.slot,
                           io.fixed_slot, io.rpi, M, c.L, c.R, A, D, w.att_a_weight, w.att_s_weight, c.att, c.zsum, io.alpha_out, att16, att_exp, np);
    }
    // ---- S5
    GateLogitArgs gate_args;
    {
        GemmBuilder g;
        GemmProb& p0 = g.prob(M, 4 * H, c.scratch, 4 * H);
        GemmBuilder::seg(p0, h1n, H, nullptr, w.lstm2_weight_ih, in2, H, h1n16, H2A_UNIT);
        GemmBuilder::seg(p0, c.att, D, nullptr, w.lstm2_weight_ih + H, in2, D, att16, H2A_ATT);
        if (io.t > 0) GemmBuilder::seg(p0, h2o, H, io.parent, w.lstm2_weight_hh, H, H, h2o16, H2A_UNIT);
        GemmProb& p1 = g.prob(M, A, nullptr, A);
        GemmBuilder::seg(p1, c.g_t, H, nullptr, w.att_ga_weight, H, H, g_t16, H2A_UNIT);
        // Round 6 (split_pre1): the h1 part of the NEXT step's LSTM1 / sentinel-gate sums (h1_new . [W_hh1 ; W1_hs]) rides HERE instead
        // of in the vocabulary launch: S5's k-aligned plan left 56 of 256 CUs idle (64 LSTM2 tiles x 3 pieces + 8 att_ga tiles), and
        // without these K = H tiles the vocabulary launch is UNIFORM (every tile K = H: one k-aligned piece per tile, no slabs at all).
        // Measured on the beam-5 shapes (tools/gemm_bench GEMM_REPACK=1, profiles/r06_e_*): 148-153 -> 133 us for the two launches.
        bool split_pre1 = io.s1_for_next && h->split_pre1 && d.h2_first_lstm;
        int ns_h1 = 0;
        auto add_h1 = [&](GemmBuilder& gb) {
            GemmProb& p2 = gb.prob(M, 4 * H, c.pre1, 6 * H);
            GemmBuilder::seg(p2, h1n, H, nullptr, w.lstm1_weight_hh, H, H, h1n16, H2A_UNIT);
            GemmProb& p3 = gb.prob(M, H, c.pre1 + 4 * H, 6 * H);
            GemmBuilder::seg(p3, h1n, H, nullptr, w.W1_hs_weight, H, H, h1n16, H2A_UNIT);
        };
        if (split_pre1) {
            // both parts land in the 8 slabs of pre1: plan the two launches first and keep the round-5 composition when they would not fit
            // (the exact-fp32 flavour cuts its 64 x 64 tiles into up to 8 stream-K pieces)
            GemmBuilder t5 = g, t6;
            add_h1(t5);
            t5.finish(h);
            GemmProb& q0 = t6.prob(M, V, c.scratch, V);
            GemmBuilder::seg(q0, h2n, H, nullptr, w.out_fc_weight, H, H, h2n16, H2A_UNIT);
            const float* Wih[3] = {w.lstm1_weight_ih, w.W1_is_weight, w.W1_ig_weight};
            const int Nn[3] = {4 * H, H, H};
            for (int i = 0; i < 3; ++i) {
                GemmProb& q = t6.prob(M, Nn[i], c.pre1, 6 * H);
                GemmBuilder::seg(q, h2n, H, nullptr, Wih[i], in1, H, h2n16, H2A_UNIT);
            }
            t6.finish(h);
            int n6 = 1;
            for (int i = 1; i < 4; ++i) n6 = std::max(n6, gemm_tight_slabs(t6.a, i));
            if (std::max(gemm_tight_slabs(t5.a, 2), gemm_tight_slabs(t5.a, 3)) + n6 > 8) split_pre1 = false;
        }
        if (split_pre1) add_h1(g);
        const int ns = g.finish(h);
        const long long stride = (long long)M * 4 * H, stride_g = (long long)M * A;
        g.a.p[0].slab_stride = stride;
        g.a.p[0].nslab = gemm_tight_slabs(g.a, 0);
        g.a.p[1].C = c.ga_slabs; g.a.p[1].slab_stride = stride_g;
        g.a.p[1].nslab = gemm_tight_slabs(g.a, 1);
        if (split_pre1) {
            ns_h1 = std::max(gemm_tight_slabs(g.a, 2), gemm_tight_slabs(g.a, 3));
            g.a.p[2].slab_stride = g.a.p[3].slab_stride = (long long)M * 6 * H;
            g.a.p[2].nslab = g.a.p[3].nslab = ns_h1;
        }
        c.pre1_skip5 = ns_h1;
        const int ns_lstm2 = g.a.p[0].nslab;
        if ((size_t)stride * ns > c.scratch_floats) return fail("S5: slabs exceed the workspace scratch");
        if (g.launch(s, h)) return fail("S5 gemm launch failed");
        hipLaunchKernelGGL(k_lstm2, dim3(cdiv((long long)M * H, 256)), dim3(256), 0, s, c.scratch, ns_lstm2, stride, w.lstm2_bias_ih,
                           w.lstm2_bias_hh, d.img_second_lstm ? c.vproj2 : nullptr, io.rpi, io.parent, c2o, M, H, h2n, c2n, h2n16, isc);
        // the gate logits (z_g, log_softmax([z_g, zsum]), step :185-188) are nobody's input before the selection: the
        // vocabulary kernel's row blocks compute them on the side instead of a launch of their own
        // (att_ga has a quarter of LSTM2's K: fewer stream-K pieces per tile, fewer slabs for the gate logits to add)
        gate_args = GateLogitArgs{c.ga_slabs, g.a.p[1].nslab, stride_g, c.hA, w.att_g_weight, c.zsum, io.verbs, io.slot, io.rpi, c.L, M, A,
                                  io.lg_out, io.lg_stride};
    }
    // ---- S6
    {
        GemmBuilder g;
        GemmProb& p0 = g.prob(M, V, c.scratch, V);
        GemmBuilder::seg(p0, h2n, H, nullptr, w.out_fc_weight, H, H, h2n16, H2A_UNIT);
        int nblk = 0;
        if (io.s1_for_next) {
            // LSTM1 / gate sums of step t+1 over THIS step's rows: they depend on (h2, h1) only (the word enters through the
            // decode cache, the beam re-indexing through k_lstm1's parent gather), so they ride in the same launch as the
            // vocabulary projection: 3 GEMM launches per timestep instead of 4, and a longer stream-K range per workgroup.
            const float* Wih[3] = {w.lstm1_weight_ih, w.W1_is_weight, w.W1_ig_weight};
            const float* Whh[3] = {w.lstm1_weight_hh, w.W1_hs_weight, nullptr};
            const int Nn[3] = {4 * H, H, H}, off[3] = {0, 4 * H, 5 * H};
            const bool h1_done = c.pre1_skip5 > 0;         // (split_pre1: the S5 launch above already wrote the h1 part into the leading slabs of pre1)
            for (int i = 0; i < 3; ++i) {
                if (!d.h2_first_lstm && !Whh[i]) continue;
                GemmProb& p = g.prob(M, Nn[i], c.pre1 + (long long)c.pre1_skip5 * M * 6 * H + off[i], 6 * H);
                if (d.h2_first_lstm) GemmBuilder::seg(p, h2n, H, nullptr, Wih[i], in1, H, h2n16, H2A_UNIT);
                if (Whh[i] && !h1_done) GemmBuilder::seg(p, h1n, H, nullptr, Whh[i], H, H, h1n16, H2A_UNIT);
                nblk = i == 0 ? 4 : i == 1 ? 5 : 6;
            }
        }
        const int ns = g.finish(h);
        const long long stride = (long long)M * V;
        g.a.p[0].slab_stride = stride;
        for (int i = 1; i < g.a.nprob; ++i) g.a.p[i].slab_stride = (long long)M * 6 * H;
        // ... and the LSTM1 / gate problems of the next step write (and k_lstm1 adds) only the slabs THEIR tiles can meet - all three get
        // the largest of their counts, k_lstm1 takes one count for its six gate blocks
        int ns_pre1 = 1;
        for (int i = 1; i < g.a.nprob; ++i) ns_pre1 = std::max(ns_pre1, gemm_tight_slabs(g.a, i));
        for (int i = 1; i < g.a.nprob; ++i) g.a.p[i].nslab = ns_pre1;
        c.pre1_ns = (g.a.nprob > 1 ? ns_pre1 : ns) + c.pre1_skip5; c.pre1_nblk = nblk; c.pre1_stride = (long long)M * 6 * H;
        if (c.pre1_ns > 8) return fail("S6: the LSTM1 sums of the next step would need %d slabs (8 fit)", c.pre1_ns);
        // the vocabulary tiles (K = H) are cut into fewer pieces than the LSTM1 tiles (K = 2 H) they share the launch with: k_vocab
        // adds only the slabs they wrote (60 -> 40 MB of logits per beam-5 step)
        const int ns_vocab = g.a.p[0].nslab = gemm_tight_slabs(g.a, 0);
        if (g.launch(s, h)) return fail("S6 gemm launch failed");
#define VOCAB_ARGS c.scratch, ns_vocab, stride, w.out_fc_bias, M, V, io.vmode, c.top_v, c.top_i, io.full_out, io.full_stride, io.forced, \
                   io.seed, (uint32_t)io.t, io.verbs, io.slot, io.rpi, c.L, io.gt, h->vt_ptr, h->vt_ids, h->n_verbs, lds_row, gate_args, c.nvalid_dev + 2
        const int lds_row = V <= VOCAB_LDS_MAX ? 1 : 0;          // combined logits row staged in LDS (<= 96 KB)
        const size_t vsm = lds_row ? (size_t)V * sizeof(float) : 0;
        // K = beam exactly (fewer selection rounds than rounding up to a power of two); 512 threads per row from V = 4096 up
#define VOCAB_LAUNCH(KK)                                                                                              \
    if (V >= 4096) hipLaunchKernelGGL((k_vocab<KK, 512>), dim3(M), dim3(512), vsm, s, VOCAB_ARGS);                  \
    else hipLaunchKernelGGL((k_vocab<KK, 256>), dim3(M), dim3(256), vsm, s, VOCAB_ARGS);                            \
    break;
        switch (io.K) {
            case 1: VOCAB_LAUNCH(1)
            case 2: VOCAB_LAUNCH(2)
            case 3: VOCAB_LAUNCH(3)
            case 4: VOCAB_LAUNCH(4)
            case 5: VOCAB_LAUNCH(5)
            case 6: VOCAB_LAUNCH(6)
            case 7: VOCAB_LAUNCH(7)
            default: VOCAB_LAUNCH(8)
        }
#undef VOCAB_LAUNCH
#undef VOCAB_ARGS
    }
    c.st16_ok[io.cur ^ 1] = sh;               // the new state's bf16 images exist iff this step wrote them
    LAUNCHCHK();
    return 0;
}

static int check_ready(vsr_handle* h, const char* who) {
    if (!h) return fail("%s: null handle", who);
    if (!h->bound) return fail("%s: weights not bound", who);
    if (!h->prepared) return fail("%s: vsr_prepare() has not been called", who);
    return 0;
}

static int zero_state(vsr_handle* h, int M, hipStream_t s) {
    Ctx& c = h->c;
    const size_t n = (size_t)M * h->d.rnn_size * sizeof(float);
    // the four state arrays of buffer 0 are consecutive in the workspace (carve): one memset
    HIPCHK(hipMemsetAsync(c.st[0][0], 0, (size_t)(reinterpret_cast<char*>(c.st[0][3]) - reinterpret_cast<char*>(c.st[0][0])) + n, s));
    HIPCHK(hipMemsetAsync(c.st16[0][0], 0, (size_t)M * h->d.rnn_size * 2 * sizeof(uint16_t), s));     // images of the zero h1 / h2 (bf16 or fp16 pairs)
    HIPCHK(hipMemsetAsync(c.st16[0][1], 0, (size_t)M * h->d.rnn_size * 2 * sizeof(uint16_t), s));
    c.st16_ok[0] = true;
    c.st16_ok[1] = false;
    hipLaunchKernelGGL(k_init_rows, dim3(cdiv(M, 256)), dim3(256), 0, s, c.slot[0], c.word[0], h->d.bos_idx, M);
    LAUNCHCHK();
    return 0;
}

// Number of out-of-range ids (words outside [0, V), slots outside [0, L), gates outside {0, 1}, gt verbs outside [0, V)) the
// calls since the last vsr_prepare*() / vsr_bad_ids() were handed.  Such ids are clamped on the device (no out-of-bounds
// access); the reference would raise (nn.Embedding / tensor indexing).  Synchronises the stream; resets the count.
extern "C" int vsr_bad_ids(vsr_handle* h, int32_t* count, void* stream) {
    if (!h || !count) return fail("vsr_bad_ids: null argument");
    if (!h->prepared) { *count = 0; return 0; }
    hipStream_t s = (hipStream_t)stream;
    int v[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(v, h->c.nvalid_dev + 2, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemsetAsync(h->c.nvalid_dev + 2, 0, 2 * sizeof(int), s));
    *count = v[0] + v[1];              // [1]: non-padding region rows beyond the caller's bound (vsr_set_valid_rows_bound) - they got no projection
    return 0;
}

// ---------------------------------------------------------------------------------------------- greedy / sampling
static int decode_simple(vsr_handle* h, int vmode, uint64_t seed, const int64_t* forced_w, const int64_t* forced_g,
                         const float* verbs, int gt, int64_t* words, int64_t* gates, float* lp_w, float* lp_g, hipStream_t s) {
    Ctx& c = h->c;
    const int B = c.B, T = h->d.seq_len;
    if (verbs && vmode != VM_TOPK) return fail("verb forcing is only defined for greedy / beam decoding");
    if (zero_state(h, B, s)) return 1;
    if (vmode == VM_FORCED) {
        for (int t = 0; t < T; ++t) {
            hipLaunchKernelGGL(k_i64_to_i32, dim3(cdiv(B, 256)), dim3(256), 0, s, forced_w + t, (long long)T, c.forced_w32 + (size_t)t * B, B, h->d.vocab_size, c.nvalid_dev + 2);
            hipLaunchKernelGGL(k_i64_to_i32, dim3(cdiv(B, 256)), dim3(256), 0, s, forced_g + t, (long long)T, c.forced_g32 + (size_t)t * B, B, 2, c.nvalid_dev + 2);
        }
    }
    // The selection of step t is per row and tiny: where the next step's LSTM1 kernel starts from cached sums (decode cache) it makes the
    // selection itself (k_select_simple_lstm1) and k_select_simple is not launched for that step (VSR_FUSE_SELECT=0: always launched).
    SelSimpleArgs pending{};
    bool have_pending = false;
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        StepIO io{};
        io.t = t; io.M = B; io.rpi = 1; io.cur = cur;
        io.parent = nullptr; io.word_prev = c.word[cur]; io.slot = c.slot[cur]; io.fixed_slot = 0;
        io.sel_simple = have_pending ? &pending : nullptr;
        io.vmode = vmode; io.K = 1; io.full_out = nullptr; io.full_stride = 0;
        io.forced = (vmode == VM_FORCED) ? c.forced_w32 + (size_t)t * B : nullptr;
        io.seed = seed; io.verbs = verbs; io.gt = gt; io.lg_out = c.lg; io.lg_stride = 2; io.alpha_out = nullptr;
        io.s1_from_prev = t > 0 && h->xproj != nullptr;
        io.s1_for_next = t + 1 < T && h->xproj != nullptr;
        if (run_step(h, io, s)) return 1;
        const SelSimpleArgs sa{vmode, c.top_v, c.top_i, c.lg, (vmode == VM_FORCED) ? c.forced_g32 + (size_t)t * B : nullptr, seed, (uint32_t)t,
                               c.slot[cur], c.L, B, T, c.word[cur ^ 1], c.gate[cur ^ 1], c.slot[cur ^ 1], words, gates, lp_w, lp_g};
        have_pending = (h->fuse_select & 1) && t + 1 < T && h->xproj != nullptr;       // (the next step is then s1_from_prev)
        if (have_pending) pending = sa;
        else hipLaunchKernelGGL(k_select_simple, dim3(cdiv(B, 256)), dim3(256), 0, s, sa);
        LAUNCHCHK();
    }
    return 0;
}

extern "C" int vsr_greedy(vsr_handle* h, const float* verbs, int32_t gt, int64_t* words, int64_t* gates, void* stream) {
    if (check_ready(h, "vsr_greedy")) return 1;
    if (!words || !gates) return fail("vsr_greedy: null output");
    return decode_simple(h, VM_TOPK, 0, nullptr, nullptr, verbs, gt, words, gates, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int vsr_sample(vsr_handle* h, uint64_t seed, const int64_t* forced_words, const int64_t* forced_gates,
                          int64_t* words, int64_t* gates, float* lp_words, float* lp_gates, void* stream) {
    if (check_ready(h, "vsr_sample")) return 1;
    if (!words || !gates || !lp_words || !lp_gates) return fail("vsr_sample: null output");
    if ((forced_words == nullptr) != (forced_gates == nullptr)) return fail("vsr_sample: forced_words and forced_gates go together");
    return decode_simple(h, forced_words ? VM_FORCED : VM_SAMPLE, seed, forced_words, forced_gates, nullptr, 0, words, gates,
                         lp_words, lp_gates, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- beam search
extern "C" int vsr_beam(vsr_handle* h, int32_t beam, int32_t out_size, int64_t eos_word, int64_t eos_gate, const float* verbs,
                        int32_t gt, int64_t* words, int64_t* gates, float* lp_words, float* lp_gates, float* scores, void* stream) {
    if (check_ready(h, "vsr_beam")) return 1;
    Ctx& c = h->c;
    if (beam < 1 || beam > c.beam) return fail("vsr_beam: beam %d exceeds the prepared beam %d", beam, c.beam);
    if (out_size < 1 || out_size > beam) return fail("vsr_beam: out_size %d not in [1, beam]", out_size);
    if (!words || !gates) return fail("vsr_beam: null output");
    hipStream_t s = (hipStream_t)stream;
    const int B = c.B, T = h->d.seq_len;
    const int K = beam;
    if (zero_state(h, B, s)) return 1;
    // The beam selection of step t rides in the LSTM1 kernel of step t + 1 (k_select_lstm1) wherever that kernel starts from cached sums
    // (decode cache); the last step's, and every step's with VSR_FUSE_SELECT=0, is a launch of its own (k_select_beam).
    SelBeamArgs pending{};
    bool have_pending = false;
    for (int t = 0; t < T; ++t) {
        const int cur = t & 1;
        const int cb = t == 0 ? 1 : beam;
        const int M = B * cb;
        StepIO io{};
        io.sel_beam = have_pending ? &pending : nullptr;
        io.t = t; io.M = M; io.rpi = cb; io.cur = cur;
        io.parent = t == 0 ? nullptr : c.parent; io.word_prev = c.word[cur]; io.slot = c.slot[cur]; io.fixed_slot = 0;
        io.vmode = VM_TOPK; io.K = K; io.forced = nullptr; io.seed = 0; io.verbs = verbs; io.gt = gt;
        io.lg_out = c.lg; io.lg_stride = 2; io.alpha_out = nullptr;
        io.s1_from_prev = t > 0 && h->xproj != nullptr;
        io.s1_for_next = t + 1 < T && h->xproj != nullptr;
        if (run_step(h, io, s)) return 1;
        const SelBeamArgs sa{t, cb, beam, c.L, eos_word, eos_gate, c.top_v, c.top_i, c.lg, c.slot[cur], c.word[cur], c.gate[cur], c.seq[cur],
                             c.seq[cur ^ 1], c.mask[cur], c.mask[cur ^ 1], c.word[cur ^ 1], c.gate[cur ^ 1], c.slot[cur ^ 1], c.parent,
                             c.hist_parent, c.hist_word, c.hist_gate, c.hist_lpw, c.hist_lpg, B};
        have_pending = (h->fuse_select & 2) && t + 1 < T && h->xproj != nullptr;       // (the next step is then s1_from_prev)
        if (have_pending) pending = sa;
        else switch (K) {
            case 1: hipLaunchKernelGGL((k_select_beam<1>), dim3(B), dim3(64), 0, s, sa); break;
            case 2: hipLaunchKernelGGL((k_select_beam<2>), dim3(B), dim3(64), 0, s, sa); break;
            case 3: hipLaunchKernelGGL((k_select_beam<3>), dim3(B), dim3(64), 0, s, sa); break;
            case 4: hipLaunchKernelGGL((k_select_beam<4>), dim3(B), dim3(64), 0, s, sa); break;
            case 5: hipLaunchKernelGGL((k_select_beam<5>), dim3(B), dim3(64), 0, s, sa); break;
            case 6: hipLaunchKernelGGL((k_select_beam<6>), dim3(B), dim3(64), 0, s, sa); break;
            case 7: hipLaunchKernelGGL((k_select_beam<7>), dim3(B), dim3(64), 0, s, sa); break;
            default: hipLaunchKernelGGL((k_select_beam<8>), dim3(B), dim3(64), 0, s, sa); break;
        }
        LAUNCHCHK();
    }
    hipLaunchKernelGGL(k_backtrack, dim3(B), dim3(64), (size_t)3 * T * beam * sizeof(int), s, T, B, beam, out_size, c.seq[T & 1], c.hist_parent, c.hist_word,
                       c.hist_gate, c.hist_lpw, c.hist_lpg, words, gates, lp_words, lp_gates, scores);
    LAUNCHCHK();
    return 0;
}

// ---------------------------------------------------------------------------------------------- teacher forcing
extern "C" int vsr_xe_forward(vsr_handle* h, const int64_t* captions, int32_t T, float* logp_words, float* logp_gates, void* stream) {
    if (check_ready(h, "vsr_xe_forward")) return 1;
    Ctx& c = h->c;
    if (!captions || !logp_words || !logp_gates) return fail("vsr_xe_forward: null tensor");
    if (T < 1 || T > c.L) return fail("vsr_xe_forward: captions have %d steps but prepare() saw %d region slots (need 1 <= T <= L: step t reads slot t)", T, c.L);
    if (T > h->d.seq_len) return fail("vsr_xe_forward: T %d exceeds seq_len %d (workspace is sized by seq_len)", T, h->d.seq_len);
    hipStream_t s = (hipStream_t)stream;
    const int B = c.B, V = h->d.vocab_size;
    if (zero_state(h, B, s)) return 1;
    for (int t = 0; t < T; ++t)
        hipLaunchKernelGGL(k_i64_to_i32, dim3(cdiv(B, 256)), dim3(256), 0, s, captions + t, (long long)T, c.cap32 + (size_t)t * B, B, V, c.nvalid_dev + 2);
    for (int t = 0; t < T; ++t) {
        StepIO io{};
        io.t = t; io.M = B; io.rpi = 1; io.cur = t & 1;
        io.parent = nullptr; io.word_prev = c.cap32 + (size_t)t * B; io.slot = nullptr; io.fixed_slot = t;
        io.vmode = VM_FULL; io.K = 1; io.full_out = logp_words + (size_t)t * V; io.full_stride = (long long)T * V;
        io.forced = nullptr; io.seed = 0; io.verbs = nullptr; io.gt = 0;
        io.lg_out = logp_gates + (size_t)t * 2; io.lg_stride = (long long)T * 2; io.alpha_out = nullptr;
        if (run_step(h, io, s)) return 1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- single step
__global__ void k_step_slots(int t, const int64_t* slot_in, const int64_t* prev_gate, int L, int M, int* slot32, int64_t* slot_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    long long k = slot_in[i];
    if (t > 0) {
        k += prev_gate[i];
        k = k < 0 ? 0 : (k > L - 1 ? L - 1 : k);
    }
    slot32[i] = (int)k;
    slot_out[i] = k;
}

extern "C" int vsr_step(vsr_handle* h, int32_t t, int32_t rows_per_image, const int64_t* prev_words, const int64_t* prev_gates,
                        const float* h1, const float* c1, const float* h2, const float* c2, const int64_t* slot,
                        float* h1_out, float* c1_out, float* h2_out, float* c2_out, int64_t* slot_out,
                        const float* verbs, int32_t gt, float* logp_words, float* logp_gates, void* stream) {
    if (check_ready(h, "vsr_step")) return 1;
    Ctx& c = h->c;
    if (rows_per_image < 1 || rows_per_image > c.beam) return fail("vsr_step: rows_per_image %d exceeds the prepared beam %d", rows_per_image, c.beam);
    if (t > 0 && (!prev_words || !prev_gates)) return fail("vsr_step: previous outputs required for t > 0");
    hipStream_t s = (hipStream_t)stream;
    const int M = c.B * rows_per_image, H = h->d.rnn_size, V = h->d.vocab_size;
    const size_t n = (size_t)M * H * sizeof(float);
    const float* in[4] = {h1, c1, h2, c2};
    float* out[4] = {h1_out, c1_out, h2_out, c2_out};
    for (int j = 0; j < 4; ++j) HIPCHK(hipMemcpyAsync(c.st[0][j], in[j], n, hipMemcpyDeviceToDevice, s));
    c.st16_ok[0] = false;                     // the caller's state has no bf16 image
    if (h->h2_on && !h->bf16_on && h->x3_on) {           // f16x2: the hidden states are unit-class GEMM operands (include/vsrcap.h, limits of the flavour)
        const long long ne = (long long)M * H;
        hipLaunchKernelGGL(k_count_outside_unit, dim3(cdiv(ne, 256)), dim3(256), 0, s, h1, ne, c.nvalid_dev + 2);
        hipLaunchKernelGGL(k_count_outside_unit, dim3(cdiv(ne, 256)), dim3(256), 0, s, h2, ne, c.nvalid_dev + 2);
    }
    hipLaunchKernelGGL(k_step_slots, dim3(cdiv(M, 256)), dim3(256), 0, s, t, slot, prev_gates, c.L, M, c.slot[0], slot_out);
    if (t == 0) hipLaunchKernelGGL(k_fill_i32, dim3(cdiv(M, 256)), dim3(256), 0, s, c.word[0], h->d.bos_idx, M);
    else hipLaunchKernelGGL(k_i64_to_i32, dim3(cdiv(M, 256)), dim3(256), 0, s, prev_words, 1LL, c.word[0], M, V, c.nvalid_dev + 2);
    StepIO io{};
    io.t = 1;  // state segments are always live here (the caller may pass a non-zero state at t == 0)
    io.M = M; io.rpi = rows_per_image; io.cur = 0;
    io.parent = nullptr; io.word_prev = c.word[0]; io.slot = c.slot[0]; io.fixed_slot = 0;
    io.vmode = VM_FULL; io.K = 1; io.full_out = logp_words; io.full_stride = V; io.forced = nullptr; io.seed = 0;
    io.verbs = verbs; io.gt = gt; io.lg_out = logp_gates; io.lg_stride = 2; io.alpha_out = nullptr;
    if (run_step(h, io, s)) return 1;
    for (int j = 0; j < 4; ++j) HIPCHK(hipMemcpyAsync(out[j], c.st[1][j], n, hipMemcpyDeviceToDevice, s));
    return 0;
}

// ---------------------------------------------------------------------------------------------- training path
#include "train.inc.h"
#include "cider.inc.h"
#include "ssp.inc.h"

static TrainCtx* new_train_ctx() { return new TrainCtx(); }
static void free_train_ctx(TrainCtx* t) { delete t; }
static void invalidate_train_ctx(TrainCtx* t) { if (t) t->valid = false; }

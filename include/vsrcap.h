/*
 * vsrcap.h - C ABI of the MI355X-native VSR-guided captioning decoder (libvsrcap.so).
 *
 * This is the drop-in boundary UNDER the reference's Python class
 *   models.ControllableCaptioningModel   (/root/reference/models/controllable_captioning.py:10)
 * and its decode / train loops            (/root/reference/models/CaptioningModel.py:22-294).
 * The reference has no native ABI of its own (it is 100 % Python on ATen); each entry point below names
 * the reference method it replaces.  The host-side mirror that keeps the reference's Python signatures
 * lives in vsr-guided-cic_amd/models/ and binds these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name says host_; tensors are dense row-major fp32 /
 *     int64 / int32 exactly as the reference lays them out (weights: [out, in]).
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); no call synchronises the host, with ONE
 *     exception: vsr_prepare() reads back a single integer (the number of non-padding region rows, which sizes the
 *     hoisted att_va GEMM) and therefore waits for the stream once - unless the caller has given an upper bound on that
 *     number (vsr_set_valid_rows_bound): then no call of the data path waits.
 *   - the library allocates nothing on the launch path: the caller provides one workspace of
 *     vsr_workspace_bytes() bytes (16-byte aligned) that must stay untouched between vsr_prepare() and
 *     the decode / forward calls that use it.
 *   - return value 0 = ok, non-zero = error; vsr_last_error() returns a thread-local message.
 *   - one handle per device, not re-entrant per handle.
 */
#ifndef VSRCAP_H
#define VSRCAP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSR_ABI_VERSION 1
#define VSR_MAX_BEAM 8

/* ctor arguments of ControllableCaptioningModel (controllable_captioning.py:11-12) */
typedef struct vsr_dims {
    int32_t seq_len;              /* T  */
    int32_t vocab_size;           /* V  */
    int32_t bos_idx;
    int32_t det_feat_size;        /* D  (multiple of 4) */
    int32_t input_encoding_size;  /* E  (multiple of 4) */
    int32_t rnn_size;             /* H  (multiple of 4) */
    int32_t att_size;             /* A  (multiple of 4) */
    int32_t h2_first_lstm;        /* 1: LSTM1 input is [h2 | vbar | x]   (:36-39)  */
    int32_t img_second_lstm;      /* 1: LSTM2 input is [h1 | att | vbar] (:54-57)  */
} vsr_dims;

/* the 28 state_dict tensors, borrowed (no copy: optimizer updates stay visible); layout per
 * controllable_captioning.py:23-68, row-major [out, in] */
typedef struct vsr_weights {
    const float* embed_weight;          /* (V, E) */
    const float* W1_is_weight;          /* (H, in1)   in1 = [H +] D + E */
    const float* W1_is_bias;            /* (H) */
    const float* W1_hs_weight;          /* (H, H) */
    const float* W1_hs_bias;
    const float* att_va_weight;         /* (A, D) */
    const float* att_ha_weight;         /* (A, H) */
    const float* att_a_weight;          /* (1, A) */
    const float* att_sa_weight;         /* (A, H) */
    const float* att_s_weight;          /* (1, A) */
    const float* lstm1_weight_ih;       /* (4H, in1)  gate order i,f,g,o */
    const float* lstm1_weight_hh;       /* (4H, H) */
    const float* lstm1_bias_ih;         /* (4H) */
    const float* lstm1_bias_hh;
    const float* lstm2_weight_ih;       /* (4H, in2)  in2 = H + D [+ D] */
    const float* lstm2_weight_hh;
    const float* lstm2_bias_ih;
    const float* lstm2_bias_hh;
    const float* out_fc_weight;         /* (V, H) */
    const float* out_fc_bias;           /* (V) */
    const float* s_fc_weight;           /* (D, H) */
    const float* s_fc_bias;             /* (D) */
    const float* W1_ig_weight;          /* (H, in1) */
    const float* W1_ig_bias;
    const float* W1_hg_weight;          /* (H, H) */
    const float* W1_hg_bias;
    const float* att_ga_weight;         /* (A, H) */
    const float* att_g_weight;          /* (1, A) */
} vsr_weights;

typedef struct vsr_handle vsr_handle;

/* ---- lifetime ------------------------------------------------------------------------------------- */
int vsr_abi_version(void);
const char* vsr_last_error(void);
/* replaces ControllableCaptioningModel.__init__ shape bookkeeping (:11-70) */
int vsr_create(const vsr_dims* dims, vsr_handle** out);
void vsr_destroy(vsr_handle* h);
int vsr_bind_weights(vsr_handle* h, const vsr_weights* w);
/* verb_2_vob_all table of step_v (:25-34, :283-292) as CSR: ids of verb v are
 * vocab_ids[row_ptr[v] .. row_ptr[v+1]); verbs >= n_verbs have no entry.  DEVICE pointers, borrowed. */
int vsr_set_verb_table(vsr_handle* h, const int32_t* row_ptr, const int32_t* vocab_ids, int32_t n_verbs);

/* ---- decode cache (optional, inference) ----------------------------------------------------------------
 * (V, 6H) projection of the embedding table through the x columns of lstm_cell_1.weight_ih / W1_is / W1_ig
 * (step :147-152, :181): weight-only work, so it is hoisted out of the call.  The buffer is caller-owned and must
 * outlive its use; vsr_bind_weights() or vsr_build_decode_cache(h, NULL, ...) drops it.  Rebuild after the
 * weights change (the training calls never use it). */
size_t vsr_decode_cache_floats(const vsr_handle* h);
int vsr_build_decode_cache(vsr_handle* h, float* buffer, size_t n_floats, void* stream);

/* ---- bf16 throughput mode (BASELINE configs[3]; never the parity path) -----------------------------------
 * The reference is fp32 throughout (init_state hard-codes float32, controllable_captioning.py:109-115); token parity and the
 * 1e-4 loss bound are fp32 claims.  This mode trades them for speed: every matrix product takes bf16 operands
 * (v_mfma_f32_32x32x16_bf16) with fp32 accumulation; states, activations, reductions, losses, gradients and the master
 * weights stay fp32.  vsr_refresh_bf16_weights(h, buffer, ...) converts the 14 weight matrices into the caller's buffer
 * (vsr_bf16_weight_bytes) and switches the handle to bf16; call it again after every weight update (the copies are not
 * borrowed views).  buffer = NULL switches back to fp32.  Sizes must be multiples of 8 in this mode. */
size_t vsr_bf16_weight_bytes(const vsr_handle* h);
int vsr_refresh_bf16_weights(vsr_handle* h, void* buffer, size_t bytes, void* stream);

/* fp32 GEMM flavour of a handle.  0: exact k-ordered fp32 fma chain on v_mfma_f32_32x32x2_f32 / 16x16x4_f32 for every launch.
 * 1 ("f32x3", the library's default): every fp32 operand is split into three bf16 terms on its way into the matrix core
 * (x = hi + mid + lo exactly) and a product is six v_mfma_f32_*_bf16 with fp32 accumulation; the dropped cross terms are below one
 * fp32 rounding of the product.  Operands stay fp32 in memory (no copies).  Routing by the rows of a launch (csrc/vsrcap.hip,
 * GemmBuilder::finish): at most 80 rows (VSR_X3S_MAX) the weight-streaming kernel of gemm_x3s.h; up to 128 rows the 128 x 128 tile of
 * gemm_x3.h; 129 .. 192 rows the exact kernels of gemm_f32.h; from 193 rows (VSR_X3_MIN_ROWS) the 128 x 256 tile.  The flavours are
 * not bit-identical (different summation orders); every parity test runs in each of them (tests/conftest.py).  A change of mode voids
 * the decode cache, a saved training forward and the hoisted projections: call vsr_prepare*() again. */
int vsr_set_gemm_mode(vsr_handle* h, int32_t mode);

/* ---- "f16x2": the fp32 flavour with three MFMAs per product and weights that are never split in a kernel (csrc/gemm_h2.h) ------
 * On top of mode 1.  vsr_refresh_h2_weights(h, buffer, ...) writes, into the caller's buffer (vsr_h2_weight_bytes, 256-byte aligned),
 * an fp16-PAIR image of each of the 14 weight matrices and of the embedding table - x 2^e = hi + lo with hi = f16(x 2^e),
 * lo = f16(x 2^e - hi), e chosen per tensor from its max |x| so that nothing overflows, 4 bytes per element in the byte geometry of the
 * fp32 matrix - plus the table of scale exponents (4 KB at the head of the buffer; its second half is scratch the backward pass of
 * training uses for the bounds of its gradient operands); vsr_prepare*() then also measures the bounds of the region / detection
 * operands.  From then on every launch whose W operands have images and whose A operands have a bound - all GEMMs of decoding and of
 * the training pass, forward and backward - forms a product as lo.hi + hi.lo + hi.hi on v_mfma_f32_*_f16 with fp32 accumulation.
 * The decoder's own vectors (h1, h2, s_t, g_t, the attended vector) are written as fp16-pair images by the kernels that produce them
 * and both operands go global -> LDS by DMA (csrc/gemm_h2a.h); a caller's fp32 tensors (vsr_prepare*(), a state handed to vsr_step)
 * and the gradients of the backward pass are scaled and split inside the kernel (csrc/gemm_h2.h), the gradients by bounds the
 * kernels that write them measure on the device.  Error against fp64 (tests/test_gpu_h2.py, next to the fma chain and f32x3): the
 * accumulation's, not the split's.  Call it again after every weight update (the images are not views); buffer = NULL: back to
 * mode 1 everywhere.  Sizes must be multiples of 8.  Turning it on or off voids what a change of mode voids.
 * Limits: (1) a state handed to vsr_step by the caller is taken as unit-bounded (|h| < 1, what sigmoid x tanh produces; scale 2^15):
 * values of |h| >= 2 overflow fp16 - they are counted into vsr_bad_ids()'s count; the f32x3 / f32 flavours accept any state.  (2) The backward pass keeps the measured bounds of its
 * gradient operands in 8 slots per timestep of a 512-slot device table: up to T = 62 timesteps it runs on the f16x2 kernels, from
 * T = 63 on the f32x3 kernels (the forward pass stays f16x2; tests/test_gpu_train.py covers both sides of the limit).  (3) The
 * transposed operands of an f16x2 backward pass exist ONLY as images: a launch that names one and cannot take an f16x2 kernel
 * (vsr_set_gemm_mode / a tile override changed in between) fails instead of reading unwritten fp32 buffers.
 * Environment (experiments): VSR_H2_AIMG=0 (no producer-written A images), VSR_H2S_MAX (rows up to which the weight-streaming kernel
 * is used), VSR_ALIGNED_EFF (percent of busy CUs below which a wide launch takes stream-K ranges instead of k-aligned pieces);
 * round 6: VSR_SPLIT_PRE1=0 (the h1 part of the next step's LSTM1 sums back in the vocabulary launch, as in rounds 2-5),
 * VSR_ATTEND_PARTS=1 (one attention workgroup per row also in launches of <= 128 rows), VSR_XCD_GROUPS=1 (k-aligned plans deal
 * whole m-groups of tiles to an XCD: measured slower, off).  None of them changes a result beyond fp32 summation order; every
 * default is the setting the parity suite ran with. */
size_t vsr_h2_weight_bytes(const vsr_handle* h);
int vsr_refresh_h2_weights(vsr_handle* h, void* buffer, size_t bytes, void* stream);

/* ---- workspace ------------------------------------------------------------------------------------ */
/* bytes needed for B images with L slots of R regions, R0 pooled regions, decoding with up to `beam`
 * hypotheses per image (1 for greedy / sampling / teacher forcing). */
size_t vsr_workspace_bytes(const vsr_handle* h, int32_t B, int32_t R0, int32_t L, int32_t R, int32_t beam);

/* ---- hoisted per-image work (step :126-128 pooled descriptor, :161 att_va(regions), :159 row masks) - */
/* det (B,R0,D); regions (B,L,R,D) = statics[1] for decoding or seqs[1] (L == T) for teacher forcing.
 * Both tensors are borrowed until the last call that uses this prepare. */
int vsr_prepare(vsr_handle* h, const float* det, int32_t B, int32_t R0, const float* regions, int32_t L, int32_t R,
                int32_t beam, void* workspace, size_t workspace_bytes, void* stream);

/* ---- index-list region format (decode side) -------------------------------------------
 * The reference's callers materialise statics[1] as a dense (rows, L, R, D) copy of detection features:
 * data/field.py:44-61 (COCOControlSequenceField._fill: np.take of det_features rows per slot) and
 * coco_scripts/eval_coco.py:222-247 (slot permutation, compaction, last-slot replication, then one
 * beam_search_v call per image with det expanded to n_caps rows).  The decoder itself only needs to know
 * WHICH row of the image's feature matrix each slot entry is.  These entry points take that instead:
 *   det       (n_img, R0, D)  pooled detections, one per IMAGE          (statics[0] before the .expand)
 *   bank      (n_img, Rb, D)  the feature matrix the slots were taken from (may alias det when Rb == R0)
 *   row_img   (B) int32 or NULL: image of decoder row b (several captions per image); NULL = identity, B == n_img
 *   slot_idx  (B, L, R) int32: bank row of slot entry (b, l, r), -1 = padding row
 * and are defined as vsr_prepare() on the dense tensor regions[b,l,r,:] = slot_idx < 0 ? 0 : bank[row_img[b], slot_idx, :]
 * (same masks, same tokens).  att_va runs over the n_img * Rb bank rows instead of B * L * R copies.
 * An index outside [-1, Rb) or an image outside [0, n_img) fails the call (reported through the same single
 * read-back as the row count).  The training calls reject a handle prepared this way. */
size_t vsr_workspace_bytes_indexed(const vsr_handle* h, int32_t B, int32_t R0, int32_t n_img, int32_t Rb, int32_t L,
                                   int32_t R, int32_t beam);
int vsr_prepare_indexed(vsr_handle* h, const float* det, int32_t n_img, int32_t R0, const float* bank, int32_t Rb,
                        const int32_t* row_img, int32_t B, const int32_t* slot_idx, int32_t L, int32_t R, int32_t beam,
                        void* workspace, size_t workspace_bytes, void* stream);
/* vsr_prepare*() projects only the NON-PADDING region rows (att_va(0) = 0) and needs their number to size that launch: by default the
 * count is read back (the one place where the library waits for the device).  A caller that knows an upper bound - eval_coco.py:222-237
 * builds det_seqs_recons on the host, train.py's loader pads on the host - passes it here (sticky; 0 = back to the read-back): the
 * following vsr_prepare*() calls then never synchronise (the row list is padded to the bound on the device; the backward pass of training
 * reads the padding as zero rows).  A bound that is TOO SMALL is a caller error the call itself cannot report (nothing is read back): the
 * rows beyond it get a ZERO att_va projection (a defined result, not stale workspace contents) and their number is added to
 * vsr_bad_ids()'s count, as are bad slot indices of the index-list format in this mode - pair a bound with that check when in doubt. */
int vsr_set_valid_rows_bound(vsr_handle* h, int64_t max_valid_rows);
/* mask[i] = (sum_d rows[i, :] != 0), the reference's zero-row test (controllable_captioning.py:126,159) */
int vsr_row_mask(const float* rows, int64_t n_rows, int32_t D, float* mask, void* stream);
/* eval_coco.py:222-241 on index lists, N captions at once:
 *   rank (N, L) int32: final_rank padded with -1 (position j takes slot rank[j]); slots whose rows are all
 *   padding / all-zero bank rows (bank_mask from vsr_row_mask over the bank, NULL = every indexed row counts)
 *   are dropped, the last kept slot is replicated to the end (:230-234); verbs_out[j] = verbs[rank[j]] or -1
 *   where the permutation has no row j, NOT compacted (:237-238).  verbs / verbs_out (N, L) fp32 or NULL. */
int vsr_reorder_slots(const int32_t* slot_idx, const int32_t* rank, const float* verbs, const float* bank_mask,
                      const int32_t* row_img, int32_t N, int32_t L, int32_t R, int32_t Rb, int32_t* slot_out,
                      float* verbs_out, void* stream);

/* ---- decode loops --------------------------------------------------------------------------------- */
/* verbs: (B,L) fp32 or NULL (-1 = no verb) -> step_v semantics; gt as in beam_search_v(..., gt=) */
/* CaptioningModel.test (:38-52): words/gates (B,T) int64 */
int vsr_greedy(vsr_handle* h, const float* verbs, int32_t gt, int64_t* words, int64_t* gates, void* stream);
/* CaptioningModel.sample_rl (:54-76).  forced_words/gates (B,T) int64 or NULL: replay given samples instead
 * of drawing (Philox4x32-10 keyed by seed, Gumbel-max).  lp_* (B,T) fp32 = log-prob of the sample. */
int vsr_sample(vsr_handle* h, uint64_t seed, const int64_t* forced_words, const int64_t* forced_gates,
               int64_t* words, int64_t* gates, float* lp_words, float* lp_gates, void* stream);
/* CaptioningModel.beam_search / beam_search_v (:116-294): joint (word x gate) beam search.
 * words/gates (B,out_size,T) int64; lp_* (B,out_size,T) fp32 (the reference's per-slot log-probs, quirk 2
 * of SURVEY.md 8a); scores (B,out_size) fp32 final sequence log-probs, may be NULL. */
int vsr_beam(vsr_handle* h, int32_t beam, int32_t out_size, int64_t eos_word, int64_t eos_gate, const float* verbs,
             int32_t gt, int64_t* words, int64_t* gates, float* lp_words, float* lp_gates, float* scores, void* stream);

/* ---- teacher forcing (CaptioningModel.forward :22-36) --------------------------------------------- */
/* captions (B,T) int64; prepare() must have been called with regions = seqs[1] (B,L,R,D), L >= T (forward() only
 * needs ctrl_seq.size(1) >= captions.size(1): step t reads slot t), beam = 1.  logp_words (B,T,V), logp_gates (B,T,2). */
int vsr_xe_forward(vsr_handle* h, const int64_t* captions, int32_t T, float* logp_words, float* logp_gates, void* stream);

/* ---- input contract check --------------------------------------------------------------------------------
 * Word ids outside [0, V) (nn.Embedding raises on them, controllable_captioning.py:144), slot traces outside [0, L),
 * replayed gates outside {0, 1} and gt-verb ids outside [0, V) (step_v :280) are CLAMPED on the device - never an
 * out-of-bounds access - and counted.  *count = ids clamped by the calls since the last vsr_prepare*() or vsr_bad_ids();
 * the call synchronises the stream and resets the counter.  (With vsr_set_valid_rows_bound: plus the region rows beyond the bound.) */
int vsr_bad_ids(vsr_handle* h, int32_t* count, void* stream);

/* ---- single timestep (ControllableCaptioningModel.step / step_v :117-297), feedback mode ---------- */
/* state in/out: h1,c1,h2,c2 (M,H) fp32 and slot (M) int64, M = B * rows_per_image (rows of one image
 * adjacent).  prev_words/prev_gates (M) int64 or NULL at t == 0.  logp_words (M,V), logp_gates (M,2). */
int vsr_step(vsr_handle* h, int32_t t, int32_t rows_per_image, const int64_t* prev_words, const int64_t* prev_gates,
             const float* h1, const float* c1, const float* h2, const float* c2, const int64_t* slot,
             float* h1_out, float* c1_out, float* h2_out, float* c2_out, int64_t* slot_out,
             const float* verbs, int32_t gt, float* logp_words, float* logp_gates, void* stream);

/* ---- training: forward with saved activations + hand-written BPTT backward ---------------------------
 * replaces autograd through CaptioningModel.forward (XE, coco_scripts/train.py:103-113) and through
 * sample_rl's log-probs (SCST, train.py:151-178).  vsr_prepare(beam = 1) must have been called with the region
 * tensor the steps read: seqs[1] (B,T,R,D) for XE (slots = NULL: step t reads slot t) or statics[1] (B,L,R,D)
 * plus the slot trace (B,T) for a replayed sample.
 *   word_in (B,T) int64: word fed at step t (XE: captions; SCST: [bos, sample[:, :-1]])
 *   logp_words (B,T,V), logp_gates (B,T,2): outputs; they and the training workspace must stay untouched until
 *   vsr_train_backward, which overwrites the 28 tensors of *grads (same layout / field order as vsr_weights)
 *   with dLoss/dW given dLoss/dlogp_words and dLoss/dlogp_gates. */
size_t vsr_train_workspace_bytes(const vsr_handle* h, int32_t B, int32_t T);
int vsr_train_forward(vsr_handle* h, const int64_t* word_in, const int64_t* slots, int32_t T, float* logp_words,
                      float* logp_gates, void* train_workspace, size_t train_workspace_bytes, void* stream);
int vsr_train_backward(vsr_handle* h, const float* grad_logp_words, const float* grad_logp_gates, const vsr_weights* grads,
                       void* stream);
/* More than one live forward (the reference runs under eager autograd: two forwards then (l1 + l2).backward(), micro-batches, a decode
 * call between a forward and its backward all just work, CaptioningModel.py:22-36).  A forward's saved state lives in the TWO caller
 * buffers it was given - the vsr_prepare*() workspace and the training workspace - and stays differentiable for as long as no later
 * vsr_prepare*() / vsr_train_forward() is handed memory that overlaps either of them (and the GEMM flavour / weight binding does not
 * change).  vsr_train_generation() identifies the forward just taken (0 = none; strictly increasing per handle).  A caller with
 * several forwards alive gives each its own pair of buffers, records the generation after each forward, and calls
 * vsr_train_select(h, generation, stream) before vsr_train_backward(): it makes that forward the handle's current one again (pointers,
 * image registrations, the per-batch rows of the f16x2 exponent table) or fails if its buffers have been reused.  A caller with ONE
 * pair of buffers (the round-1..5 contract) sees the old behaviour: the next vsr_prepare*() voids the saved forward. */
int64_t vsr_train_generation(const vsr_handle* h);
int vsr_train_select(vsr_handle* h, int64_t generation, void* stream);
/* Data-parallel hook (the reference is single-device, coco_scripts/train.py:22): vsr_train_backward finishes the
 * 28 gradients in buckets, largest first, and records a HIP event after each.  bucket_of[i] = bucket of gradient i (field
 * order of vsr_weights), static.  vsr_train_wait_bucket() makes `stream` wait on the device for one bucket of the LAST
 * backward, so its all-reduce (RCCL, side stream) overlaps the remaining weight-gradient GEMMs.  The *grads pointers may
 * point into one flat caller buffer laid out in bucket order: each bucket is then one contiguous collective. */
int vsr_train_bucket_map(int32_t* bucket_of, int32_t* n_buckets);
int vsr_train_wait_bucket(vsr_handle* h, int32_t bucket, void* stream);

/* test hook: copy an internal buffer of the saved training pass ("dpre1", "dpre2", "dh2_voc", "gates1", ...) */
int vsr_debug_copy(vsr_handle* h, const char* name, float* dst, size_t n_floats, void* stream);

/* ---- SCST reward on the device ------------------------------------------------------
 * Per-sample CIDEr-D on token ids: replaces the host section of the RL step, coco_scripts/train.py:154-172 (D2H of the
 * sampled ids, TextField.decode, groupby de-duplication, PTBTokenizer, speaksee.evaluation.Cider.compute_score, H2D).
 * speaksee==0.0.1 is absent: the algorithm is coco-caption's published CIDEr-D (oracle/cider_oracle.py, parity unpinned).
 *   keys[k] / idf[k] / counts[k], k = 0..3: the corpus table of order k + 1 built once from the training references
 *     (train.py:67): sorted keys = ids packed 16 bits each (first id lowest), idf = log(#samples) - log(max(1, df));
 *     ref_len = log(#samples).  All on the device except `counts` (host).
 *   cand (N, T) int64, refs (N, n_ref, Tr) int64: captions end at `eos` or `pad`; consecutive repeats collapse; ids with
 *     drop[id] != 0 (punctuation, V bytes, may be NULL) are removed; rewards (N) fp32.  T, Tr <= 64, V <= 65535. */
int vsr_cider_rewards(const uint64_t* const* keys, const double* const* idf, const int32_t* counts, double ref_len,
                      const int64_t* cand, int32_t N, int32_t T, const int64_t* refs, int32_t n_ref, int32_t Tr,
                      int64_t eos, int64_t pad, const uint8_t* drop, int32_t V, double sigma, float* rewards, void* stream);

/* ---- ordering models of the eval loop on the device ------------------------------------
 * coco_scripts/eval_coco.py:127-221 calls, per caption and per verb, S_SSP.generate (models/sort_model.py:105-183, batch
 * size 1: a 3+3-layer 512-d transformer that orders the verb's semantic roles by greedy "pick from the remaining roles")
 * and, per repeated role, SinkhornNet (models/sinkhorn_network.py:39-51) followed by munkres on the transposed matrix
 * (:185-189) - each with host round trips.  These entry points take ALL sequences / items of a loader batch at once.
 * Weights: borrowed fp32 device pointers in the reference's [out, in] layout (state_dict names in the comments). */
typedef struct vsr_ssp_layer {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *ln3_w, *ln3_b;    /* layer_norm1..3 (ln3: decoder layers only)               */
    const float *Wq, *bq, *Wk, *bk, *Wv, *bv, *Wo, *bo;             /* attention.linear_{Q,K,V,O}; the decoder's cross attention */
                                                                    /* re-uses them (sort_modules.py:88), cross_attention.* is dead */
    const float *W1, *b1, *W2, *b2;                                 /* ff_layer.w_1 (2048,512), w_2 (512,2048)                 */
} vsr_ssp_layer;
typedef struct vsr_ssp_weights {
    const float* sr_embed;       /* sr_embed_layer.weight (26, 512)  (shared by encoder.* and decoder.embed_layer) */
    const float* v_embed;        /* v_embed_layer.weight (n_verbs, 512) */
    int64_t n_verbs;
    const float *fc_w, *fc_b;    /* encoder.fc_feat */
    vsr_ssp_layer enc[3];        /* encoder.encoder_layers.N */
    const float *enc_ln_w, *enc_ln_b;
    vsr_ssp_layer dec[3];        /* decoder.encoder_layers.N */
    const float *dec_ln_w, *dec_ln_b;
    const float *exp_w, *exp_b;  /* expander_nn (26, 512) */
} vsr_ssp_weights;
typedef struct vsr_sinkhorn_weights {
    const float *W1_txt_w, *W1_txt_b, *W1_vis_w, *W1_vis_b, *W2_vis_w, *W2_vis_b, *W_fc_pos_w, *W_fc_pos_b, *W_fc_w, *W_fc_b;
    int32_t N, n_iters;          /* SinkhornNet(N, n_iters, tau): eval_coco.py:101 uses (10, 20, 0.1) */
    float tau;
} vsr_sinkhorn_weights;
typedef struct vsr_ssp vsr_ssp;
int vsr_ssp_create(vsr_ssp** out);
void vsr_ssp_destroy(vsr_ssp* e);
int vsr_ssp_bind(vsr_ssp* e, const vsr_ssp_weights* ssp /* or NULL */, const vsr_sinkhorn_weights* sinkhorn /* or NULL */);
size_t vsr_ssp_workspace_bytes(int32_t S);
/* S_SSP.generate(verb, roles, mode='not-normal') for S sequences: verbs (S) int64, roles (S,10) int32 (0 = padding) ->
 * pred (S,10) int32 (the roles in generated order, 0 beyond), logp (S,10) fp32 (the reference returns these truncated to
 * integers, sort_model.py:121, and its callers ignore them). */
int vsr_ssp_generate(vsr_ssp* e, const int64_t* verbs, const int32_t* roles, int32_t S, int32_t* pred, float* logp, void* workspace,
                     size_t workspace_bytes, void* stream);
size_t vsr_sinkhorn_workspace_bytes(int32_t Q, int32_t N);
/* SinkhornNet.forward + assignment for Q items: seq (Q,N,2352) -> tr (Q,N,N) or NULL, assign (Q,N) int32 with
 * assign[q][i] = the column munkres pairs with row i of tr[q]^T under cost max - value (eval_coco.py:185-189).
 * munkres is not in the image: the kernel computes the optimum of that cost matrix (Kuhn-Munkres, fp64); parity of the
 * assignment is pinned to the optimum, not to the package. */
int vsr_sinkhorn_assign(vsr_ssp* e, const float* seq, int32_t Q, float* tr, int32_t* assign, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ---- measurement (bench.py roofline leg) ------------------------------------------------------------ */
/* Between begin and end every fp32-MFMA GEMM launch is bracketed by a pair of pre-created HIP events on the
 * caller's stream.  end() synchronises the stream and returns the summed launch durations, the number of
 * launches and their ALGORITHMIC flops (2*M*N*K with the real, unpadded sizes). */
int vsr_profile_begin(vsr_handle* h);
/* the same, timing only every `every`-th GEMM launch: an event pair costs ~1 us of stream time, which a throughput
 * measurement taken in the same region would otherwise pay on every launch.  vsr_profile_seen() = launches since begin;
 * vsr_profile_end() then reports the timed launches only (their count, summed duration and flops). */
int vsr_profile_begin_sampled(vsr_handle* h, int32_t every);
int64_t vsr_profile_seen(const vsr_handle* h);
/* ALGORITHMIC bytes of the timed launches so far (operands + outputs once each; bf16 W copies count 2 bytes per element) */
double vsr_profile_bytes(const vsr_handle* h);
int vsr_profile_end(vsr_handle* h, void* stream, double* gemm_ms, int64_t* gemm_launches, double* gemm_flops);

#ifdef __cplusplus
}
#endif
#endif /* VSRCAP_H */

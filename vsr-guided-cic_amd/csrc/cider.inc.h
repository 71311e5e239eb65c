// SCST reward on the device: per-sample CIDEr-D on token ids, replacing the host section of the RL step
// (coco_scripts/train.py:154-172: ids -> D2H -> decode -> groupby de-duplication -> PTB tokenisation -> speaksee Cider ->
// H2D).  speaksee==0.0.1 is not in the image: the algorithm is the published CIDEr-D of coco-caption's cider_scorer.py,
// restated in oracle/cider_oracle.py (parity unpinned: no reference vector exists for this path).
//   clean-up per caption: stop at the first <eos>, collapse consecutive repeats, drop ids flagged in `drop` (punctuation)
//   n = 1..4: tf x idf vectors (idf = log(#corpus samples) - log(max(1, df)), from the corpus table), clipped cosine
//   similarity against every reference, Gaussian length penalty on the BIGRAM counts, mean over n and references, x 10.
// One wave per candidate caption; captions are <= 64 tokens; n-gram keys pack four 16-bit ids (V <= 65535).
struct CiderTable {
    const unsigned long long* keys[4];   // sorted n-gram keys of the corpus, per order
    const double* idf[4];                // log(#samples) - log(max(1, df)) of each key
    int count[4];
    double ref_len;                      // log(#samples): idf of an n-gram the corpus does not contain
};

constexpr int CIDER_MAXL = 64;

__device__ inline double cider_idf(const CiderTable& tb, int k, unsigned long long key) {
    int lo = 0, hi = tb.count[k] - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const unsigned long long v = tb.keys[k][mid];
        if (v == key) return tb.idf[k][mid];
        if (v < key) lo = mid + 1; else hi = mid - 1;
    }
    return tb.ref_len;
}

// clean-up of one caption by lane 0: returns its length
__device__ inline int cider_clean(const int64_t* src, int T, long long eos, long long pad, const unsigned char* drop, int V, int* dst) {
    int n = 0;
    long long prev = -1;
    bool have_prev = false;
    for (int t = 0; t < T; ++t) {
        const long long w = src[t];
        if (w == eos || w == pad) break;
        if (have_prev && w == prev) continue;      // itertools.groupby: consecutive repeats collapse (before punctuation is dropped)
        prev = w; have_prev = true;
        if (drop && w >= 0 && w < V && drop[w]) continue;
        if (n < CIDER_MAXL) dst[n++] = (int)w;
    }
    return n;
}

// per position i of a cleaned caption: key, tf x idf of its n-gram of order k + 1, and whether it is the first occurrence
__device__ inline void cider_vec(const CiderTable& tb, const int* s, int L, int k, int lane, unsigned long long* keys, double* vec,
                                 unsigned char* first) {
    const int npos = L - k;                       // positions of order-(k+1) n-grams
    if (lane < CIDER_MAXL) { keys[lane] = 0; vec[lane] = 0.0; first[lane] = 0; }
    __syncthreads();
    if (lane < npos) {
        unsigned long long key = 0;
        for (int j = 0; j <= k; ++j) key |= (unsigned long long)(unsigned)s[lane + j] << (16 * j);
        keys[lane] = key;
    }
    __syncthreads();
    if (lane < npos) {
        int tf = 0;
        bool fst = true;
        for (int j = 0; j < npos; ++j)
            if (keys[j] == keys[lane]) { ++tf; if (j < lane) fst = false; }
        first[lane] = fst ? 1 : 0;
        vec[lane] = fst ? (double)tf * cider_idf(tb, k, keys[lane]) : 0.0;
    }
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_cider(const CiderTable tb, const int64_t* __restrict__ cand, int T,
                                              const int64_t* __restrict__ refs, int n_ref, int Tr, long long eos, long long pad,
                                              const unsigned char* __restrict__ drop, int V, double sigma, float* __restrict__ out) {
    __shared__ int hs[CIDER_MAXL], rs[CIDER_MAXL];
    __shared__ unsigned long long hk[CIDER_MAXL], rk[CIDER_MAXL];
    __shared__ double hv[CIDER_MAXL], rv[CIDER_MAXL];
    __shared__ unsigned char hf[CIDER_MAXL], rf[CIDER_MAXL];
    __shared__ int Lh, Lr;
    __shared__ double total;
    const int n = blockIdx.x, lane = threadIdx.x;
    if (lane == 0) { Lh = cider_clean(cand + (long long)n * T, T, eos, pad, drop, V, hs); total = 0.0; }
    __syncthreads();
    for (int r = 0; r < n_ref; ++r) {
        if (lane == 0) Lr = cider_clean(refs + ((long long)n * n_ref + r) * Tr, Tr, eos, pad, drop, V, rs);
        __syncthreads();
        const double delta = (double)((Lh > 1 ? Lh - 1 : 0) - (Lr > 1 ? Lr - 1 : 0));     // lengths in bigrams
        const double pen = exp(-(delta * delta) / (2.0 * sigma * sigma));
        for (int k = 0; k < 4; ++k) {
            cider_vec(tb, hs, Lh, k, lane, hk, hv, hf);
            cider_vec(tb, rs, Lr, k, lane, rk, rv, rf);
            if (lane == 0) {
                double nh = 0.0, nr = 0.0, val = 0.0;
                const int ph = Lh - k, pr = Lr - k;
                for (int i = 0; i < ph; ++i) if (hf[i]) nh += hv[i] * hv[i];
                for (int j = 0; j < pr; ++j) if (rf[j]) nr += rv[j] * rv[j];
                for (int i = 0; i < ph; ++i) {
                    if (!hf[i]) continue;
                    for (int j = 0; j < pr; ++j)
                        if (rf[j] && rk[j] == hk[i]) { val += fmin(hv[i], rv[j]) * rv[j]; break; }
                }
                nh = sqrt(nh); nr = sqrt(nr);
                if (nh != 0.0 && nr != 0.0) val /= nh * nr;
                total += val * pen;
            }
            __syncthreads();
        }
    }
    if (lane == 0) out[n] = (float)(10.0 * total / 4.0 / (double)n_ref);
}

extern "C" int vsr_cider_rewards(const uint64_t* const* keys, const double* const* idf, const int32_t* counts, double ref_len,
                                 const int64_t* cand, int32_t N, int32_t T, const int64_t* refs, int32_t n_ref, int32_t Tr,
                                 int64_t eos, int64_t pad, const uint8_t* drop, int32_t V, double sigma, float* rewards, void* stream) {
    if (!keys || !idf || !counts || !cand || !refs || !rewards) return fail("vsr_cider_rewards: null argument");
    if (N <= 0 || T <= 0 || n_ref <= 0 || Tr <= 0) return fail("vsr_cider_rewards: empty batch");
    if (T > CIDER_MAXL || Tr > CIDER_MAXL) return fail("vsr_cider_rewards: captions longer than %d tokens", CIDER_MAXL);
    if (V <= 0 || V > 65535) return fail("vsr_cider_rewards: vocabulary of %d ids does not fit the 16-bit n-gram key", V);
    CiderTable tb;
    for (int k = 0; k < 4; ++k) {
        tb.keys[k] = reinterpret_cast<const unsigned long long*>(keys[k]);
        tb.idf[k] = idf[k];
        tb.count[k] = counts[k];
    }
    tb.ref_len = ref_len;
    hipLaunchKernelGGL(k_cider, dim3(N), dim3(64), 0, (hipStream_t)stream, tb, cand, T, refs, n_ref, Tr, (long long)eos, (long long)pad, drop,
                       V, sigma, rewards);
    LAUNCHCHK();
    return 0;
}

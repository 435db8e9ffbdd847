/* ORACLE (test infrastructure, never shipped): sequential C restatement of the reference's
 * dense-CRF (tools/torchCRF, CUDA-only: cannot be compiled or run in this image).
 *
 * PARITY UNPINNED: the reference holds no golden vectors for this path and its CUDA source
 * cannot run here, so this file is pinned only by construction (line-by-line citations
 * below) and by the invariants in tests/test_crf_oracle.py.  See DESIGN.md.
 *
 * What is restated (paths under /root/reference/tools/torchCRF):
 *   src/torchcrf.cu:26-51,90-103,106-143   potentials added when weight>0 && sigma>0, MAP copy
 *   src/densecrf_base.cpp:15-46            start (softmax(-U)), T x (next=-U; next+=w*filter(Q); softmax)
 *   src/densecrf_gpu.cu:40-72,84-108,145-164,179-190   softmax w/ max-subtraction, hard-label unary,
 *                                          strict '>' argmax, negate
 *   src/pairwise_gpu.cu:10-36,110-116      features (x/sxy, y/sxy, rgb/srgb), out += w * filtered
 *   src/permutohedral_gpu.cu:76-166        hash (mul-add 2531011, mod 2*capacity, linear probing)
 *   src/permutohedral_gpu.cu:169-275       elevate / round / rank / barycentric / keys
 *   src/permutohedral_gpu.cu:303-378       splat (incl. homogeneous channel)
 *   src/permutohedral_gpu.cu:381-424       blur 1/4-1/2-1/4, axis 0..pd, missing neighbour = 0
 *   src/permutohedral_gpu.cu:427-451       slice + normalisation by the homogeneous channel
 *   src/permutohedral_gpu.cu:454-467       scale factors
 * Deliberate deviations: `expNormKernel` blends with the uninitialised output
 * ((1-relax)*out, relax=1) -- here the output is simply overwritten; insertion is sequential so
 * there is no duplicate-key race and no cleanHashTable pass; float sums run in a fixed order
 * (the CUDA atomics make the reference itself non-deterministic at the ulp level).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MLAB 2           /* torchcrf.cu:16  binary CRF */
#define VD (MLAB + 1)
#define PD_MAX 5

typedef struct {
    int pd, n, capacity;  /* capacity = n*(pd+1), table has 2*capacity buckets */
    int *entries;         /* bucket -> slot, -1 empty */
    int16_t *keys;        /* slot*pd */
    int *mat_index;       /* n*(pd+1): canonical slot per (pixel, vertex) */
    float *mat_weight;    /* n*(pd+1): barycentric weight */
    int *canon;           /* list of canonical slots, in creation order */
    int ncanon;
    int *nb_plus, *nb_minus; /* per canonical slot and axis: neighbour slot or -1 (computed once) */
    float *values, *new_values;
    float w;              /* Potts weight */
    float *norm;          /* symmetric mode only: 1 / sqrt(filter(1) + 1e-20) per pixel (see lattice_filter) */
} lattice_t;

static unsigned int key_hash(const int16_t *key, int pd) {
    unsigned int k = 0;                                  /* permutohedral_gpu.cu:83-92 */
    for (int i = 0; i < pd; i++) { k += (unsigned int)(int)key[i]; k = k * 2531011u; }
    return k;
}

static int table_insert(lattice_t *L, const int16_t *key, int slot) {
    const int pd = L->pd, nb = 2 * L->capacity;          /* :96-139, sequential */
    int h = (int)(key_hash(key, pd) % (unsigned int)nb);
    for (;;) {
        int e = L->entries[h];
        if (e == -1) {
            memcpy(L->keys + (size_t)slot * pd, key, sizeof(int16_t) * pd);
            L->entries[h] = slot;
            return slot;
        }
        if (memcmp(L->keys + (size_t)e * pd, key, sizeof(int16_t) * pd) == 0) return e;
        if (++h == nb) h = 0;
    }
}

static int table_retrieve(const lattice_t *L, const int16_t *key) {
    const int pd = L->pd, nb = 2 * L->capacity;          /* :143-166 */
    int h = (int)(key_hash(key, pd) % (unsigned int)nb);
    for (;;) {
        int e = L->entries[h];
        if (e == -1) return -1;
        if (memcmp(L->keys + (size_t)e * pd, key, sizeof(int16_t) * pd) == 0) return e;
        if (++h == nb) h = 0;
    }
}

static void lattice_free(lattice_t *L) {
    free(L->entries); free(L->keys); free(L->mat_index); free(L->mat_weight); free(L->canon);
    free(L->nb_plus); free(L->nb_minus); free(L->values); free(L->new_values); free(L->norm);
    memset(L, 0, sizeof(*L));
}

/* features: n*pd floats.  Builds keys, barycentric weights and the neighbour lists. */
static int lattice_build(lattice_t *L, const float *feat, int n, int pd, float w) {
    memset(L, 0, sizeof(*L));
    L->pd = pd; L->n = n; L->capacity = n * (pd + 1); L->w = w;
    const size_t cap = (size_t)L->capacity;
    L->entries = (int *)malloc(sizeof(int) * 2 * cap);
    L->keys = (int16_t *)calloc(cap * pd, sizeof(int16_t));
    L->mat_index = (int *)malloc(sizeof(int) * cap);
    L->mat_weight = (float *)malloc(sizeof(float) * cap);
    L->canon = (int *)malloc(sizeof(int) * cap);
    if (!L->entries || !L->keys || !L->mat_index || !L->mat_weight || !L->canon) return -1;
    memset(L->entries, 0xff, sizeof(int) * 2 * cap);

    float scale[PD_MAX];                                  /* :454-461 */
    const float inv_std = (pd + 1) * sqrtf(2.0f / 3);
    for (int i = 0; i < pd; i++) scale[i] = 1.0f / (sqrtf((float)(i + 1) * (i + 2))) * inv_std;

    for (int idx = 0; idx < n; idx++) {                   /* createLattice :169-275 */
        float elevated[PD_MAX + 1];
        int rem0[PD_MAX + 1], rank[PD_MAX + 1];
        const float *pos = feat + (size_t)idx * pd;
        float sm = 0;
        for (int i = pd; i > 0; i--) {
            float cf = pos[i - 1] * scale[i - 1];
            elevated[i] = sm - i * cf;
            sm += cf;
        }
        elevated[0] = sm;
        short sum = 0;
        for (int i = 0; i <= pd; i++) {
            float v = (float)(elevated[i] * (1.0 / (pd + 1)));
            float up = ceilf(v) * (pd + 1);
            float down = floorf(v) * (pd + 1);
            rem0[i] = (up - elevated[i] < elevated[i] - down) ? (short)up : (short)down;
            sum = (short)(sum + rem0[i]);
        }
        sum = (short)(sum / (pd + 1));
        for (int i = 0; i <= pd; i++) rank[i] = 0;
        for (int i = 0; i < pd; i++) {
            double di = elevated[i] - rem0[i];
            for (int j = i + 1; j <= pd; j++) {
                if (di < elevated[j] - rem0[j]) rank[i]++; else rank[j]++;
            }
        }
        for (int i = 0; i <= pd; i++) {
            rank[i] += sum;
            if (rank[i] < 0) { rank[i] += pd + 1; rem0[i] += pd + 1; }
            else if (rank[i] > pd) { rank[i] -= pd + 1; rem0[i] -= pd + 1; }
        }
        float bary[PD_MAX + 2];
        for (int i = 0; i <= pd + 1; i++) bary[i] = 0;
        for (int i = 0; i <= pd; i++) {
            float delta = (float)((elevated[i] - rem0[i]) * (1.0 / (pd + 1)));
            bary[pd - rank[i]] += delta;
            bary[pd + 1 - rank[i]] -= delta;
        }
        bary[0] = (float)(bary[0] + (1.0 + bary[pd + 1]));
        for (int r = 0; r <= pd; r++) {
            int16_t key[PD_MAX];
            for (int i = 0; i < pd; i++) {
                key[i] = (int16_t)(rem0[i] + r);
                if (rank[i] > pd - r) key[i] = (int16_t)(key[i] - (pd + 1));
            }
            int slot = idx * (pd + 1) + r;
            int c = table_insert(L, key, slot);
            if (c == slot) L->canon[L->ncanon++] = slot;
            L->mat_index[slot] = c;
            L->mat_weight[slot] = bary[r];
        }
    }
    /* neighbour slots per axis (what blur's two hash retrieves resolve to, :392-407) */
    const size_t nc = (size_t)L->ncanon;
    L->nb_plus = (int *)malloc(sizeof(int) * nc * (pd + 1));
    L->nb_minus = (int *)malloc(sizeof(int) * nc * (pd + 1));
    L->values = (float *)calloc(cap * VD, sizeof(float));
    L->new_values = (float *)calloc(cap * VD, sizeof(float));
    if (!L->nb_plus || !L->nb_minus || !L->values || !L->new_values) return -1;
    for (size_t ci = 0; ci < nc; ci++) {
        const int16_t *k = L->keys + (size_t)L->canon[ci] * pd;
        for (int color = 0; color <= pd; color++) {
            int16_t np[PD_MAX], nm[PD_MAX];
            for (int i = 0; i < pd; i++) { np[i] = (int16_t)(k[i] + 1); nm[i] = (int16_t)(k[i] - 1); }
            if (color < pd) { np[color] = (int16_t)(np[color] - (pd + 1)); nm[color] = (int16_t)(nm[color] + (pd + 1)); }
            L->nb_plus[ci * (pd + 1) + color] = table_retrieve(L, np);
            L->nb_minus[ci * (pd + 1) + color] = table_retrieve(L, nm);
        }
    }
    return 0;
}

/* out[n*M] = filter(in[n*M])  (permutohedral_gpu.cu:551-573): normalised by the filtered homogeneous channel.
 *
 * L->norm != NULL selects the OTHER normalisation this repo needs: the symmetric one of Kraehenbuehl & Koltun's
 * DenseCRF (densecrf/src/pairwise.cpp, DenseKernel::filter with NORMALIZE_SYMMETRIC -- the default of
 * pydensecrf's DenseCRF2D.addPairwiseBilateral, which tools/pydenseCRF/crf.py:73 and models/crf_head.py:82 call):
 *     out = N^(1/2) K N^(1/2) in,   N = diag(1 / (K 1 + 1e-20)),   K = the unnormalised lattice filter
 * (DenseKernel::initLattice: norm_ = lattice_.compute(ones); norm_[i] = 1 / sqrt(norm_[i] + 1e-20);
 *  filter: out = in * norm_; lattice_.compute(out, out); out = out * norm_).  Constant factors of K (blur taps
 * 1/4-1/2-1/4 here against 1/2-1-1/2 and the slice's alpha there) cancel between K and N.  pydensecrf is not
 * vendored in the reference (requirements.txt:8, unpinned git HEAD): this branch restates the published algorithm and
 * is PARITY-UNPINNED (no reference vectors exist for it). */
static void lattice_filter(lattice_t *L, float *out, const float *in) {
    const int pd = L->pd, n = L->n;
    const float *nrm = L->norm;
    float *val = L->values, *nv = L->new_values;
    memset(val, 0, sizeof(float) * (size_t)L->capacity * VD);
    for (int p = 0; p < n; p++) {                          /* splat :303-378 */
        for (int r = 0; r <= pd; r++) {
            const int e = p * (pd + 1) + r;
            float wgt = L->mat_weight[e];
            float *v = val + (size_t)L->mat_index[e] * VD;
            for (int j = 0; j < MLAB; j++) v[j] += (nrm ? in[(size_t)p * MLAB + j] * nrm[p] : in[(size_t)p * MLAB + j]) * wgt;
            v[VD - 1] += wgt;
        }
    }
    for (int color = 0; color <= pd; color++) {            /* blur :381-424 */
        for (int ci = 0; ci < L->ncanon; ci++) {
            const int s = L->canon[ci];
            const int a = L->nb_plus[(size_t)ci * (pd + 1) + color];
            const int b = L->nb_minus[(size_t)ci * (pd + 1) + color];
            for (int j = 0; j < VD; j++) {
                float vp = a >= 0 ? val[(size_t)a * VD + j] : 0.0f;
                float vm = b >= 0 ? val[(size_t)b * VD + j] : 0.0f;
                nv[(size_t)s * VD + j] = (float)(0.25 * vp + 0.5 * val[(size_t)s * VD + j] + 0.25 * vm);
            }
        }
        float *t = val; val = nv; nv = t;
    }
    L->values = val; L->new_values = nv;
    for (int p = 0; p < n; p++) {                          /* slice :427-451 */
        float acc[MLAB] = {0}, wsum = 0;
        for (int r = 0; r <= pd; r++) {
            const int e = p * (pd + 1) + r;
            const float *v = val + (size_t)L->mat_index[e] * VD;
            for (int j = 0; j < MLAB; j++) acc[j] += L->mat_weight[e] * v[j];
            wsum += L->mat_weight[e] * v[VD - 1];
        }
        wsum = nrm ? nrm[p] : (float)(1.0 / wsum);
        for (int j = 0; j < MLAB; j++) out[(size_t)p * MLAB + j] = acc[j] * wsum;
    }
}

/* symmetric mode: norm[p] = 1 / sqrt((K 1)[p] + 1e-20), from the homogeneous channel of one plain filter pass */
static int lattice_make_symmetric(lattice_t *L) {
    const int pd = L->pd, n = L->n;
    float *zeros = (float *)calloc((size_t)n * MLAB, sizeof(float)), *tmp = (float *)malloc(sizeof(float) * (size_t)n * MLAB);
    float *norm = (float *)malloc(sizeof(float) * (size_t)n);
    if (!zeros || !tmp || !norm) { free(zeros); free(tmp); free(norm); return -1; }
    lattice_filter(L, tmp, zeros);                       /* leaves the blurred homogeneous channel in L->values */
    for (int p = 0; p < n; p++) {
        float wsum = 0;
        for (int r = 0; r <= pd; r++) {
            const int e = p * (pd + 1) + r;
            wsum += L->mat_weight[e] * L->values[(size_t)L->mat_index[e] * VD + (VD - 1)];
        }
        norm[p] = (float)(1.0 / sqrt((double)wsum + 1e-20));
    }
    L->norm = norm;
    free(zeros); free(tmp);
    return 0;
}

static void exp_normalize(float *out, const float *in, int n, float scale) {
    for (int p = 0; p < n; p++) {                          /* densecrf_gpu.cu:40-72 */
        const float *b = in + (size_t)p * MLAB;
        float mx = scale * b[0];
        for (int j = 1; j < MLAB; j++) if (mx < scale * b[j]) mx = scale * b[j];
        float V[MLAB], tt = 0.0f;
        for (int j = 0; j < MLAB; j++) { V[j] = expf(scale * b[j] - mx); tt += V[j]; }
        for (int j = 0; j < MLAB; j++) out[(size_t)p * MLAB + j] = V[j] / tt;
    }
}

static void image_features(float *out, int pd, int W, int H, const float *rgb, float posdev, float featdev) {
    for (int hi = 0; hi < H; hi++)                         /* pairwise_gpu.cu:21-36 */
        for (int wi = 0; wi < W; wi++) {
            const size_t idx = (size_t)hi * W + wi;
            out[idx * pd + 0] = (float)wi / posdev;
            out[idx * pd + 1] = (float)hi / posdev;
            for (int i = 2; i < pd; i++) out[idx * pd + i] = rgb[idx * (pd - 2) + (i - 2)] / featdev;
        }
}

/* Shared inference. unary: n*2 energies.  Optional outputs may be NULL.
 * num_vertices[0] = smoothness lattice size (0 if off), [1] = appearance lattice size. */
static int g_symmetric = 0;     /* set around a call by crf_ref_soft_symmetric */
static int crf_run(const float *rgbf, const float *unary, int W, int H, float scomp_smooth, float sxy_smooth,
                   float scomp_app, float sxy_app, float srgb_app, int iters, int16_t *out_map, float *out_q,
                   int *num_vertices) {
    const int n = W * H;
    lattice_t Ls, La;
    int has_s = 0, has_a = 0, rc = 0;
    float *feat = (float *)malloc(sizeof(float) * (size_t)n * 5);
    float *cur = (float *)malloc(sizeof(float) * (size_t)n * MLAB);
    float *nxt = (float *)malloc(sizeof(float) * (size_t)n * MLAB);
    float *tmp = (float *)malloc(sizeof(float) * (size_t)n * MLAB);
    if (!feat || !cur || !nxt || !tmp) { rc = -1; goto done; }
    if (scomp_smooth > 0.0f && sxy_smooth > 0.0f) {        /* torchcrf.cu:26-37 */
        image_features(feat, 2, W, H, NULL, sxy_smooth, 1.0f);
        if (lattice_build(&Ls, feat, n, 2, scomp_smooth)) { rc = -1; goto done; }
        has_s = 1;
        if (g_symmetric && lattice_make_symmetric(&Ls)) { rc = -1; goto done; }
    }
    if (scomp_app > 0.0f && sxy_app > 0.0f) {              /* torchcrf.cu:39-51 */
        image_features(feat, 5, W, H, rgbf, sxy_app, srgb_app);
        if (lattice_build(&La, feat, n, 5, scomp_app)) { rc = -1; goto done; }
        has_a = 1;
        if (g_symmetric && lattice_make_symmetric(&La)) { rc = -1; goto done; }
    }
    if (num_vertices) { num_vertices[0] = has_s ? Ls.ncanon : 0; num_vertices[1] = has_a ? La.ncanon : 0; }
    exp_normalize(cur, unary, n, -1.0f);                   /* densecrf_base.cpp:31-34 */
    for (int it = 0; it < iters; it++) {                   /* :36-46 */
        for (size_t i = 0; i < (size_t)n * MLAB; i++) nxt[i] = -unary[i];
        if (has_s) { lattice_filter(&Ls, tmp, cur); for (size_t i = 0; i < (size_t)n * MLAB; i++) nxt[i] += Ls.w * tmp[i]; }
        if (has_a) { lattice_filter(&La, tmp, cur); for (size_t i = 0; i < (size_t)n * MLAB; i++) nxt[i] += La.w * tmp[i]; }
        exp_normalize(cur, nxt, n, 1.0f);
    }
    for (int p = 0; p < n; p++) {                          /* densecrf_gpu.cu:145-164 */
        const float *q = cur + (size_t)p * MLAB;
        float mx = q[0]; int16_t im = 0;
        for (int16_t m = 1; m < MLAB; m++) if (mx < q[m]) { mx = q[m]; im = m; }
        out_map[p] = im;
    }
    if (out_q) memcpy(out_q, cur, sizeof(float) * (size_t)n * MLAB);
done:
    if (has_s) lattice_free(&Ls);
    if (has_a) lattice_free(&La);
    free(feat); free(cur); free(nxt); free(tmp);
    return rc;
}

/* torchcrfSoft (torchcrf.cu:126-143).  rgb: H*W*3 floats (the binding converts any dtype to f32). */
int crf_ref_soft(const float *rgb, const float *unary, int W, int H, float scomp_smooth, float sxy_smooth,
                 float scomp_app, float sxy_app, float srgb_app, int iters, int16_t *out_map, float *out_q,
                 int *num_vertices) {
    return crf_run(rgb, unary, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters, out_map,
                   out_q, num_vertices);
}

/* DenseCRF2D(W, H, 2) + setUnaryEnergy + addPairwiseGaussian / addPairwiseBilateral (symmetric normalisation, the
 * pydensecrf default) + inference(iters): Q <- softmax(-U + sum_k w_k * Ksym_k Q)  (densecrf.cpp DenseCRF::inference,
 * pairwise.cpp PottsCompatibility::apply).  Same argument meaning as crf_ref_soft.  See lattice_filter. */
int crf_ref_soft_symmetric(const float *rgb, const float *unary, int W, int H, float scomp_smooth, float sxy_smooth,
                           float scomp_app, float sxy_app, float srgb_app, int iters, int16_t *out_map, float *out_q,
                           int *num_vertices) {
    g_symmetric = 1;
    int rc = crf_run(rgb, unary, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters, out_map, out_q,
                     num_vertices);
    g_symmetric = 0;
    return rc;
}

/* torchcrfHard (torchcrf.cu:106-124) with setUnaryEnergyFromLabel (densecrf_gpu.cu:84-143). */
int crf_ref_hard(const float *rgb, const int16_t *label, int W, int H, float scomp_smooth, float sxy_smooth,
                 float scomp_app, float sxy_app, float srgb_app, float confidence, int iters, int16_t *out_map,
                 float *out_q, int *num_vertices) {
    const int n = W * H;
    float *unary = (float *)malloc(sizeof(float) * (size_t)n * MLAB);
    if (!unary) return -1;
    const float u_energy = -logf(1.0f / MLAB);
    const float n_energy = -logf((1.0f - confidence) / (MLAB - 1));
    const float p_energy = -logf(confidence);
    for (int p = 0; p < n; p++) {
        int16_t l = label[p];
        for (int m = 0; m < MLAB; m++) unary[(size_t)p * MLAB + m] = (l == -1) ? u_energy : n_energy;
        if (l != -1) unary[(size_t)p * MLAB + l] = p_energy;
    }
    int rc = crf_run(rgb, unary, W, H, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters, out_map,
                     out_q, num_vertices);
    free(unary);
    return rc;
}

/* Exposes the lattice itself for parity tests: sorted-key comparison and weights.
 * keys_out: n*(pd+1)*pd int16 (per pixel-vertex key), weights_out: n*(pd+1). Returns #vertices. */
int crf_ref_lattice(const float *feat, int n, int pd, int16_t *keys_out, float *weights_out) {
    lattice_t L;
    if (lattice_build(&L, feat, n, pd, 1.0f)) return -1;
    for (int e = 0; e < n * (pd + 1); e++) {
        memcpy(keys_out + (size_t)e * pd, L.keys + (size_t)L.mat_index[e] * pd, sizeof(int16_t) * pd);
        weights_out[e] = L.mat_weight[e];
    }
    int nv = L.ncanon;
    lattice_free(&L);
    return nv;
}

/* out = filter(in) on the appearance lattice only (invariant tests: constants are preserved). */
int crf_ref_filter(const float *rgb, int W, int H, float sxy, float srgb, const float *in, float *out) {
    const int n = W * H;
    lattice_t L;
    float *feat = (float *)malloc(sizeof(float) * (size_t)n * 5);
    if (!feat) return -1;
    image_features(feat, 5, W, H, rgb, sxy, srgb);
    if (lattice_build(&L, feat, n, 5, 1.0f)) { free(feat); return -1; }
    lattice_filter(&L, out, in);
    lattice_free(&L);
    free(feat);
    return 0;
}

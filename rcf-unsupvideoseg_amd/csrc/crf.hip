// Dense-CRF mean-field inference on the permutohedral lattice, batched over frames.
// Stands in for torchcrf_cpp (tools/torchCRF/src/torchcrf.cu:106-149): DenseCRFGPU<2> with Potts
// potentials over a PermutohedralLatticeGPU<float, 5 or 2, 3>.  Algorithm restated from
//   src/permutohedral_gpu.cu:169-275 (embedding, rounding, rank, barycentric, int16 keys), :76-92
//   (hash), :303-378 (splat), :381-424 (blur), :427-451 (slice), :454-467 (scale factors);
//   src/pairwise_gpu.cu:10-36; src/densecrf_gpu.cu:40-72,84-108,145-190; src/densecrf_base.cpp:15-46.
//
// MI355X design (HBM / L2 gather-scatter bound, no MFMA):
//  * build once per frame: every (pixel, vertex) entry writes its key, then claims a hash bucket with
//    ONE device-scope atomicCAS whose payload is a slot whose key is already globally visible (written
//    by the previous kernel) -- no lock states, no duplicate keys, hence no cleanHashTable pass;
//    atomicMin makes the bucket's representative the smallest slot => vertex numbering (hand-written
//    3-kernel scan) is deterministic.  Blur neighbours are resolved ONCE into index lists instead of
//    2 hash probes per vertex per axis per iteration.
//    The (pixel, vertex) entries are also grouped by vertex once (count -> scan -> fill: a CSR list).
//  * per iteration: splat is a GATHER, one wavefront per vertex over its CSR list, accumulating in
//    64-bit fixed point (2^-40) so the sum is independent of the list order: run-to-run bit-identical
//    MAPs and no atomics at all in the iteration (the reference does 3 float atomics per entry per
//    iteration); 6 blur passes stream float4 vertex values; slice + Potts weight + softmax (+ MAP on
//    the last iteration) are one kernel.
//    Round 6: frames whose 16 x 16 pixel tiles share most of their vertices (natural images) take the splat as per-tile sums in
//    LDS + a vertex pass over a vertex's few, contiguous per-tile partial sums, and build no CSR list (see Lattice::tslot,
//    splat_tile_kernel, slice_kernel SPLAT) -- still integer sums, the same bits.
//  * persistent caller-owned workspace, all frames of a batch in every launch (grid.y = frame).
//  Algorithmic bytes per iteration: 192*N + 348*L, L = lattice vertices.
#include "rcf_common.h"

namespace {

constexpr int MLAB = 2;
constexpr int PD_MAX = 5;
constexpr double FIX_SCALE = 1099511627776.0;   // 2^40

struct Lattice {           // device pointers of one potential, for all frames (frame stride in elements)
    int pd;
    int N;                 // pixels per frame
    int W;                 // image width (N = W * H): the per-pixel kernels walk 16 x 16 tiles
    long E;                // entries per frame = N*(pd+1)
    float w;               // Potts weight
    int tune;              // per call (rcf_crf_soft_ex `normalization` bits 10-15): bit 0 RCF_CRF_BLUR_SEQUENTIAL (one launch per blur
                           // axis instead of one per PAIR of axes), bits 1-3 the vertex kernels' grid (lab: 0 = the default)
    int build;             // per call (rcf_crf_soft_ex `normalization` bits): 0 packed build when the keys fit, 1 always the
                           // array-of-keys build, 2 packed build whose first-attempt table is tiny (exercises the overflow path)
    uint4 *keys;           // [F][E]   5 x int16 packed, zero padded
    float *weight;         // [F][E]
    int *entries;          // [F][2E]  bucket -> representative slot (-1 empty)
    int *vid;              // [F][E]   bucket index after insert, then dense vertex id
    int *slot_vid;         // [F][E]   scan scratch: vertex id of representative slots
    int *vrep;             // [F][E]   vertex -> representative slot
    int *nb;               // [F][pd+1][E][2] neighbour vertex ids (-1 = none): axis-major, so that a blur pass reads 8 contiguous
                           // bytes per vertex (vertex-major [E][2(pd+1)] made every pass fetch all 48 bytes of a vertex's lists)
    int *cnt;              // [F][E]   entries per vertex, then fill cursor
    int *off;              // [F][E]   CSR offsets (exclusive scan of cnt)
    int2 *csr;             // [F][E]   (pixel, weight bits) grouped by vertex
    float2 *val0, *val1;   // [F][E]   ping-pong blurred label values (float [F][E] for the homogeneous channel at build time)
    float *inv;            // [F][N]   1 / (sliced homogeneous channel): iteration invariant, computed at build time
    int *blocksum;         // [F][nblk+1]
    int *L;                // [F]      vertex counts
    // packed build (12-bit key coordinates): the hash table holds the 64-bit keys themselves
    unsigned long long *table;   // [F][2E]  (aliases `keys`)
    int *rel;                    // [F][E]   position of the entry inside its vertex's CSR list
    int *slot_vid2;              // [F][2E]  bucket -> dense vertex id (-1 empty)
    int *slot_off;               // [F][2E]  bucket -> CSR offset
    int *blocksum2;              // [F][2][nblk2+1]
    // The table is first tried with pk_cap(f) buckets -- `cap_small` (a natural 480x854 frame has ~7e4 distinct keys: the table
    // then stays in the frame's L2 and the bucket scans are short) or, from the sampled estimate, a larger power of two; a
    // frame that fills more than half of them, or needs a probe longer than PK_PROBE_LIMIT, is inserted again into all 2E buckets.
    int cap_small;
    int est;                     // host side: pk_estimate_kernel ran for this build (stat[4 f + 2] sizes the first attempt: pk_cap)
    int *stat;                   // [F][4]  distinct keys of the first attempt, probe-limit flag, sampled distinct keys, total length of the tiles' lists
    int sym;                     // 1: DenseCRF2D's symmetric kernel normalisation (rcf_crf_soft_ex), set per call by crf_infer
    int norm_pending;            // host side: the normaliser of this lattice is still to come out of its first filter pass
    // sort build (build 3): rocPRIM's temporary storage, sized for F * E pairs
    char *sort_tmp;
    size_t sort_tmp_bytes;
    // rcf_crf_soft_f32: the colour features as the caller's floats [F][N][3] (torchcrf.cu:84-85 converts rgbFeat to float
    // unrounded); nullptr = the u8 image.  Set per call by crf_infer.
    const float *featf;
    // tile splat (packed build, frames whose 16 x 16 pixel tiles share most of their vertices: natural images): a workgroup adds
    // its tile's 256 (pd + 1) entries into the tile's short list of distinct vertices in LDS (64-bit fixed point: the sums do not
    // depend on the order) and stores one partial sum per distinct vertex; a vertex's partial sums -- one per tile that touches it
    // -- lie next to each other (the build numbers the tiles of a vertex with the same atomic that reserves its CSR range), and
    // the vertex pass adds them up: ~10x fewer values to gather than entries, no CSR list for such frames, the same bits.
    unsigned long long *cursor64;    // [F][2E]  per bucket: entries (low half) and tiles (high half) counted by the build
    unsigned short *tslot;       // [F][E]   the entry's position in its tile's list
    int *tlist;                  // [F][tiles * 256 (pd + 1)]  per tile and list position: the bucket, then (tile_list_kernel) the
                                 //          index of the partial sum this tile owns
    int *tpos;                   // [F][tiles * 256 (pd + 1)]  the tile's ordinal among the tiles of that vertex, then (tile_list_kernel)
                                 //          the vertex id (the slice stages the tile's vertex values in LDS through it)
    int *tcnt;                   // [F][tiles]  length of the tile's list
    int *tnew;                   // [F][tiles]  keys the tile was first to insert into the small table (overflow statistics)
    long long *accg;             // [F][3E]  partial sums in 2^-40 fixed point, grouped by vertex (Lt.off / Lt.cnt in tile mode):
                                 //          [E][2] label sums, then [E] homogeneous-channel sums
    int tiles;                   // 16 x 16 tiles per frame
    long Ep;                     // frame stride of `weight` and `tslot`: tiles * 256 * (pd + 1) >= E (whole tiles)
    int wtile;                   // host side: `weight` / `tslot` are TILE-major ((tile (pd + 1) + r) 256 + position in the 16 x 16 tile:
                                 // the packed build -- its kernels and every kernel of a filter pass walk the image in these tiles, and a
                                 // tile's 256 weights of one remainder are then one contiguous KB instead of sixteen 64-byte row pieces:
                                 // -7 % per pass); 0: remainder-major r N + p (array-of-keys and sort builds)
    int tile_splat;              // host side: this build wrote the tile lists (stat[4 f + 3] = their total length per frame)
};

constexpr int PK_PROBE_LIMIT = 256;
// pk_estimate_kernel: PK_SAMPLES blocks of 256 pixels per frame, spread over the image; the number of keys that are distinct
// within their block (noise-like content: ~1 key per entry; clean natural content: a third) sizes the frame's table (pk_cap below)
// -- without it every workgroup of a noise frame's first attempt probed a table that was 10x over-subscribed (uniform-noise frames:
// 12.7 ms of the 25 ms build per 8 frames, round 3).
constexpr int PK_SAMPLES = 64;
// Buckets of frame f's FIRST attempt (round 6).  Rounds 3-5 knew two sizes: 2^18 for natural frames, all 2E for the rest -- and the
// rest began at 1.3 x 10^5 vertices, i.e. at natural frames with +-4 of sensor noise, which then cleared, filled, scanned and probed a
// 4.9 M-bucket table (480x854: 1.7 of a call's 3.2 ms at +-8).  Now the size follows the sampled estimate: s = the fraction of the
// sampled entries that are distinct within their 256-pixel block; the vertices per entry of a frame stay under an envelope of s
// (measured over natural-looking frames with +-0 ... +-64 of pixel noise, profiles/r06_crf_texture_sweep.txt: s 0.30 -> 0.03-0.04,
// 0.6 -> 0.06-0.09, 0.8 -> 0.09-0.13, 0.87 -> 0.11-0.17, 0.94 -> 0.19-0.27, 0.97 -> 0.3-0.37); the table gets the next
// 2^k - 1 above twice that many buckets.  A frame that fills more than half of it, or probes longer than PK_PROBE_LIMIT, is redone in
// all 2E buckets as before.  No estimate (small frames; the tests' tiny table): cap_small.
__device__ __forceinline__ long pk_cap(const Lattice &Lt, int f) {
    const long full = 2 * Lt.E;
    if (!Lt.est) return (long)Lt.cap_small < full ? (long)Lt.cap_small : full;
    const long samp = (long)PK_SAMPLES * 256 * (Lt.pd + 1), S = Lt.stat[4 * f + 2];
    const long pct = S * 100 / samp;
    const double vpe = pct <= 37 ? 0.045 : pct <= 57 ? 0.075 : pct <= 80 ? 0.13 : pct <= 90 ? 0.175 : pct <= 95 ? 0.27 : 1.0;
    const long want = (long)(2.0 * vpe * (double)Lt.E);
    long cap = Lt.cap_small;
    while (cap < want) cap = 2 * cap + 1;
    return cap < full ? cap : full;
}
__device__ __forceinline__ bool pk_overflowed(const Lattice &Lt, int f) {
    const long cap = pk_cap(Lt, f);
    return cap < 2 * Lt.E && (Lt.stat[4 * f] > cap / 2 || Lt.stat[4 * f + 1] != 0);
}
__device__ __forceinline__ long pk_buckets(const Lattice &Lt, int f) { return pk_overflowed(Lt, f) ? 2 * Lt.E : pk_cap(Lt, f); }
// does frame f take the tile splat?  Its tiles' lists together hold at most 0.6 x as many vertices as the frame has entries.  Measured on
// natural-looking 480x854 frames with +-0 ... +-64 of pixel noise on top (tools/crf_texture_sweep.py, profiles/r06_crf_texture_sweep.txt):
// the tile splat beats the list walk while the lists total <= ~0.67 E (+-16: 0.655 against 0.684 ms per frame; +-20, 0.7 E: 0.82 against
// 0.81), by 35-40 % on clean frames (lists ~0.1 E); beyond, every entry is nearly its own vertex and the sort build is the one to take.
// With the sampled estimate (Lt.est: frames of >= 256 x 256 pixels) the decision is taken BEFORE the build from the sampled distinct
// fraction -- over that sweep lists of 0.6 E correspond to 88 % (+-16: 83-87.5 % and 0.56-0.59 E; +-24: 93-94.5 % and 0.78 E) -- so that the
// build writes only what the frame's mode reads (tile mode: no per-entry bucket / CSR position, 8 of 14 bytes per entry; list walk: no
// tile lists).  Without it, from the lists' exact total after the build.
// tune bit 4 (RCF_CRF_SPLAT_GATHER): never; bit 5 (RCF_CRF_SPLAT_TILES): whenever the lists exist (tests).
constexpr long TILE_SPLAT_NUM = 3, TILE_SPLAT_DEN = 5;
constexpr long TILE_SPLAT_EST_PCT = 88;
__device__ __forceinline__ bool tile_mode_known(const Lattice &Lt) { return Lt.est || (Lt.tune & (16 | 32)) || !Lt.tile_splat; }
__device__ __forceinline__ bool tile_mode(const Lattice &Lt, int f) {
    if (!Lt.tile_splat || (Lt.tune & 16)) return false;
    if (Lt.tune & 32) return true;
    if (Lt.est) return (long)Lt.stat[4 * f + 2] * 100 <= TILE_SPLAT_EST_PCT * ((long)PK_SAMPLES * 256 * (Lt.pd + 1));
    return (long)Lt.stat[4 * f + 3] * TILE_SPLAT_DEN <= TILE_SPLAT_NUM * Lt.E;
}
// where entry (remainder r, pixel p) of frame f lives in Lt.weight / Lt.tslot (see Lattice::wtile)
__device__ __forceinline__ long widx(const Lattice &Lt, int f, int r, int p) {
    const long base = (long)f * Lt.Ep;
    if (!Lt.wtile) return base + (long)r * Lt.N + p;
    const int py = p / Lt.W, px = p - py * Lt.W, tiles_x = (Lt.W + 15) >> 4;
    return base + ((long)((py >> 4) * tiles_x + (px >> 4)) * (Lt.pd + 1) + r) * 256 + (((py & 15) << 4) | (px & 15));
}
// ... for thread t of the workgroup that owns tile `tile` (pixel p): no divisions
__device__ __forceinline__ long widx_tile(const Lattice &Lt, int f, int tile, int r, int t, int p) {
    return (long)f * Lt.Ep + (Lt.wtile ? ((long)tile * (Lt.pd + 1) + r) * 256 + t : (long)r * Lt.N + p);
}

__device__ __forceinline__ unsigned key_hash(const short *key, int pd) {
    unsigned k = 0;
    for (int i = 0; i < pd; i++) {
        k += (unsigned)(int)key[i];
        k *= 2531011u;
    }
    return k;
}
__device__ __forceinline__ uint4 pack_key(const short *key, int pd) {
    unsigned short k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < pd; i++) k[i] = (unsigned short)key[i];
    return make_uint4(k[0] | ((unsigned)k[1] << 16), k[2] | ((unsigned)k[3] << 16), k[4] | ((unsigned)k[5] << 16), 0u);
}
__device__ __forceinline__ bool key_eq(const uint4 &a, const uint4 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// ---------------------------------------------------------------------------------- build: keys
// createLattice arithmetic for one pixel (permutohedral_gpu.cu:169-275), kept operation for operation: the lattice
// point's remainder-0 coordinates, the ranks and the pd+1 barycentric weights
__device__ __forceinline__ void lattice_point(int pd, int p, int f, int N, int W, const uint8_t *__restrict__ rgb,
                                              float posdev, float featdev, int (&rem0)[PD_MAX + 1],
                                              int (&rank)[PD_MAX + 1], float (&bary)[PD_MAX + 2],
                                              const float *__restrict__ featf = nullptr) {
    float pos[PD_MAX];
    const int wi = p % W, hi = p / W;
    pos[0] = (float)wi / posdev;
    pos[1] = (float)hi / posdev;
    if (pd == 5) {
        if (featf) {
            const float *c = featf + ((long)f * N + p) * 3;
            pos[2] = c[0] / featdev;
            pos[3] = c[1] / featdev;
            pos[4] = c[2] / featdev;
        } else {
            const uint8_t *c = rgb + ((long)f * N + p) * 3;
            pos[2] = (float)c[0] / featdev;
            pos[3] = (float)c[1] / featdev;
            pos[4] = (float)c[2] / featdev;
        }
    }
    float elevated[PD_MAX + 1];
    const float inv_std = (pd + 1) * sqrtf(2.0f / 3);
    float sm = 0;
    for (int i = pd; i > 0; i--) {
        const float scale = 1.0f / (sqrtf((float)(i) * (i + 1))) * inv_std;   // scaleFactor[i-1]
        const float cf = pos[i - 1] * scale;
        elevated[i] = sm - i * cf;
        sm += cf;
    }
    elevated[0] = sm;
    short sum = 0;
    for (int i = 0; i <= pd; i++) {
        const float v = (float)(elevated[i] * (1.0 / (pd + 1)));
        const float up = ceilf(v) * (pd + 1);
        const float down = floorf(v) * (pd + 1);
        rem0[i] = (up - elevated[i] < elevated[i] - down) ? (short)up : (short)down;
        sum = (short)(sum + rem0[i]);
    }
    sum = (short)(sum / (pd + 1));
    for (int i = 0; i <= pd; i++) rank[i] = 0;
    for (int i = 0; i < pd; i++) {
        const double di = elevated[i] - rem0[i];
        for (int j = i + 1; j <= pd; j++) {
            if (di < elevated[j] - rem0[j]) rank[i]++;
            else rank[j]++;
        }
    }
    for (int i = 0; i <= pd; i++) {
        rank[i] += sum;
        if (rank[i] < 0) { rank[i] += pd + 1; rem0[i] += pd + 1; }
        else if (rank[i] > pd) { rank[i] -= pd + 1; rem0[i] -= pd + 1; }
    }
    for (int i = 0; i <= pd + 1; i++) bary[i] = 0;
    for (int i = 0; i <= pd; i++) {
        const float delta = (float)((elevated[i] - rem0[i]) * (1.0 / (pd + 1)));
        bary[pd - rank[i]] += delta;
        bary[pd + 1 - rank[i]] -= delta;
    }
    bary[0] = (float)(bary[0] + (1.0 + bary[pd + 1]));
}
__device__ __forceinline__ void lattice_key(int pd, int r, const int (&rem0)[PD_MAX + 1], const int (&rank)[PD_MAX + 1],
                                            short *key) {
    for (int i = 0; i < pd; i++) {
        key[i] = (short)(rem0[i] + r);
        if (rank[i] > pd - r) key[i] = (short)(key[i] - (pd + 1));
    }
}

template <int PD>      // 2 / 5: the key arithmetic unrolls with a compile-time dimension; 0: Lt.pd at run time
__global__ void __launch_bounds__(256) lattice_keys_kernel(Lattice Lt, const uint8_t *__restrict__ rgb, int W, int H,
                                                           float posdev, float featdev) {
    const int pd = PD ? PD : Lt.pd;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= Lt.N) return;
    int rem0[PD_MAX + 1], rank[PD_MAX + 1];
    float bary[PD_MAX + 2];
    lattice_point(pd, p, f, Lt.N, W, rgb, posdev, featdev, rem0, rank, bary, Lt.featf);
    // entries are stored remainder-major (e = r*N + p): every later pass is coalesced along the pixels
    const long base = (long)f * Lt.E + p;
    for (int r = 0; r <= pd; r++) {
        short key[PD_MAX];
        lattice_key(pd, r, rem0, rank, key);
        Lt.keys[base + (long)r * Lt.N] = pack_key(key, pd);
        Lt.weight[(long)f * Lt.Ep + (long)r * Lt.N + p] = bary[r];
    }
}

__device__ __forceinline__ void unpack_key(const uint4 &k, short *key) {
    key[0] = (short)(k.x & 0xffff); key[1] = (short)(k.x >> 16);
    key[2] = (short)(k.y & 0xffff); key[3] = (short)(k.y >> 16);
    key[4] = (short)(k.z & 0xffff);
}

// ---------------------------------------------------------------------------------- wavefront run helpers
// Entries are laid out remainder-major (e = r*N + p), so the 64 lanes of a wavefront are 64 consecutive
// pixels of one lattice remainder and long runs of lanes share a vertex.  `head` flags the
// first lane of a run of equal items; run_head() / run_tail() give every lane its run's first / last lane.
__device__ __forceinline__ int run_head(bool head, int lane) {
    int seg = head ? lane : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int so = __shfl_up(seg, o, 64);
        if (lane >= o && so > seg) seg = so;
    }
    return seg;
}
__device__ __forceinline__ int run_tail(bool tail, int lane) {
    int t = tail ? lane : 63;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int to = __shfl_down(t, o, 64);
        if (lane + o < 64 && to < t) t = to;
    }
    return t;
}
__device__ __forceinline__ long entry_of(long idx, int, int) { return idx; }

// ---------------------------------------------------------------------------------- build: insert
__global__ void __launch_bounds__(256) lattice_insert_kernel(Lattice Lt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const bool live = idx < Lt.E;
    const long e = live ? entry_of(idx, Lt.N, Lt.pd + 1) : 0;
    const uint4 *keys = Lt.keys + (long)f * Lt.E;
    int *entries = Lt.entries + (long)f * 2 * Lt.E;
    uint4 mine = live ? keys[e] : make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, (unsigned)lane);
    if (!live) mine.w = 1u + (unsigned)lane;                  // dead lanes never join a run
    // only the first lane of a run of identical keys probes the table; e grows with the lane, so the run
    // head also holds the smallest slot of the run
    const uint4 prev = make_uint4(__shfl_up(mine.x, 1, 64), __shfl_up(mine.y, 1, 64), __shfl_up(mine.z, 1, 64),
                                  __shfl_up(mine.w, 1, 64));
    const bool head = lane == 0 || !(prev.x == mine.x && prev.y == mine.y && prev.z == mine.z && prev.w == mine.w);
    const int seg = run_head(head, lane);
    unsigned h = 0;
    if (live && head) {
        short key[8];
        unpack_key(mine, key);
        const unsigned nb = (unsigned)(2 * Lt.E);
        h = key_hash(key, Lt.pd) % nb;
        for (;;) {
            int cur = __hip_atomic_load(entries + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == -1) cur = atomicCAS(entries + h, -1, (int)e);
            if (cur == -1 || key_eq(keys[cur], mine)) break;   // claimed, or bucket already holds my key
            if (++h == nb) h = 0;
        }
        // representative = smallest slot with this key; blocks run roughly in slot order, so most runs find a
        // smaller slot already there and skip the atomic
        if (__hip_atomic_load(entries + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (int)e) atomicMin(entries + h, (int)e);
    }
    h = (unsigned)__shfl((int)h, seg, 64);
    if (live) Lt.vid[(long)f * Lt.E + e] = (int)h;
}

// ---------------------------------------------------------------------------------- build: numbering
constexpr int SCAN_ITEMS = 8, SCAN_BLOCK = 256, SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

__device__ __forceinline__ int block_exclusive_scan(int v, int *total) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < SCAN_BLOCK / 64; i++) {
        if (i < wv) off += wsum[i];
        tot += wsum[i];
    }
    __syncthreads();
    *total = tot;
    return off + inc - v;
}

// flag[e] = entry e is the representative of its bucket; local exclusive scan + block totals
__global__ void __launch_bounds__(SCAN_BLOCK) lattice_scan_local_kernel(Lattice Lt) {
    const int f = blockIdx.y;
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
    const int *bucket = Lt.vid + (long)f * Lt.E;
    const int *entries = Lt.entries + (long)f * 2 * Lt.E;
    int flags[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const long e = base + i;
        flags[i] = (e < Lt.E && entries[bucket[e]] == (int)e) ? 1 : 0;
        s += flags[i];
    }
    int total;
    int off = block_exclusive_scan(s, &total);
    int *out = Lt.slot_vid + (long)f * Lt.E;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const long e = base + i;
        if (e < Lt.E) out[e] = flags[i] ? off : -1;
        off += flags[i];
    }
    if (threadIdx.x == 0) Lt.blocksum[(long)f * (gridDim.x + 1) + blockIdx.x] = total;
}

__global__ void __launch_bounds__(SCAN_BLOCK) lattice_scan_blocks_kernel(Lattice Lt, int nblk) {
    const int f = blockIdx.x;
    int *bs = Lt.blocksum + (long)f * (nblk + 1);
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += SCAN_BLOCK) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblk ? bs[b] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        const int carry = carry_s;
        if (b < nblk) bs[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { bs[nblk] = carry_s; Lt.L[f] = carry_s; }
}

// slot_vid += block offset; vrep[vid] = slot; entry -> dense vertex id is resolved in the next kernel
__global__ void __launch_bounds__(SCAN_BLOCK) lattice_scan_apply_kernel(Lattice Lt) {
    const int f = blockIdx.y;
    const int off = Lt.blocksum[(long)f * (gridDim.x + 1) + blockIdx.x];
    int *sv = Lt.slot_vid + (long)f * Lt.E;
    int *vrep = Lt.vrep + (long)f * Lt.E;
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        const long e = base + i;
        if (e < Lt.E && sv[e] >= 0) {
            sv[e] += off;
            vrep[sv[e]] = (int)e;
        }
    }
}

__global__ void __launch_bounds__(256) lattice_entry_vid_kernel(Lattice Lt) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (e >= Lt.E) return;
    int *vid = Lt.vid + (long)f * Lt.E;
    const int rep = Lt.entries[(long)f * 2 * Lt.E + vid[e]];
    vid[e] = Lt.slot_vid[(long)f * Lt.E + rep];
}

// neighbour lists: what blur's two hash retrieves per axis resolve to (permutohedral_gpu.cu:392-407).
// u = v + e_axis  <=>  v = u - e_axis, so only the '+' neighbour is probed; the same thread also writes v as
// the '-' neighbour of u (each link has exactly one writer).
__global__ void __launch_bounds__(256) neighbours_init_kernel(Lattice Lt) {
    const int f = blockIdx.y;
    const long per_axis = 2L * Lt.L[f], total = per_axis * (Lt.pd + 1);
    int *nb = Lt.nb + (long)f * Lt.E * 2 * (Lt.pd + 1);
    if (per_axis == 0) return;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long axis = i / per_axis;
        nb[axis * 2 * Lt.E + (i - axis * per_axis)] = -1;        // [axis][vertex][+/-], axis stride 2E
    }
}
__global__ void __launch_bounds__(256) lattice_neighbours_kernel(Lattice Lt) {
    const int f = blockIdx.y;
    const int pd = Lt.pd, nax = pd + 1;
    const long Lf = Lt.L[f];
    const uint4 *keys = Lt.keys + (long)f * Lt.E;
    const int *entries = Lt.entries + (long)f * 2 * Lt.E;
    const int *sv = Lt.slot_vid + (long)f * Lt.E;
    int *nb = Lt.nb + (long)f * Lt.E * 2 * nax;
    const unsigned nbk = (unsigned)(2 * Lt.E);
    const long total = Lf * nax;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = i / nax;
        const int axis = (int)(i - v * nax);
        short key[8];
        unpack_key(keys[Lt.vrep[(long)f * Lt.E + v]], key);
        for (int k = 0; k < pd; k++) key[k] = (short)(key[k] + 1);
        if (axis < pd) key[axis] = (short)(key[axis] - (pd + 1));
        const uint4 want = pack_key(key, pd);
        unsigned h = key_hash(key, pd) % nbk;
        for (;;) {
            const int s = entries[h];
            if (s == -1) break;
            if (key_eq(keys[s], want)) {
                const int u = sv[s];
                nb[((long)axis * Lt.E + v) * 2] = u;              // v's '+' neighbour ([axis][vertex][+/-]: a blur pass reads 8 contiguous bytes per vertex)
                nb[((long)axis * Lt.E + u) * 2 + 1] = (int)v;     // u's '-' neighbour
                break;
            }
            if (++h == nbk) h = 0;
        }
    }
}

// ---------------------------------------------------------------------------------- build: CSR by vertex
__global__ void __launch_bounds__(256) csr_count_kernel(Lattice Lt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const bool live = idx < Lt.E;
    const long fb = (long)f * Lt.E;
    const int v = live ? Lt.vid[fb + entry_of(idx, Lt.N, Lt.pd + 1)] : -1 - lane;
    const int vp = __shfl_up(v, 1, 64), vn = __shfl_down(v, 1, 64);
    const bool head = lane == 0 || vp != v, tail = lane == 63 || vn != v;
    const int seg = run_head(head, lane);
    if (live && tail) atomicAdd(Lt.cnt + fb + v, lane - seg + 1);     // one atomic per run of equal vertices
}
// exclusive scan of cnt over [0, E) (entries beyond L are zero): local pass, block totals in blocksum
__global__ void __launch_bounds__(SCAN_BLOCK) csr_scan_local_kernel(Lattice Lt) {
    const int f = blockIdx.y;
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
    const int *cnt = Lt.cnt + (long)f * Lt.E;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        v[i] = (base + i < Lt.E) ? cnt[base + i] : 0;
        s += v[i];
    }
    int total;
    int off = block_exclusive_scan(s, &total);
    int *out = Lt.off + (long)f * Lt.E;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < Lt.E) out[base + i] = off;
        off += v[i];
    }
    if (threadIdx.x == 0) Lt.blocksum[(long)f * (gridDim.x + 1) + blockIdx.x] = total;
}
__global__ void __launch_bounds__(SCAN_BLOCK) csr_scan_blocks_kernel(Lattice Lt, int nblk) {
    const int f = blockIdx.x;
    int *bs = Lt.blocksum + (long)f * (nblk + 1);
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += SCAN_BLOCK) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblk ? bs[b] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        const int carry = carry_s;
        if (b < nblk) bs[b] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
}
// off += block offset; cnt is reset to serve as the fill cursor
__global__ void __launch_bounds__(SCAN_BLOCK) csr_scan_apply_kernel(Lattice Lt) {
    const int f = blockIdx.y;
    const int add = Lt.blocksum[(long)f * (gridDim.x + 1) + blockIdx.x];
    int *off = Lt.off + (long)f * Lt.E, *cnt = Lt.cnt + (long)f * Lt.E;
    const long base = (long)blockIdx.x * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++)
        if (base + i < Lt.E) { off[base + i] += add; cnt[base + i] = 0; }
}
__global__ void __launch_bounds__(256) csr_fill_kernel(Lattice Lt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const bool live = idx < Lt.E;
    const long fb = (long)f * Lt.E;
    const long e = live ? entry_of(idx, Lt.N, Lt.pd + 1) : 0;
    const int v = live ? Lt.vid[fb + e] : -1 - lane;
    const int vp = __shfl_up(v, 1, 64), vn = __shfl_down(v, 1, 64);
    const bool head = lane == 0 || vp != v, tail = lane == 63 || vn != v;
    const int seg = run_head(head, lane), tl = run_tail(tail, lane);
    int base = 0;
    if (live && tail) base = atomicAdd(Lt.cnt + fb + v, lane - seg + 1);   // the run reserves its slots at once
    base = __shfl(base, tl, 64);
    if (live) {
        // payload in CSR order: (pixel, barycentric weight).  The order inside a vertex is free: the
        // fixed-point sum of the gather does not depend on it.
        const int pos = Lt.off[fb + v] + base + (lane - seg);
        const int p = (int)(idx - (idx / Lt.N) * Lt.N);
        Lt.csr[fb + pos] = make_int2(p, __float_as_int(Lt.weight[(long)f * Lt.Ep + e]));
    }
}

// ---------------------------------------------------------------------------------- packed build
// When every key coordinate fits 12 bits (|coordinate| < 2048: sxy 60 / srgb 5 at 480x854 gives < 260) the five
// coordinates are the hash table's 64-bit value: one 64-bit CAS inserts a vertex and nothing ever re-reads a key
// array.  A workgroup of 256 pixels first de-duplicates its 1536 (pixel, remainder) entries in an LDS table --
// neighbouring pixels share most lattice vertices -- so only the distinct keys of the block touch the global table,
// and the same pass reserves each entry's slot in its vertex's CSR list (LDS rank inside the block + one global
// atomicAdd per distinct key), which removes the count pass and every atomic of the fill pass.
constexpr unsigned long long PK_EMPTY = ~0ull;
constexpr int LT_SLOTS = 2048;                      // >= 1536 entries of a block, power of two

__device__ __forceinline__ unsigned long long pack64(const short *key, int pd) {
    unsigned long long k = 0;
    for (int i = 0; i < pd; i++) k |= (unsigned long long)((unsigned)(key[i] + 2048) & 0xfffu) << (12 * i);
    return k;
}
__device__ __forceinline__ void unpack64(unsigned long long k, int pd, short *key) {
    for (int i = 0; i < pd; i++) key[i] = (short)((int)((k >> (12 * i)) & 0xfffu) - 2048);
}

// the per-frame statistics <- 0 (before the estimate, which the table's size depends on)
__global__ void pk_stat_reset_kernel(Lattice Lt, int F) {
    for (int i = threadIdx.x; i < 4 * F; i += blockDim.x) Lt.stat[i] = 0;
}
// buckets [0, n) of every frame's table <- empty (phase 0: n = the first attempt's size, pk_cap; phase 1: n = 2E, only for the
// frames whose first attempt overflowed)
__global__ void __launch_bounds__(256) pk_clear_kernel(Lattice Lt, int phase) {
    const int f = blockIdx.x;
    long n = pk_cap(Lt, f);
    if (phase == 1) {
        if (!pk_overflowed(Lt, f)) return;
        n = 2 * Lt.E;
    }
    unsigned long long *table = Lt.table + (long)f * 2 * Lt.E;
    unsigned long long *cursor = Lt.cursor64 + (long)f * 2 * Lt.E;
    for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.y * blockDim.x) {
        table[i] = PK_EMPTY;
        cursor[i] = 0ull;
    }
}

// distinct keys of PK_SAMPLES sampled pixel blocks (block-local de-duplication only) -> stat[4 f + 2]
template <int PD>      // 2 / 5: the key arithmetic unrolls with a compile-time dimension; 0: Lt.pd at run time
__global__ void __launch_bounds__(256) pk_estimate_kernel(Lattice Lt, const uint8_t *__restrict__ rgb, int W, float posdev,
                                                          float featdev) {
    __shared__ unsigned long long lkey[LT_SLOTS];
    __shared__ int distinct;
    const int pd = PD ? PD : Lt.pd, nax = pd + 1;
    const int f = blockIdx.y;
    const int nblk = (Lt.N + 255) / 256;
    const int blk = (int)(((long)blockIdx.x * nblk) / gridDim.x);
    const int p = blk * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < LT_SLOTS; i += blockDim.x) lkey[i] = PK_EMPTY;
    if (threadIdx.x == 0) distinct = 0;
    __syncthreads();
    int mine = 0;
    if (p < Lt.N) {
        int rem0[PD_MAX + 1], rank[PD_MAX + 1];
        float bary[PD_MAX + 2];
        lattice_point(pd, p, f, Lt.N, W, rgb, posdev, featdev, rem0, rank, bary, Lt.featf);
        for (int r = 0; r < nax; r++) {
            short key[PD_MAX];
            lattice_key(pd, r, rem0, rank, key);
            const unsigned long long k = pack64(key, pd);
            unsigned h = key_hash(key, pd) & (LT_SLOTS - 1);
            for (;;) {
                const unsigned long long prev = atomicCAS(&lkey[h], PK_EMPTY, k);
                if (prev == PK_EMPTY) ++mine;
                if (prev == PK_EMPTY || prev == k) break;
                h = (h + 1) & (LT_SLOTS - 1);
            }
        }
    }
    if (mine) atomicAdd(&distinct, mine);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(Lt.stat + 4 * f + 2, distinct);
}

template <int PD>      // 2 / 5: the key arithmetic unrolls with a compile-time dimension; 0: Lt.pd at run time
__global__ void __launch_bounds__(256) lattice_build_packed_kernel(Lattice Lt, const uint8_t *__restrict__ rgb, int W,
                                                                   int H, float posdev, float featdev, int phase) {
    __shared__ unsigned long long lkey[LT_SLOTS];
    __shared__ unsigned lcnt[LT_SLOTS], lgs[LT_SLOTS], lbase[LT_SLOTS];
    __shared__ int newkeys, skip, ntile;
    const int pd = PD ? PD : Lt.pd, nax = pd + 1;
    const int f = blockIdx.x;                             // frame fastest: with 8 frames per call a frame's workgroups share one XCD
    const long cap0 = pk_cap(Lt, f);
    const bool small = phase == 0 && cap0 < 2 * Lt.E;                    // an attempt that may overflow
    // what the frame's mode will read (tile_mode): everything while the mode is still open
    const bool known = tile_mode_known(Lt), tmode = known && tile_mode(Lt, f);
    const bool want_tile = !known || tmode, want_list = !known || !tmode, want_vid = want_list || Lt.sym;
    if (phase == 1 && !pk_overflowed(Lt, f)) return;                     // (uniform: the statistics are final by now)
    // a workgroup walks every gridDim.y-th tile: ONE each in the first attempt's launch; the second attempt's launch -- which returns
    // here for every frame on nearly every call -- is small (12 960 workgroups that only start and stop cost 15 us)
    for (int tile = (int)blockIdx.y; tile < Lt.tiles; tile += (int)gridDim.y) {
    int *tl = Lt.tlist + ((long)f * Lt.tiles + tile) * 256 * nax;
    int *tp = Lt.tpos + ((long)f * Lt.tiles + tile) * 256 * nax;
    if (threadIdx.x == 0) {
        newkeys = 0;
        ntile = 0;
        // the attempt already failed for this frame: nothing this workgroup inserts will be used
        skip = small && __hip_atomic_load(Lt.stat + 4 * f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    }
    __syncthreads();
    if (skip) return;
    // A workgroup takes a 16 x 16 pixel tile, not 256 pixels of one row: neighbours in BOTH directions share lattice
    // vertices, so the tile's 1 536 entries collapse to fewer distinct keys and fewer of them go to the global table
    // (the global inserts -- dependent device-scope atomics -- are what this kernel waits for).
    const int tiles_x = (W + 15) >> 4;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int py = ty * 16 + (threadIdx.x >> 4), px = tx * 16 + (threadIdx.x & 15);
    const bool live = py < H && px < W;
    const int p = live ? py * W + px : 0;
    for (int i = threadIdx.x; i < LT_SLOTS; i += blockDim.x) { lkey[i] = PK_EMPTY; lcnt[i] = 0u; }
    __syncthreads();
    float wgt[PD_MAX + 1];
    int lh[PD_MAX + 1], lrank[PD_MAX + 1];
    if (live) {
        int rem0[PD_MAX + 1], rank[PD_MAX + 1];
        float bary[PD_MAX + 2];
        lattice_point(pd, p, f, Lt.N, W, rgb, posdev, featdev, rem0, rank, bary, Lt.featf);
        for (int r = 0; r < nax; r++) {
            short key[PD_MAX];
            lattice_key(pd, r, rem0, rank, key);
            const unsigned long long k = pack64(key, pd);
            unsigned h = key_hash(key, pd) & (LT_SLOTS - 1);
            for (;;) {
                const unsigned long long prev = atomicCAS(&lkey[h], PK_EMPTY, k);
                if (prev == PK_EMPTY || prev == k) break;
                h = (h + 1) & (LT_SLOTS - 1);
            }
            lh[r] = (int)h;
            lrank[r] = (int)atomicAdd(&lcnt[h], 1u);
            wgt[r] = bary[r];
        }
    }
    __syncthreads();
    // the block's distinct keys: insert into the frame's table, reserve the block's share of the vertex's list
    unsigned long long *table = Lt.table + (long)f * 2 * Lt.E;
    unsigned long long *cursor = Lt.cursor64 + (long)f * 2 * Lt.E;
    const unsigned nb = phase == 0 ? (unsigned)cap0 : (unsigned)(2 * Lt.E);
    int mine = 0;
    // The distinct keys first become a dense list (its order is the tile's vertex list): a key's insert is two or three DEPENDENT
    // device-scope round trips, and a thread that owned several of the ~150 occupied slots among its eight walked them one after the
    // other -- dense, every thread has at most one key on a natural frame and the tile waits for one chain (round 6).
    for (int i = threadIdx.x; i < LT_SLOTS; i += blockDim.x)
        if (lkey[i] != PK_EMPTY) lgs[atomicAdd(&ntile, 1)] = (unsigned)i;           // (lgs is free until the inserts write it)
    __syncthreads();
    constexpr int KPT = LT_SLOTS / 256;                      // keys per thread at most (the list has <= 256 (pd + 1) <= LT_SLOTS entries)
    int mykey[KPT];
#pragma unroll
    for (int j = 0; j < KPT; j++) mykey[j] = (int)threadIdx.x + 256 * j < ntile ? (int)lgs[threadIdx.x + 256 * j] : -1;
    __syncthreads();                                         // every index is in registers: lgs may be overwritten
#pragma unroll
    for (int j = 0; j < KPT; j++) {
        const int i = mykey[j], ti = threadIdx.x + 256 * j;
        if (i < 0) continue;
        const unsigned long long k = lkey[i];
        short key[PD_MAX];
        unpack64(k, pd, key);
        unsigned h = key_hash(key, pd) % nb;
        int probes = 0;
        bool placed = true;
        for (;;) {
            unsigned long long cur = __hip_atomic_load(table + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == PK_EMPTY) {
                cur = atomicCAS(table + h, PK_EMPTY, k);
                if (cur == PK_EMPTY) ++mine;
            }
            if (cur == PK_EMPTY || cur == k) break;
            if (++h == nb) h = 0;
            if (small && ++probes > PK_PROBE_LIMIT) {          // the small table is too full: the frame is redone
                atomicExch(Lt.stat + 4 * f + 1, 1);
                placed = false;
                break;
            }
        }
        lgs[i] = h;
        // ONE atomic reserves the tile's share of the vertex's CSR range (low half) and numbers the tile among the vertex's tiles
        const unsigned long long old = placed ? atomicAdd(cursor + h, (unsigned long long)lcnt[i] | (1ull << 32)) : 0ull;
        lbase[i] = (unsigned)old;
        lcnt[i] = (unsigned)ti;                                  // (the count is spent) the key's place in the tile's vertex list
        if (want_tile) {
            tl[ti] = (int)h;
            tp[ti] = (int)(old >> 32);
        }
    }
    if (small && mine) atomicAdd(&newkeys, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        // per tile, summed per frame by pk_totals_kernel: 1 620 atomics onto one counter per frame are served one after the other
        // (measured: 130 us of this kernel per counter at 8 frames of 480x854)
        Lt.tnew[(long)f * Lt.tiles + tile] = small ? newkeys : 0;
        Lt.tcnt[(long)f * Lt.tiles + tile] = ntile;
    }
    if (live) {
        const long base = (long)f * Lt.E + p;
        for (int r = 0; r < nax; r++) {
            const long e = base + (long)r * Lt.N;
            if (want_vid) Lt.vid[e] = (int)lgs[lh[r]];           // bucket for now; the fill pass turns it into the vertex id
            if (want_list) Lt.rel[e] = (int)lbase[lh[r]] + lrank[r];
            const long we = widx_tile(Lt, f, tile, r, threadIdx.x, p);
            Lt.weight[we] = wgt[r];
            if (want_tile) Lt.tslot[we] = (unsigned short)lcnt[lh[r]];
        }
    }
    __syncthreads();                                     // the next tile reuses the LDS tables
    }
}

// stat[4 f] = distinct keys the frame's tiles inserted into the (small) table, stat[4 f + 3] = total length of its tiles' lists
__global__ void __launch_bounds__(256) pk_totals_kernel(Lattice Lt, int phase) {
    __shared__ int sh[2][4];
    const int f = blockIdx.x;
    if (phase == 1 && !pk_overflowed(Lt, f)) return;                     // (uniform; phase 1 rewrote the overflowed frames only)
    int a = 0, b = 0;
    for (int t = threadIdx.x; t < Lt.tiles; t += blockDim.x) {
        a += Lt.tnew[(long)f * Lt.tiles + t];
        b += Lt.tcnt[(long)f * Lt.tiles + t];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (phase == 0) Lt.stat[4 * f] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        Lt.stat[4 * f + 3] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// exclusive scans over the table's buckets: vertex numbering (occupied flag) and CSR offsets (entry counts)
// (grid-stride over the nblk tiles of the full table: a natural frame's table fills 128 of the 2 402 -- the launch is 256 workgroups
// per frame, not 2 402 that mostly start and stop)
__global__ void __launch_bounds__(SCAN_BLOCK) pk_scan_local_kernel(Lattice Lt, int nblk) {
    const int f = blockIdx.y;
    const long SS = 2 * Lt.E, S = pk_buckets(Lt, f);          // frame stride of the bucket arrays, buckets in use
    int *bs = Lt.blocksum2 + (long)f * 2 * (nblk + 1);
    const unsigned long long *cursor = Lt.cursor64 + (long)f * SS;
    const bool tile = tile_mode(Lt, f);         // the lists scanned are then the vertices' partial sums (one per tile), not their entries
    int *sv = Lt.slot_vid2 + (long)f * SS, *so = Lt.slot_off + (long)f * SS;
    for (int vb = blockIdx.x; vb < nblk; vb += gridDim.x) {
        if ((long)vb * SCAN_TILE >= S) {                       // past the table in use: empty tile
            if (threadIdx.x == 0) {
                bs[vb] = 0;
                bs[nblk + 1 + vb] = 0;
            }
            continue;
        }
        const long base = (long)vb * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
        int fl[SCAN_ITEMS], cn[SCAN_ITEMS], sf = 0, sc = 0;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const long b = base + i;
            const unsigned long long cu = (b < S) ? cursor[b] : 0ull;
            cn[i] = tile ? (int)(cu >> 32) : (int)(unsigned)cu;
            fl[i] = cu != 0ull ? 1 : 0;             // every inserted key counted at least one entry
            sf += fl[i];
            sc += cn[i];
        }
        int tf, tc;
        int of = block_exclusive_scan(sf, &tf);
        int oc = block_exclusive_scan(sc, &tc);
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const long b = base + i;
            if (b < S) {
                sv[b] = fl[i] ? of : -1;
                if (fl[i]) so[b] = oc;              // offsets are only ever read for occupied buckets
            }
            of += fl[i];
            oc += cn[i];
        }
        if (threadIdx.x == 0) {
            bs[vb] = tf;
            bs[nblk + 1 + vb] = tc;
        }
    }
}
__global__ void __launch_bounds__(SCAN_BLOCK) pk_scan_blocks_kernel(Lattice Lt, int nblk) {
    const int f = blockIdx.x;
    __shared__ int carry_s;
    for (int which = 0; which < 2; which++) {
        int *bs = Lt.blocksum2 + (long)f * 2 * (nblk + 1) + which * (nblk + 1);
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
        for (int b0 = 0; b0 < nblk; b0 += SCAN_BLOCK) {
            const int b = b0 + threadIdx.x;
            const int v = b < nblk ? bs[b] : 0;
            int total;
            const int ex = block_exclusive_scan(v, &total);
            const int carry = carry_s;
            if (b < nblk) bs[b] = carry + ex;
            __syncthreads();
            if (threadIdx.x == 0) carry_s = carry + total;
            __syncthreads();
        }
        if (threadIdx.x == 0 && which == 0) Lt.L[f] = carry_s;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(SCAN_BLOCK) pk_scan_apply_kernel(Lattice Lt, int nblk) {
    const int f = blockIdx.y;
    const long SS = 2 * Lt.E, S = pk_buckets(Lt, f);
    const int *bs = Lt.blocksum2 + (long)f * 2 * (nblk + 1);
    int *sv = Lt.slot_vid2 + (long)f * SS, *so = Lt.slot_off + (long)f * SS;
    const unsigned long long *cursor = Lt.cursor64 + (long)f * SS;
    const bool tile = tile_mode(Lt, f);
    const long fb = (long)f * Lt.E;
    for (int vb = blockIdx.x; vb < nblk && (long)vb * SCAN_TILE < S; vb += gridDim.x) {
        const int addf = bs[vb], addc = bs[nblk + 1 + vb];
        const long base = (long)vb * SCAN_TILE + (long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const long b = base + i;
            if (b >= S) continue;
            if (sv[b] >= 0) {
                const int off = so[b] + addc;
                so[b] = off;
                const int v = sv[b] + addf;
                sv[b] = v;
                Lt.vrep[fb + v] = (int)b;
                Lt.off[fb + v] = off;
                Lt.cnt[fb + v] = tile ? (int)(cursor[b] >> 32) : (int)(unsigned)cursor[b];
            }
        }
    }
}
// entry -> (vertex id, CSR slot): a streaming pass, no atomics
__global__ void __launch_bounds__(256) pk_fill_kernel(Lattice Lt) {
    const int f = blockIdx.x;
    // tile mode: no list walk, and the slice finds its vertices through the tile's list (slice_kernel) -- nothing to do here, unless
    // the stand-alone normaliser pass of the symmetric normalisation (slice_norm_kernel) is going to read the entries' vertex ids
    const bool tile = tile_mode(Lt, f);
    if (tile && !Lt.sym) return;
    const long fb = (long)f * Lt.E, S = 2 * Lt.E;
    for (long idx = (long)blockIdx.y * blockDim.x + threadIdx.x; idx < Lt.E; idx += (long)gridDim.y * blockDim.x) {
        const int b = Lt.vid[fb + idx];
        const int v = Lt.slot_vid2[(long)f * S + b];
        if (!tile) {
            const int pos = Lt.slot_off[(long)f * S + b] + Lt.rel[fb + idx];
            const int r = (int)(idx / Lt.N), p = (int)(idx - (long)r * Lt.N);
            Lt.csr[fb + pos] = make_int2(p, __float_as_int(Lt.weight[widx(Lt, f, r, p)]));
        }
        Lt.vid[fb + idx] = v;
    }
}
// tile splat: a tile's list position -> the index of the partial sum it owns: the vertex's range (the scan's offset of its bucket)
// + the tile's ordinal among the vertex's tiles
__global__ void __launch_bounds__(256) tile_list_kernel(Lattice Lt) {
    const int f = blockIdx.x;
    if (!tile_mode(Lt, f)) return;
    const int nax = Lt.pd + 1;
    const int *so = Lt.slot_off + (long)f * 2 * Lt.E, *sv = Lt.slot_vid2 + (long)f * 2 * Lt.E;
    for (int t = blockIdx.y; t < Lt.tiles; t += gridDim.y) {
        int *tl = Lt.tlist + ((long)f * Lt.tiles + t) * 256 * nax;
        int *tp = Lt.tpos + ((long)f * Lt.tiles + t) * 256 * nax;
        const int n = Lt.tcnt[(long)f * Lt.tiles + t];
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int b = tl[i];
            tl[i] = so[b] + tp[i];
            tp[i] = sv[b];                     // the ordinal is spent: the vertex itself, for the slice
        }
    }
}
template <int PD>      // 2 / 5: the key arithmetic unrolls with a compile-time dimension; 0: Lt.pd at run time
__global__ void __launch_bounds__(256) pk_neighbours_kernel(Lattice Lt) {
    const int f = blockIdx.x;
    const int pd = PD ? PD : Lt.pd, nax = pd + 1;
    const long Lf = Lt.L[f], S = 2 * Lt.E;
    const unsigned long long *table = Lt.table + (long)f * S;
    const int *sv = Lt.slot_vid2 + (long)f * S;
    int *nb = Lt.nb + (long)f * Lt.E * 2 * nax;
    const unsigned nbk = (unsigned)pk_buckets(Lt, f);          // the modulus the insert used
    const long total = Lf * nax;
    for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.y * blockDim.x) {
        // axis-major: the lanes of a wavefront take consecutive vertices of ONE axis (their representative slots and their own
        // neighbour entries are then consecutive in memory; vertex-major, six lanes shared a vertex and wrote six arrays)
        const int axis = (int)(i / Lf);
        const long v = i - (long)axis * Lf;
        short key[8];
        unpack64(table[Lt.vrep[(long)f * Lt.E + v]], pd, key);
        for (int k = 0; k < pd; k++) key[k] = (short)(key[k] + 1);
        if (axis < pd) key[axis] = (short)(key[axis] - (pd + 1));
        const unsigned long long want = pack64(key, pd);
        unsigned h = key_hash(key, pd) % nbk;
        for (;;) {
            const unsigned long long cur = table[h];
            if (cur == PK_EMPTY) break;
            if (cur == want) {
                const int u = sv[h];
                nb[((long)axis * Lt.E + v) * 2] = u;
                nb[((long)axis * Lt.E + u) * 2 + 1] = (int)v;
                break;
            }
            if (++h == nbk) h = 0;
        }
    }
}

// ---------------------------------------------------------------------------------- sort build
// RCF_CRF_BUILD_SORT (build 3): the lattice by SORTING instead of hashing.  Every (pixel, remainder) entry of every frame of
// the call becomes a 64-bit key (frame << 60 | packed lattice key) with its entry index as the value; one device-wide radix sort
// (rocPRIM) puts a frame's entries in key order, i.e. grouped by vertex: the sorted sequence IS the CSR list, run heads are the
// vertices, an inclusive scan numbers them.  Vertices come out numbered in KEY order, which is what makes this build worth its
// sort on noise-like frames (millions of vertices met by one entry each): a blur pass gathers each vertex's two neighbours along
// an axis, key + constant -- in key order those are two more nearly sequential streams instead of random 8-byte reads -- and the
// neighbour search itself is a merge of two sorted sequences instead of six hash probes per vertex.  On natural frames (a few
// 10^4 vertices, everything in L2 either way) the tile-local de-duplication of the packed build is cheaper than sorting 2.5 M
// entries per frame (tools/lab/sort_lab.hip; profiles/r05_crf_sort_build.txt), so the packed build stays the default there.
// Same lattice, same sums in fixed point: results are bit-identical to the other builds (tests/test_crf_gpu.py).
constexpr unsigned long long KEY60 = (1ull << 60) - 1;

template <int PD>
__global__ void __launch_bounds__(256) sort_keys_kernel(Lattice Lt, const uint8_t *__restrict__ rgb, int W, float posdev,
                                                        float featdev, unsigned long long *__restrict__ K, unsigned *__restrict__ V) {
    const int pd = PD ? PD : Lt.pd;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = blockIdx.y;
    if (p >= Lt.N) return;
    int rem0[PD_MAX + 1], rank[PD_MAX + 1];
    float bary[PD_MAX + 2];
    lattice_point(pd, p, f, Lt.N, W, rgb, posdev, featdev, rem0, rank, bary, Lt.featf);
    const long base = (long)f * Lt.E + p;
    for (int r = 0; r <= pd; r++) {
        short key[PD_MAX];
        lattice_key(pd, r, rem0, rank, key);
        const long e = base + (long)r * Lt.N;
        K[e] = ((unsigned long long)f << 60) | pack64(key, pd);
        V[e] = (unsigned)(r * Lt.N + p);
        Lt.weight[(long)f * Lt.Ep + (long)r * Lt.N + p] = bary[r];
    }
}

// flag = 1 at the first entry of every vertex (and of every frame)
__global__ void __launch_bounds__(256) sort_heads_kernel(Lattice Lt, const unsigned long long *__restrict__ K, int *__restrict__ flag,
                                                         long n) {
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long)gridDim.x * blockDim.x)
        flag[g] = (g % Lt.E == 0 || K[g] != K[g - 1]) ? 1 : 0;
}

// sorted position g (frame f, CSR position i) -> the entry's vertex id, its CSR record, the vertex's list offset and key
__global__ void __launch_bounds__(256) sort_scatter_kernel(Lattice Lt, const unsigned long long *__restrict__ K,
                                                           const unsigned *__restrict__ V, const int *__restrict__ flag,
                                                           const int *__restrict__ runs, unsigned long long *__restrict__ ukey) {
    const int f = blockIdx.x;
    const long fb = (long)f * Lt.E;
    const int base = runs[fb] - 1;                        // inclusive scan: the frame's first entry is a head
    for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < Lt.E; i += (long)gridDim.y * blockDim.x) {
        const long g = fb + i;
        const int v = runs[g] - 1 - base;
        const unsigned e = V[g];
        Lt.vid[fb + e] = v;
        Lt.csr[g] = make_int2((int)(e % (unsigned)Lt.N), __float_as_int(Lt.weight[(long)f * Lt.Ep + e]));
        if (flag[g]) {
            Lt.off[fb + v] = (int)i;
            ukey[fb + v] = K[g] & KEY60;
        }
        if (i == Lt.E - 1) Lt.L[f] = v + 1;
    }
}
__global__ void __launch_bounds__(256) sort_cnt_kernel(Lattice Lt) {
    const int f = blockIdx.x;
    const long fb = (long)f * Lt.E, Lf = Lt.L[f];
    for (long v = (long)blockIdx.y * blockDim.x + threadIdx.x; v < Lf; v += (long)gridDim.y * blockDim.x)
        Lt.cnt[fb + v] = (v + 1 < Lf ? Lt.off[fb + v + 1] : (int)Lt.E) - Lt.off[fb + v];
}

// neighbours along every axis: the neighbour's packed key is this vertex's key PLUS A CONSTANT (no 12-bit field can wrap:
// keys_fit_12bit), so the wanted keys of consecutive vertices are sorted like the keys themselves and their positions t(v) in the
// key array are non-decreasing: a MERGE.  Two levels: a coarse kernel finds t for every 256th vertex (independent binary searches,
// all in flight at once), the fine kernel searches each vertex inside [t(chunk start), t(next chunk start)] -- a window of a few
// hundred keys that its workgroup keeps in L1/L2 -- without any barrier.
__device__ __forceinline__ unsigned long long axis_delta(int pd, int axis) {
    unsigned long long ones = 0;
    for (int k = 0; k < pd; k++) ones |= 1ull << (12 * k);
    return ones - (axis < pd ? ((unsigned long long)(pd + 1) << (12 * axis)) : 0ull);
}
__device__ __forceinline__ long lower_bound64(const unsigned long long *__restrict__ a, long lo, long hi, unsigned long long want) {
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (a[mid] < want) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <int PD>
__global__ void __launch_bounds__(256) sort_neighbours_coarse_kernel(Lattice Lt, const unsigned long long *__restrict__ ukey,
                                                                     int *__restrict__ coarse, int cstride) {
    const int pd = PD ? PD : Lt.pd, nax = pd + 1;
    const int f = blockIdx.x;
    const long fb = (long)f * Lt.E, Lf = Lt.L[f];
    const unsigned long long *uk = ukey + fb;
    const long nchunk = (Lf + 255) / 256;
    for (long i = (long)blockIdx.y * blockDim.x + threadIdx.x; i < nchunk * nax; i += (long)gridDim.y * blockDim.x) {
        const int axis = (int)(i / nchunk);
        const long c = i - (long)axis * nchunk;
        coarse[((long)f * nax + axis) * cstride + c] = (int)lower_bound64(uk, 0, Lf, uk[c * 256] + axis_delta(pd, axis));
    }
}
template <int PD>
__global__ void __launch_bounds__(256) sort_neighbours_kernel(Lattice Lt, const unsigned long long *__restrict__ ukey,
                                                              const int *__restrict__ coarse, int cstride) {
    const int pd = PD ? PD : Lt.pd, nax = pd + 1;
    const int f = blockIdx.x;
    const long fb = (long)f * Lt.E, Lf = Lt.L[f];
    const unsigned long long *uk = ukey + fb;
    int *nb = Lt.nb + fb * 2 * nax;
    const long nchunk = (Lf + 255) / 256;
    for (long c = blockIdx.y; c < nchunk * nax; c += gridDim.y) {
        const int axis = (int)(c / nchunk);
        const long ch = c - (long)axis * nchunk, v = ch * 256 + threadIdx.x;
        if (v >= Lf) continue;
        const int *cw = coarse + ((long)f * nax + axis) * cstride;
        const long lo = cw[ch], hi = ch + 1 < nchunk ? min((long)cw[ch + 1] + 1, Lf) : Lf;
        const unsigned long long want = uk[v] + axis_delta(pd, axis);
        const long t = lower_bound64(uk, lo, hi, want);
        if (t < Lf && uk[t] == want) {
            nb[((long)axis * Lt.E + v) * 2] = (int)t;
            nb[((long)axis * Lt.E + t) * 2 + 1] = (int)v;
        }
    }
}

// ---------------------------------------------------------------------------------- iteration
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// splat as a gather over the CSR lists, accumulating (Q0*w, Q1*w, w) in 2^-40 fixed point (exact integer
// adds: order independent) and writing the float4 vertex value directly.  A wavefront takes 64 vertices:
// lists of up to SHORT entries are summed by their own lane (noise-like images: ~1 entry per vertex),
// longer lists (smooth images: tens to thousands of entries) by the whole wavefront, one after another.
constexpr int SHORT_LIST = 6;
// (Round 6, measured and dropped: round(p 2^40) in integer arithmetic -- the float's significand shifted by its exponent, rounded to
// even by hand; bit-identical on 144 calls (tools/crf_hashes.py) and 3-7 % SLOWER than __double2ll_rn((double)p * 2^40): fp64
// multiply and the conversion run at full rate on this chip, the 64-bit shifts and selects of the integer form do not pay.)
// MODE 0: the two label channels (Q0*w, Q1*w) -> float2.  MODE 1: the homogeneous channel (w) -> float; it does not
// depend on Q, so it is splatted, blurred and sliced ONCE per lattice (build_lattice) instead of every iteration
// (same operations on the same inputs as the reference's third value channel: bit-identical normalisation).
template <int MODE>
__device__ __forceinline__ void acc_entry(const int2 pw, const float *__restrict__ Qf, long long &a0, long long &a1, long long &a2) {
    const float wgt = __int_as_float(pw.y);
    if (MODE == 0 || MODE == 2) {
        const float2 q = *reinterpret_cast<const float2 *>(Qf + (long)pw.x * MLAB);
        a0 += __double2ll_rn((double)(q.x * wgt) * FIX_SCALE);
        a1 += __double2ll_rn((double)(q.y * wgt) * FIX_SCALE);
        if (MODE == 2) a2 += __double2ll_rn((double)wgt * FIX_SCALE);
    } else {
        a0 += __double2ll_rn((double)wgt * FIX_SCALE);
    }
}
__device__ __forceinline__ float fixed_to_float(long long a) { return (float)((double)a * (1.0 / FIX_SCALE)); }
// MODE 2: the label channels AND the homogeneous channel of one list walk (the first filter pass after a build: the
// normaliser comes out of the same splat / blur / slice as the first message -- same operations on the same inputs as the
// stand-alone MODE 1 pass, so the same bits)
template <int MODE>
__device__ __forceinline__ void store_val(void *out, float *outz, long i, long long a0, long long a1, long long a2) {
    if (MODE == 0 || MODE == 2) reinterpret_cast<float2 *>(out)[i] = make_float2(fixed_to_float(a0), fixed_to_float(a1));
    else reinterpret_cast<float *>(out)[i] = fixed_to_float(a0);
    if (MODE == 2) outz[i] = fixed_to_float(a2);
}
// The per-iteration kernels take grid (frames, chunks): with the frame as the FASTEST grid index a frame's workgroups
// all run on XCD frame % 8 (for 8 k frames per call), so its label values, lattice values and neighbour lists stay in
// that XCD's 4 MB L2 from the splat through the blurs to the slice instead of being fetched by all eight.
// Tried in round 2 and rejected: a position-parallel splat (one lane per CSR position or per four, segmented scan over the
// wavefront, 64-bit integer atomics into per-vertex accumulators, bit-identical sums): 0.344-0.360 ms/frame against this
// kernel's 0.353-0.359 on smooth frames, slower on mixed batches -- the ~0.6 M 64-bit L2 atomics per frame and iteration
// cost what the list walk costs.
template <int MODE>
__global__ void __launch_bounds__(256) splat_gather_kernel(Lattice Lt, const float *__restrict__ Q, void *__restrict__ out,
                                                           float *__restrict__ outz = nullptr) {
    const int f = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const long Lf = Lt.L[f];
    const long fb = (long)f * Lt.E;
    if (tile_mode(Lt, f)) {
        // the frame's entries were summed per tile (splat_tile_kernel): a vertex's value is the sum of its tiles' partial sums,
        // which lie next to each other -- one lane per vertex (a handful of partial sums each)
        const long long *g = Lt.accg + fb * 3;
        for (long v = (long)blockIdx.y * blockDim.x + threadIdx.x; v < Lf; v += (long)gridDim.y * blockDim.x) {
            const long beg = Lt.off[fb + v];
            const int n = Lt.cnt[fb + v];
            long long a0 = 0, a1 = 0, a2 = 0;
            for (int i = 0; i < n; i++) {
                if (MODE == 1) {
                    a0 += g[2 * Lt.E + beg + i];
                } else {
                    const longlong2 pr = *reinterpret_cast<const longlong2 *>(g + (beg + i) * 2);
                    a0 += pr.x;
                    a1 += pr.y;
                    if (MODE == 2) a2 += g[2 * Lt.E + beg + i];
                }
            }
            store_val<MODE>(out, outz, fb + v, a0, a1, a2);
        }
        return;
    }
    const float *Qf = Q + (long)f * Lt.N * MLAB;
    const int wpb = blockDim.x >> 6;
    if (Lt.E > 4 * Lf) {
        // Long lists (natural images: ~35 entries per vertex).  A list is two dependent memory round trips (its
        // entries, then the label values they point to), so the kernel is bound by how many lists are in flight:
        // four 16-lane groups per wavefront take one list each (lists of more than GROUP_LIST entries are left to
        // the whole wavefront afterwards).
        constexpr int GROUP_LIST = 64;
        const int g = lane >> 4, sl = lane & 15;
        const long nw = (long)gridDim.y * wpb, w = (long)blockIdx.y * wpb + (threadIdx.x >> 6);
        for (long vb = w * 4; vb < Lf; vb += nw * 4) {         // vb is wavefront-uniform
            const long v = vb + g;
            int beg = 0, n = 0;
            if (v < Lf) { beg = Lt.off[fb + v]; n = Lt.cnt[fb + v]; }
            const bool big = n > GROUP_LIST;
            long long a0 = 0, a1 = 0, a2 = 0;
            if (!big) {
                // up to GROUP_LIST / 16 records per lane, all loaded before the first label value is asked for: the list
                // costs three dependent round trips (its extent, its records, their labels) whatever its length
                int2 pw[GROUP_LIST / 16];
#pragma unroll
                for (int k = 0; k < GROUP_LIST / 16; ++k) {
                    const int i = sl + 16 * k;
                    pw[k] = i < n ? Lt.csr[fb + beg + i] : make_int2(-1, 0);
                }
#pragma unroll
                for (int k = 0; k < GROUP_LIST / 16; ++k)
                    if (pw[k].x >= 0) acc_entry<MODE>(pw[k], Qf, a0, a1, a2);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {                   // sums inside the 16-lane group
                a0 += __shfl_xor(a0, o, 64);
                if (MODE != 1) a1 += __shfl_xor(a1, o, 64);
                if (MODE == 2) a2 += __shfl_xor(a2, o, 64);
            }
            if (sl == 0 && v < Lf && !big) store_val<MODE>(out, outz, fb + v, a0, a1, a2);
            unsigned long long bigm = __ballot(big && sl == 0);
            while (bigm) {
                const int j = __ffsll((long long)bigm) - 1;     // lane 16 * group of a long list
                bigm &= bigm - 1;
                const int bj = __shfl(beg, j, 64), nj = __shfl(n, j, 64);
                long long b0 = 0, b1 = 0, b2 = 0;
                for (int i = lane; i < nj; i += 64) acc_entry<MODE>(Lt.csr[fb + bj + i], Qf, b0, b1, b2);
                b0 = wave_sum_ll(b0);
                if (MODE != 1) b1 = wave_sum_ll(b1);
                if (MODE == 2) b2 = wave_sum_ll(b2);
                if (lane == 0) store_val<MODE>(out, outz, fb + vb + (j >> 4), b0, b1, b2);
            }
        }
        return;
    }
    // short lists (noise-like images: ~1 entry per vertex): 64 vertices per wavefront, every lane owns one
    const int chunk = 64;
    for (long v0 = ((long)blockIdx.y * wpb + (threadIdx.x >> 6)) * chunk; v0 < Lf; v0 += (long)gridDim.y * wpb * chunk) {
        const long v = v0 + lane;
        int beg = 0, n = 0;
        if (lane < chunk && v < Lf) { beg = Lt.off[fb + v]; n = Lt.cnt[fb + v]; }
        if (n > 0 && n <= SHORT_LIST) {
            long long a0 = 0, a1 = 0, a2 = 0;
            for (int i = 0; i < n; i++) acc_entry<MODE>(Lt.csr[fb + beg + i], Qf, a0, a1, a2);
            store_val<MODE>(out, outz, fb + v, a0, a1, a2);
        }
        unsigned long long longm = __ballot(n > SHORT_LIST);
        while (longm) {
            const int j = __ffsll((long long)longm) - 1;
            longm &= longm - 1;
            const int bj = __shfl(beg, j, 64), nj = __shfl(n, j, 64);
            long long a0 = 0, a1 = 0, a2 = 0;
            for (int i = lane; i < nj; i += 64) acc_entry<MODE>(Lt.csr[fb + bj + i], Qf, a0, a1, a2);
            a0 = wave_sum_ll(a0);
            if (MODE != 1) a1 = wave_sum_ll(a1);
            if (MODE == 2) a2 = wave_sum_ll(a2);
            if (lane == 0) store_val<MODE>(out, outz, fb + v0 + j, a0, a1, a2);
        }
    }
}

// The splat's first half on tile-mode frames (see Lattice::tslot): a pixel's pd + 1 entries add the SAME fixed-point terms acc_entry
// adds, into the tile's list in LDS; the tile's sums go out as plain stores, one per distinct vertex and channel, to the place the
// build gave this (tile, vertex) pair; splat_gather_kernel's tile-mode branch adds a vertex's partial sums.  Integer adds: the
// result does not depend on any order -- the same bits as the list walk's.  (Measured on the way: 64-bit atomics from the tiles
// straight into per-vertex sums cost 136 us per pass for 8 frames -- the L2 takes them at ~1.5 per clock and XCD -- against 41 us
// for everything else in this kernel; the LDS atomics, same-address conflicts included, are ~2 us of it.)
// Neighbouring pixels mostly hit the SAME vertex, and same-address LDS atomics of one instruction are served one after the other:
// a short list is kept in R copies, lane (x, y) of the tile adds into copy (x + 4 y) mod R -- 64 / R lanes per copy and wavefront
__device__ __forceinline__ int tile_copies(int n, int stride) {
    int R = 1;
    while (R < 16 && 2 * R * n <= stride) R *= 2;
    return R;
}
__device__ __forceinline__ int tile_copy_of(int tid, int R) { return ((tid & 15) + 4 * (tid >> 4)) & (R - 1); }
// the tile's sums (channel c of list position i, copy k: acc[c * CAP + k * n + i]) -> the partial sums this tile owns
// pos0: tl[threadIdx.x], asked for by the caller before its barriers (natural frames: the whole list is <= 256 long)
template <int MODE>
__device__ __forceinline__ void tile_store_sums(const Lattice &Lt, int f, int t, int n, int R, const unsigned long long *acc, int pos0) {
    constexpr int NCH = MODE == 2 ? 3 : (MODE == 1 ? 1 : 2);
    constexpr int CAP = 256 * (PD_MAX + 1);
    const int *tl = Lt.tlist + ((long)f * Lt.tiles + t) * 256 * (Lt.pd + 1);
    unsigned long long *g = reinterpret_cast<unsigned long long *>(Lt.accg) + (long)f * Lt.E * 3;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long pos = i < 256 ? pos0 : tl[i];             // this tile's partial sum of the vertex: nobody else writes it
        unsigned long long a[NCH];
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            a[c] = 0ull;
            for (int k = 0; k < R; k++) a[c] += acc[c * CAP + k * n + i];
        }
        // [pos][2] label sums (one 16-byte store) in the first 2E values of the frame's 3E, [pos] homogeneous sums behind them
        if (MODE != 1) *reinterpret_cast<ulonglong2 *>(g + pos * 2) = make_ulonglong2(a[0], a[1]);
        if (MODE != 0) g[2 * Lt.E + pos] = a[NCH - 1];
    }
}
template <int MODE>
__global__ void __launch_bounds__(256) splat_tile_kernel(Lattice Lt, const float *__restrict__ Q) {
    constexpr int NCH = MODE == 2 ? 3 : (MODE == 1 ? 1 : 2);
    constexpr int CAP = 256 * (PD_MAX + 1);                  // list entries per channel
    __shared__ unsigned long long acc[NCH * CAP];
    const int f = blockIdx.x;
    if (!tile_mode(Lt, f)) return;
    const int nax = Lt.pd + 1, stride = 256 * nax;
    const int W = Lt.W, H = Lt.N / Lt.W, tiles_x = (W + 15) >> 4;
    const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
    const int py = ty * 16 + (threadIdx.x >> 4), px = tx * 16 + (threadIdx.x & 15);
    const int n = Lt.tcnt[(long)f * Lt.tiles + blockIdx.y];
    const int R = tile_copies(n, stride);
    const int rep = tile_copy_of(threadIdx.x, R);
    const bool live = py < H && px < W;
    const int p = live ? py * W + px : 0;
    // everything the pixel reads from memory is asked for before the first barrier
    float wr[PD_MAX + 1];
    int sr[PD_MAX + 1];
    float2 q = make_float2(0.f, 0.f);
    if (live) {
        if (MODE != 1) q = *reinterpret_cast<const float2 *>(Q + ((long)f * Lt.N + p) * MLAB);
#pragma unroll
        for (int r = 0; r <= PD_MAX; r++) {
            if (r >= nax) break;
            const long e = widx_tile(Lt, f, blockIdx.y, r, threadIdx.x, p);
            wr[r] = Lt.weight[e];
            sr[r] = Lt.tslot[e];
        }
    }
    const int pos0 = (int)threadIdx.x < n ? Lt.tlist[((long)f * Lt.tiles + blockIdx.y) * stride + threadIdx.x] : 0;
    for (int i = threadIdx.x; i < R * n; i += blockDim.x) {
#pragma unroll
        for (int c = 0; c < NCH; c++) acc[c * CAP + i] = 0ull;
    }
    __syncthreads();
    if (live) {
#pragma unroll
        for (int r = 0; r <= PD_MAX; r++) {
            if (r >= nax) break;
            const float wgt = wr[r];
            const int s = rep * n + sr[r];
            if (MODE == 1) {
                atomicAdd(&acc[s], (unsigned long long)__double2ll_rn((double)wgt * FIX_SCALE));
            } else {
                atomicAdd(&acc[s], (unsigned long long)__double2ll_rn((double)(q.x * wgt) * FIX_SCALE));
                atomicAdd(&acc[CAP + s], (unsigned long long)__double2ll_rn((double)(q.y * wgt) * FIX_SCALE));
                if (MODE == 2) atomicAdd(&acc[2 * CAP + s], (unsigned long long)__double2ll_rn((double)wgt * FIX_SCALE));
            }
        }
    }
    __syncthreads();
    tile_store_sums<MODE>(Lt, f, blockIdx.y, n, R, acc, pos0);
}

__device__ __forceinline__ float2 blur3(float2 p, float2 m, float2 q) {
    return make_float2(0.25f * p.x + 0.5f * m.x + 0.25f * q.x, 0.25f * p.y + 0.5f * m.y + 0.25f * q.y);
}
__device__ __forceinline__ float blur3(float p, float m, float q) { return 0.25f * p + 0.5f * m + 0.25f * q; }
__device__ __forceinline__ void zero_of(float2 &v) { v = make_float2(0.f, 0.f); }
__device__ __forceinline__ void zero_of(float &v) { v = 0.f; }

// one blur pass along `axis`: new = 1/4 n+ + 1/2 me + 1/4 n-, missing neighbour = 0.  T = float2 (labels) / float
template <class T>
__global__ void __launch_bounds__(256) blur_kernel(Lattice Lt, int axis, const T *__restrict__ in, T *__restrict__ out) {
    const int f = blockIdx.x;
    const int nax2 = 2 * (Lt.pd + 1);
    const long Lf = Lt.L[f];
    const long fb = (long)f * Lt.E;
    for (long v = (long)blockIdx.y * blockDim.x + threadIdx.x; v < Lf; v += (long)gridDim.y * blockDim.x) {
        const int2 n = *reinterpret_cast<const int2 *>(Lt.nb + (fb * (nax2 / 2) + (long)axis * Lt.E + v) * 2);
        const T me = in[fb + v];
        T vp, vm;
        zero_of(vp);
        zero_of(vm);
        if (n.x >= 0) vp = in[fb + n.x];
        if (n.y >= 0) vm = in[fb + n.y];
        out[fb + v] = blur3(vp, me, vm);
    }
}

// the label channels and the homogeneous channel in one pass (first filter pass after a build): one read of the neighbour
// pair serves both arrays
__global__ void __launch_bounds__(256) blur2_kernel(Lattice Lt, int axis, const float2 *__restrict__ in, float2 *__restrict__ out,
                                                    const float *__restrict__ zin, float *__restrict__ zout) {
    const int f = blockIdx.x;
    const int nax2 = 2 * (Lt.pd + 1);
    const long Lf = Lt.L[f];
    const long fb = (long)f * Lt.E;
    for (long v = (long)blockIdx.y * blockDim.x + threadIdx.x; v < Lf; v += (long)gridDim.y * blockDim.x) {
        const int2 n = *reinterpret_cast<const int2 *>(Lt.nb + (fb * (nax2 / 2) + (long)axis * Lt.E + v) * 2);
        const float2 me = in[fb + v];
        const float zme = zin[fb + v];
        float2 vp = make_float2(0.f, 0.f), vm = make_float2(0.f, 0.f);
        float zp = 0.f, zm = 0.f;
        if (n.x >= 0) { vp = in[fb + n.x]; zp = zin[fb + n.x]; }
        if (n.y >= 0) { vm = in[fb + n.y]; zm = zin[fb + n.y]; }
        out[fb + v] = blur3(vp, me, vm);
        zout[fb + v] = blur3(zp, zme, zm);
    }
}

// TWO blur passes (axes `axis` and `axis + 1`) in one launch: out[v] = blur3(T[n+], T[v], T[n-]) along axis + 1 with
// T[u] = blur3(in[m+(u)], in[u], in[m-(u)]) along `axis` evaluated on the fly for the three vertices the second pass reads -- the
// same float operations on the same values as the two launches (whose intermediate array holds exactly these T), so the same
// bits.  A blur launch over a natural frame's 7 x 10^4 vertices is two dependent round trips (neighbour pair, then values)
// behind a launch boundary: 14 us for 23 MB; the pair costs three round trips and ONE boundary (round 6).  WITHZ: the
// homogeneous channel rides along (first filter pass after a build, blur2_kernel's job).
template <bool WITHZ>
__global__ void __launch_bounds__(256) blur_pair_kernel(Lattice Lt, int axis, const float2 *__restrict__ in, float2 *__restrict__ out,
                                                        const float *__restrict__ zin, float *__restrict__ zout) {
    const int f = blockIdx.x;
    const int nax = Lt.pd + 1;
    const long Lf = Lt.L[f];
    const long fb = (long)f * Lt.E;
    const int2 *nbA = reinterpret_cast<const int2 *>(Lt.nb + (fb * nax + (long)axis * Lt.E) * 2);
    const int2 *nbB = reinterpret_cast<const int2 *>(Lt.nb + (fb * nax + (long)(axis + 1) * Lt.E) * 2);
    for (long v = (long)blockIdx.y * blockDim.x + threadIdx.x; v < Lf; v += (long)gridDim.y * blockDim.x) {
        const int2 nB = nbB[v];
        const int2 nA0 = nbA[v];
        int2 nAp = make_int2(-1, -1), nAm = make_int2(-1, -1);
        if (nB.x >= 0) nAp = nbA[nB.x];
        if (nB.y >= 0) nAm = nbA[nB.y];
        auto first = [&](long u, const int2 n, float2 &t, float &tz) {      // the first pass's value at vertex u
            const float2 me = in[fb + u];
            float2 vp = make_float2(0.f, 0.f), vm = make_float2(0.f, 0.f);
            float zme = 0.f, zp = 0.f, zm = 0.f;
            if (WITHZ) zme = zin[fb + u];
            if (n.x >= 0) { vp = in[fb + n.x]; if (WITHZ) zp = zin[fb + n.x]; }
            if (n.y >= 0) { vm = in[fb + n.y]; if (WITHZ) zm = zin[fb + n.y]; }
            t = blur3(vp, me, vm);
            if (WITHZ) tz = blur3(zp, zme, zm);
        };
        float2 t0, tp = make_float2(0.f, 0.f), tm = make_float2(0.f, 0.f);
        float z0 = 0.f, zp = 0.f, zm = 0.f;
        first(v, nA0, t0, z0);
        if (nB.x >= 0) first(nB.x, nAp, tp, zp);
        if (nB.y >= 0) first(nB.y, nAm, tm, zm);
        out[fb + v] = blur3(tp, t0, tm);
        if (WITHZ) zout[fb + v] = blur3(zp, z0, zm);
    }
}

// build time: inv[p] = 1 / sum_r w_r * z[vid_r]  (z = blurred homogeneous channel)
// sym: inv[p] = 1 / sqrt(that + 1e-20), the symmetric normalisation of DenseCRF2D (see rcf_crf_soft_ex)
template <int PD>
__global__ void __launch_bounds__(256) slice_norm_kernel(Lattice Lt, const float *__restrict__ z, int sym) {
    const int f = blockIdx.x;
    const int p = blockIdx.y * blockDim.x + threadIdx.x;
    if (p >= Lt.N) return;
    const long fb = (long)f * Lt.E;
    const int pd = PD ? PD : Lt.pd;
    float sw = 0;
#pragma unroll
    for (int r = 0; r <= pd; r++) {
        const long pe = fb + (long)r * Lt.N + p;
        sw += Lt.weight[widx(Lt, f, r, p)] * z[fb + Lt.vid[pe]];
    }
    Lt.inv[(long)f * Lt.N + p] = sym ? (float)(1.0 / sqrt((double)sw + 1e-20)) : (float)(1.0 / sw);
}

// slice + Potts weight + (optionally) softmax and MAP.
//   first: next = -U, else next = next_in;  next += w * slice;  last: Q = softmax(next) (+ MAP)
// NORM: the first filter pass after a build -- the blurred homogeneous channel z is sliced beside the labels and the
// pixel's normaliser (slice_norm_kernel's operations in its order) is computed, stored for the later passes and used here
// SPLAT (single potential, not the last iteration): the marginals this pass produces are what the NEXT pass splats, and on a
// tile-mode frame that splat's first half is per tile as well -- the pixel's new marginals go straight into the tile's sums (the
// entries' weights and list positions are in registers already), splat_tile_kernel<0>'s adds and stores; Q itself is not written
// (its only reader would have been that kernel).  Frames in list-walk mode write Q as always.
template <int PD, bool NORM = false, bool SPLAT = false>      // compile-time dimension: the pd + 1 (weight, vertex, value) chains of a pixel are all in flight at once
__global__ void __launch_bounds__(256) slice_kernel(Lattice Lt, const float2 *__restrict__ val,
                                                    const float *__restrict__ unary, float *__restrict__ next,
                                                    float *__restrict__ Q, short *__restrict__ map, int first,
                                                    int last, int write_map, int sym, const float *__restrict__ z = nullptr) {
    const int f = blockIdx.x;
    // 16 x 16 pixel tiles: the pixels of a tile share most of their lattice vertices (the values stay in the CU's L1)
    const int W = Lt.W, H = Lt.N / Lt.W, tiles_x = (W + 15) >> 4;
    const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
    const int py = ty * 16 + (threadIdx.x >> 4), px = tx * 16 + (threadIdx.x & 15);
    const int nax = (PD ? PD : Lt.pd) + 1;
    constexpr int NAXC = PD ? PD + 1 : PD_MAX + 1;        // compile-time loop bound (PD = 0: guarded by nax)
    const long fb = (long)f * Lt.E;
    // tile mode: the tile's distinct vertices are a short list (the build's): their values are fetched ONCE into LDS and a pixel
    // reads them by its entries' list positions (2 bytes each) instead of gathering 8 bytes per entry by vertex id
    constexpr int CAP = 256 * (PD_MAX + 1);
    // SPLAT: the sums' 24 KB hold the staged values first (12 KB + 6 KB; dead once every lane has sliced)
    __shared__ unsigned long long acc[SPLAT ? 2 * CAP : 1];
    __shared__ float2 sval_own[SPLAT ? 1 : CAP];
    __shared__ float szv_own[(NORM && !SPLAT) ? CAP : 1];
    float2 *sval = SPLAT ? reinterpret_cast<float2 *>(acc) : sval_own;
    float *szv = SPLAT ? reinterpret_cast<float *>(acc + CAP) : szv_own;
    const bool tile = tile_mode(Lt, f);                   // (uniform)
    const bool live = py < H && px < W;
    const int p = live ? py * W + px : 0;
    // the pixel's own streams (weights, list positions, normaliser, unary) are asked for BEFORE the tile's values are staged: they
    // depend on nothing in LDS, and behind the barrier their latency would come on top of the staging gather's
    float wr[PD_MAX + 1];
    int sr[PD_MAX + 1];
    float inv_pre = 0.f;
    float2 un_pre = make_float2(0.f, 0.f);
    if (tile && live) {
#pragma unroll
        for (int r = 0; r < NAXC; r++) {
            if (!PD && r >= nax) break;
            const long pe = widx_tile(Lt, f, blockIdx.y, r, threadIdx.x, p);
            wr[r] = Lt.weight[pe];
            sr[r] = Lt.tslot[pe];
        }
        if (!NORM) inv_pre = Lt.inv[(long)f * Lt.N + p];
        if (first) un_pre = *reinterpret_cast<const float2 *>(unary + ((long)f * Lt.N + p) * MLAB);
    }
    int n = 0, R = 1, pos0 = 0;
    if (tile) {
        n = Lt.tcnt[(long)f * Lt.tiles + blockIdx.y];
        if (SPLAT && (int)threadIdx.x < n) pos0 = Lt.tlist[((long)f * Lt.tiles + blockIdx.y) * 256 * nax + threadIdx.x];
        const int *tv = Lt.tpos + ((long)f * Lt.tiles + blockIdx.y) * 256 * nax;
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int vi = tv[i];
            sval[i] = val[fb + vi];
            if (NORM) szv[i] = z[fb + vi];
        }
        if (SPLAT) R = tile_copies(n, 256 * nax);
        __syncthreads();
    }
    float qs0 = 0.f, qs1 = 0.f;
    bool have_q = false;
    if (live) {
        float s0 = 0, s1 = 0, sw = 0;
#pragma unroll
        for (int r = 0; r < NAXC; r++) {
            if (!PD && r >= nax) break;
            const long pe = fb + (long)r * Lt.N + p;
            float wgt;
            float2 v;
            float zz = 0.f;
            if (tile) {
                wgt = wr[r];
                v = sval[sr[r]];
                if (NORM) zz = szv[sr[r]];
            } else {
                wgt = Lt.weight[widx_tile(Lt, f, blockIdx.y, r, threadIdx.x, p)];
                const int vi = Lt.vid[pe];
                v = val[fb + vi];
                if (NORM) zz = z[fb + vi];
            }
            s0 += wgt * v.x;
            s1 += wgt * v.y;
            if (NORM) sw += wgt * zz;
        }
        float inv;
        if (NORM) {
            inv = sym ? (float)(1.0 / sqrt((double)sw + 1e-20)) : (float)(1.0 / sw);
            Lt.inv[(long)f * Lt.N + p] = inv;
        } else {
            inv = tile ? inv_pre : Lt.inv[(long)f * Lt.N + p];
        }
        const long qi = ((long)f * Lt.N + p) * MLAB;
        float n0, n1;
        if (first) {
            if (!tile) un_pre = *reinterpret_cast<const float2 *>(unary + qi);
            n0 = -un_pre.x;
            n1 = -un_pre.y;
        } else {
            n0 = next[qi];
            n1 = next[qi + 1];
        }
        n0 += Lt.w * (s0 * inv);
        n1 += Lt.w * (s1 * inv);
        if (!last) {
            next[qi] = n0;
            next[qi + 1] = n1;
        } else {
            const float mx = n0 < n1 ? n1 : n0;
            const float e0 = __expf(n0 - mx), e1 = __expf(n1 - mx);
            const float tt = e0 + e1;
            const float q0 = e0 / tt, q1 = e1 / tt;
            // symmetric normalisation: what the next iteration splats is Q * norm (inv holds norm = 1/sqrt(K 1) there); the final
            // iteration (write_map) leaves the marginals themselves
            const float sc = (sym && !write_map) ? inv : 1.f;
            qs0 = q0 * sc;
            qs1 = q1 * sc;
            have_q = true;
            if (!(SPLAT && tile)) {
                Q[qi] = qs0;
                Q[qi + 1] = qs1;
            }
            if (write_map) map[(long)f * Lt.N + p] = (q0 < q1) ? 1 : 0;
        }
    }
    if (SPLAT && tile) {
        __syncthreads();                                   // every lane has read its values: the space becomes the sums
        for (int i = threadIdx.x; i < R * n; i += blockDim.x) { acc[i] = 0ull; acc[CAP + i] = 0ull; }
        __syncthreads();
        if (have_q) {
            const int rep = tile_copy_of(threadIdx.x, R);
#pragma unroll
            for (int r = 0; r < NAXC; r++) {
                if (!PD && r >= nax) break;
                const int sl = rep * n + sr[r];
                atomicAdd(&acc[sl], (unsigned long long)__double2ll_rn((double)(qs0 * wr[r]) * FIX_SCALE));
                atomicAdd(&acc[CAP + sl], (unsigned long long)__double2ll_rn((double)(qs1 * wr[r]) * FIX_SCALE));
            }
        }
        __syncthreads();
        tile_store_sums<0>(Lt, f, blockIdx.y, n, R, acc, pos0);
    }
}

// symmetric normalisation: Q *= norm before the first splat
__global__ void __launch_bounds__(256) scale_q_kernel(float *__restrict__ Q, const float *__restrict__ nrm, long n) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    Q[p * 2] *= nrm[p];
    Q[p * 2 + 1] *= nrm[p];
}

// Q = softmax(scale * in)  (startInference: scale -1 on the unary; also T=0 MAP)
__global__ void __launch_bounds__(256) softmax2_kernel(const float *__restrict__ in, float *__restrict__ Q,
                                                       short *__restrict__ map, long n, float scale, int write_map) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const float a = scale * in[p * 2], b = scale * in[p * 2 + 1];
    const float mx = a < b ? b : a;
    const float e0 = __expf(a - mx), e1 = __expf(b - mx);
    const float tt = e0 + e1;
    Q[p * 2] = e0 / tt;
    Q[p * 2 + 1] = e1 / tt;
    if (write_map) map[p] = (e0 / tt < e1 / tt) ? 1 : 0;
}

__global__ void __launch_bounds__(256) unary_from_label_kernel(const short *__restrict__ label,
                                                               float *__restrict__ unary, long n, float u_energy,
                                                               float n_energy, float p_energy) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const short l = label[p];
    float a = (l == -1) ? u_energy : n_energy, b = a;
    if (l == 0) a = p_energy;
    if (l == 1) b = p_energy;
    unary[p * 2] = a;
    unary[p * 2 + 1] = b;
}

__global__ void vertex_counts_kernel(const int *__restrict__ Ls, const int *__restrict__ La, int32_t *__restrict__ nvert, int F) {
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        nvert[2 * f] = Ls ? Ls[f] : 0;
        nvert[2 * f + 1] = La ? La[f] : 0;
    }
}

// ---------------------------------------------------------------------------------- CRFHead prologue
__device__ __forceinline__ unsigned to_u8(float v) {
    v = v * 255.f;
    v = fminf(fmaxf(v, 0.f), 255.f);
    return (unsigned)(uint8_t)v;                                      // truncation, like .type(torch.uint8)
}
// 4 pixels per thread: float4 loads from the three planes, 12 output bytes as three 32-bit stores
__global__ void __launch_bounds__(256) prepare_image_kernel(const float *__restrict__ img,
                                                            const float *__restrict__ mean3,
                                                            const float *__restrict__ std3, int unstd,
                                                            uint8_t *__restrict__ rgb, int HW) {
    const int f = blockIdx.y;
    const int p4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p4 < HW) {
        const bool full = (p4 + 3 < HW) && (HW % 4 == 0);
        unsigned bytes[12];
        if (full) {                                    // 16-byte loads: (f*3 + c)*HW + p4 is a multiple of 4 floats
            float4 pl[3];
#pragma unroll
            for (int c = 0; c < 3; c++) pl[c] = *reinterpret_cast<const float4 *>(img + ((long)f * 3 + c) * HW + p4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float v = j == 0 ? pl[c].x : (j == 1 ? pl[c].y : (j == 2 ? pl[c].z : pl[c].w));
                    if (unstd) v = v * std3[c] + mean3[c];
                    bytes[3 * j + c] = to_u8(v);
                }
            }
        } else {
            for (int j = 0; j < 4; j++) {
                const int p = p4 + j;
                if (p >= HW) { bytes[3 * j] = bytes[3 * j + 1] = bytes[3 * j + 2] = 0; continue; }
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    float v = img[((long)f * 3 + c) * HW + p];
                    if (unstd) v = v * std3[c] + mean3[c];
                    bytes[3 * j + c] = to_u8(v);
                }
            }
        }
        uint8_t *dst = rgb + ((long)f * HW + p4) * 3;
        if (full) {
            unsigned *d32 = reinterpret_cast<unsigned *>(dst);        // (f*HW + p4)*3 is a multiple of 4 bytes
#pragma unroll
            for (int k = 0; k < 3; k++)
                d32[k] = bytes[4 * k] | (bytes[4 * k + 1] << 8) | (bytes[4 * k + 2] << 16) | (bytes[4 * k + 3] << 24);
        } else {
            for (int j = 0; j < 4 && p4 + j < HW; j++)
                for (int c = 0; c < 3; c++) dst[3 * j + c] = (uint8_t)bytes[3 * j + c];
        }
    }
}

// per-frame maximum of the quantised mask (models/crf_head.py:43-55 divides by it): QMAX_BLOCKS workgroups per frame, one atomicMax
// each.  Until round 6 this sat at the end of prepare_image_kernel: 401 workgroups per frame, each waiting at a barrier for a
// device-scope load of the running maximum and, while it still read 0, adding its own atomic -- 67 of that kernel's 77 us.
constexpr int QMAX_BLOCKS = 128;
__global__ void __launch_bounds__(256) mask_qmax_kernel(const float *__restrict__ mask, float crf_scale, unsigned *__restrict__ qmax, int HW) {
    const int f = blockIdx.y;
    float mx = 0.f;                         // the quantisation is monotone: the maximum of the masks, quantised once
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) mx = fmaxf(mx, mask[(long)f * HW + p]);
    __shared__ float wq[4];
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) wq[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = fmaxf(fmaxf(wq[0], wq[1]), fmaxf(wq[2], wq[3])) * 255.f / crf_scale;
        m = fminf(fmaxf(m, 0.f), 255.f);
        const unsigned q = (unsigned)(uint8_t)m;
        if (q > __hip_atomic_load(qmax + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(qmax + f, q);
    }
}

__global__ void __launch_bounds__(256) prepare_unary_kernel(const float *__restrict__ mask, float crf_scale,
                                                            const unsigned *__restrict__ qmax,
                                                            float *__restrict__ unary, int HW) {
    const int f = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    float m = mask[(long)f * HW + p] * 255.f / crf_scale;
    m = fminf(fmaxf(m, 0.f), 255.f);
    const float q = (float)(uint8_t)m;
    float U = q / ((float)qmax[f] + 1e-8f);
    const float lo = 1e-6f, hi = (float)(1.0 - 1e-6);
    U = fminf(fmaxf(U, lo), hi);
    unary[((long)f * HW + p) * 2] = -logf(1.0f - U);
    unary[((long)f * HW + p) * 2 + 1] = -logf(U);
}

// ---------------------------------------------------------------------------------- host side
inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Carver {
    char *base;
    size_t off;
    template <class T>
    T *take(size_t count) {
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += align_up(count * sizeof(T));
        return p;
    }
};

inline int scan_blocks(long E) { return (int)((E + SCAN_TILE - 1) / SCAN_TILE); }

void carve_lattice(Carver &c, Lattice &L, int pd, int N, int F, int W) {
    L.pd = pd;
    L.N = N;
    L.W = W;
    L.E = (long)N * (pd + 1);
    const size_t FE = (size_t)F * L.E;
    L.tiles = rcf_cdiv(W, 16) * rcf_cdiv(N / W, 16);
    L.Ep = (long)L.tiles * 256 * (pd + 1);
    L.wtile = 0;
    L.keys = c.take<uint4>(FE);
    L.weight = c.take<float>((size_t)F * L.Ep);
    L.entries = c.take<int>(2 * FE);
    L.vid = c.take<int>(FE);
    L.slot_vid = c.take<int>(FE);
    L.vrep = c.take<int>(FE);
    L.nb = c.take<int>(FE * 2 * (pd + 1));
    L.cnt = c.take<int>(FE);
    L.off = c.take<int>(FE);
    L.csr = c.take<int2>(FE);
    L.val0 = c.take<float2>(FE);
    L.val1 = c.take<float2>(FE);
    L.inv = c.take<float>((size_t)F * N);
    L.blocksum = c.take<int>((size_t)F * (scan_blocks(L.E) + 1));
    L.L = c.take<int>(F);
    L.table = reinterpret_cast<unsigned long long *>(L.keys);      // 2E x 8 B == E x 16 B
    L.rel = c.take<int>(FE);
    L.slot_vid2 = c.take<int>(2 * FE);
    L.slot_off = c.take<int>(2 * FE);
    L.blocksum2 = c.take<int>((size_t)F * 2 * (scan_blocks(2 * L.E) + 1));
    L.stat = c.take<int>((size_t)F * 4);
    L.cursor64 = c.take<unsigned long long>(2 * FE);
    L.tslot = c.take<unsigned short>((size_t)F * L.Ep);
    L.tlist = c.take<int>((size_t)F * L.tiles * 256 * (pd + 1));
    L.tpos = c.take<int>((size_t)F * L.tiles * 256 * (pd + 1));
    L.tcnt = c.take<int>((size_t)F * L.tiles);
    L.tnew = c.take<int>((size_t)F * L.tiles);
    L.accg = c.take<long long>(3 * FE);
    L.tile_splat = 0;
    L.cap_small = 0;
    L.est = 0;
    L.sort_tmp_bytes = rcf_crf_sort_tmp_bytes(FE);
    L.sort_tmp = c.take<char>(L.sort_tmp_bytes);
}

struct CrfBuffers {
    Lattice smooth, app;
    float *cur, *next, *unary_own;
};

size_t carve_all(char *base, int W, int H, int F, CrfBuffers &b) {
    Carver c{base, 0};
    const int N = W * H;
    carve_lattice(c, b.smooth, 2, N, F, W);
    carve_lattice(c, b.app, 5, N, F, W);
    b.cur = c.take<float>((size_t)F * N * MLAB);
    b.next = c.take<float>((size_t)F * N * MLAB);
    b.unary_own = c.take<float>((size_t)F * N * MLAB);
    return c.off;
}

#define CK(expr)                                 \
    do {                                         \
        hipError_t e__ = (expr);                 \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

int build_lattice_norm(Lattice &L, int F, hipStream_t st);
int lattice_built(Lattice &L, int F, hipStream_t st);

// kernels templated on the lattice dimension: the two potentials of the reference are pd = 2 and pd = 5
#define PD_LAUNCH(pd_, kern, grid, block, st_, ...)                                              \
    do {                                                                                        \
        if ((pd_) == 5) hipLaunchKernelGGL(kern<5>, grid, block, 0, st_, __VA_ARGS__);          \
        else if ((pd_) == 2) hipLaunchKernelGGL(kern<2>, grid, block, 0, st_, __VA_ARGS__);     \
        else hipLaunchKernelGGL(kern<0>, grid, block, 0, st_, __VA_ARGS__);                     \
    } while (0)

// bound on |key coordinate| (see lattice_point): elevated[i] in [-i*cf_i, sum_j cf_j], keys within pd+1 of it
bool keys_fit_12bit(int pd, int W, int H, float posdev, float featdev) {
    double posmax[PD_MAX] = {(double)W / posdev, (double)H / posdev, 255.0 / featdev, 255.0 / featdev, 255.0 / featdev};
    const double inv_std = (pd + 1) * sqrt(2.0 / 3.0);
    double sum = 0, worst = 0;
    for (int i = 1; i <= pd; i++) {
        const double cf = posmax[i - 1] / sqrt((double)i * (i + 1)) * inv_std;
        sum += cf;
        if (i * cf > worst) worst = i * cf;
    }
    return (sum > worst ? sum : worst) + 4.0 * (pd + 1) < 2040.0;
}

int build_lattice_packed(Lattice &L, const uint8_t *rgb, int W, int H, int F, float posdev, float featdev,
                         hipStream_t st) {
    const dim3 gp(rcf_cdiv(L.N, 256), F), ge(F, rcf_cdiv(L.E, 256));       // ge: (frames, chunks), frame = XCD
    const dim3 gpf(F, rcf_cdiv(W, 16) * rcf_cdiv(H, 16));        // (frames, 16 x 16 pixel tiles)
    const int nblk2 = scan_blocks(2 * L.E);
    // first attempt: 2^18 - 1 buckets (3 MB of keys + cursors per frame, room for 131 k distinct keys; measured at
    // 480x854, T=5: 2^21 0.388, 2^19 0.371, 2^18 0.361 ms/frame); L.build 2 forces a tiny table (tests)
    const long small = L.build == 2 ? 1021 : ((1L << 18) - 1);
    L.cap_small = (int)(small < 2 * L.E ? small : 2 * L.E);
    L.tile_splat = 1;                                            // every kernel below sees it (the struct travels by value)
    L.wtile = 1;
    L.est = (L.build != 2 && L.cap_small < 2 * L.E && rcf_cdiv(L.N, 256) >= 4 * PK_SAMPLES) ? 1 : 0;
    hipLaunchKernelGGL(pk_stat_reset_kernel, dim3(1), dim3(64), 0, st, L, F);
    if (L.est) PD_LAUNCH(L.pd, pk_estimate_kernel, dim3(PK_SAMPLES, F), dim3(256), st, L, rgb, W, posdev, featdev);
    hipLaunchKernelGGL(pk_clear_kernel, dim3(F, 256), dim3(256), 0, st, L, 0);
    PD_LAUNCH(L.pd, lattice_build_packed_kernel, gpf, dim3(256), st, L, rgb, W, H, posdev, featdev, 0);
    hipLaunchKernelGGL(pk_totals_kernel, dim3(F), dim3(256), 0, st, L, 0);
    if (L.cap_small < 2 * L.E) {           // frames that overflowed the small table: all 2E buckets (others return at once)
        hipLaunchKernelGGL(pk_clear_kernel, dim3(F, 256), dim3(256), 0, st, L, 1);
        PD_LAUNCH(L.pd, lattice_build_packed_kernel, dim3(F, 256), dim3(256), st, L, rgb, W, H, posdev, featdev, 1);
        hipLaunchKernelGGL(pk_totals_kernel, dim3(F), dim3(256), 0, st, L, 1);
    }
    const int gscan = nblk2 < 256 ? nblk2 : 256;
    hipLaunchKernelGGL(pk_scan_local_kernel, dim3(gscan, F), dim3(SCAN_BLOCK), 0, st, L, nblk2);
    hipLaunchKernelGGL(pk_scan_blocks_kernel, dim3(F), dim3(SCAN_BLOCK), 0, st, L, nblk2);
    hipLaunchKernelGGL(pk_scan_apply_kernel, dim3(gscan, F), dim3(SCAN_BLOCK), 0, st, L, nblk2);
    hipLaunchKernelGGL(pk_fill_kernel, dim3(F, 2048), dim3(256), 0, st, L);        // (returns at once on tile-mode frames)
    hipLaunchKernelGGL(tile_list_kernel, dim3(F, L.tiles), dim3(256), 0, st, L);      // one workgroup per tile: two dependent gathers per entry
    hipLaunchKernelGGL(neighbours_init_kernel, dim3(2048, F), dim3(256), 0, st, L);
    PD_LAUNCH(L.pd, pk_neighbours_kernel, dim3(F, 2048), dim3(256), st, L);
    RCF_LAUNCH_CHECK();
    return 0;
}

// RCF_CRF_BUILD_SORT: see "sort build" above.  Storage: the two key arrays alias `keys` (16 B per entry), the two value
// arrays `entries`, the run flags `slot_vid`, the scanned run numbers `rel`, the vertices' keys `slot_vid2` / `slot_off`.
int build_lattice_sorted(Lattice &L, const uint8_t *rgb, int W, int H, int F, float posdev, float featdev, hipStream_t st) {
    const size_t FE = (size_t)F * L.E;
    unsigned long long *K0 = reinterpret_cast<unsigned long long *>(L.keys), *K1 = K0 + FE;
    unsigned *V0 = reinterpret_cast<unsigned *>(L.entries), *V1 = V0 + FE;
    int *flag = L.slot_vid, *runs = L.rel;
    unsigned long long *ukey = reinterpret_cast<unsigned long long *>(L.slot_vid2);
    static_assert(sizeof(uint4) == 2 * sizeof(unsigned long long), "two key arrays in `keys`");
    PD_LAUNCH(L.pd, sort_keys_kernel, dim3(rcf_cdiv(L.N, 256), F), dim3(256), st, L, rgb, W, posdev, featdev, K0, V0);
    RCF_LAUNCH_CHECK();
    int fbits = 0;
    while ((1 << fbits) < F) ++fbits;
    if (int e = rcf_crf_sort_pairs_u64(L.sort_tmp, L.sort_tmp_bytes, K0, K1, V0, V1, FE, 60 + fbits, st)) return e;
    hipLaunchKernelGGL(sort_heads_kernel, dim3(4096), dim3(256), 0, st, L, (const unsigned long long *)K1, flag, (long)FE);
    RCF_LAUNCH_CHECK();
    if (int e = rcf_crf_inclusive_scan_i32(L.sort_tmp, L.sort_tmp_bytes, flag, runs, FE, st)) return e;
    hipLaunchKernelGGL(sort_scatter_kernel, dim3(F, 2048), dim3(256), 0, st, L, (const unsigned long long *)K1, (const unsigned *)V1,
                       (const int *)flag, (const int *)runs, ukey);
    hipLaunchKernelGGL(sort_cnt_kernel, dim3(F, 1024), dim3(256), 0, st, L);
    hipLaunchKernelGGL(neighbours_init_kernel, dim3(2048, F), dim3(256), 0, st, L);
    int *coarse = L.slot_off;                                   // [F][pd + 1][E / 256 + 1] of its 2 F E ints
    const int cstride = (int)(L.E / 256 + 1);
    PD_LAUNCH(L.pd, sort_neighbours_coarse_kernel, dim3(F, 256), dim3(256), st, L, (const unsigned long long *)ukey, coarse, cstride);
    PD_LAUNCH(L.pd, sort_neighbours_kernel, dim3(F, 4096), dim3(256), st, L, (const unsigned long long *)ukey, (const int *)coarse, cstride);
    RCF_LAUNCH_CHECK();
    return 0;
}

int build_lattice(Lattice &L, const uint8_t *rgb, int W, int H, int F, float posdev, float featdev, float weight,
                  hipStream_t st) {
    L.w = weight;
    L.tile_splat = 0;                                            // only the packed build writes the tile lists
    L.wtile = 0;                                                 // ... and stores the weights tile-major
    const dim3 gp(rcf_cdiv(L.N, 256), F), ge(rcf_cdiv(L.E, 256), F);
    const int nblk = scan_blocks(L.E);
    if (L.build != 1 && keys_fit_12bit(L.pd, W, H, posdev, featdev)) {
        // the sort build for the appearance lattice only (the position lattice has a few hundred vertices whatever the frame
        // shows); 60 key bits + up to 4 frame bits
        if (L.build == 3 && L.pd == 5 && F <= 16) {
            if (int e = build_lattice_sorted(L, rgb, W, H, F, posdev, featdev, st)) return e;
        } else if (int e = build_lattice_packed(L, rgb, W, H, F, posdev, featdev, st)) {
            return e;
        }
        return lattice_built(L, F, st);
    }
    CK(hipMemsetAsync(L.entries, 0xff, (size_t)F * 2 * L.E * sizeof(int), st));
    PD_LAUNCH(L.pd, lattice_keys_kernel, gp, dim3(256), st, L, rgb, W, H, posdev, featdev);
    hipLaunchKernelGGL(lattice_insert_kernel, ge, dim3(256), 0, st, L);
    hipLaunchKernelGGL(lattice_scan_local_kernel, dim3(nblk, F), dim3(SCAN_BLOCK), 0, st, L);
    hipLaunchKernelGGL(lattice_scan_blocks_kernel, dim3(F), dim3(SCAN_BLOCK), 0, st, L, nblk);
    hipLaunchKernelGGL(lattice_scan_apply_kernel, dim3(nblk, F), dim3(SCAN_BLOCK), 0, st, L);
    hipLaunchKernelGGL(lattice_entry_vid_kernel, ge, dim3(256), 0, st, L);
    hipLaunchKernelGGL(neighbours_init_kernel, dim3(2048, F), dim3(256), 0, st, L);
    hipLaunchKernelGGL(lattice_neighbours_kernel, dim3(2048, F), dim3(256), 0, st, L);
    CK(hipMemsetAsync(L.cnt, 0, (size_t)F * L.E * sizeof(int), st));
    hipLaunchKernelGGL(csr_count_kernel, ge, dim3(256), 0, st, L);
    hipLaunchKernelGGL(csr_scan_local_kernel, dim3(nblk, F), dim3(SCAN_BLOCK), 0, st, L);
    hipLaunchKernelGGL(csr_scan_blocks_kernel, dim3(F), dim3(SCAN_BLOCK), 0, st, L, nblk);
    hipLaunchKernelGGL(csr_scan_apply_kernel, dim3(nblk, F), dim3(SCAN_BLOCK), 0, st, L);
    hipLaunchKernelGGL(csr_fill_kernel, ge, dim3(256), 0, st, L);
    RCF_LAUNCH_CHECK();
    return lattice_built(L, F, st);
}

// The normaliser of a freshly built lattice comes out of its FIRST filter pass (apply_lattice: splat MODE 2, blur2_kernel,
// slice_kernel<PD, true>) -- one list walk, one set of neighbour reads and one slice less per lattice than the stand-alone pass
// below, same bits.  The symmetric normalisation needs it BEFORE the first pass (the marginals are scaled by it): stand-alone.
int lattice_built(Lattice &L, int F, hipStream_t st) {
    if (L.sym) return build_lattice_norm(L, F, st);
    L.norm_pending = 1;
    return 0;
}

// the splat of one filter pass: the per-tile sums of the frames in tile mode (the kernel returns at once on the others), then the
// vertex pass -- the list walk over the entries, or the sum of a vertex's few per-tile partial sums
template <int MODE>
void launch_splat(Lattice &L, int F, const float *Q, void *out, float *outz, hipStream_t st, bool tiles_done = false) {
    // tiles_done: the previous pass's slice already summed this pass's marginals per tile (slice_kernel SPLAT)
    if (L.tile_splat && !(L.tune & 16) && !tiles_done) hipLaunchKernelGGL(splat_tile_kernel<MODE>, dim3(F, L.tiles), dim3(256), 0, st, L, Q);
    // (frames, 768): measured over 256 ... 4096 workgroups per frame -- the list walk of natural frames 233 us per pass of 8 frames
    // against 240 at 4096, the tile-mode vertex pass 154 against 162 (a few 10^4 vertices per frame: most of 4096 x 256 lanes only
    // start and stop), noise frames the same at every size
    hipLaunchKernelGGL(splat_gather_kernel<MODE>, dim3(F, 768), dim3(256), 0, st, L, Q, out, outz);
}

// homogeneous channel: splat the weights, blur, slice -> per-pixel normaliser (once per lattice)
int build_lattice_norm(Lattice &L, int F, hipStream_t st) {
    const dim3 gp(F, rcf_cdiv(L.N, 256));                      // (frames, chunks): see splat_gather_kernel
    float *za = reinterpret_cast<float *>(L.val0), *zb = reinterpret_cast<float *>(L.val1);
    launch_splat<1>(L, F, nullptr, (void *)za, nullptr, st);
    for (int axis = 0; axis <= L.pd; axis++) {
        hipLaunchKernelGGL(blur_kernel<float>, dim3(F, 1024), dim3(256), 0, st, L, axis, (const float *)za, zb);
        float *t = za; za = zb; zb = t;
    }
    PD_LAUNCH(L.pd, slice_norm_kernel, gp, dim3(256), st, L, (const float *)za, L.sym);
    RCF_LAUNCH_CHECK();
    L.norm_pending = 0;
    return 0;
}

// tmp-free filter + Potts + softmax epilogue for one potential
#define SLICE_LAUNCH(NORMv, SPLATv, zptr)                                                                                        \
    do {                                                                                                                         \
        if (L.pd == 5) hipLaunchKernelGGL((slice_kernel<5, NORMv, SPLATv>), gp, dim3(256), 0, st, L, (const float2 *)a, unary, next, Qout, map, first, last, write_map, L.sym, (const float *)(zptr)); \
        else if (L.pd == 2) hipLaunchKernelGGL((slice_kernel<2, NORMv, SPLATv>), gp, dim3(256), 0, st, L, (const float2 *)a, unary, next, Qout, map, first, last, write_map, L.sym, (const float *)(zptr)); \
        else hipLaunchKernelGGL((slice_kernel<0, NORMv, SPLATv>), gp, dim3(256), 0, st, L, (const float2 *)a, unary, next, Qout, map, first, last, write_map, L.sym, (const float *)(zptr)); \
    } while (0)

// tiles_done: this pass's per-tile sums exist already (the previous pass's slice made them); fuse_next: this pass's slice makes the
// next pass's (single potential, not the last iteration, tile splat available: crf_infer)
int apply_lattice(Lattice &L, int F, const float *Q, const float *unary, float *next, float *Qout, short *map,
                  int first, int last, int write_map, hipStream_t st, bool tiles_done = false, bool fuse_next = false) {
    static const int GRIDS[8] = {1024, 256, 384, 512, 768, 2048, 128, 192};    // (L.tune >> 1) & 7: lab only
    const dim3 gv(F, GRIDS[(L.tune >> 1) & 7]), gp(F, rcf_cdiv(L.W, 16) * rcf_cdiv(L.N / L.W, 16));     // (frames, 16 x 16 tiles): see splat_gather_kernel
    const bool pairs = !(L.tune & 1);                          // RCF_CRF_BLUR_SEQUENTIAL: one launch per axis (tests, A/B)
    float2 *a = L.val0, *b = L.val1;
    if (L.norm_pending) {
        // the build's key array is dead by now: its 16 bytes per entry hold the two homogeneous-channel buffers
        float *za = reinterpret_cast<float *>(L.keys), *zb = za + (size_t)F * L.E;
        launch_splat<2>(L, F, Q, (void *)a, za, st);             // (never tiles_done: the first pass of a call)
        for (int axis = 0; axis <= L.pd; axis++) {
            if (pairs && axis + 1 <= L.pd) {
                hipLaunchKernelGGL(blur_pair_kernel<true>, gv, dim3(256), 0, st, L, axis, (const float2 *)a, b, (const float *)za, zb);
                ++axis;
            } else {
                hipLaunchKernelGGL(blur2_kernel, gv, dim3(256), 0, st, L, axis, (const float2 *)a, b, (const float *)za, zb);
            }
            float2 *t = a; a = b; b = t;
            float *tz = za; za = zb; zb = tz;
        }
        if (fuse_next) SLICE_LAUNCH(true, true, za);
        else SLICE_LAUNCH(true, false, za);
        RCF_LAUNCH_CHECK();
        L.norm_pending = 0;
        return 0;
    }
    launch_splat<0>(L, F, Q, (void *)a, nullptr, st, tiles_done);
    for (int axis = 0; axis <= L.pd; axis++) {
        if (pairs && axis + 1 <= L.pd) {
            hipLaunchKernelGGL(blur_pair_kernel<false>, gv, dim3(256), 0, st, L, axis, (const float2 *)a, b, (const float *)nullptr, (float *)nullptr);
            ++axis;
        } else {
            hipLaunchKernelGGL(blur_kernel<float2>, gv, dim3(256), 0, st, L, axis, (const float2 *)a, b);
        }
        float2 *t = a; a = b; b = t;
    }
    if (fuse_next) SLICE_LAUNCH(false, true, nullptr);
    else SLICE_LAUNCH(false, false, nullptr);
    RCF_LAUNCH_CHECK();
    return 0;
}

int crf_infer(const uint8_t *rgb, const float *unary, int W, int H, int F, float scomp_smooth, float sxy_smooth,
              float scomp_app, float sxy_app, float srgb_app, int iters, int16_t *out_map, float *q_out,
              int32_t *nvert, CrfBuffers &b, hipStream_t st, int sym = 0, int build = 0, const float *featf = nullptr) {
    b.smooth.sym = b.app.sym = sym;                                // per call, not per process: concurrent callers differ
    // float features are unbounded: the array-of-keys build (16-bit key coordinates, the reference's `short`) takes them
    b.smooth.tune = b.app.tune = build >> 4;
    build &= 3;
    b.smooth.build = b.app.build = featf ? 1 : build;
    b.smooth.featf = b.app.featf = featf;
    const bool has_s = scomp_smooth > 0.f && sxy_smooth > 0.f;     // torchcrf.cu:28
    const bool has_a = scomp_app > 0.f && sxy_app > 0.f;           // torchcrf.cu:41
    const long n = (long)F * W * H;
    if (has_s) if (int e = build_lattice(b.smooth, rgb, W, H, F, sxy_smooth, 1.f, scomp_smooth, st)) return e;
    if (has_a) if (int e = build_lattice(b.app, rgb, W, H, F, sxy_app, srgb_app, scomp_app, st)) return e;
    const int npot = (has_s ? 1 : 0) + (has_a ? 1 : 0);
    const bool direct = (iters == 0 || npot == 0);
    hipLaunchKernelGGL(softmax2_kernel, dim3(rcf_cdiv(n, 256)), dim3(256), 0, st, unary, b.cur, (short *)out_map, n,
                       -1.0f, direct ? 1 : 0);
    RCF_LAUNCH_CHECK();
    if (npot == 0 && iters > 0) {
        // no pairwise term: every step is softmax(-U) again
        iters = 0;
    }
    if (sym && iters > 0) {
        if (npot != 1) return RCF_EINVAL;                // the scaled marginals belong to ONE kernel's normaliser
        hipLaunchKernelGGL(scale_q_kernel, dim3(rcf_cdiv(n, 256)), dim3(256), 0, st, b.cur, (has_a ? b.app : b.smooth).inv, n);
        RCF_LAUNCH_CHECK();
    }
    // one potential: the slice of a pass hands its marginals straight to the next pass's per-tile sums (slice_kernel SPLAT) on the
    // frames in tile mode; tune bit 6 (RCF_CRF_SLICE_SPLAT_SEPARATE): two kernels as with two potentials (tests, A/B)
    Lattice &one = has_a ? b.app : b.smooth;
    const bool fuse = npot == 1 && one.tile_splat && !(one.tune & (16 | 64));
    for (int it = 0; it < iters; it++) {
        const int wm = (it == iters - 1) ? 1 : 0;
        const bool done = fuse && it > 0, nxt = fuse && it + 1 < iters;
        if (has_s) if (int e = apply_lattice(b.smooth, F, b.cur, unary, b.next, b.cur, (short *)out_map, 1, has_a ? 0 : 1, wm, st, done, nxt)) return e;
        if (has_a) if (int e = apply_lattice(b.app, F, b.cur, unary, b.next, b.cur, (short *)out_map, has_s ? 0 : 1, 1, wm, st, done, nxt)) return e;
    }
    if (q_out) CK(hipMemcpyAsync(q_out, b.cur, n * MLAB * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (nvert) {                                        // [F][2]: vertices of the position / appearance lattice (one launch)
        hipLaunchKernelGGL(vertex_counts_kernel, dim3(1), dim3(64), 0, st, has_s ? b.smooth.L : (const int *)nullptr,
                           has_a ? b.app.L : (const int *)nullptr, nvert, F);
        RCF_LAUNCH_CHECK();
    }
    return 0;
}

}  // namespace

extern "C" size_t rcf_crf_workspace_bytes(int W, int H, int batch) {
    if (W <= 0 || H <= 0 || batch <= 0) return 0;
    CrfBuffers b;
    return carve_all(nullptr, W, H, batch, b);
}

namespace {
int crf_soft_impl(const uint8_t *rgb, const float *unary, int W, int H, int batch, float scomp_smooth, float sxy_smooth,
                  float scomp_app, float sxy_app, float srgb_app, int iters, int sym, int16_t *out_map, float *q_out,
                  int32_t *nvert, void *workspace, size_t workspace_bytes, void *stream, int build = 0,
                  const float *featf = nullptr) {
    if ((!rgb && !featf) || !unary || !out_map || W <= 0 || H <= 0 || batch <= 0 || iters < 0) return RCF_EINVAL;
    if ((long)W * H * 6 >= (1L << 30)) return RCF_EINVAL;
    if (!workspace || workspace_bytes < rcf_crf_workspace_bytes(W, H, batch) || !rcf_aligned16(workspace)) return RCF_EWORKSPACE;
    CrfBuffers b;
    carve_all((char *)workspace, W, H, batch, b);
    return crf_infer(rgb, unary, W, H, batch, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters, out_map,
                     q_out, nvert, b, rcf_stream(stream), sym, build, featf);
}
}  // namespace

extern "C" int rcf_crf_soft(const uint8_t *rgb, const float *unary, int W, int H, int batch, float scomp_smooth,
                            float sxy_smooth, float scomp_app, float sxy_app, float srgb_app, int iters,
                            int16_t *out_map, float *q_out, int32_t *nvert, void *workspace, size_t workspace_bytes,
                            void *stream) {
    return crf_soft_impl(rgb, unary, W, H, batch, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters, 0, out_map,
                         q_out, nvert, workspace, workspace_bytes, stream);
}

/* normalization 0: rcf_crf_soft.  normalization 1: the symmetric kernel normalisation of Kraehenbuehl & Koltun's DenseCRF
 * (pydensecrf DenseCRF2D.addPairwiseBilateral / addPairwiseGaussian defaults, NORMALIZE_SYMMETRIC): the filter is
 * N^1/2 K N^1/2 with N = diag(1 / (K 1 + 1e-20)) instead of diag(1 / K 1) K -- what tools/pydenseCRF/crf.py:58-89 and
 * models/crf_head.py:62-91 (crf_cpu) compute.  Exactly one potential may be active in that mode.  The mode travels as a
 * parameter (no process-wide state): calls with different normalisations may run concurrently from several threads. */
extern "C" int rcf_crf_soft_ex(const uint8_t *rgb, const float *unary, int W, int H, int batch, float scomp_smooth,
                               float sxy_smooth, float scomp_app, float sxy_app, float srgb_app, int iters,
                               int normalization, int16_t *out_map, float *q_out, int32_t *nvert, void *workspace,
                               size_t workspace_bytes, void *stream) {
    // bits 8-9: lattice build (RCF_CRF_BUILD_*: tests and A/B measurements; identical results)
    // bits 10-17: iteration variants (RCF_CRF_BLUR_SEQUENTIAL; lab grids; RCF_CRF_SPLAT_*; RCF_CRF_SLICE_SPLAT_SEPARATE) -- they
    // travel to crf_infer above the build's two bits
    const int build = ((normalization >> 8) & 3) | (((normalization >> 10) & 0xff) << 4);
    normalization &= 0xff;
    if (normalization != 0 && normalization != 1) return RCF_EINVAL;
    return crf_soft_impl(rgb, unary, W, H, batch, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters,
                         normalization, out_map, q_out, nvert, workspace, workspace_bytes, stream, build);
}

/* rcf_crf_soft_ex on float colour features [batch,H,W,3]: torchcrf_cpp.crf_soft converts ANY rgbFeat dtype to float without
 * rounding (tools/torchCRF/src/torchcrf.cu:84-85), so a caller with non-integer features gets the lattice of exactly those
 * values.  Always the array-of-keys build (the packed builds assume the u8 range). */
extern "C" int rcf_crf_soft_f32(const float *rgbf, const float *unary, int W, int H, int batch, float scomp_smooth,
                                float sxy_smooth, float scomp_app, float sxy_app, float srgb_app, int iters,
                                int normalization, int16_t *out_map, float *q_out, int32_t *nvert, void *workspace,
                                size_t workspace_bytes, void *stream) {
    normalization &= 0xff;
    if (!rgbf || (normalization != 0 && normalization != 1)) return RCF_EINVAL;
    return crf_soft_impl(nullptr, unary, W, H, batch, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters,
                         normalization, out_map, q_out, nvert, workspace, workspace_bytes, stream, 1, rgbf);
}

extern "C" int rcf_crf_hard(const uint8_t *rgb, const int16_t *label, int W, int H, int batch, float scomp_smooth,
                            float sxy_smooth, float scomp_app, float sxy_app, float srgb_app, float confidence,
                            int iters, int16_t *out_map, float *q_out, int32_t *nvert, void *workspace,
                            size_t workspace_bytes, void *stream) {
    if (!rgb || !label || !out_map || W <= 0 || H <= 0 || batch <= 0 || iters < 0) return RCF_EINVAL;
    if (!(confidence > 0.f && confidence < 1.f)) return RCF_EINVAL;
    if (!workspace || workspace_bytes < rcf_crf_workspace_bytes(W, H, batch) || !rcf_aligned16(workspace)) return RCF_EWORKSPACE;
    CrfBuffers b;
    carve_all((char *)workspace, W, H, batch, b);
    hipStream_t st = rcf_stream(stream);
    const long n = (long)batch * W * H;
    // setUnaryEnergyFromLabel, densecrf_gpu.cu:84-143 (M = 2)
    hipLaunchKernelGGL(unary_from_label_kernel, dim3(rcf_cdiv(n, 256)), dim3(256), 0, st, (const short *)label,
                       b.unary_own, n, -logf(1.0f / MLAB), -logf((1.0f - confidence) / (MLAB - 1)), -logf(confidence));
    RCF_LAUNCH_CHECK();
    return crf_infer(rgb, b.unary_own, W, H, batch, scomp_smooth, sxy_smooth, scomp_app, sxy_app, srgb_app, iters,
                     out_map, q_out, nvert, b, st);
}

extern "C" int rcf_crf_prepare(const float *img_nchw, const float *mask, const float *mean3, const float *std3,
                               int unstandardize, float crf_scale, uint8_t *rgb_out, float *unary_out,
                               uint32_t *scratch, int batch, int H, int W, void *stream) {
    if (!img_nchw || !mask || !rgb_out || !unary_out || !scratch || batch <= 0 || H <= 0 || W <= 0) return RCF_EINVAL;
    if (unstandardize && (!mean3 || !std3)) return RCF_EINVAL;
    if (!(crf_scale > 0.f)) return RCF_EINVAL;
    hipStream_t st = rcf_stream(stream);
    const int HW = H * W;
    CK(hipMemsetAsync(scratch, 0, batch * sizeof(uint32_t), st));
    const dim3 g(rcf_cdiv(HW, 256), batch), g4(rcf_cdiv(rcf_cdiv(HW, 4), 256), batch);
    hipLaunchKernelGGL(mask_qmax_kernel, dim3(QMAX_BLOCKS, batch), dim3(256), 0, st, mask, crf_scale, (unsigned *)scratch, HW);
    hipLaunchKernelGGL(prepare_image_kernel, g4, dim3(256), 0, st, img_nchw, mean3, std3, unstandardize, rgb_out, HW);
    hipLaunchKernelGGL(prepare_unary_kernel, g, dim3(256), 0, st, mask, crf_scale, (const unsigned *)scratch, unary_out,
                       HW);
    RCF_LAUNCH_CHECK();
    return 0;
}

// RPN proposal path on the device: anchors + box decode + clip, per-image descending
// sort (bitonic, 64-bit score|index keys), bitmask NMS whose suppression scan runs in a
// single workgroup on the device (no host round trip), top-N gather and zero padding.
// Replaces rpn/proposal_layer.py:49-163 and nms/nms_cpu.py:6-34.
//
// Box arithmetic is fp32 with one rounding per operation (built with
// -ffp-contract=off) so IoU threshold decisions match the reference's CPU path bit
// for bit on the same boxes.
#include "common.h"
#include <algorithm>

namespace {

// ------------------------------------------------------------------ decode + clip
// One thread per anchor (b, y, x, a); NHWC inputs make (y,x,a) the memory order of both
// the scores and the deltas, which is exactly the reference's permute(0,2,3,1) order
// (proposal_layer.py:99-104).
__global__ void rpn_decode_kernel(const float* __restrict__ cls, int is_prob, const float* __restrict__ bbox,
                                  const float* __restrict__ im_info, const float* __restrict__ base, int B, int H,
                                  int W, int A, int stride, float* __restrict__ prop, float* __restrict__ score) {
    const long long n_per = (long long)H * W * A;
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= n_per * B) return;
    const int b = idx / n_per;
    const long long i = idx % n_per;
    const int a = i % A;
    const long long cell = i / A;
    const int x = cell % W, y = cell / W;
    // fg probability: rpn.py:69-71 reshapes to (B,2,A*H,W) and softmaxes the pair
    const float* cp = cls + ((long long)b * H * W + cell) * 2 * A;
    float fg;
    if (is_prob) {
        fg = cp[A + a];
    } else {
        float s0 = cp[a], s1 = cp[A + a], m = fmaxf(s0, s1);
        float e0 = expf(s0 - m), e1 = expf(s1 - m);
        fg = e1 / (e0 + e1);
    }
    score[idx] = fg;
    // anchor = base[a] + (x*stride, y*stride, x*stride, y*stride)   proposal_layer.py:81-95
    const float sx = (float)(x * stride), sy = (float)(y * stride);
    const float ax1 = base[4 * a] + sx, ay1 = base[4 * a + 1] + sy;
    const float ax2 = base[4 * a + 2] + sx, ay2 = base[4 * a + 3] + sy;
    const float* d = bbox + ((long long)b * H * W + cell) * 4 * A + 4 * a;
    // bbox_transform.py:77-103
    const float w = ax2 - ax1 + 1.0f, h = ay2 - ay1 + 1.0f;
    const float cx = ax1 + 0.5f * w, cy = ay1 + 0.5f * h;
    const float pcx = d[0] * w + cx, pcy = d[1] * h + cy;
    const float pw = (float)exp((double)d[2]) * w, ph = (float)exp((double)d[3]) * h;
    const float xmax = im_info[3 * b + 1] - 1.0f, ymax = im_info[3 * b] - 1.0f;
    // clip_boxes :125-133
    float4 o;
    o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), xmax);
    o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), ymax);
    o.z = fminf(fmaxf(pcx + 0.5f * pw, 0.f), xmax);
    o.w = fminf(fmaxf(pcy + 0.5f * ph, 0.f), ymax);
    *(float4*)(prop + 4 * idx) = o;
}

// ------------------------------------------------------------------ bitonic sort
// key = (order-preserving u32 of the score) << 32 | (0xFFFFFFFF - index): sorting keys
// DESCENDING gives descending score with ascending index on ties; padding keys are 0.
constexpr int SORT_TILE = 4096;     // u64 keys per LDS tile (32 KiB)
constexpr int SORT_THREADS = 512;

__device__ inline unsigned int f2ord(float f) {
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void sort_make_keys(const float* __restrict__ keys, int n, int P, unsigned long long* __restrict__ out) {
    const int seg = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    unsigned long long k = 0;
    if (i < n) k = ((unsigned long long)f2ord(keys[(long long)seg * n + i]) << 32) | (0xFFFFFFFFu - (unsigned)i);
    out[(long long)seg * P + i] = k;
}

__device__ inline void cmpswap_desc(unsigned long long& a, unsigned long long& b, bool desc) {
    if ((a < b) == desc) { unsigned long long t = a; a = b; b = t; }
}

// all stages with partner distance < SORT_TILE, for k from k_lo up to k_hi (inclusive)
__global__ void __launch_bounds__(SORT_THREADS)
sort_local(unsigned long long* __restrict__ data, int P, int k_lo, int k_hi) {
    __shared__ unsigned long long s[SORT_TILE];
    const long long base = (long long)blockIdx.y * P + (long long)blockIdx.x * SORT_TILE;
    for (int i = threadIdx.x; i < SORT_TILE; i += SORT_THREADS) s[i] = data[base + i];
    __syncthreads();
    const int gbase = blockIdx.x * SORT_TILE;
    for (int k = k_lo; k <= k_hi; k <<= 1) {
        int j0 = (k >> 1) < SORT_TILE ? (k >> 1) : (SORT_TILE >> 1);
        for (int j = j0; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < SORT_TILE / 2; t += SORT_THREADS) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));     // lower index of the pair
                bool desc = (((gbase + i) & k) == 0);
                cmpswap_desc(s[i], s[i | j], desc);
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < SORT_TILE; i += SORT_THREADS) data[base + i] = s[i];
}

__global__ void sort_global_step(unsigned long long* __restrict__ data, int P, int k, int j) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P / 2) return;
    unsigned long long* d = data + (long long)blockIdx.y * P;
    int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
    bool desc = ((i & k) == 0);
    unsigned long long a = d[i], b = d[i | j];
    if ((a < b) == desc) { d[i] = b; d[i | j] = a; }
}

__global__ void sort_emit_order(const unsigned long long* __restrict__ keys, int n, int P, int* __restrict__ order) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = keys[(long long)blockIdx.y * P + i];
    order[(long long)blockIdx.y * n + i] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
}

inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }
inline int sort_padded(int n) { int p = next_pow2(n); return p < SORT_TILE ? SORT_TILE : p; }

int launch_sort(const float* keys, int n_seg, int n, unsigned long long* buf, hipStream_t st) {
    const int P = sort_padded(n);
    sort_make_keys<<<dim3(i2v_cdiv(P, 256), n_seg), 256, 0, st>>>(keys, n, P, buf);
    dim3 tiles(P / SORT_TILE, n_seg);
    sort_local<<<tiles, SORT_THREADS, 0, st>>>(buf, P, 2, SORT_TILE);
    for (int k = SORT_TILE * 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= SORT_TILE; j >>= 1)
            sort_global_step<<<dim3(i2v_cdiv(P / 2, 256), n_seg), 256, 0, st>>>(buf, P, k, j);
        sort_local<<<tiles, SORT_THREADS, 0, st>>>(buf, P, k, k);
    }
    return P;
}

// ------------------------------------------------------------------ gather sorted dets
__global__ void gather_dets(const unsigned long long* __restrict__ keys, int P, const float* __restrict__ prop,
                            const float* __restrict__ score, int n_all, int n_top, float* __restrict__ dets,
                            int* __restrict__ src_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n_top) return;
    unsigned long long k = keys[(long long)b * P + i];
    int idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
    float4 p = *(const float4*)(prop + ((long long)b * n_all + idx) * 4);
    float* d = dets + ((long long)b * n_top + i) * 5;
    d[0] = p.x; d[1] = p.y; d[2] = p.z; d[3] = p.w; d[4] = score[(long long)b * n_all + idx];
    src_idx[(long long)b * n_top + i] = idx;
}

// ------------------------------------------------------------------ NMS
// (1) mask kernel: 64x64 tiles of the upper triangle; one wave per tile, lane = row.
//     nms_cpu.py:13,20-29: areas with +1, IoU = inter / (area_i + area_j - inter); a column suppresses when !(IoU <= thresh).
//
// The kernel is VALU-bound (72 M pairs at N = 12000: the loop was 40 instructions per pair, 11 of them the IEEE division, 4
// canonicalising moves in front of fmaxf / fminf, 6 for a variable 64-bit shift).  Round 5: the 64 columns are a compile-time
// loop (bit positions are constants), max / min are the hardware instructions, and the division is replaced by an EXACT test
// that needs it only inside a band of 2^-21 around the threshold:
//     ovr_f = fl(inter / d), d = fl(fl(area_i + area_j) - inter) > 0; rounding is monotone, so inter/d <= thresh implies
//     ovr_f <= thresh, and inter/d >= thresh (1 + 2^-22) implies ovr_f > thresh.  With thi = fl(thresh (1 + 2^-21)),
//     tlo = fl(thresh (1 - 2^-21)) (the 2^-21 absorbs the two roundings of each product):
//     inter < fl(tlo d)  =>  inter < thresh d  =>  not suppressed;
//     inter > fl(thi d)  =>  inter > thresh d (1 + 2^-22)  =>  suppressed;
//     otherwise (or a box with a non-positive side, or a NaN): the division, as before, after the loop.
//     The keep lists stay bit-exact against nms_cpu (golden vectors: 36 cases, uniform and clustered).
__device__ inline float hw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ inline float hw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

constexpr int NMS_MASK_WAVES = 1;        // row blocks (waves) per workgroup sharing one column block's boxes: measured 1 / 2 / 4 waves 47.2 / 47.6 / 52.5 us at N = 12000 (18.0 / 19.4 / 18.4 at 6000)

__global__ void __launch_bounds__(64 * NMS_MASK_WAVES)
nms_mask_kernel(const float* __restrict__ dets, int n, int nblk, float thresh,
                unsigned long long* __restrict__ mask) {
    const int cb = blockIdx.x, img = blockIdx.z;
    const int lane = threadIdx.x & 63, rb = blockIdx.y * NMS_MASK_WAVES + (threadIdx.x >> 6);
    if (cb < (int)blockIdx.y * NMS_MASK_WAVES) return;            // the whole group lies below the diagonal (uniform)
    __shared__ __attribute__((aligned(16))) float sc[5][64];      // column boxes, one field per row: x1, y1, x2, y2, area
    const float* D = dets + (long long)img * n * 5;
    const int ccount = min(64, n - cb * 64);
    __shared__ unsigned long long s_cbad;                         // columns whose box has a non-positive side (or a NaN): decided by the division
    if (threadIdx.x < 64) {
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
        if (lane < ccount) {
            const float* p = D + ((long long)cb * 64 + lane) * 5;
            c0 = p[0]; c1 = p[1]; c2 = p[2]; c3 = p[3];
        }
        const float cw = c2 - c0 + 1.0f, ch = c3 - c1 + 1.0f;
        sc[0][lane] = c0; sc[1][lane] = c1; sc[2][lane] = c2; sc[3][lane] = c3;
        sc[4][lane] = cw * ch;
        const unsigned long long bad = __ballot(!(cw > 0.f && ch > 0.f));
        if (lane == 0) s_cbad = bad;
    }
    __syncthreads();
    const int row = rb * 64 + lane;
    if (cb < rb || rb >= nblk || row >= n) return;
    const float* p = D + (long long)row * 5;
    const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    const float rw = x2 - x1 + 1.0f, rh = y2 - y1 + 1.0f;
    const float area = rw * rh;
    // For boxes with positive sides d = fl(fl(area_i + area_j) - inter) > 0 always (inter <= min of the two areas: the clipped
    // extents are not larger than either box's, and rounding is monotone), so the loop need not test it; a row or a column
    // with a non-positive side or a NaN goes to the division as a whole.
    const float thi = thresh * 1.000000476837158203125f, tlo = thresh * 0.999999523162841796875f;     // thresh (1 +- 2^-21)
    unsigned w[2] = {0u, 0u}, unc[2] = {0u, 0u};          // suppression bits; pairs the band test could not decide
#pragma unroll
    for (int j4 = 0; j4 < 64; j4 += 4) {
        const float4 X1 = *(const float4*)&sc[0][j4], Y1 = *(const float4*)&sc[1][j4], X2 = *(const float4*)&sc[2][j4],
                     Y2 = *(const float4*)&sc[3][j4], AR = *(const float4*)&sc[4][j4];
        const float cx1[4] = {X1.x, X1.y, X1.z, X1.w}, cy1[4] = {Y1.x, Y1.y, Y1.z, Y1.w}, cx2[4] = {X2.x, X2.y, X2.z, X2.w},
                    cy2[4] = {Y2.x, Y2.y, Y2.z, Y2.w}, car[4] = {AR.x, AR.y, AR.z, AR.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j4 + u;
            const unsigned bit = 1u << (j & 31);
            const float xx1 = hw_max(x1, cx1[u]), yy1 = hw_max(y1, cy1[u]);
            const float xx2 = hw_min(x2, cx2[u]), yy2 = hw_min(y2, cy2[u]);
            const float ww = hw_max(0.0f, xx2 - xx1 + 1.0f), hh = hw_max(0.0f, yy2 - yy1 + 1.0f);
            const float inter = ww * hh;
            const float d = area + car[u] - inter;
            // two compares, two selects; nothing in this loop branches
            const int over = inter > thi * d, under = inter < tlo * d;
            w[j >> 5] |= over ? bit : 0u;
            unc[j >> 5] |= (over | under) ? 0u : bit;
        }
    }
    {
        const unsigned long long bad = !(rw > 0.f && rh > 0.f) ? ~0ull : s_cbad;
        unc[0] |= (unsigned)bad; unc[1] |= (unsigned)(bad >> 32);
    }
    // the undecided pairs (inter within 2^-21 of thresh * d, or d <= 0 / NaN): the reference's own expression
    for (int h = 0; h < 2; ++h) {
        unsigned m = unc[h];
        while (m) {
            const int jj = __builtin_ctz(m), j = 32 * h + jj;
            m &= m - 1;
            const float xx1 = fmaxf(x1, sc[0][j]), yy1 = fmaxf(y1, sc[1][j]);
            const float xx2 = fminf(x2, sc[2][j]), yy2 = fminf(y2, sc[3][j]);
            const float ww = fmaxf(0.0f, xx2 - xx1 + 1.0f), hh = fmaxf(0.0f, yy2 - yy1 + 1.0f);
            const float inter = ww * hh;
            const float ovr = inter / (area + sc[4][j] - inter);
            if (!(ovr <= thresh)) w[h] |= 1u << jj;         // nms_cpu.py:31 keeps ovr <= thresh
            else w[h] &= ~(1u << jj);
        }
    }
    unsigned long long bits = ((unsigned long long)w[1] << 32) | w[0];
    if (cb == rb) bits &= lane == 63 ? 0ull : (~0ull << (lane + 1));      // the diagonal tile: columns after the row
    if (ccount < 64) bits &= (1ull << ccount) - 1ull;                                     // columns past the end
    mask[((long long)img * n + row) * nblk + cb] = bits;
}

// (2) scan kernel: one 1024-thread workgroup per image walks the 64-row blocks in score
//     order.  Wave 0 resolves the block's diagonal word serially (64 readlane steps);
//     all 16 waves then OR the mask rows of the rows just kept into the LDS "removed"
//     bitmap.  Stops after max_keep kept rows.  Replaces the host scan of
//     nms_cuda_kernel.cu:132-144 and the Python while-loop of nms_cpu.py:18-32.
constexpr int SCAN_THREADS = 1024;
constexpr int SCAN_MAX_BLK = 1024;      // n <= 65536

__global__ void __launch_bounds__(SCAN_THREADS)
nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int nblk, int max_keep,
                int* __restrict__ keep_out, int* __restrict__ num_out) {
    __shared__ unsigned long long removed[SCAN_MAX_BLK];
    __shared__ unsigned long long s_kept;
    __shared__ int s_count;
    const int img = blockIdx.x;
    const unsigned long long* M = mask + (long long)img * n * nblk;
    int* keep = keep_out + (long long)img * n;
    for (int i = threadIdx.x; i < nblk; i += SCAN_THREADS) removed[i] = 0;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int limit = max_keep > 0 ? max_keep : n;
    for (int rb = 0; rb < nblk; ++rb) {
        if (wave == 0) {
            const int row = rb * 64 + lane;
            unsigned long long diag = (row < n) ? M[(long long)row * nblk + rb] : 0ull;
            unsigned long long cur = removed[rb];
            const int valid = min(64, n - rb * 64);
            if (valid < 64) cur |= ~0ull << valid;
            unsigned long long kept = 0;
            int count = s_count;
            for (int i = 0; i < valid && count < limit; ++i) {
                unsigned long long d = __shfl(diag, i);
                if (!((cur >> i) & 1ull)) { kept |= 1ull << i; cur |= d; ++count; }
            }
            if ((kept >> lane) & 1ull) {
                int pos = s_count + __popcll(kept & ((1ull << lane) - 1ull));
                keep[pos] = row;
            }
            if (lane == 0) { s_kept = kept; s_count = count; }
        }
        __syncthreads();
        const unsigned long long kept = s_kept;
        const int count = s_count;
        if (count >= limit) break;
        // OR the kept rows' masks into the columns to the right of the diagonal
        for (int i = wave; i < 64; i += SCAN_THREADS / 64) {
            if (!((kept >> i) & 1ull)) continue;
            const unsigned long long* mr = M + (long long)(rb * 64 + i) * nblk;
            for (int j = rb + 1 + lane; j < nblk; j += 64) {
                unsigned long long v = mr[j];
                if (v) atomicOr(&removed[j], v);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) num_out[img] = s_count;
}

// Round 5 built and measured a form with the ROLES SPLIT (wave 0 only resolves and carries the next three verdict words of the
// rows it keeps in scalar registers; the other 15 waves fetch the KEPT rows' words only and fold them two blocks later; one
// barrier per block, then none: LDS counters polled both ways): bit-exact on every golden case and SLOWER -- 264 us with one
// barrier, 370-420 us polling, against 186 us for the form below (N = 12000 clustered; DESIGN.md 5.10 has the stamps).  Neither
// the barrier count nor a drained prefetch (LDS-only barriers: 185 us) is what a block's ~2k cycles are made of.
// Pipelined variant for n <= 12288 (every RPN case): the mask rows of block rb+1 are fetched into
// registers (4 rows per wave x 3 words per lane) while block rb is being resolved and folded in, so
// global-load latency is off the serial chain; the chain per 64-row block is the 64-step resolve in
// wave 0 plus two workgroup barriers.
constexpr int SCAN_PIPE_WORDS = 3;      // words per lane: covers nblk - 1 <= 192 columns

// a value every lane holds alike (read from LDS: the compiler must assume otherwise), moved to scalar registers
__device__ inline unsigned long long uniform64(unsigned long long v) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}

__global__ void __launch_bounds__(SCAN_THREADS)
nms_scan_pipelined_kernel(const unsigned long long* __restrict__ mask, int n, int nblk, int max_keep,
                          int* __restrict__ keep_out, int* __restrict__ num_out) {
    __shared__ unsigned long long removed[64 * SCAN_PIPE_WORDS + 1];
    __shared__ unsigned long long s_kept;
    __shared__ int s_count;
    const int img = blockIdx.x;
    const unsigned long long* M = mask + (long long)img * n * nblk;
    int* keep = keep_out + (long long)img * n;
    for (int i = threadIdx.x; i < nblk; i += SCAN_THREADS) removed[i] = 0;
    if (threadIdx.x == 0) s_count = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int limit = max_keep > 0 ? max_keep : n;
    unsigned long long pc[4][SCAN_PIPE_WORDS], pn[4][SCAN_PIPE_WORDS], dc = 0, dn = 0;
    // A row that is already suppressed when its block is fetched can never be kept, so its mask words are never used: they
    // are not fetched.  ``removed[rb]`` holds, at that point, the verdict of every block but the one being resolved; with
    // clustered proposals (12000 -> 2000 kept) most rows are gone by then, and the scan workgroup's ingest -- one CU pulling
    // 96 KB per block, 18 MB per image, was ~40 % of its time -- shrinks with them.  (Uniform per wave: a row is a wave's.)
    // The fetch itself must be cheap: 16 waves issue 13 loads each per block, and written as ``live ? M[row * nblk + j] : 0`` every
    // load carried a 64-bit multiply-add, a bounds compare and a branch of its own -- ~400 instructions per wave and block, four
    // waves per SIMD: the chain per block (2.4k cycles) was instruction ISSUE, not the resolve, the barriers or memory latency
    // (a deeper prefetch and a one-barrier form both measured slower: they added instructions).  Now: one buffer descriptor over
    // the image's mask (the launcher keeps it below 2 GiB), byte offsets carried from block to block (+ (64 nblk + 1) * 8), the
    // three words of a row through the instruction's immediate offset, a word past the end of its row = the 2 GiB bit (zeros, no traffic;
    // rows past n fall off the descriptor by themselves).
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)M, 0, (unsigned)((long long)n * nblk * 8), 0x00020000);
    const unsigned ostep = (unsigned)(64 * nblk + 1) * 8u;
    unsigned noff[4], ndiag = (unsigned)(lane * nblk) * 8u;          // offsets of the NEXT block to fetch
#pragma unroll
    for (int r = 0; r < 4; ++r) noff[r] = (unsigned)((wave * 4 + r) * nblk + 1 + lane) * 8u;
    auto fetch = [&](int rb, unsigned long long (&P)[4][SCAN_PIPE_WORDS], unsigned long long& diag) {
        const unsigned long long gone = removed[rb];
        const int wgone = __builtin_amdgcn_readfirstlane((int)((gone >> (wave * 4)) & 15ull));      // my four rows' verdicts: scalar
        unsigned past[SCAN_PIPE_WORDS];                                   // a word past the end of its row
#pragma unroll
        for (int c = 0; c < SCAN_PIPE_WORDS; ++c) past[c] = (rb + 1 + lane + 64 * c < nblk) ? 0u : 0x80000000u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < SCAN_PIPE_WORDS; ++c) {
                P[r][c] = 0ull;
                // a dead row or a column group past the end of the rows issues NO instruction (scalar branches): one CU's
                // memory pipe takes these 16 x 13 wave-loads per block one after the other -- issued unconditionally, with
                // the 2 GiB bit on the dead ones, the scan took 410 us instead of 280
                if (!((wgone >> r) & 1) && rb + 1 + 64 * c < nblk)
                    P[r][c] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(mrs, (noff[r] | past[c]) + 512u * c, 0, 0));
            }
            noff[r] += ostep;
        }
        if (wave == 0) {
            const unsigned dead = ((gone >> lane) & 1ull) ? 0x80000000u : 0u;
            diag = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(mrs, ndiag | dead, 0, 0));
        }
        ndiag += ostep;
    };
    __syncthreads();                                         // ``removed`` is clear (fetch reads it)
    fetch(0, pc, dc);
    // The wait for a block's words is written out, HERE and at the end of every iteration, where the loads have had a whole
    // iteration to land.  Left to the compiler it sat at the first use of dc / pc in the NEXT iteration -- behind the fetch that
    // iteration had just issued, and vector loads return in order: vmcnt(0) there waited for the NEW block's words, a full memory
    // round trip on the serial chain of every block (in-kernel clocks: 950 of a block's 2.5k cycles in wave 0's resolve, for
    // three kept rows).
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
    __syncthreads();
    for (int rb = 0; rb < nblk; ++rb) {
        if (rb + 1 < nblk) fetch(rb + 1, pn, dn);            // lands during this iteration
        if (wave == 0) {
            const int row = rb * 64 + lane;
            // ``removed[rb]`` and ``s_count`` come from LDS: without the readfirstlane the compiler keeps them (and everything
            // derived: todo, kept, count) in VECTOR registers and runs the loop below under exec masks -- 950 cycles per block
            // for three kept rows (in-kernel clocks), the largest phase of the chain
            unsigned long long cur = uniform64(removed[rb]);
            const int valid = min(64, n - rb * 64);
            if (valid < 64) cur |= ~0ull << valid;
            unsigned long long kept = 0;
            const int count0 = __builtin_amdgcn_readfirstlane(s_count);
            int count = count0;
            // serial greedy resolve of the 64 rows of this block, entirely in scalar registers: jump
            // to the next unsuppressed row with ffs, fetch its diagonal word with v_readlane
            const unsigned dlo = (unsigned)dc, dhi = (unsigned)(dc >> 32);
            unsigned long long todo = ~cur;
            while (todo && count < limit) {
                const int i = __builtin_ctzll(todo);
                const unsigned long long d = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                                             (unsigned)__builtin_amdgcn_readlane((int)dlo, i);
                kept |= 1ull << i;
                ++count;
                cur |= d;
                todo = ~cur & ~((2ull << i) - 1ull);       // rows after i that are still alive
            }
            if ((kept >> lane) & 1ull) keep[count0 + __popcll(kept & ((1ull << lane) - 1ull))] = row;
            if (lane == 0) { s_kept = kept; s_count = count; }
        }
        __syncthreads();
        const unsigned long long kept = uniform64(s_kept);
        if (__builtin_amdgcn_readfirstlane(s_count) >= limit) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (!((kept >> (wave * 4 + r)) & 1ull)) continue;                  // scalar
#pragma unroll
            for (int c = 0; c < SCAN_PIPE_WORDS; ++c) {
                const int j = rb + 1 + lane + 64 * c;
                if (j < nblk && pc[r][c]) atomicOr(&removed[j], pc[r][c]);
            }
        }
        __syncthreads();
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): the next block's words, requested an iteration ago
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < SCAN_PIPE_WORDS; ++c) pc[r][c] = pn[r][c];
        dc = dn;
    }
    if (threadIdx.x == 0) num_out[img] = s_count;
}

// Round 6 (the one algorithmic experiment round 5's review left open): the scan in SUPER-BLOCKS of 1024 rows whose 1024 x 1024
// triangle of the mask (16 words per row, 128 KB) sits in LDS.  The workgroup meets a handful of times per 1024 rows instead of
// twice per 64: to resolve the super-block, to fold the kept rows' words for LATER columns into the ``removed`` bitmap (kept rows
// only, from the mask in global memory: sixteen waves, a 64-row block each) and to bring the next triangle in (requested a
// super-block ahead, into registers).
//   ``fixed_point`` > 0 (the default): the super-block is resolved by fixed-point sweeps over the triangle, at most that many;
//   ``fixed_point`` = 0, or a super-block that has not settled: ONE wave walks the sixteen 64-row blocks serially -- the verdict
//   words of the super-block's own columns live in its lanes (lane w = word w), a block's rows are resolved in scalar registers
//   as in the pipelined kernel, the rows just kept are folded into those lanes by one LDS read each.  Alone this form is no
//   faster than the pipelined kernel (181.8 against 185.2 us): what the scan costs is the resolve's ~190 cycles of dependent
//   scalar instructions per KEPT row, not the barriers, the fetch or the fold.
// Either way the greedy keep set, row by row: bit-exact.
constexpr int SUPER_ROWS = 1024, SUPER_WORDS = 16, NMS_SWEEPS = 48;

__global__ void __launch_bounds__(SCAN_THREADS)
nms_scan_super_kernel(const unsigned long long* __restrict__ mask, int n, int nblk, int max_keep,
                      int* __restrict__ keep_out, int* __restrict__ num_out, int fixed_point /* sweeps at most; 0: serial resolve */) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long tri[];       // [1024][16]
    __shared__ unsigned long long removed[64 * SCAN_PIPE_WORDS + SUPER_WORDS + 1];
    __shared__ unsigned long long s_kept[SUPER_WORDS];
    __shared__ int s_count;
    const int img = blockIdx.x;
    const unsigned long long* M = mask + (long long)img * n * nblk;
    int* keep = keep_out + (long long)img * n;
    for (int i = threadIdx.x; i < 64 * SCAN_PIPE_WORDS + SUPER_WORDS + 1; i += SCAN_THREADS) removed[i] = 0;
    if (threadIdx.x == 0) s_count = 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int limit = max_keep > 0 ? max_keep : n;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc((void*)M, 0, (unsigned)((long long)n * nblk * 8), 0x00020000);
    const int nsuper = (n + SUPER_ROWS - 1) / SUPER_ROWS;
    // the triangle of super-block sb, 16 of its 16384 words per thread: pass p brings rows 64 p .. 64 p + 63, a thread one word of
    // one row (16 lanes = one row's 128 contiguous bytes: four cache lines per wave instruction.  One ROW per thread -- 64 lines
    // per instruction -- made the fetch 8 of a super-block's 12.5 us).  Words left of the row's own block were never written by
    // nms_mask_kernel (below the diagonal), words past the end of the row do not exist: both read as zero (the 2 GiB bit); a
    // row past n falls off the descriptor by itself
    unsigned long long tw[SUPER_WORDS];
    auto fetch_tri = [&](int sb) {
        const int w = (int)threadIdx.x & (SUPER_WORDS - 1), wa = sb * SUPER_WORDS + w;
#pragma unroll
        for (int p = 0; p < SUPER_ROWS / 64; ++p) {
            const int row = sb * SUPER_ROWS + 64 * p + ((int)threadIdx.x >> 4), rb = row >> 6;
            const unsigned dead = (row < n && wa >= rb && wa < nblk) ? 0u : 0x80000000u;
            tw[p] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(mrs, ((unsigned)(row * nblk + wa) * 8u) | dead, 0, 0));
        }
    };
    fetch_tri(0);
    __syncthreads();                                         // ``removed`` is clear
    for (int sb = 0; sb < nsuper; ++sb) {
        const int r0 = sb * SUPER_ROWS, w0 = sb * SUPER_WORDS;
        __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): my row's words, requested a super-block ago
#pragma unroll
        for (int p = 0; p < SUPER_ROWS / 64; ++p) tri[(64 * p) * SUPER_WORDS + threadIdx.x] = tw[p];
        __syncthreads();
        if (sb + 1 < nsuper) fetch_tri(sb + 1);              // lands while this super-block is resolved
        // ---- fixed-point resolve (round 5's review: "cluster-NMS"): keep <- alive & ~OR_{j kept} row_j, from keep = alive, until it
        // stops moving.  A row's verdict is final once the rows before it are (induction from row 0), so the fixed point is the
        // greedy keep set, reached after as many sweeps as the longest suppress-release chain in the 1024 rows is deep; a sweep is
        // 16 LDS reads per thread, two shuffles and one LDS atomic per 16 lanes.  The serial resolve a kept row at a time costs
        // ~190 cycles of one wave's dependent scalar instructions per KEPT row (2000 of them: the 182 us of either scan kernel).
        // Not settled after NMS_SWEEPS sweeps: the serial resolve below does the super-block (same answer, by construction).
        int settled = 0;
        if (fixed_point) {
            __shared__ unsigned long long s_alive[SUPER_WORDS], s_keepw[SUPER_WORDS], s_supp[SUPER_WORDS];
            __shared__ int s_changed;
            if (threadIdx.x < SUPER_WORDS) {
                const int rb = w0 + (int)threadIdx.x;
                unsigned long long a = 0ull;
                if (rb < nblk) {
                    a = ~removed[rb];
                    const int valid = n - rb * 64;
                    if (valid < 64) a &= (1ull << valid) - 1ull;
                }
                s_alive[threadIdx.x] = a; s_keepw[threadIdx.x] = a; s_supp[threadIdx.x] = 0ull;
            }
            __syncthreads();
            const int w = threadIdx.x & (SUPER_WORDS - 1), g = threadIdx.x >> 4;         // my word; my 16 rows (16 g .. 16 g + 15)
            for (int sweep = 0; sweep < fixed_point; ++sweep) {
                const unsigned kb = (unsigned)(s_keepw[g >> 2] >> ((g & 3) * 16)) & 0xFFFFu;
                unsigned long long acc = 0ull;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned long long v = tri[(16 * g + r) * SUPER_WORDS + w];
                    acc |= ((kb >> r) & 1u) ? v : 0ull;
                }
                acc |= __shfl_xor(acc, 16);
                acc |= __shfl_xor(acc, 32);
                if (lane < SUPER_WORDS && acc) atomicOr(&s_supp[w], acc);
                if (threadIdx.x == 0) s_changed = 0;
                __syncthreads();
                if (threadIdx.x < SUPER_WORDS) {
                    const unsigned long long nk = s_alive[threadIdx.x] & ~s_supp[threadIdx.x];
                    if (nk != s_keepw[threadIdx.x]) s_changed = 1;
                    s_keepw[threadIdx.x] = nk;
                    s_supp[threadIdx.x] = 0ull;
                }
                __syncthreads();
                if (!s_changed) { settled = 1; break; }
            }
            if (settled) {
                // the keep list: rows in order, cut at the limit
                const int count0 = s_count;
                int before = 0;
                for (int q = 0; q < wave; ++q) before += __popcll(s_keepw[q]);
                const unsigned long long mine = s_keepw[wave];
                const int rank = before + __popcll(mine & ((1ull << lane) - 1ull));
                const bool k = ((mine >> lane) & 1ull) && count0 + rank < limit;
                __syncthreads();                             // everyone has read s_count / s_keepw
                if (k) keep[count0 + rank] = r0 + 64 * wave + lane;
                const unsigned long long cut = __ballot(k);
                if (lane == 0) s_kept[wave] = cut;
                if (threadIdx.x == 0) {
                    int total = 0;
                    for (int q = 0; q < SUPER_WORDS; ++q) total += __popcll(s_keepw[q]);
                    s_count = min(limit, count0 + total);
                }
            }
        }
        if (!settled && wave == 0) {
            // the verdicts on this super-block's columns so far: lane w holds word w0 + w
            unsigned long long rem = lane < SUPER_WORDS ? removed[min(w0 + lane, nblk)] : 0ull;
            int count = __builtin_amdgcn_readfirstlane(s_count);
            for (int q = 0; q < SUPER_WORDS && r0 + 64 * q < n; ++q) {
                const int rb = w0 + q;
                unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rem >> 32), q) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane((int)(unsigned)rem, q);
                const int valid = min(64, n - rb * 64);
                if (valid < 64) cur |= ~0ull << valid;
                const unsigned long long dq = tri[(64 * q + lane) * SUPER_WORDS + q];       // my row's diagonal word
                const unsigned dlo = (unsigned)dq, dhi = (unsigned)(dq >> 32);
                unsigned long long kept = 0, todo = ~cur;
                const int count0 = count;
                while (todo && count < limit) {              // serial greedy resolve in scalar registers (as the pipelined kernel's)
                    const int i = __builtin_ctzll(todo);
                    const unsigned long long d = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
                                                 (unsigned)__builtin_amdgcn_readlane((int)dlo, i);
                    kept |= 1ull << i;
                    ++count;
                    cur |= d;
                    todo = ~cur & ~((2ull << i) - 1ull);
                }
                if ((kept >> lane) & 1ull) keep[count0 + __popcll(kept & ((1ull << lane) - 1ull))] = rb * 64 + lane;
                if (lane == 0) s_kept[q] = kept;
                // the kept rows' words for the REST of this super-block's columns: one LDS read per row, all issued before the
                // first is used (lane w reads word w; words up to q are zero or already applied)
                unsigned long long k2 = kept;
                while (k2) {
                    unsigned long long v[8];
                    int m = 0;
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        v[u] = 0ull;
                        if (k2) {
                            const int i = __builtin_ctzll(k2);
                            k2 &= k2 - 1ull;
                            v[u] = tri[(64 * q + i) * SUPER_WORDS + (lane & (SUPER_WORDS - 1))];
                            ++m;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) rem |= v[u];
                    (void)m;
                }
                if (count >= limit) {                        // the blocks behind keep nothing
                    for (int q2 = q + 1; q2 < SUPER_WORDS; ++q2) if (lane == 0) s_kept[q2] = 0ull;
                    break;
                }
            }
            for (int q2 = (n - r0 + 63) / 64; q2 < SUPER_WORDS; ++q2) if (lane == 0 && q2 >= 0) s_kept[q2] = 0ull;
            if (lane == 0) s_count = count;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(s_count) >= limit || sb + 1 == nsuper) break;
        // fold: the kept rows of block w0 + wave, their words for the columns after this super-block (kept rows only; sixteen
        // rows' loads in flight per pass), OR-ed in registers and added to the bitmap once per wave
        {
            unsigned long long kk = uniform64(s_kept[wave]);
            unsigned long long acc[SCAN_PIPE_WORDS];
            unsigned past[SCAN_PIPE_WORDS];
#pragma unroll
            for (int c = 0; c < SCAN_PIPE_WORDS; ++c) { acc[c] = 0ull; past[c] = (w0 + SUPER_WORDS + lane + 64 * c < nblk) ? 0u : 0x80000000u; }
            while (kk) {
                unsigned long long v[8][SCAN_PIPE_WORDS];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
#pragma unroll
                    for (int c = 0; c < SCAN_PIPE_WORDS; ++c) v[u][c] = 0ull;
                    if (kk) {
                        const int i = __builtin_ctzll(kk);
                        kk &= kk - 1ull;
                        const unsigned off = (unsigned)((r0 + 64 * wave + i) * nblk + w0 + SUPER_WORDS + lane) * 8u;
#pragma unroll
                        for (int c = 0; c < SCAN_PIPE_WORDS; ++c)
                            if (w0 + SUPER_WORDS + 64 * c < nblk)
                                v[u][c] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(mrs, (off | past[c]) + 512u * c, 0, 0));
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < SCAN_PIPE_WORDS; ++c) acc[c] |= v[u][c];
            }
#pragma unroll
            for (int c = 0; c < SCAN_PIPE_WORDS; ++c) {
                const int j = w0 + SUPER_WORDS + lane + 64 * c;
                if (j < nblk && acc[c]) atomicOr(&removed[j], acc[c]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) num_out[img] = s_count;
}

// ------------------------------------------------------------------ emit rois
__global__ void write_rois(const float* __restrict__ dets, const int* __restrict__ src_idx,
                           const int* __restrict__ keep, const int* __restrict__ num, int n_top, int post,
                           float* __restrict__ rois, int* __restrict__ kept_idx, int* __restrict__ num_kept) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (k >= post) return;
    const int cnt = min(num[b], post);
    float* r = rois + ((long long)b * post + k) * 5;
    r[0] = (float)b;
    int src = -1;
    if (k < cnt) {
        const int row = keep[(long long)b * n_top + k];
        const float* d = dets + ((long long)b * n_top + row) * 5;
        r[1] = d[0]; r[2] = d[1]; r[3] = d[2]; r[4] = d[3];
        src = src_idx[(long long)b * n_top + row];
    } else {
        r[1] = r[2] = r[3] = r[4] = 0.f;
    }
    if (kept_idx) kept_idx[(long long)b * post + k] = src;
    if (num_kept && k == 0) num_kept[b] = cnt;
}

// ------------------------------------------------------------------ IoU vs ground truth
// bbox_transform.py:168-257.  One thread per box row, loops the K gt boxes (K <= 64).
__global__ void bbox_overlaps_kernel(const float* __restrict__ boxes, int box_stride, int box_off, int batched,
                                     const float* __restrict__ gt, int N, int K, float* __restrict__ ov,
                                     float* __restrict__ max_ov, int* __restrict__ arg_ov) {
    extern __shared__ float sg[];          // K x 6: x1,y1,x2,y2,area,zero
    const int b = blockIdx.y;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float* g = gt + ((long long)b * K + k) * 5;
        float gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f;
        sg[k * 6] = g[0]; sg[k * 6 + 1] = g[1]; sg[k * 6 + 2] = g[2]; sg[k * 6 + 3] = g[3];
        sg[k * 6 + 4] = gw * gh;
        sg[k * 6 + 5] = (gw == 1.0f && gh == 1.0f) ? 1.f : 0.f;
    }
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* p = boxes + ((long long)(batched ? b : 0) * N + i) * box_stride + box_off;
    const float x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    const float bw = x2 - x1 + 1.0f, bh = y2 - y1 + 1.0f;
    const float area = bw * bh;
    const bool bzero = (bw == 1.0f && bh == 1.0f);
    float best = -INFINITY;
    int besti = 0;
    for (int k = 0; k < K; ++k) {
        float iw = fminf(x2, sg[k * 6 + 2]) - fmaxf(x1, sg[k * 6]) + 1.0f;
        float ih = fminf(y2, sg[k * 6 + 3]) - fmaxf(y1, sg[k * 6 + 1]) + 1.0f;
        iw = iw < 0.f ? 0.f : iw;
        ih = ih < 0.f ? 0.f : ih;
        float inter = iw * ih;
        float o = inter / (area + sg[k * 6 + 4] - inter);
        if (sg[k * 6 + 5] != 0.f) o = 0.f;
        if (bzero) o = -1.f;
        if (ov) ov[((long long)b * N + i) * K + k] = o;
        if (o > best) { best = o; besti = k; }
    }
    if (max_ov) max_ov[(long long)b * N + i] = best;
    if (arg_ov) arg_ov[(long long)b * N + i] = besti;
}

struct ProposalWs {
    float* prop; float* score; unsigned long long* keys; float* dets; int* src; int* keep; int* num;
    unsigned long long* mask; size_t total;
};

ProposalWs carve(void* ws, int B, long long n_all, int n_top) {
    ProposalWs w;
    char* p = (char*)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* q = p ? (void*)(p + off) : nullptr; off += i2v_align(bytes); return q; };
    const int P = sort_padded((int)n_all);
    const int nblk = (n_top + 63) / 64;
    w.prop = (float*)take((size_t)B * n_all * 16);
    w.score = (float*)take((size_t)B * n_all * 4);
    w.keys = (unsigned long long*)take((size_t)B * P * 8);
    w.dets = (float*)take((size_t)B * n_top * 20);
    w.src = (int*)take((size_t)B * n_top * 4);
    w.keep = (int*)take((size_t)B * n_top * 4);
    w.num = (int*)take((size_t)B * 4);
    w.mask = (unsigned long long*)take((size_t)B * n_top * nblk * 8);
    w.total = off;
    return w;
}

int launch_nms(const float* dets, int n_img, int n, float thresh, int max_keep, int* keep, int* num,
               unsigned long long* mask, hipStream_t st) {
    const int nblk = (n + 63) / 64;
    nms_mask_kernel<<<dim3(nblk, i2v_cdiv(nblk, NMS_MASK_WAVES), n_img), 64 * NMS_MASK_WAVES, 0, st>>>(dets, n, nblk, thresh, mask);
    static bool once = [] {
        (void)hipFuncSetAttribute((const void*)nms_scan_super_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SUPER_ROWS * SUPER_WORDS * 8);
        return true;
    }();
    (void)once;
    if (nblk - 1 <= 64 * SCAN_PIPE_WORDS && g_i2v_tuning[I2V_TUNE_NMS_SCAN] && (long long)n * nblk * 8 < (1ll << 31))
        nms_scan_super_kernel<<<n_img, SCAN_THREADS, SUPER_ROWS * SUPER_WORDS * 8, st>>>(mask, n, nblk, max_keep, keep, num,
                                                                                         g_i2v_tuning[I2V_TUNE_NMS_SCAN] < 2 ? 0 : g_i2v_tuning[I2V_TUNE_NMS_SCAN] == 2 ? NMS_SWEEPS : g_i2v_tuning[I2V_TUNE_NMS_SCAN]);
    else if (nblk - 1 <= 64 * SCAN_PIPE_WORDS)
        nms_scan_pipelined_kernel<<<n_img, SCAN_THREADS, 0, st>>>(mask, n, nblk, max_keep, keep, num);
    else
        nms_scan_kernel<<<n_img, SCAN_THREADS, 0, st>>>(mask, n, nblk, max_keep, keep, num);
    return 0;
}


// ------------------------------------------------------------------ detection post-processing (eval)
// test_net_instance_styleD_bilinear.py:151-221 for one image, all foreground classes at once: de-normalise the
// regression deltas, decode against the rois, clip, undo the image scale, per class threshold / sort / NMS, then the
// image-wide top-`max_per_image` cut.  The reference does 1 + (n_classes-1) host NMS calls per frame.
__global__ void det_decode_kernel(const float* __restrict__ rois, const float* __restrict__ prob,
                                  const float* __restrict__ pred, int agnostic, int normalize, float4 stds, float4 means,
                                  float im_h, float im_w, float scale, const float* __restrict__ info, int R, int C,
                                  float thresh, float4* __restrict__ boxes, float* __restrict__ keys, int* __restrict__ n_valid) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (C - 1) * R) return;
    if (info) { im_h = info[0]; im_w = info[1]; scale = info[2]; }       // the frame's im_info row, on the device (a captured eval step)
    const int seg = idx / R, i = idx % R, j = seg + 1;
    const float* d = pred + (long long)i * (agnostic ? 4 : 4 * C) + (agnostic ? 0 : 4 * j);
    float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
    if (normalize) {          // box_deltas.view(-1, 4) * stds + means, one rounding per op (:158-163)
        d0 = d0 * stds.x + means.x; d1 = d1 * stds.y + means.y;
        d2 = d2 * stds.z + means.z; d3 = d3 * stds.w + means.w;
    }
    const float* a = rois + 5 * (long long)i + 1;
    // bbox_transform.py:77-103
    const float w = a[2] - a[0] + 1.0f, h = a[3] - a[1] + 1.0f;
    const float cx = a[0] + 0.5f * w, cy = a[1] + 0.5f * h;
    const float pcx = d0 * w + cx, pcy = d1 * h + cy;
    const float pw = (float)exp((double)d2) * w, ph = (float)exp((double)d3) * h;
    const float xmax = im_w - 1.0f, ymax = im_h - 1.0f;
    float4 o;                  // clip_boxes :125-133, then pred_boxes /= scale (:171)
    o.x = fminf(fmaxf(pcx - 0.5f * pw, 0.f), xmax) / scale;
    o.y = fminf(fmaxf(pcy - 0.5f * ph, 0.f), ymax) / scale;
    o.z = fminf(fmaxf(pcx + 0.5f * pw, 0.f), xmax) / scale;
    o.w = fminf(fmaxf(pcy + 0.5f * ph, 0.f), ymax) / scale;
    boxes[idx] = o;
    const float sc = prob[(long long)i * C + j];
    const bool ok = sc > thresh;                      // :182
    keys[idx] = ok ? sc : -INFINITY;
    if (ok) atomicAdd(n_valid + seg, 1);
}

// rows in descending-score order; rows past the class's valid count become far-away unit boxes that overlap nothing
__global__ void det_gather_kernel(const float4* __restrict__ boxes, const float* __restrict__ keys,
                                  const int* __restrict__ order, const int* __restrict__ n_valid, int R, int C,
                                  float* __restrict__ dets) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (C - 1) * R) return;
    const int seg = idx / R, r = idx % R;
    float* o = dets + 5 * (long long)idx;
    if (r < n_valid[seg]) {
        const int src = seg * R + order[idx];
        const float4 b = boxes[src];
        o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = keys[src];
    } else {
        const float x = -1.0e8f - 64.0f * (float)r;
        o[0] = x; o[1] = -1.0e8f; o[2] = x + 1.0f; o[3] = -1.0e8f + 1.0f; o[4] = -INFINITY;
    }
}

// one workgroup per class: kept rows (in order) that are real detections -> tmp, their scores -> all_scores
__global__ void det_compact_kernel(const float* __restrict__ dets, const int* __restrict__ keep,
                                   const int* __restrict__ num, const int* __restrict__ n_valid, int R,
                                   float* __restrict__ tmp, float* __restrict__ all_scores, int* __restrict__ cnt,
                                   int* __restrict__ total) {
    const int seg = blockIdx.x;
    __shared__ int pos;
    if (threadIdx.x == 0) pos = 0;
    __syncthreads();
    const int nk = num[seg], nv = n_valid[seg];
    // keep[] is ascending (rows are visited in score order), so the real detections are a prefix of it
    int mine = 0;
    for (int t = threadIdx.x; t < nk; t += blockDim.x) mine += keep[(long long)seg * R + t] < nv;
    atomicAdd(&pos, mine);
    __syncthreads();
    const int n = pos;
    for (int t = threadIdx.x; t < R; t += blockDim.x) {
        float sc = -INFINITY;
        if (t < n) {
            const float* src = dets + 5 * ((long long)seg * R + keep[(long long)seg * R + t]);
            float* dst = tmp + 5 * ((long long)seg * R + t);
            dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[4] = src[4];
            sc = src[4];
        }
        all_scores[(long long)seg * R + t] = sc;
    }
    if (threadIdx.x == 0) { cnt[seg] = n; atomicAdd(total, n); }
}

// image_thresh = the max_per_image-th largest kept score when more than that many were kept (:214-221)
__global__ void det_final_kernel(const float* __restrict__ tmp, const int* __restrict__ cnt, const int* __restrict__ total,
                                 const float* __restrict__ all_scores, const int* __restrict__ order2, int max_per_image,
                                 int R, int C, float* __restrict__ out, int* __restrict__ counts) {
    const int j = blockIdx.x;               // class, 0 = background (never reported)
    if (j == 0) { if (threadIdx.x == 0) counts[0] = 0; return; }
    const int seg = j - 1;
    float image_thresh = -INFINITY;
    if (max_per_image > 0 && *total > max_per_image) image_thresh = all_scores[order2[max_per_image - 1]];
    __shared__ int n_out;
    if (threadIdx.x == 0) n_out = 0;
    __syncthreads();
    const int n = cnt[seg];
    // scores within a class are descending: the rows that pass are a prefix
    int mine = 0;
    for (int t = threadIdx.x; t < n; t += blockDim.x) mine += tmp[5 * ((long long)seg * R + t) + 4] >= image_thresh;
    atomicAdd(&n_out, mine);
    __syncthreads();
    for (int t = threadIdx.x; t < n_out; t += blockDim.x) {
        const float* src = tmp + 5 * ((long long)seg * R + t);
        float* dst = out + 5 * ((long long)j * R + t);
        dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3]; dst[4] = src[4];
    }
    if (threadIdx.x == 0) counts[j] = n_out;
}

// ------------------------------------------------------------------ relation triplet ranking (eval)
// lib/utils.py:584-628 (detection_output): rel_prob[i][r] * conf[ixs[i]] * conf[ixo[i]] (two fp32 roundings, as numpy
// evaluates float32_row * python_float * python_float), then the top k of the flattened (pair, predicate) grid.
__global__ void rel_scale_kernel(const float* __restrict__ rel, const float* __restrict__ conf,
                                 const long long* __restrict__ ixs, const long long* __restrict__ ixo, int n_pairs,
                                 int n_rel, float* __restrict__ prob) {
    const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (idx >= (long long)n_pairs * n_rel) return;
    const int i = (int)(idx / n_rel);
    float v = rel[idx] * conf[ixs[i]];
    v = v * conf[ixo[i]];
    prob[idx] = v;
}
__global__ void rel_emit_kernel(const float* __restrict__ prob, const int* __restrict__ order, int k, int n_rel,
                                int* __restrict__ pair, int* __restrict__ pred, float* __restrict__ out_conf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= k) return;
    const int flat = order[t];
    pair[t] = flat / n_rel;
    pred[t] = flat % n_rel;
    out_conf[t] = prob[flat];
}
}  // namespace

extern "C" size_t i2v_nms_workspace_bytes(int32_t n_img, int32_t n) {
    if (n_img <= 0 || n <= 0) return 256;
    return i2v_align((size_t)n_img * n * ((n + 63) / 64) * 8);
}

extern "C" int32_t i2v_nms_sorted(const float* dets, int32_t n_img, int32_t n, float thresh, int32_t max_keep,
                                  int32_t* keep_out, int32_t* num_out, void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(n_img > 0 && n >= 0, "nms_sorted: bad shape");
    I2V_CHECK_ARG(num_out, "nms_sorted: null num_out");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {            // nms_wrapper.py:15-16: empty input -> empty keep
        hipMemsetAsync(num_out, 0, sizeof(int) * n_img, st);
        return I2V_OK;
    }
    I2V_CHECK_ARG(dets && keep_out, "nms_sorted: null pointer");
    I2V_CHECK_ARG(n <= 64 * SCAN_MAX_BLK, "nms_sorted: n > 65536 unsupported");
    if (ws_bytes < i2v_nms_workspace_bytes(n_img, n) || !ws) {
        i2v_set_error("nms_sorted: workspace %zu < %zu", ws_bytes, i2v_nms_workspace_bytes(n_img, n));
        return I2V_ERR_WORKSPACE;
    }
    launch_nms(dets, n_img, n, thresh, max_keep, keep_out, num_out, (unsigned long long*)ws, st);
    I2V_CHECK_LAUNCH("nms_sorted");
    return I2V_OK;
}

extern "C" int32_t i2v_rpn_decode(const float* cls, int32_t is_prob, const float* bbox, const float* im_info,
                                  const float* base, int32_t B, int32_t H, int32_t W, int32_t A, int32_t stride,
                                  float* prop, float* score, void* stream) {
    I2V_CHECK_ARG(cls && bbox && im_info && base && prop && score, "rpn_decode: null pointer");
    I2V_CHECK_ARG(B > 0 && H > 0 && W > 0 && A > 0 && A <= 64, "rpn_decode: bad shape");
    long long total = (long long)B * H * W * A;
    rpn_decode_kernel<<<i2v_cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(cls, is_prob, bbox, im_info, base, B, H,
                                                                              W, A, stride, prop, score);
    I2V_CHECK_LAUNCH("rpn_decode");
    return I2V_OK;
}

extern "C" size_t i2v_sort_desc_workspace_bytes(int32_t n_seg, int32_t n) {
    if (n_seg <= 0 || n <= 0) return 256;
    return i2v_align((size_t)n_seg * sort_padded(n) * 8);
}

extern "C" int32_t i2v_sort_desc(const float* keys, int32_t n_seg, int32_t n, int32_t* order, void* ws,
                                 size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(n_seg > 0 && n >= 0, "sort_desc: bad shape");
    if (n == 0) return I2V_OK;
    I2V_CHECK_ARG(keys && order, "sort_desc: null pointer");
    I2V_CHECK_ARG(n <= (1 << 24), "sort_desc: n too large");
    if (!ws || ws_bytes < i2v_sort_desc_workspace_bytes(n_seg, n)) {
        i2v_set_error("sort_desc: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    int P = launch_sort(keys, n_seg, n, (unsigned long long*)ws, st);
    sort_emit_order<<<dim3(i2v_cdiv(n, 256), n_seg), 256, 0, st>>>((unsigned long long*)ws, n, P, order);
    I2V_CHECK_LAUNCH("sort_desc");
    return I2V_OK;
}

extern "C" size_t i2v_rpn_proposal_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t A, int32_t pre) {
    if (B <= 0 || H <= 0 || W <= 0 || A <= 0) return 256;
    long long n_all = (long long)H * W * A;
    // proposal_layer.py:140: truncate only when 0 < pre < B*N (numel of the whole batch)
    int n_top = (pre > 0 && pre < (long long)B * n_all && pre < n_all) ? pre : (int)n_all;
    return carve(nullptr, B, n_all, n_top).total;
}

extern "C" int32_t i2v_rpn_proposal(const float* cls, int32_t is_prob, const float* bbox, const float* im_info,
                                    const float* base, int32_t B, int32_t H, int32_t W, int32_t A, int32_t stride,
                                    int32_t pre, int32_t post, float thresh, float* rois, int32_t* kept_idx,
                                    int32_t* num_kept, void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(cls && bbox && im_info && base && rois, "rpn_proposal: null pointer");
    I2V_CHECK_ARG(B > 0 && H > 0 && W > 0 && A > 0 && A <= 64 && post > 0, "rpn_proposal: bad shape");
    const long long n_all = (long long)H * W * A;
    I2V_CHECK_ARG(n_all <= 64 * SCAN_MAX_BLK, "rpn_proposal: more than 65536 anchors per image");
    const int n_top = (pre > 0 && pre < (long long)B * n_all && pre < n_all) ? pre : (int)n_all;
    if (!ws || ws_bytes < carve(nullptr, B, n_all, n_top).total) {
        i2v_set_error("rpn_proposal: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    ProposalWs w = carve(ws, B, n_all, n_top);
    rpn_decode_kernel<<<i2v_cdiv(B * n_all, 256), 256, 0, st>>>(cls, is_prob, bbox, im_info, base, B, H, W, A,
                                                               stride, w.prop, w.score);
    const int P = launch_sort(w.score, B, (int)n_all, w.keys, st);
    gather_dets<<<dim3(i2v_cdiv(n_top, 256), B), 256, 0, st>>>(w.keys, P, w.prop, w.score, (int)n_all, n_top,
                                                               w.dets, w.src);
    launch_nms(w.dets, B, n_top, thresh, post, w.keep, w.num, w.mask, st);
    write_rois<<<dim3(i2v_cdiv(post, 256), B), 256, 0, st>>>(w.dets, w.src, w.keep, w.num, n_top, post, rois,
                                                             kept_idx, num_kept);
    I2V_CHECK_LAUNCH("rpn_proposal");
    return I2V_OK;
}

extern "C" int32_t i2v_bbox_overlaps(const float* boxes, int32_t box_stride, int32_t box_off, int32_t batched,
                                     const float* gt, int32_t B, int32_t N, int32_t K, float* ov, float* max_ov,
                                     int32_t* arg_ov, void* stream) {
    I2V_CHECK_ARG(boxes && gt, "bbox_overlaps: null pointer");
    I2V_CHECK_ARG(B > 0 && N >= 0 && K > 0 && K <= 1024 && box_stride >= box_off + 4, "bbox_overlaps: bad shape");
    if (N == 0) return I2V_OK;
    bbox_overlaps_kernel<<<dim3(i2v_cdiv(N, 256), B), 256, (size_t)K * 24, (hipStream_t)stream>>>(
        boxes, box_stride, box_off, batched, gt, N, K, ov, max_ov, arg_ov);
    I2V_CHECK_LAUNCH("bbox_overlaps");
    return I2V_OK;
}

namespace {
struct DetWs {
    float4* boxes; float* keys; int* order; float* dets; int* n_valid; int* keep; int* num; float* tmp;
    float* all_scores; int* order2; int* cnt; int* total; void* sort_ws; size_t sort_ws_bytes; void* nms_ws;
    size_t nms_ws_bytes; size_t bytes;
};
DetWs det_carve(char* base, int R, int C) {
    DetWs w;
    const size_t n = (size_t)(C - 1) * R;
    size_t off = 0;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += i2v_align(b); return p; };
    w.boxes = (float4*)take(n * 16); w.keys = (float*)take(n * 4); w.order = (int*)take(n * 4);
    w.dets = (float*)take(n * 20); w.n_valid = (int*)take((size_t)C * 4); w.keep = (int*)take(n * 4);
    w.num = (int*)take((size_t)C * 4); w.tmp = (float*)take(n * 20); w.all_scores = (float*)take(n * 4);
    w.order2 = (int*)take(n * 4); w.cnt = (int*)take((size_t)C * 4); w.total = (int*)take(256);
    w.sort_ws_bytes = std::max(i2v_sort_desc_workspace_bytes(C - 1, R), i2v_sort_desc_workspace_bytes(1, (int)n));
    w.sort_ws = take(w.sort_ws_bytes);
    w.nms_ws_bytes = i2v_nms_workspace_bytes(C - 1, R);
    w.nms_ws = take(w.nms_ws_bytes);
    w.bytes = off;
    return w;
}
}  // namespace

extern "C" size_t i2v_det_postprocess_workspace_bytes(int32_t R, int32_t C) {
    if (R <= 0 || C <= 1) return 256;
    return det_carve(nullptr, R, C).bytes;
}

static int32_t det_postprocess(const float* rois, const float* cls_prob, const float* bbox_pred,
                               int32_t class_agnostic, const float* stds, const float* means, float im_h,
                               float im_w, float im_scale, const float* info, int32_t R, int32_t C, float score_thresh,
                               float nms_thresh, int32_t max_per_image, float* dets, int32_t* counts, void* ws,
                               size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(rois && cls_prob && bbox_pred && dets && counts, "det_postprocess: null pointer");
    I2V_CHECK_ARG(R > 0 && C > 1 && (info || im_scale > 0.f), "det_postprocess: bad shape");
    I2V_CHECK_ARG((stds == nullptr) == (means == nullptr), "det_postprocess: stds and means go together");
    if (!ws || ws_bytes < i2v_det_postprocess_workspace_bytes(R, C)) {
        i2v_set_error("det_postprocess: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    DetWs w = det_carve((char*)ws, R, C);
    const int n = (C - 1) * R;
    hipMemsetAsync(w.n_valid, 0, sizeof(int) * C, st);
    hipMemsetAsync(w.total, 0, sizeof(int), st);
    const float4 sd = stds ? make_float4(stds[0], stds[1], stds[2], stds[3]) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 mn = means ? make_float4(means[0], means[1], means[2], means[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    det_decode_kernel<<<i2v_cdiv(n, 256), 256, 0, st>>>(rois, cls_prob, bbox_pred, class_agnostic, stds != nullptr, sd, mn,
                                                        im_h, im_w, im_scale, info, R, C, score_thresh, w.boxes, w.keys,
                                                        w.n_valid);
    int rc = i2v_sort_desc(w.keys, C - 1, R, w.order, w.sort_ws, w.sort_ws_bytes, stream);
    if (rc) return rc;
    det_gather_kernel<<<i2v_cdiv(n, 256), 256, 0, st>>>(w.boxes, w.keys, w.order, w.n_valid, R, C, w.dets);
    rc = i2v_nms_sorted(w.dets, C - 1, R, nms_thresh, 0, w.keep, w.num, w.nms_ws, w.nms_ws_bytes, stream);
    if (rc) return rc;
    det_compact_kernel<<<C - 1, 256, 0, st>>>(w.dets, w.keep, w.num, w.n_valid, R, w.tmp, w.all_scores, w.cnt, w.total);
    if (max_per_image > 0) {
        rc = i2v_sort_desc(w.all_scores, 1, n, w.order2, w.sort_ws, w.sort_ws_bytes, stream);
        if (rc) return rc;
    }
    det_final_kernel<<<C, 256, 0, st>>>(w.tmp, w.cnt, w.total, w.all_scores, w.order2, max_per_image, R, C, dets, counts);
    I2V_CHECK_LAUNCH("det_postprocess");
    return I2V_OK;
}

extern "C" int32_t i2v_det_postprocess(const float* rois, const float* cls_prob, const float* bbox_pred,
                                       int32_t class_agnostic, const float* stds, const float* means, float im_h,
                                       float im_w, float im_scale, int32_t R, int32_t C, float score_thresh,
                                       float nms_thresh, int32_t max_per_image, float* dets, int32_t* counts, void* ws,
                                       size_t ws_bytes, void* stream) {
    return det_postprocess(rois, cls_prob, bbox_pred, class_agnostic, stds, means, im_h, im_w, im_scale, nullptr, R, C,
                           score_thresh, nms_thresh, max_per_image, dets, counts, ws, ws_bytes, stream);
}

extern "C" int32_t i2v_det_postprocess_info(const float* rois, const float* cls_prob, const float* bbox_pred,
                                            int32_t class_agnostic, const float* stds, const float* means,
                                            const float* im_info, int32_t R, int32_t C, float score_thresh,
                                            float nms_thresh, int32_t max_per_image, float* dets, int32_t* counts,
                                            void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(im_info, "det_postprocess_info: null im_info");
    return det_postprocess(rois, cls_prob, bbox_pred, class_agnostic, stds, means, 0.f, 0.f, 0.f, im_info, R, C,
                           score_thresh, nms_thresh, max_per_image, dets, counts, ws, ws_bytes, stream);
}

extern "C" size_t i2v_relation_topk_workspace_bytes(int32_t n_pairs, int32_t n_rel) {
    if (n_pairs <= 0 || n_rel <= 0) return 256;
    const size_t n = (size_t)n_pairs * n_rel;
    return i2v_align(n * 4) + i2v_align(n * 4) + i2v_sort_desc_workspace_bytes(1, (int)n);
}

extern "C" int32_t i2v_relation_topk(const float* rel_score, const float* conf, const int64_t* ixs, const int64_t* ixo,
                                     int32_t n_pairs, int32_t n_rel, int32_t k, int32_t* pair_out, int32_t* pred_out,
                                     float* conf_out, void* ws, size_t ws_bytes, void* stream) {
    I2V_CHECK_ARG(rel_score && conf && ixs && ixo && pair_out && pred_out && conf_out, "relation_topk: null pointer");
    I2V_CHECK_ARG(n_pairs > 0 && n_rel > 0 && k > 0 && (long long)n_pairs * n_rel <= (1 << 24), "relation_topk: bad shape");
    I2V_CHECK_ARG((long long)k <= (long long)n_pairs * n_rel, "relation_topk: k exceeds the number of (pair, predicate) cells");
    if (!ws || ws_bytes < i2v_relation_topk_workspace_bytes(n_pairs, n_rel)) {
        i2v_set_error("relation_topk: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int n = n_pairs * n_rel;
    char* base = (char*)ws;
    float* prob = (float*)base;
    int* order = (int*)(base + i2v_align((size_t)n * 4));
    void* sws = base + 2 * i2v_align((size_t)n * 4);
    rel_scale_kernel<<<i2v_cdiv(n, 256), 256, 0, st>>>(rel_score, conf, (const long long*)ixs, (const long long*)ixo, n_pairs,
                                                       n_rel, prob);
    int rc = i2v_sort_desc(prob, 1, n, order, sws, i2v_sort_desc_workspace_bytes(1, n), stream);
    if (rc) return rc;
    rel_emit_kernel<<<i2v_cdiv(k, 256), 256, 0, st>>>(prob, order, k, n_rel, pair_out, pred_out, conf_out);
    I2V_CHECK_LAUNCH("relation_topk");
    return I2V_OK;
}

// fp32 implicit-GEMM convolution / linear layers on the CDNA4 matrix cores.
//
//   y[m][n] = epi( sum_k  A[m][k] * Wt[n][k] )      m = (b,oy,ox) output pixel
//                                                     k = (ky,kx,c) filter tap, c fastest
// A is never materialised: each workgroup gathers its BM x 16 slice of the im2col
// matrix straight from the NHWC activation (16 B per lane, channel-contiguous), stages
// it in LDS next to the BN x 16 weight slice, and the waves feed
// v_mfma_f32_32x32x2_f32 (exact fp32, fp32 accumulate) from ds_read_b128 fragments.
// The frozen-BatchNorm scale/shift, the bias, the residual add and the ReLU are fused
// into the accumulator epilogue, so a Bottleneck is 3-4 launches instead of ~10.
//
// LDS tile rows are 16 floats + 4 pad (80 B): a 16-lane ds_read_b128 group then hits 16
// distinct 16-B slots of the 256-B bank row (row*5 mod 16 is a bijection) - conflict
// free.  K order inside a 8-deep step is permuted (lane half h owns k = 4h..4h+3) so a
// fragment is ONE b128 read; A and B use the same permutation, so the sum is unchanged.
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BK = 16;          // k per LDS stage of the wgrad kernel
constexpr int KTAB_MAX = 2560;  // filter-tap table entries (K/4): KH*KW*Cin <= 10240 for non-1x1 filters
constexpr int BKS = 32;         // k per LDS stage of the forward/dgrad kernel (= floats per LDS row)
constexpr int LDS_ROW = 20;     // floats per staged row (16 + 4 pad)
constexpr int THREADS = 256;
constexpr int kSplitInKernelMax = 4;   // most splits the in-kernel split-K finish sums (else: atomics)

struct ConvP {
    const float* x; const float* w; const float* scale; const float* shift; const float* res; float* y;
    const float* mask;           // I2V_EPI_MASK: y = mask > 0 ? y : 0, same shape as y (the ReLU of the tensor a data gradient flows into)
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo;    // pad = top padding; may be negative (a crop)
    int pad_x;                   // left padding (run_conv callers set both; the sub-filters of a strided dgrad differ)
    int M, N, K;                 // GEMM sizes
    int flags;
    int splitk, k_per_split;     // k_per_split multiple of BK
    int ostride;                 // output pixel stride (dgrad of strided 1x1): y is (B,Ho*os..,Wo*os..,N)
    int Hy, Wy;                  // spatial size of the y buffer
    int lgCin;                   // log2(Cin) if power of two else -1
    int force_tile;              // >=0: tile config override (tuning / tests), -1: cost model
    unsigned x_bytes, w_bytes;   // sizes of x and w for the buffer descriptors (< 2 GiB each)
    int ablate;                  // diagnostic (i2v_conv_set_tile bits 10-11): 1 = skip staging in the K loop, 2 = skip MFMAs
    int ktab_entries;            // tap-table entries in LDS (>= 1; K/4 rounded up to whole stages for KxK filters)
    int dry;                     // plan only: run_conv returns the chosen split-K factor instead of launching
    unsigned long long* clk;     // diagnostic only (i2v_conv_debug_clock): per-workgroup {shader cycles, 100 MHz ticks}
    float* ws;                   // split-K partial tiles [split][tile][BM*BN] (nullptr: fp32 atomics into y)
    int* cnt;                    // split-K arrival counters, one per tile, zero between launches
    int nbatch;                  // > 1: blockIdx.z selects one of nbatch independent GEMMs (Winograd planes)
    long long bsx, bsw, bsy;     // element strides between the batches of x, w and y
};

__device__ inline void split_k(const ConvP& p, int k, int& ky, int& kx, int& c) {
    int kpos;
    if (p.lgCin >= 0) { kpos = k >> p.lgCin; c = k & (p.Cin - 1); }
    else { kpos = k / p.Cin; c = k - kpos * p.Cin; }
    if (p.KW == 1) { ky = kpos; kx = 0; }
    else { ky = kpos / p.KW; kx = kpos - ky * p.KW; }
}

// Tile = (WAVES_M*TM*16) x (WAVES_N*TN*16) outputs, 4 waves, v_mfma_f32_16x16x4_f32.
// A 32-deep K stage is staged per buffer; a lane (i = lane&15, g = lane>>4) owns k = 4g..4g+3 of
// each 16-deep half, so one ds_read_b128 feeds four MFMAs (A and B use the same permutation).
// LDS rows are 128 B, unpadded, with the 16-B column XOR-swizzled by (row>>1)&7: the four
// 16-lane groups of a ds_read_b128 then touch 16 distinct slots of the 256-B bank row.
template <int WAVES_M, int WAVES_N, int TM, int TN>
__global__ void __launch_bounds__(THREADS)
conv_igemm_f32(const ConvP p_in) {
    ConvP p = p_in;
    if (p.nbatch > 1) {              // batched GEMM: same shapes, different operands (uniform: blockIdx.z)
        p.x += (long long)blockIdx.z * p.bsx;
        p.w += (long long)blockIdx.z * p.bsw;
        p.y += (long long)blockIdx.z * p.bsy;
    }
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    constexpr int NT = THREADS;      // threads per workgroup
    constexpr int A_LD = (BM * 8 + THREADS - 1) / THREADS, B_LD = (BN * 8 + THREADS - 1) / THREADS;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    // one LDS block: [A stage 0 | A stage 1 | B stage 0 | B stage 1 | tap table]; after the K loop the
    // same bytes stage the BM x BN output tile (row stride BN+4 floats) for a coalesced epilogue
    constexpr int STAGE_FLOATS = 2 * (BM + BN) * BKS;       // + the tap table (p.ktab_entries), dynamic
    constexpr int CROW = BN + 4;
    constexpr int SMEM_FLOATS = STAGE_FLOATS > BM * CROW ? STAGE_FLOATS : BM * CROW;
    // dynamic: max(stage buffers + tap table of THIS filter, epilogue tile).  A 3x3x256 filter needs 2.3 KB of
    // table, not the 10 KB worst case: 39 KB per workgroup instead of 47 -> four workgroups per CU instead of three
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*As)[BM * BKS] = reinterpret_cast<float (*)[BM * BKS]>(smem);
    float (*Bs)[BN * BKS] = reinterpret_cast<float (*)[BN * BKS]>(smem + 2 * BM * BKS);
    unsigned* ktab = reinterpret_cast<unsigned*>(smem + 2 * (BM + BN) * BKS);

    const int gtid = threadIdx.x;                 // 0..NT-1
    unsigned long long t0c = 0, t0r = 0;
    if (p.clk) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    const int tid = gtid & (THREADS - 1);         // staging role / epilogue lane
    const int lane = gtid & 63, wave = (gtid >> 6) & 3;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tiles_n = (p.N + BN - 1) / BN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share one), so
    // give each XCD a contiguous run of tiles -- the n-tiles of one m-tile then share an L2 and the
    // activation rows cross the fabric once instead of once per XCD.  (Bijective for any grid size;
    // placement only affects speed.)
    int tile;
    {
        const int nt = gridDim.x, q = nt >> 3, r = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    // staging roles: slot = tid + q*256 -> row = slot>>3, 16-B column = slot&7 (= tid&7 for every q)
    const int kc = tid & 7, kg = kc * 4;
    const bool is1x1 = (p.KH == 1 && p.KW == 1 && p.pad == 0 && p.pad_x == 0);
    const unsigned m1x1 = is1x1 ? 0xFFFFFFFFu : 0u;
    // filter-tap table (only for KHxKW > 1): entry e = k/4 -> (byte offset of tap (ky,kx,c)) << 6 | tap id.
    // Built once per workgroup, so the K loop has no integer division and no per-tap bounds math.
    if (!is1x1) {
        const int n_e = p.ktab_entries;
        for (int e = gtid; e < n_e; e += NT) {
            unsigned v = 0;
            if ((e << 2) < p.K) {
                int ky, kx, c;
                split_k(p, e << 2, ky, kx, c);
                v = (unsigned)((((ky * p.W + kx) * p.Cin + c) * 4) << 6) | (unsigned)(ky * p.KW + kx);
            }
            ktab[e] = v;
        }
    }
    // Loads go through buffer resources: a masked lane gets an out-of-range offset and the hardware
    // returns zeros -- no branches, no select-of-loads, 32-bit byte offsets instead of 64-bit pointers.
    // INV (2 GiB) + any in-tile delta stays beyond every buffer (< 2 GiB), so rows outside the tile need
    // no per-stage test at all.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned INV = 0x80000000u, OOB = 0xFFFFFFF0u;
    unsigned a_off4[A_LD];            // byte offset of x[b][iy0][ix0][0] (mod 2^32: padded taps are masked)
    unsigned a_mlo[A_LD], a_mhi[A_LD];  // bit t: tap t of this output pixel reads inside the image
    const bool ident = is1x1 && p.stride == 1 && p.Ho == p.H && p.Wo == p.W;      // no index arithmetic at all
#pragma unroll
    for (int q = 0; q < A_LD; ++q) {
        const int row = (tid >> 3) + q * (THREADS / 8);
        const int m = m0 + row;
        const bool ok = row < BM && m < p.M;
        const int mm = ok ? m : 0;
        if (ident) {                  // pointwise, stride 1, same grid: input pixel index == output pixel index
            a_off4[q] = ok ? (unsigned)(mm * p.Cin) * 4u : INV;
            a_mlo[q] = ok ? 1u : 0u;
            a_mhi[q] = 0u;
            continue;
        }
        const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, b = t / p.Ho;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad_x;
        // 1x1 filters have no tap mask: a window that starts outside the input (possible for the sub-filters of
        // a strided dgrad, whose output grid can overhang gy) reads zeros through an invalid row offset
        const bool inside = !is1x1 || (iy0 >= 0 && ix0 >= 0 && iy0 < p.H && ix0 < p.W);
        a_off4[q] = (ok && inside) ? (unsigned)(((b * p.H + iy0) * p.W + ix0) * p.Cin) * 4u : INV;
        unsigned long long mask = 0;
        if (ok && is1x1 && inside) mask = 1;
        if (ok && !is1x1) {
            // taps (ky,kx) inside the image form a rectangle: kx in [kx_lo,kx_hi) for ky in [ky_lo,ky_hi)
            const int kx_lo = max(0, -ix0), kx_hi = min(p.KW, p.W - ix0);
            const int ky_lo = max(0, -iy0), ky_hi = min(p.KH, p.H - iy0);
            if (kx_hi > kx_lo) {
                const unsigned long long rowbits = ((1ull << kx_hi) - 1ull) & ~((1ull << kx_lo) - 1ull);
                for (int ky = ky_lo; ky < ky_hi; ++ky) mask |= rowbits << (ky * p.KW);
            }
        }
        a_mlo[q] = (unsigned)mask;
        a_mhi[q] = (unsigned)(mask >> 32);
    }
    unsigned b_off4[B_LD];
#pragma unroll
    for (int q = 0; q < B_LD; ++q) {
        const int row = (tid >> 3) + q * (THREADS / 8);
        const int n = n0 + row;
        b_off4[q] = (row < BN && n < p.N) ? (unsigned)(n * p.K) * 4u : INV;
    }
    if (!is1x1) __syncthreads();      // ktab visible

    // one stage of global loads into a register set.  Masking is pure integer arithmetic (OR-ing INV
    // into the offset): with `cond ? offset : OOB` the compiler threads the condition into divergent
    // branches that each hold a copy of the load writing the same registers, and guards the second copy
    // with s_waitcnt vmcnt(0) -- which drains every stage still in flight.
    // Fast path (whole stages of the 4-wave kernel): the k-dependent part of every address is UNIFORM across
    // the workgroup -- k0*4 for 1x1 filters and for the weights, the tap's byte offset for KxK filters whose
    // Cin is a multiple of 32 (a stage is then 32 channels of ONE tap) -- so it rides in the scalar offset
    // operand of the buffer load and the per-lane offsets are loop invariants: no address VALU in the K loop
    // (it was ~45 of the ~64 vector instructions per stage, and vector instructions of co-resident waves
    // delay MFMA issue).  The buffer base is moved back by the largest negative halo offset so that the
    // per-lane part is never negative (the hardware range-checks it before adding the scalar part).
    const bool tap_uni = !is1x1 && (p.Cin % BKS) == 0 && p.KH * p.KW <= 32;
    const unsigned halo = (unsigned)max(0, (p.pad * p.W + p.pad_x) * p.Cin) * 4u;
    const __amdgpu_buffer_rsrc_t xrb =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.x - halo), 0, p.x_bytes + halo, 0x00020000);
    unsigned a_vk[A_LD], b_vk[B_LD];
#pragma unroll
    for (int q = 0; q < A_LD; ++q) a_vk[q] = a_off4[q] == INV ? INV : a_off4[q] + halo + (unsigned)kg * 4u;
#pragma unroll
    for (int q = 0; q < B_LD; ++q) b_vk[q] = b_off4[q] == INV ? INV : b_off4[q] + (unsigned)kg * 4u;
    auto stage_load = [&](float4 (&A)[A_LD], float4 (&Bq)[B_LD], int k0) {
            if (k0 + BKS <= kend && (is1x1 || tap_uni)) {
                unsigned so_a = (unsigned)k0 * 4u, kp = 0;
                if (!is1x1) {
                    const unsigned e = __builtin_amdgcn_readfirstlane(ktab[k0 >> 2]);
                    so_a = e >> 6;
                    kp = e & 31u;
                }
#pragma unroll
                for (int q = 0; q < A_LD; ++q) {
                    const unsigned tinv = (((a_mlo[q] >> kp) & 1u) - 1u) & INV;     // 1x1: bit 0 is set for valid rows
                    A[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrb, a_vk[q] | tinv, so_a, 0));
                }
#pragma unroll
                for (int q = 0; q < B_LD; ++q)
                    Bq[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, b_vk[q], (unsigned)k0 * 4u, 0));
                return;
            }
        
        const int k = k0 + kg;
        const unsigned kinv = ~(unsigned)((k - kend) >> 31) & INV;      // INV when k >= kend
        const unsigned k4 = (unsigned)k * 4u;
        // 1x1 filters have no tap table: their "entry" is (k*4) << 6 | tap 0.  One load sequence serves
        // both cases (selected by mask arithmetic, not a branch: two copies of the loads under a branch
        // write the same registers and cost a vmcnt(0) each)
        const unsigned e = ktab[(unsigned)min(k >> 2, p.ktab_entries - 1) & ~m1x1];
        const unsigned ee = (e & ~m1x1) | ((k4 << 6) & m1x1);
        const unsigned d4 = ee >> 6, kp = ee & 63u;
#pragma unroll
        for (int q = 0; q < A_LD; ++q) {
            const unsigned long long m64 = ((unsigned long long)a_mhi[q] << 32) | a_mlo[q];
            const unsigned tinv = (((unsigned)(m64 >> kp) & 1u) - 1u) & INV;  // INV when the tap is padding
            A[q] = __builtin_bit_cast(float4,
                                      __builtin_amdgcn_raw_buffer_load_b128(xr, (a_off4[q] + d4) | tinv | kinv, 0, 0));
        }
#pragma unroll
        for (int q = 0; q < B_LD; ++q)
            Bq[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, (b_off4[q] + k4) | kinv, 0, 0));
    };
    // register prefetch: the loads of stage k+1 are in flight while stage k computes.  (A second
    // register set, two stages in flight, measured slower in the 4-wave kernel: the extra VGPRs cost more
    // occupancy than the deeper prefetch hides.)
    float4 ra[A_LD], rb[B_LD];
    auto gload = [&](int k0) { stage_load(ra, rb, k0); };
    auto sstore = [&](int S) {
#pragma unroll
        for (int q = 0; q < A_LD; ++q) {
            const int row = (tid >> 3) + q * (THREADS / 8);
            if (row < BM) *(float4*)&As[S][row * BKS + ((kc ^ ((row >> 1) & 7)) << 2)] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < B_LD; ++q) {
            const int row = (tid >> 3) + q * (THREADS / 8);
            if (row < BN) *(float4*)&Bs[S][row * BKS + ((kc ^ ((row >> 1) & 7)) << 2)] = rb[q];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    // fragment reads of one 16-deep half (h = 0/1) of a stage, and the 4*TM*TN MFMAs that consume them
    auto rd = [&](float4 (&av)[TM], float4 (&bv)[TN], int buf, int h) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + fr;
                av[i] = *(const float4*)&As[buf][row * BKS + (((h * 4 + fg) ^ ((row >> 1) & 7)) << 2)];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wn * TN + j) * 16 + fr;
                bv[j] = *(const float4*)&Bs[buf][row * BKS + (((h * 4 + fg) ^ ((row >> 1) & 7)) << 2)];
            }
        
    };
    auto mm = [&](const float4 (&av)[TM], const float4 (&bv)[TN]) {
        // k component outermost: consecutive MFMAs hit DIFFERENT accumulators (the 16x16x4 f32
        // MFMA issues every 32 cycles but a dependent one waits 40)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                    const float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
                }
    };
    auto compute = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 av[TM], bv[TN];
            rd(av, bv, buf, h);
            mm(av, bv);
        }
    };
    // residual tile: issued before the K loop so that its latency hides behind the MFMAs (small
    // tiles only: C_LD float4 registers per thread)
    constexpr int C_LD = (BM * (BN / 4) + NT - 1) / NT;
    constexpr bool PREFETCH_RES = C_LD <= 8;
    const bool vec_epi = p.splitk <= 1 && (p.N & 3) == 0;
    float4 rres[PREFETCH_RES ? C_LD : 1];
    if (PREFETCH_RES && vec_epi && (p.flags & I2V_EPI_RESIDUAL) && p.ostride == 1) {
#pragma unroll
        for (int it = 0; it < C_LD; ++it) {
            const int e = gtid + it * NT;
            const int row = e / (BN / 4), col = (e % (BN / 4)) * 4;
            const int m = m0 + row, n = n0 + col;
            // non-temporal: in a bottleneck this is the last use of the block input
            rres[it] = (e < BM * (BN / 4) && m < p.M && n < p.N)
                           ? __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4*)(p.res + (long long)m * p.N + n)))
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    unsigned long long t1c = 0;
    if (p.clk) t1c = __builtin_amdgcn_s_memtime();
    gload(kbeg);
    sstore(0);
    // one fragment register set: 118 VGPRs -> four waves per SIMD (a software-pipelined two-set form needed 138 -> three, and
    // measured no faster with four co-resident waves: DESIGN_HISTORY.md)
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BKS) {
        const bool more = k0 + BKS < kend;
        if (more) gload(k0 + BKS);
        compute(buf);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (p.clk && gtid == 0) {     // diagnostic build path: stamps go to their own buffer, never to an output
        const unsigned long long t2c = __builtin_amdgcn_s_memtime();
        unsigned long long* o = p.clk + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        o[0] = t2c - t1c;                                  // K loop (incl. its prologue stage)
        o[1] = __builtin_amdgcn_s_memrealtime() - t0r;     // 100 MHz ticks, whole kernel so far
        o[2] = t1c - t0c;                                  // setup: tap table, row masks, residual prefetch
        o[3] = t2c - t0c;
    }
    // epilogue.  C/D map of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + r.  The tile goes
    // through LDS so that global stores (and the residual loads) are whole 16-B-per-lane rows
    // instead of 64-B fragments of a line.  (Every wave's last fragment reads completed before the
    // barrier of the last loop iteration, so the stage buffers are free.)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                smem[((wm * TM + i) * 16 + 4 * fg + r) * CROW + (wn * TN + j) * 16 + fr] = acc[i][j][r];

    __syncthreads();
    const bool split = p.splitk > 1;
    auto out_index = [&](int m) -> long long {
        if (p.ostride == 1) return (long long)m * p.N;
        const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
        return (((long long)b * p.Hy + oy * p.ostride) * p.Wy + ox * p.ostride) * p.N;
    };
    if (split && p.ws) {
        // Split-K finish inside the kernel: every split stores its partial tile (plain 16-B stores, tile-
        // contiguous), the last one to arrive at the tile's counter adds the partials in split order and
        // runs the fused epilogue.  No clear of y, no atomics on y (fp32 atomics move ~1.3 TB/s chip-wide),
        // no separate epilogue pass, and the sum no longer depends on arrival order.
        //
        // The 8 XCDs have private, mutually non-coherent L2s: a release/acquire fence pair at agent scope
        // writes back and invalidates a whole L2 per fence (measured: the layer3 3x3 went 65 -> 153 us with
        // __threadfence()).  Instead the partials are stored and re-read at agent scope (sc1, the cache policy of
        // a relaxed agent-scope atomic: coherent across the XCDs without fences) -- so only these 20-64 KB tiles pay
        // for coherence; s_waitcnt vmcnt(0) orders the stores before the (device-scope) arrival count.
        constexpr int SC01 = 16;          // buffer aux bit 4 = sc1: agent scope (sc0 sc1 = system scope is not needed)
        const size_t split_stride = (size_t)gridDim.x * (BM * BN);      // floats between splits of a tile
        const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.ws + (size_t)tile * (BM * BN)), 0, 0x7FFFFFF0, 0x00020000);
        const unsigned mine = (unsigned)(blockIdx.y * split_stride * sizeof(float));
        for (int e = gtid; e < BM * (BN / 4); e += NT) {
            const int row = e / (BN / 4), col = (e % (BN / 4)) * 4;
            const float4 v = *(const float4*)&smem[row * CROW + col];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wsr,
                                                   mine + (unsigned)(row * BN + col) * 4u, 0, SC01);
        }
        __builtin_amdgcn_s_waitcnt(0);    // vmcnt(0): my stores have been acknowledged by memory
        __syncthreads();
        __shared__ int last_flag;
        int* flag = &last_flag;
        if (gtid == 0) {
            const int arrived = __hip_atomic_fetch_add(p.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = arrived == (int)gridDim.y - 1;
            if (last) __hip_atomic_store(p.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch
            *flag = last;
        }
        __syncthreads();
        if (!*flag) {
            if (p.clk && gtid == 0)
                p.clk[8 * (blockIdx.y * gridDim.x + blockIdx.x) + 4] = __builtin_amdgcn_s_memtime() - t0c;
            return;
        }
        // The finisher's reads come from beyond the L2 (~2 us a round trip): every partial of up to FIN_CH
        // elements per thread, and their residuals, are put in flight before the first sum.  Its own
        // partial is still in LDS.  Summation is in split order whichever workgroup arrives last.
        const int nsplit = gridDim.y, my = blockIdx.y;
        constexpr int FIN_CH = C_LD < 6 ? C_LD : 6;
        const bool vec = (p.N & 3) == 0;
        for (int it0 = 0; it0 < C_LD; it0 += FIN_CH) {
            float4 u[FIN_CH][kSplitInKernelMax], rr[FIN_CH], acc4[FIN_CH] = {};
            // rounds of kSplitInKernelMax splits, summed in split order: ((((p0 + p1) + p2) + p3) + p4) + ... -- one round for
            // the backbone's splits (<= 4), more for the long skinny GEMMs of the relation head (round 5: they took the
            // atomic finish before, whose sum depends on arrival order)
            for (int s0 = 0; s0 < nsplit; s0 += kSplitInKernelMax) {
#pragma unroll
                for (int c = 0; c < FIN_CH; ++c) {
                    const int e = gtid + (it0 + c) * NT;
                    const int row = e / (BN / 4), col = (e % (BN / 4)) * 4;
                    const int m = m0 + row, n = n0 + col;
                    const bool ok = e < BM * (BN / 4) && m < p.M && n < p.N;
                    const unsigned off = (unsigned)(row * BN + col) * 4u;
#pragma unroll
                    for (int sp = 0; sp < kSplitInKernelMax; ++sp)
                        u[c][sp] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                            wsr, (ok && s0 + sp < nsplit && s0 + sp != my) ? off + (unsigned)((s0 + sp) * split_stride * sizeof(float)) : 0xFFFFFFF0u,
                            0, SC01));
                    if (s0 == 0) {
                        rr[c] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok && vec && (p.flags & I2V_EPI_RESIDUAL)) rr[c] = __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4*)(p.res + (long long)m * p.N + n)));
                    }
                }
#pragma unroll
                for (int c = 0; c < FIN_CH; ++c) {
                    const int e = gtid + (it0 + c) * NT;
                    const int row = e / (BN / 4), col = (e % (BN / 4)) * 4;
                    const float4 mine4 = (e < BM * (BN / 4)) ? *(const float4*)&smem[row * CROW + col] : make_float4(0.f, 0.f, 0.f, 0.f);
                    float4 v = acc4[c];
#pragma unroll
                    for (int sp = 0; sp < kSplitInKernelMax; ++sp) {   // slots >= nsplit were read out of range: zeros
                        const float4 t = s0 + sp == my ? mine4 : u[c][sp];
                        if (s0 == 0 && sp == 0) v = t;
                        else { v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                    }
                    acc4[c] = v;
                }
            }
#pragma unroll
            for (int c = 0; c < FIN_CH; ++c) {
                const int e = gtid + (it0 + c) * NT;
                const int row = e / (BN / 4), col = (e % (BN / 4)) * 4;
                const int m = m0 + row, n = n0 + col;
                if (e >= BM * (BN / 4) || m >= p.M || n >= p.N) continue;
                const float4 v = acc4[c];
                const long long o = (long long)m * p.N + n;      // split-K only runs with ostride == 1
                float vv[4] = {v.x, v.y, v.z, v.w};
                if (vec) {
                    if (p.flags & I2V_EPI_SCALE) {
                        const float4 sc = *(const float4*)(p.scale + n);
                        vv[0] *= sc.x; vv[1] *= sc.y; vv[2] *= sc.z; vv[3] *= sc.w;
                    }
                    if ((p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && p.shift) {
                        const float4 sh = *(const float4*)(p.shift + n);
                        vv[0] += sh.x; vv[1] += sh.y; vv[2] += sh.z; vv[3] += sh.w;
                    }
                    if (p.flags & I2V_EPI_RESIDUAL) {
                        vv[0] += rr[c].x; vv[1] += rr[c].y; vv[2] += rr[c].z; vv[3] += rr[c].w;
                    }
                    if (p.flags & I2V_EPI_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) vv[q] = fmaxf(vv[q], 0.f);
                    }
                    if (p.flags & I2V_EPI_MASK) {
                        const float4 mk = *(const float4*)(p.mask + o);
                        vv[0] = mk.x > 0.f ? vv[0] : 0.f; vv[1] = mk.y > 0.f ? vv[1] : 0.f;
                        vv[2] = mk.z > 0.f ? vv[2] : 0.f; vv[3] = mk.w > 0.f ? vv[3] : 0.f;
                    }
                    *(float4*)(p.y + o) = make_float4(vv[0], vv[1], vv[2], vv[3]);
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (n + q >= p.N) break;
                        float t = vv[q];
                        if (p.flags & I2V_EPI_SCALE) t *= p.scale[n + q];
                        if ((p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && p.shift) t += p.shift[n + q];
                        if (p.flags & I2V_EPI_RESIDUAL) t += p.res[o + q];
                        if (p.flags & I2V_EPI_RELU) t = fmaxf(t, 0.f);
                        if ((p.flags & I2V_EPI_MASK) && !(p.mask[o + q] > 0.f)) t = 0.f;
                        p.y[o + q] = t;
                    }
                }
            }
        }
    } else if (split) {               // fallback: fp32 atomics into a zeroed y (64 lanes = 256 contiguous bytes)
        for (int e = gtid; e < BM * BN; e += NT) {
            const int row = e / BN, col = e % BN;
            const int m = m0 + row, n = n0 + col;
            if (m < p.M && n < p.N) atomicAdd(p.y + out_index(m) + n, smem[row * CROW + col]);
        }
    } else if ((p.N & 3) == 0) {
#pragma unroll
        for (int it = 0; it < C_LD; ++it) {
            const int e = gtid + it * NT;
            if (e >= BM * (BN / 4)) break;
            const int row = e / (BN / 4), col = (e % (BN / 4)) * 4;
            const int m = m0 + row, n = n0 + col;
            if (m >= p.M || n >= p.N) continue;           // N % 4 == 0: a float4 never straddles N
            float4 v = *(const float4*)&smem[row * CROW + col];
            if (p.flags & I2V_EPI_SCALE) {
                const float4 sc = *(const float4*)(p.scale + n);
                v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
            }
            if ((p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && p.shift) {
                const float4 sh = *(const float4*)(p.shift + n);
                v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w;
            }
            const long long o = out_index(m) + n;
            if (p.flags & I2V_EPI_RESIDUAL) {
                const float4 rr = (PREFETCH_RES && p.ostride == 1) ? rres[PREFETCH_RES ? it : 0]
                                                                   : *(const float4*)(p.res + o);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            if (p.flags & I2V_EPI_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if (p.flags & I2V_EPI_MASK) {
                const float4 mk = *(const float4*)(p.mask + o);
                v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
            }
            *(float4*)(p.y + o) = v;
        }
    } else {
        for (int e = gtid; e < BM * BN; e += NT) {
            const int row = e / BN, col = e % BN;
            const int m = m0 + row, n = n0 + col;
            if (m >= p.M || n >= p.N) continue;
            float v = smem[row * CROW + col];
            if (p.flags & I2V_EPI_SCALE) v *= p.scale[n];
            if ((p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && p.shift) v += p.shift[n];
            const long long o = out_index(m) + n;
            if (p.flags & I2V_EPI_RESIDUAL) v += p.res[o];
            if (p.flags & I2V_EPI_RELU) v = fmaxf(v, 0.f);
            if ((p.flags & I2V_EPI_MASK) && !(p.mask[o] > 0.f)) v = 0.f;
            p.y[o] = v;
        }
    }
    if (p.clk && gtid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* o = p.clk + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
        o[4] = __builtin_amdgcn_s_memtime() - t0c;           // whole kernel, this workgroup
        o[5] = 1;                                            // this workgroup ran the epilogue (split-K: the last arrival)
    }
}

// ---------------------------------------------------------------- pointwise / plain-GEMM specialisation
// y[m][n] = epi(sum_k A[m][k] * Wt[n][k]) where row m of A IS row m of x (1x1 filter, stride 1, same grid; linear layers;
// the element-wise planes of a Winograd convolution).  The layer3 bottleneck GEMMs are short (K = 256: 8 stages of MFMA
// work, ~12k cycles), so what a workgroup does OUTSIDE the K loop decides: this kernel has no tap table, no row masks, no
// integer division per row, prefetches the epilogue's operands (residual tile, BN scale / shift) before the K loop, and
// its epilogue never touches LDS: the MFMA operands are swapped (weights as the row operand), so a lane's four
// accumulator registers are four CONSECUTIVE output channels of one pixel -- scale/shift/residual/ReLU in registers, one
// 16-B store per fragment, no LDS round trip, no barrier.  Staging, swizzled LDS image and the K loop are conv_igemm_f32's.
// Split-K (<= kSplitInKernelMax): partial tiles go to the caller's workspace in REGISTER order (fragment, wave, lane), the
// last workgroup to arrive sums them in split order -- same protocol as conv_igemm_f32, whole-wave 1-KB rows.
// KG > 1 (round 4): INTRA-WORKGROUP K split.  A GEMM too short in M x N to fill the chip (layer3 conv1 of a frame pair: 240
// tiles of 80x64 for 1024 slots, K = 1024) used to run as 3 K-splits of 4 waves whose partial tiles crossed the fabric
// (sc1 write-through + re-read by the last arriver: 14.7 + 9.8 MB per launch beside the layer's own 25.6 MB).  Here ONE
// workgroup of KG x 4 waves owns the tile: wave group g stages and multiplies k in [g K/KG, (g+1) K/KG) in an LDS region of
// its own, the groups' partial tiles meet in LDS (register order, conflict-free 16-byte rows) and group 0 adds them IN GROUP
// ORDER ((p0 + p1) + p2) + p3 -- deterministic -- and runs the one fused epilogue.  Same waves per SIMD as four co-resident
// split workgroups, no partial ever leaves the CU, no arrival counter.  K must be a multiple of KG x 32 (the host checks).
// STG (round 6): how a stage reaches LDS.  0: global -> VGPR -> ds_write_b128 (rounds 2-5).  1 / 2: LDS-DMA
// (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, no LDS store-path cycles; the DMA lands lane-linear
// (wave-uniform base + lane x 16 bytes), so the image's column swizzle rides on the SOURCE address -- each 128-byte (64-byte)
// row piece is still one contiguous run of the operand row, its 16-byte columns permuted among the lanes that fetch it.
// Two buffers, one raw s_barrier per stage: [vmcnt(0): my pieces of stage i have landed | barrier: everyone's have, and everyone
// is done with the other buffer | request stage i + 1 into it | multiply stage i].  1: 32-k stages in the rounds-2-5 image
// (128-byte rows, column ^ (row >> 1) & 7).  2: 16-k stages, 64-byte rows, column ^ (-(row >> 2)) & 3 (the four 16-lane groups of
// a ds_read_b128 still touch 16 distinct 16-byte slots): half the LDS per workgroup, twice the barriers.  Same fragment
// ownership, same k order per accumulator: bit-equal to STG 0 (tools/micro/gemm_lab.hip measured the forms side by side).
// The requests are inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and drains vmcnt(0) in front of every
// ds_read.  Its own loads (epilogue operands) may sit in the same queue: returns are in order, so an extra load can only make a
// counted wait wait longer, never shorter.
__device__ inline void lds_dma16(unsigned dst, unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(r), "s"(soff) : "memory");
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool CLK = false, bool MASK = false, int KG = 1, int STG = 0>
__global__ void __launch_bounds__(THREADS * KG)
conv_gemm_f32(const ConvP p_in) {
    ConvP p = p_in;
    // CLK: diagnostic instantiation (i2v_conv_debug_clock): per workgroup {start, end in 100 MHz ticks, prologue / K-loop /
    // epilogue shader cycles, HW_ID, XCC_ID} into a buffer of their own; the product instantiation has no stamp
    unsigned long long c_rt0 = 0, c_t0 = 0, c_t1 = 0, c_t2 = 0;
    if constexpr (CLK) { c_rt0 = __builtin_amdgcn_s_memrealtime(); c_t0 = __builtin_amdgcn_s_memtime(); }
    if (p.nbatch > 1) {
        p.x += (long long)blockIdx.z * p.bsx;
        p.w += (long long)blockIdx.z * p.bsw;
        p.y += (long long)blockIdx.z * p.bsy;
    }
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    constexpr int A_LD = (BM * 8 + THREADS - 1) / THREADS, B_LD = (BN * 8 + THREADS - 1) / THREADS;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    static_assert(KG == 1 || !CLK, "the K-group form has no diagnostic instantiation");
    static_assert(STG == 0 || (KG == 1 && !CLK), "the LDS-DMA form is the plain 4-wave kernel");
    constexpr int BKL = STG == 2 ? 16 : BKS;                   // k per LDS stage
    // KG > 1: the four waves of a K group synchronise among THEMSELVES between stages (an arrival counter in LDS: release,
    // add, poll, acquire) -- an s_barrier would march all 16 waves in lock step, every SIMD's four waves staging together and
    // multiplying together; free-running groups drift apart like co-resident workgroups do, one group's staging under another's
    // MFMAs.  The counters only grow (4 per barrier), so a fast wave's next arrival cannot release a slow wave early.
    __shared__ int kcnt[KG > 1 ? KG : 1];
    int kphase = 0;
    auto stage_barrier = [&]() {
        if constexpr (KG == 1) {
            __syncthreads();
        } else {
            kphase += 4;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // my LDS stores have landed
            if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&kcnt[threadIdx.x >> 8], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (__hip_atomic_load(&kcnt[threadIdx.x >> 8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < kphase)
                __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
    };
    if constexpr (KG > 1) {
        if (threadIdx.x < KG) kcnt[threadIdx.x] = 0;
    }
    constexpr int GROUP_FLOATS = 2 * (BM + BN) * BKL;          // one K group's two stage buffers
    extern __shared__ __attribute__((aligned(16))) float smem_all[];
    const int gid = KG > 1 ? (int)(threadIdx.x >> 8) : 0;      // K group of this wave (4 waves per group)
    float* smem = smem_all + gid * GROUP_FLOATS;
    float (*As)[BM * BKL] = reinterpret_cast<float (*)[BM * BKL]>(smem);
    float (*Bs)[BN * BKL] = reinterpret_cast<float (*)[BN * BKL]>(smem + 2 * BM * BKL);

    const int tid = KG > 1 ? (int)(threadIdx.x & (THREADS - 1)) : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN;
    int tile;
    {       // XCD-aware tile order (conv_igemm_f32)
        const int nt = gridDim.x, q = nt >> 3, r = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
        tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kbeg = KG > 1 ? gid * (p.K / KG) : blockIdx.y * p.k_per_split;
    const int kend = KG > 1 ? kbeg + p.K / KG : min(p.K, kbeg + p.k_per_split);
    const int kc = tid & 7, kg = kc * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    constexpr unsigned INV = 0x80000000u;
    const unsigned g0_inv = gid == 0 ? 0u : INV;               // only K group 0 runs the epilogue: the others fetch no operand of it
    unsigned a_vk[A_LD], b_vk[B_LD];
#pragma unroll
    for (int q = 0; q < A_LD; ++q) {
        const int row = (tid >> 3) + q * (THREADS / 8);
        const int m = m0 + row;
        a_vk[q] = (row < BM && m < p.M) ? ((unsigned)(m * p.K) + (unsigned)kg) * 4u : INV;
    }
#pragma unroll
    for (int q = 0; q < B_LD; ++q) {
        const int row = (tid >> 3) + q * (THREADS / 8);
        const int n = n0 + row;
        b_vk[q] = (row < BN && n < p.N) ? ((unsigned)(n * p.K) + (unsigned)kg) * 4u : INV;
    }
    float4 ra[A_LD], rb[B_LD];
    // the k offset of a stage is uniform: it rides in the scalar offset of the buffer load; a lane beyond kend (the last,
    // partial stage of a K that is not a multiple of 32) gets the 2 GiB bit OR-ed in and reads zeros
    auto gload = [&](int k0) {
        const unsigned so = (unsigned)k0 * 4u;
        const unsigned kinv = ~(unsigned)((k0 + kg - kend) >> 31) & INV;
#pragma unroll
        for (int q = 0; q < A_LD; ++q)
            ra[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, a_vk[q] | kinv, so, 0));
#pragma unroll
        for (int q = 0; q < B_LD; ++q)
            rb[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, b_vk[q] | kinv, so, 0));
    };
    auto sstore = [&](int S) {
#pragma unroll
        for (int q = 0; q < A_LD; ++q) {
            const int row = (tid >> 3) + q * (THREADS / 8);
            if (row < BM) *(float4*)&As[S][row * BKS + ((kc ^ ((row >> 1) & 7)) << 2)] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < B_LD; ++q) {
            const int row = (tid >> 3) + q * (THREADS / 8);
            if (row < BN) *(float4*)&Bs[S][row * BKS + ((kc ^ ((row >> 1) & 7)) << 2)] = rb[q];
        }
    };
    // ---- LDS-DMA staging (STG > 0).  A stage = NP 1-KB pieces (PR rows of the A tile, then of the B tile); wave w requests
    // pieces w, w + 4, ...  A lane's logical 16-byte column is the same for all pieces of its wave: the swizzle term of a row
    // repeats with the piece stride (8 rows x an even piece count for the 128-byte rows; 16 rows for the 64-byte rows).
    constexpr int PR = 256 / BKL, CPR = BKL / 4;
    constexpr int NPA = BM / PR, NPB = BN / PR, NP = NPA + NPB, PMAX = (NP + 3) / 4, REM = NP % 4;
    static_assert(STG == 0 || (BM % PR == 0 && BN % PR == 0 && (BKL == 16 || NPA % 2 == 0)), "whole pieces, one swizzle term per wave");
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lchunk = BKL == 32 ? ((lane & 7) ^ ((wave_s * 4 + (lane >> 4)) & 7)) : ((lane & 3) ^ ((-(lane >> 4)) & 3));
    unsigned d_vk[STG > 0 ? PMAX : 1];
    __amdgpu_buffer_rsrc_t d_rs[STG > 0 ? PMAX : 1];
    if constexpr (STG > 0) {
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int pc = q * 4 + wave_s;
            const bool isA = pc < NPA;                                      // wave-uniform
            const int row = (isA ? pc : pc - NPA) * PR + lane / CPR;
            const int g = (isA ? m0 : n0) + row;
            d_vk[q] = (pc < NP && g < (isA ? p.M : p.N)) ? ((unsigned)(g * p.K) + (unsigned)(lchunk * 4)) * 4u : INV;
            d_rs[q] = isA ? xr : wr;
        }
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    auto dma_issue = [&](int k0, int S) {
        const unsigned so = (unsigned)k0 * 4u;
        const unsigned kinv = ~(unsigned)((k0 + lchunk * 4 - kend) >> 31) & INV;      // a partial last stage: lanes beyond kend read zeros
        const unsigned a_dst = lds_base + (unsigned)(S * BM * BKL * 4), b_dst = lds_base + (unsigned)((2 * BM + S * BN) * BKL * 4);
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int pc = q * 4 + wave_s;
            if (REM == 0 || q < PMAX - 1 || wave_s < REM)
                lds_dma16(pc < NPA ? a_dst + (unsigned)(pc * 1024) : b_dst + (unsigned)((pc - NPA) * 1024), d_vk[q] | kinv, d_rs[q], so);
        }
    };
    auto dma_wait_barrier = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of the stage have landed ...
        __builtin_amdgcn_s_barrier();                          // ... everyone's have; and everyone is done with the other buffer
    };
    if constexpr (STG > 0) dma_issue(kbeg, 0);
    else gload(kbeg);                     // first: everything below hides behind this round trip

    // Operands of the epilogue.  A lane owns channels n .. n+3 of pixel m for every fragment (i, j):
    //   m = m0 + (wm*TM + i)*16 + (lane & 15),   n = n0 + (wn*TN + j)*16 + 4*(lane >> 4)
    // They are fetched by buffer loads that are ALWAYS issued (an absent operand or an out-of-range lane gets the 2 GiB
    // offset and reads zeros without traffic): a load under a condition makes the number of loads in flight unknowable
    // to the compiler, which then drains everything (s_waitcnt vmcnt(0)) in front of the first LDS store -- with the
    // 19.6 MB residual of a layer3 conv3 in flight that cost 4.5k of 8.1k prologue cycles (tools/gemm_phase.py).
    // The residual tile is requested one stage before the last, behind that stage's operand loads (returns are in
    // order): its latency lies under the last stage's MFMAs, and no operand load ever waits for it.
    const bool split = p.splitk > 1;
    constexpr bool PREFETCH_RES = TM * TN <= 8;
    const unsigned n_bytes = (unsigned)p.N * 4u;
    const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.scale ? p.scale : p.shift ? p.shift : (const float*)p.x), 0, n_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.shift ? p.shift : (const float*)p.x), 0, n_bytes, 0x00020000);
    const unsigned long long y_bytes = (unsigned long long)p.M * p.N * 4ull;
    const __amdgpu_buffer_rsrc_t resr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.res ? p.res : (const float*)p.x), 0, (unsigned)(y_bytes < 0x7FFFFFF0ull ? y_bytes : 0x7FFFFFF0ull), 0x00020000);
    const unsigned sc_inv = ((p.flags & I2V_EPI_SCALE) ? 0u : INV) | g0_inv;
    const unsigned sh_inv = (((p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && p.shift) ? 0u : INV) | g0_inv;     // scale without shift: the data-gradient epilogue
    const unsigned res_inv = (((p.flags & I2V_EPI_RESIDUAL) && !split) ? 0u : INV) | g0_inv;
    const __amdgpu_buffer_rsrc_t mskr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.mask ? p.mask : (const float*)p.x), 0, (unsigned)(y_bytes < 0x7FFFFFF0ull ? y_bytes : 0x7FFFFFF0ull), 0x00020000);
    float4 sc[TN], sh[TN], rres[PREFETCH_RES ? TM : 1][PREFETCH_RES ? TN : 1];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 16 + 4 * fg;
        const unsigned off = n < p.N ? (unsigned)n * 4u : INV;
        sc[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(scr, off | sc_inv, 0, 0));
        sh[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(shr, off | sh_inv, 0, 0));
    }
    auto issue_res = [&]() {
        if constexpr (PREFETCH_RES) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int m = m0 + (wm * TM + i) * 16 + fr, n = n0 + (wn * TN + j) * 16 + 4 * fg;
                    const unsigned off = (m < p.M && n < p.N) ? (unsigned)(m * p.N + n) * 4u : INV;
                    rres[i][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(resr, off | res_inv, 0, 2));   // nt: last use of the block input
                }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        auto swzl = [](int row) { return BKL == 32 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); };
#pragma unroll
        for (int h = 0; h < BKL / 16; ++h) {
            float4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + fr;
                av[i] = *(const float4*)&As[buf][row * BKL + (((h * 4 + fg) ^ swzl(row)) << 2)];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wn * TN + j) * 16 + fr;
                bv[j] = *(const float4*)&Bs[buf][row * BKL + (((h * 4 + fg) ^ swzl(row)) << 2)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                        const float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                        // weights are the ROW operand: D[row = channel][col = pixel], a lane holds 4 consecutive channels
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i][j], 0, 0, 0);
                    }
        }
    };
    int buf = 0, k0 = kbeg;
    if constexpr (STG > 0) {
        for (; k0 + 2 * BKL < kend; k0 += BKL) {      // every stage but the last two
            dma_wait_barrier();
            dma_issue(k0 + BKL, buf ^ 1);
            compute(buf);
            buf ^= 1;
        }
        if (k0 + BKL < kend) {                        // two stages left: the last one's operands, then the residual tile
            dma_wait_barrier();
            dma_issue(k0 + BKL, buf ^ 1);
            issue_res();
            compute(buf);
            buf ^= 1;
        } else {
            issue_res();
        }
        dma_wait_barrier();
        compute(buf);
    } else {
    if constexpr (KG > 1) __syncthreads();           // the arrival counters are zero (the first stage's loads are in flight)
    sstore(0);
    stage_barrier();
    if constexpr (CLK) c_t1 = __builtin_amdgcn_s_memtime();
    for (; k0 + 2 * BKS < kend; k0 += BKS) {          // every stage but the last two
        gload(k0 + BKS);
        // the next stage's loads are REQUESTED here: left alone the scheduler sinks them below 35 of the stage's 40 MFMAs (shorter
        // live ranges), and a wave then waits out the L2 round trip between its last MFMA and its LDS stores.  With four waves
        // per SIMD the others cover that (the step did not move: 4.45-4.47 against 4.49-4.53 ms, box noise); pinned, a wave
        // covers it alone, which is what the source meant
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        sstore(buf ^ 1);
        stage_barrier();
        buf ^= 1;
    }
    if (k0 + BKS < kend) {                            // two stages left: operands of the last one, then the residual tile
        gload(k0 + BKS);
        issue_res();
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        sstore(buf ^ 1);
        stage_barrier();
        buf ^= 1;
    } else {
        issue_res();
    }
    compute(buf);                                     // last stage (no barrier: nothing is staged after it)
    }
    if constexpr (CLK) c_t2 = __builtin_amdgcn_s_memtime();
    if constexpr (KG > 1) {
        // the K groups' partial tiles meet in LDS: every group but the first leaves its accumulators in its own (now idle)
        // stage region, in register order -- (fragment, wave, lane) x 16 bytes: whole-wave 1-KB rows both ways -- and group 0
        // adds them in group order
        static_assert(TM * TN * THREADS * 4 <= GROUP_FLOATS, "a partial tile fits a group's stage buffers");
        __syncthreads();                              // every wave has read its last stage
        if (gid > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *(f32x4*)&smem[(((i * TN + j) * 4 + wave) * 64 + lane) * 4] = acc[i][j];
        }
        __syncthreads();
        if (gid > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 t = *(const f32x4*)&smem_all[g * GROUP_FLOATS + (((i * TN + j) * 4 + wave) * 64 + lane) * 4];
                    acc[i][j] = (f32x4){acc[i][j][0] + t[0], acc[i][j][1] + t[1], acc[i][j][2] + t[2], acc[i][j][3] + t[3]};
                }
    }
    auto stamp = [&]() {
        if constexpr (CLK) {
            __builtin_amdgcn_s_waitcnt(0);
            if (tid == 0) {
                unsigned long long* o = p.clk + 8 * ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
                o[0] = c_rt0; o[1] = __builtin_amdgcn_s_memrealtime();
                o[2] = c_t1 - c_t0; o[3] = c_t2 - c_t1; o[4] = __builtin_amdgcn_s_memtime() - c_t2;
                o[5] = __builtin_amdgcn_s_getreg(63492); o[6] = __builtin_amdgcn_s_getreg(63508); o[7] = 1;
            }
        }
    };

    auto finish = [&](int i, int j, f32x4 v, float4 rr, float4 mk) {
        const int m = m0 + (wm * TM + i) * 16 + fr, n = n0 + (wn * TN + j) * 16 + 4 * fg;
        if (m >= p.M || n >= p.N) return;
        float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (p.flags & I2V_EPI_SCALE) { o.x *= sc[j].x; o.y *= sc[j].y; o.z *= sc[j].z; o.w *= sc[j].w; }   // an absent operand was read as zeros
        if ((p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && p.shift) { o.x += sh[j].x; o.y += sh[j].y; o.z += sh[j].z; o.w += sh[j].w; }
        if (p.flags & I2V_EPI_RESIDUAL) { o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w; }
        if (p.flags & I2V_EPI_RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if constexpr (MASK) { o.x = mk.x > 0.f ? o.x : 0.f; o.y = mk.y > 0.f ? o.y : 0.f; o.z = mk.z > 0.f ? o.z : 0.f; o.w = mk.w > 0.f ? o.w : 0.f; }
        *(float4*)(p.y + (long long)m * p.N + n) = o;
    };
    auto res_at = [&](int i, int j) -> float4 {
        const int m = m0 + (wm * TM + i) * 16 + fr, n = n0 + (wn * TN + j) * 16 + 4 * fg;
        const unsigned off = (m < p.M && n < p.N && (p.flags & I2V_EPI_RESIDUAL)) ? (unsigned)(m * p.N + n) * 4u : INV;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(resr, off, 0, 2));
    };
    auto mask_at = [&](int i, int j) -> float4 {
        const int m = m0 + (wm * TM + i) * 16 + fr, n = n0 + (wn * TN + j) * 16 + 4 * fg;
        const unsigned off = (m < p.M && n < p.N && (p.flags & I2V_EPI_MASK)) ? (unsigned)(m * p.N + n) * 4u : INV;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(mskr, off, 0, 0));
    };
    if (!split) {
        // the mask tile (data gradients only) is fetched here, all fragments at once, into registers the K loop no longer
        // needs: prefetched beside the residual it would cost the forward layers 4 x TM x TN registers they never use
        float4 mk[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mk[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (MASK) {                   // the MASK instantiation serves data gradients only (launch_tile)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) mk[i][j] = mask_at(i, j);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float4 rr;
                if constexpr (PREFETCH_RES) rr = rres[PREFETCH_RES ? i : 0][PREFETCH_RES ? j : 0];
                else rr = res_at(i, j);
                finish(i, j, acc[i][j], rr, mk[i][j]);
            }
        stamp();
        return;
    }
    // ---- split-K: partials in register order through the caller's workspace (sc1 stores / loads: coherent across the
    // XCDs without fences; conv_igemm_f32 has the protocol's rationale)
    constexpr int SC01 = 16;
    const size_t split_stride = (size_t)gridDim.x * (BM * BN);
    const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.ws + (size_t)tile * (BM * BN)), 0, 0x7FFFFFF0, 0x00020000);
    const unsigned mine = (unsigned)(blockIdx.y * split_stride * sizeof(float));
    auto slot = [&](int i, int j) { return (unsigned)((((i * TN + j) * 4 + wave) * 64 + lane) * 16); };
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), wsr, mine + slot(i, j), 0, SC01);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    __shared__ int last_flag;
    if (tid == 0) {
        const int arrived = __hip_atomic_fetch_add(p.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = arrived == (int)gridDim.y - 1;
        if (last) __hip_atomic_store(p.cnt + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) { stamp(); return; }
    const int nsplit = gridDim.y, my = blockIdx.y;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        float4 u[TN][kSplitInKernelMax], rr[TN], mk[TN], v4[TN] = {};
        // rounds of kSplitInKernelMax splits, summed in split order (conv_igemm_f32 has the rationale): one round for the
        // backbone's splits, more for the relation head's skinny GEMMs
        for (int s0 = 0; s0 < nsplit; s0 += kSplitInKernelMax) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int sp = 0; sp < kSplitInKernelMax; ++sp)
                    u[j][sp] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                        wsr, (s0 + sp < nsplit && s0 + sp != my) ? slot(i, j) + (unsigned)((s0 + sp) * split_stride * sizeof(float)) : 0xFFFFFFF0u, 0, SC01));
                if (s0 == 0) {
                    rr[j] = res_at(i, j);
                    if constexpr (MASK) mk[j] = mask_at(i, j);
                    else mk[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const float4 m4 = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                float4 v = v4[j];
#pragma unroll
                for (int sp = 0; sp < kSplitInKernelMax; ++sp) {      // slots >= nsplit were read out of range: zeros
                    const float4 t = s0 + sp == my ? m4 : u[j][sp];
                    if (s0 == 0 && sp == 0) v = t;
                    else { v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                }
                v4[j] = v;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) finish(i, j, (f32x4){v4[j].x, v4[j].y, v4[j].z, v4[j].w}, rr[j], mk[j]);
    }
    stamp();
}


// epilogue of the split-K path (partials were accumulated with fp32 atomics)
__global__ void conv_epilogue_kernel(float* __restrict__ y, const float* __restrict__ scale,
                                     const float* __restrict__ shift, const float* __restrict__ res,
                                     const float* __restrict__ mask, long long total4, int N, int flags) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 v = ((float4*)y)[i];
        const int n = (int)((i * 4) % N);
        float4 sc = make_float4(1, 1, 1, 1), sh = make_float4(0, 0, 0, 0);
        if (flags & I2V_EPI_SCALE) sc = *(const float4*)(scale + n);
        if ((flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && shift) sh = *(const float4*)(shift + n);
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        if (flags & I2V_EPI_RESIDUAL) {
            float4 r = ((const float4*)res)[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (flags & I2V_EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (flags & I2V_EPI_MASK) {
            const float4 mk = ((const float4*)mask)[i];
            v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
        ((float4*)y)[i] = v;
    }
}

__global__ void conv_epilogue_scalar_kernel(float* __restrict__ y, const float* __restrict__ scale,
                                            const float* __restrict__ shift, const float* __restrict__ res,
                                            const float* __restrict__ mask, long long total, int N, int flags) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % N);
        float v = y[i];
        if (flags & I2V_EPI_SCALE) v *= scale[n];
        if ((flags & (I2V_EPI_SCALE | I2V_EPI_BIAS)) && shift) v += shift[n];
        if (flags & I2V_EPI_RESIDUAL) v += res[i];
        if (flags & I2V_EPI_RELU) v = fmaxf(v, 0.f);
        if ((flags & I2V_EPI_MASK) && !(mask[i] > 0.f)) v = 0.f;
        y[i] = v;
    }
}

inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

constexpr int NUM_CU = 256;
int g_force_tile = -1;           // i2v_conv_set_tile(): tuning hook (the tile index)
int g_ablate = 0;
unsigned long long* g_clk = nullptr;   // i2v_conv_debug_clock()
// Tuning knobs live in g_i2v_tuning (i2v_set_tuning; the library itself reads no environment variable).
// SPLIT_TARGET (workgroups per CU a split-K launch aims for), measured inside the step: 3 for the skinny FC GEMMs makes the
// step 1 % faster (4.89 vs 4.94 ms) although fc6 forward alone goes from 410 to 556 us and its time becomes unstable; 3
// for everything the same, 4 slower (5.21).  The default (2) is the setting that is best for the kernels on their own.
#define g_split_target g_i2v_tuning[I2V_TUNE_SPLIT_TARGET]
#define g_split_target_skinny g_i2v_tuning[I2V_TUNE_SPLIT_TARGET_SKINNY]
#define g_split_below g_i2v_tuning[I2V_TUNE_SPLIT_BELOW]
#define g_split_atomics g_i2v_tuning[I2V_TUNE_SPLIT_ATOMICS]
#define g_big_fc_tile g_i2v_tuning[I2V_TUNE_BIG_FC_TILE]
#define g_wgrad_v2 g_i2v_tuning[I2V_TUNE_WGRAD_V2]
#define g_wgrad_fused_tile g_i2v_tuning[I2V_TUNE_WGRAD_FUSED_TILE]

// Split-K workspace, provided by the CALLER (i2v_conv_split_workspace_bytes): [kSplitCounters arrival counters | slab of
// partial tiles].  The counters must be zero before the first launch that uses the workspace; every launch leaves them
// zero again (the last workgroup to arrive at a tile resets its counter).  Launches that share a workspace must be
// ordered on the device (same stream, or graph edges): two concurrently running launches need two workspaces.
static int g_ordered_fallbacks = 0;      // i2v_ordered_fallbacks(): reductions asked to be ordered that ran on fp32 atomics
constexpr int kSplitCounters = 1024;
constexpr size_t kSplitCounterBytes = sizeof(int) * kSplitCounters;

template <class K>
void set_max_lds(K kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

constexpr int kKGroups = 4;      // wave groups of the intra-workgroup K split (16 waves = 4 per SIMD, one workgroup per CU)

// The intra-workgroup K split (conv_gemm_f32<.., KG>): one workgroup of KG x 4 waves per tile, KG stage-buffer sets in LDS.
// KG = 4 (round 4): 16 waves, 115-156 KB -- the workgroup owns its CU.  KG = 2 (round 5): 8 waves, 58-78 KB -- two of them, or
// one and the 4-wave workgroups of the step's other branches, share a CU.
template <int WAVES_M, int WAVES_N, int TM, int TN, int KG = kKGroups>
int launch_kgroups(const ConvP& p, hipStream_t st) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    constexpr int need = KG * 2 * (BM + BN) * BKS * 4;
    static_assert(need <= 159 * 1024, "the stage-buffer sets fit the CU's LDS");
    // exactly what the launch asks for: the kernel also has a few bytes of static LDS, and the attribute call fails (leaving
    // the 64 KB default in force) when dynamic + static would exceed the CU's 160 KB
    static const hipError_t once_k = [] {
        hipError_t e = hipFuncSetAttribute((const void*)conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, false, KG>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, need);
        hipError_t e2 = hipFuncSetAttribute((const void*)conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true, KG>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, need);
        return e != hipSuccess ? e : e2;
    }();
    if (once_k != hipSuccess) {
        (void)hipGetLastError();
        i2v_set_error("conv: %d bytes of LDS refused for the K-group kernel: %s", need, hipGetErrorString(once_k));
        return I2V_ERR_LAUNCH;
    }
    const int tiles = i2v_cdiv(p.M, BM) * i2v_cdiv(p.N, BN);
    if (p.flags & I2V_EPI_MASK)
        conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true, KG><<<dim3(tiles, 1, 1), THREADS * KG, need, st>>>(p);
    else
        conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, false, KG><<<dim3(tiles, 1, 1), THREADS * KG, need, st>>>(p);
    return I2V_OK;
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
void launch_tile(const ConvP& p, hipStream_t st) {
    constexpr int BM = WAVES_M * TM * 16, BN = WAVES_N * TN * 16;
    const int tiles = i2v_cdiv(p.M, BM) * i2v_cdiv(p.N, BN);
    const size_t stage = (size_t)(2 * (BM + BN) * BKS + p.ktab_entries) * sizeof(float);
    const size_t epi = (size_t)BM * (BN + 4) * sizeof(float);
    const size_t lds = stage > epi ? stage : epi;
    static bool once = [] {
        set_max_lds(conv_igemm_f32<WAVES_M, WAVES_N, TM, TN>);
        return true;
    }();
    (void)once;
    const dim3 grid(tiles, p.splitk, p.nbatch > 1 ? p.nbatch : 1);
    // pointwise layers / plain GEMMs whose split-K (if any) is finished in the kernel: the lean specialisation
    const bool pointwise = p.KH == 1 && p.KW == 1 && p.pad == 0 && p.pad_x == 0 && p.stride == 1 && p.ostride == 1 &&
                           p.Ho == p.H && p.Wo == p.W && (p.N & 3) == 0 && (p.K & 3) == 0 && (p.splitk <= 1 || p.ws);
    if (pointwise && g_i2v_tuning[I2V_TUNE_CONV_GEMM]) {
        static bool once_g = [] {
            set_max_lds(conv_gemm_f32<WAVES_M, WAVES_N, TM, TN>);
            set_max_lds(conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, true>);
            set_max_lds(conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true>);
            set_max_lds(conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, false, 1, 1>);
            set_max_lds(conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true, 1, 1>);
            return true;
        }();
        (void)once_g;
        const size_t lds_g = (size_t)(2 * (BM + BN) * BKS) * sizeof(float);
        // round 6: LDS-DMA staging (I2V_TUNE_GEMM_DMA: 1 = 32-k stages, 2 = 16-k stages / half the LDS; 0 = through registers)
        const int dma = g_i2v_tuning[I2V_TUNE_GEMM_DMA];
        if (dma > 0 && !p.clk) {
            const bool mask = (p.flags & I2V_EPI_MASK) != 0;
            if (dma == 2) {
                if (mask) conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true, 1, 2><<<grid, THREADS, lds_g / 2, st>>>(p);
                else conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, false, 1, 2><<<grid, THREADS, lds_g / 2, st>>>(p);
            } else {
                if (mask) conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true, 1, 1><<<grid, THREADS, lds_g, st>>>(p);
                else conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, false, 1, 1><<<grid, THREADS, lds_g, st>>>(p);
            }
            return;
        }
        if (p.flags & I2V_EPI_MASK) conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, false, true><<<grid, THREADS, lds_g, st>>>(p);
        else if (p.clk) conv_gemm_f32<WAVES_M, WAVES_N, TM, TN, true><<<grid, THREADS, lds_g, st>>>(p);
        else conv_gemm_f32<WAVES_M, WAVES_N, TM, TN><<<grid, THREADS, lds_g, st>>>(p);
        return;
    }
    conv_igemm_f32<WAVES_M, WAVES_N, TM, TN><<<grid, THREADS, lds, st>>>(p);
}

struct TileCfg { int bm, bn; float eff; };
// eff: relative MFMA efficiency of the tile shape (operand reuse per LDS byte), from measurements
constexpr TileCfg kTiles[] = {{128, 128, 1.00f}, {128, 64, 0.97f}, {96, 64, 0.95f}, {80, 64, 0.95f}, {64, 64, 0.93f},
                              {32, 64, 0.80f}};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

int run_conv(ConvP p, hipStream_t st, void* split_ws = nullptr, size_t split_ws_bytes = 0) {
    p.M = p.B * p.Ho * p.Wo;
    p.N = p.Cout;
    p.K = p.KH * p.KW * p.Cin;
    p.lgCin = ilog2_exact(p.Cin);
    if (!(p.KH == 1 && p.KW == 1 && p.pad == 0 && p.pad_x == 0)) {
        if (p.K > KTAB_MAX * 4 || p.KH * p.KW > 64 || ((long long)(p.KH * p.W + p.KW) * p.Cin) >= (1ll << 24)) {
            i2v_set_error("conv: filter %dx%dx%d too large for the tap table", p.KH, p.KW, p.Cin);
            return I2V_ERR_UNSUPPORTED;
        }
    }
    p.ktab_entries = (p.KH == 1 && p.KW == 1 && p.pad == 0 && p.pad_x == 0) ? 4 : ((p.K + BKS - 1) / BKS) * (BKS / 4);
    const long long xb = (long long)p.B * p.H * p.W * p.Cin * 4, wb = (long long)p.N * p.K * 4;
    const long long halo = std::max(0ll, (long long)(p.pad * p.W + p.pad_x) * p.Cin * 4);    // the kernel's descriptor starts this much earlier
    if (xb + halo >= (1ll << 31) || wb >= (1ll << 31)) {
        i2v_set_error("conv: operand larger than 2 GiB (32-bit buffer offsets)");
        return I2V_ERR_UNSUPPORTED;
    }
    p.x_bytes = (unsigned)xb;
    p.w_bytes = (unsigned)wb;
    p.clk = g_clk;
    p.ablate = g_ablate;
    const int force = p.force_tile;
    const int ksteps = i2v_cdiv(p.K, BKS);
    // tile + split-K choice: minimise (rounds over the 256 CUs) x (MACs per workgroup) / efficiency.
    // The M of a 600x1000 frame pair at stride 16 is only 4788 rows, so wave quantisation decides
    // the shape; skinny GEMMs (vrd FCs: M = 128 rows) fill the chip by splitting K.
    auto plan = [&](int c, int& splitk) {
        const long long t = (long long)i2v_cdiv(p.M, kTiles[c].bm) * i2v_cdiv(p.N, kTiles[c].bn) * (p.nbatch > 1 ? p.nbatch : 1);
        splitk = 1;
        if (t < g_split_below && ksteps >= 8 && p.ostride == 1 && p.nbatch <= 1) {
            const int target = (p.M <= 256 && g_split_target_skinny > 0) ? g_split_target_skinny : g_split_target;
            splitk = (int)((target * NUM_CU + t - 1) / t);
            splitk = splitk > ksteps / 4 ? ksteps / 4 : splitk;
            if (splitk < 1) splitk = 1;
            // Under round 4's rule (SPLIT_ATOMICS == 2) a large output split more than kSplitInKernelMax ways leaves the in-kernel
            // finish: fp32 atomics, a clear in front, a separate epilogue pass and the generic kernel -- none of which the 1.05
            // below prices.  Found on 600x801 frames (round 6, tools/size_probe.py): layer3 conv1 of ONE frame (M = 1900) took
            // 128x128 tiles x 8 splits = 240 workgroups, "one round", and ran at 42 TF on conv_igemm_f32 where the 64x64 x 4 plan
            // runs at 80 on conv_gemm_f32 -- the loader-fed step was 7 % slower on the SMALLER frames.  Such outputs split at most
            // kSplitInKernelMax ways.
            if (g_split_atomics == 2 && splitk > kSplitInKernelMax && (long long)p.M * p.N >= (1 << 18)) splitk = kSplitInKernelMax;
        }
        const long long blocks = t * splitk;
        const long long rounds = (blocks + NUM_CU - 1) / NUM_CU;
        // beyond ~4 rounds several workgroups share a CU and the tail matters less
        const double r = rounds <= 4 ? (double)rounds : (double)blocks / NUM_CU + 0.5;
        double cost = r * kTiles[c].bm * kTiles[c].bn * (double)i2v_cdiv(ksteps, splitk) / kTiles[c].eff;
        if (splitk > 1) cost *= 1.05;     // memset + atomics + separate epilogue pass
        return cost;
    };
    int cfg = 0, splitk = 1;
    double best = 1e300;
    for (int c = 0; c < kNumTiles; ++c) {
        int sk;
        const double cost = plan(c, sk);
        if (cost < best) { best = cost; cfg = c; splitk = sk; }
    }
    // HBM-bound pointwise layers (at most four K stages over many rows: the 64 -> 256 / 128 -> 512 expansions of layer1 / layer2 and
    // their data gradients): the model's MAC count cannot tell the tiles apart (all within 2 %) and picks 128x64; measured, the
    // 80x64 tile streams best (tools/pers_bench.py, two frames: layer1 conv3 44.0 against 50.7 us, layer2 conv3 33.3 against 37.6)
    if (g_i2v_tuning[I2V_TUNE_STREAM_TILE] && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.nbatch <= 1 && ksteps <= 4 && p.M >= 16384) {
        cfg = 3;
        plan(cfg, splitk);
    }
    // batched launches with at most four K stages (the Winograd planes of the 64- and 128-channel layers): per-workgroup
    // set-up and epilogue dominate and the model underrates the smallest tile (measured 23.5 vs 27.4 us at 64 channels)
    if (p.nbatch > 1 && ksteps <= 4) { cfg = kNumTiles - 1; plan(cfg, splitk); }
    // the long skinny GEMM of the relation head (fc6 forward: 128 rows, K = 50176): 128x64 tiles instead of the 128x128 the
    // model picks -- twice the workgroups, each half as heavy.  Alone 436 vs 412 us, inside the two-stream step 4.88 vs
    // 4.93 ms (the same effect as with the fused update's tile: lighter workgroups give the other stream its turn sooner)
    if (g_big_fc_tile >= 0 && g_big_fc_tile < kNumTiles && p.M <= 256 && p.K >= 16384) { cfg = g_big_fc_tile; plan(cfg, splitk); }
    if (force >= 0 && force < kNumTiles) { cfg = force; plan(cfg, splitk); }
    p.splitk = splitk;
    p.k_per_split = i2v_cdiv(ksteps, splitk) * BKS;
    p.splitk = i2v_cdiv(p.K, p.k_per_split);
    // Intra-workgroup K split (conv_gemm_f32<.., KG>): a pointwise GEMM the plan would split over K through memory runs as ONE
    // 16-wave workgroup per tile whose four wave groups each take a quarter of K and meet in LDS -- same waves per SIMD, no partial
    // tile leaves the CU.  Its four stage-buffer sets leave room for one workgroup per CU, so it pays when the tiles of ONE round
    // cover most of the chip: of the tiles 80x64 / 64x64 / 48x64 / 32x64 the smallest (least work per CU) with at most 256 tiles,
    // if that is at least 180 (70 % of the CUs); otherwise the split across workgroups stays (layer3 conv1: 240 tiles of 80x64 for
    // a frame pair, 200 of 48x64 for one frame; tools/kgroup_bench.py).  K a multiple of 4 x 32 with >= 2 stages per group.
    int kg_tile = -1;
    {
        const bool pw = p.KH == 1 && p.KW == 1 && p.pad == 0 && p.pad_x == 0 && p.stride == 1 && p.ostride == 1 && p.Ho == p.H &&
                        p.Wo == p.W && (p.N & 3) == 0 && (p.K & 3) == 0 && g_i2v_tuning[I2V_TUNE_CONV_GEMM];
        const int kgn = g_i2v_tuning[I2V_TUNE_KGROUPS] == 2 ? 2 : kKGroups;       // wave groups: 2 (round 5), or 4 (1 / 4)
        if (g_i2v_tuning[I2V_TUNE_KGROUPS] && pw && !p.clk && p.nbatch <= 1 && p.splitk >= 2 && force < 0 &&
            p.K % (kgn * BKS) == 0 && p.K / kgn >= 2 * BKS &&
            (long long)p.M * p.N >= (1 << 18)) {
            static const int kg_bm[4] = {80, 64, 48, 32};
            const int nt = i2v_cdiv(p.N, 64);
            for (int c = 3; c >= 0; --c) {                       // smallest tile first
                const int t = i2v_cdiv(p.M, kg_bm[c]) * nt;
                if (t <= NUM_CU) { if (t >= (NUM_CU * 7) / 10) kg_tile = c; break; }
            }
            if (kg_tile >= 0) { p.splitk = 1; p.k_per_split = p.K; }
        }
    }
    const long long ntiles = (long long)i2v_cdiv(p.M, kTiles[cfg].bm) * i2v_cdiv(p.N, kTiles[cfg].bn);
    p.ws = nullptr;
    p.cnt = nullptr;
    const size_t ws_need = (size_t)p.splitk * ntiles * kTiles[cfg].bm * kTiles[cfg].bn * sizeof(float);
    // in-kernel finish pays where the output is large (atomics and the extra epilogue pass scale with it);
    // for the small FC outputs of the vrd head the atomics are cheap and a serial sum of many splits is not
    // round 5: the ordered finish takes ANY number of splits (rounds of kSplitInKernelMax) and any output size, so that no
    // forward or data-gradient GEMM of the relation head depends on arrival order (SPLIT_ATOMICS = 2 restores round 4's rule
    // everywhere: atomics beyond four splits and for outputs under 2^18 elements; 1: atomics always)
    // SPLIT_ATOMICS == 0 (what the relation step's head context selects, ops.LaunchContext(ordered=True)): ordered for every
    // shape.  The process default is 2, round 4's rule: measured on configs[2], ordering every reduction of the step -- its
    // 8-16-way filter-gradient splits, the bias sums of netD_style's 37500-row projections -- costs 46.2 -> 48.1 ms.
    const bool r4_ok = p.splitk <= kSplitInKernelMax && (long long)p.M * p.N >= (1 << 18);
    const bool wants_ws = p.splitk > 1 && g_split_atomics != 1 && (r4_ok || g_split_atomics == 0) &&
                          ntiles <= kSplitCounters && ws_need < (1ull << 31) - (64u << 20);
    // dry == 2: the workspace this shape would use (0: none)
    if (p.dry == 2) return wants_ws ? (int)std::min<size_t>(kSplitCounterBytes + ws_need, 0x7FFFFFFF) : 0;
    const bool in_kernel = wants_ws && split_ws && kSplitCounterBytes + ws_need <= split_ws_bytes;
    // dry == 1: 0 = y needs no clear (no split, or the split is finished in-kernel), else the split factor
    if (p.dry) return (p.splitk > 1 && !in_kernel) ? p.splitk : 0;
    if (p.splitk > 1 && !in_kernel && g_split_atomics == 0) ++g_ordered_fallbacks;      // order was asked for; the workspace (or a size cap) refused
    if (in_kernel) {
        p.cnt = reinterpret_cast<int*>(split_ws);
        p.ws = reinterpret_cast<float*>(static_cast<char*>(split_ws) + kSplitCounterBytes);
    }
    const long long ytotal = (long long)p.M * p.N;
    if (p.splitk > 1 && !p.ws && !(p.flags & I2V_EPI_ZEROED)) hipMemsetAsync(p.y, 0, (size_t)ytotal * sizeof(float), st);
    if (kg_tile >= 0 && g_i2v_tuning[I2V_TUNE_KGROUPS] == 2) {
        switch (kg_tile) {
            case 0: return launch_kgroups<1, 4, 5, 1, 2>(p, st);
            case 1: return launch_kgroups<2, 2, 2, 2, 2>(p, st);
            case 2: return launch_kgroups<1, 4, 3, 1, 2>(p, st);
            default: return launch_kgroups<2, 2, 1, 2, 2>(p, st);
        }
    }
    if (kg_tile >= 0) {
        switch (kg_tile) {
            case 0: return launch_kgroups<1, 4, 5, 1>(p, st);
            case 1: return launch_kgroups<2, 2, 2, 2>(p, st);
            case 2: return launch_kgroups<1, 4, 3, 1>(p, st);
            default: return launch_kgroups<2, 2, 1, 2>(p, st);
        }
    }
    switch (cfg) {
        case 0: launch_tile<2, 2, 4, 4>(p, st); break;
        case 1: launch_tile<2, 2, 4, 2>(p, st); break;
        case 2: launch_tile<2, 2, 3, 2>(p, st); break;
        case 3: launch_tile<1, 4, 5, 1>(p, st); break;
        case 4: launch_tile<2, 2, 2, 2>(p, st); break;
        default: launch_tile<2, 2, 1, 2>(p, st); break;
    }
    if (p.splitk > 1 && !p.ws && (p.flags & (I2V_EPI_SCALE | I2V_EPI_BIAS | I2V_EPI_RESIDUAL | I2V_EPI_RELU | I2V_EPI_MASK))) {
        if (p.N % 4 == 0)
            conv_epilogue_kernel<<<(int)fmin((double)i2v_cdiv(ytotal / 4, 256), 4096.0), 256, 0, st>>>(
                p.y, p.scale, p.shift, p.res, p.mask, ytotal / 4, p.N, p.flags);
        else
            conv_epilogue_scalar_kernel<<<(int)fmin((double)i2v_cdiv(ytotal, 256), 4096.0), 256, 0, st>>>(
                p.y, p.scale, p.shift, p.res, p.mask, ytotal, p.N, p.flags);
    }
    return I2V_OK;
}


// ---------------------------------------------------------------- dgrad helper
// wt[c][KH-1-ky][KW-1-kx][n] = w[n][ky][kx][c]: the filter of the transposed conv.
__global__ void weight_dgrad_layout(const float* __restrict__ w, float* __restrict__ wt, int Cout, int KH, int KW,
                                    int Cin, const float* __restrict__ nscale = nullptr) {
    const long long total = (long long)Cout * KH * KW * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        // i indexes wt: (c, ky', kx', n) with n fastest (coalesced writes)
        int n = i % Cout;
        long long t = i / Cout;
        int kx = t % KW; t /= KW;
        int ky = t % KH;
        int c = t / KH;
        // nscale: a per-filter factor on gy (the frozen-BN scale between the conv and the tensor gy belongs to) folded into
        // the transposed filter: dgrad(gy * s, w) == dgrad(gy, diag(s) w)
        const float v = w[(((long long)n * KH + (KH - 1 - ky)) * KW + (KW - 1 - kx)) * Cin + c];
        wt[i] = nscale ? v * nscale[n] : v;
    }
}

// Sub-filter of a strided dgrad: the input pixels of one parity class (iy % s, ix % s) only see the taps
// ky = ky0 + s*t: wt[c][Ty-1-ty][Tx-1-tx][n] = w[n][ky0 + s*ty][kx0 + s*tx][c].
__global__ void weight_dgrad_sub_layout(const float* __restrict__ w, float* __restrict__ wt, int Cout, int KH, int KW,
                                        int Cin, int s, int ky0, int kx0, int Ty, int Tx) {
    const long long total = (long long)Cin * Ty * Tx * Cout;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        int n = i % Cout;
        long long t = i / Cout;
        int ux = t % Tx; t /= Tx;
        int uy = t % Ty;
        int c = t / Ty;
        const int ky = ky0 + s * (Ty - 1 - uy), kx = kx0 + s * (Tx - 1 - ux);
        wt[i] = w[(((long long)n * KH + ky) * KW + kx) * Cin + c];
    }
}

// ---------------------------------------------------------------- wgrad
//   gw[n][k] (+)= sum_m gy[m][n] * A[m][k]     reduction over the output pixels m.
// Both operands arrive reduction-major from HBM (gy rows are n-contiguous, im2col rows
// are c-contiguous), so the staging pass transposes them into the [row][kk] LDS image
// the MFMA fragments want; split over m across blockIdx.y with fp32 atomics.
struct WgP {
    const float* x; const float* gy; float* gw; int direct;
    const float* row_scale;                // gw[n][:] = row_scale[n] * sum (a frozen-BN scale on gy applied where the sum ends), or NULL
    int xcd_remap;                         // (tile, split, plane) from the dispatch index so that a split's tiles share an XCD (launch_wgrad)
    int r_tiles, r_splits, r_total;        // with xcd_remap: the logical grid (tiles x splits x planes = r_total workgroups) behind the 1-D launch
    int nbatch;                            // > 1: blockIdx.z selects one of nbatch independent GEMMs (the planes of a Winograd filter gradient)
    long long bsx, bsg, bsw;               // element strides between the batches of x, gy and gw
    float* sgd_m; float lr, mom, wd;       // sgd_m != NULL: gw is the PARAMETER, updated in place (fused SGD)
    unsigned x_bytes, gy_bytes;            // buffer descriptor sizes (v2 kernel)
    int B, H, W, Cin, Cout, KH, KW, stride, pad, Ho, Wo, M, N, K, m_per_split, lgCin;
    unsigned long long* clk;               // diagnostic (i2v_conv_debug_clock): per-workgroup stamps, CLK instantiation only
    int abl;                               // diagnostic instantiation only: ablation bits (i2v_conv_set_tile bits 10-12)
    // ordered finish of a split over pixels (round 5): partial tiles through the caller's split workspace, summed in split
    // order by the last workgroup to arrive at the tile's counter (the protocol of conv_igemm_f32's split-K finish): the sum
    // does not depend on arrival order, no clear of gw in front.  ord_ws == NULL: fp32 atomics into a cleared gw.
    float* ord_ws; int* ord_cnt; int ord_splits, ord_tiles, ord_acc;      // ord_acc: gw += sum (beta = 1) instead of gw = sum
    // two-pass ordered finish (round 6): a split of MORE parts than one finisher should read (or of the first-generation kernel,
    // which has no in-kernel finish): every split stores its partial filter in gw's own layout at part_ws[(split * planes + plane)
    // * N * K ...] with plain stores, and a reduce pass (wgrad_reduce_kernel, or the Winograd filter gradient's final transform)
    // sums the parts in split order.  No counter, no clear of gw, all of the chip reads the parts.
    float* part_ws;
    int part_cap;                          // a caller-owned part_ws (launch_wgrad's ext_part): the parts it has room for
};

template <int BM, int BN>   // BM over n (Cout), BN over k; 4 waves as 2x2, 64x64 tiles: BM=BN=64 -> wave 32x32
__global__ void __launch_bounds__(THREADS)
conv_wgrad_f32(const WgP p) {
    constexpr int MI = BM / 64, NI = BN / 64;
    __shared__ __attribute__((aligned(16))) float As[BM * LDS_ROW];   // [n][mm]
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_ROW];   // [k][mm]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = (p.K + BN - 1) / BN;
    const int n0 = (blockIdx.x / tiles_k) * BM, k0 = (blockIdx.x % tiles_k) * BN;
    const int mbeg = blockIdx.y * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging roles: a stage is 16 reduction rows (m) x BM (or BN) columns; one float4 = 4 columns
    constexpr int A_V = 16 * BM / 4 / THREADS, B_V = 16 * BN / 4 / THREADS;
    const int fr = lane & 31, fh = lane >> 5;
    for (int ms = mbeg; ms < mend; ms += 16) {
        float4 ra[A_V], rb[B_V];
#pragma unroll
        for (int q = 0; q < A_V; ++q) {
            const int slot = tid + q * THREADS;
            const int mm = slot / (BM / 4), col = (slot % (BM / 4)) * 4;
            const int m = ms + mm, n = n0 + col;
            ra[q] = make_float4(0, 0, 0, 0);
            if (m < mend && n < p.N) {
                const float* g = p.gy + (long long)m * p.N + n;
                if (n + 3 < p.N && (p.N & 3) == 0) ra[q] = *(const float4*)g;
                else { ra[q].x = g[0]; if (n + 1 < p.N) ra[q].y = g[1]; if (n + 2 < p.N) ra[q].z = g[2]; if (n + 3 < p.N) ra[q].w = g[3]; }
            }
        }
#pragma unroll
        for (int q = 0; q < B_V; ++q) {
            const int slot = tid + q * THREADS;
            const int mm = slot / (BN / 4), col = (slot % (BN / 4)) * 4;
            const int m = ms + mm, k = k0 + col;
            rb[q] = make_float4(0, 0, 0, 0);
            if (m < mend && k < p.K) {
                int kpos, c;
                if (p.lgCin >= 0) { kpos = k >> p.lgCin; c = k & (p.Cin - 1); } else { kpos = k / p.Cin; c = k - kpos * p.Cin; }
                const int ky = kpos / p.KW, kx = kpos - ky * p.KW;
                const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
                const int iy = oy * p.stride - p.pad + ky, ix = ox * p.stride - p.pad + kx;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    rb[q] = *(const float4*)(p.x + (((long long)b * p.H + iy) * p.W + ix) * p.Cin + c);
            }
        }
        __syncthreads();      // previous stage fully consumed
#pragma unroll
        for (int q = 0; q < A_V; ++q) {
            const int slot = tid + q * THREADS;
            const int mm = slot / (BM / 4), col = (slot % (BM / 4)) * 4;
            As[(col + 0) * LDS_ROW + mm] = ra[q].x; As[(col + 1) * LDS_ROW + mm] = ra[q].y;
            As[(col + 2) * LDS_ROW + mm] = ra[q].z; As[(col + 3) * LDS_ROW + mm] = ra[q].w;
        }
#pragma unroll
        for (int q = 0; q < B_V; ++q) {
            const int slot = tid + q * THREADS;
            const int mm = slot / (BN / 4), col = (slot % (BN / 4)) * 4;
            Bs[(col + 0) * LDS_ROW + mm] = rb[q].x; Bs[(col + 1) * LDS_ROW + mm] = rb[q].y;
            Bs[(col + 2) * LDS_ROW + mm] = rb[q].z; Bs[(col + 3) * LDS_ROW + mm] = rb[q].w;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float4 av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = *(const float4*)&As[(wm * (BM / 2) + i * 32 + fr) * LDS_ROW + s * 8 + fh * 4];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = *(const float4*)&Bs[(wn * (BN / 2) + j * 32 + fr) * LDS_ROW + s * 8 + fh * 4];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int k = k0 + wn * (BN / 2) + j * 32 + fr;
        if (k >= p.K) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (n < p.N) {
                    const long long o = (long long)n * p.K + k;
                    if (p.sgd_m) {          // g' = g + wd*p ; m = mom*m + g' ; p -= lr*m   (same order as sgd_momentum_kernel)
                        const float pv = p.gw[o];
                        const float mv = p.mom * p.sgd_m[o] + (acc[i][j][r] + p.wd * pv);
                        p.sgd_m[o] = mv;
                        p.gw[o] = pv - p.lr * mv;
                    } else if (p.direct) {
                        p.gw[o] = acc[i][j][r];
                    } else if (p.part_ws) {
                        p.part_ws[(long long)blockIdx.y * p.N * p.K + o] = acc[i][j][r];
                    } else {
                        atomicAdd(p.gw + o, acc[i][j][r]);
                    }
                }
            }
    }
}

// ---------------------------------------------------------------- wgrad v2
// Same GEMM as conv_wgrad_f32 on the conv_igemm_f32 machinery: 16x16x4 MFMAs, 32 reduction rows per
// stage, swizzled 128-B LDS rows, register prefetch + double-buffered LDS.  Both operands arrive
// reduction-major, so every thread owns 4x4 blocks (4 consecutive pixels x 4 columns): four 16-B loads,
// an in-register transpose, four ds_write_b128 -- no scalar LDS traffic, no bank-conflicted transposes.
// FUSED_SGD: the instantiation that runs the SGD update in its epilogue (prefetches the filter / momentum tiles:
// +34 VGPRs, four waves per SIMD instead of five -- which the plain gradient kernel should not pay)
// LDS column swizzle of the filter-gradient kernel.  A fragment read touches 16 consecutive rows of one 16-row group: any
// bijection of (row >> 1) & 7 keeps it conflict-free, and a term in row >> 4 is constant there.  The transposing stores of a
// 16-lane group go to rows 4 cg + i (cg = 0..15, four columns per thread): (row >> 1) & 7 alone takes FOUR values there -- a
// four-way bank conflict on every ds_write_b128; with bit 4 of the row folded in it takes eight (two-way, the best a
// 4-column block allows: the rows of a group share their parity, which picks the half of the 256-byte bank row).
__device__ inline int wswz(int row) { return ((row >> 1) & 7) ^ ((row >> 4) & 1); }
constexpr bool WGRAD_ROWMAJOR = false;     // LDS image of the filter-gradient kernel: the transposed [column][pixel] one (false), or [pixel][column] as the operands
                                           // arrive (true: no register transposes, conflict-free 16-byte stores, one-float fragment reads merged into
                                           // ds_read2st64_b32 -- bit-equal, measured 3 % slower on the layer3 shapes: 113.8 vs 110.3 us)

// DMA (round 6, I2V_TUNE_WGRAD_DMA; pointwise / linear problems only: pixel index in == pixel index out): the stage tiles go
// from global memory to LDS without passing through registers (lds_dma16), as in conv_gemm_f32.  The DMA lands lane-linear, so the
// LDS image is the [pixel][column] one (WGRAD_ROWMAJOR's) with its 16-column-group swizzle applied to the SOURCE column: a 1-KB
// piece is 4 (2) consecutive pixel rows of a 64- (128-) column tile.  No staging registers (64 of the 128 VGPRs of the 64x64
// form), no register transposes, no ds_write; the stage offset rides in the scalar offset of the request, the per-lane offsets
// are loop constants.  Same MFMA order per accumulator: bit-equal to the register-staged forms.
template <int TM, int TN, bool FUSED_SGD = false, bool CLK = false, bool DMA = false>       // tile = (2*TM*16) filters x (2*TN*16) taps, 4 waves as 2x2
__global__ void __launch_bounds__(THREADS)
conv_wgrad2_f32(const WgP p_in) {
    WgP p = p_in;
    unsigned long long c_rt0 = 0, c_t0 = 0, c_t1 = 0, c_t2 = 0;
    if constexpr (CLK) { c_rt0 = __builtin_amdgcn_s_memrealtime(); c_t0 = __builtin_amdgcn_s_memtime(); }
    // XCD-aware order (p.xcd_remap): workgroups are dealt to the 8 XCDs round robin in dispatch order, and every XCD has its own
    // L2.  All tiles of one pixel range (one split of one plane: a GROUP) read the same rows of gy and x; dealt in (tile, split)
    // order they land on all 8 XCDs and every L2 fetches those rows again (PMC: 3.6x the algorithmic bytes on the layer3
    // shapes).  Round 2 put group s on XCD s % 8 -- possible only when the group count is a multiple of 8, which the
    // pixel split rarely is (254 splits for layer1's expansion at 8 frames).  Round 4: the launch is 1-D and XCD g owns the
    // CONTIGUOUS range [g W / 8, (g + 1) W / 8) of the group-major workgroup order (W = tiles x groups): every XCD gets the
    // same number of workgroups (+-1) whatever the group count, a group lives on one XCD (two where a range boundary cuts
    // it).  The launch is padded to a multiple of 8; the <= 7 surplus workgroups leave at once.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_remap) {
        const int L = (int)blockIdx.x, g = L & 7, r = L >> 3;
        const long long W = p.r_total;
        const int start = (int)((g * W) >> 3), count = (int)(((g + 1) * W) >> 3) - start;
        if (r >= count) return;
        const int idx = start + r, grp = idx / p.r_tiles;
        bx = idx - grp * p.r_tiles;
        by = grp % p.r_splits;
        bz = grp / p.r_splits;
    }
    if (p.nbatch > 1) {
        p.x += (long long)bz * p.bsx;
        p.gy += (long long)bz * p.bsg;
        p.gw += (long long)bz * p.bsw;
    }
    constexpr int BMW = 2 * TM * 16, BNW = 2 * TN * 16;
    constexpr bool ROWMAJOR = WGRAD_ROWMAJOR || DMA;
    static_assert(!DMA || (!FUSED_SGD && !CLK), "the LDS-DMA form is the plain gradient kernel");
    constexpr int A_BLK = BMW * 2, B_BLK = BNW * 2;             // (cols/4) * 8 row-groups
    constexpr int NBLK = DMA ? 1 : (A_BLK + B_BLK + THREADS - 1) / THREADS;     // (DMA: the register-staging roles below are dead code)
    constexpr int STAGE_FLOATS = 2 * (BMW + BNW) * BKS;
    constexpr int CROW = BNW + 4;
    constexpr int SMEM_FLOATS = STAGE_FLOATS > BMW * CROW ? STAGE_FLOATS : BMW * CROW;
    __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
    float (*As)[BMW * BKS] = reinterpret_cast<float (*)[BMW * BKS]>(smem);
    float (*Bs)[BNW * BKS] = reinterpret_cast<float (*)[BNW * BKS]>(smem + 2 * BMW * BKS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_k = (p.K + BNW - 1) / BNW;
    const int n0 = (bx / tiles_k) * BMW, k0 = (bx % tiles_k) * BNW;
    const int mbeg = by * p.m_per_split, mend = min(p.M, mbeg + p.m_per_split);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc((void*)p.gy, 0, p.gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFFFF0u;

    // per-thread block roles (fixed over the m loop)
    bool is_a[NBLK], live[NBLK];
    int cg[NBLK], m4[NBLK], coff[NBLK], ky[NBLK], kx[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q) {
        const int id = tid + q * THREADS;
        is_a[q] = id < A_BLK;
        const int j = is_a[q] ? id : id - A_BLK;
        const int groups = (is_a[q] ? BMW : BNW) / 4;
        live[q] = id < A_BLK + B_BLK;
        cg[q] = j % groups;
        m4[q] = j / groups;
        ky[q] = kx[q] = 0;
        if (is_a[q]) {
            const int n = n0 + 4 * cg[q];
            coff[q] = n;
            live[q] = live[q] && n < p.N;
        } else {
            const int k = k0 + 4 * cg[q];
            live[q] = live[q] && k < p.K;
            int kpos, c;
            const int kk = live[q] ? k : 0;
            if (p.lgCin >= 0) { kpos = kk >> p.lgCin; c = kk & (p.Cin - 1); } else { kpos = kk / p.Cin; c = kk - kpos * p.Cin; }
            ky[q] = kpos / p.KW;
            kx[q] = kpos - ky[q] * p.KW;
            coff[q] = c;
        }
    }
    float4 r[NBLK][4];
    // pointwise layers on the same grid (and linear layers): input pixel index == output pixel index, no pixel arithmetic
    const bool lin = p.KH == 1 && p.KW == 1 && p.pad == 0 && p.stride == 1;     // uniform
    // other filters: the (ox, oy, b) of a block's first pixel is divided out ONCE and then carried from stage to stage
    // (stages advance by 32 pixels; one conditional wrap suffices while a row holds at least 32 pixels) -- the three
    // integer divisions per stage and block were ~100 of the ~250 vector instructions a stage issues beside its 32 MFMAs
    // (tools/wgrad_phase.py: 6.2k cycles per stage against 4.1k of MFMA time at 4 workgroups per CU)
    const bool carry = p.Wo >= BKS;
    int sx[NBLK], sy[NBLK], sb[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q) {
        sx[q] = sy[q] = sb[q] = 0;
        if (lin) continue;                   // uniform: the skinny GEMMs of the relation head live ~4 stages, set-up counts
        const int m0 = mbeg + 4 * m4[q];
        sx[q] = m0 % p.Wo;
        const int tt = m0 / p.Wo;
        sy[q] = tt % p.Ho;
        sb[q] = tt / p.Ho;
    }
    // No divergent control flow around the loads: the A/B role of a block is uniform per wave (A_BLK is a multiple of
    // 128), so the descriptor is picked with a scalar select, and a masked element gets the 2 GiB bit OR-ed into its
    // offset (the buffer returns 0) -- written as `cond ? off : OOB` the compiler wraps every load in its own branch.
    auto gload = [&](int ms) {
#pragma unroll
        for (int q = 0; q < NBLK; ++q) {
            const bool a_u = __builtin_amdgcn_readfirstlane((int)is_a[q]) != 0;
            const __amdgpu_buffer_rsrc_t rs = a_u ? gr : xr;
            const int m0 = ms + 4 * m4[q];
            int ox = sx[q], oy = sy[q], b = sb[q];
            if (!a_u && !lin) {
                if (carry) {                // next stage: 32 pixels on
                    int nx = ox + BKS, ny = oy, nb = b;
                    if (nx >= p.Wo) { nx -= p.Wo; ++ny; if (ny == p.Ho) { ny = 0; ++nb; } }
                    sx[q] = nx; sy[q] = ny; sb[q] = nb;
                } else {                    // short rows (the 7x7 / 4x4 maps of the ROI head): divide
                    ox = m0 % p.Wo;
                    const int tt = m0 / p.Wo;
                    oy = tt % p.Ho;
                    b = tt / p.Ho;
                }
            }
            // The offsets are computed under uniform branches, the LOADS are not: a load inside a branch makes the number of
            // loads in flight unknowable to the compiler, which then drains everything (vmcnt(0)) before the first LDS
            // store -- the fused-SGD form would wait for its prefetched filter / momentum tiles there (fc6: +17 %).
            unsigned offs[4];
            if (a_u || lin) {               // row m of gy / of x: no pixel arithmetic
                const int rowlen = a_u ? p.N : p.Cin;
#pragma unroll
                for (int t = 0; t < 4; ++t) offs[t] = (unsigned)((m0 + t) * rowlen + coff[q]) * 4u;
            } else if (p.Wo >= 4) {
                // the block's 4 consecutive pixels: offsets from the first one's by increments (a pixel past the end of the
                // row moves to the next row -- or the next image -- by one precomputed delta: rows hold >= 4 pixels)
                const int iy0 = oy * p.stride - p.pad + ky[q], ix0 = ox * p.stride - p.pad + kx[q];
                const int base = ((b * p.H + iy0) * p.W + ix0) * p.Cin + coff[q];
                const bool last_row = oy + 1 == p.Ho;
                const int iy1 = last_row ? ky[q] - p.pad : iy0 + p.stride;                     // row of the pixels after a wrap
                const int wrap_delta = ((last_row ? p.H - (p.Ho - 1) * p.stride : p.stride) * p.W - p.Wo * p.stride) * p.Cin;
                const bool y0_in = iy0 >= 0 && iy0 < p.H, y1_in = iy1 >= 0 && iy1 < p.H;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool w = ox + t >= p.Wo;
                    const int ix = ix0 + t * p.stride - (w ? p.Wo * p.stride : 0);
                    const bool inside = (w ? y1_in : y0_in) && ix >= 0 && ix < p.W;
                    offs[t] = ((unsigned)(base + t * p.stride * p.Cin + (w ? wrap_delta : 0)) * 4u) | (inside ? 0u : 0x80000000u);
                }
            } else {                        // rows shorter than a block: pixel by pixel
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int iy = oy * p.stride - p.pad + ky[q], ix = ox * p.stride - p.pad + kx[q];
                    const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                    offs[t] = ((unsigned)(((b * p.H + iy) * p.W + ix) * p.Cin + coff[q]) * 4u) | (inside ? 0u : 0x80000000u);
                    const bool wx = ox + 1 == p.Wo;
                    const bool wy = wx && oy + 1 == p.Ho;
                    ox = wx ? 0 : ox + 1;
                    oy = wy ? 0 : (wx ? oy + 1 : oy);
                    b += wy ? 1 : 0;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned off = offs[t] | ((live[q] && m0 + t < mend) ? 0u : 0x80000000u);
                r[q][t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NBLK; ++q) {
            if (tid + q * THREADS >= A_BLK + B_BLK) continue;
            float* base = is_a[q] ? As[buf] : Bs[buf];
            if constexpr (ROWMAJOR) {
                // [pixel][column] image, as the operands arrive: no transposition, four conflict-free 16-byte stores (the 16 lanes
                // of a store cover one 256-byte row segment); the 16-column groups of a row are XOR-ed with (pixel >> 2) & 3 so
                // that the four pixel rows a fragment read touches fall into four different bank quarters
                const int cols = is_a[q] ? BMW : BNW;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    *(float4*)&base[(4 * m4[q] + t) * cols + ((((cg[q] >> 2) ^ (m4[q] & 3)) << 4) | ((cg[q] & 3) << 2))] = r[q][t];
                continue;
            }
            const float4 c0 = make_float4(r[q][0].x, r[q][1].x, r[q][2].x, r[q][3].x);
            const float4 c1 = make_float4(r[q][0].y, r[q][1].y, r[q][2].y, r[q][3].y);
            const float4 c2 = make_float4(r[q][0].z, r[q][1].z, r[q][2].z, r[q][3].z);
            const float4 c3 = make_float4(r[q][0].w, r[q][1].w, r[q][2].w, r[q][3].w);
            const int row = 4 * cg[q];
            *(float4*)&base[(row + 0) * BKS + ((m4[q] ^ wswz(row + 0)) << 2)] = c0;
            *(float4*)&base[(row + 1) * BKS + ((m4[q] ^ wswz(row + 1)) << 2)] = c1;
            *(float4*)&base[(row + 2) * BKS + ((m4[q] ^ wswz(row + 2)) << 2)] = c2;
            *(float4*)&base[(row + 3) * BKS + ((m4[q] ^ wswz(row + 3)) << 2)] = c3;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int buf) {
        if constexpr (ROWMAJOR) {
            // the MFMA (s2, t) reduces over the pixels 16 s2 + 4 fg + t (the same sets, in the same order, as the transposed
            // image's float4 columns): a lane reads ONE float per fragment and MFMA, 16 lanes a 64-byte run of one pixel row
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = 16 * s2 + 4 * fg + t;
                    float a[TM], b[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = As[buf][m * BMW + ((((wm * TM + i) ^ fg) << 4) | fr)];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[j] = Bs[buf][m * BNW + ((((wn * TN + j) ^ fg) << 4) | fr)];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            return;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            float4 av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 16 + fr;
                av[i] = *(const float4*)&As[buf][row * BKS + (((s2 * 4 + fg) ^ wswz(row)) << 2)];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = (wn * TN + j) * 16 + fr;
                bv[j] = *(const float4*)&Bs[buf][row * BKS + (((s2 * 4 + fg) ^ wswz(row)) << 2)];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                        const float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
                    }
        }
    };

    // fused SGD: the filter and momentum tiles are fetched NOW, so their HBM latency hides behind the pixel
    // reduction (read in the epilogue loop they cost one exposed round trip per iteration: the stores of one
    // iteration alias the loads of the next as far as the compiler knows)
    constexpr int W_LD = (BMW * (BNW / 4) + THREADS - 1) / THREADS;
    constexpr bool PREFETCH_W = FUSED_SGD && W_LD <= 8;
    float4 pw[PREFETCH_W ? W_LD : 1], pm[PREFETCH_W ? W_LD : 1];
    if (PREFETCH_W && p.sgd_m) {
#pragma unroll
        for (int it = 0; it < W_LD; ++it) {
            const int e = tid + it * THREADS;
            const int row = e / (BNW / 4), col = (e % (BNW / 4)) * 4;
            const int n = n0 + row, k = k0 + col;
            const bool ok = e < BMW * (BNW / 4) && n < p.N && k < p.K;
            const long long o = (long long)n * p.K + k;
            // streamed once: non-temporal, so the filter / momentum tiles do not evict the x and gy slices that the
            // other workgroups of this XCD re-read from L2
            pw[it] = ok ? __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4*)(p.gw + o))) : make_float4(0.f, 0.f, 0.f, 0.f);
            pm[it] = ok ? __builtin_bit_cast(float4, __builtin_nontemporal_load((const f32x4*)(p.sgd_m + o))) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }

    int buf = 0;
    if constexpr (DMA) {
        // A stage = NPA + NPB pieces of 1 KB (RP pixel rows of the gy tile, then of the x tile); wave w requests pieces w, w + 4, ...
        constexpr int NPA = BMW / 8, NPB = BNW / 8, NP = NPA + NPB, PMAX = NP / 4;
        static_assert(NP % 4 == 0, "whole rounds of four pieces");
        constexpr unsigned INV = 0x80000000u;
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        unsigned d_vk[PMAX];
        int d_row[PMAX];
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int pc = q * 4 + wave_s;
            const bool isA = pc < NPA;                                   // wave-uniform
            const int cols = isA ? BMW : BNW, cp = cols / 4, rp = 256 / cols;
            const int mi = (isA ? pc : pc - NPA) * rp + lane / cp, ch = lane % cp;
            const int col = ((((ch >> 2) ^ ((mi >> 2) & 3)) << 4) | ((ch & 3) << 2));
            const int g = (isA ? n0 : k0) + col;
            d_row[q] = mi;
            d_vk[q] = g < (isA ? p.N : p.K) ? (unsigned)(mi * (isA ? p.N : p.Cin) + g) * 4u : INV;
        }
        const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
        auto dma_issue = [&](int ms, int S) {
            const unsigned so_a = (unsigned)ms * (unsigned)p.N * 4u, so_b = (unsigned)ms * (unsigned)p.Cin * 4u;
            const unsigned a_dst = lds_base + (unsigned)(S * BMW * BKS * 4), b_dst = lds_base + (unsigned)((2 * BMW + S * BNW) * BKS * 4);
            const bool tail = ms + BKS > mend;                           // uniform: only a split's last stage can be partial
#pragma unroll
            for (int q = 0; q < PMAX; ++q) {
                const int pc = q * 4 + wave_s;
                const bool isA = pc < NPA;
                const unsigned v = tail ? (d_vk[q] | (ms + d_row[q] < mend ? 0u : INV)) : d_vk[q];
                lds_dma16(isA ? a_dst + (unsigned)(pc * 1024) : b_dst + (unsigned)((pc - NPA) * 1024), v, isA ? gr : xr, isA ? so_a : so_b);
            }
        };
        dma_issue(mbeg, 0);
        for (int ms = mbeg; ms < mend; ms += BKS) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my pieces of this stage have landed ...
            __builtin_amdgcn_s_barrier();                          // ... everyone's have; and everyone is done with the other buffer
            if (ms + BKS < mend) dma_issue(ms + BKS, buf ^ 1);
            compute(buf);
            buf ^= 1;
        }
        __syncthreads();                                           // the epilogue reuses the stage buffers
    } else {
    gload(mbeg);
    sstore(0);
    __syncthreads();
    if constexpr (CLK) c_t1 = __builtin_amdgcn_s_memtime();
    for (int ms = mbeg; ms < mend; ms += BKS) {
        const bool more = ms + BKS < mend;
        if constexpr (CLK) {        // diagnostic instantiation only (tools/wgrad_phase.py ABL=..): 1 = no staging after the first stage, 2 = no MFMAs
            if (more && !(p.abl & 1)) gload(ms + BKS);
            if (!(p.abl & 2)) compute(buf);
            if (more && !(p.abl & 1)) sstore(buf ^ 1);
            if (!(p.abl & 4)) __syncthreads();
            buf ^= (p.abl & 1) ? 0 : 1;
            continue;
        }
        if (more) gload(ms + BKS);
        compute(buf);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    }
    if constexpr (CLK) c_t2 = __builtin_amdgcn_s_memtime();

    // epilogue through LDS: rows = filters n, columns = taps k (contiguous in gw)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = (wm * TM + i) * 16 + 4 * fg + rr;
            float rsc = 1.f;
            if (p.row_scale) rsc = n0 + row < p.N ? p.row_scale[n0 + row] : 0.f;      // uniform branch
#pragma unroll
            for (int j = 0; j < TN; ++j)
                smem[row * CROW + (wn * TN + j) * 16 + fr] = p.row_scale ? acc[i][j][rr] * rsc : acc[i][j][rr];
        }
    __syncthreads();
    if (PREFETCH_W && p.sgd_m) {
#pragma unroll
        for (int it = 0; it < W_LD; ++it) {
            const int e = tid + it * THREADS;
            const int row = e / (BNW / 4), col = (e % (BNW / 4)) * 4;
            const int n = n0 + row, k = k0 + col;
            if (e >= BMW * (BNW / 4) || n >= p.N || k >= p.K) continue;
            const long long o = (long long)n * p.K + k;
            const float4 g = *(const float4*)&smem[row * CROW + col];
            float4 pv = pw[it], mv = pm[it];      // g' = g + wd*p ; m = mom*m + g' ; p -= lr*m   (same order as sgd_momentum_kernel)
            mv.x = p.mom * mv.x + (g.x + p.wd * pv.x); mv.y = p.mom * mv.y + (g.y + p.wd * pv.y);
            mv.z = p.mom * mv.z + (g.z + p.wd * pv.z); mv.w = p.mom * mv.w + (g.w + p.wd * pv.w);
            pv.x -= p.lr * mv.x; pv.y -= p.lr * mv.y; pv.z -= p.lr * mv.z; pv.w -= p.lr * mv.w;
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, mv), (f32x4*)(p.sgd_m + o));
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, pv), (f32x4*)(p.gw + o));
        }
    } else if (p.sgd_m || p.direct) {
        for (int e = tid; e < BMW * (BNW / 4); e += THREADS) {
            const int row = e / (BNW / 4), col = (e % (BNW / 4)) * 4;
            const int n = n0 + row, k = k0 + col;
            if (n >= p.N || k >= p.K) continue;                     // K % 4 == 0
            const long long o = (long long)n * p.K + k;
            float4 g = *(const float4*)&smem[row * CROW + col];
            if (p.sgd_m) {      // g' = g + wd*p ; m = mom*m + g' ; p -= lr*m   (same order as sgd_momentum_kernel)
                float4 pv = *(const float4*)(p.gw + o), mv = *(const float4*)(p.sgd_m + o);
                mv.x = p.mom * mv.x + (g.x + p.wd * pv.x); mv.y = p.mom * mv.y + (g.y + p.wd * pv.y);
                mv.z = p.mom * mv.z + (g.z + p.wd * pv.z); mv.w = p.mom * mv.w + (g.w + p.wd * pv.w);
                pv.x -= p.lr * mv.x; pv.y -= p.lr * mv.y; pv.z -= p.lr * mv.z; pv.w -= p.lr * mv.w;
                *(float4*)(p.sgd_m + o) = mv;
                *(float4*)(p.gw + o) = pv;
            } else {
                *(float4*)(p.gw + o) = g;
            }
        }
    } else if (p.part_ws) {
        // ---- two-pass ordered finish: my partial tile in gw's layout, slot (split, plane); the reduce pass sums the slots in order
        float* dst = p.part_ws + ((long long)by * (p.nbatch > 1 ? p.nbatch : 1) + bz) * ((long long)p.N * p.K);
        for (int e = tid; e < BMW * (BNW / 4); e += THREADS) {
            const int row = e / (BNW / 4), col = (e % (BNW / 4)) * 4;
            const int n = n0 + row, k = k0 + col;
            if (n >= p.N || k >= p.K) continue;                     // K % 4 == 0
            *(float4*)(dst + (long long)n * p.K + k) = *(const float4*)&smem[row * CROW + col];
        }
    } else if (p.ord_ws) {
        // ---- ordered finish: my partial tile to the workspace (sc1: coherent across the XCDs without fences), arrival count,
        // the last workgroup of the tile sums the partials in split order -- four in flight per round -- and writes gw
        constexpr int SC01 = 16;
        const int tile_lin = bz * p.ord_tiles + bx, nsplit = p.ord_splits, my = by;
        const size_t split_stride = (size_t)p.ord_tiles * (p.nbatch > 1 ? p.nbatch : 1) * (BMW * BNW);      // floats between splits
        const __amdgpu_buffer_rsrc_t wsr = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.ord_ws + (size_t)tile_lin * (BMW * BNW)), 0, 0x7FFFFFF0, 0x00020000);
        for (int e = tid; e < BMW * (BNW / 4); e += THREADS) {
            const int row = e / (BNW / 4), col = (e % (BNW / 4)) * 4;
            const float4 v = *(const float4*)&smem[row * CROW + col];
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wsr,
                                                   (unsigned)(my * split_stride * sizeof(float)) + (unsigned)(row * BNW + col) * 4u, 0, SC01);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        __shared__ int ord_last;
        if (tid == 0) {
            const int arrived = __hip_atomic_fetch_add(p.ord_cnt + tile_lin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = arrived == nsplit - 1;
            if (last) __hip_atomic_store(p.ord_cnt + tile_lin, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
            ord_last = last;
        }
        __syncthreads();
        if (ord_last) {
            for (int e = tid; e < BMW * (BNW / 4); e += THREADS) {
                const int row = e / (BNW / 4), col = (e % (BNW / 4)) * 4;
                const int n = n0 + row, k = k0 + col;
                if (n >= p.N || k >= p.K) continue;                     // K % 4 == 0
                const long long o = (long long)n * p.K + k;
                const unsigned off = (unsigned)(row * BNW + col) * 4u;
                const float4 mine4 = *(const float4*)&smem[row * CROW + col];
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.ord_acc) v = *(const float4*)(p.gw + o);
                for (int s0 = 0; s0 < nsplit; s0 += 8) {               // eight parts in flight (round 5: four -- twice the round trips)
                    float4 u[8];
#pragma unroll
                    for (int sp = 0; sp < 8; ++sp)
                        u[sp] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                            wsr, (s0 + sp < nsplit && s0 + sp != my) ? off + (unsigned)((s0 + sp) * split_stride * sizeof(float)) : 0xFFFFFFF0u, 0, SC01));
#pragma unroll
                    for (int sp = 0; sp < 8; ++sp) {                    // slots >= nsplit were read out of range: zeros
                        const float4 t = s0 + sp == my ? mine4 : u[sp];
                        if (s0 == 0 && sp == 0 && !p.ord_acc) v = t;
                        else { v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
                    }
                }
                *(float4*)(p.gw + o) = v;
            }
        }
    } else {
        for (int e = tid; e < BMW * BNW; e += THREADS) {
            const int row = e / BNW, col = e % BNW;
            const int n = n0 + row, k = k0 + col;
            if (n < p.N && k < p.K) atomicAdd(p.gw + (long long)n * p.K + k, smem[row * CROW + col]);
        }
    }
    if constexpr (CLK) {
        __builtin_amdgcn_s_waitcnt(0);
        if (tid == 0 && p.clk) {
            unsigned long long* o = p.clk + 8 * (blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = c_rt0; o[1] = __builtin_amdgcn_s_memrealtime();
            o[2] = c_t1 - c_t0; o[3] = c_t2 - c_t1; o[4] = __builtin_amdgcn_s_memtime() - c_t2;
            o[5] = __builtin_amdgcn_s_getreg(63492); o[6] = __builtin_amdgcn_s_getreg(63508); o[7] = 1;
        }
    }
}

// ---------------------------------------------------------------- small elementwise pieces
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int* __restrict__ arg, int B,
                                    int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)B * Ho * Wo * (C / 4);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (i % (C / 4)) * 4;
        long long t = i / (C / 4);
        const int ox = t % Wo; t /= Wo;
        const int oy = t % Ho;
        const int b = t / Ho;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int4 a = make_int4(-1, -1, -1, -1);
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 + ky;
            if (iy >= H) break;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 + kx;
                if (ix >= W) break;
                float4 v = *(const float4*)(x + (((long long)b * H + iy) * W + ix) * C + c);
                const int id = iy * W + ix;
                if (v.x > m.x) { m.x = v.x; a.x = id; }
                if (v.y > m.y) { m.y = v.y; a.y = id; }
                if (v.z > m.z) { m.z = v.z; a.z = id; }
                if (v.w > m.w) { m.w = v.w; a.w = id; }
            }
        }
        const long long o = (((long long)b * Ho + oy) * Wo + ox) * C + c;
        *(float4*)(y + o) = m;
        if (arg) *(int4*)(arg + o) = a;
    }
}

__global__ void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                    long long n4, long long n, float lr, float mom, float wd) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
         i += (long long)gridDim.x * blockDim.x) {
        float4 pv = ((float4*)p)[i], gv = ((const float4*)g)[i], mv = ((float4*)m)[i];
        mv.x = mom * mv.x + (gv.x + wd * pv.x); mv.y = mom * mv.y + (gv.y + wd * pv.y);
        mv.z = mom * mv.z + (gv.z + wd * pv.z); mv.w = mom * mv.w + (gv.w + wd * pv.w);
        pv.x -= lr * mv.x; pv.y -= lr * mv.y; pv.z -= lr * mv.z; pv.w -= lr * mv.w;
        ((float4*)m)[i] = mv;
        ((float4*)p)[i] = pv;
    }
    // tail (n not a multiple of 4)
    const long long i = n4 * 4 + blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (i < n) {
        float mv = mom * m[i] + (g[i] + wd * p[i]);
        m[i] = mv;
        p[i] -= lr * mv;
    }
}

// Many small tensors in one launch (the per-tensor launch, not the bytes, is what a 300-float bias costs).  The
// table travels by value in the kernel arguments; block b works on the tensor whose block range contains it.
constexpr int SGD_MULTI_MAX = 48;
struct SgdMulti {
    float* p[SGD_MULTI_MAX]; const float* g[SGD_MULTI_MAX]; float* m[SGD_MULTI_MAX];
    long long n[SGD_MULTI_MAX];
    float lr[SGD_MULTI_MAX], wd[SGD_MULTI_MAX];
    int first_block[SGD_MULTI_MAX + 1];
    int count;
    float mom;
};
constexpr int SGD_MULTI_PER_BLOCK = 256 * 16;     // elements per block
__global__ void __launch_bounds__(256) sgd_momentum_multi_kernel(const SgdMulti t) {
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const long long base = (long long)(blockIdx.x - t.first_block[k]) * SGD_MULTI_PER_BLOCK;
    float* p = t.p[k]; const float* g = t.g[k]; float* m = t.m[k];
    const float lr = t.lr[k], wd = t.wd[k], mom = t.mom;
    for (int j = threadIdx.x; j < SGD_MULTI_PER_BLOCK; j += 256) {
        const long long i = base + j;
        if (i >= t.n[k]) break;
        const float mv = mom * m[i] + (g[i] + wd * p[i]);       // same order as sgd_momentum_kernel
        m[i] = mv;
        p[i] -= lr * mv;
    }
}

// torch.optim.Adam (amsgrad off) for up to SGD_MULTI_MAX tensors per launch, in torch's operation order:
//   g' = g + wd p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g' g';  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// The step count t lives in DEVICE memory (adam_step_kernel increments it once per optimizer step), so a captured training step
// replays with the right bias corrections.
struct AdamMulti {
    float* p[SGD_MULTI_MAX]; const float* g[SGD_MULTI_MAX]; float* m[SGD_MULTI_MAX]; float* v[SGD_MULTI_MAX];
    long long n[SGD_MULTI_MAX];
    double lr[SGD_MULTI_MAX];
    float wd[SGD_MULTI_MAX];
    int first_block[SGD_MULTI_MAX + 1];
    int count;
    double b1, b2, eps;          // the betas / eps as the host holds them (Python floats are doubles)
    const int* step;
};
__global__ void adam_step_kernel(int* step) { *step += 1; }
__global__ void __launch_bounds__(256) adam_multi_kernel(const AdamMulti t) {
    int k = 0;
    while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;
    const long long base = (long long)(blockIdx.x - t.first_block[k]) * SGD_MULTI_PER_BLOCK;
    float* p = t.p[k]; const float* g = t.g[k]; float* m = t.m[k]; float* v = t.v[k];
    // torch.optim.Adam (_single_tensor_adam) computes the scalars of a step on the host in DOUBLE: bias_correction = 1 - beta ** step,
    // step_size = lr / bias_correction1, bias_correction2_sqrt = bias_correction2 ** 0.5 -- and hands the tensor kernels their
    // float roundings.  The same here, once per workgroup (round-3 advice: powf on float betas is off by ~3e-5 at small t).
    const double st = (double)*t.step;
    const double bc1 = 1.0 - pow(t.b1, st), bc2 = 1.0 - pow(t.b2, st);
    const float step_size = (float)(t.lr[k] / bc1), bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - t.b1), b2 = (float)t.b2, w2 = (float)(1.0 - t.b2), eps = (float)t.eps, wd = t.wd[k];
    for (int j = threadIdx.x; j < SGD_MULTI_PER_BLOCK; j += 256) {
        const long long i = base + j;
        if (i >= t.n[k]) break;
        const float gp = g[i] + wd * p[i];                    // grad.add(param, alpha=weight_decay)
        const float mv = m[i] + w1 * (gp - m[i]);             // exp_avg.lerp_(grad, 1 - beta1)
        const float vv = b2 * v[i] + w2 * gp * gp;            // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        m[i] = mv;
        v[i] = vv;
        p[i] -= step_size * (mv / (sqrtf(vv) / bc2_sqrt + eps));      // param.addcdiv_(exp_avg, denom, value=-step_size)
    }
}

// Ordered finish of per-workgroup column sums (round 5): ``s`` = this workgroup's sum of four columns n..n+3 over ITS rows.
// With a workspace the sums of one column block (blockIdx.y) meet there: every workgroup stores its row of partials (sc1),
// counts its arrival, and the LAST one adds the gridDim.x partials of every column in block order -- eight in flight per round --
// onto gbias: bit-reproducible where fp32 atomics (the workspace-free form) add in arrival order.  Every thread of the
// workgroup must call it (barriers inside); ``live`` = the thread owns four columns.
// Round 6: TWO LEVELS when there are more than kColsumGroup row blocks (netD_style's 37500-row projections keep their hundreds
// of row blocks -- they must stream at full rate -- and were left on atomics): the blocks of a group of kColsumGroup meet first,
// the group's last arriver adds the group's partials in block order and stores the group sum; the last GROUP to finish adds
// the group sums in group order.  Nobody reads more than kColsumGroup + #groups rows, the order of every addition is fixed by
// the block indices.  Counters: cnt[blockIdx.y * (1 + groups)] for the groups' meeting, + 1 + g for group g; rows of partials:
// part[block] then part2 = part + gridDim.x rows: [group].
constexpr int kColsumGroup = 32;
__device__ inline void colsum_finish4(float4 s, int n, int N, bool live, float* __restrict__ gbias, float* part, int* cnt) {
    if (!part) {
        if (live) {
            atomicAdd(gbias + n, s.x); atomicAdd(gbias + n + 1, s.y);
            atomicAdd(gbias + n + 2, s.z); atomicAdd(gbias + n + 3, s.w);
        }
        return;
    }
    constexpr int SC01 = 16;
    const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc((void*)part, 0, 0x7FFFFFF0, 0x00020000);
    const unsigned rowb = (unsigned)N * 4u, off = (unsigned)n * 4u;
    const int nb = gridDim.x, my = blockIdx.x;
    const int ngroups = (nb + kColsumGroup - 1) / kColsumGroup, grp = my / kColsumGroup;
    const int g0 = grp * kColsumGroup, gn = min(kColsumGroup, nb - g0);          // my group: blocks g0 .. g0 + gn - 1
    int* cbase = cnt + blockIdx.y * (1 + ngroups);
    if (live) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), pr, (unsigned)my * rowb + off, 0, SC01);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    __shared__ int cs_last;
    if (threadIdx.x == 0) {
        int* c = ngroups > 1 ? cbase + 1 + grp : cbase;
        const int arrived = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = arrived == gn - 1;
        if (last) __hip_atomic_store(c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cs_last = last;
    }
    __syncthreads();
    if (!cs_last) return;
    // the group's partials in block order (mine from its register); one level: onto gbias directly, as in round 5
    float4 t = ngroups > 1 ? make_float4(0.f, 0.f, 0.f, 0.f) : (live ? *(const float4*)(gbias + n) : make_float4(0.f, 0.f, 0.f, 0.f));
    if (live) {
        for (int b0 = 0; b0 < gn; b0 += 8) {
            float4 u[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                u[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                    pr, (b0 + k < gn && g0 + b0 + k != my) ? (unsigned)(g0 + b0 + k) * rowb + off : 0xFFFFFFF0u, 0, SC01));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float4 v = g0 + b0 + k == my ? s : u[k];       // slots beyond the group were read out of range: zeros
                t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
            }
        }
    }
    if (ngroups == 1) {
        if (live) *(float4*)(gbias + n) = t;
        return;
    }
    // second level: my group's sum to row (nb + grp); the last group to arrive adds the group sums in group order onto gbias
    if (live) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, t), pr, (unsigned)(nb + grp) * rowb + off, 0, SC01);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int arrived = __hip_atomic_fetch_add(cbase, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = arrived == ngroups - 1;
        if (last) __hip_atomic_store(cbase, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cs_last = last;
    }
    __syncthreads();
    if (!cs_last || !live) return;
    float4 r = *(const float4*)(gbias + n);
    for (int b0 = 0; b0 < ngroups; b0 += 8) {
        float4 u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            u[k] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(
                pr, (b0 + k < ngroups && b0 + k != grp) ? (unsigned)(nb + b0 + k) * rowb + off : 0xFFFFFFF0u, 0, SC01));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float4 v = b0 + k == grp ? t : u[k];
            r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w;
        }
    }
    *(float4*)(gbias + n) = r;
}

// g_pre = gy * (y > 0); g = g_pre * scale[n]; gbias[n] += sum_m g_pre.  One streaming pass: thread = 4 columns
// (float4), a workgroup covers rows_per_blk rows x 1024 columns; either output may be NULL.
__global__ void __launch_bounds__(256)
epilogue_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ scale,
                    float* __restrict__ g, float* __restrict__ gpre, float* __restrict__ gbias, long long M, int N,
                    int relu, int rows_per_blk, float* __restrict__ g_t, float* part, int* cnt) {
    const int n = (blockIdx.y * 256 + threadIdx.x) * 4;
    const bool live = n < N;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = !live ? r0 : (r0 + rows_per_blk < M ? r0 + rows_per_blk : M);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
    if (scale && live) sc = *(const float4*)(scale + n);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (long long r = r0; r < r1; ++r) {
        float4 v = *(const float4*)(gy + r * N + n);
        if (relu) {
            const float4 yy = *(const float4*)(y + r * N + n);
            v.x = yy.x > 0.f ? v.x : 0.f; v.y = yy.y > 0.f ? v.y : 0.f;
            v.z = yy.z > 0.f ? v.z : 0.f; v.w = yy.w > 0.f ? v.w : 0.f;
        }
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        if (gpre) *(float4*)(gpre + r * N + n) = v;
        if (g) *(float4*)(g + r * N + n) = make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w);
        if (g_t) {          // the same gradient column-major, (N x M): what a linear layer's dgrad on the wgrad kernel reads
            g_t[(long long)n * M + r] = v.x * sc.x; g_t[(long long)(n + 1) * M + r] = v.y * sc.y;
            g_t[(long long)(n + 2) * M + r] = v.z * sc.z; g_t[(long long)(n + 3) * M + r] = v.w * sc.w;
        }
    }
    if (gbias) colsum_finish4(s, n, N, live, gbias, part, cnt);
}

// Narrow tensors (N <= 1024 columns, tall M: the conv_lo feature maps of the relation head are 16384 x 96):
// the 256 threads split into N/4 column groups x row lanes, a lane strides over the rows of the block, and the
// column sums are reduced across the lanes in LDS -> ONE atomic per column per workgroup.  (With one thread per
// 4 columns only 24 of 256 threads had work and 512 workgroups hammered the same 96 addresses: 58 us.)
__global__ void __launch_bounds__(256)
epilogue_bwd_narrow_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ scale,
                           float* __restrict__ g, float* __restrict__ gpre, float* __restrict__ gbias, long long M,
                           int N, int relu, int rows_per_blk, float* part, int* cnt) {
    __shared__ float red[256 * 4];
    const int cg = N >> 2;                       // column groups (<= 256)
    const int lanes = 256 / cg;                  // row lanes (>= 1)
    const int c = threadIdx.x % cg, lane = threadIdx.x / cg;
    const int n = c * 4;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < lanes) {
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
        if (scale) sc = *(const float4*)(scale + n);
#pragma unroll 4
        for (long long r = r0 + lane; r < r1; r += lanes) {
            float4 v = *(const float4*)(gy + r * N + n);
            if (relu) {
                const float4 yy = *(const float4*)(y + r * N + n);
                v.x = yy.x > 0.f ? v.x : 0.f; v.y = yy.y > 0.f ? v.y : 0.f;
                v.z = yy.z > 0.f ? v.z : 0.f; v.w = yy.w > 0.f ? v.w : 0.f;
            }
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            if (gpre) *(float4*)(gpre + r * N + n) = v;
            if (g) *(float4*)(g + r * N + n) = make_float4(v.x * sc.x, v.y * sc.y, v.z * sc.z, v.w * sc.w);
        }
    }
    if (!gbias) return;
    *(float4*)&red[threadIdx.x * 4] = s;
    __syncthreads();
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (threadIdx.x < cg) {
        for (int l = 0; l < lanes; ++l) {
            const float4 u = *(const float4*)&red[(l * cg + threadIdx.x) * 4];
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
    }
    colsum_finish4(t, (int)(threadIdx.x % cg) * 4, N, threadIdx.x < cg, gbias, part, cnt);
}

__global__ void __launch_bounds__(256)
epilogue_bwd_scalar_kernel(const float* __restrict__ gy, const float* __restrict__ y, const float* __restrict__ scale,
                           float* __restrict__ g, float* __restrict__ gpre, float* __restrict__ gbias, long long M,
                           int N, int relu, int rows_per_blk, float* __restrict__ g_t, float* __restrict__ part) {
    const int n = blockIdx.y * 256 + threadIdx.x;
    if (n >= N) return;
    const long long r0 = (long long)blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    const float sc = scale ? scale[n] : 1.f;
    float s = 0.f;
    for (long long r = r0; r < r1; ++r) {
        float v = gy[r * N + n];
        if (relu && !(y[r * N + n] > 0.f)) v = 0.f;
        s += v;
        if (gpre) gpre[r * N + n] = v;
        if (g) g[r * N + n] = v * sc;
        if (g_t) g_t[(long long)n * M + r] = v * sc;
    }
    if (gbias && part) part[(long long)blockIdx.x * N + n] = s;       // ordered: the row blocks' sums side by side, added in block order by a reduce pass
    else if (gbias) atomicAdd(gbias + n, s);
}

}  // namespace


// Reductions that were asked to be ordered (I2V_TUNE_SPLIT_ATOMICS == 0) and fell back to fp32 atomics because the caller's
// workspace was absent or too small (round-5 advice: the fallback was silent).  i2v_ordered_fallbacks() reads / resets it.
extern "C" int32_t i2v_ordered_fallbacks(int32_t reset) {
    const int n = g_ordered_fallbacks;
    if (reset) g_ordered_fallbacks = 0;
    return n;
}

namespace {
// second pass of the two-pass ordered filter gradient: gw[plane][i] = (acc ? gw : 0) + part[0][plane][i] + part[1][plane][i] + ...
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, int splits, int planes, long long nk4, long long bsw, int acc) {
    const long long total = (long long)planes * nk4;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += 256ll * gridDim.x) {
        const long long plane = e / nk4, i = e - plane * nk4;
        float4* o = (float4*)(gw + plane * bsw) + i;
        float4 v = acc ? *o : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4* src = (const float4*)part + plane * nk4 + i;
        for (int s0 = 0; s0 < splits; s0 += 8) {
            float4 u[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) u[k] = s0 + k < splits ? src[(long long)(s0 + k) * planes * nk4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s0 + k < splits) { v.x += u[k].x; v.y += u[k].y; v.z += u[k].z; v.w += u[k].w; }
        }
        *o = v;
    }
}
__global__ void __launch_bounds__(256)
wgrad_reduce_scalar_kernel(const float* __restrict__ part, float* __restrict__ gw, int splits, long long nk, int acc) {
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < nk; e += 256ll * gridDim.x) {
        float v = acc ? gw[e] : 0.f;
        for (int s0 = 0; s0 < splits; s0 += 16) {           // sixteen loads in flight, added in split order
            float u[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) u[k] = s0 + k < splits ? part[(long long)(s0 + k) * nk + e] : 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (s0 + k < splits) v += u[k];
        }
        gw[e] = v;
    }
}
}  // namespace

// picks the kernel + pixel split for one wgrad problem; returns false when v2 cannot be used
constexpr int kWgradOrderedMax = 16;     // most splits the ordered finish of a filter gradient sums (one workgroup reads them all)

// the pixel split launch_wgrad gives an unfused problem of `planes` x (N x K) filters over M reduction rows
static int wgrad_split_count(long long M, int N, int K, int planes, int tm, int tk, int rs) {
    const long long tiles = (long long)i2v_cdiv(N, tm) * i2v_cdiv(K, tk), all_tiles = tiles * planes;
    const int msteps = i2v_cdiv(M, rs), per_cu = g_i2v_tuning[I2V_TUNE_WGRAD_PER_CU];
    int splits = (int)((long long)per_cu * NUM_CU / all_tiles);
    if (splits < 2) splits = (int)(((long long)per_cu * NUM_CU + all_tiles - 1) / all_tiles);
    if (splits > msteps / 4) splits = msteps / 4;
    if (splits < 1) splits = 1;
    return splits;
}

// (csrc/winograd.hip) the parts a 36-plane Winograd filter gradient is split into when its sum is ordered: what its workspace holds
int i2v_internal_wgrad_plane_splits(long long T, int Cout, int Cin) {
    const int s = wgrad_split_count(T, Cout, Cin, 36, 64, 64, BKS);
    const int mps = i2v_cdiv(i2v_cdiv(T, BKS), s) * BKS;
    return i2v_cdiv(T, mps);
}

static bool launch_wgrad(WgP& p, float beta, bool fused, hipStream_t st, void* split_ws = nullptr, size_t split_ws_bytes = 0) {
    const long long xb = (long long)p.B * p.H * p.W * p.Cin * 4, gb = (long long)p.M * p.N * 4;
    const bool v2 = (p.N % 4 == 0) && xb < (1ll << 31) && gb < (1ll << 31) && g_wgrad_v2;
    if (p.row_scale && !v2) { i2v_set_error("conv_wgrad_scaled: shape outside the v2 kernel (Cout % 4, 2 GiB operands)"); return false; }
    // bigger tiles raise the FLOP per staged byte (the reduction dim is streamed): 128x128 = 32 FLOP/B vs 16
    int tm = 64, tk = 64;
    // fused update: 128 filters x 64 taps -- the x tile is shared by twice the filters and half as many workgroups go
    // through the dispatcher.  Alone the kernel is slower than the 64x64 form (fc6: 792 vs 736 us), inside the step it is
    // faster (4.93 vs 5.00 ms, four alternating pairs): the rest of the step gets the chip back sooner
    if (v2 && fused && g_wgrad_fused_tile == 128 && p.N >= 128) tm = 128;
    if (v2 && g_wgrad_v2 >= 2) {
        if (p.N >= 128) tm = 128;
        if (p.K >= 128 && tm == 128 && g_wgrad_v2 == 2) tk = 128;
    }
    const long long tiles = (long long)i2v_cdiv(p.N, tm) * i2v_cdiv(p.K, tk);
    const int rs = v2 ? BKS : 16;
    int splits = 1;
    const int msteps = i2v_cdiv(p.M, rs);
    // one round of workgroups: floor, not ceil (144 tiles x 8 splits = 1152 workgroups on 1024 slots ran 1.5 rounds)
    if (!fused) splits = wgrad_split_count(p.M, p.N, p.K, p.nbatch > 1 ? p.nbatch : 1, tm, tk, rs);
    // Ordered finish (round 5; a caller that passes its split workspace): the split of a SMALL problem (under I2V_TUNE_WGRAD_ORDERED_GFLOP = 8 GFLOP: the
    // relation head's conv_lo filters, its linear layers' data gradients), capped at kWgradOrderedMax parts, is summed in split
    // order by the tile's last workgroup instead of with atomics -- bit-reproducible, and no clear of gw.  A large one (the
    // instance_styleD backbone: up to 254 splits to fill the chip) keeps the atomics.
    const int planes = p.nbatch > 1 ? p.nbatch : 1;
    // Round 6: every split a caller wants ordered IS ordered, whatever its size.  Up to kWgradOrderedMax parts of the
    // second-generation kernel meet in the workspace as tiles and the tile's last workgroup sums them (round 5); more parts --
    // the instance_styleD backbone splits up to 254 ways to fill the chip -- and the first-generation kernel (Cout % 4 != 0)
    // store their partial FILTERS side by side and a reduce pass adds them in split order (part_ws; round 5 left these on
    // atomics, and ran the first-generation kernel UNSPLIT when order was asked for: the RPN's 18-row cls_score gradient over
    // 9576 pixels on eight workgroups, 300 us instead of 9 -- that alone was the "+4 %" ordered sums cost configs[2]).
    const bool ext_part = p.part_ws != nullptr;       // the caller reduces (the Winograd filter gradient's final transform): its own slab
    const double ord_flops = 1e9 * g_i2v_tuning[I2V_TUNE_WGRAD_ORDERED_GFLOP], flops = 2.0 * p.M * p.N * p.K * planes;
    const bool want_ord = !fused && !ext_part && splits > 1 && g_i2v_tuning[I2V_TUNE_SPLIT_ATOMICS] == 0 && flops < ord_flops;
    // a split beyond kWgradOrderedMax parts is capped where that costs nothing (under 1 GFLOP: conv_lo.0's 128-way split of a
    // 0.3 GFLOP problem): the in-kernel finish needs no second launch
    if (want_ord && split_ws && v2 && splits > kWgradOrderedMax && flops < 1e9) splits = kWgradOrderedMax;
    if (ext_part && splits > p.part_cap) splits = p.part_cap > 0 ? p.part_cap : 1;
    p.m_per_split = i2v_cdiv(msteps, splits) * rs;
    splits = i2v_cdiv(p.M, p.m_per_split);
    p.direct = (splits == 1 && beta == 0.f) || fused;
    bool two_pass = false;
    if (ext_part) {
        if (splits == 1) p.part_ws = nullptr;         // one part: written straight to gw (the caller passed its slot 0 as gw)
    } else if (want_ord && splits > 1) {
        const size_t need_ord = kSplitCounterBytes + (size_t)splits * tiles * planes * (size_t)(tm * tk) * sizeof(float);
        const size_t need_part = kSplitCounterBytes + (size_t)splits * planes * (size_t)p.N * p.K * sizeof(float);
        if (split_ws && v2 && splits <= kWgradOrderedMax && tiles * planes <= kSplitCounters && need_ord <= split_ws_bytes && need_ord < (1ull << 31)) {
            p.ord_cnt = reinterpret_cast<int*>(split_ws);
            p.ord_ws = reinterpret_cast<float*>(static_cast<char*>(split_ws) + kSplitCounterBytes);
            p.ord_splits = splits; p.ord_tiles = (int)tiles; p.ord_acc = beta != 0.f;
        } else if (split_ws && need_part <= split_ws_bytes) {
            p.part_ws = reinterpret_cast<float*>(static_cast<char*>(split_ws) + kSplitCounterBytes);      // the counters in front stay zero
            two_pass = true;
        } else {
            ++g_ordered_fallbacks;                    // no workspace, or too small: fp32 atomics (i2v_ordered_fallbacks() tells)
        }
    }
    if (beta == 0.f && !p.direct && !p.ord_ws && !p.part_ws)
        hipMemsetAsync(p.gw, 0, (p.nbatch > 1 ? (size_t)(p.nbatch - 1) * p.bsw : 0) * sizeof(float) + (size_t)p.N * p.K * sizeof(float), st);
    p.x_bytes = (unsigned)xb;
    p.gy_bytes = (unsigned)gb;
    dim3 grid((unsigned)tiles, splits, p.nbatch > 1 ? p.nbatch : 1);
    const long long groups = (long long)splits * (p.nbatch > 1 ? p.nbatch : 1), total = tiles * groups;
    // v2 kernels only (the remap lives there); one group needs no grouping; the 1-D launch must fit an int
    p.xcd_remap = (v2 && groups >= 2 && total < (1ll << 30) && g_i2v_tuning[I2V_TUNE_WGRAD_XCD]) ? 1 : 0;
    if (p.xcd_remap) {
        p.r_tiles = (int)tiles; p.r_splits = splits; p.r_total = (int)total;
        grid = dim3((unsigned)((total + 7) / 8 * 8), 1, 1);
    }
    // round 6: LDS-DMA staging for the pointwise / linear problems (most of a backbone's filter-gradient time: the 1x1 layers and
    // the Winograd-domain plane GEMMs)
    const bool lin = p.KH == 1 && p.KW == 1 && p.pad == 0 && p.stride == 1;
    const bool dma = v2 && !fused && !g_clk && lin && (p.K % 4 == 0) && g_i2v_tuning[I2V_TUNE_WGRAD_DMA];
    if (!v2) conv_wgrad_f32<64, 64><<<grid, THREADS, 0, st>>>(p);
    else if (dma && tm == 128 && tk == 128) conv_wgrad2_f32<4, 4, false, false, true><<<grid, THREADS, 0, st>>>(p);
    else if (dma && tm == 128) conv_wgrad2_f32<4, 2, false, false, true><<<grid, THREADS, 0, st>>>(p);
    else if (dma) conv_wgrad2_f32<2, 2, false, false, true><<<grid, THREADS, 0, st>>>(p);
    else if (fused && tm == 128 && tk == 64) conv_wgrad2_f32<4, 2, true><<<grid, THREADS, 0, st>>>(p);
    else if (tm == 128 && tk == 128) conv_wgrad2_f32<4, 4><<<grid, THREADS, 0, st>>>(p);
    else if (tm == 128) conv_wgrad2_f32<4, 2><<<grid, THREADS, 0, st>>>(p);
    else if (fused) conv_wgrad2_f32<2, 2, true><<<grid, THREADS, 0, st>>>(p);
    else if (g_clk) { p.clk = g_clk; p.abl = g_ablate; conv_wgrad2_f32<2, 2, false, true><<<grid, THREADS, 0, st>>>(p); }
    else conv_wgrad2_f32<2, 2><<<grid, THREADS, 0, st>>>(p);
    if (ext_part) p.ord_splits = splits;              // what the caller's reduce pass must sum
    if (two_pass) {
        const long long nk = (long long)p.N * p.K;
        if ((nk & 3) == 0) {
            const long long total = (long long)planes * (nk / 4);
            wgrad_reduce_kernel<<<(unsigned)std::min<long long>(i2v_cdiv(total, 256), 2048), 256, 0, st>>>(
                p.part_ws, p.gw, splits, planes, nk / 4, planes > 1 ? p.bsw : nk, beta != 0.f);
        } else {
            wgrad_reduce_scalar_kernel<<<(unsigned)std::min<long long>(i2v_cdiv(nk, 256), 2048), 256, 0, st>>>(p.part_ws, p.gw, splits, nk, beta != 0.f);
        }
    }
    return true;
}

static int check_conv(const char* who, const void* a, const void* b, const void* c, int B, int H, int W, int Cin,
                      int Cout, int KH, int KW, int stride, int pad) {
    if (!a || !b || !c) { i2v_set_error("%s: null pointer", who); return I2V_ERR_ARG; }
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0) {
        i2v_set_error("%s: bad shape", who); return I2V_ERR_ARG;
    }
    if (Cin % 4) { i2v_set_error("%s: Cin must be a multiple of 4 (pad the stem input to 4 channels)", who); return I2V_ERR_ARG; }
    if ((H + 2 * pad - KH) < 0 || (W + 2 * pad - KW) < 0) { i2v_set_error("%s: kernel larger than input", who); return I2V_ERR_ARG; }
    return I2V_OK;
}

extern "C" int32_t i2v_conv_debug_clock(void* buf) {
    g_clk = (unsigned long long*)buf;      // device buffer of 2 u64 per workgroup, or NULL to switch off
    return I2V_OK;
}

namespace {
__global__ void clock_stamp_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memtime(); out[1] = __builtin_amdgcn_s_memrealtime(); }
}
}  // namespace

extern "C" int32_t i2v_debug_clock_stamp(void* out2, void* stream) {
    clock_stamp_kernel<<<1, 64, 0, (hipStream_t)stream>>>((unsigned long long*)out2);
    I2V_CHECK_LAUNCH("i2v_debug_clock_stamp");
    return I2V_OK;
}

extern "C" int32_t i2v_conv_set_tile(int32_t cfg) {
    if (cfg < 0) { g_force_tile = -1; g_ablate = 0; return I2V_OK; }
    if (cfg >> 8) {         // rounds 1-5: bits 8-9 selected the 8-wave loader / MFMA specialisation, bits 10+ ablation instantiations
        i2v_set_error("conv_set_tile: bits above the tile index selected experiment kernels that left the library in round 6");
        return I2V_ERR_UNSUPPORTED;
    }
    g_force_tile = (cfg & 0xFF) == 0xFF ? -1 : (cfg & 0xFF);
    return I2V_OK;
}

static int plan_conv(int dry, size_t ws_bytes, int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH,
                     int32_t KW, int32_t stride, int32_t pad) {
    int dummy = 0;
    int rc = check_conv("conv_fwd_splits", &dummy, &dummy, &dummy, B, H, W, Cin, Cout, KH, KW, stride, pad);
    if (rc) return rc;
    ConvP p = {};
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pad_x = pad;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.ostride = 1; p.force_tile = g_force_tile; p.dry = dry;
    return run_conv(p, nullptr, ws_bytes ? &dummy : nullptr, ws_bytes);
}

// 1 if i2v_conv_fwd, given a split-K workspace of ws_bytes, will accumulate split-K partials with atomics for this
// shape (its output must then start at zero: the call clears it itself unless the caller passes I2V_EPI_ZEROED),
// 0 otherwise, < 0 on error.
extern "C" int32_t i2v_conv_fwd_splits(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH,
                                       int32_t KW, int32_t stride, int32_t pad, size_t ws_bytes) {
    const int rc = plan_conv(1, ws_bytes, B, H, W, Cin, Cout, KH, KW, stride, pad);
    return rc < 0 ? rc : (rc > 1 ? 1 : 0);
}

// Bytes of split-K workspace i2v_conv_fwd would use for this shape (0: it does not split K, or it splits into so many
// parts / so small an output that fp32 atomics are the better finish).
extern "C" size_t i2v_conv_split_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH,
                                                 int32_t KW, int32_t stride, int32_t pad) {
    const int rc = plan_conv(2, 0, B, H, W, Cin, Cout, KH, KW, stride, pad);
    return rc > 0 ? (size_t)rc : 0;
}

extern "C" int32_t i2v_conv_fwd(const float* x, const float* w, const float* scale, const float* shift,
                                const float* res, float* y, int32_t B, int32_t H, int32_t W, int32_t Cin,
                                int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad, int32_t flags,
                                void* split_ws, size_t split_ws_bytes, void* stream) {
    int rc = check_conv("conv_fwd", x, w, y, B, H, W, Cin, Cout, KH, KW, stride, pad);
    if (rc) return rc;
    I2V_CHECK_ARG(!(flags & I2V_EPI_SCALE) || (scale && shift), "conv_fwd: EPI_SCALE needs scale and shift");
    I2V_CHECK_ARG(!(flags & I2V_EPI_BIAS) || shift, "conv_fwd: EPI_BIAS needs shift");
    I2V_CHECK_ARG(!(flags & I2V_EPI_RESIDUAL) || res, "conv_fwd: EPI_RESIDUAL needs res");
    ConvP p = {};
    p.x = x; p.w = w; p.scale = scale; p.shift = shift; p.res = res; p.y = y;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad; p.pad_x = pad;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.flags = flags; p.ostride = 1; p.Hy = p.Ho; p.Wy = p.Wo;
    p.force_tile = g_force_tile;
    rc = run_conv(p, (hipStream_t)stream, split_ws, split_ws_bytes);
    if (rc) return rc;
    I2V_CHECK_LAUNCH("conv_fwd");
    return I2V_OK;
}

// nbatch independent GEMMs of one shape in ONE launch: C_z (M x N) = A_z (M x K) * B_z (N x K)^T, fp32, operand z at
// base + z * stride (elements).  The 16 element-wise planes of a Winograd convolution are such a batch.
extern "C" int32_t i2v_gemm_nt_batched(const float* a, const float* b, float* c, int32_t M, int32_t N, int32_t K,
                                       int32_t nbatch, int64_t stride_a, int64_t stride_b, int64_t stride_c,
                                       void* split_ws, size_t split_ws_bytes, void* stream) {
    I2V_CHECK_ARG(a && b && c && M > 0 && N > 0 && K > 0 && nbatch > 0 && nbatch <= 65535, "gemm_nt_batched: bad argument");
    I2V_CHECK_ARG(K % 4 == 0, "gemm_nt_batched: K must be a multiple of 4");
    ConvP p = {};
    p.x = a; p.w = b; p.y = c;
    p.B = 1; p.H = M; p.W = 1; p.Cin = K; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.pad_x = 0;
    p.Ho = M; p.Wo = 1; p.flags = 0; p.ostride = 1; p.Hy = M; p.Wy = 1;
    p.force_tile = g_force_tile;
    p.nbatch = nbatch; p.bsx = stride_a; p.bsw = stride_b; p.bsy = stride_c;
    int rc = run_conv(p, (hipStream_t)stream, split_ws, split_ws_bytes);
    if (rc) return rc;
    I2V_CHECK_LAUNCH("gemm_nt_batched");
    return I2V_OK;
}

extern "C" size_t i2v_conv_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout,
                                                 int32_t KH, int32_t KW, int32_t stride, int32_t pad) {
    (void)B; (void)H; (void)W; (void)stride; (void)pad;
    (void)Cin; (void)Cout; (void)KH; (void)KW;
    return 256;   // reserved (the split-m reduction uses atomics directly into gw)
}

extern "C" size_t i2v_conv_dgrad_workspace_bytes(int32_t Cin, int32_t Cout, int32_t KH, int32_t KW) {
    return i2v_align((size_t)Cout * KH * KW * Cin * sizeof(float));
}

// dgrad = forward conv of gy with the flipped/transposed filter (staged in the workspace).
//
// stride s > 1, KxK filter: an input pixel iy = s*a + r only receives taps ky = (r + pad) % s + s*t, from
// output rows a + c0 - t with c0 = (r + pad - ky0) / s.  So every parity class (ry, rx) is its own stride-1
// correlation of gy with a flipped sub-filter, written to every s-th pixel of gx: s*s launches whose MACs add up
// to exactly the dense count (the zero-insertion form did s*s times that).
static int conv_dgrad_impl(const float* gy, const float* w, const float* gy_scale, const float* out_scale,
                           const float* res, const float* mask, float* gx, int32_t B, int32_t H, int32_t W,
                           int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                           void* ws, size_t ws_bytes, void* split_ws, size_t split_ws_bytes, void* stream) {
    int rc = check_conv("conv_dgrad", gy, w, gx, B, H, W, Cin, Cout, KH, KW, stride, pad);
    if (rc) return rc;
    const bool fused = gy_scale || out_scale || res || mask;
    if (fused && stride != 1 && !(KH == 1 && KW == 1 && pad == 0)) {
        i2v_set_error("conv_dgrad_fused: stride 1, or a strided 1x1 layer (other strided layers take i2v_conv_dgrad and separate passes)");
        return I2V_ERR_UNSUPPORTED;
    }
    I2V_CHECK_ARG(Cout % 4 == 0, "conv_dgrad: Cout must be a multiple of 4");
    if (!ws || ws_bytes < (size_t)Cout * KH * KW * Cin * sizeof(float)) {
        i2v_set_error("conv_dgrad: workspace too small");
        return I2V_ERR_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    ConvP p = {};
    p.x = gy; p.y = gx;
    p.B = B; p.H = Ho; p.W = Wo; p.Cin = Cout; p.Cout = Cin; p.stride = 1;
    p.flags = 0;
    if (out_scale) { p.scale = out_scale; p.shift = nullptr; p.flags |= I2V_EPI_SCALE; }
    if (res) { p.res = res; p.flags |= I2V_EPI_RESIDUAL; }
    if (mask) { p.mask = mask; p.flags |= I2V_EPI_MASK; }
    p.force_tile = g_force_tile;
    p.Hy = H; p.Wy = W;
    const bool pointwise = KH == 1 && KW == 1 && pad == 0;
    if (stride == 1 || pointwise) {
        const long long wn = (long long)Cout * KH * KW * Cin;
        weight_dgrad_layout<<<(int)fmin((double)i2v_cdiv(wn, 256), 4096.0), 256, 0, st>>>(w, (float*)ws, Cout, KH, KW, Cin, gy_scale);
        p.w = (const float*)ws;
        p.KH = KH; p.KW = KW;
        if (stride == 1) {
            I2V_CHECK_ARG(KH - 1 - pad >= 0 && KW - 1 - pad >= 0, "conv_dgrad: padding larger than the filter");
            p.pad = KH - 1 - pad; p.pad_x = KW - 1 - pad;
            p.Ho = H; p.Wo = W; p.ostride = 1;
        } else {                    // strided 1x1: every s-th pixel gets a value (epilogue operands are read at that pixel), the
            p.pad = 0; p.pad_x = 0; p.Ho = Ho; p.Wo = Wo; p.ostride = stride;       // rest are zero -- res must be zero there too
            hipMemsetAsync(gx, 0, (size_t)B * H * W * Cin * sizeof(float), st);
        }
        rc = run_conv(p, st, split_ws, split_ws_bytes);
        if (rc) return rc;
        I2V_CHECK_LAUNCH("conv_dgrad");
        return I2V_OK;
    }
    // parity decomposition
    bool all_written = true;
    for (int r = 0; r < stride; ++r) {
        if ((r + pad) % stride >= KH || (r + pad) % stride >= KW) all_written = false;
    }
    // pixels beyond the last window (iy + pad - ky > s*(Ho-1) for every tap) still get zeros from the masked taps
    if (!all_written) hipMemsetAsync(gx, 0, (size_t)B * H * W * Cin * sizeof(float), st);
    float* wsub = (float*)ws;
    for (int ry = 0; ry < stride && ry < H; ++ry) {
        const int ky0 = (ry + pad) % stride;
        if (ky0 >= KH) continue;
        const int Ty = (KH - ky0 + stride - 1) / stride, c0y = (ry + pad - ky0) / stride;
        for (int rx = 0; rx < stride && rx < W; ++rx) {
            const int kx0 = (rx + pad) % stride;
            if (kx0 >= KW) continue;
            const int Tx = (KW - kx0 + stride - 1) / stride, c0x = (rx + pad - kx0) / stride;
            const long long wn = (long long)Cin * Ty * Tx * Cout;
            weight_dgrad_sub_layout<<<(int)fmin((double)i2v_cdiv(wn, 256), 4096.0), 256, 0, st>>>(
                w, wsub, Cout, KH, KW, Cin, stride, ky0, kx0, Ty, Tx);
            ConvP q = p;
            q.w = wsub;
            q.KH = Ty; q.KW = Tx;
            q.pad = Ty - 1 - c0y; q.pad_x = Tx - 1 - c0x;          // may be negative: the window starts inside gy
            q.Ho = (H - ry + stride - 1) / stride;                    // pixels of this parity class
            q.Wo = (W - rx + stride - 1) / stride;
            q.ostride = stride;
            q.y = gx + ((long long)ry * W + rx) * Cin;
            rc = run_conv(q, st, split_ws, split_ws_bytes);
            if (rc) return rc;
            wsub += wn;
        }
    }
    I2V_CHECK_LAUNCH("conv_dgrad");
    return I2V_OK;
}

extern "C" int32_t i2v_conv_dgrad(const float* gy, const float* w, float* gx, int32_t B, int32_t H, int32_t W,
                                  int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                  void* ws, size_t ws_bytes, void* split_ws, size_t split_ws_bytes, void* stream) {
    return conv_dgrad_impl(gy, w, nullptr, nullptr, nullptr, nullptr, gx, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes,
                           split_ws, split_ws_bytes, stream);
}

extern "C" int32_t i2v_conv_dgrad_fused(const float* gy, const float* w, const float* gy_scale, const float* out_scale,
                                        const float* res, const float* mask, float* gx, int32_t B, int32_t H, int32_t W,
                                        int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                        void* ws, size_t ws_bytes, void* split_ws, size_t split_ws_bytes, void* stream) {
    return conv_dgrad_impl(gy, w, gy_scale, out_scale, res, mask, gx, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes,
                           split_ws, split_ws_bytes, stream);
}

static int conv_wgrad_impl(const float* x, const float* gy, float* gw, const float* row_scale, int32_t B, int32_t H,
                           int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                           float beta, void* stream, void* split_ws = nullptr, size_t split_ws_bytes = 0) {
    int rc = check_conv("conv_wgrad", x, gy, gw, B, H, W, Cin, Cout, KH, KW, stride, pad);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    WgP p = {};
    p.x = x; p.gy = gy; p.gw = gw; p.row_scale = row_scale;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = KH * KW * Cin;
    p.lgCin = ilog2_exact(Cin);
    I2V_CHECK_ARG(beta == 0.f || beta == 1.f, "conv_wgrad: beta must be 0 or 1");
    if (!launch_wgrad(p, beta, false, st, split_ws, split_ws_bytes)) return I2V_ERR_UNSUPPORTED;
    I2V_CHECK_LAUNCH("conv_wgrad");
    return I2V_OK;
}

// gw[z] (N x K) = gy[z]^T (M x N) . x[z] (M x K) for z < nbatch: the element-wise planes of a Winograd filter gradient
// (csrc/winograd.hip).  gw is overwritten; the batches of gw must be contiguous when the reduction is split (one clear).
// beta is an explicit argument of the shared implementation (round-3 advice: the accumulating entry point used to pass it
// through a thread_local global, where an early return could have left it at 1).
static int32_t gemm_tn_batched_impl(const float* x, const float* gy, float* gw, int32_t M, int32_t N, int32_t K, int32_t nbatch,
                                    long long stride_x, long long stride_gy, long long stride_gw, float beta, void* stream,
                                    int part_cap = 0, int* part_splits = nullptr) {
    I2V_CHECK_ARG(x && gy && gw && M > 0 && N > 0 && K > 0 && nbatch > 0, "gemm_tn_batched: bad argument");
    I2V_CHECK_ARG(N % 4 == 0 && K % 4 == 0, "gemm_tn_batched: N and K must be multiples of 4");
    I2V_CHECK_ARG(nbatch == 1 || stride_gw == (long long)N * K, "gemm_tn_batched: gw batches must be contiguous");
    WgP p = {};
    p.x = x; p.gy = gy; p.gw = gw;
    p.B = 1; p.H = 1; p.W = M; p.Cin = K; p.Cout = N; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0;
    p.Ho = 1; p.Wo = M;
    p.M = M; p.N = N; p.K = K;
    p.lgCin = ilog2_exact(K);
    p.nbatch = nbatch; p.bsx = stride_x; p.bsg = stride_gy; p.bsw = stride_gw;
    if (!g_wgrad_v2 || (long long)M * K * 4 >= (1ll << 31) || (long long)M * N * 4 >= (1ll << 31)) {
        i2v_set_error("gemm_tn_batched: operand larger than 2 GiB per batch");
        return I2V_ERR_UNSUPPORTED;
    }
    if (part_splits) {          // i2v_internal_gemm_tn_batched_parts: gw = a slab of part_cap slots of nbatch x (N x K); slot s = split s
        p.part_ws = gw; p.part_cap = part_cap;
    }
    launch_wgrad(p, beta, false, (hipStream_t)stream);
    if (part_splits) *part_splits = p.part_ws ? p.ord_splits : 1;
    I2V_CHECK_LAUNCH("gemm_tn_batched");
    return I2V_OK;
}

// (csrc/winograd.hip) parts[0] = parts[0] + parts[1] + ... in part order, all planes in parallel (in place: an element is read and
// written by one thread)
int32_t i2v_internal_reduce_parts(float* parts, int nparts, int planes, long long nk, void* stream) {
    const long long total = (long long)planes * (nk / 4);
    wgrad_reduce_kernel<<<(unsigned)std::min<long long>(i2v_cdiv(total, 256), 4096), 256, 0, (hipStream_t)stream>>>(
        parts, parts, nparts, planes, nk / 4, nk, 0);
    return I2V_OK;
}

// (csrc/winograd.hip) the same GEMMs with the split parts left SIDE BY SIDE in ``parts`` ([split][plane][N x K], room for
// ``cap`` splits) for the caller's own ordered sum; *splits = how many were written (1: the result itself)
int32_t i2v_internal_gemm_tn_batched_parts(const float* x, const float* gy, float* parts, int32_t M, int32_t N, int32_t K,
                                           int32_t nbatch, long long stride_x, long long stride_gy, int cap, int* splits, void* stream) {
    return gemm_tn_batched_impl(x, gy, parts, M, N, K, nbatch, stride_x, stride_gy, (long long)N * K, 0.f, stream, cap, splits);
}

extern "C" int32_t i2v_gemm_tn_batched(const float* x, const float* gy, float* gw, int32_t M, int32_t N, int32_t K,
                                       int32_t nbatch, long long stride_x, long long stride_gy, long long stride_gw,
                                       void* stream) {
    return gemm_tn_batched_impl(x, gy, gw, M, N, K, nbatch, stride_x, stride_gy, stride_gw, 0.f, stream);
}

// The same accumulating into gw (gw += sum; the caller has cleared or pre-loaded it): no memset node in front.
extern "C" int32_t i2v_gemm_tn_batched_acc(const float* x, const float* gy, float* gw, int32_t M, int32_t N, int32_t K,
                                           int32_t nbatch, long long stride_x, long long stride_gy, long long stride_gw,
                                           void* stream) {
    return gemm_tn_batched_impl(x, gy, gw, M, N, K, nbatch, stride_x, stride_gy, stride_gw, 1.f, stream);
}

extern "C" int32_t i2v_conv_wgrad(const float* x, const float* gy, float* gw, int32_t B, int32_t H, int32_t W,
                                  int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride, int32_t pad,
                                  float beta, void* ws, size_t ws_bytes, void* stream) {
    // ws: the caller's split workspace (i2v_conv_fwd's: zeroed counters + slab; NULL: splits are summed with fp32 atomics)
    return conv_wgrad_impl(x, gy, gw, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, beta, stream, ws, ws_bytes);
}

extern "C" int32_t i2v_conv_wgrad_scaled(const float* x, const float* gy, const float* row_scale, float* gw, int32_t B,
                                         int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW,
                                         int32_t stride, int32_t pad, float beta, void* ws, size_t ws_bytes, void* stream) {
    // ws: as i2v_conv_wgrad's (round 6: the trained bottlenecks' 1x1 filter gradients are ordered through it too)
    return conv_wgrad_impl(x, gy, gw, row_scale, B, H, W, Cin, Cout, KH, KW, stride, pad, beta, stream, ws, ws_bytes);
}

// wgrad with the SGD(momentum) update of that filter fused into the accumulator epilogue: the gradient
// never goes to HBM (for vrd.fc6 that is 822 MB written + 822 MB read back per step).  Only when the whole
// reduction over the pixels fits one workgroup pass (no split over m), i.e. the skinny relation-head GEMMs.
extern "C" int32_t i2v_conv_wgrad_sgd(const float* x, const float* gy, float* w, float* m, int32_t B, int32_t H,
                                      int32_t W, int32_t Cin, int32_t Cout, int32_t KH, int32_t KW, int32_t stride,
                                      int32_t pad, float lr, float momentum, float weight_decay, void* stream) {
    int rc = check_conv("conv_wgrad_sgd", x, gy, w, B, H, W, Cin, Cout, KH, KW, stride, pad);
    if (rc) return rc;
    I2V_CHECK_ARG(m, "conv_wgrad_sgd: null momentum buffer");
    WgP p = {};
    p.x = x; p.gy = gy; p.gw = w; p.sgd_m = m; p.lr = lr; p.mom = momentum; p.wd = weight_decay;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
    p.Ho = (H + 2 * pad - KH) / stride + 1;
    p.Wo = (W + 2 * pad - KW) / stride + 1;
    p.M = B * p.Ho * p.Wo; p.N = Cout; p.K = KH * KW * Cin;
    p.lgCin = ilog2_exact(Cin);
    const long long tiles = (long long)i2v_cdiv(p.N, 64) * i2v_cdiv(p.K, 64);
    if (tiles < 2 * NUM_CU || p.M > 4096) {
        i2v_set_error("conv_wgrad_sgd: shape needs a split over pixels; use i2v_conv_wgrad + i2v_sgd_momentum");
        return I2V_ERR_UNSUPPORTED;
    }
    launch_wgrad(p, 0.f, true, (hipStream_t)stream);
    I2V_CHECK_LAUNCH("conv_wgrad_sgd");
    return I2V_OK;
}

extern "C" int32_t i2v_epilogue_bwd(const float* gy, const float* y, const float* scale, float* g, float* gpre,
                                    float* gbias, int64_t M, int32_t N, int32_t relu, float* g_t, void* split_ws,
                                    size_t split_ws_bytes, void* stream) {
    I2V_CHECK_ARG(gy && M >= 0 && N > 0, "epilogue_bwd: bad argument");
    I2V_CHECK_ARG(!relu || y, "epilogue_bwd: relu needs y");
    if (M == 0) return I2V_OK;
    const bool vec = (N & 3) == 0;
    const int cols = vec ? 1024 : 256;
    // ordered column sums (round 5): with the caller's split workspace (i2v_conv_fwd's: zeroed counters + slab) the row
    // blocks' partial sums are added in block order by the last block to arrive -- few blocks then, the finisher reads them all
    // ... and small tensors only (at most 2^21 elements: the relation head's layers): a large one (netD_style's 37500 x 2560
    // projections) needs its hundreds of row blocks to stream at full rate (measured: configs[2] 46.3 -> 48.1 ms with every
    // tensor held to <= 32 row blocks), so it keeps the atomics
    // round 6: large tensors too -- they keep their row blocks (full streaming rate) and the sums meet in two levels
    // (colsum_finish4); `small` = the tensors round 5 ordered by cutting them into few row blocks
    const bool want_ord = gbias && vec && g_i2v_tuning[I2V_TUNE_SPLIT_ATOMICS] == 0;
    const bool small = M * (long long)N <= (1ll << 21);
    auto ordered = [&](long long nblk, int ncolblk, float*& part, int*& cnt) {
        part = nullptr; cnt = nullptr;
        if (!want_ord || nblk < 2) return;
        const long long groups = (nblk + kColsumGroup - 1) / kColsumGroup;
        const size_t need = kSplitCounterBytes + (size_t)(nblk + groups) * N * sizeof(float);
        if (!split_ws || (long long)ncolblk * (1 + groups) > kSplitCounters || need > split_ws_bytes || need >= (1ull << 31)) {
            ++g_ordered_fallbacks;
            return;
        }
        cnt = reinterpret_cast<int*>(split_ws);
        part = reinterpret_cast<float*>(static_cast<char*>(split_ws) + kSplitCounterBytes);
    };
    float* part; int* cnt;
    // enough workgroups to cover the chip even for the 64..128-row tensors of the relation head
    int rows = 64;
    while (rows > 4 && (long long)i2v_cdiv(M, rows) * i2v_cdiv(N, cols) < 2 * NUM_CU) rows >>= 1;
    if (vec && N <= 512 && M >= 1024 && !g_t) {
        // tall and narrow: all 256 threads on one row block, one atomic per column per workgroup
        const int lanes = 256 / (N >> 2);
        // few workgroups: same-address atomics retire one per ~150 ns, so 256 contenders cost more than the rows
        int rpb = lanes * 64;
        while (rpb > lanes && i2v_cdiv(M, rpb) < 48) rpb >>= 1;
        if (want_ord && small) while (i2v_cdiv(M, rpb) > 64) rpb <<= 1;  // small tensors: at most 64 partials per column (two groups)
        ordered(i2v_cdiv(M, rpb), 1, part, cnt);
        epilogue_bwd_narrow_kernel<<<(unsigned)i2v_cdiv(M, rpb), 256, 0, (hipStream_t)stream>>>(gy, y, scale, g, gpre,
                                                                                              gbias, M, N, relu, rpb, part, cnt);
        I2V_CHECK_LAUNCH("epilogue_bwd");
        return I2V_OK;
    }
    if (want_ord && vec && small) while (i2v_cdiv(M, rows) > 32 && rows < 1024) rows <<= 1;   // small tensors: at most 32 row blocks (one level)
    dim3 grid(i2v_cdiv(M, rows), i2v_cdiv(N, cols));
    ordered(grid.x, (int)grid.y, part, cnt);
    if (vec) {
        epilogue_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gy, y, scale, g, gpre, gbias, M, N, relu, rows, g_t, part, cnt);
    } else {
        // N % 4 != 0 (the RPN's 18-channel cls_score): ordered = partial rows + the reduce pass of the filter gradients
        float* sp = nullptr;
        if (gbias && grid.x > 1 && g_i2v_tuning[I2V_TUNE_SPLIT_ATOMICS] == 0) {
            if (split_ws && kSplitCounterBytes + (size_t)grid.x * N * sizeof(float) <= split_ws_bytes)
                sp = reinterpret_cast<float*>(static_cast<char*>(split_ws) + kSplitCounterBytes);
            else ++g_ordered_fallbacks;
        }
        epilogue_bwd_scalar_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(gy, y, scale, g, gpre, gbias, M, N, relu, rows, g_t, sp);
        if (sp) wgrad_reduce_scalar_kernel<<<(unsigned)i2v_cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(sp, gbias, (int)grid.x, N, 1);
    }
    I2V_CHECK_LAUNCH("epilogue_bwd");
    return I2V_OK;
}

extern "C" int32_t i2v_maxpool3x3s2_fwd(const float* x, float* y, int32_t* argmax, int32_t B, int32_t H, int32_t W,
                                        int32_t C, void* stream) {
    I2V_CHECK_ARG(x && y && B > 0 && H >= 3 && W >= 3 && C > 0 && C % 4 == 0, "maxpool: bad argument");
    // ceil_mode, pad 0: Ho = ceil((H-3)/2)+1, and the last window must start inside the input
    int Ho = (H - 3 + 1) / 2 + 1, Wo = (W - 3 + 1) / 2 + 1;
    if ((Ho - 1) * 2 >= H) --Ho;
    if ((Wo - 1) * 2 >= W) --Wo;
    const long long total = (long long)B * Ho * Wo * (C / 4);
    maxpool3x3s2_kernel<<<(int)fmin((double)i2v_cdiv(total, 256), 8192.0), 256, 0, (hipStream_t)stream>>>(
        x, y, argmax, B, H, W, C, Ho, Wo);
    I2V_CHECK_LAUNCH("maxpool3x3s2");
    return I2V_OK;
}

extern "C" int32_t i2v_sgd_momentum_multi(float* const* p, const float* const* g, float* const* m, const int64_t* n,
                                          const float* lr, const float* weight_decay, int32_t count, float momentum,
                                          void* stream) {
    I2V_CHECK_ARG(count >= 0 && (count == 0 || (p && g && m && n && lr && weight_decay)), "sgd_momentum_multi: bad argument");
    for (int32_t c0 = 0; c0 < count; c0 += SGD_MULTI_MAX) {
        SgdMulti t;
        t.count = 0;
        t.mom = momentum;
        int blocks = 0;
        for (int32_t c = c0; c < count && t.count < SGD_MULTI_MAX; ++c) {
            I2V_CHECK_ARG(p[c] && g[c] && m[c] && n[c] >= 0, "sgd_momentum_multi: bad tensor");
            if (n[c] == 0) continue;
            const int k = t.count++;
            t.p[k] = p[c]; t.g[k] = g[c]; t.m[k] = m[c]; t.n[k] = n[c]; t.lr[k] = lr[c]; t.wd[k] = weight_decay[c];
            t.first_block[k] = blocks;
            blocks += (int)i2v_cdiv(n[c], (long long)SGD_MULTI_PER_BLOCK);
        }
        t.first_block[t.count] = blocks;
        if (blocks == 0) continue;
        sgd_momentum_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(t);
        I2V_CHECK_LAUNCH("sgd_momentum_multi");
    }
    return I2V_OK;
}

extern "C" int32_t i2v_adam_step(int32_t* step_counter, void* stream) {
    I2V_CHECK_ARG(step_counter, "adam_step: null counter");
    adam_step_kernel<<<1, 1, 0, (hipStream_t)stream>>>(step_counter);
    I2V_CHECK_LAUNCH("adam_step");
    return I2V_OK;
}

extern "C" int32_t i2v_adam_multi(float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                                  const double* lr, const float* weight_decay, int32_t count, double beta1, double beta2,
                                  double eps, const int32_t* step_counter, void* stream) {
    I2V_CHECK_ARG(count >= 0 && step_counter && (count == 0 || (p && g && m && v && n && lr && weight_decay)), "adam_multi: bad argument");
    for (int32_t c0 = 0; c0 < count;) {
        AdamMulti t;
        t.count = 0;
        t.b1 = beta1; t.b2 = beta2; t.eps = eps; t.step = step_counter;
        int blocks = 0;
        int32_t c = c0;
        for (; c < count && t.count < SGD_MULTI_MAX; ++c) {
            I2V_CHECK_ARG(p[c] && g[c] && m[c] && v[c] && n[c] >= 0, "adam_multi: bad tensor");
            if (n[c] == 0) continue;
            const long long nb = i2v_cdiv(n[c], (long long)SGD_MULTI_PER_BLOCK);
            if (t.count && blocks + nb > (1 << 20)) break;        // a very large tensor starts its own launch
            const int k = t.count++;
            t.p[k] = p[c]; t.g[k] = g[c]; t.m[k] = m[c]; t.v[k] = v[c]; t.n[k] = n[c]; t.lr[k] = lr[c]; t.wd[k] = weight_decay[c];
            t.first_block[k] = blocks;
            blocks += (int)nb;
        }
        c0 = c;
        t.first_block[t.count] = blocks;
        if (blocks == 0) continue;
        adam_multi_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(t);
        I2V_CHECK_LAUNCH("adam_multi");
    }
    return I2V_OK;
}

extern "C" int32_t i2v_sgd_momentum(float* p, const float* g, float* m, int64_t n, float lr, float momentum,
                                    float weight_decay, void* stream) {
    I2V_CHECK_ARG(p && g && m && n >= 0, "sgd_momentum: bad argument");
    if (n == 0) return I2V_OK;
    const long long n4 = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) ? 0 : n / 4;
    long long work = n4 > 0 ? n4 : n;
    int grid = (int)fmin((double)i2v_cdiv(work, 256), 8192.0);
    if (n4 == 0) grid = i2v_cdiv(n, 256);
    sgd_momentum_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p, g, m, n4, n, lr, momentum, weight_decay);
    I2V_CHECK_LAUNCH("sgd_momentum");
    return I2V_OK;
}

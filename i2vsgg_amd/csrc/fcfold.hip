// One pass over a large linear layer's filter per training step (vrd.fc6: 4096 x 50176 fp32 = 822 MB, resnet_SGG_emb.py:83,
// :146-151): the SGD(momentum) update that the PREVIOUS step's backward left pending is applied while THIS step's forward
// streams the filter --
//
//     gw[n][k] = sum_m gp[m][n] * xp[m][k]                 (pending filter gradient: rows of the previous minibatch)
//     m[n][k]  = mom * m[n][k] + (gw[n][k] + wd * w[n][k]);   w[n][k] -= lr * m[n][k]
//     y[r][n] += sum_k x[r][k] * w[n][k]                    (forward of the current minibatch, on the FRESH tile)
//
// so the filter and its momentum are read once and written once per step (3.3 GB) where forward + fused wgrad/SGD read the
// filter twice (4.1 GB), and the update's memory phases lie under the forward's MFMA phases.  Both GEMMs have <= 128 rows:
// the pair is matrix-pipe bound (2 x 52.6 GFLOP for fc6).
//
// Workgroup = 64 filters x one K range, 8 waves as 4 (k tiles of 16) x 2 (filter halves), v_mfma_f32_16x16x4_f32.
//   * gp[:, 64 filters] is transposed once into LDS (Gs[n][m]); every iteration transposes a 64-k slice of xp into LDS
//     (Xs[k][m]) -- both operands of the gradient GEMM are reduction-major in memory, the 4x4 register transposes of
//     conv_wgrad2_f32.  The MFMA operands are swapped (D[k][n]): a lane's four accumulators are four consecutive k of one
//     filter row, i.e. one 16-byte piece of w / m -- the update happens in registers on 16-byte loads and stores.
//   * that same register quad IS the B fragment of the forward MFMA for k-group (k tile) and filter tile of the wave (the
//     K permutation "lane group g owns k = 4g..4g+3" of conv_igemm_f32): the fresh filter never visits LDS.  The A fragments
//     of the forward (x[r][k..k+3]) are 16-byte row pieces: loaded straight from memory (L2) into registers.
//   * a wave sums its own 16-k quarter of every 64-k slice into its y partial (128 rows x 32 filters); at the end the four
//     quarters are folded in LDS and the sums go to y with fp32 atomics (y arrives zeroed; the K range 0 workgroup adds the bias).
//   * workgroups of one K range share an XCD (blockIdx % 8 = range): the 64 filter tiles of a range stream the same columns
//     of x / xp through one L2.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int FT = 512;              // threads: 8 waves as 4 (k tiles) x 2 (filter halves)
constexpr int FBN = 64, FBK = 64;    // filters per workgroup, k per iteration
constexpr int FM = 128;              // rows (pending and current minibatch) the kernel is built for
constexpr int FSPLITS = 8;           // K ranges: 8 x (N / 64) workgroups; one range per XCD

struct FoldP {
    const float* x; const float* xp; const float* gp; const int* valid;
    float* w; float* m; const float* bias; float* y;
    int M, Mp, N, K, kper;
    unsigned x_bytes, xp_bytes, gp_bytes, w_bytes;
    float lr, mom, wd;
    int ablate;                  // I2V_TUNE_FC_FOLD: diagnostic only
};

// LDS images [row][128 m]: 512-byte rows; the 16-byte column is XOR-swizzled so that the 16 lanes of a fragment read (rows
// r0 + 0..15, one column) and the 16 lanes of a transposing store (rows 4 * (0..15) + c, one column) each hit 16 distinct slots
__device__ inline int sw(int row, int c4) { return row * FM + ((c4 ^ ((row & 15) ^ ((row >> 4) & 3))) << 2); }

// 8 waves of ~200 VGPRs: one workgroup per CU, two waves per SIMD.  Every load of the K loop is issued unconditionally (a
// slice past the end of the K range gets the out-of-range bit in its buffer offset and costs no traffic): with loads under
// `if (more)` the compiler cannot count the loads in flight and drains vmcnt(0) in front of the forward MFMAs -- behind the
// filter / momentum loads of the NEXT slice, i.e. one exposed HBM round trip per iteration (measured: 1.42 ms -> see DESIGN.md).
template <bool UPD>
__device__ __forceinline__ void fc_fold_body(const FoldP& p, float* smem, f32x4 (&hacc)[8][2], int n0, int kbeg, int kend) {
    float* Gs = smem;
    float* Xs0 = smem + FBN * FM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xpr = __builtin_amdgcn_make_buffer_rsrc((void*)p.xp, 0, p.xp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gpr = __builtin_amdgcn_make_buffer_rsrc((void*)p.gp, 0, p.gp_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc((void*)p.m, 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    constexpr int NT = 2;                                            // cache policy "nt": the filter and its momentum are streamed once

    // roles of the transposing loads: a 4 (m) x 4 (column) block per thread, 16 column groups x 32 row blocks
    const int cgp = tid & 15, mb = tid >> 4;
    f32x4 xq[4];
    auto tload = [&](const __amdgpu_buffer_rsrc_t rs, int rowlen, int col0, int rows, bool live) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int mrow = 4 * mb + t;
            const unsigned off = ((unsigned)(mrow * rowlen + col0 + 4 * cgp) * 4u) | ((live && mrow < rows) ? 0u : OOB);
            xq[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        }
    };
    auto tstore = [&](float* dst) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *(f32x4*)&dst[sw(4 * cgp + c, mb)] = (f32x4){xq[0][c], xq[1][c], xq[2][c], xq[3][c]};
    };
    if (UPD) {                                                       // gp[:, n0 .. n0+63] -> Gs[n][m], once
        tload(gpr, p.N, n0, p.Mp, true);
        tstore(Gs);
        tload(xpr, p.K, kbeg, p.Mp, true);                           // first slice of xp, in flight
    }
    // this wave's filter / momentum pieces: w[n0 + (2 wn + nt) 16 + fr][k0 + wk 16 + 4 fg .. +3]; byte offsets fit 31 bits
    unsigned woff[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) woff[nt] = (unsigned)((n0 + (2 * wn + nt) * 16 + fr) * p.K + wk * 16 + 4 * fg) * 4u;
    // three slices of filter / momentum pieces in flight per wave (96 KB per CU): with one slice the kernel is bound by the
    // latency of its own streaming loads -- 2.7 TB/s, 1.2 ms for fc6 with every MFMA removed (tools/fold_probe.py)
    constexpr int PF = 3;
    f32x4 wq[PF][2], mq[PF][2];
    auto wload = [&](int st, int k0, bool live) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const unsigned off = (woff[nt] + (unsigned)k0 * 4u) | (live ? 0u : OOB);
            wq[st][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, NT));
            if (UPD) mq[st][nt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mr, off, 0, NT));
        }
    };
#pragma unroll
    for (int st = 0; st < PF; ++st) wload(st, kbeg + st * FBK, kbeg + st * FBK < kend);
    // forward A fragments of this wave's k tile: x[r = mt 16 + fr][k0 + wk 16 + 4 fg .. +3], straight from L2
    unsigned xoff[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) xoff[mt] = ((unsigned)((mt * 16 + fr) * p.K + wk * 16 + 4 * fg) * 4u) | (mt * 16 + fr < p.M ? 0u : OOB);

    int buf = 0;
    for (int kb = kbeg; kb < kend; kb += PF * FBK)
#pragma unroll
    for (int st = 0; st < PF; ++st) {
        const int k0 = kb + st * FBK;
        if (k0 >= kend) break;               // uniform
        const bool more = k0 + FBK < kend;
        float* Xs = Xs0 + buf * FBK * FM;
        if (UPD && !(p.ablate & 16)) {
            tstore(Xs);                      // double buffered: the readers of this buffer finished two iterations ago
            __syncthreads();
            tload(xpr, p.K, k0 + FBK, p.Mp, more);                   // next slice, in flight over both GEMMs
        }
        f32x4 xa[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
            xa[mt] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, (xoff[mt] + (unsigned)k0 * 4u) | ((p.ablate & 4) ? OOB : 0u), 0, 0));
        f32x4 wnew[2];
        if (UPD) {
            // ---- pending filter gradient of the 16 k x 32 filters of this wave: D[k][n] = sum_m xp[m][k] gp[m][n]
            f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            if (!(p.ablate & 1))
#pragma unroll
            for (int q = 0; q < FM / 16; ++q) {
                const f32x4 av = *(const f32x4*)&Xs[sw(wk * 16 + fr, 4 * q + fg)];
                f32x4 bv[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) bv[nt] = *(const f32x4*)&Gs[sw((2 * wn + nt) * 16 + fr, 4 * q + fg)];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[nt][t], acc[nt], 0, 0, 0);
            }
            // ---- the update, in registers: g' = g + wd*w ; m = mom*m + g' ; w -= lr*m   (the order of sgd_momentum_kernel)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                f32x4 pv = wq[st][nt], mv = mq[st][nt];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    mv[c] = p.mom * mv[c] + (acc[nt][c] + p.wd * pv[c]);
                    pv[c] -= p.lr * mv[c];
                }
                const unsigned off = (woff[nt] + (unsigned)k0 * 4u) | ((p.ablate & 8) ? OOB : 0u);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, mv), mr, off, 0, NT);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, pv), wr, off, 0, NT);
                wnew[nt] = pv;
            }
        } else {
            wnew[0] = wq[st][0];
            wnew[1] = wq[st][1];
        }
        wload(st, k0 + PF * FBK, k0 + PF * FBK < kend);              // this stage's registers are free: the slice three ahead
        // ---- forward on the fresh tile: D[r][n] += sum_k x[r][k] w[n][k], this wave's 16 k
        if (!(p.ablate & 2))
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    hacc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[mt][t], wnew[nt][t], hacc[mt][nt], 0, 0, 0);
        buf ^= 1;
    }
}

__global__ void __launch_bounds__(FT, 2) fc_fold_kernel(const FoldP p) {
    __shared__ __attribute__((aligned(16))) float smem[(FBN + 2 * FBK) * FM];   // Gs + two Xs buffers: 96 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int ntiles = p.N / FBN;
    const int bid = blockIdx.x;
    const int split = (bid & 7) + 8 * (bid / (8 * ntiles));
    const int n0 = ((bid >> 3) % ntiles) * FBN;
    const int kbeg = split * p.kper, kend = min(p.K, kbeg + p.kper);
    if (kbeg >= kend) return;
    f32x4 hacc[8][2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) hacc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (*p.valid != 0) fc_fold_body<true>(p, smem, hacc, n0, kbeg, kend);       // uniform: a pending update exists
    else fc_fold_body<false>(p, smem, hacc, n0, kbeg, kend);

    // ---- the four k quarters of a filter half meet in LDS (two folding rounds), then y += partial with fp32 atomics (the K
    // ranges meet there; y arrives zeroed; range 0 adds the bias)
    float* red = smem;                                               // 4 slots of 64 registers x 64 lanes = 64 KB
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int half = round == 0 ? 2 : 1;                         // waves wk >= half hand their partial to wk - half
        __syncthreads();
        if (wk >= half && wk < 2 * half) {
            float* slot = red + ((wk - half) * 2 + wn) * 64 * 64;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) slot[((mt * 2 + nt) * 4 + rr) * 64 + lane] = hacc[mt][nt][rr];
        }
        __syncthreads();
        if (wk < half) {
            const float* slot = red + (wk * 2 + wn) * 64 * 64;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) hacc[mt][nt][rr] += slot[((mt * 2 + nt) * 4 + rr) * 64 + lane];
        }
    }
    if (wk == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + (2 * wn + nt) * 16 + fr;
            const float b = (p.bias && split == 0) ? p.bias[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = mt * 16 + 4 * fg + rr;
                    if (r < p.M) atomicAdd(p.y + (long long)r * p.N + n, hacc[mt][nt][rr] + b);
                }
        }
    }
}

}  // namespace

extern "C" int32_t i2v_fc_fold_supported(int32_t M, int32_t Mp, int32_t N, int32_t K) {
    return M > 0 && M <= FM && Mp >= 0 && Mp <= FM && N % FBN == 0 && K % FBK == 0 && (long long)FM * K * 4 < (1ll << 31) &&
           (long long)FM * N * 4 < (1ll << 31) && (long long)N * K * 4 < (1ll << 31);
}

extern "C" int32_t i2v_fc_fold_fwd(const float* x, const float* x_pending, const float* g_pending, const int32_t* pending_valid,
                                   float* w, float* m, const float* bias, float* y, int32_t M, int32_t M_pending, int32_t N,
                                   int32_t K, float lr, float momentum, float weight_decay, void* stream) {
    I2V_CHECK_ARG(x && x_pending && g_pending && pending_valid && w && m && y, "fc_fold_fwd: null pointer");
    if (!i2v_fc_fold_supported(M, M_pending, N, K)) {
        i2v_set_error("fc_fold_fwd: needs rows <= %d, N %% %d == 0, K %% %d == 0 (got M %d / %d, N %d, K %d)", FM, FBN, FBK, M,
                      M_pending, N, K);
        return I2V_ERR_UNSUPPORTED;
    }
    FoldP p;
    p.x = x; p.xp = x_pending; p.gp = g_pending; p.valid = pending_valid; p.w = w; p.m = m; p.bias = bias; p.y = y;
    p.M = M; p.Mp = M_pending; p.N = N; p.K = K;
    p.kper = i2v_cdiv(i2v_cdiv(K, FBK), FSPLITS) * FBK;
    p.x_bytes = (unsigned)((long long)M * K * 4);
    p.xp_bytes = (unsigned)((long long)(M_pending > 0 ? M_pending : 1) * K * 4);
    p.gp_bytes = (unsigned)((long long)(M_pending > 0 ? M_pending : 1) * N * 4);
    p.w_bytes = (unsigned)((long long)N * K * 4);
    p.lr = lr; p.mom = momentum; p.wd = weight_decay;
    p.ablate = g_i2v_tuning[I2V_TUNE_FC_FOLD];
    fc_fold_kernel<<<FSPLITS * (N / FBN), FT, 0, (hipStream_t)stream>>>(p);
    I2V_CHECK_LAUNCH("fc_fold_fwd");
    return I2V_OK;
}

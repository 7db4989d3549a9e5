// K-loop laboratory for conv_gemm_f32 (round 6): y[m][n] = sum_k a[m][k] * w[n][k], fp32 in / fp32 accumulate on
// v_mfma_f32_16x16x4_f32, same LDS image (128-byte rows, 16-byte column XOR (row >> 1) & 7), same fragment ownership and the
// same k order per accumulator as the product kernel -- every variant must be BIT-EQUAL to variant R (the product's loop).
//   R : the product's loop -- two LDS buffers, global -> VGPR -> ds_write_b128, one __syncthreads per 32-k stage
//   D : LDS-DMA ring (buffer_load_dwordx4 ... lds; the swizzle rides on the SOURCE address, the LDS image is lane-linear),
//       NBUF buffers, loads NBUF-1 stages ahead, counted vmcnt, raw s_barrier, one barrier per stage
//       FP = 0: one fragment set (12 ds_read_b128 in two groups, as R); 1: both halves' fragments read up front (two sets);
//       FP = 2: stage i+1 is published at barrier i, its first half's fragments are read under stage i's second-half MFMAs
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/micro/gemm_lab.hip -o tools/micro/gemm_lab
// Run:   tools/micro/gemm_lab [rounds]     (prints one line per shape x variant: median / min us, TF, bit-equal to R)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <type_traits>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int BKS = 32, THREADS = 256;
constexpr unsigned INV = 0x80000000u;

struct GP {
    const float* a; const float* w; float* y;
    int M, N, K, k_per_split;
    unsigned a_bytes, w_bytes;
    unsigned long long* clk;        // non-null: per workgroup {prologue, K loop, epilogue} shader cycles + {start, end} in 100 MHz ticks
};

template <int N> __device__ inline void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one LDS-DMA piece: 64 lanes x 16 bytes land at LDS byte address `dst` (wave-uniform) + lane * 16; the source is per lane
__device__ inline void dma16(unsigned dst, unsigned voff, __amdgpu_buffer_rsrc_t r, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dst), "v"(voff), "s"(r), "s"(soff) : "memory");
}

struct Stamp {
    unsigned long long t0, t1, t2, r0;
    __device__ inline void begin(const GP& p) { if (p.clk) { r0 = __builtin_amdgcn_s_memrealtime(); t0 = __builtin_amdgcn_s_memtime(); } }
    __device__ inline void loop(const GP& p) { if (p.clk) t1 = __builtin_amdgcn_s_memtime(); }
    __device__ inline void done(const GP& p) { if (p.clk) t2 = __builtin_amdgcn_s_memtime(); }
    __device__ inline void end(const GP& p) {
        if (p.clk && threadIdx.x == 0) {
            unsigned long long* o = p.clk + 8 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            o[0] = t1 - t0; o[1] = t2 - t1; o[2] = __builtin_amdgcn_s_memtime() - t2; o[3] = r0; o[4] = __builtin_amdgcn_s_memrealtime();
        }
    }
};

__device__ inline int xcd_tile() {
    const int nt = gridDim.x, q = nt >> 3, r = nt & 7, x = blockIdx.x & 7, i = blockIdx.x >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

template <int WM, int WN, int TM, int TN>
__device__ inline void store_tile(const GP& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int fr, int fg) {
    float* y = p.y + (size_t)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m = m0 + (wm * TM + i) * 16 + fr, n = n0 + (wn * TN + j) * 16 + 4 * fg;
            if (m < p.M && n < p.N) *(f32x4*)(y + (size_t)m * p.N + n) = acc[i][j];
        }
}

// ------------------------------------------------------------------ R: the product's loop
template <int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(THREADS) gemm_r(const GP p) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int A_LD = (BM * 8 + THREADS - 1) / THREADS, B_LD = (BN * 8 + THREADS - 1) / THREADS;
    Stamp st;
    st.begin(p);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float (*As)[BM * BKS] = reinterpret_cast<float (*)[BM * BKS]>(smem);
    float (*Bs)[BN * BKS] = reinterpret_cast<float (*)[BN * BKS]>(smem + 2 * BM * BKS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN, fr = lane & 15, fg = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = xcd_tile();
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kbeg = blockIdx.y * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    const int kc = tid & 7, kg = kc * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    unsigned a_vk[A_LD], b_vk[B_LD];
#pragma unroll
    for (int q = 0; q < A_LD; ++q) {
        const int row = (tid >> 3) + q * 32, m = m0 + row;
        a_vk[q] = (row < BM && m < p.M) ? ((unsigned)(m * p.K) + (unsigned)kg) * 4u : INV;
    }
#pragma unroll
    for (int q = 0; q < B_LD; ++q) {
        const int row = (tid >> 3) + q * 32, n = n0 + row;
        b_vk[q] = (row < BN && n < p.N) ? ((unsigned)(n * p.K) + (unsigned)kg) * 4u : INV;
    }
    float4 ra[A_LD], rb[B_LD];
    auto gload = [&](int k0) {
        const unsigned so = (unsigned)k0 * 4u;
#pragma unroll
        for (int q = 0; q < A_LD; ++q) ra[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, a_vk[q], so, 0));
#pragma unroll
        for (int q = 0; q < B_LD; ++q) rb[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wr, b_vk[q], so, 0));
    };
    auto sstore = [&](int S) {
#pragma unroll
        for (int q = 0; q < A_LD; ++q) {
            const int row = (tid >> 3) + q * 32;
            if (row < BM) *(float4*)&As[S][row * BKS + ((kc ^ ((row >> 1) & 7)) << 2)] = ra[q];
        }
#pragma unroll
        for (int q = 0; q < B_LD; ++q) {
            const int row = (tid >> 3) + q * 32;
            if (row < BN) *(float4*)&Bs[S][row * BKS + ((kc ^ ((row >> 1) & 7)) << 2)] = rb[q];
        }
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 av[TM], bv[TN];
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
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                        const float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i][j], 0, 0, 0);
                    }
        }
    };
    gload(kbeg);
    sstore(0);
    __syncthreads();
    st.loop(p);
    int buf = 0, k0 = kbeg;
    for (; k0 + BKS < kend; k0 += BKS) {
        gload(k0 + BKS);
        __builtin_amdgcn_sched_barrier(0);
        compute(buf);
        sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    compute(buf);
    st.done(p);
    store_tile<WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, fr, fg);
    st.end(p);
}

// ------------------------------------------------------------------ D: LDS-DMA ring
// BK = 32: 128-byte LDS rows, 16-byte column XOR (row >> 1) & 7 (the product's image).  BK = 16: 64-byte rows, a stage is ONE
// 16-k half, column XOR (-(row >> 2)) & 3 -- the four 16-lane groups of a ds_read_b128 still touch 16 distinct 16-byte slots of
// the 256-byte bank row; half the LDS per buffer, twice the barriers.
template <int BK> __device__ inline int swz(int row) { return BK == 32 ? ((row >> 1) & 7) : ((-(row >> 2)) & 3); }

template <int WM, int WN, int TM, int TN, int NBUF, int FP, int BK>
__device__ __forceinline__ void gemm_d_body(const GP& p) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    constexpr int PR = 256 / BK;                                   // rows per 1-KB piece: 8 (BK 32) or 16 (BK 16)
    constexpr int CPR = BK / 4;                                    // 16-byte columns per row
    constexpr int NPA = BM / PR, NPB = BN / PR, NP = NPA + NPB;
    constexpr int PMAX = (NP + 3) / 4, REM = NP % 4;               // pieces per wave: PMAX for waves < REM (all when REM == 0), else PMAX - 1
    constexpr int STAGE = (BM + BN) * BK;                          // floats per stage buffer: A rows, then B rows
    constexpr int D = NBUF - 1;                                    // stages requested ahead of the one being multiplied
    constexpr int H = BK / 16;                                     // 16-k halves per stage
    static_assert(BM % PR == 0 && BN % PR == 0, "whole pieces");
    static_assert(NBUF >= 2 && (FP != 2 || NBUF >= 3), "the early-publish form needs three buffers");
    Stamp st;
    st.begin(p);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN, fr = lane & 15, fg = lane >> 4;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile = xcd_tile();
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int kbeg = blockIdx.y * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    const int nst = (kend - kbeg + BK - 1) / BK;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    // piece q of this wave = piece q * 4 + wave of the stage: PR rows of [A rows | B rows]; lane l lands at byte l * 16 of the
    // piece = (row l / CPR, physical column l % CPR) and therefore FETCHES logical column (l % CPR) ^ swz(row)
    unsigned voff[PMAX];
#pragma unroll
    for (int q = 0; q < PMAX; ++q) {
        const int pc = q * 4 + wave;
        const bool isA = pc < NPA;
        const int row = (isA ? pc : pc - NPA) * PR + lane / CPR;
        const int g = (isA ? m0 : n0) + row;
        const int chunk = (lane % CPR) ^ swz<BK>(row);
        voff[q] = (pc < NP && g < (isA ? p.M : p.N)) ? ((unsigned)(g * p.K) + (unsigned)(chunk * 4)) * 4u : INV;
    }
    __amdgpu_buffer_rsrc_t rs[PMAX];          // the operand a piece belongs to is wave-uniform: a scalar select, once
#pragma unroll
    for (int q = 0; q < PMAX; ++q) rs[q] = (q * 4 + wave < NPA) ? xr : wr;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;      // LDS byte address of the ring
    auto issue = [&](int s) {           // stage s -> buffer s % NBUF; a stage beyond the end is requested out of range (zeros, no traffic)
        const int k0 = kbeg + s * BK;
        const unsigned so = (unsigned)k0 * 4u;
        const unsigned kinv = k0 < kend ? 0u : INV;
        const unsigned base = lds0 + (unsigned)((s % NBUF) * STAGE * 4);
#pragma unroll
        for (int q = 0; q < PMAX; ++q) {
            const int pc = q * 4 + wave;
            // inline asm: hipcc counts a builtin LDS-DMA as a pending LDS write and drains vmcnt(0) in front of EVERY ds_read
            if (REM == 0 || q < PMAX - 1 || wave < REM) dma16(base + (unsigned)(pc * 1024), voff[q] | kinv, rs[q], so);
        }
    };
    auto wait_keep = [&](auto keep_c) {          // all of this wave's pieces landed except those of the newest `keep` stages
        constexpr int keep = decltype(keep_c)::value;
        if (REM == 0 || wave < REM) wait_vm<keep * PMAX>();
        else wait_vm<keep * (PMAX - 1)>();
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto rd = [&](int s, int h, float4 (&av)[TM], float4 (&bv)[TN]) {
        const float* A = smem + (s % NBUF) * STAGE;
        const float* B = A + BM * BK;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = (wm * TM + i) * 16 + fr;
            av[i] = *(const float4*)&A[row * BK + (((h * 4 + fg) ^ swz<BK>(row)) << 2)];
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = (wn * TN + j) * 16 + fr;
            bv[j] = *(const float4*)&B[row * BK + (((h * 4 + fg) ^ swz<BK>(row)) << 2)];
        }
    };
    auto mm = [&](const float4 (&av)[TM], const float4 (&bv)[TN]) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float a = t == 0 ? av[i].x : t == 1 ? av[i].y : t == 2 ? av[i].z : av[i].w;
                    const float b = t == 0 ? bv[j].x : t == 1 ? bv[j].y : t == 2 ? bv[j].z : bv[j].w;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i][j], 0, 0, 0);
                }
    };
#pragma unroll
    for (int s = 0; s < D; ++s) issue(s);
    wait_keep(std::integral_constant<int, D - 1>{});      // (the loop's first wait, taken here so that the K-loop stamp starts with stage 0 in LDS)
    st.loop(p);
    float4 a0[TM], b0[TN], a1[TM], b1[TN];
    if constexpr (FP < 2) {
        for (int i = 0; i < nst; ++i) {
            wait_keep(std::integral_constant<int, D - 1>{});     // stage i has landed (mine)
            __builtin_amdgcn_s_barrier();                        // ... everyone's; and everyone is done with stage i - 1
            issue(i + D);                                        // into the buffer stage i - 1 occupied
            if constexpr (H == 1) {
                rd(i, 0, a0, b0); mm(a0, b0);
            } else if constexpr (FP == 0) {
                rd(i, 0, a0, b0); mm(a0, b0);
                rd(i, 1, a0, b0); mm(a0, b0);
            } else {
                rd(i, 0, a0, b0); rd(i, 1, a1, b1);
                __builtin_amdgcn_sched_barrier(0);      // left alone the scheduler sinks the second set's reads below the first 16 MFMAs
                mm(a0, b0); mm(a1, b1);
            }
        }
    } else if constexpr (H == 2) {
        // stage i + 1 is published at barrier i (one stage early): the fragments of its first half are read while the second
        // half of stage i multiplies, so no fragment read waits behind a barrier
        wait_keep(std::integral_constant<int, D - 1>{});
        __builtin_amdgcn_s_barrier();                            // stage 0 visible
        rd(0, 0, a0, b0);
        for (int i = 0; i < nst; ++i) {
            wait_keep(std::integral_constant<int, D - 2>{});     // stage i + 1 has landed (mine)
            __builtin_amdgcn_s_barrier();                        // ... everyone's; everyone is done with stage i - 1
            issue(i + D);                                        // D = NBUF - 1: the buffer of stage i - 1
            rd(i, 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);                   // pinned: the scheduler would fold the two sets into one
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            rd(i + 1, 0, a0, b0);                                // beyond the end: zeros or stale data, never multiplied
            __builtin_amdgcn_sched_barrier(0);
            mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // 16-k stages, early publish: stage i + 1's only fragment set is read under stage i's MFMAs (two stages per trip)
        wait_keep(std::integral_constant<int, D - 1>{});
        __builtin_amdgcn_s_barrier();
        rd(0, 0, a0, b0);
        for (int i = 0; i < nst; i += 2) {
            wait_keep(std::integral_constant<int, D - 2>{});
            __builtin_amdgcn_s_barrier();
            issue(i + D);
            rd(i + 1, 0, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            wait_keep(std::integral_constant<int, D - 2>{});
            __builtin_amdgcn_s_barrier();
            issue(i + 1 + D);
            rd(i + 2, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < nst) mm(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    st.done(p);
    wait_vm<0>();            // the out-of-range requests of the tail have landed before the LDS goes back
    store_tile<WM, WN, TM, TN>(p, acc, m0, n0, wm, wn, fr, fg);
    st.end(p);
}

template <int WM, int WN, int TM, int TN, int NBUF, int FP, int BK>
__global__ void __launch_bounds__(THREADS) gemm_d(const GP p) { gemm_d_body<WM, WN, TM, TN, NBUF, FP, BK>(p); }

// ------------------------------------------------------------------ host
__global__ void spin_kernel(long long ticks) {      // holds one CU for ~ticks of the 100 MHz clock: lets the host queue launches behind it
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

struct Variant {
    std::string name;
    int bm, bn;
    size_t lds;
    void (*fn)(const GP);
};
template <int WM, int WN, int TM, int TN> Variant vr(const char* nm) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    auto k = gemm_r<WM, WN, TM, TN>;
    const size_t lds = (size_t)2 * (BM + BN) * BKS * 4;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return {nm, BM, BN, lds, k};
}
template <int WM, int WN, int TM, int TN, int NBUF, int FP, int BK = 32> Variant vd(const char* nm) {
    constexpr int BM = WM * TM * 16, BN = WN * TN * 16;
    auto k = gemm_d<WM, WN, TM, TN, NBUF, FP, BK>;
    const size_t lds = (size_t)NBUF * (BM + BN) * BK * 4;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    return {nm, BM, BN, lds, k};
}

struct Shape { const char* name; int M, N, K, splitk; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 7;
    const int reps = 10;
    std::vector<Variant> vs;
    vs.push_back(vr<1, 4, 5, 1>("R  80x64"));
    vs.push_back(vd<1, 4, 5, 1, 2, 0>("D2 80x64 fp0"));
    vs.push_back(vd<1, 4, 5, 1, 2, 0, 16>("D2 80x64 k16"));
    vs.push_back(vd<1, 4, 5, 1, 3, 0, 16>("D3 80x64 k16"));
    vs.push_back(vd<1, 4, 5, 1, 3, 2, 16>("D3 80x64 k16 fp2"));
    vs.push_back(vd<1, 4, 5, 1, 4, 0, 16>("D4 80x64 k16"));
    vs.push_back(vd<1, 4, 5, 1, 4, 2, 16>("D4 80x64 k16 fp2"));
    vs.push_back(vr<2, 2, 3, 2>("R  96x64"));
    vs.push_back(vd<2, 2, 3, 2, 2, 0>("D2 96x64 fp0"));
    vs.push_back(vd<2, 2, 3, 2, 2, 0, 16>("D2 96x64 k16"));
    vs.push_back(vd<2, 2, 3, 2, 3, 0, 16>("D3 96x64 k16"));
    vs.push_back(vd<2, 2, 3, 2, 3, 2, 16>("D3 96x64 k16 fp2"));
    vs.push_back(vr<2, 2, 2, 2>("R  64x64"));
    vs.push_back(vd<2, 2, 2, 2, 2, 0>("D2 64x64 fp0"));
    vs.push_back(vd<2, 2, 2, 2, 2, 0, 16>("D2 64x64 k16"));
    vs.push_back(vd<2, 2, 2, 2, 3, 0, 16>("D3 64x64 k16"));
    vs.push_back(vd<2, 2, 2, 2, 3, 2, 16>("D3 64x64 k16 fp2"));
    vs.push_back(vr<2, 2, 4, 2>("R  128x64"));
    vs.push_back(vd<2, 2, 4, 2, 2, 0>("D2 128x64 fp0"));
    vs.push_back(vd<2, 2, 4, 2, 2, 0, 16>("D2 128x64 k16"));
    vs.push_back(vd<2, 2, 4, 2, 3, 0, 16>("D3 128x64 k16"));
    vs.push_back(vd<2, 2, 4, 2, 3, 2, 16>("D3 128x64 k16 fp2"));
    vs.push_back(vd<2, 2, 4, 2, 3, 0>("D3 128x64 fp0"));
    vs.push_back(vd<2, 2, 4, 2, 3, 1>("D3 128x64 fp1"));
    vs.push_back(vd<2, 2, 4, 2, 4, 0>("D4 128x64 fp0"));
    vs.push_back(vd<2, 2, 4, 2, 4, 2>("D4 128x64 fp2"));
    vs.push_back(vd<2, 2, 4, 2, 4, 0, 16>("D4 128x64 k16"));
    vs.push_back(vd<2, 2, 4, 2, 6, 0, 16>("D6 128x64 k16"));
    vs.push_back(vr<2, 2, 4, 4>("R  128x128"));
    vs.push_back(vd<2, 2, 4, 4, 2, 0>("D2 128x128 fp0"));
    vs.push_back(vd<2, 2, 4, 4, 2, 0, 16>("D2 128x128 k16"));
    vs.push_back(vd<2, 2, 4, 4, 3, 0, 16>("D3 128x128 k16"));
    vs.push_back(vd<2, 2, 4, 4, 4, 2, 16>("D4 128x128 k16 fp2"));
    const Shape shapes[] = {
        {"4096^3", 4096, 4096, 4096, 1},
        {"l3c1x4 (9576x256x1024)", 9576, 256, 1024, 1},
        {"l3c1x4 split2", 9576, 256, 1024, 2},
        {"l3c3x2 (4788x1024x256)", 4788, 1024, 256, 1},
        {"l3c1x2 split3 (4788x256x1024)", 4788, 256, 1024, 3},
        {"l3c3x1 (2394x1024x256)", 2394, 1024, 256, 1},
        {"fc6 forward (128x4096x50176, 8 splits; the 822 MB filter streams from HBM)", 128, 4096, 50176, 8},
        {"fc6 forward, 4 splits", 128, 4096, 50176, 4},
    };
    const char* only = argc > 2 ? argv[2] : nullptr;
    for (const auto& v : vs) {
        int nb = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, v.fn, THREADS, v.lds));
        hipFuncAttributes fa;
        CK(hipFuncGetAttributes(&fa, (const void*)v.fn));
        printf("variant %-16s lds %6zu B  vgpr %3d  workgroups/CU %d\n", v.name.c_str(), v.lds, fa.numRegs, nb);
    }
    hipStream_t st[3];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1, ec[3];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto& e : ec) CK(hipEventCreate(&e));
    for (const auto& sh : shapes) {
        if (only && !strstr(sh.name, only)) continue;
        const size_t an = (size_t)sh.M * sh.K, wn = (size_t)sh.N * sh.K, yn = (size_t)sh.M * sh.N * sh.splitk;
        std::vector<float> ha(an), hw(wn);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
        for (auto& x : ha) x = rnd();
        for (auto& x : hw) x = rnd();
        float *da[3], *dw[3], *dy[3], *dref;
        for (int c = 0; c < 3; ++c) {
            CK(hipMalloc(&da[c], an * 4)); CK(hipMalloc(&dw[c], wn * 4)); CK(hipMalloc(&dy[c], yn * 4));
            CK(hipMemcpy(da[c], ha.data(), an * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(dw[c], hw.data(), wn * 4, hipMemcpyHostToDevice));
        }
        CK(hipMalloc(&dref, yn * 4));
        std::vector<float> href(yn), hy(yn);
        const double flop = 2.0 * sh.M * sh.N * sh.K;
        printf("== shape %s  (%.2f GFLOP)\n", sh.name, flop * 1e-9);
        auto params = [&](const Variant& v, int c) {
            GP p;
            p.a = da[c]; p.w = dw[c]; p.y = dy[c]; p.M = sh.M; p.N = sh.N; p.K = sh.K;
            const int ksteps = sh.K / BKS;
            p.k_per_split = ((ksteps + sh.splitk - 1) / sh.splitk) * BKS;
            p.a_bytes = (unsigned)(an * 4); p.w_bytes = (unsigned)(wn * 4);
            p.clk = nullptr;
            return p;
        };
        auto grid = [&](const Variant& v) { return dim3(((sh.M + v.bm - 1) / v.bm) * ((sh.N + v.bn - 1) / v.bn), sh.splitk, 1); };
        // correctness: every variant against the first R variant, bit for bit; R against a host spot check
        std::vector<int> equal(vs.size(), 0);
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            const auto& v = vs[vi];
            CK(hipMemset(dy[0], 0xFF, yn * 4));
            GP p = params(v, 0);
            hipLaunchKernelGGL(v.fn, grid(v), dim3(THREADS), v.lds, st[0], p);
            CK(hipStreamSynchronize(st[0]));
            CK(hipMemcpy(hy.data(), dy[0], yn * 4, hipMemcpyDeviceToHost));
            if (vi == 0) {
                href = hy;
                double worst = 0;
                for (int t = 0; t < 64; ++t) {
                    const int m = (t * 7919) % sh.M, n = (t * 104729) % sh.N;
                    double ref = 0;
                    for (int k = 0; k < sh.K; ++k) ref += (double)ha[(size_t)m * sh.K + k] * hw[(size_t)n * sh.K + k];
                    double got = 0;
                    for (int sp = 0; sp < sh.splitk; ++sp) got += hy[(size_t)sp * sh.M * sh.N + (size_t)m * sh.N + n];
                    worst = std::max(worst, fabs(got - ref));
                }
                printf("   R vs fp64 host spot check: max abs err %.3g\n", worst);
                equal[vi] = worst < 1e-2;
            } else {
                equal[vi] = memcmp(hy.data(), href.data(), yn * 4) == 0;
            }
        }
        // timing: interleaved rounds, `reps` back-to-back launches per measurement
        std::vector<std::vector<float>> us(vs.size()), us3(vs.size());
        for (int r = 0; r < rounds + 1; ++r)
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                const auto& v = vs[vi];
                GP p = params(v, 0);
                CK(hipEventRecord(e0, st[0]));
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(v.fn, grid(v), dim3(THREADS), v.lds, st[0], p);
                CK(hipEventRecord(e1, st[0]));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r > 0) us[vi].push_back(ms * 1000.f / reps);
            }
        // co-run: three independent chains of the same launch on three streams (what a step's three branches look like to a CU)
        for (int r = 0; r < 3; ++r)
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                const auto& v = vs[vi];
                CK(hipDeviceSynchronize());
                hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st[0], 60000LL);      // 0.6 ms
                CK(hipEventRecord(e0, st[0]));
                for (int c = 1; c < 3; ++c) CK(hipStreamWaitEvent(st[c], e0, 0));
                for (int i = 0; i < reps; ++i)
                    for (int c = 0; c < 3; ++c) {
                        GP p = params(v, c);
                        hipLaunchKernelGGL(v.fn, grid(v), dim3(THREADS), v.lds, st[c], p);
                    }
                for (int c = 0; c < 3; ++c) CK(hipEventRecord(ec[c], st[c]));
                float worst = 0;
                for (int c = 0; c < 3; ++c) {
                    CK(hipEventSynchronize(ec[c]));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, ec[c]));
                    worst = std::max(worst, ms);
                }
                if (r > 0) us3[vi].push_back(worst * 1000.f / (3 * reps));
            }
        // stamped passes (their own launches, never timed): one launch alone; one launch of chain 0 while chains 1 and 2 run the
        // same kernel beside it.  Per workgroup: prologue / K loop / epilogue shader cycles; medians over the workgroups
        std::vector<std::string> stamp_txt(vs.size());
        {
            unsigned long long* dclk;
            const size_t maxwg = 1 << 16;
            CK(hipMalloc(&dclk, maxwg * 8 * sizeof(unsigned long long)));
            std::vector<unsigned long long> hc(maxwg * 8);
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                const auto& v = vs[vi];
                const dim3 g = grid(v);
                const size_t nwg = (size_t)g.x * g.y;
                if (nwg > maxwg) continue;
                char buf[512]; std::string txt;
                for (int mode = 0; mode < 2; ++mode) {
                    CK(hipDeviceSynchronize());
                    CK(hipMemset(dclk, 0, nwg * 8 * sizeof(unsigned long long)));
                    if (mode == 1)
                        for (int i = 0; i < 6; ++i)
                            for (int c = 1; c < 3; ++c) { GP q = params(v, c); hipLaunchKernelGGL(v.fn, g, dim3(THREADS), v.lds, st[c], q); }
                    GP p = params(v, 0);
                    if (mode == 1) for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(v.fn, g, dim3(THREADS), v.lds, st[0], p);
                    p.clk = dclk;
                    hipLaunchKernelGGL(v.fn, g, dim3(THREADS), v.lds, st[0], p);
                    CK(hipDeviceSynchronize());
                    CK(hipMemcpy(hc.data(), dclk, nwg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                    std::vector<double> pro, loop, epi;
                    unsigned long long r0 = ~0ull, r1 = 0;
                    for (size_t w = 0; w < nwg; ++w) {
                        pro.push_back((double)hc[8 * w]); loop.push_back((double)hc[8 * w + 1]); epi.push_back((double)hc[8 * w + 2]);
                        r0 = std::min(r0, hc[8 * w + 3]); r1 = std::max(r1, hc[8 * w + 4]);
                    }
                    auto med = [](std::vector<double>& x) { std::sort(x.begin(), x.end()); return x[x.size() / 2]; };
                    const int ksteps = sh.K / BKS;
                    const int stages32 = (ksteps + sh.splitk - 1) / sh.splitk;       // 32-k stages per workgroup
                    const double lp = med(loop);
                    snprintf(buf, sizeof buf, " | %s: prologue %5.0f loop %6.0f (%5.0f per 32 k) epilogue %5.0f cyc, span %5.1f us",
                             mode ? "beside 2 chains" : "alone", med(pro), lp, lp / stages32, med(epi), (double)(r1 - r0) / 100.0);
                    txt += buf;
                }
                stamp_txt[vi] = txt;
            }
            CK(hipFree(dclk));
        }
        for (size_t vi = 0; vi < vs.size(); ++vi) {
            auto& u = us[vi];
            std::sort(u.begin(), u.end());
            auto& u3 = us3[vi];
            std::sort(u3.begin(), u3.end());
            const float med = u[u.size() / 2], mn = u[0], c3 = u3[0];
            printf("   %-16s med %8.2f us  min %8.2f us  %6.1f TF | 3 chains: %8.2f us/launch %6.1f TF | %s\n", vs[vi].name.c_str(), med, mn,
                   flop / med * 1e-6, c3, flop / c3 * 1e-6, equal[vi] ? "bit-equal" : "MISMATCH");
            if (!stamp_txt[vi].empty()) printf("   %-16s stamps%s\n", "", stamp_txt[vi].c_str());
        }
        for (int c = 0; c < 3; ++c) { CK(hipFree(da[c])); CK(hipFree(dw[c])); CK(hipFree(dy[c])); }
        CK(hipFree(dref));
        fflush(stdout);
    }
    return 0;
}

// Head projection h = lin(x) and its backward as fp32-input MFMA GEMMs on gfx950
// (v_mfma_f32_32x32x2_f32: exact f32 fma chains, so the result is an ordinary fp32
// GEMM).  Stands for `self.lin(x)` inside GATConv.forward and its autograd backward
// (call sites /root/reference/gat/model.py:40,43,46,54).
//
// One kernel covers the three products through two layout switches:
//   forward   hp[N,Dp]  = x[N,F]   * Wp[Dp,F]^T      A row-major, B n-major
//   d_x       dx[N,F]   = dhp[N,Dp]* Wp[Dp,F]        A row-major, B k-major
//   d_weight  dWp[Dp,F] = dhp[N,Dp]^T * x[N,F]       A k-major,  B k-major, split over N
// Wp is the weight with each head's rows padded from C to Cp (zero rows), so the output
// lands directly in the head-padded layout the attention kernels gather from.
#include "common.h"
#include <stdlib.h>

namespace gatmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

// AFix, GemmBatch, GemmDesc: common.h

// Tile BM x BN (64 or 128 each), 4 waves as 2 x 2, each wave (BM/64) x (BN/64) MFMA tiles of
// 32 x 32.  blockIdx.z = batch * nsplit + split: `batch` walks independent products (one per
// attention head: pointer strides sA/sB/sC), `split` cuts K (partial results sSplit apart).
// Columns [Nn, n_store) of C are written as zeros (padding of a head-padded row); bias, when
// given, is added per column.
template <bool A_KMAJOR, bool B_KMAJOR, int BM, int BN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float *__restrict__ A, int lda,
                                                       const float *__restrict__ B, int ldb,
                                                       float *__restrict__ C, int ldc, int M,
                                                       int Nn, int K, int k_per_split,
                                                       const AFix fix, const GemmBatch gb) {
    constexpr int MI = BM / 64, NJ = BN / 64;
    constexpr int EA = BM * BK / 256, EB = BN * BK / 256;   // staged elements per thread
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int split = blockIdx.z % gb.nsplit, batch = blockIdx.z / gb.nsplit;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    A += (size_t)batch * gb.sA;
    B += (size_t)batch * gb.sB;
    C += (size_t)batch * gb.sC + (size_t)split * gb.sSplit;
    const float *bias = gb.bias ? gb.bias + (size_t)batch * gb.sBias : nullptr;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Global -> registers for one K-tile (issued one tile ahead of its use), registers ->
    // LDS, MFMA over the tile.
    float asum = 0.f;               // k-major A: sum of the staged column (gb.asum), for d_bias
    float ra[EA], rb[EB], rd[EA];   // rd: d_a_dst of the element's node (AFix), applied at LDS-store
    // time so the fetched tile stays in flight during the MFMA loop
    // AFix, k-major A (d_weight): this thread's column m0 + tid % BM never changes
    int fix_head = 0;
    float fix_att = 0.f, fix_att_tile = 0.f;
    if (A_KMAJOR && fix.da_dst && m0 + tid % BM < M) {
        fix_head = (int)(m0 + tid % BM) / fix.Cp;
        fix_att = fix.att_p[m0 + tid % BM];
    }
    auto fetch = [&](int k0) {
        if (!A_KMAJOR && fix.da_dst) {   // row-major A (d_x): column k0 + tid % BK, once per tile
            const int col = k0 + tid % BK;
            fix_head = col / fix.Cp;
            fix_att = col < kend ? fix.att_p[col] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < EA; ++i) {
            const int idx = tid + i * 256;
            int m, k;
            if (A_KMAJOR) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
            const long long gm = m0 + m;
            const int gk = k0 + k;
            const bool ok = gm < M && gk < kend;
            ra[i] = ok ? (A_KMAJOR ? A[(size_t)gk * lda + gm] : A[(size_t)gm * lda + gk]) : 0.f;
            if (A_KMAJOR) asum += ra[i];
            if (fix.da_dst) {
                const long long node = A_KMAJOR ? gk : gm;   // row of the d_h table
                rd[i] = ok ? fix.da_dst[node * fix.H + fix_head] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int idx = tid + i * 256;
            int n, k;
            if (B_KMAJOR) { n = idx % BN; k = idx / BN; } else { k = idx % BK; n = idx / BK; }
            const int gn = n0 + n;
            const int gk2 = k0 + k;
            const bool okb = gn < Nn && gk2 < kend;
            rb[i] = okb ? (B_KMAJOR ? B[(size_t)gk2 * ldb + gn] : B[(size_t)gn * ldb + gk2]) : 0.f;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        fix_att_tile = fix_att;   // att of the tile now in registers (fetch() overwrites fix_att)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < EA; ++i) {
            const int idx = tid + i * 256;
            int m, k;
            if (A_KMAJOR) { m = idx % BM; k = idx / BM; } else { k = idx % BK; m = idx / BK; }
            As[k][m] = fix.da_dst ? fmaf(rd[i], fix_att_tile, ra[i]) : ra[i];
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
            const int idx = tid + i * 256;
            int n, k;
            if (B_KMAJOR) { n = idx % BN; k = idx / BN; } else { k = idx % BK; n = idx / BK; }
            Bs[k][n] = rb[i];
        }
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);   // in flight while this tile is multiplied
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5);
            float a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = As[kr][wm * (BM / 2) + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = Bs[kr][wn * (BN / 2) + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    if (A_KMAJOR && gb.asum && blockIdx.y == 0) {
        // column sums of this block's A slice: thread tid staged column tid % BM only
        __shared__ float red[256];
        red[tid] = asum;
        __syncthreads();
        if (tid < BM && m0 + tid < M) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 256 / BM; ++q) s += red[tid + q * BM];
            gb.asum[(size_t)blockIdx.z * M + m0 + tid] = s;
        }
    }
    // C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cn = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
            const float bv = (bias && cn < Nn) ? bias[cn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long cm =
                    m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (cm < M && cn < gb.n_store) C[(size_t)cm * ldc + cn] = acc[i][j][r] + bv;
            }
        }
}

// ---------------------------------------------------------------------------------
// The same product for the shapes the hidden layers have (tile 128 x 128, operands with
// 16-byte aligned rows): float4 global loads along the contiguous dimension, two LDS buffers
// so the next tile is stored while this one is multiplied (one barrier per K-tile instead of
// two), K-tiles of 32, and a flattened grid whose logical order is XCD-contiguous with the
// tiles that share an operand (the n-tiles of an m-tile; the four tiles of a d_W split) next to
// each other, so the shared operand comes out of that XCD's L2.
// Both operand images are written with ds_write_b128 and read conflict-free (see the kernel):
// transposing a k-contiguous source into [k][m] with four scalar stores per float4 cost 10-15 %.
// ---------------------------------------------------------------------------------
constexpr int BK2 = 32;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool A_KMAJOR, bool B_KMAJOR, bool FIX>
__global__ __launch_bounds__(256, 2) void gemm_f32_v2_kernel(const float *__restrict__ A, int lda,
                                                             const float *__restrict__ B, int ldb,
                                                             float *__restrict__ C, int ldc, int M,
                                                             int Nn, int K, int k_per_split,
                                                             const AFix fix, const GemmBatch gb,
                                                             int tiles_m, int tiles_n, int total) {
    constexpr int BM = 128, BN = 128, MI = 2, NJ = 2;
    // LDS images: a source contiguous in m is kept k-major, [k][132]; one contiguous in k is kept
    // as it comes, [m][36] -- both are written with ds_write_b128
    constexpr int SK = 132, SM = 36;
    constexpr int NV = BM * BK2 / 4 / 256;                        // float4 per thread and operand: 4
    constexpr int ASZ = A_KMAJOR ? BK2 * SK : BM * SM, BSZ = B_KMAJOR ? BK2 * SK : BN * SM;
    __shared__ __attribute__((aligned(16))) float As[2][ASZ];
    __shared__ __attribute__((aligned(16))) float Bs[2][BSZ];
    const int L = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (L >= total) return;
    const int tn = L % tiles_n, tm = (L / tiles_n) % tiles_m, bz = L / (tiles_n * tiles_m);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const long long m0 = (long long)tm * BM;
    const int n0 = tn * BN;
    const int split = bz % gb.nsplit, batch = bz / gb.nsplit;
    const int kbeg = split * k_per_split;
    const int kend = min(K, kbeg + k_per_split);
    A += (size_t)batch * gb.sA;
    B += (size_t)batch * gb.sB;
    C += (size_t)batch * gb.sC + (size_t)split * gb.sSplit;
    const float *bias = gb.bias ? gb.bias + (size_t)batch * gb.sBias : nullptr;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging maps: a source contiguous in k -> (row = idx / 8, q = idx % 8: k = 4 q ..);
    //               a source contiguous in m -> (m4 = idx % 32, k = idx / 32)
    f32x4 ra[NV], rb[NV], asum = {0.f, 0.f, 0.f, 0.f};
    float rd[NV];
    f32x4 att_fix[A_KMAJOR ? 1 : NV];
    int fix_head = 0;
    if (A_KMAJOR && FIX) {
        const long long gm = m0 + 4 * (tid & 31);
        att_fix[0] = gm < M ? *reinterpret_cast<const f32x4 *>(fix.att_p + gm) : f32x4{0.f, 0.f, 0.f, 0.f};
        fix_head = gm < M ? (int)gm / fix.Cp : 0;
    }
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if (A_KMAJOR) {
                const long long gm = m0 + 4 * (idx & 31);
                const int gk = k0 + (idx >> 5);
                const bool ok = gm < M && gk < kend;
                ra[i] = ok ? *reinterpret_cast<const f32x4 *>(A + (size_t)gk * lda + gm) : f32x4{0.f, 0.f, 0.f, 0.f};
                asum += ra[i];
                if (FIX) rd[i] = ok ? fix.da_dst[(size_t)gk * fix.H + fix_head] : 0.f;
            } else {
                const long long gm = m0 + (idx >> 3);
                const int gk = k0 + 4 * (idx & 7);
                const bool ok = gm < M && gk < kend;
                ra[i] = ok ? *reinterpret_cast<const f32x4 *>(A + (size_t)gm * lda + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
                if (FIX) {
                    rd[i] = ok ? fix.da_dst[(size_t)gm * fix.H + gk / fix.Cp] : 0.f;
                    att_fix[A_KMAJOR ? 0 : i] =
                        ok ? *reinterpret_cast<const f32x4 *>(fix.att_p + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if (B_KMAJOR) {
                const int gn = n0 + 4 * (idx & 31);
                const int gk = k0 + (idx >> 5);
                const bool ok = gn < Nn && gk < kend;
                rb[i] = ok ? *reinterpret_cast<const f32x4 *>(B + (size_t)gk * ldb + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                const int gn = n0 + (idx >> 3);
                const int gk = k0 + 4 * (idx & 7);
                const bool ok = gn < Nn && gk < kend;
                rb[i] = ok ? *reinterpret_cast<const f32x4 *>(B + (size_t)gn * ldb + gk) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    // Whole K-tiles take this path: no bounds tests (a row or column past the edge is clamped to
    // the last valid one -- what it produces lands in rows that are never stored and in columns
    // the epilogue writes as zeros), a block-uniform base that advances per tile and per-thread
    // byte offsets fixed for the kernel.  The tests and 64-bit address arithmetic of fetch()
    // were ~150 instructions per tile and wave; with every block of the grid in the same phase
    // nothing hides them.
    unsigned offA[NV], offB[NV], offD[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + i * 256;
        if (A_KMAJOR) {
            const long long gm = min(m0 + 4 * (idx & 31), (long long)M - 4);
            offA[i] = (unsigned)(((size_t)(idx >> 5) * lda + gm) * 4);
            offD[i] = (unsigned)(((idx >> 5) * fix.H + fix_head) * 4);
        } else {
            const long long r = min(m0 + (idx >> 3), (long long)M - 1) - m0;
            offA[i] = (unsigned)(((size_t)r * lda + 4 * (idx & 7)) * 4);
            offD[i] = (unsigned)(r * fix.H * 4);
        }
        if (B_KMAJOR) {
            const int gn = min(n0 + 4 * (idx & 31), Nn - 4);
            offB[i] = (unsigned)(((size_t)(idx >> 5) * ldb + gn) * 4);
        } else {
            const int r = min(n0 + (idx >> 3), Nn - 1);           // absolute: n0 may lie in the zero pad
            offB[i] = (unsigned)(((size_t)r * ldb + 4 * (idx & 7)) * 4);
        }
    }
    auto fetch_whole = [&](int k0) {
        const char *pa = reinterpret_cast<const char *>(A_KMAJOR ? A + (size_t)k0 * lda : A + (size_t)m0 * lda + k0);
        const char *pb = reinterpret_cast<const char *>(B_KMAJOR ? B + (size_t)k0 * ldb : B + k0);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            ra[i] = *reinterpret_cast<const f32x4 *>(pa + offA[i]);
            if (A_KMAJOR) asum += ra[i];
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) rb[i] = *reinterpret_cast<const f32x4 *>(pb + offB[i]);
        if (FIX) {
            if (A_KMAJOR) {
                const char *pd = reinterpret_cast<const char *>(fix.da_dst + (size_t)k0 * fix.H);
#pragma unroll
                for (int i = 0; i < NV; ++i) rd[i] = *reinterpret_cast<const float *>(pd + offD[i]);
            } else {
                const int gk = k0 + 4 * (tid & 7);                // the same four columns for every i
                const char *pd = reinterpret_cast<const char *>(fix.da_dst + (size_t)m0 * fix.H);
                const unsigned hoff = (unsigned)(gk / fix.Cp) * 4u;
                const f32x4 att = *reinterpret_cast<const f32x4 *>(fix.att_p + gk);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    rd[i] = *reinterpret_cast<const float *>(pd + offD[i] + hoff);
                    att_fix[A_KMAJOR ? 0 : i] = att;
                }
            }
        }
    };
    auto stage = [&](int buf) {
        float *as = As[buf], *bs = Bs[buf];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            f32x4 v = ra[i];
            if (FIX) v += rd[i] * att_fix[A_KMAJOR ? 0 : i];
            if (A_KMAJOR) *reinterpret_cast<f32x4 *>(as + (idx >> 5) * SK + 4 * (idx & 31)) = v;
            else *reinterpret_cast<f32x4 *>(as + (idx >> 3) * SM + 4 * (idx & 7)) = v;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + i * 256;
            if (B_KMAJOR) *reinterpret_cast<f32x4 *>(bs + (idx >> 5) * SK + 4 * (idx & 31)) = rb[i];
            else *reinterpret_cast<f32x4 *>(bs + (idx >> 3) * SM + 4 * (idx & 7)) = rb[i];
        }
    };
    auto compute = [&](int buf) {
        // Contraction order: step (j, i), j < 8, i < 4, multiplies k = 8 j + i in lane half 0 and
        // k = 8 j + 4 + i in half 1 -- a [m][k] image hands a lane its four k of a j as ONE
        // ds_read_b128 (chunk 2 j + half), a [k][m] image as four rows.  The operands of j + 1 are
        // requested before the 16 MFMAs of j issue and waited for after them (inline asm and
        // sched_barrier pin that order: hipcc sinks ordinary reads to their use).
        const int half = lane >> 5, l31 = lane & 31;
        const unsigned a_addr =
            (unsigned)(uintptr_t)((const __attribute__((address_space(3))) float *)As[buf]) +
            4u * (A_KMAJOR ? (4 * half) * SK + wm * 64 + l31 : (wm * 64 + l31) * SM + 4 * half);
        const unsigned b_addr =
            (unsigned)(uintptr_t)((const __attribute__((address_space(3))) float *)Bs[buf]) +
            4u * (B_KMAJOR ? (4 * half) * SK + wn * 64 + l31 : (wn * 64 + l31) * SM + 4 * half);
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        // the asm outputs themselves are what the waits below are tied to: nothing may touch
        // them in between (a repack here would be scheduled before the data has landed)
        struct Ops { f32x4 v[2]; f32x2 r[4]; };                   // [m][k] image: v[tile][i]; [k][m]: r[i][tile]
        auto ldop = [&](Ops &o, unsigned base, bool kmajor, int j) {
            if (kmajor) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    asm volatile("ds_read2_b32 %0, %1 offset1:32" : "=v"(o.r[i]) : "v"(base + (8 * j + i) * (SK * 4)));
            } else {
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    asm volatile("ds_read_b128 %0, %1" : "=v"(o.v[t]) : "v"(base + t * (32 * SM * 4) + j * 32));
            }
        };
        auto mfma16 = [&](const Ops &a, const Ops &b) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int nj = 0; nj < NJ; ++nj)
                        acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            A_KMAJOR ? a.r[i][mi] : a.v[mi][i], B_KMAJOR ? b.r[i][nj] : b.v[nj][i], acc[mi][nj], 0, 0, 0);
        };
        auto wait_ops = [&](Ops &a, Ops &b) {
            if (A_KMAJOR && B_KMAJOR)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.r[0]), "+v"(a.r[1]), "+v"(a.r[2]), "+v"(a.r[3]),
                             "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]));
            else if (B_KMAJOR)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]),
                             "+v"(b.r[0]), "+v"(b.r[1]), "+v"(b.r[2]), "+v"(b.r[3]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(b.v[0]), "+v"(b.v[1]));
        };
        Ops a0, b0, a1, b1;
        ldop(a0, a_addr, A_KMAJOR, 0);
        ldop(b0, b_addr, B_KMAJOR, 0);
        wait_ops(a0, b0);
#pragma unroll
        for (int j = 0; j < BK2 / 8; j += 2) {
            ldop(a1, a_addr, A_KMAJOR, j + 1);
            ldop(b1, b_addr, B_KMAJOR, j + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            wait_ops(a1, b1);
            if (j + 2 < BK2 / 8) {
                ldop(a0, a_addr, A_KMAJOR, j + 2);
                ldop(b0, b_addr, B_KMAJOR, j + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma16(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            if (j + 2 < BK2 / 8)
                wait_ops(a0, b0);
        }
    };
    // Whole tiles run in a loop of their own and a last partial tile (K not a multiple of 32)
    // after it: with both fetch paths inside one loop their results meet in copies at the join,
    // and hipcc waits for the loads there instead of after the MFMAs.
    const int nwhole = kbeg < kend ? (kend - kbeg) / BK2 : 0;
    const bool part = kbeg < kend && (kend - kbeg) % BK2 != 0;
    int buf = 0;
    if (nwhole > 0) {
        fetch_whole(kbeg);
        stage(0);
    } else if (part) {
        fetch(kbeg);
        stage(0);
    }
    __syncthreads();
    for (int t = 1; t < nwhole; ++t, buf ^= 1) {
        fetch_whole(kbeg + t * BK2);                              // in flight during the MFMAs
        compute(buf);
        stage(buf ^ 1);                                           // the buffer read one tile ago
        __syncthreads();
    }
    if (nwhole > 0 && part) {
        fetch(kbeg + nwhole * BK2);
        compute(buf);
        stage(buf ^ 1);
        __syncthreads();
        compute(buf ^ 1);
    } else if (nwhole > 0 || part) {
        compute(buf);
    }
    if (A_KMAJOR && gb.asum && tn == 0) {
        // column sums of this block's A slice: thread tid staged columns 4 (tid % 32) .. + 3
        __shared__ f32x4 red[256];
        red[tid] = asum;
        __syncthreads();
        if (tid < BM && m0 + tid < M) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += red[(tid >> 2) + 32 * q][tid & 3];
            gb.asum[(size_t)bz * M + m0 + tid] = s;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cn = n0 + wn * 64 + j * 32 + (lane & 31);
            const float bv = (bias && cn < Nn) ? bias[cn] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long cm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (cm < M && cn < gb.n_store) C[(size_t)cm * ldc + cn] = cn < Nn ? acc[i][j][r] + bv : 0.f;
            }
        }
}

// Wp[(h*Cp + c), f] = c < C ? W[(h*C + c), f] : 0
__global__ void pad_weight_kernel(const float *w, float *wp, int H, int C, int Cp, int F) {
    const long long tot = (long long)H * Cp * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot;
         i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int r = (int)(i / F);
        const int h = r / Cp, c = r % Cp;
        wp[i] = c < C ? w[((size_t)h * C + c) * F + f] : 0.f;
    }
}

// d_weight[(h*C + c), f] = sum_z slab[z][(h*Cp + c), f]   (fixed order: reproducible)
__global__ void reduce_dweight_kernel(const float *slab, int nsplit, int H, int C, int Cp, int F,
                                      float *dw) {
    const long long tot = (long long)H * C * F;
    const size_t slab_elems = (size_t)H * Cp * F;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot;
         i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F);
        const int r = (int)(i / F);
        const int h = r / C, c = r % C;
        const size_t src = ((size_t)h * Cp + c) * F + f;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // 4 loads in flight, fixed order
        int z = 0;
        for (; z + 3 < nsplit; z += 4) {
            s0 += slab[(size_t)z * slab_elems + src];
            s1 += slab[(size_t)(z + 1) * slab_elems + src];
            s2 += slab[(size_t)(z + 2) * slab_elems + src];
            s3 += slab[(size_t)(z + 3) * slab_elems + src];
        }
        for (; z < nsplit; ++z) s0 += slab[(size_t)z * slab_elems + src];
        dw[i] = (s0 + s1) + (s2 + s3);
    }
}

// a_dst[n,h] = <hp[n,h,:], att_dst[h,:]>.  A block covers R = 256 / (Cp/4) (node, head)
// groups; every thread moves one float4.
__global__ __launch_bounds__(256) void row_dots_kernel(const float *hp, const float *att,
                                                       float *a_dst, long long total_groups,
                                                       int H, int C, int Cp) {
    __shared__ float part[256];
    const int lph = Cp >> 2;
    const int R = 256 / lph;
    const long long g0 = (long long)blockIdx.x * R;
    const int t = threadIdx.x, gl = t / lph, sub = t % lph;
    const long long grp = g0 + gl;
    float p = 0.f;
    if (gl < R && grp < total_groups) {
        const int head = (int)(grp % H);
        const float4 v = *reinterpret_cast<const float4 *>(hp + (size_t)grp * Cp + sub * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = sub * 4 + k;
            if (c < C) p = fmaf(vv[k], att[head * C + c], p);
        }
    }
    part[t] = p;
    __syncthreads();
    if (t < R && g0 + t < total_groups) {
        float s = 0.f;
        for (int q = 0; q < lph; ++q) s += part[t * lph + q];
        a_dst[g0 + t] = s;
    }
}

// row_dots for power-of-two lanes per head: flat float4 stream, shuffle butterfly.
__global__ __launch_bounds__(256) void row_dots_flat_kernel(const float *hp, const float *att_p,
                                                            float *a_dst, long long nq, int H,
                                                            int lph_log, int Cp) {
    const int lph = 1 << lph_log;
    const long long stride = (long long)gridDim.x * 256;
    constexpr int UN = 8;
    for (long long q0 = (long long)blockIdx.x * 256 + threadIdx.x; q0 - threadIdx.x % 64 < nq;
         q0 += stride * UN) {
        float4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long q = q0 + u * stride;
            v[u] = q < nq ? *reinterpret_cast<const float4 *>(hp + q * 4)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long q = q0 + u * stride;
            const long long grp = q >> lph_log;
            // (a 64-bit remainder is some forty instructions: 32-bit whenever the index allows)
            const int head = nq < (1ll << 31) ? (int)((unsigned)grp % (unsigned)H) : (int)(grp % H);
            const int sub = (int)(q & (lph - 1));
            const float4 a4 = *reinterpret_cast<const float4 *>(att_p + head * Cp + sub * 4);
            float p = fmaf(v[u].w, a4.w, fmaf(v[u].z, a4.z, fmaf(v[u].y, a4.y, v[u].x * a4.x)));
            for (int w = 1; w < lph; w <<= 1) p += __shfl_xor(p, w, 64);
            if (q < nq && sub == 0) a_dst[grp] = p;
        }
    }
}

// ---------------------------------------------------------------------------------
// K-resident GEMM for a tall row-major A and a small B (K <= 128, N <= 128 per batch): the
// out = Z W^T and q = g W products of the input-side aggregation path, one batch per head.
// The tiled kernel above spends half its wave cycles parked at waitcnt/barriers on these
// shapes (K is 4-7 of its k-tiles: every block is mostly prologue and epilogue).  Here
//   * B (zero padded to [Kp][NT*32]) sits in LDS for the lifetime of a persistent block;
//   * every wave owns tiles of 32 rows and stages them itself, global -> LDS by LDS-DMA
//     (no VGPRs), two tile buffers per wave, the next tile in flight while this one is
//     multiplied -- waves never meet at a barrier after the B fill;
//   * A operands are read with ds_read_b128: lane (l, half) takes chunk 2j + half of row l, and
//     the contraction order is permuted to match (k = 8j + 4 half + i in step (j, i)).  A row
//     stride of nf4 * 16 bytes is conflict-free for odd nf4; nf4 == 16 uses an XOR swizzle
//     applied on the SOURCE address of the DMA (its LDS image is lane-linear).
// ---------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void kres_dma16(const void *g, void *lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_uniform, 16,
                                     0, 0);
}
template <int N> __device__ __forceinline__ void kres_wait_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void kres_wait_vm(int n) {   // n wave-uniform, <= 32 here
    switch (n) {
        case 0: kres_wait_imm<0>(); break;
        case 1: kres_wait_imm<1>(); break;
        case 2: kres_wait_imm<2>(); break;
        case 3: kres_wait_imm<3>(); break;
        case 4: kres_wait_imm<4>(); break;
        case 5: kres_wait_imm<5>(); break;
        case 6: kres_wait_imm<6>(); break;
        case 7: kres_wait_imm<7>(); break;
        case 8: kres_wait_imm<8>(); break;
        case 9: kres_wait_imm<9>(); break;
        case 10: kres_wait_imm<10>(); break;
        case 11: kres_wait_imm<11>(); break;
        case 12: kres_wait_imm<12>(); break;
        case 13: kres_wait_imm<13>(); break;
        case 14: kres_wait_imm<14>(); break;
        case 15: kres_wait_imm<15>(); break;
        case 16: kres_wait_imm<16>(); break;
        case 32: kres_wait_imm<32>(); break;
        default: kres_wait_imm<0>(); break;
    }
}

template <int NT, bool SWZ>
__global__ __launch_bounds__(256, 1) void gemm_kres_kernel(const float *__restrict__ A, int lda,
                                                           const float *__restrict__ B, int ldb,
                                                           int b_kmajor, float *__restrict__ C,
                                                           int ldc, int M, int Nn, int K, int nf4,
                                                           const GemmBatch gb) {
    constexpr int LDB = NT * 32;
    // B in a static array, the A tile buffers in the dynamic one: hipcc makes every LDS read that
    // may alias an LDS-DMA in flight wait for it (vmcnt(0) inside the MFMA loop: no prefetch left);
    // two distinct objects let it see that the B reads do not.  The A reads go through inline
    // asm for the same reason -- their ordering against the DMA is the counted wait below.
    __shared__ float Bl[128 * LDB];
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Kp = (nf4 * 4 + 7) & ~7;
    const int batch = blockIdx.y;
    A += (size_t)batch * gb.sA;
    B += (size_t)batch * gb.sB;
    C += (size_t)batch * gb.sC;
    const float *bias = gb.bias ? gb.bias + (size_t)batch * gb.sBias : nullptr;
    for (int idx = threadIdx.x; idx < Kp * LDB; idx += 256) {
        const int k = idx / LDB, n = idx % LDB;
        float v = 0.f;
        if (k < K && n < Nn) v = b_kmajor ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
        Bl[idx] = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    const unsigned rowb = (unsigned)nf4 * 16u, tileb = 32u * rowb;
    char *abuf = smem + (size_t)wave * 2 * tileb;
    const int rpi = 64 / nf4;                   // rows per DMA instruction
    const int ninstr = (32 + rpi - 1) / rpi;
    const int er = lane / nf4, ep = lane - er * nf4;
    const bool dma_on = er < rpi;
    const long long ntiles = ((long long)M + 31) / 32;
    const long long tstride = (long long)gridDim.x * 4;
    auto issue = [&](long long tile, int buf) {
        char *dst = abuf + buf * tileb;
        for (int i = 0; i < ninstr; ++i) {
            const int rl = i * rpi + er;                        // row inside the tile
            long long row = tile * 32 + (rl < 32 ? rl : 31);
            row = row < M ? row : (long long)M - 1;
            const int piece = SWZ ? (ep ^ (rl & 15)) : ep;
            const float *g = A + row * lda + piece * 4;
            if (dma_on && rl < 32) kres_dma16(g, dst + (unsigned)(i * rpi) * rowb);
        }
    };
    const float *bl = Bl + (4 * half) * LDB + l;
    const int KC = (nf4 + 1) >> 1;
    // bias of this lane's columns, fetched (and waited for) before any DMA is in flight: the use
    // of an ordinary load's result later on would make hipcc drain the DMA queue with vmcnt(0)
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cn = nt * 32 + l;
        bv[nt] = (bias && cn < Nn) ? bias[cn] : 0.f;
        asm volatile("" : "+v"(bv[nt]));
    }
    // Order of a wave's vector-memory operations: ... stores(t-1), DMA(t+1) | wait(t), MFMA(t),
    // stores(t), DMA(t+2) ...  vmcnt counts loads and stores alike; loads retire in order, so
    // "at most ninstr outstanding" = tile t has landed.  The stores younger than DMA(t) were
    // issued a whole tile ago, so the wait does not stall on them (it does when a tile's DMA is
    // issued right after the previous tile's stores: 6.4 instead of 3.1 ms for q = g W).
    long long t = (long long)blockIdx.x * 4 + wave;
    int buf = 0;
    if (t < ntiles) issue(t, 0);
    if (t + tstride < ntiles) issue(t + tstride, 1);
    for (; t < ntiles; t += tstride, buf ^= 1) {
        kres_wait_vm(t + tstride < ntiles ? ninstr : 0);
        const char *at = abuf + buf * tileb + (unsigned)l * rowb;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        int boff = 0;                            // keep the B reads inside the tile loop: an opaque
        asm volatile("" : "+v"(boff));           // OFFSET (laundering the pointer itself makes it
        const float *blt = bl + boff;            // generic: flat loads that wait on vmcnt too)

        // software pipeline over the contraction steps: the operands of step j + 1 (one
        // ds_read_b128 of A, 4 NT floats of B) are requested before the MFMAs of step j issue
        auto a_read = [&](int j) {
            const int c = 2 * j + half;
            const int cc = c < nf4 ? c : 0;       // the pad chunk multiplies zero rows of B
            const unsigned addr = (unsigned)(uintptr_t)(
                (const __attribute__((address_space(3))) char *)(at + (SWZ ? (cc ^ (l & 15)) : cc) * 16));
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
            return v;
        };
        f32x4 a4 = a_read(0);
        float b4[4][NT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) b4[i][nt] = blt[i * LDB + nt * 32];
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4));
        for (int j = 0; j < KC; ++j) {
            const int jn = j + 1 < KC ? j + 1 : j;
            f32x4 an = a_read(jn);
            float bn[4][NT];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bn[i][nt] = blt[(8 * jn + i) * LDB + nt * 32];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], b4[i][nt], acc[nt], 0, 0, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an));
            a4 = an;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) b4[i][nt] = bn[i][nt];
        }
        const long long row0 = t * 32;
        const bool full = row0 + 32 <= M;        // one branch per tile, not one per store
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int cn = nt * 32 + l;
            if (cn < gb.n_store) {
                float *cp = C + (size_t)(row0 + 4 * half) * ldc + cn;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (full || row0 + 4 * half + dr < M) cp[(size_t)dr * ldc] = acc[nt][r] + bv[nt];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this buffer's reads have landed
        if (t + 2 * tstride < ntiles) issue(t + 2 * tstride, buf);
    }
}

// ---------------------------------------------------------------------------------
// The same product with B in REGISTERS.  The LDS kernel above is latency bound: 39 % of its wave
// cycles sit at the tile wait (profiles/r1/pmc_gemm_waits.txt) because B takes 32 KB of LDS and
// the rest holds only two tiles per wave -- one in flight while the other is multiplied, ~50 KB
// per CU, a quarter of what a 7-TB/s stream needs.  One wave per SIMD owns 512 registers and uses
// 100: here each lane keeps its B operands (k = 8j + 4 half + i, column nt*32 + l: KC*4*NT
// floats, at most 128) for the lifetime of the block, the contraction is unrolled over KC
// steps, and all 160 KB of LDS are tile buffers: NBUF = 3-4 per wave, two or three in flight.
// ---------------------------------------------------------------------------------
template <int NT, int KC, bool SWZ, bool TAIL, bool DBUF = false>
__global__ __launch_bounds__(256, 1) void gemm_kreg_kernel(const float *__restrict__ A, int lda,
                                                           const float *__restrict__ B, int ldb,
                                                           int b_kmajor, float *__restrict__ C,
                                                           int ldc, int M, int Nn, int K, int nf4,
                                                           int nbuf, const GemmBatch gb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int batch = blockIdx.y;
    A += (size_t)batch * gb.sA;
    B += (size_t)batch * gb.sB;
    C += (size_t)batch * gb.sC;
    const float *bias = gb.bias ? gb.bias + (size_t)batch * gb.sBias : nullptr;
    const int lane = threadIdx.x & 63, l = lane & 31, half = lane >> 5, wave = threadIdx.x >> 6;
    // operands and bias are fetched (and waited for: the empty asm uses them) before any DMA is
    // in flight -- a later first use would make hipcc drain the DMA queue with vmcnt(0)
    float breg[KC][4][NT], bv[NT];
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int k = 8 * j + 4 * half + i, n = nt * 32 + l;
                float v = 0.f;
                if (k < K && n < Nn) v = b_kmajor ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k];
                breg[j][i][nt] = v;
            }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int cn = nt * 32 + l;
        bv[nt] = (bias && cn < Nn) ? bias[cn] : 0.f;
    }
    // TAIL: up to four more columns (NT*32 ...) go through v_mfma_f32_4x4x1_16B_f32 -- 16
    // independent 4x4 blocks, lane 4b + i supplies the A of row i of block b, lane 4b + j the B
    // of column j.  Blocks 0-7 are the tile's 32 rows on the k values of half 0, blocks 8-15 the
    // same rows on those of half 1 -- exactly what a4[i] already holds; the halves are added in
    // the epilogue.  8 cycles per step instead of the 64 of a fourth 32-wide tile.
    float btail[TAIL ? KC : 1][4], bvt = 0.f;
    if constexpr (TAIL) {
        const int n = NT * 32 + (lane & 3);
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kk = 8 * j + 4 * half + i;
                float v = 0.f;
                if (kk < K && n < Nn) v = b_kmajor ? B[(size_t)kk * ldb + n] : B[(size_t)n * ldb + kk];
                btail[j][i] = v;
            }
        bvt = (bias && n < Nn) ? bias[n] : 0.f;
#pragma unroll
        for (int j = 0; j < KC; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(btail[j][i]));
        asm volatile("" : "+v"(bvt));
    }
#pragma unroll
    for (int j = 0; j < KC; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(breg[j][i][nt]));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) asm volatile("" : "+v"(bv[nt]));

    const unsigned rowb = (unsigned)nf4 * 16u, tileb = 32u * rowb;
    char *abuf = smem + (size_t)wave * nbuf * tileb;
    const int rpi = 64 / nf4;                   // rows per DMA instruction
    const int ninstr = (32 + rpi - 1) / rpi;
    const int er = lane / nf4, ep = lane - er * nf4;
    const bool dma_on = er < rpi;
    const long long ntiles = ((long long)M + 31) / 32;
    const long long tstride = (long long)gridDim.x * 4;
    // DMA source of instruction i = (wave-uniform tile base + i * rows-per-instruction) + a 32-bit
    // lane offset: scalar address arithmetic, no 64-bit vector multiplies in the tile loop.  Only
    // the last, partial tile clamps its rows.
    const unsigned lane_off = ((unsigned)er * (unsigned)lda + (unsigned)ep * 4u) * 4u;
    const unsigned step_off = (unsigned)rpi * (unsigned)lda * 4u;
    auto issue = [&](long long tile, int buf) {
        char *dst = abuf + buf * tileb;
        const char *src = reinterpret_cast<const char *>(A + tile * 32 * lda);   // uniform
        const bool whole = __builtin_amdgcn_readfirstlane((int)(tile * 32 + 32 <= M)) != 0;
        if (whole && (32 % rpi) == 0) {
            for (int i = 0; i < ninstr; ++i) {
                unsigned off = lane_off;
                if (SWZ) off = ((unsigned)er * (unsigned)lda + (unsigned)((ep ^ ((i * rpi + er) & 15)) * 4)) * 4u;
                if (dma_on) kres_dma16(src + (size_t)i * step_off + off, dst + (unsigned)(i * rpi) * rowb);
            }
        } else {
            for (int i = 0; i < ninstr; ++i) {
                const int rl = i * rpi + er;                        // row inside the tile
                long long row = tile * 32 + (rl < 32 ? rl : 31);
                row = row < M ? row : (long long)M - 1;
                const int piece = SWZ ? (ep ^ (rl & 15)) : ep;
                const float *g = A + row * lda + piece * 4;
                if (dma_on && rl < 32) kres_dma16(g, dst + (unsigned)(i * rpi) * rowb);
            }
        }
    };
    // A wave's vector-memory operations: ... stores(t-1), DMA(t+nbuf-1) | wait(t), MFMA(t),
    // stores(t), DMA(t+nbuf) ...  Loads retire in order, so "at most (tiles issued after t) *
    // ninstr operations outstanding" means tile t has landed whatever the stores do.
    long long t = (long long)blockIdx.x * 4 + wave;
    for (int b = 0; b < nbuf; ++b)
        if (t + b * tstride < ntiles) issue(t + b * tstride, b);
    int buf = 0;
    long long after = t < ntiles ? (ntiles - 1 - t) / tstride : 0;   // this wave's tiles after t
    if constexpr (DBUF) {
        // Two accumulator sets: the stores of tile t-1 are issued in slices BETWEEN the MFMA groups
        // of tile t (a lone wave per SIMD issues in order: with one set, a tile's 52 stores and
        // their address arithmetic sit between its last MFMA and the next tile's first).  Only the
        // wave's last tile can be the matrix's partial one, and that one is flushed after the loop
        // with the predicated form; every tile stored inside the loop is whole.
        f32x16 accs[2][NT];
        f32x4 accts[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        constexpr int kStores = NT * 16;
        constexpr int kSlice = (kStores + KC - 1) / KC;
        const unsigned ldcb = (unsigned)ldc * 4u;
        auto store_one = [&](const f32x16(&pa)[NT], char *cb, int s_) {
            const int nt = s_ / 16, r = s_ % 16;
            const int cn = nt * 32 + l;
            const unsigned lo = (unsigned)(4 * half) * ldcb + (unsigned)cn * 4u;
            const unsigned dr = (r & 3) + 8 * (r >> 2);
            if (cn < gb.n_store) *reinterpret_cast<float *>(cb + (size_t)(dr * ldcb) + lo) = pa[nt][r] + bv[nt];
        };
        auto store_tail = [&](const f32x4 &pt, long long row0) {
            if constexpr (TAIL) {
                const int cn = NT * 32 + (lane & 3);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float tot = pt[i] + __shfl_xor(pt[i], 32, 64);
                    const long long row = row0 + 4 * (l >> 2) + i;
                    if (half == 0 && cn < gb.n_store && row < M) C[(size_t)row * ldc + cn] = tot + bvt;
                }
            }
        };
        auto body = [&](f32x16(&acc)[NT], f32x4 &acct, const f32x16(&pa)[NT], const f32x4 &pt, bool have_prev,
                        long long prev_row0) {
            kres_wait_vm((int)(after < nbuf - 1 ? after : nbuf - 1) * ninstr);
            const char *at = abuf + buf * tileb + (unsigned)l * rowb;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
            acct = f32x4{0.f, 0.f, 0.f, 0.f};
            auto a_read = [&](int j) {
                const int c = 2 * j + half;
                const int cc = c < nf4 ? c : 0;
                const unsigned addr = (unsigned)(uintptr_t)(
                    (const __attribute__((address_space(3))) char *)(at + (SWZ ? (cc ^ (l & 15)) : cc) * 16));
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
                return v;
            };
            char *pcb = reinterpret_cast<char *>(C + prev_row0 * ldc);     // wave-uniform
            f32x4 a4 = a_read(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4));
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                f32x4 an = a_read(j + 1 < KC ? j + 1 : j);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], breg[j][i][nt], acc[nt], 0, 0, 0);
                if constexpr (TAIL) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        acct = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[i], btail[j][i], acct, 0, 0, 0);
                }
                if (have_prev) {                       // wave-uniform: a slice of tile t-1's stores
#pragma unroll
                    for (int s_ = j * kSlice; s_ < (j + 1) * kSlice && s_ < kStores; ++s_) store_one(pa, pcb, s_);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an));
                a4 = an;
                __builtin_amdgcn_sched_barrier(0);     // keep each slice between its MFMA groups
            }
            if (have_prev) store_tail(pt, prev_row0);
            if (t + (long long)nbuf * tstride < ntiles) issue(t + (long long)nbuf * tstride, buf);
            buf = buf + 1 < nbuf ? buf + 1 : 0;
        };
        bool have_prev = false;
        long long prev_row0 = 0;
        int par = 0;
        for (; t < ntiles; t += tstride, --after) {
            if (par == 0) body(accs[0], accts[0], accs[1], accts[1], have_prev, prev_row0);
            else body(accs[1], accts[1], accs[0], accts[0], have_prev, prev_row0);
            have_prev = true;
            prev_row0 = t * 32;
            par ^= 1;
        }
        if (have_prev) {                               // the wave's last tile (maybe the partial one)
            const int cur = par ^ 1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cn = nt * 32 + l;
                if (cn < gb.n_store) {
                    float *cp = C + (size_t)(prev_row0 + 4 * half) * ldc + cn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        const float v = cur == 0 ? accs[0][nt][r] : accs[1][nt][r];
                        if (prev_row0 + 4 * half + dr < M) cp[(size_t)dr * ldc] = v + bv[nt];
                    }
                }
            }
            store_tail(cur == 0 ? accts[0] : accts[1], prev_row0);
        }
        return;
    }
    for (; t < ntiles; t += tstride, --after) {
        kres_wait_vm((int)(after < nbuf - 1 ? after : nbuf - 1) * ninstr);
        const char *at = abuf + buf * tileb + (unsigned)l * rowb;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        f32x4 acct = {0.f, 0.f, 0.f, 0.f};
        auto a_read = [&](int j) {
            const int c = 2 * j + half;
            const int cc = c < nf4 ? c : 0;       // a pad chunk meets zero operands of B
            const unsigned addr = (unsigned)(uintptr_t)(
                (const __attribute__((address_space(3))) char *)(at + (SWZ ? (cc ^ (l & 15)) : cc) * 16));
            f32x4 v;
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
            return v;
        };
        f32x4 a4 = a_read(0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a4));
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            f32x4 an = a_read(j + 1 < KC ? j + 1 : j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], breg[j][i][nt], acc[nt], 0, 0, 0);
            if constexpr (TAIL) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acct = __builtin_amdgcn_mfma_f32_4x4x1f32(a4[i], btail[j][i], acct, 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(an));
            a4 = an;
        }
        const long long row0 = t * 32;
        if (__builtin_amdgcn_readfirstlane((int)(row0 + 32 <= M))) {
            // whole tile: uniform row base + 32-bit lane offset, no per-store predicate
            char *cb = reinterpret_cast<char *>(C + row0 * ldc);
            const unsigned ldcb = (unsigned)ldc * 4u;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cn = nt * 32 + l;
                const unsigned lo = (unsigned)(4 * half) * ldcb + (unsigned)cn * 4u;
                if (cn < gb.n_store) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned dr = (r & 3) + 8 * (r >> 2);
                        *reinterpret_cast<float *>(cb + (size_t)(dr * ldcb) + lo) = acc[nt][r] + bv[nt];
                    }
                }
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int cn = nt * 32 + l;
                if (cn < gb.n_store) {
                    float *cp = C + (size_t)(row0 + 4 * half) * ldc + cn;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = (r & 3) + 8 * (r >> 2);
                        if (row0 + 4 * half + dr < M) cp[(size_t)dr * ldc] = acc[nt][r] + bv[nt];
                    }
                }
            }
        }
        if constexpr (TAIL) {
            // lane 4b + j holds rows 4b .. 4b+3 (register index) of column NT*32 + j
            const int cn = NT * 32 + (lane & 3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float tot = acct[i] + __shfl_xor(acct[i], 32, 64);
                const long long row = row0 + 4 * (l >> 2) + i;
                if (half == 0 && cn < gb.n_store && row < M) C[(size_t)row * ldc + cn] = tot + bvt;
            }
        }
        if (t + (long long)nbuf * tstride < ntiles) issue(t + (long long)nbuf * tstride, buf);
        buf = buf + 1 < nbuf ? buf + 1 : 0;
    }
}

static size_t kres_dyn_lds(int nf4) { return (size_t)4 * 2 * 32 * nf4 * 16; }
static size_t kres_lds_bytes(int nf4, int NT) {
    return (size_t)128 * NT * 32 * 4 + kres_dyn_lds(nf4);
}

// true when the K-resident kernel covers the product (else use launch_gemm)
bool kres_covers(const GemmDesc &d) {
    if (option("GATMI_KRES", 1) == 0) return false;   // A/B switch: 0 = tiled kernel
    const int ka = d.a_kpad > d.K ? d.a_kpad : d.K;  // A's zero-padded row length, if longer
    const int nf4 = ka / 4;
    const int nstore = d.n_store > d.Nn ? d.n_store : d.Nn;
    if (d.a_kmajor || d.nsplit > 1 || d.K < 1 || ka > 128 || (ka & 3) || nstore > 128) return false;
    if ((d.lda & 3) || (d.sA & 3) || (reinterpret_cast<uintptr_t>(d.A) & 15) || d.fix.da_dst) return false;
    // conflict-free LDS image: odd stride or swizzle; 128-column rows (nf4 32) only fit the
    // B-in-registers kernel, with two tile buffers per wave and at most 64 output columns
    const bool wide = nf4 == 32 && nstore <= 64 && option("GATMI_KREG", 1) != 0 && option("GATMI_KREG_WIDE", 1) != 0;
    if (!((nf4 & 1) || nf4 == 16 || wide || (d.even_ok && nf4 <= 16))) return false;
    if (d.nbatch > 65535 || d.M < 4096) return false;
    return kres_lds_bytes(nf4, nstore <= 32 ? 1 : (nstore <= 64 ? 2 : 4)) <= 160 * 1024;
}

template <int NT, bool SWZ>
static void kres_launch(const GemmDesc &d, int nf4, hipStream_t st) {
    const size_t lds = kres_dyn_lds(nf4), lds_all = kres_lds_bytes(nf4, NT);
    const int cus = device_cus();
    // dynamic LDS beyond 64 KiB has to be granted per kernel AND per device (the attribute applies
    // to the current device only: a process driving several GPUs needs it on each)
    static size_t dyn_cap[kMaxDevices] = {};
    const int di = device_index();
    if (lds > dyn_cap[di]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kres_kernel<NT, SWZ>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            dyn_cap[di] = lds;
    }
    const int per_cu = (int)((160 * 1024) / lds_all) < 1 ? 1 : (int)((160 * 1024) / lds_all);
    long long bx = ((long long)cus * per_cu + d.nbatch - 1) / d.nbatch;   // one resident set
    const long long need = (((long long)d.M + 31) / 32 + 3) / 4;
    if (bx > need) bx = need;
    if (bx < 1) bx = 1;
    GemmBatch gb{d.sA, d.sB, d.sC, d.sSplit, d.sBias, d.bias, 1, d.n_store, nullptr};
    hipLaunchKernelGGL((gemm_kres_kernel<NT, SWZ>), dim3((unsigned)bx, d.nbatch), dim3(256), lds, st,
                       d.A, d.lda, d.B, d.ldb, d.b_kmajor ? 1 : 0, d.C, d.ldc, d.M, d.Nn, d.K, nf4, gb);
}

int device_index() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
    return dev;
}

int device_cus() {
    static int cus[kMaxDevices] = {};
    const int di = device_index();
    if (cus[di] == 0) {
        int c = 0;
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, di) != hipSuccess || c < 1)
            c = 256;
        cus[di] = c;
    }
    return cus[di];
}

template <int NT, int KC, bool SWZ, bool TAIL = false, bool DBUF = false>
static void kreg_launch(const GemmDesc &d, int nf4, hipStream_t st) {
    if constexpr (!DBUF && NT <= 3) {        // A/B: two accumulator sets (GATMI_KREG_DBUF)
        if (option("GATMI_KREG_DBUF", 0) != 0) return kreg_launch<NT, KC, SWZ, TAIL, true>(d, nf4, st);
    }
    const size_t tileb = (size_t)32 * nf4 * 16;
    int nbuf = (int)((160 * 1024) / (4 * tileb));
    nbuf = nbuf > 4 ? 4 : nbuf;
    const size_t lds = (size_t)4 * nbuf * tileb;
    static size_t dyn_cap[kMaxDevices] = {};   // granted per kernel and per device (see kres_launch)
    const int di = device_index();
    if (lds > dyn_cap[di]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_kreg_kernel<NT, KC, SWZ, TAIL, DBUF>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess)
            dyn_cap[di] = lds;
    }
    long long bx = ((long long)device_cus() + d.nbatch - 1) / d.nbatch;   // one resident set
    const long long need = (((long long)d.M + 31) / 32 + 3) / 4;
    if (bx > need) bx = need;
    if (bx < 1) bx = 1;
    GemmBatch gb{d.sA, d.sB, d.sC, d.sSplit, d.sBias, d.bias, 1, d.n_store, nullptr};
    hipLaunchKernelGGL((gemm_kreg_kernel<NT, KC, SWZ, TAIL, DBUF>), dim3((unsigned)bx, d.nbatch), dim3(256), lds, st,
                       d.A, d.lda, d.B, d.ldb, d.b_kmajor ? 1 : 0, d.C, d.ldc, d.M, d.Nn, d.K, nf4, nbuf, gb);
}

// B-in-registers variant: at most 128 operand registers per lane, at least 3 tile buffers per wave
template <int NT>
static bool kreg_dispatch(const GemmDesc &d, int nf4, hipStream_t st) {
    if (option("GATMI_KREG", 1) == 0) return false;   // A/B switch: 0 = B in LDS
    const int kc = (nf4 + 1) / 2;
    if (nf4 == 32) {   // K = 128: two 16-KiB tile buffers per wave
        if constexpr (NT <= 2) { kreg_launch<NT, 16, true>(d, nf4, st); return true; }
        return false;
    }
    if ((size_t)4 * 3 * 32 * nf4 * 16 > 160 * 1024) return false;
    if (nf4 == 16) { kreg_launch<NT, 8, true>(d, nf4, st); return true; }
    if (kc <= 4) { kreg_launch<NT, 4, false>(d, nf4, st); return true; }
    if (kc <= 8) { kreg_launch<NT, 8, false>(d, nf4, st); return true; }
    if constexpr (NT <= 2) {
        if (kc <= 13) { kreg_launch<NT, 13, false>(d, nf4, st); return true; }
        if (kc <= 16) { kreg_launch<NT, 16, false>(d, nf4, st); return true; }
    }
    return false;
}

static bool kreg_tail3(const GemmDesc &d, int nf4, hipStream_t st) {
    if (option("GATMI_KREG", 1) == 0 || option("GATMI_KREG_TAIL", 1) == 0) return false;
    const int kc = (nf4 + 1) / 2;
    if ((size_t)4 * 3 * 32 * nf4 * 16 > 160 * 1024) return false;
    if (nf4 == 16) { kreg_launch<3, 8, true, true>(d, nf4, st); return true; }
    if (kc <= 4) { kreg_launch<3, 4, false, true>(d, nf4, st); return true; }
    if (kc <= 8) { kreg_launch<3, 8, false, true>(d, nf4, st); return true; }
    return false;
}

int launch_kres(const GemmDesc &d0, hipStream_t st) {
    GemmDesc d = d0;
    if (d.n_store < d.Nn) d.n_store = d.Nn;
    if (d.nbatch < 1) d.nbatch = 1;
    if (d.M == 0 || d.Nn == 0) return GATMI_OK;
    GATMI_REQUIRE(kres_covers(d), "K-resident GEMM does not cover this product");
    const int nf4 = (d.a_kpad > d.K ? d.a_kpad : d.K) / 4;
    const bool swz = nf4 == 16 || nf4 == 32;
    if (d.n_store > 96 && d.n_store <= 100 && kreg_tail3(d, nf4, st)) {
        // three 32-wide tiles + a 4-column tail (F = 100: the products-shaped layer)
    } else if (d.n_store <= 32) {
        if (!kreg_dispatch<1>(d, nf4, st)) swz ? kres_launch<1, true>(d, nf4, st) : kres_launch<1, false>(d, nf4, st);
    } else if (d.n_store <= 64) {
        if (!kreg_dispatch<2>(d, nf4, st)) swz ? kres_launch<2, true>(d, nf4, st) : kres_launch<2, false>(d, nf4, st);
    } else {
        if (!kreg_dispatch<4>(d, nf4, st)) swz ? kres_launch<4, true>(d, nf4, st) : kres_launch<4, false>(d, nf4, st);
    }
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

template <bool AK, bool BK_, int BM, int BN>
static void launch_tile(const GemmDesc &d, int kps, hipStream_t st) {
    dim3 grid((d.M + BM - 1) / BM, (d.n_store + BN - 1) / BN,
              d.nsplit * d.nbatch);
    GemmBatch gb{d.sA, d.sB, d.sC, d.sSplit, d.sBias, d.bias, d.nsplit, d.n_store, d.asum};
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, BM, BN>), grid, dim3(256), 0, st, d.A, d.lda, d.B,
                       d.ldb, d.C, d.ldc, d.M, d.Nn, d.K, kps, d.fix, gb);
}

// gemm_f32_v2_kernel: full-size tiles, every float4 of an operand inside one row and aligned
static bool v2_covers(const GemmDesc &d) {
    if (option("GATMI_GEMM_V2", 1) == 0) return false;
    if (d.M <= 64 || d.n_store <= 64 || d.lda % 4 || d.ldb % 4 || d.sA % 4 || d.sB % 4) return false;
    if (d.Nn < 4 || d.lda > (1 << 20) || d.ldb > (1 << 20)) return false;   // 32-bit offsets inside a tile
    if (!d.b_kmajor && (long long)d.Nn * d.ldb >= (1ll << 29)) return false;   // and over an n-major B
    if ((!d.a_kmajor || !d.b_kmajor) && d.K % 4) return false;   // a float4 along k
    if (((uintptr_t)d.A | (uintptr_t)d.B) & 15) return false;
    if (d.a_kmajor && d.M % 4) return false;
    if (d.b_kmajor && d.Nn % 4) return false;
    if (d.fix.da_dst && (d.fix.Cp % 4 || ((uintptr_t)d.fix.att_p & 15))) return false;
    return true;
}

template <bool AK, bool BK_>
static void launch_v2(const GemmDesc &d, hipStream_t st) {
    const int kps = ((d.K + d.nsplit - 1) / d.nsplit + BK2 - 1) / BK2 * BK2;
    const int tm = (d.M + 127) / 128, tn = (d.n_store + 127) / 128;
    const long long total = (long long)tm * tn * d.nsplit * d.nbatch;
    const int grid = (int)((total + 7) / 8 * 8);
    GemmBatch gb{d.sA, d.sB, d.sC, d.sSplit, d.sBias, d.bias, d.nsplit, d.n_store, d.asum};
    if (d.fix.da_dst)
        hipLaunchKernelGGL((gemm_f32_v2_kernel<AK, BK_, true>), dim3(grid), dim3(256), 0, st, d.A, d.lda,
                           d.B, d.ldb, d.C, d.ldc, d.M, d.Nn, d.K, kps, d.fix, gb, tm, tn, (int)total);
    else
        hipLaunchKernelGGL((gemm_f32_v2_kernel<AK, BK_, false>), dim3(grid), dim3(256), 0, st, d.A, d.lda,
                           d.B, d.ldb, d.C, d.ldc, d.M, d.Nn, d.K, kps, d.fix, gb, tm, tn, (int)total);
}

template <bool AK, bool BK_>
static void launch_layout(const GemmDesc &d, int kps, hipStream_t st) {
    if (v2_covers(d)) {
        launch_v2<AK, BK_>(d, st);
        return;
    }
    const bool m64 = d.M <= 64, n64 = d.n_store <= 64;
    if (m64 && n64) launch_tile<AK, BK_, 64, 64>(d, kps, st);
    else if (m64) launch_tile<AK, BK_, 64, 128>(d, kps, st);
    else if (n64) launch_tile<AK, BK_, 128, 64>(d, kps, st);
    else launch_tile<AK, BK_, 128, 128>(d, kps, st);
}

int launch_gemm(const GemmDesc &d0, hipStream_t st) {
    GemmDesc d = d0;
    if (d.n_store < d.Nn) d.n_store = d.Nn;
    if (d.nbatch < 1) d.nbatch = 1;
    if (d.nsplit < 1) d.nsplit = 1;
    if (d.M == 0 || d.Nn == 0) return GATMI_OK;
    GATMI_REQUIRE((long long)d.nsplit * d.nbatch <= 65535, "gemm: %d splits x %d batches", d.nsplit,
                  d.nbatch);
    const int kps = ((d.K + d.nsplit - 1) / d.nsplit + BK - 1) / BK * BK;
    if (!d.a_kmajor && !d.b_kmajor) launch_layout<false, false>(d, kps, st);
    else if (!d.a_kmajor && d.b_kmajor) launch_layout<false, true>(d, kps, st);
    else if (d.a_kmajor && d.b_kmajor) launch_layout<true, true>(d, kps, st);
    else {
        set_error("gemm: A k-major with B n-major is not built");
        return GATMI_ERR_ARG;
    }
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

static int launch_gemm(bool a_km, bool b_km, const float *A, int lda, const float *B, int ldb,
                       float *C, int ldc, int M, int Nn, int K, int nsplit, hipStream_t st,
                       AFix fix = AFix{nullptr, nullptr, 1, 1}) {
    GemmDesc d{};
    d.a_kmajor = a_km; d.b_kmajor = b_km;
    d.A = A; d.lda = lda; d.B = B; d.ldb = ldb; d.C = C; d.ldc = ldc;
    d.M = M; d.Nn = Nn; d.K = K; d.n_store = Nn; d.nbatch = 1; d.nsplit = nsplit;
    d.sSplit = (long long)M * ldc;
    d.fix = fix;
    return launch_gemm(d, st);
}

// out = sum of nsplit slabs of n4 float4 each (split-K forward of a small graph)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const f32x4 *slab, int nsplit, long long n4,
                                                         f32x4 *out) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
        f32x4 s = slab[i];
        for (int k = 1; k < nsplit; ++k) s += slab[(size_t)k * n4 + i];
        out[i] = s;
    }
}

// h = x W^T of a SMALL graph with many features (Cora: 2708 x 1433 -> 64) is 22 tiles of 90
// serial K-steps on 22 of 256 CUs: cut K so that ~256 blocks run, partial products into the
// slab region of the workspace (sized for d_W), then one summing pass.
static int fwd_splits(int N, int Dp, int F, size_t slab_floats) {
    const long long blocks = ((N + 127) / 128) * (long long)((Dp + 127) / 128);
    if (blocks >= 128 || F < 256 || (size_t)N * Dp % 4) return 1;
    long long want = 256 / blocks;
    if (want > F / 128) want = F / 128;
    const long long fit = (long long)(slab_floats / ((size_t)N * Dp));
    if (want > fit) want = fit;
    return want < 2 ? 1 : (int)want;
}

// ---------------------------------------------------------------------------------
// Narrow last layer: conv3 of /root/reference/gat/model.py:30 projects 8 * hidden features onto
// num_classes channels of one head (2 classes: Dp = 4).  The MFMA tiles ran these products at 2.2 TB/s
// (a 128 x 64 tile for 4 columns); they are streams of x: a row per group of lanes (a float4 of the row
// each), Dp dot products per row reduced by Dp-to-1 trading stages and a butterfly, and in the
// backward ONE pass over x that writes d_x = d_h W and accumulates d_W = d_h^T x in registers
// (reads x once, writes d_x once: the two GEMMs read x and d_h twice and wrote a slab per split).
// Dp = 4 or 8, F % 4 == 0, at least Dp lanes per row, Dp * (float4 per lane) <= 8.
// ---------------------------------------------------------------------------------
#ifndef GATMI_NARROW_UN
#define GATMI_NARROW_UN 2
#endif
// d_weight[(h*C + c), f] = sum_z slab[z][(h*Cp + c), f] for the hundreds of slabs of the streaming
// kernel: 16 outputs x 16 groups of slabs per block, four loads in flight per thread (one thread per
// output walking 1024 slabs took 0.16 ms for conv3's 480 outputs)
__global__ __launch_bounds__(256) void reduce_dweight_wide_kernel(const float *slab, int nsplit, int H, int C, int Cp,
                                                                  int F, float *dw) {
    __shared__ float red[256];
    const long long tot = (long long)H * C * F;
    const size_t slab_elems = (size_t)H * Cp * F;
    const long long i = blockIdx.x * 16ll + (threadIdx.x & 15);
    const int rg = threadIdx.x >> 4;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < tot) {
        const int f = (int)(i % F), r = (int)(i / F), h = r / C, c = r % C;
        const float *src = slab + ((size_t)h * Cp + c) * F + f;
        int z = rg;
        for (; z + 48 < nsplit; z += 64) {
            s0 += src[(size_t)z * slab_elems];
            s1 += src[(size_t)(z + 16) * slab_elems];
            s2 += src[(size_t)(z + 32) * slab_elems];
            s3 += src[(size_t)(z + 48) * slab_elems];
        }
        for (; z < nsplit; z += 16) s0 += src[(size_t)z * slab_elems];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && i < tot) {
        float t = 0.f;
        for (int q = 0; q < 16; ++q) t += red[threadIdx.x + q * 16];
        dw[i] = t;
    }
}

template <int S> __device__ __forceinline__ float narrow_xor(float v) {
    if constexpr (S == 1) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    else if constexpr (S == 2) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    else return __shfl_xor(v, S, 64);
}
// v[d] = this lane's part of dot d; returns the whole dot of index bitrev(lane mod DP) over the gl lanes of the row
template <int DP> __device__ __forceinline__ float narrow_reduce(float (&v)[DP], int lane, int gl) {
    {
        constexpr int n = DP / 2;
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const float keep = up ? v[n + i] : v[i], send = up ? v[i] : v[n + i];
            v[i] = keep + narrow_xor<1>(send);
        }
    }
    {
        constexpr int n = DP / 4;
        const bool up = (lane >> 1) & 1;
#pragma unroll
        for (int i = 0; i < n; ++i) {
            const float keep = up ? v[n + i] : v[i], send = up ? v[i] : v[n + i];
            v[i] = keep + narrow_xor<2>(send);
        }
    }
    if constexpr (DP == 8) {
        const bool up = (lane >> 2) & 1;
        const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
        v[0] = keep + narrow_xor<4>(send);
    }
    float c = v[0];
    for (int o = DP; o < gl; o <<= 1) c += __shfl_xor(c, o, 64);
    return c;
}
template <int DP> __device__ __forceinline__ int narrow_owner(int lane) {   // the dot index lane owns after narrow_reduce
    constexpr int L = DP == 4 ? 2 : 3;
    int d = 0;
#pragma unroll
    for (int b = 0; b < L; ++b) d |= ((lane >> b) & 1) << (L - 1 - b);
    return d;
}

typedef float nf4 __attribute__((ext_vector_type(4)));

// hp[n, d] = <x[n, :], wp[d, :]> (d < DP = H Cp, wp zero in the padded channels), a_dst[n, h] = <hp[n, h, :], att_p[h, :]>
template <int DP, int KS>
__global__ __launch_bounds__(256) void proj_narrow_fwd_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                              const float *__restrict__ att_p, long long n, int F,
                                                              int glog, int H, int Cp, float *__restrict__ hp,
                                                              float *__restrict__ a_dst) {
    constexpr int UN = KS == 1 ? 4 : 2;                           // rows in flight per lane group
    const int lane = threadIdx.x & 63, gl = 1 << glog, sl = lane & (gl - 1), rsub = lane >> glog;
    const int rpw = 64 >> glog, lph = F >> 2;
    nf4 w[DP][KS];
#pragma unroll
    for (int d = 0; d < DP; ++d)
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sub = sl + k * gl;
            w[d][k] = sub < lph ? *reinterpret_cast<const nf4 *>(wp + (size_t)d * F + sub * 4) : nf4{0.f, 0.f, 0.f, 0.f};
        }
    const int dl = narrow_owner<DP>(lane);
    const float attv = att_p ? att_p[dl] : 0.f;
    const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6), nwaves = gridDim.x * 4ll;
    for (long long row0 = wave * rpw * UN; row0 < n; row0 += nwaves * rpw * UN) {
        nf4 xv[UN][KS];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long row = row0 + u * rpw + rsub;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int sub = sl + k * gl;
                xv[u][k] = (row < n && sub < lph) ? *reinterpret_cast<const nf4 *>(x + row * F + sub * 4)
                                                   : nf4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long row = row0 + u * rpw + rsub;
            float v[DP];
#pragma unroll
            for (int d = 0; d < DP; ++d) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < KS; ++k)
                    acc += (xv[u][k][0] * w[d][k][0] + xv[u][k][1] * w[d][k][1]) +
                           (xv[u][k][2] * w[d][k][2] + xv[u][k][3] * w[d][k][3]);
                v[d] = acc;
            }
            const float dot = narrow_reduce<DP>(v, lane, gl);     // of channel dl, in every lane of the row
            if (row < n && sl < DP) hp[row * DP + dl] = dot;
            if (a_dst) {
                // <hp[n, h, :], att_p[h, :]>: the Cp (4 or 8) owners of a head are neighbouring lanes
                // the owners' channel indices are bit-reversed lane indices: with one head all DP
                // owners add up; with two heads of 4 channels (DP 8) the head is lane bit 0
                float t = dot * attv;
                if (H == 1) {
                    t += narrow_xor<1>(t);
                    t += narrow_xor<2>(t);
                    if (DP == 8) t += narrow_xor<4>(t);
                } else {
                    t += narrow_xor<2>(t);
                    t += narrow_xor<4>(t);
                }
                if (row < n && sl < H) a_dst[row * H + dl / Cp] = t;
            }
        }
    }
}

// d_x[n, :] = sum_d dh[n, d] wp[d, :],  slab[block][d, :] = sum over the block's rows of dh[n, d] x[n, :],
// dh[n, d] = d_hp[n, d] + d_a_dst[n, head(d)] att_p[d]
template <int DP, int KS>
__global__ __launch_bounds__(256) void proj_narrow_bwd_kernel(const float *__restrict__ x, const float *__restrict__ wp,
                                                              const float *__restrict__ d_hp,
                                                              const float *__restrict__ d_a_dst,
                                                              const float *__restrict__ att_p, long long n, int F,
                                                              int glog, int H, int Cp, float *__restrict__ d_x,
                                                              float *__restrict__ slab) {
    constexpr int UN = GATMI_NARROW_UN;                           // rows in flight per lane group
    __shared__ nf4 red[4][DP][64];
    const int lane = threadIdx.x & 63, gl = 1 << glog, sl = lane & (gl - 1), rsub = lane >> glog;
    const int rpw = 64 >> glog, lph = F >> 2, wv = threadIdx.x >> 6;
    nf4 w[DP][KS], acc[DP][KS];
    float att[DP];
#pragma unroll
    for (int d = 0; d < DP; ++d) {
        att[d] = d_a_dst ? att_p[d] : 0.f;
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const int sub = sl + k * gl;
            w[d][k] = (d_x && sub < lph) ? *reinterpret_cast<const nf4 *>(wp + (size_t)d * F + sub * 4) : nf4{0.f, 0.f, 0.f, 0.f};
            acc[d][k] = nf4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const long long wave = blockIdx.x * 4ll + wv, nwaves = gridDim.x * 4ll;
    for (long long row0 = wave * rpw * UN; row0 < n; row0 += nwaves * rpw * UN) {
        nf4 xv[UN][KS];
        float dh[UN][DP];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long row = row0 + u * rpw + rsub;
            const bool on = row < n;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int sub = sl + k * gl;
                xv[u][k] = (on && sub < lph) ? *reinterpret_cast<const nf4 *>(x + row * F + sub * 4) : nf4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int q = 0; q < DP / 4; ++q) {
                const nf4 t = on ? *reinterpret_cast<const nf4 *>(d_hp + row * DP + 4 * q) : nf4{0.f, 0.f, 0.f, 0.f};
                dh[u][4 * q] = t[0]; dh[u][4 * q + 1] = t[1]; dh[u][4 * q + 2] = t[2]; dh[u][4 * q + 3] = t[3];
            }
            if (d_a_dst) {
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    const float da = on ? d_a_dst[row * H + d / Cp] : 0.f;
                    dh[u][d] = fmaf(da, att[d], dh[u][d]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long row = row0 + u * rpw + rsub;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                nf4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < DP; ++d) {
                    o += w[d][k] * dh[u][d];
                    acc[d][k] += xv[u][k] * dh[u][d];
                }
                const int sub = sl + k * gl;
                if (d_x && row < n && sub < lph) *reinterpret_cast<nf4 *>(d_x + row * F + sub * 4) = o;
            }
        }
    }
    if (!slab) return;
    // the row groups of a wave, then the block's four waves: one slab per block
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
        for (int d = 0; d < DP; ++d) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                for (int o = gl; o < 64; o <<= 1) acc[d][k][c] += __shfl_xor(acc[d][k][c], o, 64);
            red[wv][d][lane] = acc[d][k];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < DP * gl; i += 256) {
            const int d = i / gl, l = i - d * gl, sub = l + k * gl;
            if (sub < lph) {
                const nf4 t = (red[0][d][l] + red[1][d][l]) + (red[2][d][l] + red[3][d][l]);
                *reinterpret_cast<nf4 *>(slab + ((size_t)blockIdx.x * DP + d) * F + sub * 4) = t;
            }
        }
        __syncthreads();
    }
}

struct NarrowPlan {
    bool ok;
    int dp, ks, glog, blocks;
};
static NarrowPlan narrow_plan(const gatmi_shape &s) {
    NarrowPlan p{false, 0, 0, 0, 0};
    if (option("GATMI_NARROW", 1) == 0 || s.num_nodes < 4096) return p;
    const int Cp = round_up4(s.channels), Dp = Cp * s.heads, F = s.in_features;
    if ((Dp != 4 && Dp != 8) || (Cp != 4 && Cp != 8) || F % 4 != 0 || F < 4 * Dp) return p;
    const int lph = F / 4;
    int gl = next_pow2(lph);
    if (gl > 64) gl = 64;
    const int ks = (lph + gl - 1) / gl;
    if (gl < Dp || (ks != 1 && ks != 2) || Dp * ks > 8) return p;
    p.ok = true;
    p.dp = Dp;
    p.ks = ks;
    p.glog = ilog2(gl);
    const long long rows_per_block = 4ll * (64 / gl) * 8;         // eight trips of a block at least
    long long b = (s.num_nodes + rows_per_block - 1) / rows_per_block;
    const long long cap = option("GATMI_NARROW_BLOCKS", 1024);
    p.blocks = (int)(b > cap ? cap : (b < 1 ? 1 : b));
    return p;
}

template <int DP, int KS>
static void narrow_launch(bool fwd, const NarrowPlan &p, const float *x, const float *wp, const float *d_hp,
                          const float *d_a_dst, const float *att_p, long long n, int F, int H, int Cp, float *out,
                          float *a_dst_or_slab, hipStream_t st) {
    if (fwd)
        hipLaunchKernelGGL((proj_narrow_fwd_kernel<DP, KS>), dim3(p.blocks), dim3(256), 0, st, x, wp, att_p, n, F,
                           p.glog, H, Cp, out, a_dst_or_slab);
    else
        hipLaunchKernelGGL((proj_narrow_bwd_kernel<DP, KS>), dim3(p.blocks), dim3(256), 0, st, x, wp, d_hp, d_a_dst,
                           att_p, n, F, p.glog, H, Cp, out, a_dst_or_slab);
}
static void narrow_dispatch(bool fwd, const NarrowPlan &p, const float *x, const float *wp, const float *d_hp,
                            const float *d_a_dst, const float *att_p, long long n, int F, int H, int Cp, float *out,
                            float *a_dst_or_slab, hipStream_t st) {
    if (p.dp == 4 && p.ks == 1) narrow_launch<4, 1>(fwd, p, x, wp, d_hp, d_a_dst, att_p, n, F, H, Cp, out, a_dst_or_slab, st);
    else if (p.dp == 4) narrow_launch<4, 2>(fwd, p, x, wp, d_hp, d_a_dst, att_p, n, F, H, Cp, out, a_dst_or_slab, st);
    else narrow_launch<8, 1>(fwd, p, x, wp, d_hp, d_a_dst, att_p, n, F, H, Cp, out, a_dst_or_slab, st);
}

static int dweight_splits(int N, int Dp, int F) {
    const int tiles = ((Dp + 127) / 128) * ((F + 127) / 128);
    int want = 512 / tiles;   // ~512 blocks: enough to fill 256 CUs, half the slab traffic of 1024
    const int maxs = (N + 4 * BK - 1) / (4 * BK);
    if (want > maxs) want = maxs;
    return want < 1 ? 1 : want;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" size_t gatmi_project_workspace_bytes(const gatmi_shape *s) {
    if (!s) return 0;
    const size_t Cp = round_up4(s->channels), Dp = Cp * s->heads, F = s->in_features;
    const size_t wp = align_up(Dp * F * 4);
    size_t nslab = (size_t)dweight_splits(s->num_nodes, (int)Dp, (int)F);
    if (const NarrowPlan np = narrow_plan(*s); np.ok && (size_t)np.blocks > nslab) nslab = (size_t)np.blocks;
    const size_t slab = align_up(nslab * Dp * F * 4);
    return wp + slab + align_up(Dp * 4) + 256 + split3_workspace_bytes(*s);
}

// where the split-GEMM region of the workspace starts (behind everything the f32 kernels use)
static void *split3_ws(const gatmi_shape *s, void *ws) {
    return static_cast<char *>(ws) + (gatmi_project_workspace_bytes(s) - split3_workspace_bytes(*s));
}
// GATMI_SPLIT3 bits: 1 forward product, 2 d_x, 4 d_weight (0: all three on the f32 MFMA kernels)
static int split3_mask(const gatmi_shape *s) {
    if (!split3_covers(*s)) return 0;
    // d_W's row-streaming kernel needs rows to fill the chip; below, the split-K tiles are faster
    const int dflt = s->num_nodes >= 32768 ? 7 : 3;
    return option("GATMI_SPLIT3", dflt) & 7;
}

static const float *padded_weight(const gatmi_shape *s, const float *weight, float *wp_buf,
                                  hipStream_t st) {
    const int C = s->channels, Cp = round_up4(C);
    if (C == Cp) return weight;
    const long long tot = (long long)s->heads * Cp * s->in_features;
    const int blocks = (int)((tot + 255) / 256 > 1024 ? 1024 : (tot + 255) / 256);
    hipLaunchKernelGGL(pad_weight_kernel, dim3(blocks), dim3(256), 0, st, weight, wp_buf, s->heads,
                       C, Cp, s->in_features);
    return wp_buf;
}

extern "C" int gatmi_project_fwd(const gatmi_shape *s, const float *x, const float *weight,
                                 const float *att_dst, float *hp, float *a_dst, void *ws,
                                 size_t ws_bytes, void *stream) {
    GATMI_REQUIRE(s && x && weight && hp, "null argument");
    GATMI_REQUIRE(s->heads >= 1 && s->channels >= 1 && s->in_features >= 1 && s->num_nodes >= 0,
                  "bad layer shape");
    if (ws_bytes < gatmi_project_workspace_bytes(s)) {
        set_error("project workspace too small: %zu < %zu", ws_bytes,
                  gatmi_project_workspace_bytes(s));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = s->num_nodes, F = s->in_features, H = s->heads, C = s->channels;
    const int Cp = round_up4(C), Dp = H * Cp;
    GATMI_REQUIRE(Cp <= 1024, "channels %d too wide", C);
    if (N == 0) return GATMI_OK;
    if (const NarrowPlan np = narrow_plan(*s); np.ok) {
        // a narrow last layer: one streaming pass over x, a_dst from the same kernel
        const float *wp = padded_weight(s, weight, static_cast<float *>(ws), st);
        GATMI_HIP(hipGetLastError());
        float *att_p = nullptr;
        if (a_dst) {
            GATMI_REQUIRE(att_dst, "a_dst requested without att_dst");
            att_p = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)Dp * F * 4));
            hipLaunchKernelGGL(pad_weight_kernel, dim3(1), dim3(256), 0, st, att_dst, att_p, H, C, Cp, 1);
            GATMI_HIP(hipGetLastError());
        }
        ProfScope ps("proj_fwd_gemm", st);
        narrow_dispatch(true, np, x, wp, nullptr, nullptr, att_p, N, F, H, Cp, hp, a_dst, st);
        GATMI_HIP(hipGetLastError());
        return GATMI_OK;
    }
    if (split3_mask(s) & 1) {
        // f32-accurate product from three bf16 pieces per operand on the bf16 matrix cores
        ProfScope ps("proj_fwd_gemm", st);
        const bool fused = a_dst && att_dst && split3_fwd_makes_a_dst(*s) && option("GATMI_ADST_FUSED", 1);
        if (int e = split3_project_fwd(s, x, weight, fused ? att_dst : nullptr, hp, fused ? a_dst : nullptr,
                                       split3_ws(s, ws), st))
            return e;
        if (fused) a_dst = nullptr;         // done in the GEMM's epilogue
    } else {
    const float *wp = padded_weight(s, weight, static_cast<float *>(ws), st);
    GATMI_HIP(hipGetLastError());
    {
        ProfScope ps("proj_fwd_gemm", st);
        const size_t slab_floats = (size_t)dweight_splits(N, Dp, F) * Dp * F;
        const int nsplit = fwd_splits(N, Dp, F, slab_floats);
        if (nsplit > 1) {
            // workspace: padded weight | padded att_dst | slab
            float *slab = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)Dp * F * 4) +
                                                    align_up((size_t)Dp * 4));
            if (int e = launch_gemm(false, false, x, F, wp, F, slab, Dp, N, Dp, F, nsplit, st)) return e;
            const long long n4 = (long long)N * Dp / 4;
            const int blocks = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
            hipLaunchKernelGGL(sum_slabs_kernel, dim3(blocks), dim3(256), 0, st,
                               reinterpret_cast<const f32x4 *>(slab), nsplit, n4,
                               reinterpret_cast<f32x4 *>(hp));
            GATMI_HIP(hipGetLastError());
        } else if (int e = launch_gemm(false, false, x, F, wp, F, hp, Dp, N, Dp, F, 1, st)) {
            return e;
        }
    }
    }
    if (a_dst) {
        ProfScope ps("proj_row_dots", st);
        GATMI_REQUIRE(att_dst, "a_dst requested without att_dst");
        const int lph = Cp / 4;
        const long long groups = (long long)N * H;
        if ((lph & (lph - 1)) == 0 && lph <= 64) {
            // padded att_dst behind the padded weight in the workspace
            float *att_p = reinterpret_cast<float *>(static_cast<char *>(ws) +
                                                     align_up((size_t)Dp * F * 4));
            hipLaunchKernelGGL(pad_weight_kernel, dim3(1), dim3(256), 0, st, att_dst, att_p, H, C,
                               Cp, 1);
            const long long nq = groups * lph;
            const long long want = (nq + 256 * 8 - 1) / (256 * 8);
            hipLaunchKernelGGL(row_dots_flat_kernel, dim3((unsigned)(want > 4096 ? 4096 : want)),
                               dim3(256), 0, st, hp, att_p, a_dst, nq, H, ilog2(lph), Cp);
        } else {
            const int R = 256 / lph;
            hipLaunchKernelGGL(row_dots_kernel, dim3((unsigned)((groups + R - 1) / R)), dim3(256),
                               0, st, hp, att_dst, a_dst, groups, H, C, Cp);
        }
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

extern "C" int gatmi_project_bwd(const gatmi_shape *s, const float *x, const float *weight,
                                 const float *d_hp, const float *d_a_dst, const float *att_dst,
                                 float *d_weight, float *d_x, void *ws, size_t ws_bytes,
                                 void *stream) {
    GATMI_REQUIRE(s && x && weight && d_hp, "null argument");
    if (ws_bytes < gatmi_project_workspace_bytes(s)) {
        set_error("project workspace too small: %zu < %zu", ws_bytes,
                  gatmi_project_workspace_bytes(s));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = s->num_nodes, F = s->in_features, H = s->heads, C = s->channels;
    const int Cp = round_up4(C), Dp = H * Cp;
    if (const NarrowPlan np = narrow_plan(*s); np.ok && N > 0) {
        // a narrow last layer: d_x and d_weight from ONE pass over x (a slab of d_weight per block)
        GATMI_REQUIRE(!d_a_dst || att_dst, "d_a_dst given without att_dst");
        if (!d_weight && !d_x) return GATMI_OK;
        const float *wp = d_x ? padded_weight(s, weight, static_cast<float *>(ws), st) : nullptr;
        GATMI_HIP(hipGetLastError());
        float *slab = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)Dp * F * 4));
        size_t nslab = (size_t)dweight_splits(N, Dp, F);
        if ((size_t)np.blocks > nslab) nslab = (size_t)np.blocks;
        float *att_p = reinterpret_cast<float *>(reinterpret_cast<char *>(slab) + align_up(nslab * Dp * F * 4));
        if (d_a_dst) {
            hipLaunchKernelGGL(pad_weight_kernel, dim3(1), dim3(256), 0, st, att_dst, att_p, H, C, Cp, 1);
            GATMI_HIP(hipGetLastError());
        }
        {
            ProfScope ps("proj_bwd_x_gemm", st);
            narrow_dispatch(false, np, x, wp, d_hp, d_a_dst, att_p, N, F, H, Cp, d_x, d_weight ? slab : nullptr, st);
            GATMI_HIP(hipGetLastError());
        }
        if (d_weight) {
            ProfScope ps("proj_bwd_w_gemm", st);
            const long long tot = (long long)H * C * F;
            hipLaunchKernelGGL(reduce_dweight_wide_kernel, dim3((unsigned)((tot + 15) / 16)), dim3(256), 0, st, slab,
                               np.blocks, H, C, Cp, F, d_weight);
            GATMI_HIP(hipGetLastError());
        }
        return GATMI_OK;
    }
    if (const int sp = N > 0 ? split3_mask(s) : 0; sp & 6) {
        GATMI_REQUIRE(!d_a_dst || att_dst, "d_a_dst given without att_dst");
        if (int e = split3_project_bwd(s, x, weight, d_hp, d_a_dst, att_dst,
                                       (sp & 4) ? d_weight : nullptr, (sp & 2) ? d_x : nullptr,
                                       split3_ws(s, ws), st))
            return e;
        if (sp & 4) d_weight = nullptr;
        if (sp & 2) d_x = nullptr;
        if (!d_weight && !d_x) return GATMI_OK;
    }
    float *wp_buf = static_cast<float *>(ws);
    float *slab = reinterpret_cast<float *>(static_cast<char *>(ws) + align_up((size_t)Dp * F * 4));
    AFix fix{nullptr, nullptr, H, Cp};
    if (d_a_dst) {
        GATMI_REQUIRE(att_dst, "d_a_dst given without att_dst");
        float *att_p = reinterpret_cast<float *>(reinterpret_cast<char *>(slab) +
                                                 align_up((size_t)dweight_splits(N, Dp, F) * Dp * F * 4));
        hipLaunchKernelGGL(pad_weight_kernel, dim3(1), dim3(256), 0, st, att_dst, att_p, H, C, Cp, 1);
        GATMI_HIP(hipGetLastError());
        fix.da_dst = d_a_dst;
        fix.att_p = att_p;
    }
    if (d_weight) {
        if (N == 0) {
            GATMI_HIP(hipMemsetAsync(d_weight, 0, (size_t)H * C * F * 4, st));
        } else {
            ProfScope ps("proj_bwd_w_gemm", st);
            const int ns = dweight_splits(N, Dp, F);
            if (int e = launch_gemm(true, true, d_hp, Dp, x, F, slab, F, Dp, F, N, ns, st, fix)) return e;
            const long long tot = (long long)H * C * F;
            const int blocks = (int)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256);
            hipLaunchKernelGGL(reduce_dweight_kernel, dim3(blocks), dim3(256), 0, st, slab, ns, H,
                               C, Cp, F, d_weight);
            GATMI_HIP(hipGetLastError());
        }
    }
    if (d_x && N > 0) {
        const float *wp = padded_weight(s, weight, wp_buf, st);
        GATMI_HIP(hipGetLastError());
        ProfScope ps("proj_bwd_x_gemm", st);
        if (int e = launch_gemm(false, true, d_hp, Dp, wp, F, d_x, F, N, F, Dp, 1, st, fix)) return e;
    }
    return GATMI_OK;
}

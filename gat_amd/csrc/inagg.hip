// Input-side aggregation: the GAT layer of /root/reference/gat/model.py:40 (the INPUT layer,
// whose x is data and needs no gradient) with the projection moved behind the aggregation.
//
//   out[i,h,:] = sum_e alpha~_e (W_h x_j) + b_h  =  W_h (sum_e alpha~_e x_j) + b_h  =  W_h Z[i,h,:] + b_h
//   a_src[j,h] = <W_h x_j, att_src_h> = <x_j, ws_h>,   ws_h = W_h^T att_src_h   (a_dst, wd likewise)
//
// so the per-edge gather moves an x row (F floats) instead of a projected row (H*C floats):
// 400 B instead of 1 KiB per edge for the products-shaped layer 100 -> 4x64, 112 B instead
// of 1 KiB for run.py's conv1 25 -> 8x30.  Same function as attn.hip's path (SURVEY.md 8a
// rows a5-a11), sums re-associated; chosen by the host when F < H*Cp and d_x is not needed.
//
// The backward stays on the DESTINATION CSR (no transposed pass, no per-edge output):
//   q[i,h,:]  = W_h^T g[i,h,:]                     (GEMM)           d alpha~_e = <q[i,h], x_j>
//   S[i,h]    = <q[i,h], Z[i,h]>                   (= sum_e alpha_e d alpha_e)
//   d_u_e     = (alpha~_e d alpha~_e - alpha_e S) leaky'(u_e)
//   d_ws_h    = sum_e d_u_e x_j      d_wd_h = sum_i (sum_{e in in(i)} d_u_e) x_i     (per-wave partial sums)
//   d_W_h     = g_h^T Z_h (GEMM) + att_src_h (x) d_ws_h + att_dst_h (x) d_wd_h
//   d_att_src_h = W_h d_ws_h       d_att_dst_h = W_h d_wd_h         d_bias = sum_i g_i
//
// Lane mapping: a destination row is owned by G = H2 * LH lanes (H2 = next_pow2(H)); the LH
// consecutive lanes of a head hold T float4 of the x row each (sub = sl + t*LH), every head
// group reads the same row.  Per-head dots are butterflies over LH lanes.
#ifndef GATMI_INAGG_U2
#define GATMI_INAGG_U2 2
#endif
#include "attn_device.h"
#include "common.h"

namespace gatmi {

struct InArgs {
    const int *rowptr, *col, *item_row, *item_beg, *item_slot;
    int n_items, seg_len;
    int H, Fp, lph, lh_log, dst_off;
    float neg_slope, inv_keep;
    uint32_t keep_thr;
    int mask_mode;
    const uint8_t *mask;
    uint64_t seed;
    const float *xp;    // [num_src, Fp]
    const float *wsd;   // [2][H*Fp]: ws, wd (padded columns zero)
    // forward
    float *z, *lse, *part_m, *part_s, *part_acc;
    // backward
    const float *q, *zin, *lsein;
    float *part_y;      // [waves][2][H*Fp]
};

#define INAGG_PROLOGUE                                                                    \
    const int lane = threadIdx.x & 63;                                                    \
    const int lig = lane & (G - 1);                                                       \
    const int gbase = lane & ~(G - 1);                                                    \
    const int lh = 1 << A.lh_log;                                                         \
    const int head_raw = lig >> A.lh_log, sl = lig & (lh - 1);                            \
    const bool hok = head_raw < A.H;                                                      \
    const int head = hok ? head_raw : 0;                                                  \
    const int HF = A.H * A.Fp;                                                            \
    bool ok[T];                                                                           \
    unsigned offb[T];                                                                     \
    int roff[T];                                                                          \
    float4 ws4[T], wd4[T];                                                                \
    _Pragma("unroll") for (int t = 0; t < T; ++t) {                                       \
        const int sub = sl + t * lh;                                                      \
        ok[t] = hok && sub < A.lph;                                                       \
        offb[t] = ok[t] ? (unsigned)sub * 16u : 0u;                                       \
        roff[t] = ok[t] ? head * A.Fp + sub * 4 : 0;                                      \
        ws4[t] = ok[t] ? ld4(A.wsd + roff[t]) : make_float4(0.f, 0.f, 0.f, 0.f);          \
        wd4[t] = ok[t] ? ld4(A.wsd + HF + roff[t]) : make_float4(0.f, 0.f, 0.f, 0.f);     \
    }                                                                                     \
    const unsigned hbit = 1u << head;                                                     \
    constexpr int GPB = 256 / G;                                                          \
    const int stride = gridDim.x * GPB;

#define INAGG_ITEM_LOOP                                                                   \
    for (int it = xcd_contiguous_block() * GPB + (int)threadIdx.x / G; it < A.n_items;    \
         it += stride) {                                                                  \
        const int row = uniform<G>(A.item_row[it]);                                       \
        const int beg = uniform<G>(A.item_beg[it]);                                       \
        const int slot = uniform<G>(A.item_slot[it]);                                     \
        int end = uniform<G>(A.rowptr[row + 1]);                                          \
        end = min(end, beg + A.seg_len);                                                  \
        float4 xi[T];                                                                     \
        float adi = 0.f;                                                                  \
        _Pragma("unroll") for (int t = 0; t < T; ++t) {                                   \
            xi[t] = ok[t] ? ld4(A.xp + (size_t)(row + A.dst_off) * A.Fp + (offb[t] >> 2)) \
                          : make_float4(0.f, 0.f, 0.f, 0.f);                              \
            adi += dot4(xi[t], wd4[t]);                                                   \
        }                                                                                 \
        adi = head_sum<LH_T>(adi, lh);

// ---------------------------------------------------------------------------------
// Forward: online softmax over the row's edges, Z[i,h,:] = sum_e alpha~_e x_j, lse.
// Same pipeline as attn_fwd_kernel (double-buffered branch-free gathers).
// ---------------------------------------------------------------------------------
template <int LH_T, int G, int T>
__global__ __launch_bounds__(256) void inagg_fwd_kernel(const InArgs A) {
    constexpr int U = (T == 1) ? 4 : (T == 2 ? GATMI_INAGG_U2 : 1);
    INAGG_PROLOGUE
    INAGG_ITEM_LOOP
        float m = -INFINITY, ssum = 0.f;
        float4 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int c = beg; c < end; c += G) {
            const int n = min(G, end - c);
            const int cl = min(lig, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(A.mask_mode, A.mask, A.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](float4(&v)[U][T], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = group_bcast<G>(cj, min(k0 + u, n - 1), gbase);
#pragma unroll
                    for (int t = 0; t < T; ++t) v[u][t] = ld4_row<G>(A.xp, j, A.Fp, offb[t]);
                }
            };
            auto reduce = [&](const float4(&v)[U][T], int k0) {
                float z[U];
                float mn = m;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float d = 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t) d += dot4(v[u][t], ws4[t]);
                    const float uu = head_sum<LH_T>(d, lh) + adi;
                    const float zz = uu > 0.f ? uu : A.neg_slope * uu;
                    z[u] = (k0 + u < n) ? zz : -INFINITY;
                    mn = fmaxf(mn, z[u]);
                }
                const float sc = exp_fast(m - mn);
                ssum *= sc;
#pragma unroll
                for (int t = 0; t < T; ++t) scale4(acc[t], sc);
                m = mn;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = k0 + u < n;
                    const unsigned sb = (unsigned)group_bcast<G>(kb, min(k0 + u, n - 1), gbase);
                    const float p = live ? exp_fast(z[u] - m) : 0.f;
                    ssum += p;
                    const float w = (sb & hbit) ? p * A.inv_keep : 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t) fma4(acc[t], w, v[u][t]);
                }
            };
            float4 va[U][T], vb[U][T];
            load(va, 0);
            for (int k0 = 0; k0 < n; k0 += 2 * U) {
                load(vb, k0 + U);
                reduce(va, k0);
                load(va, k0 + 2 * U);
                reduce(vb, k0 + U);
            }
        }
        if (slot < 0) {
            const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (ok[t]) {
                    float4 o = acc[t];
                    scale4(o, inv);
                    st4(A.z + (size_t)row * HF + roff[t], o);
                }
            if (A.lse && hok && sl == 0) A.lse[(size_t)row * A.H + head] = m + logf(ssum);
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (ok[t]) st4(A.part_acc + (size_t)slot * HF + roff[t], acc[t]);
            if (hok && sl == 0) {
                A.part_m[(size_t)slot * A.H + head] = m;
                A.part_s[(size_t)slot * A.H + head] = ssum;
            }
        }
    }
}

// Merge of the partial (m, s, acc) of destination rows cut into segments (fixed order).
__global__ __launch_bounds__(256) void inagg_fwd_merge_kernel(
    const int *split_row, const int *split_slot, int n_split, int H, int Fp, const float *part_m,
    const float *part_s, const float *part_acc, float *z, float *lse) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (w >= n_split) return;
    const int lane = threadIdx.x & 63;
    const int row = split_row[w], s0 = split_slot[w], s1 = split_slot[w + 1];
    const int lph = Fp >> 2, HF = H * Fp;
    for (int q = lane; q < H * lph; q += 64) {
        const int head = q / lph, sub = q % lph;
        float M = -INFINITY;
        for (int s = s0; s < s1; ++s) M = fmaxf(M, part_m[(size_t)s * H + head]);
        float tot = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = s0; s < s1; ++s) {
            const float f = expf(part_m[(size_t)s * H + head] - M);
            tot = fmaf(part_s[(size_t)s * H + head], f, tot);
            fma4(acc, f, ld4(part_acc + (size_t)s * HF + head * Fp + sub * 4));
        }
        scale4(acc, tot > 0.f ? 1.0f / tot : 0.f);
        st4(z + (size_t)row * HF + head * Fp + sub * 4, acc);
        if (lse && sub == 0) lse[(size_t)row * H + head] = M + logf(tot);
    }
}

// ---------------------------------------------------------------------------------
// Backward edge pass over the destination CSR.  Nothing is written per edge or per row:
// each wave keeps d_ws / d_wd partial sums in registers and writes them once.
// ---------------------------------------------------------------------------------
template <int LH_T, int G, int T>
__global__ __launch_bounds__(256) void inagg_bwd_kernel(const InArgs A) {
    constexpr int U = (T == 1) ? 4 : (T == 2 ? GATMI_INAGG_U2 : 1);
    INAGG_PROLOGUE
    float4 ys[T], yd[T];
#pragma unroll
    for (int t = 0; t < T; ++t) ys[t] = yd[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    INAGG_ITEM_LOOP
        (void)slot;
        float4 q4[T];
        float S = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            q4[t] = ok[t] ? ld4(A.q + (size_t)row * HF + roff[t]) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 z4 = ok[t] ? ld4(A.zin + (size_t)row * HF + roff[t])
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
            S += dot4(q4[t], z4);
        }
        S = head_sum<LH_T>(S, lh);
        const float lse_i = A.lsein[(size_t)row * A.H + head];
        float dad = 0.f;
        for (int c = beg; c < end; c += G) {
            const int n = min(G, end - c);
            const int cl = min(lig, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(A.mask_mode, A.mask, A.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](float4(&v)[U][T], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = group_bcast<G>(cj, min(k0 + u, n - 1), gbase);
#pragma unroll
                    for (int t = 0; t < T; ++t) v[u][t] = ld4_row<G>(A.xp, j, A.Fp, offb[t]);
                }
            };
            auto reduce = [&](const float4(&v)[U][T], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float d1 = 0.f, d2 = 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        d1 += dot4(v[u][t], ws4[t]);
                        d2 += dot4(v[u][t], q4[t]);
                    }
                    d1 = head_sum<LH_T>(d1, lh);
                    d2 = head_sum<LH_T>(d2, lh);
                    const bool live = k0 + u < n;
                    const unsigned sb = (unsigned)group_bcast<G>(kb, min(k0 + u, n - 1), gbase);
                    const float uu = d1 + adi;
                    const float zz = uu > 0.f ? uu : A.neg_slope * uu;
                    const float alpha = live ? exp_fast(zz - lse_i) : 0.f;
                    const float at = (sb & hbit) ? alpha * A.inv_keep : 0.f;
                    const float du = (at * d2 - alpha * S) * (uu > 0.f ? 1.0f : A.neg_slope);
                    dad += du;
#pragma unroll
                    for (int t = 0; t < T; ++t) fma4(ys[t], du, v[u][t]);
                }
            };
            float4 va[U][T], vb[U][T];
            load(va, 0);
            for (int k0 = 0; k0 < n; k0 += 2 * U) {
                load(vb, k0 + U);
                reduce(va, k0);
                load(va, k0 + 2 * U);
                reduce(vb, k0 + U);
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) fma4(yd[t], dad, xi[t]);
    }
    // fold the wave's 64/G row groups, then one partial row per wave
    if constexpr (G < 64) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            for (int o = G; o < 64; o <<= 1) {
                ys[t].x += __shfl_xor(ys[t].x, o, 64); ys[t].y += __shfl_xor(ys[t].y, o, 64);
                ys[t].z += __shfl_xor(ys[t].z, o, 64); ys[t].w += __shfl_xor(ys[t].w, o, 64);
                yd[t].x += __shfl_xor(yd[t].x, o, 64); yd[t].y += __shfl_xor(yd[t].y, o, 64);
                yd[t].z += __shfl_xor(yd[t].z, o, 64); yd[t].w += __shfl_xor(yd[t].w, o, 64);
            }
    }
    if (lane < G) {
        float *dst = A.part_y + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 * HF;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (ok[t]) {
                st4(dst + roff[t], ys[t]);
                st4(dst + HF + roff[t], yd[t]);
            }
    }
}

// ---------------------------------------------------------------------------------
// Small kernels around the two passes
// ---------------------------------------------------------------------------------
// wsd[k][h*Fp + f] = sum_c W[(h*C + c), f] * att_k[h*C + c]  (k = 0: att_src, 1: att_dst), pad = 0
__global__ void inagg_wsd_kernel(const float *w, const float *att_src, const float *att_dst, int H,
                                 int C, int F, int Fp, float *wsd) {
    const int tot = 2 * H * Fp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x) {
        const int k = i / (H * Fp), r = i % (H * Fp), h = r / Fp, f = r % Fp;
        const float *att = k == 0 ? att_src : att_dst;
        float s = 0.f;
        if (f < F)
            for (int c = 0; c < C; ++c) s = fmaf(w[((size_t)h * C + c) * F + f], att[h * C + c], s);
        wsd[i] = s;
    }
}

// xp[n, 0..Fp) = x[n, 0..F), zero padded (only when F % 4 != 0)
__global__ void inagg_pad_x_kernel(const float *x, float *xp, long long n, int F, int Fp) {
    const long long tot = n * Fp;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot;
         i += (long long)gridDim.x * blockDim.x) {
        const int f = (int)(i % Fp);
        xp[i] = f < F ? x[(i / Fp) * F + f] : 0.f;
    }
}

// partial[b][col] = sum over block b's stripe of rows of g[row][col]   (d_bias, stage 1)
__global__ __launch_bounds__(256) void inagg_gsum_kernel(const float *g, long long n, int D,
                                                         float *partial) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long r0 = blockIdx.x * per, r1 = min(n, r0 + per);
    for (int col = threadIdx.x; col < D; col += 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        long long r = r0;
        for (; r + 3 < r1; r += 4) {
            s0 += g[r * D + col];
            s1 += g[(r + 1) * D + col];
            s2 += g[(r + 2) * D + col];
            s3 += g[(r + 3) * D + col];
        }
        for (; r < r1; ++r) s0 += g[r * D + col];
        partial[(size_t)blockIdx.x * D + col] = (s0 + s1) + (s2 + s3);
    }
}

// dst[col] = sum of nrows partial rows, fixed order.  One block per (job, 16 columns).
struct SumJob {
    const float *src;
    float *dst;
    int nrows, ld, ncols;
};
__global__ __launch_bounds__(256) void inagg_sum_rows_kernel(SumJob j0, SumJob j1, SumJob j2) {
    __shared__ float red[256];
    const SumJob J = blockIdx.y == 0 ? j0 : (blockIdx.y == 1 ? j1 : j2);
    const int pc = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;
    const bool on = J.dst && pc < J.ncols;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (on) {
        int b = rg;
        for (; b + 48 < J.nrows; b += 64) {
            s0 += J.src[(size_t)b * J.ld + pc];
            s1 += J.src[(size_t)(b + 16) * J.ld + pc];
            s2 += J.src[(size_t)(b + 32) * J.ld + pc];
            s3 += J.src[(size_t)(b + 48) * J.ld + pc];
        }
        for (; b < J.nrows; b += 16) s0 += J.src[(size_t)b * J.ld + pc];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && on) {
        float tot = 0.f;
        for (int q = 0; q < 16; ++q) tot += red[threadIdx.x + q * 16];
        J.dst[pc] = tot;
    }
}

// d_W = sum_split slab + att_src (x) d_ws + att_dst (x) d_wd;  d_att_k = W d_w{s,d}
__global__ __launch_bounds__(256) void inagg_finalize_kernel(
    const float *slab, int nsplit, int H, int C, int F, int Fp, const float *w,
    const float *att_src, const float *att_dst, const float *dwsd, float *d_w, float *d_att_src,
    float *d_att_dst) {
    const long long tot = (long long)H * C * F;
    const float *dws = dwsd, *dwd = dwsd + (size_t)H * Fp;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < tot; i += gridDim.x * 256ll) {
        const int f = (int)(i % F), hc = (int)(i / F), h = hc / C;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int z = 0;
        for (; z + 3 < nsplit; z += 4) {
            s0 += slab[(size_t)z * tot + i];
            s1 += slab[(size_t)(z + 1) * tot + i];
            s2 += slab[(size_t)(z + 2) * tot + i];
            s3 += slab[(size_t)(z + 3) * tot + i];
        }
        for (; z < nsplit; ++z) s0 += slab[(size_t)z * tot + i];
        float v = (s0 + s1) + (s2 + s3);
        v = fmaf(att_src[hc], dws[h * Fp + f], v);
        v = fmaf(att_dst[hc], dwd[h * Fp + f], v);
        if (d_w) d_w[i] = v;
    }
    const int HC = H * C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * HC; i += gridDim.x * 256) {
        const int k = i / HC, hc = i % HC, h = hc / C;
        const float *dv = (k == 0 ? dws : dwd) + h * Fp;
        float s = 0.f;
        for (int f = 0; f < F; ++f) s = fmaf(w[(size_t)hc * F + f], dv[f], s);
        float *dst = k == 0 ? d_att_src : d_att_dst;
        if (dst) dst[hc] = s;
    }
}

// ---------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------
struct InGeo {
    int Fp, lph, H2, LH, lh_log, G, T, HF;
};

static bool make_in_geo(const gatmi_shape &s, InGeo *g) {
    if (s.heads < 1 || s.heads > 32 || s.channels < 1 || s.in_features < 1) return false;
    g->Fp = round_up4(s.in_features);
    g->lph = g->Fp / 4;
    g->H2 = next_pow2(s.heads);
    const int lmax = 64 / g->H2;
    const int lneed = next_pow2(g->lph);
    g->LH = lneed < lmax ? lneed : lmax;
    g->lh_log = ilog2(g->LH);
    g->T = (g->lph + g->LH - 1) / g->LH;
    if (g->T > 4) return false;
    if (g->T == 3) g->T = 4;
    g->G = g->H2 * g->LH;
    g->HF = s.heads * g->Fp;
    return true;
}

enum InKind { IN_FWD, IN_BWD };

template <int LH_T, int G, int T>
static void in_launch(InKind k, const InArgs &A, int grid, hipStream_t st) {
    if (k == IN_FWD)
        hipLaunchKernelGGL((inagg_fwd_kernel<LH_T, G, T>), dim3(grid), dim3(256), 0, st, A);
    else
        hipLaunchKernelGGL((inagg_bwd_kernel<LH_T, G, T>), dim3(grid), dim3(256), 0, st, A);
}

template <int G>
static void in_launch_g(InKind k, const InGeo &g, const InArgs &A, int grid, hipStream_t st) {
    if (g.T == 1) in_launch<0, G, 1>(k, A, grid, st);
    else if (g.T == 2) in_launch<0, G, 2>(k, A, grid, st);
    else in_launch<0, G, 4>(k, A, grid, st);
}

static int in_grid(const InGeo &g, int n_items) { return item_grid(n_items, 256 / g.G); }

static int in_launch_rows(InKind k, const InGeo &g, const InArgs &A, int grid, hipStream_t st) {
    // compile-time butterflies for the layers BASELINE.json names: 100 -> 4x64 / 128 -> 4x64
    // (LH 16, T 2) and run.py's 25 -> 8x30 (LH 8, T 1)
    if (g.G == 64 && g.LH == 16 && g.T == 2) in_launch<16, 64, 2>(k, A, grid, st);
    else if (g.G == 64 && g.LH == 8 && g.T == 1) in_launch<8, 64, 1>(k, A, grid, st);
    else
        switch (g.G) {
            case 64: in_launch_g<64>(k, g, A, grid, st); break;
            case 32: in_launch_g<32>(k, g, A, grid, st); break;
            case 16: in_launch_g<16>(k, g, A, grid, st); break;
            case 8: in_launch_g<8>(k, g, A, grid, st); break;
            case 4: in_launch_g<4>(k, g, A, grid, st); break;
            case 2: in_launch_g<2>(k, g, A, grid, st); break;
            default: in_launch_g<1>(k, g, A, grid, st); break;
        }
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

static int check_inagg(const gatmi_graph *g, const gatmi_inagg *a, InGeo *geo) {
    GATMI_REQUIRE(g && a, "null graph/layer descriptor");
    GATMI_REQUIRE(make_in_geo(a->shape, geo),
                  "input-side aggregation does not cover F=%d H=%d (needs H <= 32 and "
                  "round_up(F,4)/4 <= 4 * 64/next_pow2(H))", a->shape.in_features, a->shape.heads);
    GATMI_REQUIRE(g->num_dst == a->shape.num_nodes, "graph has %d destination rows, layer %d",
                  g->num_dst, a->shape.num_nodes);
    GATMI_REQUIRE(g->num_src >= g->num_dst && g->dst_offset >= 0 &&
                      g->dst_offset + g->num_dst <= g->num_src,
                  "destination rows [%d, %d) outside the %d source rows", g->dst_offset,
                  g->dst_offset + g->num_dst, g->num_src);
    GATMI_REQUIRE(g->seg_len >= 64 && g->seg_len % 64 == 0, "seg_len %d must be a multiple of 64",
                  g->seg_len);
    GATMI_REQUIRE(a->mask_mode >= 0 && a->mask_mode <= 2, "bad mask_mode %d", a->mask_mode);
    GATMI_REQUIRE(a->mask_mode != GATMI_MASK_BYTES || a->mask, "mask_mode BYTES needs a mask");
    GATMI_REQUIRE(a->dropout_p >= 0.f && a->dropout_p <= 1.f, "dropout_p %f outside [0,1]",
                  a->dropout_p);
    GATMI_REQUIRE(a->x && a->weight && a->att_src && a->att_dst, "null x/weight/att_src/att_dst");
    return GATMI_OK;
}

static void fill_in_args(InArgs &A, const gatmi_graph *g, const gatmi_inagg *a, const InGeo &geo) {
    memset(&A, 0, sizeof(A));
    A.rowptr = g->rowptr;
    A.col = g->col;
    A.item_row = g->plan.item_row;
    A.item_beg = g->plan.item_beg;
    A.item_slot = g->plan.item_slot;
    A.n_items = g->plan.n_items;
    A.seg_len = g->seg_len;
    A.H = a->shape.heads;
    A.Fp = geo.Fp;
    A.lph = geo.lph;
    A.lh_log = geo.lh_log;
    A.dst_off = g->dst_offset;
    A.neg_slope = a->negative_slope;
    A.inv_keep = a->dropout_p < 1.0f ? 1.0f / (1.0f - a->dropout_p) : 0.0f;
    if (a->mask_mode == GATMI_MASK_NONE) A.inv_keep = 1.0f;
    A.keep_thr = (uint32_t)((double)a->dropout_p * 16777216.0);
    A.mask_mode = a->mask_mode;
    A.mask = a->mask;
    A.seed = a->seed;
}

// ws, wd and (when F % 4 != 0) the padded copy of x: the start of both workspaces
struct InCommonWs {
    float *wsd, *xp;
};
static InCommonWs carve_common(Carver &cv, const gatmi_graph *g, const gatmi_shape *s,
                               const InGeo &geo) {
    InCommonWs w;
    w.wsd = cv.take<float>((size_t)2 * geo.HF);
    w.xp = (s->in_features & 3) ? cv.take<float>((size_t)g->num_src * geo.Fp) : nullptr;
    return w;
}

static int prepare_common(const InCommonWs &w, const gatmi_graph *g, const gatmi_inagg *a,
                          const InGeo &geo, InArgs &A, hipStream_t st) {
    const int H = a->shape.heads, C = a->shape.channels, F = a->shape.in_features;
    ProfScope ps("inagg_prep", st);
    hipLaunchKernelGGL(inagg_wsd_kernel, dim3((2 * geo.HF + 255) / 256), dim3(256), 0, st, a->weight,
                       a->att_src, a->att_dst, H, C, F, geo.Fp, w.wsd);
    GATMI_HIP(hipGetLastError());
    A.wsd = w.wsd;
    A.xp = a->x;
    if (w.xp && g->num_src > 0) {
        const long long tot = (long long)g->num_src * geo.Fp;
        const int blocks = (int)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256);
        hipLaunchKernelGGL(inagg_pad_x_kernel, dim3(blocks), dim3(256), 0, st, a->x, w.xp,
                           (long long)g->num_src, F, geo.Fp);
        GATMI_HIP(hipGetLastError());
        A.xp = w.xp;
    }
    return GATMI_OK;
}

constexpr int kGsumBlocks = 1024;

static int dw_splits(int N, int H, int C, int F) {
    const int tiles = ((C + 127) / 128) * ((F + 127) / 128) * H;
    int want = 1024 / tiles;
    const int maxs = (N + 4 * 16 - 1) / (4 * 16);
    if (want > maxs) want = maxs;
    if ((long long)want * H > 65535) want = 65535 / H;
    return want < 1 ? 1 : want;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" int gatmi_inagg_supported(const gatmi_shape *s) {
    InGeo geo;
    return s && make_in_geo(*s, &geo) ? 1 : 0;
}

extern "C" size_t gatmi_inagg_z_floats(const gatmi_shape *s) {
    InGeo geo;
    if (!s || !make_in_geo(*s, &geo)) return 0;
    return (size_t)geo.HF;
}

extern "C" size_t gatmi_inagg_fwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s) {
    InGeo geo;
    if (!g || !s || !make_in_geo(*s, &geo)) return 0;
    Carver cv(nullptr);
    carve_common(cv, g, s, geo);
    const size_t slots = (size_t)g->plan.n_slots;
    cv.take<float>(slots * s->heads);
    cv.take<float>(slots * s->heads);
    cv.take<float>(slots * geo.HF);
    return cv.off + 256;
}

extern "C" int gatmi_inagg_fwd(const gatmi_graph *g, const gatmi_inagg *a, float *out, float *z,
                               float *lse, void *ws, size_t ws_bytes, void *stream) {
    InGeo geo;
    if (int e = check_inagg(g, a, &geo)) return e;
    GATMI_REQUIRE(out && z, "null out/z");
    const size_t need = gatmi_inagg_fwd_workspace_bytes(g, &a->shape);
    if (ws_bytes < need) {
        set_error("inagg_fwd workspace too small: %zu < %zu", ws_bytes, need);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = g->num_dst, H = a->shape.heads, C = a->shape.channels, F = a->shape.in_features;
    if (N == 0) return GATMI_OK;
    Carver cv(ws);
    const InCommonWs cw = carve_common(cv, g, &a->shape, geo);
    InArgs A;
    fill_in_args(A, g, a, geo);
    if (int e = prepare_common(cw, g, a, geo, A, st)) return e;
    const size_t slots = (size_t)g->plan.n_slots;
    A.part_m = cv.take<float>(slots * H);
    A.part_s = cv.take<float>(slots * H);
    A.part_acc = cv.take<float>(slots * geo.HF);
    A.z = z;
    A.lse = lse;
    {
        ProfScope ps("inagg_fwd", st);
        if (int e = in_launch_rows(IN_FWD, geo, A, in_grid(geo, A.n_items), st)) return e;
    }
    if (g->plan.n_split_rows > 0) {
        ProfScope ps("inagg_fwd_merge", st);
        hipLaunchKernelGGL(inagg_fwd_merge_kernel, dim3((g->plan.n_split_rows + 3) / 4), dim3(256), 0,
                           st, g->plan.split_row, g->plan.split_slot, g->plan.n_split_rows, H,
                           geo.Fp, A.part_m, A.part_s, A.part_acc, z, lse);
        GATMI_HIP(hipGetLastError());
    }
    {
        // out[:, h, :] = Z[:, h, :] W_h^T + bias_h
        ProfScope ps("inagg_out_gemm", st);
        GemmDesc d{};
        d.a_kmajor = false; d.b_kmajor = false;
        d.A = z; d.lda = geo.HF; d.sA = geo.Fp;
        d.B = a->weight; d.ldb = F; d.sB = (long long)C * F;
        d.C = out; d.ldc = H * C; d.sC = C; d.sSplit = 0;
        d.M = N; d.Nn = C; d.K = F; d.n_store = C; d.nbatch = H; d.nsplit = 1;
        d.bias = a->bias; d.sBias = C;
        d.fix = AFix{nullptr, nullptr, 1, 1};
        if (int e = launch_gemm(d, st)) return e;
    }
    return GATMI_OK;
}

namespace {
struct InBwdWs {
    InCommonWs cw;
    float *q, *part_y, *dwsd, *slab, *gsum;
    int waves, nsplit;
    size_t bytes;
};
InBwdWs carve_in_bwd(const gatmi_graph *g, const gatmi_shape *s, const InGeo &geo, void *ws) {
    Carver cv(ws);
    InBwdWs w;
    w.cw = carve_common(cv, g, s, geo);
    const size_t N = (size_t)g->num_dst;
    w.q = cv.take<float>(N * geo.HF);
    w.waves = in_grid(geo, g->plan.n_items) * 4;
    w.part_y = cv.take<float>((size_t)w.waves * 2 * geo.HF);
    w.dwsd = cv.take<float>((size_t)2 * geo.HF);
    w.nsplit = dw_splits(g->num_dst, s->heads, s->channels, s->in_features);
    w.slab = cv.take<float>((size_t)w.nsplit * s->heads * s->channels * s->in_features);
    w.gsum = cv.take<float>((size_t)kGsumBlocks * s->heads * s->channels);
    w.bytes = cv.off + 256;
    return w;
}
}  // namespace

extern "C" size_t gatmi_inagg_bwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s) {
    InGeo geo;
    if (!g || !s || !make_in_geo(*s, &geo)) return 0;
    return carve_in_bwd(g, s, geo, nullptr).bytes;
}

extern "C" int gatmi_inagg_bwd(const gatmi_graph *g, const gatmi_inagg *a, const float *z,
                               const float *lse, const float *grad_out, float *d_weight,
                               float *d_att_src, float *d_att_dst, float *d_bias, void *ws,
                               size_t ws_bytes, void *stream) {
    InGeo geo;
    if (int e = check_inagg(g, a, &geo)) return e;
    GATMI_REQUIRE(z && lse && grad_out, "null z/lse/grad_out");
    const size_t need = gatmi_inagg_bwd_workspace_bytes(g, &a->shape);
    if (ws_bytes < need) {
        set_error("inagg_bwd workspace too small: %zu < %zu", ws_bytes, need);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int N = g->num_dst, H = a->shape.heads, C = a->shape.channels, F = a->shape.in_features;
    const InBwdWs w = carve_in_bwd(g, &a->shape, geo, ws);
    InArgs A;
    fill_in_args(A, g, a, geo);
    if (int e = prepare_common(w.cw, g, a, geo, A, st)) return e;
    if (N > 0) {
        // q[:, h, :] = g[:, h, :] W_h   (padded columns written as zeros)
        ProfScope ps("inagg_q_gemm", st);
        GemmDesc d{};
        d.a_kmajor = false; d.b_kmajor = true;
        d.A = grad_out; d.lda = H * C; d.sA = C;
        d.B = a->weight; d.ldb = F; d.sB = (long long)C * F;
        d.C = w.q; d.ldc = geo.HF; d.sC = geo.Fp; d.sSplit = 0;
        d.M = N; d.Nn = F; d.K = C; d.n_store = geo.Fp; d.nbatch = H; d.nsplit = 1;
        d.bias = nullptr; d.sBias = 0;
        d.fix = AFix{nullptr, nullptr, 1, 1};
        if (int e = launch_gemm(d, st)) return e;
    }
    A.q = w.q;
    A.zin = z;
    A.lsein = lse;
    A.part_y = w.part_y;
    {
        ProfScope ps("inagg_bwd_edge", st);
        // every wave of the grid writes its partial row, also when it had no work item
        const int grid = w.waves / 4;
        if (A.n_items == 0) {
            GATMI_HIP(hipMemsetAsync(w.part_y, 0, (size_t)w.waves * 2 * geo.HF * 4, st));
        } else if (int e = in_launch_rows(IN_BWD, geo, A, grid, st)) {
            return e;
        }
    }
    const int gblocks = N < kGsumBlocks ? (N > 0 ? N : 1) : kGsumBlocks;
    {
        ProfScope ps("inagg_colsums", st);
        if (N > 0 && d_bias) {
            hipLaunchKernelGGL(inagg_gsum_kernel, dim3(gblocks), dim3(256), 0, st, grad_out,
                               (long long)N, H * C, w.gsum);
            GATMI_HIP(hipGetLastError());
        }
        SumJob j0{w.part_y, w.dwsd, w.waves, 2 * geo.HF, geo.HF};
        SumJob j1{w.part_y + geo.HF, w.dwsd + geo.HF, w.waves, 2 * geo.HF, geo.HF};
        SumJob j2{w.gsum, d_bias, N > 0 ? gblocks : 0, H * C, H * C};
        const int widest = geo.HF > H * C ? geo.HF : H * C;
        hipLaunchKernelGGL(inagg_sum_rows_kernel, dim3((widest + 15) / 16, 3), dim3(256), 0, st, j0,
                           j1, j2);
        GATMI_HIP(hipGetLastError());
    }
    {
        // slab[split][h] = g[:, h, :]^T Z[:, h, :] over the split's rows
        ProfScope ps("inagg_dw_gemm", st);
        if (N > 0) {
            GemmDesc d{};
            d.a_kmajor = true; d.b_kmajor = true;
            d.A = grad_out; d.lda = H * C; d.sA = C;
            d.B = z; d.ldb = geo.HF; d.sB = geo.Fp;
            d.C = w.slab; d.ldc = F; d.sC = (long long)C * F; d.sSplit = (long long)H * C * F;
            d.M = C; d.Nn = F; d.K = N; d.n_store = F; d.nbatch = H; d.nsplit = w.nsplit;
            d.bias = nullptr; d.sBias = 0;
            d.fix = AFix{nullptr, nullptr, 1, 1};
            if (int e = launch_gemm(d, st)) return e;
        }
        const long long tot = (long long)H * C * F;
        const int blocks = (int)((tot + 255) / 256 > 2048 ? 2048 : (tot + 255) / 256);
        hipLaunchKernelGGL(inagg_finalize_kernel, dim3(blocks), dim3(256), 0, st, w.slab,
                           N > 0 ? w.nsplit : 0, H, C, F, geo.Fp, a->weight, a->att_src, a->att_dst,
                           w.dwsd, d_weight, d_att_src, d_att_dst);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

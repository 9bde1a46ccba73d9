// Fused GAT attention kernels for gfx950: forward (logits + LeakyReLU + segment softmax
// + attention dropout + aggregation + bias), backward pass A (dst-CSR) and pass B
// (transposed CSR), the merge kernels for cut rows, and the parameter-gradient column
// reductions.  Stands for GATConv.forward/backward at /root/reference/gat/model.py:
// 40,43,46,54.  HBM-bound: each kernel gathers every source
// (or destination) feature row exactly once, one float4 per lane.
#ifndef GATMI_U
#define GATMI_U 4
#endif
#include "attn_device.h"
#include "common.h"
#include <stdlib.h>

#include <mutex>

namespace gatmi {

struct RowArgs {
    // CSR half being walked
    const int *rowptr;
    const int *col;
    const int *aux;        // backward edge pass: pos_t (dst-CSR id of each transposed edge)
    const int *item_row;
    const int *item_beg;
    const int *item_slot;
    int n_items;
    int seg_len;
    // layer
    int N, H, C, Cp, Dp, lph2_log;
    float neg_slope, p_drop, inv_keep;
    uint32_t keep_thr;     // p * 2^24
    int mask_mode;
    int feat_bf16;         // hp / hself are bf16 tables (G == 64 layouts only)
    const uint8_t *mask;
    uint64_t seed;
    const void *hp;        // fwd: projected features; bwd edge pass: padded grad_out -- f32 or bf16
                           // rows (the kernel's RAW type), [N, Dp]
    const float *a_dst;
    const float *att_src;
    const float *att_dst;
    const float *bias;
    // fwd
    float *out;
    float *lse;
    float *part_m, *part_s, *part_acc;
    float *row_m, *row_s, *row_acc;   // source chunks: un-normalised (m, s, acc) per destination row
                                      // instead of out / lse (NULL: the whole table is here)
    // backward edge pass (transposed CSR; hp = padded grad_out there)
    const void *hself;     // [N, Dp] projected features (row-local), same element type as hp
    int hp_row;            // floats per row of hp: Dp, or H*C when g_compact
    int g_compact;         // f32, C % 4 != 0, C > 4: hp is grad_out itself (rows of H*C floats, heads not padded).
                           // A lane's four channels start at head*C + 4*sub; the last lane of a head loads the
                           // head's last four channels instead and shifts (nothing is read outside the row)
    const float *nodeinfo; // [N, H, 4] (a_dst, lse, S, 0)
    float *du_t;           // [E', H] d_u in transposed-CSR order
    float *dhp;            // [N, Dp]
    float *da_src;         // [N, H]
    float *part_das;       // [slots_t, H]
    float *part_dh;        // [slots_t, Dp]
};

// The work-item loop of the row kernels.  An item's description is a chain of dependent loads (item
// arrays -> row pointer -> col of its first chunk) ahead of the first gathered row.  The loop runs ahead:
// at the top of pass k it asks for the item arrays of pass k + 3, the row pointer of pass k + 2 and the
// col (and aux) entries of the first chunk of pass k + 1.  Full-wave rows (G == 64) index the arrays with
// a scalar and read them through the constant address space, so what is fetched ahead waits in scalar
// registers.
#define GATMI_ITEM_ARRAYS(i_, r_, b_, s_)                                                 \
    {                                                                                     \
        r_ = 0;                                                                           \
        b_ = 0;                                                                           \
        s_ = -1;                                                                          \
        if ((i_) < A.n_items) {                                                           \
            r_ = ld_layout(A.item_row, (i_));                                             \
            b_ = ld_layout(A.item_beg, (i_));                                             \
            s_ = ld_layout(A.item_slot, (i_));                                            \
        }                                                                                 \
    }
#define GATMI_ITEM_END(i_, r_) ((i_) < A.n_items ? ld_layout(A.rowptr, (r_) + 1) : 0)
#define GATMI_ITEM_FIRST_COLS(i_, b_, e_, c_, x_)                                         \
    {                                                                                     \
        const int nb_ = (b_);                                                             \
        const int nn_ = min(G, min((e_), nb_ + A.seg_len) - nb_);                         \
        if ((i_) < A.n_items && nn_ > 0) {                                                \
            const int el_ = nb_ + min(lig, nn_ - 1);                                      \
            c_ = A.col[el_];                                                              \
            if (A.aux) x_ = A.aux[el_];                                                   \
        }                                                                                 \
    }

#define GATMI_ITEM_LOOP_BEGIN                                                             \
    const int lane = threadIdx.x & 63;                                                    \
    const int lig = lane & (G - 1);                                                       \
    const int gbase = lane & ~(G - 1);                                                    \
    const int lph2 = 1 << A.lph2_log;                                                     \
    (void)lph2;                                                                           \
    const Slots<T> S = make_slots<T, G>(lig, A.H, A.C, A.Cp, A.lph2_log);                 \
    const bool vec_ok = (A.C & 3) == 0;                                                   \
    constexpr int GPB = 256 / G;                                                          \
    const int stride = gridDim.x * GPB;                                                   \
    const int grp_ = G == 64 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6)      \
                             : (int)threadIdx.x / G;                                      \
    int it = xcd_contiguous_block() * GPB + grp_;                                         \
    int i0_row, i0_beg, i0_slot, i0_end, i0_col = 0, i0_aux = 0;                          \
    int i1_row, i1_beg, i1_slot, i1_end, i2_row, i2_beg, i2_slot;                         \
    GATMI_ITEM_ARRAYS(it, i0_row, i0_beg, i0_slot)                                        \
    i0_end = GATMI_ITEM_END(it, i0_row);                                                  \
    GATMI_ITEM_FIRST_COLS(it, i0_beg, i0_end, i0_col, i0_aux)                             \
    GATMI_ITEM_ARRAYS(it + stride, i1_row, i1_beg, i1_slot)                               \
    i1_end = GATMI_ITEM_END(it + stride, i1_row);                                         \
    GATMI_ITEM_ARRAYS(it + 2 * stride, i2_row, i2_beg, i2_slot)                           \
    for (; it < A.n_items; it += stride) {                                                \
        const int row = uniform<G>(i0_row);                                               \
        const int beg = uniform<G>(i0_beg);                                               \
        const int slot = uniform<G>(i0_slot);                                             \
        int end = uniform<G>(i0_end);                                                     \
        end = min(end, beg + A.seg_len);                                                  \
        const int first_col = i0_col, first_aux = i0_aux;                                 \
        (void)first_col;                                                                  \
        (void)first_aux;                                                                  \
        i0_row = i1_row;                                                                  \
        i0_beg = i1_beg;                                                                  \
        i0_slot = i1_slot;                                                                \
        i0_end = i1_end;                                                                  \
        GATMI_ITEM_FIRST_COLS(it + stride, i0_beg, i0_end, i0_col, i0_aux)                \
        i1_row = i2_row;                                                                  \
        i1_beg = i2_beg;                                                                  \
        i1_slot = i2_slot;                                                                \
        i1_end = GATMI_ITEM_END(it + 2 * stride, i1_row);                                 \
        GATMI_ITEM_ARRAYS(it + 3 * stride, i2_row, i2_beg, i2_slot)

#define GATMI_ITEM_LOOP_END }

// ---------------------------------------------------------------------------------
// Forward.  Online softmax over the row's edges: the source logit a_src[j] is
// recomputed from the gathered row (a 4-FMA dot + head butterfly), so the only per-edge
// HBM traffic is the col index and the feature row itself.  Rows are gathered in
// batches of U edges, double-buffered: batch b+1 is in flight while batch b is reduced.
// Loads are branch-free: batch slots past the row's end re-read its last edge with
// weight exp(-inf) = 0, idle lanes read offset 0.
// ---------------------------------------------------------------------------------
template <int LPH2_T, int G, int T, class RAW>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const RowArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
#ifndef GATMI_U16
#define GATMI_U16 GATMI_U
#endif
    // edges per batch (two batches are in flight); bf16 rows are half as wide
    constexpr int U = (T == 1) ? (sizeof(RAW) == 8 ? GATMI_U16 : GATMI_U) : (T == 2 ? 2 : 1);
    GATMI_ITEM_LOOP_BEGIN
    float4 att[T], acc[T];
    float adi[T], m[T], ssum[T];
    unsigned offb[T], hbit[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        att[t] = ldc(A.att_src + S.coff[t], S.nval[t], vec_ok);
        adi[t] = S.nval[t] ? A.a_dst[(size_t)row * A.H + S.head[t]] : 0.f;
        m[t] = -INFINITY;
        ssum[t] = 0.f;
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        offb[t] = (unsigned)S.poff[t] * 4u;
        hbit[t] = 1u << S.head[t];
    }
    for (int c = beg; c < end; c += G) {
        const int n = min(G, end - c);
        const int cl = min(lig, n - 1);
        const int cj = c == beg ? first_col : A.col[c + cl];
        const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
        auto load = [&](RAW(&v)[U][T], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = group_bcast<G>(cj, min(k0 + u, n - 1), gbase);
#pragma unroll
                for (int t = 0; t < T; ++t) v[u][t] = ld_slot<G, RAW>(A.hp, j, A.Dp, offb[t]);
            }
        };
        auto reduce = [&](const RAW(&raw)[U][T], int k0) {
            float4 v[U][T];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) v[u][t] = widen(raw[u][t]);
            float z[U][T];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = k0 + u < n;
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float d = head_sum<LPH2_T>(dot4(v[u][t], att[t]), lph2);
                    const float uu = d + adi[t];
                    const float zz = uu > 0.f ? uu : A.neg_slope * uu;
                    z[u][t] = live ? zz : -INFINITY;
                }
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float mn = m[t];
#pragma unroll
                for (int u = 0; u < U; ++u) mn = fmaxf(mn, z[u][t]);
                const float sc = exp_quick(m[t] - mn);
                ssum[t] *= sc;
                scale4(acc[t], sc);
                m[t] = mn;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = k0 + u < n;
                const unsigned sb = (unsigned)group_bcast<G>(kb, min(k0 + u, n - 1), gbase);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float p = live ? exp_quick(z[u][t] - m[t]) : 0.f;
                    ssum[t] += p;
                    fma4(acc[t], (sb & hbit[t]) ? p * A.inv_keep : 0.f, v[u][t]);
                }
            }
        };
        RAW va[U][T], vb[U][T];
        load(va, 0);
        for (int k0 = 0; k0 < n; k0 += 2 * U) {
            load(vb, k0 + U);
            reduce(va, k0);
            load(va, k0 + 2 * U);
            reduce(vb, k0 + U);   // unconditional: keeps the prefetch out of a branch
        }
    }
    if (slot < 0 && A.row_acc) {
        // one source chunk of several: the row's partial triple, normalised by the combine pass
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (S.nval[t]) {
                st4(A.row_acc + (size_t)row * A.Dp + S.poff[t], acc[t]);
                if ((S.poff[t] % A.Cp) == 0) {
                    A.row_m[(size_t)row * A.H + S.head[t]] = m[t];
                    A.row_s[(size_t)row * A.H + S.head[t]] = ssum[t];
                }
            }
        }
    } else if (slot < 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 bia = A.bias ? ldc(A.bias + S.coff[t], S.nval[t], vec_ok)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            const float inv = ssum[t] > 0.f ? 1.0f / ssum[t] : 0.f;
            float4 o;
            o.x = fmaf(acc[t].x, inv, bia.x);
            o.y = fmaf(acc[t].y, inv, bia.y);
            o.z = fmaf(acc[t].z, inv, bia.z);
            o.w = fmaf(acc[t].w, inv, bia.w);
            stc(A.out + (size_t)row * A.H * A.C + S.coff[t], o, S.nval[t], vec_ok);
            if (A.lse && S.nval[t] && (S.poff[t] % A.Cp) == 0)
                A.lse[(size_t)row * A.H + S.head[t]] = m[t] + logf(ssum[t]);
        }
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (S.nval[t]) {
                st4(A.part_acc + (size_t)slot * A.Dp + S.poff[t], acc[t]);
                if ((S.poff[t] % A.Cp) == 0) {
                    A.part_m[(size_t)slot * A.H + S.head[t]] = m[t];
                    A.part_s[(size_t)slot * A.H + S.head[t]] = ssum[t];
                }
            }
        }
    }
    GATMI_ITEM_LOOP_END
}

// Merge of the partial (m, s, acc) triples of rows that were cut into segments.
// One wave per cut row; fixed segment order -> bitwise reproducible.
__global__ __launch_bounds__(256) void attn_fwd_merge_kernel(
    const int *split_row, const int *split_slot, int n_split, int H, int C, int Cp, int Dp,
    const float *part_m, const float *part_s, const float *part_acc, const float *bias,
    float *out, float *lse, float *row_m, float *row_s, float *row_acc) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (w >= n_split) return;
    const int lane = threadIdx.x & 63;
    const int row = split_row[w], s0 = split_slot[w], s1 = split_slot[w + 1];
    const int lph = Cp >> 2;
    for (int q = lane; q < H * lph; q += 64) {
        const int head = q / lph, sub = q % lph;
        float M = -INFINITY;
        for (int s = s0; s < s1; ++s) M = fmaxf(M, part_m[(size_t)s * H + head]);
        float tot = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = s0; s < s1; ++s) {
            const float f = expf(part_m[(size_t)s * H + head] - M);
            tot = fmaf(part_s[(size_t)s * H + head], f, tot);
            fma4(acc, f, ld4(part_acc + (size_t)s * Dp + head * Cp + sub * 4));
        }
        if (row_acc) {      // one source chunk of several: leave the merged triple un-normalised
            st4(row_acc + (size_t)row * Dp + head * Cp + sub * 4, acc);
            if (sub == 0) {
                row_m[(size_t)row * H + head] = M;
                row_s[(size_t)row * H + head] = tot;
            }
            continue;
        }
        const float inv = tot > 0.f ? 1.0f / tot : 0.f;
        const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
        for (int k = 0; k < 4; ++k) {
            const int c = sub * 4 + k;
            if (c < C) {
                const float b = bias ? bias[head * C + c] : 0.f;
                out[(size_t)row * H * C + head * C + c] = fmaf(a4[k], inv, b);
            }
        }
        if (lse && sub == 0) lse[(size_t)row * H + head] = M + logf(tot);
    }
}

// ---------------------------------------------------------------------------------
// Backward.  One heavy pass over the TRANSPOSED CSR (source rows) does all the per-edge
// work, so the projected features are not gathered a second time:
//
//   prep      per node i:   S_i = <grad_out[i], out[i] - bias>  (= sum_e alpha_e d_alpha_e,
//             because out[i] - bias = sum_e alpha~_e h[j]); packs (a_dst, lse, S) per head
//             into a 16-byte record; pads grad_out when C % 4 != 0.
//   edge      per source j (h[j] and a_src[j] row-local), per out-edge e = (j -> i):
//               gather grad_out[i] (one float4 per lane) and node record i
//               alpha  = exp(leaky(a_src[j] + a_dst[i]) - lse[i])
//               d_at   = <grad_out[i,h,:], h[j,h,:]>
//               d_u    = (alpha~ d_at - alpha S_i) * leaky'(u)        -> du_t[e] (src order)
//               d_h[j] += alpha~ grad_out[i];   d_a_src[j] += d_u
//             then d_h[j] += d_a_src[j] att_src
//   dst       per destination i: d_a_dst[i] = sum over in-edges of du_t (light: 4H bytes
//             per edge through d2s)
//   colsum    d_h[i] += d_a_dst[i] att_dst (row-local fix-up) fused with the parameter
//             column sums.
// ---------------------------------------------------------------------------------
// (v[sh], v[sh + 1], ..) with zeros behind: sh = 0 keeps v, sh = 4 clears it
__device__ __forceinline__ float4 shift_out4(const float4 v, int sh) {
    float4 r;
    r.x = sh == 0 ? v.x : (sh == 1 ? v.y : (sh == 2 ? v.z : (sh == 3 ? v.w : 0.f)));
    r.y = sh == 0 ? v.y : (sh == 1 ? v.z : (sh == 2 ? v.w : 0.f));
    r.z = sh == 0 ? v.z : (sh == 1 ? v.w : 0.f);
    r.w = sh == 0 ? v.w : 0.f;
    return r;
}

// Fast form of bwd_prep_kernel for C % 4 == 0 with power-of-two lanes per head and slots per
// row (compact == padded layout): the tables are streamed as flat float4 arrays, 4 in flight
// per thread, the per-head dot is a shuffle butterfly -- no LDS, no barrier in the loop.
// COMPACT: C % 4 != 0, C > 4, no padded copy wanted (gather_compact): the same loop over rows of H*C
// floats, a lane's four channels by one unaligned float4 load (the head's last lane: the last four
// channels, shifted down -- attn_bwd_edge_kernel<.., GC = true> reads grad_out the same way).
template <bool COMPACT>
__global__ __launch_bounds__(256) void bwd_prep_flat_kernel(
    const float *gout, const float *outc, const float *bias, const float *a_dst, const float *lse,
    long long nq /* float4 count = N*Dp/4 */, int H, int C, int lph_log, int nslot, float4 *nodeinfo,
    float *gsum_partial, bf16x4 *g16 /* NULL, or the bf16 copy the edge pass gathers */) {
    const int lph = 1 << lph_log;
    const long long stride = (long long)gridDim.x * 256;   // multiple of nslot
    const int slot = threadIdx.x & (nslot - 1);
    int sh = 0, coff = slot * 4, nslot_log = 0;
    if constexpr (COMPACT) {
        const int head = slot >> lph_log, sub = slot & (lph - 1);
        sh = 4 - min(C - 4 * sub, 4);                       // sub < lph = ceil(C / 4): 1..4 channels are real
        coff = head * C + 4 * sub - sh;
        nslot_log = 31 - __builtin_clz(nslot);
    }
    (void)nslot_log;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) b4 = COMPACT ? shift_out4(ld4_unaligned(bias + coff), sh) : ld4(bias + slot * 4);
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int UN = 8;
    for (long long q0 = (long long)blockIdx.x * 256 + threadIdx.x; q0 - threadIdx.x % 64 < nq;
         q0 += stride * UN) {
        float4 g4[UN], o4[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long q = q0 + u * stride;
            const bool on = q < nq;
            if constexpr (COMPACT) {
                const size_t at = (size_t)(on ? q >> nslot_log : 0) * (size_t)(H * C) + coff;
                g4[u] = ld4_unaligned(gout + at);
                o4[u] = ld4_unaligned(outc + at);
            } else {
                g4[u] = on ? ld4(gout + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                o4[u] = on ? ld4(outc + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long q = q0 + u * stride;
            if constexpr (COMPACT) {
                const bool on = q < nq;
                g4[u] = on ? shift_out4(g4[u], sh) : make_float4(0.f, 0.f, 0.f, 0.f);
                o4[u] = on ? shift_out4(o4[u], sh) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float4 o = o4[u];
            o.x -= b4.x; o.y -= b4.y; o.z -= b4.z; o.w -= b4.w;
            gsum.x += g4[u].x; gsum.y += g4[u].y; gsum.z += g4[u].z; gsum.w += g4[u].w;
            float4 gq = g4[u];
            if (g16) {          // the edge pass sees grad_out rounded to bf16: so does S (d_bias does not)
                const bf16x4 r = narrow(gq);
                if (q < nq) g16[q] = r;
                gq = widen(r);
            }
            float p = dot4(gq, o);
            for (int w = 1; w < lph; w <<= 1) p += __shfl_xor(p, w, 64);
            if (q < nq && (q & (lph - 1)) == 0) {
                const long long grp = q >> lph_log;
                nodeinfo[grp] = make_float4(a_dst[grp], lse[grp], p, 0.f);
            }
        }
    }
    __shared__ float4 red[256];
    red[threadIdx.x] = gsum;
    __syncthreads();
    if (threadIdx.x < nslot) {
        float4 s4 = red[threadIdx.x];
        for (int q = threadIdx.x + nslot; q < 256; q += nslot) {
            const float4 o = red[q];
            s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
        }
        st4(gsum_partial + (size_t)blockIdx.x * nslot * 4 + slot * 4, s4);
    }
}

__global__ __launch_bounds__(256) void bwd_prep_kernel(
    const float *gout, const float *outc, const float *bias, const float *a_dst, const float *lse,
    int N, int H, int C, int Cp, int Dp, float4 *nodeinfo, float *gp,
    float *gsum_partial /* [blocks, Dp]: column sums of grad_out, for d_bias */,
    bf16x4 *g16 /* NULL, or [N, Dp] bf16: the padded copy the edge pass gathers instead of gp */) {
    __shared__ float part[256];
    const int nslot = Dp >> 2;
    const int rows_per_pass = 256 / nslot;
    const int slot = threadIdx.x % nslot, rsub = threadIdx.x / nslot;
    const int lph = Cp >> 2;
    const int head = slot / lph, sub = slot % lph;
    const int coff = head * C + sub * 4;
    int nval = C - sub * 4;
    nval = nval > 4 ? 4 : nval;
    const bool vec_ok = (C & 3) == 0;
    const bool lane_ok = rsub < rows_per_pass;
    const float4 b4 = (bias && lane_ok) ? ldc(bias + coff, nval, vec_ok) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 gsum = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long rows_per_block = ((long long)N + gridDim.x - 1) / gridDim.x;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = min((long long)N, r0 + rows_per_block);
    for (long long base = r0; base < r1; base += rows_per_pass) {
        const long long r = base + rsub;
        const bool on = lane_ok && r < r1;
        float p = 0.f;
        if (on) {
            const float4 g4 = ldc(gout + r * H * C + coff, nval, vec_ok);
            float4 o4 = ldc(outc + r * H * C + coff, nval, vec_ok);
            o4.x -= b4.x; o4.y -= b4.y; o4.z -= b4.z; o4.w -= b4.w;
            gsum.x += g4.x; gsum.y += g4.y; gsum.z += g4.z; gsum.w += g4.w;
            float4 gq = g4;
            if (g16) {
                const bf16x4 rr = narrow(g4);
                g16[r * (Dp >> 2) + slot] = rr;
                gq = widen(rr);
            } else if (gp) {
                st4(gp + r * Dp + slot * 4, g4);
            }
            p = dot4(gq, o4);
        }
        part[threadIdx.x] = p;
        __syncthreads();
        if (on && sub == 0) {
            float sd = 0.f;
            for (int q = 0; q < lph; ++q) sd += part[threadIdx.x + q];
            const long long grp = r * H + head;
            nodeinfo[grp] = make_float4(a_dst[grp], lse[grp], sd, 0.f);
        }
        __syncthreads();
    }
    // block-level column sum of grad_out (fixed order over rsub)
    __shared__ float4 red[256];
    red[threadIdx.x] = gsum;
    __syncthreads();
    if (threadIdx.x < nslot) {
        float4 s4 = red[threadIdx.x];
        for (int q = 1; q < rows_per_pass; ++q) {
            const float4 o = red[threadIdx.x + q * nslot];
            s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
        }
        st4(gsum_partial + (size_t)blockIdx.x * Dp + slot * 4, s4);
    }
}

template <int LPH2_T, int G, int T, class RAW, bool GC = false>
__global__ __launch_bounds__(256) void attn_bwd_edge_kernel(const RowArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
#ifndef GATMI_U16
#define GATMI_U16 GATMI_U
#endif
    // edges per batch (two batches are in flight); bf16 rows are half as wide
    constexpr int U = (T == 1) ? (sizeof(RAW) == 8 ? GATMI_U16 : GATMI_U) : (T == 2 ? 2 : 1);
    GATMI_ITEM_LOOP_BEGIN
    float4 att[T], hj[T], acc[T];
    float asj[T], das[T];
    unsigned offb[T], hbit[T], noff[T];
    int gsh[T];                                                   // GC: elements the lane's load is shifted by
#pragma unroll
    for (int t = 0; t < T; ++t) {
        att[t] = ldc(A.att_src + S.coff[t], S.nval[t], vec_ok);
        hj[t] = S.nval[t] ? ld_own<RAW>(A.hself, (size_t)row, A.Dp, S.poff[t])
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        asj[t] = head_sum<LPH2_T>(dot4(hj[t], att[t]), lph2);
        acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        das[t] = 0.f;
        offb[t] = (unsigned)S.poff[t] * 4u;
        gsh[t] = 0;
        if constexpr (GC) {                                       // idle lanes: the row's first bytes, all shifted out
            gsh[t] = 4 - S.nval[t];
            offb[t] = S.nval[t] ? (unsigned)(S.coff[t] - gsh[t]) * 4u : 0u;
        }
        noff[t] = (unsigned)S.head[t] * 16u;
        hbit[t] = 1u << S.head[t];
    }
    for (int c = beg; c < end; c += G) {
        const int n = min(G, end - c);
        const int cl = min(lig, n - 1);
        const int ci = c == beg ? first_col : A.col[c + cl];
        const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr,
                                      (long long)(c == beg ? first_aux : A.aux[c + cl]), A.H);
        auto load = [&](RAW(&v)[U][T], float4(&q)[U][T], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = group_bcast<G>(ci, min(k0 + u, n - 1), gbase);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    v[u][t] = ld_slot<G, RAW>(A.hp, i, A.hp_row, offb[t]);
                    q[u][t] = ld4_row<G>(A.nodeinfo, i, A.H * 4, noff[t]);
                }
            }
        };
        auto reduce = [&](const RAW(&raw)[U][T], const float4(&q)[U][T], int k0) {
            float4 v[U][T];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    v[u][t] = widen(raw[u][t]);
                    if constexpr (GC) v[u][t] = shift_out4(v[u][t], gsh[t]);
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = k0 + u < n;
                const int kk = min(k0 + u, n - 1);
                const unsigned sb = (unsigned)group_bcast<G>(kb, kk, gbase);
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float d2 = head_sum<LPH2_T>(dot4(v[u][t], hj[t]), lph2);
                    const float uu = asj[t] + q[u][t].x;
                    const float zz = uu > 0.f ? uu : A.neg_slope * uu;
                    const float alpha = live ? exp_quick(zz - q[u][t].y) : 0.f;
                    const float at = (sb & hbit[t]) ? alpha * A.inv_keep : 0.f;
                    const float du = (at * d2 - alpha * q[u][t].z) * (uu > 0.f ? 1.0f : A.neg_slope);
                    fma4(acc[t], at, v[u][t]);
                    das[t] += du;
                    if (live && S.nval[t] && (S.poff[t] % A.Cp) == 0)
                        A.du_t[(size_t)(c + kk) * A.H + S.head[t]] = du;
                }
            }
        };
        RAW va[U][T], vb[U][T];
        float4 qa[U][T], qb[U][T];
        load(va, qa, 0);
        for (int k0 = 0; k0 < n; k0 += 2 * U) {
            load(vb, qb, k0 + U);
            reduce(va, qa, k0);
            load(va, qa, k0 + 2 * U);
            reduce(vb, qb, k0 + U);
        }
    }
    if (slot < 0) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (S.nval[t]) {
                float4 o = acc[t];
                fma4(o, das[t], att[t]);
                st4(A.dhp + (size_t)row * A.Dp + S.poff[t], o);
                if ((S.poff[t] % A.Cp) == 0) A.da_src[(size_t)row * A.H + S.head[t]] = das[t];
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (S.nval[t]) {
                st4(A.part_dh + (size_t)slot * A.Dp + S.poff[t], acc[t]);
                if ((S.poff[t] % A.Cp) == 0) A.part_das[(size_t)slot * A.H + S.head[t]] = das[t];
            }
        }
    }
    GATMI_ITEM_LOOP_END
}

// The same pass with the softmax chain hoisted out of the row loop (full-wave rows of H = 64 / LPH2
// heads: 4x64, 8x32).  Above, every edge costs a second 64-lane load instruction (the destination's
// 16-byte record per head, one address per head group), an exponential and the chain's selects, and
// one store instruction with H live lanes for d_u.  Here the chain of a 64-edge chunk runs
// lane-parallel, one (edge, head) pair per lane and pass: the records arrive in H coalesced
// loads, (a~, a~ slope, -a S slope) go to this wave's LDS table, the row loop reads its head's entry
// (one ds_read_b128), d_u = d2 * c1 + c0 goes back into the table and leaves in H coalesced stores
// when the chunk is done.
template <int LPH2_T, class RAW>
__global__ __launch_bounds__(256) void attn_bwd_edge2_kernel(const RowArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int G = 64, T = 1, H = 64 / LPH2_T, HLOG = (H == 8 ? 3 : (H == 4 ? 2 : (H == 2 ? 1 : 0))),
                  EPP = 64 / H;
    static_assert(H * LPH2_T == 64 && (1 << HLOG) == H, "full-wave rows of 1, 2, 4 or 8 heads");
    constexpr int U = sizeof(RAW) == 8 ? GATMI_U16 : GATMI_U;
    __shared__ float4 cs_all[4][64 * H];
    float4 *cs = cs_all[threadIdx.x >> 6];                       // [edge of the chunk][head]
    GATMI_ITEM_LOOP_BEGIN
    (void)S.coff; (void)vec_ok; (void)gbase; (void)lig;
    const int myhead = lane / LPH2_T;                            // the head whose channels this lane holds
    const int hh = lane & (H - 1);                               // the head of this lane's chain pairs
    const float4 att = ld4(A.att_src + myhead * A.C + (lane % LPH2_T) * 4);
    const float4 hj = ld_own<RAW>(A.hself, (size_t)row, A.Dp, lane * 4);
    const float asj = head_sum<LPH2_T>(dot4(hj, att), lph2);
    const float as_h = __shfl(asj, hh * LPH2_T, 64);             // head hh's source logit
    const unsigned offb = (unsigned)lane * 16u;
    const bool writer = (lane % LPH2_T) == 0;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float das = 0.f;
    for (int c = beg; c < end; c += 64) {
        const int n = min(64, end - c);
        const int cl = min(lane, n - 1);
        const int ci = c == beg ? first_col : A.col[c + cl];
        const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr,
                                      (long long)(c == beg ? first_aux : A.aux[c + cl]), A.H);
        auto load = [&](RAW(&v)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = __builtin_amdgcn_readlane(ci, min(k0 + u, n - 1));
                v[u] = ld_slot<G, RAW>(A.hp, i, A.Dp, offb);
            }
        };
        RAW va[U], vb[U];
        load(va, 0);
        // the chain: pass ps covers edges EPP * ps .. EPP * ps + EPP - 1, lane -> (edge, head hh)
#pragma unroll
        for (int ps = 0; ps < H; ++ps) {
            if (EPP * ps < n) {                                   // wave-uniform
                const int e = EPP * ps + (lane >> HLOG);
                const int ec = min(e, n - 1);
                const int i = __shfl(ci, ec, 64);
                const unsigned sb = (unsigned)__shfl(kb, ec, 64);
                const float4 q = ld4(A.nodeinfo + ((size_t)i * H + hh) * 4);
                const float uu = as_h + q.x;
                const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                const float alpha = e < n ? exp_quick(uu * slope - q.y) : 0.f;
                const float at = ((sb >> hh) & 1u) ? alpha * A.inv_keep : 0.f;
                cs[64 * ps + lane] = make_float4(at, at * slope, -alpha * q.z * slope, 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
        auto reduce = [&](const RAW(&raw)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool live = k0 + u < n;
                const int kk = min(k0 + u, n - 1);
                const float4 v = widen(raw[u]);
                const float d2 = head_sum<LPH2_T>(dot4(v, hj), lph2);
                const float4 cc = cs[kk * H + myhead];
                const float du = live ? fmaf(d2, cc.y, cc.z) : 0.f;
                fma4(acc, live ? cc.x : 0.f, v);
                das += du;
                if (live && writer) cs[kk * H + myhead].w = du;
            }
        };
        for (int k0 = 0; k0 < n; k0 += 2 * U) {
            load(vb, k0 + U);
            reduce(va, k0);
            load(va, k0 + 2 * U);
            reduce(vb, k0 + U);
        }
        __builtin_amdgcn_wave_barrier();
        // d_u of the chunk: [edge][head] is contiguous in du_t
#pragma unroll
        for (int ps = 0; ps < H; ++ps)
            if (EPP * ps + (lane >> HLOG) < n)
                A.du_t[(size_t)(c + EPP * ps) * H + lane] = cs[64 * ps + lane].w;
        __builtin_amdgcn_wave_barrier();
    }
    if (slot < 0) {
        float4 o = acc;
        fma4(o, das, att);
        st4(A.dhp + (size_t)row * A.Dp + lane * 4, o);
        if (writer) A.da_src[(size_t)row * A.H + myhead] = das;
    } else {
        st4(A.part_dh + (size_t)slot * A.Dp + lane * 4, acc);
        if (writer) A.part_das[(size_t)slot * A.H + myhead] = das;
    }
    GATMI_ITEM_LOOP_END
}

// Merge of the partial (d_h, d_a_src) of source rows cut into segments; segment order fixed.
__global__ __launch_bounds__(256) void attn_bwd_edge_merge_kernel(
    const int *split_row, const int *split_slot, int n_split, int H, int C, int Cp, int Dp,
    const float *part_dh, const float *part_das, const float *att_src, float *dhp,
    float *da_src) {
    const int w = (blockIdx.x * 256 + threadIdx.x) >> 6;
    if (w >= n_split) return;
    const int lane = threadIdx.x & 63;
    const int row = split_row[w], s0 = split_slot[w], s1 = split_slot[w + 1];
    const int lph = Cp >> 2;
    for (int q = lane; q < H * lph; q += 64) {
        const int head = q / lph, sub = q % lph;
        float das = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = s0; s < s1; ++s) {
            das += part_das[(size_t)s * H + head];
            const float4 p = ld4(part_dh + (size_t)s * Dp + head * Cp + sub * 4);
            acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
        }
        float o[4] = {acc.x, acc.y, acc.z, acc.w};
        for (int k = 0; k < 4; ++k) {
            const int c = sub * 4 + k;
            if (c < C) o[k] = fmaf(das, att_src[head * C + c], o[k]);
        }
        st4(dhp + (size_t)row * Dp + head * Cp + sub * 4, make_float4(o[0], o[1], o[2], o[3]));
        if (sub == 0) da_src[(size_t)row * H + head] = das;
    }
}

// d_a_dst[i,h] = sum of du_t over the in-edges of i, in dst-CSR order.  One wave per work
// item; lane = (edge slot, head) with the head count padded to a power of two.
__global__ __launch_bounds__(256) void attn_bwd_dst_kernel(
    const int *rowptr, const int *d2s, const int *item_row, const int *item_beg,
    const int *item_slot, int n_items, int seg_len, int H, int hp2_log, const float *du_t,
    float *da_dst, float *part_dad, int accumulate /* source chunks: da_dst += */) {
    const int lane = threadIdx.x & 63;
    const int hp2 = 1 << hp2_log;
    const int h = lane & (hp2 - 1), eo = lane >> hp2_log, epw = 64 >> hp2_log;
    const int waves = gridDim.x * 4;
    for (int it = xcd_contiguous_block() * 4 + (threadIdx.x >> 6); it < n_items; it += waves) {
        const int row = item_row[it], beg = item_beg[it], slot = item_slot[it];
        const int end = min(rowptr[row + 1], beg + seg_len);
        float s = 0.f;
        for (int e = beg + eo; e < end; e += epw)
            if (h < H) s += du_t[(size_t)d2s[e] * H + h];
        // sum over the lanes that share h: strides hp2, 2*hp2, ... 32 -- a fixed tree
        for (int o = hp2; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        if (eo == 0 && h < H) {
            if (slot < 0) da_dst[(size_t)row * H + h] = accumulate ? da_dst[(size_t)row * H + h] + s : s;
            else part_dad[(size_t)slot * H + h] = s;
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dst_merge_kernel(
    const int *split_row, const int *split_slot, int n_split, int H, const float *part_dad,
    float *da_dst, int accumulate) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_split * H) return;
    const int w = idx / H, head = idx % H;
    float acc = 0.f;
    for (int s = split_slot[w]; s < split_slot[w + 1]; ++s) acc += part_dad[(size_t)s * H + head];
    float *o = da_dst + (size_t)split_row[w] * H + head;
    *o = accumulate ? *o + acc : acc;
}

// ---------------------------------------------------------------------------------
// Parameter gradients: column sums over the N nodes.
//   d_att_src[h,c] = sum_n d_a_src[n,h] * h[n,h,c]   d_att_dst likewise   d_bias = sum_n grad_out[n]
// Stage 1: each block sums a stripe of rows into one partial row; stage 2 adds the
// partial rows in block order (reproducible) and writes the compact [H*C] vectors.
// ---------------------------------------------------------------------------------
constexpr int kColsumBlocks = 1024;

template <class RAW>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(
    const void *hp, const float *da_src, const float *da_dst, const float *att_dst,
    float *dhp /* NULL: no fix-up */, int n_src, int n_dst, int dst_off, int H, int C, int Cp,
    int Dp, float *partial /* [blocks, 2, Dp] */) {
    __shared__ float4 red[2][256];
    const int nslot = Dp >> 2;  // float4 slots per row, <= 256
    const int rows_per_pass = 256 / nslot;
    const int slot = threadIdx.x % nslot;
    const int rsub = threadIdx.x / nslot;
    const int head = (slot * 4) / Cp;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    float4 ad4 = a;  // att_dst of this slot (0 in the padding)
    if (dhp) {
        const int c0 = slot * 4 - head * Cp;
        if (c0 + 0 < C) ad4.x = att_dst[head * C + c0 + 0];
        if (c0 + 1 < C) ad4.y = att_dst[head * C + c0 + 1];
        if (c0 + 2 < C) ad4.z = att_dst[head * C + c0 + 2];
        if (c0 + 3 < C) ad4.w = att_dst[head * C + c0 + 3];
    }
    if (rsub < rows_per_pass) {
        const long long n_max = n_src > n_dst ? n_src : n_dst;
        const long long rows_per_block = (n_max + gridDim.x - 1) / gridDim.x;
        const long long r0 = (long long)blockIdx.x * rows_per_block;
        const long long r1 = min(n_max, r0 + rows_per_block);
        const bool same = dst_off == 0;
        long long r = r0 + rsub;
        if (same && n_src == n_dst && !dhp) {
            // the common case (one shard, the d_a_dst term deferred to the GEMMs): four rows in flight
            // per thread, every load issued before the first use
            for (; r + 3 * rows_per_pass < r1; r += 4 * rows_per_pass) {
                float4 hv[4];
                float ws[4], wd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long ru = r + u * rows_per_pass;
                    hv[u] = ld_own<RAW>(hp, (size_t)ru, Dp, slot * 4);
                    ws[u] = da_src[ru * H + head];
                    wd[u] = da_dst[ru * H + head];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fma4(a, ws[u], hv[u]);
                    fma4(b, wd[u], hv[u]);
                }
            }
        }
#pragma unroll 4
        for (; r < r1; r += rows_per_pass) {
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < n_src) {
                hv = ld_own<RAW>(hp, (size_t)r, Dp, slot * 4);
                fma4(a, da_src[r * H + head], hv);
            }
            if (r < n_dst) {
                const float dd = da_dst[r * H + head];
                if (!same || r >= n_src) hv = ld_own<RAW>(hp, (size_t)(r + dst_off), Dp, slot * 4);
                fma4(b, dd, hv);
                if (dhp) {  // d_h[r] += d_a_dst[r] att_dst
                    float4 dh = ld4(dhp + (r + dst_off) * Dp + slot * 4);
                    fma4(dh, dd, ad4);
                    st4(dhp + (r + dst_off) * Dp + slot * 4, dh);
                }
            }
        }
    }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if (threadIdx.x < nslot) {
        for (int k = 0; k < 2; ++k) {
            float4 s = red[k][threadIdx.x];
            for (int q = 1; q < rows_per_pass; ++q) {
                const float4 o = red[k][threadIdx.x + q * nslot];
                s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
            }
            st4(partial + ((size_t)blockIdx.x * 2 + k) * Dp + slot * 4, s);
        }
    }
}

// d_h[i] += d_a_dst[i] att_dst for a block of rows (the partitioned layer applies it
// after the reduce-scatter of d_h).
__global__ __launch_bounds__(256) void add_dst_term_kernel(float *dh, const float *da_dst,
                                                           const float *att_dst, long long n, int H,
                                                           int C, int Cp) {
    const int lph = Cp >> 2;
    const long long total = n * H * lph;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
        const int sub = (int)(i % lph);
        const long long grp = i / lph;  // node * H + head
        const int head = (int)(grp % H);
        const float dd = da_dst[grp];
        float4 v = ld4(dh + grp * Cp + sub * 4);
        float o[4] = {v.x, v.y, v.z, v.w};
        for (int k = 0; k < 4; ++k) {
            const int c = sub * 4 + k;
            if (c < C) o[k] = fmaf(dd, att_dst[head * C + c], o[k]);
        }
        st4(dh + grp * Cp + sub * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// One block per (term k, 16 padded columns): 16 row groups each sum every 16th partial row
// (4 loads in flight), then a fixed-order 16-way add -- reproducible and short.
__global__ __launch_bounds__(256) void colsum_stage2_kernel(
    const float *partial, int nblocks, const float *gsum_partial, int nblocks_g, int H, int C,
    int Cp, int Dp, float *d_att_src, float *d_att_dst, float *d_bias) {
    __shared__ float red[256];
    const int k = blockIdx.y;
    const int pc = blockIdx.x * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;
    float *dst = k == 0 ? d_att_src : (k == 1 ? d_att_dst : d_bias);
    const float *src = k < 2 ? partial + (size_t)k * Dp : gsum_partial;
    const size_t ld = k < 2 ? (size_t)2 * Dp : (size_t)Dp;
    const int nb = k < 2 ? nblocks : nblocks_g;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (dst && pc < Dp) {
        int b = rg;
        for (; b + 48 < nb; b += 64) {
            s0 += src[(size_t)b * ld + pc];
            s1 += src[(size_t)(b + 16) * ld + pc];
            s2 += src[(size_t)(b + 32) * ld + pc];
            s3 += src[(size_t)(b + 48) * ld + pc];
        }
        for (; b < nb; b += 16) s0 += src[(size_t)b * ld + pc];
    }
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && dst && pc < Dp) {
        float tot = 0.f;
        for (int q = 0; q < 16; ++q) tot += red[threadIdx.x + q * 16];
        const int head = pc / Cp, c = pc % Cp;
        if (c < C) dst[head * C + c] = tot;
    }
}

// ---------------------------------------------------------------------------------
// Host side: shape -> kernel instantiation
// ---------------------------------------------------------------------------------
struct Geometry {
    int Cp, Dp, lph, lph2, lph2_log, G, T;
};

static int make_geometry(const gatmi_shape &s, Geometry *g) {
    GATMI_REQUIRE(s.heads >= 1 && s.channels >= 1 && s.num_nodes >= 0 && s.in_features >= 1,
                  "bad layer shape N=%d F=%d H=%d C=%d", s.num_nodes, s.in_features, s.heads,
                  s.channels);
    g->Cp = round_up4(s.channels);
    g->Dp = s.heads * g->Cp;
    g->lph = g->Cp / 4;
    g->lph2 = next_pow2(g->lph);
    g->lph2_log = ilog2(g->lph2);
    const int slots = s.heads * g->lph2;
    if (slots <= 64) {
        g->G = next_pow2(slots);
        g->T = 1;
    } else {
        g->G = 64;
        g->T = (slots + 63) / 64;
        GATMI_REQUIRE(g->T <= 4, "heads*channels too wide: %d x %d (max 1024 padded floats)",
                      s.heads, s.channels);
        g->T = g->T == 3 ? 4 : g->T;
    }
    GATMI_REQUIRE(g->Dp <= 1024, "heads*round_up(channels,4) = %d exceeds 1024", g->Dp);
    return GATMI_OK;
}

enum Kind { FWD, BWD_EDGE };

template <int LPH2_T, int G, int T>
static void launch_kind(Kind k, const RowArgs &A, int grid, hipStream_t st) {
    if constexpr (G == 64 && T == 1 && (LPH2_T == 16 || LPH2_T == 8)) {
        // full-wave rows without idle lanes (4x64, 8x32): the chain runs per chunk, not per edge
        if (k == BWD_EDGE && A.H * LPH2_T == 64 && A.C == LPH2_T * 4 && option("GATMI_BWD_HOIST", 1) != 0) {
            if (A.feat_bf16)
                hipLaunchKernelGGL((attn_bwd_edge2_kernel<LPH2_T, bf16x4>), dim3(grid), dim3(256), 0, st, A);
            else
                hipLaunchKernelGGL((attn_bwd_edge2_kernel<LPH2_T, float4>), dim3(grid), dim3(256), 0, st, A);
            return;
        }
    }
    if constexpr (G == 64) {      // bf16 feature tables: full-wave rows only (make_geometry.G == 64)
        if (A.feat_bf16) {
            if (k == FWD)
                hipLaunchKernelGGL((attn_fwd_kernel<LPH2_T, G, T, bf16x4>), dim3(grid), dim3(256), 0, st, A);
            else
                hipLaunchKernelGGL((attn_bwd_edge_kernel<LPH2_T, G, T, bf16x4>), dim3(grid), dim3(256), 0, st, A);
            return;
        }
    }
    if (k == FWD)
        hipLaunchKernelGGL((attn_fwd_kernel<LPH2_T, G, T, float4>), dim3(grid), dim3(256), 0, st, A);
    else if (A.g_compact)     // its own instantiation: the shifts cost the padded-row kernel a wave of occupancy
        hipLaunchKernelGGL((attn_bwd_edge_kernel<LPH2_T, G, T, float4, true>), dim3(grid), dim3(256), 0, st, A);
    else
        hipLaunchKernelGGL((attn_bwd_edge_kernel<LPH2_T, G, T, float4>), dim3(grid), dim3(256), 0, st, A);
}

static int launch_rows(Kind k, const Geometry &g, const RowArgs &A, hipStream_t st) {
    if (A.n_items == 0) return GATMI_OK;
    // Large graphs: 16384 blocks instead of 2048.  About 1024 blocks are resident, so 2048 ran as two
    // rounds whose second one tails off; many short blocks let the dispatcher even out the
    // power-law rows (products size: forward 13.6 -> 13.0 ms, bf16 rows 7.8 -> 6.9 ms; arxiv
    // size is 3 % slower that way and keeps the small grid).
    const int grid = item_grid(A.n_items, 256 / g.G, A.n_items >= (1 << 20) ? 16384 : 2048);
    if (g.G == 64) {
        if (g.T == 1) {
            if (g.lph2 == 16) launch_kind<16, 64, 1>(k, A, grid, st);
            else if (g.lph2 == 8) launch_kind<8, 64, 1>(k, A, grid, st);
            else launch_kind<0, 64, 1>(k, A, grid, st);
        } else if (g.T == 2) {
            if (g.lph2 == 16) launch_kind<16, 64, 2>(k, A, grid, st);
            else launch_kind<0, 64, 2>(k, A, grid, st);
        } else {
            launch_kind<0, 64, 4>(k, A, grid, st);
        }
    } else {
        // common narrow layers get the DPP butterfly at compile time: one head (G == lanes
        // per head), and 8 heads of <= 8 channels (Cora-shaped layer 1)
        const int key = g.G * 100 + g.lph2;
        switch (key) {
            case 1616: launch_kind<16, 16, 1>(k, A, grid, st); break;
            case 808: launch_kind<8, 8, 1>(k, A, grid, st); break;
            case 404: launch_kind<4, 4, 1>(k, A, grid, st); break;
            case 202: launch_kind<2, 2, 1>(k, A, grid, st); break;
            case 1602: launch_kind<2, 16, 1>(k, A, grid, st); break;
            case 3204: launch_kind<4, 32, 1>(k, A, grid, st); break;
            case 101: launch_kind<1, 1, 1>(k, A, grid, st); break;
            case 801: launch_kind<1, 8, 1>(k, A, grid, st); break;
            default:
                switch (g.G) {
                    case 32: launch_kind<0, 32, 1>(k, A, grid, st); break;
                    case 16: launch_kind<0, 16, 1>(k, A, grid, st); break;
                    case 8: launch_kind<0, 8, 1>(k, A, grid, st); break;
                    case 4: launch_kind<0, 4, 1>(k, A, grid, st); break;
                    case 2: launch_kind<0, 2, 1>(k, A, grid, st); break;
                    default: launch_kind<0, 1, 1>(k, A, grid, st); break;
                }
        }
    }
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

static void fill_common(RowArgs &A, const gatmi_attn &a, const Geometry &g) {
    memset(&A, 0, sizeof(A));
    A.N = a.shape.num_nodes;
    A.H = a.shape.heads;
    A.C = a.shape.channels;
    A.Cp = g.Cp;
    A.Dp = g.Dp;
    A.lph2_log = g.lph2_log;
    A.neg_slope = a.negative_slope;
    A.p_drop = a.dropout_p;
    A.inv_keep = a.dropout_p < 1.0f ? 1.0f / (1.0f - a.dropout_p) : 0.0f;
    if (a.mask_mode == GATMI_MASK_NONE) A.inv_keep = 1.0f;
    A.keep_thr = (uint32_t)((double)a.dropout_p * 16777216.0);
    A.mask_mode = a.mask_mode;
    A.feat_bf16 = a.feat_bf16 ? 1 : 0;
    A.mask = a.mask;
    A.seed = a.seed;
    A.hp = a.hp;
    A.hp_row = g.Dp;
    A.a_dst = a.a_dst;
    A.att_src = a.att_src;
    A.att_dst = a.att_dst;
    A.bias = a.bias;
}

static int check_attn(const gatmi_graph *g, const gatmi_attn *a) {
    GATMI_REQUIRE(g && a, "null graph/attn descriptor");
    GATMI_REQUIRE(g->num_dst == a->shape.num_nodes, "graph has %d destination rows, layer %d",
                  g->num_dst, a->shape.num_nodes);
    GATMI_REQUIRE(g->num_src >= g->num_dst && g->dst_offset >= 0 &&
                      g->dst_offset + g->num_dst <= g->num_src,
                  "destination rows [%d, %d) outside the %d source rows", g->dst_offset,
                  g->dst_offset + g->num_dst, g->num_src);
    GATMI_REQUIRE(g->seg_len >= 64 && g->seg_len % 64 == 0, "seg_len %d must be a multiple of 64",
                  g->seg_len);
    GATMI_REQUIRE(a->mask_mode >= 0 && a->mask_mode <= 3, "bad mask_mode %d", a->mask_mode);
    GATMI_REQUIRE((a->mask_mode != GATMI_MASK_BYTES && a->mask_mode != GATMI_MASK_HASH_DEV) || a->mask,
                  "mask_mode BYTES / HASH_DEV needs `mask` (the keep flags / the device seed)");
    GATMI_REQUIRE(a->dropout_p >= 0.f && a->dropout_p <= 1.f, "dropout_p %f outside [0,1]",
                  a->dropout_p);
    GATMI_REQUIRE(a->mask_mode == GATMI_MASK_NONE || a->shape.heads <= 32,
                  "attention dropout supports at most 32 heads, got %d", a->shape.heads);
    GATMI_REQUIRE(a->hp && a->a_dst && a->att_src, "null hp/a_dst/att_src");
    GATMI_REQUIRE(!a->feat_bf16 || gatmi_attn_bf16_supported(&a->shape),
                  "bf16 feature tables need rows of more than 32 float4 slots (heads * "
                  "next_pow2(round_up(channels, 4) / 4) > 32): got %d x %d", a->shape.heads,
                  a->shape.channels);
    return GATMI_OK;
}

}  // namespace gatmi

using namespace gatmi;

namespace gatmi {
__global__ void keep_mask_kernel(uint8_t *out, long long total, int H, uint64_t seed, float p) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = keep_hash((uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)((double)p * 16777216.0),
                           (uint32_t)(i / H), (uint32_t)(i % H));
}
}  // namespace gatmi

extern "C" int gatmi_dropout_mask(const gatmi_graph *g, int heads, float dropout_p, uint64_t seed,
                                  uint8_t *mask, void *stream) {
    GATMI_REQUIRE(g && mask && heads >= 1, "null graph/mask");
    const long long total = (long long)g->num_edges * heads;
    if (total == 0) return GATMI_OK;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(keep_mask_kernel, dim3(blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), mask, total, heads, seed, dropout_p);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" size_t gatmi_attn_fwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s) {
    Geometry geo;
    if (!g || !s || make_geometry(*s, &geo)) return 0;
    const size_t slots = (size_t)g->plan.n_slots;
    return align_up(slots * s->heads * 4) * 2 + align_up(slots * geo.Dp * 4) + 256;
}

// Body shared by gatmi_attn_fwd (workspace from the caller) and gatmi_attn_fwd_static (workspace
// in this library's own device array): arguments are already checked, `ws` holds at least
// gatmi_attn_fwd_workspace_bytes(g, &a->shape).
static int attn_fwd_run(const gatmi_graph *g, const gatmi_attn *a, const Geometry &geo, float *out,
                        float *lse, void *ws, hipStream_t st) {
    RowArgs A;
    fill_common(A, *a, geo);
    A.rowptr = g->rowptr;
    A.col = g->col;
    A.item_row = g->plan.item_row;
    A.item_beg = g->plan.item_beg;
    A.item_slot = g->plan.item_slot;
    A.n_items = g->plan.n_items;
    A.seg_len = g->seg_len;
    A.out = out;
    A.lse = lse;
    const size_t slots = (size_t)g->plan.n_slots;
    Carver cv(ws);
    A.part_m = cv.take<float>(slots * A.H);
    A.part_s = cv.take<float>(slots * A.H);
    A.part_acc = cv.take<float>(slots * geo.Dp);
    {
        ProfScope ps("attn_fwd", st);
        if (int e = launch_rows(FWD, geo, A, st)) return e;
    }
    if (g->plan.n_split_rows > 0) {
        ProfScope ps("attn_fwd_merge", st);
        const int blocks = (g->plan.n_split_rows + 3) / 4;
        hipLaunchKernelGGL(attn_fwd_merge_kernel, dim3(blocks), dim3(256), 0, st,
                           g->plan.split_row, g->plan.split_slot, g->plan.n_split_rows, A.H, A.C,
                           A.Cp, A.Dp, A.part_m, A.part_s, A.part_acc, A.bias, out, lse, A.row_m,
                           A.row_s, A.row_acc);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

extern "C" int gatmi_attn_fwd(const gatmi_graph *g, const gatmi_attn *a, float *out, float *lse,
                              void *ws, size_t ws_bytes, void *stream) {
    if (int e = check_attn(g, a)) return e;
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    GATMI_REQUIRE(out, "null out");
    if (ws_bytes < gatmi_attn_fwd_workspace_bytes(g, &a->shape)) {
        set_error("attn_fwd workspace too small: %zu < %zu", ws_bytes,
                  gatmi_attn_fwd_workspace_bytes(g, &a->shape));
        return GATMI_ERR_WS;
    }
    return attn_fwd_run(g, a, geo, out, lse, ws, static_cast<hipStream_t>(stream));
}

// ---- The forward for callers that cannot allocate (HIP graph capture, serving loops): the
// partial (m, s, acc) triples of cut rows -- the only scratch the forward uses -- live in a
// device array of this library instead of a caller's workspace.  Every GPU the library runs on
// holds its own copy of the array at its own address; `scratch` below is only the host-side
// handle of the symbol, the address kernels may use comes from hipGetSymbolAddress, once per
// device, and is kept in g_scratch_ptr.
static __device__ __align__(256) unsigned char scratch[GATMI_STATIC_SCRATCH_BYTES];

namespace {
std::mutex g_scratch_mu;
void *g_scratch_ptr[kMaxDevices] = {};

int static_scratch(void **ptr) {
    int dev = -1;
    GATMI_HIP(hipGetDevice(&dev));
    GATMI_REQUIRE(dev >= 0 && dev < kMaxDevices, "device ordinal %d outside [0, %d)", dev,
                  kMaxDevices);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        GATMI_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        GATMI_REQUIRE(p && ((uintptr_t)p & 255) == 0, "static scratch of device %d at %p", dev, p);
        g_scratch_ptr[dev] = p;
    }
    *ptr = g_scratch_ptr[dev];
    return GATMI_OK;
}
}  // namespace

extern "C" int gatmi_attn_fwd_static(const gatmi_graph *g, const gatmi_attn *a, float *out,
                                     float *lse, void *stream) {
    if (int e = check_attn(g, a)) return e;
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    GATMI_REQUIRE(out, "null out");
    const size_t need = gatmi_attn_fwd_workspace_bytes(g, &a->shape);
    if (need > (size_t)GATMI_STATIC_SCRATCH_BYTES) {
        set_error("gatmi_attn_fwd_static: %d cut-row segments of %d floats need %zu bytes of "
                  "scratch, the static array holds %zu; use gatmi_attn_fwd with a workspace or a "
                  "larger seg_len", g->plan.n_slots, geo.Dp + 2 * a->shape.heads, need,
                  (size_t)GATMI_STATIC_SCRATCH_BYTES);
        return GATMI_ERR_WS;
    }
    void *ws = nullptr;
    if (int e = static_scratch(&ws)) return e;
    // nothing in the scratch has to start at zero: every slot the merge reads is written by the
    // segment that owns it in the same call
    return attn_fwd_run(g, a, geo, out, lse, ws, static_cast<hipStream_t>(stream));
}

namespace {
struct BwdWs {
    float *nodeinfo, *du_t, *gp, *da_dst, *da_src, *part_dad, *part_das, *part_dh, *colsum;
    bf16x4 *g16;
    size_t bytes;
};
// C % 4 != 0 in f32: the edge pass gathers grad_out rows as they are (RowArgs::g_compact) instead of a
// head-padded copy.  The copy is a third of the prep kernel's bytes (run.py's conv2, 8 x 30 at 10^6 nodes
// and 6 edges a row: prep 0.78 -> 0.47 ms), but rows that are no multiple of a line cost the gather half
// a line more per edge (edge pass 1.80 -> 1.92 ms): a gain up to about 12 edges a row, hence the default
// for graphs of at most 8 -- the reference's 5-neighbour graphs.  GATMI_GCOMPACT=0: always the padded
// copy, 2: never.
static bool gather_compact(const gatmi_graph *g, int C) {
    const int mode = option("GATMI_GCOMPACT", 1);
    if ((C & 3) == 0 || C <= 4 || mode == 0) return false;
    return mode == 2 || (long long)g->num_edges <= 8ll * g->num_dst;
}

BwdWs carve_bwd(const gatmi_graph *g, const gatmi_shape *s, const Geometry &geo, void *ws,
                bool feat_bf16) {
    Carver cv(ws);
    BwdWs w;
    const size_t N = (size_t)g->num_dst, NS = (size_t)g->num_src, H = (size_t)s->heads;
    w.nodeinfo = cv.take<float>(N * H * 4);
    w.du_t = cv.take<float>((size_t)g->num_edges * H);
    // the grad_out rows the edge pass gathers: grad_out itself, a padded f32 copy (C % 4 != 0),
    // or -- bf16 feature tables -- a padded bf16 copy at half the bytes
    w.gp = (!feat_bf16 && (s->channels & 3) && !gather_compact(g, s->channels)) ? cv.take<float>(N * geo.Dp) : nullptr;
    w.g16 = feat_bf16 ? cv.take<bf16x4>(N * geo.Dp / 4) : nullptr;
    w.da_dst = cv.take<float>(N * H);
    w.da_src = cv.take<float>(NS * H);
    w.part_dad = cv.take<float>((size_t)g->plan.n_slots * H);
    w.part_das = cv.take<float>((size_t)g->plan_t.n_slots * H);
    w.part_dh = cv.take<float>((size_t)g->plan_t.n_slots * geo.Dp);
    w.colsum = cv.take<float>((size_t)kColsumBlocks * 3 * geo.Dp);  // [blocks,2,Dp] + [blocks,Dp]
    w.bytes = cv.off + 256;
    return w;
}
}  // namespace

extern "C" size_t gatmi_attn_bwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s) {
    Geometry geo;
    if (!g || !s || make_geometry(*s, &geo)) return 0;
    return carve_bwd(g, s, geo, nullptr, false).bytes;
}

extern "C" size_t gatmi_attn_bwd_workspace_bytes_bf16(const gatmi_graph *g, const gatmi_shape *s) {
    Geometry geo;
    if (!g || !s || make_geometry(*s, &geo)) return 0;
    return carve_bwd(g, s, geo, nullptr, true).bytes;
}

extern "C" int gatmi_attn_bf16_supported(const gatmi_shape *s) {
    Geometry geo;
    return s && make_geometry(*s, &geo) == GATMI_OK && geo.G == 64;
}

extern "C" int gatmi_attn_bwd(const gatmi_graph *g, const gatmi_attn *a, const float *out,
                              const float *lse, const float *grad_out, float *d_hp,
                              float *d_att_src, float *d_att_dst, float *d_bias,
                              float *d_a_dst_out, void *ws, size_t ws_bytes, void *stream) {
    if (int e = check_attn(g, a)) return e;
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    GATMI_REQUIRE(out && lse && grad_out && d_hp && a->att_dst, "null out/lse/grad_out/d_hp/att_dst");
    const bool bf = a->feat_bf16 != 0;
    const size_t need = bf ? gatmi_attn_bwd_workspace_bytes_bf16(g, &a->shape)
                           : gatmi_attn_bwd_workspace_bytes(g, &a->shape);
    if (ws_bytes < need) {
        set_error("attn_bwd workspace too small: %zu < %zu", ws_bytes, need);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    BwdWs w = carve_bwd(g, &a->shape, geo, ws, bf);
    const int N = g->num_dst, NS = g->num_src, H = a->shape.heads, C = a->shape.channels;
    GATMI_REQUIRE(d_a_dst_out || N == NS,
                  "a destination partition must take d_a_dst (the d_h fix-up follows its reduce-scatter)");
    if (NS == 0) return GATMI_OK;
    // the caller's d_a_dst (destination rows owned elsewhere) is written in place: no copy node in
    // a captured step
    if (d_a_dst_out) w.da_dst = d_a_dst_out;
    const int prep_blocks = N < kColsumBlocks ? (N > 0 ? N : 1) : kColsumBlocks;
    float *gsum_partial = w.colsum + (size_t)kColsumBlocks * 2 * geo.Dp;

    {
        ProfScope ps("attn_bwd_prep", st);
        const bool pow2 = geo.lph == geo.lph2 && geo.lph <= 64 &&
                          ((geo.Dp / 4) & (geo.Dp / 4 - 1)) == 0 && geo.Dp / 4 <= 256;
        const bool flat = (C & 3) == 0 && pow2;
        const bool flat_compact = (C & 3) != 0 && pow2 && !w.gp && !w.g16;   // gather_compact
        if (N > 0 && flat) {
            hipLaunchKernelGGL(bwd_prep_flat_kernel<false>, dim3(prep_blocks), dim3(256), 0, st, grad_out,
                               out, a->bias, a->a_dst, lse, (long long)N * (geo.Dp / 4), H, C,
                               geo.lph2_log, geo.Dp / 4, reinterpret_cast<float4 *>(w.nodeinfo),
                               gsum_partial, w.g16);
            GATMI_HIP(hipGetLastError());
        } else if (N > 0 && flat_compact) {
            hipLaunchKernelGGL(bwd_prep_flat_kernel<true>, dim3(prep_blocks), dim3(256), 0, st, grad_out,
                               out, a->bias, a->a_dst, lse, (long long)N * (geo.Dp / 4), H, C,
                               geo.lph2_log, geo.Dp / 4, reinterpret_cast<float4 *>(w.nodeinfo),
                               gsum_partial, nullptr);
            GATMI_HIP(hipGetLastError());
        } else if (N > 0) {
            hipLaunchKernelGGL(bwd_prep_kernel, dim3(prep_blocks), dim3(256), 0, st, grad_out, out,
                               a->bias, a->a_dst, lse, N, H, C, geo.Cp, geo.Dp,
                               reinterpret_cast<float4 *>(w.nodeinfo), w.gp, gsum_partial, w.g16);
            GATMI_HIP(hipGetLastError());
        }
    }
    const void *gpad = w.g16 ? static_cast<const void *>(w.g16)
                             : static_cast<const void *>(w.gp ? w.gp : grad_out);

    RowArgs B;
    fill_common(B, *a, geo);
    B.hp = gpad;
    if (!w.g16 && !w.gp && (C & 3)) {        // gather_compact: grad_out's own rows
        B.hp_row = H * C;
        B.g_compact = 1;
    }
    B.hself = a->hp;
    B.nodeinfo = w.nodeinfo;
    B.rowptr = g->rowptr_t;
    B.col = g->col_t;
    B.aux = g->pos_t;
    B.item_row = g->plan_t.item_row;
    B.item_beg = g->plan_t.item_beg;
    B.item_slot = g->plan_t.item_slot;
    B.n_items = g->plan_t.n_items;
    B.seg_len = g->seg_len;
    B.du_t = w.du_t;
    B.dhp = d_hp;
    B.da_src = w.da_src;
    B.part_das = w.part_das;
    B.part_dh = w.part_dh;
    {
        ProfScope ps("attn_bwd_edge", st);
        if (int e = launch_rows(BWD_EDGE, geo, B, st)) return e;
    }
    if (g->plan_t.n_split_rows > 0) {
        ProfScope ps("attn_bwd_edge_merge", st);
        const int blocks = (g->plan_t.n_split_rows + 3) / 4;
        hipLaunchKernelGGL(attn_bwd_edge_merge_kernel, dim3(blocks), dim3(256), 0, st,
                           g->plan_t.split_row, g->plan_t.split_slot, g->plan_t.n_split_rows, H, C,
                           geo.Cp, geo.Dp, w.part_dh, w.part_das, a->att_src, d_hp, w.da_src);
        GATMI_HIP(hipGetLastError());
    }
    {
        ProfScope ps("attn_bwd_dst", st);
        const int hp2_log = ilog2(next_pow2(H));
        GATMI_REQUIRE(hp2_log <= 6, "more than 64 heads");
        const int grid = item_grid(g->plan.n_items, 4);
        hipLaunchKernelGGL(attn_bwd_dst_kernel, dim3(grid), dim3(256), 0, st, g->rowptr, g->d2s,
                           g->plan.item_row, g->plan.item_beg, g->plan.item_slot, g->plan.n_items,
                           g->seg_len, H, hp2_log, w.du_t, w.da_dst, w.part_dad, 0);
        GATMI_HIP(hipGetLastError());
        if (g->plan.n_split_rows > 0) {
            const int tot = g->plan.n_split_rows * H;
            hipLaunchKernelGGL(attn_bwd_dst_merge_kernel, dim3((tot + 255) / 256), dim3(256), 0, st,
                               g->plan.split_row, g->plan.split_slot, g->plan.n_split_rows, H,
                               w.part_dad, w.da_dst, 0);
            GATMI_HIP(hipGetLastError());
        }
    }
    {
        ProfScope ps("param_colsum", st);
        const int blocks = NS < kColsumBlocks ? NS : kColsumBlocks;
        if (bf)
            hipLaunchKernelGGL(colsum_stage1_kernel<bf16x4>, dim3(blocks), dim3(256), 0, st, a->hp,
                               w.da_src, w.da_dst, a->att_dst, d_a_dst_out ? nullptr : d_hp, NS, N,
                               g->dst_offset, H, C, geo.Cp, geo.Dp, w.colsum);
        else
            hipLaunchKernelGGL(colsum_stage1_kernel<float4>, dim3(blocks), dim3(256), 0, st, a->hp,
                               w.da_src, w.da_dst, a->att_dst, d_a_dst_out ? nullptr : d_hp, NS, N,
                               g->dst_offset, H, C, geo.Cp, geo.Dp, w.colsum);
        GATMI_HIP(hipGetLastError());
        if (d_att_src || d_att_dst || d_bias) {
            hipLaunchKernelGGL(colsum_stage2_kernel, dim3((geo.Dp + 15) / 16, 3), dim3(256), 0, st,
                               w.colsum, blocks, gsum_partial, N > 0 ? prep_blocks : 0, H, C,
                               geo.Cp, geo.Dp, d_att_src, d_att_dst, d_bias);
            GATMI_HIP(hipGetLastError());
        }
    }
    return GATMI_OK;
}

extern "C" int gatmi_add_dst_term(const gatmi_shape *s, float *d_hp, const float *d_a_dst,
                                  const float *att_dst, void *stream) {
    GATMI_REQUIRE(s && d_hp && d_a_dst && att_dst, "null argument");
    Geometry geo;
    if (int e = make_geometry(*s, &geo)) return e;
    const long long total = (long long)s->num_nodes * s->heads * geo.lph;
    if (total == 0) return GATMI_OK;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(add_dst_term_kernel, dim3(blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), d_hp, d_a_dst, att_dst,
                       (long long)s->num_nodes, s->heads, s->channels, geo.Cp);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

// ---------------------------------------------------------------------------------
// Source chunks: the same layer when the feature table arrives in pieces (a pipelined
// all-gather on a destination partition, SURVEY.md 8e "chunk the all-gather by source range").
// Chunk k has a rectangular layout of its own -- the edges whose sources lie in the chunk,
// sources renumbered from the chunk's first row -- and the kernels above run on it unchanged:
//   forward   gatmi_attn_fwd_partial per chunk: the un-normalised online-softmax triple
//             (m, s, acc) of every destination row over the chunk's edges;
//             gatmi_attn_fwd_combine: the triples of all chunks -> out, lse (the merge of cut
//             rows, across chunks; chunk order fixed -> bitwise reproducible)
//   backward  gatmi_attn_bwd_begin (per-node preparation, once), gatmi_attn_bwd_chunk per chunk
//             (edge pass over the chunk's transposed CSR -> d_h of the chunk's source rows,
//             complete, ready for its reduce-scatter; d_a_dst accumulated), gatmi_attn_bwd_end
//             (d_att_src, d_att_dst, d_bias).
// ---------------------------------------------------------------------------------
namespace gatmi {

struct CombineArgs {
    const float *m[GATMI_MAX_PARTS], *s[GATMI_MAX_PARTS], *acc[GATMI_MAX_PARTS];
    int n_parts;
};

// One thread per float4 slot of a destination row; the per-head scalars are re-read by the
// lanes of a head (cache hits).
__global__ __launch_bounds__(256) void attn_fwd_combine_kernel(
    const CombineArgs P, long long total /* rows * Dp / 4 */, int H, int C, int Cp, int Dp,
    const float *bias, float *out, float *lse) {
    const int nslot = Dp >> 2, lph = Cp >> 2;
    for (long long idx = blockIdx.x * 256ll + threadIdx.x; idx < total; idx += gridDim.x * 256ll) {
        const long long row = idx / nslot;
        const int q = (int)(idx - row * nslot);
        const int head = q / lph, sub = q - head * lph;
        float M = -INFINITY;
        for (int k = 0; k < P.n_parts; ++k) M = fmaxf(M, P.m[k][row * H + head]);
        float tot = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < P.n_parts; ++k) {
            const float mk = P.m[k][row * H + head];
            if (mk == -INFINITY) continue;          // no edge of this row in chunk k
            const float f = expf(mk - M);
            tot = fmaf(P.s[k][row * H + head], f, tot);
            fma4(acc, f, ld4(P.acc[k] + row * Dp + q * 4));
        }
        const float inv = tot > 0.f ? 1.0f / tot : 0.f;
        const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
        for (int k = 0; k < 4; ++k) {
            const int c = sub * 4 + k;
            if (c < C) {
                const float b = bias ? bias[head * C + c] : 0.f;
                out[row * H * C + head * C + c] = fmaf(a4[k], inv, b);
            }
        }
        if (lse && sub == 0) lse[row * H + head] = M + logf(tot);
    }
}

// Checks shared by the chunk entry points: as check_attn, but a chunk's sources are a slice of
// the table that need not hold the destination rows (dst_offset is not used).
static int check_chunk(const gatmi_graph *g, const gatmi_attn *a) {
    GATMI_REQUIRE(g && a, "null graph/attn descriptor");
    GATMI_REQUIRE(g->num_dst == a->shape.num_nodes, "graph has %d destination rows, layer %d",
                  g->num_dst, a->shape.num_nodes);
    GATMI_REQUIRE(g->seg_len >= 64 && g->seg_len % 64 == 0, "seg_len %d must be a multiple of 64",
                  g->seg_len);
    GATMI_REQUIRE(a->mask_mode >= 0 && a->mask_mode <= 3, "bad mask_mode %d", a->mask_mode);
    GATMI_REQUIRE((a->mask_mode != GATMI_MASK_BYTES && a->mask_mode != GATMI_MASK_HASH_DEV) || a->mask,
                  "mask_mode BYTES / HASH_DEV needs `mask` (the keep flags / the device seed)");
    GATMI_REQUIRE(a->dropout_p >= 0.f && a->dropout_p <= 1.f, "dropout_p %f outside [0,1]",
                  a->dropout_p);
    GATMI_REQUIRE(a->mask_mode == GATMI_MASK_NONE || a->shape.heads <= 32,
                  "attention dropout supports at most 32 heads, got %d", a->shape.heads);
    GATMI_REQUIRE((a->hp || g->num_src == 0) && a->a_dst && a->att_src, "null hp/a_dst/att_src");
    GATMI_REQUIRE(!a->feat_bf16 || gatmi_attn_bf16_supported(&a->shape),
                  "bf16 feature tables need rows of more than 32 float4 slots: got %d x %d",
                  a->shape.heads, a->shape.channels);
    return GATMI_OK;
}

// Per-node state of a chunked backward: what gatmi_attn_bwd keeps at the head of its workspace,
// sized by the destination rows alone so that every chunk call finds it at the same place.
struct NodeWs {
    float *nodeinfo, *gp, *gsum_partial, *colsum;
    bf16x4 *g16;
    size_t bytes;
};
static NodeWs carve_node(const gatmi_shape *s, const Geometry &geo, int max_chunks, bool feat_bf16,
                         void *ws) {
    Carver cv(ws);
    NodeWs w;
    const size_t N = (size_t)s->num_nodes, H = (size_t)s->heads;
    w.nodeinfo = cv.take<float>(N * H * 4);
    w.gp = (!feat_bf16 && (s->channels & 3)) ? cv.take<float>(N * geo.Dp) : nullptr;
    w.g16 = feat_bf16 ? cv.take<bf16x4>(N * geo.Dp / 4) : nullptr;
    w.gsum_partial = cv.take<float>((size_t)kColsumBlocks * geo.Dp);
    // column-sum partials: one [blocks, 2, Dp] slab per chunk (d_att_src) + one for d_att_dst
    w.colsum = cv.take<float>((size_t)(max_chunks + 1) * kColsumBlocks * 2 * geo.Dp);
    w.bytes = cv.off + 256;
    return w;
}

struct ChunkWs {
    float *du_t, *da_src, *part_dad, *part_das, *part_dh;
    size_t bytes;
};
static ChunkWs carve_chunk(const gatmi_graph *g, const gatmi_shape *s, const Geometry &geo, void *ws) {
    Carver cv(ws);
    ChunkWs w;
    const size_t H = (size_t)s->heads;
    w.du_t = cv.take<float>((size_t)g->num_edges * H);
    w.da_src = cv.take<float>((size_t)g->num_src * H);
    w.part_dad = cv.take<float>((size_t)g->plan.n_slots * H);
    w.part_das = cv.take<float>((size_t)g->plan_t.n_slots * H);
    w.part_dh = cv.take<float>((size_t)g->plan_t.n_slots * geo.Dp);
    w.bytes = cv.off + 256;
    return w;
}

}  // namespace gatmi

extern "C" int gatmi_attn_fwd_partial(const gatmi_graph *g, const gatmi_attn *a,
                                      const gatmi_attn_partial *p, void *ws, size_t ws_bytes,
                                      void *stream) {
    if (int e = check_chunk(g, a)) return e;
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    GATMI_REQUIRE(p && p->m && p->s && p->acc, "null partial (m, s, acc)");
    if (ws_bytes < gatmi_attn_fwd_workspace_bytes(g, &a->shape)) {
        set_error("attn_fwd_partial workspace too small: %zu < %zu", ws_bytes,
                  gatmi_attn_fwd_workspace_bytes(g, &a->shape));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    RowArgs A;
    fill_common(A, *a, geo);
    A.rowptr = g->rowptr;
    A.col = g->col;
    A.item_row = g->plan.item_row;
    A.item_beg = g->plan.item_beg;
    A.item_slot = g->plan.item_slot;
    A.n_items = g->plan.n_items;
    A.seg_len = g->seg_len;
    A.row_m = p->m;
    A.row_s = p->s;
    A.row_acc = p->acc;
    const size_t slots = (size_t)g->plan.n_slots;
    Carver cv(ws);
    A.part_m = cv.take<float>(slots * A.H);
    A.part_s = cv.take<float>(slots * A.H);
    A.part_acc = cv.take<float>(slots * geo.Dp);
    {
        ProfScope ps("attn_fwd", st);
        if (int e = launch_rows(FWD, geo, A, st)) return e;
    }
    if (g->plan.n_split_rows > 0) {
        ProfScope ps("attn_fwd_merge", st);
        const int blocks = (g->plan.n_split_rows + 3) / 4;
        hipLaunchKernelGGL(attn_fwd_merge_kernel, dim3(blocks), dim3(256), 0, st,
                           g->plan.split_row, g->plan.split_slot, g->plan.n_split_rows, A.H, A.C,
                           A.Cp, A.Dp, A.part_m, A.part_s, A.part_acc, A.bias, nullptr, nullptr,
                           A.row_m, A.row_s, A.row_acc);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

extern "C" int gatmi_attn_fwd_combine(const gatmi_shape *s, int n_parts,
                                      const gatmi_attn_partial *parts, const float *bias,
                                      float *out, float *lse, void *stream) {
    GATMI_REQUIRE(s && parts && out, "null shape/parts/out");
    GATMI_REQUIRE(n_parts >= 1 && n_parts <= GATMI_MAX_PARTS, "n_parts %d outside [1, %d]", n_parts,
                  GATMI_MAX_PARTS);
    Geometry geo;
    if (int e = make_geometry(*s, &geo)) return e;
    CombineArgs P;
    memset(&P, 0, sizeof(P));
    P.n_parts = n_parts;
    for (int k = 0; k < n_parts; ++k) {
        GATMI_REQUIRE(parts[k].m && parts[k].s && parts[k].acc, "null partial %d", k);
        P.m[k] = parts[k].m;
        P.s[k] = parts[k].s;
        P.acc[k] = parts[k].acc;
    }
    const long long total = (long long)s->num_nodes * (geo.Dp / 4);
    if (total == 0) return GATMI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps("attn_fwd_combine", st);
    const long long want = (total + 255) / 256;
    const int blocks = (int)(want > 16384 ? 16384 : want);
    hipLaunchKernelGGL(attn_fwd_combine_kernel, dim3(blocks), dim3(256), 0, st, P, total, s->heads,
                       s->channels, geo.Cp, geo.Dp, bias, out, lse);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" size_t gatmi_attn_bwd_node_workspace_bytes(const gatmi_shape *s, int max_chunks,
                                                      int feat_bf16) {
    Geometry geo;
    if (!s || max_chunks < 1 || max_chunks > GATMI_MAX_PARTS || make_geometry(*s, &geo)) return 0;
    return carve_node(s, geo, max_chunks, feat_bf16 != 0, nullptr).bytes;
}

extern "C" size_t gatmi_attn_bwd_chunk_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s) {
    Geometry geo;
    if (!g || !s || make_geometry(*s, &geo)) return 0;
    return carve_chunk(g, s, geo, nullptr).bytes;
}

extern "C" int gatmi_attn_bwd_begin(const gatmi_attn *a, int max_chunks, const float *out,
                                    const float *lse, const float *grad_out, void *node_ws,
                                    size_t node_ws_bytes, void *stream) {
    GATMI_REQUIRE(a && out && lse && grad_out && a->a_dst, "null attn/out/lse/grad_out/a_dst");
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    const bool bf = a->feat_bf16 != 0;
    GATMI_REQUIRE(!bf || gatmi_attn_bf16_supported(&a->shape), "bf16 feature tables do not cover %d x %d",
                  a->shape.heads, a->shape.channels);
    const size_t need = gatmi_attn_bwd_node_workspace_bytes(&a->shape, max_chunks, bf);
    if (need == 0 || node_ws_bytes < need) {
        set_error("attn_bwd node workspace too small (or max_chunks %d outside [1, %d]): %zu < %zu",
                  max_chunks, GATMI_MAX_PARTS, node_ws_bytes, need);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    NodeWs w = carve_node(&a->shape, geo, max_chunks, bf, node_ws);
    const int N = a->shape.num_nodes, H = a->shape.heads, C = a->shape.channels;
    // every column-sum slab reads as zero until its chunk (or the end call) fills it: chunks a
    // caller skips (no rows, no edges) then contribute nothing
    GATMI_HIP(hipMemsetAsync(w.colsum, 0,
                             (size_t)(max_chunks + 1) * kColsumBlocks * 2 * geo.Dp * sizeof(float), st));
    if (N == 0) return GATMI_OK;
    const int prep_blocks = N < kColsumBlocks ? N : kColsumBlocks;
    ProfScope ps("attn_bwd_prep", st);
    const bool flat = (C & 3) == 0 && geo.lph == geo.lph2 && geo.lph <= 64 &&
                      ((geo.Dp / 4) & (geo.Dp / 4 - 1)) == 0 && geo.Dp / 4 <= 256;
    if (flat)
        hipLaunchKernelGGL(bwd_prep_flat_kernel<false>, dim3(prep_blocks), dim3(256), 0, st, grad_out, out,
                           a->bias, a->a_dst, lse, (long long)N * (geo.Dp / 4), H, C, geo.lph2_log,
                           geo.Dp / 4, reinterpret_cast<float4 *>(w.nodeinfo), w.gsum_partial, w.g16);
    else
        hipLaunchKernelGGL(bwd_prep_kernel, dim3(prep_blocks), dim3(256), 0, st, grad_out, out,
                           a->bias, a->a_dst, lse, N, H, C, geo.Cp, geo.Dp,
                           reinterpret_cast<float4 *>(w.nodeinfo), w.gp, w.gsum_partial, w.g16);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" int gatmi_attn_bwd_chunk(const gatmi_graph *g, const gatmi_attn *a, int chunk,
                                    int max_chunks, int accumulate, const float *grad_out,
                                    void *node_ws, size_t node_ws_bytes, float *d_hp, float *d_a_dst,
                                    void *ws, size_t ws_bytes, void *stream) {
    if (int e = check_chunk(g, a)) return e;
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    GATMI_REQUIRE(grad_out && d_a_dst && (d_hp || g->num_src == 0), "null grad_out/d_a_dst/d_hp");
    GATMI_REQUIRE(max_chunks >= 1 && max_chunks <= GATMI_MAX_PARTS && chunk >= 0 && chunk < max_chunks,
                  "chunk %d of %d (at most %d)", chunk, max_chunks, GATMI_MAX_PARTS);
    const bool bf = a->feat_bf16 != 0;
    const size_t need_node = gatmi_attn_bwd_node_workspace_bytes(&a->shape, max_chunks, bf);
    const size_t need = gatmi_attn_bwd_chunk_workspace_bytes(g, &a->shape);
    if (node_ws_bytes < need_node || ws_bytes < need) {
        set_error("attn_bwd_chunk workspace too small: node %zu < %zu or chunk %zu < %zu",
                  node_ws_bytes, need_node, ws_bytes, need);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    NodeWs nw = carve_node(&a->shape, geo, max_chunks, bf, node_ws);
    ChunkWs w = carve_chunk(g, &a->shape, geo, ws);
    const int N = g->num_dst, NS = g->num_src, H = a->shape.heads, C = a->shape.channels;
    float *colsum = nw.colsum + (size_t)chunk * kColsumBlocks * 2 * geo.Dp;
    const int cs_blocks = NS < kColsumBlocks ? NS : kColsumBlocks;
    if (NS == 0 || N == 0) {
        // an empty chunk contributes nothing (its column-sum slab was zeroed by the begin call);
        // without destination rows the chunk's d_h is all zeros, and it is reduced over the ranks
        if (N == 0 && NS > 0) GATMI_HIP(hipMemsetAsync(d_hp, 0, (size_t)NS * geo.Dp * sizeof(float), st));
        if (!accumulate && N > 0) GATMI_HIP(hipMemsetAsync(d_a_dst, 0, (size_t)N * H * sizeof(float), st));
        return GATMI_OK;
    }
    const void *gpad = nw.g16 ? static_cast<const void *>(nw.g16)
                              : static_cast<const void *>(nw.gp ? nw.gp : grad_out);
    RowArgs B;
    fill_common(B, *a, geo);
    B.hp = gpad;
    B.hself = a->hp;
    B.nodeinfo = nw.nodeinfo;
    B.rowptr = g->rowptr_t;
    B.col = g->col_t;
    B.aux = g->pos_t;
    B.item_row = g->plan_t.item_row;
    B.item_beg = g->plan_t.item_beg;
    B.item_slot = g->plan_t.item_slot;
    B.n_items = g->plan_t.n_items;
    B.seg_len = g->seg_len;
    B.du_t = w.du_t;
    B.dhp = d_hp;
    B.da_src = w.da_src;
    B.part_das = w.part_das;
    B.part_dh = w.part_dh;
    {
        ProfScope ps("attn_bwd_edge", st);
        if (int e = launch_rows(BWD_EDGE, geo, B, st)) return e;
    }
    if (g->plan_t.n_split_rows > 0) {
        ProfScope ps("attn_bwd_edge_merge", st);
        const int blocks = (g->plan_t.n_split_rows + 3) / 4;
        hipLaunchKernelGGL(attn_bwd_edge_merge_kernel, dim3(blocks), dim3(256), 0, st,
                           g->plan_t.split_row, g->plan_t.split_slot, g->plan_t.n_split_rows, H, C,
                           geo.Cp, geo.Dp, w.part_dh, w.part_das, a->att_src, d_hp, w.da_src);
        GATMI_HIP(hipGetLastError());
    }
    {
        ProfScope ps("attn_bwd_dst", st);
        const int hp2_log = ilog2(next_pow2(H));
        GATMI_REQUIRE(hp2_log <= 6, "more than 64 heads");
        const int grid = item_grid(g->plan.n_items, 4);
        hipLaunchKernelGGL(attn_bwd_dst_kernel, dim3(grid), dim3(256), 0, st, g->rowptr, g->d2s,
                           g->plan.item_row, g->plan.item_beg, g->plan.item_slot, g->plan.n_items,
                           g->seg_len, H, hp2_log, w.du_t, d_a_dst, w.part_dad, accumulate ? 1 : 0);
        GATMI_HIP(hipGetLastError());
        if (g->plan.n_split_rows > 0) {
            const int tot = g->plan.n_split_rows * H;
            hipLaunchKernelGGL(attn_bwd_dst_merge_kernel, dim3((tot + 255) / 256), dim3(256), 0, st,
                               g->plan.split_row, g->plan.split_slot, g->plan.n_split_rows, H,
                               w.part_dad, d_a_dst, accumulate ? 1 : 0);
            GATMI_HIP(hipGetLastError());
        }
    }
    {
        // sum_j d_a_src[j] h[j] over the chunk's source rows (n_dst = 0: no destination term here)
        ProfScope ps("param_colsum", st);
        if (bf)
            hipLaunchKernelGGL(colsum_stage1_kernel<bf16x4>, dim3(cs_blocks), dim3(256), 0, st, a->hp,
                               w.da_src, d_a_dst, a->att_dst, nullptr, NS, 0, 0, H, C, geo.Cp, geo.Dp,
                               colsum);
        else
            hipLaunchKernelGGL(colsum_stage1_kernel<float4>, dim3(cs_blocks), dim3(256), 0, st, a->hp,
                               w.da_src, d_a_dst, a->att_dst, nullptr, NS, 0, 0, H, C, geo.Cp, geo.Dp,
                               colsum);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

extern "C" int gatmi_attn_bwd_end(const gatmi_attn *a, int max_chunks, const float *d_a_dst,
                                  void *node_ws, size_t node_ws_bytes, float *d_att_src,
                                  float *d_att_dst, float *d_bias, void *stream) {
    GATMI_REQUIRE(a && d_a_dst && (a->hp || a->shape.num_nodes == 0), "null attn/d_a_dst/hp");
    Geometry geo;
    if (int e = make_geometry(a->shape, &geo)) return e;
    const bool bf = a->feat_bf16 != 0;
    const size_t need = gatmi_attn_bwd_node_workspace_bytes(&a->shape, max_chunks, bf);
    if (need == 0 || node_ws_bytes < need) {
        set_error("attn_bwd node workspace too small: %zu < %zu", node_ws_bytes, need);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    NodeWs nw = carve_node(&a->shape, geo, max_chunks, bf, node_ws);
    const int N = a->shape.num_nodes, H = a->shape.heads, C = a->shape.channels;
    float *colsum = nw.colsum + (size_t)max_chunks * kColsumBlocks * 2 * geo.Dp;
    ProfScope ps("param_colsum", st);
    // sum_i d_a_dst[i] h[i] over the OWN rows (a->hp = their table): slab max_chunks, n_src = 0
    const int blocks = N < kColsumBlocks ? N : kColsumBlocks;
    if (N > 0) {
        if (bf)
            hipLaunchKernelGGL(colsum_stage1_kernel<bf16x4>, dim3(blocks), dim3(256), 0, st, a->hp,
                               d_a_dst, d_a_dst, a->att_dst, nullptr, 0, N, 0, H, C, geo.Cp, geo.Dp,
                               colsum);
        else
            hipLaunchKernelGGL(colsum_stage1_kernel<float4>, dim3(blocks), dim3(256), 0, st, a->hp,
                               d_a_dst, d_a_dst, a->att_dst, nullptr, 0, N, 0, H, C, geo.Cp, geo.Dp,
                               colsum);
        GATMI_HIP(hipGetLastError());
    }
    if (d_att_src || d_att_dst || d_bias) {
        const int prep_blocks = N < kColsumBlocks ? N : kColsumBlocks;
        hipLaunchKernelGGL(colsum_stage2_kernel, dim3((geo.Dp + 15) / 16, 3), dim3(256), 0, st,
                           nw.colsum, (max_chunks + 1) * kColsumBlocks, nw.gsum_partial, prep_blocks, H,
                           C, geo.Cp, geo.Dp, d_att_src, d_att_dst, d_bias);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

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
#ifndef GATMI_E8_UF      // 8 heads, rows of 17..32 float4: instructions (of two edges) per bank, forward / backward
#define GATMI_E8_UF 2
#endif
#ifndef GATMI_E8_UB
#define GATMI_E8_UB 2
#endif
#ifndef GATMI_E8_OCC     // blocks per CU the 8-head backward is compiled for
#define GATMI_E8_OCC 3
#endif
#include "attn_device.h"
#include "common.h"
#include <stdlib.h>

namespace gatmi {

// packed-f32 helpers: v_pk_fma_f32 does two lanes-worth of fp32 FMA per issue slot, and the
// edge loops below are VALU-issue bound (an x row is short), so the 4-wide row arithmetic is
// written on 2-vectors explicitly (the build runs with -fno-slp-vectorize)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 lo2(const f4 v) { return __builtin_shufflevector(v, v, 0, 1); }
__device__ __forceinline__ f2 hi2(const f4 v) { return __builtin_shufflevector(v, v, 2, 3); }
__device__ __forceinline__ void pk_fma4(f4 &acc, float s, const f4 v) {
    const f2 s2 = {s, s};
    const f2 a = __builtin_elementwise_fma(s2, lo2(v), lo2(acc));
    const f2 b = __builtin_elementwise_fma(s2, hi2(v), hi2(acc));
    acc = __builtin_shufflevector(a, b, 0, 1, 2, 3);
}
__device__ __forceinline__ void pk_dot4(f2 &d, const f4 a, const f4 b) {
    d = __builtin_elementwise_fma(lo2(a), lo2(b), d);
    d = __builtin_elementwise_fma(hi2(a), hi2(b), d);
}
__device__ __forceinline__ void pk_scale4(f4 &acc, float s) {
    const f2 s2 = {s, s};
    const f2 a = lo2(acc) * s2, b = hi2(acc) * s2;
    acc = __builtin_shufflevector(a, b, 0, 1, 2, 3);
}
__device__ __forceinline__ f4 ldv4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ void stv4(float *p, const f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ f4 zero4() { return f4{0.f, 0.f, 0.f, 0.f}; }

// 16 bytes / 4 bytes of row j of the packed table at a lane byte offset (see ld4_row)
template <int G>
__device__ __forceinline__ f4 ldv4_row(const float *table, int j, int row_floats, unsigned off_bytes) {
    const char *rowp = reinterpret_cast<const char *>(table) + (size_t)j * (size_t)row_floats * 4;
    if constexpr (G == 64) {
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(rowp), 0, row_floats * 4, 0x00020000);
        return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_bytes, 0, 0));
    } else {
        return *reinterpret_cast<const f4 *>(rowp + off_bytes);
    }
}
template <int G>
__device__ __forceinline__ float ld1_row(const float *table, int j, int row_floats, unsigned off_bytes) {
    const char *rowp = reinterpret_cast<const char *>(table) + (size_t)j * (size_t)row_floats * 4;
    if constexpr (G == 64) {
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(rowp), 0, row_floats * 4, 0x00020000);
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off_bytes, 0, 0));
    } else {
        return *reinterpret_cast<const float *>(rowp + off_bytes);
    }
}

struct InArgs {
    const int *rowptr, *col, *item_row, *item_beg, *item_slot;
    int n_items, seg_len;
    int H, Fp, Fs, lph, lh_log, dst_off;
    float neg_slope, inv_keep;
    uint32_t keep_thr;
    int mask_mode;
    const uint8_t *mask;
    uint64_t seed;
    const float *xs;     // [num_src, Fs]: x row (Fp floats, zero padded) | a_src (H floats) | pad
    const void *xs16;    // bf16 copy of that table in 256-byte rows (inagg_table16_kernel), or null
    const float *a_dst;  // [num_dst, H]
    // forward
    float *z, *lse, *part_m, *part_s, *part_acc;
    // backward
    const float *q, *zin, *lsein;
    const float *sin;    // [num_dst, H] S = <q, Z> per (row, head) when the products pass made it (inagg_bwdg only), or null
    float *part_y;       // [waves][2][H*Fp]
    // streaming kernels (LDS-DMA ring): chunk queue
    int *queue;
    const int *chunk_item;   // [n_chunks + 1]
    int n_chunks, E;
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
    _Pragma("unroll") for (int t = 0; t < T; ++t) {                                       \
        const int sub = sl + t * lh;                                                      \
        ok[t] = hok && sub < A.lph;                                                       \
        offb[t] = ok[t] ? (unsigned)sub * 16u : 0u;                                       \
        roff[t] = ok[t] ? head * A.Fp + sub * 4 : 0;                                      \
    }                                                                                     \
    const unsigned aoff = (unsigned)(A.Fp + head) * 4u;                                   \
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
        const float adi = A.a_dst[(size_t)row * A.H + head];

// ---------------------------------------------------------------------------------
// Forward: online softmax over the row's edges, Z[i,h,:] = sum_e alpha~_e x_j, lse.
// Same pipeline as attn_fwd_kernel (double-buffered branch-free gathers); the source logit
// rides in the gathered row (one dword per lane at the end of it).
// ---------------------------------------------------------------------------------
template <int LH_T, int G, int T>
__global__ __launch_bounds__(256) void inagg_fwd_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int U = (T == 1) ? 4 : (T == 2 ? GATMI_INAGG_U2 : 2);
    INAGG_PROLOGUE
    INAGG_ITEM_LOOP
        float m = -INFINITY, ssum = 0.f;
        f4 acc[T];
#pragma unroll
        for (int t = 0; t < T; ++t) acc[t] = zero4();
        for (int c = beg; c < end; c += G) {
            const int n = min(G, end - c);
            const int cl = min(lig, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](f4(&v)[U][T], float(&as)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = group_bcast<G>(cj, min(k0 + u, n - 1), gbase);
#pragma unroll
                    for (int t = 0; t < T; ++t) v[u][t] = ldv4_row<G>(A.xs, j, A.Fs, offb[t]);
                    as[u] = ld1_row<G>(A.xs, j, A.Fs, aoff);
                }
            };
            auto reduce = [&](const f4(&v)[U][T], const float(&as)[U], int k0) {
                float z[U];
                float mn = m;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float uu = as[u] + adi;
                    const float zz = uu > 0.f ? uu : A.neg_slope * uu;
                    z[u] = (k0 + u < n) ? zz : -INFINITY;
                    mn = fmaxf(mn, z[u]);
                }
                const float sc = exp_quick(m - mn);
                ssum *= sc;
#pragma unroll
                for (int t = 0; t < T; ++t) pk_scale4(acc[t], sc);
                m = mn;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool live = k0 + u < n;
                    const unsigned sb = (unsigned)group_bcast<G>(kb, min(k0 + u, n - 1), gbase);
                    const float p = live ? exp_quick(z[u] - m) : 0.f;
                    ssum += p;
                    const float w = (sb & hbit) ? p * A.inv_keep : 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t) pk_fma4(acc[t], w, v[u][t]);
                }
            };
            f4 va[U][T], vb[U][T];
            float sa[U], sb_[U];
            load(va, sa, 0);
            for (int k0 = 0; k0 < n; k0 += 2 * U) {
                load(vb, sb_, k0 + U);
                reduce(va, sa, k0);
                load(va, sa, k0 + 2 * U);
                reduce(vb, sb_, k0 + U);
            }
        }
        if (slot < 0) {
            const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (ok[t]) {
                    f4 o = acc[t];
                    pk_scale4(o, inv);
                    stv4(A.z + (size_t)row * HF + roff[t], o);
                }
            if (A.lse && hok && sl == 0) A.lse[(size_t)row * A.H + head] = m + logf(ssum);
        } else {
#pragma unroll
            for (int t = 0; t < T; ++t)
                if (ok[t]) stv4(A.part_acc + (size_t)slot * HF + roff[t], acc[t]);
            if (hok && sl == 0) {
                A.part_m[(size_t)slot * A.H + head] = m;
                A.part_s[(size_t)slot * A.H + head] = ssum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// Streaming forward (G == 64).  The register pipeline above keeps every gathered row four
// times in VGPRs (once per head group), which caps the bytes in flight per CU at ~40 KB and
// the kernel at ~3 TB/s for 400-B rows.  Here the rows travel global -> LDS by LDS-DMA
// (global_load_lds_dwordx4, one copy per row, no VGPRs) through a per-wave ring of 1-KiB
// slots; the head groups read their pieces back with ds_read_b128.
//
//   chunk     the work items that begin inside one bucket of kChunkEdges consecutive CSR
//             edges (table built per launch by inagg_chunk_table_kernel), claimed from an
//             atomic queue and walked in groups of <= kChunkItems items; a group's edges
//             are one contiguous range [cb, ce) of the CSR.  Bucketing by edges bounds the
//             serial work of a wave (a hub's 512-edge segments are 1-2 per chunk)
//   producer  walks that range: one DMA instruction (a "stage") moves rpi = 64 / (Fs/4) rows,
//             lane (er, ep) -> piece ep of the row of edge pe + er; stages restart at item
//             boundaries.  col indices come from 64-entry windows DMA'd into LDS one window
//             ahead, so no ordinary load result is ever needed while rows are in flight
//             (hipcc drains the DMA queue at such a use)
//   consumer  per item and stage: tops the ring up, waits with a COUNTED vmcnt for its stage
//             (loads retire in order; counting only the stages issued after it under-counts
//             the younger operations, which only waits longer), reads the rows from LDS
// Every loop is bounded by the chunk's item/edge counts; there is no spin-wait.
// ---------------------------------------------------------------------------------
#ifndef GATMI_RING
#define GATMI_RING 4
#endif
constexpr int kChunkItems = 32;       // items per group (lane-parallel metadata: < 64)
constexpr int kChunkEdgesLog = 9;     // bucket of 512 edges
constexpr int kColWin = 4;   // col windows in the LDS ring (power of two, >= 4)
constexpr int kStreamMaxHeads = 8;

__device__ __forceinline__ void dma16(const void *g, void *lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_uniform, 16,
                                     0, 0);
}
__device__ __forceinline__ void dma4(const void *g, void *lds_wave_uniform) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_uniform, 4,
                                     0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm_imm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// wait until at most n (wave-uniform) vector-memory operations are outstanding
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: wait_vm_imm<0>(); break;
        case 1: wait_vm_imm<1>(); break;
        case 2: wait_vm_imm<2>(); break;
        case 3: wait_vm_imm<3>(); break;
        case 4: wait_vm_imm<4>(); break;
        case 5: wait_vm_imm<5>(); break;
        case 6: wait_vm_imm<6>(); break;
        case 7: wait_vm_imm<7>(); break;
        case 8: wait_vm_imm<8>(); break;
        case 9: wait_vm_imm<9>(); break;
        case 10: wait_vm_imm<10>(); break;
        case 11: wait_vm_imm<11>(); break;
        case 12: wait_vm_imm<12>(); break;
        case 13: wait_vm_imm<13>(); break;
        case 14: wait_vm_imm<14>(); break;
        default: wait_vm_imm<15>(); break;
    }
}
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Shared by the streaming forward and backward: chunk claim, metadata, col windows, producer.
#define STREAM_PROLOGUE                                                                   \
    constexpr int G = 64, R = GATMI_RING, CH = kChunkItems;                               \
    __shared__ __attribute__((aligned(16))) char s_ring[4][R * 1024];                     \
    __shared__ int s_col[4][kColWin * 64];                                                \
    INAGG_PROLOGUE                                                                        \
    (void)gbase; (void)stride; (void)lig;                                                 \
    const int wv = threadIdx.x >> 6;                                                      \
    char *ring = s_ring[wv];                                                              \
    int *colr = s_col[wv];                                                                \
    const int nf4 = A.Fs >> 2, rpi = 64 / nf4;                                            \
    const int er = lane / nf4, ep = lane - er * nf4;                                      \
    const bool dma_on = er < rpi;                                                         \
    const unsigned rstride = (unsigned)nf4 * 16u;                                         \
    const size_t row_bytes = (size_t)A.Fs * 4;

#define STREAM_CLAIM_CHUNK                                                                \
    int k = 0;                                                                            \
    if (lane == 0) k = atomicAdd(A.queue, 1);                                             \
    k = __builtin_amdgcn_readfirstlane(k);                                                \
    if (k >= A.n_chunks) break;                                                           \
    const int ck0 = __builtin_amdgcn_readfirstlane(A.chunk_item[k]);                      \
    const int ck1 = __builtin_amdgcn_readfirstlane(A.chunk_item[k + 1]);

#define STREAM_GROUP_META                                                                 \
    const int ni = min(CH, ck1 - i0);                                                     \
    const int il = i0 + lane;                                                             \
    const int vbeg = lane <= ni ? (il < A.n_items ? A.item_beg[il] : A.E) : 0;            \
    const int vrow = lane < ni ? A.item_row[il] : 0;                                      \
    const int vslot = lane < ni ? A.item_slot[il] : -1;                                   \
    const int cb = __builtin_amdgcn_readlane(vbeg, 0);                                    \
    const int ce = __builtin_amdgcn_readlane(vbeg, ni);

// chunk_item[k] = first work item that begins at or after edge k << kChunkEdgesLog
// (chunk_item[n_chunks] = n_items); items are in CSR order, so a chunk is a contiguous range.
// Also resets the chunk queue of the kernel that follows (a plain store from a kernel node: a
// hipMemsetAsync node in front of it was seen NOT to take effect in replays of a captured HIP graph).
__global__ void inagg_chunk_table_kernel(const int *item_beg, int n_items, int n_chunks,
                                         int *chunk_item, int *queue) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *queue = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n_items; i += gridDim.x * blockDim.x) {
        const int b = i < n_items ? item_beg[i] >> kChunkEdgesLog : n_chunks;
        const int bp = i > 0 ? item_beg[i - 1] >> kChunkEdgesLog : -1;
        for (int k = bp + 1; k <= b; ++k) chunk_item[k] = i;
    }
}

template <int LH_T, int T>
__global__ __launch_bounds__(256) void inagg_fwd_stream_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    STREAM_PROLOGUE
    __shared__ float s_adst[4][kChunkItems * kStreamMaxHeads];
    float *adl = s_adst[wv];
    for (;;) {
        STREAM_CLAIM_CHUNK
        for (int i0 = ck0; i0 < ck1; i0 += CH) {
        STREAM_GROUP_META
        for (int t0 = 0; t0 < ni * A.H; t0 += 64) {
            const int t = t0 + lane;
            const int itx = min(t / A.H, ni - 1);
            const int r = __shfl(vrow, itx, 64);
            if (t < ni * A.H) adl[t] = A.a_dst[(size_t)r * A.H + (t - itx * A.H)];
        }
        // col windows 0 and 1
        int wiss = 0;
        auto issue_win = [&]() {
            const int e = min(cb + wiss * 64 + lane, A.E - 1);
            dma4(A.col + e, colr + (wiss & (kColWin - 1)) * 64);
            ++wiss;
        };
        if (ce > cb) {
            issue_win();
            if (ce - cb > 64) issue_win();
        }
        wait_vm_imm<0>();
        // producer
        int Pst = 0, Cst = 0, pi = 0, pe = cb, pend = __builtin_amdgcn_readlane(vbeg, 1);
        int wcur = 0, st_since = 0;
        auto produce = [&]() {
            while (pe >= pend) {
                ++pi;
                if (pi >= ni) return;
                pend = __builtin_amdgcn_readlane(vbeg, pi + 1);
            }
            const int ehi = min(pe + rpi, pend) - 1;
            const int wneed = (ehi - cb) >> 6;
            if (wneed > wcur) {          // first stage that reads window wcur + 1
                wait_vm(st_since);       // stages issued after that window's DMA
                wcur = wneed;
                if (wiss * 64 < ce - cb) {
                    issue_win();
                    st_since = 0;
                }
            }
            wait_lds();                  // reads of the slot about to be refilled have landed
            const int e = min(pe + er, ehi);
            const int j = colr[(e - cb) & (kColWin * 64 - 1)];
            const char *g = reinterpret_cast<const char *>(A.xs) + (size_t)j * row_bytes + ep * 16;
            if (dma_on) dma16(g, ring + (Pst & (R - 1)) * 1024);
            ++Pst;
            ++st_since;
            pe = ehi + 1;
        };
        for (int ci = 0; ci < ni; ++ci) {
            const int row = __builtin_amdgcn_readlane(vrow, ci);
            const int beg = __builtin_amdgcn_readlane(vbeg, ci);
            const int end = __builtin_amdgcn_readlane(vbeg, ci + 1);
            const int slot = __builtin_amdgcn_readlane(vslot, ci);
            const float adi = adl[ci * A.H + head];
            float m = -INFINITY, ssum = 0.f;
            f4 acc[T];
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = zero4();
            int kbase = beg - 64, kb = 0;
            for (int e0 = beg; e0 < end; e0 += rpi) {
                while (Pst - Cst < R && pi < ni) produce();
                wait_vm(Pst - Cst - 1);
                const char *sb = ring + (Cst & (R - 1)) * 1024;
                const int cnt = min(rpi, end - e0);
                for (int r = 0; r < cnt; r += 2) {
                    const int e = e0 + r;
                    if (e - kbase >= 64) {
                        kbase = e;
                        kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr,
                                            (long long)min(e + lane, end - 1), A.H);
                    }
                    const bool two = r + 1 < cnt;
                    const char *r0 = sb + r * rstride;
                    const char *r1 = two ? r0 + rstride : r0;
                    f4 v0[T], v1[T];
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        v0[t] = *reinterpret_cast<const f4 *>(r0 + offb[t]);
                        v1[t] = *reinterpret_cast<const f4 *>(r1 + offb[t]);
                    }
                    const float as0 = *reinterpret_cast<const float *>(r0 + aoff);
                    const float as1 = *reinterpret_cast<const float *>(r1 + aoff);
                    const float u0 = as0 + adi, u1 = as1 + adi;
                    const float z0 = u0 > 0.f ? u0 : A.neg_slope * u0;
                    const float z1 = two ? (u1 > 0.f ? u1 : A.neg_slope * u1) : -INFINITY;
                    const float mn = fmaxf(m, fmaxf(z0, z1));
                    const float sc = exp_quick(m - mn);
                    const float p0 = exp_quick(z0 - mn);
                    const float p1 = two ? exp_quick(z1 - mn) : 0.f;
                    ssum = fmaf(ssum, sc, p0 + p1);
                    // the second edge of the pair may sit in the next 64-edge mask window
                    unsigned sb0 = (unsigned)__builtin_amdgcn_readlane(kb, e - kbase);
                    unsigned sb1 = sb0;
                    if (two) {
                        if (e + 1 - kbase >= 64) {
                            kbase = e + 1;
                            kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr,
                                                (long long)min(e + 1 + lane, end - 1), A.H);
                        }
                        sb1 = (unsigned)__builtin_amdgcn_readlane(kb, e + 1 - kbase);
                    }
                    const float w0 = (sb0 & hbit) ? p0 * A.inv_keep : 0.f;
                    const float w1 = (sb1 & hbit) ? p1 * A.inv_keep : 0.f;
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        pk_scale4(acc[t], sc);
                        pk_fma4(acc[t], w0, v0[t]);
                        pk_fma4(acc[t], w1, v1[t]);
                    }
                    m = mn;
                }
                ++Cst;
            }
            if (slot < 0) {
                const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (ok[t]) {
                        f4 o = acc[t];
                        pk_scale4(o, inv);
                        stv4(A.z + (size_t)row * HF + roff[t], o);
                    }
                if (A.lse && hok && sl == 0) A.lse[(size_t)row * A.H + head] = m + logf(ssum);
            } else {
#pragma unroll
                for (int t = 0; t < T; ++t)
                    if (ok[t]) stv4(A.part_acc + (size_t)slot * HF + roff[t], acc[t]);
                if (hok && sl == 0) {
                    A.part_m[(size_t)slot * A.H + head] = m;
                    A.part_s[(size_t)slot * A.H + head] = ssum;
                }
            }
        }
        wait_lds();
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
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int U = (T == 1) ? 4 : (T == 2 ? GATMI_INAGG_U2 : 2);
    INAGG_PROLOGUE
    f4 ys[T], yd[T];
#pragma unroll
    for (int t = 0; t < T; ++t) ys[t] = yd[t] = zero4();
    INAGG_ITEM_LOOP
        (void)slot;
        f4 q4[T];
        f2 s2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < T; ++t) {
            q4[t] = ok[t] ? ldv4(A.q + (size_t)row * HF + roff[t]) : zero4();
            const f4 z4 = ok[t] ? ldv4(A.zin + (size_t)row * HF + roff[t]) : zero4();
            pk_dot4(s2, q4[t], z4);
        }
        const float S = head_sum<LH_T>(s2.x + s2.y, lh);
        const float lse_i = A.lsein[(size_t)row * A.H + head];
        float dad = 0.f;
        for (int c = beg; c < end; c += G) {
            const int n = min(G, end - c);
            const int cl = min(lig, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](f4(&v)[U][T], float(&as)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = group_bcast<G>(cj, min(k0 + u, n - 1), gbase);
#pragma unroll
                    for (int t = 0; t < T; ++t) v[u][t] = ldv4_row<G>(A.xs, j, A.Fs, offb[t]);
                    as[u] = ld1_row<G>(A.xs, j, A.Fs, aoff);
                }
            };
            auto reduce = [&](const f4(&v)[U][T], const float(&as)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    f2 d = {0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < T; ++t) pk_dot4(d, v[u][t], q4[t]);
                    const float d2 = head_sum<LH_T>(d.x + d.y, lh);
                    const bool live = k0 + u < n;
                    const unsigned sb = (unsigned)group_bcast<G>(kb, min(k0 + u, n - 1), gbase);
                    const float uu = as[u] + adi;
                    const float zz = uu > 0.f ? uu : A.neg_slope * uu;
                    const float alpha = live ? exp_quick(zz - lse_i) : 0.f;
                    const float at = (sb & hbit) ? alpha * A.inv_keep : 0.f;
                    const float du = (at * d2 - alpha * S) * (uu > 0.f ? 1.0f : A.neg_slope);
                    dad += du;
#pragma unroll
                    for (int t = 0; t < T; ++t) pk_fma4(ys[t], du, v[u][t]);
                }
            };
            f4 va[U][T], vb[U][T];
            float sa[U], sb_[U];
            load(va, sa, 0);
            for (int k0 = 0; k0 < n; k0 += 2 * U) {
                load(vb, sb_, k0 + U);
                reduce(va, sa, k0);
                load(va, sa, k0 + 2 * U);
                reduce(vb, sb_, k0 + U);
            }
        }
        // d_wd += (sum of the row's d_u) * x_i
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f4 xi = ok[t] ? ldv4(A.xs + (size_t)(row + A.dst_off) * A.Fs + (offb[t] >> 2)) : zero4();
            pk_fma4(yd[t], dad, xi);
        }
    }
    // fold the wave's 64/G row groups, then one partial row per wave
    if constexpr (G < 64) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            for (int o = G; o < 64; o <<= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ys[t][k] += __shfl_xor(ys[t][k], o, 64);
                    yd[t][k] += __shfl_xor(yd[t][k], o, 64);
                }
    }
    if (lane < G) {
        float *dst = A.part_y + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 * HF;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (ok[t]) {
                stv4(dst + roff[t], ys[t]);
                stv4(dst + HF + roff[t], yd[t]);
            }
    }
}

// ---------------------------------------------------------------------------------
// The same pass with the per-(edge, head) scalar chain hoisted out of the row loop.  In the
// kernel above every lane of a head group repeats  u -> LeakyReLU -> exp -> keep -> slope  for
// each edge (about 22 of the ~35 vector instructions an edge costs; the pass is VALU-issue
// bound).  Here lane (head h, sub-lane s) does that once for edge s, s + LH, ... of the chunk --
// one (edge, head) pair per lane, a_src gathered as one dword per lane -- and leaves
//   c1 = alpha~ * slope,   c0 = -alpha * S * slope        (d_u = c1 * <q, x_j> + c0)
// in a wave-private LDS table; the row loop then reads the pair of its edge (one ds_read_b64,
// broadcast within the head group) and is left with the dot, the butterfly and the FMAs.
// H2 <= 8 (table of 64 * H2 * 8 bytes per wave).
// ---------------------------------------------------------------------------------
constexpr int kBwd2MaxH2 = 8;
#ifndef GATMI_BWD2_U2
#define GATMI_BWD2_U2 2   // edges per batch (3: 128 VGPRs with spills in the loop, 9.9 ms)
#endif
#ifndef GATMI_BWD2_OCC
#define GATMI_BWD2_OCC 4   // waves per SIMD the register allocation is held to (114 VGPRs; at 5
                         // it spills 20 registers into the edge loop: 11.5 instead of 8.9 ms)
#endif

template <int LH_T, int G, int T>
__global__ __launch_bounds__(256, GATMI_BWD2_OCC) void inagg_bwd2_kernel(const InArgs A, int H2) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int U = (T == 1) ? 4 : (T == 2 ? GATMI_BWD2_U2 : 2);
    __shared__ f2 cs_all[4][64 * kBwd2MaxH2];
    INAGG_PROLOGUE
    f2 *cs = cs_all[threadIdx.x >> 6] + gbase * H2;      // this group's table: [edge][H2]
    (void)aoff;
    f4 ys[T], yd[T];
#pragma unroll
    for (int t = 0; t < T; ++t) ys[t] = yd[t] = zero4();
    INAGG_ITEM_LOOP
        (void)slot;
        f4 q4[T];
        f2 s2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < T; ++t) {
            q4[t] = ok[t] ? ldv4(A.q + (size_t)row * HF + roff[t]) : zero4();
            const f4 z4 = ok[t] ? ldv4(A.zin + (size_t)row * HF + roff[t]) : zero4();
            pk_dot4(s2, q4[t], z4);
        }
        const float S = head_sum<LH_T>(s2.x + s2.y, lh);
        const float lse_i = A.lsein[(size_t)row * A.H + head];
        float dad = 0.f;
        for (int c = beg; c < end; c += G) {
            const int n = min(G, end - c);
            const int cl = min(lig, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](f4(&v)[U][T], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = group_bcast<G>(cj, min(k0 + u, n - 1), gbase);
#pragma unroll
                    for (int t = 0; t < T; ++t) v[u][t] = ldv4_row<G>(A.xs, j, A.Fs, offb[t]);
                }
            };
            f4 va[U][T], vb[U][T];
            load(va, 0);
            // the chain, one (edge, head) pair per lane
            for (int e0 = 0; e0 < n; e0 += lh) {
                const int e = e0 + sl;
                const int ec = min(e, n - 1);
                const int js = __shfl(cj, gbase + ec, 64);
                const unsigned sb = (unsigned)__shfl(kb, gbase + ec, 64);
                const float as = A.xs[(size_t)js * A.Fs + A.Fp + head];
                const float uu = as + adi;
                const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                const float alpha = e < n ? exp_fast(uu * slope - lse_i) : 0.f;
                const float at = (sb & hbit) ? alpha * A.inv_keep : 0.f;
                if (hok) cs[e * H2 + head_raw] = f2{at * slope, -alpha * S * slope};
            }
            auto reduce = [&](const f4(&v)[U][T], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    f2 d = {0.f, 0.f};
#pragma unroll
                    for (int t = 0; t < T; ++t) pk_dot4(d, v[u][t], q4[t]);
                    const float d2 = head_sum<LH_T>(d.x + d.y, lh);
                    const f2 cc = cs[min(k0 + u, n - 1) * H2 + head];
                    const float du = k0 + u < n ? fmaf(d2, cc.x, cc.y) : 0.f;
                    dad += du;
#pragma unroll
                    for (int t = 0; t < T; ++t) pk_fma4(ys[t], du, v[u][t]);
                }
            };
            for (int k0 = 0; k0 < n; k0 += 2 * U) {
                load(vb, k0 + U);
                reduce(va, k0);
                load(va, k0 + 2 * U);
                reduce(vb, k0 + U);
            }
        }
        // d_wd += (sum of the row's d_u) * x_i
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const f4 xi = ok[t] ? ldv4(A.xs + (size_t)(row + A.dst_off) * A.Fs + (offb[t] >> 2)) : zero4();
            pk_fma4(yd[t], dad, xi);
        }
    }
    if constexpr (G < 64) {
#pragma unroll
        for (int t = 0; t < T; ++t)
            for (int o = G; o < 64; o <<= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ys[t][k] += __shfl_xor(ys[t][k], o, 64);
                    yd[t][k] += __shfl_xor(yd[t][k], o, 64);
                }
    }
    if (lane < G) {
        float *dst = A.part_y + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 * HF;
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (ok[t]) {
                stv4(dst + roff[t], ys[t]);
                stv4(dst + HF + roff[t], yd[t]);
            }
    }
}

// ---------------------------------------------------------------------------------
// The same pass in the "two edges per instruction" layout (H <= 4, Fp <= 128).  With head
// groups (above) every gathered row sits H2 times in registers, so a wave keeps four rows in
// flight and the kernel is bound by that (25 % less gather traffic buys 5 %).  Here the 32
// lanes of a wave HALF hold one x row once (one float4 each), the two halves work on two
// consecutive edges of the destination row, and every lane carries all heads:
//   * one load instruction = two rows; eight pairs in flight per wave (16 rows, 32 VGPRs);
//   * the H per-head dots of a lane are reduced by a multi-value butterfly: lanes trade half
//     of their values with lane^1, then lane^2 (after which a lane owns head 2(l&1) + ((l>>1)&1)
//     of its quad), then sum over the quads of the half;
//   * d_u = c1 <q, x_j> + c0 with (c1, c0) from the hoisted chain's LDS table as above, spread
//     to the other heads' lanes by four quad broadcasts, then the H FMAs per lane.
// ---------------------------------------------------------------------------------
constexpr int kBwd3U = 2;   // pairs per bank; two banks

__device__ __forceinline__ float quad_heads_sum(float v0, float v1, float v2, float v3, bool o1, bool o2) {
    float a = o1 ? v2 : v0, b = o1 ? v3 : v1;
    const float sa = o1 ? v0 : v2, sb = o1 ? v1 : v3;
    a += dpp_mov<0xB1>(sa);                     // quad_perm [1,0,3,2]
    b += dpp_mov<0xB1>(sb);
    float c = o2 ? b : a;
    const float sc = o2 ? a : b;
    c += dpp_mov<0x4E>(sc);                     // quad_perm [2,3,0,1]
    c += dpp_mov<0x124>(c);                     // row_ror:4
    c += dpp_mov<0x128>(c);                     // row_ror:8
    return c + __shfl_xor(c, 16, 64);           // the other 16-lane row of this half
}

// ---------------------------------------------------------------------------------
// Work items of the wave-per-item bf16 kernels, fetched ahead of their pass (products-shaped layer: forward
// 5.11 -> 4.77 ms, backward 6.33 -> 6.29; the f32 multi-edge kernels and inagg_fwd3 measured the same with
// and without and keep the plain loop).  An item's description is a chain
// of dependent loads -- item arrays -> row pointer -> col entries of its first chunk -- in front of the
// first gathered row.  At the top of pass k the wave asks for the item arrays of pass k + 3, the row
// pointer of pass k + 2 and its lane's col entry of the first chunk of pass k + 1; the wave index is a
// scalar, so everything but the col entry is a scalar load and waits in scalar registers.
// ---------------------------------------------------------------------------------
struct ItemsAhead {
    int r0, b0, s0, e0, c0;   // pass k: row, first edge, slot, row pointer of row + 1, col entry
    int r1, b1, s1, e1;       // pass k + 1
    int r2, b2, s2;           // pass k + 2
};
template <bool SLOT>
__device__ __forceinline__ void items_arrays(const InArgs &A, int it, int &r, int &b, int &s) {
    r = 0;
    b = 0;
    s = -1;
    if (it < A.n_items) {
        r = ld_layout(A.item_row, it);
        b = ld_layout(A.item_beg, it);
        if (SLOT) s = ld_layout(A.item_slot, it);
    }
}
__device__ __forceinline__ int items_end(const InArgs &A, int it, int r) {
    return it < A.n_items ? ld_layout(A.rowptr, r + 1) : 0;
}
__device__ __forceinline__ int items_col(const InArgs &A, int it, int b, int e, int lane) {
    const int n = min(64, min(e, b + A.seg_len) - b);
    return (it < A.n_items && n > 0) ? A.col[b + min(lane, n - 1)] : 0;
}
template <bool SLOT>
__device__ __forceinline__ void items_begin(const InArgs &A, int it, int stride, int lane, ItemsAhead &I) {
    items_arrays<SLOT>(A, it, I.r0, I.b0, I.s0);
    I.e0 = items_end(A, it, I.r0);
    I.c0 = items_col(A, it, I.b0, I.e0, lane);
    items_arrays<SLOT>(A, it + stride, I.r1, I.b1, I.s1);
    I.e1 = items_end(A, it + stride, I.r1);
    items_arrays<SLOT>(A, it + 2 * stride, I.r2, I.b2, I.s2);
}
// called at the top of pass `it`, once its own fields have been read
template <bool SLOT>
__device__ __forceinline__ void items_advance(const InArgs &A, int it, int stride, int lane, ItemsAhead &I) {
    I.r0 = I.r1; I.b0 = I.b1; I.s0 = I.s1; I.e0 = I.e1;
    I.c0 = items_col(A, it + stride, I.b0, I.e0, lane);
    I.r1 = I.r2; I.b1 = I.b2; I.s1 = I.s2;
    I.e1 = items_end(A, it + 2 * stride, I.r1);
    items_arrays<SLOT>(A, it + 3 * stride, I.r2, I.b2, I.s2);
}
#define GATMI_WAVE_ITEMS(SLOT)                                                            \
    int it = xcd_contiguous_block() * 4 + __builtin_amdgcn_readfirstlane(wv);             \
    ItemsAhead items;                                                                     \
    items_begin<SLOT>(A, it, stride, lane, items);                                        \
    for (; it < A.n_items; it += stride) {                                                \
        const int row = __builtin_amdgcn_readfirstlane(items.r0);                         \
        const int beg = __builtin_amdgcn_readfirstlane(items.b0);                         \
        const int slot = __builtin_amdgcn_readfirstlane(items.s0);                        \
        (void)slot;                                                                       \
        const int end = min(__builtin_amdgcn_readfirstlane(items.e0), beg + A.seg_len);   \
        const int first_col = items.c0;                                                   \
        items_advance<SLOT>(A, it, stride, lane, items);

__global__ __launch_bounds__(256, 4) void inagg_bwd3_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int U = kBwd3U;
    __shared__ f2 cs_all[4][64 * 4];
    const int lane = threadIdx.x & 63, half = lane >> 5, p = lane & 31, wv = threadIdx.x >> 6;
    const bool o1 = lane & 1, o2 = lane & 2;
    const int hl_raw = 2 * (int)o1 + (int)(o2 ? 1 : 0);          // the head this lane ends up owning
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(A.xs);
    f2 *cs = cs_all[wv];                                         // [edge of the chunk][4 heads]
    const unsigned hbit = 1u << hl;
    // d_wd accumulates once per destination row: it lives in LDS (this wave's 4 x 128 floats,
    // lane p of half 0 owns its float4s), not in 16 registers of the edge loop
    __shared__ __attribute__((aligned(16))) float yd_all[4][4 * 128];
    float *ydl = yd_all[wv];
    f4 ys[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        ys[h] = zero4();
        if (half == 0) stv4(ydl + h * 128 + p * 4, zero4());
    }
    const int stride = gridDim.x * 4;
    // This kernel keeps its own two-passes-ahead form (the row pointer of pass k + 1 and the item arrays of
    // pass k + 2 at the top, the next item's col entries at the END of a pass): measured against the
    // three-passes-ahead helper above on the arxiv-shaped layer, 0.349 against 0.360 ms
    int it = xcd_contiguous_block() * 4 + __builtin_amdgcn_readfirstlane(wv);
    int i1_row = 0, i1_beg = 0, i1_end = 0, i2_row = 0, i2_beg = 0;
    int i1_col = 0;                                               // col entry of this lane in the item's first chunk
    auto first_cols = [&]() {
        const int nn = min(64, min(i1_end, i1_beg + A.seg_len) - i1_beg);
        if (nn > 0) i1_col = A.col[i1_beg + min(lane, nn - 1)];
    };
    if (it < A.n_items) {
        i1_row = ld_layout(A.item_row, it);
        i1_beg = ld_layout(A.item_beg, it);
        i1_end = ld_layout(A.rowptr, i1_row + 1);
        first_cols();
    }
    if (it + stride < A.n_items) {
        i2_row = ld_layout(A.item_row, it + stride);
        i2_beg = ld_layout(A.item_beg, it + stride);
    }
    for (; it < A.n_items; it += stride) {
        const int row = __builtin_amdgcn_readfirstlane(i1_row);
        const int beg = __builtin_amdgcn_readfirstlane(i1_beg);
        int end = __builtin_amdgcn_readfirstlane(i1_end);
        end = min(end, beg + A.seg_len);
        const int first_col = i1_col;
        i1_row = i2_row;
        i1_beg = i2_beg;
        i1_end = it + stride < A.n_items ? ld_layout(A.rowptr, i1_row + 1) : 0;
        if (it + 2 * stride < A.n_items) {
            i2_row = ld_layout(A.item_row, it + 2 * stride);
            i2_beg = ld_layout(A.item_beg, it + 2 * stride);
        }
        f4 q4[4];
        float sv[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bool on = pok && h < A.H;
            q4[h] = on ? ldv4(A.q + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
            const f4 z4 = on ? ldv4(A.zin + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
            f2 d = {0.f, 0.f};
            pk_dot4(d, q4[h], z4);
            sv[h] = d.x + d.y;
        }
        const float S = quad_heads_sum(sv[0], sv[1], sv[2], sv[3], o1, o2);
        const float lse_i = A.lsein[(size_t)row * A.H + hl];
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float dad = 0.f;
        for (int c = beg; c < end; c += 64) {
            const int n = min(64, end - c);
            const int cl = min(lane, n - 1);
            const int cj = c == beg ? first_col : A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](f4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + 2 * u;
                    const int j0 = __builtin_amdgcn_readlane(cj, min(k, n - 1));
                    const int j1 = __builtin_amdgcn_readlane(cj, min(k + 1, n - 1));
                    const int j = half ? j1 : j0;
                    v[u] = *reinterpret_cast<const f4 *>(xsb + (size_t)j * row_bytes + offp);
                }
            };
            f4 va[U], vb[U];
            load(va, 0);
            // the chain, one (edge, head) pair per lane: edge = 16 * pass + lane / 4, head = hl
            for (int e0 = 0; e0 < n; e0 += 16) {
                const int e = e0 + (lane >> 2);
                const int ec = min(e, n - 1);
                const int js = __shfl(cj, ec, 64);
                const unsigned sb = (unsigned)__shfl(kb, ec, 64);
                const float as = A.xs[(size_t)js * A.Fs + A.Fp + hl];
                const float uu = as + adi;
                const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                const float alpha = (e < n && hlok) ? exp_fast(uu * slope - lse_i) : 0.f;
                const float at = (sb & hbit) ? alpha * A.inv_keep : 0.f;
                cs[e * 4 + hl_raw] = f2{at * slope, -alpha * S * slope};
            }
            auto reduce = [&](const f4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ke = k0 + 2 * u + half;             // this half's edge
                    float dv[4];
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        f2 d = {0.f, 0.f};
                        pk_dot4(d, v[u], q4[h]);
                        dv[h] = d.x + d.y;
                    }
                    const float d2 = quad_heads_sum(dv[0], dv[1], dv[2], dv[3], o1, o2);
                    const f2 cc = cs[min(ke, n - 1) * 4 + hl_raw];
                    const float du = ke < n ? fmaf(d2, cc.x, cc.y) : 0.f;
                    dad += du;
                    pk_fma4(ys[0], dpp_mov<0x00>(du), v[u]);      // quad lane 0 owns head 0
                    pk_fma4(ys[1], dpp_mov<0xAA>(du), v[u]);      // quad lane 2 owns head 1
                    pk_fma4(ys[2], dpp_mov<0x55>(du), v[u]);      // quad lane 1 owns head 2
                    pk_fma4(ys[3], dpp_mov<0xFF>(du), v[u]);      // quad lane 3 owns head 3
                }
            };
            for (int k0 = 0; k0 < n; k0 += 4 * U) {
                load(vb, k0 + 2 * U);
                reduce(va, k0);
                load(va, k0 + 4 * U);
                reduce(vb, k0 + 2 * U);
            }
        }
        // d_wd += (sum of the row's d_u) * x_i: both halves hold partial sums of d_u
        {
            if (it + stride < A.n_items) first_cols();           // the next item's row pointer has arrived
            const float dt = dad + __shfl_xor(dad, 32, 64);
            const f4 xi = pok ? ldv4(A.xs + (size_t)(row + A.dst_off) * A.Fs + p * 4) : zero4();
            const float d0 = dpp_mov<0x00>(dt), d1 = dpp_mov<0xAA>(dt), d2h = dpp_mov<0x55>(dt),
                        d3 = dpp_mov<0xFF>(dt);
            if (half == 0) {                                      // count the row once
                const float dh[4] = {d0, d1, d2h, d3};
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    f4 acc = ldv4(ydl + h * 128 + p * 4);
                    pk_fma4(acc, dh[h], xi);
                    stv4(ydl + h * 128 + p * 4, acc);
                }
            }
        }
    }
    // the two halves accumulated different edges: add them; then the block's four waves are
    // added in LDS and ONE partial row per block goes out (part_y [blocks][2][H*Fp])
    __shared__ __attribute__((aligned(16))) float ys_all[4][4 * 128];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ys[h][k] += __shfl_xor(ys[h][k], 32, 64);
        if (half == 0) stv4(ys_all[wv] + h * 128 + p * 4, ys[h]);
    }
    __syncthreads();
    float *dst = A.part_y + (size_t)blockIdx.x * 2 * HF;
    for (int idx = threadIdx.x; idx < 2 * 4 * 128; idx += 256) {
        const int which = idx >> 9, r = idx & 511, hh = r >> 7, ff = r & 127;
        if (hh < A.H && ff < A.Fp) {
            const float(*src)[4 * 128] = which ? yd_all : ys_all;
            dst[which * HF + hh * A.Fp + ff] = (src[0][r] + src[1][r]) + (src[2][r] + src[3][r]);
        }
    }
}

// ---------------------------------------------------------------------------------
// Forward in the two-edges-per-instruction layout (H <= 4, Fp <= 128).  No dots here, so
// once the softmax weights are hoisted the row loop is a load, one broadcast ds_read_b128 of
// the edge's four weights and the H FMAs per lane.  The chain runs lane-parallel per 64-edge
// chunk: lane (edge = lane / 4 + 16 pass, head = lane's quad head) computes the logit, the
// chunk maximum and sum are butterflies over the 16 edge lanes, the running (m, s, acc) are
// rescaled once per chunk (online softmax at chunk granularity).
// ---------------------------------------------------------------------------------
constexpr int kFwd3U = 4;

// all-reduce over the 16 lanes of this wave that share lane & 3 (op: 0 = max, 1 = sum)
template <int OP> __device__ __forceinline__ float edge_lanes_reduce(float v) {
    auto comb = [](float a, float b) { return OP == 0 ? fmaxf(a, b) : a + b; };
    v = comb(v, dpp_mov<0x124>(v));             // row_ror:4
    v = comb(v, dpp_mov<0x128>(v));             // row_ror:8
    v = comb(v, __shfl_xor(v, 16, 64));
    return comb(v, __shfl_xor(v, 32, 64));
}

__global__ __launch_bounds__(256, 4) void inagg_fwd3_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int U = kFwd3U;
    __shared__ __attribute__((aligned(16))) float ws_all[4][64 * 4];
    const int lane = threadIdx.x & 63, half = lane >> 5, p = lane & 31, wv = threadIdx.x >> 6;
    const int hl_raw = 2 * (lane & 1) + ((lane >> 1) & 1);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(A.xs);
    float *wt = ws_all[wv];                                      // [edge of the chunk][4 heads]
    const unsigned hbit = 1u << hl;
    const int stride = gridDim.x * 4;
    for (int it = xcd_contiguous_block() * 4 + wv; it < A.n_items; it += stride) {
        const int row = __builtin_amdgcn_readfirstlane(A.item_row[it]);
        const int beg = __builtin_amdgcn_readfirstlane(A.item_beg[it]);
        const int slot = __builtin_amdgcn_readfirstlane(A.item_slot[it]);
        int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
        end = min(end, beg + A.seg_len);
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float m = -INFINITY, ssum = 0.f;
        f4 acc[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) acc[h] = zero4();
        for (int c = beg; c < end; c += 64) {
            const int n = min(64, end - c);
            const int cl = min(lane, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](f4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + 2 * u;
                    const int j0 = __builtin_amdgcn_readlane(cj, min(k, n - 1));
                    const int j1 = __builtin_amdgcn_readlane(cj, min(k + 1, n - 1));
                    const int j = half ? j1 : j0;
                    v[u] = *reinterpret_cast<const f4 *>(xsb + (size_t)j * row_bytes + offp);
                }
            };
            f4 va[U], vb[U];
            load(va, 0);
            // logits of the chunk, one (edge, head) pair per lane and pass
            float zz[4];
            float zmax = -INFINITY;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int e = 16 * ps + (lane >> 2);
                zz[ps] = -INFINITY;
                if (16 * ps < n) {                                // wave-uniform
                    const int js = __shfl(cj, min(e, n - 1), 64);
                    const float uu = A.xs[(size_t)js * A.Fs + A.Fp + hl] + adi;
                    zz[ps] = e < n ? (uu > 0.f ? uu : A.neg_slope * uu) : -INFINITY;
                }
                zmax = fmaxf(zmax, zz[ps]);
            }
            const float mn = fmaxf(m, edge_lanes_reduce<0>(zmax));
            const float sc = exp_quick(m - mn);                   // first chunk: exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                if (16 * ps < n) {
                    const int e = 16 * ps + (lane >> 2);
                    const unsigned sb = (unsigned)__shfl(kb, min(e, n - 1), 64);
                    const float pe = exp_quick(zz[ps] - mn);      // 0 past the end
                    psum += pe;
                    wt[e * 4 + hl_raw] = (sb & hbit) ? pe * A.inv_keep : 0.f;
                }
            }
            ssum = fmaf(ssum, sc, edge_lanes_reduce<1>(psum));
            m = mn;
            {   // rescale the running sums: every lane needs the factor of every head
                const float s0 = dpp_mov<0x00>(sc), s1 = dpp_mov<0xAA>(sc), s2 = dpp_mov<0x55>(sc),
                            s3 = dpp_mov<0xFF>(sc);
                pk_scale4(acc[0], s0);
                pk_scale4(acc[1], s1);
                pk_scale4(acc[2], s2);
                pk_scale4(acc[3], s3);
            }
            auto reduce = [&](const f4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ke = k0 + 2 * u + half;
                    f4 w4 = *reinterpret_cast<const f4 *>(wt + min(ke, n - 1) * 4);
                    if (ke >= n) w4 = zero4();
                    pk_fma4(acc[0], w4[0], v[u]);
                    pk_fma4(acc[1], w4[1], v[u]);
                    pk_fma4(acc[2], w4[2], v[u]);
                    pk_fma4(acc[3], w4[3], v[u]);
                }
            };
            for (int k0 = 0; k0 < n; k0 += 4 * U) {
                load(vb, k0 + 2 * U);
                reduce(va, k0);
                load(va, k0 + 4 * U);
                reduce(vb, k0 + 2 * U);
            }
        }
        // the halves summed different edges
#pragma unroll
        for (int h = 0; h < 4; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[h][k] += __shfl_xor(acc[h][k], 32, 64);
        if (slot < 0) {
            const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
            const float i0 = dpp_mov<0x00>(inv), i1 = dpp_mov<0xAA>(inv), i2 = dpp_mov<0x55>(inv),
                        i3 = dpp_mov<0xFF>(inv);
            const float iv[4] = {i0, i1, i2, i3};
            if (half == 0 && pok) {
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    if (h < A.H) {
                        f4 o = acc[h];
                        pk_scale4(o, iv[h]);
                        stv4(A.z + (size_t)row * HF + h * A.Fp + p * 4, o);
                    }
            }
            if (A.lse && lane < 4 && hlok) A.lse[(size_t)row * A.H + hl] = m + logf(ssum);
        } else {
            if (half == 0 && pok) {
#pragma unroll
                for (int h = 0; h < 4; ++h)
                    if (h < A.H) stv4(A.part_acc + (size_t)slot * HF + h * A.Fp + p * 4, acc[h]);
            }
            if (lane < 4 && hlok) {
                A.part_m[(size_t)slot * A.H + hl] = m;
                A.part_s[(size_t)slot * A.H + hl] = ssum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// The same layout for up to 8 heads and any row of at most 32 float4: LR = 2^LRL lanes hold
// one packed row (one float4 each), a wave instruction covers EP = 64 / LR consecutive edges,
// all HP = 2^HPL heads sit in every lane.  run.py's 25 -> 8 x 30 layer (lph 7) runs 8 edges
// per instruction instead of one.
//   * per-head dots: HP partial sums per lane, HPL trading stages (a lane keeps half of its
//     values and receives the partner's sums of that half), then an all-reduce over the
//     remaining LR / HP lanes.  Lane l ends up owning head bitrev(l mod HP).
//   * a head's scalar goes back to all lanes of its HP-lane cluster by one DPP (HP 4) or one
//     ds_swizzle (HP 8) per head.
// ---------------------------------------------------------------------------------
template <int M> __device__ __forceinline__ float lane_xor(float v) {   // the value lane ^ M holds
    if constexpr (M == 1) return dpp_mov<0xB1>(v);                      // quad_perm [1,0,3,2]
    else if constexpr (M == 2) return dpp_mov<0x4E>(v);                 // quad_perm [2,3,0,1]
    else if constexpr (M == 4)                                          // swizzle, xor_mask 4
        return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x101F));
    else if constexpr (M == 8) return dpp_mov<0x128>(v);                // row_ror:8
    else return __shfl_xor(v, M, 64);
}

template <int S, int HP> __device__ __forceinline__ void trade_stage(float (&v)[HP], int lane) {
    constexpr int n2 = HP >> (S + 1);
    const bool up = (lane >> S) & 1;
#pragma unroll
    for (int i = 0; i < n2; ++i) {
        const float keep = up ? v[n2 + i] : v[i];
        const float send = up ? v[i] : v[n2 + i];
        v[i] = keep + lane_xor<(1 << S)>(send);
    }
}

// v[h] = this lane's part of head h's dot; returns the whole dot of the head this lane owns
template <int HPL, int LRL> __device__ __forceinline__ float heads_sum_e(float (&v)[1 << HPL], int lane) {
    constexpr int HP = 1 << HPL;
    if constexpr (HPL >= 1) trade_stage<0, HP>(v, lane);
    if constexpr (HPL >= 2) trade_stage<1, HP>(v, lane);
    if constexpr (HPL >= 3) trade_stage<2, HP>(v, lane);
    float c = v[0];
    if constexpr (HPL <= 2 && LRL > 2) c += lane_xor<4>(c);
    if constexpr (HPL <= 3 && LRL > 3) c += lane_xor<8>(c);
    if constexpr (LRL > 4) c += lane_xor<16>(c);
    return c;
}

template <int HPL> __device__ __forceinline__ int head_of_lane(int lane) {
    int h = 0;
#pragma unroll
    for (int s = 0; s < HPL; ++s) h |= ((lane >> s) & 1) << (HPL - 1 - s);
    return h;
}

// head h's value, from the lane of this lane's HP-cluster that owns it
template <int HPL> __device__ __forceinline__ float bcast_head(float v, int h) {
    if constexpr (HPL == 2) {
        switch (h) {
            case 0: return dpp_mov<0x00>(v);
            case 1: return dpp_mov<0xAA>(v);
            case 2: return dpp_mov<0x55>(v);
            default: return dpp_mov<0xFF>(v);
        }
    } else {   // swizzle: lane' = (lane & 0x18) | owner, owner = bitrev3(h)
        const int i = __float_as_int(v);
        switch (h) {
            case 0: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (0 << 5)));
            case 1: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (4 << 5)));
            case 2: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (2 << 5)));
            case 3: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (6 << 5)));
            case 4: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (1 << 5)));
            case 5: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (5 << 5)));
            case 6: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (3 << 5)));
            default: return __int_as_float(__builtin_amdgcn_ds_swizzle(i, 0x18 | (7 << 5)));
        }
    }
}

// all-reduce over the lanes that differ in lane bits LOW .. 5 (OP: 0 = max, 1 = sum)
template <int OP, int LOW> __device__ __forceinline__ float high_lanes_reduce(float v) {
    auto comb = [](float a, float b) { return OP == 0 ? fmaxf(a, b) : a + b; };
    if constexpr (LOW <= 2) v = comb(v, lane_xor<4>(v));
    if constexpr (LOW <= 3) v = comb(v, lane_xor<8>(v));
    if constexpr (LOW <= 4) v = comb(v, lane_xor<16>(v));
    if constexpr (LOW <= 5) v = comb(v, lane_xor<32>(v));
    return v;
}

// B16: the gathered rows come from the bf16 table (gatmi_inagg.xs16, laid out further down): a lane's
// piece is 8 bytes = the same 4 channels, widened to f32; the logits sit at byte 224 of a 256-byte row.
__device__ __forceinline__ f4 widen4(const uint2 r) {
    return f4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
              __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
}
template <bool B16> struct EdgeRow { typedef f4 raw; };
template <> struct EdgeRow<true> { typedef uint2 raw; };
__device__ __forceinline__ f4 edge_widen(const f4 v) { return v; }
__device__ __forceinline__ f4 edge_widen(const uint2 v) { return widen4(v); }

template <int LRL, int HPL, int U, bool B16 = false>
__global__ __launch_bounds__(256, HPL == 3 ? GATMI_E8_OCC : 4) void inagg_bwde_kernel(const InArgs A) {
    typedef typename EdgeRow<B16>::raw RAWV;
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int HP = 1 << HPL, LR = 1 << LRL, EP = 64 >> LRL, ROWF = HP * LR * 4, EPP = 64 >> HPL;
    __shared__ f2 cs_all[4][64 * HP];
    __shared__ __attribute__((aligned(16))) float yd_all[4][ROWF];
    __shared__ __attribute__((aligned(16))) float ys_all[4][ROWF];
    const int lane = threadIdx.x & 63, eg = lane >> LRL, p = lane & (LR - 1), wv = threadIdx.x >> 6;
    const int hl_raw = head_of_lane<HPL>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * (B16 ? 8u : 16u) : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = B16 ? (size_t)256 : (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(B16 ? A.xs16 : (const void *)A.xs);
    const size_t logit_off = B16 ? (size_t)224 : (size_t)A.Fp * 4;
    f2 *cs = cs_all[wv];                                         // [edge of the chunk][HP heads]
    const unsigned hbit = 1u << hl;
    float *ydl = yd_all[wv];
    f4 ys[HP];
#pragma unroll
    for (int h = 0; h < HP; ++h) {
        ys[h] = zero4();
        if (eg == 0) stv4(ydl + h * LR * 4 + p * 4, zero4());
    }
    const int stride = gridDim.x * 4;
    for (int it = xcd_contiguous_block() * 4 + wv; it < A.n_items; it += stride) {
        const int row = __builtin_amdgcn_readfirstlane(A.item_row[it]);
        const int beg = __builtin_amdgcn_readfirstlane(A.item_beg[it]);
        int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
        end = min(end, beg + A.seg_len);
        f4 q4[HP];
        float sv[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            const bool on = pok && h < A.H;
            q4[h] = on ? ldv4(A.q + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
            const f4 z4 = on ? ldv4(A.zin + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
            f2 d = {0.f, 0.f};
            pk_dot4(d, q4[h], z4);
            sv[h] = d.x + d.y;
        }
        const float S = heads_sum_e<HPL, LRL>(sv, lane);
        const float lse_i = A.lsein[(size_t)row * A.H + hl];
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float dad = 0.f;
        for (int c = beg; c < end; c += 64) {
            const int n = min(64, end - c);
            const int cl = min(lane, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](RAWV(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = __shfl(cj, min(k0 + EP * u + eg, n - 1), 64);
                    v[u] = *reinterpret_cast<const RAWV *>(xsb + (size_t)j * row_bytes + offp);
                }
            };
            RAWV va[U], vb[U];
            load(va, 0);
            // the chain, one (edge, head) pair per lane: edge = EPP * pass + lane / HP, head = hl
            for (int e0 = 0; e0 < n; e0 += EPP) {
                const int e = e0 + (lane >> HPL);
                const int ec = min(e, n - 1);
                const int js = __shfl(cj, ec, 64);
                const unsigned sb = (unsigned)__shfl(kb, ec, 64);
                const float as = *reinterpret_cast<const float *>(xsb + (size_t)js * row_bytes + logit_off + 4 * hl);
                const float uu = as + adi;
                const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                const float alpha = (e < n && hlok) ? exp_fast(uu * slope - lse_i) : 0.f;
                const float at = (sb & hbit) ? alpha * A.inv_keep : 0.f;
                cs[e * HP + hl_raw] = f2{at * slope, -alpha * S * slope};
            }
            auto reduce = [&](const RAWV(&raw)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ke = k0 + EP * u + eg;              // this lane group's edge
                    const f4 vu = edge_widen(raw[u]);
                    float dv[HP];
#pragma unroll
                    for (int h = 0; h < HP; ++h) {
                        f2 d = {0.f, 0.f};
                        pk_dot4(d, vu, q4[h]);
                        dv[h] = d.x + d.y;
                    }
                    const float d2 = heads_sum_e<HPL, LRL>(dv, lane);
                    const f2 cc = cs[min(ke, n - 1) * HP + hl_raw];
                    const float du = ke < n ? fmaf(d2, cc.x, cc.y) : 0.f;
                    dad += du;
#pragma unroll
                    for (int h = 0; h < HP; ++h) pk_fma4(ys[h], bcast_head<HPL>(du, h), vu);
                }
            };
            for (int k0 = 0; k0 < n; k0 += 2 * U * EP) {
                load(vb, k0 + U * EP);
                reduce(va, k0);
                load(va, k0 + 2 * U * EP);
                reduce(vb, k0 + U * EP);
            }
        }
        // d_wd += (sum of the row's d_u) * x_i: every lane group holds partial sums of d_u
        {
            const float dt = high_lanes_reduce<1, LRL>(dad);
            const f4 xi = pok ? ldv4(A.xs + (size_t)(row + A.dst_off) * A.Fs + p * 4) : zero4();
            float dh[HP];
#pragma unroll
            for (int h = 0; h < HP; ++h) dh[h] = bcast_head<HPL>(dt, h);
            if (eg == 0) {                                        // count the row once
#pragma unroll
                for (int h = 0; h < HP; ++h) {
                    f4 acc = ldv4(ydl + h * LR * 4 + p * 4);
                    pk_fma4(acc, dh[h], xi);
                    stv4(ydl + h * LR * 4 + p * 4, acc);
                }
            }
        }
    }
    // the lane groups accumulated different edges: add them; then the block's four waves are
    // added in LDS and ONE partial row per block goes out (part_y [blocks][2][H*Fp])
#pragma unroll
    for (int h = 0; h < HP; ++h) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ys[h][k] = high_lanes_reduce<1, LRL>(ys[h][k]);
        if (eg == 0) stv4(ys_all[wv] + h * LR * 4 + p * 4, ys[h]);
    }
    __syncthreads();
    float *dst = A.part_y + (size_t)blockIdx.x * 2 * HF;
    for (int idx = threadIdx.x; idx < 2 * ROWF; idx += 256) {
        const int which = idx / ROWF, r = idx % ROWF, hh = r / (LR * 4), ff = r % (LR * 4);
        if (hh < A.H && ff < A.Fp) {
            const float(*src)[ROWF] = which ? yd_all : ys_all;
            dst[which * HF + hh * A.Fp + ff] = (src[0][r] + src[1][r]) + (src[2][r] + src[3][r]);
        }
    }
}

template <int LRL, int HPL, int U, bool B16 = false>
__global__ __launch_bounds__(256, 4) void inagg_fwde_kernel(const InArgs A) {
    typedef typename EdgeRow<B16>::raw RAWV;
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int HP = 1 << HPL, LR = 1 << LRL, EP = 64 >> LRL, EPP = 64 >> HPL;
    __shared__ __attribute__((aligned(16))) float ws_all[4][64 * HP];
    const int lane = threadIdx.x & 63, eg = lane >> LRL, p = lane & (LR - 1), wv = threadIdx.x >> 6;
    const int hl_raw = head_of_lane<HPL>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * (B16 ? 8u : 16u) : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = B16 ? (size_t)256 : (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(B16 ? A.xs16 : (const void *)A.xs);
    const size_t logit_off = B16 ? (size_t)224 : (size_t)A.Fp * 4;
    float *wt = ws_all[wv];                                      // [edge of the chunk][HP heads]
    const unsigned hbit = 1u << hl;
    const int stride = gridDim.x * 4;
    for (int it = xcd_contiguous_block() * 4 + wv; it < A.n_items; it += stride) {
        const int row = __builtin_amdgcn_readfirstlane(A.item_row[it]);
        const int beg = __builtin_amdgcn_readfirstlane(A.item_beg[it]);
        const int slot = __builtin_amdgcn_readfirstlane(A.item_slot[it]);
        int end = __builtin_amdgcn_readfirstlane(A.rowptr[row + 1]);
        end = min(end, beg + A.seg_len);
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float m = -INFINITY, ssum = 0.f;
        f4 acc[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) acc[h] = zero4();
        for (int c = beg; c < end; c += 64) {
            const int n = min(64, end - c);
            const int cl = min(lane, n - 1);
            const int cj = A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](RAWV(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = __shfl(cj, min(k0 + EP * u + eg, n - 1), 64);
                    v[u] = *reinterpret_cast<const RAWV *>(xsb + (size_t)j * row_bytes + offp);
                }
            };
            RAWV va[U], vb[U];
            load(va, 0);
            // logits of the chunk, one (edge, head) pair per lane and pass
            float zz[HP];
            float zmax = -INFINITY;
#pragma unroll
            for (int ps = 0; ps < HP; ++ps) {
                const int e = EPP * ps + (lane >> HPL);
                zz[ps] = -INFINITY;
                if (EPP * ps < n) {                               // wave-uniform
                    const int js = __shfl(cj, min(e, n - 1), 64);
                    const float uu = *reinterpret_cast<const float *>(xsb + (size_t)js * row_bytes + logit_off +
                                                                      4 * hl) + adi;
                    zz[ps] = e < n ? (uu > 0.f ? uu : A.neg_slope * uu) : -INFINITY;
                }
                zmax = fmaxf(zmax, zz[ps]);
            }
            const float mn = fmaxf(m, high_lanes_reduce<0, HPL>(zmax));
            const float sc = exp_quick(m - mn);                   // first chunk: exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int ps = 0; ps < HP; ++ps) {
                if (EPP * ps < n) {
                    const int e = EPP * ps + (lane >> HPL);
                    const unsigned sb = (unsigned)__shfl(kb, min(e, n - 1), 64);
                    const float pe = exp_quick(zz[ps] - mn);      // 0 past the end
                    psum += pe;
                    wt[e * HP + hl_raw] = (sb & hbit) ? pe * A.inv_keep : 0.f;
                }
            }
            ssum = fmaf(ssum, sc, high_lanes_reduce<1, HPL>(psum));
            m = mn;
#pragma unroll
            for (int h = 0; h < HP; ++h) pk_scale4(acc[h], bcast_head<HPL>(sc, h));
            auto reduce = [&](const RAWV(&raw)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ke = k0 + EP * u + eg;
                    const f4 vu = edge_widen(raw[u]);
                    const float *wp = wt + min(ke, n - 1) * HP;
#pragma unroll
                    for (int hq = 0; hq < HP / 4; ++hq) {
                        f4 w4 = *reinterpret_cast<const f4 *>(wp + 4 * hq);
                        if (ke >= n) w4 = zero4();
                        pk_fma4(acc[4 * hq + 0], w4[0], vu);
                        pk_fma4(acc[4 * hq + 1], w4[1], vu);
                        pk_fma4(acc[4 * hq + 2], w4[2], vu);
                        pk_fma4(acc[4 * hq + 3], w4[3], vu);
                    }
                }
            };
            for (int k0 = 0; k0 < n; k0 += 2 * U * EP) {
                load(vb, k0 + U * EP);
                reduce(va, k0);
                load(va, k0 + 2 * U * EP);
                reduce(vb, k0 + U * EP);
            }
        }
        // the lane groups summed different edges
#pragma unroll
        for (int h = 0; h < HP; ++h)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[h][k] = high_lanes_reduce<1, LRL>(acc[h][k]);
        if (slot < 0) {
            const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
            float iv[HP];
#pragma unroll
            for (int h = 0; h < HP; ++h) iv[h] = bcast_head<HPL>(inv, h);
            if (eg == 0 && pok) {
#pragma unroll
                for (int h = 0; h < HP; ++h)
                    if (h < A.H) {
                        f4 o = acc[h];
                        pk_scale4(o, iv[h]);
                        stv4(A.z + (size_t)row * HF + h * A.Fp + p * 4, o);
                    }
            }
            if (A.lse && lane < HP && hlok) A.lse[(size_t)row * A.H + hl] = m + logf(ssum);
        } else {
            if (eg == 0 && pok) {
#pragma unroll
                for (int h = 0; h < HP; ++h)
                    if (h < A.H) stv4(A.part_acc + (size_t)slot * HF + h * A.Fp + p * 4, acc[h]);
            }
            if (lane < HP && hlok) {
                A.part_m[(size_t)slot * A.H + hl] = m;
                A.part_s[(size_t)slot * A.H + hl] = ssum;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// bf16 source rows (gatmi_inagg.xs16; H <= 4, Fp <= 112).  The packed f32 row of the products-
// shaped layer is 416 bytes = four 128-byte lines per edge, and lines per edge is what these
// kernels pay for.  The bf16 table keeps a node in ONE 256-byte row, two lines:
//   piece p = 0..13 (16 bytes): channels 8p .. 8p+7 of x as bf16 (zero beyond F)
//   piece 14: a_src[0..3] in f32 (logits are NOT rounded: they come from the f32 x)   piece 15: zero
// Same layout family as the multi-edge kernels above with LR = 16 lanes per row: one wave
// instruction covers FOUR consecutive edges, a lane holds 8 channels of its edge's row and all
// four heads (two float4 accumulators per head).  Arithmetic is f32 on the widened values; Z, q,
// the logits, the softmax and every output stay f32.
// ---------------------------------------------------------------------------------
constexpr int kRow16 = 256;              // bytes per row of the bf16 table
constexpr int kRow16Logits = 224;        // byte offset of a_src[0] in a row

__device__ __forceinline__ void widen8(const uint4 r, f4 &lo, f4 &hi) {
    lo = f4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u),
            __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
    hi = f4{__uint_as_float(r.z << 16), __uint_as_float(r.z & 0xffff0000u),
            __uint_as_float(r.w << 16), __uint_as_float(r.w & 0xffff0000u)};
}

// xs [n, Fs] (f32) -> xs16 [n, 256 bytes]; one thread per 16-byte piece
__global__ __launch_bounds__(256) void inagg_table16_kernel(const float *xs, long long n, int Fp, int Fs,
                                                            int H, uint4 *out) {
    const long long total = n * 16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i >> 4;
        const int p = (int)(i & 15);
        const float *r = xs + row * Fs;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if (p < 14) {                 // Fp and Fs are multiples of 4: whole float4 are x or are not
            const f4 lo = 8 * p < Fp ? ldv4(r + 8 * p) : zero4();
            const f4 hi = 8 * p + 4 < Fp ? ldv4(r + 8 * p + 4) : zero4();
            o = make_uint4(pack_bf16(lo[0], lo[1]), pack_bf16(lo[2], lo[3]), pack_bf16(hi[0], hi[1]),
                           pack_bf16(hi[2], hi[3]));
        } else {                      // pieces 14, 15: logits of heads 0..3, 4..7
            const int h0 = 4 * (p - 14);
            float a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = h0 + k < H ? r[Fp + h0 + k] : 0.f;
            o = make_uint4(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]),
                           __float_as_uint(a[3]));
        }
        out[i] = o;
    }
}

template <int U>
__global__ __launch_bounds__(256, 4) void inagg_fwd16_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int HPL = 2, LRL = 4, HP = 4, EP = 4, EPP = 16;
    __shared__ __attribute__((aligned(16))) float ws_all[4][64 * HP];
    const int lane = threadIdx.x & 63, eg = lane >> LRL, p = lane & 15, wv = threadIdx.x >> 6;
    const int hl_raw = head_of_lane<HPL>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool ok0 = 8 * p < A.Fp, ok1 = 8 * p + 4 < A.Fp;
    const unsigned offp = ok0 ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const char *tb = reinterpret_cast<const char *>(A.xs16);
    float *wt = ws_all[wv];                                      // [edge of the chunk][HP heads]
    const unsigned hbit = 1u << hl;
    const int stride = gridDim.x * 4;
    GATMI_WAVE_ITEMS(true)
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float m = -INFINITY, ssum = 0.f;
        f4 acc[HP][2];
#pragma unroll
        for (int h = 0; h < HP; ++h) acc[h][0] = acc[h][1] = zero4();
        for (int c = beg; c < end; c += 64) {
            const int n = min(64, end - c);
            const int cl = min(lane, n - 1);
            const int cj = c == beg ? first_col : A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](uint4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = __shfl(cj, min(k0 + EP * u + eg, n - 1), 64);
                    v[u] = *reinterpret_cast<const uint4 *>(tb + (size_t)j * kRow16 + offp);
                }
            };
            uint4 va[U], vb[U];
            load(va, 0);
            // logits of the chunk, one (edge, head) pair per lane and pass
            float zz[HP];
            float zmax = -INFINITY;
#pragma unroll
            for (int ps = 0; ps < HP; ++ps) {
                const int e = EPP * ps + (lane >> HPL);
                zz[ps] = -INFINITY;
                if (EPP * ps < n) {                               // wave-uniform
                    const int js = __shfl(cj, min(e, n - 1), 64);
                    const float uu = *reinterpret_cast<const float *>(tb + (size_t)js * kRow16 +
                                                                      kRow16Logits + 4 * hl) + adi;
                    zz[ps] = e < n ? (uu > 0.f ? uu : A.neg_slope * uu) : -INFINITY;
                }
                zmax = fmaxf(zmax, zz[ps]);
            }
            const float mn = fmaxf(m, high_lanes_reduce<0, HPL>(zmax));
            const float sc = exp_quick(m - mn);                   // first chunk: exp(-inf) = 0
            float psum = 0.f;
#pragma unroll
            for (int ps = 0; ps < HP; ++ps) {
                if (EPP * ps < n) {
                    const int e = EPP * ps + (lane >> HPL);
                    const unsigned sb = (unsigned)__shfl(kb, min(e, n - 1), 64);
                    const float pe = exp_quick(zz[ps] - mn);      // 0 past the end
                    psum += pe;
                    wt[e * HP + hl_raw] = (sb & hbit) ? pe * A.inv_keep : 0.f;
                }
            }
            ssum = fmaf(ssum, sc, high_lanes_reduce<1, HPL>(psum));
            m = mn;
#pragma unroll
            for (int h = 0; h < HP; ++h) {
                const float sh = bcast_head<HPL>(sc, h);
                pk_scale4(acc[h][0], sh);
                pk_scale4(acc[h][1], sh);
            }
            auto reduce = [&](const uint4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ke = k0 + EP * u + eg;
                    f4 w4 = *reinterpret_cast<const f4 *>(wt + min(ke, n - 1) * HP);
                    if (ke >= n) w4 = zero4();
                    f4 x0, x1;
                    widen8(v[u], x0, x1);
#pragma unroll
                    for (int h = 0; h < HP; ++h) {
                        pk_fma4(acc[h][0], w4[h], x0);
                        pk_fma4(acc[h][1], w4[h], x1);
                    }
                }
            };
            for (int k0 = 0; k0 < n; k0 += 2 * U * EP) {
                load(vb, k0 + U * EP);
                reduce(va, k0);
                load(va, k0 + 2 * U * EP);
                reduce(vb, k0 + U * EP);
            }
        }
        // the four lane groups summed different edges
#pragma unroll
        for (int h = 0; h < HP; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[h][q][k] = high_lanes_reduce<1, LRL>(acc[h][q][k]);
        if (slot < 0) {
            const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
            float iv[HP];
#pragma unroll
            for (int h = 0; h < HP; ++h) iv[h] = bcast_head<HPL>(inv, h);
            if (eg == 0) {
#pragma unroll
                for (int h = 0; h < HP; ++h)
                    if (h < A.H) {
                        f4 o0 = acc[h][0], o1 = acc[h][1];
                        pk_scale4(o0, iv[h]);
                        pk_scale4(o1, iv[h]);
                        float *zr = A.z + (size_t)row * HF + h * A.Fp + p * 8;
                        if (ok0) stv4(zr, o0);
                        if (ok1) stv4(zr + 4, o1);
                    }
            }
            if (A.lse && lane < HP && hlok) A.lse[(size_t)row * A.H + hl] = m + logf(ssum);
        } else {
            if (eg == 0) {
#pragma unroll
                for (int h = 0; h < HP; ++h)
                    if (h < A.H) {
                        float *zr = A.part_acc + (size_t)slot * HF + h * A.Fp + p * 8;
                        if (ok0) stv4(zr, acc[h][0]);
                        if (ok1) stv4(zr + 4, acc[h][1]);
                    }
            }
            if (lane < HP && hlok) {
                A.part_m[(size_t)slot * A.H + hl] = m;
                A.part_s[(size_t)slot * A.H + hl] = ssum;
            }
        }
    }
}

// blocks per CU / instructions (of four edges) per bank.  Two accumulators and two q pieces per
// head take 64 registers: at 4 blocks per CU one instruction per bank fits (6.4 ms at products
// size); two need 163 registers and 3 blocks per CU (7.2 ms), 2 blocks x 3 instructions 9.0 ms.
#ifndef GATMI_B16_OCC
#define GATMI_B16_OCC 4
#endif
#ifndef GATMI_B16_U
#define GATMI_B16_U 1
#endif
template <int U>
__global__ __launch_bounds__(256, GATMI_B16_OCC) void inagg_bwd16_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int HPL = 2, LRL = 4, HP = 4, EP = 4, EPP = 16, ROWF = HP * 128;
    __shared__ f2 cs_all[4][64 * HP];
    __shared__ __attribute__((aligned(16))) float yd_all[4][ROWF];
    __shared__ __attribute__((aligned(16))) float ys_all[4][ROWF];
    const int lane = threadIdx.x & 63, eg = lane >> LRL, p = lane & 15, wv = threadIdx.x >> 6;
    const int hl_raw = head_of_lane<HPL>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool ok0 = 8 * p < A.Fp, ok1 = 8 * p + 4 < A.Fp;
    const unsigned offp = ok0 ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const char *tb = reinterpret_cast<const char *>(A.xs16);
    f2 *cs = cs_all[wv];                                         // [edge of the chunk][HP heads]
    const unsigned hbit = 1u << hl;
    float *ydl = yd_all[wv];
    f4 ys[HP][2];
#pragma unroll
    for (int h = 0; h < HP; ++h) {
        ys[h][0] = ys[h][1] = zero4();
        if (eg == 0) {
            stv4(ydl + h * 128 + p * 8, zero4());
            stv4(ydl + h * 128 + p * 8 + 4, zero4());
        }
    }
    const int stride = gridDim.x * 4;
    GATMI_WAVE_ITEMS(false)
        f4 q8[HP][2];
        float sv[HP];
#pragma unroll
        for (int h = 0; h < HP; ++h) {
            const size_t o = (size_t)row * HF + h * A.Fp + p * 8;
            const bool on0 = ok0 && h < A.H, on1 = ok1 && h < A.H;
            q8[h][0] = on0 ? ldv4(A.q + o) : zero4();
            q8[h][1] = on1 ? ldv4(A.q + o + 4) : zero4();
            const f4 z0 = on0 ? ldv4(A.zin + o) : zero4();
            const f4 z1 = on1 ? ldv4(A.zin + o + 4) : zero4();
            f2 d = {0.f, 0.f};
            pk_dot4(d, q8[h][0], z0);
            pk_dot4(d, q8[h][1], z1);
            sv[h] = d.x + d.y;
        }
        const float S = heads_sum_e<HPL, LRL>(sv, lane);
        const float lse_i = A.lsein[(size_t)row * A.H + hl];
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float dad = 0.f;
        for (int c = beg; c < end; c += 64) {
            const int n = min(64, end - c);
            const int cl = min(lane, n - 1);
            const int cj = c == beg ? first_col : A.col[c + cl];
            const int kb = (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)c + cl, A.H);
            auto load = [&](uint4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int j = __shfl(cj, min(k0 + EP * u + eg, n - 1), 64);
                    v[u] = *reinterpret_cast<const uint4 *>(tb + (size_t)j * kRow16 + offp);
                }
            };
            uint4 va[U], vb[U];
            load(va, 0);
            // the chain, one (edge, head) pair per lane: edge = EPP * pass + lane / HP, head = hl
            for (int e0 = 0; e0 < n; e0 += EPP) {
                const int e = e0 + (lane >> HPL);
                const int ec = min(e, n - 1);
                const int js = __shfl(cj, ec, 64);
                const unsigned sb = (unsigned)__shfl(kb, ec, 64);
                const float as = *reinterpret_cast<const float *>(tb + (size_t)js * kRow16 +
                                                                  kRow16Logits + 4 * hl);
                const float uu = as + adi;
                const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                const float alpha = (e < n && hlok) ? exp_fast(uu * slope - lse_i) : 0.f;
                const float at = (sb & hbit) ? alpha * A.inv_keep : 0.f;
                cs[e * HP + hl_raw] = f2{at * slope, -alpha * S * slope};
            }
            auto reduce = [&](const uint4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ke = k0 + EP * u + eg;              // this lane group's edge
                    f4 x0, x1;
                    widen8(v[u], x0, x1);
                    float dv[HP];
#pragma unroll
                    for (int h = 0; h < HP; ++h) {
                        f2 d = {0.f, 0.f};
                        pk_dot4(d, x0, q8[h][0]);
                        pk_dot4(d, x1, q8[h][1]);
                        dv[h] = d.x + d.y;
                    }
                    const float d2 = heads_sum_e<HPL, LRL>(dv, lane);
                    const f2 cc = cs[min(ke, n - 1) * HP + hl_raw];
                    const float du = ke < n ? fmaf(d2, cc.x, cc.y) : 0.f;
                    dad += du;
#pragma unroll
                    for (int h = 0; h < HP; ++h) {
                        const float dh = bcast_head<HPL>(du, h);
                        pk_fma4(ys[h][0], dh, x0);
                        pk_fma4(ys[h][1], dh, x1);
                    }
                }
            };
            for (int k0 = 0; k0 < n; k0 += 2 * U * EP) {
                load(vb, k0 + U * EP);
                reduce(va, k0);
                load(va, k0 + 2 * U * EP);
                reduce(vb, k0 + U * EP);
            }
        }
        // d_wd += (sum of the row's d_u) * x_i (the f32 row: this term is not gathered)
        {
            const float dt = high_lanes_reduce<1, LRL>(dad);
            const float *xr = A.xs + (size_t)(row + A.dst_off) * A.Fs + p * 8;
            const f4 xi0 = ok0 ? ldv4(xr) : zero4();
            const f4 xi1 = ok1 ? ldv4(xr + 4) : zero4();
            float dh[HP];
#pragma unroll
            for (int h = 0; h < HP; ++h) dh[h] = bcast_head<HPL>(dt, h);
            if (eg == 0) {                                        // count the row once
#pragma unroll
                for (int h = 0; h < HP; ++h) {
                    f4 a0 = ldv4(ydl + h * 128 + p * 8), a1 = ldv4(ydl + h * 128 + p * 8 + 4);
                    pk_fma4(a0, dh[h], xi0);
                    pk_fma4(a1, dh[h], xi1);
                    stv4(ydl + h * 128 + p * 8, a0);
                    stv4(ydl + h * 128 + p * 8 + 4, a1);
                }
            }
        }
    }
    // the lane groups accumulated different edges: add them; then the block's four waves are
    // added in LDS and ONE partial row per block goes out (part_y [blocks][2][H*Fp])
#pragma unroll
    for (int h = 0; h < HP; ++h) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) ys[h][q][k] = high_lanes_reduce<1, LRL>(ys[h][q][k]);
            if (eg == 0) stv4(ys_all[wv] + h * 128 + p * 8 + 4 * q, ys[h][q]);
        }
    }
    __syncthreads();
    float *dst = A.part_y + (size_t)blockIdx.x * 2 * HF;
    for (int idx = threadIdx.x; idx < 2 * ROWF; idx += 256) {
        const int which = idx / ROWF, r = idx % ROWF, hh = r >> 7, ff = r & 127;
        if (hh < A.H && ff < A.Fp) {
            const float(*src)[ROWF] = which ? yd_all : ys_all;
            dst[which * HF + hh * A.Fp + ff] = (src[0][r] + src[1][r]) + (src[2][r] + src[3][r]);
        }
    }
}

// ---------------------------------------------------------------------------------
// A destination row per 8-lane group (5..8 heads, rows of at most 8 float4, LOW degree: run.py's
// 25 -> 8x30 layer on a 5-nearest-neighbour graph).  With one row per wave an item there is a
// handful of edges under a whole wave's prologue.  Here a wave takes 8 items at once: the 8
// lanes of group g hold one packed row (a float4 each) and all 8 heads, lane p of a group owns
// head bitrev(p) for the scalar chain, 8 edges of the group's row make a chunk (lane p loads
// col and keep bits of edge p).  Nothing crosses between groups until the partial sums of the
// backward are added at the end; the wave loops to its longest row.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int groups_max_i(int v) {
    v = max(v, __shfl_xor(v, 8, 64));
    v = max(v, __shfl_xor(v, 16, 64));
    return max(v, __shfl_xor(v, 32, 64));
}

#ifndef GATMI_FWDG_OCC
#define GATMI_FWDG_OCC 3
#endif
#ifndef GATMI_BWDG_OCC
#define GATMI_BWDG_OCC 3
#endif
__global__ __launch_bounds__(256, GATMI_FWDG_OCC) void inagg_fwdg_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    const int lane = threadIdx.x & 63, g = lane >> 3, p = lane & 7, wv = threadIdx.x >> 6;
    const int hl_raw = head_of_lane<3>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(A.xs);
    const unsigned hbit = 1u << hl;
    const int stride = gridDim.x * 32;
    for (int it0 = (xcd_contiguous_block() * 4 + wv) * 8; it0 < A.n_items; it0 += stride) {
        const int it = it0 + g;
        const bool on = it < A.n_items;
        int row = 0, beg = 0, end = 0, slot = -1;
        if (on) {
            row = A.item_row[it];
            beg = A.item_beg[it];
            slot = A.item_slot[it];
            end = min(A.rowptr[row + 1], beg + A.seg_len);
        }
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float m = -INFINITY, ssum = 0.f;
        f4 acc[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) acc[h] = zero4();
        const int nmax = __builtin_amdgcn_readfirstlane(groups_max_i(end - beg));
        for (int c = 0; c < nmax; c += 8) {
            const int n = min(max(end - beg - c, 0), 8);          // this group's edges of the chunk
            const int nw = min(nmax - c, 8);                      // the wave's longest
            const int el = beg + c + min(p, max(n - 1, 0));
            const int cj = n > 0 ? A.col[el] : 0;
            const int kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
            f4 v[8];
            float z[8];
            unsigned sb[8];
            float zmax = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                z[e] = -INFINITY;
                if (e < nw) {                                     // wave-uniform
                    const int src = (lane & 56) | min(e, max(n - 1, 0));
                    const int j = __shfl(cj, src, 64);
                    sb[e] = (unsigned)__shfl(kb, src, 64);
                    v[e] = *reinterpret_cast<const f4 *>(xsb + (size_t)j * row_bytes + offp);
                    const float uu = A.xs[(size_t)j * A.Fs + A.Fp + hl] + adi;
                    z[e] = e < n ? (uu > 0.f ? uu : A.neg_slope * uu) : -INFINITY;
                }
                zmax = fmaxf(zmax, z[e]);
            }
            const float mn = fmaxf(m, zmax);
            const float sc = mn == -INFINITY ? 1.0f : exp_quick(m - mn);
            float psum = 0.f, w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                w[e] = 0.f;
                if (e < nw) {
                    const float pe = e < n ? exp_quick(z[e] - mn) : 0.f;
                    psum += pe;
                    w[e] = (sb[e] & hbit) ? pe * A.inv_keep : 0.f;
                }
            }
            ssum = fmaf(ssum, sc, psum);
            m = mn;
            if (c > 0) {
#pragma unroll
                for (int h = 0; h < 8; ++h) pk_scale4(acc[h], bcast_head<3>(sc, h));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < nw) {
#pragma unroll
                    for (int h = 0; h < 8; ++h) pk_fma4(acc[h], bcast_head<3>(w[e], h), v[e]);
                }
        }
        const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
        float iv[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) iv[h] = bcast_head<3>(inv, h);
        if (on && slot < 0) {
            if (pok) {
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    if (h < A.H) {
                        f4 o = acc[h];
                        pk_scale4(o, iv[h]);
                        stv4(A.z + (size_t)row * HF + h * A.Fp + p * 4, o);
                    }
            }
            if (A.lse && hlok) A.lse[(size_t)row * A.H + hl] = m + logf(ssum);
        } else if (on) {
            if (pok) {
#pragma unroll
                for (int h = 0; h < 8; ++h)
                    if (h < A.H) stv4(A.part_acc + (size_t)slot * HF + h * A.Fp + p * 4, acc[h]);
            }
            if (hlok) {
                A.part_m[(size_t)slot * A.H + hl] = m;
                A.part_s[(size_t)slot * A.H + hl] = ssum;
            }
        }
    }
}

#ifndef GATMI_BWDG_E
#define GATMI_BWDG_E 2
#endif
constexpr int kBwdgE = GATMI_BWDG_E;
__global__ __launch_bounds__(256, GATMI_BWDG_OCC) void inagg_bwdg_kernel(const InArgs A) {
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    constexpr int ROWF = 8 * 8 * 4;                               // [8 heads][8 lanes x float4]
    __shared__ f4 yd_all[4][8 * 64];                              // [wave][head][lane]
    __shared__ __attribute__((aligned(16))) float red_all[2][4][ROWF];
    const int lane = threadIdx.x & 63, g = lane >> 3, p = lane & 7, wv = threadIdx.x >> 6;
    const int hl_raw = head_of_lane<3>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(A.xs);
    const unsigned hbit = 1u << hl;
    f4 *ydl = yd_all[wv];
    f4 ys[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        ys[h] = zero4();
        ydl[h * 64 + lane] = zero4();
    }
    const int stride = gridDim.x * 32;
    // The item of the next pass is fetched during this one, a level of its dependent loads at a time
    // (item arrays -> row pointer -> col and keep bits of its first eight edges): four round trips a
    // pass instead of six
    struct Ahead { int row, beg, end, cj, kb; bool on; };
    auto ahead_item = [&](int it0, Ahead &M) {
        const int it = it0 + g;
        M.on = it < A.n_items;
        M.row = M.beg = M.end = M.cj = M.kb = 0;
        if (M.on) {
            M.row = A.item_row[it];
            M.beg = A.item_beg[it];
        }
    };
    auto ahead_end = [&](Ahead &M) {
        if (M.on) M.end = min(A.rowptr[M.row + 1], M.beg + A.seg_len);
    };
    auto ahead_cols = [&](Ahead &M) {
        const int n = min(max(M.end - M.beg, 0), 8);
        const int el = M.beg + min(p, max(n - 1, 0));
        M.cj = n > 0 ? A.col[el] : 0;
        M.kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
    };
    Ahead N1;
    {
        const int first = (xcd_contiguous_block() * 4 + wv) * 8;
        ahead_item(first, N1);
        ahead_end(N1);
        ahead_cols(N1);
    }
    for (int it0 = (xcd_contiguous_block() * 4 + wv) * 8; it0 < A.n_items; it0 += stride) {
        const Ahead M = N1;
        const bool on = M.on;
        const int row = M.row, beg = M.beg, end = M.end;
        ahead_item(it0 + stride, N1);
        f4 q4[8];
        float S;
        if (A.sin) {                                              // wave-uniform: S came with q from the products pass
#pragma unroll
            for (int h = 0; h < 8; ++h)
                q4[h] = (on && pok && h < A.H) ? ldv4(A.q + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
            S = (on && hlok) ? A.sin[(size_t)row * A.H + hl] : 0.f;
        } else {
            float sv[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const bool ld = on && pok && h < A.H;
                q4[h] = ld ? ldv4(A.q + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
                const f4 z4 = ld ? ldv4(A.zin + (size_t)row * HF + h * A.Fp + p * 4) : zero4();
                f2 d = {0.f, 0.f};
                pk_dot4(d, q4[h], z4);
                sv[h] = d.x + d.y;
            }
            S = heads_sum_e<3, 3>(sv, lane);
        }
        const float lse_i = A.lsein[(size_t)row * A.H + hl];
        const float adi = A.a_dst[(size_t)row * A.H + hl];
        float dad = 0.f;
        const int nmax = __builtin_amdgcn_readfirstlane(groups_max_i(end - beg));
        for (int c = 0; c < nmax; c += 8) {
            const int n = min(max(end - beg - c, 0), 8);
            const int nw = min(nmax - c, 8);
            int cj = M.cj, kb = M.kb;
            if (c > 0) {                                          // wave-uniform
                const int el = beg + c + min(p, max(n - 1, 0));
                cj = n > 0 ? A.col[el] : 0;
                kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
            }
            for (int e0 = 0; e0 < nw; e0 += kBwdgE) {             // kBwdgE rows in flight
                if (c == 0 && e0 == kBwdgE) ahead_end(N1);        // the next item's row has arrived by now
                f4 v[kBwdgE];
                float as[kBwdgE];
                unsigned sb[kBwdgE];
#pragma unroll
                for (int e = 0; e < kBwdgE; ++e)
                    if (e0 + e < nw) {                            // wave-uniform
                        const int src = (lane & 56) | min(e0 + e, max(n - 1, 0));
                        const int j = __shfl(cj, src, 64);
                        sb[e] = (unsigned)__shfl(kb, src, 64);
                        v[e] = *reinterpret_cast<const f4 *>(xsb + (size_t)j * row_bytes + offp);
                        as[e] = A.xs[(size_t)j * A.Fs + A.Fp + hl];
                    }
#pragma unroll
                for (int e = 0; e < kBwdgE; ++e)
                    if (e0 + e < nw) {
                        // the chain of (this edge, the head this lane owns)
                        const float uu = as[e] + adi;
                        const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                        const float alpha = (e0 + e < n && hlok) ? exp_fast(uu * slope - lse_i) : 0.f;
                        const float at = (sb[e] & hbit) ? alpha * A.inv_keep : 0.f;
                        float dv[8];
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            f2 d = {0.f, 0.f};
                            pk_dot4(d, v[e], q4[h]);
                            dv[h] = d.x + d.y;
                        }
                        const float d2 = heads_sum_e<3, 3>(dv, lane);
                        const float du = slope * (at * d2 - alpha * S);
                        dad += du;
#pragma unroll
                        for (int h = 0; h < 8; ++h) pk_fma4(ys[h], bcast_head<3>(du, h), v[e]);
                    }
            }
        }
        // d_wd += (sum of the row's d_u) * x_i
        {
            const f4 xi = (on && pok) ? ldv4(A.xs + (size_t)(row + A.dst_off) * A.Fs + p * 4) : zero4();
            if (nmax <= kBwdgE) ahead_end(N1);                    // rows of at most kBwdgE edges all round: not yet asked for
            ahead_cols(N1);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                const float dh = bcast_head<3>(dad, h);
                f4 acc = ydl[h * 64 + lane];
                pk_fma4(acc, dh, xi);
                ydl[h * 64 + lane] = acc;
            }
        }
    }
    // add the 8 groups, then the block's four waves in LDS: ONE partial row per block
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        f4 yd = ydl[h * 64 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            ys[h][k] = high_lanes_reduce<1, 3>(ys[h][k]);
            yd[k] = high_lanes_reduce<1, 3>(yd[k]);
        }
        if (g == 0) {
            stv4(red_all[0][wv] + h * 32 + p * 4, ys[h]);
            stv4(red_all[1][wv] + h * 32 + p * 4, yd);
        }
    }
    __syncthreads();
    float *dst = A.part_y + (size_t)blockIdx.x * 2 * HF;
    for (int idx = threadIdx.x; idx < 2 * ROWF; idx += 256) {
        const int which = idx / ROWF, r = idx % ROWF, hh = r / 32, ff = r % 32;
        if (hh < A.H && ff < A.Fp) {
            const float(*src)[ROWF] = red_all[which];
            dst[which * HF + hh * A.Fp + ff] = (src[0][r] + src[1][r]) + (src[2][r] + src[3][r]);
        }
    }
}

// ---------------------------------------------------------------------------------
// Small kernels around the two passes
// ---------------------------------------------------------------------------------
// Block-diagonal weight of a group of G heads, k-major: out[m][hh*C + c][hh2*Fp + f] =
// W[(m*G + hh)*C + c, f] when hh == hh2 and f < F, else 0.  With C not a multiple of 4 the rows of
// one head of grad_out are not 16-byte aligned; those of a group of G = 4 / gcd(C, 4) heads are,
// and q = g W for the group is one aligned product of K = G C by G Fp columns.
__global__ void inagg_bdiag_kernel(const float *w, int H, int C, int F, int Fp, int G, float *out) {
    const int K = G * C, Nn = G * Fp;
    const long long tot = (long long)(H / G) * K * Nn;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot;
         i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % Nn), k = (int)((i / Nn) % K), m = (int)(i / ((long long)Nn * K));
        const int hh = k / C, c = k % C, hh2 = n / Fp, f = n % Fp;
        out[i] = (hh == hh2 && f < F) ? w[((size_t)(m * G + hh) * C + c) * F + f] : 0.f;
    }
}

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

// Packed source table: xs[n] = x[n] (Fp floats, zero padded) | a_src[n, 0..H) | zero pad, and
// a_dst[n, h] = <x[n], wd_h>.  A row is owned by gl = next_pow2(Fp/4) (<= 64) lanes, each holding
// up to KS float4 of it; per-head dots are butterflies over those lanes.
// GL = lanes per row when known at compile time (DPP butterflies), 0 = run-time width
template <int KS, int GL>
__global__ __launch_bounds__(256) void inagg_pack_kernel(const float *x, const float *wsd, long long n,
                                                         int F, int Fp, int Fs, int H, int glog,
                                                         float *xs, float *a_dst, uint2 *xs16) {
    constexpr int UN = KS == 1 ? 4 : (KS == 2 ? 2 : 1);   // row groups in flight per wave
    extern __shared__ float s_wsd[];                      // ws | wd, [2][H*Fp]
    const int lane = threadIdx.x & 63, gl = 1 << glog, sl = lane & (gl - 1), rsub = lane >> glog;
    const int rpw = 64 >> glog, lph = Fp >> 2, HF = H * Fp;
    for (int i = threadIdx.x; i < 2 * HF; i += 256) s_wsd[i] = wsd[i];
    __syncthreads();
    const bool vec = (F & 3) == 0;
    const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6), nwaves = gridDim.x * 4ll;
    for (long long row0 = wave * rpw * UN; row0 < n; row0 += nwaves * rpw * UN) {
        f4 xv[UN][KS];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long row = row0 + u * rpw + rsub;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int sub = sl + k * gl;
                xv[u][k] = zero4();
                if (row < n && sub < lph) {
                    const float *src = x + row * F + sub * 4;
                    if (vec) {
                        xv[u][k] = ldv4(src);
                    } else {
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (sub * 4 + c < F) xv[u][k][c] = src[c];
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long row = row0 + u * rpw + rsub;
            const bool on = row < n;
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const int sub = sl + k * gl;
                if (on && sub < lph) stv4(xs + row * Fs + sub * 4, xv[u][k]);
            }
            // the bf16 table of the same rows (gatmi_inagg_pack16; rows of 256 bytes = 32 uint2: 28
            // pieces of four channels, zero past Fp, then the f32 logits): the bytes
            // inagg_table16_kernel makes of xs, without reading xs back.  KS == 1 there (Fp <= 112).
            if (KS == 1 && xs16 && on) {
                uint2 *r16 = xs16 + row * (kRow16 / 8);
                const f4 v = xv[u][0];
                if (sl < kRow16Logits / 8)
                    r16[sl] = sl < lph ? make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])) : make_uint2(0u, 0u);
                for (int sb = sl + gl; sb < kRow16Logits / 8; sb += gl) r16[sb] = make_uint2(0u, 0u);
            }
            // four heads at a time: the row's logits leave as one float4 per table (the pad of
            // round_up(H, 4) is the zeros of the heads past H)
            for (int h0 = 0; h0 < H; h0 += 4) {
                f4 a4 = zero4(), d4 = zero4();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int h = h0 + i;
                    if (h < H) {
                        f2 ds = {0.f, 0.f}, dd = {0.f, 0.f};
#pragma unroll
                        for (int k = 0; k < KS; ++k) {
                            const int sub = sl + k * gl;
                            if (sub < lph) {
                                pk_dot4(ds, xv[u][k], ldv4(s_wsd + h * Fp + sub * 4));
                                pk_dot4(dd, xv[u][k], ldv4(s_wsd + HF + h * Fp + sub * 4));
                            }
                        }
                        a4[i] = head_sum<GL>(ds.x + ds.y, gl);
                        d4[i] = head_sum<GL>(dd.x + dd.y, gl);
                    }
                }
                if (on && sl == 0) {
                    stv4(xs + row * Fs + Fp + h0, a4);
                    if (KS == 1 && xs16 && h0 < 8) {
                        uint4 *l16 = reinterpret_cast<uint4 *>(xs16 + row * (kRow16 / 8) + kRow16Logits / 8);
                        l16[h0 >> 2] = make_uint4(__float_as_uint(a4[0]), __float_as_uint(a4[1]),
                                                  __float_as_uint(a4[2]), __float_as_uint(a4[3]));
                        if (H <= 4) l16[1] = make_uint4(0u, 0u, 0u, 0u);
                    }
                    if ((H & 3) == 0) {
                        stv4(a_dst + row * H + h0, d4);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (h0 + i < H) a_dst[row * H + h0 + i] = d4[i];
                    }
                }
            }
        }
    }
}

// dst[col] = sum of nrows partial rows, fixed order.  One block per (job, 16 columns).
// ---------------------------------------------------------------------------------
// d_W slabs  slab[split][h] = g[rows, h, :]^T Z[rows, h, :]  without LDS: the contraction runs
// over ROWS, so both MFMA operands are coalesced dword loads straight into registers
// (v_mfma_f32_32x32x2_f32: lane (l, half) supplies A[m = l][k = half] = g[n0 + half][c = l] and
// B[k = half][n = l] = Z[n0 + half][f = l]).  A wave owns one head and one contiguous row
// range, keeps the whole [32 CT x 32 NT] result in accumulators and D row pairs of operands
// in flight; consecutive waves take the heads of the same rows, so a block reads whole rows.
// The tiled kernel needs LDS staging, barriers and a transposing read for the same product
// and sits at 46 % waitcnt (profiles/r1/pmc_gemm_waits.txt).  Column sums of g (d_bias) ride
// on the A operands.
// ---------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4t __attribute__((ext_vector_type(4)));
constexpr int kDwBank = 6;

template <int CT, int NT, bool TAIL>
__global__ __launch_bounds__(256, 2) void inagg_dw_stream_kernel(
    const float *__restrict__ g, int ldg, const float *__restrict__ z, int ldz, int N, int H, int C,
    int F, int Fp, int nsplit, int rows_per_split, float *__restrict__ slab, float *__restrict__ gsum) {
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= H * nsplit) return;
    const int split = w / H, h = w - split * H;
    const int lane = threadIdx.x & 63, l = lane & 31, half = lane >> 5;
    const long long n_beg = (long long)split * rows_per_split;
    const long long n_end = n_beg + rows_per_split < N ? n_beg + rows_per_split : N;
    const int nrows = n_end > n_beg ? (int)(n_end - n_beg) : 0;
    const int np = nrows >> 1;                               // whole row pairs; an odd last row below
    // addresses = wave-uniform row-pair base (scalar registers) + a 32-bit lane offset
    unsigned go[CT], zo[NT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) go[ct] = (unsigned)(half * ldg + h * C + min(l + 32 * ct, C - 1)) * 4u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) zo[nt] = (unsigned)(half * ldz + h * Fp + min(l + 32 * nt, Fp - 1)) * 4u;
    // TAIL (CT == 2): columns NT*32 .. NT*32+3 through v_mfma_f32_4x4x1_16B_f32, one row per
    // instruction: lane c supplies g[n][c] (the two halves of the 32x32 A operands, swapped into
    // place), lane 4b + j supplies Z[n][NT*32 + j]; block b holds rows 4b .. 4b+3 of the result.
    unsigned zt[2];
    zt[0] = (unsigned)(h * Fp + min(NT * 32 + (lane & 3), Fp - 1)) * 4u;
    zt[1] = zt[0] + (unsigned)ldz * 4u;
    f32x4t acct = {0.f, 0.f, 0.f, 0.f};
    const char *gb0 = reinterpret_cast<const char *>(g + (size_t)n_beg * ldg);
    const char *zb0 = reinterpret_cast<const char *>(z + (size_t)n_beg * ldz);
    f32x16 acc[CT][NT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][nt][r] = 0.f;
    float gs[CT], cmask[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        gs[ct] = 0.f;
        cmask[ct] = l + 32 * ct < C ? 1.f : 0.f;
    }
    // Two banks of kDwBank row pairs, one in flight while the other is multiplied (the pattern
    // the edge kernels use; hipcc counts vmcnt across the loop for it, which it does not for a
    // per-step ring: it drained every load at the top of each trip).
    struct Bank {
        float ga[kDwBank][CT], zb[kDwBank][NT], zt[kDwBank][TAIL ? 2 : 1];
    };
    auto load = [&](Bank &bk, int p0) {
#pragma unroll
        for (int d = 0; d < kDwBank; ++d) {
            const int pc = p0 + d < np ? p0 + d : np - 1;     // past the end: reload, unused
            const char *gr = gb0 + (size_t)pc * (size_t)(2 * ldg) * 4;
            const char *zr = zb0 + (size_t)pc * (size_t)(2 * ldz) * 4;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bk.ga[d][ct] = *reinterpret_cast<const float *>(gr + go[ct]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bk.zb[d][nt] = *reinterpret_cast<const float *>(zr + zo[nt]);
            if constexpr (TAIL) {
                bk.zt[d][0] = *reinterpret_cast<const float *>(zr + zt[0]);
                bk.zt[d][1] = *reinterpret_cast<const float *>(zr + zt[1]);
            }
        }
    };
    auto mult = [&](const Bank &bk, int p0) {
#pragma unroll
        for (int d = 0; d < kDwBank; ++d) {
            if (p0 + d < np) {                                // wave-uniform
                float a[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    a[ct] = bk.ga[d][ct] * cmask[ct];
                    gs[ct] += a[ct];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ct], bk.zb[d][nt], acc[ct][nt], 0, 0, 0);
                }
                if constexpr (TAIL) {
                    const float o0 = __shfl_xor(a[0], 32, 64), o1 = __shfl_xor(a[CT - 1], 32, 64);
                    const float r0 = half ? o1 : a[0];            // g[n0][lane]
                    const float r1 = half ? a[CT - 1] : o0;       // g[n0 + 1][lane]
                    acct = __builtin_amdgcn_mfma_f32_4x4x1f32(r0, bk.zt[d][0], acct, 0, 0, 0);
                    acct = __builtin_amdgcn_mfma_f32_4x4x1f32(r1, bk.zt[d][1], acct, 0, 0, 0);
                }
            }
        }
    };
    if (np > 0) {
        Bank ba, bb;
        load(ba, 0);
        for (int p0 = 0; p0 < np; p0 += 2 * kDwBank) {
            load(bb, p0 + kDwBank);
            mult(ba, p0);
            load(ba, p0 + 2 * kDwBank);
            mult(bb, p0 + kDwBank);
        }
    }
    if (nrows & 1) {                                          // the odd last row: k = 1 is zero
        const size_t row = (size_t)(n_end - 1);
        if constexpr (TAIL) {
            const float gv = g[row * ldg + h * C + min(lane, C - 1)] * (lane < C ? 1.f : 0.f);
            const float zv = z[row * ldz + h * Fp + min(NT * 32 + (lane & 3), Fp - 1)];
            acct = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, zv, acct, 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float v = g[row * ldg + h * C + min(l + 32 * ct, C - 1)];
            const float a = v * (half == 0 ? cmask[ct] : 0.f);
            gs[ct] += a;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float bv = z[row * ldz + h * Fp + min(l + 32 * nt, Fp - 1)];
                acc[ct][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[ct][nt], 0, 0, 0);
            }
        }
    }
    float *dst = slab + ((size_t)split * H + h) * C * F;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int f = nt * 32 + l;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (c < C && f < F) dst[(size_t)c * F + f] = acc[ct][nt][r];
            }
        }
    if constexpr (TAIL) {
        const int f = NT * 32 + (lane & 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 4 * (lane >> 2) + i;
            if (c < C && f < F) dst[(size_t)c * F + f] = acct[i];
        }
    }
    if (gsum) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const float tot = gs[ct] + __shfl_xor(gs[ct], 32, 64);
            const int c = ct * 32 + l;
            if (half == 0 && c < C) gsum[((size_t)h * nsplit + split) * C + c] = tot;
        }
    }
}

struct SumJob {
    const float *src;
    float *dst;
    int nrows, ld, ncols;
};
__global__ __launch_bounds__(256) void inagg_sum_rows_kernel(SumJob j0, SumJob j1, SumJob j2, SumJob j3 = SumJob{}) {
    __shared__ float red[256];
    const SumJob J = blockIdx.y == 0 ? j0 : (blockIdx.y == 1 ? j1 : (blockIdx.y == 2 ? j2 : j3));
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
    const float *att_src, const float *att_dst, const float *dwsd, const float *gsum /* [H][nsplit][C] */,
    float *d_w, float *d_att_src, float *d_att_dst, float *d_bias) {
    const long long tot = (long long)H * C * F;
    const float *dws = dwsd, *dwd = dwsd + (size_t)H * Fp;
    // A block sums 64 consecutive outputs; its four waves take the splits z = wave, wave + 4, ...
    // with eight loads in flight each (the splits number in the hundreds: one thread walking
    // them all is a chain of memory latencies), and meet in LDS.
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long long base = blockIdx.x * 64ll; base < tot; base += gridDim.x * 64ll) {
        const long long i = base + lane;
        float s[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) s[u] = 0.f;
        if (i < tot)
            for (int z = wave; z < nsplit; z += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (z + 4 * u < nsplit) s[u] += slab[(size_t)(z + 4 * u) * tot + i];
            }
        red[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        __syncthreads();
        if (wave == 0 && i < tot) {
            const int f = (int)(i % F), hc = (int)(i / F), h = hc / C;
            float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            v = fmaf(att_src[hc], dws[h * Fp + f], v);
            v = fmaf(att_dst[hc], dwd[h * Fp + f], v);
            if (d_w) d_w[i] = v;
        }
        __syncthreads();
    }
    const int HC = H * C;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 2 * HC; i += gridDim.x * 256) {
        const int k = i / HC, hc = i % HC, h = hc / C;
        const float *dv = (k == 0 ? dws : dwd) + h * Fp;
        float sacc = 0.f;
        for (int f = 0; f < F; ++f) sacc = fmaf(w[(size_t)hc * F + f], dv[f], sacc);
        float *dst = k == 0 ? d_att_src : d_att_dst;
        if (dst) dst[hc] = sacc;
    }
    if (d_bias)   // column sums of grad_out, collected by the d_W kernel from its A operands
        for (int base = blockIdx.x * 64; base < HC; base += gridDim.x * 64) {
            const int i = base + lane;
            float s[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] = 0.f;
            if (i < HC) {
                const int h = i / C, c = i % C;
                const float *src = gsum + (size_t)h * nsplit * C + c;
                for (int z = wave; z < nsplit; z += 32) {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (z + 4 * u < nsplit) s[u] += src[(size_t)(z + 4 * u) * C];
                }
            }
            red[wave][lane] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
            __syncthreads();
            if (wave == 0 && i < HC)
                d_bias[i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
            __syncthreads();
        }
}

// ---------------------------------------------------------------------------------
// Fused tiles of the row-per-lane-group layout: run.py's conv1, 25 -> 8 x 30 on
// a 5-nearest-neighbour graph, and its neighbours in the reference's searches (5..8 heads, rows
// of at most 8 float4, C <= 64, no row cut into segments).  A block owns tiles of 32 CONSECUTIVE
// destination rows, a wave 8 of them (one per 8-lane group, as in inagg_fwdg / inagg_bwdg), and
// the products that the separate GEMMs made of whole [N, H*Fp] tensors are MFMA tiles of the
// block, each wave multiplying the two heads 2 wv, 2 wv + 1:
//   forward   Z tile (registers -> LDS) -> out tile = Z W^T + bias        (Z still goes to HBM once:
//             the backward reads it for S and d_W; the out GEMM's re-read of it is gone)
//   backward  g tile (HBM -> the wave's LDS) -> q tile = g W (-> LDS, never in HBM),
//             d_W += g^T Z and the column sums of g in persistent accumulators, then the edge
//             pass of the tile with q from LDS.  One slab of d_W per block, summed by
//             inagg_finalize_kernel as the split-K slabs were.
// MFMA operands (v_mfma_f32_32x32x2_f32): a lane supplies A[m = lane % 32][k = lane / 32].  Tiles
// whose m is the ROW index are read from LDS as ds_read_b64 -- two consecutive k of one row,
// used as two MFMA steps (the K order of a product is free as long as A and B agree) -- with a
// row stride of 2 mod 4 floats: 32 rows x 2 banks cover the 64 banks once.
// ---------------------------------------------------------------------------------
#ifndef GATMI_BWDGF_E   // gathered rows in flight per lane group of the fused backward: 3 / 4 / 8 run 1.75 / 1.95 / 2.11 against 1.64 ms
                        // at 10^6 rows (spills), and no better with q of the row read from LDS per use instead of held (1.74 / 2.08 / 2.41)
#define GATMI_BWDGF_E 2
#endif
#ifndef GATMI_FWDGF_OCC  // blocks per CU the fused forward (C <= 32) is compiled for: 4 (128 registers) 0.89 ms at 10^6 rows, 3: 0.97
#define GATMI_FWDGF_OCC 4
#endif
struct FuseArgs {
    const float *w;      // [H*C, F]
    const float *wimg;   // forward: [H][Fp][32 CT] = W^T per head, zero padded (inagg_wimg_kernel)
    const float *bias;   // [H*C] or null (forward)
    const float *gout;   // grad_out [N, H*C] (backward)
    float *out;          // [N, H*C] (forward)
    float *slab;         // [blocks][H][C][F] (backward)
    float *gsum;         // [blocks][H*C] or null (backward)
    float *qout;         // backward without the edge pass: q [N, H*Fp]
    float *sout;         //   and S [N, H] = <q, Z> per (row, head), what the edge pass needs of Z
    int C, F, N, ntiles;
    int dbg;             // GATMI_FUSE_DBG: ablation bits (wrong results; timing only)
    int zs;              // forward: row stride of the Z tile; backward: of the q tile (floats)
    int gs, cp4;         // backward: row stride of a wave's g tile, columns per head in it
    int lpr_log;         // backward: log2 of the lanes that load one head's columns of a row of grad_out
};

__device__ __forceinline__ void st_lds2(float *p, float a, float b) {
    *reinterpret_cast<f2 *>(p) = f2{a, b};
}

// Barrier of the block for LDS hand-overs only: waits for the wave's LDS operations, not for its
// global stores (__syncthreads drains vmcnt too: the Z and out rows of a tile would be waited for
// twice per tile)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// What a lane needs of its row before the first gather of a tile, fetched one tile ahead: the
// rowptr pair and the row's a_dst / lse while the tile before is in its edge pass, col and keep
// bits of the first 8 edges while it is in its products -- a tile then starts with the gathers
// themselves instead of a chain of three dependent loads
struct TileMeta {
    int beg, end, cj, kb;
    float adi, lse;
};

// wimg[h][k][c] = W[h C + c][k] for c < C, k < F, else 0 (k < Fp, c < cw = 32 CT)
__global__ void inagg_wimg_kernel(const float *w, int H, int C, int F, int Fp, int cw, float *wimg) {
    const int tot = H * Fp * cw;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x) {
        const int c = i % cw, k = (i / cw) % Fp, h = i / (cw * Fp);
        wimg[i] = (c < C && k < F) ? w[((size_t)h * C + c) * F + k] : 0.f;
    }
}

template <int CT>
__global__ __launch_bounds__(256, CT == 1 ? GATMI_FWDGF_OCC : 3) void inagg_fwdgf_kernel(const InArgs A, const FuseArgs P) {
    extern __shared__ __attribute__((aligned(16))) float fuse_lds[];
    float *zt = fuse_lds;                                         // [32 rows][zs]
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    const int lane = threadIdx.x & 63, g = lane >> 3, p = lane & 7;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: heads, LDS corners and offsets follow from it
    const int l = lane & 31, half = lane >> 5;
    const int hl_raw = head_of_lane<3>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(A.xs);
    const unsigned hbit = 1u << hl;
    // B operands of out[:, h, :] = Z[:, h, :] W_h^T for the wave's two heads: lane (n = l, half)
    // holds W[h C + c][k] for the k pairs 4 t + 2 half, + 1 the A reads deliver.  Loaded per tile
    // (24 KiB of weights: cache hits) once the edge pass has let go of its registers, ahead of the barrier
    auto load_wf = [&](float (&wf)[2][CT][16], const float *wimg) {
        // from the k-major image inagg_wimg_kernel makes of W ([H][Fp][32 CT], zero padded): a
        // half-wave reads 128 consecutive bytes (from W itself each lane would touch a line of its
        // own: 64 tag look-ups an instruction, 2048 a tile and wave -- the kernel was bound by them);
        // head, k and column tile in the scalar offset, the quads past Fp run off the image and read 0
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(wimg), 0, A.H * A.Fp * CT * 32 * 4, 0x00020000);
        const int vo = (2 * half * CT * 32 + l) * 4;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = min(2 * wv + hh, A.H - 1);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int so = t < A.lph ? (((h * A.Fp + 4 * t + j) * CT + ct) * 32) * 4 : 0x7ffffff0;
                        wf[hh][ct][2 * t + j] =
                            __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, vo, so, 0));
                    }
        }
    };
    auto meta_row = [&](int tile, TileMeta &M) {
        const int row = tile * 32 + wv * 8 + g;
        const bool on = tile < P.ntiles && row < P.N;
        M.beg = M.end = 0;
        if (on) {
            M.beg = A.rowptr[row];
            M.end = A.rowptr[row + 1];
        }
        M.adi = A.a_dst[(size_t)(on ? row : 0) * A.H + hl];
    };
    auto meta_cols = [&](TileMeta &M) {
        const int n = min(max(M.end - M.beg, 0), 8);
        const int el = M.beg + min(p, max(n - 1, 0));
        M.cj = n > 0 ? A.col[el] : 0;
        M.kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
    };
    int tile = xcd_contiguous_block();
    TileMeta N1;
    meta_row(tile, N1);
    meta_cols(N1);
    for (; tile < P.ntiles; tile += gridDim.x) {
        const int row = tile * 32 + wv * 8 + g;
        const bool on = row < P.N;
        const TileMeta M = N1;
        meta_row(tile + gridDim.x, N1);                           // in flight across the edge pass
        const int beg = M.beg, end = M.end;
        const float adi = M.adi;
        float m = -INFINITY, ssum = 0.f;
        f4 acc[8];
#pragma unroll
        for (int h = 0; h < 8; ++h) acc[h] = zero4();
        const int nmax = __builtin_amdgcn_readfirstlane(groups_max_i(end - beg));
        for (int c = 0; c < nmax; c += 8) {
            const int n = min(max(end - beg - c, 0), 8);          // this group's edges of the chunk
            const int nw = min(nmax - c, 8);                      // the wave's longest
            int cj = M.cj, kb = M.kb;
            if (c > 0) {                                          // wave-uniform
                const int el = beg + c + min(p, max(n - 1, 0));
                cj = n > 0 ? A.col[el] : 0;
                kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
            }
            f4 v[8];
            float z[8];
            unsigned sb[8];
            float zmax = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                z[e] = -INFINITY;
                if (e < nw) {                                     // wave-uniform
                    const int src = (lane & 56) | min(e, max(n - 1, 0));
                    const int j = __shfl(cj, src, 64);
                    sb[e] = (unsigned)__shfl(kb, src, 64);
                    v[e] = *reinterpret_cast<const f4 *>(xsb + (size_t)j * row_bytes + offp);
                    const float uu = A.xs[(size_t)j * A.Fs + A.Fp + hl] + adi;
                    z[e] = e < n ? (uu > 0.f ? uu : A.neg_slope * uu) : -INFINITY;
                }
                zmax = fmaxf(zmax, z[e]);
            }
            const float mn = fmaxf(m, zmax);
            const float sc = mn == -INFINITY ? 1.0f : exp_quick(m - mn);
            float psum = 0.f, w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                w[e] = 0.f;
                if (e < nw) {
                    const float pe = e < n ? exp_quick(z[e] - mn) : 0.f;
                    psum += pe;
                    w[e] = (sb[e] & hbit) ? pe * A.inv_keep : 0.f;
                }
            }
            ssum = fmaf(ssum, sc, psum);
            m = mn;
            if (c > 0) {
#pragma unroll
                for (int h = 0; h < 8; ++h) pk_scale4(acc[h], bcast_head<3>(sc, h));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e < nw) {
#pragma unroll
                    for (int h = 0; h < 8; ++h) pk_fma4(acc[h], bcast_head<3>(w[e], h), v[e]);
                }
        }
        meta_cols(N1);                                            // in flight across the products
        const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
        // Sizes the compiler cannot see through: everything derived from them (lane addresses of the
        // weight loads, of the tile reads and of the 16 output rows) is loop-invariant, would be
        // hoisted out of the tile loop, held across the edge pass and spilled
        int zs = P.zs, C = P.C, F = P.F;
        asm volatile("" : "+s"(zs), "+s"(C), "+s"(F));
        const int HCl = A.H * C;
        const int trows = min(32, P.N - tile * 32);
        lds_barrier();                                            // the tile before has left the LDS
        // the Z row -> the LDS tile (zeros past N); it goes to HBM from there
        float *ztr = zt + (wv * 8 + g) * zs + p * 4;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const float ivh = bcast_head<3>(inv, h);
            if (pok && h < A.H) {
                f4 o = acc[h];
                pk_scale4(o, ivh);
                st_lds2(ztr + h * A.Fp, o[0], o[1]);
                st_lds2(ztr + h * A.Fp + 2, o[2], o[3]);
            }
        }
        if (on && A.lse && hlok) A.lse[(size_t)row * A.H + hl] = m + logf(ssum);
        float wf[2][CT][16];
        load_wf(wf, P.wimg + (zs - P.zs));
        lds_barrier();                                            // the Z tile is whole
        // Z rows of the tile are one contiguous block of HBM: whole lines, 512 bytes a wave
        // instruction (stored from the edge pass they were 112-byte pieces of 8 rows an instruction:
        // 3.9 TB/s, and the out rows in 120-byte pieces off the MFMA fragments 2.1 TB/s)
        {
            const int hf2 = HF >> 1;                              // float2 per row
            float *zg = A.z + (size_t)tile * 32 * HF;
            for (int r = wv; r < trows; r += 4)
                for (int c2 = lane; c2 < hf2; c2 += 64)
                    *reinterpret_cast<f2 *>(zg + r * HF + 2 * c2) = *reinterpret_cast<const f2 *>(zt + r * zs + 2 * c2);
        }
        f32x16 d[2][CT];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = min(2 * wv + hh, A.H - 1);
            const float *ap = zt + l * zs + h * A.Fp + 2 * half;
            // branch-free: the quads past the row re-read its last one against zero weights
            f2 a2[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) a2[t] = *reinterpret_cast<const f2 *>(ap + 4 * min(t, A.lph - 1));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[hh][ct][r] = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    d[hh][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t].x, wf[hh][ct][2 * t], d[hh][ct], 0, 0, 0);
                    d[hh][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t].y, wf[hh][ct][2 * t + 1], d[hh][ct], 0, 0, 0);
                }
            }
        }
        lds_barrier();                                            // the Z tile has been read
        // out tile = products + bias -> LDS as the image of its 32 rows in HBM (over the Z tile)
        float *ot = fuse_lds;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int h = 2 * wv + hh;
            if (h < A.H) {                                        // wave-uniform
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    const int c = ct * 32 + l;
                    if (c < C) {
                        const float bv = P.bias ? P.bias[h * C + c] : 0.f;
                        float *op = ot + 4 * half * HCl + h * C + c;
#pragma unroll
                        for (int r = 0; r < 16; ++r) op[((r & 3) + 8 * (r >> 2)) * HCl] = d[hh][ct][r] + bv;
                    }
                }
            }
        }
        lds_barrier();                                            // the out tile is whole
        {
            float *og = P.out + (size_t)tile * 32 * HCl;
            const int n4 = trows * HCl >> 2;                      // 32 HC is a multiple of 4; a ragged last tile
            for (int i = threadIdx.x; i < n4; i += 256)           // with HC % 4 != 0 ends in single floats
                stv4(og + 4 * i, ldv4(ot + 4 * i));
            for (int i = 4 * n4 + threadIdx.x; i < trows * HCl; i += 256) og[i] = ot[i];
        }
    }
}

// EDGE == false: the products only -- q = g W written to HBM (P.qout, whole lines through the LDS tile),
// d_W slabs and g sums as above; the edge pass then runs as inagg_bwdg_kernel.  For graphs beyond
// kFuseBwdMaxRows: one pass over grad_out instead of the two of the separate q and d_W products.
template <int CT, int VW, bool EDGE>
__global__ __launch_bounds__(256, 2) void inagg_bwdgf_kernel(const InArgs A, const FuseArgs P) {
    extern __shared__ __attribute__((aligned(16))) float fuse_lds[];
    // LDS: q of the tile [32 rows][zs] | per wave: g, its two heads [32 rows][gs] | per wave: [8][8]
    const KeepSel ks = keep_sel(A.mask_mode, A.mask, A.seed);
    const int lane = threadIdx.x & 63, g = lane >> 3, p = lane & 7;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // scalar: heads, LDS corners and offsets follow from it
    const int l = lane & 31, half = lane >> 5;
    const int hl_raw = head_of_lane<3>(lane);
    const bool hlok = hl_raw < A.H;
    const int hl = hlok ? hl_raw : 0;
    const bool pok = p < A.lph;
    const unsigned offp = pok ? (unsigned)p * 16u : 0u;
    const int HF = A.H * A.Fp;
    const size_t row_bytes = (size_t)A.Fs * 4;
    const char *xsb = reinterpret_cast<const char *>(A.xs);
    const unsigned hbit = 1u << hl;
    const int h0 = 2 * wv;
    const int nh = max(min(2, A.H - h0), 0);                      // heads this wave multiplies
    constexpr int K4 = CT * 8;                                    // k quads of q = g W (C <= 32 CT)
    constexpr int kE = GATMI_BWDGF_E;                             // gathered rows in flight per lane group
    constexpr int kOob = 0x7ffffff0;                              // an offset past every buffer: the load returns 0
    constexpr int NI = CT * 16 / VW;                              // load instructions per head of a g tile at most
    for (int i = lane; i < 32 * P.gs; i += 64)                    // the pad columns stay zero
        fuse_lds[32 * P.zs + wv * 32 * P.gs + i] = 0.f;
    // B operands of q[:, h, :] = g[:, h, :] W_h: lane (half, n = l) holds W[h C + k][l] for the k
    // pairs 4 t + 2 half, + 1.  Loaded per tile (cache hits): held across the edge pass they spill
    float wq[2][2 * K4];
    auto load_wq = [&](const float *wbase, int C, int F, int nk4) {
        // branch-free: a buffer over each head's block of W, row k (scalar offset) at column l (one
        // lane offset); rows beyond C run past the block and read 0, as do the columns beyond F
        const int vo = l < F ? (2 * half * F + l) * 4 : 0x7ffffff0;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(wbase) + (size_t)min(h0 + hh, A.H - 1) * C * F, 0, C * F * 4, 0x00020000);
#pragma unroll
            for (int t = 0; t < K4; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    wq[hh][2 * t + j] = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(wrs, vo, (4 * t + j) * F * 4, 0));
        }
    };
    auto meta_row = [&](int tile, TileMeta &M) {
        const int row = tile * 32 + wv * 8 + g;
        const bool on = tile < P.ntiles && row < P.N;
        M.beg = M.end = 0;
        if (on) {
            M.beg = A.rowptr[row];
            M.end = A.rowptr[row + 1];
        }
        M.adi = A.a_dst[(size_t)(on ? row : 0) * A.H + hl];
        M.lse = A.lsein[(size_t)(on ? row : 0) * A.H + hl];
    };
    auto meta_cols = [&](TileMeta &M) {
        const int n = min(max(M.end - M.beg, 0), 8);
        const int el = M.beg + min(p, max(n - 1, 0));
        M.cj = n > 0 ? A.col[el] : 0;
        M.kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
    };
    // B of d_W: Z[row][h Fp + l] of the tile `zrs` spans.  Step s of the contraction over the tile's rows
    // takes row (s & 3) + 8 (s >> 2) + 4 half -- the order in which an MFMA result holds its rows (the K
    // order of a product is free), so that zb[s] lines up with element s of the q fragment: S = <q, Z> of
    // a (row, head) is then a reduction of 16 products over the 32 lanes of a half-wave
    const int zvo = l < A.Fp ? (4 * half * HF + l) * 4 : kOob;
    auto load_zb = [&](const __amdgpu_buffer_rsrc_t zrs, int hh, float (&zb)[16]) {
        const int so = min(h0 + hh, A.H - 1) * A.Fp * 4;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            zb[s] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(zrs, zvo, so + ((s & 3) + 8 * (s >> 2)) * HF * 4, 0));
    };
    // The wave's columns of the 32 rows of grad_out of a tile: per head, LPR lanes a row (VW floats
    // each), 64 / LPR rows an instruction; buffer loads over the tile's rows (one lane offset, the
    // rest in scalar offsets, rows past N and tiles past the last read 0).  Requested a tile ahead
    // (request_g), written to the wave's LDS tile when the tile starts (place_g).
    const int lpr_log = P.lpr_log, rpi = 64 >> lpr_log, ni = 32 >> (6 - lpr_log);
    const int le = lane & ((1 << lpr_log) - 1), lr = lane >> lpr_log;
    auto request_g = [&](int t, int C, float (&vg)[2][NI][VW]) {
        const int HC = A.H * C, r0 = t * 32;
        const int rows = t < P.ntiles ? min(32, P.N - r0) : 0;
        const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(P.gout) + (size_t)(rows > 0 ? r0 : 0) * HC, 0, rows * HC * 4, 0x00020000);
        const int gvo = le * VW < C ? (lr * HC + le * VW) * 4 : kOob;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int so = min(h0 + hh, A.H - 1) * C * 4;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                float(&vv)[VW] = vg[hh][u];
                if (u < ni && hh < nh) {                          // wave-uniform
                    const int sof = so + u * rpi * HC * 4;
                    if constexpr (VW == 4) {
                        const f4 t4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(grs, gvo, sof, 0));
                        vv[0] = t4[0]; vv[1] = t4[1]; vv[2] = t4[2]; vv[3] = t4[3];
                    } else if constexpr (VW == 2) {
                        const f2 t2 = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(grs, gvo, sof, 0));
                        vv[0] = t2.x; vv[1] = t2.y;
                    } else {
                        vv[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, gvo, sof, 0));
                    }
                }
            }
        }
    };
    auto place_g = [&](const float (&vg)[2][NI][VW], float *gtw, int gs, int cp4, int C) {
        if (le * VW < C) {
            float *gl = gtw + lr * gs + le * VW;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int u = 0; u < NI; ++u)
                    if (u < ni && hh < nh) {
                        const float(&vv)[VW] = vg[hh][u];
                        float *dp = gl + u * rpi * gs + hh * cp4;
                        if constexpr (VW == 4) {
                            st_lds2(dp, vv[0], vv[1]);
                            st_lds2(dp + 2, vv[2], vv[3]);
                        } else if constexpr (VW == 2) {
                            st_lds2(dp, vv[0], vv[1]);
                        } else {
                            *dp = vv[0];
                        }
                    }
        }
    };
    f32x16 dw[2][CT];
    float gsa[2][CT];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            gsa[hh][ct] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dw[hh][ct][r] = 0.f;
        }
    f4 ys[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) ys[h] = zero4();
    f32x4t ydm[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // d_wd [head][f] by 16x16x4 MFMAs, f < 16 | f >= 16
    int tile = xcd_contiguous_block();
    TileMeta N1{};
    if constexpr (EDGE) {
        meta_row(tile, N1);
        meta_cols(N1);
    }
    float vg[2][NI][VW], zb[16];
    request_g(tile, P.C, vg);
    {
        const int rows = tile < P.ntiles ? min(32, P.N - tile * 32) : 0;
        load_zb(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.zin) + (size_t)(rows > 0 ? tile * 32 : 0) * HF, 0,
                                                  rows * HF * 4, 0x00020000), 0, zb);
    }
    for (; tile < P.ntiles; tile += gridDim.x) {
        const int row0 = tile * 32;
        // Sizes the compiler cannot see through: what is derived from them (lane addresses of the
        // weight loads, of the tile reads and stores) is loop-invariant, would be hoisted out of the
        // tile loop, held across the edge pass and spilled
        int zs = P.zs, gs = P.gs, cp4 = P.cp4, C = P.C, F = P.F;
        asm volatile("" : "+s"(zs), "+s"(gs), "+s"(cp4), "+s"(C), "+s"(F));
        const int nk4 = cp4 >> 2;
        float *qt = fuse_lds;
        float *gtw = fuse_lds + 32 * zs + wv * 32 * gs;
        float *dtw = fuse_lds + 32 * zs + 4 * 32 * gs + wv * 64;
        float *stile = fuse_lds + 32 * zs + 4 * 32 * gs + 4 * 64;    // [32 rows][8 heads]: S of the tile
        (void)stile;
        // 0. grad_out and the first head's Z tile were requested at the end of the tile before (or
        //    ahead of the loop): grad_out goes to the wave's LDS tile now
        const int trows = min(32, P.N - row0);
        const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(A.zin) + (size_t)row0 * HF, 0, trows * HF * 4, 0x00020000);
        place_g(vg, gtw, gs, cp4, C);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // the loads the products and the edge pass wait for later, issued once the staging registers
        // of grad_out are free: weight fragments, the second head's Z tile
        if (!(P.dbg & 32)) load_wq(P.w, C, F, nk4);
        float zn[16];
        if (nh > 1 && !(P.dbg & 32)) load_zb(zrs, 1, zn);
        __builtin_amdgcn_sched_barrier(0);
        // 2. d_W += g^T Z with the column sums of g on the A operands (needs no weights: runs while
        //    they arrive);  3. q tile of the wave's heads -> LDS
#pragma unroll
        for (int hh = 0; hh < 2; ++hh)
            if (hh < nh) {                                        // wave-uniform
                const int h = h0 + hh;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    if (P.dbg & 2) break;
                    const int c = ct * 32 + l;
                    const float *gp = gtw + 4 * half * gs + hh * cp4 + min(c, C - 1);
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        float a = gp[((s & 3) + 8 * (s >> 2)) * gs];
                        a = c < C ? a : 0.f;
                        gsa[hh][ct] += a;
                        dw[hh][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, hh == 0 ? zb[s] : zn[s], dw[hh][ct], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);                // operands of one product at a time in registers
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                const float *ap = gtw + l * gs + hh * cp4 + 2 * half;
#pragma unroll
                for (int t = 0; t < K4; ++t) {                    // branch-free: quads past C against zero weights
                    if (P.dbg & 1) break;
                    if (CT > 1 && t >= nk4) break;                // two channel tiles: no steps against zero weights
                    const f2 a2 = *reinterpret_cast<const f2 *>(ap + 4 * min(t, nk4 - 1));
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, wq[hh][2 * t], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, wq[hh][2 * t + 1], d, 0, 0, 0);
                }
                if (l < A.Fp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rb = (r & 3) + 8 * (r >> 2);
                        (qt + rb * zs)[4 * half * zs + h * A.Fp + l] = d[r];
                    }
                }
                {   // S[row, h] = <q[row, h, :], Z[row, h, :]> = sum_e alpha_e d alpha_e of the row: 16 rows per
                    // lane, summed over the half-wave's 32 columns by 16-to-1 trading stages and one step
                    float t[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = d[r] * (hh == 0 ? zb[r] : zn[r]);
                    trade_stage<0, 16>(t, lane);
                    trade_stage<1, 16>(t, lane);
                    trade_stage<2, 16>(t, lane);
                    trade_stage<3, 16>(t, lane);
                    const float sv = t[0] + lane_xor<16>(t[0]);
                    // lane l of a half now holds row index bitrev4(l mod 16) of the fragment order
                    const int r = ((l & 1) << 3) | ((l & 2) << 1) | ((l & 4) >> 1) | ((l & 8) >> 3);
                    const int rowt = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (l < 16) {
                        if constexpr (EDGE) stile[rowt * 8 + h] = sv;
                        else if (row0 + rowt < P.N) P.sout[(size_t)(row0 + rowt) * A.H + h] = sv;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        lds_barrier();                                            // the q tile is whole
        if constexpr (!EDGE) {
            // q rows of the tile are one contiguous block of HBM: whole lines from the LDS tile
            float *qg = P.qout + (size_t)row0 * HF;
            for (int r = wv; r < trows; r += 4)
                for (int c4 = lane; c4 < (HF >> 2); c4 += 64)
                    stv4(qg + r * HF + 4 * c4, ldv4(qt + r * zs + 4 * c4));
            {   // the next tile's grad_out columns and first Z tile
                const int tn = tile + gridDim.x;
                request_g(tn, C, vg);
                const int rows = tn < P.ntiles ? min(32, P.N - tn * 32) : 0;
                load_zb(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.zin) + (size_t)(rows > 0 ? tn * 32 : 0) * HF,
                                                          0, rows * HF * 4, 0x00020000), 0, zb);
            }
            lds_barrier();                                        // before the next tile's q stores
            continue;
        }
        // 4. the edge pass of the wave's 8 rows (inagg_bwdg_kernel's, q from LDS)
        const TileMeta M = N1;
        meta_row(tile + gridDim.x, N1);                           // in flight across the edge pass
        const int beg = M.beg, end = M.end;
        f4 q4[8];
        const float *qr = qt + (wv * 8 + g) * zs + p * 4;
#pragma unroll
        for (int h = 0; h < 8; ++h) q4[h] = (pok && h < A.H) ? ldv4(qr + h * A.Fp) : zero4();
        const float S = hlok ? stile[(wv * 8 + g) * 8 + hl] : 0.f;
        // B of the d_wd steps after the edge pass: the wave's own x rows, requested ahead of it
        float xb[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int rr = row0 + wv * 8 + 4 * s + (lane >> 4);
            const float *xr = A.xs + (size_t)(min(rr, P.N - 1) + A.dst_off) * A.Fs;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int f = 16 * nt + (lane & 15);
                xb[s][nt] = 16 * nt < A.Fp ? xr[min(f, A.Fp - 1)] : 0.f;
                xb[s][nt] = (rr < P.N && f < A.Fp) ? xb[s][nt] : 0.f;
            }
        }
        const float lse_i = M.lse;
        const float adi = M.adi;
        float dad = 0.f;
        const int nmax = (P.dbg & 8) ? 0 : __builtin_amdgcn_readfirstlane(groups_max_i(end - beg));
        for (int c = 0; c < nmax; c += 8) {
            const int n = min(max(end - beg - c, 0), 8);
            const int nw = min(nmax - c, 8);
            int cj = M.cj, kb = M.kb;
            if (c > 0) {                                          // wave-uniform
                const int el = beg + c + min(p, max(n - 1, 0));
                cj = n > 0 ? A.col[el] : 0;
                kb = n > 0 ? (int)keep_bits(ks.mode, A.mask, ks.seed, A.keep_thr, (long long)el, A.H) : 0;
            }
            for (int e0 = 0; e0 < nw; e0 += kE) {             // kE rows in flight
                f4 v[kE];
                float as[kE];
                unsigned sb[kE];
#pragma unroll
                for (int e = 0; e < kE; ++e)
                    if (e0 + e < nw) {                            // wave-uniform
                        const int src = (lane & 56) | min(e0 + e, max(n - 1, 0));
                        const int j = __shfl(cj, src, 64);
                        sb[e] = (unsigned)__shfl(kb, src, 64);
                        v[e] = *reinterpret_cast<const f4 *>(xsb + (size_t)j * row_bytes + offp);
                        as[e] = A.xs[(size_t)j * A.Fs + A.Fp + hl];
                    }
#pragma unroll
                for (int e = 0; e < kE; ++e)
                    if (e0 + e < nw) {
                        const float uu = as[e] + adi;
                        const float slope = uu > 0.f ? 1.0f : A.neg_slope;
                        const float alpha = (e0 + e < n && hlok) ? exp_fast(uu * slope - lse_i) : 0.f;
                        const float at = (sb[e] & hbit) ? alpha * A.inv_keep : 0.f;
                        float dv[8];
#pragma unroll
                        for (int h = 0; h < 8; ++h) {
                            f2 d = {0.f, 0.f};
                            pk_dot4(d, v[e], q4[h]);
                            dv[h] = d.x + d.y;
                        }
                        const float d2 = heads_sum_e<3, 3>(dv, lane);
                        const float du = slope * (at * d2 - alpha * S);
                        dad += du;
#pragma unroll
                        for (int h = 0; h < 8; ++h) pk_fma4(ys[h], bcast_head<3>(du, h), v[e]);
                    }
            }
        }
        meta_cols(N1);                                            // in flight across the next tile's products
        {   // the next tile's grad_out columns and first Z tile: the edge pass has let go of its registers
            const int tn = tile + gridDim.x;
            request_g(tn, C, vg);
            const int rows = tn < P.ntiles ? min(32, P.N - tn * 32) : 0;
            load_zb(__builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(A.zin) + (size_t)(rows > 0 ? tn * 32 : 0) * HF,
                                                      0, rows * HF * 4, 0x00020000), 0, zb);
        }
        {   // d_wd[h][f] += sum over the wave's 8 rows of (the row's sum of d_u)[h] * x_i[f]: two
            // 16x16x4 steps per 16 columns, A[m = head][k = row] through the wave's LDS corner
            dtw[g * 8 + hl_raw] = dad;                            // lane (g, p) owns head bitrev(p) of row g
            __builtin_amdgcn_wave_barrier();
            const int l16 = lane & 15, k4 = lane >> 4;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (P.dbg & 16) break;
                const int rl = 4 * s + k4;
                const float av = l16 < 8 ? dtw[rl * 8 + l16] : 0.f;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    if (16 * nt < A.Fp)                           // wave-uniform
                        ydm[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xb[s][nt], ydm[nt], 0, 0, 0);
            }
        }
        lds_barrier();                                            // before the next tile's q stores
    }
    const int C = P.C;
    // the block's partial sums: d_ws | d_wd rows (8 groups, then 4 waves through LDS), its d_W slab, g sums
    float *red = fuse_lds;                                        // [2][4][256], over the q tile
    if constexpr (EDGE) {
#pragma unroll
    for (int h = 0; h < 8; ++h) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ys[h][k] = high_lanes_reduce<1, 3>(ys[h][k]);
        if (g == 0) stv4(red + (0 * 4 + wv) * 256 + h * 32 + p * 4, ys[h]);
    }
    if (lane < 32) {                                              // D[m = 4 (lane / 16) + r][n = lane % 16]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[(1 * 4 + wv) * 256 + (4 * (lane >> 4) + r) * 32 + 16 * nt + (lane & 15)] = ydm[nt][r];
    }
    __syncthreads();
    float *dst = A.part_y + (size_t)blockIdx.x * 2 * HF;
    for (int idx = threadIdx.x; idx < 2 * 256; idx += 256) {
        const int which = idx >> 8, r = idx & 255, hh = r >> 5, ff = r & 31;
        if (hh < A.H && ff < A.Fp) {
            const float *src = red + which * 4 * 256 + r;
            dst[which * HF + hh * A.Fp + ff] = (src[0] + src[256]) + (src[512] + src[768]);
        }
    }
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
        if (hh < nh) {
            const int h = h0 + hh;
            float *sl = P.slab + ((size_t)blockIdx.x * A.H + h) * C * P.F;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (c < C && l < P.F) sl[(size_t)c * P.F + l] = dw[hh][ct][r];
                }
                if (P.gsum) {
                    const float tot = gsa[hh][ct] + __shfl_xor(gsa[hh][ct], 32, 64);
                    const int c = ct * 32 + l;
                    if (half == 0 && c < C) P.gsum[(size_t)blockIdx.x * A.H * C + h * C + c] = tot;
                }
            }
        }
}

// ---------------------------------------------------------------------------------
// Host side
// ---------------------------------------------------------------------------------
struct InGeo {
    int Fp, Fs, lph, H2, LH, lh_log, G, T, HF;
};

static bool make_in_geo(const gatmi_shape &s, InGeo *g) {
    if (s.heads < 1 || s.heads > 32 || s.channels < 1 || s.in_features < 1) return false;
    g->Fp = round_up4(s.in_features);
    g->Fs = g->Fp + round_up4(s.heads);
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

static bool bwd2_on() {   // A/B switch: GATMI_BWD2=0 runs the kernel with the chain in the row loop
    return option("GATMI_BWD2", 1) != 0;
}

template <int LH_T, int G, int T>
static void in_launch(InKind k, const InArgs &A, int grid, hipStream_t st) {
    if (k == IN_FWD)
        hipLaunchKernelGGL((inagg_fwd_kernel<LH_T, G, T>), dim3(grid), dim3(256), 0, st, A);
    else if (bwd2_on() && next_pow2(A.H) <= kBwd2MaxH2)
        hipLaunchKernelGGL((inagg_bwd2_kernel<LH_T, G, T>), dim3(grid), dim3(256), 0, st, A, next_pow2(A.H));
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

// the two-edge backward writes one partial row per BLOCK, the others one per wave
// The multi-edge layout for 5..8 heads and rows of 9..32 float4 (4 or 2 edges per instruction;
// 2.8 against 8.3 ms on 128 -> 8x32 at degree 25).  Rows of at most 8 float4 stay with the
// one-edge kernels: a lane holds one float4 there too, and at run.py's degree 5 the row loop is
// a small part of an item (0.56 against 0.50 ms on runpy100k; 1.43 against 1.51 at degree 25).
// GATMI_EDGE8=0: one-edge kernels everywhere; 2: multi-edge also for short rows and, in place
// of the two-edge kernels, for H <= 4 (parity tests, A/B runs).
static int edge8_lrl(const InGeo &g, int H) {
    const int mode = option("GATMI_EDGE8", 1);
    if (mode == 0 || g.G != 64 || g.lph > 32) return 0;
    const int lrl = ilog2(next_pow2(g.lph));
    if (g.H2 == 8) return lrl > 3 ? lrl : (mode == 2 ? 3 : 0);
    if (mode == 2 && g.H2 == 4 && g.lph > 16) return 5;
    return 0;
}
static bool bwd3_covers(const InGeo &g, int H) {
    if (!bwd2_on()) return false;
    if (edge8_lrl(g, H)) return true;
    return g.G == 64 && H <= 4 && g.lph <= 32 && option("GATMI_BWD3", 1) != 0;
}
// A destination row per 8-lane group: 5..8 heads, rows of at most 8 float4, items of few edges
// (GATMI_GROUPROWS=0 off, 2 whatever the degree).
static bool group_rows(const InGeo &g, const InArgs &A) {
    const int mode = option("GATMI_GROUPROWS", 1);
    if (mode == 0 || g.G != 64 || g.H2 != 8 || g.lph > 8 || !bwd2_on()) return false;
    if (mode == 2) return true;
    return A.n_items > 0 && (long long)A.E <= 12ll * A.n_items;
}
// these kernels take 32 items per block and pass
static int group_rows_grid(const InArgs &A, int grid) {
    const int need = item_grid(A.n_items, 32);
    return need < grid ? need : grid;
}
template <int LRL, int HPL, int UF, int UB>
static void edge8_launch(InKind k, const InArgs &A, int grid, hipStream_t st) {
    if (k == IN_FWD)
        hipLaunchKernelGGL((inagg_fwde_kernel<LRL, HPL, UF>), dim3(grid), dim3(256), 0, st, A);
    else
        hipLaunchKernelGGL((inagg_bwde_kernel<LRL, HPL, UB>), dim3(grid), dim3(256), 0, st, A);
}

// The fused tiles (inagg_fwdgf / inagg_bwdgf): where the row-per-group kernels run, every row is
// one work item in row order (no row cut into segments) and a head's channels fit two 32-wide
// MFMA tiles.  GATMI_FUSED=0: the separate kernels (A/B runs, the tests of the fallbacks); 1 / 2:
// only the forward / the backward fused.
constexpr int kFuseMaxC = 64;
constexpr int kFuseBwdMaxRows = 65536;
static bool fused_covers(const InGeo &g, const InArgs &A, const gatmi_graph *gr, int C, bool bf16_rows, int pass) {
    // GATMI_FUSED: bit 0 the forward, bit 1 the backward (they do not depend on each other), bit 2:
    // the backward whatever the size.  By default the backward is fused up to kFuseBwdMaxRows rows:
    // beyond, with every CU busy for long, the separate kernels' higher occupancy wins (runpy1m: 1.64
    // against 1.52 ms; runpy100k: 0.19 against 0.24) -- and so does the products-only pass
    // with the edge kernel it spares the Z rows, once that kernel fetched its next item ahead (100 k rows:
    // kernels 0.199 against 0.207 ms, a step replayed as a HIP graph 0.296 against 0.303, eager 0.334
    // against 0.330; 30 k rows: kernels 0.088 against 0.077)
    const int mode = option("GATMI_FUSED", 11);
    if ((mode & pass) == 0 || bf16_rows || C > kFuseMaxC) return false;
    if (pass == 2 && !(mode & 4) && gr->num_dst > kFuseBwdMaxRows) return false;
    // the backward holds a head pair's d_W in registers beside the edge pass: one 32-channel tile
    // per head fits (C = 33..64 spills: runpy100k at 8 x 40 0.41 against 0.27 ms); C even: rows of
    // grad_out are staged as float2 / float4
    if (pass == 2 && (C > 32 || C % 2 != 0)) return false;
    if (gr->plan.n_split_rows != 0 || gr->plan.n_items != gr->num_dst) return false;
    return group_rows(g, A);
}
// q = g W and d_W = g^T Z (+ column sums of g) from one pass over grad_out (inagg_bwdgf_kernel<.., false>):
// any graph -- the kernel never looks at it --, up to 8 heads, rows of at most 8 float4, C <= 40 (two
// 32-channel tiles per head from 33 on; up to 40 two blocks' LDS tiles still fit a CU; odd C: grad_out
// staged float by float -- without the edge pass in the kernel that costs no registers).
// GATMI_FUSED bit 3.
static bool fused_products_cover(const InGeo &g, int H, int C) {
    if ((option("GATMI_FUSED", 11) & 8) == 0) return false;
    return g.G == 64 && H <= 8 && g.lph <= 8 && C <= 40;
}
struct FusePlan {
    int grid, grid_fwd, ct, vw;
    size_t lds_fwd, lds_bwd;
    FuseArgs P;
};
static FusePlan fuse_plan(const InGeo &g, int N, int H, int C, int F) {
    FusePlan f{};
    f.P.C = C;
    f.P.F = F;
    f.P.N = N;
    f.P.ntiles = (N + 31) / 32;
#ifdef GATMI_ABLATION                        // timing-only switches that skip phases of the fused backward (results
    f.P.dbg = option("GATMI_FUSE_DBG", 0);   // are wrong): only in builds made for tools/fused_dbg.sh
#else
    f.P.dbg = 0;
#endif
    f.P.zs = g.HF + 2;                       // forward (the backward sets its own): 2 mod 4
    f.P.cp4 = round_up4(C);
    f.P.gs = 2 * f.P.cp4 + 2;                // 2 mod 4
    f.ct = (C + 31) / 32;
    f.vw = C % 4 == 0 ? 4 : (C % 2 == 0 ? 2 : 1);
    f.lds_fwd = (size_t)32 * (f.P.zs > H * C ? f.P.zs : H * C) * 4;   // the Z tile, then the out tile over it
    f.lds_bwd = ((size_t)32 * (g.HF + 4) + (size_t)4 * 32 * f.P.gs + 4 * 64 + 32 * 8) * 4;   // q tile | g tiles | d_u sums | S tile
    if (f.lds_bwd < 2 * 4 * 256 * 4) f.lds_bwd = 2 * 4 * 256 * 4;   // the closing sums of the block
    // two blocks per CU are resident in the backward (three in the forward); a block walks its
    // tiles with barriers, so blocks beyond the resident ones only add tails
    const int want = device_cus() * 2;
    int grid = f.P.ntiles < want ? f.P.ntiles : want;
    if (grid < 8) grid = 8;
    f.grid = (grid + 7) / 8 * 8;
    const int want3 = device_cus() * (f.ct == 1 ? GATMI_FWDGF_OCC : 3);
    grid = f.P.ntiles < want3 ? f.P.ntiles : want3;
    if (grid < 8) grid = 8;
    f.grid_fwd = (grid + 7) / 8 * 8;
    return f;
}
template <class K> static int grant_lds(K kern, size_t lds, size_t *granted) {
    if (*granted < lds) {
        GATMI_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        *granted = lds;
    }
    return GATMI_OK;
}
static int fused_fwd_launch(const FusePlan &f, const InArgs &A, float *wimg, hipStream_t st) {
    const int dev = device_index();
    static size_t granted[2][kMaxDevices] = {};
    FuseArgs P = f.P;
    {
        const int tot = A.H * A.Fp * f.ct * 32;
        hipLaunchKernelGGL(inagg_wimg_kernel, dim3((tot + 255) / 256), dim3(256), 0, st, P.w, A.H, P.C, P.F, A.Fp,
                           f.ct * 32, wimg);
        GATMI_HIP(hipGetLastError());
        P.wimg = wimg;
    }
    if (f.ct == 1) {
        if (int e = grant_lds(inagg_fwdgf_kernel<1>, f.lds_fwd, &granted[0][dev])) return e;
        hipLaunchKernelGGL(inagg_fwdgf_kernel<1>, dim3(f.grid_fwd), dim3(256), f.lds_fwd, st, A, P);
    } else {
        if (int e = grant_lds(inagg_fwdgf_kernel<2>, f.lds_fwd, &granted[1][dev])) return e;
        hipLaunchKernelGGL(inagg_fwdgf_kernel<2>, dim3(f.grid_fwd), dim3(256), f.lds_fwd, st, A, P);
    }
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}
template <int CT, int VW, bool EDGE>
static int fused_bwd_launch_t(const FusePlan &f, const InArgs &A, const FuseArgs &P, hipStream_t st) {
    static size_t granted[kMaxDevices] = {};
    if (int e = grant_lds(inagg_bwdgf_kernel<CT, VW, EDGE>, f.lds_bwd, &granted[device_index()])) return e;
    hipLaunchKernelGGL((inagg_bwdgf_kernel<CT, VW, EDGE>), dim3(f.grid), dim3(256), f.lds_bwd, st, A, P);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}
static int fused_bwd_launch(const FusePlan &f, const InGeo &g, const InArgs &A, hipStream_t st) {
    FuseArgs P = f.P;
    P.zs = g.HF + 4;                         // the q tile is read back as float4
    const int lpr = next_pow2(P.C / f.vw);   // lanes per row of one head's columns of grad_out
    P.lpr_log = ilog2(lpr < 2 ? 2 : lpr);
    if (P.qout && f.ct == 2)
        return f.vw == 4 ? fused_bwd_launch_t<2, 4, false>(f, A, P, st)
                         : (f.vw == 2 ? fused_bwd_launch_t<2, 2, false>(f, A, P, st) : fused_bwd_launch_t<2, 1, false>(f, A, P, st));
    if (P.qout)
        return f.vw == 4 ? fused_bwd_launch_t<1, 4, false>(f, A, P, st)
                         : (f.vw == 2 ? fused_bwd_launch_t<1, 2, false>(f, A, P, st) : fused_bwd_launch_t<1, 1, false>(f, A, P, st));
    GATMI_REQUIRE(f.ct == 1 && f.vw != 1, "the fused backward takes even C <= 32 (fused_covers), got %d", P.C);
    return f.vw == 4 ? fused_bwd_launch_t<1, 4, true>(f, A, P, st) : fused_bwd_launch_t<1, 2, true>(f, A, P, st);
}

static int in_launch_rows(InKind k, const InGeo &g, const InArgs &A, int grid, hipStream_t st) {
    if (group_rows(g, A)) {
        const int gg = group_rows_grid(A, grid);
        if (k == IN_FWD) hipLaunchKernelGGL(inagg_fwdg_kernel, dim3(gg), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(inagg_bwdg_kernel, dim3(gg), dim3(256), 0, st, A);
        GATMI_HIP(hipGetLastError());
        return GATMI_OK;
    }
    if (const int lrl = edge8_lrl(g, A.H); lrl && (k == IN_FWD || bwd2_on())) {
        if (g.H2 == 4) edge8_launch<5, 2, 4, 2>(k, A, grid, st);
        else if (lrl == 3) edge8_launch<3, 3, 1, 1>(k, A, grid, st);
        else if (lrl == 4) edge8_launch<4, 3, 2, 2>(k, A, grid, st);
        else edge8_launch<5, 3, GATMI_E8_UF, GATMI_E8_UB>(k, A, grid, st);
        GATMI_HIP(hipGetLastError());
        return GATMI_OK;
    }
    // backward, H <= 4 and a row of at most 32 float4: the two-edges-per-instruction layout
    if (k == IN_FWD && g.G == 64 && A.H <= 4 && g.lph <= 32 && option("GATMI_FWD3", 1) != 0) {
        hipLaunchKernelGGL(inagg_fwd3_kernel, dim3(grid), dim3(256), 0, st, A);
        GATMI_HIP(hipGetLastError());
        return GATMI_OK;
    }
    if (k == IN_BWD && bwd3_covers(g, A.H)) {
        hipLaunchKernelGGL(inagg_bwd3_kernel, dim3(grid), dim3(256), 0, st, A);
        GATMI_HIP(hipGetLastError());
        return GATMI_OK;
    }
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

// Streaming (LDS-DMA) kernels: one row per wave (G == 64), a packed row fits one DMA
// instruction (Fs/4 <= 64 lanes), per-chunk logit tables sized for <= kStreamMaxHeads heads.
// They pay when the packed table is far larger than the 256 MiB Infinity Cache (rows come
// from HBM and bytes in flight decide); a cache-resident table is served faster by the
// register pipeline (lower per-edge overhead).  GATMI_STREAM=0/1 forces the choice (A/B runs).
static bool stream_ok(const InGeo &g, int H, long long num_src, int mode) {
    const int forced = option("GATMI_STREAM", -1);
    if (!(g.G == 64 && g.Fs / 4 <= 64 && H <= kStreamMaxHeads)) return false;
    if (mode == 2 || (mode == 0 && forced == 0)) return false;
    if (mode == 1 || (mode == 0 && forced == 1)) return true;
    // 5..8 heads on rows of 9..32 float4: every head group reads the whole row back from the ring
    // (eight LDS reads of a row per edge); the multi-edge register kernel holds a row once and is
    // faster there also from HBM (100 -> 8x32 at products size: 8.8 against 10.7 ms)
    if (edge8_lrl(g, H)) return false;
    return num_src * g.Fs * 4 > (256ll << 20);
}

// bf16 source rows (gatmi_inagg.xs16): rows of at most 112 channels; up to four heads on the
// four-edge kernels (inagg_fwd16 / inagg_bwd16), 5..8 heads on the multi-edge kernels reading 8-byte
// pieces (rows of 9..28 float4: 4 or 2 edges per instruction)
static bool rows16_cover(const InGeo &g, int H) {
    if (g.G != 64 || g.Fp > 112) return false;
    if (H <= 4) return true;
    return g.H2 == 8 && g.lph > 8;
}
static void rows16_launch(InKind k, const InGeo &g, const InArgs &A, int grid, hipStream_t st) {
    if (A.H <= 4) {
        // forward: two instructions of four edges per bank (one: 5.3 ms, three: 5.9 ms at products size)
        if (k == IN_FWD) hipLaunchKernelGGL(inagg_fwd16_kernel<2>, dim3(grid), dim3(256), 0, st, A);
        else hipLaunchKernelGGL(inagg_bwd16_kernel<GATMI_B16_U>, dim3(grid), dim3(256), 0, st, A);
    } else if (g.lph > 16) {
        if (k == IN_FWD) hipLaunchKernelGGL((inagg_fwde_kernel<5, 3, GATMI_E8_UF, true>), dim3(grid), dim3(256), 0, st, A);
        else hipLaunchKernelGGL((inagg_bwde_kernel<5, 3, GATMI_E8_UB, true>), dim3(grid), dim3(256), 0, st, A);
    } else {
        if (k == IN_FWD) hipLaunchKernelGGL((inagg_fwde_kernel<4, 3, 2, true>), dim3(grid), dim3(256), 0, st, A);
        else hipLaunchKernelGGL((inagg_bwde_kernel<4, 3, 2, true>), dim3(grid), dim3(256), 0, st, A);
    }
}

// Persistent grid: as many blocks as fit on the device at once; chunks come from the queue.
template <class K> static int resident_blocks(K kernel) {
    int dev = 0, cus = 0, per = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kernel, 256, 0) != hipSuccess)
        return 1024;
    if (cus < 1) cus = 256;
    if (per < 1) per = 1;
    return cus * per;
}

template <int LH_T, int T>
static void stream_launch(InKind k, const InArgs &A, hipStream_t st) {
    if (k == IN_FWD) {
        static int grids[kMaxDevices] = {};      // per device: occupancy and CU count are the device's
        const int di = device_index();
        if (grids[di] == 0) grids[di] = resident_blocks(inagg_fwd_stream_kernel<LH_T, T>);
        const int grid = grids[di];
        const int need = (A.n_chunks + 3) / 4;
        hipLaunchKernelGGL((inagg_fwd_stream_kernel<LH_T, T>), dim3(need < grid ? need : grid),
                           dim3(256), 0, st, A);
    }
}

static size_t stream_ws_ints(const gatmi_graph *g) {
    return 64 + ((size_t)g->num_edges >> kChunkEdgesLog) + 2;   // queue | chunk_item
}

static int stream_launch_rows(InKind k, const InGeo &g, InArgs &A, int *ws, hipStream_t st) {
    if (A.n_items == 0) return GATMI_OK;
    A.n_chunks = (A.E >> kChunkEdgesLog) + 1;
    A.queue = ws;
    int *table = ws + 64;
    A.chunk_item = table;
    {
        const int blocks = (A.n_items + 256) / 256;
        hipLaunchKernelGGL(inagg_chunk_table_kernel, dim3(blocks > 1024 ? 1024 : blocks), dim3(256),
                           0, st, A.item_beg, A.n_items, A.n_chunks, table, A.queue);
        GATMI_HIP(hipGetLastError());
    }
    if (g.LH == 16 && g.T == 2) stream_launch<16, 2>(k, A, st);
    else if (g.LH == 8 && g.T == 1) stream_launch<8, 1>(k, A, st);
    else if (g.T == 1) stream_launch<0, 1>(k, A, st);
    else if (g.T == 2) stream_launch<0, 2>(k, A, st);
    else stream_launch<0, 4>(k, A, st);
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
    GATMI_REQUIRE(a->mask_mode >= 0 && a->mask_mode <= 3, "bad mask_mode %d", a->mask_mode);
    GATMI_REQUIRE((a->mask_mode != GATMI_MASK_BYTES && a->mask_mode != GATMI_MASK_HASH_DEV) || a->mask,
                  "mask_mode BYTES / HASH_DEV needs `mask` (the keep flags / the device seed)");
    GATMI_REQUIRE(a->dropout_p >= 0.f && a->dropout_p <= 1.f, "dropout_p %f outside [0,1]",
                  a->dropout_p);
    GATMI_REQUIRE(a->stream >= 0 && a->stream <= 2, "bad stream mode %d", a->stream);
    GATMI_REQUIRE(a->xs && a->a_dst && a->weight && a->att_src && a->att_dst,
                  "null xs/a_dst/weight/att_src/att_dst");
    GATMI_REQUIRE(!a->xs16 || rows16_cover(*geo, a->shape.heads),
                  "bf16 source rows cover F <= 112 with H <= 4, or 5..8 heads and F > 32 (got F=%d H=%d)", a->shape.in_features,
                  a->shape.heads);
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
    A.Fs = geo.Fs;
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
    A.xs = a->xs;
    A.xs16 = a->xs16;
    A.a_dst = a->a_dst;
    A.E = g->num_edges;
}

static int dw_splits(int N, int H, int C, int F) {
    const int tiles = ((C + 127) / 128) * ((F + 127) / 128) * H;
    int want = 1024 / tiles;
    const int maxs = (N + 4 * 16 - 1) / (4 * 16);
    if (want > maxs) want = maxs;
    if ((long long)want * H > 65535) want = 65535 / H;
    return want < 1 ? 1 : want;
}


// d_W by the register-streaming kernel when the head's result fits the accumulators
struct DwPlan {
    bool stream;
    int nsplit, rows_per_split, ct, nt;
    bool tail;
};
static DwPlan dw_plan(int N, int H, int C, int F) {
    const bool stream_on = option("GATMI_DW_STREAM", 1) != 0;   // A/B switch: 0 = tiled GEMM
    DwPlan p{false, dw_splits(N, H, C, F), 0, (C + 31) / 32, (F + 31) / 32, false};
    if (!stream_on) return p;
    if (p.ct > 2 || p.nt > 4 || p.ct * p.nt > 8 || N < 4096) return p;
    // A/B switch GATMI_DW_TAIL: 0 = a fourth 32-wide tile
    if (p.ct == 2 && F > 96 && F <= 100 && option("GATMI_DW_TAIL", 1) != 0) {
        p.nt = 3;
        p.tail = true;
    } else if (p.nt == 3) {
        p.nt = 4;
    }
    long long waves = (long long)device_cus() * 8;         // two blocks of four waves per CU
    int ns = (int)(waves / H);
    const int maxs = N / 256;                              // >= 128 row pairs per wave
    if (ns > maxs) ns = maxs;
    if (ns < 1) ns = 1;
    int rps = (N + ns - 1) / ns;
    rps += rps & 1;                                        // even: pairs never straddle splits
    ns = (N + rps - 1) / rps;
    p.stream = true;
    p.nsplit = ns;
    p.rows_per_split = rps;
    return p;
}

template <int CT, int NT, bool TAIL = false>
static void dw_stream_launch(const DwPlan &p, const float *g, const float *z, int N, int H, int C, int F,
                             int Fp, float *slab, float *gsum, hipStream_t st) {
    const int blocks = (H * p.nsplit + 3) / 4;
    hipLaunchKernelGGL((inagg_dw_stream_kernel<CT, NT, TAIL>), dim3(blocks), dim3(256), 0, st, g, H * C, z,
                       H * Fp, N, H, C, F, Fp, p.nsplit, p.rows_per_split, slab, gsum);
}
static void dw_stream(const DwPlan &p, const float *g, const float *z, int N, int H, int C, int F, int Fp,
                      float *slab, float *gsum, hipStream_t st) {
    if (p.tail) dw_stream_launch<2, 3, true>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
    else if (p.ct == 1 && p.nt == 1) dw_stream_launch<1, 1>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
    else if (p.ct == 1 && p.nt == 2) dw_stream_launch<1, 2>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
    else if (p.ct == 1) dw_stream_launch<1, 4>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
    else if (p.nt == 1) dw_stream_launch<2, 1>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
    else if (p.nt == 2) dw_stream_launch<2, 2>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
    else dw_stream_launch<2, 4>(p, g, z, N, H, C, F, Fp, slab, gsum, st);
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

extern "C" size_t gatmi_inagg_xs_floats(const gatmi_shape *s) {
    InGeo geo;
    if (!s || !make_in_geo(*s, &geo)) return 0;
    return (size_t)geo.Fs;
}

extern "C" size_t gatmi_inagg_pack_workspace_bytes(const gatmi_shape *s) {
    InGeo geo;
    if (!s || !make_in_geo(*s, &geo)) return 0;
    return align_up((size_t)2 * geo.HF * 4) + 256;
}

static int inagg_pack_rows(const gatmi_shape *s, const float *x, const float *weight,
                           const float *att_src, const float *att_dst, float *xs, float *a_dst,
                           void *xs16, void *ws, size_t ws_bytes, void *stream) {
    GATMI_REQUIRE(s && x && weight && att_src && att_dst && xs && a_dst, "null argument");
    InGeo geo;
    GATMI_REQUIRE(make_in_geo(*s, &geo), "input-side aggregation does not cover F=%d H=%d",
                  s->in_features, s->heads);
    GATMI_REQUIRE(s->num_nodes >= 0, "negative row count");
    if (ws_bytes < gatmi_inagg_pack_workspace_bytes(s)) {
        set_error("inagg_pack workspace too small: %zu < %zu", ws_bytes,
                  gatmi_inagg_pack_workspace_bytes(s));
        return GATMI_ERR_WS;
    }
    if (s->num_nodes == 0) return GATMI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *wsd = static_cast<float *>(ws);
    const int H = s->heads, C = s->channels, F = s->in_features;
    ProfScope ps(xs16 ? "inagg_pack16" : "inagg_pack", st);
    hipLaunchKernelGGL(inagg_wsd_kernel, dim3((2 * geo.HF + 255) / 256), dim3(256), 0, st, weight,
                       att_src, att_dst, H, C, F, geo.Fp, wsd);
    GATMI_HIP(hipGetLastError());
    int gl = next_pow2(geo.lph);
    if (gl > 64) gl = 64;
    const int ks = (geo.lph + gl - 1) / gl;   // <= 4: lph <= 256 whenever make_in_geo accepts
    GATMI_REQUIRE(ks <= 4, "in_features %d too wide to pack", F);
    const long long rpw = 64 / gl;
    long long blocks = (s->num_nodes + rpw * 4 - 1) / (rpw * 4);
    if (blocks > 2048) blocks = 2048;
    const int glog = ilog2(gl);
    const long long n = s->num_nodes;
    const size_t lds = (size_t)2 * geo.HF * sizeof(float);
#define GATMI_PACK(KS, GL)                                                                       \
    hipLaunchKernelGGL((inagg_pack_kernel<KS, GL>), dim3((unsigned)blocks), dim3(256), lds, st, x, wsd, n, \
                       F, geo.Fp, geo.Fs, H, glog, xs, a_dst, static_cast<uint2 *>(xs16))
    GATMI_REQUIRE(!xs16 || (ks == 1 && rows16_cover(geo, H)),
                  "bf16 source rows cover F <= 112 with H <= 4, or 5..8 heads and F > 32 (got F=%d H=%d)", F, H);
    if (ks == 1 && gl == 32) GATMI_PACK(1, 32);         // F = 65 .. 128: the layers BASELINE.json names
    else if (ks == 1 && gl == 8) GATMI_PACK(1, 8);      // run.py's 25 features
    else if (ks == 1) GATMI_PACK(1, 0);
    else if (ks == 2) GATMI_PACK(2, 64);
    else GATMI_PACK(4, 64);
#undef GATMI_PACK
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" int gatmi_inagg_pack(const gatmi_shape *s, const float *x, const float *weight,
                                const float *att_src, const float *att_dst, float *xs, float *a_dst,
                                void *ws, size_t ws_bytes, void *stream) {
    return inagg_pack_rows(s, x, weight, att_src, att_dst, xs, a_dst, nullptr, ws, ws_bytes, stream);
}

extern "C" int gatmi_inagg_pack16(const gatmi_shape *s, const float *x, const float *weight,
                                  const float *att_src, const float *att_dst, float *xs, float *a_dst,
                                  void *xs16, void *ws, size_t ws_bytes, void *stream) {
    GATMI_REQUIRE(xs16, "null xs16");
    return inagg_pack_rows(s, x, weight, att_src, att_dst, xs, a_dst, xs16, ws, ws_bytes, stream);
}

extern "C" size_t gatmi_inagg_xs16_bytes(const gatmi_shape *s) {
    InGeo g;
    if (!s || !make_in_geo(*s, &g) || !rows16_cover(g, s->heads)) return 0;
    return kRow16;
}

extern "C" int gatmi_inagg_table16(const gatmi_shape *s, const float *xs, void *xs16, void *stream) {
    GATMI_REQUIRE(s && xs && xs16, "null shape/xs/xs16");
    InGeo g;
    GATMI_REQUIRE(make_in_geo(*s, &g) && rows16_cover(g, s->heads),
                  "bf16 source rows cover F <= 112 with H <= 4, or 5..8 heads and F > 32 (got F=%d H=%d)", s->in_features, s->heads);
    if (s->num_nodes == 0) return GATMI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProfScope ps("inagg_table16", st);
    const long long pieces = (long long)s->num_nodes * 16;
    const int blocks = (int)((pieces + 255) / 256 > 16384 ? 16384 : (pieces + 255) / 256);
    hipLaunchKernelGGL(inagg_table16_kernel, dim3(blocks), dim3(256), 0, st, xs, (long long)s->num_nodes,
                       g.Fp, g.Fs, s->heads, static_cast<uint4 *>(xs16));
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" size_t gatmi_inagg_fwd_workspace_bytes(const gatmi_graph *g, const gatmi_shape *s) {
    InGeo geo;
    if (!g || !s || !make_in_geo(*s, &geo)) return 0;
    Carver cv(nullptr);
    const size_t slots = (size_t)g->plan.n_slots;
    cv.take<float>(slots * s->heads);
    cv.take<float>(slots * s->heads);
    cv.take<float>(slots * geo.HF);
    cv.take<int>(stream_ws_ints(g));   // chunk queue + table of the streaming kernel
    cv.take<float>((size_t)s->heads * s->channels * 4 * geo.Fp);   // block-diagonal weight of head groups
    cv.take<float>((size_t)s->heads * geo.Fp * 64);                 // W^T image of the fused forward
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
    InArgs A;
    fill_in_args(A, g, a, geo);
    const size_t slots = (size_t)g->plan.n_slots;
    A.part_m = cv.take<float>(slots * H);
    A.part_s = cv.take<float>(slots * H);
    A.part_acc = cv.take<float>(slots * geo.HF);
    int *stream_ws = cv.take<int>(stream_ws_ints(g));
    float *bdiag = cv.take<float>((size_t)H * C * 4 * geo.Fp);
    float *wimg = cv.take<float>((size_t)H * geo.Fp * 64);    // W^T image of the fused forward
    A.z = z;
    A.lse = lse;
    if (fused_covers(geo, A, g, C, a->xs16 != nullptr, 1)) {
        // edge pass and out = Z W^T + bias of a 32-row tile in one kernel
        ProfScope ps("inagg_fwd_fused", st);
        FusePlan f = fuse_plan(geo, N, H, C, F);
        f.P.w = a->weight;
        f.P.bias = a->bias;
        f.P.out = out;
        return fused_fwd_launch(f, A, wimg, st);
    }
    {
        ProfScope ps("inagg_fwd", st);
        // table beyond the Infinity Cache: the LDS-DMA streaming kernel (6.67 ms against 6.93 ms
        // for the two-edge register layout on the products-shaped layer); else in_launch_rows
        // picks the two-edge layout when it covers the shape (H <= 4), the head-group kernel if not
        if (a->xs16) {
            if (A.n_items > 0) {
                rows16_launch(IN_FWD, geo, A, in_grid(geo, A.n_items), st);
                GATMI_HIP(hipGetLastError());
            }
        } else if (stream_ok(geo, H, g->num_src, a->stream)) {
            if (int e = stream_launch_rows(IN_FWD, geo, A, stream_ws, st)) return e;
        } else if (int e = in_launch_rows(IN_FWD, geo, A, in_grid(geo, A.n_items), st)) {
            return e;
        }
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
        d.a_kpad = geo.Fp;   // the padded columns of Z are zero
        // tiny heads (run.py's 28 x 30): a 32-row tile is 14 MFMAs under a tile's fixed waits and
        // stores; pairs of heads against the block-diagonal weight (the image inagg_bdiag_kernel
        // builds reads as n-major [G C][G Fp]) halve the tiles: 0.57 -> 0.47 ms at 10^6 nodes, but
        // 0.046 -> 0.061 ms at 10^5 (one more launch, half the blocks), hence the size test.
        // GATMI_OUTGROUP=0: per head; 2: pairs whatever the size
        GemmDesc dg = d;
        const int G = 2;
        dg.sA = (long long)G * geo.Fp;
        dg.B = bdiag; dg.ldb = G * geo.Fp; dg.sB = (long long)G * C * G * geo.Fp;
        dg.sC = (long long)G * C; dg.sBias = (long long)G * C;
        dg.Nn = G * C; dg.K = G * geo.Fp; dg.a_kpad = G * geo.Fp; dg.n_store = G * C; dg.nbatch = H / G;
        dg.even_ok = true;
        const int og = option("GATMI_OUTGROUP", 1);
        if (geo.Fp <= 32 && C <= 32 && H % G == 0 && (og == 2 || (og == 1 && N >= 300000)) && kres_covers(dg)) {
            const long long tot = (long long)H * C * G * geo.Fp;
            hipLaunchKernelGGL(inagg_bdiag_kernel, dim3((unsigned)((tot + 255) / 256 > 256 ? 256 : (tot + 255) / 256)),
                               dim3(256), 0, st, a->weight, H, C, F, geo.Fp, G, bdiag);
            GATMI_HIP(hipGetLastError());
            if (int e = launch_kres(dg, st)) return e;
        } else if (int e = kres_covers(d) ? launch_kres(d, st) : launch_gemm(d, st)) {
            return e;
        }
    }
    return GATMI_OK;
}

namespace {
struct InBwdWs {
    float *q, *part_y, *dwsd, *slab, *gsum, *bdiag, *slab_sum, *sdot;
    int waves, nsplit;
    size_t bytes;
};
InBwdWs carve_in_bwd(const gatmi_graph *g, const gatmi_shape *s, const InGeo &geo, void *ws) {
    Carver cv(ws);
    InBwdWs w;
    const size_t N = (size_t)g->num_dst;
    w.q = cv.take<float>(N * geo.HF);
    w.waves = in_grid(geo, g->plan.n_items) * 4;
    w.part_y = cv.take<float>((size_t)w.waves * 2 * geo.HF);
    w.dwsd = cv.take<float>((size_t)2 * geo.HF);
    w.nsplit = dw_plan(g->num_dst, s->heads, s->channels, s->in_features).nsplit;
    int slabs = w.nsplit;
    if (s->channels <= kFuseMaxC) {        // the fused backward writes one d_W slab per block
        const int fg = fuse_plan(geo, g->num_dst, s->heads, s->channels, s->in_features).grid;
        if (fg > slabs) slabs = fg;
    }
    w.slab = cv.take<float>((size_t)slabs * s->heads * s->channels * s->in_features);
    w.gsum = cv.take<float>((size_t)slabs * s->heads * s->channels);
    w.bdiag = cv.take<float>((size_t)s->heads * s->channels * 4 * geo.Fp);   // head groups of <= 4
    w.slab_sum = cv.take<float>((size_t)s->heads * s->channels * (s->in_features + 1));   // fused backward: summed slabs | g sums
    w.sdot = cv.take<float>(N * s->heads);   // S = <q, Z> per (row, head) from the products pass
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
    if (N > 0 && fused_covers(geo, A, g, C, a->xs16 != nullptr, 2)) {
        // q = g W, the edge pass, d_W = g^T Z and the column sums of g per 32-row tile in one kernel
        FusePlan f = fuse_plan(geo, N, H, C, F);
        A.zin = z;
        A.lsein = lse;
        A.part_y = w.part_y;
        f.P.w = a->weight;
        f.P.gout = grad_out;
        f.P.slab = w.slab;
        f.P.gsum = d_bias ? w.gsum : nullptr;
        {
            ProfScope ps("inagg_bwd_fused", st);
            if (int e = fused_bwd_launch(f, geo, A, st)) return e;
        }
        ProfScope ps("inagg_colsums", st);
        // the blocks' partial rows, slabs and g sums are added by one launch of the row-sum kernel (a
        // block per 16 columns: 375 blocks for run.py's d_W) into row 0 of each; the closing kernel then
        // reads single rows (walking 512 slabs itself it took 60 us)
        const int tot = H * C * F;
        SumJob j0{w.part_y, w.dwsd, f.grid, 2 * geo.HF, geo.HF};
        SumJob j1{w.part_y + geo.HF, w.dwsd + geo.HF, f.grid, 2 * geo.HF, geo.HF};
        SumJob j2{w.slab, w.slab_sum, f.grid, tot, tot};
        SumJob j3{w.gsum, d_bias ? w.slab_sum + tot : nullptr, f.grid, H * C, H * C};
        hipLaunchKernelGGL(inagg_sum_rows_kernel, dim3((max(tot, geo.HF) + 15) / 16, 4), dim3(256), 0, st, j0, j1, j2, j3);   // the widest job: H C F, or H Fp when C F < Fp
        GATMI_HIP(hipGetLastError());
        const int blocks = (int)((tot + 63) / 64 > 2048 ? 2048 : (tot + 63) / 64);
        hipLaunchKernelGGL(inagg_finalize_kernel, dim3(blocks), dim3(256), 0, st, w.slab_sum, 1, H, C, F,
                           geo.Fp, a->weight, a->att_src, a->att_dst, w.dwsd, w.slab_sum + tot, d_weight, d_att_src,
                           d_att_dst, d_bias);
        GATMI_HIP(hipGetLastError());
        return GATMI_OK;
    }
    const bool k1 = N > 0 && fused_products_cover(geo, H, C);
    FusePlan fk{};
    if (k1) {
        // q = g W and the d_W slabs from one pass over grad_out; the edge pass below reads q from HBM
        ProfScope ps("inagg_q_dw_fused", st);
        fk = fuse_plan(geo, N, H, C, F);
        InArgs Ak = A;
        Ak.zin = z;
        fk.P.w = a->weight;
        fk.P.gout = grad_out;
        fk.P.slab = w.slab;
        fk.P.gsum = d_bias ? w.gsum : nullptr;
        fk.P.qout = w.q;
        fk.P.sout = w.sdot;
        if (int e = fused_bwd_launch(fk, geo, Ak, st)) return e;
    } else if (N > 0) {
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
        // heads of 8, 16, 24, 32 ... channels give A rows of an even number of float4: 2-way
        // conflicted LDS reads on the K-resident kernels, against the tiled kernel at a third of their
        // rate (8 x 32 on 2.45 M rows: 4.7 ms tiled)
        d.even_ok = true;
        // C % 4 != 0 (run.py's 8 x 30): per-head rows of grad_out are unaligned for the
        // K-resident kernels; groups of G heads are aligned (GATMI_QGROUP=0: per head, tiled)
        const int G = C % 4 == 0 ? 1 : (C % 2 == 0 ? 2 : 4);
        GemmDesc dg = d;
        dg.sA = (long long)G * C;
        dg.B = w.bdiag; dg.ldb = G * geo.Fp; dg.sB = (long long)G * C * G * geo.Fp;
        dg.sC = (long long)G * geo.Fp;
        dg.Nn = G * geo.Fp; dg.K = G * C; dg.n_store = G * geo.Fp; dg.nbatch = H / G;
        if (G > 1 && H % G == 0 && option("GATMI_QGROUP", 1) != 0 && kres_covers(dg)) {
            const long long tot = (long long)H * C * G * geo.Fp;
            hipLaunchKernelGGL(inagg_bdiag_kernel, dim3((unsigned)((tot + 255) / 256 > 256 ? 256 : (tot + 255) / 256)),
                               dim3(256), 0, st, a->weight, H, C, F, geo.Fp, G, w.bdiag);
            GATMI_HIP(hipGetLastError());
            if (int e = launch_kres(dg, st)) return e;
        } else if (int e = kres_covers(d) ? launch_kres(d, st) : launch_gemm(d, st)) {
            return e;
        }
    }
    A.q = w.q;
    A.zin = z;
    A.lsein = lse;
    A.part_y = w.part_y;
    if (k1 && !a->xs16 && group_rows(geo, A)) A.sin = w.sdot;   // inagg_bwdg_kernel then leaves the Z rows alone
    {
        ProfScope ps("inagg_bwd_edge", st);
        // every wave of the grid writes its partial row, also when it had no work item
        const int grid = w.waves / 4;
        if (A.n_items == 0) {
            GATMI_HIP(hipMemsetAsync(w.part_y, 0, (size_t)w.waves * 2 * geo.HF * 4, st));
        } else if (a->xs16) {
            rows16_launch(IN_BWD, geo, A, grid, st);
            GATMI_HIP(hipGetLastError());
        } else if (int e = in_launch_rows(IN_BWD, geo, A, grid, st)) {
            return e;
        }
    }
    {
        ProfScope ps("inagg_colsums", st);
        int prow = w.waves;                                       // partial rows written
        if (A.n_items > 0 && a->xs16) prow = w.waves / 4;           // one partial row per block
        else if (A.n_items > 0 && group_rows(geo, A)) prow = group_rows_grid(A, w.waves / 4);
        else if (A.n_items > 0 && bwd3_covers(geo, H)) prow = w.waves / 4;
        SumJob j0{w.part_y, w.dwsd, prow, 2 * geo.HF, geo.HF};
        SumJob j1{w.part_y + geo.HF, w.dwsd + geo.HF, prow, 2 * geo.HF, geo.HF};
        if (k1) {
            // with the slabs and g sums of the products pass: ONE row-sum launch (as behind the fused backward),
            // then the closing kernel on single rows
            const int tot = H * C * F;
            SumJob j2{w.slab, w.slab_sum, fk.grid, tot, tot};
            SumJob j3{w.gsum, d_bias ? w.slab_sum + tot : nullptr, fk.grid, H * C, H * C};
            hipLaunchKernelGGL(inagg_sum_rows_kernel, dim3((max(tot, geo.HF) + 15) / 16, 4), dim3(256), 0, st, j0, j1, j2, j3);   // the widest job: H C F, or H Fp when C F < Fp
            GATMI_HIP(hipGetLastError());
            const int blocks = (int)((tot + 63) / 64 > 2048 ? 2048 : (tot + 63) / 64);
            hipLaunchKernelGGL(inagg_finalize_kernel, dim3(blocks), dim3(256), 0, st, w.slab_sum, 1, H, C, F, geo.Fp,
                               a->weight, a->att_src, a->att_dst, w.dwsd, w.slab_sum + tot, d_weight, d_att_src,
                               d_att_dst, d_bias);
            GATMI_HIP(hipGetLastError());
            return GATMI_OK;
        }
        hipLaunchKernelGGL(inagg_sum_rows_kernel, dim3((geo.HF + 15) / 16, 2), dim3(256), 0, st, j0,
                           j1, j1);
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
            d.asum = d_bias ? w.gsum : nullptr;   // [h][split][C]: d_bias rides on the staging of g
            const DwPlan dp = dw_plan(N, H, C, F);
            if (dp.stream) {
                dw_stream(dp, grad_out, z, N, H, C, F, geo.Fp, w.slab, d_bias ? w.gsum : nullptr, st);
                GATMI_HIP(hipGetLastError());
            } else if (int e = launch_gemm(d, st)) {
                return e;
            }
        }
        const long long tot = (long long)H * C * F;
        const int blocks = (int)((tot + 63) / 64 > 2048 ? 2048 : (tot + 63) / 64);
        hipLaunchKernelGGL(inagg_finalize_kernel, dim3(blocks), dim3(256), 0, st, w.slab,
                           N > 0 ? w.nsplit : 0, H, C, F, geo.Fp, a->weight, a->att_src, a->att_dst,
                           w.dwsd, w.gsum, d_weight, d_att_src, d_att_dst, d_bias);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

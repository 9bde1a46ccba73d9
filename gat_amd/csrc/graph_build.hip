// Graph layout on the device: edge_index -> destination-sorted CSR with self-loops, its
// transpose, and the segment plans of both.  Stands for remove_self_loops +
// add_self_loops inside GATConv.forward (call sites
// /root/reference/gat/model.py:40,43,46) applied to the edge_index of
// /root/reference/gat/converter.py:12-15.  The reference redoes this step on every
// forward call; here it runs once per graph and is bit exact: the order of a row's
// in-edges is their order in the normalised edge list (stable sort by destination).
//
// Stable ordering comes from rocPRIM's LSD radix sort over (key = destination,
// value = position in the un-normalised list ++ appended loops); self-loops of the input
// get the sentinel key N and fall off the end.
#include <string.h>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "common.h"

namespace gatmi {

__global__ void gb_make_keys(const int64_t *ei, long long E, long long NS, long long N, int asl,
                             uint32_t *keys, uint32_t *vals, uint32_t *selfflag, int *err) {
    // NS = source rows, N = destination rows (equal unless the graph is a partition)
    const long long M = E + (asl ? N : 0);
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < M;
         e += (long long)gridDim.x * blockDim.x) {
        if (e < E) {
            const long long s = ei[e], d = ei[E + e];
            const bool bad = s < 0 || s >= NS || d < 0 || d >= N;
            if (bad) atomicOr(err, 1);
            const bool self = asl && s == d;
            keys[e] = bad ? (uint32_t)N : (self ? (uint32_t)N : (uint32_t)d);
            selfflag[e] = (self || bad) ? 1u : 0u;
        } else {
            keys[e] = (uint32_t)(e - E);
        }
        vals[e] = (uint32_t)e;
    }
}

// counts[0] = number of dropped input edges (self-loops)
__global__ void gb_count_self(const uint32_t *selfflag, const uint32_t *selfrank, long long E,
                              int *counts) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        counts[0] = E > 0 ? (int)(selfrank[E - 1] + selfflag[E - 1]) : 0;
}

__global__ void gb_fill_csr(const int64_t *ei, long long E, long long N, long long M,
                            const uint32_t *skeys, const uint32_t *svals,
                            const uint32_t *selfrank, const int *counts, int *col, int *perm,
                            int *rowid) {
    const long long kept = E - counts[0];
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < M;
         p += (long long)gridDim.x * blockDim.x) {
        const uint32_t k = skeys[p];
        if (k >= (uint32_t)N) continue;  // dropped edges sit at the end
        const long long v = svals[p];
        if (v < E) {
            col[p] = (int)ei[v];
            perm[p] = (int)(v - selfrank[v]);
        } else {
            col[p] = (int)(v - E);
            perm[p] = (int)(kept + (v - E));
        }
        rowid[p] = (int)k;
    }
}

// rowptr[n] = first position whose key is >= n, n = 0..nrows
__global__ void gb_rowptr(const uint32_t *skeys, long long size, long long nrows, int *rowptr) {
    for (long long n = blockIdx.x * (long long)blockDim.x + threadIdx.x; n <= nrows;
         n += (long long)gridDim.x * blockDim.x) {
        long long lo = 0, hi = size;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (skeys[mid] < (uint32_t)n) lo = mid + 1; else hi = mid;
        }
        rowptr[n] = (int)lo;
    }
}

__global__ void gb_iota_keys(const int *col, long long size, uint32_t *keys, uint32_t *vals) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < size;
         p += (long long)gridDim.x * blockDim.x) {
        keys[p] = (uint32_t)col[p];
        vals[p] = (uint32_t)p;
    }
}

__global__ void gb_fill_transpose(const uint32_t *svals, const int *rowid, long long size,
                                  int *col_t, int *pos_t, int *d2s) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < size;
         t += (long long)gridDim.x * blockDim.x) {
        const int p = (int)svals[t];
        pos_t[t] = p;
        col_t[t] = rowid[p];
        d2s[p] = (int)t;
    }
}

// Segment plan ------------------------------------------------------------------
struct SegCount {
    const int *rowptr;
    int seg;
    int which;  // 0: segments per row, 1: slots per row, 2: cut-row flag
    __device__ int operator()(int n) const {
        const int deg = rowptr[n + 1] - rowptr[n];
        int ns = (deg + seg - 1) / seg;
        ns = ns < 1 ? 1 : ns;
        if (which == 0) return ns;
        if (which == 1) return ns > 1 ? ns : 0;
        return ns > 1 ? 1 : 0;
    }
};

__global__ void gb_fill_plan(const int *rowptr, int nrows, int seg, const int *item_ptr,
                             const int *slot_ptr, const int *split_ptr, int *item_row,
                             int *item_beg, int *item_slot, int *split_row, int *split_slot,
                             int *totals /* [3] items, slots, splits */) {
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nrows; n += gridDim.x * blockDim.x) {
        const int b = rowptr[n], deg = rowptr[n + 1] - b;
        int ns = (deg + seg - 1) / seg;
        ns = ns < 1 ? 1 : ns;
        const int ip = item_ptr[n], sp = slot_ptr[n];
        for (int s = 0; s < ns; ++s) {
            item_row[ip + s] = n;
            item_beg[ip + s] = b + s * seg;
            item_slot[ip + s] = ns > 1 ? sp + s : -1;
        }
        if (ns > 1) {
            split_row[split_ptr[n]] = n;
            split_slot[split_ptr[n]] = sp;
        }
        if (n == nrows - 1) {
            totals[0] = ip + ns;
            totals[1] = sp + (ns > 1 ? ns : 0);
            totals[2] = split_ptr[n] + (ns > 1 ? 1 : 0);
            split_slot[totals[2]] = totals[1];
        }
    }
}

static size_t sort_temp_bytes(size_t m) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (uint32_t *)nullptr, m, 0, 32);
    return bytes;
}
static size_t scan_temp_bytes(size_t n) {
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, (int *)nullptr, (int *)nullptr, 0, n,
                                  rocprim::plus<int>());
    return bytes;
}

struct BuildWs {
    uint32_t *keys_a, *keys_b, *vals_a, *vals_b, *selfflag, *selfrank;
    int *rowid, *scan_a, *scan_b, *scan_c, *counts;
    void *temp;
    size_t temp_bytes, total;
};

static BuildWs carve_build(long long E, long long NS, long long ND, int asl, void *ws) {
    const long long N = NS > ND ? NS : ND;
    const size_t M = (size_t)(E + (asl ? ND : 0));
    Carver cv(ws);
    BuildWs w;
    w.keys_a = cv.take<uint32_t>(M);
    w.keys_b = cv.take<uint32_t>(M);
    w.vals_a = cv.take<uint32_t>(M);
    w.vals_b = cv.take<uint32_t>(M);
    w.selfflag = cv.take<uint32_t>((size_t)E + 1);
    w.selfrank = cv.take<uint32_t>((size_t)E + 1);
    w.rowid = cv.take<int>(M);
    w.scan_a = cv.take<int>((size_t)N + 1);
    w.scan_b = cv.take<int>((size_t)N + 1);
    w.scan_c = cv.take<int>((size_t)N + 1);
    w.counts = cv.take<int>(16);
    size_t t = sort_temp_bytes(M ? M : 1);
    const size_t t2 = scan_temp_bytes((size_t)(E > N ? E : N) + 1);
    w.temp_bytes = align_up(t > t2 ? t : t2);
    w.temp = cv.take<char>(w.temp_bytes);
    w.total = cv.off + 256;
    return w;
}

static inline int grid_for(long long n) {
    long long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

static int build_plan(const int *rowptr, int nrows, int seg, gatmi_plan *plan, BuildWs &w,
                      int *totals_dev, hipStream_t st) {
    if (nrows == 0) {
        plan->n_items = plan->n_split_rows = plan->n_slots = 0;
        return GATMI_OK;
    }
    int *ptrs[3] = {w.scan_a, w.scan_b, w.scan_c};
    for (int k = 0; k < 3; ++k) {
        auto in = rocprim::make_transform_iterator(rocprim::make_counting_iterator<int>(0),
                                                   SegCount{rowptr, seg, k});
        size_t tb = w.temp_bytes;
        GATMI_HIP(rocprim::exclusive_scan(w.temp, tb, in, ptrs[k], 0, (size_t)nrows,
                                          rocprim::plus<int>(), st));
    }
    hipLaunchKernelGGL(gb_fill_plan, dim3(grid_for(nrows)), dim3(256), 0, st, rowptr, nrows, seg,
                       w.scan_a, w.scan_b, w.scan_c, const_cast<int *>(plan->item_row),
                       const_cast<int *>(plan->item_beg), const_cast<int *>(plan->item_slot),
                       const_cast<int *>(plan->split_row), const_cast<int *>(plan->split_slot),
                       totals_dev);
    GATMI_HIP(hipGetLastError());
    int h[3];
    GATMI_HIP(hipMemcpyAsync(h, totals_dev, sizeof(h), hipMemcpyDeviceToHost, st));
    GATMI_HIP(hipStreamSynchronize(st));
    plan->n_items = h[0];
    plan->n_slots = h[1];
    plan->n_split_rows = h[2];
    return GATMI_OK;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" int gatmi_graph_sizes_query(int64_t E, int64_t NS, int64_t ND, int asl, int seg_len,
                                       gatmi_graph_sizes *out) {
    GATMI_REQUIRE(out, "null out");
    GATMI_REQUIRE(E >= 0 && NS >= 0 && ND >= 0 && seg_len >= 64 && seg_len % 64 == 0,
                  "bad E=%lld num_src=%lld num_dst=%lld seg_len=%d", (long long)E, (long long)NS,
                  (long long)ND, seg_len);
    GATMI_REQUIRE(!asl || NS == ND, "add_self_loops needs a square graph (num_src == num_dst)");
    GATMI_REQUIRE(E + NS + ND < (1ll << 31) - 1, "E + N = %lld does not fit int32 indices",
                  (long long)(E + NS + ND));
    const int64_t M = E + (asl ? ND : 0);
    const int64_t NM = NS > ND ? NS : ND;
    out->max_edges = M;
    out->max_items = M / seg_len + NM + 1;
    out->max_split_rows = M / seg_len + 1;
    out->workspace_bytes = (int64_t)carve_build(E, NS, ND, asl, nullptr).total;
    return GATMI_OK;
}

extern "C" int gatmi_graph_build(const int64_t *edge_index, int64_t E, int64_t NS, int64_t ND,
                                 int asl, int seg_len, gatmi_graph *g, void *ws, size_t ws_bytes,
                                 void *stream) {
    gatmi_graph_sizes sz;
    if (int e = gatmi_graph_sizes_query(E, NS, ND, asl, seg_len, &sz)) return e;
    GATMI_REQUIRE(g && (edge_index || E == 0), "null graph / edge_index");
    if (ws_bytes < (size_t)sz.workspace_bytes) {
        set_error("graph_build workspace too small: %zu < %lld", ws_bytes,
                  (long long)sz.workspace_bytes);
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    BuildWs w = carve_build(E, NS, ND, asl, ws);
    const long long M = E + (asl ? ND : 0);
    int *col = const_cast<int *>(g->col), *perm = const_cast<int *>(g->perm);
    int *rowptr = const_cast<int *>(g->rowptr), *rowptr_t = const_cast<int *>(g->rowptr_t);
    int *col_t = const_cast<int *>(g->col_t), *pos_t = const_cast<int *>(g->pos_t);
    int *d2s = const_cast<int *>(g->d2s);
    g->num_src = (int)NS;
    g->num_dst = (int)ND;
    g->seg_len = seg_len;

    GATMI_HIP(hipMemsetAsync(w.counts, 0, 16 * sizeof(int), st));
    GATMI_HIP(hipMemsetAsync(w.selfflag, 0, ((size_t)E + 1) * sizeof(uint32_t), st));
    int ep = 0;
    if (M > 0) {
        hipLaunchKernelGGL(gb_make_keys, dim3(grid_for(M)), dim3(256), 0, st, edge_index,
                           (long long)E, (long long)NS, (long long)ND, asl, w.keys_a, w.vals_a,
                           w.selfflag, w.counts + 8);
        GATMI_HIP(hipGetLastError());
        size_t tb = w.temp_bytes;
        GATMI_HIP(rocprim::exclusive_scan(w.temp, tb, w.selfflag, w.selfrank, 0u, (size_t)E + 1,
                                          rocprim::plus<uint32_t>(), st));
        hipLaunchKernelGGL(gb_count_self, dim3(1), dim3(64), 0, st, w.selfflag, w.selfrank,
                           (long long)E, w.counts);
        unsigned bits = 1;
        while ((1ull << bits) <= (unsigned long long)ND) ++bits;
        tb = w.temp_bytes;
        GATMI_HIP(rocprim::radix_sort_pairs(w.temp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b,
                                            (size_t)M, 0u, bits, st));
        hipLaunchKernelGGL(gb_fill_csr, dim3(grid_for(M)), dim3(256), 0, st, edge_index,
                           (long long)E, (long long)ND, M, w.keys_b, w.vals_b, w.selfrank, w.counts,
                           col, perm, w.rowid);
        GATMI_HIP(hipGetLastError());
        int h[16];
        GATMI_HIP(hipMemcpyAsync(h, w.counts, sizeof(h), hipMemcpyDeviceToHost, st));
        GATMI_HIP(hipStreamSynchronize(st));
        if (h[8]) {
            set_error("edge_index holds a node id outside [0, %lld) x [0, %lld)", (long long)NS,
                      (long long)ND);
            return GATMI_ERR_RANGE;
        }
        ep = (int)(M - h[0]);
    }
    g->num_edges = ep;
    hipLaunchKernelGGL(gb_rowptr, dim3(grid_for(ND + 1)), dim3(256), 0, st, w.keys_b, (long long)ep,
                       (long long)ND, rowptr);
    GATMI_HIP(hipGetLastError());
    if (ep > 0) {
        hipLaunchKernelGGL(gb_iota_keys, dim3(grid_for(ep)), dim3(256), 0, st, col, (long long)ep,
                           w.keys_a, w.vals_a);
        unsigned bits = 1;
        while ((1ull << bits) <= (unsigned long long)NS) ++bits;
        size_t tb = w.temp_bytes;
        GATMI_HIP(rocprim::radix_sort_pairs(w.temp, tb, w.keys_a, w.keys_b, w.vals_a, w.vals_b,
                                            (size_t)ep, 0u, bits, st));
        hipLaunchKernelGGL(gb_fill_transpose, dim3(grid_for(ep)), dim3(256), 0, st, w.vals_b,
                           w.rowid, (long long)ep, col_t, pos_t, d2s);
        GATMI_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL(gb_rowptr, dim3(grid_for(NS + 1)), dim3(256), 0, st, w.keys_b, (long long)ep,
                       (long long)NS, rowptr_t);
    GATMI_HIP(hipGetLastError());
    if (int e = build_plan(rowptr, (int)ND, seg_len, &g->plan, w, w.counts, st)) return e;
    if (int e = build_plan(rowptr_t, (int)NS, seg_len, &g->plan_t, w, w.counts + 4, st)) return e;
    GATMI_HIP(hipStreamSynchronize(st));
    return GATMI_OK;
}

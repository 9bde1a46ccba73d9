// On-device construction of the reference's input graph: create_knn_graph(X, k=5)
// (/root/reference/gat/converter.py:7-15 -- sklearn.neighbors.kneighbors_graph(X, k,
// mode="connectivity", include_self=True), brute force for F > 15, then COO -> edge_index).
// SURVEY.md 8a row a1, 8f rank 2.  The reference rebuilds this graph on the CPU in every
// run_experiment call; it is O(N^2 F).
//
// Exact brute force in fp64, the precision scikit-learn ranks in (X.values is float64): every
// query keeps its k best candidates, ordered by the same surrogate scikit-learn uses,
//     s(q, c) = |c|^2 - 2 <q, c>          (+ |q|^2, constant per query),
// ties broken towards the lower index.  fp64 FMA runs at the full vector rate on CDNA, and
// with F = 25 the work is far too narrow for a distance GEMM + selection to pay: one thread
// owns one query (its row in registers, its k best in registers), candidates stream through
// LDS in tiles and are read by broadcast.  100 k points: 2.5e11 FMA.
//
// Ties.  Among candidates at EQUAL distance scikit-learn's order is whatever its bounded max-heap,
// its final sort, its thread count (1024 x threads queries or fewer: per-thread heaps merged
// afterwards) and the rounding of its BLAS middle term leave behind -- implementation-defined and
// host-dependent, so it is detected rather than imitated: with `tie_flags`, the kernel keeps one
// candidate more than k and marks every query in whose k + 1 best two neighbouring distances are
// closer than 2^-40 of their magnitude (exact duplicates, and near-ties that scikit-learn's
// differently rounded surrogate could order the other way).  Unmarked queries have a strictly
// ordered neighbour list, the one scikit-learn finds; gat_amd.data.create_knn_graph's "auto"
// backend hands the whole graph to scikit-learn when any query is marked.
//
// Output layout = what converter.py:8-15 hands the model: edge_index[0] = the query i, k times
// in a row, ascending; edge_index[1] = its neighbours by ascending distance (itself first, at
// distance 0, unless an exact duplicate with a lower index ties).
#include "common.h"

namespace gatmi {

constexpr int kKnnTile = 64;     // candidates per LDS tile
constexpr int kKnnMaxK = 16;

__global__ __launch_bounds__(256) void knn_norms_kernel(const double *__restrict__ x, long long n,
                                                        int f, double *__restrict__ norms) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += gridDim.x * 256ll) {
        double s = 0.0;
        for (int c = 0; c < f; ++c) s = fma(x[i * f + c], x[i * f + c], s);
        norms[i] = s;
    }
}

// FP = feature count padded to even (zero filled); KK = list capacity >= k
template <int FP, int KK>
__global__ __launch_bounds__(256) void knn_kernel(const double *__restrict__ x,
                                                  const double *__restrict__ norms, long long n,
                                                  int f, int k, long long *__restrict__ edge_index,
                                                  unsigned char *__restrict__ tie_flags) {
    __shared__ double tile[kKnnTile][FP + 2];         // candidate row | norm | pad (16-byte rows)
    const long long qi = blockIdx.x * 256ll + threadIdx.x;
    const bool qok = qi < n;
    double q[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) q[c] = (qok && c < f) ? -2.0 * x[qi * f + c] : 0.0;
    double bd[KK];
    int bi[KK];
#pragma unroll
    for (int s = 0; s < KK; ++s) {
        bd[s] = INFINITY;
        bi[s] = 0x7fffffff;
    }
    for (long long c0 = 0; c0 < n; c0 += kKnnTile) {
        __syncthreads();
        for (int e = threadIdx.x; e < kKnnTile * (FP + 1); e += 256) {
            const int r = e / (FP + 1), c = e - r * (FP + 1);
            const long long cr = c0 + r;
            double v = 0.0;
            if (cr < n) v = c < f ? x[cr * f + c] : (c == FP ? norms[cr] : 0.0);
            else if (c == FP) v = INFINITY;           // past the end: never accepted
            tile[r][c] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < kKnnTile; ++r) {
            double s = tile[r][FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) s = fma(q[c], tile[r][c], s);
            if (s < bd[KK - 1]) {                     // strict: equal distances keep the lower index
                bd[KK - 1] = s;
                bi[KK - 1] = (int)(c0 + r);
#pragma unroll
                for (int t = KK - 1; t > 0; --t) {
                    if (bd[t] < bd[t - 1]) {
                        const double td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
                        const int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
                    }
                }
            }
        }
    }
    if (qok) {
        // the list holds KK >= k (+ 1 with tie_flags) entries: only the k best are the answer
#pragma unroll
        for (int s = 0; s < KK; ++s)
            if (s < k) {
                edge_index[qi * k + s] = qi;
                edge_index[n * k + qi * k + s] = bi[s];
            }
        if (tie_flags) {
            const double qn = norms[qi];
            bool tie = false;
#pragma unroll
            for (int s = 0; s + 1 < KK; ++s)
                if (s < k && bi[s + 1] != 0x7fffffff) {       // pairs (0,1) .. (k-1,k) that exist
                    const double tol = (qn + fabs(bd[s]) + fabs(bd[s + 1])) * 0x1p-40;
                    tie |= bd[s + 1] - bd[s] <= tol;
                }
            tie_flags[qi] = tie ? 1 : 0;
        }
    }
}

template <int FP>
static void knn_launch_k(const double *x, const double *norms, long long n, int f, int k,
                         long long *ei, unsigned char *ties, hipStream_t st) {
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int need = k + (ties ? 1 : 0);      // list capacity: one more than k to see a tie at the k-th place
    if (need <= 8)
        hipLaunchKernelGGL((knn_kernel<FP, 8>), dim3(blocks), dim3(256), 0, st, x, norms, n, f, k, ei, ties);
    else if (need <= kKnnMaxK)
        hipLaunchKernelGGL((knn_kernel<FP, kKnnMaxK>), dim3(blocks), dim3(256), 0, st, x, norms, n, f, k, ei, ties);
    else
        hipLaunchKernelGGL((knn_kernel<FP, kKnnMaxK + 1>), dim3(blocks), dim3(256), 0, st, x, norms, n, f, k, ei, ties);
}

}  // namespace gatmi

using namespace gatmi;

extern "C" size_t gatmi_knn_workspace_bytes(int64_t n) {
    return align_up((size_t)(n > 0 ? n : 0) * sizeof(double)) + 256;
}

extern "C" int gatmi_knn_graph(const double *x, int64_t n, int32_t f, int32_t k, int64_t *edge_index,
                               void *ws, size_t ws_bytes, void *stream) {
    return gatmi_knn_graph_ties(x, n, f, k, edge_index, nullptr, ws, ws_bytes, stream);
}

extern "C" int gatmi_knn_graph_ties(const double *x, int64_t n, int32_t f, int32_t k,
                                    int64_t *edge_index, uint8_t *tie_flags, void *ws,
                                    size_t ws_bytes, void *stream) {
    GATMI_REQUIRE(n >= 0 && f >= 1 && f <= 64, "knn_graph: features must be in [1, 64]");
    GATMI_REQUIRE(k >= 1 && k <= kKnnMaxK, "knn_graph: k must be in [1, 16]");
    GATMI_REQUIRE(k <= n || n == 0, "knn_graph: k exceeds the number of points (scikit-learn raises too)");
    GATMI_REQUIRE(n < (1ll << 31), "knn_graph: at most 2^31 - 1 points");
    if (n == 0) return GATMI_OK;
    GATMI_REQUIRE(x && edge_index && ws && ws_bytes >= gatmi_knn_workspace_bytes(n), "knn_graph: null argument or workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *norms = static_cast<double *>(ws);
    ProfScope ps("knn_graph", st);
    const int nb = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    hipLaunchKernelGGL(knn_norms_kernel, dim3(nb), dim3(256), 0, st, x, (long long)n, f, norms);
    long long *ei = reinterpret_cast<long long *>(edge_index);
    if (f <= 8) knn_launch_k<8>(x, norms, n, f, k, ei, tie_flags, st);
    else if (f <= 16) knn_launch_k<16>(x, norms, n, f, k, ei, tie_flags, st);
    else if (f <= 26) knn_launch_k<26>(x, norms, n, f, k, ei, tie_flags, st);
    else if (f <= 32) knn_launch_k<32>(x, norms, n, f, k, ei, tie_flags, st);
    else if (f <= 48) knn_launch_k<48>(x, norms, n, f, k, ei, tie_flags, st);
    else knn_launch_k<64>(x, norms, n, f, k, ei, tie_flags, st);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

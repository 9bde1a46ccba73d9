// Shared host/device helpers for libgat_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/gat_mi355.h"

namespace gatmi {

void set_error(const char *fmt, ...);

#define GATMI_HIP(expr)                                                                   \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            gatmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                             __FILE__, __LINE__);                                         \
            return GATMI_ERR_HIP;                                                         \
        }                                                                                 \
    } while (0)

#define GATMI_REQUIRE(cond, ...)                                                          \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            gatmi::set_error(__VA_ARGS__);                                                \
            return GATMI_ERR_ARG;                                                         \
        }                                                                                 \
    } while (0)

// Optional per-kernel timing with HIP events on the launch stream (off by default);
// see gatmi_prof_enable / gatmi_prof_read.
void prof_begin(const char *tag, hipStream_t st);
void prof_end(hipStream_t st);
struct ProfScope {
    hipStream_t st;
    ProfScope(const char *tag, hipStream_t s) : st(s) { prof_begin(tag, s); }
    ~ProfScope() { prof_end(st); }
};

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Carves 256-byte aligned pieces out of a caller-provided workspace.
struct Carver {
    char *base;
    size_t off = 0;
    explicit Carver(void *p) : base(static_cast<char *>(p)) {}
    template <class T> T *take(size_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off += align_up(count * sizeof(T));
        return p;
    }
};

inline int round_up4(int c) { return (c + 3) & ~3; }
inline int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

// Grid for the grid-stride work-item kernels: enough 256-thread blocks to fill
// 256 CUs x 8 blocks, a multiple of 8 so the XCD remap is a bijection.
inline int item_grid(long long n_groups_needed, int groups_per_block, int max_blocks = 2048) {
    long long blocks = (n_groups_needed + groups_per_block - 1) / groups_per_block;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 8) blocks = 8;
    return (int)((blocks + 7) / 8 * 8);
}

// ---- fp32 MFMA GEMM of project.hip, shared with the input-side aggregation path ----
// Optional correction of the A operand on its way into LDS (backward GEMMs only): A holds
// d_h without the destination-logit term, the tile is staged as
//   d_h[n, col] + d_a_dst[n, head(col)] * att_dst_p[col]
// so that term never makes a round trip through HBM.
struct AFix {
    const float *da_dst;   // [rows, H], NULL = no correction
    const float *att_p;    // [Dp] head-padded att_dst
    int H, Cp;
};
struct GemmBatch {
    long long sA, sB, sC, sSplit, sBias;
    const float *bias;
    int nsplit, n_store;
    float *asum;   // k-major A only: [batch * nsplit + split][M] column sums of A (or NULL)
};
// C[batch][split] (M x n_store, row stride ldc) = A (M x K) * B (K x Nn) over this split's K range.
//   a_kmajor: A[k * lda + m] else A[m * lda + k];   b_kmajor: B[k * ldb + n] else B[n * ldb + k]
struct GemmDesc {
    bool a_kmajor, b_kmajor;
    const float *A; int lda; long long sA;
    const float *B; int ldb; long long sB;
    float *C; int ldc; long long sC, sSplit;
    int M, Nn, K, n_store, nbatch, nsplit;
    int a_kpad;   // K-resident kernel only: A rows are zero padded to this many columns (0: K)
    bool even_ok; // K-resident kernel only: take an even number of float4 per A row (LDS reads 2-way conflicted)
    const float *bias; long long sBias;
    AFix fix;
    float *asum;   // see GemmBatch
};
int launch_gemm(const GemmDesc &d, hipStream_t st);
// tall row-major A (16-byte aligned rows, K <= 128), small B kept in LDS: see project.hip
bool kres_covers(const GemmDesc &d);
int launch_kres(const GemmDesc &d, hipStream_t st);
constexpr int kMaxDevices = 64;
int device_index(); // ordinal of the current device, for per-device caches (0 when out of range)
int device_cus();   // CUs of the current device (256 when it cannot be queried)
int option(const char *name, int dflt);   // gatmi_set_option / environment switch, see errors.cpp

// f32-accurate projection GEMMs on the bf16 matrix cores (three-way operand split), gemm_split3.hip
bool split3_covers(const gatmi_shape &s);
size_t split3_workspace_bytes(const gatmi_shape &s);
bool split3_fwd_makes_a_dst(const gatmi_shape &s);   // a_dst comes out of the forward GEMM's epilogue
int split3_project_fwd(const gatmi_shape *s, const float *x, const float *weight, const float *att_dst,
                       float *hp, float *a_dst, void *ws, hipStream_t st);
int split3_project_bwd(const gatmi_shape *s, const float *x, const float *weight, const float *d_hp,
                       const float *d_a_dst, const float *att_dst, float *d_weight, float *d_x,
                       void *ws, hipStream_t st);
// the row-streaming d_W kernel of gemm_split3.hip with 3 (exact split) or 1 (bf16 rounding) pieces
int rows_dw_groups(const gatmi_shape &s, int nblocks);   // slabs of `part` (Dp * F floats each) rows_dw needs
int rows_dw(const gatmi_shape *s, const float *x, const float *d_hp, const float *d_a_dst,
            const float *att_p, float *d_weight, float *part, int nblocks, int pieces, hipStream_t st);

}  // namespace gatmi

// Fused inter-layer glue of the reference's network: F.elu -> BatchNorm1d -> Dropout between
// two GAT layers (/root/reference/gat/model.py:40-42 and :43-45; SURVEY.md 8a row a12, 8f
// rank 1).  Stock PyTorch runs it as ~9 kernels and 18 passes over the [N, D] activations per
// training step (forward + backward); here it is 2 + 2 passes plus tiny finalisers:
//
//   forward   stats   per column: mean and M2 of e = ELU(x) (per-block shifted sums, combined
//                     with Chan's formula in double -> no E[e^2] - mean^2 cancellation)
//             apply   y = keep * (gamma (e - mean) rstd + beta) / (1 - p)
//   backward  reduce  d_beta = sum g', d_gamma = sum g' xhat      (g' = grad_y keep / (1 - p))
//             apply   d_x = elu'(x) gamma rstd (g' - d_beta/N - xhat d_gamma/N)
//
// BatchNorm1d semantics as in torch: biased variance for the normalisation, unbiased for
// running_var, momentum update, eval mode uses the running statistics (and then the backward
// has no batch terms).  The dropout mask is the counter hash (seed, row, column) of
// attn_device.h, never materialised (gatmi_glue_mask writes it out for tests).
#include "attn_device.h"
#include "common.h"

namespace gatmi {

constexpr int kGlueBlocks = 2048;
constexpr int kFinCols = 4;   // glue_stats_finalize_kernel: columns per block (64 row groups walk the partials)

struct GlueArgs {
    const float *x, *gy;
    float *y, *gx;
    long long rows;
    int cols, nslot;          // nslot = ceil(cols / 4) float4 slots per row (<= 256)
    int training;
    float inv_keep, inv_rows;
    uint32_t keep_thr, seed_lo, seed_hi;
    const uint64_t *seed_ptr;  // gatmi_glue.seed_on_device: the seed is read from device memory
    const float *gamma, *beta, *mean, *rstd;   // [cols]
    const float *dsum;        // [2][cols]: d_beta, d_gamma (backward apply)
    float *partial;           // [blocks][2][nslot*4]
};

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

// this thread's 4 columns of row r (guarded when cols % 4 != 0)
__device__ __forceinline__ float4 glue_ld(const float *p, long long r, int cols, int c0, bool vec) {
    const float *q = p + r * cols + c0;
    if (vec) return *reinterpret_cast<const float4 *>(q);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 + 0 < cols) v.x = q[0];
    if (c0 + 1 < cols) v.y = q[1];
    if (c0 + 2 < cols) v.z = q[2];
    if (c0 + 3 < cols) v.w = q[3];
    return v;
}
__device__ __forceinline__ void glue_st(float *p, long long r, int cols, int c0, bool vec, float4 v) {
    float *q = p + r * cols + c0;
    if (vec) {
        *reinterpret_cast<float4 *>(q) = v;
        return;
    }
    if (c0 + 0 < cols) q[0] = v.x;
    if (c0 + 1 < cols) q[1] = v.y;
    if (c0 + 2 < cols) q[2] = v.z;
    if (c0 + 3 < cols) q[3] = v.w;
}
__device__ __forceinline__ float4 glue_ldc(const float *p, int cols, int c0) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!p) return v;
    if (c0 + 0 < cols) v.x = p[c0 + 0];
    if (c0 + 1 < cols) v.y = p[c0 + 1];
    if (c0 + 2 < cols) v.z = p[c0 + 2];
    if (c0 + 3 < cols) v.w = p[c0 + 3];
    return v;
}

#define GLUE_LAYOUT                                                                       \
    const int slot = threadIdx.x % A.nslot, rsub = threadIdx.x / A.nslot;                 \
    const int rpp = 256 / A.nslot;                                                        \
    const int c0 = slot * 4;                                                              \
    const bool vec = (A.cols & 3) == 0;                                                   \
    const bool lane_on = rsub < rpp;                                                      \
    const long long per = (A.rows + gridDim.x - 1) / gridDim.x;                           \
    const long long r0 = (long long)blockIdx.x * per;                                     \
    const long long r1 = r0 + per < A.rows ? r0 + per : A.rows;

// per block and column: n_b (= r1 - r0), mean_b, M2_b of e = ELU(x) over the block's rows
__global__ __launch_bounds__(256) void glue_stats_kernel(const GlueArgs A) {
    __shared__ float4 red[2][256];
    GLUE_LAYOUT
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1, sh = s1;
    if (r0 < r1) {   // shift = the block's first row: keeps the sums small
        const float4 v = glue_ld(A.x, r0, A.cols, c0, vec);
        sh = make_float4(elu1(v.x), elu1(v.y), elu1(v.z), elu1(v.w));
    }
    if (lane_on) {
#pragma unroll 4
        for (long long r = r0 + rsub; r < r1; r += rpp) {
            const float4 v = glue_ld(A.x, r, A.cols, c0, vec);
            const float a = elu1(v.x) - sh.x, b = elu1(v.y) - sh.y, c = elu1(v.z) - sh.z,
                        d = elu1(v.w) - sh.w;
            s1.x += a; s1.y += b; s1.z += c; s1.w += d;
            s2.x = fmaf(a, a, s2.x); s2.y = fmaf(b, b, s2.y); s2.z = fmaf(c, c, s2.z);
            s2.w = fmaf(d, d, s2.w);
        }
    }
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.x < A.nslot) {
        float4 a = red[0][threadIdx.x], b = red[1][threadIdx.x];
        for (int q = 1; q < rpp; ++q) {
            const float4 o = red[0][threadIdx.x + q * A.nslot], p = red[1][threadIdx.x + q * A.nslot];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            b.x += p.x; b.y += p.y; b.z += p.z; b.w += p.w;
        }
        const float n = (float)(r1 > r0 ? r1 - r0 : 1);
        float4 mean, m2;
        mean.x = sh.x + a.x / n; m2.x = b.x - a.x * a.x / n;
        mean.y = sh.y + a.y / n; m2.y = b.y - a.y * a.y / n;
        mean.z = sh.z + a.z / n; m2.z = b.z - a.z * a.z / n;
        mean.w = sh.w + a.w / n; m2.w = b.w - a.w * a.w / n;
        float *dst = A.partial + (size_t)blockIdx.x * 2 * A.nslot * 4 + c0;
        *reinterpret_cast<float4 *>(dst) = mean;
        *reinterpret_cast<float4 *>(dst + A.nslot * 4) = m2;
    }
}

// Chan's combination of the per-block (n_b, mean_b, M2_b), in double.  One block per kFinCols
// columns: 256 / kFinCols row groups each take every such partial row, fixed-order adds through LDS.
__global__ __launch_bounds__(256) void glue_stats_finalize_kernel(
    const float *partial, int nblocks, long long rows, int cols, int nslot, float eps, float momentum,
    float *save_mean, float *save_rstd, float *running_mean, float *running_var,
    float *m2_out /* row partition: M2 = sum (e - mean)^2 of these rows instead of rstd; the caller
                     combines (rows, mean, M2) over the ranks */) {
    __shared__ double red[256];
    constexpr int CG = kFinCols, RG = 256 / CG;   // columns per block, row groups
    const int c = blockIdx.x * CG + (threadIdx.x & (CG - 1)), rg = threadIdx.x / CG;
    const bool on = c < cols;
    const long long per = nblocks > 0 ? (rows + nblocks - 1) / nblocks : 0;    // no partials: zero rows
    const int ld = 2 * nslot * 4;
    auto nb = [&](int b) -> double {
        const long long r0 = (long long)b * per, r1 = r0 + per < rows ? r0 + per : rows;
        return r1 > r0 ? (double)(r1 - r0) : 0.0;
    };
    double w = 0.0;
    if (on)
        for (int b = rg; b < nblocks; b += RG) w += nb(b) * partial[(size_t)b * ld + c];
    red[threadIdx.x] = w;
    __syncthreads();
    double wsum = 0.0;
    for (int q = 0; q < RG; ++q) wsum += red[(threadIdx.x & (CG - 1)) + q * CG];
    const double mean = rows > 0 ? wsum / (double)rows : 0.0;
    __syncthreads();
    double m = 0.0;
    if (on)
        for (int b = rg; b < nblocks; b += RG) {
            const double n = nb(b);
            if (n > 0.0) {
                const double d = partial[(size_t)b * ld + c] - mean;
                m += partial[(size_t)b * ld + nslot * 4 + c] + n * d * d;
            }
        }
    red[threadIdx.x] = m;
    __syncthreads();
    if (rg == 0 && on) {
        double m2 = 0.0;
        for (int q = 0; q < RG; ++q) m2 += red[threadIdx.x + q * CG];
        const double var = rows > 0 ? m2 / (double)rows : 0.0;
        save_mean[c] = (float)mean;
        if (m2_out) {
            m2_out[c] = (float)m2;
            return;
        }
        save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unb = rows > 1 ? m2 / (double)(rows - 1) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
}

// eval mode: rstd from the running variance
__global__ void glue_eval_stats_kernel(const float *running_mean, const float *running_var, int cols,
                                       float eps, float *save_mean, float *save_rstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    save_mean[c] = running_mean[c];
    save_rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

__device__ __forceinline__ float keepf(const GlueArgs &A, long long r, int c) {
    if (A.keep_thr == 0) return 1.0f;
    uint32_t lo = A.seed_lo, hi = A.seed_hi;
    if (A.seed_ptr) {
        const uint64_t sd = *A.seed_ptr;
        lo = (uint32_t)sd;
        hi = (uint32_t)(sd >> 32);
    }
    return keep_hash(lo, hi, A.keep_thr, (uint32_t)r, (uint32_t)c) ? A.inv_keep : 0.f;
}

__global__ __launch_bounds__(256) void glue_apply_fwd_kernel(const GlueArgs A) {
    GLUE_LAYOUT
    if (!lane_on) return;
    const float4 g = glue_ldc(A.gamma, A.cols, c0), b = glue_ldc(A.beta, A.cols, c0);
    const float4 mu = glue_ldc(A.mean, A.cols, c0), rs = glue_ldc(A.rstd, A.cols, c0);
    const float4 sc = make_float4(g.x * rs.x, g.y * rs.y, g.z * rs.z, g.w * rs.w);
#pragma unroll 4
    for (long long r = r0 + rsub; r < r1; r += rpp) {
        const float4 v = glue_ld(A.x, r, A.cols, c0, vec);
        float4 o;
        o.x = keepf(A, r, c0 + 0) * fmaf(elu1(v.x) - mu.x, sc.x, b.x);
        o.y = keepf(A, r, c0 + 1) * fmaf(elu1(v.y) - mu.y, sc.y, b.y);
        o.z = keepf(A, r, c0 + 2) * fmaf(elu1(v.z) - mu.z, sc.z, b.z);
        o.w = keepf(A, r, c0 + 3) * fmaf(elu1(v.w) - mu.w, sc.w, b.w);
        glue_st(A.y, r, A.cols, c0, vec, o);
    }
}

// per block and column: sum g', sum g' xhat
__global__ __launch_bounds__(256) void glue_reduce_bwd_kernel(const GlueArgs A) {
    __shared__ float4 red[2][256];
    GLUE_LAYOUT
    const float4 mu = glue_ldc(A.mean, A.cols, c0), rs = glue_ldc(A.rstd, A.cols, c0);
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
    if (lane_on) {
#pragma unroll 4
        for (long long r = r0 + rsub; r < r1; r += rpp) {
            const float4 v = glue_ld(A.x, r, A.cols, c0, vec);
            const float4 gy = glue_ld(A.gy, r, A.cols, c0, vec);
            const float a = gy.x * keepf(A, r, c0 + 0), b = gy.y * keepf(A, r, c0 + 1),
                        c = gy.z * keepf(A, r, c0 + 2), d = gy.w * keepf(A, r, c0 + 3);
            sb.x += a; sb.y += b; sb.z += c; sb.w += d;
            sg.x = fmaf(a, (elu1(v.x) - mu.x) * rs.x, sg.x);
            sg.y = fmaf(b, (elu1(v.y) - mu.y) * rs.y, sg.y);
            sg.z = fmaf(c, (elu1(v.z) - mu.z) * rs.z, sg.z);
            sg.w = fmaf(d, (elu1(v.w) - mu.w) * rs.w, sg.w);
        }
    }
    red[0][threadIdx.x] = sb;
    red[1][threadIdx.x] = sg;
    __syncthreads();
    if (threadIdx.x < A.nslot) {
        float4 a = red[0][threadIdx.x], b = red[1][threadIdx.x];
        for (int q = 1; q < rpp; ++q) {
            const float4 o = red[0][threadIdx.x + q * A.nslot], p = red[1][threadIdx.x + q * A.nslot];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
            b.x += p.x; b.y += p.y; b.z += p.z; b.w += p.w;
        }
        float *dst = A.partial + (size_t)blockIdx.x * 2 * A.nslot * 4 + c0;
        *reinterpret_cast<float4 *>(dst) = a;
        *reinterpret_cast<float4 *>(dst + A.nslot * 4) = b;
    }
}

// dsum[k][c] = sum over blocks of partial[b][k][c], fixed order; 16 row groups per column.  The two
// rows also go to copy0 / copy1 when given (d_beta, d_gamma: stored from this kernel, not copied by
// hipMemcpyAsync -- copy and memset nodes of a captured HIP graph were seen to run out of order).
__global__ __launch_bounds__(256) void glue_sum_partials_kernel(const float *partial, int nblocks,
                                                                int cols, int nslot, float *dsum,
                                                                float *copy0, float *copy1) {
    __shared__ float red[256];
    const int k = blockIdx.y;
    constexpr int CG = kFinCols, RG = 256 / CG;
    const int c = blockIdx.x * CG + (threadIdx.x & (CG - 1)), rg = threadIdx.x / CG;
    const size_t ld = (size_t)2 * nslot * 4;
    float s = 0.f;
    if (c < cols)
        for (int b = rg; b < nblocks; b += RG) s += partial[(size_t)b * ld + (size_t)k * nslot * 4 + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && c < cols) {
        float tot = 0.f;
        for (int q = 0; q < RG; ++q) tot += red[threadIdx.x + q * CG];
        dsum[(size_t)k * cols + c] = tot;
        float *cp = k == 0 ? copy0 : copy1;
        if (cp) cp[c] = tot;
    }
}

__global__ __launch_bounds__(256) void glue_apply_bwd_kernel(const GlueArgs A) {
    GLUE_LAYOUT
    if (!lane_on) return;
    const float4 g = glue_ldc(A.gamma, A.cols, c0);
    const float4 mu = glue_ldc(A.mean, A.cols, c0), rs = glue_ldc(A.rstd, A.cols, c0);
    float4 db = glue_ldc(A.dsum, A.cols, c0), dg = glue_ldc(A.dsum + A.cols, A.cols, c0);
    const float t = A.training ? A.inv_rows : 0.f;   // eval: no batch terms
    db.x *= t; db.y *= t; db.z *= t; db.w *= t;
    dg.x *= t; dg.y *= t; dg.z *= t; dg.w *= t;
    const float4 sc = make_float4(g.x * rs.x, g.y * rs.y, g.z * rs.z, g.w * rs.w);
#pragma unroll 4
    for (long long r = r0 + rsub; r < r1; r += rpp) {
        const float4 v = glue_ld(A.x, r, A.cols, c0, vec);
        const float4 gy = glue_ld(A.gy, r, A.cols, c0, vec);
        float4 o;
#define GLUE_DX(F, K)                                                                     \
        {                                                                                 \
            const float e = elu1(v.F);                                                    \
            const float xh = (e - mu.F) * rs.F;                                           \
            const float gp = gy.F * keepf(A, r, c0 + K);                                  \
            o.F = (v.F > 0.f ? 1.0f : e + 1.0f) * sc.F * (gp - db.F - xh * dg.F);          \
        }
        GLUE_DX(x, 0) GLUE_DX(y, 1) GLUE_DX(z, 2) GLUE_DX(w, 3)
#undef GLUE_DX
        glue_st(A.gx, r, A.cols, c0, vec, o);
    }
}

__global__ void glue_mask_kernel(uint8_t *out, long long rows, int cols, uint32_t thr, uint32_t lo,
                                 uint32_t hi) {
    const long long total = rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x)
        out[i] = thr == 0 ? 1 : keep_hash(lo, hi, thr, (uint32_t)(i / cols), (uint32_t)(i % cols));
}

static int glue_check(const gatmi_glue *g) {
    GATMI_REQUIRE(g, "null glue descriptor");
    GATMI_REQUIRE(g->rows >= 0 && g->rows < (1ll << 32) && g->cols >= 1 && g->cols <= 1024,
                  "glue: rows %lld / cols %d out of range (cols <= 1024)", (long long)g->rows, g->cols);
    GATMI_REQUIRE(g->dropout_p >= 0.f && g->dropout_p < 1.f, "glue: dropout_p %f outside [0,1)",
                  g->dropout_p);
    GATMI_REQUIRE(g->training || (g->running_mean && g->running_var),
                  "glue: eval mode needs running statistics");
    return GATMI_OK;
}

static void glue_fill(GlueArgs &A, const gatmi_glue *g) {
    memset(&A, 0, sizeof(A));
    A.rows = g->rows;
    A.cols = g->cols;
    A.nslot = (g->cols + 3) / 4;
    A.training = g->training;
    const bool drop = g->training && g->dropout_p > 0.f;
    A.inv_keep = drop ? 1.0f / (1.0f - g->dropout_p) : 1.0f;
    A.keep_thr = drop ? (uint32_t)((double)g->dropout_p * 16777216.0) : 0u;
    A.seed_lo = (uint32_t)g->seed;
    A.seed_hi = (uint32_t)(g->seed >> 32);
    A.seed_ptr = g->seed_on_device ? reinterpret_cast<const uint64_t *>((uintptr_t)g->seed) : nullptr;
    A.inv_rows = g->rows > 0 ? 1.0f / (float)g->rows : 0.f;
    A.gamma = g->gamma;
    A.beta = g->beta;
}

static int glue_blocks(long long rows) {
    long long b = (rows + 63) / 64;
    if (b > kGlueBlocks) b = kGlueBlocks;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace gatmi

using namespace gatmi;

extern "C" size_t gatmi_glue_workspace_bytes(int cols) {
    const size_t nslot = ((size_t)cols + 3) / 4;
    return align_up((size_t)kGlueBlocks * 2 * nslot * 4 * sizeof(float)) + align_up(2 * (size_t)cols * 4) + 256;
}

extern "C" int gatmi_glue_fwd(const gatmi_glue *g, const float *x, float *y, float *save_mean,
                              float *save_rstd, void *ws, size_t ws_bytes, void *stream) {
    if (int e = glue_check(g)) return e;
    GATMI_REQUIRE(x && y && save_mean && save_rstd && g->gamma && g->beta, "glue_fwd: null argument");
    if (ws_bytes < gatmi_glue_workspace_bytes(g->cols)) {
        set_error("glue workspace too small: %zu < %zu", ws_bytes, gatmi_glue_workspace_bytes(g->cols));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    GlueArgs A;
    glue_fill(A, g);
    A.x = x;
    A.y = y;
    A.partial = static_cast<float *>(ws);
    const int blocks = glue_blocks(g->rows);
    const int cb = (g->cols + 255) / 256;
    if (g->training) {
        ProfScope ps("glue_stats", st);
        GATMI_REQUIRE(g->rows > 0, "glue: training-mode BatchNorm over zero rows");
        hipLaunchKernelGGL(glue_stats_kernel, dim3(blocks), dim3(256), 0, st, A);
        hipLaunchKernelGGL(glue_stats_finalize_kernel, dim3((g->cols + kFinCols - 1) / kFinCols), dim3(256), 0, st,
                           A.partial, blocks,
                           (long long)g->rows, g->cols, A.nslot, g->eps, g->momentum, save_mean,
                           save_rstd, g->running_mean, g->running_var, nullptr);
    } else {
        hipLaunchKernelGGL(glue_eval_stats_kernel, dim3(cb), dim3(256), 0, st, g->running_mean,
                           g->running_var, g->cols, g->eps, save_mean, save_rstd);
    }
    GATMI_HIP(hipGetLastError());
    if (g->rows == 0) return GATMI_OK;
    A.mean = save_mean;
    A.rstd = save_rstd;
    {
        ProfScope ps("glue_apply_fwd", st);
        hipLaunchKernelGGL(glue_apply_fwd_kernel, dim3(blocks), dim3(256), 0, st, A);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

extern "C" int gatmi_glue_bwd(const gatmi_glue *g, const float *x, const float *grad_y,
                              const float *save_mean, const float *save_rstd, float *grad_x,
                              float *grad_gamma, float *grad_beta, void *ws, size_t ws_bytes,
                              void *stream) {
    if (int e = glue_check(g)) return e;
    GATMI_REQUIRE(x && grad_y && save_mean && save_rstd && grad_x && g->gamma,
                  "glue_bwd: null argument");
    if (ws_bytes < gatmi_glue_workspace_bytes(g->cols)) {
        set_error("glue workspace too small: %zu < %zu", ws_bytes, gatmi_glue_workspace_bytes(g->cols));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    GlueArgs A;
    glue_fill(A, g);
    A.x = x;
    A.gy = grad_y;
    A.gx = grad_x;
    A.mean = save_mean;
    A.rstd = save_rstd;
    A.partial = static_cast<float *>(ws);
    float *dsum = reinterpret_cast<float *>(static_cast<char *>(ws) +
                                            align_up((size_t)kGlueBlocks * 2 * A.nslot * 4 * sizeof(float)));
    A.dsum = dsum;
    const int blocks = glue_blocks(g->rows);
    {
        ProfScope ps("glue_reduce_bwd", st);
        if (g->rows > 0) hipLaunchKernelGGL(glue_reduce_bwd_kernel, dim3(blocks), dim3(256), 0, st, A);
        hipLaunchKernelGGL(glue_sum_partials_kernel, dim3((g->cols + kFinCols - 1) / kFinCols, 2), dim3(256), 0, st,
                           A.partial, g->rows > 0 ? blocks : 0, g->cols, A.nslot, dsum, grad_beta, grad_gamma);
        GATMI_HIP(hipGetLastError());
    }
    if (g->rows == 0) return GATMI_OK;
    {
        ProfScope ps("glue_apply_bwd", st);
        hipLaunchKernelGGL(glue_apply_bwd_kernel, dim3(blocks), dim3(256), 0, st, A);
        GATMI_HIP(hipGetLastError());
    }
    return GATMI_OK;
}

extern "C" int gatmi_glue_mask(int64_t rows, int32_t cols, float dropout_p, uint64_t seed,
                               uint8_t *mask, void *stream) {
    GATMI_REQUIRE(mask && rows >= 0 && cols >= 1 && dropout_p >= 0.f && dropout_p < 1.f,
                  "glue_mask: bad argument");
    const long long total = rows * cols;
    if (total == 0) return GATMI_OK;
    const uint32_t thr = dropout_p > 0.f ? (uint32_t)((double)dropout_p * 16777216.0) : 0u;
    const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(glue_mask_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       mask, (long long)rows, cols, thr, (uint32_t)seed, (uint32_t)(seed >> 32));
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

// ---- The same glue on a ROW PARTITION (a destination-partitioned network: SURVEY.md 8e, "BatchNorm
// batch statistics need a cross-GPU all-reduce"): the two reductions and the two passes as calls of
// their own, so that the caller can combine the per-rank statistics / sums in between.
//   forward   gatmi_glue_stats (this rank's rows -> mean, M2 per column)  | caller: Chan's combination
//             over the ranks -> mean, rstd of ALL rows (+ running statistics)  | gatmi_glue_apply
//   backward  gatmi_glue_bwd_sums (this rank's sum g', sum g' xhat)  | caller: all-reduce  |
//             gatmi_glue_bwd_apply (with the sums and the row count of all ranks)
extern "C" int gatmi_glue_stats(const gatmi_glue *g, const float *x, float *mean, float *m2, void *ws,
                                size_t ws_bytes, void *stream) {
    if (int e = glue_check(g)) return e;
    GATMI_REQUIRE((x || g->rows == 0) && mean && m2, "glue_stats: null argument");
    if (ws_bytes < gatmi_glue_workspace_bytes(g->cols)) {
        set_error("glue workspace too small: %zu < %zu", ws_bytes, gatmi_glue_workspace_bytes(g->cols));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    GlueArgs A;
    glue_fill(A, g);
    A.x = x;
    A.partial = static_cast<float *>(ws);
    const int blocks = glue_blocks(g->rows);
    ProfScope ps("glue_stats", st);
    // zero rows: the finaliser walks no partial (nblocks = 0) and writes mean = M2 = 0
    if (g->rows > 0) hipLaunchKernelGGL(glue_stats_kernel, dim3(blocks), dim3(256), 0, st, A);
    hipLaunchKernelGGL(glue_stats_finalize_kernel, dim3((g->cols + kFinCols - 1) / kFinCols), dim3(256), 0, st,
                       A.partial, g->rows > 0 ? blocks : 0, (long long)g->rows, g->cols, A.nslot, g->eps,
                       g->momentum, mean, nullptr, nullptr, nullptr, m2);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" int gatmi_glue_apply(const gatmi_glue *g, const float *x, const float *mean, const float *rstd,
                                float *y, void *stream) {
    if (int e = glue_check(g)) return e;
    GATMI_REQUIRE(mean && rstd && g->gamma && g->beta && ((x && y) || g->rows == 0), "glue_apply: null argument");
    if (g->rows == 0) return GATMI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    GlueArgs A;
    glue_fill(A, g);
    A.x = x;
    A.y = y;
    A.mean = mean;
    A.rstd = rstd;
    ProfScope ps("glue_apply_fwd", st);
    hipLaunchKernelGGL(glue_apply_fwd_kernel, dim3(glue_blocks(g->rows)), dim3(256), 0, st, A);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" int gatmi_glue_bwd_sums(const gatmi_glue *g, const float *x, const float *grad_y,
                                   const float *mean, const float *rstd, float *sums, void *ws,
                                   size_t ws_bytes, void *stream) {
    if (int e = glue_check(g)) return e;
    GATMI_REQUIRE(mean && rstd && sums && ((x && grad_y) || g->rows == 0), "glue_bwd_sums: null argument");
    if (ws_bytes < gatmi_glue_workspace_bytes(g->cols)) {
        set_error("glue workspace too small: %zu < %zu", ws_bytes, gatmi_glue_workspace_bytes(g->cols));
        return GATMI_ERR_WS;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    GlueArgs A;
    glue_fill(A, g);
    A.x = x;
    A.gy = grad_y;
    A.mean = mean;
    A.rstd = rstd;
    A.partial = static_cast<float *>(ws);
    const int blocks = glue_blocks(g->rows);
    ProfScope ps("glue_reduce_bwd", st);
    if (g->rows > 0) hipLaunchKernelGGL(glue_reduce_bwd_kernel, dim3(blocks), dim3(256), 0, st, A);
    hipLaunchKernelGGL(glue_sum_partials_kernel, dim3((g->cols + kFinCols - 1) / kFinCols, 2), dim3(256), 0, st,
                       A.partial, g->rows > 0 ? blocks : 0, g->cols, A.nslot, sums, nullptr, nullptr);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" int gatmi_glue_bwd_apply(const gatmi_glue *g, int64_t total_rows, const float *x,
                                    const float *grad_y, const float *mean, const float *rstd,
                                    const float *sums, float *grad_x, void *stream) {
    if (int e = glue_check(g)) return e;
    GATMI_REQUIRE(mean && rstd && sums && g->gamma && ((x && grad_y && grad_x) || g->rows == 0),
                  "glue_bwd_apply: null argument");
    GATMI_REQUIRE(total_rows >= g->rows, "glue_bwd_apply: %lld rows in all, %lld of them here",
                  (long long)total_rows, (long long)g->rows);
    if (g->rows == 0) return GATMI_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    GlueArgs A;
    glue_fill(A, g);
    A.inv_rows = total_rows > 0 ? 1.0f / (float)total_rows : 0.f;    // the batch is all ranks' rows
    A.x = x;
    A.gy = grad_y;
    A.gx = grad_x;
    A.mean = mean;
    A.rstd = rstd;
    A.dsum = sums;
    ProfScope ps("glue_apply_bwd", st);
    hipLaunchKernelGGL(glue_apply_bwd_kernel, dim3(glue_blocks(g->rows)), dim3(256), 0, st, A);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

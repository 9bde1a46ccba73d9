// The tail of the reference's network: F.log_softmax(x, dim=1) and F.nll_loss(out, y)
// (/root/reference/gat/model.py:47 and :53).  The tensor is
// small ([N, classes], classes = 2 ... 40), but the stock nll_loss kernels reduce it with a
// single workgroup: 0.25 + 0.16 ms per step on the arxiv-shaped graph, 7 % of a training
// epoch.  Here: one pass forward (log-probabilities + per-block loss partials, summed in
// double by a finaliser) and one pass backward
//     d_x = g/count (softmax - onehot)  [+ d_logp - softmax * sum_c d_logp, if logp is used elsewhere].
// LPR = next_pow2(classes) (<= 64) consecutive lanes own a row (coalesced loads, shuffle
// reductions); classes <= 256 (4 values per lane).  Targets outside [0, classes) are ignored
// (not counted), which covers torch's ignore_index = -100 default.
#include "attn_device.h"
#include "common.h"

namespace gatmi {

constexpr int kTailBlocks = 1024;
constexpr int kTailV = 4;

template <class T> __device__ __forceinline__ T group_red_sum(T v, int lpr) {
    for (int o = 1; o < lpr; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float group_red_max(float v, int lpr) {
    for (int o = 1; o < lpr; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(256) void tail_fwd_kernel(const float *__restrict__ x,
                                                       const long long *__restrict__ y,
                                                       float *__restrict__ logp, double *partial,
                                                       long long rows, int K, int lpr) {
    const int rpb = 256 / lpr;                       // rows per block and step
    const int sub = threadIdx.x % lpr;
    double loss = 0.0, cnt = 0.0;
    for (long long r = (long long)blockIdx.x * rpb + threadIdx.x / lpr; r < rows;
         r += (long long)gridDim.x * rpb) {
        float v[kTailV];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < kTailV; ++t) {
            const int c = sub + t * lpr;
            v[t] = c < K ? x[r * K + c] : -INFINITY;
            m = fmaxf(m, v[t]);
        }
        m = group_red_max(m, lpr);
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < kTailV; ++t)
            if (sub + t * lpr < K) s += expf(v[t] - m);
        s = group_red_sum(s, lpr);
        const float lse = m + logf(s);
        const long long tgt = y[r];
#pragma unroll
        for (int t = 0; t < kTailV; ++t) {
            const int c = sub + t * lpr;
            if (c < K) {
                const float lp = v[t] - lse;
                logp[r * K + c] = lp;
                if (c == tgt) loss -= (double)lp;
            }
        }
        if (sub == 0 && tgt >= 0 && tgt < K) cnt += 1.0;
    }
    __shared__ double sh[2][256];
    sh[0][threadIdx.x] = loss;
    sh[1][threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sh[0][0];
        partial[2 * blockIdx.x + 1] = sh[1][0];
    }
}

// stats[0] = mean loss over the counted rows (NaN when none, as torch), stats[1] = 1 / count
__global__ __launch_bounds__(256) void tail_finalize_kernel(const double *partial, int nblocks,
                                                            float *stats) {
    __shared__ double sh[2][256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = sh[1][0];
        stats[0] = (float)(sh[0][0] / n);
        stats[1] = n > 0.0 ? (float)(1.0 / n) : 0.f;
    }
}

__global__ __launch_bounds__(256) void tail_bwd_kernel(const float *__restrict__ logp,
                                                       const long long *__restrict__ y,
                                                       const float *__restrict__ stats,
                                                       const float *__restrict__ grad_loss,
                                                       const float *__restrict__ grad_logp,
                                                       float *__restrict__ dx, long long rows, int K,
                                                       int lpr) {
    const int rpb = 256 / lpr;
    const int sub = threadIdx.x % lpr;
    const float gw = grad_loss ? grad_loss[0] * stats[1] : 0.f;
    for (long long r = (long long)blockIdx.x * rpb + threadIdx.x / lpr; r < rows;
         r += (long long)gridDim.x * rpb) {
        const long long tgt = y[r];
        const bool counted = tgt >= 0 && tgt < K;
        float p[kTailV], g[kTailV];
        float gs = 0.f;
#pragma unroll
        for (int t = 0; t < kTailV; ++t) {
            const int c = sub + t * lpr;
            p[t] = c < K ? expf(logp[r * K + c]) : 0.f;
            g[t] = (grad_logp && c < K) ? grad_logp[r * K + c] : 0.f;
            gs += g[t];
        }
        if (grad_logp) gs = group_red_sum(gs, lpr);
#pragma unroll
        for (int t = 0; t < kTailV; ++t) {
            const int c = sub + t * lpr;
            if (c < K) {
                float d = g[t] - p[t] * gs;
                if (counted) d += gw * (p[t] - (c == tgt ? 1.f : 0.f));
                dx[r * K + c] = d;
            }
        }
    }
}

static int tail_geometry(long long rows, int K, int *lpr, int *blocks) {
    int l = 1;
    while (l * kTailV < K && l < 64) l <<= 1;   // a lane takes up to kTailV classes: 40 classes -> 16 lanes per row
    *lpr = l;
    const long long need = (rows + 256 / l - 1) / (256 / l);
    *blocks = (int)(need < 1 ? 1 : (need > kTailBlocks ? kTailBlocks : need));
    return GATMI_OK;
}

}  // namespace gatmi

using namespace gatmi;

extern "C" size_t gatmi_tail_workspace_bytes(void) { return (size_t)kTailBlocks * 2 * sizeof(double); }

extern "C" int gatmi_tail_fwd(int64_t rows, int32_t classes, const float *logits,
                              const int64_t *target, float *logp, float *loss_stats, void *ws,
                              size_t ws_bytes, void *stream) {
    GATMI_REQUIRE(rows >= 0 && classes >= 1 && classes <= 64 * kTailV, "tail_fwd: classes must be in [1, 256]");
    GATMI_REQUIRE(loss_stats && ws && ws_bytes >= gatmi_tail_workspace_bytes(), "tail_fwd: workspace too small");
    GATMI_REQUIRE(rows == 0 || (logits && target && logp), "tail_fwd: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int lpr, blocks;
    tail_geometry(rows, classes, &lpr, &blocks);
    ProfScope ps("tail_fwd", st);
    double *partial = static_cast<double *>(ws);
    if (rows > 0)
        hipLaunchKernelGGL(tail_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits,
                           reinterpret_cast<const long long *>(target), logp, partial,
                           (long long)rows, classes, lpr);
    hipLaunchKernelGGL(tail_finalize_kernel, dim3(1), dim3(256), 0, st, partial, rows > 0 ? blocks : 0,
                       loss_stats);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

extern "C" int gatmi_tail_bwd(int64_t rows, int32_t classes, const float *logp,
                              const int64_t *target, const float *loss_stats,
                              const float *grad_loss, const float *grad_logp, float *grad_logits,
                              void *stream) {
    GATMI_REQUIRE(rows >= 0 && classes >= 1 && classes <= 64 * kTailV, "tail_bwd: classes must be in [1, 256]");
    if (rows == 0) return GATMI_OK;
    GATMI_REQUIRE(logp && target && loss_stats && grad_logits, "tail_bwd: null argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int lpr, blocks;
    tail_geometry(rows, classes, &lpr, &blocks);
    ProfScope ps("tail_bwd", st);
    hipLaunchKernelGGL(tail_bwd_kernel, dim3(blocks), dim3(256), 0, st, logp,
                       reinterpret_cast<const long long *>(target), loss_stats, grad_loss, grad_logp,
                       grad_logits, (long long)rows, classes, lpr);
    GATMI_HIP(hipGetLastError());
    return GATMI_OK;
}

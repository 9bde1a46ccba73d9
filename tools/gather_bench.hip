// What bounds a random-row gather on MI355X: lines or bytes?  Stand-alone microbenchmark of the access
// pattern of the input-side edge kernels (inagg.hip): a wave instruction loads two rows, one float4 per
// lane, eight pairs in flight; rows are picked by a random index list read 64 at a time.
//   rows of R bytes at a pitch of P bytes, optionally plus one 16-byte piece per edge from a second,
//   compact table of 32-byte entries ("tail": where the last 16 bytes of a 400-byte row and the
//   logits could live so that the row itself is exactly three 128-byte lines).
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_bench.hip -o gpurun_out/gather_bench
// run:   gpurun_out/gather_bench            (prints one line per layout)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, int OCC>
__global__ __launch_bounds__(256, OCC) void gather_kernel(const char *__restrict__ rows, long long pitch, int lanes,
                                                        const char *__restrict__ tail, const int *__restrict__ idx,
                                                        long long n_edges, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, half = lane >> 5, p = lane & 31;
    const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6), nwaves = gridDim.x * 4ll;
    const bool row_lane = p < lanes, tail_lane = tail != nullptr && p == lanes;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long c = wave * 64; c < n_edges; c += nwaves * 64) {
        const int n = (int)(n_edges - c < 64 ? n_edges - c : 64);
        const int cj = idx[c + (lane < n ? lane : n - 1)];
        auto load = [&](f4(&v)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 2 * u + half;
                const int j = __shfl(cj, k < n ? k : n - 1, 64);
                // one unconditional load per lane (idle lanes re-read the row's first piece, as in inagg.hip)
                const char *src = tail_lane ? tail + (size_t)j * 32 : rows + (size_t)j * pitch + (row_lane ? p * 16 : 0);
                v[u] = *reinterpret_cast<const f4 *>(src);
            }
        };
        f4 va[U], vb[U];
        load(va, 0);
        for (int k0 = 0; k0 < n; k0 += 4 * U) {
            load(vb, k0 + 2 * U);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += va[u];
            load(va, k0 + 4 * U);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += vb[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];   // keep the loads alive
}


// one 1024-byte row per wave instruction (the projected-row kernels of attn.hip: 256 channels f32)
template <int U>
__global__ __launch_bounds__(256, 4) void gather1k_kernel(const char *__restrict__ rows, const int *__restrict__ idx,
                                                          long long n_edges, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave = blockIdx.x * 4ll + (threadIdx.x >> 6), nwaves = gridDim.x * 4ll;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long long c = wave * 64; c < n_edges; c += nwaves * 64) {
        const int n = (int)(n_edges - c < 64 ? n_edges - c : 64);
        const int cj = idx[c + (lane < n ? lane : n - 1)];
        auto load = [&](f4(&v)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = __builtin_amdgcn_readlane(cj, k0 + u < n ? k0 + u : n - 1);
                v[u] = *reinterpret_cast<const f4 *>(rows + (size_t)j * 1024 + lane * 16);
            }
        };
        f4 va[U], vb[U];
        load(va, 0);
        for (int k0 = 0; k0 < n; k0 += 2 * U) {
            load(vb, k0 + U);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += va[u];
            load(va, k0 + 2 * U);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += vb[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// the same with the projected-row forward's output: per 26 gathered rows one 1024-byte row is written
// (SM 0 plain, 1 nontemporal, 2 plain but only every row's first line, 3 none)
template <int U, int SM>
__global__ __launch_bounds__(256, 4) void gather1k_store_kernel(const char *__restrict__ rows, const int *__restrict__ idx,
                                                                int n_items, int deg, float *__restrict__ dst,
                                                                float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    f4 keep = {0.f, 0.f, 0.f, 0.f}, prev = {0.f, 0.f, 0.f, 0.f};
    int prev_it = 0;
    bool have_prev = false;
    for (int it = wave; it < n_items; it += nwaves) {
        const long long c = (long long)it * deg;
        const int n = deg;
        const int cj = idx[c + (lane < n ? lane : n - 1)];
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        auto load = [&](f4(&v)[U], int k0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = __builtin_amdgcn_readlane(cj, k0 + u < n ? k0 + u : n - 1);
                v[u] = *reinterpret_cast<const f4 *>(rows + (size_t)j * 1024 + lane * 16);
            }
        };
        f4 va[U], vb[U];
        load(va, 0);
        if (SM == 4) {
            // the PREVIOUS item's row leaves now, behind this item's first loads: the waits on those
            // loads (vmcnt counts loads and stores in issue order on gfx9) no longer include its ack
            if (have_prev) *reinterpret_cast<f4 *>(dst + (size_t)prev_it * 256 + lane * 4) = prev;
        }
        for (int k0 = 0; k0 < n; k0 += 2 * U) {
            load(vb, k0 + U);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += va[u];
            load(va, k0 + 2 * U);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += vb[u];
        }
        f4 *d = reinterpret_cast<f4 *>(dst + (size_t)it * 256 + lane * 4);
        if (SM == 0) *d = acc;
        else if (SM == 1) __builtin_nontemporal_store(acc, d);
        else if (SM == 2) { if (lane < 8) *d = acc; else keep += acc; }
        else if (SM == 4) { prev = acc; prev_it = it; have_prev = true; }
        else keep += acc;
    }
    if (SM == 4 && have_prev) *reinterpret_cast<f4 *>(dst + (size_t)prev_it * 256 + lane * 4) = prev;
    if (keep[0] + keep[1] + keep[2] + keep[3] == 12345.678f) out[0] = keep[0];
}

// streams: read-only, write-only, copy (what a written byte costs next to a read one, no gathers involved)
template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const f4 *__restrict__ src, f4 *__restrict__ dst, long long n4,
                                                     float *__restrict__ out) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += 4 * stride) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long k = i + u * stride;
            if (k < n4) {
                if (MODE == 0) acc += src[k];
                else if (MODE == 1) dst[k] = f4{1.f, 2.f, 3.f, (float)u};
                else dst[k] = src[k];
            }
        }
    }
    if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// ---------------------------------------------------------------------------------------------------
// From the pure gather towards inagg_fwd3_kernel, one ingredient at a time, on a uniform-degree CSR
// (row r owns edges [deg r, deg (r + 1))): which ingredient costs what.
//   STAGE 1: items -- a wave takes a destination row, loads its col indices, gathers its rows in pairs
//   STAGE 2: + the hoisted softmax chain: per-edge logit gather, a_dst, exp, weights through LDS, the
//              four weighted FMAs per lane and pair
//   STAGE 3: + the row's Z [4][Fp] and lse stored
//   STAGE 4: + item metadata read from arrays (item_row, item_beg, rowptr) instead of computed
// ---------------------------------------------------------------------------------------------------
template <int STAGE, int U, int SM = 0>
__global__ __launch_bounds__(256, 4) void ladder_kernel(const char *__restrict__ rows, const int *__restrict__ col,
                                                        const int *__restrict__ item_row, const int *__restrict__ item_beg,
                                                        const int *__restrict__ rowptr, const float *__restrict__ a_dst,
                                                        int n_items, int deg, float *__restrict__ z, float *__restrict__ lse,
                                                        float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float ws_all[4][64 * 4];
    __shared__ __attribute__((aligned(16))) float zrow_all[4][400];
    const int lane = threadIdx.x & 63, half = lane >> 5, p = lane & 31, wv = threadIdx.x >> 6;
    const int hl = 2 * (lane & 1) + ((lane >> 1) & 1);
    const bool pok = p < 25;
    const unsigned offp = pok ? p * 16u : 0u;
    float *wt = ws_all[wv];
    f4 keep = {0.f, 0.f, 0.f, 0.f};
    const int stride = gridDim.x * 4;
    for (int it = blockIdx.x * 4 + wv; it < n_items; it += stride) {
        int row = it, beg = it * deg, end = beg + deg;
        if (STAGE >= 4) {
            row = __builtin_amdgcn_readfirstlane(item_row[it]);
            beg = __builtin_amdgcn_readfirstlane(item_beg[it]);
            end = __builtin_amdgcn_readfirstlane(rowptr[row + 1]);
        }
        const float adi = STAGE >= 2 ? a_dst[(size_t)row * 4 + hl] : 0.f;
        float m = -INFINITY, ssum = 0.f;
        f4 acc[4];
        for (int h = 0; h < 4; ++h) acc[h] = f4{0.f, 0.f, 0.f, 0.f};
        for (int c = beg; c < end; c += 64) {
            const int n = end - c < 64 ? end - c : 64;
            const int cj = col[c + (lane < n ? lane : n - 1)];
            auto load = [&](f4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = k0 + 2 * u;
                    const int j0 = __builtin_amdgcn_readlane(cj, k < n ? k : n - 1);
                    const int j1 = __builtin_amdgcn_readlane(cj, k + 1 < n ? k + 1 : n - 1);
                    const int j = half ? j1 : j0;
                    v[u] = *reinterpret_cast<const f4 *>(rows + (size_t)j * 416 + offp);
                }
            };
            f4 va[U], vb[U];
            load(va, 0);
            if (STAGE >= 2) {
                float zz[4], zmax = -INFINITY;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int e = 16 * ps + (lane >> 2);
                    zz[ps] = -INFINITY;
                    if (16 * ps < n) {
                        const int js = __shfl(cj, e < n ? e : n - 1, 64);
                        const float uu = *reinterpret_cast<const float *>(rows + (size_t)js * 416 + 400 + hl * 4) + adi;
                        zz[ps] = e < n ? (uu > 0.f ? uu : 0.2f * uu) : -INFINITY;
                    }
                    zmax = fmaxf(zmax, zz[ps]);
                }
                for (int o = 4; o < 64; o <<= 1) zmax = fmaxf(zmax, __shfl_xor(zmax, o, 64));
                const float mn = fmaxf(m, zmax);
                const float sc = __expf(m - mn);
                float psum = 0.f;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps)
                    if (16 * ps < n) {
                        const int e = 16 * ps + (lane >> 2);
                        const float pe = __expf(zz[ps] - mn);
                        psum += pe;
                        wt[e * 4 + hl] = pe;
                    }
                for (int o = 4; o < 64; o <<= 1) psum += __shfl_xor(psum, o, 64);
                ssum = ssum * sc + psum;
                m = mn;
                for (int h = 0; h < 4; ++h) acc[h] *= __shfl(sc, (lane & ~3) | (2 * (h & 1) + (h >> 1)), 64);
            }
            auto reduce = [&](const f4(&v)[U], int k0) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (STAGE >= 2) {
                        const int ke = k0 + 2 * u + half;
                        f4 w4 = *reinterpret_cast<const f4 *>(wt + (ke < n ? ke : n - 1) * 4);
                        if (ke >= n) w4 = f4{0.f, 0.f, 0.f, 0.f};
                        acc[0] += w4[0] * v[u];
                        acc[1] += w4[1] * v[u];
                        acc[2] += w4[2] * v[u];
                        acc[3] += w4[3] * v[u];
                    } else {
                        acc[0] += v[u];
                    }
                }
            };
            for (int k0 = 0; k0 < n; k0 += 4 * U) {
                load(vb, k0 + 2 * U);
                reduce(va, k0);
                load(va, k0 + 4 * U);
                reduce(vb, k0 + 2 * U);
            }
        }
        if (STAGE >= 3) {
            for (int h = 0; h < 4; ++h)
                for (int k = 0; k < 4; ++k) acc[h][k] += __shfl_xor(acc[h][k], 32, 64);
            const float inv = ssum > 0.f ? 1.0f / ssum : 0.f;
            // SM: 0 plain stores, 1 nontemporal, 2 rows padded to 13 whole lines (pitch 416 floats),
            //     3 both, 4 two store instructions (heads 0,1 by half 0; 2,3 by half 1), 5 = 4 + nontemporal + padded
            const size_t zp = (SM == 2 || SM == 3 || SM == 5 || SM == 7) ? 416 : 400;
            if (SM >= 6) {
                // through LDS: the row leaves as two full-wave instructions of contiguous float4 (whole lines
                // when the pitch is 13 lines)
                float *zb = zrow_all[wv];
                if (half == 0 && pok)
                    for (int h = 0; h < 4; ++h)
                        *reinterpret_cast<f4 *>(zb + h * 100 + p * 4) =
                            acc[h] * __shfl(inv, (lane & ~3) | (2 * (h & 1) + (h >> 1)), 64);
                __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
                const f4 o0 = *reinterpret_cast<const f4 *>(zb + lane * 4);
                const f4 o1 = *reinterpret_cast<const f4 *>(zb + (lane < 36 ? 256 + lane * 4 : 0));
                *reinterpret_cast<f4 *>(z + (size_t)row * zp + lane * 4) = o0;
                if (lane < 36) *reinterpret_cast<f4 *>(z + (size_t)row * zp + 256 + lane * 4) = o1;
            } else if (SM >= 4) {
                if (pok)
                    for (int hh = 0; hh < 2; ++hh) {
                        const int h = 2 * half + hh;
                        const f4 o = (half ? acc[2 + hh] : acc[hh]) * __shfl(inv, (lane & ~3) | (2 * (h & 1) + (h >> 1)), 64);
                        f4 *dst = reinterpret_cast<f4 *>(z + (size_t)row * zp + h * 100 + p * 4);
                        if (SM == 5) __builtin_nontemporal_store(o, dst); else *dst = o;
                    }
            } else if (half == 0 && pok)
                for (int h = 0; h < 4; ++h) {
                    const f4 o = acc[h] * __shfl(inv, (lane & ~3) | (2 * (h & 1) + (h >> 1)), 64);
                    f4 *dst = reinterpret_cast<f4 *>(z + (size_t)row * zp + h * 100 + p * 4);
                    if (SM == 1 || SM == 3) __builtin_nontemporal_store(o, dst); else *dst = o;
                }
            if (lane < 4) lse[(size_t)row * 4 + hl] = m + __logf(ssum);
        } else {
            for (int h = 0; h < 4; ++h) keep += acc[h];
            keep[0] += ssum;
        }
    }
    if (keep[0] + keep[1] + keep[2] + keep[3] == 12345.678f) out[0] = keep[0];
}

int main() {
    const long long N = 2449029, E = 64449007;
    std::vector<int> h_idx(E);
    unsigned long long s = 88172645463325252ull;
    for (long long i = 0; i < E; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        h_idx[i] = (int)(s % (unsigned long long)N);
    }
    int *idx; char *rows, *tail; float *out;
    CHECK(hipMalloc(&idx, E * 4));
    CHECK(hipMalloc(&rows, (size_t)N * 1024 + 1024));
    CHECK(hipMalloc(&tail, (size_t)N * 32 + 32));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemcpy(idx, h_idx.data(), E * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(rows, 0, (size_t)N * 1024 + 1024));
    CHECK(hipMemset(tail, 0, (size_t)N * 32 + 32));
    struct Case { const char *name; long long pitch; int lanes; bool tail; };
    const Case cases[] = {
        {"416-byte rows (today: x 400 B | logits 16 B), 26 lanes", 416, 26, false},
        {"400-byte rows, 25 lanes", 400, 25, false},
        {"384-byte rows (three lines), 24 lanes", 384, 24, false},
        {"384-byte rows + 16 B from a 32-byte-entry tail table", 384, 24, true},
        {"512-byte rows (four lines), 32 lanes", 512, 32, false},
        {"256-byte rows (two lines), 16 lanes", 256, 16, false},
        {"256-byte rows + 16 B tail", 256, 16, true},
        {"128-byte rows (one line), 8 lanes", 128, 8, false},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto run = [&](const Case &cs, int variant, float *best_out) -> int {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            const char *tl = cs.tail ? tail : nullptr;
#define GB_LAUNCH(U_, OCC_, GRID_) hipLaunchKernelGGL((gather_kernel<U_, OCC_>), dim3(GRID_), dim3(256), 0, 0, rows, cs.pitch, cs.lanes, tl, idx, E, out)
            switch (variant) {
                case 0: GB_LAUNCH(4, 4, 2048); break;
                case 1: GB_LAUNCH(8, 4, 2048); break;
                case 2: GB_LAUNCH(2, 8, 4096); break;
                case 3: GB_LAUNCH(4, 8, 4096); break;
                case 4: GB_LAUNCH(2, 4, 2048); break;
                case 5: GB_LAUNCH(1, 4, 2048); break;
                case 6: GB_LAUNCH(1, 8, 4096); break;
                case 7: GB_LAUNCH(3, 4, 2048); break;
                case 8: GB_LAUNCH(2, 2, 1024); break;
                default: GB_LAUNCH(4, 2, 1024); break;
            }
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        *best_out = best;
        return 0;
    };
    const char *vname[] = {"U=4 occ 4", "U=8 occ 4", "U=2 occ 8", "U=4 occ 8", "U=2 occ 4", "U=1 occ 4", "U=1 occ 8", "U=3 occ 4", "U=2 occ 2", "U=4 occ 2"};
    for (const Case &cs : cases) {
        for (int v = 0; v < 10; ++v) {
            if (v > 0 && v != 4) continue;
            float best;
            if (run(cs, v, &best)) return 1;
            const double useful = (double)E * (cs.lanes * 16 + (cs.tail ? 16 : 0) + 4);
            printf("%-58s %-10s %7.3f ms  %6.2f G edges/s  useful %5.2f TB/s\n", cs.name, vname[v], best,
                   (double)E / best / 1e6, useful / best / 1e9);
            fflush(stdout);
        }
    }
    for (int uu = 0; uu < 3; ++uu) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (uu == 0) hipLaunchKernelGGL((gather1k_kernel<2>), dim3(2048), dim3(256), 0, 0, rows, idx, E, out);
            else if (uu == 1) hipLaunchKernelGGL((gather1k_kernel<4>), dim3(2048), dim3(256), 0, 0, rows, idx, E, out);
            else hipLaunchKernelGGL((gather1k_kernel<8>), dim3(2048), dim3(256), 0, 0, rows, idx, E, out);
            CHECK(hipGetLastError());
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("1024-byte rows (eight lines), one row per instruction, U=%d  %7.3f ms  %6.2f G edges/s  useful %5.2f TB/s\n",
               uu == 0 ? 2 : (uu == 1 ? 4 : 8), best, (double)E / best / 1e6, (double)E * 1028 / best / 1e9);
        fflush(stdout);
    }
    {
        const int deg = 26, n_items = (int)(E / deg);
        float *dst;
        CHECK(hipMalloc(&dst, (size_t)n_items * 1024 + 1024));
        const char *nm[] = {"plain 1-KiB row stores", "nontemporal stores", "only the first line of every row stored", "no stores",
                            "row stored behind the next item's first loads"};
        for (int sm = 0; sm < 5; ++sm) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (sm == 0) hipLaunchKernelGGL((gather1k_store_kernel<4, 0>), dim3(2048), dim3(256), 0, 0, rows, idx, n_items, deg, dst, out);
                else if (sm == 1) hipLaunchKernelGGL((gather1k_store_kernel<4, 1>), dim3(2048), dim3(256), 0, 0, rows, idx, n_items, deg, dst, out);
                else if (sm == 2) hipLaunchKernelGGL((gather1k_store_kernel<4, 2>), dim3(2048), dim3(256), 0, 0, rows, idx, n_items, deg, dst, out);
                else if (sm == 3) hipLaunchKernelGGL((gather1k_store_kernel<4, 3>), dim3(2048), dim3(256), 0, 0, rows, idx, n_items, deg, dst, out);
                else hipLaunchKernelGGL((gather1k_store_kernel<4, 4>), dim3(2048), dim3(256), 0, 0, rows, idx, n_items, deg, dst, out);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("1024-byte rows, 26 per item, %-42s %7.3f ms\n", nm[sm], best);
            fflush(stdout);
        }
        CHECK(hipFree(dst));
    }
    {
        const long long bytes = 3900000000ll, n4 = bytes / 16;
        f4 *a, *b;
        CHECK(hipMalloc(&a, bytes));
        CHECK(hipMalloc(&b, bytes));
        CHECK(hipMemset(a, 0, bytes));
        const char *nm[] = {"read 3.9 GB", "write 3.9 GB", "copy 3.9 GB (read + write)"};
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (mode == 0) hipLaunchKernelGGL((stream_kernel<0>), dim3(8192), dim3(256), 0, 0, a, b, n4, out);
                else if (mode == 1) hipLaunchKernelGGL((stream_kernel<1>), dim3(8192), dim3(256), 0, 0, a, b, n4, out);
                else hipLaunchKernelGGL((stream_kernel<2>), dim3(8192), dim3(256), 0, 0, a, b, n4, out);
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("stream: %-30s %7.3f ms  %5.2f TB/s\n", nm[mode], best, (mode == 2 ? 2.0 : 1.0) * bytes / best / 1e9);
            fflush(stdout);
        }
        CHECK(hipFree(a));
        CHECK(hipFree(b));
    }
    // ---- the ladder on a uniform-degree CSR (26 edges per destination row)
    {
        const int deg = 26;
        const int n_items = (int)(E / deg);
        std::vector<int> h_item(n_items), h_beg(n_items), h_ptr(n_items + 1);
        for (int i = 0; i < n_items; ++i) { h_item[i] = i; h_beg[i] = i * deg; h_ptr[i] = i * deg; }
        h_ptr[n_items] = n_items * deg;
        int *item_row, *item_beg, *rowptr; float *a_dst, *z, *lse;
        CHECK(hipMalloc(&item_row, (size_t)n_items * 4));
        CHECK(hipMalloc(&item_beg, (size_t)n_items * 4));
        CHECK(hipMalloc(&rowptr, (size_t)(n_items + 1) * 4));
        CHECK(hipMalloc(&a_dst, (size_t)n_items * 16));
        CHECK(hipMalloc(&z, (size_t)n_items * 1664 + 1664));
        CHECK(hipMalloc(&lse, (size_t)n_items * 16));
        CHECK(hipMemcpy(item_row, h_item.data(), (size_t)n_items * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(item_beg, h_beg.data(), (size_t)n_items * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(rowptr, h_ptr.data(), (size_t)(n_items + 1) * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(a_dst, 0, (size_t)n_items * 16));
        const char *names[] = {"", "1: items (col load + paired row gathers)", "2: + hoisted chain, weights via LDS, 4 FMAs",
                               "3: + Z / lse stores", "4: + item metadata from arrays"};
        for (int stage = 1; stage <= 4; ++stage)
            for (int uu = 0; uu < 2; ++uu) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(e0, 0));
#define GB_LADDER(ST_, U_) hipLaunchKernelGGL((ladder_kernel<ST_, U_>), dim3(2048), dim3(256), 0, 0, rows, idx, item_row, item_beg, rowptr, a_dst, n_items, deg, z, lse, out)
                    if (uu == 0) { if (stage == 1) GB_LADDER(1, 4); else if (stage == 2) GB_LADDER(2, 4); else if (stage == 3) GB_LADDER(3, 4); else GB_LADDER(4, 4); }
                    else { if (stage == 1) GB_LADDER(1, 2); else if (stage == 2) GB_LADDER(2, 2); else if (stage == 3) GB_LADDER(3, 2); else GB_LADDER(4, 2); }
                    CHECK(hipGetLastError());
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                printf("ladder %-48s U=%d  %7.3f ms\n", names[stage], uu == 0 ? 4 : 2, best);
                fflush(stdout);
            }
    }
    {
        const int deg = 26;
        const int n_items = (int)(E / deg);
        int *item_row = nullptr, *item_beg = nullptr, *rowptr = nullptr; float *a_dst, *z, *lse;
        CHECK(hipMalloc(&a_dst, (size_t)n_items * 16));
        CHECK(hipMalloc(&z, (size_t)n_items * 1664 + 1664));
        CHECK(hipMalloc(&lse, (size_t)n_items * 16));
        CHECK(hipMemset(a_dst, 0, (size_t)n_items * 16));
        const char *names[] = {"plain stores", "nontemporal stores", "rows padded to 13 lines", "nontemporal + padded",
                               "two store instructions (both halves)", "two instructions + nontemporal + padded",
                               "row through LDS, two full-wave stores", "through LDS + rows padded to 13 lines"};
        for (int sm = 0; sm < 8; ++sm) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
#define GB_STORE(SM_) hipLaunchKernelGGL((ladder_kernel<3, 4, SM_>), dim3(2048), dim3(256), 0, 0, rows, idx, item_row, item_beg, rowptr, a_dst, n_items, deg, z, lse, out)
                switch (sm) { case 0: GB_STORE(0); break; case 1: GB_STORE(1); break; case 2: GB_STORE(2); break;
                              case 3: GB_STORE(3); break; case 4: GB_STORE(4); break; case 5: GB_STORE(5); break;
                              case 6: GB_STORE(6); break; default: GB_STORE(7); break; }
                CHECK(hipGetLastError());
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            printf("stage 3 with %-45s %7.3f ms\n", names[sm], best);
            fflush(stdout);
        }
    }
    // the same 416-byte rows picked in ascending order (every line used once, DRAM pages in order):
    // what the memory system gives when locality is perfect
    {
        std::vector<int> sorted(E);
        for (long long i = 0; i < E; ++i) sorted[i] = (int)(i * N / E);
        CHECK(hipMemcpy(idx, sorted.data(), E * 4, hipMemcpyHostToDevice));
        float best;
        if (run(cases[0], 0, &best)) return 1;
        printf("%-58s %-10s %7.3f ms  (ascending indices: each row read ~26 times in a row)\n", cases[0].name, vname[0], best);
    }
    return 0;
}

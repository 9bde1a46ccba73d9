// Device-side building blocks of the fused attention kernels (gfx950, wave64).
//
// Lane mapping.  A destination (or source) row is owned by a GROUP of G lanes
// (G = 64: one wave per row; G < 64: 64/G rows per wave, for narrow layers).  Inside a
// group, lane `lig` owns T float4 "slots": slot = t*G + lig -> (head, sub) with
// head = slot / LPH2, sub = slot % LPH2, LPH2 = next_pow2(Cp/4) lanes per head; slots
// with sub >= Cp/4 or head >= H idle.  Per-head sums are butterfly reductions over the
// LPH2 lanes of a head: DPP inside a 16-lane row, permute across rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gatmi {

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// Sum over the LPH2 consecutive lanes that hold one head; every lane gets the total.
// LPH2_T == 0: width only known at run time (generic kernels).
template <int LPH2_T> __device__ __forceinline__ float head_sum(float v, int lph2) {
    if constexpr (LPH2_T == 0) {
        for (int o = 1; o < lph2; o <<= 1) v += __shfl_xor(v, o, 64);
    } else {
        if constexpr (LPH2_T >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
        if constexpr (LPH2_T >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
        if constexpr (LPH2_T >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
        if constexpr (LPH2_T >= 16) v += dpp_mov<0x140>(v);  // row_mirror
        if constexpr (LPH2_T >= 32) v += __shfl_xor(v, 16, 64);
        if constexpr (LPH2_T >= 64) v += __shfl_xor(v, 32, 64);
    }
    return v;
}

// Value held by lane k of this lane's group.  G == 64: k is wave-uniform -> v_readlane.
template <int G> __device__ __forceinline__ int group_bcast(int v, int k, int gbase) {
    if constexpr (G == 64)
        return __builtin_amdgcn_readlane(v, k);
    else
        return __shfl(v, gbase + k, 64);
}

template <int G> __device__ __forceinline__ int uniform(int v) {
    if constexpr (G == 64)
        return __builtin_amdgcn_readfirstlane(v);
    else
        return v;
}

__device__ __forceinline__ float dot4(const float4 &a, const float4 &b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ void fma4(float4 &acc, float s, const float4 &v) {
    acc.x = fmaf(s, v.x, acc.x);
    acc.y = fmaf(s, v.y, acc.y);
    acc.z = fmaf(s, v.z, acc.z);
    acc.w = fmaf(s, v.w, acc.w);
}
__device__ __forceinline__ void scale4(float4 &acc, float s) {
    acc.x *= s; acc.y *= s; acc.z *= s; acc.w *= s;
}

__device__ __forceinline__ float4 ld4(const float *p) {
    return *reinterpret_cast<const float4 *>(p);
}
__device__ __forceinline__ void st4(float *p, const float4 &v) {
    *reinterpret_cast<float4 *>(p) = v;
}

// Load/store of up to 4 floats of a COMPACT [.., H*C] row (heads not 16-byte aligned
// unless C % 4 == 0).  nval = number of real channels in this slot (0..4).
__device__ __forceinline__ float4 ldc(const float *p, int nval, bool vec_ok) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok) {
        if (nval) r = ld4(p);
    } else {
        if (nval > 0) r.x = p[0];
        if (nval > 1) r.y = p[1];
        if (nval > 2) r.z = p[2];
        if (nval > 3) r.w = p[3];
    }
    return r;
}
__device__ __forceinline__ void stc(float *p, const float4 &v, int nval, bool vec_ok) {
    if (vec_ok) {
        if (nval) st4(p, v);
    } else {
        if (nval > 0) p[0] = v.x;
        if (nval > 1) p[1] = v.y;
        if (nval > 2) p[2] = v.z;
        if (nval > 3) p[3] = v.w;
    }
}

// Attention-dropout keep factor of (edge e in dst-CSR order, head): 0 or 1/(1-p).
// GATMI_MASK_HASH: a counter-based draw, the same in forward and backward.
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float keep_factor(int mode, const uint8_t *mask, uint64_t seed,
                                             float p, float inv_keep, long long e, int heads,
                                             int head) {
    if (mode == 0) return 1.0f;
    const long long idx = e * heads + head;
    if (mode == 1) return mask[idx] ? inv_keep : 0.0f;
    uint32_t h = mix32((uint32_t)idx ^ (uint32_t)seed);
    h = mix32(h + (uint32_t)(idx >> 32) * 0x9E3779B9u + (uint32_t)(seed >> 32));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);  // [0,1)
    return u >= p ? inv_keep : 0.0f;
}

// blockIdx -> work position such that blocks sharing an XCD (equal blockIdx % 8)
// walk a contiguous range of work items.  gridDim.x is a multiple of 8.
__device__ __forceinline__ int xcd_contiguous_block() {
    const int per = gridDim.x >> 3;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// Per-lane slot description for T tiles.
template <int T> struct Slots {
    int head[T];
    int poff[T];  // float offset inside a head-padded row
    int coff[T];  // float offset inside a compact row
    int nval[T];  // real channels in the slot, 0 = idle
};

template <int T, int G>
__device__ __forceinline__ Slots<T> make_slots(int lig, int H, int C, int Cp, int lph2_log) {
    Slots<T> s;
    const int lph = Cp >> 2;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int slot = t * G + lig;
        const int head = slot >> lph2_log;
        const int sub = slot & ((1 << lph2_log) - 1);
        const bool ok = head < H && sub < lph;
        s.head[t] = ok ? head : 0;
        s.poff[t] = ok ? head * Cp + sub * 4 : 0;
        s.coff[t] = ok ? head * C + sub * 4 : 0;
        int nv = C - sub * 4;
        nv = nv > 4 ? 4 : nv;
        s.nval[t] = ok ? nv : 0;
    }
    return s;
}

}  // namespace gatmi

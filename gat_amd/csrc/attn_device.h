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

// A load through the constant address space: the graph layout is never written while a kernel runs, and with
// a scalar index this is a scalar load whatever the kernel stores elsewhere (as a plain global load the
// compiler has to keep it a vector load once the loop holds a store that might alias)
typedef const int __attribute__((address_space(4))) *layout_ptr;
__device__ __forceinline__ int ld_layout(const int *p, int i) { return ((layout_ptr)(p))[i]; }

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

// a float4 at any 4-byte boundary (global loads and stores need no more)
__device__ __forceinline__ float4 ld4_unaligned(const float *p) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    const f4u r = *reinterpret_cast<const f4u *>(p);
    return make_float4(r[0], r[1], r[2], r[3]);
}
__device__ __forceinline__ void st4_unaligned(float *p, const float4 &v) {
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
    *reinterpret_cast<f4u *>(p) = f4u{v.x, v.y, v.z, v.w};
}

// Load/store of up to 4 floats of a COMPACT [.., H*C] row (heads not 16-byte aligned
// unless C % 4 == 0).  nval = number of real channels in this slot (0..4).  Whole slots of unaligned
// heads move as one 16-byte access all the same (C = 30: seven lanes of a head; the eighth two floats).
__device__ __forceinline__ float4 ldc(const float *p, int nval, bool vec_ok) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec_ok) {
        if (nval) r = ld4(p);
    } else if (nval == 4) {
        r = ld4_unaligned(p);
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
    } else if (nval == 4) {
        st4_unaligned(p, v);
    } else {
        if (nval > 0) p[0] = v.x;
        if (nval > 1) p[1] = v.y;
        if (nval > 2) p[2] = v.z;
        if (nval > 3) p[3] = v.w;
    }
}

// exp(x) through v_exp_f32 with a compensated range reduction: t = x*log2(e) rounded,
// r = the rounding error of that product (+ the low word of log2(e)), exp(x) =
// 2^t * (1 + r ln 2).  About 1.5 ulp for x in [-87, 0].  x is clamped at -87 (the result,
// 1.6e-38, is 0 against a softmax denominator >= 1), which also keeps -inf from turning
// into inf - inf in the compensation term.
__device__ __forceinline__ float exp_fast(float x) {
    x = fmaxf(x, -87.0f);
    const float L2E = 1.44269502162933349609375f;      // float(log2 e)
    const float L2E_LO = 1.925963033500011e-8f;        // log2 e - float(log2 e)
    const float t = x * L2E;
    float r = fmaf(x, L2E, -t);
    r = fmaf(x, L2E_LO, r);
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.693147182464599609375f, e);
}

// exp(x) for x <= 0 as v_exp_f32(x * log2 e), no compensation: the product's rounding leaves a
// relative error of |x| * 4e-8 (6e-7 at x = -15, where the weight itself is 3e-7 of the row's
// largest).  Two instructions instead of seven; used where the exponential sits in the per-edge
// loop of a VALU-bound kernel.
__device__ __forceinline__ float exp_quick(float x) {   // x = -inf gives 0 (no clamp needed)
#ifdef GATMI_EXACT_EXP
    return exp_fast(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269502162933349609375f);
#endif
}

// Attention-dropout keep decision of (edge e in dst-CSR order, head): a counter-based
// hash of (seed, e, head), identical in forward, backward and gatmi_dropout_mask.
__device__ __forceinline__ bool keep_hash(uint32_t seed_lo, uint32_t seed_hi, uint32_t thr,
                                          uint32_t e, uint32_t head) {
    uint32_t h = e * 0x9E3779B1u + seed_lo;
    h ^= head * 0x85EBCA77u + seed_hi;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return (h >> 8) >= thr;   // thr = p * 2^24
}

// GATMI_MASK_HASH_DEV: the seed lives in device memory (`mask` points at it), so that a captured graph
// replays with new masks.  Read ONCE at the top of a kernel: inside keep_bits the load sat in the edge
// loops, and its wait drained the LDS-DMA queue of the streaming forward (6.68 -> 7.31 ms).
struct KeepSel { int mode; uint64_t seed; };
__device__ __forceinline__ KeepSel keep_sel(int mode, const uint8_t *mask, uint64_t seed) {
    if (mode == 3) return KeepSel{2, *reinterpret_cast<const uint64_t *>(mask)};
    return KeepSel{mode, seed};
}

// Keep flags of all H (<= 32) heads of one edge, bit h = keep.  Computed lane-parallel
// (one edge per lane) once per chunk; the per-edge loop then only tests a bit.
// `mode` and `seed` are what keep_sel returned (GATMI_MASK_HASH_DEV resolved to HASH once per kernel).
__device__ __forceinline__ uint32_t keep_bits(int mode, const uint8_t *mask, uint64_t seed,
                                              uint32_t thr, long long e, int heads) {
    if (mode == 0) return 0xFFFFFFFFu;
    uint32_t bits = 0;
    if (mode == 1) {
        const uint8_t *m = mask + e * heads;
        for (int h = 0; h < heads; ++h) bits |= (m[h] ? 1u : 0u) << h;
    } else {
        for (int h = 0; h < heads; ++h)
            bits |= (keep_hash((uint32_t)seed, (uint32_t)(seed >> 32), thr, (uint32_t)e, h) ? 1u : 0u) << h;
    }
    return bits;
}

// Row j of a head-padded table, this lane's 16 bytes.  G == 64: j is wave-uniform, the
// row base lives in SGPRs as a buffer descriptor (scalar address arithmetic only) and the
// lane offset is a loop-invariant VGPR.  G < 64: j differs per group -> plain global load.
template <int G>
__device__ __forceinline__ float4 ld4_row(const float *table, int j, int row_floats,
                                          unsigned off_bytes) {
    const char *rowp = reinterpret_cast<const char *>(table) + (size_t)j * (size_t)row_floats * 4;
    if constexpr (G == 64) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(rowp), 0, row_floats * 4, 0x00020000);
        const f32x4 r = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off_bytes, 0, 0));
        return make_float4(r[0], r[1], r[2], r[3]);
    } else {
        return *reinterpret_cast<const float4 *>(rowp + off_bytes);
    }
}

// ---- bf16 feature tables (gatmi_attn.feat_bf16): the same head-padded rows at 2 bytes per
// channel.  A lane's slot is 4 channels = 8 bytes; values are widened to f32 in registers and all
// arithmetic stays f32.
struct bf16x4 { uint32_t lo, hi; };   // channels (0,1) and (2,3), low half first

__device__ __forceinline__ float4 widen(const bf16x4 &r) {
    return make_float4(__uint_as_float(r.lo << 16), __uint_as_float(r.lo & 0xffff0000u),
                       __uint_as_float(r.hi << 16), __uint_as_float(r.hi & 0xffff0000u));
}
__device__ __forceinline__ float4 widen(const float4 &r) { return r; }

// two f32 -> packed bf16, round to nearest even (what torch's .to(torch.bfloat16) does)
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ bf16x4 narrow(const float4 &v) {
    return bf16x4{pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)};
}
// v rounded to bf16 and widened again: the value a bf16 table hands back
__device__ __forceinline__ float4 round_bf16(const float4 &v) { return widen(narrow(v)); }

// This lane's slot of row j of a table of type RAW (float4: f32 table, bf16x4: bf16 table);
// off_bytes is the slot's byte offset in an f32 row.  See ld4_row.
template <int G, class RAW>
__device__ __forceinline__ RAW ld_slot(const void *table, int j, int row_floats, unsigned off_bytes) {
    if constexpr (sizeof(RAW) == 16) {
        return ld4_row<G>(static_cast<const float *>(table), j, row_floats, off_bytes);
    } else {
        const char *rowp = static_cast<const char *>(table) + (size_t)j * (size_t)row_floats * 2;
        if constexpr (G == 64) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(rowp), 0, row_floats * 2, 0x00020000);
            const u32x2 r = __builtin_bit_cast(
                u32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(off_bytes >> 1), 0, 0));
            return bf16x4{r[0], r[1]};
        } else {
            const uint2 r = *reinterpret_cast<const uint2 *>(rowp + (off_bytes >> 1));
            return bf16x4{r.x, r.y};
        }
    }
}
// the same slot of a row this lane owns itself (plain load)
template <class RAW>
__device__ __forceinline__ float4 ld_own(const void *table, size_t row, int row_floats, int poff) {
    if constexpr (sizeof(RAW) == 16) {
        return ld4(static_cast<const float *>(table) + row * row_floats + poff);
    } else {
        const uint2 r = *reinterpret_cast<const uint2 *>(
            static_cast<const uint16_t *>(table) + row * row_floats + poff);
        return widen(bf16x4{r.x, r.y});
    }
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

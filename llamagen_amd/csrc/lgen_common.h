// Common device helpers for the lgen HIP library (gfx950 / CDNA4 only).
//
// Storage dtypes: BF16 (the reference's default --precision bf16) and F32
// (--precision none, the bit-exact-token parity mode).  Arithmetic is always fp32 with
// the reference's storage rounding points reproduced explicitly.
//
// Fragment-packed layouts (DESIGN.md "data layout in HBM"):
//   one "chunk" = what one wave-wide 16-byte-per-lane load brings in = 1 KiB =
//   a [16 rows x KC k] tile in MFMA operand order: lane = g*16 + r holds row r,
//   k-slice [g*EPL, g*EPL+EPL).  KC = 32 / EPL = 8 for bf16 (mfma_f32_16x16x32_bf16),
//   KC = 16 / EPL = 4 for fp32 (4 x mfma_f32_16x16x4_f32, k-slot j <-> element j).
//   weights      WP[nt][kc][lane][EPL]   (nt = 16-row tile of N)
//   activations  XP[kc][mt][lane][EPL]   (mt = 16-row tile of M, M padded to MT*16)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define LGEN_DEV __device__ __forceinline__

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

LGEN_DEV float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// fp32 -> bf16, round-to-nearest-even (NaN stays NaN): the hardware v_cvt_pk_bf16_f32 of gfx950
LGEN_DEV uint32_t f2bf2(float lo, float hi) {  // packed pair, lo in bits [15:0]
    f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
LGEN_DEV uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
LGEN_DEV float rnd_bf(float f) { return bf2f(f2bf(f)); }

struct BF16 {
    static constexpr int EPL = 8;    // elements per lane per 16-byte load
    static constexpr int KC = 32;    // k elements per chunk
    static constexpr int ESZ = 2;
    static constexpr int CODE = 0;
    typedef uint16_t elem_t;
    LGEN_DEV static float rnd(float f) { return rnd_bf(f); }
    LGEN_DEV static float fexp(float x) { return __expf(x); }  // softmax weights: bf16 output hides the 2-ulp fast exp
    LGEN_DEV static float ld(const void* p, size_t i) { return bf2f(((const uint16_t*)p)[i]); }
    LGEN_DEV static void st(void* p, size_t i, float v) { ((uint16_t*)p)[i] = f2bf(v); }
    // unpack the EPL values of one lane's 16 bytes
    LGEN_DEV static void unpack(const uint4& u, float (&f)[8]) {
        f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
        f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
        f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
    }
    LGEN_DEV static uint4 pack(const float (&f)[8]) {
        uint4 u;
        u.x = f2bf2(f[0], f[1]);
        u.y = f2bf2(f[2], f[3]);
        u.z = f2bf2(f[4], f[5]);
        u.w = f2bf2(f[6], f[7]);
        return u;
    }
    // store 4 consecutive elements (8 bytes)
    LGEN_DEV static void st4(void* p, size_t i, float a, float b, float c, float d) {
        uint2 u;
        u.x = f2bf2(a, b);
        u.y = f2bf2(c, d);
        *(uint2*)((uint16_t*)p + i) = u;
    }
    LGEN_DEV static void ld4(const void* p, size_t i, float& a, float& b, float& c, float& d) {
        uint2 u = *(const uint2*)((const uint16_t*)p + i);
        a = __uint_as_float(u.x << 16); b = __uint_as_float(u.x & 0xffff0000u);
        c = __uint_as_float(u.y << 16); d = __uint_as_float(u.y & 0xffff0000u);
    }
    LGEN_DEV static void unpack4(const uint2& u, float& a, float& b, float& c, float& d) {
        a = __uint_as_float(u.x << 16); b = __uint_as_float(u.x & 0xffff0000u);
        c = __uint_as_float(u.y << 16); d = __uint_as_float(u.y & 0xffff0000u);
    }
    LGEN_DEV static void ld8(const void* p, size_t i, float (&f)[8]) {  // 8 consecutive elements, 16-byte aligned
        unpack(*(const uint4*)((const uint16_t*)p + i), f);
    }
    // element offset inside a packed activation buffer of (m-tile mt, lane-row r, k index)
    LGEN_DEV static size_t xp_off(int k, int mt, int r, int MTs) {
        return ((size_t)((k >> 5) * MTs + mt) * 64 + ((k >> 3) & 3) * 16 + r) * 8 + (k & 7);
    }
    // RMSNorm of one operand chunk (gpt.py:143-148): rnd(rnd(x * rinv) * w), 8 elements
    LGEN_DEV static uint4 norm_chunk(const uint4& x, float ri, const uint4& w) {
        float f[8], wf[8];
        unpack(x, f);
        unpack(w, wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * ri;
        const uint4 t = pack(f);
        unpack(t, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * wf[e];
        return pack(f);
    }
    // sum of the 8 products of two packed operands, fp32 accumulation (v_dot2c_f32_bf16 x 4; bf16 x bf16 products are exact in fp32)
    LGEN_DEV static float dot8(const uint4& a, const uint4& b) {
        float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), 0.f, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), d, false);
        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), d, false);
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), d, false);
    }
    LGEN_DEV static f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                        __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};

struct F32 {
    static constexpr int EPL = 4;
    static constexpr int KC = 16;
    static constexpr int ESZ = 4;
    static constexpr int CODE = 1;
    typedef float elem_t;
    LGEN_DEV static float rnd(float f) { return f; }
    LGEN_DEV static float fexp(float x) { return expf(x); }
    LGEN_DEV static float ld(const void* p, size_t i) { return ((const float*)p)[i]; }
    LGEN_DEV static void st(void* p, size_t i, float v) { ((float*)p)[i] = v; }
    LGEN_DEV static void unpack(const uint4& u, float (&f)[4]) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y);
        f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    LGEN_DEV static uint4 pack(const float (&f)[4]) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
    LGEN_DEV static void st4(void* p, size_t i, float a, float b, float c, float d) {
        *(float4*)((float*)p + i) = make_float4(a, b, c, d);
    }
    LGEN_DEV static void ld4(const void* p, size_t i, float& a, float& b, float& c, float& d) {
        float4 v = *(const float4*)((const float*)p + i);
        a = v.x; b = v.y; c = v.z; d = v.w;
    }
    LGEN_DEV static void ld8(const void* p, size_t i, float (&f)[8]) {
        const float4 a = *(const float4*)((const float*)p + i), b = *(const float4*)((const float*)p + i + 4);
        f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
    }
    LGEN_DEV static size_t xp_off(int k, int mt, int r, int MTs) {
        return ((size_t)((k >> 4) * MTs + mt) * 64 + ((k >> 2) & 3) * 16 + r) * 4 + (k & 3);
    }
    LGEN_DEV static uint4 norm_chunk(const uint4& x, float ri, const uint4& w) {
        float f[4], wf[4];
        unpack(x, f);
        unpack(w, wf);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = (f[e] * ri) * wf[e];
        return pack(f);
    }
    LGEN_DEV static f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
        return c;
    }
};


// fp16 storage (the reference's --precision fp16, sample_c2i.py:108): same layouts as BF16 (2-byte elements, KC = 32, EPL = 8),
// IEEE half rounding points, v_mfma_f32_16x16x32_f16 with fp32 accumulation.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
struct F16 {
    static constexpr int EPL = 8;
    static constexpr int KC = 32;
    static constexpr int ESZ = 2;
    static constexpr int CODE = 2;
    typedef uint16_t elem_t;
    LGEN_DEV static float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
    LGEN_DEV static uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
    LGEN_DEV static float rnd(float f) { return h2f(f2h(f)); }
    LGEN_DEV static float fexp(float x) { return __expf(x); }
    LGEN_DEV static float ld(const void* p, size_t i) { return h2f(((const uint16_t*)p)[i]); }
    LGEN_DEV static void st(void* p, size_t i, float v) { ((uint16_t*)p)[i] = f2h(v); }
    LGEN_DEV static void unpack(const uint4& u, float (&f)[8]) {
        const f32x8_t v = __builtin_convertvector(__builtin_bit_cast(f16x8_t, u), f32x8_t);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = v[e];
    }
    LGEN_DEV static uint4 pack(const float (&f)[8]) {
        f32x8_t v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = f[e];
        return __builtin_bit_cast(uint4, __builtin_convertvector(v, f16x8_t));
    }
    LGEN_DEV static void st4(void* p, size_t i, float a, float b, float c, float d) {
        const f32x4_t v = {a, b, c, d};
        *(uint2*)((uint16_t*)p + i) = __builtin_bit_cast(uint2, __builtin_convertvector(v, f16x4_t));
    }
    LGEN_DEV static void ld4(const void* p, size_t i, float& a, float& b, float& c, float& d) {
        const uint2 u = *(const uint2*)((const uint16_t*)p + i);
        const f32x4_t v = __builtin_convertvector(__builtin_bit_cast(f16x4_t, u), f32x4_t);
        a = v[0]; b = v[1]; c = v[2]; d = v[3];
    }
    LGEN_DEV static void unpack4(const uint2& u, float& a, float& b, float& c, float& d) {
        const f32x4_t v = __builtin_convertvector(__builtin_bit_cast(f16x4_t, u), f32x4_t);
        a = v[0]; b = v[1]; c = v[2]; d = v[3];
    }
    LGEN_DEV static void ld8(const void* p, size_t i, float (&f)[8]) { unpack(*(const uint4*)((const uint16_t*)p + i), f); }
    LGEN_DEV static size_t xp_off(int k, int mt, int r, int MTs) { return BF16::xp_off(k, mt, r, MTs); }
    LGEN_DEV static uint4 norm_chunk(const uint4& x, float ri, const uint4& w) {
        float f[8], wf[8];
        unpack(x, f);
        unpack(w, wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * ri;
        const uint4 t = pack(f);
        unpack(t, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = f[e] * wf[e];
        return pack(f);
    }
    LGEN_DEV static float dot8(const uint4& a, const uint4& b) {   // v_dot2_f32_f16 x 4
        typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
        float d = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.x), __builtin_bit_cast(f16x2_t, b.x), 0.f, false);
        d = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.y), __builtin_bit_cast(f16x2_t, b.y), d, false);
        d = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.z), __builtin_bit_cast(f16x2_t, b.z), d, false);
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.w), __builtin_bit_cast(f16x2_t, b.w), d, false);
    }
    LGEN_DEV static f32x4_t mma(const uint4& a, const uint4& b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

// 2-D RoPE rotation of one interleaved (even, odd) pair (gpt.py:420-430): fp32, four separately rounded products, one add, one sub.
// Kept SCALAR on purpose.  Left to itself hipcc's SLP vectoriser pairs the products into packed-fp32 instructions with CROSSED operand
// selection (`v_pk_mul_f32 v[44:45], v[138:139], v[30:31] op_sel:[0,1] op_sel_hi:[0,0]`: low result = src0.lo * src1.HI), and on MI355X
// that instruction intermittently returns a wrong low product in lanes 48-63 -- the wrong q / k elements of GPUTEST_r05
// (tools/isa_run_qkv.py reproduces it, profiles/r06_isa_ab*.log).  The empty asm statements make every
// intermediate an opaque value, so there is no vectorisable tree; tools/isa_lint.py (run by tests/test_abi.py on the built library)
// fails the build if a crossed packed-fp32 instruction shows up anywhere.
LGEN_DEV void rope_pair(float& x0, float& x1, float c, float s) {
    float a = x0 * c, b = x1 * s, e = x1 * c, f = x0 * s;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(e), "+v"(f));
    float y0 = a - b, y1 = e + f;
    asm volatile("" : "+v"(y0), "+v"(y1));
    x0 = y0;
    x1 = y1;
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
LGEN_DEV uint4 ldg_nt(const uint4* p) {  // streamed-once data (weights): non-temporal
    u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

LGEN_DEV float4 ldg_nt_f4(const float4* p) {  // streamed-once fp32 data
    f32x4_t v = __builtin_nontemporal_load((const f32x4_t*)p);
    return make_float4(v[0], v[1], v[2], v[3]);
}
LGEN_DEV void stg_nt_f4(float4* p, float a, float b, float c, float d) {
    f32x4_t v = {a, b, c, d};
    __builtin_nontemporal_store(v, (f32x4_t*)p);
}

LGEN_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
LGEN_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Cross-lane butterflies without the LDS crossbar (__shfl_xor compiles to ds_bpermute_b32: an LDS instruction and an lgkmcnt round
// trip of ~100 cycles per step -- with one wave per SIMD nothing hides it; round 4).  DPP modifiers fold into the consuming VALU
// op (v_add_f32_dpp ...); the 16- and 32-lane exchanges are gfx950's v_permlane16_swap / v_permlane32_swap (both operands = the
// value: the two results are "rows 0,0,2,2 | 1,1,3,3" resp. "low half twice | high half twice" of it).
template <int CTRL>
LGEN_DEV float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140, DPP_ROR8 = 0x128;
// (inline asm: with the builtin, hipcc of ROCm 7.2 loses track of which register holds which result when both inputs are the same
// value -- `op(r0, r0)` / a stale copy after the swap; found with tools/ubench/dpp_check.hip.  The s_nop covers the VALU-write ->
// permlane-swap-read wait states the compiler would have inserted.)
LGEN_DEV void lane_swap16(float v, float& a, float& b) {   // a, b: the values of this lane's row pair (xor 16 partner in one of them)
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
LGEN_DEV void lane_swap32(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// sum over the LPK lanes of a key group (LPK = 4 .. 32 consecutive lanes, aligned): every lane ends with the group total.  Bitwise
// equal to the __shfl_xor(1, 2, 4, ...) butterfly: after each step all lanes of a sub-group hold the SAME partial, and a + b == b + a.
template <int LPK>
LGEN_DEV float group_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    if constexpr (LPK >= 8) v += dpp_f<DPP_HALF_MIRROR>(v);
    if constexpr (LPK >= 16) v += dpp_f<DPP_MIRROR>(v);
    if constexpr (LPK >= 32) { float a, b; lane_swap16(v, a, b); v = a + b; }
    return v;
}
// combine across the 64 / LPK key groups of a wave, lane-aligned (partner = lane ^ LPK, ^ 2 LPK, ...), LPK >= 8
template <int LPK, typename F>
LGEN_DEV float across_groups(float v, F&& op) {
    if constexpr (LPK <= 8) v = op(v, dpp_f<DPP_ROR8>(v));
    if constexpr (LPK <= 16) { float a, b; lane_swap16(v, a, b); v = op(a, b); }
    { float a, b; lane_swap32(v, a, b); v = op(a, b); }
    return v;
}

// Per-device one-time state of the launchers (kernel attributes, CU count): the deployment is one process per GPU, but nothing in
// the ABI stops a caller from switching devices, and a function attribute set on one device says nothing about another.
static inline int lgen_cur_dev() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = 0;
    return d & 63;
}
static inline int lgen_cu_count() {
    static int n_cu[64] = {0};
    const int d = lgen_cur_dev();
    if (!n_cu[d]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) v = 256;
        n_cu[d] = v > 0 ? v : 256;
    }
    return n_cu[d];
}
#define LGEN_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)

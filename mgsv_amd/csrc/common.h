// Shared device/host helpers for libmade_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/made_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define WAVE 64

// ---- error plumbing (host) -----------------------------------------------------------------
void made_set_error(const char* fmt, ...);

#define MADE_REQUIRE(cond, ...)                                                   \
    do {                                                                          \
        if (!(cond)) {                                                            \
            made_set_error(__VA_ARGS__);                                          \
            return MADE_ERR_INVALID_ARG;                                          \
        }                                                                         \
    } while (0)

#define MADE_UNSUPPORTED(cond, ...)                                               \
    do {                                                                          \
        if (!(cond)) {                                                            \
            made_set_error(__VA_ARGS__);                                          \
            return MADE_ERR_UNSUPPORTED;                                          \
        }                                                                         \
    } while (0)

// ---- measurement knobs ------------------------------------------------------------------------
// Every MADE_* environment variable the library reads selects a VARIANT that an A/B measurement once needed (tile shapes, the split
// attention backward, debug stamps ...; docs/EXPERIMENTS.md says what each one decided).  They are honoured only under
// MADE_DEBUG_VARIANTS=1: a production process has ONE code path, whatever its environment holds.
#include <stdlib.h>
#include <string.h>
static inline const char* made_variant_env(const char* name) {
    const char* on = getenv("MADE_DEBUG_VARIANTS");
    if (on == nullptr || on[0] == '\0' || strcmp(on, "0") == 0) return nullptr;
    return getenv(name);
}

// ---- launch tape (include/made_hip.h: made_tape_*) ------------------------------------------------
// Every kernel of the library is launched through made_launch (the hipLaunchKernelGGL spelling below is redirected to it): the
// launch goes to the stream as always and, while the calling thread records a tape, a copy of (function, grid, block, LDS bytes,
// stream, argument bytes) is appended to it, so that the sequence can be replayed later from one C loop.
extern thread_local void* g_made_tape;                      // the tape this thread is recording (NULL: none)
void made_tape_push_kernel(const void* fn, dim3 grid, dim3 block, unsigned lds, hipStream_t st, void** arg_ptrs, const size_t* arg_sizes, int n);

#include <tuple>
#include <utility>
template <typename... KArgs, size_t... I, typename Tup>
inline void made_launch_impl(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Tup& t, std::index_sequence<I...>) {
    void* ptrs[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {(void*)&std::get<I>(t)...};
    const size_t sizes[sizeof...(KArgs) ? sizeof...(KArgs) : 1] = {sizeof(std::get<I>(t))...};
    if (g_made_tape) made_tape_push_kernel((const void*)kernel, grid, block, (unsigned)lds, st, ptrs, sizes, (int)sizeof...(KArgs));
    (void)hipLaunchKernel((const void*)kernel, grid, block, ptrs, lds, st);
}
template <typename... KArgs, typename... Args>
inline void made_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "made_launch: argument count");
    std::tuple<KArgs...> t{static_cast<KArgs>(args)...};      // by value, in the kernel's own parameter types
    made_launch_impl(kernel, grid, block, lds, st, t, std::index_sequence_for<KArgs...>{});
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) made_launch(kernel, grid, block, lds, stream, __VA_ARGS__)

static inline int made_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        made_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return MADE_ERR_HIP;
    }
    return MADE_OK;
}

// ---- device helpers ------------------------------------------------------------------------
// Row index inside a 32x32 MFMA accumulator tile held by (reg r in [0,16), lane half hh):
// col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * hh   (cdna_hip_programming.md section 3).
__device__ __forceinline__ int acc_row(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <typename T> struct elem_traits;
template <> struct elem_traits<float> {
    static constexpr int per16 = 4;      // elements per 16-byte fragment
};
template <> struct elem_traits<bf16_t> {
    static constexpr int per16 = 8;
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// load one element of runtime dtype as f32 / store f32 as runtime dtype
__device__ __forceinline__ float load_as_f32(const void* p, int dtype, int64_t idx) {
    return dtype == MADE_F32 ? ((const float*)p)[idx] : (float)((const bf16_t*)p)[idx];
}
__device__ __forceinline__ void store_from_f32(void* p, int dtype, int64_t idx, float v) {
    if (dtype == MADE_F32) ((float*)p)[idx] = v;
    else ((bf16_t*)p)[idx] = (bf16_t)v;
}

// Slot of query q (0 .. 31 inside its group of 32) in the dropout bit cache of made_attention (MadeAttnArgs.keep_bits): register e of
// a 32 x 32 MFMA accumulator tile holds row (e & 3) + 8 (e >> 2) + 4 hh, so the words at slots 2 e and 2 e + 1 are the two halves of
// the 64-lane keep mask of register e.
__host__ __device__ __forceinline__ int made_keep_slot(int q) { return 2 * ((q & 3) + 4 * (q >> 3)) + ((q >> 2) & 1); }

// made_attention_bwd's single-pass kernel (attention_bwd_fused.hip): MADE_OK after launching, a HIP error, or -1000 when it does not apply
int made_attention_bwd_fused_try(const MadeAttnBwdArgs& a, hipStream_t st);

// ---- f32 products on the bf16 matrix pipe (made_set_f32_products(1); engine dtype "f32x3") ----------------------------------------
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate (157 TFLOP/s).  With x = hi + lo + O(2^-17 |x|) (hi = bf16(x), lo = bf16(x - hi))
// a . b ~ hi.hi + hi.lo + lo.hi: three v_mfma_f32_32x32x8_bf16 (32 cycles each) replace four f32 MFMAs (64 cycles each) per 8-deep
// step on the SAME fragment layout (lane half hh holds k = 4 hh .. 4 hh + 3 in both), f32 accumulation as before.  Storage, statistics and
// every elementwise step stay f32; only the products lose the operands' last 7 bits (relative 8e-6 per operand, random sign).
extern int g_made_f32_products;
typedef __attribute__((ext_vector_type(4))) short s16x4;
struct SplitF32x4 { s16x4 hi, lo; };
__device__ __forceinline__ SplitF32x4 made_split4(const f32x4 x) {
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = (bf16_t)x[j]; l[j] = (bf16_t)(x[j] - (float)h[j]); }
    SplitF32x4 r;
    r.hi = __builtin_bit_cast(s16x4, h); r.lo = __builtin_bit_cast(s16x4, l);
    return r;
}
__device__ __forceinline__ f32x16 made_mfma_x3(const SplitF32x4& a, const SplitF32x4& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a.lo, b.hi, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a.hi, b.lo, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a.hi, b.hi, c, 0, 0, 0);
}

// two 8-deep steps at once: v_mfma_f32_32x32x16_bf16 (the same 32 cycles for twice the depth); lane half hh holds the first step's
// k = 4 hh .. + 3 in slots 0 .. 3 and the second step's in slots 4 .. 7 -- any assignment is right as long as both operands use it
__device__ __forceinline__ f32x16 made_mfma_x3_16(const SplitF32x4& a0, const SplitF32x4& a1, const SplitF32x4& b0, const SplitF32x4& b1, f32x16 c) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const bf16x8 ah = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0.hi, a1.hi, 0, 1, 2, 3, 4, 5, 6, 7));
    const bf16x8 al = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0.lo, a1.lo, 0, 1, 2, 3, 4, 5, 6, 7));
    const bf16x8 bh = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0.hi, b1.hi, 0, 1, 2, 3, 4, 5, 6, 7));
    const bf16x8 bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0.lo, b1.lo, 0, 1, 2, 3, 4, 5, 6, 7));
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

// keep ? x : 0 on a 16-byte fragment without control flow (conditional LOADS make hipcc branch around every load and
// wait for each one in turn -- cdna_hip_programming.md, "three .s-level traps" (c); load always, mask afterwards)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
template <typename F> __device__ __forceinline__ F keep_or_zero(F x, bool keep) {
    u32x4 b = __builtin_bit_cast(u32x4, x);
    const unsigned int m = keep ? 0xFFFFFFFFu : 0u;
    b[0] &= m; b[1] &= m; b[2] &= m; b[3] &= m;
    return __builtin_bit_cast(F, b);
}

// Keep bits of N <= 8 CONSECUTIVE element indices idx0 .. idx0 + N - 1 of a dropout site (bit j = element idx0 + j is kept).  The mix
// is fmix32(lo32(idx) ^ key(seed, site, hi32(idx))): called per element with a 64-bit index the compiler recomputes the key --
// a second fmix32 round and the 64-bit carry chain, ~30 VALU instructions per element -- here it is made once per group
// (made_rng_mix's definition, bit for bit; the slow path covers a group that straddles a 2^32 boundary).
template <int N>
__device__ __forceinline__ uint32_t made_keep_bits(uint64_t seed, uint32_t site, uint32_t thr, uint64_t idx0) {
    const uint32_t lo = (uint32_t)idx0;
    uint32_t m = 0;
    if (lo <= 0xFFFFFFFFu - (uint32_t)N) {
        const uint32_t k = made_rng_key(seed, site, (uint32_t)(idx0 >> 32));
#pragma unroll
        for (int j = 0; j < N; ++j) m |= ((made_rng_fmix32((lo + (uint32_t)j) ^ k) >> 8) >= thr) ? (1u << j) : 0u;
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) m |= ((made_rng_mix(seed, site, idx0 + (uint64_t)j) >> 8) >= thr) ? (1u << j) : 0u;
    }
    return m;
}

// Wave-wide reductions on the DPP path (VALU only).  The butterfly of six __shfl_xor compiles to six ds_bpermute_b32, which go
// through the LDS crossbar: ~100 cycles each and one LDS unit per CU -- 96 of them per wave cost made_dec_stage 4 us of 10.
// Here: four DPP steps reduce each row of 16 lanes (every lane of the row ends with the row's value), two row broadcasts chain the
// four rows, lane 63 holds the total and is read back as a scalar.  Every lane returns the same bits.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
#ifdef MADE_DEBUG_WAVE_SUM_SHFL                                  // (bisection builds of tools/build_variants.sh: the same sums through ds_bpermute)
    for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_xor(v, o2);
    return v;
#endif
    v += dpp_f32<0xB1, 0xF>(v, v);                               // quad_perm [1,0,3,2]
    v += dpp_f32<0x4E, 0xF>(v, v);                               // quad_perm [2,3,0,1]
    v += dpp_f32<0x124, 0xF>(v, v);                              // row_ror:4
    v += dpp_f32<0x128, 0xF>(v, v);                              // row_ror:8
    v += dpp_f32<0x142, 0xA>(0.f, v);                            // row_bcast:15 -> rows 1, 3
    v += dpp_f32<0x143, 0xC>(0.f, v);                            // row_bcast:31 -> rows 2, 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x124, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x128, 0xF>(v, v));
    v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));                     // rows 0, 2 keep their own value (old = v)
    v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Shared helpers for the gfx950 kernels and the C ABI (no torch types anywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <type_traits>

#include "../../include/mi355_nnunet.h"

namespace mi355 {

void set_error(const char *fmt, ...);

// One process per GPU: the library binds to the HIP device that is current at its first compute call (weights, the
// activation arena, the scratch buffers below and the raised dynamic-LDS limits all live on / apply to that device)
// and every later entry point fails with MI355_ERR_INVALID if another device is current.  Also fails with
// MI355_ERR_NO_DEVICE when no gfx950 device is visible.
int bind_device();

// Process-wide scratch on the bound device, one buffer per (lane, slot) - a lane = one of the first few streams the caller
// launches on (unet.hip, "Lanes") -, grown on demand (synchronise, free, allocate) under a mutex and never freed per call;
// `zeroed` buffers are cleared when (re)allocated.  Work that uses a slot is ordered by the stream it was asked for
// ("Stream semantics").
enum ScratchSlot { SCR_ARENA = 0, SCR_SW_AGG, SCR_SPLITK_F32, SCR_SPLITK_F16, SCR_ZEROS, SCR_ZERO_BIAS, SCR_SMALL, SCR_TOPK,
                   SCR_CROP, SCR_ZSCORE, SCR_RESAMPLE, SCR_RESAMPLE_MM, SCR_COUNT };
int device_scratch(int slot, hipStream_t stream, size_t bytes, void **out, bool zeroed = false);

#define MI355_HIP(expr)                                                                    \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            ::mi355::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),     \
                               __FILE__, __LINE__);                                        \
            return MI355_ERR_HIP;                                                          \
        }                                                                                  \
    } while (0)

#define MI355_TRY(expr)            \
    do {                           \
        int rc__ = (expr);         \
        if (rc__ != MI355_OK)      \
            return rc__;           \
    } while (0)

#define MI355_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            ::mi355::set_error(__VA_ARGS__);  \
            return MI355_ERR_INVALID;         \
        }                                     \
    } while (0)

// Exact unsigned division by a runtime constant for n < 2^31 (round-up method).
struct FastDiv {
    uint32_t d, m, s;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    uint32_t s = 0;
    while ((1u << s) < d)
        ++s;
    f.s = s;
    f.m = (uint32_t)(((uint64_t(1) << 32) * ((uint64_t(1) << s) - d)) / d + 1);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    return (__umulhi(n, f.m) + n) >> f.s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous
// range of the logical tile list so halo re-reads hit that XCD's L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Order-independent accumulation of the Instance/GroupNorm sums.  Every workgroup adds its partial (sum, sum of squares)
// of a channel to one fp64 word per (sample, channel) with atomicAdd, i.e. in arrival order, and fp64 addition is not
// associative.  Rounding each partial to a multiple of a fixed quantum q first makes every addition EXACT (all operands and
// all partial totals are multiples of q, and a double holds multiples of q exactly up to 2^53 q), so the total does not
// depend on the order: run-to-run and rank-to-rank bit-identical statistics without a second pass.  q scales with the
// number of voxels V the statistic runs over (2^lv <= V): 2^(lv-40) for sum x, 2^(lv-44) for sum x^2.
//   * What the rounding costs: a partial (one wave's or one workgroup's share of a channel) covers >= 32 voxels,
//     so there are at most V / 32 partials and the worst case - every partial off by q / 2 in the same direction - moves
//     mean x by q / 64 = 2^(lv-46) and mean x^2 by 2^(lv-50): 3.0e-8 and 1.9e-9 at a 128^3 patch (lv = 21), nothing against
//     eps = 1e-5 however small the channel's pre-norm rms is (round 2's quantum for sum x^2, 2^(lv-30) = 2e-3 at that patch,
//     zeroed the variance of a channel with rms 1e-3: every partial rounded to 0).
//   * Where the additions stay exact (all totals below 2^53 q): |mean x| < 8192 and rms x < 16.  Beyond those
//     magnitudes the additions merely stop being exact: the statistics stay correct to fp64 rounding, only the order
//     independence is lost.
__device__ __forceinline__ double quantise_partial(double v, int k, long voxels) {
    const int lv = 63 - __clzll((unsigned long long)(voxels > 0 ? voxels : 1));
    const int e = lv - (k == 0 ? 40 : 44);
    return ldexp(rint(ldexp(v, -e)), e);
}

// Sum over the 32 lanes of a wave half (lanes 0-31 / 32-63), returned in every lane of that half.  Four DPP steps (quad_perm,
// quad_perm, row_half_mirror, row_mirror: plain VALU adds with a lane-permuted operand, 4 cycles each) give every lane of a
// 16-lane row its row total; one cross-row exchange finishes.  Round 3: the Instance/GroupNorm statistics epilogues used a
// five-step __shfl_xor butterfly per value - hipcc lowers every step to ds_swizzle / ds_bpermute, i.e. an LDS round trip,
// and 640 of them in dependent chains cost the fp16 LDS-DMA kernel 16.6k cycles per tile (20.2k against 3.5k for the
// epilogue without statistics, tools/h16_probe.hip stamps: 31 % of a 4-chunk tile).
template <int CTRL>
__device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));  // (bound_ctrl: no 'old' operand to materialise)
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_perm<0xB1>(v);   // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_perm<0x4E>(v);   // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_perm<0x141>(v);  // row_half_mirror: lane <-> 7 - lane within 8: the other quad's total
    v += dpp_perm<0x140>(v);  // row_mirror: lane <-> 15 - lane within 16: the other half-row's total
    return v;
}
__device__ __forceinline__ float half32_sum(float v) {
    v = row16_sum(v);
    return v + __shfl_xor(v, 16);
}

// Transposing reduction of the statistics of a 32x32 accumulator fragment whose LANES are voxels (lane & 31) and whose 16
// registers are channels: a lane holds 16 partial sums (s1) and 16 partial sums of squares (s2), and the totals over the 32
// lanes of its half-wave are wanted.  Reducing each of the 32 values with a butterfly is 32 x 5 cross-lane additions per
// lane (the compiler made 768 VALU instructions of them, h16 stamps: +6.1k cycles per tile).  Here every step HALVES the
// values a lane still carries: the two partners keep one half each and send the other, so the steps cost 16+8+4+2+1 = 31
// additions, and each of the 32 lanes ends with the total of exactly ONE value - the one it then adds to the statistics
// buffer, with no trip through LDS:
//   index r = (lane >> 1 & 1) + 2 (lane & 1) + 4 (lane >> 2 & 1) + 8 (lane >> 3 & 1) of s1 (lane & 16 == 0) or s2 (lane & 16)
// Step 1 is gfx950's v_permlane16_swap (rows 1 and 3 of the first operand change places with rows 0 and 2 of the second:
// one instruction is the exchange of both directions); the 16-lane rows then use DPP.  The half-mirror step comes before
// the two quad steps because its partner (lane ^ 7) differs in the quad bits too: the later partners (lane ^ 1, lane ^ 2)
// have made the same choice in it, which is all a halving step needs.
__device__ __forceinline__ int stat_slot_r(int lane) { return ((lane >> 1) & 1) + 2 * (lane & 1) + (lane & 12); }
typedef float stat_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float half32_reduce_scatter(const stat_f32x2 (&s1)[8], const stat_f32x2 (&s2)[8], int lane) {
    // (inline asm: hipcc 7.2 miscompiles __builtin_amdgcn_permlane16_swap - both members of the returned pair come out as the
    //  same register, tools/stat_probe.hip.  The hazard recogniser does not look into inline asm: the s_nop 1 in front covers
    //  the two wait states a VALU write needs before a permlane reads the register, the one behind covers its readers.)
    float y[16];
#pragma unroll
    for (int q = 0; q < 16; q += 4) {
        float a0 = s1[(q >> 1)][0], a1 = s1[(q >> 1)][1], a2 = s1[(q >> 1) + 1][0], a3 = s1[(q >> 1) + 1][1];
        float c0 = s2[(q >> 1)][0], c1 = s2[(q >> 1)][1], c2 = s2[(q >> 1) + 1][0], c3 = s2[(q >> 1) + 1][1];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                     "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        y[q] = a0 + c0; y[q + 1] = a1 + c1; y[q + 2] = a2 + c2; y[q + 3] = a3 + c3;
    }
    const bool b3 = lane & 8, b2 = lane & 4, b0 = lane & 1, b1 = lane & 2;
    float z[8], w[4], u[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (b3 ? y[i + 8] : y[i]) + dpp_perm<0x128>(b3 ? y[i] : y[i + 8]);  // row_ror:8 = lane ^ 8
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (b2 ? z[i + 4] : z[i]) + dpp_perm<0x141>(b2 ? z[i] : z[i + 4]);  // row_half_mirror = lane ^ 7
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (b0 ? w[i + 2] : w[i]) + dpp_perm<0xB1>(b0 ? w[i] : w[i + 2]);   // lane ^ 1
    return (b1 ? u[1] : u[0]) + dpp_perm<0x4E>(b1 ? u[0] : u[1]);                                      // lane ^ 2
}

// Two 8-cout blocks of one voxel, 16 bytes per lane (round 3).  In the accumulator layout a lane holds couts 4 half .. + 3 of
// a block for its voxel l31 and lane ^ 32 the other four.  v_permlane32_swap exchanges the upper half-wave of its first
// operand with the lower half-wave of its second: afterwards lanes 0-31 hold all 16 bytes of block g (`v0`) and lanes 32-63
// those of block g + 1 (`v1`), so a wave stores 16 B per lane - half the store instructions of the 8-byte form, and in the
// channel-blocked layout (common.h) x-consecutive voxels of a block are whole lines.  Must run with every lane active (the
// caller predicates the STORE, not this).  Inline asm: the pair-returning builtin is miscompiled by this hipcc (common.h);
// s_nop 1 = the two wait states between a VALU write of an operand and the swap, and before a reader of the result.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4_t pair_blocks_f16(f16x4 v0, f16x4 v1) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
    return u32x4_t{a[0], a[1], b[0], b[1]};
}

// fp16 activation layout (round 3): channel-blocked NDHWC, [N][C / 8][D][H][W][8] ("B8", oneDNN's nCdhw8c).  A voxel's eight
// channels of one block are the 16 bytes ONE lane feeds the matrix cores (v_mfma_f32_32x32x16_f16 B operand: lane = voxel,
// channels 8 (lane >> 5) .. + 7), and x-consecutive voxels of a block are contiguous, so
//   * an LDS-DMA piece (64 lanes x 16 B) of a brick reads rows of 160+ contiguous bytes instead of 64 lines that each give
//     16 of their 128 bytes: tools/dma_probe.hip measures 84-128 cycles per instruction for such rows against 256 for the
//     16- or 32-byte pieces of plain NDHWC, and 5-8 times the bytes per clock once the lines come from beyond the L2
//     (profiles/r03_dma_probe.txt);
//   * the 16-channel chunks of a convolution no longer share 128-B lines: with plain NDHWC a chunk fetched a quarter of
//     each line and the next chunk found the line evicted (2.4-6.8 x the algorithmic bytes through the fabric, round 2);
//   * epilogues store whole lines (8 voxels x 16 B per block row) without gathering 64 couts of a voxel first.
// The fp32 path keeps plain NDHWC (16 B = 4 channels there; its kernels stage 8- and 16-channel chunks of whole lines).
// Element (n, c, v) of a tensor with C channels and V voxels per sample:
__host__ __device__ __forceinline__ size_t b8_index(int64_t n, int c, int64_t v, int C, int64_t V) {
    return (size_t)(((n * (C >> 3) + (c >> 3)) * V + v) * 8 + (c & 7));
}

// Compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N).  Indices are constant EXPRESSIONS inside the body
// (inline-asm immediates, register-array subscripts), whatever the optimiser decides about a "#pragma unroll" loop.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Blocked tile order of the persistent kernels.  Within a sample, tile ids run through blocks of 2^lbx x 2^lby x 2^lbz
// tiles (32 tiles when the grid allows), x fastest inside a block and over the blocks.  The 32 workgroups of an XCD work
// on 32 consecutive ids, so at any time they cover one compact 3-D block whose halos overlap inside that XCD's L2 (4x4x2
// tiles of 4x4x32 voxels: 1.43 input voxels fetched per output voxel instead of 1.62 for a 4x8x1 slab), and the partial
// lines a narrow channel chunk leaves behind are still there when the next chunk of the same voxels is fetched.
struct TileOrder {
    int lbx, lby, lbz;
    FastDiv div_nbx, div_nby;  // blocks per row / per slab
};
inline TileOrder make_tile_order(int tiles_x, int tiles_y, int tiles_z) {
    auto pow2 = [](int v) { return v > 0 && (v & (v - 1)) == 0; };
    int lbx = 0, lby = 0, lbz = 0;
    if (pow2(tiles_x) && pow2(tiles_y) && pow2(tiles_z) && (long)tiles_x * tiles_y * tiles_z >= 32) {
        while ((1 << lbx) < tiles_x && lbx < 2) ++lbx;  // x tiles are the widest (16 or 32 voxels): at most 4 of them
        int rest = 5 - lbx;
        while (rest > 0) {  // deal the remaining factors of two to y and z alternately, y first
            if ((1 << lby) < tiles_y && (lby <= lbz || (1 << lbz) >= tiles_z)) { ++lby; --rest; }
            else if ((1 << lbz) < tiles_z) { ++lbz; --rest; }
            else break;
        }
        if (rest > 0) lbx = lby = lbz = 0;  // (a grid that cannot be cut into 32-tile blocks keeps the linear order)
    }
    TileOrder o;
    o.lbx = lbx; o.lby = lby; o.lbz = lbz;
    o.div_nbx = make_fastdiv(tiles_x >> lbx);
    o.div_nby = make_fastdiv(tiles_y >> lby);
    return o;
}
// tile index within a sample -> (tile_x, tile_y, tile_z)
__device__ __forceinline__ void tile_from_id(int tt, const TileOrder &o, int &tile_x, int &tile_y, int &tile_z) {
    const int lb = o.lbx + o.lby + o.lbz;
    const int w = tt & ((1 << lb) - 1), blk = tt >> lb;  // tile within its block, block within the sample
    const int bzy = (int)fdiv((uint32_t)blk, o.div_nbx);
    const int bxi = blk - bzy * (int)o.div_nbx.d;
    const int bzi = (int)fdiv((uint32_t)bzy, o.div_nby);
    const int byi = bzy - bzi * (int)o.div_nby.d;
    tile_x = (bxi << o.lbx) + (w & ((1 << o.lbx) - 1));
    tile_y = (byi << o.lby) + ((w >> o.lbx) & ((1 << o.lby) - 1));
    tile_z = (bzi << o.lbz) + (w >> (o.lbx + o.lby));
}

inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v)
        ++l;
    return l;
}
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace mi355

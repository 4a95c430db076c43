// 3x3x3 convolution, NDHWC fp32, on the gfx950 matrix cores.
//
// Replaces torch.nn.Conv3d(k=3, p=1, stride 1|2) inside ConvDropoutNormNonlin
// (reference model_architecture/generic_UNet.py:56,69) - 99.7 % of the path's flops.
//
// GEMM view: D[cout][voxel] = W x X with K = 27 taps x Cin.  The weights are the MFMA A operand, the voxels the B
// operand, so a lane holds one voxel (column lane&31) and 16 couts of a 32x32 tile.
//   * voxels come from an LDS image of the input halo brick, planar [16-B channel quad][brick voxel]: x-consecutive
//     lanes read consecutive 16-B slots (conflict-free); one ds_read_b128 feeds four v_mfma_f32_32x32x2_f32 (lane l
//     holds channels g*8 + 4*(l>>5) + j of voxel l&31; MFMA j contracts the channel pair {g*8+j, g*8+4+j});
//   * weights are pre-permuted on the host into fragment order (1 KiB per (tap, 8-channel group, 32 couts));
//   * f32-input MFMA is an exact k-ordered fmaf chain (no TF32 on gfx950).
// Kernels in this file (dispatch: conv3d_mfma_f32 at the bottom; DESIGN.md section 5 has the measurements):
//   conv3_f32_wino2_kernel      stride 1, large launches: Winograd F(2x2,3x3) over (z,y), LDS-DMA bricks   [default]
//   conv3_f32_wino_kernel       stride 1, large launches: Winograd F(2,3) along y                   (MI355_WINOGRAD=1)
//   conv3_f32_s2dma_kernel      stride 2, large launches: LDS-DMA bricks, weight planes through an LDS ring
//   conv3_f32_mfma_pipe_kernel  stride 1, mid-size launches: persistent, register-staged double-buffered brick
//   conv3_f32_mfma_kernel       everything else (one tile per workgroup, stride 1|2), also the split-K slices of the
//                               deep levels (+ splitk_finish_kernel)
//   conv3_direct_kernel         reference-order direct convolution (tests, odd channel counts)
// Shared epilogue: + bias, LeakyReLU, predicated wide NDHWC stores; optionally per-(n, channel) sum / sum of squares
// for Instance/GroupNorm (wave reduction -> LDS -> one fp64 atomic per channel and workgroup) or the fused 1x1x1 head.
#include "kernels.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace mi355 {

struct ConvArgs {
    const float *in0, *in1;
    const float *wp, *bias;
    float *out;
    double *stats;
    int C0, C1;
    int N, Di, Hi, Wi, Do, Ho, Wo, Cout;
    int lx, ly, lz;  // log2 of the output tile dims
    int tiles_x, tiles_y, tiles_z;
    int IX, IY, IZ;  // input brick dims
    FastDiv div_tiles_per_n, div_tiles_x, div_tiles_y, div_IX, div_IY;
    int nchunks;
    int act;
    float slope;
    const float *head_w, *head_b;  // fused 1x1x1 head (see ConvCall)
    float *head_out;
    int head_ncls;
    // split-K (simple kernel, small launches): blockIdx.z = slice of the channel chunks; raw partial sums go to `partial`
    int ksplit;
    float *partial;
    const float *zero_bias;
    long out_elems;
};

// Shared epilogue.  The MFMAs are issued as D = W x X (weights are the A operand, voxels the B operand),
// so a lane holds ONE voxel (column lane&31) and, in its 16 accumulator registers, 16 couts:
// cout = (r&3) + 8*(r>>2) + 4*(lane>>5).  Registers 4g..4g+3 are four consecutive couts -> one 16-B store;
// 4 stores per 32x32 tile instead of 16 dword stores (the store tail is issue-bound, not bandwidth-bound).
// Adds bias, optional LeakyReLU, predicated on the volume bounds; optional per-(n, cout) sum / sum of squares:
// butterfly over the 32 voxel lanes, then LDS across the 4 waves, then one fp64 atomic per cout.
template <int MF, int NF, int NW = 4>
__device__ __forceinline__ void conv_epilogue(f32x16 (&acc)[MF][NF], const ConvArgs &p, int n, int oz0, int oy0,
                                              int ox0, int co_blk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    if (p.head_out) {
        // fused segmentation head: logits[k] = sum_cout w[k][cout] * act(conv[cout]) + b[k]; a lane holds 16*NF couts of
        // its voxel, lanes l / l+32 the two halves -> one cross-half add; the feature map is never written
        constexpr int KMAX = 4;
        const int64_t Vo = (int64_t)p.Do * p.Ho * p.Wo;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int v = (wave * MF + mf) * 32 + l31;
            const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
            const bool ok = (oz < p.Do) && (oy < p.Ho) && (ox < p.Wo);
            float part[KMAX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = co_blk + nf * 32 + 8 * g + 4 * half;
                    const f32x4 bias = *(const f32x4 *)(p.bias + co);
                    f32x4 x;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float y = acc[mf][nf][4 * g + k] + bias[k];
                        if (p.act == ACT_LRELU) y = y > 0.f ? y : y * p.slope;
                        x[k] = y;
                    }
#pragma unroll
                    for (int c = 0; c < KMAX; ++c)
                        if (c < p.head_ncls) {
                            const f32x4 hw = *(const f32x4 *)(p.head_w + c * p.Cout + co);
                            part[c] += x[0] * hw[0] + x[1] * hw[1] + x[2] * hw[2] + x[3] * hw[3];
                        }
                }
#pragma unroll
            for (int c = 0; c < KMAX; ++c) part[c] += __shfl_xor(part[c], 32);
            if (ok && half == 0) {
                const int64_t vi = ((int64_t)oz * p.Ho + oy) * p.Wo + ox;
#pragma unroll
                for (int c = 0; c < KMAX; ++c)
                    if (c < p.head_ncls) p.head_out[((int64_t)n * p.head_ncls + c) * Vo + vi] = part[c] + p.head_b[c];
            }
        }
        return;
    }
    if (!p.stats) {
        // plain path: bias, LeakyReLU as max(x, slope * x) (slope 1 = none), wide stores; packed f32 math - on a SIMD whose
        // matrix pipe waits for this epilogue every VALU instruction counts (tools/coissue_probe.hip)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const float slope = p.act == ACT_LRELU ? p.slope : 1.0f;
        const f32x2 slope2 = {slope, slope};
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int v = (wave * MF + mf) * 32 + l31;
            const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
            const bool ok = (oz < p.Do) && (oy < p.Ho) && (ox < p.Wo);
            float *orow = p.out + ((((size_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Cout + co_blk + 4 * half;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 bias = *(const f32x4 *)(p.bias + co_blk + nf * 32 + 8 * g + 4 * half);
                    f32x4 val;
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        f32x2 x = {acc[mf][nf][4 * g + k], acc[mf][nf][4 * g + k + 1]};
                        const f32x2 b2 = {bias[k], bias[k + 1]};
                        f32x2 y;
                        asm("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b2));
                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(slope2));
                        asm("v_max_f32 %0, %1, %2" : "=v"(val[k]) : "v"(x[0]), "v"(y[0]));      // bare max: fmaxf would add a
                        asm("v_max_f32 %0, %1, %2" : "=v"(val[k + 1]) : "v"(x[1]), "v"(y[1]));  // canonicalising max per value
                    }
                    if (ok) *(f32x4 *)(orow + nf * 32 + 8 * g) = val;
                }
        }
        return;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 s1[NF][8], s2[NF][8];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 8; ++r) { s1[nf][r] = f32x2{0.f, 0.f}; s2[nf][r] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = (wave * MF + mf) * 32 + l31;
        const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
        const bool ok = (oz < p.Do) && (oy < p.Ho) && (ox < p.Wo);
        const float in = ok ? 1.f : 0.f;  // a voxel beyond a ragged edge adds nothing to the statistics
        float *orow = p.out + ((((size_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Cout + co_blk + 4 * half;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bias = *(const f32x4 *)(p.bias + co_blk + nf * 32 + 8 * g + 4 * half);
                f32x4 val;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float x = acc[mf][nf][4 * g + k] + bias[k];
                    if (p.act == ACT_LRELU) x = x > 0.f ? x : x * p.slope;
                    val[k] = x;
                }
#pragma unroll
                for (int k = 0; k < 4; k += 2) {  // v_pk_add_f32 / v_pk_fma_f32: one instruction per value pair
                    const f32x2 m = f32x2{val[k], val[k + 1]} * f32x2{in, in};
                    s1[nf][2 * g + (k >> 1)] += m;
                    s2[nf][2 * g + (k >> 1)] = __builtin_elementwise_fma(m, m, s2[nf][2 * g + (k >> 1)]);
                }
                if (ok) *(f32x4 *)(orow + nf * 32 + 8 * g) = val;
            }
        }
    }
    // (round 3) transposing reduction over the 32 voxel lanes (common.h): each lane ends with the total of ONE (cout, statistic)
    // of this wave's MF * 32 voxels and adds it itself - no LDS, no barrier (round 2: 32 NF butterflies, a cross-wave reduction
    // through `red` behind two __syncthreads()).  Quantised partials: exact additions, hence order-independent.
    (void)tid;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const float tot = half32_reduce_scatter(s1[nf], s2[nf], lane);
        const int r = stat_slot_r(lane), k = (lane >> 4) & 1;
        const int c = nf * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        atomicAdd(p.stats + ((size_t)n * p.Cout + co_blk + c) * 2 + k, quantise_partial((double)tot, k, (long)p.Do * p.Ho * p.Wo));
    }
}

// Whole-line variant of the plain epilogue (no statistics, no fused head).  In the accumulator layout a lane holds 16
// couts of ONE voxel, so a 16-B store instruction touches 32 different 128-B lines, a quarter of each: the store tail of
// a 4x4x32x32 tile took 13.7k cycles of a workgroup's time (tools/wino2_probe.hip stamps, round 2) - issue-bound on
// partial lines, not bandwidth-bound.  Here each 32-voxel x 32-cout fragment is transposed through LDS (a wave-private
// 32 x 36-float image: ds_write_b128 by voxel row, ds_read_b128 by line) so that 8 consecutive lanes store one whole
// 128-B line and an instruction covers 8 full lines; bias and LeakyReLU are applied after the transposition (one bias
// quad per lane and cout block).  LDS traffic is free next to the f32 MFMAs (tools/coissue_probe.hip), VALU work is
// the same as before.  `stage` = this wave's 32 * 36 floats; DS operations of one wave execute in order, so no barrier
// is needed between the writes and the reads.
constexpr int EPI_PITCH = 36;                      // floats per staged voxel row: 128 B + 16 B keeps ds_write_b128 conflict-free
constexpr int EPI_STAGE_FLOATS = 32 * EPI_PITCH;   // per wave
template <int MF, int NF>
__device__ __forceinline__ void conv_epilogue_lines(f32x16 (&acc)[MF][NF], const ConvArgs &p, int n, int oz0, int oy0,
                                                    int ox0, int co_blk, float *stage) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    const float slope = p.act == ACT_LRELU ? p.slope : 1.0f;
    const f32x2 slope2 = {slope, slope};
    const int srow = lane >> 3, spiece = lane & 7;
    float *wr = stage + l31 * EPI_PITCH + 4 * half;
    const float *rd = stage + srow * EPI_PITCH + spiece * 4;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const f32x4 bias = *(const f32x4 *)(p.bias + co_blk + nf * 32 + spiece * 4);
        const f32x2 b01 = {bias[0], bias[1]}, b23 = {bias[2], bias[3]};
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 raw = {acc[mf][nf][4 * g], acc[mf][nf][4 * g + 1], acc[mf][nf][4 * g + 2], acc[mf][nf][4 * g + 3]};
                *(f32x4 *)(wr + 8 * g) = raw;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4 raw = *(const f32x4 *)(rd + 8 * t * EPI_PITCH);
                const int v = (wave * MF + mf) * 32 + 8 * t + srow;
                const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
                const bool ok = (oz < p.Do) && (oy < p.Ho) && (ox < p.Wo);
                f32x2 x0 = {raw[0], raw[1]}, x1 = {raw[2], raw[3]}, y0, y1;
                f32x4 val;
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(x0) : "v"(x0), "v"(b01));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(x1) : "v"(x1), "v"(b23));
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(slope2));
                asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(slope2));
                asm("v_max_f32 %0, %1, %2" : "=v"(val[0]) : "v"(x0[0]), "v"(y0[0]));  // bare max: fmaxf would add a
                asm("v_max_f32 %0, %1, %2" : "=v"(val[1]) : "v"(x0[1]), "v"(y0[1]));  // canonicalising max per value
                asm("v_max_f32 %0, %1, %2" : "=v"(val[2]) : "v"(x1[0]), "v"(y1[0]));
                asm("v_max_f32 %0, %1, %2" : "=v"(val[3]) : "v"(x1[1]), "v"(y1[1]));
#ifdef MI355_W2_ABL_NOSTORE
                asm volatile("" :: "v"(val), "v"(ok), "v"(oz), "v"(oy), "v"(ox));
#else
                if (ok)
                    *(f32x4 *)(p.out + ((((size_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Cout + co_blk + nf * 32 + spiece * 4) = val;
#endif
            }
        }
    }
}

template <int STRIDE, int CC, int MF, int NF>
__global__ __launch_bounds__(256, 2) void conv3_f32_mfma_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int Q = CC / 4;  // 16-B channel quads per voxel = LDS planes
    constexpr int G = CC / 8;  // 8-channel groups per chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = (int)fdiv((uint32_t)bid, p.div_tiles_per_n);
    int t = bid - n * (int)p.div_tiles_per_n.d;
    const int tzy = (int)fdiv((uint32_t)t, p.div_tiles_x);
    const int tile_x = t - tzy * p.tiles_x;
    const int tile_z = (int)fdiv((uint32_t)tzy, p.div_tiles_y);
    const int tile_y = tzy - tile_z * p.tiles_y;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    const int oz0 = tile_z << p.lz, oy0 = tile_y << p.ly, ox0 = tile_x << p.lx;
    const int iz0 = oz0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const int IX = p.IX, IY = p.IY;
    const int brickvox = IX * IY * p.IZ;
    const int npieces = brickvox * Q;
    const int plane = brickvox * 4;  // floats per plane; LDS image is planar [quad][brick voxel][4 floats]:
                                     // x-consecutive lanes read consecutive 16-B slots (conflict-free, no padding)

    // per-lane LDS offset (floats) of each voxel fragment at tap (0,0,0) in plane `half`
    int a_base[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = (wave * MF + mf) * 32 + l31;
        const int x = v & TXm, y = (v >> p.lx) & TYm, z = v >> (p.lx + p.ly);
        a_base[mf] = half * plane + ((z * STRIDE * IY + y * STRIDE) * IX + x * STRIDE) * 4;
    }

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[mf][nf][r] = 0.f;

    const float *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (27 * G * NF * 256) + lane * 4;

    const int ks = (int)blockIdx.z;
    const int ch_begin = p.ksplit > 1 ? ks * p.nchunks / p.ksplit : 0, ch_end = p.ksplit > 1 ? (ks + 1) * p.nchunks / p.ksplit : p.nchunks;
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        // ---- stage the CC-channel input halo brick (zero outside the volume); Q lanes share one voxel,
        //      i.e. CC*4 contiguous bytes per voxel (whole 64-B half lines for CC = 16)
        const int cglob = ch * CC;
        const float *src;
        int Csrc, coff;
        if (cglob < p.C0) {
            src = p.in0; Csrc = p.C0; coff = cglob;
        } else {
            src = p.in1; Csrc = p.C1; coff = cglob - p.C0;
        }
        src += (size_t)n * p.Di * p.Hi * p.Wi * Csrc + coff;
        constexpr int U = 4;
        for (int i0 = tid; i0 < npieces; i0 += 256 * U) {
            f32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                const int bv = i / Q, q = i - bv * Q;
                const int r = (int)fdiv((uint32_t)bv, p.div_IX);
                const int bx = bv - r * IX;
                const int bz = (int)fdiv((uint32_t)r, p.div_IY);
                const int by = r - bz * IY;
                const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
                const bool ok = (i < npieces) && ((unsigned)iz < (unsigned)p.Di) &&
                                ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
                dst[u] = (i < npieces) ? q * plane + bv * 4 : -1;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (ok)
                    val = *(const f32x4 *)(src + ((size_t)(iz * p.Hi + iy) * p.Wi + ix) * Csrc + q * 4);
                v[u] = val;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0)
                    *(f32x4 *)(lds + dst[u]) = v[u];
        }
        __syncthreads();

        // ---- 27 taps x G channel groups, fragments prefetched one step ahead
        const float *wch = wblk + (size_t)ch * (27 * G * NF * 256);
        f32x4 a_cur[MF], b_cur[NF], a_nxt[MF], b_nxt[NF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
            a_cur[mf] = *(const f32x4 *)(lds + a_base[mf]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
            b_cur[nf] = *(const f32x4 *)(wch + nf * 256);

        for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // next step = (tap, g+1) or (tap+1, 0)
                int ntap = tap, ng = g + 1;
                if (ng == G) { ng = 0; ntap = tap + 1; }
                if (ntap < 27) {
                    const int dz = ntap / 9, rr = ntap - dz * 9, dy = rr / 3, dx = rr - dy * 3;
                    const int off = ((dz * IY + dy) * IX + dx) * 4 + ng * 2 * plane;
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
                        a_nxt[mf] = *(const f32x4 *)(lds + a_base[mf] + off);
                    const float *wn = wch + (size_t)(ntap * G + ng) * (NF * 256);
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        b_nxt[nf] = *(const f32x4 *)(wn + nf * 256);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NF; ++nf)
                            acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[nf][j], a_cur[mf][j],
                                                                               acc[mf][nf], 0, 0, 0);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
                    a_cur[mf] = a_nxt[mf];
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    b_cur[nf] = b_nxt[nf];
            }
        }
        __syncthreads();  // brick is overwritten by the next chunk
    }

    if (p.ksplit > 1) {  // raw partial sums of this channel slice; bias / activation happen in splitk_finish_kernel
        ConvArgs q = p;
        q.out = p.partial + (size_t)ks * p.out_elems;
        q.bias = p.zero_bias; q.act = ACT_NONE; q.stats = nullptr; q.head_out = nullptr;
        conv_epilogue<MF, NF>(acc, q, n, oz0, oy0, ox0, (int)blockIdx.y * NF * 32);
        return;
    }
    conv_epilogue<MF, NF>(acc, p, n, oz0, oy0, ox0, (int)blockIdx.y * NF * 32);
}

// out = act(sum_s partial[s] + bias[c]): the slices are added in slice order, so the result does not depend on scheduling
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float *partial, int S, long total4, int cout4, const float *bias,
                                                            int act, float slope, float *out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    f32x4 v = *(const f32x4 *)(partial + i * 4);
    for (int s2 = 1; s2 < S; ++s2) {
        const f32x4 t = *(const f32x4 *)(partial + ((long)s2 * total4 + i) * 4);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const f32x4 b = *(const f32x4 *)(bias + (i % cout4) * 4);
    const float sl = act == ACT_LRELU ? slope : 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x = v[k] + b[k];
        v[k] = act == ACT_LRELU ? (x > 0.f ? x : x * sl) : x;
    }
    *(f32x4 *)(out + i * 4) = v;
}


// ------------------------------------------------------------------ pipelined variant (stride 1)
// Same GEMM mapping, restructured so the matrix pipe never waits for staging (tools/conv_probe2.hip is the
// micro-benchmark this structure was developed on: +20 % over the kernel above on the 32->32 layer shape):
//   * persistent workgroups walk a contiguous, XCD-local range of output tiles;
//   * the input brick is staged in 8-channel chunks into a DOUBLE-buffered LDS image while the 27 tap steps of
//     the previous chunk run: slot r (one 16-B piece per thread) is fetched at step 2r - after that step's
//     weight fetch, so it is the youngest entry of the in-order vmcnt queue - and written to the other buffer
//     five steps later.  Every staging load is unconditional (out-of-volume / unused slots read the tensor
//     base and are zeroed or dropped at the write), so the compiler's vmcnt counts are static;
//   * LDS image is planar [16-B channel quad][brick voxel]: x-consecutive lanes read consecutive 16-B slots
//     (conflict-free ds_read_b128, no padding);
//   * weight fragments come through a 3-deep register ring (fetched 3 steps ahead), voxel fragments are
//     double-registered (no copies); MF = 4 voxel fragments per wave when Cout = 32 halves the number of
//     weight fetches per MFMA;
//   * one barrier per chunk; 2 workgroups per CU.
struct PipeArgs {
    ConvArgs c;
    int total_tiles;   // N * tiles per sample
    int plane;         // floats per LDS plane (= brickvox * 4)
    int buf_floats;    // floats per LDS buffer (= 2 * plane)
    int w_split;       // the weight pack interleaves NF*w_split cout fragments per tap; this launch takes NF of them per block
};

template <int MF, int NF>
__global__ __launch_bounds__(256, 2) void conv3_f32_mfma_pipe_kernel(PipeArgs pa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs &p = pa.c;
    constexpr int CC = 8;
    constexpr int SLOTS = MF == 4 ? 11 : 8;  // 16-B staging pieces per thread and chunk (brick <= SLOTS*128 voxels)
    constexpr int BD = 3;                    // weight fragments are fetched BD tap-steps ahead
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    const int IX = p.IX, IY = p.IY;
    const int brickvox = IX * IY * p.IZ;
    const int npieces = 2 * brickvox;

    // this workgroup's tile sequence: XCD group x owns the contiguous range [lo, hi); its workgroups stride through it
    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int nl = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = pa.total_tiles >> 3, r8 = pa.total_tiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + li;
    if (tile >= hi) return;

    // per-lane LDS offsets (floats) of each voxel fragment at tap (0,0,0), in plane `half`
    int a_base[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = (wave * MF + mf) * 32 + l31;
        const int x = v & TXm, y = (v >> p.lx) & TYm, z = v >> (p.lx + p.ly);
        a_base[mf] = half * pa.plane + ((z * IY + y) * IX + x) * 4;
    }
    const int qoff = (tid & 1) * 4;

    struct TileCoord { int n, oz0, oy0, ox0; };
    auto decode = [&](int t) {
        TileCoord tc;
        tc.n = (int)fdiv((uint32_t)t, p.div_tiles_per_n);
        const int tt = t - tc.n * (int)p.div_tiles_per_n.d;
        const int tzy = (int)fdiv((uint32_t)tt, p.div_tiles_x);
        const int tile_x = tt - tzy * p.tiles_x;
        const int tile_z = (int)fdiv((uint32_t)tzy, p.div_tiles_y);
        const int tile_y = tzy - tile_z * p.tiles_y;
        tc.oz0 = tile_z << p.lz; tc.oy0 = tile_y << p.ly; tc.ox0 = tile_x << p.lx;
        return tc;
    };
    // staging slot r = piece i = r*256 + tid = (brick voxel i>>1, quad i&1); lanes (2k, 2k+1) fetch the 32
    // contiguous bytes of one voxel.  dst = LDS offset in floats or -1; inside = voxel lies in the volume.
    auto stage_issue = [&](const TileCoord &tc, int ch, int r, int &dst, bool &inside) {
        const int cglob = ch * CC;
        const float *src; int Csrc, coff;
        if (cglob < p.C0) { src = p.in0; Csrc = p.C0; coff = cglob; }
        else { src = p.in1; Csrc = p.C1; coff = cglob - p.C0; }
        const int i = r * 256 + tid;
        const int bv = i >> 1;
        const int rr = (int)fdiv((uint32_t)bv, p.div_IX);
        const int bx = bv - rr * IX;
        const int bz = (int)fdiv((uint32_t)rr, p.div_IY);
        const int by = rr - bz * IY;
        const int iz = tc.oz0 - 1 + bz, iy = tc.oy0 - 1 + by, ix = tc.ox0 - 1 + bx;
        dst = (i < npieces) ? (i & 1) * pa.plane + bv * 4 : -1;
        inside = (i < npieces) && ((unsigned)iz < (unsigned)p.Di) && ((unsigned)iy < (unsigned)p.Hi) &&
                 ((unsigned)ix < (unsigned)p.Wi);
        size_t off = ((((size_t)tc.n * p.Di + iz) * p.Hi + iy) * p.Wi + ix) * Csrc + coff + qoff;
        off = inside ? off : 0;
        return *(const f32x4 *)(src + off);
    };

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[mf][nf][r] = 0.f;

    const int NFP = NF * pa.w_split;  // fragments per tap in the pack
    const float *wblk = p.wp + (size_t)((int)blockIdx.y / pa.w_split) * p.nchunks * (27 * NFP * 256) +
                        ((int)blockIdx.y % pa.w_split) * (NF * 256) + lane * 4;
    const int co_blk = (int)blockIdx.y * NF * 32;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // prologue: stage (tile, chunk 0) into buffer 0; first BD weight fragments
    TileCoord cur = decode(tile);
#pragma unroll
    for (int r = 0; r < SLOTS; ++r) {
        bool inside; int dst;
        const f32x4 v = stage_issue(cur, 0, r, dst, inside);
        if (dst >= 0) *(f32x4 *)(lds + dst) = inside ? v : zero4;
    }
    f32x4 bq[BD][NF];
#pragma unroll
    for (int k = 0; k < BD; ++k)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bq[k][nf] = *(const f32x4 *)(wblk + (size_t)k * (NFP * 256) + nf * 256);
    __syncthreads();

    int ch = 0, buf = 0;
    while (true) {
        // what comes after (tile, ch)
        int ntile = tile, nch = ch + 1;
        if (nch == p.nchunks) { nch = 0; ntile = tile + nl; }
        const bool have_next = ntile < hi;
        const TileCoord nxt = (nch == 0 && have_next) ? decode(ntile) : cur;
        const int nch_eff = have_next ? nch : ch;  // last chunk of this workgroup: harmless re-read
        const float *bufc = lds + buf * pa.buf_floats;
        float *bufn = lds + (buf ^ 1) * pa.buf_floats;
        const float *wch = wblk + (size_t)ch * (27 * NFP * 256);
        const float *wnx = wblk + (size_t)nch_eff * (27 * NFP * 256);

        f32x4 a[2][MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a[0][mf] = *(const f32x4 *)(bufc + a_base[mf]);
        f32x4 st_v[SLOTS];
        int st_dst[SLOTS];
        bool st_in[SLOTS];

#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) {
                const int nt = tap + 1;
                const int dz = nt / 9, rr = nt - dz * 9, dy = rr / 3, dx = rr - dy * 3;
                const int off = ((dz * IY + dy) * IX + dx) * 4;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a[(tap + 1) & 1][mf] = *(const f32x4 *)(bufc + a_base[mf] + off);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int nf = 0; nf < NF; ++nf)
                        acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[tap % BD][nf][j], a[tap & 1][mf][j], acc[mf][nf], 0, 0, 0);
            {   // refill this ring slot with the fragment BD steps ahead (possibly of the next chunk)
                const int k = tap + BD;
                const float *wsrc = (k < 27) ? wch + (size_t)k * (NFP * 256) : wnx + (size_t)(k - 27) * (NFP * 256);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) bq[tap % BD][nf] = *(const f32x4 *)(wsrc + nf * 256);
            }
            if ((tap & 1) == 0 && tap / 2 < SLOTS) {
                const int r = tap / 2;
                st_v[r] = stage_issue(nxt, nch_eff, r, st_dst[r], st_in[r]);
            }
#pragma unroll
            for (int r = 0; r < SLOTS; ++r) {
                const int wr = 2 * r + 5 < 26 ? 2 * r + 5 : 26;   // write step of slot r
                if (wr == tap && have_next && st_dst[r] >= 0) *(f32x4 *)(bufn + st_dst[r]) = st_in[r] ? st_v[r] : zero4;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();  // next brick complete and visible; this brick free for the chunk after next

        if (ch == p.nchunks - 1) {
            conv_epilogue<MF, NF>(acc, p, cur.n, cur.oz0, cur.oy0, cur.ox0, co_blk);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
        }
        if (!have_next) break;
        tile = ntile; ch = nch; cur = nxt; buf ^= 1;
    }
}

// ------------------------------------------------------------------ direct kernel (any shape)
// One thread per (voxel, cout); used for shapes the MFMA path does not cover and as an
// independent on-device cross-check of it in the parity tests.
__global__ void conv3_direct_kernel(const float *in0, const float *in1, int C0, int C1, const float *w,
                                    const float *bias, float *out, double *stats, int N, int Di, int Hi,
                                    int Wi, int Do, int Ho, int Wo, int Cout, int stride, int act,
                                    float slope) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * Do * Ho * Wo * Cout;
    if (idx >= total)
        return;
    const int co = (int)(idx % Cout);
    size_t v = idx / Cout;
    const int ox = (int)(v % Wo); v /= Wo;
    const int oy = (int)(v % Ho); v /= Ho;
    const int oz = (int)(v % Do);
    const int n = (int)(v / Do);
    const int Cin = C0 + C1;
    float acc = 0.f;
    for (int dz = 0; dz < 3; ++dz) {
        const int iz = oz * stride - 1 + dz;
        if ((unsigned)iz >= (unsigned)Di) continue;
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * stride - 1 + dy;
            if ((unsigned)iy >= (unsigned)Hi) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * stride - 1 + dx;
                if ((unsigned)ix >= (unsigned)Wi) continue;
                const size_t vox = (((size_t)n * Di + iz) * Hi + iy) * Wi + ix;
                const int tap = (dz * 3 + dy) * 3 + dx;
                for (int c = 0; c < C0; ++c)
                    acc = fmaf(in0[vox * C0 + c], w[((size_t)co * Cin + c) * 27 + tap], acc);
                for (int c = 0; c < C1; ++c)
                    acc = fmaf(in1[vox * C1 + c], w[((size_t)co * Cin + C0 + c) * 27 + tap], acc);
            }
        }
    }
    float val = acc + (bias ? bias[co] : 0.f);
    if (act == ACT_LRELU)
        val = val > 0.f ? val : val * slope;
    out[idx] = val;
    if (stats) {
        atomicAdd(stats + ((size_t)n * Cout + co) * 2 + 0, (double)val);
        atomicAdd(stats + ((size_t)n * Cout + co) * 2 + 1, (double)val * (double)val);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Winograd F(2,3) along y.  The f32 convs are MFMA-bound (exact fp32 multiplies, no TF32), so the lever left is the
// number of multiplies: for an output row pair (2p, 2p+1) of the 4x4x32 tile the three dy taps
//     y0 = w0 d0 + w1 d1 + w2 d2,   y1 = w0 d1 + w1 d2 + w2 d3       (d0..d3 = input rows 2p-1 .. 2p+2)
// are computed as        m0 = (d0-d2) w0, m1 = (d1+d2)(w0+w1+w2)/2, m2 = (d2-d1)(w0-w1+w2)/2, m3 = (d1-d3) w2,
//                        y0 = m0+m1+m2,   y1 = m1-m2-m3
// i.e. 4 MFMA K-steps per (dz, dx, channel) for two output rows instead of 6: 2/3 of the direct kernel's MFMAs.
// The row direction is the one that costs nothing else: lane = x as before (conflict-free ds_read_b128), the
// transform is 16 VALU ops per pair and step on registers the lane already holds, a wave (one z plane of the tile)
// reads the SIX brick rows of its two pairs once per step (the direct kernel reads 12 fragments for the same 3 dy
// taps), and the brick, the staging and the epilogue are those of conv3_f32_mfma_kernel<1,16,4,1>.
// The transformed weights U_f = G w are computed in fp64 on the host (conv_weights_upload) and rounded once to fp32.
// Accumulators: 2 pairs x 4 components x 16 = 128 VGPRs per wave.
// Rounding: the 1-D transform adds one fp32 rounding on the input differences and one on U; measured against the
// CPU oracle the logits move by < 2x the direct kernel's summation-order noise (tests/test_gpu_ops.py).
__global__ __launch_bounds__(256, 2) void conv3_f32_wino_kernel(ConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CC = 16, Q = CC / 4, G = CC / 8;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;

    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = (int)fdiv((uint32_t)bid, p.div_tiles_per_n);
    int t = bid - n * (int)p.div_tiles_per_n.d;
    const int tzy = (int)fdiv((uint32_t)t, p.div_tiles_x);
    const int tile_x = t - tzy * p.tiles_x;
    const int tile_z = (int)fdiv((uint32_t)tzy, p.div_tiles_y);
    const int tile_y = tzy - tile_z * p.tiles_y;
    const int oz0 = tile_z << 2, oy0 = tile_y << 2, ox0 = tile_x << 5;  // tile is fixed: 4 x 4 x 32
    const int iz0 = oz0 - 1, iy0 = oy0 - 1, ix0 = ox0 - 1;
    constexpr int IX = 34, IY = 6, IZ = 6;
    constexpr int brickvox = IX * IY * IZ;
    constexpr int npieces = brickvox * Q;
    constexpr int plane = brickvox * 4;

    // wave = z plane of the tile; lane's brick row 0 at tap (dz, dx) = (0, 0), channel quad `half`
    const int a_base = half * plane + (wave * IY * IX + l31) * 4;

    f32x16 acc[2][4];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pr][f][r] = 0.f;

    // packed U: [cout block][chunk][step = dz*3+dx][g][f][lane][4]
    const float *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (9 * G * 4 * 256) + lane * 4;

    for (int ch = 0; ch < p.nchunks; ++ch) {
        const int cglob = ch * CC;
        const float *src;
        int Csrc, coff;
        if (cglob < p.C0) {
            src = p.in0; Csrc = p.C0; coff = cglob;
        } else {
            src = p.in1; Csrc = p.C1; coff = cglob - p.C0;
        }
        src += (size_t)n * p.Di * p.Hi * p.Wi * Csrc + coff;
        constexpr int U = 4;
        for (int i0 = tid; i0 < npieces; i0 += 256 * U) {
            f32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                const int bv = i / Q, q = i - bv * Q;
                const int r = bv / IX, bx = bv - r * IX;
                const int bz = r / IY, by = r - bz * IY;
                const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
                const bool ok = (i < npieces) && ((unsigned)iz < (unsigned)p.Di) &&
                                ((unsigned)iy < (unsigned)p.Hi) && ((unsigned)ix < (unsigned)p.Wi);
                dst[u] = (i < npieces) ? q * plane + bv * 4 : -1;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (ok)
                    val = *(const f32x4 *)(src + ((size_t)(iz * p.Hi + iy) * p.Wi + ix) * Csrc + q * 4);
                v[u] = val;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0)
                    *(f32x4 *)(lds + dst[u]) = v[u];
        }
        __syncthreads();

        const float *wch = wblk + (size_t)ch * (9 * G * 4 * 256);
        f32x4 rows[6], u_cur[4], u_nxt[4];
#pragma unroll
        for (int r = 0; r < 6; ++r) rows[r] = *(const f32x4 *)(lds + a_base + r * IX * 4);
#pragma unroll
        for (int f = 0; f < 4; ++f) u_cur[f] = *(const f32x4 *)(wch + f * 256);

        for (int st = 0; st < 9; ++st) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                // input transform B^T d of both row pairs (rows 0..3 and 2..5)
                f32x4 V[2][4];
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    V[pr][0] = rows[2 * pr] - rows[2 * pr + 2];
                    V[pr][1] = rows[2 * pr + 1] + rows[2 * pr + 2];
                    V[pr][2] = rows[2 * pr + 2] - rows[2 * pr + 1];
                    V[pr][3] = rows[2 * pr + 1] - rows[2 * pr + 3];
                }
                int nst = st, ng = g + 1;
                if (ng == G) { ng = 0; nst = st + 1; }
                if (nst < 9) {
                    const int dz = nst / 3, dx = nst - dz * 3;
                    const int off = (dz * IY * IX + dx) * 4 + ng * 2 * plane;
#pragma unroll
                    for (int r = 0; r < 6; ++r) rows[r] = *(const f32x4 *)(lds + a_base + off + r * IX * 4);
                    const float *wn = wch + (size_t)(nst * G + ng) * (4 * 256);
#pragma unroll
                    for (int f = 0; f < 4; ++f) u_nxt[f] = *(const f32x4 *)(wn + f * 256);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                        for (int f = 0; f < 4; ++f)
                            acc[pr][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(u_cur[f][j], V[pr][f][j], acc[pr][f], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < 4; ++f) u_cur[f] = u_nxt[f];
            }
        }
        __syncthreads();  // brick is overwritten by the next chunk
    }

    // output transform A^T m -> the four rows of this wave's z plane, then the shared epilogue (voxel fragment mf = row)
    f32x16 out[4][1];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            out[2 * pr][0][r] = acc[pr][0][r] + acc[pr][1][r] + acc[pr][2][r];
            out[2 * pr + 1][0][r] = acc[pr][1][r] - acc[pr][2][r] - acc[pr][3][r];
        }
    conv_epilogue<4, 1>(out, p, n, oz0, oy0, ox0, (int)blockIdx.y * 32);
}

// ------------------------------------------------------------------ host side
// Which stride-1 kernel a launch gets (env MI355_CONV_IMPL: "0" = always the one-tile-per-workgroup kernel,
// "1" = always the pipelined persistent kernel, default "auto").  Measured on MI355X (bench.py, config 2):
// the simple kernel with 512-voxel tiles (MF = 4) and 16-channel chunks reaches 0.81 (Cout 32) / 0.87 (Cout >= 64)
// of the f32 MFMA peak and moves ~40 % fewer bytes through the fabric, but needs >= 512 workgroups of that size;
// launches with fewer tiles (deep levels, small batches) and the 8-channel stem go to the pipelined kernel, which
// keeps 2 workgroups per CU busy with 256-voxel tiles.  Auto mode therefore keeps two weight packs per layer.
static int conv_impl() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MI355_CONV_IMPL");
        v = !e ? 2 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2));
    }
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) over (z, y): the nested form of the kernel above.  A wave owns one 2x2 (z, y) output block of
// the 4x4x32 tile at 32 x positions; per (dx, channel) it reads the block's 4x4 input rows, transforms them to 16
// components V = B^T D B (32 packed adds on 2-channel pairs) and issues 16 MFMA K-steps - 4 per output row instead
// of 9 (direct) or 6 (F(2,3) along y only).  The 16 component accumulators are 256 registers, so the kernel runs ONE
// wave per SIMD (512-register budget, accumulators in AGPRs), one persistent workgroup per CU, and everything that a
// second workgroup used to hide is overlapped inside the wave:
//   * a step is (channel quad, dx): one 8-B LDS read per row (lane half h holds channels 2h, 2h+1 of the quad), two
//     MFMAs per component; rows and weights of step s+1 are fetched while the 32 MFMAs of step s run;
//   * the 16-channel halo brick is double-buffered (2 x 77 KB of the 160 KB LDS) and filled by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write): a wave-instruction writes 64 consecutive 16-B
//     slots of one quad plane, out-of-volume voxels read a zero page; the 4 quads of a 64-voxel range are issued
//     back to back so the 128-B lines they share are still in the vector L1.  The chunk-end __syncthreads retires
//     the DMA (vmcnt(0)) before the barrier, which is the ordering a ds_read of DMA data needs.
struct Wino2Args {
    ConvArgs c;
    int total_tiles;
    const float *zeros;  // >= 16 B of zeros in global memory: the source of every out-of-volume piece
    TileOrder order;  // blocked tile order (common.h)
};

// Ablation switches of the diagnostic harness (tools/wino2_probe.hip -DMI355_W2_ABL=<bits>; results are then wrong by
// design): 1 no epilogue, 2 no accumulator reset, 4 no brick DMA, 8 no weight loads, 16 no input transform,
// 32 no global stores, 64 no store phase (read-back, bias, activation, stores).
#ifndef MI355_W2_ABL
#define MI355_W2_ABL 0
#endif
#ifdef MI355_W2_STAMPS
// Diagnostic build only (tools/wino2_probe.hip): per-workgroup cycle sums of the kernel's phases, wave 0, via s_memtime.
// Slots: 0 = chunk prologue, 1 = step loop, 2 = chunk-end drain + barrier, 3 = output transform, 4 = shared epilogue,
// 5 = whole kernel, 6 = chunks, 7 = tiles, 8 = barrier after the epilogue, 9 = accumulator reset + tile set-up.  The shipped kernel contains no stamp.
__device__ unsigned long long w2_stamps[1024 * 16];
#define W2_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define W2_ACC(slot, a, b) do { if (threadIdx.x == 0) w2_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] += (b) - (a); } while (0)
#define W2_CNT(slot) do { if (threadIdx.x == 0) w2_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] += 1; } while (0)
#else
#define W2_T(var)
#define W2_ACC(slot, a, b)
#define W2_CNT(slot)
#endif

constexpr int W2_IX = 34, W2_IY = 6, W2_IZ = 6, W2_BV = W2_IX * W2_IY * W2_IZ;  // 1224 brick voxels
constexpr int W2_BUF_FLOATS = (4 * W2_BV + 56) * 4;  // 4 quad planes + the overrun of the last DMA range
constexpr size_t W2_LDS_BYTES = (size_t)(2 * W2_BUF_FLOATS + 4 * 32 * 2) * sizeof(float);

// PLAIN = bias + LeakyReLU + store only (BatchNorm-folded / un-normalised layers): the whole-line epilogue below, and no
// code for statistics or the fused head in the instantiation - their register demand made the allocator spill loop
// invariants at kernel entry, and the epilogue's reloads missed every cache level after a chunk of streaming DMA traffic
// (9-11k cycles per tile, tools/wino2_probe.hip stamps, round 2).
// EPI: 0 = plain (above), 1 = fused 1x1x1 segmentation head (the network's last conv: only the logits are written),
//      2 = the shared epilogue (Instance/GroupNorm statistics).
template <int EPI>
__global__ __launch_bounds__(256, 1) void conv3_f32_wino2_kernel(Wino2Args pa) {
    constexpr bool PLAIN = EPI == 0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const ConvArgs &p = pa.c;
    constexpr int IX = W2_IX, IY = W2_IY, BV = W2_BV;
    constexpr int plane = BV * 4;  // floats per channel-quad plane
    constexpr int STEPS = 12;      // (quad 0..3) x (dx 0..2) per 16-channel chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it stays in SGPRs
    const int half = lane >> 5;
    const int l31 = lane & 31;
    const int bz = wave >> 1, by = wave & 1;
    float *red = lds + 2 * W2_BUF_FLOATS;

    // this workgroup's tile sequence: XCD group x owns the contiguous range [lo, hi); its workgroups stride through it
    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int nl = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = pa.total_tiles >> 3, r8 = pa.total_tiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + li;
    if (tile >= hi) return;

    struct TileCoord { int n, oz0, oy0, ox0; };
    auto decode = [&](int t) {
        TileCoord tc;
        tc.n = (int)fdiv((uint32_t)t, p.div_tiles_per_n);
        const int tt = t - tc.n * (int)p.div_tiles_per_n.d;
        int tile_x, tile_y, tile_z;
        tile_from_id(tt, pa.order, tile_x, tile_y, tile_z);
        tc.oz0 = tile_z << 2; tc.oy0 = tile_y << 2; tc.ox0 = tile_x << 5;
        return tc;
    };
    // LDS-DMA of slots [q*BV + 64*rng, +64) of `buf`: slot i holds channel quad i / BV of brick voxel i % BV (the tail
    // of range 19 runs into the next plane with exactly the data that belongs there; past the last plane into padding).
    // Every VALU instruction in the tap loop costs the matrix pipe 5-10 cycles (tools/coissue_probe.hip), so the
    // lane's part of the address is kept per range as one packed dword (rz | ry << 4 | bx << 8 | over << 16): a DMA
    // group is ~15 VALU ops + 4 DMAs whose quad offset is the instruction's immediate (the immediate is added to the
    // global AND the LDS address, so the LDS base is moved back by as much).
    unsigned dma_pk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int bv = (wave + 4 * k) * 64 + lane;
        const int over = bv >= BV ? 1 : 0;  // tail of range 19: slots of the NEXT quad plane, voxels 0..55
        bv -= over * BV;
        const int rr = bv / IX, bx = bv - rr * IX;
        const int rz = rr / IY, ry = rr - rz * IY;
        dma_pk[k] = (unsigned)(rz | (ry << 4) | (bx << 8) | (over << 16));
    }
    auto dma_group = [&](const TileCoord &tc, int ch, int k, float *buf) {
        const int rng = wave + 4 * k;
        const int cglob = ch * 16;
        const float *src; int Csrc, coff;
        if (cglob < p.C0) { src = p.in0; Csrc = p.C0; coff = cglob; }
        else { src = p.in1; Csrc = p.C1; coff = cglob - p.C0; }
        // wave-uniform part (SALU); the per-lane part fits 32 bits (host check)
        src += ((((size_t)tc.n * p.Di + (tc.oz0 - 1)) * p.Hi + (tc.oy0 - 1)) * p.Wi + (tc.ox0 - 1)) * (long)Csrc + coff;
#ifdef MI355_W2_RECOMPUTE_PK
        int ln = lane;
        asm volatile("" : "+v"(ln));
        int bvv = rng * 64 + ln;
        const int over = bvv >= BV ? 1 : 0;
        bvv -= over * BV;
        const int rrr = bvv / IX, bx = bvv - rrr * IX;
        const int rz = rrr / IY, ry = rrr - rz * IY;
#else
        unsigned pk = dma_pk[k];
        asm volatile("" : "+v"(pk));  // unpack HERE, every time: hoisted out of the tile loop the unpacked fields and the 64-bit
                                      // offsets built from them are ~40 registers that get spilled to scratch at kernel entry
        const int rz = pk & 15, ry = (pk >> 4) & 15, bx = (pk >> 8) & 255, over = pk >> 16;
#endif
        const bool in_vol = ((unsigned)(tc.oz0 - 1 + rz) < (unsigned)p.Di) && ((unsigned)(tc.oy0 - 1 + ry) < (unsigned)p.Hi) &&
                            ((unsigned)(tc.ox0 - 1 + bx) < (unsigned)p.Wi);
        const int voff = ((rz * p.Hi + ry) * p.Wi + bx) * Csrc + over * 4;
        const float *g = in_vol ? src + voff : pa.zeros;  // the zero page holds 4 quads
        asm volatile("" : "+v"(g));  // one DMA per quad for every lane (a branchy select would issue two and break the vmcnt count)
        float *dst = buf + rng * 64 * 4;
        const float *g3 = g;
        if (k == 4) {  // quad 3 of the last range: its overrun lanes fill padding; keep them inside the tensor
            g3 = (over != 0) ? pa.zeros : g;
            asm volatile("" : "+v"(g3));
        }
#define W2_DMA(G, Q)                                                                                    \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(G),              \
                                     (__attribute__((address_space(3))) void *)(dst + (Q) * (BV * 4 - 4)), 16, (Q) * 16, 0)
        W2_DMA(g, 0);
        W2_DMA(g, 1);
        W2_DMA(g, 2);
        W2_DMA(g3, 3);
#undef W2_DMA
    };

    // floats: block row (0,0) at dx = 0, quad 0, this lane's channel pair
    const int a_base = ((2 * bz * IY + 2 * by) * IX + l31) * 4 + half * 2;

    // packed U: [cout block][chunk][step = q*3 + dx][fragment pair 0..7][lane][f&1][2] - one 16-B load per lane fetches
    // the fragments 2k, 2k+1.  The loads are inline asm with hand-counted waits: while an LDS-DMA is in flight hipcc
    // retires EVERY vector-memory operation (vmcnt(0)) before the first use of an ordinary load's result, which would
    // make each step wait for the brick DMA issued just before (HBM latency > one 2048-cycle step).  Ring uq[2][8],
    // indexed by step parity (no copies: a register must not be read before its wait).
    const float *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (STEPS * 16 * 128);
    const unsigned wl0 = lane * 16, wl1 = lane * 16 + 4096;  // byte offsets of this lane in pairs 0..3 / 4..7
#define W2_ULOAD(DST, VOFF, SBASE, IMM) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(DST) : "v"(VOFF), "s"(SBASE), "n"(IMM) : "memory")
#define W2_UWAIT(U, N)                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                             \
                 : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7])      \
                 :                                                                                                     \
                 : "memory")

#ifdef MI355_W2_STAGGER
    {   // experiment: desynchronise the workgroups so that their store bursts do not coincide
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        const unsigned long long wait = (unsigned long long)(((int)blockIdx.x >> 3) & 7) * (MI355_W2_STAGGER);
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
    }
#endif
    W2_T(t_kernel0);
    TileCoord cur = decode(tile);
#pragma unroll
    for (int k = 0; k < 5; ++k) dma_group(cur, 0, k, lds);
    f32x4 uq[2][8];
    static_for<0, 8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        auto &u0 = uq[0]; const unsigned wl = k < 4 ? wl0 : wl1; const float *wb = wblk;  // (named: asm operands alone do not capture)
        W2_ULOAD(u0[k], wl, wb, (k & 3) * 1024);
    });
    __syncthreads();

    // rows of step `st` (16 x ds_read_b64) / the two halves of the transform V = B^T D B (first along y within each z
    // row a, then along z), one packed add per call so that they can be dealt out between the MFMAs
    auto row_read = [&](const float *bufc, int st, int k) {
        const int q = st / 3, dx = st - q * 3;
        return *(const f32x2 *)(bufc + a_base + q * plane + dx * 4 + ((k >> 2) * IY + (k & 3)) * IX * 4);
    };
    // (packed adds spelled out: left to itself the compiler splits most of them into two v_add_f32, and every VALU
    //  instruction between two MFMAs costs the matrix pipe its issue cycles)
    auto pk_add = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto pk_sub = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto t_op = [&](const f32x2 (&d)[16], f32x2 (&T)[16], int i) {
        const int a = i >> 2, k = i & 3;
        T[i] = k == 0 ? pk_sub(d[a * 4 + 0], d[a * 4 + 2]) : k == 1 ? pk_add(d[a * 4 + 1], d[a * 4 + 2])
             : k == 2 ? pk_sub(d[a * 4 + 2], d[a * 4 + 1]) : pk_sub(d[a * 4 + 1], d[a * 4 + 3]);
    };
    auto v_op = [&](const f32x2 (&T)[16], f32x2 (&V)[16], int m) {
        const int fz = m >> 2, b = m & 3;
        V[m] = fz == 0 ? pk_sub(T[0 * 4 + b], T[2 * 4 + b]) : fz == 1 ? pk_add(T[1 * 4 + b], T[2 * 4 + b])
             : fz == 2 ? pk_sub(T[2 * 4 + b], T[1 * 4 + b]) : pk_sub(T[1 * 4 + b], T[3 * 4 + b]);
    };

    int buf = 0;
    // tile loop outside, chunk loop inside, accumulators scoped to one tile: a conditional reset inside a single
    // flattened loop makes the register allocator spill the 256 accumulators at every back edge
    for (; tile < hi; tile += nl) {
        W2_T(t_t0);
        f32x16 acc[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            if constexpr ((MI355_W2_ABL & 2) != 0) { asm volatile("" : "=a"(acc[f])); continue; }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
        }
        const int ntile = tile + nl;
        const TileCoord nxt_tile = ntile < hi ? decode(ntile) : cur;

#ifdef MI355_W2_STAMPS
#pragma unroll
        for (int f = 0; f < 16; ++f) asm volatile("" : "+a"(acc[f]));
#endif
        W2_T(t_t1);
        W2_ACC(9, t_t0, t_t1);
        for (int ch = 0; ch < p.nchunks; ++ch) {
            const bool last_ch = ch == p.nchunks - 1;
            const bool have_next = !last_ch || ntile < hi;
            const TileCoord nxt = last_ch ? nxt_tile : cur;
            // (without a next chunk the DMAs re-stage the current one into the idle buffer: the wait counts stay fixed)
            const int nch_eff = have_next ? (last_ch ? 0 : ch + 1) : ch;
            const float *bufc = lds + buf * W2_BUF_FLOATS;
            float *bufn = lds + (buf ^ 1) * W2_BUF_FLOATS;
            const float *wch = wblk + (size_t)ch * (STEPS * 16 * 128);
            const float *wnx = wblk + (size_t)nch_eff * (STEPS * 16 * 128);

            // chunk prologue (exposed once per chunk): V of step 0, rows of step 1
            W2_T(t_c0);
            f32x2 d[16], T[16], V[2][16];
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = row_read(bufc, 0, k);
#pragma unroll
            for (int i = 0; i < 16; ++i) t_op(d, T, i);
#pragma unroll
            for (int m = 0; m < 16; ++m) v_op(T, V[0], m);
#pragma unroll
            for (int k = 0; k < 16; ++k) d[k] = row_read(bufc, 1, k);
            __builtin_amdgcn_sched_barrier(0);
            W2_T(t_c1);
            W2_ACC(0, t_c0, t_c1);

            // One step = 32 MFMAs (64 cycles each); everything else of the pipeline is dealt out between them, one
            // scheduling fence per MFMA: the transform of step st+1 (its rows were read during step st-1), the weight
            // loads of step st+1, the row reads of step st+2 and, on even steps, 4 brick DMAs of the next chunk.
            static_for<0, STEPS>([&](auto st_c) {
                constexpr int st = decltype(st_c)::value;
                constexpr int pp = st & 1;
                // this step's weights: everything older than the 4 brick DMAs of the previous step must have landed
                // (the DMAs were issued after the weight loads and may stay in flight: they get a step and a half)
                auto &uc = uq[pp];
                if constexpr (st > 0 && ((st - 1) & 1) == 0 && st - 1 < 10) W2_UWAIT(uc, 4);
                else W2_UWAIT(uc, 0);
                const float *wn = (st + 1 < STEPS) ? wch + (size_t)(st + 1) * (16 * 128) : wnx;
                static_for<0, 32>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    constexpr int f = i & 15, j = i >> 4;
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(uq[pp][f >> 1][(f & 1) * 2 + j], V[pp][f][j], acc[f], 0, 0, 0);
                    // the transform of step st+1 in four bunches of eight packed adds: a gap that holds any VALU work costs
                    // the matrix pipe ~5 cycles plus ~4.4 per instruction (tools/coissue_probe.hip), so 32 adds dealt one
                    // per gap cost twice what they cost in four gaps
                    if constexpr ((MI355_W2_ABL & 16) != 0) {
                    } else if constexpr (st + 1 < STEPS && (i == 0 || i == 2)) {
                        static_for<0, 8>([&](auto u) { t_op(d, T, 4 * i + decltype(u)::value); });
                    } else if constexpr (st + 1 < STEPS && (i == 16 || i == 18)) {
                        static_for<0, 8>([&](auto u) { v_op(T, V[pp ^ 1], 4 * (i - 16) + decltype(u)::value); });
                    }
                    // (the last step of a tile's last chunk fetches nothing: the next tile's first fragments are loaded
                    //  behind the epilogue, so that they are not 32 live registers across it)
                    if constexpr (i < 16 && (i & 1) == 0) {
                        constexpr int k = i >> 1;
                        auto &un = uq[pp ^ 1]; const unsigned wl = k < 4 ? wl0 : wl1; const float *wb = wn;
                        if constexpr ((MI355_W2_ABL & 8) == 0)
                            if (st + 1 < STEPS || !last_ch) W2_ULOAD(un[k], wl, wb, (k & 3) * 1024);
                    }
                    if constexpr (st + 2 < STEPS && i >= 8 && i < 16) {  // two rows per group (one ds_read2_b64), right after
                        d[2 * (i - 8)] = row_read(bufc, st + 2, 2 * (i - 8));          // the T ops released d: the data is
                        d[2 * (i - 8) + 1] = row_read(bufc, st + 2, 2 * (i - 8) + 1);  // needed 16 MFMAs later
                    }
                    if constexpr ((MI355_W2_ABL & 4) == 0 && (st & 1) == 0 && st < 10 && i == 20) dma_group(nxt, nch_eff, st >> 1, bufn);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            W2_T(t_c2);
            W2_ACC(1, t_c1, t_c2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: a ds_read is ordered behind an LDS-DMA only by the issuer's vmcnt + a barrier
            __syncthreads();  // retires this chunk's DMA (vmcnt(0)) and orders it before the next chunk's ds_reads
            buf ^= 1;
            W2_T(t_c3);
            W2_ACC(2, t_c2, t_c3);
            W2_CNT(6);
        }
        W2_T(t_e0);

        // Y = A^T M A: along y within each z component, then along z; rows ordered mf = 2*zrow + yrow (packed over
        // accumulator register pairs: the epilogue is pure VALU time on a SIMD that has nothing else to run)
        if constexpr ((MI355_W2_ABL & 1) != 0) {
#pragma unroll
            for (int f = 0; f < 16; ++f) asm volatile("" :: "a"(acc[f]));
        } else if constexpr (EPI == 0 || EPI == 2) {
            // (EPI == 2, round 3: the Instance/GroupNorm statistics ride on this epilogue - eight registers of running sums
            //  over the values a lane stores - instead of materialising the 64 outputs for the shared conv_epilogue, which
            //  spilled 59 registers and reloaded them from scratch in every tile)
            // Whole-line stores (see conv_epilogue_lines) through the brick buffer that has just been consumed - the other
            // one already holds the next tile's first chunk.  The output transform streams straight into the wave's LDS
            // image [fragment mf][voxel][cout] (8-B writes of two adjacent couts), so the 64 output values never exist
            // as registers; the image is read back by 128-B lines.  One wave alone on its SIMD issues an instruction
            // every 4-5 cycles, so the store phase is written for instruction count: the (z, y) row of a fragment is
            // wave-uniform (scalar address and bounds), the lane's part of the address is one 32-bit offset per tile,
            // the x bound one compare per 8-voxel group.
            // (lane-derived values are rebuilt from the hardware lane id here: as loop invariants of the tile loop they were
            //  hoisted to the kernel entry, spilled, and reloaded from scratch in this epilogue - six dependent round trips
            //  to memory per tile)
            int lane_e;  // volatile: the mbcnt pair is pure and would be hoisted (and spilled) like everything else
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            float *stage = lds + (buf ^ 1) * W2_BUF_FLOATS + wave * (4 * EPI_STAGE_FLOATS);
            float *wr = stage + (lane_e & 31) * EPI_PITCH + 4 * (lane_e >> 5);
            const f32x4 bias = *(const f32x4 *)(p.bias + (int)blockIdx.y * 32 + (lane_e & 7) * 4);  // (lands during the transform)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // (round 4: the accumulators are "redefined" by an empty asm per register pair, so that the pair's 32 v_accvgpr_read
                //  cannot be hoisted above it - the scheduling barrier below alone did not hold them, see conv3d_wino3.hip)
#pragma unroll
                for (int f = 0; f < 16; ++f) asm volatile("" : "+a"(acc[f]));
                f32x2 P[4][2];
#pragma unroll
                for (int fz = 0; fz < 4; ++fz) {
                    const f32x2 a0 = {acc[fz * 4 + 0][r], acc[fz * 4 + 0][r + 1]}, a1 = {acc[fz * 4 + 1][r], acc[fz * 4 + 1][r + 1]};
                    const f32x2 a2 = {acc[fz * 4 + 2][r], acc[fz * 4 + 2][r + 1]}, a3 = {acc[fz * 4 + 3][r], acc[fz * 4 + 3][r + 1]};
                    P[fz][0] = pk_add(pk_add(a0, a1), a2);
                    P[fz][1] = pk_sub(pk_sub(a1, a2), a3);
                }
                const int co = (r & 3) + 8 * (r >> 2);  // + 4 * half: in wr
#pragma unroll
                for (int yy = 0; yy < 2; ++yy) {
                    *(f32x2 *)(wr + (0 + yy) * EPI_STAGE_FLOATS + co) = pk_add(pk_add(P[0][yy], P[1][yy]), P[2][yy]);
                    *(f32x2 *)(wr + (2 + yy) * EPI_STAGE_FLOATS + co) = pk_sub(pk_sub(P[1][yy], P[2][yy]), P[3][yy]);
                }
                // one register pair of the 16 accumulators at a time: left alone the scheduler reads all 256 accumulator
                // registers first, and that peak is what spills the tile loop's invariants to scratch
                __builtin_amdgcn_sched_barrier(0);
            }
            W2_T(t_e1);
            W2_ACC(3, t_e0, t_e1);
            const int srow = lane_e >> 3, spiece = lane_e & 7;
            const float *rd = stage + srow * EPI_PITCH + spiece * 4;
            const int zb = cur.oz0 + 2 * bz, yb = cur.oy0 + 2 * by;
            const int co0 = (int)blockIdx.y * 32;
            const size_t row_elems = (size_t)p.Wo * p.Cout;
            float *obase = p.out + (((size_t)cur.n * p.Do + zb) * p.Ho + yb) * row_elems + (size_t)cur.ox0 * p.Cout + co0;  // wave-uniform
            const unsigned lane_off = (unsigned)(srow * p.Cout + spiece * 4);
            const unsigned t_stride = (unsigned)(8 * p.Cout);
            const f32x2 b01 = {bias[0], bias[1]}, b23 = {bias[2], bias[3]};
            // (a scalar across the tile loop; the vector copy is made here)
            // (the scalar is laundered, not the vector: {slope, slope} built from a loop-invariant scalar is itself loop-invariant - hipcc
            //  made the pair at kernel entry, spilled it, and this epilogue began with a cache-cold scratch reload and a vmcnt(0))
            float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.act == ACT_LRELU ? p.slope : 1.0f)));
            asm volatile("" : "+s"(slope));
            const f32x2 slope2 = {slope, slope};
            bool xok[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) xok[t] = cur.ox0 + 8 * t + srow < p.Wo;
            float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};  // EPI == 2: sum x, sum x^2 of couts 4 spiece .. + 3
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                if constexpr ((MI355_W2_ABL & 64) != 0) continue;
                f32x4 raw[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) raw[t] = *(const f32x4 *)(rd + mf * EPI_STAGE_FLOATS + 8 * t * EPI_PITCH);
                if (zb + (mf >> 1) >= p.Do || yb + (mf & 1) >= p.Ho) continue;  // wave-uniform
                float *rowp = obase + ((size_t)(mf >> 1) * p.Ho + (mf & 1)) * row_elems;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    f32x2 x0 = {raw[t][0], raw[t][1]}, x1 = {raw[t][2], raw[t][3]}, y0, y1;
                    f32x4 val;
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(x0) : "v"(x0), "v"(b01));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(x1) : "v"(x1), "v"(b23));
                    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(slope2));
                    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(slope2));
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[0]) : "v"(x0[0]), "v"(y0[0]));  // bare max: fmaxf would add a
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[1]) : "v"(x0[1]), "v"(y0[1]));  // canonicalising max per value
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[2]) : "v"(x1[0]), "v"(y1[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[3]) : "v"(x1[1]), "v"(y1[1]));
                    if constexpr (EPI == 2) {
                        if (xok[t]) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) { st1[k] += val[k]; st2[k] = fmaf(val[k], val[k], st2[k]); }
                        }
                    }
                    if constexpr ((MI355_W2_ABL & 32) != 0) asm volatile("" :: "v"(val));
                    // sc1: the line leaves the XCD's L2 with the store.  Nothing on this XCD reads it again, and kept in L2 the
                    // output (as many bytes as the input at Cin = Cout) evicts brick lines between the two half-line chunks of
                    // a 32-channel voxel: FETCH_SIZE of the 32 -> 32 layer at 128^3 fell by 38 % with this flag alone.
                    else if (xok[t]) { float *gp = rowp + lane_off + t * t_stride; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 0" :: "v"(gp), "v"(val) : "memory"); }
                }
            }
            if constexpr (EPI == 2) {
                // the eight lanes with the same spiece (lane bits 3..5) hold the same four couts of different voxels
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a = st1[k], b = st2[k];
#pragma unroll
                    for (int m = 8; m < 64; m <<= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
                    if (lane_e < 8) {
                        red[(wave * 32 + 4 * lane_e + k) * 2 + 0] = a;
                        red[(wave * 32 + 4 * lane_e + k) * 2 + 1] = b;
                    }
                }
            }
            W2_T(t_e3);
            // the barrier keeps the next chunk's DMAs of a faster wave out of the staging area until every wave has read its
            // image back.  Raw barrier + lgkmcnt only: a __syncthreads() would also drain the stores (vmcnt(0)).
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if constexpr (EPI == 2) {
                // (red is written again one tile later, behind this tile's barrier and the next tile's chunk barriers)
                if (wave == 0) {  // (scalar test, lane id from the hardware: nothing here is a spilled invariant of the tile loop)
                    const int c = lane_e >> 1, k = lane_e & 1;
                    double tot = 0.0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) tot += (double)red[(w * 32 + c) * 2 + k];
                    atomicAdd(p.stats + ((size_t)cur.n * p.Cout + co0 + c) * 2 + k, quantise_partial(tot, k, (long)p.Do * p.Ho * p.Wo));  // exact, hence order-independent (common.h)
                }
            }
            W2_T(t_e4);
            W2_ACC(8, t_e3, t_e4);
        } else if constexpr (EPI == 1) {
            // Fused segmentation head: logit[c] = sum_cout w[c][cout] * act(y[cout] + b[cout]) + hb[c].  A lane holds 16 couts of
            // its voxel (the other 16 sit in lane ^ 32), so the head is a dot product over registers plus one cross-half add,
            // done inside the output transform one cout pair at a time: the 32-channel feature map exists neither in memory
            // nor as 64 registers.  All weights this lane needs (4 bias quads, ncls x 4 head quads) are fetched once per tile,
            // up front - loading them where they were used cost a memory round trip each (26k cycles per tile).
            int lane_e;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            const int co_l = 4 * (lane_e >> 5);  // this lane's couts: 8 g + co_l + k
            constexpr int KMAX = 4;
            f32x4 bq[4], hq[KMAX][4];
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *(const f32x4 *)(p.bias + (int)blockIdx.y * 32 + 8 * g + co_l);
#pragma unroll
            for (int c = 0; c < KMAX; ++c)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    hq[c][g] = c < p.head_ncls ? *(const f32x4 *)(p.head_w + c * p.Cout + (int)blockIdx.y * 32 + 8 * g + co_l) : f32x4{0.f, 0.f, 0.f, 0.f};
            float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.act == ACT_LRELU ? p.slope : 1.0f)));
            asm volatile("" : "+s"(slope));  // (see the plain epilogue: the vector pair must not become a tile-loop invariant)
            const f32x2 slope2 = {slope, slope};
            f32x2 part[4][KMAX];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int c = 0; c < KMAX; ++c) part[mf][c] = f32x2{0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // (round 4: the accumulators are "redefined" by an empty asm per register pair, so that the pair's 32 v_accvgpr_read
                //  cannot be hoisted above it - the scheduling barrier below alone did not hold them, see conv3d_wino3.hip)
#pragma unroll
                for (int f = 0; f < 16; ++f) asm volatile("" : "+a"(acc[f]));
                f32x2 P[4][2];
#pragma unroll
                for (int fz = 0; fz < 4; ++fz) {
                    const f32x2 a0 = {acc[fz * 4 + 0][r], acc[fz * 4 + 0][r + 1]}, a1 = {acc[fz * 4 + 1][r], acc[fz * 4 + 1][r + 1]};
                    const f32x2 a2 = {acc[fz * 4 + 2][r], acc[fz * 4 + 2][r + 1]}, a3 = {acc[fz * 4 + 3][r], acc[fz * 4 + 3][r + 1]};
                    P[fz][0] = pk_add(pk_add(a0, a1), a2);
                    P[fz][1] = pk_sub(pk_sub(a1, a2), a3);
                }
                const f32x2 b2 = {bq[r >> 2][r & 3], bq[r >> 2][(r & 3) + 1]};
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {  // mf = 2 * zrow + yrow
                    const int yy = mf & 1;
                    f32x2 x = (mf >> 1) == 0 ? pk_add(pk_add(P[0][yy], P[1][yy]), P[2][yy]) : pk_sub(pk_sub(P[1][yy], P[2][yy]), P[3][yy]);
                    f32x2 y;
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(b2));
                    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(slope2));
                    asm("v_max_f32 %0, %1, %2" : "=v"(x[0]) : "v"(x[0]), "v"(y[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(x[1]) : "v"(x[1]), "v"(y[1]));
#pragma unroll
                    for (int c = 0; c < KMAX; ++c) {
                        const f32x2 w2 = {hq[c][r >> 2][r & 3], hq[c][r >> 2][(r & 3) + 1]};
                        asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(part[mf][c]) : "v"(x), "v"(w2));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int zb = cur.oz0 + 2 * bz, yb = cur.oy0 + 2 * by;
            const int ox = cur.ox0 + (lane_e & 31);
            const int64_t Vo = (int64_t)p.Do * p.Ho * p.Wo;
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int oz = zb + (mf >> 1), oy = yb + (mf & 1);
                const bool ok = oz < p.Do && oy < p.Ho && ox < p.Wo && lane_e < 32;
                const int64_t vi = ((int64_t)oz * p.Ho + oy) * p.Wo + ox;
#pragma unroll
                for (int c = 0; c < KMAX; ++c) {
                    if (c >= p.head_ncls) continue;  // uniform
                    float v = part[mf][c][0] + part[mf][c][1];
                    v += __shfl_xor(v, 32);
                    if (ok) p.head_out[((int64_t)cur.n * p.head_ncls + c) * Vo + vi] = v + p.head_b[c];
                }
            }
        }
        cur = nxt_tile;
        if (ntile < hi) {  // the next tile's first weight fragments (same cout block, chunk 0); they land while the accumulators are reset
            static_for<0, 8>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                auto &u0 = uq[0]; const unsigned wl = k < 4 ? wl0 : wl1; const float *wb = wblk;
                W2_ULOAD(u0[k], wl, wb, (k & 3) * 1024);
            });
        }
        W2_T(t_e2);
        W2_ACC(4, t_e0, t_e2);  // transform + stores (slot 3 = the transform part, PLAIN kernel only)
        W2_CNT(7);
    }
    W2_T(t_kernel1);
    W2_ACC(5, t_kernel0, t_kernel1);
#undef W2_ULOAD
#undef W2_UWAIT
}

// ---------------------------------------------------------------------------------------------------------------
// Stride-2 convolution (the encoder's "convolutional pooling", generic_UNet.py:285-288,314-315) as ONE persistent
// workgroup per CU.  What the experiments in DESIGN.md say about these layers: a 2x2x32 output tile needs a 5x5x65 input
// brick (12.7 voxels per output), at one voxel fragment per wave every MFMA group would stream its own KiB of weights
// through the L1, and synchronous staging is not hidden by a second workgroup.  So, with the machinery of the Winograd
// kernel above: the 8-channel brick is double-buffered and filled by LDS-DMA while the previous chunk is computed, and
// the weights come through LDS as well - one dz plane of the chunk (9 taps x 64 couts = 18 KiB) per step in a two-slot
// ring, fetched by DMA one step ahead and shared by the four waves.  The tap loop then holds LDS reads and MFMAs only
// (8 per tap: 4 channel pairs x 2 cout fragments); the VALU work left is the DMA address arithmetic.
// Step = (chunk, dz): 72 MFMAs per wave; the barrier that ends it retires the step's DMAs (vmcnt(0)) before anybody
// reads what they fetched.
// Tile: 2 x 2 x 32 outputs (TXL = 5, brick 5 x 5 x 65) or 2 x 4 x 16 (TXL = 4, brick 5 x 9 x 33) for narrow volumes.
template <int TXL>
struct S2Geom {
    static constexpr int TX = 1 << TXL, TY = 64 >> TXL;                           // a wave = one z plane half: 32 voxels
    static constexpr int IX = 2 * TX + 1, IY = 2 * TY + 1, IZ = 5, BV = IX * IY * IZ;  // brick voxels, 2 quads each
    static constexpr int RANGES = 28, RSTRIDE = (BV - 64 + 26) / 27;              // DMA ranges [RSTRIDE*r, +64) per quad plane
    static constexpr int PAD = 27 * RSTRIDE + 64 - BV;                            // slots written past the last plane
    static constexpr int BUF_FLOATS = (2 * BV + PAD) * 4;
    static constexpr int WSLOT_FLOATS = 9 * 2 * 256;                              // one dz plane: 9 taps x 2 cout fragments
    static constexpr size_t LDS_BYTES = (size_t)(2 * BUF_FLOATS + 2 * WSLOT_FLOATS + 4 * 64 * 2) * sizeof(float);
};

template <int TXL>
__global__ __launch_bounds__(256, 1) void conv3_f32_s2dma_kernel(Wino2Args pa) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ConvArgs &p = pa.c;
    typedef S2Geom<TXL> GM;
    constexpr int IX = GM::IX, IY = GM::IY, BV = GM::BV;
    constexpr int S2_BUF_FLOATS = GM::BUF_FLOATS, S2_WSLOT_FLOATS = GM::WSLOT_FLOATS, S2_RSTRIDE = GM::RSTRIDE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int l31 = lane & 31;
    float *wring = lds + 2 * S2_BUF_FLOATS;

    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int nl = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = pa.total_tiles >> 3, r8 = pa.total_tiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + li;
    if (tile >= hi) return;

    struct TileCoord { int n, oz0, oy0, ox0; };
    auto decode = [&](int t) {
        TileCoord tc;
        tc.n = (int)fdiv((uint32_t)t, p.div_tiles_per_n);
        const int tt = t - tc.n * (int)p.div_tiles_per_n.d;
        int tile_x, tile_y, tile_z;
        tile_from_id(tt, pa.order, tile_x, tile_y, tile_z);
        tc.oz0 = tile_z << 1; tc.oy0 = tile_y * GM::TY; tc.ox0 = tile_x << TXL;
        return tc;
    };

    // brick DMA: 7 ranges per wave, both quads of a range back to back.  Range r = voxels [59r, 59r + 64) of a quad
    // plane; overlaps carry identical data, the tail of range 27 runs into the next plane (or the padding).
    unsigned dma_pk[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        int bv = (wave + 4 * k) * S2_RSTRIDE + lane;
        const int over = bv >= BV ? 1 : 0;
        bv -= over * BV;
        const int rr = bv / IX, bx = bv - rr * IX;
        const int rz = rr / IY, ry = rr - rz * IY;
        dma_pk[k] = (unsigned)(rz | (ry << 4) | (bx << 8) | (over << 16));
    }
    auto dma_brick = [&](const TileCoord &tc, int ch, int k, float *buf) {
        const int rng = wave + 4 * k;
        const int cglob = ch * 8;
        const float *src; int Csrc, coff;
        if (cglob < p.C0) { src = p.in0; Csrc = p.C0; coff = cglob; }
        else { src = p.in1; Csrc = p.C1; coff = cglob - p.C0; }
        src += ((((size_t)tc.n * p.Di + (2 * tc.oz0 - 1)) * p.Hi + (2 * tc.oy0 - 1)) * p.Wi + (2 * tc.ox0 - 1)) * (long)Csrc + coff;
        const unsigned pk = dma_pk[k];
        const int rz = pk & 15, ry = (pk >> 4) & 15, bx = (pk >> 8) & 255, over = pk >> 16;
        const bool in_vol = ((unsigned)(2 * tc.oz0 - 1 + rz) < (unsigned)p.Di) && ((unsigned)(2 * tc.oy0 - 1 + ry) < (unsigned)p.Hi) &&
                            ((unsigned)(2 * tc.ox0 - 1 + bx) < (unsigned)p.Wi);
        const int voff = ((rz * p.Hi + ry) * p.Wi + bx) * Csrc + over * 4;
        const float *g0 = in_vol ? src + voff : pa.zeros;
        const float *g1 = (in_vol && over == 0) ? src + voff : pa.zeros;  // quad 1's overrun lanes fill padding
        asm volatile("" : "+v"(g0), "+v"(g1));
        float *dst = buf + rng * S2_RSTRIDE * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g0,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g1,
                                         (__attribute__((address_space(3))) void *)(dst + BV * 4 - 4), 16, 16, 0);
    };
    // weight DMA: dz plane `dz` of chunk `ch` = 18 KiB contiguous in the pack; KiB i goes to wave i & 3
    const float *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (27 * 2 * 256);
    auto dma_weights = [&](int ch, int dz, float *slot, int i_lo = 0, int i_hi = 5) {
        const float *wsrc = wblk + ((size_t)ch * 27 + dz * 9) * (2 * 256) + lane * 4;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            if (i < i_lo || i >= i_hi) continue;  // (compile-time at the call sites)
            // (no branch: KiB 18 and 19 do not exist - waves 2 and 3 fetch KiB 16 and 17 a second time in the last round,
            //  the same bytes into the same slot as waves 0 and 1)
            const int kib = wave + 4 * i < 18 ? wave + 4 * i : wave + 4 * i - 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(wsrc + kib * 256),
                                             (__attribute__((address_space(3))) void *)(slot + kib * 256), 16, 0, 0);
        }
    };

    // wave w = z plane w >> 1, y rows (w & 1) * TY/2 ..; lane = (y row, x); floats: input voxel (2z, 2y, 2x) at tap 0, quad `half`
    const int ay = (wave & 1) * (GM::TY / 2) + (l31 >> TXL), ax = l31 & (GM::TX - 1);
    const int a_base = half * BV * 4 + (((wave >> 1) * 2 * IY + ay * 2) * IX + 2 * ax) * 4;

    TileCoord cur = decode(tile);
#pragma unroll
    for (int k = 0; k < 7; ++k) dma_brick(cur, 0, k, lds);
    dma_weights(0, 0, wring);
    __syncthreads();

    int buf = 0, wslot = 0;
    for (; tile < hi; tile += nl) {
        f32x16 acc[1][2];
#pragma unroll
        for (int nf = 0; nf < 2; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][nf][r] = 0.f;
        const int ntile = tile + nl;
        const TileCoord nxt_tile = ntile < hi ? decode(ntile) : cur;

        for (int ch = 0; ch < p.nchunks; ++ch) {
            const bool last_ch = ch == p.nchunks - 1;
            const bool have_next = !last_ch || ntile < hi;
            // (without a next chunk the fetches re-stage the current one into the idle buffers: no branch in the MFMA stream)
            const TileCoord nxt = last_ch ? nxt_tile : cur;   // (nxt_tile = cur past the last tile)
            const int nch = have_next ? (last_ch ? 0 : ch + 1) : ch;
            const float *bufc = lds + buf * S2_BUF_FLOATS;
            float *bufn = lds + (buf ^ 1) * S2_BUF_FLOATS;
#pragma unroll
            for (int dz = 0; dz < 3; ++dz) {
                // fetches of the step: the next weight plane and a third of the next chunk's brick - issued from inside the tap loop
                // (round 4: in front of it, their ~11 DMAs and address arithmetic ran with the matrix pipe idle, once per step)
                auto step_fetch = [&](int piece) {
                    if (piece < 2) {  // the weight plane's 18 KiB: rounds 0-1, then 2-4
                        const int lo = piece == 0 ? 0 : 2, hi = piece == 0 ? 2 : 5;
                        if (dz < 2) dma_weights(ch, dz + 1, wring + (wslot ^ 1) * S2_WSLOT_FLOATS, lo, hi);
                        else dma_weights(nch, 0, wring + (wslot ^ 1) * S2_WSLOT_FLOATS, lo, hi);
                    } else {
                        const int q = piece - 1;
                        const int k = dz == 0 ? q - 1 : (dz == 1 ? 2 + q : 4 + q);   // dz 0: ranges 0 1 2, dz 1: 3 4, dz 2: 5 6
                        if (q <= (dz == 0 ? 3 : 2)) dma_brick(nxt, nch, k, bufn);
                    }
                };
                const float *wcur = wring + wslot * S2_WSLOT_FLOATS + lane * 4;
                f32x4 a_cur, a_nxt, b_cur[2], b_nxt[2];
                a_cur = *(const f32x4 *)(bufc + a_base + dz * IY * IX * 4);
                b_cur[0] = *(const f32x4 *)(wcur);
                b_cur[1] = *(const f32x4 *)(wcur + 256);
                // The next tap's fragments are read BEHIND the first two MFMAs of this tap (round 4).  With an LDS-DMA in flight hipcc
                // does not count LDS reads (every wait is lgkmcnt(0)): read at the top of the tap, as before, the three reads
                // were waited for on the spot - their latency exposed nine times a step, which is what kept this kernel's matrix
                // pipe at 0.72.  Now the wait comes in front of the NEXT tap's first MFMA, six MFMAs later.
#pragma unroll
                for (int t = 0; t < 9; ++t) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            acc[0][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b_cur[nf][j], a_cur[j], acc[0][nf], 0, 0, 0);
                            if (j == 0 && nf == 1) {
                                __builtin_amdgcn_sched_barrier(0);
                                if (t + 1 < 9) {
                                    const int dy = (t + 1) / 3, dx = (t + 1) - dy * 3;
                                    a_nxt = *(const f32x4 *)(bufc + a_base + ((dz * IY + dy) * IX + dx) * 4);
                                    b_nxt[0] = *(const f32x4 *)(wcur + (t + 1) * 512);
                                    b_nxt[1] = *(const f32x4 *)(wcur + (t + 1) * 512 + 256);
                                }
                                if (t < 5) step_fetch(t);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    a_cur = a_nxt; b_cur[0] = b_nxt[0]; b_cur[1] = b_nxt[1];
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // explicit: a ds_read is ordered behind an LDS-DMA only by the issuer's vmcnt + a barrier
                __syncthreads();  // retires the step's DMAs (vmcnt(0)); the other weight slot / brick buffer may be read now
                wslot ^= 1;
            }
            buf ^= 1;
        }
        ConvArgs q = p;
        q.lx = TXL; q.ly = 6 - TXL; q.lz = 1;  // voxel v = wave * 32 + lane: x = v & (TX-1), y = (v >> TXL) & (TY-1), z = v >> 6
        conv_epilogue<1, 2>(acc, q, cur.n, cur.oz0, cur.oy0, cur.ox0, (int)blockIdx.y * 64);
        cur = nxt_tile;
    }
}

// MI355_WINOGRAD: 0 = direct kernels only, 1 = F(2,3) along y, 2 (default) = F(2x2,3x3) over (z, y)
static int winograd_mode() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("MI355_WINOGRAD");
        v = !e ? 2 : (e[0] == '0' ? 0 : (e[0] == '1' ? 1 : 2));
    }
    return v;
}

// Winograd-y pack (floats): [cout block of 32][chunk of 16][step = dz*3+dx][g][f 0..3][lane][j] with
//   cout = block*32 + (lane&31), cin = chunk*16 + g*8 + (lane>>5)*4 + j and U_f = (G w)_f over the dy taps,
//   G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]], evaluated in fp64 and rounded once.
static void pack_conv_weights_wino(const float *w, int cin, int cin_pad, int cout, std::vector<float> &out) {
    const int nchunks = cin_pad / 16, nblk = cout / 32;
    out.assign((size_t)nblk * nchunks * 9 * 2 * 4 * 256, 0.f);
    size_t o = 0;
    for (int b = 0; b < nblk; ++b)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int st = 0; st < 9; ++st)
                for (int g = 0; g < 2; ++g)
                    for (int f = 0; f < 4; ++f)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j, ++o) {
                                const int co = b * 32 + (lane & 31);
                                const int ci = ch * 16 + g * 8 + (lane >> 5) * 4 + j;
                                if (ci >= cin) continue;
                                const int dz = st / 3, dx = st % 3;
                                const float *wk = &w[((size_t)co * cin + ci) * 27 + dz * 9 + dx];  // dy stride 3
                                const double w0 = wk[0], w1 = wk[3], w2 = wk[6];
                                const double u = f == 0 ? w0 : f == 1 ? 0.5 * (w0 + w1 + w2) : f == 2 ? 0.5 * (w0 - w1 + w2) : w2;
                                out[o] = (float)u;
                            }
}

// 2-D Winograd pack (floats): [cout block of 32][chunk of 16][step = q*3+dx][f/2][lane][f&1][j 0..1], f = fz*4+fy, with
//   cout = block*32 + (lane&31), cin = chunk*16 + q*4 + (lane>>5)*2 + j, U = G w G^T over the (dz, dy) taps.
static void pack_conv_weights_wino2(const float *w, int cin, int cin_pad, int cout, std::vector<float> &out) {
    const int nchunks = cin_pad / 16, nblk = cout / 32;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    out.assign((size_t)nblk * nchunks * 12 * 16 * 128, 0.f);
    size_t o = 0;
    for (int b = 0; b < nblk; ++b)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int st = 0; st < 12; ++st)
                for (int fp = 0; fp < 8; ++fp)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int fj = 0; fj < 4; ++fj, ++o) {
                            const int f = fp * 2 + (fj >> 1), j = fj & 1;
                            const int co = b * 32 + (lane & 31);
                            const int q = st / 3, dx = st % 3;
                            const int ci = ch * 16 + q * 4 + (lane >> 5) * 2 + j;
                            if (ci >= cin) continue;
                            const float *wk = &w[((size_t)co * cin + ci) * 27 + dx];  // [dz][dy] at strides 9, 3
                            const int fz = f >> 2, fy = f & 3;
                            double u = 0.0;
                            for (int dz = 0; dz < 3; ++dz)
                                for (int dy = 0; dy < 3; ++dy) u += Gm[fz][dz] * Gm[fy][dy] * (double)wk[dz * 9 + dy * 3];
                            out[o] = (float)u;
                        }
}

// Packed layout (floats): [cout_block][chunk][tap][g][nf][lane 0..63][j 0..3] with
//   cout = (cout_block*NF + nf)*32 + (lane&31),  cin = chunk*CC + g*8 + (lane>>5)*4 + j.
static void pack_conv_weights_f32(const float *w, int cin, int cin_pad, int cout, int cc, int nf,
                                  std::vector<float> &out) {
    const int nchunks = cin_pad / cc, G = cc / 8, nblk = cout / (32 * nf);
    out.assign((size_t)nblk * nchunks * 27 * G * nf * 256, 0.f);
    size_t o = 0;
    for (int b = 0; b < nblk; ++b)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int tap = 0; tap < 27; ++tap)
                for (int g = 0; g < G; ++g)
                    for (int f = 0; f < nf; ++f)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j, ++o) {
                                const int co = (b * nf + f) * 32 + (lane & 31);
                                const int ci = ch * cc + g * 8 + (lane >> 5) * 4 + j;
                                out[o] = (ci < cin) ? w[((size_t)co * cin + ci) * 27 + tap] : 0.f;
                            }
}

int conv_weights_upload(const float *w_host, const float *bias_host, int cin, int cin_pad, int cout,
                        int stride, bool keep_plain, ConvWeights *out) {
    MI355_REQUIRE(stride == 1 || stride == 2, "conv stride %d unsupported", stride);
    MI355_REQUIRE(cin_pad >= cin && cin_pad % 4 == 0, "bad cin_pad %d for cin %d", cin_pad, cin);
    ConvWeights cw;
    cw.cin = cin; cw.cin_pad = cin_pad; cw.cout = cout; cw.stride = stride;
    const bool mfma_ok = (cout % 32 == 0) && (cin_pad % 8 == 0);
    if (mfma_ok) {
        // stride 2 bricks are ~8x the output tile: keep them to 8 channels per pass
        cw.pipe = (stride == 1) && conv_impl() != 0;
        cw.cc = (!cw.pipe && stride == 1 && cin_pad % 16 == 0) ? 16 : 8;
        cw.nf = (cout % 64 == 0) ? 2 : 1;
        std::vector<float> packed;
        pack_conv_weights_f32(w_host, cin, cin_pad, cout, cw.cc, cw.nf, packed);
        cw.wp_bytes = packed.size() * sizeof(float);
        MI355_HIP(hipMalloc(&cw.wp_dev, cw.wp_bytes));
        MI355_HIP(hipMemcpy(cw.wp_dev, packed.data(), cw.wp_bytes, hipMemcpyHostToDevice));
        if (stride == 1 && conv_impl() == 2 && cin_pad % 16 == 0) {
            pack_conv_weights_f32(w_host, cin, cin_pad, cout, 16, cw.nf, packed);
            MI355_HIP(hipMalloc(&cw.wp16_dev, packed.size() * sizeof(float)));
            MI355_HIP(hipMemcpy(cw.wp16_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
            if (winograd_mode() != 0) {
                cw.wino2 = winograd_mode() == 2;
                if (cw.wino2) pack_conv_weights_wino2(w_host, cin, cin_pad, cout, packed);
                else pack_conv_weights_wino(w_host, cin, cin_pad, cout, packed);
                MI355_HIP(hipMalloc(&cw.wpw_dev, packed.size() * sizeof(float)));
                MI355_HIP(hipMemcpy(cw.wpw_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
                if (cw.wino2 && conv3d_wino3_enabled()) {  // F(2x2x2, 3x3x3): the launches that are whole 4 x 8 x 8 tiles (conv3d_wino3.hip)
                    pack_conv_weights_wino3(w_host, cin, cin_pad, cout, packed);
                    MI355_HIP(hipMalloc(&cw.wp3_dev, packed.size() * sizeof(float)));
                    MI355_HIP(hipMemcpy(cw.wp3_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
                }
            }
        }
    }
    if (keep_plain || !mfma_ok) {
        // the direct kernel indexes channels of the (possibly zero-padded) input tensors
        std::vector<float> plain((size_t)cout * cin_pad * 27, 0.f);
        for (int co = 0; co < cout; ++co)
            for (int ci = 0; ci < cin; ++ci)
                memcpy(&plain[((size_t)co * cin_pad + ci) * 27], &w_host[((size_t)co * cin + ci) * 27],
                       27 * sizeof(float));
        MI355_HIP(hipMalloc(&cw.w_plain_dev, plain.size() * sizeof(float)));
        MI355_HIP(hipMemcpy(cw.w_plain_dev, plain.data(), plain.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    MI355_HIP(hipMalloc(&cw.bias_dev, cout * sizeof(float)));
    if (bias_host)
        MI355_HIP(hipMemcpy(cw.bias_dev, bias_host, cout * sizeof(float), hipMemcpyHostToDevice));
    else
        MI355_HIP(hipMemset(cw.bias_dev, 0, cout * sizeof(float)));
    *out = cw;
    return MI355_OK;
}

void conv_weights_free(ConvWeights *w) {
    if (w->wp_dev) (void)hipFree(w->wp_dev);
    if (w->wp16_dev) (void)hipFree(w->wp16_dev);
    if (w->wpw_dev) (void)hipFree(w->wpw_dev);
    if (w->wp3_dev) (void)hipFree(w->wp3_dev);
    if (w->bias_dev) (void)hipFree(w->bias_dev);
    if (w->w_plain_dev) (void)hipFree(w->w_plain_dev);
    *w = ConvWeights();
}

// Output tile (power-of-two dims, 128*MF voxels): x as long as the volume allows (up to 32:
// x-consecutive lanes are the conflict-free LDS pattern and give the longest contiguous global
// rows), then the (y, z) split with the smallest input brick.
static void choose_tile(int Do, int Ho, int Wo, int stride, int voxels, int *lz, int *ly, int *lx) {
    auto p2cap = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    const int cz = p2cap(Do), cy = p2cap(Ho), cx = p2cap(Wo);
    const int L = ilog2_exact(voxels);
    int x = cx < 5 ? cx : 5;
    if (x > L) x = L;
    long best = -1;
    int bz = L - x, by = 0;
    for (int y = 0; x + y <= L; ++y) {
        const int z = L - x - y;
        const int oy = y > cy ? y - cy : 0, oz = z > cz ? z - cz : 0;  // lanes wasted past the volume
        const long brick = (long)(((1 << y) - 1) * stride + 3) * (((1 << z) - 1) * stride + 3);
        const long cost = ((long)(oy + oz) << 32) + brick;
        if (best < 0 || cost < best) {
            best = cost; bz = z; by = y;
        }
    }
    *lz = bz; *ly = by; *lx = x;
}

template <int STRIDE, int CC, int MF, int NF>
static int launch_conv(const ConvArgs &a, dim3 grid, size_t lds_bytes, hipStream_t s) {
    auto kern = conv3_f32_mfma_kernel<STRIDE, CC, MF, NF>;
    static size_t attr_bytes = 48 * 1024;  // raise the dynamic-LDS limit on demand (one process per GPU)
    if (lds_bytes > attr_bytes) {
        MI355_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, a);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

template <int MF, int NF>
static int launch_pipe(const PipeArgs &a, dim3 grid, size_t lds_bytes, hipStream_t s) {
    auto kern = conv3_f32_mfma_pipe_kernel<MF, NF>;
    static size_t attr_bytes = 48 * 1024;
    if (lds_bytes > attr_bytes) {
        MI355_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, a);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

static void fill_geometry(ConvArgs &a, int st, int voxels) {
    choose_tile(a.Do, a.Ho, a.Wo, st, voxels, &a.lz, &a.ly, &a.lx);
    const int TX = 1 << a.lx, TY = 1 << a.ly, TZ = 1 << a.lz;
    a.tiles_x = ceil_div(a.Wo, TX); a.tiles_y = ceil_div(a.Ho, TY); a.tiles_z = ceil_div(a.Do, TZ);
    a.IX = (TX - 1) * st + 3; a.IY = (TY - 1) * st + 3; a.IZ = (TZ - 1) * st + 3;
    a.div_tiles_per_n = make_fastdiv(a.tiles_x * a.tiles_y * a.tiles_z);
    a.div_tiles_x = make_fastdiv(a.tiles_x);
    a.div_tiles_y = make_fastdiv(a.tiles_y);
    a.div_IX = make_fastdiv(a.IX);
    a.div_IY = make_fastdiv(a.IY);
}

int conv3d_mfma_f32(const ConvWeights &w, const ConvCall &c, hipStream_t s, const char **kernel_name) {
    const char *kn_dummy;
    if (!kernel_name) kernel_name = &kn_dummy;
    MI355_REQUIRE(w.wp_dev != nullptr, "conv %d->%d has no MFMA weight pack", w.cin, w.cout);
    MI355_REQUIRE(c.C0 + c.C1 == w.cin_pad, "conv input channels %d+%d != %d", c.C0, c.C1, w.cin_pad);
    MI355_REQUIRE(c.C0 % w.cc == 0 && c.C1 % w.cc == 0, "concat split %d/%d not a multiple of %d", c.C0, c.C1, w.cc);
    MI355_REQUIRE(c.C1 == 0 || c.in1 != nullptr, "second conv input missing");
    ConvArgs a;
    a.in0 = c.in0; a.in1 = c.in1; a.C0 = c.C0; a.C1 = c.C1;
    a.wp = w.wp_dev; a.bias = w.bias_dev; a.out = c.out; a.stats = c.stats;
    a.head_w = c.head_w; a.head_b = c.head_b; a.head_out = c.head_out; a.head_ncls = c.head_ncls;
    MI355_REQUIRE(!c.head_out || (w.cout == 32 * w.nf && !c.stats && c.head_ncls >= 1 && c.head_ncls <= 4 && c.head_w && c.head_b),
                  "fused head needs Cout (%d) == one workgroup's couts, no statistics, 1..4 classes", w.cout);
    a.N = c.N; a.Di = c.Di; a.Hi = c.Hi; a.Wi = c.Wi;
    const int st = w.stride;
    a.Do = (c.Di - 1) / st + 1; a.Ho = (c.Hi - 1) / st + 1; a.Wo = (c.Wi - 1) / st + 1;  // k=3, p=1
    a.Cout = w.cout;
    a.nchunks = w.cin_pad / w.cc;
    a.act = c.act; a.slope = c.slope;
    a.ksplit = 1; a.partial = nullptr; a.zero_bias = nullptr; a.out_elems = 0;
    if (w.wp3_dev) {
        bool taken = false;
        MI355_TRY(conv3d_wino3_f32(w, c, s, kernel_name, &taken));
        if (taken) return MI355_OK;
    }
    MI355_REQUIRE(!c.in_scale, "conv %d->%d: a pending input normalisation reached a kernel that cannot apply it", w.cin, w.cout);
    if (w.wpw_dev) {
        // auto mode, large launches: Winograd F(2,3) along y on fixed 4x4x32 tiles, 32 couts per workgroup
        ConvArgs b = a;
        b.lz = 2; b.ly = 2; b.lx = 5;
        b.tiles_x = ceil_div(b.Wo, 32); b.tiles_y = ceil_div(b.Ho, 4); b.tiles_z = ceil_div(b.Do, 4);
        b.IX = 34; b.IY = 6; b.IZ = 6;
        b.div_tiles_per_n = make_fastdiv(b.tiles_x * b.tiles_y * b.tiles_z);
        b.div_tiles_x = make_fastdiv(b.tiles_x);
        b.div_tiles_y = make_fastdiv(b.tiles_y);
        b.div_IX = make_fastdiv(b.IX);
        b.div_IY = make_fastdiv(b.IY);
        const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
        const size_t brick_bytes = (size_t)34 * 6 * 6 * 16 * sizeof(float);
        // the fixed tile wastes lanes on thin volumes: only when every tile dim is at least half used
        if (tiles * (w.cout / 32) >= 512 && tiles < (1l << 30) && b.Wo >= 16 && b.Ho >= 4 && b.Do >= 4 &&
            (long)c.Di * c.Hi * c.Wi * (c.C0 > c.C1 ? c.C0 : c.C1) < (1l << 31) &&
            (!c.head_out || w.cout == 32)) {
            b.wp = w.wpw_dev;
            b.nchunks = w.cin_pad / 16;
            dim3 grid((unsigned)tiles, w.cout / 32);
            static bool attr_set = false;
            if (!attr_set) {
                MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)brick_bytes));
                MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES));
                MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES));
                MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W2_LDS_BYTES));
                attr_set = true;
            }
            float *zeros = nullptr;  // the zero page out-of-volume DMA pieces read
            MI355_TRY(device_scratch(SCR_ZEROS, s, 256, (void **)&zeros, true));
            const int epi = c.head_out ? 1 : (c.stats ? 2 : 0);
            static const char *const w2_names[3] = {"conv3_f32_wino2_kernel<0>", "conv3_f32_wino2_kernel<1>", "conv3_f32_wino2_kernel<2>"};
            *kernel_name = w.wino2 ? w2_names[epi] : "conv3_f32_wino_kernel";
            if (w.wino2) {
                Wino2Args wa;
                wa.c = b; wa.total_tiles = (int)tiles; wa.zeros = zeros;
                wa.order = make_tile_order(b.tiles_x, b.tiles_y, b.tiles_z);
                const int gy = w.cout / 32;
                int gx = 256 / gy;                      // one persistent workgroup per CU
                gx = gx < 8 ? 8 : (gx / 8) * 8;         // multiple of 8: blockIdx.x & 7 labels the XCD group
                const int need = (int)((tiles + 7) / 8) * 8;
                if (gx > need) gx = need;
                if (epi == 0) hipLaunchKernelGGL(conv3_f32_wino2_kernel<0>, dim3(gx, gy), dim3(256), W2_LDS_BYTES, s, wa);
                else if (epi == 1) hipLaunchKernelGGL(conv3_f32_wino2_kernel<1>, dim3(gx, gy), dim3(256), W2_LDS_BYTES, s, wa);
                else hipLaunchKernelGGL(conv3_f32_wino2_kernel<2>, dim3(gx, gy), dim3(256), W2_LDS_BYTES, s, wa);
            } else {
                hipLaunchKernelGGL(conv3_f32_wino_kernel, grid, dim3(256), brick_bytes, s, b);
            }
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
    }
    if (w.wp16_dev) {
        // auto mode: 512-voxel tiles + 16-channel chunks when that fills the chip
        ConvArgs b = a;
        fill_geometry(b, 1, 512);
        const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
        const size_t brick_bytes = (size_t)b.IX * b.IY * b.IZ * 16 * sizeof(float);
        if (tiles * (w.cout / (32 * w.nf)) >= 512 && brick_bytes <= 80 * 1024 && tiles < (1l << 30)) {
            b.wp = w.wp16_dev;
            b.nchunks = w.cin_pad / 16;
            dim3 grid((unsigned)tiles, w.cout / (32 * w.nf));
            if (w.nf == 2) { *kernel_name = "conv3_f32_mfma_kernel<1, 16, 4, 2>"; return launch_conv<1, 16, 4, 2>(b, grid, brick_bytes, s); }
            *kernel_name = "conv3_f32_mfma_kernel<1, 16, 4, 1>";
            return launch_conv<1, 16, 4, 1>(b, grid, brick_bytes, s);
        }
    }
    {
        // Small launches (deep levels: few voxels, hundreds of channels) leave most CUs idle and run one long serial chain
        // of chunks per workgroup: split the channel chunks over blockIdx.z, write raw partial sums, add them in slice
        // order in a finishing pass (deterministic).  Only without run-time statistics / fused head.
        static int splitk = -1;
        if (splitk < 0) { const char *e = getenv("MI355_SPLITK"); splitk = (e && e[0] == '0') ? 0 : 1; }
        if (splitk && !c.stats && !c.head_out && w.cc == 8 && a.nchunks >= 8) {
            ConvArgs b = a;
            const int MFs = st == 1 ? 2 : 1;
            fill_geometry(b, st, 128 * MFs);
            const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
            const int gy = w.cout / (32 * w.nf);
            const long units = tiles * gy;
            const size_t brick_bytes = (size_t)b.IX * b.IY * b.IZ * 8 * sizeof(float);
            // as many slices as still fit the chip in ONE round of workgroups (256 CUs x 2): rounding up (round 2) gave the 8^3 level
            // 80 x 7 = 560 workgroups - 48 of them ran behind the other 512 and doubled the launch's critical path
            int S = (int)(512 / units);
            if (S > a.nchunks / 4) S = a.nchunks / 4;
            if (S > 8) S = 8;
            if (units < 256 && S >= 2 && brick_bytes <= 80 * 1024 && w.cout <= 4096) {
                float *partial = nullptr, *zero_bias = nullptr;
                const long out_elems = (long)c.N * a.Do * a.Ho * a.Wo * w.cout;
                const size_t need = (size_t)S * out_elems * sizeof(float);
                MI355_TRY(device_scratch(SCR_ZERO_BIAS, s, 4096 * sizeof(float), (void **)&zero_bias, true));
                MI355_TRY(device_scratch(SCR_SPLITK_F32, s, need, (void **)&partial));
                b.ksplit = S; b.partial = partial; b.zero_bias = zero_bias; b.out_elems = out_elems;
                const size_t lds_bytes = brick_bytes < 4096 ? 4096 : brick_bytes;
                dim3 grid((unsigned)tiles, gy, S);
                int rc;
                if (st == 1 && w.nf == 1) { *kernel_name = "conv3_f32_mfma_kernel<1, 8, 2, 1> split-K"; rc = launch_conv<1, 8, 2, 1>(b, grid, lds_bytes, s); }
                else if (st == 1) { *kernel_name = "conv3_f32_mfma_kernel<1, 8, 2, 2> split-K"; rc = launch_conv<1, 8, 2, 2>(b, grid, lds_bytes, s); }
                else if (w.nf == 1) { *kernel_name = "conv3_f32_mfma_kernel<2, 8, 1, 1> split-K"; rc = launch_conv<2, 8, 1, 1>(b, grid, lds_bytes, s); }
                else { *kernel_name = "conv3_f32_mfma_kernel<2, 8, 1, 2> split-K"; rc = launch_conv<2, 8, 1, 2>(b, grid, lds_bytes, s); }
                if (rc != MI355_OK) return rc;
                const long total4 = out_elems / 4;
                hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, partial, S, total4, w.cout / 4,
                                   w.bias_dev, c.act, c.slope, c.out);
                MI355_HIP(hipGetLastError());
                return MI355_OK;
            }
        }
    }
    if (w.pipe) {
        // tile size / couts per workgroup: as large as still gives the chip >= ~2 workgroups per CU; launches with
        // few voxels and many channels (deep levels) fall back to 256- then 128-voxel tiles and 32-cout blocks
        int MF = 4, NF = 1;
        long tiles = 0;
        auto geom = [&](int mf) { fill_geometry(a, 1, 128 * mf); tiles = (long)a.tiles_x * a.tiles_y * a.tiles_z * c.N; };
        geom(4);
        if (w.nf == 2 || tiles * (w.cout / 32) < 512 || a.IX * a.IY * a.IZ > 11 * 128) {
            MF = 2; NF = w.nf;
            geom(2);
            if (tiles * (w.cout / (32 * NF)) < 384) NF = 1;
            if (tiles * (w.cout / (32 * NF)) < 384) { MF = 1; geom(1); }
        }
        const int gy = w.cout / (32 * NF);
        MI355_REQUIRE(tiles < (1l << 30), "conv grid too large");
        const int brickvox = a.IX * a.IY * a.IZ;
        MI355_REQUIRE(brickvox <= (MF == 4 ? 11 : 8) * 128, "conv brick of %d voxels exceeds the staging slots", brickvox);
        PipeArgs pa;
        pa.c = a;
        pa.total_tiles = (int)tiles;
        pa.plane = brickvox * 4;
        pa.buf_floats = 2 * pa.plane;
        pa.w_split = w.nf / NF;
        const size_t lds_bytes = (size_t)(2 * pa.buf_floats + 4 * NF * 32 * 2) * sizeof(float);
        MI355_REQUIRE(lds_bytes <= 160 * 1024, "conv brick needs %zu B of LDS", lds_bytes);
        int gx = 512 / gy;                      // ~2 resident workgroups per CU in total
        gx = gx < 8 ? 8 : (gx / 8) * 8;         // multiple of 8: blockIdx.x & 7 labels the XCD group
        const int need = (int)((tiles + 7) / 8) * 8;
        if (gx > need) gx = need;
        dim3 grid(gx, gy);
        if (MF == 4) { *kernel_name = "conv3_f32_mfma_pipe_kernel<4, 1>"; return launch_pipe<4, 1>(pa, grid, lds_bytes, s); }
        if (MF == 2 && NF == 2) { *kernel_name = "conv3_f32_mfma_pipe_kernel<2, 2>"; return launch_pipe<2, 2>(pa, grid, lds_bytes, s); }
        if (MF == 2) { *kernel_name = "conv3_f32_mfma_pipe_kernel<2, 1>"; return launch_pipe<2, 1>(pa, grid, lds_bytes, s); }
        *kernel_name = "conv3_f32_mfma_pipe_kernel<1, 1>";
        return launch_pipe<1, 1>(pa, grid, lds_bytes, s);
    }
    if (st == 2 && w.cc == 8 && w.nf == 2 && !c.head_out) {
        static int s2dma = -1;
        if (s2dma < 0) { const char *e = getenv("MI355_S2_DMA"); s2dma = (e && e[0] == '0') ? 0 : 1; }
        ConvArgs b = a;
        const int txl = b.Wo >= 24 ? 5 : 4;  // 2 x 2 x 32 tiles, or 2 x 4 x 16 on narrow volumes
        const int TX = 1 << txl, TY = 64 >> txl;
        b.lz = 1; b.ly = 6 - txl; b.lx = txl;
        b.tiles_x = ceil_div(b.Wo, TX); b.tiles_y = ceil_div(b.Ho, TY); b.tiles_z = ceil_div(b.Do, 2);
        b.IX = 2 * TX + 1; b.IY = 2 * TY + 1; b.IZ = 5;
        b.div_tiles_per_n = make_fastdiv(b.tiles_x * b.tiles_y * b.tiles_z);
        b.div_tiles_x = make_fastdiv(b.tiles_x);
        b.div_tiles_y = make_fastdiv(b.tiles_y);
        b.div_IX = make_fastdiv(b.IX);
        b.div_IY = make_fastdiv(b.IY);
        const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
        const int gy = w.cout / 64;
        // persistent workgroups need a few tiles each, and the fixed tiles waste lanes on very small volumes
        if (s2dma && tiles * gy >= 768 && tiles < (1l << 30) && b.Wo >= 12 && b.Ho >= 3 &&
            (long)c.Di * c.Hi * c.Wi * (c.C0 > c.C1 ? c.C0 : c.C1) < (1l << 31)) {
            static bool attr_set = false;
            if (!attr_set) {
                MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_s2dma_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2Geom<5>::LDS_BYTES));
                MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_s2dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S2Geom<4>::LDS_BYTES));
                attr_set = true;
            }
            float *zeros = nullptr;
            MI355_TRY(device_scratch(SCR_ZEROS, s, 256, (void **)&zeros, true));
            Wino2Args wa;
            wa.c = b; wa.total_tiles = (int)tiles; wa.zeros = zeros;
            wa.order = make_tile_order(b.tiles_x, b.tiles_y, b.tiles_z);
            int gx = 256 / gy;
            gx = gx < 8 ? 8 : (gx / 8) * 8;
            const int need = (int)((tiles + 7) / 8) * 8;
            if (gx > need) gx = need;
            *kernel_name = txl == 5 ? "conv3_f32_s2dma_kernel<5>" : "conv3_f32_s2dma_kernel<4>";
            if (txl == 5) hipLaunchKernelGGL(conv3_f32_s2dma_kernel<5>, dim3(gx, gy), dim3(256), S2Geom<5>::LDS_BYTES, s, wa);
            else hipLaunchKernelGGL(conv3_f32_s2dma_kernel<4>, dim3(gx, gy), dim3(256), S2Geom<4>::LDS_BYTES, s, wa);
            MI355_HIP(hipGetLastError());
            return MI355_OK;
        }
    }
    int MF = (st == 1) ? 4 : 1;
    fill_geometry(a, st, 128 * MF);
    if (st == 1 && ((long)a.tiles_x * a.tiles_y * a.tiles_z * c.N * (w.cout / (32 * w.nf)) < 512 ||
                    (size_t)a.IX * a.IY * a.IZ * w.cc * 4 > 80 * 1024 || (w.cc == 8 && w.nf == 2))) {
        MF = 2;
        fill_geometry(a, st, 128 * MF);
    }
    const int tiles_per_n = a.tiles_x * a.tiles_y * a.tiles_z;
    MI355_REQUIRE((long)tiles_per_n * c.N < (1l << 30), "conv grid too large");
    const int brickvox = a.IX * a.IY * a.IZ;
    const size_t brick_bytes = (size_t)brickvox * w.cc * sizeof(float);
    const size_t lds_bytes = brick_bytes < 4096 ? 4096 : brick_bytes;  // >= the stats scratch
    MI355_REQUIRE(lds_bytes <= 160 * 1024, "conv brick needs %zu B of LDS", lds_bytes);
    dim3 grid(tiles_per_n * c.N, w.cout / (32 * w.nf));
    if (st == 1 && w.cc == 16 && MF == 4 && w.nf == 2) { *kernel_name = "conv3_f32_mfma_kernel<1, 16, 4, 2>"; return launch_conv<1, 16, 4, 2>(a, grid, lds_bytes, s); }
    if (st == 1 && w.cc == 16 && MF == 4) { *kernel_name = "conv3_f32_mfma_kernel<1, 16, 4, 1>"; return launch_conv<1, 16, 4, 1>(a, grid, lds_bytes, s); }
    if (st == 1 && w.cc == 8 && MF == 4 && w.nf == 1) { *kernel_name = "conv3_f32_mfma_kernel<1, 8, 4, 1>"; return launch_conv<1, 8, 4, 1>(a, grid, lds_bytes, s); }
    if (st == 1 && w.cc == 16 && w.nf == 1) { *kernel_name = "conv3_f32_mfma_kernel<1, 16, 2, 1>"; return launch_conv<1, 16, 2, 1>(a, grid, lds_bytes, s); }
    if (st == 1 && w.cc == 16 && w.nf == 2) { *kernel_name = "conv3_f32_mfma_kernel<1, 16, 2, 2>"; return launch_conv<1, 16, 2, 2>(a, grid, lds_bytes, s); }
    if (st == 1 && w.cc == 8 && w.nf == 1) { *kernel_name = "conv3_f32_mfma_kernel<1, 8, 2, 1>"; return launch_conv<1, 8, 2, 1>(a, grid, lds_bytes, s); }
    if (st == 1 && w.cc == 8 && w.nf == 2) { *kernel_name = "conv3_f32_mfma_kernel<1, 8, 2, 2>"; return launch_conv<1, 8, 2, 2>(a, grid, lds_bytes, s); }
    if (st == 2 && w.cc == 8 && w.nf == 1) { *kernel_name = "conv3_f32_mfma_kernel<2, 8, 1, 1>"; return launch_conv<2, 8, 1, 1>(a, grid, lds_bytes, s); }
    if (st == 2 && w.cc == 8 && w.nf == 2) { *kernel_name = "conv3_f32_mfma_kernel<2, 8, 1, 2>"; return launch_conv<2, 8, 1, 2>(a, grid, lds_bytes, s); }
    set_error("no conv kernel for stride %d cc %d nf %d", st, w.cc, w.nf);
    return MI355_ERR_UNSUPPORTED;
}

int conv3d_direct_f32(const ConvWeights &w, const ConvCall &c, hipStream_t s) {
    MI355_REQUIRE(w.w_plain_dev != nullptr, "conv %d->%d has no plain weights", w.cin, w.cout);
    MI355_REQUIRE(c.C0 + c.C1 == w.cin_pad, "conv input channels %d+%d != %d", c.C0, c.C1, w.cin_pad);
    const int st = w.stride;
    const int Do = (c.Di - 1) / st + 1, Ho = (c.Hi - 1) / st + 1, Wo = (c.Wi - 1) / st + 1;
    const size_t total = (size_t)c.N * Do * Ho * Wo * w.cout;
    const size_t blocks = (total + 255) / 256;
    MI355_REQUIRE(blocks < (1ull << 31), "direct conv grid too large");
    hipLaunchKernelGGL(conv3_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, s, c.in0, c.in1, c.C0, c.C1,
                       w.w_plain_dev, w.bias_dev, c.out, c.stats, c.N, c.Di, c.Hi, c.Wi, Do, Ho, Wo, w.cout, st,
                       c.act, c.slope);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

}  // namespace mi355

// 3x3x3 convolution, fp16 storage / fp32 accumulate, on v_mfma_f32_32x32x16_f16 (gfx950).
// Activations are channel-blocked NDHWC since round 3: [N][C / 8][D][H][W][8] (common.h, "fp16 activation layout") - a
// lane's 16-byte MFMA fragment is one 8-channel block of one voxel, x-consecutive voxels of a block are contiguous.
//
// BASELINE.json configs[2] ("fp16") path of the same op as conv3d.hip (reference
// model_architecture/generic_UNet.py:56,69 run under autocast upstream).  Same GEMM mapping as the
// f32 kernels: D[cout][voxel] = W x X, weights are the MFMA A operand, voxels the B operand, so a lane
// holds one voxel and 16 couts and the epilogue stores 4 consecutive couts (8 B) at a time.
//   * one MFMA contracts a 16-channel chunk of one tap: lane l holds channels 8*(l>>5) .. +7 of
//     cout / voxel l&31, i.e. ONE 16-byte read per fragment;
//   * LDS image of the input brick is planar [8-channel half][brick voxel][16 B]: x-consecutive lanes
//     read consecutive 16-B slots (conflict-free ds_read_b128), 32 B per voxel and chunk;
//   * weights are packed on the host as [cout block][chunk][tap][nf][lane][8 halfs] (1 KiB fragments).
// Kernels: conv3_f16_dma_kernel (stride 1, Cout % 64 == 0, volumes that are whole 8^3 tiles: one wave per SIMD, brick by
// LDS-DMA, weights in a hand-waited register ring, whole-line stores), conv3_f16_mfma_pipe_kernel (the other stride-1
// launches: persistent, double-buffered brick staged through a buffer descriptor while the 27 tap steps of the previous
// chunk run; INAFF variant applies the producer's normalisation while staging, HEAD variant the 1x1x1 head with four
// extra MFMAs; its STRIDE = 2 instantiation with 128-output tiles serves the large stride-2 launches), and the simple
// conv3_f16_mfma_kernel (one tile per workgroup, stride 1|2; also the split-K slices of the deep levels, finished by
// splitk_finish_f16_kernel).
#include "kernels.h"

#include <cstdlib>
#include <cstring>
#include <vector>

namespace mi355 {

typedef _Float16 half_t;

struct ConvArgsH {
    const half_t *in0, *in1;
    const half_t *wp;
    const float *bias;
    half_t *out;
    double *stats;
    int C0, C1;
    int N, Di, Hi, Wi, Do, Ho, Wo, Cout;
    int lx, ly, lz;
    int tiles_x, tiles_y, tiles_z;
    int IX, IY, IZ;
    FastDiv div_tiles_per_n, div_tiles_x, div_tiles_y, div_IX, div_IY;
    int nchunks;
    int act;
    float slope;
    int total_tiles;  // pipelined kernel
    const float *head_w, *head_b;  // fused 1x1x1 head (see ConvCall)
    float *head_out;
    int head_ncls;
    int plane_bytes;  // brickvox * 16
    const void *zeros;  // >= 32 B of zeros in global memory (LDS-DMA kernel: the source of out-of-volume pieces)
    TileOrder order;    // blocked tile order of the LDS-DMA kernel (common.h)
    // split-K (simple kernel, small launches): blockIdx.z = slice of the channel chunks; fp32 partial sums go to `partial`
    int ksplit;
    float *partial;
    long out_elems;
    // normalisation of the PRODUCER fused into this conv's staging (pipelined kernel, INAFF): in0 holds the raw conv output
    // of the previous block, x' = act(x * in_scale[n][c] + in_shift[n][c]) is what the reference feeds this conv
    // (generic_UNet.py:62-72: lrelu(instnorm(conv(x)))); out-of-volume voxels stay exactly zero (padding follows the norm)
    const float *in_scale, *in_shift;  // [N][C0] fp32
    int in_act;
};

// Epilogue shared by both kernels (C/D map of the 32x32 MFMA: col = lane&31 = voxel,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5) = cout).  The accumulators already hold the bias (acc_init_bias), so the
// per-element work is LeakyReLU as max(x, slope*x) and the fp16 conversion; the statistics are only computed when
// the layer needs them (Instance/GroupNorm).
template <int MF, int NF>
__device__ __forceinline__ void acc_init_bias(f32x16 (&acc)[MF][NF], const float *bias, int co_blk, int half) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *(const f32x4 *)(bias + co_blk + nf * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[mf][nf][4 * g + k] = b[k];
        }
}

// SC1: the output lines leave the XCD's L2 with the store (global_store ... sc1).  For the persistent LDS-DMA kernels: nothing
// on this XCD reads the output again, and kept in L2 it evicts the brick lines that the next channel chunk of the same
// voxels is about to fetch a second part of (conv3_f32_wino2_kernel: FETCH_SIZE -38 % with this flag alone).
template <bool SC1>
__device__ __forceinline__ void store_f16x4(half_t *ptr, f16x4 v) {
    if constexpr (SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
    else *(f16x4 *)ptr = v;
}

template <bool SC1>
__device__ __forceinline__ void store_16b(half_t *ptr, u32x4_t v) {
    // (s_nop 1 behind the asm store: a VALU write of a 16-byte store's data registers needs a wait state after its issue)
    if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
    else *(u32x4_t *)ptr = v;
}

// PATH: -1 = statistics iff p.stats (runtime), 0 = never, 1 = always (conv3_f16_dma_kernel: with both paths inlined behind
// its 72-register weight ring the allocator spills)
template <int MF, int NF, bool HEAD = false, bool SC1 = false, int PATH = -1>
__device__ __forceinline__ void conv_epilogue_f16(f32x16 (&acc)[MF][NF], const ConvArgsH &p, int n, int oz0, int oy0,
                                                  int ox0, int co_blk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    const bool lrelu = p.act == ACT_LRELU;
    const float slope = lrelu ? p.slope : 1.0f;  // max(x, 1*x) = x
    if (HEAD) {
        // Fused segmentation head on the matrix cores.  logits[c][voxel] = sum_cout Wh[c][cout] * act[cout][voxel] is one
        // more small GEMM, and the activation tile already sits in the accumulators in exactly the lane = voxel layout of
        // an MFMA B operand: registers 0..7 and 8..15 of a lane are two K = 16 slices (K order = the lane's cout order,
        // matched by the A operand built below).  The head weights stay fp32-accurate: Wh = hi + lo in fp16, two MFMAs
        // each.  4 MFMAs per voxel fragment replace ~100 VALU instructions (FMAs + a cross-half shuffle per class).
        // (separate instantiation: carrying this path in the plain kernels costs them 220 B of spills)
        const int64_t Vo = (int64_t)p.Do * p.Ho * p.Wo;
        // (launched with NF == 1 only: one 32-cout fragment per workgroup)
        f16x8 wh[2][2];  // [K slice][hi / lo]: lane (class = l31, half) holds Wh[class][cout(r)] for r = 8*slice .. +7
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = sl * 8 + j;
                const int co = co_blk + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float w = l31 < p.head_ncls ? p.head_w[l31 * p.Cout + co] : 0.f;
                const half_t hi = (half_t)w;
                wh[sl][0][j] = hi;
                wh[sl][1][j] = (half_t)(w - (float)hi);
            }
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int v = (wave * MF + mf) * 32 + l31;
            const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
            const bool ok = (oz < p.Do) && (oy < p.Ho) && (ox < p.Wo);
            f16x8 act[2];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float y = acc[mf][0][r];
                // the unfused path rounds the activation to fp16 before the head reads it: keep that rounding
                act[r >> 3][r & 7] = (half_t)fmaxf(y, y * slope);
            }
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sl][1], act[sl], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sl][0], act[sl], d, 0, 0, 0);
            }
            // D rows = classes: rows 0..3 are registers 0..3 of the half-0 lanes; column = this lane's voxel
            if (ok && half == 0) {
                const int64_t vi = ((int64_t)oz * p.Ho + oy) * p.Wo + ox;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < p.head_ncls) p.head_out[((int64_t)n * p.head_ncls + c) * Vo + vi] = d[c] + p.head_b[c];
            }
        }
        return;
    }
    if (PATH == 0 || (PATH < 0 && !p.stats)) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int v = (wave * MF + mf) * 32 + l31;
            const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
            const bool ok = (oz < p.Do) && (oy < p.Ho) && (ox < p.Wo);
            // couts co_blk + 32 nf + 8 g + 4 half .. + 3: block (co_blk >> 3) + 4 nf + g of the blocked output; after pair_blocks_f16
            // this lane stores block g + half of a block pair (g = 0, 2), all 16 bytes of its voxel
            const size_t Vo = (size_t)p.Do * p.Ho * p.Wo;
            half_t *orow = p.out + (((size_t)n * (p.Cout >> 3) + (co_blk >> 3) + half) * Vo + ((size_t)oz * p.Ho + oy) * p.Wo + ox) * 8;
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    // max(x, slope * x) with packed multiplies and bare v_max_f32 (fmaxf adds a canonicalising max per
                    // value; every VALU instruction of the epilogue is time the matrix pipe stands still)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 slope2 = {slope, slope};
                    f16x4 val2[2];
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            const int g = gp + gi;
                            const f32x2 x = {acc[mf][nf][4 * g + k], acc[mf][nf][4 * g + k + 1]};
                            f32x2 y;
                            float m0, m1;
                            asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(slope2));
                            asm("v_max_f32 %0, %1, %2" : "=v"(m0) : "v"(x[0]), "v"(y[0]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(m1) : "v"(x[1]), "v"(y[1]));
                            val2[gi][k] = (half_t)m0;
                            val2[gi][k + 1] = (half_t)m1;
                        }
                    const u32x4_t v16 = pair_blocks_f16(val2[0], val2[1]);
                    if (ok) store_16b<SC1>(orow + (size_t)(nf * 4 + gp) * Vo * 8, v16);
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
        const float in = ok ? 1.f : 0.f;  // a voxel beyond a ragged edge adds nothing
        const size_t Vo = (size_t)p.Do * p.Ho * p.Wo;
        half_t *orow = p.out + (((size_t)n * (p.Cout >> 3) + (co_blk >> 3) + half) * Vo + ((size_t)oz * p.Ho + oy) * p.Wo + ox) * 8;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                f16x4 val2[2];
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const int g = gp + gi;
                        f32x2 x = {acc[mf][nf][4 * g + k], acc[mf][nf][4 * g + k + 1]};
                        if (lrelu) x = f32x2{fmaxf(x[0], x[0] * slope), fmaxf(x[1], x[1] * slope)};  // (wave-uniform; conv -> norm -> LeakyReLU has none here)
                        val2[gi][k] = (half_t)x[0];
                        val2[gi][k + 1] = (half_t)x[1];
                        const f32x2 m = x * f32x2{in, in};
                        s1[nf][2 * g + (k >> 1)] += m;  // v_pk_add_f32 / v_pk_fma_f32: one instruction per value pair
                        s2[nf][2 * g + (k >> 1)] = __builtin_elementwise_fma(m, m, s2[nf][2 * g + (k >> 1)]);
                    }
                const u32x4_t v16 = pair_blocks_f16(val2[0], val2[1]);  // (every lane active here: only the store is predicated)
                if (ok) store_16b<SC1>(orow + (size_t)(nf * 4 + gp) * Vo * 8, v16);
            }
        }
    }
    // (round 3) transposing reduction over the 32 voxel lanes (common.h): every lane ends with the total of ONE (cout, statistic)
    // of this wave's MF * 32 voxels and adds it itself - no LDS, no barrier.  Quantised partials: exact additions in any order.
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const float tot = half32_reduce_scatter(s1[nf], s2[nf], lane);
        const int r = stat_slot_r(lane), k = (lane >> 4) & 1;
        const int c = nf * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        atomicAdd(p.stats + ((size_t)n * p.Cout + co_blk + c) * 2 + k, quantise_partial((double)tot, k, (long)p.Do * p.Ho * p.Wo));
    }
}

// ------------------------------------------------------------------ simple kernel (stride 1 | 2)
template <int STRIDE, int MF, int NF>
__global__ __launch_bounds__(256, 2) void conv3_f16_mfma_kernel(ConvArgsH p) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int n = (int)fdiv((uint32_t)bid, p.div_tiles_per_n);
    const int t = bid - n * (int)p.div_tiles_per_n.d;
    const int tzy = (int)fdiv((uint32_t)t, p.div_tiles_x);
    const int tile_x = t - tzy * p.tiles_x;
    const int tile_z = (int)fdiv((uint32_t)tzy, p.div_tiles_y);
    const int tile_y = tzy - tile_z * p.tiles_y;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    const int oz0 = tile_z << p.lz, oy0 = tile_y << p.ly, ox0 = tile_x << p.lx;
    const int iz0 = oz0 * STRIDE - 1, iy0 = oy0 * STRIDE - 1, ix0 = ox0 * STRIDE - 1;
    const int IX = p.IX, IY = p.IY;
    const int brickvox = IX * IY * p.IZ;
    const int npieces = 2 * brickvox;

    int a_base[MF];  // bytes
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = (wave * MF + mf) * 32 + l31;
        const int x = v & TXm, y = (v >> p.lx) & TYm, z = v >> (p.lx + p.ly);
        a_base[mf] = half * p.plane_bytes + ((z * STRIDE * IY + y * STRIDE) * IX + x * STRIDE) * 16;
    }
    f32x16 acc[MF][NF];
    const int ks = (int)blockIdx.z;
    const int ch_begin = p.ksplit > 1 ? ks * p.nchunks / p.ksplit : 0, ch_end = p.ksplit > 1 ? (ks + 1) * p.nchunks / p.ksplit : p.nchunks;
    if (ks == 0) {
        acc_init_bias<MF, NF>(acc, p.bias, (int)blockIdx.y * NF * 32, half);  // the bias travels with slice 0
    } else {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
    }

    const half_t *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (27 * NF * 512) + lane * 8;
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int cglob = ch * 16;
        const half_t *src; int Csrc, coff;
        if (cglob < p.C0) { src = p.in0; Csrc = p.C0; coff = cglob; }
        else { src = p.in1; Csrc = p.C1; coff = cglob - p.C0; }
        const size_t Vi = (size_t)p.Di * p.Hi * p.Wi;
        src += ((size_t)n * (Csrc >> 3) + (coff >> 3)) * Vi * 8;  // block coff / 8 of sample n; piece q = the next block
        constexpr int U = 4;
        for (int i0 = tid; i0 < npieces; i0 += 256 * U) {
            f32x4 v[U];
            int dst[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + u * 256;
                const int bv = i >> 1, q = i & 1;
                const int r = (int)fdiv((uint32_t)bv, p.div_IX);
                const int bx = bv - r * IX;
                const int bz = (int)fdiv((uint32_t)r, p.div_IY);
                const int by = r - bz * IY;
                const int iz = iz0 + bz, iy = iy0 + by, ix = ix0 + bx;
                const bool ok = (i < npieces) && ((unsigned)iz < (unsigned)p.Di) && ((unsigned)iy < (unsigned)p.Hi) &&
                                ((unsigned)ix < (unsigned)p.Wi);
                dst[u] = (i < npieces) ? q * p.plane_bytes + bv * 16 : -1;
                f32x4 val = {0.f, 0.f, 0.f, 0.f};
                if (ok) val = *(const f32x4 *)(src + ((size_t)q * Vi + (size_t)(iz * p.Hi + iy) * p.Wi + ix) * 8);
                v[u] = val;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (dst[u] >= 0) *(f32x4 *)(lds_raw + dst[u]) = v[u];
        }
        __syncthreads();
        const half_t *wch = wblk + (size_t)ch * (27 * NF * 512);
        f16x8 a_cur[MF], b_cur[NF], a_nxt[MF], b_nxt[NF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a_cur[mf] = *(const f16x8 *)(lds_raw + a_base[mf]);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) b_cur[nf] = *(const f16x8 *)(wch + nf * 512);
        for (int tap = 0; tap < 27; ++tap) {
            if (tap + 1 < 27) {
                const int nt = tap + 1;
                const int dz = nt / 9, rr = nt - dz * 9, dy = rr / 3, dx = rr - dy * 3;
                const int off = ((dz * IY + dy) * IX + dx) * 16;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a_nxt[mf] = *(const f16x8 *)(lds_raw + a_base[mf] + off);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) b_nxt[nf] = *(const f16x8 *)(wch + (size_t)nt * (NF * 512) + nf * 512);
            }
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_cur[nf], a_cur[mf], acc[mf][nf], 0, 0, 0);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) a_cur[mf] = a_nxt[mf];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) b_cur[nf] = b_nxt[nf];
        }
        __syncthreads();
    }
    if (p.ksplit > 1) {
        // raw fp32 partial sums of this channel slice, NDHWC like the output; activation / rounding in splitk_finish_f16_kernel
        float *part = p.partial + (size_t)ks * p.out_elems;
        const int co_blk = (int)blockIdx.y * NF * 32;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int v = (wave * MF + mf) * 32 + l31;
            const int oz = oz0 + (v >> (p.lx + p.ly)), oy = oy0 + ((v >> p.lx) & TYm), ox = ox0 + (v & TXm);
            if ((oz < p.Do) && (oy < p.Ho) && (ox < p.Wo)) {
                float *orow = part + ((((size_t)n * p.Do + oz) * p.Ho + oy) * p.Wo + ox) * p.Cout + co_blk + 4 * half;
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *(f32x4 *)(orow + nf * 32 + 8 * g) = f32x4{acc[mf][nf][4 * g], acc[mf][nf][4 * g + 1], acc[mf][nf][4 * g + 2], acc[mf][nf][4 * g + 3]};
            }
        }
        return;
    }
    conv_epilogue_f16<MF, NF>(acc, p, n, oz0, oy0, ox0, (int)blockIdx.y * NF * 32);
}

// out = fp16(act(sum_s partial[s])), slices added in slice order (slice 0 carries the bias)
// (partial sums are plain NDHWC fp32, 4 consecutive couts per thread; the fp16 output is channel-blocked, common.h)
__global__ __launch_bounds__(256) void splitk_finish_f16_kernel(const float *partial, int S, long total4, int act, float slope, half_t *out,
                                                                int C, long V) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    f32x4 v = *(const f32x4 *)(partial + i * 4);
    for (int s2 = 1; s2 < S; ++s2) {
        const f32x4 t = *(const f32x4 *)(partial + ((long)s2 * total4 + i) * 4);
        v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
    }
    const float sl = act == ACT_LRELU ? slope : 1.0f;
    f16x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (half_t)fmaxf(v[k], v[k] * sl);
    const long e = i * 4, nv = e / C;
    const int c = (int)(e - nv * C);
    *(f16x4 *)(out + b8_index(nv / V, c, nv % V, C, V)) = o;
}

// ------------------------------------------------------------------ pipelined persistent kernel (stride 1)
// STRIDE = 2 (round 2): the same kernel on the encoder's stride-2 convs with 128-output tiles (MF = 1).  Its staging fetches
// the 32 contiguous bytes of a voxel's chunk with two adjacent lanes, so an instruction touches half the 128-B lines that
// the LDS-DMA stride-2 kernel's plane-wise 16-B pieces touch - and the line rate of the L1, not the matrix pipe or the
// L2, is what held that kernel at 0.13-0.16 of the fp16 peak.
// WHOLE (round 3): the volume is a whole number of tiles (host check), i.e. every network layer at the reference's patch
// sizes: the instantiation carries neither the exact per-axis test of overhanging tiles nor its 11 wave-uniform branches
// per chunk (each a basic-block boundary with waits of its own); INAFF is only built with WHOLE.
template <int MF, int NF, bool HEAD = false, bool INAFF = false, int STRIDE = 1, bool WHOLE = false>
__global__ __launch_bounds__(256, 2) void conv3_f16_mfma_pipe_kernel(ConvArgsH p) {
    static_assert(!INAFF || WHOLE, "the fused input normalisation is built for whole-tile volumes only");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    constexpr int SLOTS = STRIDE == 2 ? 13 : (MF == 4 ? 11 : 8);  // 16-B staging pieces per thread and chunk
    constexpr int BD = 3;                    // weight fragments fetched BD tap-steps ahead; the ring phase must
                                             // be the same in every chunk, so BD divides 27
    constexpr int FLIGHT = 7;               // tap-steps between a staging fetch and its LDS write
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int TXm = (1 << p.lx) - 1, TYm = (1 << p.ly) - 1;
    const int IX = p.IX, IY = p.IY;
    const int brickvox = IX * IY * p.IZ;
    const int npieces = 2 * brickvox;
    const int buf_bytes = 2 * p.plane_bytes;

    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int nl = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = p.total_tiles >> 3, r8 = p.total_tiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + li;
    if (tile >= hi) return;

    int a_base[MF];  // bytes
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = (wave * MF + mf) * 32 + l31;
        const int x = v & TXm, y = (v >> p.lx) & TYm, z = v >> (p.lx + p.ly);
        a_base[mf] = half * p.plane_bytes + ((z * STRIDE * IY + y * STRIDE) * IX + x * STRIDE) * 16;
    }
    const unsigned qoff = (unsigned)(tid & 1) * (unsigned)((long)p.Di * p.Hi * p.Wi * 16);  // bytes to the lane's 8-channel block (host check: < 2^31)

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
    // Tile-invariant part of every staging slot, computed once per kernel: the voxel offset of the piece relative to
    // the brick origin voxel (24 bits, per lane) and the brick faces it lies on (bit 0/1: z lo/hi, 2/3: y, 4/5: x;
    // bit 6 = unused slot).  A fetch is then five VALU ops: AND + compare (faces), AND + 24-bit multiply-add (byte
    // offset), one select - the fetch itself is a raw BUFFER load from a descriptor based at the brick origin, and an
    // out-of-volume piece gets the offset 0xffffffff: the hardware range check returns zeros, so neither a 64-bit
    // address select nor a select on the loaded value is needed.  (VALU instructions are what the tap loop pays
    // for: each one costs the matrix pipe 5-10 cycles, tools/coissue_probe.hip.)
    int st_pk[SLOTS];  // relative voxel offset (24 bits) | face bits << 24
    const int Cs0 = p.C0;  // relative offsets are kept for in0's channel stride; in1 (concat half) rescales below
#pragma unroll
    for (int r = 0; r < SLOTS; ++r) {
        const int i = r * 256 + tid;
        const int bv = i >> 1;
        const int rr = (int)fdiv((uint32_t)bv, p.div_IX);
        const int bx = bv - rr * IX;
        const int bz = (int)fdiv((uint32_t)rr, p.div_IY);
        const int by = rr - bz * IY;
        const int rel = (bz * p.Hi + by) * p.Wi + bx;   // in voxels (< 2^24: checked on the host)
        const int face = (bz == 0) | ((bz == p.IZ - 1) << 1) | ((by == 0) << 2) | ((by == IY - 1) << 3) | ((bx == 0) << 4) |
                         ((bx == IX - 1) << 5);
        st_pk[r] = (i < npieces) ? (rel | (face << 24)) : (64 << 24);
    }
    (void)Cs0;
    // which faces of the brick of tile tc stick out of the volume (wave-uniform)
    auto tile_faces = [&](const TileCoord &tc) {
        // a brick spans [o0-1, o0+T+1): the low face is outside iff o0 == 0, the high face iff o0 + T >= dim.
        // voxels beyond the high face + 1 (tiles overhanging a ragged volume) are caught by the exact test below.
        // (stride S: the brick's last input voxel is S (o0 + T - 1) + 1)
        return (tc.oz0 == 0) | ((STRIDE * (tc.oz0 + (1 << p.lz) - 1) + 1 >= p.Di) << 1) | ((tc.oy0 == 0) << 2) |
               ((STRIDE * (tc.oy0 + (1 << p.ly) - 1) + 1 >= p.Hi) << 3) | ((tc.ox0 == 0) << 4) |
               ((STRIDE * (tc.ox0 + (1 << p.lx) - 1) + 1 >= p.Wi) << 5);
    };
    auto tile_ragged = [&](const TileCoord &tc) {
        if constexpr (WHOLE) return false;
        else return (bool)((STRIDE * (tc.oz0 + (1 << p.lz) - 1) + 1 > p.Di) | (STRIDE * (tc.oy0 + (1 << p.ly) - 1) + 1 > p.Hi) |
                           (STRIDE * (tc.ox0 + (1 << p.lx) - 1) + 1 > p.Wi));
    };
    const int dst0 = (tid & 1) * p.plane_bytes + (tid >> 1) * 16;  // LDS byte offset of slot 0; slot r is 128 voxels further
    // (tried in round 3 and dropped: a 4-KiB junk area so that slots beyond the brick are written unconditionally instead of
    //  behind an exec-mask branch - the extra LDS pushed the 512-voxel-tile kernels from two workgroups per CU to one,
    //  1018 -> 826 TFLOP/s)
    auto stage_issue = [&](const TileCoord &tc, int faces, bool ragged, int ch, int r) {
        const int cglob = ch * 16;
        const half_t *src; int Csrc, coff;
        if (cglob < p.C0) { src = p.in0; Csrc = p.C0; coff = cglob; }
        else { src = p.in1; Csrc = p.C1; coff = cglob - p.C0; }
        const int pk = st_pk[r];
        bool inside = (pk & ((faces | 64) << 24)) == 0;
        if (ragged) {  // rare: tile overhangs the volume by more than the halo -> exact per-axis test
            const int bv = (r * 256 + tid) >> 1;
            const int rr = (int)fdiv((uint32_t)bv, p.div_IX);
            const int bx = bv - rr * IX;
            const int bz = (int)fdiv((uint32_t)rr, p.div_IY);
            const int by = rr - bz * IY;
            inside = ((pk >> 30) == 0) && ((unsigned)(STRIDE * tc.oz0 - 1 + bz) < (unsigned)p.Di) && ((unsigned)(STRIDE * tc.oy0 - 1 + by) < (unsigned)p.Hi) &&
                     ((unsigned)(STRIDE * tc.ox0 - 1 + bx) < (unsigned)p.Wi);
        }
        // wave-uniform descriptor based at the brick origin voxel (may lie one voxel outside the tensor: only pieces
        // inside the volume are addressed through it)
        // (blocked tensors: block coff / 8 of sample n, brick origin voxel within the sample; the lane's 8-channel half is the
        //  next block, Vi voxels further)
        const long Vi = (long)p.Di * p.Hi * p.Wi;
        const long base_vox = ((long)(STRIDE * tc.oz0 - 1) * p.Hi + (STRIDE * tc.oy0 - 1)) * p.Wi + (STRIDE * tc.ox0 - 1);
        const half_t *sbase = src + (((long)tc.n * (Csrc >> 3) + (coff >> 3)) * Vi + base_vox) * 8;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)sbase, 0, 0x7fffffff, 0x00020000);
        const unsigned off = (((unsigned)pk & 0xffffffu) << 4) + qoff;
        const unsigned voff = inside ? off : 0xffffffffu;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0);
        return __builtin_bit_cast(f32x4, v);
    };
    auto slot_valid = [&](int r) { return (st_pk[r] >> 30) == 0; };
    // INAFF: whether slot r's piece of the brick being staged lies inside the volume (same test as stage_issue)
    auto slot_inside = [&](const TileCoord &tc, int faces, bool ragged, int r) {
        const int pk = st_pk[r];
        bool inside = (pk & ((faces | 64) << 24)) == 0;
        if (ragged) {
            const int bv = (r * 256 + tid) >> 1;
            const int rr = (int)fdiv((uint32_t)bv, p.div_IX);
            const int bx = bv - rr * IX;
            const int bz = (int)fdiv((uint32_t)rr, p.div_IY);
            const int by = rr - bz * IY;
            inside = ((pk >> 30) == 0) && ((unsigned)(STRIDE * tc.oz0 - 1 + bz) < (unsigned)p.Di) && ((unsigned)(STRIDE * tc.oy0 - 1 + by) < (unsigned)p.Hi) &&
                     ((unsigned)(STRIDE * tc.ox0 - 1 + bx) < (unsigned)p.Wi);
        }
        return inside;
    };
    // INAFF: the producer's normalisation + activation on one staged piece (8 channels of one voxel), all in packed fp16:
    // v_pk_fma_f16 (x * scale + shift, fused: one rounding), v_pk_mul_f16 + v_pk_max_f16 (LeakyReLU) - 12 VALU per piece.
    // The kernel is VALU-bound with this work in it (two waves per SIMD: SQ counters), so every instruction
    // counts: scale and shift are held as fp16 (that rounds them to 2^-11 relative, the precision of the activation
    // itself), and out-of-volume pieces are not selected to zero here but overwritten in LDS by a second, masked write.
    auto in_affine = [&](f32x4 raw, f16x8 sc, f16x8 sh, half_t slope_h) {
        f16x8 y = __builtin_elementwise_fma(__builtin_bit_cast(f16x8, raw), sc, sh);
        const f16x8 sl8 = {slope_h, slope_h, slope_h, slope_h, slope_h, slope_h, slope_h, slope_h};
        y = __builtin_elementwise_max(y, y * sl8);
        return __builtin_bit_cast(f32x4, y);
    };
    const float slope_in = p.in_act == ACT_LRELU ? p.slope : 1.0f;
    // per-(sample, channel) scale / shift of the 8 channels this thread stages in chunk `ch` of sample n (in0 only: the
    // second half of a virtual concat is a skip tensor that was normalised when it was written)
    auto load_aff = [&](int n, int ch, f16x8 &sc, f16x8 &sh) {
        const int c0 = ch * 16 + (tid & 1) * 8;
        const bool from_in0 = c0 < p.C0;
        const float *ps = p.in_scale + (size_t)n * p.C0 + (from_in0 ? c0 : 0), *ph = p.in_shift + (size_t)n * p.C0 + (from_in0 ? c0 : 0);
        const f32x4 s0 = *(const f32x4 *)ps, s1 = *(const f32x4 *)(ps + 4), h0 = *(const f32x4 *)ph, h1 = *(const f32x4 *)(ph + 4);
        // (chunks of the second input get the identity: scale 1, shift 0, slope 1 - no branch in the tap loop)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sc[j] = from_in0 ? (half_t)s0[j] : (half_t)1.f; sc[4 + j] = from_in0 ? (half_t)s1[j] : (half_t)1.f;
            sh[j] = from_in0 ? (half_t)h0[j] : (half_t)0.f; sh[4 + j] = from_in0 ? (half_t)h1[j] : (half_t)0.f;
        }
        return (half_t)(from_in0 ? slope_in : 1.0f);
    };

    f32x16 acc[MF][NF];
    acc_init_bias<MF, NF>(acc, p.bias, (int)blockIdx.y * NF * 32, half);

    // wave-uniform weight base (SGPRs) + a 32-bit per-lane offset: the tap / chunk offsets are scalar arithmetic
    const half_t *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (27 * NF * 512);
    // weights: raw buffer loads, per-lane offset in a VGPR that never changes, tap / chunk offset in an SGPR (soffset):
    // no VALU address arithmetic in the tap loop
    const unsigned wlane = lane * 16;
    typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wblk, 0, 0x7fffffff, 0x00020000);
    auto wload = [&](unsigned soff_halfs) {
        return __builtin_bit_cast(f16x8, (wu32x4)__builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, soff_halfs * 2, 0));
    };
    const int co_blk = (int)blockIdx.y * NF * 32;

    TileCoord cur = decode(tile);
    {
        f16x8 sc0, sh0;
        half_t sl0 = (half_t)1.f;
        if constexpr (INAFF) sl0 = load_aff(cur.n, 0, sc0, sh0);
#pragma unroll
        for (int r = 0; r < SLOTS; ++r) {
            f32x4 v = stage_issue(cur, tile_faces(cur), tile_ragged(cur), 0, r);
            if constexpr (INAFF) {  // padding follows the norm: out-of-volume pieces are zero, not act(shift)
                v = in_affine(v, sc0, sh0, sl0);
                const bool in = slot_inside(cur, tile_faces(cur), tile_ragged(cur), r);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = in ? v[j] : 0.f;
            }
            if (slot_valid(r)) *(f32x4 *)(lds_raw + dst0 + r * 2048) = v;
        }
    }
    f16x8 bq[BD][NF];
#pragma unroll
    for (int k = 0; k < BD; ++k)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bq[k][nf] = wload(k * (NF * 512) + nf * 512);
    __syncthreads();

    int ch = 0, buf = 0;
    while (true) {
        int ntile = tile, nch = ch + 1;
        if (nch == p.nchunks) { nch = 0; ntile = tile + nl; }
        const bool have_next = ntile < hi;
        const TileCoord nxt = (nch == 0 && have_next) ? decode(ntile) : cur;
        const int nch_eff = have_next ? nch : ch;
        const int nfaces = tile_faces(nxt);
        const bool nragged = tile_ragged(nxt);
        const char *bufc = lds_raw + buf * buf_bytes;
        char *bufn = lds_raw + (buf ^ 1) * buf_bytes;
        const unsigned wch = ch * (27 * NF * 512), wnx = nch_eff * (27 * NF * 512);  // halfs from wblk

        // per-chunk LDS addresses of the voxel fragments, pinned in VGPRs: the 27 tap offsets are instruction immediates
        typedef const __attribute__((address_space(3))) char lds_cchar;
        typedef const __attribute__((address_space(3))) f16x8 lds_cf16x8;
        lds_cchar *ab[MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            unsigned t = (unsigned)(size_t)(lds_cchar *)bufc + a_base[mf];
            asm volatile("" : "+v"(t));
            ab[mf] = (lds_cchar *)t;
        }
        f16x8 a[2][MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) a[0][mf] = *(lds_cf16x8 *)(ab[mf]);
        f32x4 st_v[SLOTS];
        f16x8 sc_n, sh_n;  // INAFF: scale / shift of the chunk being staged (loaded with the first fetch, used FLIGHT taps later)
        half_t sl_n = (half_t)1.f;
        if constexpr (INAFF) sl_n = load_aff(nxt.n, nch_eff, sc_n, sh_n);

        // (compile-time tap and slot indices: a "#pragma unroll" the optimiser declines turns a[tap & 1], bq[tap % BD] and
        //  st_v[tap] into select chains over whole register arrays)
        static_for<0, 27>([&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            // ---- all memory instructions of the step first (next step's voxel fragments, the weight fragment
            // BD steps ahead, one staging fetch), pinned ahead of the MFMAs so that every fragment has a full
            // step of MFMA time to arrive (the compiler otherwise sinks the LDS reads to just before their use)
            f16x8 bnew[NF];
            if constexpr (tap + 1 < 27) {
                constexpr int nt = tap + 1;
                constexpr int dz = nt / 9, rr = nt - dz * 9, dy = rr / 3, dx = rr - dy * 3;
                const int off = ((dz * IY + dy) * IX + dx) * 16;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf) a[(tap + 1) & 1][mf] = *(lds_cf16x8 *)(ab[mf] + off);
            }
            {
                constexpr int k = tap + BD;
                const unsigned wsrc = (k < 27) ? wch + k * (NF * 512) : wnx + (k - 27) * (NF * 512);
#pragma unroll
                for (int nf = 0; nf < NF; ++nf) bnew[nf] = wload(wsrc + nf * 512);
            }
            if constexpr (tap < SLOTS) st_v[tap] = stage_issue(nxt, nfaces, nragged, nch_eff, tap);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < NF; ++nf)
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[tap % BD][nf], a[tap & 1][mf], acc[mf][nf], 0, 0, 0);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) bq[tap % BD][nf] = bnew[nf];
            static_for<0, SLOTS>([&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                constexpr int wr = r + FLIGHT < 26 ? r + FLIGHT : 26;
                if constexpr (wr == tap) {
                    if constexpr (INAFF) {
                        // (round 3: the out-of-volume pieces are zeroed by four selects - as a second, masked LDS write the test
                        //  was a branch per slot, 49 more basic-block boundaries in the tap loop, each with its own waits)
                        st_v[r] = in_affine(st_v[r], sc_n, sh_n, sl_n);
                        const bool in = slot_inside(nxt, nfaces, nragged, r);
#pragma unroll
                        for (int j = 0; j < 4; ++j) st_v[r][j] = in ? st_v[r][j] : 0.f;
                    }
                    if (have_next && slot_valid(r)) *(f32x4 *)(bufn + dst0 + r * 2048) = st_v[r];
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();

        if (ch == p.nchunks - 1) {
            conv_epilogue_f16<MF, NF, HEAD>(acc, p, cur.n, cur.oz0, cur.oy0, cur.ox0, co_blk);
            acc_init_bias<MF, NF>(acc, p.bias, co_blk, half);
        }
        if (!have_next) break;
        tile = ntile; ch = nch; cur = nxt; buf ^= 1;
    }
}

// ------------------------------------------------------------------ stride 1, Cout % 64 == 0: LDS-DMA kernel (round 2)
// The pipelined kernel above spends a quarter of its time on the staging of the brick (ablation: -26 % without it):
// per piece a buffer load into a register, five VALU instructions and a ds_write, all of it issued between MFMAs that
// are only 32 cycles long in fp16.  This kernel is conv3_f32_wino2_kernel's skeleton (conv3d.hip) applied to the direct
// fp16 conv: ONE wave per SIMD that owns 128 voxels x 64 couts (8 MFMAs per tap, 128 accumulators in AGPRs), the
// 10 x 10 x 10 halo brick of an 8 x 8 x 8 tile double-buffered in LDS and filled by LDS-DMA (no staging registers, no
// ds_write; ~8 VALU per DMA for the address), the weights in a nine-tap register ring fetched with inline-asm loads and
// hand-counted waits (while an LDS-DMA is in flight hipcc retires EVERY vector-memory operation before the first use of
// an ordinary load).  All 27 tap offsets are instruction immediates (the brick geometry is a compile-time constant).
//   vmcnt bookkeeping (loads return in order; stores only make a count more conservative): tap t = [wait for W(t)]
//   [2 MFMAs] [LDS reads of tap t+1] [a DMA in every third tap] [6 MFMAs] [2 loads W(t+9) into the ring slot just consumed].
//   The loads issued after W(t) are those of taps t-8 .. t-1: 16 + the DMAs among them (DmaGeomH::pending); after the last
//   DMA of a chunk come the weight loads of the remaining taps, and that many may be outstanding at the barrier that
//   publishes the brick.
#ifdef MI355_H16_STAMPS
// tools/h16_probe.hip: cycle sums of wave 0 per workgroup. 0 taps 0-8, 1 taps 9-17, 2 taps 18-26, 3 drain + barrier, 4 epilogue,
// 5 whole kernel, 6 chunks, 7 tiles, 8 accumulator init + set-up
__device__ unsigned long long h16_stamps[1024 * 16];
#define H16_T(v) unsigned long long v = 0; if (stamp_on) v = __builtin_readcyclecounter()
#define H16_ACC(k, a, b) if (stamp_on) h16_acc[k] += (b) - (a)
#else
#define H16_T(v)
#define H16_ACC(k, a, b)
#endif
template <int EVERY_, int NF_ = 2, int D_ = 9>
struct DmaGeomH {
    static constexpr int IX = 10, IY = 10, IZ = 10, BV = IX * IY * IZ;
    static constexpr int PLANE_SLOTS = 1024, PLANE_BYTES = PLANE_SLOTS * 16, BUF_BYTES = 2 * PLANE_BYTES;
    // Brick layout in LDS: planar [8-channel half][voxel][16 B] (round 3) - with channel-blocked activations (common.h) the 64
    // lanes of a piece fetch 64 consecutive brick voxels of ONE block, i.e. rows of 10 x 16 contiguous bytes, and the fragment
    // reads (lane = voxel, stride 16 B) are conflict-free.  (Round 2's interleaved [voxel][half][16 B] brick for plain NDHWC
    // tensors went in round 5 together with the LDS transposition image of the epilogue: neither had a user left.)
    static constexpr bool INTERLEAVED = false;
    static constexpr int VOX_BYTES = 16;
    static constexpr int NF = NF_;  // 32-cout fragments per wave: 2 = conv3_f16_dma_kernel (64 couts per workgroup), 1 = conv3_f16_c32_kernel
    static constexpr int D = D_;   // weight ring depth in taps (divides 27: the ring phase is the same in every chunk)
    static constexpr int KD = 8;   // DMAs per wave and chunk: range wave + 4 (k & 3) of plane k >> 2
    static constexpr int EVERY = EVERY_;  // DMA k goes out in tap EVERY * k: four waves issuing 64-line DMAs in the same tap ask the
                                          // L1 for more lines than a tap has cycles (stamps: +1.3-2.2k cycles per chunk at EVERY = 1)
    static constexpr bool dma_tap(int t) { return t % EVERY == 0 && t / EVERY < KD; }
    // loads issued after the weight loads of tap t (which went out at the end of tap t - D): NF per tap of taps t-8 .. t-1
    // and the DMAs among those taps (taps < 0 are the previous chunk's)
    static constexpr int pending(int t) {
        int n = NF * (D - 1);
        for (int j = t - (D - 1); j <= t - 1; ++j) n += dma_tap((j + 27) % 27) ? 1 : 0;
        return n;
    }
    // weight loads issued after the chunk's last DMA
    static constexpr int after_last_dma = NF * (27 - EVERY * (KD - 1));
    // LDS: two brick buffers, the bias of the workgroup's couts, and for INAFF a 1-KiB junk area (the write target of
    // out-of-volume pieces) and the fp16 scale / shift tables [N][C0].  (Round 5: the 72-KiB transposition image of round 2's
    // epilogue is gone - the launch asked for 140 KiB of which it used 66, which kept every kernel that needs LDS off the CU
    // while this one ran, and capped the tables at 19 KiB: the 256- and 512-channel levels fell back to the register-staged kernel.)
    static constexpr int BIAS_OFF = 2 * BUF_BYTES, JUNK_OFF = BIAS_OFF + 64 * 4, TAB_OFF = JUNK_OFF + 1024;
    static constexpr size_t LDS_BYTES = (size_t)JUNK_OFF;
    static constexpr int WG_PER_CU = NF == 1 ? 2 : 1;   // (NF = 1: 64 accumulators per wave, two waves per SIMD)
    static constexpr int TAB_MAX_BYTES = 160 * 1024 / WG_PER_CU - TAB_OFF;
    // INAFF: piece k (DMA in tap EVERY k) has landed once the weights of tap EVERY k + 10 have been waited for (they were
    // issued after it); it is read back in that tap and normalised + written in the next one (NF = 1: in the next two)
    static constexpr int aff_read_tap(int k) { return EVERY * k + 10; }
};

// INAFF (round 2): the producer's Instance/GroupNorm + LeakyReLU applied to the brick IN LDS.  The DMA cannot touch the data
// in flight, so each lane normalises the eight pieces it fetched itself, in place, ten taps after their DMA (the weight
// waits guarantee they have landed): ds_read_b128, 12 packed fp16 ops, ds_write_b128, dealt out over the taps.  Out-of-volume
// pieces (zeros from the zero page; padding follows the norm) are written to a junk area instead.  Same arithmetic as the
// pipelined kernel's INAFF path: scale and shift as fp16, one fused multiply-add, max(y, slope y).
template <bool STATS, bool INAFF, int NF, bool HEAD, int RING = 9>
__device__ __forceinline__ void conv3_f16_dma_body(const ConvArgsH &p) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    typedef DmaGeomH<INAFF ? 2 : 3, NF, RING> G;
    static_assert(NF == 1 || NF == 2, "one or two 32-cout fragments per wave");
    static_assert(!HEAD || (NF == 1 && !STATS && !INAFF), "the fused head is built for the Cout = 32, BatchNorm-folded last conv only");
    // INAFF: the last piece must be normalised within its chunk (NF = 2: read in tap aff_read_tap, finished in the next one;
    // NF = 1 has two MFMA gaps per tap to spare, so a piece takes the next TWO taps)
    static_assert(!INAFF || G::aff_read_tap(G::KD - 1) + (NF == 1 ? 2 : 1) <= 26, "the last piece must be normalised within its chunk");
    constexpr int MF = 4;
    constexpr int IX = G::IX, IY = G::IY;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *bias_lds = (float *)(lds_raw + G::BIAS_OFF);
    if (tid < NF * 32) bias_lds[tid] = p.bias[(int)blockIdx.y * NF * 32 + tid];  // (published by the prologue's barrier)
    const int tab_n = p.N * p.C0;  // INAFF: fp16 scale [N][C0], then shift [N][C0]
    if constexpr (INAFF) {
        half_t *tab = (half_t *)(lds_raw + G::TAB_OFF);
        for (int i = tid; i < tab_n; i += 256) { tab[i] = (half_t)p.in_scale[i]; tab[tab_n + i] = (half_t)p.in_shift[i]; }
    }

    // this workgroup's tile sequence: XCD group x owns the contiguous range [lo, hi); its workgroups stride through it
    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int nl = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = p.total_tiles >> 3, r8 = p.total_tiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + li;
    if (tile >= hi) return;
    // (Measured in round 5 and removed: starting the second half of the Cout = 32 grid - the CUs' second workgroups - half a chunk late,
    //  MI355X_MICROARCH.md "Two waves per SIMD" item 9: s_sleep 40 / 80 / 127 against none, tools/h16_probe c: no difference beyond
    //  the +-4 % between repeats, profiles/r05_c32_stagger.txt.  The two workgroups of a CU drift apart by themselves.)

    struct TileCoord { int n, oz0, oy0, ox0; };
    auto decode = [&](int t) {
        TileCoord tc;
        tc.n = (int)fdiv((uint32_t)t, p.div_tiles_per_n);
        const int tt = t - tc.n * (int)p.div_tiles_per_n.d;
        int tile_x, tile_y, tile_z;
        tile_from_id(tt, p.order, tile_x, tile_y, tile_z);
        tc.oz0 = tile_z << 3; tc.oy0 = tile_y << 3; tc.ox0 = tile_x << 3;
        return tc;
    };
    // which faces of the brick [o0 - 1, o0 + 8] stick out of the volume (bit 0/1: z lo/hi, 2/3: y, 4/5: x); the volume is a
    // whole number of tiles (host check), so nothing else of a brick can lie outside
    auto tile_faces = [&](const TileCoord &tc) {
        return (tc.oz0 == 0) | ((tc.oz0 + 8 >= p.Di) << 1) | ((tc.oy0 == 0) << 2) | ((tc.oy0 + 8 >= p.Hi) << 3) |
               ((tc.ox0 == 0) << 4) | ((tc.ox0 + 8 >= p.Wi) << 5);
    };

    // tile-invariant lane part of the four DMA ranges of a plane: voxel offset from the brick origin (24 bits) | the
    // brick faces the voxel lies on << 24 (bit 6: a padding slot beyond the brick)
    unsigned dma_pk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int bv = (wave + 4 * k) * 64 + lane;
        const int bz = bv / (IX * IY), rr = bv - bz * (IX * IY), by = rr / IX, bx = rr - by * IX;
        const int face = (bz == 0) | ((bz == G::IZ - 1) << 1) | ((by == 0) << 2) | ((by == IY - 1) << 3) | ((bx == 0) << 4) | ((bx == IX - 1) << 5);
        dma_pk[k] = bv < G::BV ? (unsigned)(((bz * p.Hi + by) * p.Wi + bx) | (face << 24)) : (64u << 24);
    }
    auto dma = [&](const TileCoord &tc, int faces, int ch, auto k_c, char *buf) {
        constexpr int k = decltype(k_c)::value;
        const int cglob = ch * 16;
        const half_t *src; int Csrc, coff;
        if (cglob < p.C0) { src = p.in0; Csrc = p.C0; coff = cglob; }
        else { src = p.in1; Csrc = p.C1; coff = cglob - p.C0; }
        // wave-uniform part (SALU): block coff / 8 + the piece's half of sample n, at the brick origin voxel, which may lie one
        // voxel outside the tensor
        const long Vi = (long)p.Di * p.Hi * p.Wi;
        src += (((long)tc.n * (Csrc >> 3) + (coff >> 3) + (k >> 2)) * Vi + ((long)(tc.oz0 - 1) * p.Hi + (tc.oy0 - 1)) * p.Wi + (tc.ox0 - 1)) * 8;
        unsigned pk = dma_pk[k & 3];
        asm volatile("" : "+v"(pk));
        bool inside = (pk & ((unsigned)(faces | 64) << 24)) == 0;
#if defined(MI355_H16_ABL_DMA) && MI355_H16_ABL_DMA == 1   // (probe only: every piece reads the 16-B zero page - the same instructions, nothing fetched from beyond the L1; results wrong)
        inside = false;
#endif
        unsigned off = (pk & 0xffffffu) << 4;  // bytes: 16 per voxel of a block (< 2^32: host check)
#if defined(MI355_H16_ABL_DMA) && MI355_H16_ABL_DMA == 2   // (probe only: real data, but every piece comes from the first 256 KiB of the tensor - L2-resident; results wrong.
        const char *gin = (const char *)p.in0 + (off & 0x3fff0u);   //  Based at the TENSOR, not at the brick origin, which may lie in front of it)
#else
        const char *gin = (const char *)src + off;
#endif
        asm volatile("" : "+v"(gin));  // (computed for every lane: left to itself the compiler branches around it, and a basic-block
                                       //  boundary between the MFMAs of a tap makes it wait for every outstanding LDS read there)
        const char *g = inside ? gin : (const char *)p.zeros;  // the zero page holds both planes' pieces
        asm volatile("" : "+v"(g));
        char *dst = buf + (k >> 2) * G::PLANE_BYTES + (wave + 4 * (k & 3)) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };

    // INAFF: this lane's piece k of a brick buffer, read back / normalised and written in place (see above the kernel)
    typedef __attribute__((address_space(3))) char lds_char_t;
    const half_t slope_in = (half_t)(p.in_act == ACT_LRELU ? p.slope : 1.0f);
    // (per chunk: `pb` = this lane's slot 0 of the buffer, `ta` / `tb` = the chunk's rows of the scale / shift tables - the
    //  per-piece parts are instruction immediates; recomputed per piece these addresses were half of the pass's VALU work)
    unsigned aff_junk = (unsigned)(size_t)(lds_char_t *)(lds_raw + G::JUNK_OFF) + lane * 16;
    auto aff_bases = [&](char *buf, const TileCoord &tc, int ch, unsigned &pb, unsigned &ta, unsigned &tb) {
        pb = (unsigned)(size_t)(lds_char_t *)buf + wave * 1024 + lane * 16;
        ta = (unsigned)(size_t)(lds_char_t *)(lds_raw + G::TAB_OFF) + (tc.n * p.C0 + ch * 16) * 2;
        tb = ta + tab_n * 2;
        asm volatile("" : "+v"(pb), "+v"(ta), "+v"(tb));
    };
    auto aff_read = [&](unsigned pb, auto k_c) {
        constexpr int k = decltype(k_c)::value;
        return *(const __attribute__((address_space(3))) f32x4 *)(pb + ((k >> 2) * G::PLANE_BYTES + 4 * (k & 3) * 1024));
    };
    auto aff_apply = [&](int faces, auto k_c, unsigned pb, unsigned ta, unsigned tb, f32x4 raw) {
        constexpr int k = decltype(k_c)::value;
        const f16x8 sc = *(const __attribute__((address_space(3))) f16x8 *)(ta + (k >> 2) * 16);
        const f16x8 sh = *(const __attribute__((address_space(3))) f16x8 *)(tb + (k >> 2) * 16);
        f16x8 y = __builtin_elementwise_fma(__builtin_bit_cast(f16x8, raw), sc, sh);
        const f16x8 sl8 = {slope_in, slope_in, slope_in, slope_in, slope_in, slope_in, slope_in, slope_in};
        y = __builtin_elementwise_max(y, y * sl8);
        unsigned pk = dma_pk[k & 3];
        asm volatile("" : "+v"(pk));
        const bool inside = (pk & ((unsigned)(faces | 64) << 24)) == 0;
        // (padding follows the norm: out-of-volume pieces stay the zeros the DMA wrote; their result goes to the junk area)
        unsigned dst = inside ? pb + ((k >> 2) * G::PLANE_BYTES + 4 * (k & 3) * 1024) : aff_junk;
        asm volatile("" : "+v"(dst));
        *(__attribute__((address_space(3))) f32x4 *)dst = __builtin_bit_cast(f32x4, y);
    };

    // The same work in five stages dealt over the gaps between a tap's MFMAs (round 3).  As one block behind the tap's second
    // MFMA it cost 17 % of the kernel (conv3_f16_dma_kernel<true, true> 1010 against 1215 TFLOP/s without it): a gap hides about
    // six VALU instructions (MI355X_MICROARCH.md, vector-instruction issue cost) and the block had 17 plus a ds_write, and the
    // scale / shift rows were read from LDS in the gap that used them, i.e. with their full latency exposed.  Now: tap t reads
    // the piece AND its scale / shift rows (stage 0); NF = 2: tap t + 1 does fma | mul | max | select + write behind MFMAs 2, 3, 4,
    // 5; NF = 1 (four MFMAs per tap): fma | mul behind MFMAs 2, 3 of tap t + 1, max | select + write behind those of tap t + 2 -
    // stage 0 of the NEXT piece (tap t + 2, before MFMA 2) only overwrites raw / sc / sh, which stages 3, 4 no longer read.
    struct AffStage { f32x4 raw; f16x8 sc, sh, y, ys; };
    auto aff_s0 = [&](AffStage &st, unsigned pb, unsigned ta, unsigned tb, auto k_c) {
        constexpr int k = decltype(k_c)::value;
        st.raw = aff_read(pb, k_c);
        st.sc = *(const __attribute__((address_space(3))) f16x8 *)(ta + (k >> 2) * 16);
        st.sh = *(const __attribute__((address_space(3))) f16x8 *)(tb + (k >> 2) * 16);
    };
    auto aff_s1 = [&](AffStage &st) { st.y = __builtin_elementwise_fma(__builtin_bit_cast(f16x8, st.raw), st.sc, st.sh); };
    auto aff_s2 = [&](AffStage &st) {
        const f16x8 sl8 = {slope_in, slope_in, slope_in, slope_in, slope_in, slope_in, slope_in, slope_in};
        st.ys = st.y * sl8;
    };
    auto aff_s3 = [&](AffStage &st) { st.y = __builtin_elementwise_max(st.y, st.ys); };
    auto aff_s4 = [&](AffStage &st, int faces, unsigned pb, auto k_c) {
        constexpr int k = decltype(k_c)::value;
        unsigned pk = dma_pk[k & 3];
        asm volatile("" : "+v"(pk));
        const bool inside = (pk & ((unsigned)(faces | 64) << 24)) == 0;
        unsigned dst = inside ? pb + ((k >> 2) * G::PLANE_BYTES + 4 * (k & 3) * 1024) : aff_junk;
        asm volatile("" : "+v"(dst));
        *(__attribute__((address_space(3))) f32x4 *)dst = __builtin_bit_cast(f32x4, st.y);
    };

    // LDS byte offsets of this lane's voxel fragments in a brick buffer (tap (0,0,0)); fragment mf of wave w holds the
    // voxels v = (4 w + mf) * 32 + l31 of the tile: x = v & 7, y = (v >> 3) & 7, z = v >> 6
    int a_base[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = (wave * MF + mf) * 32 + l31;
        a_base[mf] = half * G::PLANE_BYTES + (((v >> 6) * IY + ((v >> 3) & 7)) * IX + (v & 7)) * 16;
    }
    const int co_blk = (int)blockIdx.y * NF * 32;
    const char *wblk = (const char *)(p.wp + (size_t)blockIdx.y * p.nchunks * (27 * NF * 512));
    const unsigned wlane = lane * 16;
#define H16_WLOAD(DST, SBASE, IMM) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(DST) : "v"(wl), "s"(SBASE), "n"(IMM) : "memory")
    // (the ring registers are operands of the wait: the compiler must not read or move them before it)
    auto wwait = [&](auto &w, auto n_c) {
        constexpr int N = decltype(n_c)::value;
        if constexpr (NF == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(w[0]), "+v"(w[1]) : "n"(N) : "memory");
        else asm volatile("s_waitcnt vmcnt(%1)" : "+v"(w[0]) : "n"(N) : "memory");
    };
    typedef std::integral_constant<int, 0> Zero;

    // HEAD: the fused 1x1x1 head's weights as MFMA A operands (hi + lo fp16 halves, conv_epilogue_f16), kernel invariants
    f16x8 wh[2][2];
    if constexpr (HEAD) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = sl * 8 + j;
                const int co = co_blk + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float w = l31 < p.head_ncls ? p.head_w[l31 * p.Cout + co] : 0.f;
                const half_t hi_h = (half_t)w;
                wh[sl][0][j] = hi_h;
                wh[sl][1][j] = (half_t)(w - (float)hi_h);
            }
    }

#ifdef MI355_H16_STAMPS
    const bool stamp_on = tid == 0 && blockIdx.y == 0;
    unsigned long long h16_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    H16_T(t_k0);
    TileCoord cur = decode(tile);
    f16x8 wq[G::D][NF];
    {
        const int f0 = tile_faces(cur);
        static_for<0, G::KD>([&](auto k_c) { dma(cur, f0, 0, k_c, lds_raw); });
        static_for<0, G::D>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            const char *wb = wblk + t * (NF * 1024);
            const unsigned wl = wlane;  // (asm operands alone do not capture)
            auto &w = wq[t];
            H16_WLOAD(w[0], wb, 0);
            if constexpr (NF == 2) H16_WLOAD(w[1], wb, 1024);
        });
        static_for<0, G::D>([&](auto t_c) { wwait(wq[decltype(t_c)::value], Zero{}); });
        if constexpr (INAFF) {
            __syncthreads();  // the tables
            unsigned pb, ta, tb;
            aff_bases(lds_raw, cur, 0, pb, ta, tb);
            static_for<0, G::KD>([&](auto k_c) { aff_apply(f0, k_c, pb, ta, tb, aff_read(pb, k_c)); });
        }
        __syncthreads();
    }

    int buf = 0;
    for (; tile < hi; tile += nl) {
        H16_T(t_t0);
        f32x16 acc[MF][NF];
        {   // bias from LDS (a global load here is an L2 round trip per tile with nothing to hide it behind)
            typedef const __attribute__((address_space(3))) f32x4 lds_cf32x4;
            unsigned bl = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)bias_lds + half * 16;
            asm volatile("" : "+v"(bl));  // (the reads stay inside the tile loop: hoisted they are 32 live registers)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b = *(lds_cf32x4 *)(bl + (nf * 32 + 8 * g) * 4);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[mf][nf][4 * g + k] = b[k];
                }
        }
        const int ntile = tile + nl;
        const TileCoord nxt_tile = ntile < hi ? decode(ntile) : cur;
#ifdef MI355_H16_STAMPS
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) asm volatile("" : "+a"(acc[mf][nf]));
#endif
        H16_T(t_t1);
        H16_ACC(8, t_t0, t_t1);
        for (int ch = 0; ch < p.nchunks; ++ch) {
            H16_T(t_c0);
            const bool last_ch = ch == p.nchunks - 1;
            const bool have_next = !last_ch || ntile < hi;
            const TileCoord nxt = last_ch ? nxt_tile : cur;
            // (without a next chunk the DMAs re-stage the current one into the idle buffer: the wait counts stay fixed)
            const int nch_eff = have_next ? (last_ch ? 0 : ch + 1) : ch;
            const int nfaces = tile_faces(nxt);
            const char *bufc = lds_raw + buf * G::BUF_BYTES;
            char *bufn = lds_raw + (buf ^ 1) * G::BUF_BYTES;
            const char *wch = wblk + (size_t)ch * (27 * NF * 1024), *wnx = wblk + (size_t)nch_eff * (27 * NF * 1024);

            typedef const __attribute__((address_space(3))) char lds_cchar;
            typedef const __attribute__((address_space(3))) f16x8 lds_cf16x8;
            lds_cchar *ab[MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                unsigned t = (unsigned)(size_t)(lds_cchar *)bufc + a_base[mf];
                asm volatile("" : "+v"(t));
                ab[mf] = (lds_cchar *)t;
            }
            f16x8 a[2][MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) a[0][mf] = *(lds_cf16x8 *)(ab[mf]);

#ifdef MI355_H16_STAMPS
            unsigned long long t_seg = t_c0;
#endif
            AffStage aff_st;
            unsigned aff_pb = 0, aff_ta = 0, aff_tb = 0;
            if constexpr (INAFF) aff_bases(bufn, nxt, nch_eff, aff_pb, aff_ta, aff_tb);
            static_for<0, 27>([&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int slot = tap % G::D;
                auto &wc = wq[slot];
                wwait(wc, std::integral_constant<int, G::pending(tap)>{});
                // the first MFMAs go out before the tap's memory instructions: the compiler waits for ALL outstanding LDS
                // reads before the first MFMA of a tap (lgkmcnt(0): it will not count past an LDS-DMA), so the reads of tap
                // t+1 are issued behind two MFMAs of tap t (NF = 2: six more to land behind) or behind one (NF = 1: three more)
#ifndef MI355_C32_MEM_AT
#define MI355_C32_MEM_AT 0
#endif
                constexpr int MEM_AT = NF == 2 ? 1 : MI355_C32_MEM_AT;
                static_for<0, MF * NF>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    constexpr int mf = i / NF, nf = i % NF;
                    acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[slot][nf], a[tap & 1][mf], acc[mf][nf], 0, 0, 0);
                    // INAFF: which piece is in which stage in this tap.  tr = the tap of a piece's DMA whose read-back (stage 0) is
                    // due now; ta1 / ta2 = those of the pieces one / two taps further on
                    constexpr int tr = tap - 10, ta1 = tap - 11, ta2 = tap - 12;
                    constexpr bool due1 = INAFF && ta1 >= 0 && ta1 % G::EVERY == 0 && ta1 / G::EVERY < G::KD;
                    constexpr bool due2 = INAFF && NF == 1 && ta2 >= 0 && ta2 % G::EVERY == 0 && ta2 / G::EVERY < G::KD;
                    if constexpr (i == MEM_AT) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (INAFF) {  // (before this tap's LDS reads: the piece landed long ago, nothing is waited for)
                            if constexpr (tr >= 0 && tr % G::EVERY == 0 && tr / G::EVERY < G::KD)
                                aff_s0(aff_st, aff_pb, aff_ta, aff_tb, std::integral_constant<int, tr / G::EVERY>{});
                        }
                        if constexpr (tap + 1 < 27) {
                            constexpr int nt = tap + 1;
                            constexpr int dz = nt / 9, rr = nt - dz * 9, dy = rr / 3, dx = rr - dy * 3;
                            constexpr int off = ((dz * IY + dy) * IX + dx) * G::VOX_BYTES;
#pragma unroll
                            for (int m = 0; m < MF; ++m) a[(tap + 1) & 1][m] = *(lds_cf16x8 *)(ab[m] + off);
                        }
                        if constexpr (G::dma_tap(tap)) dma(nxt, nfaces, nch_eff, std::integral_constant<int, tap / G::EVERY>{}, bufn);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if constexpr (NF == 2) {
                        if constexpr (due1 && i >= 2 && i <= 5) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (i == 2) aff_s1(aff_st);
                            if constexpr (i == 3) aff_s2(aff_st);
                            if constexpr (i == 4) aff_s3(aff_st);
                            if constexpr (i == 5) aff_s4(aff_st, nfaces, aff_pb, std::integral_constant<int, (ta1 >= 0 ? ta1 : 0) / G::EVERY>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
                        if constexpr ((due1 || due2) && (i == 2 || i == 3)) {
                            static_assert(!(due1 && due2), "EVERY = 2: a tap holds either stages 1-2 of one piece or stages 3-4 of the one before");
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (due1 && i == 2) aff_s1(aff_st);
                            if constexpr (due1 && i == 3) aff_s2(aff_st);
                            if constexpr (due2 && i == 2) aff_s3(aff_st);
                            if constexpr (due2 && i == 3) aff_s4(aff_st, nfaces, aff_pb, std::integral_constant<int, (ta2 >= 0 ? ta2 : 0) / G::EVERY>{});
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
                {
                    constexpr int k = tap + G::D;
#ifdef MI355_H16_ABL_W   // (probe only: every tap reads the same 2 KB of weights - L1-resident, no L2 traffic; results wrong)
                    const char *wb = wblk;
#else
                    const char *wb = (k < 27) ? wch + k * (NF * 1024) : wnx + (k - 27) * (NF * 1024);
#endif
                    const unsigned wl = wlane;
                    H16_WLOAD(wc[0], wb, 0);
                    if constexpr (NF == 2) H16_WLOAD(wc[1], wb, 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef MI355_H16_STAMPS
                if constexpr (tap == 8 || tap == 17 || tap == 26) {
                    if (stamp_on) { const unsigned long long t = __builtin_readcyclecounter(); h16_acc[tap / 9] += t - t_seg; t_seg = t; }
                }
#endif
            });
            // this wave's DMAs have landed once at most the weight loads issued after DMA 7 are outstanding; the barrier
            // publishes the brick (a ds_read is ordered behind an LDS-DMA only by the issuer's vmcnt + a barrier)
            H16_T(t_c2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::after_last_dma) : "memory");
            if constexpr (INAFF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the in-place writes of the normalised pieces
            __builtin_amdgcn_s_barrier();
            buf ^= 1;
            H16_T(t_c3);
            H16_ACC(3, t_c2, t_c3);
#ifdef MI355_H16_STAMPS
            if (stamp_on) h16_acc[6] += 1;
#endif
        }
        H16_T(t_e0);
        // The ring already holds the next tile's first nine taps, still in flight.  The compiler knows nothing of that: if it
        // moved one of those registers during the epilogue (a spill copy to an AGPR) it would copy what was there BEFORE the
        // load landed.  So the loads are retired here, with the ring as operands of the wait.
        static_for<0, G::D>([&](auto t_c) { wwait(wq[decltype(t_c)::value], Zero{}); });
        if constexpr (HEAD) {
            // ---- fused 1x1x1 segmentation head (the network's last conv, Cout = 32): logits[c][voxel] = sum_cout Wh[c][cout] act[cout][voxel]
            // on the matrix cores - the activation tile sits in the accumulators in the lane = voxel layout of an MFMA B operand
            // (conv_epilogue_f16, HEAD); the feature map is never written.  Voxel l31 of fragment mf = (z = 2 wave + (mf >> 1),
            // y = 4 (mf & 1) + (l31 >> 3), x = l31 & 7); class c = register c of the half-0 lanes.
            const float slope = p.act == ACT_LRELU ? p.slope : 1.0f;
            const unsigned Vo_h = (unsigned)(p.Do * p.Ho * p.Wo);   // (host: head_ncls * Vo * 4 < 2^32)
            int ln = lane;
            asm volatile("" : "+v"(ln));   // (a tile-loop invariant of the plain lane id would be hoisted and spilled)
            const unsigned lane_off_h = (unsigned)((((ln & 31) >> 3) * p.Wo + (ln & 7)) * 4);
            float *hbase = p.head_out + (size_t)cur.n * p.head_ncls * Vo_h + ((size_t)(cur.oz0 + 2 * wave) * p.Ho + cur.oy0) * p.Wo + cur.ox0;  // wave-uniform
            float hb[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) hb[c] = c < p.head_ncls ? p.head_b[c] : 0.f;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                f16x8 act[2];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = acc[mf][0][r];
                    act[r >> 3][r & 7] = (half_t)fmaxf(y, y * slope);  // (the unfused path rounds the activation to fp16 before the head reads it)
                }
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sl][1], act[sl], d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[sl][0], act[sl], d, 0, 0, 0);
                }
                if (ln < 32) {
                    char *row = (char *)(hbase + ((size_t)(mf >> 1) * p.Ho + (mf & 1) * 4) * p.Wo) + lane_off_h;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < p.head_ncls) *(float *)(row + (size_t)c * Vo_h * 4) = d[c] + hb[c];
                }
            }
        } else {
            // ---- epilogue: LeakyReLU + fp16 in registers, whole-line stores straight from registers (v_permlane32_swap, round 3)
            const float slope = p.act == ACT_LRELU ? p.slope : 1.0f;  // max(x, 1*x) = x
#ifndef MI355_H16_SC1
#define MI355_H16_SC1 1
#endif
            // voxel l31 of fragment mf = (z = 2 wave + (mf >> 1), y = 4 (mf & 1) + (l31 >> 3), x = l31 & 7); lanes 32-63 store the
            // next cout block (one block plane = Vo voxels x 16 B further)
            const size_t Vo_sw = (size_t)p.Do * p.Ho * p.Wo;
            // (32-bit, from a laundered lane id, HERE: as a loop invariant of the tile loop the 64-bit form was hoisted to the kernel
            //  entry and spilled to scratch - and a scratch access is a vector-memory operation the hand-counted vmcnt waits of
            //  the weight ring do not know of: the statistics instantiations read weights that had not landed)
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const unsigned lane_off_sw = (unsigned)(ln >> 5) * (unsigned)(Vo_sw * 16) + (unsigned)((((ln & 31) >> 3) * p.Wo + (ln & 7)) * 16);  // (< 2^32: host check)
            const half_t *obase_sw = p.out + (((size_t)cur.n * (p.Cout >> 3) + (co_blk >> 3)) * Vo_sw + ((size_t)(cur.oz0 + 2 * wave) * p.Ho + cur.oy0) * p.Wo + cur.ox0) * 8;
            // (cout fragment outermost: the statistics of one fragment are 32 live registers, not 64)
            // A statistics epilogue usually follows a convolution WITHOUT activation (conv -> norm -> LeakyReLU): that case
            // skips the 3 VALU instructions per value pair of max(x, slope x) - a wave-uniform choice of two instantiations.
            auto transpose_out = [&](auto act_c) {
                constexpr bool ACT = decltype(act_c)::value;
                static_for<0, NF>([&](auto nf_c) {
                    constexpr int nf = decltype(nf_c)::value;
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 s1[8], s2[8];
                    if constexpr (STATS) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) { s1[r] = f32x2{0.f, 0.f}; s2[r] = f32x2{0.f, 0.f}; }
                    }
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                        for (int gp = 0; gp < 4; gp += 2) {  // cout blocks g = gp, gp + 1 (8 couts each: 4 in this lane, 4 in lane ^ 32)
                            const f32x2 slope2 = {slope, slope};
                            f16x4 val2[2];
#pragma unroll
                            for (int gi = 0; gi < 2; ++gi) {
                                const int g = gp + gi;
#pragma unroll
                                for (int k = 0; k < 4; k += 2) {
                                    const f32x2 x = {acc[mf][nf][4 * g + k], acc[mf][nf][4 * g + k + 1]};
                                    f32x2 m = x;
                                    if constexpr (ACT) {
                                        f32x2 y;
                                        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(slope2));
                                        asm("v_max_f32 %0, %1, %2" : "=v"(m[0]) : "v"(x[0]), "v"(y[0]));
                                        asm("v_max_f32 %0, %1, %2" : "=v"(m[1]) : "v"(x[1]), "v"(y[1]));
                                    }
                                    val2[gi][k] = (half_t)m[0];
                                    val2[gi][k + 1] = (half_t)m[1];
                                    if constexpr (STATS) {  // v_pk_add_f32 + v_pk_fma_f32: one instruction each per value PAIR
                                        s1[2 * g + (k >> 1)] += m;
                                        s2[2 * g + (k >> 1)] = __builtin_elementwise_fma(m, m, s2[2 * g + (k >> 1)]);
                                    }
                                }
                            }
                            // Whole-line stores WITHOUT a trip through LDS (round 3; cdna_hip_programming.md T21).  This lane holds
                            // couts 0-3 (half 0) or 4-7 (half 1) of blocks gp and gp + 1 for voxel l31; v_permlane32_swap exchanges the
                            // upper half-wave of its first operand with the lower half-wave of its second, after which lanes 0-31 hold
                            // all 16 bytes of block gp and lanes 32-63 those of block gp + 1 for voxel l31.  The fragment's 32 voxels
                            // are four x-rows of 8 (128 B each in a block's plane): one store = 8 whole lines.
                            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                            u32x2 a = __builtin_bit_cast(u32x2, val2[0]), b = __builtin_bit_cast(u32x2, val2[1]);
                            // (inline asm: the pair-returning builtin is miscompiled by this hipcc, see common.h; s_nop 1 = the two wait
                            //  states between a VALU write of an operand and the swap, and again before the store reads the result)
                            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"
                                         : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
                            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                            const u32x4 v = {a[0], a[1], b[0], b[1]};
                            const char *row = (const char *)(obase_sw + (((size_t)(mf >> 1) * p.Ho + (mf & 1) * 4) * p.Wo + (size_t)(nf * 4 + gp) * Vo_sw) * 8);
                            const unsigned lo = lane_off_sw;
                            // (s_nop 4 in front: `row` may have been reloaded from an SGPR spill lane by v_readlane_b32 right before
                            //  the statement - five wait states hipcc does not pad inside inline asm, _isa_gate.py H1.  s_nop 1 behind
                            //  the store: a VALU write of the data registers of a 16-byte store needs a wait state after its issue - hipcc
                            //  pads its own stores, it does not look into inline asm, and the next pair's v_cvt_pk reuses these four
                            //  registers at once: without the pad the statistics instantiations stored garbage)
                            if constexpr (MI355_H16_SC1 != 0) asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(lo), "v"(v), "s"(row) : "memory");
                            else asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lo), "v"(v), "s"(row) : "memory");
                        }
                    if constexpr (STATS) {
                        // (round 3) transposing reduction, common.h: every lane ends with ONE total over the 32 voxel lanes of its
                        // half-wave (x 4 fragments = this wave's 128 voxels) and adds it itself - no LDS, no barrier.
                        // Quantised partials: exact additions in any order (common.h).
                        // (lane-dependent address parts from the laundered lane id `ln`: as tile-loop invariants of the plain lane id the
                        //  64-bit statistics address was hoisted to the kernel entry and spilled - 12 bytes of scratch, reloaded here by
                        //  vector-memory operations the hand-counted waits of the weight ring know nothing of; round 5)
                        const float tot = half32_reduce_scatter(s1, s2, ln);
                        const int r = stat_slot_r(ln), k = (ln >> 4) & 1;
                        const int c = nf * 32 + 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);
                        atomicAdd(p.stats + ((size_t)cur.n * p.Cout + co_blk + c) * 2 + k, quantise_partial((double)tot, k, (long)p.Do * p.Ho * p.Wo));
                    }
                });
            };
            if constexpr (STATS) {
                if (p.act == ACT_LRELU) transpose_out(std::true_type{}); else transpose_out(std::false_type{});
            } else {
                transpose_out(std::true_type{});
            }
        }
        cur = nxt_tile;
        H16_T(t_e1);
        H16_ACC(4, t_e0, t_e1);
#ifdef MI355_H16_STAMPS
        if (stamp_on) h16_acc[7] += 1;
#endif
    }
#ifdef MI355_H16_STAMPS
    if (stamp_on) {
        const unsigned long long t_k1 = __builtin_readcyclecounter();
        h16_acc[5] = t_k1 - t_k0;
        for (int k = 0; k < 10; ++k) h16_stamps[(blockIdx.x & 1023) * 16 + k] += h16_acc[k];
    }
#endif
#undef H16_WLOAD
}

// 64 couts per workgroup, one wave per SIMD (128 accumulators per lane)
template <bool STATS, bool INAFF = false>
__global__ __launch_bounds__(256, 1) void conv3_f16_dma_kernel(ConvArgsH p) {
    conv3_f16_dma_body<STATS, INAFF, 2, false>(p);
}

// Measured in round 5 and removed: the 64-cout body with a THREE-tap weight ring (211 - 225 registers per wave, no spills) and two
// workgroups per CU, so that the epilogue, chunk barrier and DMA issue of one run beside the MFMAs of the other: 1 318 against
// 1 320 TFLOP/s on <true, false>, 1 220 - 1 234 against 1 224 on <true, true>, config 3 fp16 264.0 / 263.8 against 264.5 / 264.7 ms
// in alternating runs on one box (profiles/r05_dma2_ab.txt).  These launches run at the board's power limit: a fuller matrix
// pipe is paid back in clock.
// Cout = 32 (round 5): the same body with ONE cout fragment per wave - 64 accumulators per lane, so TWO workgroups per CU (two
// waves per SIMD): a wave's DMA issue stalls, its chunk barrier and its epilogue run beside the other workgroup's MFMAs.  With
// half the MFMAs per brick a single workgroup per CU could not hide them (every DMA costs the wave ~130 cycles of issue; the
// register-staged conv3_f16_mfma_pipe_kernel<4, 1, ...> these launches used before spent a quarter of its time staging: 870-980
// TFLOP/s at 3.3-6.0 VALU instructions per MFMA).  HEAD: the network's last conv with the fused 1x1x1 head (model A).
template <bool STATS, bool INAFF = false, bool HEAD = false>
__global__ __launch_bounds__(256, 2) void conv3_f16_c32_kernel(ConvArgsH p) {
    conv3_f16_dma_body<STATS, INAFF, 1, HEAD>(p);
}

// ------------------------------------------------------------------ host side
// Packed layout (halfs): [cout_block][chunk][tap][nf][lane 0..63][j 0..7] with
//   cout = (cout_block*NF + nf)*32 + (lane&31),  cin = chunk*16 + (lane>>5)*8 + j.
int conv_weights_upload_f16(const float *w_host, const float *bias_host, int cin, int cin_pad, int cout, int stride,
                            ConvWeightsH *out) {
    MI355_REQUIRE(stride == 1 || stride == 2, "conv stride %d unsupported", stride);
    MI355_REQUIRE(cin_pad >= cin && cin_pad % 16 == 0, "fp16 conv needs cin_pad %% 16 == 0 (got %d for cin %d)", cin_pad, cin);
    MI355_REQUIRE(cout % 32 == 0, "fp16 conv needs cout %% 32 == 0 (got %d)", cout);
    ConvWeightsH cw;
    cw.cin = cin; cw.cin_pad = cin_pad; cw.cout = cout; cw.stride = stride;
    cw.nf = (cout % 64 == 0) ? 2 : 1;
    const int nchunks = cin_pad / 16, nblk = cout / (32 * cw.nf);
    std::vector<half_t> packed((size_t)nblk * nchunks * 27 * cw.nf * 512);
    size_t o = 0;
    for (int b = 0; b < nblk; ++b)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int tap = 0; tap < 27; ++tap)
                for (int f = 0; f < cw.nf; ++f)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j, ++o) {
                            const int co = (b * cw.nf + f) * 32 + (lane & 31);
                            const int ci = ch * 16 + (lane >> 5) * 8 + j;
                            packed[o] = (half_t)((ci < cin) ? w_host[((size_t)co * cin + ci) * 27 + tap] : 0.f);
                        }
    MI355_HIP(hipMalloc(&cw.wp_dev, packed.size() * sizeof(half_t)));
    MI355_HIP(hipMemcpy(cw.wp_dev, packed.data(), packed.size() * sizeof(half_t), hipMemcpyHostToDevice));
    MI355_HIP(hipMalloc(&cw.bias_dev, cout * sizeof(float)));
    if (bias_host) MI355_HIP(hipMemcpy(cw.bias_dev, bias_host, cout * sizeof(float), hipMemcpyHostToDevice));
    else MI355_HIP(hipMemset(cw.bias_dev, 0, cout * sizeof(float)));
    *out = cw;
    return MI355_OK;
}

void conv_weights_free_f16(ConvWeightsH *w) {
    if (w->wp_dev) (void)hipFree(w->wp_dev);
    if (w->bias_dev) (void)hipFree(w->bias_dev);
    *w = ConvWeightsH();
}

static void choose_tile_h(int Do, int Ho, int Wo, int stride, int voxels, int *lz, int *ly, int *lx) {
    auto p2cap = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    const int cz = p2cap(Do), cy = p2cap(Ho), cx = p2cap(Wo);
    const int L = ilog2_exact(voxels);
    int x = cx < 5 ? cx : 5;
    if (x > L) x = L;
    long best = -1;
    int bz = L - x, by = 0;
    for (int y = 0; x + y <= L; ++y) {
        const int z = L - x - y;
        const int oy = y > cy ? y - cy : 0, oz = z > cz ? z - cz : 0;
        const long brick = (long)(((1 << y) - 1) * stride + 3) * (((1 << z) - 1) * stride + 3);
        const long cost = ((long)(oy + oz) << 32) + brick;
        if (best < 0 || cost < best) { best = cost; bz = z; by = y; }
    }
    *lz = bz; *ly = by; *lx = x;
}

static void fill_geometry_h(ConvArgsH &a, int st, int voxels) {
    choose_tile_h(a.Do, a.Ho, a.Wo, st, voxels, &a.lz, &a.ly, &a.lx);
    const int TX = 1 << a.lx, TY = 1 << a.ly, TZ = 1 << a.lz;
    a.tiles_x = ceil_div(a.Wo, TX); a.tiles_y = ceil_div(a.Ho, TY); a.tiles_z = ceil_div(a.Do, TZ);
    a.IX = (TX - 1) * st + 3; a.IY = (TY - 1) * st + 3; a.IZ = (TZ - 1) * st + 3;
    a.div_tiles_per_n = make_fastdiv(a.tiles_x * a.tiles_y * a.tiles_z);
    a.div_tiles_x = make_fastdiv(a.tiles_x);
    a.div_tiles_y = make_fastdiv(a.tiles_y);
    a.div_IX = make_fastdiv(a.IX);
    a.div_IY = make_fastdiv(a.IY);
    a.plane_bytes = a.IX * a.IY * a.IZ * 16;
}

template <typename K>
static int launch_h(K kern, const ConvArgsH &a, dim3 grid, size_t lds_bytes, hipStream_t s, size_t *attr_bytes) {
    if (lds_bytes > *attr_bytes) {
        MI355_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        *attr_bytes = lds_bytes;
    }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, s, a);
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

static bool use_pipe_h() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MI355_CONV_IMPL"); v = (e && e[0] == '0') ? 0 : 1; }
    return v == 1;
}

// Whether conv3d_mfma_f16 would run this call on a kernel that can apply the producer's normalisation to its input: the
// pipelined stride-1 kernel (while staging through registers) or the LDS-DMA kernel (in LDS); the split-K and stride-2
// kernels cannot.  Mirrors the dispatch below.
bool conv3d_f16_fuses_input_norm(const ConvWeightsH &w, const ConvCallH &c) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("MI355_FUSE_NORM"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || w.stride != 1 || !use_pipe_h() || c.head_out || c.C0 % 16 != 0) return false;
    static int splitk = -1;
    if (splitk < 0) { const char *e = getenv("MI355_SPLITK"); splitk = (e && e[0] == '0') ? 0 : 1; }
    if (splitk && !c.stats && w.cin_pad / 16 >= 8) return false;  // might take the split-K path: keep it simple
    // the register-staged kernels apply the norm on volumes that are whole tiles only (512- or 256-voxel tiles, whichever
    // the launch picks; the LDS-DMA kernel's 8 x 8 x 8 tiles divide whatever these divide)
    for (int vox : {512, 256}) {
        int lz, ly, lx;
        choose_tile_h(c.Di, c.Hi, c.Wi, 1, vox, &lz, &ly, &lx);
        if (c.Di % (1 << lz) || c.Hi % (1 << ly) || c.Wi % (1 << lx)) return false;
    }
    return true;
}

int conv3d_mfma_f16(const ConvWeightsH &w, const ConvCallH &c, hipStream_t s, const char **kernel_name) {
    MI355_REQUIRE(c.C0 + c.C1 == w.cin_pad, "conv input channels %d+%d != %d", c.C0, c.C1, w.cin_pad);
    MI355_REQUIRE(c.C0 % 16 == 0 && c.C1 % 16 == 0, "fp16 concat split %d/%d not a multiple of 16", c.C0, c.C1);
    MI355_REQUIRE(c.C1 == 0 || c.in1 != nullptr, "second conv input missing");
    ConvArgsH a;
    a.in0 = c.in0; a.in1 = c.in1; a.C0 = c.C0; a.C1 = c.C1;
    a.wp = w.wp_dev; a.bias = w.bias_dev; a.out = c.out; a.stats = c.stats;
    a.head_w = c.head_w; a.head_b = c.head_b; a.head_out = c.head_out; a.head_ncls = c.head_ncls;
    a.N = c.N; a.Di = c.Di; a.Hi = c.Hi; a.Wi = c.Wi;
    const int st = w.stride;
    MI355_REQUIRE(!c.head_out || (st == 1 && use_pipe_h() && w.cout == 32 * w.nf && !c.stats && c.head_ncls >= 1 && c.head_ncls <= 4 && c.head_w && c.head_b),
                  "fused head needs Cout (%d) == one workgroup's couts, no statistics, 1..4 classes", w.cout);
    a.Do = (c.Di - 1) / st + 1; a.Ho = (c.Hi - 1) / st + 1; a.Wo = (c.Wi - 1) / st + 1;
    a.Cout = w.cout;
    a.nchunks = w.cin_pad / 16;
    a.act = c.act; a.slope = c.slope;
    a.total_tiles = 0;
    a.ksplit = 1; a.partial = nullptr; a.out_elems = 0;
    a.in_scale = c.in_scale; a.in_shift = c.in_shift; a.in_act = c.in_act;
    MI355_REQUIRE(!c.in_scale || (c.in_shift && conv3d_f16_fuses_input_norm(w, c)), "input normalisation can only be fused into the stride-1 kernels");
    const int gy = w.cout / (32 * w.nf);
    static size_t attr[8] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
    {
        // split-K for small launches (deep levels), as in conv3d.hip: fp32 partial sums, deterministic finishing pass
        static int splitk = -1;
        if (splitk < 0) { const char *e = getenv("MI355_SPLITK"); splitk = (e && e[0] == '0') ? 0 : 1; }
        if (splitk && !c.stats && !c.head_out && a.nchunks >= 8) {
            ConvArgsH b = a;
            const int MFs = st == 1 ? 2 : 1;
            fill_geometry_h(b, st, 128 * MFs);
            const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
            const long units = tiles * gy;
            // as many slices as still fit the chip in ONE round of workgroups (256 CUs x 2): rounding up (round 2) gave the 8^3 level
            // 80 x 7 = 560 workgroups - 48 of them ran behind the other 512 and doubled the launch's critical path
            int S = (int)(512 / units);
            if (S > a.nchunks / 4) S = a.nchunks / 4;
            if (S > 8) S = 8;
            size_t lds_bytes = (size_t)2 * b.plane_bytes;
            if (lds_bytes < 4096) lds_bytes = 4096;
            if (units < 256 && S >= 2 && lds_bytes <= 80 * 1024) {
                float *partial = nullptr;
                const long out_elems = (long)c.N * a.Do * a.Ho * a.Wo * w.cout;
                const size_t need = (size_t)S * out_elems * sizeof(float);
                MI355_TRY(device_scratch(SCR_SPLITK_F16, s, need, (void **)&partial));
                b.ksplit = S; b.partial = partial; b.out_elems = out_elems;
                dim3 grid((unsigned)tiles, gy, S);
                int rc;
                static size_t attr_sk[4] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
                if (st == 1 && w.nf == 1) { if (kernel_name) *kernel_name = "conv3_f16_mfma_kernel<1, 2, 1> split-K"; rc = launch_h(conv3_f16_mfma_kernel<1, 2, 1>, b, grid, lds_bytes, s, &attr_sk[0]); }
                else if (st == 1) { if (kernel_name) *kernel_name = "conv3_f16_mfma_kernel<1, 2, 2> split-K"; rc = launch_h(conv3_f16_mfma_kernel<1, 2, 2>, b, grid, lds_bytes, s, &attr_sk[1]); }
                else if (w.nf == 1) { if (kernel_name) *kernel_name = "conv3_f16_mfma_kernel<2, 1, 1> split-K"; rc = launch_h(conv3_f16_mfma_kernel<2, 1, 1>, b, grid, lds_bytes, s, &attr_sk[2]); }
                else { if (kernel_name) *kernel_name = "conv3_f16_mfma_kernel<2, 1, 2> split-K"; rc = launch_h(conv3_f16_mfma_kernel<2, 1, 2>, b, grid, lds_bytes, s, &attr_sk[3]); }
                if (rc != MI355_OK) return rc;
                const long total4 = out_elems / 4;
                hipLaunchKernelGGL(splitk_finish_f16_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, partial, S, total4, c.act, c.slope, c.out,
                                   w.cout, (long)a.Do * a.Ho * a.Wo);
                MI355_HIP(hipGetLastError());
                return MI355_OK;
            }
        }
    }
    if (st == 1 && use_pipe_h() && (w.nf == 2 ? !c.head_out : (w.cout == 32 && (!c.head_out || (!c.stats && !c.in_scale))))) {
        // LDS-DMA kernels: 8 x 8 x 8 tiles; conv3_f16_dma_kernel = 64 couts per workgroup, one workgroup per CU; conv3_f16_c32_kernel
        // (round 5) = the Cout = 32 layers, two workgroups per CU (MI355_F16_DMA=0: the register-staged kernels)
        static int dmak = -1;
        if (dmak < 0) { const char *e = getenv("MI355_F16_DMA"); dmak = (e && e[0] == '0') ? 0 : 1; }
        static int c32k = -1;
        if (c32k < 0) { const char *e = getenv("MI355_F16_C32"); c32k = (e && e[0] == '0') ? 0 : 1; }
        const bool c32 = w.nf == 1;
        ConvArgsH b = a;
        b.lx = b.ly = b.lz = 3;
        b.tiles_x = ceil_div(b.Wo, 8); b.tiles_y = ceil_div(b.Ho, 8); b.tiles_z = ceil_div(b.Do, 8);
        b.IX = b.IY = b.IZ = 10;
        b.div_tiles_per_n = make_fastdiv(b.tiles_x * b.tiles_y * b.tiles_z);
        const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
        typedef DmaGeomH<2, 2> GA;
        typedef DmaGeomH<2, 1> GA1;
        const int wg_per_cu = c32 ? GA1::WG_PER_CU : GA::WG_PER_CU;
        const size_t tab_bytes = c.in_scale ? (size_t)c.N * c.C0 * 4 : 0;  // fused input norm: fp16 scale + shift tables in LDS
        const size_t tab_max = c32 ? (size_t)GA1::TAB_MAX_BYTES : (size_t)GA::TAB_MAX_BYTES;
        if (dmak && (!c32 || c32k) && tiles * gy >= 256 * wg_per_cu && tiles < (1l << 30) && b.Do % 8 == 0 && b.Ho % 8 == 0 && b.Wo % 8 == 0 &&
            (long)10 * c.Hi * c.Wi < (1l << 24) && ((long)10 * c.Hi * c.Wi + 8l * c.Di * c.Hi * c.Wi) * 16 < (1l << 32) &&
            (!c.in_scale || (c.C1 == 0 && tab_bytes <= tab_max)) &&
            (!c.head_out || (long)c.head_ncls * c.Di * c.Hi * c.Wi * 4 < (1l << 32))) {
            void *zeros = nullptr;  // the zero page out-of-volume DMA pieces read
            MI355_TRY(device_scratch(SCR_ZEROS, s, 256, &zeros, true));
            b.zeros = zeros;
            b.total_tiles = (int)tiles;
            b.order = make_tile_order(b.tiles_x, b.tiles_y, b.tiles_z);
            int gx = 256 * wg_per_cu / gy;
            gx = gx < 8 ? 8 : (gx / 8) * 8;
            const int need = (int)((tiles + 7) / 8) * 8;
            if (gx > need) gx = need;
            const size_t lds_plain = GA::LDS_BYTES, lds_aff = (size_t)GA::TAB_OFF + tab_bytes;   // (the same layout for both fragment counts)
            static_assert(GA::LDS_BYTES == GA1::LDS_BYTES && GA::TAB_OFF == GA1::TAB_OFF, "one LDS layout");
            if (c32) {
                static size_t attr_c32[5] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
                if (c.head_out) {
                    if (kernel_name) *kernel_name = "conv3_f16_c32_kernel<false, false, true>";
                    return launch_h(conv3_f16_c32_kernel<false, false, true>, b, dim3(gx, gy), lds_plain, s, &attr_c32[4]);
                }
                if (c.in_scale) {
                    if (kernel_name) *kernel_name = c.stats ? "conv3_f16_c32_kernel<true, true, false>" : "conv3_f16_c32_kernel<false, true, false>";
                    if (c.stats) return launch_h(conv3_f16_c32_kernel<true, true>, b, dim3(gx, gy), lds_aff, s, &attr_c32[2]);
                    return launch_h(conv3_f16_c32_kernel<false, true>, b, dim3(gx, gy), lds_aff, s, &attr_c32[3]);
                }
                if (kernel_name) *kernel_name = c.stats ? "conv3_f16_c32_kernel<true, false, false>" : "conv3_f16_c32_kernel<false, false, false>";
                if (c.stats) return launch_h(conv3_f16_c32_kernel<true>, b, dim3(gx, gy), lds_plain, s, &attr_c32[0]);
                return launch_h(conv3_f16_c32_kernel<false>, b, dim3(gx, gy), lds_plain, s, &attr_c32[1]);
            }
            static size_t attr_dma[4] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};
            if (c.in_scale) {
                if (kernel_name) *kernel_name = c.stats ? "conv3_f16_dma_kernel<true, true>" : "conv3_f16_dma_kernel<false, true>";
                if (c.stats) return launch_h(conv3_f16_dma_kernel<true, true>, b, dim3(gx, gy), lds_aff, s, &attr_dma[2]);
                return launch_h(conv3_f16_dma_kernel<false, true>, b, dim3(gx, gy), lds_aff, s, &attr_dma[3]);
            }
            if (kernel_name) *kernel_name = c.stats ? "conv3_f16_dma_kernel<true, false>" : "conv3_f16_dma_kernel<false, false>";
            if (c.stats) return launch_h(conv3_f16_dma_kernel<true>, b, dim3(gx, gy), lds_plain, s, &attr_dma[0]);
            return launch_h(conv3_f16_dma_kernel<false>, b, dim3(gx, gy), lds_plain, s, &attr_dma[1]);
        }
    }
    if (st == 1 && use_pipe_h()) {
        int MF = 4;
        fill_geometry_h(a, 1, 128 * MF);
        long tiles = (long)a.tiles_x * a.tiles_y * a.tiles_z * c.N;
        if (tiles * gy < 512 || a.IX * a.IY * a.IZ > 11 * 128 || w.nf == 2) {
            MF = 2;
            fill_geometry_h(a, 1, 128 * MF);
            tiles = (long)a.tiles_x * a.tiles_y * a.tiles_z * c.N;
        }
        MI355_REQUIRE(tiles < (1l << 30), "conv grid too large");
        MI355_REQUIRE(a.IX * a.IY * a.IZ <= (MF == 4 ? 11 : 8) * 128, "conv brick exceeds the staging slots");
        MI355_REQUIRE((long)a.IZ * c.Hi * c.Wi < (1l << 24) && ((long)a.IZ * c.Hi * c.Wi + (long)c.Di * c.Hi * c.Wi) * 16 < (1l << 31),
                      "volume too large for the 32-bit staging offsets");
        a.total_tiles = (int)tiles;
        const size_t lds_bytes = (size_t)4 * a.plane_bytes + 4 * w.nf * 32 * 2 * sizeof(float);
        MI355_REQUIRE(lds_bytes <= 160 * 1024, "conv brick needs %zu B of LDS", lds_bytes);
        int gx = 512 / gy;
        gx = gx < 8 ? 8 : (gx / 8) * 8;
        const int need = (int)((tiles + 7) / 8) * 8;
        if (gx > need) gx = need;
        dim3 grid(gx, gy);
        const bool whole = a.Do % (1 << a.lz) == 0 && a.Ho % (1 << a.ly) == 0 && a.Wo % (1 << a.lx) == 0;
        // (names = the instantiations as rocprofv3 prints them: <MF, NF, HEAD, INAFF, STRIDE, WHOLE>)
        if (c.head_out) {
            MI355_REQUIRE(w.nf == 1, "fused head: fp16 path supports Cout = 32 only");
            static size_t attr_head[4] = {48 * 1024, 48 * 1024, 48 * 1024, 48 * 1024};  // one slot per kernel: the attribute is per function
            if (MF == 4 && whole) { if (kernel_name) *kernel_name = "conv3_f16_mfma_pipe_kernel<4, 1, true, false, 1, true>"; return launch_h(conv3_f16_mfma_pipe_kernel<4, 1, true, false, 1, true>, a, grid, lds_bytes, s, &attr_head[0]); }
            if (MF == 4) { if (kernel_name) *kernel_name = "conv3_f16_mfma_pipe_kernel<4, 1, true, false, 1, false>"; return launch_h(conv3_f16_mfma_pipe_kernel<4, 1, true>, a, grid, lds_bytes, s, &attr_head[1]); }
            if (kernel_name) *kernel_name = "conv3_f16_mfma_pipe_kernel<2, 1, true, false, 1, false>";
            return launch_h(conv3_f16_mfma_pipe_kernel<2, 1, true>, a, grid, lds_bytes, s, &attr_head[2]);
        }
        if (c.in_scale) {  // the producer's normalisation + activation applied while the brick is staged
            MI355_REQUIRE(whole, "fused input normalisation needs a volume of whole tiles (conv3d_f16_fuses_input_norm)");
            static size_t attr_aff[3] = {48 * 1024, 48 * 1024, 48 * 1024};
            if (kernel_name) *kernel_name = MF == 4 ? "conv3_f16_mfma_pipe_kernel<4, 1, false, true, 1, true>" : (w.nf == 1 ? "conv3_f16_mfma_pipe_kernel<2, 1, false, true, 1, true>" : "conv3_f16_mfma_pipe_kernel<2, 2, false, true, 1, true>");
            if (MF == 4) return launch_h(conv3_f16_mfma_pipe_kernel<4, 1, false, true, 1, true>, a, grid, lds_bytes, s, &attr_aff[0]);
            if (w.nf == 1) return launch_h(conv3_f16_mfma_pipe_kernel<2, 1, false, true, 1, true>, a, grid, lds_bytes, s, &attr_aff[1]);
            return launch_h(conv3_f16_mfma_pipe_kernel<2, 2, false, true, 1, true>, a, grid, lds_bytes, s, &attr_aff[2]);
        }
        if (whole) {
            static size_t attr_w[3] = {48 * 1024, 48 * 1024, 48 * 1024};
            if (kernel_name) *kernel_name = MF == 4 ? "conv3_f16_mfma_pipe_kernel<4, 1, false, false, 1, true>" : (w.nf == 1 ? "conv3_f16_mfma_pipe_kernel<2, 1, false, false, 1, true>" : "conv3_f16_mfma_pipe_kernel<2, 2, false, false, 1, true>");
            if (MF == 4) return launch_h(conv3_f16_mfma_pipe_kernel<4, 1, false, false, 1, true>, a, grid, lds_bytes, s, &attr_w[0]);
            if (w.nf == 1) return launch_h(conv3_f16_mfma_pipe_kernel<2, 1, false, false, 1, true>, a, grid, lds_bytes, s, &attr_w[1]);
            return launch_h(conv3_f16_mfma_pipe_kernel<2, 2, false, false, 1, true>, a, grid, lds_bytes, s, &attr_w[2]);
        }
        if (kernel_name) *kernel_name = MF == 4 ? "conv3_f16_mfma_pipe_kernel<4, 1, false, false, 1, false>" : (w.nf == 1 ? "conv3_f16_mfma_pipe_kernel<2, 1, false, false, 1, false>" : "conv3_f16_mfma_pipe_kernel<2, 2, false, false, 1, false>");
        if (MF == 4) return launch_h(conv3_f16_mfma_pipe_kernel<4, 1>, a, grid, lds_bytes, s, &attr[0]);
        if (w.nf == 1) return launch_h(conv3_f16_mfma_pipe_kernel<2, 1>, a, grid, lds_bytes, s, &attr[1]);
        return launch_h(conv3_f16_mfma_pipe_kernel<2, 2>, a, grid, lds_bytes, s, &attr[2]);
    }
    if (st == 2 && use_pipe_h()) {  // round 3: Cout % 128 == 0 on whole 4 x 4 x 8 output tiles -> conv3d_f16_s2.hip
        bool taken = false;
        MI355_TRY(conv3d_f16_s2dma(w, c, s, kernel_name, &taken));
        if (taken) return MI355_OK;
    }
    if (st == 2 && w.nf == 2 && !c.head_out && use_pipe_h()) {
        // round 2: the pipelined kernel with STRIDE = 2 (register-staged, two lanes per voxel) for the large stride-2 launches
        static int s2pipe = -1;  // MI355_S2_DMA=0: the one-tile-per-workgroup kernel instead (the switch of the f32 stride-2 kernel)
        if (s2pipe < 0) { const char *e = getenv("MI355_S2_DMA"); s2pipe = (e && e[0] == '0') ? 0 : 1; }
        ConvArgsH b = a;
        fill_geometry_h(b, 2, 128);
        const long tiles = (long)b.tiles_x * b.tiles_y * b.tiles_z * c.N;
        const size_t lds_bytes = (size_t)4 * b.plane_bytes + 4 * w.nf * 32 * 2 * sizeof(float);
        if (s2pipe && tiles * gy >= 768 && tiles < (1l << 30) && b.IX * b.IY * b.IZ <= 13 * 128 && lds_bytes <= 160 * 1024 &&
            (long)b.IZ * c.Hi * c.Wi < (1l << 24) && ((long)b.IZ * c.Hi * c.Wi + (long)c.Di * c.Hi * c.Wi) * 16 < (1l << 31) && !c.in_scale) {
            b.total_tiles = (int)tiles;
            int gx = 256 / gy;   // one workgroup per CU: the double-buffered brick takes about 100 KB of LDS
            gx = gx < 8 ? 8 : (gx / 8) * 8;
            const int need = (int)((tiles + 7) / 8) * 8;
            if (gx > need) gx = need;
            if (kernel_name) *kernel_name = "conv3_f16_mfma_pipe_kernel<1, 2, false, false, 2, false>";
            static size_t attr_s2 = 48 * 1024;
            return launch_h(conv3_f16_mfma_pipe_kernel<1, 2, false, false, 2>, b, dim3(gx, gy), lds_bytes, s, &attr_s2);
        }
    }
    const int MF = (st == 1) ? 2 : 1;
    fill_geometry_h(a, st, 128 * MF);
    const long tiles = (long)a.tiles_x * a.tiles_y * a.tiles_z * c.N;
    MI355_REQUIRE(tiles < (1l << 30), "conv grid too large");
    size_t lds_bytes = (size_t)2 * a.plane_bytes;
    if (lds_bytes < 4096) lds_bytes = 4096;
    MI355_REQUIRE(lds_bytes <= 160 * 1024, "conv brick needs %zu B of LDS", lds_bytes);
    dim3 grid((unsigned)tiles, gy);
    if (st == 1) {
        if (kernel_name) *kernel_name = w.nf == 1 ? "conv3_f16_mfma_kernel<1, 2, 1>" : "conv3_f16_mfma_kernel<1, 2, 2>";
        if (w.nf == 1) return launch_h(conv3_f16_mfma_kernel<1, 2, 1>, a, grid, lds_bytes, s, &attr[3]);
        return launch_h(conv3_f16_mfma_kernel<1, 2, 2>, a, grid, lds_bytes, s, &attr[4]);
    }
    if (kernel_name) *kernel_name = w.nf == 1 ? "conv3_f16_mfma_kernel<2, 1, 1>" : "conv3_f16_mfma_kernel<2, 1, 2>";
    if (w.nf == 1) return launch_h(conv3_f16_mfma_kernel<2, 1, 1>, a, grid, lds_bytes, s, &attr[5]);
    return launch_h(conv3_f16_mfma_kernel<2, 1, 2>, a, grid, lds_bytes, s, &attr[6]);
}

}  // namespace mi355

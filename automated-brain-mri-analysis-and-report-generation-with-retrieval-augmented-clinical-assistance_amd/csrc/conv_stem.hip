// First convolution of the network (Cin <= 4, 3x3x3, stride 1): the x-taps are folded into the GEMM K dimension.
//
// With only 4 input channels a tap-by-tap implicit GEMM wastes the matrix cores (K per tap = 4, padded to 8 in
// fp32 and to 16 in fp16).  Here K = (dx, channel): the NDHW4 layout makes the 3 x-neighbours of a voxel contiguous
// in memory and in the LDS brick, so for every (dz, dy)
//   fp32: three 8-B LDS reads feed 6 MFMAs 32x32x2   (lane half h holds channels 2h, 2h+1 of voxel x+dx)
//   fp16: two 8-B LDS reads feed ONE MFMA 32x32x16   (lane half 0: voxels x, x+1; half 1: voxel x+2 + 4 zero-weight k)
// i.e. 54 instead of 108 MFMAs (fp32) and 9 instead of 27 (fp16) per 32-voxel fragment, on an input tensor of
// 16 B / 8 B per voxel instead of 32 B.  All 27 / 9 weight fragments of a 32-cout block live in registers.
// Same D = W x X orientation and epilogues as conv3d.hip / conv3d_f16.hip (reference generic_UNet.py:56,69).
#include "kernels.h"

#include <vector>

namespace mi355 {

struct StemArgs {
    const void *in;     // [N,D,H,W,4] (fp32 or fp16)
    const void *wp;     // packed weights
    const float *bias;
    void *out;          // [N,D,H,W,Cout]
    double *stats;
    int N, D, H, W, Cout;
    int tiles_x, tiles_y, tiles_z;
    FastDiv div_tiles_per_n, div_tiles_x, div_tiles_y;
    int act;
    float slope;
};

// tile 4 x 4 x 32 outputs (wave = z, fragment = y), brick 6 x 6 x 34 (+2 voxels of x padding for the fp16 reads)
constexpr int S_IX = 36, S_IY = 6, S_IZ = 6;
constexpr int S_BRICK = S_IX * S_IY * S_IZ;

template <typename T>
__device__ __forceinline__ void stem_stage(const T *in, char *lds, int n, int D, int H, int W, int oz0, int oy0, int ox0) {
    // one piece per voxel: 4 channels = 16 B (fp32) / 8 B (fp16).  Round 3: all six loads of a thread are issued before the
    // first LDS write - as a load / select / write loop every iteration waited for its own load (six memory round trips
    // in series per tile, with four workgroups per CU to hide them: the fp16 stem ran at 2.1 TB/s of output).
    constexpr int PB = 4 * sizeof(T);
    constexpr int ITER = (S_BRICK + 255) / 256;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef typename std::conditional<sizeof(T) == 4, f32x4, f32x2>::type piece_t;
    piece_t v[ITER];
    bool ok[ITER];
    // Address arithmetic at full rate (round 3): the kernel runs one tile per workgroup, so every thread decodes six brick
    // positions per tile - with i / 36, r / 6 as v_mul_hi and a 64-bit offset per piece that was 72 v_mul_lo_u32 + 18
    // v_mad_u64_u32 per wave and tile, quarter-rate instructions worth ~1.7k cycles on a VALU-bound kernel.  Here: exact
    // reciprocal multiplies in 24 bits (i < 1536, r < 43), the voxel offset relative to the brick origin in 24-bit multiplies
    // (host check: (S_IZ * H + S_IY) * W < 2^24), and ONE wave-uniform 64-bit base.
    const long base = (((long)n * D + (oz0 - 1)) * H + (oy0 - 1)) * W + (ox0 - 1);  // voxels; may point one voxel outside (never dereferenced there)
    const T *inb = in + base * 4;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const unsigned i = (unsigned)(it * 256) + threadIdx.x;
        const unsigned r = __umul24(i, 1821u) >> 16, bx = i - __umul24(r, (unsigned)S_IX);  // i / 36 (exact for i < 1536)
        const unsigned bz = __umul24(r, 43u) >> 8, by = r - bz * 6u;                         // r / 6  (exact for r < 43)
        const int iz = oz0 - 1 + (int)bz, iy = oy0 - 1 + (int)by, ix = ox0 - 1 + (int)bx;
        ok[it] = (i < (unsigned)S_BRICK) && ((unsigned)iz < (unsigned)D) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
        const unsigned rel = __umul24(__umul24(bz, (unsigned)H) + by, (unsigned)W) + bx;     // voxels from the brick origin
        // (an out-of-volume piece reads the sample's first voxel instead: any valid address)
        const T *src = ok[it] ? inb + (size_t)rel * 4 : in + (size_t)n * D * H * W * 4;
        v[it] = *(const piece_t *)src;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int i = it * 256 + (int)threadIdx.x;
        piece_t z;
#pragma unroll
        for (int k = 0; k < (int)(sizeof(piece_t) / 4); ++k) z[k] = 0.f;
        if (i < S_BRICK) *(piece_t *)(lds + (size_t)i * PB) = ok[it] ? v[it] : z;
    }
}

__device__ __forceinline__ void stem_tile(const StemArgs &p, int &n, int &oz0, int &oy0, int &ox0) {
    const int bid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    n = (int)fdiv((uint32_t)bid, p.div_tiles_per_n);
    const int t = bid - n * (int)p.div_tiles_per_n.d;
    const int tzy = (int)fdiv((uint32_t)t, p.div_tiles_x);
    const int tile_x = t - tzy * p.tiles_x;
    const int tile_z = (int)fdiv((uint32_t)tzy, p.div_tiles_y);
    const int tile_y = tzy - tile_z * p.tiles_y;
    oz0 = tile_z << 2; oy0 = tile_y << 2; ox0 = tile_x << 5;
}

// Epilogue for the fixed 4x4x32 tile: lane = voxel (x = lane&31, y = fragment, z = wave), 16 couts per lane.
// Round 2: the 32 voxels x 32 couts of a fragment are one x-row of the output - 4 KiB (fp32) / 2 KiB (fp16) that are
// contiguous in memory when Cout = 32 - but a lane holds 4 x 4 couts of ONE voxel, and stored from the registers every
// instruction wrote 16 (8) bytes to each of 32 lines.  Each wave now transposes the fragment through a private LDS image
// (row = voxel, padded pitch) and stores 16 B per lane over whole 128-B lines: this kernel is bound by its output stream
// (2.1 GB per launch in fp32), which was reaching the HBM at a third of its rate.
template <typename T>
struct StemEpi {
    static constexpr int ROW = 32 * sizeof(T);       // bytes of one voxel's 32 couts
    static constexpr int PITCH = ROW + 16;           // padded: consecutive voxels start 4 banks apart
    static constexpr int WAVE_BYTES = 32 * PITCH;
    static constexpr int UNITS = ROW / 16;           // 16-B pieces per voxel: 8 (fp32) / 4 (fp16)
    static constexpr int STORES = 32 * UNITS / 64;   // store instructions per fragment: 4 / 2
};

// STATS: -1 = iff p.stats (run time), 0 / 1 = compiled out / in (fp16 kernel: with the 32 statistics registers of the run-time
// form the kernel needs 134 VGPRs and spilled at four workgroups per CU)
template <typename T, int STATS = -1>
__device__ __forceinline__ void stem_epilogue(f32x16 (&acc)[4], const StemArgs &p, int n, int oz0, int oy0, int ox0,
                                              int co_blk, char *img_all) {
    const bool do_stats = STATS < 0 ? p.stats != nullptr : STATS != 0;
    typedef StemEpi<T> E;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: the row addresses derived from it stay in SGPRs
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 s1[8], s2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { s1[r] = f32x2{0.f, 0.f}; s2[r] = f32x2{0.f, 0.f}; }
    const bool lrelu = p.act == ACT_LRELU;
    const float slope = lrelu ? p.slope : 1.0f;
    char *img = img_all + wave * E::WAVE_BYTES;
    f32x4 bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = *(const f32x4 *)(p.bias + co_blk + 8 * g + 4 * half);
    const int oz = oz0 + wave;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int oy = oy0 + mf;
        const bool ok = (oz < p.D) && (oy < p.H) && (ox0 + l31 < p.W);
        const float in = ok ? 1.f : 0.f;  // a voxel beyond a ragged edge adds nothing to the statistics
        f16x4 hv[4];  // fp16: the four 8-cout blocks' halves this lane holds
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 val;
#pragma unroll
            for (int k = 0; k < 4; ++k) val[k] = acc[mf][4 * g + k] + bias[g][k];
            if (lrelu) {  // (wave-uniform: conv -> norm -> LeakyReLU has no activation here, and this kernel is VALU-bound)
#pragma unroll
                for (int k = 0; k < 4; ++k) val[k] = fmaxf(val[k], val[k] * slope);
            }
            if (do_stats) {  // v_pk_add_f32 / v_pk_fma_f32: one instruction per value pair
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const f32x2 m = f32x2{val[k], val[k + 1]} * f32x2{in, in};
                    s1[2 * g + (k >> 1)] += m;
                    s2[2 * g + (k >> 1)] = __builtin_elementwise_fma(m, m, s2[2 * g + (k >> 1)]);
                }
            }
            if constexpr (sizeof(T) == 4) *(f32x4 *)(img + l31 * E::PITCH + (8 * g + 4 * half) * 4) = val;
            else hv[g] = f16x4{(_Float16)val[0], (_Float16)val[1], (_Float16)val[2], (_Float16)val[3]};
        }
        if constexpr (sizeof(T) == 2) {
            // fp16 tensors are channel-blocked ([N][C / 8][V][8], common.h): the fragment's 32 voxels are one x-row, 512 contiguous
            // bytes in each of the four 8-cout blocks.  Round 3: no transposition through LDS - pair_blocks_f16 (common.h) hands
            // lanes 0-31 the 16 bytes of block g and lanes 32-63 those of block g + 1 for voxel l31, and a store instruction
            // writes 32 voxels x two blocks = eight whole lines
            // (address = wave-uniform 64-bit base + a 32-bit lane part: computed per lane in 64 bits it was nine quarter-rate
            //  multiplies per store on a VALU-bound kernel; V * 16 < 2^32: host check)
            const int64_t V = (int64_t)p.D * p.H * p.W;
            const int64_t vrow = ((int64_t)oz * p.H + oy) * p.W + ox0;
            const unsigned lane_off = (unsigned)half * (unsigned)(V * 16) + (unsigned)l31 * 16u;
#pragma unroll
            for (int gp = 0; gp < 4; gp += 2) {
                const u32x4_t v16 = pair_blocks_f16(hv[gp], hv[gp + 1]);  // (every lane active; only the store is predicated)
                char *rowp = (char *)p.out + (((int64_t)n * (p.Cout >> 3) + (co_blk >> 3) + gp) * V + vrow) * 16;
                if (ok) *(u32x4_t *)(rowp + lane_off) = v16;
            }
        } else {
            // (same wave wrote the image: the LDS executes a wave's accesses in order, the compiler inserts the wait)
            const bool row_ok = (oz < p.D) && (oy < p.H);
            char *orow = (char *)p.out + ((((size_t)n * p.D + oz) * p.H + oy) * p.W + ox0) * p.Cout * sizeof(T) + co_blk * sizeof(T);
#pragma unroll
            for (int j = 0; j < E::STORES; ++j) {
                const int q = j * 64 + lane, vox = q / E::UNITS, unit = q % E::UNITS;
                const f32x4 v = *(const f32x4 *)(img + vox * E::PITCH + unit * 16);
                if (row_ok && ox0 + vox < p.W) *(f32x4 *)(orow + (size_t)vox * p.Cout * sizeof(T) + unit * 16) = v;
            }
        }
    }
    if (do_stats) {
        // (round 3) transposing reduction over the 32 voxel lanes (common.h): each lane ends with the total of one (cout,
        // statistic) over this wave's 128 voxels and adds it itself - no LDS, no barrier (round 2: 32 butterflies, a cross-wave
        // reduction through the dead brick behind two __syncthreads()).  Quantised partials: exact, hence order-independent.
        const float tot = half32_reduce_scatter(s1, s2, lane);
        const int r = stat_slot_r(lane), k = (lane >> 4) & 1;
        const int c = 8 * (r >> 2) + 4 * half + (r & 3);
        atomicAdd(p.stats + ((size_t)n * p.Cout + co_blk + c) * 2 + k, quantise_partial((double)tot, k, (long)p.D * p.H * p.W));
    }
}

// ---------------------------------------------------------------- fp32
// weights: [cout block][tap 27][lane 64][2]: lane (cout = l&31, h = l>>5) holds W[cout][c = 2h + j][tap], j = 0, 1
__global__ __launch_bounds__(256, 3) void conv3_stem_f32_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int n, oz0, oy0, ox0;
    stem_tile(p, n, oz0, oy0, ox0);
    const int co_blk = (int)blockIdx.y * 32;
    f32x2 wreg[27];
    const f32x2 *wsrc = (const f32x2 *)p.wp + (size_t)blockIdx.y * 27 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 27; ++t) wreg[t] = wsrc[t * 64];
    stem_stage<float>((const float *)p.in, lds, n, p.D, p.H, p.W, oz0, oy0, ox0);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
    // lane's voxel at tap (0,0,0): brick (z = wave, y = mf, x = l31); 16 B per voxel, half h reads channels 2h, 2h+1
    const int base = ((wave * S_IY) * S_IX + l31) * 16 + half * 8;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            f32x2 a[4][3];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    a[mf][dx] = *(const f32x2 *)(lds + base + (((dz * S_IY) + dy + mf) * S_IX + dx) * 16);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        acc[mf] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[(dz * 3 + dy) * 3 + dx][j], a[mf][dx][j], acc[mf], 0, 0, 0);
        }
    stem_epilogue<float>(acc, p, n, oz0, oy0, ox0, co_blk, lds + S_BRICK * 16);
}

// ---------------------------------------------------------------- fp16
// weights: [cout block][(dz,dy) 9][lane 64][8 halfs]: lane (cout, h): h = 0 -> W[c0..3][dx0], W[c0..3][dx1];
// h = 1 -> W[c0..3][dx2], 0, 0, 0, 0
template <bool STATS>
__global__ __launch_bounds__(256, 3) void conv3_stem_f16_kernel(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    int n, oz0, oy0, ox0;
    stem_tile(p, n, oz0, oy0, ox0);
    const int co_blk = (int)blockIdx.y * 32;
    f16x8 wreg[9];
    const f16x8 *wsrc = (const f16x8 *)p.wp + (size_t)blockIdx.y * 9 * 64 + lane;
#pragma unroll
    for (int t = 0; t < 9; ++t) wreg[t] = wsrc[t * 64];
    stem_stage<_Float16>((const _Float16 *)p.in, lds, n, p.D, p.H, p.W, oz0, oy0, ox0);
    __syncthreads();
    f32x16 acc[4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
    // 8 B per voxel; half h starts at voxel x + 2h and reads 16 B (two 8-B reads: the address is only 8-B aligned)
    const int base = ((wave * S_IY) * S_IX + l31 + 2 * half) * 8;
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const char *ptr = lds + base + ((dz * S_IY) + dy + mf) * S_IX * 8;
                const f16x4 lo = *(const f16x4 *)ptr, hi = *(const f16x4 *)(ptr + 8);
                const f16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                acc[mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[dz * 3 + dy], a, acc[mf], 0, 0, 0);
            }
        }
    stem_epilogue<_Float16, STATS ? 1 : 0>(acc, p, n, oz0, oy0, ox0, co_blk, lds + S_BRICK * 8);
}

// ---------------------------------------------------------------- host
int stem_weights_upload(const float *w_host, const float *bias_host, int cin, int cout, int dtype, StemWeights *out) {
    MI355_REQUIRE(cin >= 1 && cin <= 4 && cout % 32 == 0, "stem conv: need cin <= 4 and cout %% 32 == 0 (got %d -> %d)", cin, cout);
    StemWeights sw;
    sw.cin = cin; sw.cout = cout; sw.dtype = dtype;
    const int nblk = cout / 32;
    auto W = [&](int co, int c, int tap) { return c < cin ? w_host[((size_t)co * cin + c) * 27 + tap] : 0.f; };
    if (dtype == MI355_F16) {
        std::vector<_Float16> packed((size_t)nblk * 9 * 64 * 8);
        size_t o = 0;
        for (int b = 0; b < nblk; ++b)
            for (int zy = 0; zy < 9; ++zy)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j, ++o) {
                        const int co = b * 32 + (lane & 31), h = lane >> 5;
                        const int k = 8 * h + j, dx = k >> 2, c = k & 3;
                        packed[o] = (_Float16)(dx < 3 ? W(co, c, zy * 3 + dx) : 0.f);
                    }
        MI355_HIP(hipMalloc(&sw.wp_dev, packed.size() * sizeof(_Float16)));
        MI355_HIP(hipMemcpy(sw.wp_dev, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    } else {
        std::vector<float> packed((size_t)nblk * 27 * 64 * 2);
        size_t o = 0;
        for (int b = 0; b < nblk; ++b)
            for (int tap = 0; tap < 27; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 2; ++j, ++o) {
                        const int co = b * 32 + (lane & 31), h = lane >> 5;
                        packed[o] = W(co, 2 * h + j, tap);
                    }
        MI355_HIP(hipMalloc(&sw.wp_dev, packed.size() * sizeof(float)));
        MI355_HIP(hipMemcpy(sw.wp_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    MI355_HIP(hipMalloc(&sw.bias_dev, cout * sizeof(float)));
    if (bias_host) MI355_HIP(hipMemcpy(sw.bias_dev, bias_host, cout * sizeof(float), hipMemcpyHostToDevice));
    else MI355_HIP(hipMemset(sw.bias_dev, 0, cout * sizeof(float)));
    *out = sw;
    return MI355_OK;
}

void stem_weights_free(StemWeights *w) {
    if (w->wp_dev) (void)hipFree(w->wp_dev);
    if (w->bias_dev) (void)hipFree(w->bias_dev);
    *w = StemWeights();
}

int conv3d_stem(const StemWeights &w, const void *in, int N, int D, int H, int W, void *out, double *stats, int act,
                float slope, hipStream_t s) {
    StemArgs a;
    a.in = in; a.wp = w.wp_dev; a.bias = w.bias_dev; a.out = out; a.stats = stats;
    a.N = N; a.D = D; a.H = H; a.W = W; a.Cout = w.cout;
    a.tiles_x = ceil_div(W, 32); a.tiles_y = ceil_div(H, 4); a.tiles_z = ceil_div(D, 4);
    const long tiles = (long)a.tiles_x * a.tiles_y * a.tiles_z * N;
    MI355_REQUIRE(tiles < (1l << 30), "stem conv grid too large");
    MI355_REQUIRE(H < (1 << 20) && W < (1 << 20) && (long)(S_IZ * (long)H + S_IY) * W < (1l << 31) && (long)D * H * W * 16 < (1l << 32),
                  "stem conv: volume %d x %d x %d too large for the 32-bit staging / store offsets", D, H, W);
    a.div_tiles_per_n = make_fastdiv(a.tiles_x * a.tiles_y * a.tiles_z);
    a.div_tiles_x = make_fastdiv(a.tiles_x);
    a.div_tiles_y = make_fastdiv(a.tiles_y);
    a.act = act; a.slope = slope;
    dim3 grid((unsigned)tiles, w.cout / 32);
    if (w.dtype == MI355_F16) {
        if (stats) hipLaunchKernelGGL(conv3_stem_f16_kernel<true>, grid, dim3(256), (size_t)S_BRICK * 8, s, a);  // (brick only: the fp16 epilogue stores from registers)
        else hipLaunchKernelGGL(conv3_stem_f16_kernel<false>, grid, dim3(256), (size_t)S_BRICK * 8, s, a);  // (brick only: the fp16 epilogue stores from registers)
    } else {
        hipLaunchKernelGGL(conv3_stem_f32_kernel, grid, dim3(256), (size_t)S_BRICK * 16 + 4 * StemEpi<float>::WAVE_BYTES, s, a);
    }
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

}  // namespace mi355

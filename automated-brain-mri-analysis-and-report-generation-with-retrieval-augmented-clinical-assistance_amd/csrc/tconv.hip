// ConvTranspose3d(k=2, s=2, bias=False), NDHWC fp32, on the f32 matrix cores.
//
// Replaces the `tu[u]` modules of the reference decoder
// (model_architecture/generic_UNet.py:363-364, applied at :435).  With kernel == stride the
// op is 8 independent 1x1x1 GEMMs, one per output parity (a,b,c):
//     out[n, 2z+a, 2y+b, 2x+c, co] = sum_ci in[n,z,y,x,ci] * W[ci, co, a, b, c]
// M = input voxels (flattened N*D*H*W), K = Cin, N = Cout per parity.  A fragments are read
// straight from global (16 B per lane: 4 channels of one voxel; lanes 0-31 / 32-63 take the two
// halves of an 8-channel group, same K pairing as conv3d.hip); B fragments come from a
// host-permuted pack.  1.6 % of the network's flops, so no LDS staging.
#include "kernels.h"

#include <vector>

namespace mi355 {

template <int MF>
__global__ __launch_bounds__(256) void tconv2_f32_mfma_kernel(const float *__restrict__ in,
                                                             const float *__restrict__ wp, float *out,
                                                             int M, int Cin, int Cout, int D, int H, int W,
                                                             FastDiv divW, FastDiv divH, FastDiv divD) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = Cout >> 5;
    const int pos = (int)blockIdx.y / nblk, nb = (int)blockIdx.y - pos * nblk;
    const int G = Cin >> 3;
    const int m0 = ((int)blockIdx.x * 4 + wave) * (MF * 32);

    const float *arow[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        int v = m0 + mf * 32 + l31;
        if (v >= M) v = M - 1;  // clamp: rows past the end are computed and discarded
        arow[mf] = in + (size_t)v * Cin + half * 4;
    }
    const float *wrow = wp + ((size_t)(pos * nblk + nb) * G) * 256 + lane * 4;

    f32x16 acc[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[mf][r] = 0.f;

#pragma unroll 2
    for (int g = 0; g < G; ++g) {
        f32x4 a[MF];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
            a[mf] = *(const f32x4 *)(arow[mf] + g * 8);
        const f32x4 b = *(const f32x4 *)(wrow + (size_t)g * 256);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
                acc[mf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mf][j], b[j], acc[mf], 0, 0, 0);
    }

    const int pa = pos >> 2, pb = (pos >> 1) & 1, pc = pos & 1;
    const int co = nb * 32 + l31;
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int v = m0 + mf * 32 + row;
            if (v < M) {
                const uint32_t q1 = fdiv((uint32_t)v, divW);
                const int x = v - (int)q1 * W;
                const uint32_t q2 = fdiv(q1, divH);
                const int y = (int)q1 - (int)q2 * H;
                const uint32_t n = fdiv(q2, divD);
                const int z = (int)q2 - (int)n * D;
                out[((((size_t)n * Do + 2 * z + pa) * Ho + 2 * y + pb) * Wo + 2 * x + pc) * Cout + co] =
                    acc[mf][r];
            }
        }
    }
}

// Version 2: one workgroup = 128 input voxels (4 voxel fragments) x 32 couts x ALL 8 output parities; wave w owns
// parities {2w, 2w+1}.  Per 8-channel group a wave reads 4 voxel fragments (B operand, straight from global, L1-shared
// by the four waves) and 2 weight fragments (A operand) for 32 MFMAs; D = W^T x X so a lane holds one voxel and 16
// couts -> 16-B stores.  The input is read Cout/32 times in total instead of 8*Cout/32.
__global__ __launch_bounds__(256) void tconv2_f32_mfma_v2_kernel(const float *__restrict__ in,
                                                                const float *__restrict__ wp, float *out, int M,
                                                                int Cin, int Cout, int D, int H, int W, FastDiv divW,
                                                                FastDiv divH, FastDiv divD) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = Cout >> 5;
    const int nb = (int)blockIdx.y;
    const int G = Cin >> 3;
    const int m0 = (int)blockIdx.x * 128;
    // The 128 input voxels are staged through LDS in 64-channel chunks with whole-line loads (16 consecutive lanes read
    // the 256 contiguous bytes of one voxel); a fragment-shaped load straight from global would touch 64 lines per
    // instruction for 16 B each, and the texture path, not the matrix pipe, would set the pace.  LDS image: planar
    // [16-B channel quad][voxel], plane stride padded by one slot so that the staging writes are conflict-free too.
    constexpr int XPLANE = 128 * 4 + 4;  // floats
    __shared__ __attribute__((aligned(16))) float xs[16 * XPLANE];
    const float *wrow[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) wrow[pp] = wp + ((size_t)((wave * 2 + pp) * nblk + nb) * G) * 256 + lane * 4;
    f32x16 acc[2][4];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pp][mf][r] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        const int cq = (Cin - c0 < 64 ? Cin - c0 : 64) >> 2;  // channel quads in this chunk (Cin % 8 == 0)
        if (c0) __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const int i = k2 * 256 + tid, v = i >> 4, q = i & 15;
            int vg = m0 + v;
            if (vg >= M) vg = M - 1;
            if (q < cq) *(f32x4 *)(xs + q * XPLANE + v * 4) = *(const f32x4 *)(in + (size_t)vg * Cin + c0 + q * 4);
        }
        __syncthreads();
        const int gn = cq >> 1;
#pragma unroll 2
        for (int g = 0; g < gn; ++g) {
            f32x4 x[4], wv[2];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) x[mf] = *(const f32x4 *)(xs + (2 * g + half) * XPLANE + (mf * 32 + l31) * 4);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) wv[pp] = *(const f32x4 *)(wrow[pp] + (size_t)((c0 >> 3) + g) * 256);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        acc[pp][mf] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[pp][j], x[mf][j], acc[pp][mf], 0, 0, 0);
        }
    }
    // Epilogue.  A lane holds one voxel and 16 couts, so a direct store instruction touches 32 different 128-B lines with
    // 16-B pieces - 16 K line operations per workgroup through the texture path, as many cycles as the MFMAs take.  Each
    // 32-voxel x 32-cout tile therefore goes through a 4-KB LDS transpose (XOR-swizzled 16-B pieces, conflict-free both
    // ways) and is stored as whole lines: 8 lanes per voxel, 8 lines per instruction.
    __shared__ __attribute__((aligned(16))) float tr[4][32 * 32];
    float *mytr = tr[wave];
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int v = m0 + mf * 32 + l31;
        const int vc = v < M ? v : M - 1;
        const uint32_t q1 = fdiv((uint32_t)vc, divW);
        const int x = vc - (int)q1 * W;
        const uint32_t q2 = fdiv(q1, divH);
        const int y = (int)q1 - (int)q2 * H;
        const uint32_t n = fdiv(q2, divD);
        const int z = (int)q2 - (int)n * D;
        // output voxel index of parity (0,0,0); -1 = voxel beyond the tensor
        // element offset of the voxel's parity-(0,0,0) output row; -1 = voxel beyond the tensor (the row offset, not the
        // voxel index, travels through the shuffles: no 64-bit multiply per store)
        const long vox000 = v < M ? ((((long)n * Do + 2 * z) * Ho + 2 * y) * Wo + 2 * x) * Cout : -1;
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int pos = wave * 2 + pp;
            const int pa = pos >> 2, pb = (pos >> 1) & 1, pc = pos & 1;
            const long padd = (((long)pa * Ho + pb) * Wo + pc) * Cout + nb * 32;
            // write: row = voxel l31, 16-B piece (2*g4 + half) at physical piece (piece ^ (row & 7))
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 val = {acc[pp][mf][4 * g4], acc[pp][mf][4 * g4 + 1], acc[pp][mf][4 * g4 + 2], acc[pp][mf][4 * g4 + 3]};
                *(f32x4 *)(mytr + l31 * 32 + (((2 * g4 + half) ^ (l31 & 7)) << 2)) = val;
            }
            // read back: lane = (row r, piece c); the row's output offset comes from the lane that owns the voxel
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = it * 8 + (lane >> 3), c = lane & 7;
                const f32x4 val = *(const f32x4 *)(mytr + r * 32 + ((c ^ (r & 7)) << 2));
                const int lo = __shfl((int)(vox000 & 0xffffffff), r), hi = __shfl((int)(vox000 >> 32), r);
                const long vo = ((long)hi << 32) | (unsigned)lo;
                if (vo >= 0) *(f32x4 *)(out + (vo + padd) + c * 4) = val;
            }
        }
    }
}

// Version 3 (round 4; the fp16 kernel has had it since round 3): version 2 made persistent, with the wave's weights in registers.
// A workgroup of version 2 re-reads its weight block (Cin x 1 KiB) from the L2 for every 128 voxels, waits for each fragment in
// front of the MFMAs that use it, and has nothing to overlap its load - compute - store sequence with but the other
// workgroups of its CU.  Here the two parities' fragments of a wave stay in 8 G = Cin registers (G = Cin / 8 = 8 or 16) and the
// workgroup strides over the voxel tiles; per tile only the 128 x Cin input block comes through the L1.
template <int G>
__global__ __launch_bounds__(256, G <= 8 ? 2 : 1) void tconv2_f32_mfma_v3_kernel(const float *__restrict__ in,
                                                                                const float *__restrict__ wp, float *out, int M,
                                                                                int Cout, int D, int H, int W, FastDiv divW,
                                                                                FastDiv divH, FastDiv divD) {
    constexpr int Cin = G * 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = Cout >> 5;
    const int nb = (int)blockIdx.y;
    constexpr int XPLANE = 128 * 4 + 4;  // floats (version 2's image: planar [16-B channel quad][voxel], 64 channels at a time)
    __shared__ __attribute__((aligned(16))) float xs[16 * XPLANE];
    __shared__ __attribute__((aligned(16))) float tr[4][32 * 32];
    f32x4 wreg[2][G];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int g = 0; g < G; ++g) wreg[pp][g] = *(const f32x4 *)(wp + ((size_t)((wave * 2 + pp) * nblk + nb) * G + g) * 256 + lane * 4);
    float *mytr = tr[wave];
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
    const int ntiles = (M + 127) >> 7;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int m0 = tile * 128;
        f32x16 acc[2][4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][mf][r] = 0.f;
#pragma unroll
        for (int c0 = 0; c0 < Cin; c0 += 64) {
            if (c0 || tile != (int)blockIdx.x) __syncthreads();  // the previous image's fragment reads are done
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) {  // 128 voxels x 16 channel quads, whole-line loads (16 lanes = the 256 B of a voxel)
                const int i = k2 * 256 + tid, v = i >> 4, q = i & 15;
                int vg = m0 + v;
                if (vg >= M) vg = M - 1;
                *(f32x4 *)(xs + q * XPLANE + v * 4) = *(const f32x4 *)(in + (size_t)vg * Cin + c0 + q * 4);
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                f32x4 x[4];
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) x[mf] = *(const f32x4 *)(xs + (2 * g + half) * XPLANE + (mf * 32 + l31) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
                            acc[pp][mf] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[pp][(c0 >> 3) + g][j], x[mf][j], acc[pp][mf], 0, 0, 0);
            }
        }
        // epilogue: version 2's (4-KB LDS transpose per 32-voxel x 32-cout tile, whole-line stores)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int v = m0 + mf * 32 + l31;
            const int vc = v < M ? v : M - 1;
            const uint32_t q1 = fdiv((uint32_t)vc, divW);
            const int x = vc - (int)q1 * W;
            const uint32_t q2 = fdiv(q1, divH);
            const int y = (int)q1 - (int)q2 * H;
            const uint32_t n = fdiv(q2, divD);
            const int z = (int)q2 - (int)n * D;
            const long vox000 = v < M ? ((((long)n * Do + 2 * z) * Ho + 2 * y) * Wo + 2 * x) * Cout : -1;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {
                const int pos = wave * 2 + pp;
                const int pa = pos >> 2, pb = (pos >> 1) & 1, pc = pos & 1;
                const long padd = (((long)pa * Ho + pb) * Wo + pc) * Cout + nb * 32;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 val = {acc[pp][mf][4 * g4], acc[pp][mf][4 * g4 + 1], acc[pp][mf][4 * g4 + 2], acc[pp][mf][4 * g4 + 3]};
                    *(f32x4 *)(mytr + l31 * 32 + (((2 * g4 + half) ^ (l31 & 7)) << 2)) = val;
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int r = it * 8 + (lane >> 3), c = lane & 7;
                    const f32x4 val = *(const f32x4 *)(mytr + r * 32 + ((c ^ (r & 7)) << 2));
                    const int lo = __shfl((int)(vox000 & 0xffffffff), r), hi = __shfl((int)(vox000 >> 32), r);
                    const long vo = ((long)hi << 32) | (unsigned)lo;
                    if (vo >= 0) *(f32x4 *)(out + (vo + padd) + c * 4) = val;
                }
            }
        }
    }
}

// pack: [pos = a*4+b*2+c][cout block][g][lane][j]; cout = nb*32 + (lane&31), cin = g*8 + (lane>>5)*4 + j
int tconv_weights_upload(const float *w_host, int cin, int cout, TConvWeights *out) {
    MI355_REQUIRE(cin % 8 == 0 && cout % 32 == 0, "tconv %d->%d: need cin %% 8 == 0 and cout %% 32 == 0", cin, cout);
    const int G = cin / 8, nblk = cout / 32;
    std::vector<float> packed((size_t)8 * nblk * G * 256);
    size_t o = 0;
    for (int pos = 0; pos < 8; ++pos)
        for (int nb = 0; nb < nblk; ++nb)
            for (int g = 0; g < G; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j, ++o) {
                        const int co = nb * 32 + (lane & 31);
                        const int ci = g * 8 + (lane >> 5) * 4 + j;
                        packed[o] = w_host[((size_t)ci * cout + co) * 8 + pos];
                    }
    TConvWeights tw;
    tw.cin = cin; tw.cout = cout;
    MI355_HIP(hipMalloc(&tw.wp_dev, packed.size() * sizeof(float)));
    MI355_HIP(hipMemcpy(tw.wp_dev, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
    *out = tw;
    return MI355_OK;
}

void tconv_weights_free(TConvWeights *w) {
    if (w->wp_dev) (void)hipFree(w->wp_dev);
    *w = TConvWeights();
}

int tconv2_mfma_f32(const TConvWeights &w, const float *in, int N, int D, int H, int W, float *out,
                    hipStream_t s, const char **kernel_name) {
    const long M = (long)N * D * H * W;
    MI355_REQUIRE(M > 0 && M < (1l << 30), "tconv: %ld voxels out of range", M);
    static int v1 = -1;
    if (v1 < 0) { const char *e = getenv("MI355_TCONV_V1"); v1 = (e && e[0] == '1') ? 1 : 0; }
    static int v3 = -1;
    if (v3 < 0) { const char *e = getenv("MI355_TCONV_V3"); v3 = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }
    const long ntiles = (M + 127) / 128;
    // (Cin = 128 - 128 weight registers, one workgroup per CU - measured slower than version 2: 3.4 against 3.0 ms for 128 -> 64 @ 8 x 32^3;
    //  MI355_TCONV_V3=2 still takes it)
    if (!v1 && v3 && (w.cin == 64 || (w.cin == 128 && v3 == 2)) && ntiles >= 1024) {
        // persistent: the resident slots (two workgroups per CU at Cin = 64, one at 128: 128 weight registers) shared by the cout blocks
        const int nblk = w.cout / 32;
        long gx = (w.cin == 64 ? 512 : 256) / nblk;
        if (gx < 8) gx = 8;
        if (gx > ntiles) gx = ntiles;
        dim3 grid3((unsigned)gx, nblk);
        if (kernel_name) *kernel_name = w.cin == 64 ? "tconv2_f32_mfma_v3_kernel<8>" : "tconv2_f32_mfma_v3_kernel<16>";
        if (w.cin == 64) hipLaunchKernelGGL(tconv2_f32_mfma_v3_kernel<8>, grid3, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cout, D, H, W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        else hipLaunchKernelGGL(tconv2_f32_mfma_v3_kernel<16>, grid3, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cout, D, H, W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        MI355_HIP(hipGetLastError());
        return MI355_OK;
    }
    if (kernel_name) *kernel_name = v1 ? "tconv2_f32_mfma_kernel<2>" : "tconv2_f32_mfma_v2_kernel";
    if (!v1) {
        dim3 grid((unsigned)((M + 127) / 128), w.cout / 32);
        hipLaunchKernelGGL(tconv2_f32_mfma_v2_kernel, grid, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cin, w.cout, D, H, W,
                           make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        MI355_HIP(hipGetLastError());
        return MI355_OK;
    }
    constexpr int MF = 2;
    dim3 grid((unsigned)((M + 4 * MF * 32 - 1) / (4 * MF * 32)), 8 * (w.cout / 32));
    hipLaunchKernelGGL(tconv2_f32_mfma_kernel<MF>, grid, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cin,
                       w.cout, D, H, W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

// ------------------------------------------------------------------ fp16 storage variant
// D[cout][voxel] = W^T x X per output parity on v_mfma_f32_32x32x16_f16: weights are the A operand (lane: cout l&31,
// channels 8*(l>>5)..+7), input voxels the B operand read straight from global (16 B per lane); a lane ends up with
// 4 consecutive couts per register quad -> 8-byte stores.
template <int MF>
__global__ __launch_bounds__(256) void tconv2_f16_mfma_kernel(const _Float16 *__restrict__ in,
                                                             const _Float16 *__restrict__ wp, _Float16 *out, int M,
                                                             int Cin, int Cout, int D, int H, int W, FastDiv divW,
                                                             FastDiv divH, FastDiv divD) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = Cout >> 5;
    const int pos = (int)blockIdx.y / nblk, nb = (int)blockIdx.y - pos * nblk;
    const int G = Cin >> 4;
    const int m0 = ((int)blockIdx.x * 4 + wave) * (MF * 32);
    // fp16 tensors are channel-blocked ([N][C / 8][V][8], common.h): voxel v of sample n, block 2 g + half
    const long Vi = (long)D * H * W;
    const _Float16 *xrow[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        int v = m0 + mf * 32 + l31;
        if (v >= M) v = M - 1;
        const uint32_t n = fdiv(fdiv(fdiv((uint32_t)v, divW), divH), divD);
        xrow[mf] = in + ((long)n * (Cin >> 3) * Vi + (v - (long)n * Vi) + half * Vi) * 8;
    }
    const _Float16 *wrow = wp + ((size_t)(pos * nblk + nb) * G) * 512 + lane * 8;
    f32x16 acc[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
#pragma unroll 2
    for (int g = 0; g < G; ++g) {
        const f16x8 wf = *(const f16x8 *)(wrow + (size_t)g * 512);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const f16x8 xf = *(const f16x8 *)(xrow[mf] + (long)g * Vi * 16);
            acc[mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[mf], 0, 0, 0);
        }
    }
    const int pa = pos >> 2, pb = (pos >> 1) & 1, pc = pos & 1;
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int v = m0 + mf * 32 + l31;
        if (v < M) {
            const uint32_t q1 = fdiv((uint32_t)v, divW);
            const int x = v - (int)q1 * W;
            const uint32_t q2 = fdiv(q1, divH);
            const int y = (int)q1 - (int)q2 * H;
            const uint32_t n = fdiv(q2, divD);
            const int z = (int)q2 - (int)n * D;
            // couts nb * 32 + 8 g + 4 half .. + 3 = block nb * 4 + g, position 4 half of the blocked output
            const long Vo = (long)Do * Ho * Wo;
            _Float16 *o = out + (((long)n * (Cout >> 3) + nb * 4) * Vo + (((long)2 * z + pa) * Ho + 2 * y + pb) * Wo + 2 * x + pc) * 8 + 4 * half;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f16x4 hv = {(_Float16)acc[mf][4 * g], (_Float16)acc[mf][4 * g + 1], (_Float16)acc[mf][4 * g + 2],
                            (_Float16)acc[mf][4 * g + 3]};
                *(f16x4 *)(o + (long)g * Vo * 8) = hv;
            }
        }
    }
}

// Version 2 (same decomposition as tconv2_f32_mfma_v2_kernel): 128 voxels x 32 couts x 8 parities per workgroup.
// Two workgroups per CU (round 2): left to itself hipcc spread this kernel over 146 VGPRs + 128 AGPRs, one wave per SIMD, and a
// workgroup that loads, computes and stores in sequence had nothing to overlap with (137 -> 111 us per launch; three per CU spill).
__global__ __launch_bounds__(256, 2) void tconv2_f16_mfma_v2_kernel(const _Float16 *__restrict__ in,
                                                                const _Float16 *__restrict__ wp, _Float16 *out, int M,
                                                                int Cin, int Cout, int D, int H, int W, FastDiv divW,
                                                                FastDiv divH, FastDiv divD) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = Cout >> 5;
    const int nb = (int)blockIdx.y;
    const int G = Cin >> 4;
    const int m0 = (int)blockIdx.x * 128;
    // As in the fp32 kernel: input staged through LDS in 64-channel chunks with whole-line loads (8 lanes read the 128
    // contiguous bytes of a voxel), planar [16-B piece][voxel] image with a padded plane stride; output transposed through
    // LDS so that a store instruction writes whole 128-B lines.  A wave's two parities differ in x only, so the rows
    // (voxel, parity 2w) and (voxel, parity 2w+1) are adjacent in memory: one 128-B line = 2 x 32 couts.
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int XPLANE = 128 * 8 + 8;  // halfs
    __shared__ __attribute__((aligned(16))) _Float16 xs[8 * XPLANE];
    __shared__ __attribute__((aligned(16))) _Float16 tr[4][32 * 64];
    const _Float16 *wrow[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) wrow[pp] = wp + ((size_t)((wave * 2 + pp) * nblk + nb) * G) * 512 + lane * 8;
    f32x16 acc[2][4];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pp][mf][r] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += 64) {
        const int cp = (Cin - c0 < 64 ? Cin - c0 : 64) >> 3;  // 16-B pieces (8 channels) in this chunk (Cin % 16 == 0)
        if (c0) __syncthreads();
        // (fp16 tensors are channel-blocked, [N][C / 8][V][8], common.h: consecutive lanes = consecutive voxels of one block)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) {
            const int i = k2 * 256 + tid, v = i & 127, q = i >> 7;
            int vg = m0 + v;
            if (vg >= M) vg = M - 1;
            const uint32_t ns = fdiv(fdiv(fdiv((uint32_t)vg, divW), divH), divD);
            const long Vi = (long)D * H * W;
            if (q < cp) *(f32x4_t *)(xs + q * XPLANE + v * 8) = *(const f32x4_t *)(in + (((long)ns * (Cin >> 3) + (c0 >> 3) + q) * Vi + (vg - (long)ns * Vi)) * 8);
        }
        __syncthreads();
        const int gn = cp >> 1;
        for (int g = 0; g < gn; ++g) {
            f16x8 x[4], wv[2];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) x[mf] = *(const f16x8 *)(xs + (2 * g + half) * XPLANE + (mf * 32 + l31) * 8);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) wv[pp] = *(const f16x8 *)(wrow[pp] + (size_t)((c0 >> 4) + g) * 512);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[pp][mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wv[pp], x[mf], acc[pp][mf], 0, 0, 0);
        }
    }
    _Float16 *mytr = tr[wave];
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
    const int pa = wave >> 1, pb = wave & 1;  // parities 2w, 2w+1 = (pa, pb, 0) and (pa, pb, 1)
    // blocked output ([N][Cout / 8][Vo][8]): a wave's two parities are x-neighbours, so the 32 input voxels of a fragment (one
    // x-row when W >= 32) become 64 consecutive output voxels = 1 KiB contiguous in each of the four 8-cout blocks
    const long Vo = (long)Do * Ho * Wo;
    const long padd = ((long)pa * Ho + pb) * Wo;  // voxels
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int v = m0 + mf * 32 + l31;
        const int vc = v < M ? v : M - 1;
        const uint32_t q1 = fdiv((uint32_t)vc, divW);
        const int x = vc - (int)q1 * W;
        const uint32_t q2 = fdiv(q1, divH);
        const int y = (int)q1 - (int)q2 * H;
        const uint32_t n = fdiv(q2, divD);
        const int z = (int)q2 - (int)n * D;
        // 16-B piece index of output voxel (2z, 2y, 2x) in cout block nb * 4 of sample n (-1: row past the end)
        const long vox000 = v < M ? ((long)n * (Cout >> 3) + nb * 4) * Vo + (((long)2 * z) * Ho + 2 * y) * Wo + 2 * x : -1;
        // write: row = voxel l31 (128 B = 2 parities x 32 couts); 16-B piece pp*4 + g4 at physical piece (piece ^ (row & 7))
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f16x4 hv = {(_Float16)acc[pp][mf][4 * g4], (_Float16)acc[pp][mf][4 * g4 + 1],
                                  (_Float16)acc[pp][mf][4 * g4 + 2], (_Float16)acc[pp][mf][4 * g4 + 3]};
                *(f16x4 *)(mytr + l31 * 64 + (((pp * 4 + g4) ^ (l31 & 7)) << 3) + half * 4) = hv;
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            // store `it` = cout block it: lane -> (input voxel r = lane >> 1, x parity = lane & 1)
            const int r = lane >> 1, pc = lane & 1;
            const f32x4_t val = *(const f32x4_t *)(mytr + r * 64 + (((pc * 4 + it) ^ (r & 7)) << 3));
            const int lo = __shfl((int)(vox000 & 0xffffffff), r), hi = __shfl((int)(vox000 >> 32), r);
            const long vo = ((long)hi << 32) | (unsigned)lo;
            if (vo >= 0) *(f32x4_t *)(out + (vo + (long)it * Vo + padd + pc) * 8) = val;
        }
    }
}

// Version 3 (round 3): version 2 made persistent, with the weights in registers.  A workgroup of version 2 read its whole
// weight block from the L2 for every 128 voxels - Cin x 512 B, as many bytes as the 64 KiB of output it stores (Cin = 128) -
// so the L1 moved 2.5 x the HBM traffic of a kernel that should be bound by its output stream.  Here a workgroup keeps the
// two parities' weight fragments of its wave in registers (8 G VGPRs, G = Cin / 16 <= 8 at compile time) and strides over
// the voxel tiles; per tile only the 128 x Cin input block comes through the L1.
template <int G>
__global__ __launch_bounds__(256, 2) void tconv2_f16_mfma_v3_kernel(const _Float16 *__restrict__ in,
                                                                const _Float16 *__restrict__ wp, _Float16 *out, int M,
                                                                int Cout, int D, int H, int W, FastDiv divW,
                                                                FastDiv divH, FastDiv divD) {
    constexpr int Cin = G * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nblk = Cout >> 5;
    const int nb = (int)blockIdx.y;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    constexpr int XPLANE = 128 * 8 + 8;  // halfs
    constexpr int NP = Cin / 8;          // 8-channel blocks of the input (all of them staged at once: <= 16 x 2 KiB)
    __shared__ __attribute__((aligned(16))) _Float16 xs[NP * XPLANE];
    __shared__ __attribute__((aligned(16))) _Float16 tr[4][32 * 64];
    f16x8 wreg[2][G];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int g = 0; g < G; ++g) wreg[pp][g] = *(const f16x8 *)(wp + ((size_t)((wave * 2 + pp) * nblk + nb) * G + g) * 512 + lane * 8);
    const long Vi = (long)D * H * W;
    const int Ho = 2 * H, Wo = 2 * W, Do = 2 * D;
    const long Vo = (long)Do * Ho * Wo;
    const int pa = wave >> 1, pb = wave & 1;  // parities 2w, 2w+1 = (pa, pb, 0) and (pa, pb, 1)
    const long padd = ((long)pa * Ho + pb) * Wo;  // voxels
    _Float16 *mytr = tr[wave];
    const int ntiles = (M + 127) >> 7;
    for (int tile = (int)blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
        const int m0 = tile * 128;
        if (tile != (int)blockIdx.x) __syncthreads();  // the previous tile's fragment reads are done
#pragma unroll
        for (int k2 = 0; k2 < NP / 2; ++k2) {  // 128 voxels x NP blocks = NP / 2 pieces per thread
            const int i = k2 * 256 + tid, v = i & 127, q = i >> 7;
            int vg = m0 + v;
            if (vg >= M) vg = M - 1;
            const uint32_t ns = fdiv(fdiv(fdiv((uint32_t)vg, divW), divH), divD);
            *(f32x4_t *)(xs + q * XPLANE + v * 8) = *(const f32x4_t *)(in + (((long)ns * NP + q) * Vi + (vg - (long)ns * Vi)) * 8);
        }
        __syncthreads();
        f32x16 acc[2][4];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][mf][r] = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            f16x8 x[4];
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) x[mf] = *(const f16x8 *)(xs + (2 * g + half) * XPLANE + (mf * 32 + l31) * 8);
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    acc[pp][mf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[pp][g], x[mf], acc[pp][mf], 0, 0, 0);
        }
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int v = m0 + mf * 32 + l31;
            const int vc = v < M ? v : M - 1;
            const uint32_t q1 = fdiv((uint32_t)vc, divW);
            const int x = vc - (int)q1 * W;
            const uint32_t q2 = fdiv(q1, divH);
            const int y = (int)q1 - (int)q2 * H;
            const uint32_t n = fdiv(q2, divD);
            const int z = (int)q2 - (int)n * D;
            const long vox000 = v < M ? ((long)n * (Cout >> 3) + nb * 4) * Vo + (((long)2 * z) * Ho + 2 * y) * Wo + 2 * x : -1;
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f16x4 hv = {(_Float16)acc[pp][mf][4 * g4], (_Float16)acc[pp][mf][4 * g4 + 1],
                                      (_Float16)acc[pp][mf][4 * g4 + 2], (_Float16)acc[pp][mf][4 * g4 + 3]};
                    *(f16x4 *)(mytr + l31 * 64 + (((pp * 4 + g4) ^ (l31 & 7)) << 3) + half * 4) = hv;
                }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int r = lane >> 1, pc = lane & 1;
                const f32x4_t val = *(const f32x4_t *)(mytr + r * 64 + (((pc * 4 + it) ^ (r & 7)) << 3));
                const int lo = __shfl((int)(vox000 & 0xffffffff), r), hi = __shfl((int)(vox000 >> 32), r);
                const long vo = ((long)hi << 32) | (unsigned)lo;
                if (vo >= 0) *(f32x4_t *)(out + (vo + (long)it * Vo + padd + pc) * 8) = val;
            }
        }
    }
}

// pack: [pos][cout block][g][lane][j 0..7]; cout = nb*32 + (lane&31), cin = g*16 + (lane>>5)*8 + j
int tconv_weights_upload_f16(const float *w_host, int cin, int cout, TConvWeightsH *out) {
    MI355_REQUIRE(cin % 16 == 0 && cout % 32 == 0, "fp16 tconv %d->%d: need cin %% 16 == 0 and cout %% 32 == 0", cin, cout);
    const int G = cin / 16, nblk = cout / 32;
    std::vector<_Float16> packed((size_t)8 * nblk * G * 512);
    size_t o = 0;
    for (int pos = 0; pos < 8; ++pos)
        for (int nb = 0; nb < nblk; ++nb)
            for (int g = 0; g < G; ++g)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j, ++o) {
                        const int co = nb * 32 + (lane & 31);
                        const int ci = g * 16 + (lane >> 5) * 8 + j;
                        packed[o] = (_Float16)w_host[((size_t)ci * cout + co) * 8 + pos];
                    }
    TConvWeightsH tw;
    tw.cin = cin; tw.cout = cout;
    MI355_HIP(hipMalloc(&tw.wp_dev, packed.size() * sizeof(_Float16)));
    MI355_HIP(hipMemcpy(tw.wp_dev, packed.data(), packed.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    *out = tw;
    return MI355_OK;
}

void tconv_weights_free_f16(TConvWeightsH *w) {
    if (w->wp_dev) (void)hipFree(w->wp_dev);
    *w = TConvWeightsH();
}

int tconv2_mfma_f16(const TConvWeightsH &w, const _Float16 *in, int N, int D, int H, int W, _Float16 *out,
                    hipStream_t s, const char **kernel_name) {
    if (kernel_name) *kernel_name = "tconv2_f16_mfma_v2_kernel";
    const long M = (long)N * D * H * W;
    MI355_REQUIRE(M > 0 && M < (1l << 30), "tconv: %ld voxels out of range", M);
    static int v1 = -1;
    if (v1 < 0) { const char *e = getenv("MI355_TCONV_V1"); v1 = (e && e[0] == '1') ? 1 : 0; }
    static int v3 = -1;
    if (v3 < 0) { const char *e = getenv("MI355_TCONV_V3"); v3 = (e && e[0] == '0') ? 0 : 1; }
    const long ntiles = (M + 127) / 128;
    if (!v1 && v3 && (w.cin == 32 || w.cin == 64 || w.cin == 128) && ntiles >= 1024) {
        // persistent: two workgroups per CU and cout block share the 512 resident slots
        const int nblk = w.cout / 32;
        long gx = 512 / nblk;
        if (gx < 8) gx = 8;
        if (gx > ntiles) gx = ntiles;
        dim3 grid3((unsigned)gx, nblk);
        if (kernel_name) *kernel_name = w.cin == 32 ? "tconv2_f16_mfma_v3_kernel<2>" : (w.cin == 64 ? "tconv2_f16_mfma_v3_kernel<4>" : "tconv2_f16_mfma_v3_kernel<8>");
        if (w.cin == 32) hipLaunchKernelGGL(tconv2_f16_mfma_v3_kernel<2>, grid3, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cout, D, H, W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        else if (w.cin == 64) hipLaunchKernelGGL(tconv2_f16_mfma_v3_kernel<4>, grid3, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cout, D, H, W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        else hipLaunchKernelGGL(tconv2_f16_mfma_v3_kernel<8>, grid3, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cout, D, H, W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        MI355_HIP(hipGetLastError());
        return MI355_OK;
    }
    if (!v1) {
        dim3 grid2((unsigned)((M + 127) / 128), w.cout / 32);
        hipLaunchKernelGGL(tconv2_f16_mfma_v2_kernel, grid2, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cin, w.cout, D, H,
                           W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
        MI355_HIP(hipGetLastError());
        return MI355_OK;
    }
    constexpr int MF = 2;
    if (kernel_name) *kernel_name = "tconv2_f16_mfma_kernel<2>";
    dim3 grid((unsigned)((M + 4 * MF * 32 - 1) / (4 * MF * 32)), 8 * (w.cout / 32));
    hipLaunchKernelGGL(tconv2_f16_mfma_kernel<MF>, grid, dim3(256), 0, s, in, w.wp_dev, out, (int)M, w.cin, w.cout, D, H,
                       W, make_fastdiv(W), make_fastdiv(H), make_fastdiv(D));
    MI355_HIP(hipGetLastError());
    return MI355_OK;
}

}  // namespace mi355

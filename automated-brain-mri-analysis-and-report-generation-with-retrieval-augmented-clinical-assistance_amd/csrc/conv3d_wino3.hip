// 3x3x3 convolution, stride 1, NDHWC fp32, as Winograd F(2x2x2, 3x3x3) on the gfx950 matrix cores (round 4).
//
// Replaces torch.nn.Conv3d(k=3, p=1) of ConvDropoutNormNonlin (reference model_architecture/generic_UNet.py:56,69) on the
// large stride-1 layers of the fp32 path.  gfx950 has no TF32: an f32 MFMA is an exact fmaf chain at 157 TFLOP/s, so the
// lever of the fp32 path is the NUMBER of multiplies.  conv3_f32_wino2_kernel (conv3d.hip) transforms (z, y) and walks x
// directly: 4 * 4 * 3 = 48 multiplies per 2x2 outputs and x tap set, i.e. 4/9 of the direct count.  Here all three axes
// are transformed: a 2x2x2 output block costs 4 * 4 * 4 = 64 multiplies per (cin, cout) instead of 8 * 27 = 216 - 8/27 of
// the direct count, two thirds of the 2-D kernel's.
//
// Mapping.  Workgroup = 4 waves = one tile of 32 output blocks (2 x 4 x 4 blocks = 4 x 8 x 8 voxels) x 32 couts.  The 64
// transform-domain components are dealt over the waves by their z index: wave w owns xi_z = w and its 16 (xi_y, xi_x)
// components, 16 accumulator fragments of 32 couts x 32 blocks = 256 AGPRs - one wave per SIMD, one persistent workgroup
// per CU, the skeleton of the 2-D kernel:
//   * MFMA v_mfma_f32_32x32x2_f32, A = transformed weights U[xi][cout][cin pair], B = transformed input V[xi][cin pair][block];
//     a lane is a block (lane & 31) and a channel pair of the current quad (lane >> 5);
//   * the 16-channel halo brick (6 x 10 x 10 voxels) is double-buffered in LDS and filled by LDS-DMA
//     (global_load_lds_dwordx4; out-of-volume voxels read a zero page).  Slots are 16 B = one channel quad of a voxel,
//     ordered [quad][z][y][x parity][x / 2]: a lane's four x taps are then two adjacent slots per parity and the 32 blocks
//     of a half-wave read 32 different slots whose bank groups tile the 64 banks exactly twice (the minimum for 8-B reads);
//   * a step = one channel quad: 32 reads of 8 B (two z planes x 4 x 4 positions), V = B^T d B as 16 + 16 + 16 packed
//     operations (the z stage is one fma with a per-wave sign: xi_z = 0..3 are d0 - d2, d1 + d2, d2 - d1, d1 - d3), 32 MFMAs;
//     reads, transform, weight loads and brick DMAs of the NEXT step are dealt out between the MFMAs of the current one;
//   * epilogue: every wave applies A^T along y and x to its 16 components (16 -> 4 values per cout), writes the four partial
//     images to an LDS staging area [wave][oy, ox][block][cout], and after one barrier wave (oy, ox) adds the four xi_z
//     partials with A^T along z (out0 = p0 + p1 + p2, out1 = p1 - p2 - p3), applies bias / LeakyReLU and stores whole
//     128-B lines (8 lanes x 16 B = the 32 couts of one voxel).
// Numerics: U = G w G^T per axis in fp64, rounded once; tests/diagnostics/wino3d_numerics.py (CPU emulation of the whole
// network, sequential fp32 accumulation chains) gives a smaller logit error than the 2-D form (shorter chains: K = Cin
// per component instead of 3 Cin).
#include "kernels.h"

#include <cstdlib>
#include <vector>

// (the brick DMA below names m0 in its clobber list: it is a reserved register, which clang reports; nothing else in these kernels uses it)
#pragma clang diagnostic ignored "-Winline-asm"

namespace mi355 {

struct Wino3Args {
    const float *in0, *in1;
    const float *wp, *bias;
    float *out;
    double *stats;
    int C0, C1, N, D, H, W, Cout, nchunks, act;
    float slope;
    int total_tiles;
    FastDiv div_tiles_per_n;
    TileOrder order;
    const float *zeros;  // >= 64 B of zeros in global memory: the source of every out-of-volume piece
    const float *head_w, *head_b;
    float *head_out;
    int head_ncls;
    // INAFF: in0 is the RAW output of the previous block; act(x * in_scale[n][c] + in_shift[n][c]) is applied to the brick in LDS
    const float *in_scale, *in_shift;
    float in_slope;  // LeakyReLU slope of that activation, 1 = none
};

#ifdef MI355_W3_STAMPS
// Diagnostic build only (tools/wino3_probe.hip): cycle sums per phase, wave 0 of every workgroup.
// Slots: 0 chunk prologue, 1 step loop, 2 chunk drain + barrier, 3 epilogue phase 1 (in-wave output transform), 4 whole epilogue,
// 5 kernel, 6 chunks, 7 tiles, 9 accumulator reset, 10 whole chunk body (steps 0-3 + barrier), 11-14 steps 0-3 (MFMA loop only).
__device__ unsigned long long w3_stamps[1024 * 16];
#define W3_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
// (sums live in scalar registers and are written once at the end: a read-modify-write in global memory per stamp put its own
// vmcnt waits into the phase that followed it)
#define W3_DECL unsigned long long w3_loc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define W3_ACC(slot, a, b) w3_loc[slot] += (b) - (a)
#define W3_CNT(slot) w3_loc[slot] += 1
#define W3_FLUSH do { if (threadIdx.x == 0) for (int s_ = 0; s_ < 16; ++s_) w3_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 16 + s_] += w3_loc[s_]; } while (0)
#else
#define W3_T(var)
#define W3_DECL
#define W3_ACC(slot, a, b)
#define W3_CNT(slot)
#define W3_FLUSH
#endif
#ifndef MI355_W3_PIN
#define MI355_W3_PIN 1   // epilogue phase 1: the accumulator reads are pinned every MI355_W3_PIN register pairs (see there)
#endif
#ifndef MI355_W3_TWOBODY
#define MI355_W3_TWOBODY 1  // a second copy of the chunk body for a tile's chunk 0 whose first MFMAs take C = 0 (see the tile loop); 0: one body
#endif
#ifndef MI355_W3_ABL
#define MI355_W3_ABL 0  // ablation bits (probe only, results wrong): 1 no epilogue, 4 no brick DMA, 8 no weight loads, 16 no input transform, 32 no stores
#endif

constexpr int W3_IZ = 6, W3_IY = 10, W3_IX = 10, W3_BV = W3_IZ * W3_IY * W3_IX;  // 600 brick voxels
constexpr int W3_PS = 640;                      // slots per quad plane: 10 DMA ranges of 64
constexpr int W3_BUF = 4 * W3_PS * 4;           // floats per brick buffer (4 quad planes)
constexpr int W3_PITCH = 36;                    // floats per staged block row (128 B + 16 B)
constexpr int W3_IMG = 32 * W3_PITCH;           // one staged image: 32 blocks x 32 couts
constexpr int W3_STAGE = 4 * 4 * W3_IMG;        // [wave = xi_z][oy, ox][block][cout]
constexpr int W3_JUNK = 64 * 4;                 // INAFF: where the in-place writes of out-of-volume pieces go (padding follows the norm)
constexpr size_t W3_LDS_BYTES = (size_t)(2 * W3_BUF + W3_STAGE + 4 * 32 * 2 + W3_JUNK) * sizeof(float);
static_assert(W3_LDS_BYTES <= 160 * 1024, "LDS budget");

// EPI: 0 = bias + LeakyReLU + store, 1 = fused 1x1x1 segmentation head (the network's last conv, Cout = 32: only the logits are
// written), 2 = as 0 + Instance/GroupNorm statistics (sum x, sum x^2 per sample and cout)
// INAFF (round 4): the producer's Instance/GroupNorm (+ LeakyReLU) is applied by THIS conv (generic_UNet.py:62-72 is one expression,
// lrelu(instnorm(conv(x)))): every lane normalises, in place in LDS, the ten 16-byte pieces it fetched itself - no other lane has
// seen them before the chunk barrier publishes the brick -, dealt over the MFMA gaps of the step in front of that barrier:
// ds_read_b128, two v_pk_fma_f32 (scale / shift of the piece's channel quad: wave-uniform, scalar loads), two v_pk_mul_f32 and four
// v_max_f32 (LeakyReLU as max(y, slope y)), ds_write_b128.  Pieces from outside the volume came from the zero page and must stay
// zero (padding follows the norm): their write goes to a junk slot.  fp32 arithmetic: bit-compatible with norm_apply_kernel<f32>.
template <int EPI, bool INAFF = false>
__global__ __launch_bounds__(256, 1) void conv3_f32_wino3_kernel(Wino3Args p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr int STEPS = 4;  // channel quads per 16-channel chunk
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l31 = lane & 31;
    const int bx = l31 & 3, by = (l31 >> 2) & 3, bz = l31 >> 4;
    float *stage = lds + 2 * W3_BUF;
    float *red = stage + W3_STAGE;
    float *junk = red + 4 * 32 * 2;
    unsigned inmask = 0;  // INAFF: bit k = "group k of the chunk being staged lies inside the volume (for this lane)"

    // tile sequence of this workgroup: XCD group x owns one contiguous range of tile ids
    const int xcd = (int)blockIdx.x & 7, li = (int)blockIdx.x >> 3;
    const int nl = ((int)gridDim.x - xcd + 7) >> 3;
    const int q8 = p.total_tiles >> 3, r8 = p.total_tiles & 7;
    const int lo = xcd * q8 + (xcd < r8 ? xcd : r8);
    const int hi = lo + q8 + (xcd < r8 ? 1 : 0);
    int tile = lo + li;
    if (tile >= hi) return;

    struct TileCoord { int n, oz0, oy0, ox0; };
    auto decode = [&](int t) {
        TileCoord tc;
        tc.n = (int)fdiv((uint32_t)t, p.div_tiles_per_n);
        const int tt = t - tc.n * (int)p.div_tiles_per_n.d;
        int tx, ty, tz;
        tile_from_id(tt, p.order, tx, ty, tz);
        tc.oz0 = tz << 2; tc.oy0 = ty << 3; tc.ox0 = tx << 3;
        return tc;
    };

    // brick DMA: range r = 64 consecutive slots of a quad plane, 10 ranges x 4 quads = 40 wave-instructions per chunk, ten per wave
    // (the hand-counted waits below need the same number of vector-memory operations in flight in all four waves): group 0 = range
    // w, group 1 = range w + 4 (four quads each, issued back to back: they share their 128-B lines), group 2 = two quads of range
    // 8 (waves 0, 1) or 9 (waves 2, 3).
    // Per lane and group, fixed for the whole kernel: the brick voxel (rz, ry, rx) as three one-hot fields (bit rz, bit 6 + ry,
    // bit 16 + rx; bit 31 = padding slot beyond the 600 voxels) and its byte offset from the brick's corner in either input tensor.
    // Per DMA stream position (tile, chunk), scalar: the corner's address and the same three fields with the planes / rows /
    // columns that fall outside the volume set.  A piece is inside iff (need & notok) == 0: two vector instructions instead of
    // unpacking and three range checks, and the scalar part is computed once per chunk, spread over the MFMA gaps of step 1
    // (round 4: the stamps put 2 300 of a chunk's 12 250 cycles on the three gaps that held the DMA address arithmetic).
    unsigned dma_need[3], dma_off0[3], dma_off1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = (k < 2 ? wave + 4 * k : 8 + (wave >> 1)) * 64 + lane;
        const int pad = v >= W3_BV ? 1 : 0;
        const int vv = pad ? 0 : v;
        const int rz = vv / 100, rem = vv - rz * 100;
        const int ry = rem / 10, r2 = rem - ry * 10;
        const int par = r2 / 5, xh = r2 - par * 5;
        const int rx = 2 * xh + par;
        const int q0 = k < 2 ? 0 : 2 * (wave & 1);
        dma_need[k] = pad ? 0x80000000u : ((1u << rz) | (1u << (6 + ry)) | (1u << (16 + rx)));
        const unsigned vox = (unsigned)((rz * p.H + ry) * p.W + rx);
        dma_off0[k] = (vox * (unsigned)p.C0 + 4u * q0) * 4u;
        dma_off1[k] = (vox * (unsigned)p.C1 + 4u * q0) * 4u;
    }
    // DMA stream position and what is derived from it (all scalar)
    TileCoord d_tc;
    int d_tile, d_ch;
    const char *d_src;   // address of the brick's corner voxel (may lie in front of the tensor), channel of the chunk's first quad
    unsigned d_notok;    // one-hot fields of the planes / rows / columns outside the volume, | bit 31
    bool d_sel1;         // the chunk's channels come from in1 (virtual concat)
    // A group is issued in two halves (quads 0-1, then 2-3) in two consecutive MFMA gaps: four 1-KiB DMAs in one gap hold the
    // wave's issue longer than one MFMA runs.
    const float *dma_g = nullptr;
    unsigned dma_m0 = 0;
    auto dma_group = [&](auto kc, auto part_c, float *buf) {
        constexpr int k = decltype(kc)::value, part = decltype(part_c)::value;
        // Inline asm, not __builtin_amdgcn_global_load_lds: hipcc tracks the builtin's LDS writes and, wherever it cannot prove that a
        // ds_read does not alias one in flight - at every loop back edge - it retires ALL vector memory operations (vmcnt(0)) in front
        // of the read; this pipeline keeps a DMA group in flight across the chunk loop's back edge by design (the barrier that
        // publishes the data is what orders it).  M0 = LDS byte address of lane 0's slot, one wait state between its write and its
        // use.  The instruction's immediate offset is added to the global AND the LDS address: quad Q's M0 is moved back by it.
#define W3_DMA(Q)                                                                                                         \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:%2"                                \
                 :: "s"(dma_m0 + (unsigned)((Q) * (W3_PS * 16 - 16))), "v"(dma_g), "n"((Q) * 16) : "memory", "m0")
        if constexpr (part == 0) {
            const int rng = k < 2 ? wave + 4 * k : 8 + (wave >> 1);  // scalar
            const int q0 = k < 2 ? 0 : 2 * (wave & 1);
            const bool in_vol = (dma_need[k] & d_notok) == 0;
            if constexpr (INAFF) inmask = in_vol ? (inmask | (1u << k)) : (inmask & ~(1u << k));
            const unsigned off = d_sel1 ? dma_off1[k] : dma_off0[k];
            dma_g = in_vol ? (const float *)(d_src + off) : p.zeros;
            asm volatile("" : "+v"(dma_g));  // one DMA per quad for every lane (a branchy select would break the vmcnt count)
            float *dst = buf + rng * 64 * 4 + q0 * (W3_PS * 4);
            dma_m0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)dst;
            W3_DMA(0);
            W3_DMA(1);
        } else if constexpr (k < 2) {
            W3_DMA(2);
            W3_DMA(3);
        }
#undef W3_DMA
    };

    // this lane's block origin in the brick (slot of (2 bz, 2 by, 2 bx)), floats, + its channel pair; the two z planes of the
    // wave's component: xi_z = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  T = dA + sgn * dB
    const int a_base = (((2 * bz) * W3_IY + 2 * by) * W3_IX + bx) * 4 + half * 2;
    const int zA = wave == 0 ? 0 : (wave == 2 ? 2 : 1), zB = wave == 2 ? 1 : (wave == 3 ? 3 : 2);
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const int offA = a_base + zA * (W3_IY * W3_IX * 4), offB = a_base + zB * (W3_IY * W3_IX * 4);

    // packed U: [cout block][chunk][quad][xi_z][fragment pair 0..7][lane][f & 1][j]: one 16-B load per lane = fragments 2k, 2k + 1.
    // Inline asm with hand-counted waits (conv3d.hip: with an LDS-DMA in flight hipcc retires every vector-memory operation
    // before the first use of an ordinary load).
    const float *wblk = p.wp + (size_t)blockIdx.y * p.nchunks * (STEPS * 4 * 2048) + wave * 2048;
    const unsigned wl0 = lane * 16, wl1 = lane * 16 + 4096;
// W3_ULOAD0 = the first load of a group: SBASE may have been reloaded from an SGPR spill lane (v_readlane_b32, a VALU write of an
// SGPR) right in front of the statement, and a vector-memory instruction that reads such an SGPR as its scalar base needs five
// wait states which hipcc pads for its own instructions only, not inside inline asm (cdna_hip_programming.md 5.7 item 2).
// Without them the load reads the PREVIOUS contents of the pair: a wild address - round 4's "memory access fault on every
// shape" of the two-body build (its listing carries exactly this sequence; _isa_gate.py H1 now fails the build on it).
#define W3_ULOAD0(DST, VOFF, SBASE, IMM) \
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(DST) : "v"(VOFF), "s"(SBASE), "n"(IMM) : "memory")
#define W3_ULOAD(DST, VOFF, SBASE, IMM) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(DST) : "v"(VOFF), "s"(SBASE), "n"(IMM) : "memory")
#define W3_UWAIT(U, N)                                                                                                 \
    asm volatile("s_waitcnt vmcnt(" #N ")"                                                                             \
                 : "+v"(U[0]), "+v"(U[1]), "+v"(U[2]), "+v"(U[3]), "+v"(U[4]), "+v"(U[5]), "+v"(U[6]), "+v"(U[7])      \
                 :                                                                                                     \
                 : "memory")

    auto pk_add = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
    auto pk_sub = [](f32x2 x, f32x2 y) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y)); return r; };
    // position pos = iy * 4 + ix of the 4 x 4 patch: slot offset iy * IX + (ix & 1) * 5 + (ix >> 1)
    auto slot_of = [](int pos) { return ((pos >> 2) * W3_IX + ((pos & 1) * 5) + ((pos & 3) >> 1)) * 4; };
    // reads of one half of the patch (8 positions x 2 planes) of quad q from buffer b
    auto read8 = [&](const float *b, int q, int hb, f32x2 (&dA)[8], f32x2 (&dB)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dA[i] = *(const f32x2 *)(b + offA + q * (W3_PS * 4) + slot_of(hb * 8 + i));
            dB[i] = *(const f32x2 *)(b + offB + q * (W3_PS * 4) + slot_of(hb * 8 + i));
        }
    };
    const f32x2 sg2 = {sgn, sgn};
    auto z_op = [&](const f32x2 (&dA)[8], const f32x2 (&dB)[8], f32x2 (&T)[16], int hb) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            f32x2 r;
            asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(dB[i]), "v"(sg2), "v"(dA[i]));
            T[hb * 8 + i] = r;
        }
    };
    // y stage for x column ix: T[iy][ix] -> Y[xi_y][ix]
    auto y_op = [&](const f32x2 (&T)[16], f32x2 (&Y)[16], int ix) {
        Y[0 * 4 + ix] = pk_sub(T[0 * 4 + ix], T[2 * 4 + ix]);
        Y[1 * 4 + ix] = pk_add(T[1 * 4 + ix], T[2 * 4 + ix]);
        Y[2 * 4 + ix] = pk_sub(T[2 * 4 + ix], T[1 * 4 + ix]);
        Y[3 * 4 + ix] = pk_sub(T[1 * 4 + ix], T[3 * 4 + ix]);
    };
    // x stage for row xi_y: Y[xi_y][ix] -> V[xi_y * 4 + xi_x]
    auto x_op = [&](const f32x2 (&Y)[16], f32x2 (&V)[16], int fy) {
        V[fy * 4 + 0] = pk_sub(Y[fy * 4 + 0], Y[fy * 4 + 2]);
        V[fy * 4 + 1] = pk_add(Y[fy * 4 + 1], Y[fy * 4 + 2]);
        V[fy * 4 + 2] = pk_sub(Y[fy * 4 + 2], Y[fy * 4 + 1]);
        V[fy * 4 + 3] = pk_sub(Y[fy * 4 + 1], Y[fy * 4 + 3]);
    };

    // INAFF: the five piece PAIRS a lane fetched of a chunk - groups 0 and 1: quads (0, 1) and (2, 3) of its range, group 2: its two
    // quads of range 8 / 9 - are normalised in three phases each (read, compute, write), dealt over MFMA gaps by the caller.
    struct AffPair { f32x4 r0, r1; };
    auto aff_slot = [&](float *b, auto gc, auto jc) -> float * {
        constexpr int g = decltype(gc)::value, j = decltype(jc)::value;
        const int rng = g < 2 ? wave + 4 * g : 8 + (wave >> 1);
        const int qa = g < 2 ? 2 * j : 2 * (wave & 1);
        return b + (qa * W3_PS + rng * 64) * 4 + lane * 4;
    };
    // Scale / shift of the staged chunk's 16 channels (wave-uniform): eight 16-byte VECTOR loads per chunk with a scalar base and a zero
    // lane offset, in inline asm, issued in step 0 in front of DMA group 1 - step 1's vmcnt(4) (everything but that group's four
    // DMAs) retires them with the weights, so the hand-counted waits do not change; used in step 2.
    // History: (1) plain global loads - hipcc made vector loads of them and, the asm statements around clobbering "memory", put
    // vmcnt(0) in front of every use: +17 %; (2) scalar loads through the constant address space, four per piece pair in step 2 - a
    // scalar load returns out of order, so its wait is lgkmcnt(0), which also waits for every LDS read in flight: five such stalls
    // per chunk (stamps: step 2 3 580 cycles against 2 350 without the fused norm); (3) scalar loads once per chunk, copied to
    // vector registers two gaps later - 32 more live scalar registers pushed the lane-spill registers into scratch, whose loads
    // are vector memory operations: vmcnt(0) all over step 2, 5 930 cycles.
    // (The table is written by norm_finalize, a previous launch.)
    f32x4 aff_sc[4], aff_sh[4];          // [channel quad]
    auto aff_table_load = [&](int n, int ch) {
        auto uni = [](const float *x) {
            const unsigned long long v = (unsigned long long)x;
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
            return (const float *)(((unsigned long long)hi << 32) | lo);
        };
        const float *q = uni(p.in_scale + (size_t)n * p.C0 + ch * 16), *r = uni(p.in_shift + (size_t)n * p.C0 + ch * 16);
        unsigned zoff = 0;
        asm volatile("" : "+v"(zoff));
#define W3_TLOAD(DST, SBASE, IMM) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(DST) : "v"(zoff), "s"(SBASE), "n"(IMM) : "memory")
        asm volatile("s_nop 4" ::: "memory");  // (q, r come straight from v_readfirstlane_b32: the five wait states of W3_ULOAD0)
        W3_TLOAD(aff_sc[0], q, 0); W3_TLOAD(aff_sc[1], q, 16); W3_TLOAD(aff_sc[2], q, 32); W3_TLOAD(aff_sc[3], q, 48);
        W3_TLOAD(aff_sh[0], r, 0); W3_TLOAD(aff_sh[1], r, 16); W3_TLOAD(aff_sh[2], r, 32); W3_TLOAD(aff_sh[3], r, 48);
#undef W3_TLOAD
    };
    // ("redefined here": the compiler must not read or move the table registers in front of the wait that retires their loads)
    auto aff_table_landed = [&]() {
        asm volatile("" : "+v"(aff_sc[0]), "+v"(aff_sc[1]), "+v"(aff_sc[2]), "+v"(aff_sc[3]), "+v"(aff_sh[0]), "+v"(aff_sh[1]), "+v"(aff_sh[2]), "+v"(aff_sh[3]));
    };
    auto aff_read = [&](float *b, auto gc, auto jc, AffPair &a) {
        const float *s0 = aff_slot(b, gc, jc);
        a.r0 = *(const f32x4 *)s0;
        a.r1 = *(const f32x4 *)(s0 + W3_PS * 4);
    };
    auto aff_apply1 = [&](f32x4 &r, const f32x4 &sc, const f32x4 &sh) {
        const f32x2 sl2 = {p.in_slope, p.in_slope};
        f32x2 a = {r[0], r[1]}, b = {r[2], r[3]}, ya, yb, za, zb;
        const f32x2 s01 = {sc[0], sc[1]}, s23 = {sc[2], sc[3]}, t01 = {sh[0], sh[1]}, t23 = {sh[2], sh[3]};
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(ya) : "v"(a), "v"(s01), "v"(t01));
        asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(yb) : "v"(b), "v"(s23), "v"(t23));
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(za) : "v"(ya), "v"(sl2));
        asm("v_pk_mul_f32 %0, %1, %2" : "=v"(zb) : "v"(yb), "v"(sl2));
        asm("v_max_f32 %0, %1, %2" : "=v"(r[0]) : "v"(ya[0]), "v"(za[0]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r[1]) : "v"(ya[1]), "v"(za[1]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r[2]) : "v"(yb[0]), "v"(zb[0]));
        asm("v_max_f32 %0, %1, %2" : "=v"(r[3]) : "v"(yb[1]), "v"(zb[1]));
    };
    // half 0 / 1 of a pair = its first / second quad: groups 0, 1 hold quads (2 j, 2 j + 1), group 2 quads (2 (wave & 1), + 1)
    auto aff_apply = [&](auto gc, auto jc, auto hc, AffPair &a) {
        constexpr int g = decltype(gc)::value, j = decltype(jc)::value, h = decltype(hc)::value;
        f32x4 &r = h == 0 ? a.r0 : a.r1;
        if constexpr (g < 2) aff_apply1(r, aff_sc[2 * j + h], aff_sh[2 * j + h]);
        else {
            // (wave-uniform: four v_cndmask per operand.  The operands are laundered first: hipcc turned `w1 ? t[2 + h] : t[h]` into a
            //  dynamically indexed read of a PRIVATE copy of the table - scratch loads, i.e. vector memory operations the
            //  hand-counted vmcnt waits know nothing of)
            const bool w1 = wave & 1;
            f32x4 sa = aff_sc[h], sb = aff_sc[2 + h], ta = aff_sh[h], tb = aff_sh[2 + h];
            asm volatile("" : "+v"(sa), "+v"(sb), "+v"(ta), "+v"(tb));
            const f32x4 sc = w1 ? sb : sa, sh = w1 ? tb : ta;
            aff_apply1(r, sc, sh);
        }
    };
    auto aff_write = [&](float *b, auto gc, auto jc, const AffPair &a) {
        constexpr int g = decltype(gc)::value;
        const bool in = (inmask >> g) & 1;
        float *s0 = aff_slot(b, gc, jc), *jk = junk + lane * 4;
        float *d0 = in ? s0 : jk, *d1 = in ? s0 + W3_PS * 4 : jk;
        *(f32x4 *)d0 = a.r0;
        *(f32x4 *)d1 = a.r1;
    };
    typedef std::integral_constant<int, 0> I0; typedef std::integral_constant<int, 1> I1; typedef std::integral_constant<int, 2> I2;

    // Pipeline (one barrier per chunk, nothing exposed but the very first transform of the kernel):
    //   chunk c, steps 0 1 2: MFMAs of quads 0 1 2; between them the transform of the next quad (from brick buffer c), the next
    //                         step's weights and - steps 0 and 1 - DMA groups 1 and 2 of chunk c + 1 into the other buffer;
    //   [vmcnt(0) + barrier]: chunk c + 1 is in LDS for everybody, buffer c is no longer read by anybody;
    //   step 3:               MFMAs of quad 3; between them the transform of quad 0 of chunk c + 1 (from the other buffer),
    //                         the weights of that step, and DMA group 0 of chunk c + 2 into buffer c.
    // The brick DMA therefore runs as its own stream of (tile, chunk) positions, one chunk ahead of the MFMAs and across tile
    // boundaries; past the last chunk it re-stages the last one (nobody reads it), so the wait counts stay fixed.
    W3_DECL;
    W3_T(t_kernel0);
    TileCoord cur = decode(tile);
    // The stream's next position, in eight scalar pieces (adv0 .. adv5) that the chunk loop deals over eight MFMA gaps of step 1,
    // behind the issue of group 2: branch-free (a branch would cut the MFMA stream into basic blocks), ~10-15 scalar
    // instructions each, which the scalar unit runs while the matrix pipe works on the MFMA in front of the gap.
    int a_tt = 0, a_tx = 0, a_ty = 0, a_tz = 0, a_vox = 0, a_csrc = 0, a_coff = 0;
    const float *a_ptr = nullptr;
    auto adv0 = [&]() {   // (tile, chunk) <- next; past the last chunk of the last tile the position stays (re-staged, nobody reads it)
        const bool more_ch = d_ch + 1 < p.nchunks;
        const bool more_tiles = d_tile + nl < hi;
        d_ch = more_ch ? d_ch + 1 : (more_tiles ? 0 : d_ch);
        d_tile = (!more_ch && more_tiles) ? d_tile + nl : d_tile;
    };
    auto adv1 = [&]() {
        d_tc.n = (int)fdiv((uint32_t)d_tile, p.div_tiles_per_n);
        a_tt = d_tile - d_tc.n * (int)p.div_tiles_per_n.d;
    };
    int a_w = 0, a_bzy = 0, a_bxi = 0;
    auto adv2a = [&]() {  // tile_from_id (common.h) in two halves
        const int lb = p.order.lbx + p.order.lby + p.order.lbz;
        a_w = a_tt & ((1 << lb) - 1);
        const int blk = a_tt >> lb;
        a_bzy = (int)fdiv((uint32_t)blk, p.order.div_nbx);
        a_bxi = blk - a_bzy * (int)p.order.div_nbx.d;
    };
    auto adv2b = [&]() {
        const int bzi = (int)fdiv((uint32_t)a_bzy, p.order.div_nby);
        const int byi = a_bzy - bzi * (int)p.order.div_nby.d;
        a_tx = (a_bxi << p.order.lbx) + (a_w & ((1 << p.order.lbx) - 1));
        a_ty = (byi << p.order.lby) + ((a_w >> p.order.lbx) & ((1 << p.order.lby) - 1));
        a_tz = (bzi << p.order.lbz) + (a_w >> (p.order.lbx + p.order.lby));
    };
    auto adv3 = [&]() {
        d_tc.oz0 = a_tz << 2; d_tc.oy0 = a_ty << 3; d_tc.ox0 = a_tx << 3;
        const int cglob = d_ch * 16;
        d_sel1 = cglob >= p.C0;
        a_ptr = d_sel1 ? p.in1 : p.in0;
        a_csrc = d_sel1 ? p.C1 : p.C0;
        a_coff = d_sel1 ? cglob - p.C0 : cglob;
    };
    auto adv4a = [&]() {  // (host: N * D * H * W < 2^30, wino3_fits)
        a_vox = ((d_tc.n * p.D + (d_tc.oz0 - 1)) * p.H + (d_tc.oy0 - 1)) * p.W + (d_tc.ox0 - 1);
    };
    auto adv4b = [&]() {
        d_notok = 0x80000000u | (d_tc.oz0 == 0 ? 1u : 0u) | (d_tc.oz0 + 4 == p.D ? 1u << 5 : 0u) | (d_tc.oy0 == 0 ? 1u << 6 : 0u) |
                  (d_tc.oy0 + 8 == p.H ? 1u << 15 : 0u) | (d_tc.ox0 == 0 ? 1u << 16 : 0u) | (d_tc.ox0 + 8 == p.W ? 1u << 25 : 0u);
    };
    auto adv5 = [&]() { d_src = (const char *)a_ptr + ((long)a_vox * a_csrc + a_coff) * 4; };
    d_tile = tile; d_ch = 0;
    adv1(); adv2a(); adv2b(); adv3(); adv4a(); adv4b(); adv5();
    static_for<0, 3>([&](auto kc) { dma_group(kc, I0{}, lds); dma_group(kc, I1{}, lds); });
    f32x4 uq[2][8];
    static_for<0, 8>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        auto &u0 = uq[0]; const unsigned wl = k < 4 ? wl0 : wl1; const float *wb = wblk;
        if constexpr (k == 0) W3_ULOAD0(u0[k], wl, wb, (k & 3) * 1024); else W3_ULOAD(u0[k], wl, wb, (k & 3) * 1024);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int t_n = d_tc.n, t_ch = d_ch;  // INAFF: (sample, chunk) of the brick being staged - the DMA stream moves on before the brick is normalised
    if constexpr (INAFF) {  // the kernel's first brick: normalised here, exposed once
        AffPair a;
        aff_table_load(t_n, t_ch);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        aff_table_landed();
        aff_read(lds, I0{}, I0{}, a); aff_apply(I0{}, I0{}, I0{}, a); aff_apply(I0{}, I0{}, I1{}, a); aff_write(lds, I0{}, I0{}, a);
        aff_read(lds, I0{}, I1{}, a); aff_apply(I0{}, I1{}, I0{}, a); aff_apply(I0{}, I1{}, I1{}, a); aff_write(lds, I0{}, I1{}, a);
        aff_read(lds, I1{}, I0{}, a); aff_apply(I1{}, I0{}, I0{}, a); aff_apply(I1{}, I0{}, I1{}, a); aff_write(lds, I1{}, I0{}, a);
        aff_read(lds, I1{}, I1{}, a); aff_apply(I1{}, I1{}, I0{}, a); aff_apply(I1{}, I1{}, I1{}, a); aff_write(lds, I1{}, I1{}, a);
        aff_read(lds, I2{}, I0{}, a); aff_apply(I2{}, I0{}, I0{}, a); aff_apply(I2{}, I0{}, I1{}, a); aff_write(lds, I2{}, I0{}, a);
    }
    __syncthreads();
    f32x2 dA[8], dB[8], T[16], Y[16], V[2][16];
    {   // the first transform of the kernel (exposed once)
        read8(lds, 0, 0, dA, dB);
        z_op(dA, dB, T, 0);
        read8(lds, 0, 1, dA, dB);
        z_op(dA, dB, T, 1);
#pragma unroll
        for (int ix = 0; ix < 4; ++ix) y_op(T, Y, ix);
#pragma unroll
        for (int fy = 0; fy < 4; ++fy) x_op(Y, V[0], fy);
    }
    adv0(); adv1(); adv2a(); adv2b(); adv3(); adv4a(); adv4b(); adv5();
    dma_group(I0{}, I0{}, lds + W3_BUF); dma_group(I0{}, I1{}, lds + W3_BUF);  // ("step 3 of chunk -1")
    {   // (the first tile's first step has no weight wait of its own: retire uq[0] here, behind the DMA group - once per kernel)
        auto &u0 = uq[0];
        W3_UWAIT(u0, 0);
    }

    // Kernel invariants of the epilogue's phase 2 (lane = 4 couts `piece` of a voxel): bias, and for the fused head its weights and
    // bias.  Loaded once: as loads inside phase 2 their L2 round trip was exposed once per tile (vmcnt(0) in front of the first
    // use, nothing to hide it behind).
    const f32x4 bias_k = *(const f32x4 *)(p.bias + (int)blockIdx.y * 32 + (lane & 7) * 4);
    f32x4 hq_k[4] = {};
    float hb_k = 0.f;
    if constexpr (EPI == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            hq_k[c] = c < p.head_ncls ? *(const f32x4 *)(p.head_w + c * p.Cout + (int)blockIdx.y * 32 + (lane & 7) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        hb_k = p.head_b[(lane & 7) < p.head_ncls ? (lane & 7) : 0];
    }
    double stat_acc = 0.0;  // EPI 2, wave 0: this workgroup's quantised statistics of sample stat_n (lane = cout, statistic)
    int stat_n = -1;
    int buf = 0;
    for (; tile < hi; tile += nl) {
        W3_T(t_t0);
        f32x16 acc[16];
#if !MI355_W3_TWOBODY
#pragma unroll
        for (int f = 0; f < 16; ++f)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
#endif
        const int ntile = tile + nl;
        const TileCoord nxt_tile = ntile < hi ? decode(ntile) : cur;
        W3_T(t_t1);
        W3_ACC(9, t_t0, t_t1);
        // Two bodies (MI355_W3_TWOBODY, round 5): a second copy of the chunk body for a tile's chunk 0 whose first MFMAs take the
        // constant 0 as C saves the 256 v_accvgpr_write of the accumulator clear (1 - 3 % per launch, tools/wino3_probe).  Round 4
        // abandoned it after "a memory access fault on every shape".  The cause, from that build's listing: with the second body
        // the allocator spilled the weight base of a step to a VGPR lane and reloaded it (v_readlane_b32) one instruction in
        // front of the inline-asm load that uses it as scalar base - five wait states short (W3_ULOAD0 above).  The build gate
        // (_isa_gate.py) checks every listing for that sequence; positive and negative control in profiles/r05_wino3_two_body.txt.
        auto chunk_body = [&](auto first_c, const int ch) {
            constexpr bool FIRST = decltype(first_c)::value;  // two-body build: a tile's chunk 0, whose first MFMAs take C = 0
            const bool last_ch = ch == p.nchunks - 1;
            const float *bufc = lds + buf * W3_BUF;
            float *bufn = lds + (buf ^ 1) * W3_BUF;
            const float *wch = wblk + (size_t)ch * (STEPS * 4 * 2048);
            const float *wnx = wblk + (size_t)(last_ch ? 0 : ch + 1) * (STEPS * 4 * 2048);
            W3_T(t_c1);
            AffPair aff, aff2;
            static_for<0, STEPS>([&](auto st_c) {
                constexpr int st = decltype(st_c)::value;
                constexpr int pp = st & 1;
                auto &uc = uq[pp];
                // this step's weights: everything older than the 4 brick DMAs the previous step issued behind them.  A tile's first
                // step finds its weights retired already (the epilogue waits for them before its stores); step 3 follows the
                // chunk barrier's vmcnt(0)
                if constexpr (st == 0) { if constexpr (!FIRST) { if (MI355_W3_TWOBODY || ch != 0) W3_UWAIT(uc, 4); } }
                else if constexpr (st == 1) { W3_UWAIT(uc, 4); if constexpr (INAFF) aff_table_landed(); }
                else if constexpr (st == 2) W3_UWAIT(uc, 2);  // (group 2 is two DMAs)
                else W3_UWAIT(uc, 0);
                const float *wn = (st + 1 < STEPS) ? wch + (size_t)(st + 1) * (4 * 2048) : wnx;
                const float *rb = (st + 1 < STEPS) ? bufc : bufn;   // brick the next quad is read from
                constexpr int rq = (st + 1) & 3;
                W3_T(t_s0);
                static_for<0, 32>([&](auto i_c) {
                    constexpr int i = decltype(i_c)::value;
                    constexpr int f = i & 15, j = i >> 4;
                    if constexpr (FIRST && st == 0 && j == 0) {
                        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(uq[pp][f >> 1][(f & 1) * 2 + j], V[pp][f][j], zero16, 0, 0, 0);
                    } else
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(uq[pp][f >> 1][(f & 1) * 2 + j], V[pp][f][j], acc[f], 0, 0, 0);
                    if constexpr ((MI355_W3_ABL & 16) == 0) {
                        // V of the next quad, dealt over the MFMA gaps in bunches (a gap that holds VALU work costs the matrix
                        // pipe ~5 cycles + ~4.4 per instruction, tools/coissue_probe.hip)
                        if constexpr (i == 0) read8(rb, rq, 0, dA, dB);
                        if constexpr (i == 6) z_op(dA, dB, T, 0);
                        if constexpr (i == 7) read8(rb, rq, 1, dA, dB);
                        if constexpr (i == 13) z_op(dA, dB, T, 1);
                        if constexpr (i == 14) { y_op(T, Y, 0); y_op(T, Y, 1); }
                        if constexpr (i == 15) { y_op(T, Y, 2); y_op(T, Y, 3); }
                        if constexpr (i == 17) { x_op(Y, V[pp ^ 1], 0); x_op(Y, V[pp ^ 1], 1); }
                        if constexpr (i == 18) { x_op(Y, V[pp ^ 1], 2); x_op(Y, V[pp ^ 1], 3); }
                    }
                    if constexpr (i < 16 && (i & 1) == 0) {  // the next step's weights - unconditionally: a branch per load would cut
                        constexpr int k = i >> 1;            // the MFMA stream into basic blocks.  A tile's last step fetches the next
                        auto &un = uq[pp ^ 1]; const unsigned wl = k < 4 ? wl0 : wl1; const float *wb = wn;  // tile's first fragments
                        if constexpr ((MI355_W3_ABL & 8) == 0) {                                               // (same cout block, chunk 0)
                            if constexpr (k == 0) W3_ULOAD0(un[k], wl, wb, (k & 3) * 1024); else W3_ULOAD(un[k], wl, wb, (k & 3) * 1024);
                        }
                    }
                    if constexpr (INAFF && st == 2) {
                        // the brick of chunk c + 1 (in the other buffer) is normalised here, in front of the barrier that publishes it.
                        // Groups 0 and 1 have landed (this step's weight wait left only group 2's two DMAs in flight); group 2 is
                        // retired by vmcnt(8) at i = 24: only this step's eight weight loads (i < 16) are younger.
                        // (two pairs in flight, each pair's arithmetic split over two gaps: 8 vector instructions per gap - 16 in one
                        //  gap ran over the 64 cycles an MFMA covers; stamps: step 2 of this instantiation 3 578 cycles against 2 346)
                        if constexpr (i == 1) aff_read(bufn, I0{}, I0{}, aff);
                        if constexpr (i == 2) aff_read(bufn, I0{}, I1{}, aff2);
                        if constexpr (i == 3) aff_apply(I0{}, I0{}, I0{}, aff);
                        if constexpr (i == 4) aff_apply(I0{}, I0{}, I1{}, aff);
                        if constexpr (i == 5) aff_write(bufn, I0{}, I0{}, aff);
                        if constexpr (i == 8) aff_apply(I0{}, I1{}, I0{}, aff2);
                        if constexpr (i == 9) aff_apply(I0{}, I1{}, I1{}, aff2);
                        if constexpr (i == 10) aff_write(bufn, I0{}, I1{}, aff2);
                        if constexpr (i == 11) aff_read(bufn, I1{}, I0{}, aff);
                        if constexpr (i == 12) aff_read(bufn, I1{}, I1{}, aff2);
                        if constexpr (i == 16) aff_apply(I1{}, I0{}, I0{}, aff);
                        if constexpr (i == 19) aff_apply(I1{}, I0{}, I1{}, aff);
                        if constexpr (i == 20) aff_write(bufn, I1{}, I0{}, aff);
                        if constexpr (i == 21) aff_apply(I1{}, I1{}, I0{}, aff2);
                        if constexpr (i == 22) aff_apply(I1{}, I1{}, I1{}, aff2);
                        if constexpr (i == 23) aff_write(bufn, I1{}, I1{}, aff2);
                        if constexpr (i == 24) { asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); aff_read(bufn, I2{}, I0{}, aff); }
                        if constexpr (i == 26) aff_apply(I2{}, I0{}, I0{}, aff);
                        if constexpr (i == 27) aff_apply(I2{}, I0{}, I1{}, aff);
                        if constexpr (i == 28) aff_write(bufn, I2{}, I0{}, aff);
                    }
                    if constexpr ((MI355_W3_ABL & 4) == 0 && (i == 20 || i == 21)) {
                        typedef std::integral_constant<int, i - 20> Part;
                        if constexpr (st == 0) dma_group(I1{}, Part{}, bufn);
                        if constexpr (st == 1 && i == 20) { dma_group(I2{}, I0{}, bufn); t_n = d_tc.n; t_ch = d_ch; }
                        if constexpr (st == 3) dma_group(I0{}, Part{}, const_cast<float *>(bufc));
                    }
                    if constexpr ((MI355_W3_ABL & 4) == 0 && st == 1) {  // the stream moves on: one scalar piece per gap
                        if constexpr (i == 21) adv0();
                        if constexpr (i == 22) adv1();
                        if constexpr (i == 23) adv2a();
                        if constexpr (i == 24) adv2b();
                        if constexpr (i == 25) adv3();
                        if constexpr (i == 26) adv4a();
                        if constexpr (i == 27) adv4b();
                        if constexpr (i == 28) adv5();
                    }
                    if constexpr (INAFF && st == 0 && i == 17) aff_table_load(d_tc.n, d_ch);  // (the chunk whose DMA groups 1, 2 follow)
                    __builtin_amdgcn_sched_barrier(0);
                });
                W3_T(t_s1);
                W3_ACC(11 + st, t_s0, t_s1);
                if constexpr (st == 2) {
                    W3_T(t_c2);
                    W3_ACC(1, t_c1, t_c2);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a ds_read is ordered behind an LDS-DMA only by the issuer's vmcnt + a barrier
                    __syncthreads();
                    W3_T(t_c3);
                    W3_ACC(2, t_c2, t_c3);
                }
            });
            W3_T(t_c4);
            W3_ACC(10, t_c1, t_c4);
            buf ^= 1;
            W3_CNT(6);
        };
#if MI355_W3_TWOBODY
        chunk_body(std::true_type{}, 0);
        for (int ch = 1; ch < p.nchunks; ++ch) chunk_body(std::false_type{}, ch);
#else
        for (int ch = 0; ch < p.nchunks; ++ch) chunk_body(std::false_type{}, ch);
#endif
        W3_T(t_e0);
        if constexpr ((MI355_W3_ABL & 1) != 0) {
#pragma unroll
            for (int f = 0; f < 16; ++f) asm volatile("" :: "a"(acc[f]));
        } else {
            // ---- phase 1: A^T along y and x inside the wave (16 components -> 4 partial outputs per cout), one accumulator
            // register pair at a time; the partials go to this wave's four staged images [oy * 2 + ox][block][cout]
            int lane_e;  // rebuilt from the hardware lane id: a tile-loop invariant would be hoisted to the kernel entry and spilled
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e));
            float *wr = stage + wave * (4 * W3_IMG) + (lane_e & 31) * W3_PITCH + 4 * (lane_e >> 5);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                // The accumulators are "redefined" by an empty asm at the top of every register pair: the 32 v_accvgpr_read of pair r
                // cannot be hoisted above it.  A scheduling barrier alone did not hold them: hipcc read ~170 accumulator registers
                // into VGPRs before the first staged write, which is what put every instantiation at the 256-VGPR limit with 7-10
                // spilled loop invariants (without the epilogue the kernel needs 152 registers).
                if ((r / 2) % MI355_W3_PIN == 0) {
#pragma unroll
                    for (int f = 0; f < 16; ++f) asm volatile("" : "+a"(acc[f]));
                }
                f32x2 P[2][4];  // [oy][xi_x]
#pragma unroll
                for (int fx = 0; fx < 4; ++fx) {
                    const f32x2 a0 = {acc[0 * 4 + fx][r], acc[0 * 4 + fx][r + 1]}, a1 = {acc[1 * 4 + fx][r], acc[1 * 4 + fx][r + 1]};
                    const f32x2 a2 = {acc[2 * 4 + fx][r], acc[2 * 4 + fx][r + 1]}, a3 = {acc[3 * 4 + fx][r], acc[3 * 4 + fx][r + 1]};
                    P[0][fx] = pk_add(pk_add(a0, a1), a2);
                    P[1][fx] = pk_sub(pk_sub(a1, a2), a3);
                }
                const int co = (r & 3) + 8 * (r >> 2);  // + 4 * half: in wr
#pragma unroll
                for (int oy = 0; oy < 2; ++oy) {
                    *(f32x2 *)(wr + (oy * 2 + 0) * W3_IMG + co) = pk_add(pk_add(P[oy][0], P[oy][1]), P[oy][2]);
                    *(f32x2 *)(wr + (oy * 2 + 1) * W3_IMG + co) = pk_sub(pk_sub(P[oy][1], P[oy][2]), P[oy][3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            W3_T(t_e1);
            W3_ACC(3, t_e0, t_e1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // (raw: a __syncthreads() would also drain the brick DMAs of the next tile)
            asm volatile("" ::: "memory");  // the intrinsic is IntrNoMem: without this hipcc may issue the read-back ABOVE the barrier
            // the next tile's first weight fragments (fetched during step 3) are retired BEFORE this tile's stores are issued: loads
            // and stores of a wave may complete out of order with each other, a count-based wait taken behind the stores could not
            // tell them apart.  The next tile's first step then starts without any wait.
            { auto &u0 = uq[0]; W3_UWAIT(u0, 0); }

            // ---- phase 2: wave (oy, ox) adds the four xi_z partials (A^T along z), bias, LeakyReLU, whole-line stores.
            // Lane = (block row srow + 8 t, 4 couts `piece`): 8 lanes hold the 32 couts = the 128-B line of one voxel.
            const int oy = wave >> 1, ox = wave & 1;
            const int srow = lane_e >> 3, piece = lane_e & 7;
            const float *rd = stage + wave * W3_IMG + srow * W3_PITCH + piece * 4;
            const int co0 = (int)blockIdx.y * 32;
            f32x4 bias = bias_k;
            asm volatile("" : "+v"(bias));  // (a copy per tile: the register pairs below are formed from it)
            const f32x2 b01 = {bias[0], bias[1]}, b23 = {bias[2], bias[3]};
            float slope = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.act == ACT_LRELU ? p.slope : 1.0f)));
            asm volatile("" : "+s"(slope));
            const f32x2 slope2 = {slope, slope};
            // voxel of block b = 8 t + srow: bx = b & 3, by = (b >> 2) & 3 = 2 (t & 1) + (srow >> 2), bz = b >> 4 = t >> 1
            const int vx = cur.ox0 + 2 * (srow & 3) + ox;
            const size_t row_elems = (size_t)p.W * p.Cout;
            float *obase = p.out + (((size_t)cur.n * p.D + cur.oz0) * p.H + cur.oy0 + oy) * row_elems + co0;  // wave-uniform
            const unsigned lane_off = (unsigned)(((2 * (srow >> 2)) * p.W + vx) * p.Cout + piece * 4);
            float st1[4] = {0.f, 0.f, 0.f, 0.f}, st2[4] = {0.f, 0.f, 0.f, 0.f};
            // EPI == 1: logit[c] = sum_cout w[c][cout] * act(y[cout] + b[cout]) + hb[c] (generic_UNet.py:389-391, 1x1x1, no bias in the
            // reference's head).  A lane holds four couts of its voxel and the other 28 sit in the seven lanes beside it: four fmas per
            // class, then three DPP additions across the 8-lane group; lane `piece` = c stores class c.
            // Round 4 (SQ pass: 7.6 VALU per MFMA, pipe 0.53 busy on this instantiation): the store address is a scalar base per output
            // row + ONE tile-invariant lane offset (was: a 64-bit index product per row and lane), the class a lane stores is
            // picked with v_cndmask from totals that are computed unconditionally (hipcc had sunk them into 35 branches).
            constexpr int KMAX = 4;
            f32x4 hq[KMAX];
            float hb = 0.f;
            const unsigned Vo = (unsigned)(p.D * p.H * p.W);  // (host: head_ncls * D * H * W * 4 < 2^32)
            unsigned lane_h = 0;
            int pc = 0;
            float *hrow0 = nullptr;
            if constexpr (EPI == 1) {
#pragma unroll
                for (int c = 0; c < KMAX; ++c) hq[c] = hq_k[c];
                pc = piece < p.head_ncls ? piece : 0;
                hb = hb_k;
                lane_h = (unsigned)pc * Vo + (unsigned)((2 * (srow >> 2)) * p.W + 2 * (srow & 3));
                // voxel (oz0, oy0 + oy, ox0 + ox) of class 0 of this sample: wave-uniform
                hrow0 = p.head_out + (size_t)cur.n * p.head_ncls * Vo + ((size_t)cur.oz0 * p.H + cur.oy0 + oy) * p.W + cur.ox0 + ox;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x4 pz[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) pz[k] = *(const f32x4 *)(rd + k * (4 * W3_IMG) + 8 * t * W3_PITCH);
                // row of blocks (bz = t >> 1, by = 2 (t & 1) + ...): z = oz0 + 2 bz + oz, y = oy0 + 2 by + oy
                float *rowp = obase + ((size_t)(2 * (t >> 1)) * p.H + 4 * (t & 1)) * row_elems;
#pragma unroll
                for (int oz = 0; oz < 2; ++oz) {
                    f32x2 x0, x1, y0, y1;
                    if (oz == 0) {
                        x0 = pk_add(pk_add(f32x2{pz[0][0], pz[0][1]}, f32x2{pz[1][0], pz[1][1]}), f32x2{pz[2][0], pz[2][1]});
                        x1 = pk_add(pk_add(f32x2{pz[0][2], pz[0][3]}, f32x2{pz[1][2], pz[1][3]}), f32x2{pz[2][2], pz[2][3]});
                    } else {
                        x0 = pk_sub(pk_sub(f32x2{pz[1][0], pz[1][1]}, f32x2{pz[2][0], pz[2][1]}), f32x2{pz[3][0], pz[3][1]});
                        x1 = pk_sub(pk_sub(f32x2{pz[1][2], pz[1][3]}, f32x2{pz[2][2], pz[2][3]}), f32x2{pz[3][2], pz[3][3]});
                    }
                    f32x4 val;
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(x0) : "v"(x0), "v"(b01));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(x1) : "v"(x1), "v"(b23));
                    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(slope2));
                    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(slope2));
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[0]) : "v"(x0[0]), "v"(y0[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[1]) : "v"(x0[1]), "v"(y0[1]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[2]) : "v"(x1[0]), "v"(y1[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(val[3]) : "v"(x1[1]), "v"(y1[1]));
                    if constexpr (EPI == 2) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { st1[k] += val[k]; st2[k] = fmaf(val[k], val[k], st2[k]); }
                    }
                    if constexpr (EPI == 1) {
                        float tot[KMAX];
#pragma unroll
                        for (int c = 0; c < KMAX; ++c) {
                            float v = val[0] * hq[c][0];
                            v = fmaf(val[1], hq[c][1], v); v = fmaf(val[2], hq[c][2], v); v = fmaf(val[3], hq[c][3], v);
                            v += dpp_perm<0xB1>(v);   // lane ^ 1
                            v += dpp_perm<0x4E>(v);   // lane ^ 2
                            v += dpp_perm<0x141>(v);  // row_half_mirror: the other quad of the 8-lane group
                            tot[c] = v;
                        }
                        asm volatile("" : "+v"(tot[0]), "+v"(tot[1]), "+v"(tot[2]), "+v"(tot[3]));  // (no sinking into the selects)
                        float mine = tot[0];   // (lanes beyond the last class repeat lane 0's store: same address, same value - no exec mask, no branch)
                        mine = pc == 1 ? tot[1] : mine;
                        mine = pc == 2 ? tot[2] : mine;
                        mine = pc == 3 ? tot[3] : mine;
                        // row (z = oz0 + 2 (t >> 1) + oz, y = oy0 + oy + 4 (t & 1) [+ 2 (srow >> 2): in lane_h])
                        float *hrow = hrow0 + (size_t)((2 * (t >> 1) + oz) * p.H + 4 * (t & 1)) * p.W;
                        hrow[lane_h] = mine + hb;
                    } else if constexpr ((MI355_W3_ABL & 32) != 0) asm volatile("" :: "v"(val));
                    else {
                        float *gp = rowp + (size_t)oz * p.H * row_elems + lane_off;
                        // sc1: nothing on this XCD reads the line again (conv3d.hip, FETCH_SIZE -38 % on the 32 -> 32 layer).
                        // s_nop 2: the VALU instructions of the next output row may be allocated onto these four data registers right
                        // behind the store; with one wait state (what conv3d.hip's stores carry) dword 1 of the lanes that are read out
                        // last came out as the NEXT row's intermediate (tools/wino3_probe: couts 17, 21, 25, 29 of every other block)
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 2" :: "v"(gp), "v"(val) : "memory");
                    }
                }
            }
            if constexpr (EPI == 2) {
                // The eight lanes with the same `piece` (lane bits 3..5) hold the same four couts of different voxels: 8 values per lane
                // (sum x and sum x^2 of 4 couts) to be added over those three lane bits.  A halving reduce-scatter (common.h,
                // half32_reduce_scatter): v_permlane32_swap pairs lane ^ 32 (lanes 0-31 keep the sums, lanes 32-63 the sums of
                // squares), v_permlane16_swap lane ^ 16 (rows 0, 2 keep couts 0, 1, rows 1, 3 couts 2, 3), one DPP row_ror:8 step
                // lane ^ 8 - 7 additions and 6 swaps; every lane ends with ONE total.  (Was: a three-step __shfl_xor butterfly per
                // value = 24 ds_bpermute round trips in dependent chains; stamps: this epilogue 6 100 cycles per tile against
                // 4 270 without statistics.)
                float y[4];
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    float a0 = st1[k], a1 = st1[k + 1], c0 = st2[k], c1 = st2[k + 1];
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1));
                    y[k] = a0 + c0; y[k + 1] = a1 + c1;
                }
                float z0, z1;
                {
                    float a0 = y[0], a1 = y[1], c0 = y[2], c1 = y[3];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\ts_nop 1" : "+v"(a0), "+v"(a1), "+v"(c0), "+v"(c1));
                    z0 = a0 + c0; z1 = a1 + c1;
                }
                const bool b3 = lane_e & 8;
                const float tot_l = (b3 ? z1 : z0) + dpp_perm<0x128>(b3 ? z0 : z1);  // row_ror:8 = lane ^ 8
                // this lane's value: statistic lane >> 5, cout 4 piece + 2 (lane >> 4 & 1) + (lane >> 3 & 1)
                red[(wave * 32 + 4 * piece + 2 * ((lane_e >> 4) & 1) + ((lane_e >> 3) & 1)) * 2 + (lane_e >> 5)] = tot_l;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (wave == 0) {
                    const int c = lane_e >> 1, k = lane_e & 1;
                    double tot = 0.0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) tot += (double)red[(w * 32 + c) * 2 + k];
                    // quantised partials add exactly in fp64 in any order (common.h): they are gathered per workgroup and sample and
                    // go out as ONE atomic per (sample, cout, statistic) and workgroup instead of one per tile - 65 536 tiles of a
                    // 128^3 x 8 layer were 4.2 M fp64 atomics on 512 addresses
                    if (cur.n != stat_n) {
                        if (stat_n >= 0) atomicAdd(p.stats + ((size_t)stat_n * p.Cout + co0 + c) * 2 + k, stat_acc);
                        stat_acc = 0.0; stat_n = cur.n;
                    }
                    stat_acc += quantise_partial(tot, k, (long)p.D * p.H * p.W);
                }
            }
        }
        cur = nxt_tile;
        W3_T(t_e2);
        W3_ACC(4, t_e0, t_e2);
        W3_CNT(7);
    }
    if constexpr (EPI == 2) {
        if (wave == 0 && stat_n >= 0)
            atomicAdd(p.stats + ((size_t)stat_n * p.Cout + (int)blockIdx.y * 32 + (lane >> 1)) * 2 + (lane & 1), stat_acc);
    }
    W3_T(t_kernel1);
    W3_ACC(5, t_kernel0, t_kernel1);
    W3_FLUSH;
#undef W3_ULOAD
#undef W3_UWAIT
}

// 3-D Winograd pack (floats): [cout block of 32][chunk of 16][quad q][xi_z][f / 2][lane][f & 1][j 0..1], f = xi_y * 4 + xi_x, with
//   cout = block * 32 + (lane & 31), cin = chunk * 16 + q * 4 + (lane >> 5) * 2 + j, U = G w G^T along all three tap axes,
//   G = [[1,0,0],[1/2,1/2,1/2],[1/2,-1/2,1/2],[0,0,1]], evaluated in fp64 and rounded once.
void pack_conv_weights_wino3(const float *w, int cin, int cin_pad, int cout, std::vector<float> &out) {
    const int nchunks = cin_pad / 16, nblk = cout / 32;
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    out.assign((size_t)nblk * nchunks * 4 * 4 * 2048, 0.f);
    // U of one (cout, cin): 64 components
    std::vector<double> U((size_t)cout * cin * 64);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const float *wk = &w[((size_t)co * cin + ci) * 27];
            double t1[4][3][3], t2[4][4][3];
            for (int fz = 0; fz < 4; ++fz)
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx) {
                        double s = 0.0;
                        for (int dz = 0; dz < 3; ++dz) s += Gm[fz][dz] * (double)wk[dz * 9 + dy * 3 + dx];
                        t1[fz][dy][dx] = s;
                    }
            for (int fz = 0; fz < 4; ++fz)
                for (int fy = 0; fy < 4; ++fy)
                    for (int dx = 0; dx < 3; ++dx) {
                        double s = 0.0;
                        for (int dy = 0; dy < 3; ++dy) s += Gm[fy][dy] * t1[fz][dy][dx];
                        t2[fz][fy][dx] = s;
                    }
            double *u = &U[((size_t)co * cin + ci) * 64];
            for (int fz = 0; fz < 4; ++fz)
                for (int fy = 0; fy < 4; ++fy)
                    for (int fx = 0; fx < 4; ++fx) {
                        double s = 0.0;
                        for (int dx = 0; dx < 3; ++dx) s += Gm[fx][dx] * t2[fz][fy][dx];
                        u[(fz * 4 + fy) * 4 + fx] = s;
                    }
        }
    size_t o = 0;
    for (int b = 0; b < nblk; ++b)
        for (int ch = 0; ch < nchunks; ++ch)
            for (int q = 0; q < 4; ++q)
                for (int fz = 0; fz < 4; ++fz)
                    for (int fp = 0; fp < 8; ++fp)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int fj = 0; fj < 4; ++fj, ++o) {
                                const int f = fp * 2 + (fj >> 1), j = fj & 1;
                                const int co = b * 32 + (lane & 31);
                                const int ci = ch * 16 + q * 4 + (lane >> 5) * 2 + j;
                                if (ci >= cin) continue;
                                out[o] = (float)U[((size_t)co * cin + ci) * 64 + fz * 16 + f];
                            }
}

static int wino3_mode() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("MI355_WINO3"); v = (e && e[0] == '0') ? 0 : 1; }
    return v;
}
bool conv3d_wino3_enabled() { return wino3_mode() != 0; }

// Does the F(2x2x2, 3x3x3) kernel take this call?  Stride 1, whole 4 x 8 x 8 tiles, 16-channel chunks on both halves of a virtual
// concat, enough tiles to fill the chip; the fused head needs Cout = 32; the fused input norm (INAFF) a single input tensor and
// the statistics instantiation (its consumer is a block of the same Instance/GroupNorm stage).
static bool wino3_fits(const ConvWeights &w, const ConvCall &c) {
    if (!wino3_mode() || !w.wp3_dev || w.stride != 1) return false;
    if (c.head_out && (w.cout != 32 || c.stats || c.head_ncls < 1 || c.head_ncls > 4 || !c.head_w || !c.head_b)) return false;
    if (c.head_out && (long)c.head_ncls * c.Di * c.Hi * c.Wi >= (1l << 30)) return false;  // the head's lane offset is 32 bits
    if (c.in_scale && (c.C1 != 0 || !c.stats || c.head_out || !c.in_shift)) return false;
    if (c.Di % 4 || c.Hi % 8 || c.Wi % 8 || c.C0 % 16 || c.C1 % 16 || w.cout % 32 || (c.C0 + c.C1) != w.cin_pad) return false;
    const long tiles = (long)(c.Wi / 8) * (c.Hi / 8) * (c.Di / 4) * c.N;
    // Enough (tile, cout block) units to fill the chip - or, for the deep levels (8 x 8^3 x 320 channels: 160 units of 20 chunks
    // each), enough chunks per unit that one tile per workgroup on 160 of the 256 CUs still beats the direct split-K kernel, which
    // executes 27/8 of the multiplies at 0.57 of the pipe (round 4: 0.26 -> 0.11 ms per launch).
    const long units = tiles * (w.cout / 32);
    if ((units < 1024 && !(units >= 96 && w.cin_pad >= 128)) || tiles >= (1l << 30)) return false;
    if ((long)c.Di * c.Hi * c.Wi * (c.C0 > c.C1 ? c.C0 : c.C1) >= (1l << 31)) return false;  // the per-lane part of a DMA address fits 32 bits
    if ((long)c.N * c.Di * c.Hi * c.Wi >= (1l << 30)) return false;  // the brick corner's voxel index is a 32-bit scalar (adv4)
    return true;
}

// Can the conv `w` of call shape `c` (N, Di, Hi, Wi, C0, stats set as it will be called) apply its producer's normalisation itself?
bool conv3d_wino3_fuses_input_norm(const ConvWeights &w, const ConvCall &c) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("MI355_FUSE_NORM"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on) return false;
    ConvCall t = c;
    static const float dummy = 0.f;
    t.in_scale = &dummy; t.in_shift = &dummy;
    return wino3_fits(w, t);
}

int conv3d_wino3_f32(const ConvWeights &w, const ConvCall &c, hipStream_t s, const char **kernel_name, bool *taken) {
    *taken = false;
    if (!wino3_fits(w, c)) return MI355_OK;
    const int tx = c.Wi / 8, ty = c.Hi / 8, tz = c.Di / 4;
    const long tiles = (long)tx * ty * tz * c.N;
    Wino3Args a;
    a.in0 = c.in0; a.in1 = c.in1; a.wp = w.wp3_dev; a.bias = w.bias_dev; a.out = c.out; a.stats = c.stats;
    a.C0 = c.C0; a.C1 = c.C1; a.N = c.N; a.D = c.Di; a.H = c.Hi; a.W = c.Wi; a.Cout = w.cout;
    a.nchunks = w.cin_pad / 16; a.act = c.act; a.slope = c.slope;
    a.total_tiles = (int)tiles;
    a.div_tiles_per_n = make_fastdiv(tx * ty * tz);
    a.order = make_tile_order(tx, ty, tz);
    a.head_w = c.head_w; a.head_b = c.head_b; a.head_out = c.head_out; a.head_ncls = c.head_ncls;
    a.in_scale = c.in_scale; a.in_shift = c.in_shift; a.in_slope = c.in_act == ACT_LRELU ? c.slope : 1.0f;
    float *zeros = nullptr;
    MI355_TRY(device_scratch(SCR_ZEROS, s, 256, (void **)&zeros, true));
    a.zeros = zeros;
    static bool attr_set = false;
    if (!attr_set) {
        MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS_BYTES));
        MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS_BYTES));
        MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS_BYTES));
        MI355_HIP(hipFuncSetAttribute((const void *)conv3_f32_wino3_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)W3_LDS_BYTES));
        attr_set = true;
    }
    const int gy = w.cout / 32;
    int gx = 256 / gy;                      // one persistent workgroup per CU
    gx = gx < 8 ? 8 : (gx / 8) * 8;         // multiple of 8: blockIdx.x & 7 labels the XCD group
    const int need = (int)((tiles + 7) / 8) * 8;
    if (gx > need) gx = need;
    if (c.in_scale) {
        if (kernel_name) *kernel_name = "conv3_f32_wino3_kernel<2, true>";
        hipLaunchKernelGGL((conv3_f32_wino3_kernel<2, true>), dim3(gx, gy), dim3(256), W3_LDS_BYTES, s, a);
    } else if (c.head_out) {
        if (kernel_name) *kernel_name = "conv3_f32_wino3_kernel<1, false>";
        hipLaunchKernelGGL(conv3_f32_wino3_kernel<1>, dim3(gx, gy), dim3(256), W3_LDS_BYTES, s, a);
    } else if (c.stats) {
        if (kernel_name) *kernel_name = "conv3_f32_wino3_kernel<2, false>";
        hipLaunchKernelGGL(conv3_f32_wino3_kernel<2>, dim3(gx, gy), dim3(256), W3_LDS_BYTES, s, a);
    } else {
        if (kernel_name) *kernel_name = "conv3_f32_wino3_kernel<0, false>";
        hipLaunchKernelGGL(conv3_f32_wino3_kernel<0>, dim3(gx, gy), dim3(256), W3_LDS_BYTES, s, a);
    }
    MI355_HIP(hipGetLastError());
    *taken = true;
    return MI355_OK;
}

}  // namespace mi355

// Stride-2 3x3x3 convolution, channel-blocked fp16 storage ([N][C / 8][D][H][W][8], common.h) / fp32 accumulate (round 3): the encoder's down-sampling convs
// (reference model_architecture/generic_UNet.py:285-288,314-315, `first_stride` of StackedConvLayers :128-143) with
// Cout % 128 == 0 on volumes that are whole 4 x 4 x 8 output tiles.
//
// Why a kernel of its own.  A stride-2 brick holds (2T+1)^3 input voxels for T^3 outputs - five times the staging per MFMA
// of a stride-1 brick - and round 2's kernel for these layers (conv3_f16_mfma_pipe_kernel<1,2,..,2>) spent its time there:
// 0.15 of the fp16 MFMA peak, 5.9 VALU instructions per MFMA, and with ONE voxel fragment per wave (32 voxels x 64 couts)
// every pair of MFMAs pulled 2 KB of weights through the L1.  Here:
//   * workgroup tile = 128 outputs (4 x 4 x 8) x 128 couts; the four waves SHARE the voxel fragments (LDS) and split the
//     couts: a wave owns 128 voxels x 32 couts (MF = 4, NF = 1, 64 accumulator registers), so a tap is 4 MFMAs for ONE
//     1-KiB weight fragment from the L1 (0.25 KB per MFMA instead of 1) and four 1-KiB voxel fragments from the LDS;
//   * the brick (9 x 9 x 17 voxels x 16 channels) is double-buffered and filled by LDS-DMA (global_load_lds_dwordx4: no
//     staging registers, no ds_write, ~7 VALU per 1-KiB piece), 13 pieces per wave and chunk, one every second tap;
//   * LDS image of the brick: planar [8-channel half][z][y][x parity][x / 2] with 20-slot rows (9 even + 8 odd x, 3 pad
//     slots).  The fragment of a tap reads voxels 2x + dx for x = 0..7 of four y rows: with the parity split they are
//     CONSECUTIVE 16-B slots, and the 20-slot pitch (y step = 40 slots = 8 mod 16) puts the four rows of a ds_read_b128
//     lane group on four different quarters of the 64 banks: conflict-free (a [voxel] image read with stride 2 is 2- to
//     4-way conflicted, and at one weight fragment per 4 MFMAs the LDS delivers half of what the MFMAs consume);
//   * weights: the (couts 32 w .. 32 w + 31) fragment of the tap in a nine-tap register ring, inline-asm loads with
//     hand-counted s_waitcnt vmcnt (loads return in order; see conv3_f16_dma_kernel for why the compiler cannot do it);
//   * epilogue: bias (the accumulators are initialised with it), LeakyReLU, fp16; a lane holds 4 of a block's 8 couts for its
//     voxel and lane ^ 32 the other 4: v_permlane32_swap pairs them up and a store instruction writes 2 cout blocks x 4 x-rows
//     x 8 voxels: whole 128-B lines of the channel-blocked output (common.h) without a transposition through LDS;
//     sum x and sum x^2 per cout for Instance/GroupNorm as in the other kernels (quantised partials, common.h).
// CW = couts per workgroup: 128 as above, or 64 (round 4: layers with Cout % 128 != 0, e.g. the 32 -> 64 conv of the base model's
// level 1, which used to fall back to the register-staged kernel at 0.17 of the peak): the waves then split 2 x 2 - wave & 1
// picks the 32-cout fragment, wave >> 1 the output z planes {0, 1} or {2, 3} (MF = 2) - over the SAME brick.
// The stride-2 convs read one input tensor (no virtual concat) and never carry the fused head.
// Tried in round 5 and reverted: EIGHT waves per workgroup over the same brick (two per SIMD, 64 voxels x 32 couts each, 7 pieces
// per wave and chunk) so that one wave's DMA issue, chunk barrier and epilogue run beside its partner's MFMAs - the reading of the
// 0.42-busy matrix pipe as issue stalls.  Green on every test, and SLOWER: <true, 128> 751 against 860 TFLOP/s, <false, 64> 392
// against 469 (same bench, boxes 6 % apart on the stride-1 kernels).  The pipe is not waiting for this wave's instruction
// stream: a chunk's 52-KiB brick arrives in ~8 000 cycles whoever asks for it.
#include "kernels.h"

#include <cstdlib>

namespace mi355 {

typedef _Float16 half_t;

struct S2ArgsH {
    const half_t *in;
    const half_t *wp;       // conv_weights_upload_f16 pack with nf = 2: [cout / 64][chunk][tap][2][lane][8 halfs]
    const float *bias;
    half_t *out;
    double *stats;
    int C;                  // input channels (physical, multiple of 16)
    int N, Di, Hi, Wi, Do, Ho, Wo, Cout;
    int nchunks;
    int act;
    float slope;
    int total_tiles;
    FastDiv div_tiles_per_n;
    TileOrder order;
    const void *zeros;      // >= 16 B of zeros in global memory: what out-of-volume and padding slots fetch
};

struct S2GeomH {
    static constexpr int TZ = 4, TY = 4, TX = 8;                       // output tile (z, y, x)
    static constexpr int IZ = 2 * TZ + 1, IY = 2 * TY + 1, IX = 2 * TX + 1;  // brick 9 x 9 x 17
    static constexpr int ROW = 20;                                      // 16-B slots per x row: even x at 0..8, odd x at 9..16
    static constexpr int PLANE_BLOCKS = 26;                             // 1-KiB DMA pieces per 8-channel plane (81 rows x 20 slots = 1620 <= 1664)
    static constexpr int PLANE_BYTES = PLANE_BLOCKS * 1024, BUF_BYTES = 2 * PLANE_BYTES;
    static constexpr int KD = 13;                                       // pieces per wave and chunk: piece d = wave + 4 k of the buffer's 52
    static constexpr int EVERY = 2;                                     // piece k goes out in tap 2 k
    static constexpr int D = 9;                                         // weight ring depth in taps (divides 27)
    static constexpr bool dma_tap(int t) { return t % EVERY == 0 && t / EVERY < KD; }
    // vector-memory operations issued after the weight load of tap t (which went out at the end of tap t - D): one weight
    // load per tap of taps t-8 .. t-1 and the pieces among those taps (taps < 0 are the previous chunk's)
    static constexpr int pending(int t) {
        int n = D - 1;
        for (int j = t - (D - 1); j <= t - 1; ++j) n += dma_tap((j + 27) % 27) ? 1 : 0;
        return n;
    }
    static constexpr int after_last_dma = 27 - EVERY * (KD - 1);        // weight loads issued after the chunk's last piece
    static constexpr int MF_STRIDE = 2 * IY * ROW * 16;                 // fragment mf = output plane z = mf: two input planes further
    static constexpr int BIAS_OFF = 2 * BUF_BYTES;
    static constexpr size_t LDS_BYTES = (size_t)BIAS_OFF + 128 * 4;     // (the epilogue needs no LDS: stores by v_permlane32_swap)
    static_assert(IZ * IY * ROW <= PLANE_BLOCKS * 64, "plane does not hold the brick");
    static_assert(EVERY * (KD - 1) <= 26 && 4 * KD == 2 * PLANE_BLOCKS, "piece schedule");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <bool STATS, int CW = 128>
__global__ __launch_bounds__(256, 1) void conv3_f16_s2dma_kernel(S2ArgsH p) {
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    typedef S2GeomH G;
    static_assert(CW == 128 || CW == 64, "couts per workgroup");
    constexpr int MF = CW == 128 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *bias_lds = (float *)(lds_raw + G::BIAS_OFF);
    const int co_blk = (int)blockIdx.y * CW;
    if (tid < CW) bias_lds[tid] = p.bias[co_blk + tid];  // (published by the prologue's barrier)
    const int wco = CW == 128 ? wave : (wave & 1);          // this wave's 32-cout fragment of the workgroup's couts
    const int wz = CW == 128 ? 0 : 2 * (wave >> 1);         // first output z plane of this wave's fragments

    // this workgroup's tile sequence: XCD group x owns the contiguous range [lo, hi); its workgroups stride through it
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
        int tile_x, tile_y, tile_z;
        tile_from_id(tt, p.order, tile_x, tile_y, tile_z);
        tc.oz0 = tile_z * G::TZ; tc.oy0 = tile_y * G::TY; tc.ox0 = tile_x * G::TX;
        return tc;
    };
    // The brick of tile (oz0, oy0, ox0) spans the input voxels [2 o0 - 1, 2 o0 + 2 T - 1]; the input dims are even (host
    // check: Di = 2 Do), so its high faces always lie inside the volume and only the low face of the first tile of an axis
    // sticks out (bit 0: z, 1: y, 2: x)
    auto tile_faces = [&](const TileCoord &tc) { return (tc.oz0 == 0) | ((tc.oy0 == 0) << 1) | ((tc.ox0 == 0) << 2); };

    // Tile-invariant lane part of this wave's 13 pieces: piece d = wave + 4 k of the buffer, plane d / 26, slots
    // 64 (d % 26) .. + 63 of the plane; slot -> (z, y, x) by rows of 20 slots (even x first).  Packed: voxel offset from the
    // brick origin (24 bits) | low faces the voxel lies on << 24 | bit 27: padding slot.
    unsigned dma_pk[G::KD];
#pragma unroll
    for (int k = 0; k < G::KD; ++k) {
        const int d = wave + 4 * k;
        const int s = (d >= G::PLANE_BLOCKS ? d - G::PLANE_BLOCKS : d) * 64 + lane;
        const int row = s / G::ROW, c = s - row * G::ROW;
        const int bz = row / G::IY, by = row - bz * G::IY;
        const int bx = c < 9 ? 2 * c : 2 * (c - 9) + 1;
        const bool valid = row < G::IZ * G::IY && c < G::IX;
        const int face = (bz == 0) | ((by == 0) << 1) | ((bx == 0) << 2);
        dma_pk[k] = valid ? (unsigned)(((bz * p.Hi + by) * p.Wi + bx) | (face << 24)) : (8u << 24);
    }
    auto dma = [&](const TileCoord &tc, int faces, int ch, auto k_c, char *buf) {
        constexpr int k = decltype(k_c)::value;
        const int d = wave + 4 * k;  // (wave-uniform: scalar arithmetic)
        // wave-uniform part: the brick origin voxel (may lie one voxel outside the tensor), this chunk's 16 channels, and
        // the 8-channel half of the piece's plane
        // (channel-blocked input, common.h: block 2 ch + plane of sample n; 16 B per voxel of a block)
        const long Vi = (long)p.Di * p.Hi * p.Wi;
        const half_t *src = p.in + (((long)tc.n * (p.C >> 3) + 2 * ch + (d >= G::PLANE_BLOCKS ? 1 : 0)) * Vi +
                                    ((long)(2 * tc.oz0 - 1) * p.Hi + (2 * tc.oy0 - 1)) * p.Wi + (2 * tc.ox0 - 1)) * 8;
        unsigned pk = dma_pk[k];
        asm volatile("" : "+v"(pk));
        const bool inside = (pk & ((unsigned)(faces | 8) << 24)) == 0;
        const unsigned off = (pk & 0xffffffu) << 4;  // bytes (< 2^32: host check)
        const char *gin = (const char *)src + off;
        asm volatile("" : "+v"(gin));  // (computed for every lane: a branch around it is a basic-block boundary between the MFMAs)
        const char *g = inside ? gin : (const char *)p.zeros;
        asm volatile("" : "+v"(g));
        char *dst = buf + d * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };

    // LDS byte offset of this lane's voxel fragment 0 in a brick buffer, tap (0, 0, 0): output voxel (y = l31 >> 3, x = l31 & 7)
    // of plane z = mf reads input (2 z + dz, 2 y + dy, 2 x + dx) = row (2 z + dz) * 9 + 2 y + dy, slot x + (dx & 1) * 9 + (dx >> 1)
    const int a_lane = half * G::PLANE_BYTES + ((2 * (l31 >> 3)) * G::ROW + (l31 & 7)) * 16 + wz * G::MF_STRIDE;
    // weights: cout block of 64 = 2 blockIdx.y + (wave >> 1) (CW = 128) or blockIdx.y (CW = 64), fragment nf = wave & 1 of the nf = 2 pack
    const char *wblk = (const char *)(p.wp + (size_t)(CW == 128 ? 2 * blockIdx.y + (wave >> 1) : blockIdx.y) * p.nchunks * (27 * 2 * 512)) + (wave & 1) * 1024;
    const unsigned wlane = lane * 16;
#define S2_WLOAD(DST, SBASE) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(DST) : "v"(wl), "s"(SBASE) : "memory")
#define S2_WWAIT(W, N) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(W) : "n"(N) : "memory")

    TileCoord cur = decode(tile);
    f16x8 wq[G::D];
    {
        const int f0 = tile_faces(cur);
        static_for<0, G::KD>([&](auto k_c) { dma(cur, f0, 0, k_c, lds_raw); });
        static_for<0, G::D>([&](auto t_c) {
            constexpr int t = decltype(t_c)::value;
            const char *wb = wblk + t * 2048;
            const unsigned wl = wlane;  // (asm operands alone do not capture)
            auto &w = wq[t];
            S2_WLOAD(w, wb);
        });
        // (the ring registers are operands of the wait: the compiler must not read or move them before it)
        static_for<0, G::D>([&](auto t_c) { auto &w = wq[decltype(t_c)::value]; S2_WWAIT(w, 0); });
        __syncthreads();
    }

    int buf = 0;
    for (; tile < hi; tile += nl) {
        f32x16 acc[MF];
        {   // the accumulators start from the bias: cout (r & 3) + 8 (r >> 2) + 4 half of this wave's 32
            typedef const __attribute__((address_space(3))) f32x4 lds_cf32x4;
            unsigned bl = (unsigned)(size_t)(const __attribute__((address_space(3))) char *)bias_lds + (wco * 32 + half * 4) * 4;
            asm volatile("" : "+v"(bl));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *(lds_cf32x4 *)(bl + 8 * g * 4);
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[mf][4 * g + k] = b[k];
            }
        }
        const int ntile = tile + nl;
        const TileCoord nxt_tile = ntile < hi ? decode(ntile) : cur;
        for (int ch = 0; ch < p.nchunks; ++ch) {
            const bool last_ch = ch == p.nchunks - 1;
            const bool have_next = !last_ch || ntile < hi;
            const TileCoord nxt = last_ch ? nxt_tile : cur;
            // (without a next chunk the pieces re-stage the current one into the idle buffer: the wait counts stay fixed)
            const int nch_eff = have_next ? (last_ch ? 0 : ch + 1) : ch;
            const int nfaces = tile_faces(nxt);
            const char *bufc = lds_raw + buf * G::BUF_BYTES;
            char *bufn = lds_raw + (buf ^ 1) * G::BUF_BYTES;
            const char *wch = wblk + (size_t)ch * (27 * 2048), *wnx = wblk + (size_t)nch_eff * (27 * 2048);

            typedef const __attribute__((address_space(3))) char lds_cchar;
            typedef const __attribute__((address_space(3))) f16x8 lds_cf16x8;
            unsigned abv = (unsigned)(size_t)(lds_cchar *)bufc + a_lane;
            asm volatile("" : "+v"(abv));
            lds_cchar *ab = (lds_cchar *)abv;
            f16x8 a[2][MF];
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) a[0][mf] = *(lds_cf16x8 *)(ab + mf * G::MF_STRIDE);

            static_for<0, 27>([&](auto tap_c) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int slot = tap % G::D;
                auto &wc = wq[slot];
                S2_WWAIT(wc, G::pending(tap));
                // one MFMA goes out before the tap's memory instructions: hipcc waits for ALL outstanding LDS reads before the
                // first MFMA of a tap (lgkmcnt(0): it does not count past an LDS-DMA), so the reads of tap t+1 are issued behind
                // the first MFMA of tap t and have the other three to land
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[slot], a[tap & 1][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (tap + 1 < 27) {
                    constexpr int nt = tap + 1;
                    constexpr int dz = nt / 9, rr = nt - dz * 9, dy = rr / 3, dx = rr - dy * 3;
                    constexpr int off = ((dz * G::IY + dy) * G::ROW + (dx & 1) * 9 + (dx >> 1)) * 16;
#pragma unroll
                    for (int m = 0; m < MF; ++m) a[(tap + 1) & 1][m] = *(lds_cf16x8 *)(ab + m * G::MF_STRIDE + off);
                }
                if constexpr (G::dma_tap(tap)) dma(nxt, nfaces, nch_eff, std::integral_constant<int, tap / G::EVERY>{}, bufn);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 1; m < MF; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[slot], a[tap & 1][m], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                {
                    constexpr int k = tap + G::D;
                    const char *wb = (k < 27) ? wch + k * 2048 : wnx + (k - 27) * 2048;
                    const unsigned wl = wlane;
                    S2_WLOAD(wc, wb);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // this wave's pieces have landed once at most the weight loads issued after the last one are outstanding; the
            // barrier publishes the brick (a ds_read is ordered behind an LDS-DMA only by the issuer's vmcnt + a barrier)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::after_last_dma) : "memory");
            __builtin_amdgcn_s_barrier();
            buf ^= 1;
        }
        // The ring holds the next tile's first nine taps, still in flight, and the compiler knows nothing of them: retire them
        // before it may move a ring register in the epilogue (see conv3_f16_dma_kernel)
        static_for<0, G::D>([&](auto t_c) { auto &w = wq[decltype(t_c)::value]; S2_WWAIT(w, 0); });
        {
            const float slope = p.act == ACT_LRELU ? p.slope : 1.0f;  // max(x, 1 x) = x
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const bool lrelu = p.act == ACT_LRELU;
            f32x2 s1[8], s2[8];
            if constexpr (STATS) {
#pragma unroll
                for (int r = 0; r < 8; ++r) { s1[r] = f32x2{0.f, 0.f}; s2[r] = f32x2{0.f, 0.f}; }
            }
            // Whole-line stores without a trip through LDS (as conv3_f16_dma_kernel, round 3): fragment mf is output plane z = mf,
            // its lane l31 the voxel (y = l31 >> 3, x = l31 & 7); the lane holds couts 4 half .. + 3 of the wave's four 8-cout
            // blocks.  Per block pair one v_permlane32_swap per dword hands lanes 0-31 the 16 bytes of block gp and lanes 32-63
            // those of block gp + 1: a store instruction writes 4 x-rows x 128 B in each of two block planes, eight whole lines.
            // (lane part rebuilt from a laundered lane id here: hoisted out of the tile loop it would be spilled)
            const size_t Vo = (size_t)p.Do * p.Ho * p.Wo;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const unsigned lane_off = (unsigned)(ln >> 5) * (unsigned)(Vo * 16) + (unsigned)((((ln & 31) >> 3) * p.Wo + (ln & 7)) * 16);  // (< 2^32: host check)
            const half_t *obase = p.out + (((size_t)cur.n * (p.Cout >> 3) + ((co_blk + wco * 32) >> 3)) * Vo + ((size_t)(cur.oz0 + wz) * p.Ho + cur.oy0) * p.Wo + cur.ox0) * 8;
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int gp = 0; gp < 4; gp += 2) {
                    f16x4 val2[2];
#pragma unroll
                    for (int gi = 0; gi < 2; ++gi) {
                        const int g = gp + gi;
#pragma unroll
                        for (int k = 0; k < 4; k += 2) {
                            f32x2 m = {acc[mf][4 * g + k], acc[mf][4 * g + k + 1]};
                            if (lrelu) m = f32x2{fmaxf(m[0], m[0] * slope), fmaxf(m[1], m[1] * slope)};  // (wave-uniform)
                            val2[gi][k] = (half_t)m[0];
                            val2[gi][k + 1] = (half_t)m[1];
                            if constexpr (STATS) {  // v_pk_add_f32 / v_pk_fma_f32: one instruction per value pair
                                s1[2 * g + (k >> 1)] += m;
                                s2[2 * g + (k >> 1)] = __builtin_elementwise_fma(m, m, s2[2 * g + (k >> 1)]);
                            }
                        }
                    }
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    u32x2 a = __builtin_bit_cast(u32x2, val2[0]), b = __builtin_bit_cast(u32x2, val2[1]);
                    // (s_nop 1: two wait states between a VALU write of an operand and the swap / before the store reads the result)
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1"
                                 : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]));
                    const u32x4 v = {a[0], a[1], b[0], b[1]};
                    const char *rowp = (const char *)(obase + ((size_t)mf * p.Ho * p.Wo + (size_t)gp * Vo) * 8);
                    const unsigned lo2 = lane_off;
                    // sc1: nothing on this XCD reads the output again; kept in its L2 the lines would evict brick lines.  s_nop 1: a
                    // VALU write of a 16-byte store's data registers needs a wait state after its issue (the next pair reuses them)
                    asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(lo2), "v"(v), "s"(rowp) : "memory");
                }
            if constexpr (STATS) {
                // transposing reduction over the 32 voxel lanes (common.h): each lane ends with the total of one (cout, statistic)
                // of the tile's 128 voxels and adds it itself; quantised partials: exact, hence order-independent
                const float tot = half32_reduce_scatter(s1, s2, ln);
                const int r = stat_slot_r(ln), k = (ln >> 4) & 1;
                const int c = 8 * (r >> 2) + 4 * (ln >> 5) + (r & 3);
                atomicAdd(p.stats + ((size_t)cur.n * p.Cout + co_blk + wco * 32 + c) * 2 + k, quantise_partial((double)tot, k, (long)p.Do * p.Ho * p.Wo));
            }
        }
        cur = nxt_tile;
    }
#undef S2_WLOAD
#undef S2_WWAIT
}

// Launches the kernel above when the call fits it; *taken says whether it did (the caller falls back to the older kernels).
int conv3d_f16_s2dma(const ConvWeightsH &w, const ConvCallH &c, hipStream_t s, const char **kernel_name, bool *taken) {
    typedef S2GeomH G;
    *taken = false;
    static int on = -1;
    if (on < 0) { const char *e = getenv("MI355_F16_S2"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || w.stride != 2 || w.nf != 2 || w.cout % 64 != 0 || c.head_out || c.C1 != 0 || c.in_scale || c.C0 != w.cin_pad || c.C0 % 16 != 0) return MI355_OK;
    if ((c.Di | c.Hi | c.Wi) & 1) return MI355_OK;
    const int Do = c.Di / 2, Ho = c.Hi / 2, Wo = c.Wi / 2;
    if (Do % G::TZ || Ho % G::TY || Wo % G::TX) return MI355_OK;
    const int tiles_x = Wo / G::TX, tiles_y = Ho / G::TY, tiles_z = Do / G::TZ;
    const long tiles = (long)tiles_x * tiles_y * tiles_z * c.N;
    const int cw = w.cout % 128 == 0 ? 128 : 64;
    const int gy = w.cout / cw;
    if (tiles * gy < 256 || tiles >= (1l << 30)) return MI355_OK;
    if ((long)G::IZ * c.Hi * c.Wi >= (1l << 24) || (long)G::IZ * c.Hi * c.Wi * 16 >= (1l << 32) || (long)Do * Ho * Wo * 64 >= (1l << 32)) return MI355_OK;
    S2ArgsH a;
    a.in = c.in0; a.wp = w.wp_dev; a.bias = w.bias_dev; a.out = c.out; a.stats = c.stats;
    a.C = c.C0; a.N = c.N; a.Di = c.Di; a.Hi = c.Hi; a.Wi = c.Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo; a.Cout = w.cout;
    a.nchunks = w.cin_pad / 16; a.act = c.act; a.slope = c.slope;
    a.total_tiles = (int)tiles;
    a.div_tiles_per_n = make_fastdiv(tiles_x * tiles_y * tiles_z);
    a.order = make_tile_order(tiles_x, tiles_y, tiles_z);
    void *zeros = nullptr;
    MI355_TRY(device_scratch(SCR_ZEROS, s, 256, &zeros, true));
    a.zeros = zeros;
    int gx = 256 / gy;
    gx = gx < 8 ? 8 : (gx / 8) * 8;
    const int need = (int)((tiles + 7) / 8) * 8;
    if (gx > need) gx = need;
    static bool attr_set[4] = {false, false, false, false};
    auto launch = [&](auto kern, int idx) -> int {
        if (!attr_set[idx]) {
            MI355_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES));
            attr_set[idx] = true;
        }
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(256), G::LDS_BYTES, s, a);
        MI355_HIP(hipGetLastError());
        return MI355_OK;
    };
    *taken = true;
    if (cw == 64) {
        if (kernel_name) *kernel_name = c.stats ? "conv3_f16_s2dma_kernel<true, 64>" : "conv3_f16_s2dma_kernel<false, 64>";
        if (c.stats) return launch(conv3_f16_s2dma_kernel<true, 64>, 2);
        return launch(conv3_f16_s2dma_kernel<false, 64>, 3);
    }
    if (kernel_name) *kernel_name = c.stats ? "conv3_f16_s2dma_kernel<true, 128>" : "conv3_f16_s2dma_kernel<false, 128>";
    if (c.stats) return launch(conv3_f16_s2dma_kernel<true, 128>, 0);
    return launch(conv3_f16_s2dma_kernel<false, 128>, 1);
}

}  // namespace mi355

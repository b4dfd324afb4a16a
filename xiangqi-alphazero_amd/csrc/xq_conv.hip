// xq_conv.hip -- 3x3 convolution of the residual tower (model.py:20-36) as a fused Winograd F(2x3, 3x3) kernel
// on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Why Winograd: the contract is fp32 (1e-5 against the fp32 reference; gfx950 has no xf32), and the direct
// implicit-GEMM form is already at ~88 % of the 157 TFLOP/s fp32 MFMA peak in the ROCm library -- the only way past that
// roofline is fewer multiplies.  The board is 10 rows x 9 columns: F(2,3) along the rows (5 tiles, 4 frequencies) and
// F(3,3) along the columns (3 tiles, 5 frequencies; points 0, +-1, 2, inf) cover it exactly -- 15 tiles x 20 frequencies
// = 300 multiplies per board and channel pair instead of 810: 2.7x fewer MFMA flops than the direct form, all
// arithmetic still fp32 (a float32 emulation of the 256x10 tower differs from the float64 tower by 7e-7 relative, the
// direct fp32 convolution by 3e-7).  The transformed input (3.3x the activation bytes) and the 20 per-frequency
// products never leave the CU -- an unfused Winograd would be HBM-bound and lose the gain.
//
// Transforms (the column direction's B^T rows are scaled to small integers, the inverse scales 1/2, 1/2, 1/6, 1/6, 1
// are folded into the host-side G):
//   w_c = d[r1][c] + s d[r2][c]   row p of the F(2,3) B^T d: (r1, r2, s) = (0,2,-), (1,2,+), (1,2,-), (1,3,-); row 2 is the
//                                 NEGATIVE of the textbook d2 - d1, the weights of its frequencies carry the other sign
//   t = w3 - w1;  v0 = 2 (w0 - w2) + t;  v1 = 2 w1 + w2 - w3;  v2 = 3 w2 - (2 w1 + w3);  v3 = t;  v4 = (w4 - w2) - 2 t
//   Y[a][b] = sum_p ATr[a][p] sum_j ATc[b][j] M[p][j],  ATr = [[1,1,1,0],[0,1,-1,-1]],
//                                                      ATc = [[1,1,1,1,0],[0,1,-1,2,0],[0,1,1,4,1]]
//
// Layouts (C = channels in = channels out, C % 64 == 0):
//   X, Y, R : float[B][90][C]   (NHWC, position-major)                       activations / residual
//   Ug      : float[C/64][C/8][20][2][64][4] = U[cog][chunk][xi = 5 p + j][quad][co][k], U_xi = s_p (G_r g G_c'^T)[p][j]
//             for input channel 8*chunk + 4*quad + k, s_2 = -1                 pre-transformed weights (host)
// Work decomposition: workgroup = 32 tiles (2.13 boards) x 64 output channels, 4 waves, TWO workgroups per CU.
// Wave p owns Winograd row p (frequencies 5p..5p+4) for all 64 channels: 5 xi x 1 M-tile x 2 N-tiles = 10 accumulator
// tiles of 32x32 = 160 VGPRs.  The MFMA A operand of lane (h, m) -- V[xi][tile m][ci = 4h..4h+3] -- is exactly what the
// input transform of (tile m, channel quad h, row p) produces, so every lane transforms what it multiplies: the
// transformed input never goes through LDS, and the only shared data is the raw input of the <= 4 boards a tile group
// touches, staged 8 channels at a time (double-buffered, one barrier per 8 channels).  Every weight element is used by
// exactly one wave: the B operand goes global (L2) -> registers through buffer loads (wave-uniform descriptor + scalar
// offset, one address VGPR) into a pool of FIVE fragment registers -- each fragment is fetched half a chunk (20 MFMAs)
// ahead of its use, which is what lets 160 accumulators, 20 A and 20 B registers fit under the 256 a wave may hold.
// The two workgroups of a CU are not synchronised with each other: one's prologue, barrier waits and epilogue are
// covered by the other's MFMAs.
// Epilogue: the column half of A^T M A in registers, the row half across the 4 waves through LDS (one 32-channel half
// at a time over the staging buffers), then bias + residual + ReLU and NHWC stores.
// Blocks are dealt so that each XCD works on one 64-channel slice of U at a time (1.3 MB at C=256: L2-resident).
// The WIDE variant (NT = 4, one workgroup per CU, 128 output channels, 320 accumulators in VGPRs + AGPRs) differs in three
// places, each explained where it happens: 16 weight-fragment registers (a fragment is fetched 64 MFMAs ahead), the input
// loads of the prologue before everything else, and an epilogue pipelined against its own global stores.  Cost model used
// throughout (measured, tests/microbench/valu_rates.hip): the fp32 MFMA runs on the vector ALU's multipliers, so every
// vector instruction beside it costs its ~5 issue cycles; an LDS read ~6; a store or load issued into a full queue blocks
// the wave (~260 / ~130 cycles each when all CUs do it), a spaced one costs its slot.
#include <type_traits>

#include "xq_common.h"

#pragma clang fp contract(off)

// In-situ ablation builds (tests/microbench/wino_ablate.sh; never in the shipped library): bit 1 no weight loads in the
// main loop, 2 no input transform (and no LDS reads of the raw input), 4 LDS reads kept but the transform's arithmetic
// dropped, 8 no staging of the next chunks (global -> LDS), 16 no main-loop barriers, 32 no epilogue, 64 no MFMAs,
// 128 staging loads from one contiguous run (coalesced), 256 staging loads from an L2-resident region, 512 staging issued
// at the start of a chunk instead of its middle, 1024 weight loads always from chunk 0 (L1/L2-hot), 2048 two weight slices
// per XCD, 4096 / 8192 weight loads sc1 (L1 bypass) / nt, 16384 output stores nt, 32768 residual loads nt, 65536 one
// workgroup per CU (narrow variant), 131072 staging loads nt.
// Results are wrong by construction; only the launch time is read.
#ifndef XQ_ABL
#define XQ_ABL 0
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC = 8;
constexpr int TILES = 32;                           // tiles per workgroup (2.13 boards)
// Staged raw input of one 8-channel chunk: two planes (channel quad h = 0, 1) of 16-byte units, one unit per halo'd
// board position.  Unit of position (board b, halo'd row y' = y + 1 in 0..11, halo'd column x' = x + 1 in 0..10):
//     b * XU_B + (y' >> 1) * XU_PAIR + (y' & 1) * XU_ODD + x'
// Even and odd rows are split so that a tile's base unit b*XU_B + ty*XU_PAIR + 3 tx is == 3 * (global tile index) mod 16
// (XU_PAIR == 9, XU_B == 13 mod 16): the 16 lanes that one ds_read_b128 services together -- 16 tiles with distinct
// indices mod 16 -- then fall on 16 distinct bank quads, and every patch element is the base plus an immediate, so the
// raw-input reads are bank-conflict free (the round-1 layout, 48-byte positions row-major, was 2.6-way conflicted).
// The right halo of a row is the left halo of the row that follows it in memory (both zero).
constexpr int XSTRIDE = 16;                          // bytes per unit
constexpr int XU_ODD = 10, XU_PAIR = 25, XU_B = 157;
constexpr int XPLANE = (4 * XU_B) * 16;             // 4 boards per plane
constexpr int XDUMP = 4 * XU_B - 1;                 // a unit no tile reads: target of out-of-range staging lanes
constexpr int XRAW = 2 * XPLANE;
// Row stride of an exchange plane: the epilogue's reader (tile = tid >> 3, channel quad = tid & 7) issues ds_read_b128 in the
// lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: four tiles x a 16-float window each.  With a stride == 32 mod 64
// floats the four windows fall on four disjoint quarters of the 64 banks (round 2's 36 / 68 gave 2-3-way conflicts:
// SQ_LDS_BANK_CONFLICT 1.7e7 per launch).  The writes are ds_write_b32 of 32 consecutive floats: any stride.
#ifndef XQ_EPAD_WIDE
#define XQ_EPAD_WIDE 32
#endif
#ifndef XQ_EPAD_NARROW
#define XQ_EPAD_NARROW 0
#endif
constexpr int ESTR = 32 + XQ_EPAD_NARROW;                             // floats per tile row of an exchange plane (32 + 4: the two
                                                     // lane halves of an accumulator write land on different banks)
constexpr int E_BYTES = 4 * 3 * TILES * ESTR * 4;  // epilogue exchange for one 32-channel half: [row p][b][tile][co]
constexpr int LDS_BYTES = 2 * XRAW > E_BYTES ? 2 * XRAW : E_BYTES;
constexpr int LDS_BYTES_WIDE = 4 * 3 * TILES * (64 + XQ_EPAD_WIDE) * 4;             // exchange planes of a 64-channel round

__device__ __forceinline__ f32x4 ld4(const char *p) { return *(const f32x4 *)p; }
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// 128-bit store for code that continues with INLINE-ASM vector instructions.  gfx950 hazard: a VALU write to the data VGPRs of a
// store wider than 64 bits needs 2 wait states after the store; hipcc's hazard recogniser inserts them for instructions it can
// see, but an `asm` statement is opaque to it -- a v_pk_add_f32 from asm that recycled the store's registers right behind it
// corrupted lanes 12-15 of every row of 16 (the part of the data the store path reads last; round 3, found by bisecting two
// epilogue re-orderings that only changed register allocation).  The s_nop is pinned behind the store by the memory clobber.
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
    asm volatile("s_nop 1" ::: "memory");
}
// cache-policy variants (aux: bit 0 sc0, bit 1 nt, bit 4 sc1): weight fragments are read once per workgroup and never from
// this CU's L1 again -- XQ_W_AUX selects how they pass through the caches
#ifndef XQ_W_AUX
#define XQ_W_AUX ((XQ_ABL & 4096) ? 16 : (XQ_ABL & 8192) ? 2 : 0)
#endif
__device__ __forceinline__ f32x4 buf_ld4_w(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, XQ_W_AUX));
}
__device__ __forceinline__ void st4_y(float *p, f32x4 v) {
    if (XQ_ABL & 16384) __builtin_nontemporal_store(v, (f32x4 *)p); else *(f32x4 *)p = v;
    asm volatile("s_nop 1" ::: "memory");          // store-data hazard against the inline-asm VALU that follows (see buf_st4)
}
__device__ __forceinline__ f32x4 ld4_r(const float *p) {
    if (XQ_ABL & 32768) return __builtin_nontemporal_load((const f32x4 *)p);
    return *(const f32x4 *)p;
}
// packed fp32 helpers on four floats (two instructions each); c is a {k, k} pair in SGPRs or VGPRs
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f32x2{a.x, a.y}), "v"(f32x2{b.x, b.y}));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f32x2{a.z, a.w}), "v"(f32x2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a.x, a.y}), "v"(f32x2{b.x, b.y}));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a.z, a.w}), "v"(f32x2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ float relu1(float x) {                // one v_max_f32 (fmaxf adds a canonicalising max)
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ float max1(float lo, float x) {      // one v_max_f32 against a wave-uniform bound
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(lo), "v"(x));
    return r;
}
__device__ __forceinline__ f32x2 pk_add2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_sub2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 pk_fma2(f32x2 x, f32x2 c, f32x2 y) {
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(c), "v"(y));
    return r;
}
// x * c + y  /  x * c - y
__device__ __forceinline__ f32x4 pk_fma4(f32x4 x, f32x2 c, f32x4 y) {
    f32x2 lo, hi;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(f32x2{x.x, x.y}), "v"(c), "v"(f32x2{y.x, y.y}));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(f32x2{x.z, x.w}), "v"(c), "v"(f32x2{y.z, y.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 pk_fms4(f32x4 x, f32x2 c, f32x4 y) {
    f32x2 lo, hi;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(lo) : "v"(f32x2{x.x, x.y}), "v"(c), "v"(f32x2{y.x, y.y}));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(hi) : "v"(f32x2{x.z, x.w}), "v"(c), "v"(f32x2{y.z, y.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

// NT = N-tiles (32 output channels each) per wave.  NT = 2: 64 output channels per workgroup, 160 accumulators per wave,
// two workgroups per CU (the round-1 shape).  NT = 4 ("wide"): 128 output channels per workgroup, 320 accumulators per
// wave (VGPRs + AGPRs of the unified 512-entry file), ONE workgroup per CU: every staged input chunk and every transformed
// A operand feeds twice as many MFMAs, and the input is fetched by half as many workgroups.
template <int NT>
__global__ __launch_bounds__(256, NT == 2 ? 2 : 1) void k_wino_conv(const float *__restrict__ X, const float *__restrict__ Ug,
                                                   const float *__restrict__ bias, const float *__restrict__ R,
                                                   float *__restrict__ Y, int B, int C, int flags, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Xr = lds;
    constexpr int NCO = 32 * NT;                      // output channels per workgroup
    constexpr int UBUF_BYTES = 20 * 2 * NCO * 16;    // one 8-channel chunk of weights for them
    // Weight fragments per chunk, and fragment registers.  Narrow: 5 registers, each fragment fetched half a chunk (20 MFMAs)
    // ahead.  Wide: XQ_WIDE_POOL registers (default 16 of a chunk's 20): a fragment is fetched 16 fragments = 64 MFMAs = 4096
    // cycles ahead of its use.  Loads retire in order (vmcnt), so the wait for a weight fragment also waits for every OLDER
    // load -- including the staging loads of the raw input, a quarter of which miss to HBM (one 128-byte line serves four
    // 8-channel chunks); with 10 registers they had 2560 cycles to land (round 3's in-situ ablation: staging 0.108 ms of 2.60,
    // and the weight loads cost nothing once the staging loads are gone).  16 registers: 2.528 against 2.551 ms; all 20: 2.563
    // (profiles/r03_wino_ab_weight_pool_10_16_20.log).  The slot of fragment f of chunk c is (20 c + f) mod POOL, so the code
    // repeats every PER = lcm(20, POOL) / 20 chunks.
#ifndef XQ_WIDE_POOL
#define XQ_WIDE_POOL 16
#endif
    constexpr int NF = 5 * NT, POOL = NT == 4 ? XQ_WIDE_POOL : NF / 2;   // (narrow: 8 or 10 registers spill 124 / 32 VGPRs)
    constexpr int PER = NT == 4 ? (XQ_WIDE_POOL == 16 ? 4 : XQ_WIDE_POOL == 10 || XQ_WIDE_POOL == 20 ? 1 : -1) : 1;
    static_assert(PER > 0, "XQ_WIDE_POOL must be 10, 16 or 20");

    const int tid = threadIdx.x, lane = tid & 63, wp = tid >> 6;
    const int NG = C / NCO;
    const int per = 8 / NG;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    // ablation 2048 (C = 256): two weight slices per XCD, the two blocks that share a tile group back to back on it
    const int cog = (XQ_ABL & 2048) ? 2 * (xcd & 1) + (rr & 1) : xcd % NG;
    // flags bit 1: walk the batch back to front.  A launch that reads what the previous launch wrote (the next layer of
    // the tower) then starts with the boards written last -- still in the 256 MB Infinity Cache -- instead of the oldest.
    const int tg_fwd = (XQ_ABL & 2048) ? (rr >> 1) * 4 + (xcd >> 1) : rr * per + xcd / NG;
    const int tg = (flags & 2) ? n_groups - 1 - tg_fwd : tg_fwd;
    if (tg_fwd >= n_groups) return;
    const int relu = flags & 1;
    const int T = B * 15;
    const int t0 = tg * TILES;
    const int b_lo = t0 / 15;
    const int NCH = C / KC;
    const int h = lane >> 5, l31 = lane & 31;

    // transform / MFMA role: tile l31, channel quad h, Winograd row wp
    const int gt = t0 + l31 < T ? t0 + l31 : T - 1;
    const int tb = gt / 15, tt = gt - tb * 15, ty = tt / 3, tx = tt - ty * 3;
    const int tbase = ((tb - b_lo) * XU_B + ty * XU_PAIR + 3 * tx) * XSTRIDE + h * XPLANE;   // P(tb, 2ty-1, 3tx-1)
    const int tb1 = tbase + (wp == 0 ? 0 : XU_ODD) * XSTRIDE;                                 // patch row 0 or 1
    const int tb2 = tbase + (wp == 3 ? XU_PAIR + XU_ODD : XU_PAIR) * XSTRIDE;                 // patch row 2 or 3
    const float sg = wp == 1 ? 1.0f : -1.0f;
    const f32x2 sgn = {sg, sg};
    const f32x2 two = {2.0f, 2.0f}, three = {3.0f, 3.0f}, mtwo = {-2.0f, -2.0f}, four = {4.0f, 4.0f};

    // staging role: a contiguous run of positions (first tile's halo row .. last tile's), 2 x 16 B per position and chunk
    const int tl = (t0 + TILES - 1 < T ? t0 + TILES - 1 : T - 1);
    const int b_hi = tl / 15;
    const int y_min = 2 * ((t0 - b_lo * 15) / 3) - 1, y_max = 2 * ((tl - b_hi * 15) / 3) + 2;
    const int pos_first = (y_min > 0 ? y_min : 0) * 9;
    const int pos_last = (b_hi - b_lo) * 90 + ((y_max < 9 ? y_max : 9) + 1) * 9 - 1;
    const int spos = pos_first + (tid >> 1), spart = tid & 1;
    const unsigned xgo = (unsigned)(((long long)b_lo * 90 + spos) * C + spart * 4) * 4u;
    const unsigned xstep = 128u * (unsigned)C * 4u;
    unsigned xgk[2];
    int xl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int pos = spos + 128 * k;
        const int bi = pos / 90, rem = pos - bi * 90, y = rem / 9, x = rem - y * 9;
        const bool ok = pos <= pos_last;
        xgk[k] = ok ? xgo + k * xstep : 0xFFFFFFF0u;
        xl[k] = (ok ? bi * XU_B + ((y + 1) >> 1) * XU_PAIR + ((y + 1) & 1) * XU_ODD + x + 1 : XDUMP) * XSTRIDE + spart * XPLANE;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, (int)((unsigned)B * 90u * (unsigned)C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)(Ug + (size_t)cog * NCH * (UBUF_BYTES / 4)), 0,
                                                                         NCH * UBUF_BYTES, 0x00020000);
    // B fragment (q, nt): lane (h, n) needs U[5 wp + q][8 chunk + 4h + j][64 cog + 32 nt + n], j = 0..3
    const unsigned ul = ((wp * 5 * 2 + h) * NCO + l31) * 16;

    f32x16 acc[5][NT];                                // first written by chunk 0 (its MFMAs start from a zero C operand)

    f32x4 xreg[2];
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (XQ_ABL & 128)            // ablation: same bytes per chunk, but one contiguous 8 KB run per workgroup (coalesced)
                xreg[k] = buf_ld4(xrs, (unsigned)b_lo * 90u * (unsigned)C * 4u + (unsigned)(tid + 256 * k) * 16u, chunk * 8192);
            else if (XQ_ABL & 256)       // ablation: every workgroup reads the same 256 KB (L2-resident): latency without HBM
                xreg[k] = buf_ld4(xrs, (unsigned)(tid + 256 * k) * 16u, chunk * 8192);
            else if (XQ_ABL & 131072)    // ablation: staging loads nt (streaming hint)
                xreg[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xgk[k], chunk * 32, 2));
            else
                xreg[k] = buf_ld4(xrs, xgk[k], chunk * 32);
        }
    };
    auto store_x = [&](int chunk) __attribute__((always_inline)) {
        char *dst = Xr + (chunk & 1) * XRAW;
#pragma unroll
        for (int k = 0; k < 2; ++k) *(f32x4 *)(dst + xl[k]) = xreg[k];
    };
    auto rowpair = [&](f32x4 d1, f32x4 d2) __attribute__((always_inline)) { return pk_fma4(d2, sgn, d1); };

    f32x4 a[5], ub[POOL];
    auto loop_barrier = [&]() __attribute__((always_inline)) { if (!(XQ_ABL & 16)) __syncthreads(); };
    // weight fragment f of a chunk: (q, nt) = (QO[f >> 1], f & 1), processing order of the column frequencies 1,2,3,0,4
    auto load_frag = [&](int chunk, int f, int slot) __attribute__((always_inline)) {
        const int q = (f / NT) == 0 ? 1 : (f / NT) == 1 ? 2 : (f / NT) == 2 ? 3 : (f / NT) == 3 ? 0 : 4;
        ub[slot] = buf_ld4_w(urs, ul, ((XQ_ABL & 1024) ? 0u : (unsigned)chunk * UBUF_BYTES) + q * (2 * NCO * 16) + (f % NT) * (32 * 16));
    };
    auto transform0 = [&]() __attribute__((always_inline)) {
        f32x4 w[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) w[c] = rowpair(ld4(Xr + tb1 + c * XSTRIDE), ld4(Xr + tb2 + c * XSTRIDE));
        const f32x4 t = pk_sub4(w[3], w[1]);
        a[0] = pk_fma4(pk_sub4(w[0], w[2]), two, t);
        a[1] = pk_add4(pk_fms4(w[1], two, w[3]), w[2]);
        a[2] = pk_fms4(w[2], three, pk_fma4(w[1], two, w[3]));
        a[3] = t;
        a[4] = pk_fma4(t, mtwo, pk_sub4(w[4], w[2]));
    };
    // One chunk: 40 MFMAs (10 weight fragments x 4 k-steps); the transform of the next chunk is threaded through.
    auto chunk_body = [&](int nchunk_u, int lchunk, auto xo_tag, int stage, auto first_tag, auto ph_tag) __attribute__((always_inline)) {
        constexpr int XO = decltype(xo_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int PH = decltype(ph_tag)::value;        // chunk index mod PER
        f32x4 d1, d2, w1, w3, w2, w0, t, e, fm, v0;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int g = f / NT, nt = f % NT;
            const int q = g == 0 ? 1 : g == 1 ? 2 : g == 2 ? 3 : g == 3 ? 0 : 4;
            const int col = g == 0 ? 1 : g == 1 ? 3 : g == 2 ? 2 : g == 3 ? 0 : 4;
            const int slot = (NF * PH + f) % POOL;
            if (nt == 0 && !(XQ_ABL & 2)) { d1 = ld4(Xr + tb1 + XO + col * XSTRIDE); d2 = ld4(Xr + tb2 + XO + col * XSTRIDE); }
            if (!(XQ_ABL & 6)) {
                if (f == 3 * NT) a[3] = t;                            // column frequency 3 retired with the last fragment of group 2
                if (f == 4 * NT) a[0] = v0;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (XQ_ABL & 64) {
                    if (FIRST && jj == 0) for (int e = 0; e < 16; ++e) acc[q][nt][e] = 0.0f;
                    asm volatile("" ::"v"(a[q][jj]), "v"(ub[slot][jj]));
                } else if (FIRST && jj == 0) {
                    const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    acc[q][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][jj], ub[slot][jj], zero, 0, 0, 0);
                } else if (NT == 4 && nt == 3) {
                    // wide variant: 20 accumulator tiles do not fit the 256 AGPRs; the compiler would spill four of them
                    // around the loop.  The five tiles of the last N-tile are pinned to VGPRs ("+v"), the other 15 stay in
                    // AGPRs.  (A dependent MFMA on its own vDst/SrcC needs no software wait states.)
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[q][nt]) : "v"(a[q][jj]), "v"(ub[slot][jj]));
                } else {
                    acc[q][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][jj], ub[slot][jj], acc[q][nt], 0, 0, 0);
                }
                if (jj == 1) __builtin_amdgcn_sched_barrier(0);
            }
            // weights: the fragment POOL positions further down the stream replaces this one (later in this chunk or in the next)
            if (!(XQ_ABL & 1)) { if (f + POOL < NF) load_frag(lchunk, f + POOL, slot); else load_frag(nchunk_u, f + POOL - NF, slot); }
            if (nt == NT - 1 && (XQ_ABL & 4)) {                        // ablation: reads stay live, no arithmetic
                asm volatile("" ::"v"(d1), "v"(d2));
                if (g == 3 && stage >= 0 && !(XQ_ABL & 8)) { store_x(stage); load_x(stage + 1); }
            }
            if (nt == NT - 1 && (XQ_ABL & 2) && g == 3 && stage >= 0 && !(XQ_ABL & 8)) { store_x(stage); load_x(stage + 1); }
            if (nt == NT - 1 && !(XQ_ABL & 6)) {
                if (g == 0) {
                    w1 = rowpair(d1, d2);
                    if ((XQ_ABL & 512) && stage >= 0) { store_x(stage); load_x(stage + 1); }
                }
                if (g == 1) { w3 = rowpair(d1, d2); t = pk_sub4(w3, w1); e = pk_fma4(w1, two, w3); fm = pk_fms4(w1, two, w3); }
                if (g == 2) { w2 = rowpair(d1, d2); a[1] = pk_add4(fm, w2); a[2] = pk_fms4(w2, three, e); }
                if (g == 3) {
                    w0 = rowpair(d1, d2); v0 = pk_fma4(pk_sub4(w0, w2), two, t);
                    if (stage >= 0 && !(XQ_ABL & (8 | 512))) { store_x(stage); load_x(stage + 1); }
                }
                if (g == 4) { const f32x4 w4 = rowpair(d1, d2); a[4] = pk_fma4(t, mtwo, pk_sub4(w4, w2)); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue ------------------------------------------------------------------------------------------
    // the input of chunks 0 and 1 first (one HBM latency, not two; it is what the first MFMA waits for), then the weights, then
    // the zero fill of the staging buffers while both are in flight
    f32x4 x1[2];
    load_x(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) x1[k] = buf_ld4(xrs, xgk[k], 32);
#pragma unroll
    for (int f = 0; f < POOL; ++f) { if (f < NF) load_frag(0, f, f); else load_frag(1, f - NF, f); }
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int o = tid * 16; o < 2 * XRAW; o += 256 * 16) *(f32x4 *)(Xr + o) = z;
    }
    __syncthreads();                                  // zero fill before the first stores
    store_x(0);
#pragma unroll
    for (int k = 0; k < 2; ++k) *(f32x4 *)(Xr + XRAW + xl[k]) = x1[k];
    load_x(2);
    __syncthreads();
    transform0();
    __syncthreads();                                  // chunk 0 stores into the buffer transform0 just read

    // ---- main loop: chunk c multiplies (a, fragments) of chunk c, transforms chunk c+1 out of buffer (c+1)&1,
    // stores chunk c+2 into buffer c&1 and fetches chunk c+3; one barrier per chunk
    using XA = std::integral_constant<int, XRAW>;
    using XB = std::integral_constant<int, 0>;
    chunk_body(1, 0, XA{}, 2, std::true_type{}, std::integral_constant<int, 0>{});
    loop_barrier();
    chunk_body(2, 1, XB{}, 3, std::false_type{}, std::integral_constant<int, 1 % PER>{});
    loop_barrier();
    if constexpr (PER == 4) {
        chunk_body(3, 2, XA{}, 4, std::false_type{}, std::integral_constant<int, 2>{});
        loop_barrier();
        chunk_body(4, 3, XB{}, 5, std::false_type{}, std::integral_constant<int, 3>{});
        loop_barrier();
        for (int c = 4; c < NCH; c += 4) {
            chunk_body(c + 1, c, XA{}, c + 2, std::false_type{}, std::integral_constant<int, 0>{});
            loop_barrier();
            chunk_body(c + 2, c + 1, XB{}, c + 3, std::false_type{}, std::integral_constant<int, 1>{});
            loop_barrier();
            chunk_body(c + 3, c + 2, XA{}, c + 4, std::false_type{}, std::integral_constant<int, 2>{});
            loop_barrier();
            chunk_body(c + 4 < NCH ? c + 4 : c + 3, c + 3, XB{}, c + 5, std::false_type{}, std::integral_constant<int, 3>{});
            loop_barrier();
        }
    } else {
        for (int c = 2; c < NCH; c += 2) {
            chunk_body(c + 1, c, XA{}, c + 2, std::false_type{}, std::integral_constant<int, 0>{});
            loop_barrier();
            chunk_body(c + 2 < NCH ? c + 2 : c + 1, c + 1, XB{}, c + 3, std::false_type{}, std::integral_constant<int, 0>{});
            loop_barrier();
        }
    }
    if (XQ_ABL & 32) {                                 // ablation: no epilogue (keep the accumulators observable)
        float sacc = 0.0f;
        for (int q = 0; q < 5; ++q) for (int n = 0; n < NT; ++n) for (int e = 0; e < 16; ++e) sacc += acc[q][n][e];
        if (sacc == 1234.5f) Y[tid] = sacc;
        return;
    }

    if constexpr (NT == 4) {
        // The "+v" MFMAs of N-tile 3 are invisible to the compiler's hazard recogniser: hold the first VALU read of the last
        // one's destination for its 16 passes (the other four tiles' last MFMAs are >= 256 cycles older).  Costs 64 cycles
        // per workgroup; the narrow == wide bit-identity test (tests/test_nn_parity.py) stays the functional guard.
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" : "+v"(acc[4][3]));
    }
#ifndef XQ_EPI_PIPE
#define XQ_EPI_PIPE 1
#endif
    if constexpr (NT == 4 && XQ_EPI_PIPE) {
        // ---- wide epilogue, software-pipelined against its own stores ------------------------------------------------------
        // Same exchange as below (two rounds of 64 channels through the planes, thread (tile, channel quad) finishes six
        // pixels), re-ordered around one measured fact: a 1-KB store instruction takes the CU's store path ~260 cycles with four
        // waves storing, and a wave whose store finds the queue full issues NOTHING meanwhile -- 24 stores back to back are
        // ~3 us of a 6.6 us epilogue (profiles/r03_instruction_costs_one_wave_per_simd.log).  So the twelve output vectors of
        // round 0 are kept in registers and stored one by one between the segments of round 1's column transform (16 segments
        // of ~20 vector instructions), round 1's are stored as each is finished, and the next round's residual loads are
        // issued one per finished vector.  Scheduling fences keep the compiler from re-clustering them.  No divergent branch:
        // lanes of tiles past the end carry an out-of-range buffer offset (dropped by the hardware), a null residual is a
        // descriptor of zero records (loads return 0.0f), ReLU-or-identity is one v_max_f32 against 0 / -inf.
        const int etile = tid >> 3, co = (tid & 7) * 4;
        const int eg = t0 + etile, egc = eg < T ? eg : T - 1;
        const int ebd = egc / 15, et2 = egc - ebd * 15, ety = et2 / 3, etx = et2 - ety * 3;
        const int Cs = __builtin_amdgcn_readfirstlane(C);
        const unsigned ooff_in = (unsigned)(((ebd * 90 + 2 * ety * 9 + 3 * etx) * Cs + cog * NCO + co) * 4);
        const unsigned ooff = eg < T ? ooff_in : 0xFFFFFFF0u;
        const unsigned nbytes = (unsigned)B * 90u * (unsigned)C * 4u;
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)Y, 0, (int)nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(R != nullptr ? R : X), 0, R != nullptr ? (int)nbytes : 0, 0x00020000);
        const float rlo = relu ? 0.0f : -__builtin_inff();
        constexpr int ESTR_R = 64 + XQ_EPAD_WIDE;
        auto goff = [&](int n, int it) __attribute__((always_inline)) { return (unsigned)(32 * n + ((it / 3) * 9 + it % 3) * Cs) * 4u; };
#ifndef XQ_RES_SPREAD
#define XQ_RES_SPREAD 2       // round 0's residual loads go out this many per column-transform segment (0: all twelve up front, -0.5 %: the issue of twelve back-to-back loads blocks the wave like stores do)
#endif
        f32x4 resv[2][6], bv[4], pend[12];
        if (!XQ_RES_SPREAD) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int it = 0; it < 6; ++it) resv[s][it] = buf_ld4(rrs, ooff, goff(s, it));
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) bv[n] = *(const f32x4 *)(bias + cog * NCO + 32 * n + co);
        float *E = (float *)lds;
        float *ew = E + ((wp * 3) * TILES + 4 * h) * ESTR_R + l31;
        const float *er = E + etile * ESTR_R + co;
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int n = rd * 2 + s;
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const f32x2 m0 = {acc[0][n][e], acc[0][n][e + 1]}, m1 = {acc[1][n][e], acc[1][n][e + 1]};
                    const f32x2 m2 = {acc[2][n][e], acc[2][n][e + 1]}, m3 = {acc[3][n][e], acc[3][n][e + 1]};
                    const f32x2 m4 = {acc[4][n][e], acc[4][n][e + 1]};
                    const f32x2 s12 = pk_add2(m1, m2);
                    const f32x2 y0 = pk_add2(pk_add2(m0, m3), s12);
                    const f32x2 y1 = pk_fma2(m3, two, pk_sub2(m1, m2));
                    const f32x2 y2 = pk_add2(pk_fma2(m3, four, s12), m4);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int tile = ((e + k) & 3) + 8 * ((e + k) >> 2);     // + 4 h (in ew)
                        ew[(0 * TILES + tile) * ESTR_R + 32 * s] = y0[k];
                        ew[(1 * TILES + tile) * ESTR_R + 32 * s] = y1[k];
                        ew[(2 * TILES + tile) * ESTR_R + 32 * s] = y2[k];
                    }
                    const int seg = s * 8 + e / 2;                               // 0 .. 15
                    if (rd == 1 && seg < 12) buf_st4(yrs, ooff, goff(seg / 6, seg % 6), pend[seg]);   // round 0's vector `seg`
                    if (XQ_RES_SPREAD && rd == 0) {
#pragma unroll
                        for (int j = seg * XQ_RES_SPREAD; j < (seg + 1) * XQ_RES_SPREAD && j < 12; ++j) resv[j / 6][j % 6] = buf_ld4(rrs, ooff, goff(j / 6, j % 6));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int n = rd * 2 + s;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int ya = it / 3, yb = it % 3;
                    const float *e0 = er + yb * TILES * ESTR_R + 32 * s;
                    const int pstride = 3 * TILES * ESTR_R;      // next Winograd row p
                    f32x4 y;
                    if (ya == 0) y = pk_add4(pk_add4(*(const f32x4 *)(e0), *(const f32x4 *)(e0 + pstride)), *(const f32x4 *)(e0 + 2 * pstride));
                    else y = pk_sub4(pk_sub4(*(const f32x4 *)(e0 + pstride), *(const f32x4 *)(e0 + 2 * pstride)), *(const f32x4 *)(e0 + 3 * pstride));
                    y = pk_add4(pk_add4(y, bv[n]), resv[s][it]);
                    y.x = max1(rlo, y.x); y.y = max1(rlo, y.y); y.z = max1(rlo, y.z); y.w = max1(rlo, y.w);
                    if (rd == 0) {
                        pend[s * 6 + it] = y;
                        resv[s][it] = buf_ld4(rrs, ooff, goff(2 + s, it));      // round 1's residual, one per finished vector
                    } else {
                        buf_st4(yrs, ooff, goff(n, it), y);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (rd == 0) __syncthreads();             // round 1 overwrites the planes
        }
        return;
    }
    // ---- epilogue: Y = A_r^T M A_c, bias, residual, ReLU; one 32-channel half at a time through LDS --------------
    // Thread (tile = tid >> 3, channel quad = tid & 7) writes the six pixels of its tile, so the pixel inside the tile is
    // a compile-time constant of the unrolled loop.  The residual loads come first: their HBM latency hides behind the
    // register reduction and the exchange (the main loop's operand registers are dead by now).
    const int etile = tid >> 3, co = (tid & 7) * 4;
    const int eg = t0 + etile;
    const bool eok = eg < T;
    const int ebd = eg / 15, et2 = eg - ebd * 15, ety = et2 / 3, etx = et2 - ety * 3;
    const size_t obase = eok ? ((size_t)ebd * 90 + (2 * ety) * 9 + 3 * etx) * C + cog * NCO + co : 0;
    const bool has_r = R != nullptr;
    // The exchange runs in rounds of RW N-tiles (32 RW output channels): one round = column transform + LDS writes, ONE
    // barrier, row transform + bias/residual/ReLU + stores by thread (tile, channel quad).  Narrow variant: RW = 1, planes
    // over the staging buffers, a second barrier before they are rewritten.  Wide variant (alone on its CU: every
    // dependent LDS round trip is exposed, and the LDS is all its own): RW = 2 -- two rounds instead of four
    // (104 KB of planes).  Residual loads run one round ahead.
    constexpr int RW = NT == 4 ? 2 : 1, ROUNDS = NT / RW;
    constexpr int ESTR_R = RW == 2 ? 64 + XQ_EPAD_WIDE : ESTR;     // floats per tile row of a plane
    constexpr int ESET = 4 * 3 * TILES * ESTR_R;     // floats per plane set
    constexpr bool E2 = false;                        // two 64-channel plane sets (209 KB) do not fit the CU's 160 KB
    f32x4 resv[RW][6];
#pragma unroll
    for (int s = 0; s < RW; ++s)
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            resv[s][it] = (has_r && eok) ? ld4_r(R + obase + 32 * s + ((it / 3) * 9 + it % 3) * C) : z;
        }
    float *E = (float *)lds;
    float *ew0 = E + ((wp * 3) * TILES + 4 * h) * ESTR_R + l31;     // + compile-time offsets: immediates of the LDS ops
    const float *er0 = E + etile * ESTR_R + co;
#pragma unroll
    for (int rd = 0; rd < ROUNDS; ++rd) {
        float *ew = ew0 + (E2 ? (rd & 1) * ESET : 0);
        const float *er = er0 + (E2 ? (rd & 1) * ESET : 0);
        // column half of the inverse transform on register pairs (packed fp32): y0 = m0+m1+m2+m3, y1 = m1-m2+2 m3,
        // y2 = m1+m2+4 m3+m4
#pragma unroll
        for (int s = 0; s < RW; ++s) {
            const int n = rd * RW + s;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 m0 = {acc[0][n][e], acc[0][n][e + 1]}, m1 = {acc[1][n][e], acc[1][n][e + 1]};
                const f32x2 m2 = {acc[2][n][e], acc[2][n][e + 1]}, m3 = {acc[3][n][e], acc[3][n][e + 1]};
                const f32x2 m4 = {acc[4][n][e], acc[4][n][e + 1]};
                const f32x2 s12 = pk_add2(m1, m2);
                const f32x2 y0 = pk_add2(pk_add2(m0, m3), s12);
                const f32x2 y1 = pk_fma2(m3, two, pk_sub2(m1, m2));
                const f32x2 y2 = pk_add2(pk_fma2(m3, four, s12), m4);
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int tile = ((e + k) & 3) + 8 * ((e + k) >> 2);     // + 4 h (in ew)
                    ew[(0 * TILES + tile) * ESTR_R + 32 * s] = y0[k];
                    ew[(1 * TILES + tile) * ESTR_R + 32 * s] = y1[k];
                    ew[(2 * TILES + tile) * ESTR_R + 32 * s] = y2[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < RW; ++s) {
            const int n = rd * RW + s;
            const f32x4 bv = *(const f32x4 *)(bias + cog * NCO + 32 * n + co);
            f32x4 yv[6];
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int ya = it / 3, yb = it % 3;
                const float *e0 = er + yb * TILES * ESTR_R + 32 * s;
                const int pstride = 3 * TILES * ESTR_R;      // next Winograd row p
                f32x4 y;
                if (ya == 0) y = *(const f32x4 *)(e0) + *(const f32x4 *)(e0 + pstride) + *(const f32x4 *)(e0 + 2 * pstride);
                else y = *(const f32x4 *)(e0 + pstride) - *(const f32x4 *)(e0 + 2 * pstride) - *(const f32x4 *)(e0 + 3 * pstride);
                y = y + bv + resv[s][it];
                if (relu) { y.x = relu1(y.x); y.y = relu1(y.y); y.z = relu1(y.z); y.w = relu1(y.w); }
                yv[it] = y;
            }
            if (eok) {
#pragma unroll
                for (int it = 0; it < 6; ++it) st4_y(Y + obase + 32 * n + ((it / 3) * 9 + it % 3) * C, yv[it]);
            }
            if (rd + 1 < ROUNDS && has_r && eok) {
#pragma unroll
                for (int it = 0; it < 6; ++it)        // residual of the next round: in flight during its column transform
                    resv[s][it] = ld4_r(R + obase + 32 * (n + RW) + ((it / 3) * 9 + it % 3) * C);
            }
        }
        if (rd + 1 < ROUNDS && !E2) __syncthreads();  // the next round overwrites the planes
    }
}


// Filter transform on the device (train step: the filters change every optimizer step).  One thread = one output channel x four input
// channels: 20 float4 stores, coalesced over the output channel.  float64 arithmetic as the host transform (hip.wino_transform_weights),
// rounded once to float32.  `dgrad`: the filters of the data-gradient convolution, w'[co][ci][r][s] = w[ci][co][2 - r][2 - s].
__device__ __forceinline__ void filter_transform_one(const float *__restrict__ W, float *__restrict__ U, int C, int nco, int dgrad, int co, int cq) {
    const double gr[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const double gc[5][3] = {{0.5, 0.0, 0.0}, {0.5, 0.5, 0.5}, {1.0 / 6.0, -1.0 / 6.0, 1.0 / 6.0}, {1.0 / 6.0, 2.0 / 6.0, 4.0 / 6.0}, {0.0, 0.0, 1.0}};
    float out[20][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ci = 4 * cq + k;
        double g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                g[r][q] = dgrad ? (double)W[((size_t)ci * C + co) * 9 + (2 - r) * 3 + (2 - q)] : (double)W[((size_t)co * C + ci) * 9 + r * 3 + q];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            double tr[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) tr[q] = gr[p][0] * g[0][q] + gr[p][1] * g[1][q] + gr[p][2] * g[2][q];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const double v = tr[0] * gc[j][0] + tr[1] * gc[j][1] + tr[2] * gc[j][2];
                out[5 * p + j][k] = (float)(p == 2 ? -v : v);
            }
        }
    }
    const int cog = co / nco, col = co - cog * nco, chunk = cq >> 1, quad = cq & 1;
    float *dst = U + ((((size_t)cog * (C / 8) + chunk) * 20 * 2 + quad) * nco + col) * 4;
#pragma unroll
    for (int xi = 0; xi < 20; ++xi) {
        const f32x4 v = {out[xi][0], out[xi][1], out[xi][2], out[xi][3]};
        *(f32x4 *)(dst + (size_t)xi * 2 * nco * 4) = v;
    }
}

// mode 0: forward filters; 1: data-gradient filters; 2: both, the data-gradient tensor right behind the forward one (one launch per layer
// and train step instead of two)
__global__ __launch_bounds__(256) void k_wino_filter_transform(const float *__restrict__ W, float *__restrict__ U, int C, int nco, int mode) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= C * (C / 4)) return;
    const int co = t % C, cq = t / C;
    if (mode != 1) filter_transform_one(W, U, C, nco, 0, co, cq);
    if (mode != 0) filter_transform_one(W, mode == 2 ? U + (size_t)20 * C * C : U, C, nco, 1, co, cq);
}

}  // namespace

extern "C" {

/* bytes of the pre-transformed weight tensor Ug for C channels: 20 * C * C floats */
size_t xq_wino_weight_bytes(int channels) { return (size_t)20 * channels * channels * sizeof(float); }

int xq_wino_conv3x3(const float *dev_x, const float *dev_u, const float *dev_bias, const float *dev_residual, float *dev_y,
                      int batch, int channels, int flags, void *stream) {
    if (!dev_x || !dev_u || !dev_bias || !dev_y || batch <= 0) return XQ_ERR_ARG;
    const bool wide = (flags & XQ_CONV_WIDE) != 0;
    const int nco = wide ? 128 : 64;
    if (channels < nco || channels % nco || 8 % (channels / nco)) return XQ_ERR_ARG;
    if (dev_x == dev_y || dev_residual == dev_y) return XQ_ERR_ARG;
    if (((uintptr_t)dev_x | (uintptr_t)dev_u | (uintptr_t)dev_bias | (uintptr_t)dev_residual | (uintptr_t)dev_y) & 15) return XQ_ERR_ARG;
    if ((unsigned long long)batch * 90ull * (unsigned)channels * 4ull >= (1ull << 32)) return XQ_ERR_ARG;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_WIDE));
        attr_set = true;
    }
    const int n_groups = (batch * 15 + TILES - 1) / TILES;
    const int lds_bytes = (XQ_ABL & 65536) ? 100 * 1024 : LDS_BYTES;      // ablation: one workgroup per CU
    const int per = 8 / (channels / nco);
    const int rows = (n_groups + per - 1) / per;
    if (wide)
        hipLaunchKernelGGL(k_wino_conv<4>, dim3(rows * 8), dim3(256), LDS_BYTES_WIDE, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                           dev_residual, dev_y, batch, channels, flags, n_groups);
    else
        hipLaunchKernelGGL(k_wino_conv<2>, dim3(rows * 8), dim3(256), lds_bytes, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                           dev_residual, dev_y, batch, channels, flags, n_groups);
    return xq::launch_status();
}

int xq_wino_transform_filters(const float *dev_w, float *dev_u, int channels, int flags, void *stream) {
    if (!dev_w || !dev_u || dev_w == dev_u) return XQ_ERR_ARG;
    const int nco = (flags & XQ_CONV_WIDE) ? 128 : 64;
    if (channels < nco || channels % nco || 8 % (channels / nco)) return XQ_ERR_ARG;
    if (((uintptr_t)dev_w | (uintptr_t)dev_u) & 15) return XQ_ERR_ARG;
    const int threads = channels * (channels / 4);
    hipLaunchKernelGGL(k_wino_filter_transform, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, dev_w, dev_u, channels,
                       nco, (flags & XQ_FILTER_BOTH) ? 2 : (flags & XQ_FILTER_DGRAD) ? 1 : 0);
    return xq::launch_status();
}

}  // extern "C"

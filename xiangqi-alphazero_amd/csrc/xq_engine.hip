// xq_engine.hip -- device-resident self-play engine (B3): SoA MCTS trees in HBM, one wavefront per game.
//
//   k_select : per game, advance the game/search state machine until ONE network evaluation is needed:
//                finish games (flush samples with z), start games (random opening), finish moves (visit
//                counts -> sample -> sampled action -> make_move), run simulations: PUCT descent with lanes over
//                the children of each node (coalesced N/W/P reads), moves replayed on an LDS board with the
//                12-board repetition ring, leaf terminal test (full ordered move generation).  Terminal leaves
//                are backed up in place and the next simulation starts at once, so every live slot hands exactly
//                one position to the evaluator per step.
//   k_expand : per game, consume the evaluator's output: softmax over all 8100 logits (as model.py:122), the
//                reference's sequential-float32 mask-and-normalise (mcts.py:176-188), children appended to the
//                slot's bump arena, Dirichlet noise at the root (mcts.py:117-121), resign probe
//                (parallel_selfplay.py:110-121), backup along the recorded path (mcts.py:66-73).
//
// Numeric contract (pinned by tests against the reference's MCTS under a stub evaluator): priors float32,
// PUCT evaluated in float32 as f32(q) + ((f32(c)*P)*f32(sqrt(N_parent)))/f32(1+N); at a noisy root (and for the
// uniform fallback) priors and PUCT are float64; W accumulates in float64; first maximum wins.
// Floating-point contraction is OFF for this file.
#include <math.h>
#include <string.h>

#include "xq_common.h"
#include "xq_rules.cuh"

#pragma clang fp contract(off)

using namespace xq;

static_assert(sizeof(xq_sample) == XQ_SAMPLE_BYTES, "xq_sample layout");
static_assert(sizeof(xq_game_result) == XQ_RESULT_BYTES, "xq_game_result layout");

namespace {

enum Phase : int { PH_NEWGAME = 0, PH_NEWPOS = 1, PH_WAIT_ROOT = 2, PH_SEARCH = 3, PH_WAIT_LEAF = 4, PH_FINISHED = 5,
                   PH_IDLE = 6, PH_HOLD = 7 };

enum Gi : int { GI_SIDE = 0, GI_MC, GI_NOCAP, GI_PHASE, GI_SIMS, GI_NSAMP, GI_GSEQ, GI_ALLOC, GI_PLEAF, GI_PDEPTH,
                GI_PCOUNT, GI_RSTATUS, GI_RWINNER, GI_RESIGN_N, GI_RNG0, GI_RNG1, GI_RNG2, GI_RNG3, GI_FWINNER,
                GI_FREASON, GI_MANNOISE, GI_DELAY, GI_N = 32 };

enum St : int { ST_SIMS = 0, ST_TERM, ST_LEAF, ST_ROOT, ST_MOVES, ST_GAMES, ST_RED, ST_BLACK, ST_DRAW, ST_PLIES, ST_NODES,
                ST_DEPTH, ST_SCAN, ST_RESIGN, ST_SAMP, ST_DROP, ST_OVF, ST_STARTED, ST_N = 32 };

enum Ptr : int { P_BOARD = 0, P_HIST, P_GI, P_RESIGN, P_PMOVES, P_PATH, P_TN, P_TW, P_TP, P_TA, P_TC, P_TM, P_ROOTP,
                 P_STAGE, P_OUTS, P_OUTR, P_CNT, P_STATS, P_INJECT, P_SQRT, P_MNOISE, P_STATSUM, P_REQ };

enum Rng : int { RNG_RANDINT = 0, RNG_CHOICE = 1, RNG_DIRICHLET = 2, RNG_UNIFORM = 3 };

// Device view of the engine (passed by value to kernels)
struct Dev {
    xq_engine_config cfg;
    int node_cap, path_cap, stage_cap;
    int8_t *board, *hist;
    int32_t *gi;
    double *resign;
    uint16_t *pmoves;
    int32_t *path;
    int32_t *tN; double *tW; float *tP; uint16_t *tA; int32_t *tC; uint16_t *tM;
    double *rootP;
    uint8_t *stage, *outs, *outr;
    unsigned int *cnt;              // [0] out samples, [1] out results
    unsigned long long *started;    // games started (quota)
    unsigned long long *stats;      // [G][ST_N]
    const uint64_t *inject;
    const double *sqrt_tab;
    double *mnoise;
    int32_t *req;                   // [G] legal moves of the evaluation each slot asked for this step (0: none)
};

Dev make_dev(const xq_engine *e) {
    Dev d;
    d.cfg = e->cfg;
    d.node_cap = e->node_cap; d.path_cap = e->path_cap; d.stage_cap = e->stage_cap;
    d.board = (int8_t *)e->p[P_BOARD]; d.hist = (int8_t *)e->p[P_HIST]; d.gi = (int32_t *)e->p[P_GI];
    d.resign = (double *)e->p[P_RESIGN]; d.pmoves = (uint16_t *)e->p[P_PMOVES]; d.path = (int32_t *)e->p[P_PATH];
    d.tN = (int32_t *)e->p[P_TN]; d.tW = (double *)e->p[P_TW]; d.tP = (float *)e->p[P_TP];
    d.tA = (uint16_t *)e->p[P_TA]; d.tC = (int32_t *)e->p[P_TC]; d.tM = (uint16_t *)e->p[P_TM];
    d.rootP = (double *)e->p[P_ROOTP]; d.stage = (uint8_t *)e->p[P_STAGE]; d.outs = (uint8_t *)e->p[P_OUTS];
    d.outr = (uint8_t *)e->p[P_OUTR]; d.cnt = (unsigned int *)e->p[P_CNT];
    d.started = (unsigned long long *)((char *)e->p[P_CNT] + 16);
    d.stats = (unsigned long long *)e->p[P_STATS]; d.inject = (const uint64_t *)e->p[P_INJECT];
    d.sqrt_tab = (const double *)e->p[P_SQRT]; d.mnoise = (double *)e->p[P_MNOISE]; d.req = (int32_t *)e->p[P_REQ];
    return d;
}

// ---------------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10 keyed by (seed, rank), counter (slot, kind, ctr, sub); or injected raw draws (tests).
__device__ __forceinline__ void philox_round(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__device__ inline uint64_t philox_u64(uint64_t seed, uint32_t rank, uint32_t slot, uint32_t kind, uint32_t ctr, uint32_t sub) {
    uint32_t c0 = slot, c1 = kind | (sub << 8), c2 = ctr, c3 = rank;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return ((uint64_t)c0 << 32) | c1;
}

// raw 64-bit draw number `ctr` (0-based) of stream `kind` of this slot
__device__ inline uint64_t draw_u64(const Dev &E, int slot, int kind, int ctr, unsigned long long *st) {
    if (E.cfg.inject_len > 0) {
        if (ctr >= E.cfg.inject_len) { st[ST_OVF] |= 2ull; return 0; }
        return E.inject[((size_t)slot * 4 + kind) * E.cfg.inject_len + ctr];
    }
    return philox_u64(E.cfg.seed, (uint32_t)E.cfg.rank, (uint32_t)slot, (uint32_t)kind, (uint32_t)ctr, 0);
}

__device__ inline double u64_to_unit(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

// Gamma(alpha) variate for lane-private use (Marsaglia-Tsang on alpha+1, boosted by U^(1/alpha))
__device__ inline double gamma_variate(const Dev &E, int slot, int ctr, double alpha) {
    const double d = alpha + 1.0 - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    double g = d;
    for (uint32_t it = 0; it < 64; ++it) {
        const uint64_t r0 = philox_u64(E.cfg.seed, (uint32_t)E.cfg.rank, (uint32_t)slot, RNG_DIRICHLET, (uint32_t)ctr, 1 + 2 * it);
        const uint64_t r1 = philox_u64(E.cfg.seed, (uint32_t)E.cfg.rank, (uint32_t)slot, RNG_DIRICHLET, (uint32_t)ctr, 2 + 2 * it);
        const double u1 = ((double)(r0 >> 11) + 0.5) * (1.0 / 9007199254740992.0);
        const double u2 = u64_to_unit(r1);
        const double u3 = ((double)(uint32_t)(r0 * 0x9E3779B97F4A7C15ull >> 32) + 0.5) * (1.0 / 4294967296.0);
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        double v = 1.0 + c * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(u3) < 0.5 * x * x + d * (1.0 - v + log(v))) { g = d * v; break; }
    }
    const uint64_t rb = philox_u64(E.cfg.seed, (uint32_t)E.cfg.rank, (uint32_t)slot, RNG_DIRICHLET, (uint32_t)ctr, 0);
    const double ub = ((double)(rb >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    return g * pow(ub, 1.0 / alpha);
}

// ---------------------------------------------------------------------------------------------------------
struct SelectLds {
    __attribute__((aligned(16))) int8_t root[XQ_BS];          // the game's real board
    __attribute__((aligned(16))) int8_t rhist[XQ_HIST][XQ_BS]; // its last 12 pre-move boards (slot = ply % 12)
    __attribute__((aligned(16))) int8_t board[XQ_BS];         // simulation board
    __attribute__((aligned(16))) int8_t hist[XQ_HIST][XQ_BS];  // simulation ring
    MoveGenLds mg;
    uint16_t moves[XQ_MAXM];
    uint16_t sa[XQ_MAXM];
    double sw[XQ_MAXM];
    double cdf[XQ_MAXM];
    uint16_t a_tmp[XQ_MAXM];
    double w_tmp[XQ_MAXM];
};

__device__ __forceinline__ void lds_copy_dwords(void *dst, const void *src, int ndw) {
    const int lane = lane_id();
    uint32_t *d = (uint32_t *)dst;
    const uint32_t *s = (const uint32_t *)src;
    for (int i = lane; i < ndw; i += 64) d[i] = s[i];
}

__device__ __forceinline__ void init_board_lds(int8_t *b) {
    const int lane = lane_id();
    for (int sq = lane; sq < XQ_BS; sq += 64) {
        int v = 0;
        if (sq < 90) {
            const int r = sq / 9, c = sq % 9;
            const int back = (c == 0 || c == 8) ? 5 : (c == 1 || c == 7) ? 4 : (c == 2 || c == 6) ? 3 : (c == 3 || c == 5) ? 2 : 1;
            if (r == 0) v = back;
            else if (r == 9) v = -back;
            else if (r == 2 && (c == 1 || c == 7)) v = 6;
            else if (r == 7 && (c == 1 || c == 7)) v = -6;
            else if (r == 3 && (c % 2 == 0)) v = 7;
            else if (r == 6 && (c % 2 == 0)) v = -7;
        }
        b[sq] = (int8_t)v;
    }
}

// game.py:565-616 on an LDS position.  Leaves the ordered legal moves in `moves` (count in *cnt) whenever both
// kings stand.  Wave-uniform result.
__device__ inline bool wave_game_over(const int8_t *b, const int8_t (*ring)[XQ_BS], int side, int mc, int nocap,
                                      MoveGenLds &mg, uint16_t *moves, int *cnt, int *winner, int *ovf) {
    const VMove none{-1, -1, 0};
    const int lane = lane_id();
    *cnt = 0;
    if (find_king(b, none, 1) < 0) { *winner = -1; return true; }
    if (find_king(b, none, -1) < 0) { *winner = 1; return true; }
    const int n = wave_movegen(b, side, mg, moves, ovf);
    *cnt = n;
    if (n == 0) { *winner = -side; return true; }
    if (nocap >= 120) { *winner = 0; return true; }
    if (mc >= 200) {
        int red, black;
        wave_material(b, red, black);
        const int diff = red - black;
        *winner = diff > 30 ? 1 : (diff < -30 ? -1 : 0);
        return true;
    }
    if (mc >= 6) {
        const int k = mc < XQ_HIST ? mc : XQ_HIST;
        int rep = 0;
        for (int e = 0; e < k; ++e) {
            const int8_t *h = ring[(mc - 1 - e) % XQ_HIST];
            const uint32_t x = lane < 23 ? (((const uint32_t *)h)[lane] ^ ((const uint32_t *)b)[lane]) : 0u;
            if (__ballot(x != 0u) == 0ull) ++rep;
        }
        if (rep >= 3) { *winner = 0; return true; }
    }
    *winner = 2;
    return false;
}

// game.py:528-550 on an LDS position + ring.  Wave-uniform scalars updated by reference.
__device__ __forceinline__ void wave_make_move(int8_t *b, int8_t (*ring)[XQ_BS], int action, int &side, int &mc, int &nocap) {
    const int from = action / 90, to = action - from * 90;
    lds_copy_dwords(ring[mc % XQ_HIST], b, XQ_BS / 4);
    const int captured = b[to], mover = b[from];
    wave_sync();
    if (lane_id() == 0) { b[to] = (int8_t)mover; b[from] = 0; }
    wave_sync();
    nocap = captured != 0 ? 0 : nocap + 1;
    side = -side;
    mc += 1;
}

// mcts.py:66-73 along path[0..depth]
__device__ __forceinline__ void wave_backup(int32_t *tN, double *tW, const int32_t *path, int depth, double v) {
    for (int j = lane_id(); j <= depth; j += 64) {
        const int nd = path[j];
        const double s = ((depth - j) & 1) ? -v : v;
        tN[nd] += 1;
        tW[nd] += s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Four independent games per 256-thread workgroup (one per wave, LDS carved per wave, no workgroup barrier anywhere).
// Measured: k_select's time at G = 8192 does not depend on the workgroup shape (0.15 ms either way, 0.04 ms at G = 2048):
// it saturates a CU at ~8 resident games because the leaf's move generation / legality scan is LDS-instruction bound
// (thousands of byte reads of the LDS board per game), not HBM bound.
constexpr int WAVES_PER_WG = 4;

#ifndef XQ_SELECT_WAVES_PER_EU
// 0 = the compiler's choice (240 VGPRs, two waves per SIMD, no spills) -- the shipped setting.  Capping the registers for three / four
// waves per SIMD (168 / 128 VGPRs) was measured: select 0.135 / 0.159 ms instead of 0.177 (near-uniform) and 0.350 / 0.328 instead of 0.364
// (peaked), but the 47 / 94 spilled registers are stored by EVERY wave (12 KB of scratch per game): k_select's HBM-side writes went from
// 47 MB to 146 MB per launch (TCC_EA0_WRREQ).  0.04 ms of a 50 ms step is not worth tripling the kernel's traffic.
#define XQ_SELECT_WAVES_PER_EU 0
#endif
#if XQ_SELECT_WAVES_PER_EU > 0
#define XQ_SELECT_OCC __attribute__((amdgpu_waves_per_eu(XQ_SELECT_WAVES_PER_EU, XQ_SELECT_WAVES_PER_EU)))
#else
#define XQ_SELECT_OCC
#endif
__global__ __launch_bounds__(64 * WAVES_PER_WG) XQ_SELECT_OCC void k_select(Dev E, float *__restrict__ nn_in) {
    __shared__ SelectLds Ls[WAVES_PER_WG];
    SelectLds &L = Ls[threadIdx.x >> 6];
    const int slot = blockIdx.x * WAVES_PER_WG + (int)(threadIdx.x >> 6);
    if (slot >= E.cfg.n_games) return;
    const int lane = lane_id();
    int32_t *gi = E.gi + (size_t)slot * GI_N;
    unsigned long long *st = E.stats + (size_t)slot * ST_N;
    const size_t nb = (size_t)slot * E.node_cap;
    int32_t *tN = E.tN + nb; double *tW = E.tW + nb; float *tP = E.tP + nb;
    uint16_t *tA = E.tA + nb; int32_t *tC = E.tC + nb; uint16_t *tM = E.tM + nb;
    int32_t *path = E.path + (size_t)slot * E.path_cap;
    int8_t *g_board = E.board + (size_t)slot * XQ_BS;
    int8_t *g_hist = E.hist + (size_t)slot * XQ_HIST * XQ_BS;
    uint16_t *pmoves = E.pmoves + (size_t)slot * XQ_MAXM;
    const double *rootP = E.rootP + (size_t)slot * XQ_MAXM;
    const int S = E.cfg.num_simulations;
    const bool manual = E.cfg.manual_moves == 1;     // search only (MCTS.search parity / serving)
    const bool arena = E.cfg.manual_moves == 2;      // evaluation games (train.py:453-535): T = 0, no opening, no samples

    int phase = __builtin_amdgcn_readfirstlane(gi[GI_PHASE]);
    if (phase == PH_WAIT_ROOT || phase == PH_WAIT_LEAF) return;        // still waiting: the request stands
    if (phase == PH_IDLE || phase == PH_HOLD) { if (lane == 0) E.req[slot] = 0; return; }
    const int delay = __builtin_amdgcn_readfirstlane(gi[GI_DELAY]);
    if (delay > 0) {                                  // start_stagger: not started yet
        if (lane == 0) { gi[GI_DELAY] = delay - 1; E.req[slot] = 0; }
        return;
    }
    int req_cnt = 0;

    int g_side = __builtin_amdgcn_readfirstlane(gi[GI_SIDE]);
    int g_mc = __builtin_amdgcn_readfirstlane(gi[GI_MC]);
    int g_nocap = __builtin_amdgcn_readfirstlane(gi[GI_NOCAP]);
    int sims_done = __builtin_amdgcn_readfirstlane(gi[GI_SIMS]);
    int n_samples = __builtin_amdgcn_readfirstlane(gi[GI_NSAMP]);
    int game_seq = __builtin_amdgcn_readfirstlane(gi[GI_GSEQ]);
    int rng_ctr[4] = {__builtin_amdgcn_readfirstlane(gi[GI_RNG0]), __builtin_amdgcn_readfirstlane(gi[GI_RNG1]),
                      __builtin_amdgcn_readfirstlane(gi[GI_RNG2]), __builtin_amdgcn_readfirstlane(gi[GI_RNG3])};
    int ovf = 0;
    // per-lane stat deltas are kept wave-uniform and written by lane 0 at the end
    unsigned long long d_sims = 0, d_term = 0, d_moves = 0, d_depth = 0, d_scan = 0;
    int term_run = 0;

    lds_copy_dwords(L.root, g_board, XQ_BS / 4);
    lds_copy_dwords(L.rhist, g_hist, XQ_HIST * XQ_BS / 4);
    wave_sync();

    bool state_dirty = false;  // real game state changed -> write back
    for (int guard = 0; guard < 4 * S + 64; ++guard) {
        if (phase == PH_FINISHED) {
            // ---- flush the finished game's samples with z (parallel_selfplay.py:123-132) and its result
            const int winner = __builtin_amdgcn_readfirstlane(gi[GI_FWINNER]);
            const int reason = __builtin_amdgcn_readfirstlane(gi[GI_FREASON]);
            unsigned base = 0;
            bool fits = true;
            if (n_samples > 0) {
                if (lane == 0) base = atomicAdd(&E.cnt[0], (unsigned)n_samples);
                base = __builtin_amdgcn_readfirstlane(base);
                fits = (unsigned long long)base + (unsigned)n_samples <= (unsigned)E.cfg.max_out_samples;
                if (fits) {
                    const uint8_t *src = E.stage + (size_t)slot * E.stage_cap * XQ_SAMPLE_BYTES;
                    uint8_t *dst = E.outs + (size_t)base * XQ_SAMPLE_BYTES;
                    const int ndw = n_samples * (XQ_SAMPLE_BYTES / 4);
                    for (int i = lane; i < ndw; i += 64) ((uint32_t *)dst)[i] = ((const uint32_t *)src)[i];
                    for (int i = lane; i < n_samples; i += 64) {
                        const int sside = ((const int8_t *)src)[(size_t)i * XQ_SAMPLE_BYTES + 90];
                        ((int8_t *)dst)[(size_t)i * XQ_SAMPLE_BYTES + 91] = (int8_t)(winner == 0 ? 0 : (winner == sside ? 1 : -1));
                    }
                }
            }
            if (lane == 0) {
                if (fits) st[ST_SAMP] += (unsigned)n_samples; else st[ST_DROP] += (unsigned)n_samples;
                const unsigned r = atomicAdd(&E.cnt[1], 1u);
                if (r < (unsigned)E.cfg.max_out_results) {
                    xq_game_result res;
                    res.slot = (uint32_t)slot; res.game_seq = (uint32_t)game_seq; res.winner = (int8_t)winner;
                    res.reason = (uint8_t)reason; res.steps = (uint16_t)g_mc; res.n_samples = (uint16_t)n_samples;
                    res.reserved = 0;
                    *(xq_game_result *)(E.outr + (size_t)r * XQ_RESULT_BYTES) = res;
                }
                st[ST_GAMES] += 1;
                st[winner == 1 ? ST_RED : (winner == -1 ? ST_BLACK : ST_DRAW)] += 1;
                st[ST_PLIES] += (unsigned)g_mc;
                if (reason == 3) st[ST_RESIGN] += 1;
            }
            phase = PH_NEWGAME;
        }
        if (phase == PH_NEWGAME) {
            // ---- new game + random opening (parallel_selfplay.py:58-72)
            unsigned long long idx = 0;
            if (lane == 0) idx = atomicAdd(E.started, 1ull);
            idx = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(idx >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((unsigned)idx);
            if (E.cfg.games_target > 0 && idx >= (unsigned long long)E.cfg.games_target) { phase = PH_IDLE; break; }
            game_seq += 1;
            n_samples = 0;
            init_board_lds(L.root);
            g_side = 1; g_mc = 0; g_nocap = 0;
            if (lane == 0) { gi[GI_RESIGN_N] = 0; st[ST_STARTED] += 1; }
            wave_sync();
            const int R = arena ? 0 : E.cfg.random_opening_moves;
            const int k = R > 0 ? (int)(draw_u64(E, slot, RNG_RANDINT, rng_ctr[RNG_RANDINT], st) % (uint64_t)(R + 1)) : 0;
            if (!arena) rng_ctr[RNG_RANDINT] += 1;   // random.randint is called even when R == 0
            for (int i = 0; i < k; ++i) {
                const int cnt = wave_movegen(L.root, g_side, L.mg, L.moves, &ovf);
                if (cnt == 0) break;
                const int pick = (int)(draw_u64(E, slot, RNG_CHOICE, rng_ctr[RNG_CHOICE], st) % (uint64_t)cnt);
                rng_ctr[RNG_CHOICE] += 1;
                const int action = L.moves[pick];
                wave_make_move(L.root, L.rhist, action, g_side, g_mc, g_nocap);
                int c2, w2;
                if (wave_game_over(L.root, L.rhist, g_side, g_mc, g_nocap, L.mg, L.moves, &c2, &w2, &ovf)) {
                    init_board_lds(L.root);
                    g_side = 1; g_mc = 0; g_nocap = 0;
                    wave_sync();
                    break;
                }
            }
            state_dirty = true;
            phase = PH_NEWPOS;
        }
        if (phase == PH_NEWPOS) {
            // ---- root request: terminal status of the real position + its planes + its ordered legal moves
            int cnt, winner;
            int status = 0;
            const bool done = wave_game_over(L.root, L.rhist, g_side, g_mc, g_nocap, L.mg, L.moves, &cnt, &winner, &ovf);
            if (done) status = 1;
            else if (arena && g_mc >= E.cfg.max_game_length) {     // train.py:477,494-496: not over after max plies => draw
                winner = 0;
                status = 2;
            } else if (!manual && !arena && g_mc >= E.cfg.max_game_length) {   // parallel_selfplay.py:79-89
                int red, black;
                wave_material(L.root, red, black);
                const int diff = red - black;
                winner = diff > 30 ? 1 : (diff < -30 ? -1 : 0);
                status = 2;
            }
            wave_encode(L.root, g_side, nn_in + (size_t)slot * XQ_STATE_FLOATS);
            for (int j = lane; j < cnt; j += 64) pmoves[j] = L.moves[j];
            if (lane == 0) {
                gi[GI_PCOUNT] = cnt; gi[GI_RSTATUS] = status; gi[GI_RWINNER] = winner;
                gi[GI_ALLOC] = 1;
                tN[0] = 0; tW[0] = 0.0; tC[0] = -1; tM[0] = 0; tA[0] = 0; tP[0] = 0.0f;
            }
            sims_done = 0;
            req_cnt = cnt;
            phase = PH_WAIT_ROOT;
            break;
        }
        // ---- phase == PH_SEARCH
        if (sims_done >= S) {
            if (manual) { phase = PH_HOLD; break; }
            const int nch = __builtin_amdgcn_readfirstlane((int)(tM[0] & 0x3FFF));
            const int first = __builtin_amdgcn_readfirstlane(tC[0]);
            if (arena) {
                // MCTS.get_action(temperature=0) (mcts.py:166-174, 197-200): first maximum of the visit counts, move order
                int bn = -1, bi = 0x7FFFFFFF;
                for (int i = lane; i < nch; i += 64) {
                    const int n = tN[first + i];
                    if (n > bn) { bn = n; bi = i; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const int on = __shfl_xor(bn, off), oi = __shfl_xor(bi, off);
                    if (on > bn || (on == bn && oi < bi)) { bn = on; bi = oi; }
                }
                bi = __builtin_amdgcn_readfirstlane(bi);
                const int action = __builtin_amdgcn_readfirstlane((int)tA[first + bi]);
                wave_make_move(L.root, L.rhist, action, g_side, g_mc, g_nocap);
                d_moves += 1;
                state_dirty = true;
                phase = PH_NEWPOS;
                continue;
            }
            // ---- end of move: sample (parallel_selfplay.py:97-107), pi from visit counts (mcts.py:190-206)
            const bool late = g_mc >= E.cfg.temperature_threshold;
            const double inv_t = 1.0 / E.cfg.late_temperature;
            uint8_t *rec = E.stage + ((size_t)slot * E.stage_cap + (n_samples < E.stage_cap ? n_samples : E.stage_cap - 1)) * XQ_SAMPLE_BYTES;
            if (n_samples >= E.stage_cap) ovf |= 4;
            for (int i = lane; i < XQ_SAMPLE_BYTES / 4; i += 64) ((uint32_t *)rec)[i] = 0u;
            wave_sync_mem();
            for (int i = lane; i < 90; i += 64) rec[i] = (uint8_t)L.root[i];
            if (lane == 0) {
                xq_sample *s = (xq_sample *)rec;
                s->side = (int8_t)g_side; s->z = 0; s->n_moves = (uint8_t)nch; s->late_temp = late ? 1 : 0;
                s->ply = (uint16_t)g_mc; s->slot = (uint32_t)slot; s->game_seq = (uint32_t)game_seq;
            }
            for (int i = lane; i < nch; i += 64) {
                const int a = tA[first + i], n = tN[first + i];
                ((xq_sample *)rec)->actions[i] = (uint16_t)a;
                ((xq_sample *)rec)->visits[i] = (uint16_t)(n > 65535 ? 65535 : n);
                L.a_tmp[i] = (uint16_t)a;
                L.w_tmp[i] = late ? (n > 0 ? pow((double)n, inv_t) : 0.0) : (double)n;
            }
            wave_sync();
            // np.random.choice walks the dense pi in ACTION-ID order: sort the (action, weight) pairs by id
            for (int i = lane; i < nch; i += 64) {
                const int a = L.a_tmp[i];
                int rank = 0;
                for (int j = 0; j < nch; ++j) rank += (L.a_tmp[j] < a) ? 1 : 0;
                L.sa[rank] = (uint16_t)a;
                L.sw[rank] = L.w_tmp[i];
            }
            wave_sync();
            const double u = u64_to_unit(draw_u64(E, slot, RNG_UNIFORM, rng_ctr[RNG_UNIFORM], st));
            rng_ctr[RNG_UNIFORM] += 1;
            int action = 0;
            {
                // every lane runs the same short sequential scan (LDS broadcast reads); result is wave-uniform
                double total = 0.0;
                for (int i = 0; i < nch; ++i) total += L.sw[i];
                double run = 0.0;
                for (int i = 0; i < nch; ++i) run += L.sw[i] / total;
                const double last = run;
                run = 0.0;
                int pick = nch - 1;
                for (int i = 0; i < nch; ++i) {
                    run += L.sw[i] / total;
                    if (run / last > u) { pick = i; break; }
                }
                action = L.sa[pick];
            }
            action = __builtin_amdgcn_readfirstlane(action);
            wave_make_move(L.root, L.rhist, action, g_side, g_mc, g_nocap);
            n_samples += 1;
            d_moves += 1;
            state_dirty = true;
            phase = PH_NEWPOS;
            continue;
        }
        // ---- one simulation (mcts.py:126-153): descend from the root replaying moves on the LDS board
        lds_copy_dwords(L.board, L.root, XQ_BS / 4);
        lds_copy_dwords(L.hist, L.rhist, XQ_HIST * XQ_BS / 4);
        wave_sync();
        int side = g_side, mc = g_mc, nocap = g_nocap, node = 0, depth = 0;
        if (lane == 0) path[0] = 0;
        // One dependent round trip to memory per level: every lane reads, with its candidate child's N / W / P, that child's own
        // node words (children count + kind, first child, action) as well, so the winner's are already in a register when the arg-max
        // is known -- the next level starts from a lane read instead of three more dependent loads (tM -> tC/tN -> ... -> tA).  The
        // winner's N, read here, IS the next level's parent count.  Same values, same arithmetic, same order as before.
        int m = __builtin_amdgcn_readfirstlane((int)tM[0]);
        int first = __builtin_amdgcn_readfirstlane(tC[0]);
        int pn = __builtin_amdgcn_readfirstlane(tN[0]);
        for (;;) {
            const int nch = m & 0x3FFF, kind = m >> 14;
            if (nch == 0) break;
            const double sqrtp = E.sqrt_tab[pn];
            const float sqrtp_f = (float)sqrtp, c_f = (float)E.cfg.c_puct;
            const double uni = 1.0 / (double)nch;
            double best = -INFINITY;
            int best_i = 0x7FFFFFFF;
            int c_m = 0, c_first = 0, c_n = 0, c_a = 0;              // node words of this lane's best candidate
            for (int base = 0; base < nch; base += 64) {
                const int i = base + lane;
                if (i < nch) {
                    const int n = tN[first + i];
                    const double w = tW[first + i];
                    const int cm = (int)tM[first + i], cf = tC[first + i], ca = (int)tA[first + i];
                    const double q = n ? w / (double)n : 0.0;
                    double ucb;
                    if (kind == 0) {
                        float t = c_f * tP[first + i];
                        t = t * sqrtp_f;
                        t = t / (float)(1 + n);
                        t = (float)q + t;
                        ucb = (double)t;
                    } else {
                        const double p = kind == 1 ? rootP[i] : uni;
                        double t = E.cfg.c_puct * p;
                        t = t * sqrtp;
                        t = t / (double)(1 + n);
                        ucb = q + t;
                    }
                    if (ucb > best) { best = ucb; best_i = i; c_m = cm; c_first = cf; c_n = n; c_a = ca; }
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(best, off);
                const int oi = __shfl_xor(best_i, off);
                if (ov > best || (ov == best && oi < best_i)) { best = ov; best_i = oi; }
            }
            best_i = __builtin_amdgcn_readfirstlane(best_i);
            int action;
            if (best_i == 0x7FFFFFFF) {                              // all-NaN scores: the reference would raise
                ovf |= 8; best_i = 0;
                m = __builtin_amdgcn_readfirstlane((int)tM[first]);
                pn = __builtin_amdgcn_readfirstlane(tN[first]);
                action = __builtin_amdgcn_readfirstlane((int)tA[first]);
                const int nf = __builtin_amdgcn_readfirstlane(tC[first]);
                d_scan += (unsigned)nch;
                const int child0 = first;
                first = nf;
                wave_make_move(L.board, L.hist, action, side, mc, nocap);
                depth += 1;
                if (depth >= E.path_cap) { ovf |= 16; depth = E.path_cap - 1; }
                if (lane == 0) path[depth] = child0;
                node = child0;
                continue;
            }
            d_scan += (unsigned)nch;
            const int child = first + best_i;
            const int src = best_i & 63;                              // child i was lane i % 64's candidate, and its best (it won)
            action = __builtin_amdgcn_readlane(c_a, src);
            m = __builtin_amdgcn_readlane(c_m, src);
            pn = __builtin_amdgcn_readlane(c_n, src);
            first = __builtin_amdgcn_readlane(c_first, src);
            wave_make_move(L.board, L.hist, action, side, mc, nocap);
            depth += 1;
            if (depth >= E.path_cap) { ovf |= 16; depth = E.path_cap - 1; }
            if (lane == 0) path[depth] = child;
            node = child;
        }
        d_depth += (unsigned)depth;
        int cnt, winner;
        const bool term = wave_game_over(L.board, L.hist, side, mc, nocap, L.mg, L.moves, &cnt, &winner, &ovf);
        if (term) {
            wave_sync_mem();   // path[] stores of lane 0 must be visible to the other lanes
            wave_backup(tN, tW, path, depth, winner == 0 ? 0.0 : 1.0);   // mcts.py:137-140
            wave_sync_mem();   // the next descent reads N/W written here by other lanes
            sims_done += 1; d_sims += 1; d_term += 1;
            // A root with a mating reply re-tests that terminal child on every visit (as mcts.py does); bound how
            // many such simulations one launch runs so a single slot cannot stretch the step (it resumes next step
            // and hands the evaluator no position this time).
            if (++term_run >= 48) break;
            continue;
        }
        wave_encode(L.board, side, nn_in + (size_t)slot * XQ_STATE_FLOATS);
        for (int j = lane; j < cnt; j += 64) pmoves[j] = L.moves[j];
        if (lane == 0) { gi[GI_PLEAF] = node; gi[GI_PDEPTH] = depth; gi[GI_PCOUNT] = cnt; }
        req_cnt = cnt;
        phase = PH_WAIT_LEAF;
        break;
    }

    if (state_dirty) {
        lds_copy_dwords(g_board, L.root, XQ_BS / 4);
        lds_copy_dwords(g_hist, L.rhist, XQ_HIST * XQ_BS / 4);
    }
    if (lane == 0) {
        E.req[slot] = (phase == PH_WAIT_ROOT || phase == PH_WAIT_LEAF) ? req_cnt : 0;
        gi[GI_SIDE] = g_side; gi[GI_MC] = g_mc; gi[GI_NOCAP] = g_nocap; gi[GI_PHASE] = phase; gi[GI_SIMS] = sims_done;
        gi[GI_NSAMP] = n_samples; gi[GI_GSEQ] = game_seq;
        gi[GI_RNG0] = rng_ctr[0]; gi[GI_RNG1] = rng_ctr[1]; gi[GI_RNG2] = rng_ctr[2]; gi[GI_RNG3] = rng_ctr[3];
        st[ST_SIMS] += d_sims; st[ST_TERM] += d_term; st[ST_MOVES] += d_moves; st[ST_DEPTH] += d_depth; st[ST_SCAN] += d_scan;
        if (ovf) st[ST_OVF] |= (unsigned long long)ovf << 8;
    }
}

// ---------------------------------------------------------------------------------------------------------
struct ExpandLds {
    float p[XQ_MAXM];
    double eta[XQ_MAXM];
    uint16_t act[XQ_MAXM];
};

// one game per 64-thread workgroup: this kernel streams 32 KB of logits per game and measured faster with more,
// smaller workgroups in flight (0.111 vs 0.132 ms at G = 8192) than with four games per workgroup
__global__ __launch_bounds__(64) void k_expand(Dev E, const float *__restrict__ policy, const float *__restrict__ value,
                                               int is_probs) {
    __shared__ ExpandLds L;
    const int slot = blockIdx.x;
    if (slot >= E.cfg.n_games) return;
    const int lane = lane_id();
    int32_t *gi = E.gi + (size_t)slot * GI_N;
    unsigned long long *st = E.stats + (size_t)slot * ST_N;
    int phase = __builtin_amdgcn_readfirstlane(gi[GI_PHASE]);
    if (phase != PH_WAIT_ROOT && phase != PH_WAIT_LEAF) return;
    const size_t nb = (size_t)slot * E.node_cap;
    int32_t *tN = E.tN + nb; double *tW = E.tW + nb; float *tP = E.tP + nb;
    uint16_t *tA = E.tA + nb; int32_t *tC = E.tC + nb; uint16_t *tM = E.tM + nb;
    const int32_t *path = E.path + (size_t)slot * E.path_cap;
    const uint16_t *pmoves = E.pmoves + (size_t)slot * XQ_MAXM;
    double *rootP = E.rootP + (size_t)slot * XQ_MAXM;
    const bool manual = E.cfg.manual_moves == 1;
    const bool arena = E.cfg.manual_moves == 2;
    const bool is_root = phase == PH_WAIT_ROOT;
    const double v_net = (double)value[slot];        // tensor.item(): float32 widened
    const int cnt = __builtin_amdgcn_readfirstlane(gi[GI_PCOUNT]);
    int sims_done = __builtin_amdgcn_readfirstlane(gi[GI_SIMS]);
    int ovf = 0;

    if (is_root) {
        if (lane == 0) st[ST_ROOT] += 1;
        const int side = gi[GI_SIDE];
        int fin = 0, fwinner = 0, freason = 0;
        // resign probe on the position after the move (parallel_selfplay.py:110-121)
        if (!manual && !arena && E.cfg.enable_resign && gi[GI_NSAMP] > 10) {
            const int K = E.cfg.resign_check_steps;
            double *rh = E.resign + (size_t)slot * 16;
            int rn = gi[GI_RESIGN_N];
            wave_sync_mem();
            if (lane == 0) { rh[rn % 16] = v_net; gi[GI_RESIGN_N] = rn + 1; }
            wave_sync_mem();
            rn += 1;
            if (rn >= K) {
                bool all_low = true;
                for (int i = rn - K; i < rn; ++i) all_low = all_low && (rh[i % 16] < E.cfg.resign_threshold);
                if (all_low) { fin = 1; fwinner = -side; freason = 3; }
            }
        }
        const int rstatus = gi[GI_RSTATUS];
        if (!fin && rstatus != 0) { fin = 1; fwinner = gi[GI_RWINNER]; freason = rstatus; }
        fin = __builtin_amdgcn_readfirstlane(fin);
        if (fin) {
            if (lane == 0) {
                gi[GI_FWINNER] = fwinner; gi[GI_FREASON] = freason;
                gi[GI_PHASE] = manual ? PH_HOLD : PH_FINISHED;
            }
            return;
        }
    } else {
        if (lane == 0) st[ST_LEAF] += 1;
    }

    // ---- priors of the legal moves: softmax over ALL 8100 logits (model.py:122), then mcts.py:176-188
    // is_probs: 0 logits over all 8100 actions, 1 probabilities over all 8100, 2 logits of the legal moves only
    // ([G][XQ_MAXM], move order): softmax over the legal logits -- the common factor of the full softmax cancels in
    // mcts.py:176-188's renormalisation
    const float *pol = policy + (size_t)slot * (is_probs == 2 ? XQ_MAXM : XQ_ACTION_SPACE);
    float mx = 0.0f, den = 1.0f;
    if (is_probs == 2) {
        float m = -INFINITY;
        for (int i = lane; i < cnt; i += 64) m = fmaxf(m, pol[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        mx = m;
    } else if (!is_probs) {
        const float4 *p4 = (const float4 *)pol;
        float m = -INFINITY;
        for (int i = lane; i < XQ_ACTION_SPACE / 4; i += 64) {
            const float4 x = p4[i];
            m = fmaxf(fmaxf(m, fmaxf(x.x, x.y)), fmaxf(x.z, x.w));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        float s = 0.0f;
        for (int i = lane; i < XQ_ACTION_SPACE / 4; i += 64) {
            const float4 x = p4[i];
            s += expf(x.x - m) + expf(x.y - m) + expf(x.z - m) + expf(x.w - m);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        mx = m; den = s;
    }
    for (int i = lane; i < cnt; i += 64) {
        const int a = pmoves[i];
        const float x = pol[is_probs == 2 ? i : a];
        L.p[i] = is_probs == 1 ? x : expf(x - mx) / den;
        L.act[i] = (uint16_t)a;
    }
    wave_sync();
    float sum = 0.0f;
    for (int i = 0; i < cnt; ++i) sum = sum + L.p[i];      // builtin sum(): sequential float32, move order

    const bool noisy = is_root && (E.cfg.add_noise != 0 || gi[GI_MANNOISE] != 0);
    if (noisy) {
        // eta ~ Dirichlet(alpha) over the legal moves, in move order
        if (gi[GI_MANNOISE] != 0) {
            const double *mn = E.mnoise + (size_t)slot * XQ_MAXM;
            for (int i = lane; i < cnt; i += 64) L.eta[i] = mn[i];
        } else {
            const int ctr0 = gi[GI_RNG0 + RNG_DIRICHLET];
            for (int i = lane; i < cnt; i += 64) {
                double g;
                if (E.cfg.inject_len > 0) {           // tests/draws.py Draws.dirichlet: w = (1 + (u>>40)%4096)^3
                    const double w = (double)(1 + (int)((draw_u64(E, slot, RNG_DIRICHLET, ctr0 + i, st) >> 40) % 4096ull));
                    g = w * w * w;
                } else {
                    g = gamma_variate(E, slot, ctr0 + i, E.cfg.dirichlet_alpha);
                }
                L.eta[i] = g;
            }
            wave_sync();
            double tot = 0.0;
            for (int i = 0; i < cnt; ++i) tot += L.eta[i];
            wave_sync();
            for (int i = lane; i < cnt; i += 64) L.eta[i] = L.eta[i] / tot;
            if (lane == 0) gi[GI_RNG0 + RNG_DIRICHLET] = ctr0 + cnt;
        }
        wave_sync();
    }

    const int node = is_root ? 0 : __builtin_amdgcn_readfirstlane(gi[GI_PLEAF]);
    const int first = __builtin_amdgcn_readfirstlane(gi[GI_ALLOC]);
    if (cnt > 0) {
        if (first + cnt > E.node_cap) {
            ovf |= 32;
        } else {
            int kind;
            const double eps = E.cfg.noise_eps;
            const float keep_f = (float)(1.0 - eps);
            if (sum > 0.0f) {
                kind = noisy ? 1 : 0;
                for (int i = lane; i < cnt; i += 64) {
                    const float pr = L.p[i] / sum;
                    if (noisy) { const float sc = keep_f * pr; rootP[i] = (double)sc + eps * L.eta[i]; }
                    tP[first + i] = pr;
                }
            } else {
                kind = noisy ? 1 : 2;
                const double uni = 1.0 / (double)cnt;
                for (int i = lane; i < cnt; i += 64) {
                    if (noisy) rootP[i] = (1.0 - eps) * uni + eps * L.eta[i];
                    tP[first + i] = (float)uni;
                }
            }
            for (int i = lane; i < cnt; i += 64) {
                tN[first + i] = 0; tW[first + i] = 0.0; tA[first + i] = L.act[i]; tC[first + i] = -1; tM[first + i] = 0;
            }
            if (lane == 0) {
                tC[node] = first; tM[node] = (uint16_t)(cnt | (kind << 14));
                gi[GI_ALLOC] = first + cnt;
                st[ST_NODES] += (unsigned)cnt;
            }
        }
    }
    if (is_root) {
        if (lane == 0) { gi[GI_PHASE] = PH_SEARCH; gi[GI_SIMS] = 0; if (ovf) st[ST_OVF] |= (unsigned long long)ovf << 8; }
        return;
    }
    // ---- leaf: value = -v (mcts.py:150), backup
    const int depth = __builtin_amdgcn_readfirstlane(gi[GI_PDEPTH]);
    wave_sync_mem();
    wave_backup(tN, tW, path, depth, -v_net);
    sims_done += 1;
    if (lane == 0) {
        gi[GI_SIMS] = sims_done;
        gi[GI_PHASE] = (manual && sims_done >= E.cfg.num_simulations) ? PH_HOLD : PH_SEARCH;
        st[ST_SIMS] += 1;
        if (ovf) st[ST_OVF] |= (unsigned long long)ovf << 8;
    }
}

__global__ void k_init(Dev E) {
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= E.cfg.n_games) return;
    int32_t *gi = E.gi + (size_t)slot * GI_N;
    for (int i = 0; i < GI_N; ++i) gi[i] = 0;
    gi[GI_PHASE] = E.cfg.manual_moves == 1 ? PH_HOLD : PH_NEWGAME;
    gi[GI_SIDE] = 1;
    if (E.cfg.start_stagger && E.cfg.manual_moves == 0)
        gi[GI_DELAY] = (int)(philox_u64(E.cfg.seed, (uint32_t)E.cfg.rank, (uint32_t)slot, 7u, 0u, 0u) % (uint64_t)(E.cfg.num_simulations + 1));
    unsigned long long *st = E.stats + (size_t)slot * ST_N;
    for (int i = 0; i < ST_N; ++i) st[i] = 0;
    if (slot == 0) { E.cnt[0] = 0; E.cnt[1] = 0; *E.started = 0; }
}

// Column sums of the per-slot counters [G][ST_N] (OR for the overflow word): each 256-thread block sweeps slot rows
// (8 rows x 32 columns per pass, 256 contiguous bytes per row), folds its eight partial rows through LDS and adds the
// result to the zeroed output with one atomic per column.
__global__ __launch_bounds__(256) void k_reduce_stats(Dev E, unsigned long long *out) {
    __shared__ unsigned long long part[8][ST_N];
    const int col = threadIdx.x & (ST_N - 1), row = threadIdx.x >> 5;
    unsigned long long acc = 0;
    for (int s = blockIdx.x * 8 + row; s < E.cfg.n_games; s += gridDim.x * 8) {
        const unsigned long long v = E.stats[(size_t)s * ST_N + col];
        acc = (col == ST_OVF) ? (acc | v) : (acc + v);
    }
    part[row][col] = acc;
    __syncthreads();
    if (row == 0) {
#pragma unroll
        for (int r = 1; r < 8; ++r) acc = (col == ST_OVF) ? (acc | part[r][col]) : (acc + part[r][col]);
        if (col == ST_OVF) atomicOr(&out[col], acc); else atomicAdd(&out[col], acc);
    }
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
    size_t off[32];
    size_t total;
    int node_cap, path_cap, stage_cap;
};

Layout make_layout(const xq_engine_config *c) {
    Layout l;
    memset(&l, 0, sizeof(l));
    const size_t G = (size_t)c->n_games, S = (size_t)c->num_simulations;
    l.node_cap = (int)(1 + (S + 1) * XQ_MAXM);
    l.path_cap = (int)(S + 2);
    int sc = c->max_game_length < 200 ? c->max_game_length : 200;
    if (sc < 1) sc = 1;
    l.stage_cap = c->manual_moves ? 1 : sc + 1;
    size_t o = 0;
    auto put = [&](int id, size_t bytes) { l.off[id] = o; o = align_up(o + bytes); };
    put(P_BOARD, G * XQ_BS);
    put(P_HIST, G * XQ_HIST * XQ_BS);
    put(P_GI, G * GI_N * 4);
    put(P_RESIGN, G * 16 * 8);
    put(P_PMOVES, G * XQ_MAXM * 2);
    put(P_PATH, G * (size_t)l.path_cap * 4);
    put(P_TN, G * (size_t)l.node_cap * 4);
    put(P_TW, G * (size_t)l.node_cap * 8);
    put(P_TP, G * (size_t)l.node_cap * 4);
    put(P_TA, G * (size_t)l.node_cap * 2);
    put(P_TC, G * (size_t)l.node_cap * 4);
    put(P_TM, G * (size_t)l.node_cap * 2);
    put(P_ROOTP, G * XQ_MAXM * 8);
    put(P_STAGE, G * (size_t)l.stage_cap * XQ_SAMPLE_BYTES);
    put(P_OUTS, (size_t)(c->max_out_samples > 0 ? c->max_out_samples : 1) * XQ_SAMPLE_BYTES);
    put(P_OUTR, (size_t)(c->max_out_results > 0 ? c->max_out_results : 1) * XQ_RESULT_BYTES);
    put(P_CNT, 64);
    put(P_STATS, G * ST_N * 8);
    put(P_SQRT, (S + 2) * 8);
    put(P_MNOISE, G * XQ_MAXM * 8);
    put(P_STATSUM, ST_N * 8);
    put(P_REQ, G * 4);
    l.total = o;
    return l;
}

bool config_ok(const xq_engine_config *c) {
    return c && c->n_games > 0 && c->num_simulations > 0 && c->num_simulations < 16000 && c->resign_check_steps >= 1 &&
           c->resign_check_steps <= 16 && c->random_opening_moves >= 0 && c->late_temperature > 0.0 && c->inject_len >= 0;
}

}  // namespace

extern "C" {

size_t xq_engine_workspace_bytes(const xq_engine_config *cfg) {
    if (!config_ok(cfg)) return 0;
    return make_layout(cfg).total;
}

int xq_engine_init(xq_engine *eng, const xq_engine_config *cfg, void *ws, size_t ws_bytes, const uint64_t *dev_inject,
                   void *stream) {
    if (!eng || !config_ok(cfg) || !ws || ((uintptr_t)ws & 255)) return XQ_ERR_ARG;
    if (cfg->inject_len > 0 && !dev_inject) return XQ_ERR_ARG;
    const Layout l = make_layout(cfg);
    if (ws_bytes < l.total) return XQ_ERR_WORKSPACE;
    memset(eng, 0, sizeof(*eng));
    eng->cfg = *cfg;
    eng->node_cap = l.node_cap; eng->path_cap = l.path_cap; eng->stage_cap = l.stage_cap;
    for (int i = 0; i < 32; ++i) eng->p[i] = (char *)ws + l.off[i];
    eng->p[P_INJECT] = (void *)dev_inject;
    hipStream_t s = (hipStream_t)stream;
    // small state is zeroed; tree arenas need no clearing (nodes are initialised when created)
    XQ_TRY(hipMemsetAsync(eng->p[P_BOARD], 0, l.off[P_PATH] - l.off[P_BOARD], s));
    XQ_TRY(hipMemsetAsync(eng->p[P_ROOTP], 0, (size_t)cfg->n_games * XQ_MAXM * 8, s));
    XQ_TRY(hipMemsetAsync(eng->p[P_MNOISE], 0, (size_t)cfg->n_games * XQ_MAXM * 8, s));
    XQ_TRY(hipMemsetAsync(eng->p[P_REQ], 0, (size_t)cfg->n_games * 4, s));
    {
        const int n = cfg->num_simulations + 2;
        double *tab = (double *)malloc(sizeof(double) * n);
        if (!tab) return XQ_ERR_ARG;
        for (int i = 0; i < n; ++i) tab[i] = sqrt((double)i);   // math.sqrt(visit_count), mcts.py:49
        const int rc = xq::check(hipMemcpyAsync(eng->p[P_SQRT], tab, sizeof(double) * n, hipMemcpyHostToDevice, s));
        if (rc == XQ_OK) (void)hipStreamSynchronize(s);
        free(tab);
        if (rc != XQ_OK) return rc;
    }
    const Dev d = make_dev(eng);
    hipLaunchKernelGGL(k_init, dim3((cfg->n_games + 255) / 256), dim3(256), 0, s, d);
    return launch_status();
}

int xq_engine_select(const xq_engine *eng, float *dev_nn_input, void *stream) {
    if (!eng || !dev_nn_input) return XQ_ERR_ARG;
    const Dev d = make_dev(eng);
    hipLaunchKernelGGL(k_select, dim3((eng->cfg.n_games + WAVES_PER_WG - 1) / WAVES_PER_WG), dim3(64 * WAVES_PER_WG), 0,
                       (hipStream_t)stream, d, dev_nn_input);
    return launch_status();
}

int xq_engine_expand(const xq_engine *eng, const float *dev_policy, const float *dev_value, int policy_is_probs,
                     void *stream) {
    if (!eng || !dev_policy || !dev_value) return XQ_ERR_ARG;
    const Dev d = make_dev(eng);
    hipLaunchKernelGGL(k_expand, dim3(eng->cfg.n_games), dim3(64), 0, (hipStream_t)stream, d, dev_policy, dev_value,
                       policy_is_probs ? 1 : 0);
    return launch_status();
}

int xq_engine_requests(const xq_engine *eng, const uint16_t **dev_moves, const int32_t **dev_counts) {
    if (!eng || !dev_moves || !dev_counts) return XQ_ERR_ARG;
    *dev_moves = (const uint16_t *)eng->p[P_PMOVES];
    *dev_counts = (const int32_t *)eng->p[P_REQ];
    return XQ_OK;
}

int xq_engine_expand_legal(const xq_engine *eng, const float *dev_legal_logits, const float *dev_value, void *stream) {
    if (!eng || !dev_legal_logits || !dev_value) return XQ_ERR_ARG;
    const Dev d = make_dev(eng);
    hipLaunchKernelGGL(k_expand, dim3(eng->cfg.n_games), dim3(64), 0, (hipStream_t)stream, d, dev_legal_logits, dev_value, 2);
    return launch_status();
}

int xq_engine_stats_read(const xq_engine *eng, xq_engine_stats *host_out, void *stream) {
    if (!eng || !host_out) return XQ_ERR_ARG;
    const Dev d = make_dev(eng);
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *sum = (unsigned long long *)eng->p[P_STATSUM];
    XQ_TRY(hipMemsetAsync(sum, 0, ST_N * sizeof(unsigned long long), s));
    int blocks = (eng->cfg.n_games + 63) / 64;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(k_reduce_stats, dim3(blocks), dim3(256), 0, s, d, sum);
    int rc = launch_status();
    if (rc != XQ_OK) return rc;
    unsigned long long h[ST_N];
    XQ_TRY(hipMemcpyAsync(h, sum, sizeof(h), hipMemcpyDeviceToHost, s));
    XQ_TRY(hipStreamSynchronize(s));
    memset(host_out, 0, sizeof(*host_out));
    host_out->sims = h[ST_SIMS]; host_out->terminal_sims = h[ST_TERM]; host_out->leaf_evals = h[ST_LEAF];
    host_out->root_evals = h[ST_ROOT]; host_out->moves_played = h[ST_MOVES]; host_out->games_finished = h[ST_GAMES];
    host_out->red_wins = h[ST_RED]; host_out->black_wins = h[ST_BLACK]; host_out->draws = h[ST_DRAW];
    host_out->plies_finished = h[ST_PLIES]; host_out->nodes_created = h[ST_NODES]; host_out->depth_sum = h[ST_DEPTH];
    host_out->children_scanned = h[ST_SCAN]; host_out->resigns = h[ST_RESIGN]; host_out->samples_written = h[ST_SAMP];
    host_out->samples_dropped = h[ST_DROP]; host_out->overflow = h[ST_OVF]; host_out->games_started = h[ST_STARTED];
    return h[ST_OVF] ? XQ_ERR_OVERFLOW : XQ_OK;
}

int xq_engine_drain(const xq_engine *eng, void *host_samples, int max_samples, int *n_samples, void *host_results,
                    int max_results, int *n_results, void *stream) {
    if (!eng || !n_samples || !n_results) return XQ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned cnt[2];
    XQ_TRY(hipStreamSynchronize(s));
    XQ_TRY(hipMemcpy(cnt, eng->p[P_CNT], sizeof(cnt), hipMemcpyDeviceToHost));
    unsigned ns = cnt[0] < (unsigned)eng->cfg.max_out_samples ? cnt[0] : (unsigned)eng->cfg.max_out_samples;
    unsigned nr = cnt[1] < (unsigned)eng->cfg.max_out_results ? cnt[1] : (unsigned)eng->cfg.max_out_results;
    if ((int)ns > max_samples || (int)nr > max_results) {   // caller's buffers too small: report sizes, keep the data
        *n_samples = (int)ns; *n_results = (int)nr;
        return XQ_ERR_ARG;
    }
    if (ns && host_samples) XQ_TRY(hipMemcpy(host_samples, eng->p[P_OUTS], (size_t)ns * XQ_SAMPLE_BYTES, hipMemcpyDeviceToHost));
    if (nr && host_results) XQ_TRY(hipMemcpy(host_results, eng->p[P_OUTR], (size_t)nr * XQ_RESULT_BYTES, hipMemcpyDeviceToHost));
    XQ_TRY(hipMemset(eng->p[P_CNT], 0, 8));
    *n_samples = (int)ns; *n_results = (int)nr;
    return XQ_OK;
}

int xq_engine_drain_device(const xq_engine *eng, void *dev_samples, int max_samples, int *n_samples, void *dev_results,
                           int max_results, int *n_results, void *stream) {
    if (!eng || !n_samples || !n_results) return XQ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    unsigned cnt[2];
    XQ_TRY(hipStreamSynchronize(s));
    XQ_TRY(hipMemcpy(cnt, eng->p[P_CNT], sizeof(cnt), hipMemcpyDeviceToHost));
    const unsigned ns = cnt[0] < (unsigned)eng->cfg.max_out_samples ? cnt[0] : (unsigned)eng->cfg.max_out_samples;
    const unsigned nr = cnt[1] < (unsigned)eng->cfg.max_out_results ? cnt[1] : (unsigned)eng->cfg.max_out_results;
    *n_samples = (int)ns; *n_results = (int)nr;
    if (!dev_samples && !dev_results) return XQ_OK;          // size query: nothing is consumed
    if ((int)ns > max_samples || (int)nr > max_results || (ns && !dev_samples) || (nr && !dev_results)) return XQ_ERR_ARG;
    if (ns) XQ_TRY(hipMemcpyAsync(dev_samples, eng->p[P_OUTS], (size_t)ns * XQ_SAMPLE_BYTES, hipMemcpyDeviceToDevice, s));
    if (nr) XQ_TRY(hipMemcpyAsync(dev_results, eng->p[P_OUTR], (size_t)nr * XQ_RESULT_BYTES, hipMemcpyDeviceToDevice, s));
    XQ_TRY(hipMemsetAsync(eng->p[P_CNT], 0, 8, s));
    XQ_TRY(hipStreamSynchronize(s));
    return XQ_OK;
}

int xq_engine_set_position(const xq_engine *eng, int slot, const int8_t *host_board, int side, int move_count,
                           int no_capture, const int8_t *host_hist12, const double *host_noise, void *stream) {
    if (!eng || !host_board || slot < 0 || slot >= eng->cfg.n_games || (side != 1 && side != -1) || move_count < 0)
        return XQ_ERR_ARG;
    XQ_TRY(hipStreamSynchronize((hipStream_t)stream));
    int8_t b[XQ_BS];
    memset(b, 0, sizeof(b));
    memcpy(b, host_board, 90);
    XQ_TRY(hipMemcpy((char *)eng->p[P_BOARD] + (size_t)slot * XQ_BS, b, XQ_BS, hipMemcpyHostToDevice));
    int8_t ring[XQ_HIST][XQ_BS];
    memset(ring, 0, sizeof(ring));
    const int k = move_count < XQ_HIST ? move_count : XQ_HIST;
    if (k > 0 && !host_hist12) return XQ_ERR_ARG;
    for (int e = 0; e < k; ++e) {               // entry e (oldest first) is the pre-move board of ply mc-k+e
        const int ply = move_count - k + e;
        memcpy(ring[ply % XQ_HIST], host_hist12 + (size_t)e * 90, 90);
    }
    XQ_TRY(hipMemcpy((char *)eng->p[P_HIST] + (size_t)slot * XQ_HIST * XQ_BS, ring, sizeof(ring), hipMemcpyHostToDevice));
    int32_t gi[GI_N];
    memset(gi, 0, sizeof(gi));
    gi[GI_SIDE] = side; gi[GI_MC] = move_count; gi[GI_NOCAP] = no_capture; gi[GI_PHASE] = PH_NEWPOS;
    gi[GI_MANNOISE] = host_noise ? 1 : 0;
    XQ_TRY(hipMemcpy((char *)eng->p[P_GI] + (size_t)slot * GI_N * 4, gi, sizeof(gi), hipMemcpyHostToDevice));
    if (host_noise)
        XQ_TRY(hipMemcpy((char *)eng->p[P_MNOISE] + (size_t)slot * XQ_MAXM * 8, host_noise, XQ_MAXM * 8, hipMemcpyHostToDevice));
    return XQ_OK;
}

int xq_engine_read_root(const xq_engine *eng, int slot, uint16_t *actions, int32_t *visits, double *total_value,
                        double *prior, int *prior_kind, int32_t *root_visits, int32_t *sims_done, void *stream) {
    if (!eng || slot < 0 || slot >= eng->cfg.n_games || !actions || !visits || !total_value || !prior) return XQ_ERR_ARG;
    XQ_TRY(hipStreamSynchronize((hipStream_t)stream));
    const size_t nb = (size_t)slot * eng->node_cap;
    uint16_t m; int32_t first, rn; int32_t gi[GI_N];
    XQ_TRY(hipMemcpy(&m, (uint16_t *)eng->p[P_TM] + nb, 2, hipMemcpyDeviceToHost));
    XQ_TRY(hipMemcpy(&first, (int32_t *)eng->p[P_TC] + nb, 4, hipMemcpyDeviceToHost));
    XQ_TRY(hipMemcpy(&rn, (int32_t *)eng->p[P_TN] + nb, 4, hipMemcpyDeviceToHost));
    XQ_TRY(hipMemcpy(gi, (char *)eng->p[P_GI] + (size_t)slot * GI_N * 4, sizeof(gi), hipMemcpyDeviceToHost));
    const int n = m & 0x3FFF, kind = m >> 14;
    if (root_visits) *root_visits = rn;
    if (sims_done) *sims_done = gi[GI_SIMS];
    if (prior_kind) *prior_kind = kind == 0 ? 0 : 1;
    if (n == 0) return 0;
    float pf[XQ_MAXM];
    XQ_TRY(hipMemcpy(actions, (uint16_t *)eng->p[P_TA] + nb + first, (size_t)n * 2, hipMemcpyDeviceToHost));
    XQ_TRY(hipMemcpy(visits, (int32_t *)eng->p[P_TN] + nb + first, (size_t)n * 4, hipMemcpyDeviceToHost));
    XQ_TRY(hipMemcpy(total_value, (double *)eng->p[P_TW] + nb + first, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (kind == 1) {
        XQ_TRY(hipMemcpy(prior, (double *)eng->p[P_ROOTP] + (size_t)slot * XQ_MAXM, (size_t)n * 8, hipMemcpyDeviceToHost));
    } else if (kind == 2) {
        for (int i = 0; i < n; ++i) prior[i] = 1.0 / (double)n;
    } else {
        XQ_TRY(hipMemcpy(pf, (float *)eng->p[P_TP] + nb + first, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) prior[i] = (double)pf[i];
    }
    return n;
}

}  // extern "C"

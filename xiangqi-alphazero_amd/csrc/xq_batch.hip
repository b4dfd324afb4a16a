// xq_batch.hip -- stateless batched rules kernels behind the B1 plug point (include/xq_hip.h).
// One 64-lane wavefront per position, four positions per 256-thread workgroup (LDS carved per wave, no workgroup
// barrier: lane-to-lane traffic of one wave through LDS only needs program order, see xq_rules.cuh).  Integer/LDS-bound work: 90 bytes in, <= 2*L bytes out per position.
#include "xq_common.h"
#include "xq_rules.cuh"

#pragma clang fp contract(off)

namespace xq {
thread_local hipError_t g_last_error = hipSuccess;
}

using namespace xq;

// four positions per 256-thread workgroup, one per wave; LDS carved per wave, no workgroup barrier (see xq_rules.cuh)
constexpr int WPW = 4;
#define XQ_WAVE_ITEM(n)                                             \
    const int wv = (int)(threadIdx.x >> 6);                         \
    const int i = blockIdx.x * WPW + wv;                            \
    if (i >= (n)) return;

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WPW) void k_movegen(const int8_t *__restrict__ boards, const int8_t *__restrict__ side,
                                                int n, uint16_t *__restrict__ moves, uint16_t *__restrict__ counts,
                                                uint8_t *__restrict__ in_chk, uint8_t *__restrict__ status) {
    __shared__ __attribute__((aligned(16))) int8_t s_boards[WPW][XQ_BS];
    __shared__ MoveGenLds s_mgs[WPW];
    __shared__ uint16_t s_outs[WPW][XQ_MAXM];
    XQ_WAVE_ITEM(n)
    int8_t *s_board = s_boards[wv];
    const int lane = lane_id();
    wave_load_board(boards + (size_t)i * 90, s_board);
    wave_sync();
    const int player = side[i];
    int ovf = 0;
    const int cnt = wave_movegen(s_board, player, s_mgs[wv], s_outs[wv], &ovf);
    uint16_t *dst = moves + (size_t)i * XQ_MAXM;
    for (int j = lane; j < cnt; j += 64) dst[j] = s_outs[wv][j];
    const bool chk = in_chk ? wave_in_check(s_board, player) : false;    // probes dealt to lanes, two ballots
    if (lane == 0) {
        counts[i] = (uint16_t)cnt;
        if (in_chk) in_chk[i] = chk ? 1 : 0;
        if (status) status[i] = (uint8_t)ovf;
    }
}

__global__ __launch_bounds__(64 * WPW) void k_attack_map(const int8_t *__restrict__ boards, int n, uint8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) int8_t s_boards[WPW][XQ_BS];
    XQ_WAVE_ITEM(n)
    int8_t *s_board = s_boards[wv];
    const int lane = lane_id();
    wave_load_board(boards + (size_t)i * 90, s_board);
    wave_sync();
    const VMove none{-1, -1, 0};
    for (int e = lane; e < 180; e += 64) {
        const int by = e < 90 ? 1 : -1;
        const int sq = e < 90 ? e : e - 90;
        out[(size_t)i * 180 + e] = is_attacked(s_board, none, sq / 9, sq % 9, by) ? 1 : 0;
    }
}

__global__ __launch_bounds__(64 * WPW) void k_find_king(const int8_t *__restrict__ boards, int n, int16_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) int8_t s_boards[WPW][XQ_BS];
    XQ_WAVE_ITEM(n)
    int8_t *s_board = s_boards[wv];
    wave_load_board(boards + (size_t)i * 90, s_board);
    wave_sync();
    const VMove none{-1, -1, 0};
    const int lane = lane_id();
    if (lane < 2) out[(size_t)i * 2 + lane] = (int16_t)find_king(s_board, none, lane == 0 ? 1 : -1);
}

__global__ __launch_bounds__(64 * WPW) void k_encode(const int8_t *__restrict__ boards, const int8_t *__restrict__ side, int n,
                                               float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) int8_t s_boards[WPW][XQ_BS];
    XQ_WAVE_ITEM(n)
    int8_t *s_board = s_boards[wv];
    wave_load_board(boards + (size_t)i * 90, s_board);
    wave_sync();
    wave_encode(s_board, side[i], out + (size_t)i * XQ_STATE_FLOATS);
}

__global__ __launch_bounds__(64 * WPW) void k_material(const int8_t *__restrict__ boards, int n, int32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) int8_t s_boards[WPW][XQ_BS];
    XQ_WAVE_ITEM(n)
    int8_t *s_board = s_boards[wv];
    wave_load_board(boards + (size_t)i * 90, s_board);
    wave_sync();
    int red, black;
    wave_material(s_board, red, black);
    if (lane_id() == 0) { out[(size_t)i * 2] = red; out[(size_t)i * 2 + 1] = black; }
}

// thread per (child, byte): pure copy with two patched bytes
__global__ void k_apply_moves(const int8_t *__restrict__ boards, const int8_t *__restrict__ side,
                              const uint32_t *__restrict__ parent, const uint16_t *__restrict__ action, int m,
                              int8_t *__restrict__ out_boards, int8_t *__restrict__ out_side) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long j = gid / 96;
    const int b = (int)(gid - j * 96);
    if (j >= m || b > 90) return;
    const uint32_t p = parent[j];
    if (b == 90) { out_side[j] = (int8_t)-side[p]; return; }
    const int a = action[j], from = a / 90, to = a - from * 90;
    const int8_t *src = boards + (size_t)p * 90;
    int8_t v = src[b];
    if (b == to) v = src[from];
    if (b == from) v = 0;
    out_boards[(size_t)j * 90 + b] = v;
}

// game.py:565-616 for independent states; hist = last min(12,mc) pre-move boards, oldest first
__global__ __launch_bounds__(64 * WPW) void k_game_over(const int8_t *__restrict__ boards, const int8_t *__restrict__ side,
                                                  const int32_t *__restrict__ move_count,
                                                  const int32_t *__restrict__ no_capture, const int8_t *__restrict__ hist,
                                                  int n, int8_t *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) int8_t s_boards[WPW][XQ_BS];
    __shared__ MoveGenLds s_mgs[WPW];
    __shared__ uint16_t s_outs[WPW][XQ_MAXM];
    XQ_WAVE_ITEM(n)
    int8_t *s_board = s_boards[wv];
    const int lane = lane_id();
    wave_load_board(boards + (size_t)i * 90, s_board);
    wave_sync();
    const int player = side[i], mc = move_count[i], nc = no_capture[i];
    const VMove none{-1, -1, 0};
    int done = 0, winner = 2;
    if (find_king(s_board, none, 1) < 0) { done = 1; winner = -1; }
    else if (find_king(s_board, none, -1) < 0) { done = 1; winner = 1; }
    else {
        int ovf = 0;
        const int cnt = wave_movegen(s_board, player, s_mgs[wv], s_outs[wv], &ovf);
        if (cnt == 0) { done = 1; winner = -player; }
        else if (nc >= 120) { done = 1; winner = 0; }
        else if (mc >= 200) {
            int red, black;
            wave_material(s_board, red, black);
            const int diff = red - black;
            done = 1; winner = diff > 30 ? 1 : (diff < -30 ? -1 : 0);
        } else if (mc >= 6) {
            const int k = mc < XQ_HIST ? mc : XQ_HIST;
            int rep = 0;
            for (int e = 0; e < k; ++e) {
                const int8_t *h = hist + ((size_t)i * XQ_HIST + e) * 90;
                bool diff = h[lane] != s_board[lane];
                if (lane + 64 < 90) diff = diff || (h[lane + 64] != s_board[lane + 64]);
                if (__ballot(diff) == 0ull) ++rep;
            }
            if (rep >= 3) { done = 1; winner = 0; }
        }
    }
    if (lane == 0) { out[(size_t)i * 2] = (int8_t)done; out[(size_t)i * 2 + 1] = (int8_t)winner; }
}

// ---------------------------------------------------------------------------------------------
extern "C" {

const char *xq_version(void) { return "xq_hip 0.1.0 (gfx950)"; }

const char *xq_last_hip_error(void) { return g_last_error == hipSuccess ? "" : hipGetErrorString(g_last_error); }

int xq_movegen_batch(const int8_t *dev_boards, const int8_t *dev_side, int n, uint16_t *dev_moves, uint16_t *dev_counts,
                     uint8_t *dev_in_check, uint8_t *dev_status, void *stream) {
    if (n < 0 || (n > 0 && (!dev_boards || !dev_side || !dev_moves || !dev_counts))) return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_movegen, dim3((n + WPW - 1) / WPW), dim3(64 * WPW), 0, (hipStream_t)stream, dev_boards, dev_side, n, dev_moves,
                       dev_counts, dev_in_check, dev_status);
    return launch_status();
}

int xq_attack_map_batch(const int8_t *dev_boards, int n, uint8_t *dev_out, void *stream) {
    if (n < 0 || (n > 0 && (!dev_boards || !dev_out))) return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_attack_map, dim3((n + WPW - 1) / WPW), dim3(64 * WPW), 0, (hipStream_t)stream, dev_boards, n, dev_out);
    return launch_status();
}

int xq_find_king_batch(const int8_t *dev_boards, int n, int16_t *dev_out, void *stream) {
    if (n < 0 || (n > 0 && (!dev_boards || !dev_out))) return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_find_king, dim3((n + WPW - 1) / WPW), dim3(64 * WPW), 0, (hipStream_t)stream, dev_boards, n, dev_out);
    return launch_status();
}

int xq_encode_batch(const int8_t *dev_boards, const int8_t *dev_side, int n, float *dev_out, void *stream) {
    if (n < 0 || (n > 0 && (!dev_boards || !dev_side || !dev_out))) return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_encode, dim3((n + WPW - 1) / WPW), dim3(64 * WPW), 0, (hipStream_t)stream, dev_boards, dev_side, n, dev_out);
    return launch_status();
}

int xq_material_batch(const int8_t *dev_boards, int n, int32_t *dev_out, void *stream) {
    if (n < 0 || (n > 0 && (!dev_boards || !dev_out))) return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_material, dim3((n + WPW - 1) / WPW), dim3(64 * WPW), 0, (hipStream_t)stream, dev_boards, n, dev_out);
    return launch_status();
}

int xq_apply_moves_batch(const int8_t *dev_boards, const int8_t *dev_side, const uint32_t *dev_parent,
                         const uint16_t *dev_action, int m, int8_t *dev_out_boards, int8_t *dev_out_side, void *stream) {
    if (m < 0 || (m > 0 && (!dev_boards || !dev_side || !dev_parent || !dev_action || !dev_out_boards || !dev_out_side)))
        return XQ_ERR_ARG;
    if (m == 0) return XQ_OK;
    const long long threads = (long long)m * 96;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(k_apply_moves, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dev_boards, dev_side, dev_parent,
                       dev_action, m, dev_out_boards, dev_out_side);
    return launch_status();
}

int xq_game_over_batch(const int8_t *dev_boards, const int8_t *dev_side, const int32_t *dev_move_count,
                       const int32_t *dev_no_capture, const int8_t *dev_hist, int n, int8_t *dev_out, void *stream) {
    if (n < 0 || (n > 0 && (!dev_boards || !dev_side || !dev_move_count || !dev_no_capture || !dev_hist || !dev_out)))
        return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_game_over, dim3((n + WPW - 1) / WPW), dim3(64 * WPW), 0, (hipStream_t)stream, dev_boards, dev_side, dev_move_count,
                       dev_no_capture, dev_hist, n, dev_out);
    return launch_status();
}

}  // extern "C"

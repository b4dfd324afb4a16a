/*
 * xq_hip.h -- C ABI of libxq_hip.so: the MI355X (gfx950) self-play hot path of
 * wenjunyang/xiangqi-alphazero as hand-written HIP kernels.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - every pointer named dev_* / ws is DEVICE memory owned by the caller; nothing is allocated or
 *     freed behind the caller's back.  `stream` is a hipStream_t passed as void*.
 *   - return value: XQ_OK (0) or a negative XQ_ERR_* code; no exceptions cross the ABI; all entry
 *     points are re-entrant (no global mutable state).  Launches are asynchronous on `stream`
 *     unless the function says it synchronises.
 *   - boards are the reference's layout: int8[10][9] row-major (90 bytes), red positive, black
 *     negative, pieces 1..7 = king advisor bishop knight rook cannon pawn (training/game.py:49-65);
 *     player/side is +1 (red) or -1 (black); an action id is (from_sq*90 + to_sq),
 *     sq = row*9+col (training/game.py:112-121).
 *
 * Each entry point cites the reference interface it replaces (paths relative to the reference root).
 */
#ifndef XQ_HIP_H
#define XQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XQ_OK 0
#define XQ_ERR_ARG (-1)      /* bad argument (null pointer, non-positive size, ...) */
#define XQ_ERR_HIP (-2)      /* a HIP runtime call / kernel launch failed; see xq_last_hip_error() */
#define XQ_ERR_WORKSPACE (-3) /* workspace too small */
#define XQ_ERR_OVERFLOW (-4) /* a device-side capacity was exceeded (reported by xq_engine_stats) */

#define XQ_SQUARES 90
#define XQ_ACTION_SPACE 8100
#define XQ_STATE_FLOATS 1350  /* 15 planes x 90, training/game.py:627 */
#define XQ_MAXM 128           /* legal moves kept per position (reference buffer: 200, observed max 69) */
#define XQ_SAMPLE_BYTES 640
#define XQ_RESULT_BYTES 16

const char *xq_version(void);
/* hipGetErrorString of the last failing HIP call on this thread ("" if none). */
const char *xq_last_hip_error(void);

/* =====================================================================================
 * B1 -- rules engine plug point.  Replaces training/cython_engine/game_core.pyx:493-569
 * (cy_generate_legal_moves, cy_is_in_check, cy_find_king, cy_is_attacked, cy_has_legal_moves)
 * and their Python twins training/game.py:176-265, 297-521, 552-563, 618-640, batched over n boards.
 * ===================================================================================== */

/* cy_generate_legal_moves + cy_is_in_check(board, side) + cy_has_legal_moves for n boards.
 * dev_moves[i][0..counts[i]) = action ids in the reference's emission order.
 * dev_in_check may be NULL.  dev_status (may be NULL): per board 0, or 1 if more than XQ_MAXM
 * legal moves / more than 256 pseudo-legal candidates were met (list truncated). */
int xq_movegen_batch(const int8_t *dev_boards, const int8_t *dev_side, int n, uint16_t *dev_moves,
                     uint16_t *dev_counts, uint8_t *dev_in_check, uint8_t *dev_status, void *stream);

/* cy_is_attacked(board, r, c, by) for every square and both attackers:
 * dev_out[i][0][sq] = attacked by red, dev_out[i][1][sq] = attacked by black (uint8 0/1). */
int xq_attack_map_batch(const int8_t *dev_boards, int n, uint8_t *dev_out, void *stream);

/* cy_find_king for both sides: dev_out[i][0] = red king square or -1, dev_out[i][1] = black. */
int xq_find_king_batch(const int8_t *dev_boards, int n, int16_t *dev_out, void *stream);

/* XiangqiGame.get_state_for_nn (training/game.py:618-640): dev_out[i] = float32[15][10][9]. */
int xq_encode_batch(const int8_t *dev_boards, const int8_t *dev_side, int n, float *dev_out, void *stream);

/* XiangqiGame.get_material_score (training/game.py:552-563): dev_out[i][0] red, [i][1] black. */
int xq_material_batch(const int8_t *dev_boards, int n, int32_t *dev_out, void *stream);

/* XiangqiGame.make_move on copies (training/game.py:528-550, board part): child j is
 * dev_boards[parent[j]] with action[j] applied; side flipped.  Used for perft-style expansion. */
int xq_apply_moves_batch(const int8_t *dev_boards, const int8_t *dev_side, const uint32_t *dev_parent,
                         const uint16_t *dev_action, int m, int8_t *dev_out_boards, int8_t *dev_out_side,
                         void *stream);

/* XiangqiGame.is_game_over (training/game.py:565-616) for n independent game states.
 * dev_hist[i] = the last min(12, move_count) pre-move boards (int8[12][90], oldest first, rest ignored).
 * dev_out[i][0] = done (0/1), dev_out[i][1] = winner (+1/-1/0, or 2 when not done). */
int xq_game_over_batch(const int8_t *dev_boards, const int8_t *dev_side, const int32_t *dev_move_count,
                       const int32_t *dev_no_capture, const int8_t *dev_hist, int n, int8_t *dev_out,
                       void *stream);

/* =====================================================================================
 * B3 -- self-play operator.  Replaces the search + game loop that the reference fans out over
 * processes: training/mcts.py:21-206 (MCTSNode, MCTS.search), training/parallel_selfplay.py:42-134
 * (_play_one_game) and the per-evaluation IPC of training/inference_server.py:37-497.
 *
 * One engine = n_games concurrent game slots resident on one GPU.  A *step* is
 *     xq_engine_select   every slot advances (finishing moves/games, running simulations whose leaves are
 *                        terminal) until it needs ONE network evaluation, and writes that position's
 *                        15x10x9 planes into dev_nn_input[slot]
 *     <evaluator>        B2: any batched policy/value function over dev_nn_input (the ResNet)
 *     xq_engine_expand   consumes dev_policy[slot] / dev_value[slot]: root or leaf expansion with the
 *                        reference's mask-and-normalise, backup along the recorded path
 * Simulations of one game stay strictly sequential (as in mcts.py:126-153); parallelism is across games.
 * ===================================================================================== */

typedef struct xq_engine_config {
    int32_t n_games;               /* concurrent slots G */
    int32_t num_simulations;       /* MCTS simulations per move (TrainingConfig.num_simulations) */
    double  c_puct;                /* 1.5 */
    int32_t temperature_threshold; /* plies with T=1.0, then late_temperature (parallel_selfplay.py:92) */
    int32_t max_game_length;
    int32_t random_opening_moves;
    int32_t enable_resign;
    double  resign_threshold;
    int32_t resign_check_steps;    /* <= 16 */
    int32_t add_noise;             /* Dirichlet noise at the root (mcts.py:117-121); self-play: 1 */
    double  dirichlet_alpha;       /* 0.3 */
    double  noise_eps;             /* 0.25 */
    double  late_temperature;      /* 0.3 */
    uint64_t seed;                 /* Philox key (run seed); rank goes into the key as well */
    int32_t rank;
    int32_t inject_len;            /* 0: device RNG; >0: draws come from dev_inject (tests), per slot
                                      4 streams x inject_len raw uint64 (tests/draws.py order) */
    int64_t games_target;          /* stop starting games after this many (<=0: unlimited) */
    int32_t max_out_samples;       /* capacity of the finished-sample ring */
    int32_t max_out_results;       /* capacity of the game-result ring */
    int32_t manual_moves;          /* 0: self-play.  1: search only -- never plays the move (xq_engine_set_position +
                                      num_simulations steps, then xq_engine_read_root); MCTS.search parity / serving.
                                      2: arena games (training/train.py:453-535): no opening, no noise unless add_noise,
                                      move = first maximum of the visit counts (temperature 0), no samples, no resign,
                                      a game still running after max_game_length plies is a draw */
    int32_t start_stagger;         /* 1: slot s idles hash(s) mod (num_simulations+1) steps before its first game, so a
                                      freshly initialised engine reaches the steady-state mix of search depths */
} xq_engine_config;

/* Host-side handle: plain pointers into the caller's workspace.  Treat as opaque. */
typedef struct xq_engine {
    xq_engine_config cfg;
    int32_t node_cap, path_cap, stage_cap, pad0;
    void *p[32];
} xq_engine;

typedef struct xq_engine_stats {
    uint64_t sims;            /* completed simulations (leaf evaluated or terminal) */
    uint64_t terminal_sims;
    uint64_t leaf_evals;      /* network evaluations consumed by simulations */
    uint64_t root_evals;      /* network evaluations consumed by roots (incl. the resign probe) */
    uint64_t moves_played;
    uint64_t games_finished, red_wins, black_wins, draws;
    uint64_t plies_finished;  /* sum of move_count over finished games */
    uint64_t nodes_created;
    uint64_t depth_sum;       /* sum over simulations of descent depth */
    uint64_t children_scanned;/* sum over descents of children read by PUCT select */
    uint64_t resigns;
    uint64_t samples_written, samples_dropped;
    uint64_t overflow;        /* non-zero: a device capacity was exceeded (results invalid) */
    uint64_t games_started;
    uint64_t reserved[14];
} xq_engine_stats;

/* Bytes of device workspace the engine needs for cfg (tree arenas dominate:
 * n_games * (1 + (num_simulations+1)*XQ_MAXM) nodes * 24 B -- sized for 288 GB HBM, no per-node malloc). */
size_t xq_engine_workspace_bytes(const xq_engine_config *cfg);

/* Carves `ws` (>= workspace_bytes, 256-byte aligned) and initialises the engine: boards, history rings, per-slot state
 * words (allocation marks and RNG counters among them), request counts, root prior / injected-noise tables, ring
 * counters and the statistics are zeroed, and every slot starts a new game at its first step.  The tree arenas (N, W, P, action, first-child, meta) are NOT cleared: a slot's nodes are valid only below its
 * allocation mark (bump allocator, reset at every move), nothing in the engine reads past it, and whatever the caller's
 * buffer held before stays there -- tools that walk the arenas (`SelfPlayEngine.arena_views`) must stop at the mark.
 * dev_inject: uint64[n_games][4][inject_len] or NULL. */
int xq_engine_init(xq_engine *eng, const xq_engine_config *cfg, void *ws, size_t ws_bytes,
                   const uint64_t *dev_inject, void *stream);

int xq_engine_select(const xq_engine *eng, float *dev_nn_input /* [G][15][90] */, void *stream);

/* dev_policy[slot] = float32[8100]: network LOGITS (policy_is_probs = 0; softmax over all 8100 as
 * model.py:122 does) or already-softmaxed probabilities (policy_is_probs = 1, evaluator-plugin
 * protocol of mcts.py:157-164).  dev_value[slot] = tanh output. */
int xq_engine_expand(const xq_engine *eng, const float *dev_policy, const float *dev_value,
                     int policy_is_probs, void *stream);

/* Sparse hand-off between the engine and the evaluator (replaces the dense 8100-wide policy row of mcts.py:157-188).
 * xq_engine_requests: after xq_engine_select, *dev_moves = uint16[G][XQ_MAXM] holds the ORDERED legal moves of the
 * position each slot handed to the evaluator and *dev_counts = int32[G] their number (0: the slot asked for nothing this
 * step).  Both point into the engine's workspace and stay valid for its lifetime.
 * xq_engine_expand_legal: like xq_engine_expand, but dev_legal_logits[slot][m] is the network's logit of legal move m
 * only.  Priors = softmax over the legal logits, summed sequentially in float32 in move order and divided -- the
 * reference's softmax over all 8100 followed by mask-and-normalise (model.py:122, mcts.py:176-188), whose common factor
 * exp(max_legal - max_all)/denominator cancels; identical up to float32 rounding unless the reference's float32 softmax
 * underflows (a logit gap above ~87), where the reference degrades to denormal or uniform priors and this stays exact. */
int xq_engine_requests(const xq_engine *eng, const uint16_t **dev_moves, const int32_t **dev_counts);
int xq_engine_expand_legal(const xq_engine *eng, const float *dev_legal_logits, const float *dev_value, void *stream);

/* Synchronises `stream`, copies the counters to host. */
int xq_engine_stats_read(const xq_engine *eng, xq_engine_stats *host_out, void *stream);

/* Synchronises; copies up to max_samples finished samples (XQ_SAMPLE_BYTES each, layout xq_sample) and
 * up to max_results game results (xq_game_result) to host buffers and resets the rings. */
int xq_engine_drain(const xq_engine *eng, void *host_samples, int max_samples, int *n_samples,
                    void *host_results, int max_results, int *n_results, void *stream);

/* The same into DEVICE buffers (the samples stay on the GPU for the replay buffer / the RCCL all-gather; nothing crosses
 * PCIe).  With both buffers NULL it only reports the pending counts and consumes nothing.  Synchronises. */
int xq_engine_drain_device(const xq_engine *eng, void *dev_samples, int max_samples, int *n_samples,
                           void *dev_results, int max_results, int *n_results, void *stream);

/* Test / serving hooks (MCTS.search for a given position, mcts.py:94-155). Synchronise. */
int xq_engine_set_position(const xq_engine *eng, int slot, const int8_t *host_board, int side, int move_count,
                           int no_capture, const int8_t *host_hist12 /* int8[12][90], oldest first, last
                           min(12,move_count) valid */, const double *host_noise /* eta per legal move or NULL */,
                           void *stream);
/* Root children of `slot` after a search: returns n; arrays sized XQ_MAXM. prior_kind: 0 float32, 1 float64. */
int xq_engine_read_root(const xq_engine *eng, int slot, uint16_t *actions, int32_t *visits, double *total_value,
                        double *prior, int *prior_kind, int32_t *root_visits, int32_t *sims_done, void *stream);

/* =====================================================================================
 * B2 -- evaluator plug point helpers (training/model.py:20-36, 87-107 run over the leaf batch).
 * ===================================================================================== */

/* y = act(y + bias[c] (+ residual)) in place over a channels-last float32 tensor [rows][channels]
 * (folded BatchNorm bias + ReLU + skip connection of ResBlock.forward, model.py:30-36).
 * channels % 4 == 0, pointers 16-byte aligned; dev_residual may be NULL. */
int xq_bias_act(float *dev_y, const float *dev_bias, const float *dev_residual, long long rows, int channels,
                int relu, void *stream);

/* Input convolution (model.py:87-93: Conv2d(15, C, 3, padding=1), BatchNorm folded, ReLU) straight from the encoder's
 * planes:  y = relu(conv(planes) + bias).  Exact for any input; fast because the planes are sparse (zero inputs are skipped).
 *   dev_planes : float32[games][15][10][9] (xq_engine_select's nn_input);  dev_y : float32[games][90][channels] (NHWC);
 *   dev_wt : float32[135][channels], dev_wt[plane*9 + ky*3 + kx][co] = folded filter w[co][plane][ky][kx];
 *   dev_bias : float32[channels].  channels % 4 == 0. */
int xq_stem_conv(const float *dev_planes, const float *dev_wt, const float *dev_bias, float *dev_y, int games,
                 int channels, void *stream);

/* Both heads' 1x1 convolutions (model.py:43-62: policy Conv2d(C,32,1), value Conv2d(C,4,1), BatchNorm folded, ReLU) in one
 * pass over the tower output:  out[r][o] = relu(bias[o] + sum_c h[r][c] w[o][c]),  o < 36.
 *   dev_h : float32[rows][channels] (NHWC rows);  dev_w : float32[36][channels], rows 0-31 policy, 32-35 value;
 *   dev_bias : float32[36];  dev_p : float32[rows][32];  dev_v : float32[rows][4].  channels % 16 == 0, <= 1024. */
int xq_heads_1x1(const float *dev_h, const float *dev_w, const float *dev_bias, float *dev_p, float *dev_v,
                 long long rows, int channels, void *stream);

/* Policy head's Linear(2880, 8100) (model.py:64-71) evaluated ONLY at the ordered legal moves of each pending evaluation
 * -- what mcts.py:176-188 keeps of the 8 100 logits:  dev_out[g][m] = dev_bias[a] + <dev_feat[g], dev_w[a]>,
 * a = dev_moves[g][m], m < dev_counts[g] (counts <= 0: the game is skipped, its row is left untouched).
 *   dev_feat : float32[games][2880], the policy features in NHWC order (position-major, 32 channels) as xq_heads_1x1 writes
 *              them;  dev_w : float32[8100][2880] with the columns permuted to that order
 *              (w[a][hw*32 + c] = policy_head.4.weight[a][c*90 + hw]);  dev_bias : float32[8100];
 *   dev_moves : uint16[games][XQ_MAXM], dev_counts : int32[games] (xq_engine_requests);  dev_out : float32[games][XQ_MAXM]. */
int xq_policy_head_legal(const float *dev_feat, const float *dev_w, const float *dev_bias, const uint16_t *dev_moves,
                         const int32_t *dev_counts, int games, float *dev_out, void *stream);

/* Value head's Linear(360,128) + ReLU + Linear(128,1) + tanh (model.py:73-85) over the value features float32[games][360]
 * (NHWC order, xq_heads_1x1's dev_v):  dev_w1t : float32[360][128], w1t[hw*4 + c][j] = value_head.4.weight[j][c*90 + hw];
 * dev_b1 float32[128]; dev_w2 float32[128] = value_head.6.weight[0]; dev_b2 float32[1];  dev_value : float32[games]. */
int xq_value_head(const float *dev_vfeat, const float *dev_w1t, const float *dev_b1, const float *dev_w2,
                  const float *dev_b2, int games, float *dev_value, void *stream);

/* 3x3 convolution, stride 1, pad 1, C -> C channels (ResBlock.conv1/conv2 with BatchNorm folded, model.py:25-36)
 * as fused Winograd F(2x3,3x3) -- F(2,3) along the 10 rows, F(3,3) at the points 0, +-1, 2, inf along the 9 columns -- on
 * the fp32 MFMA:  y = act(conv(x) + bias (+ residual)).
 *   dev_x, dev_y, dev_residual : float32[batch][90][channels] (NHWC; y must not alias x or residual)
 *   dev_u : pre-transformed weights, float32[C/64][C/8][20][2][64][4] with
 *           u[cog][chunk][5p+j][quad][co][k] = s_p (G_r g G_c'^T)[p][j] for output channel 64*cog+co and input channel
 *           8*chunk+4*quad+k, g = the folded 3x3 filter (cross-correlation, as torch.nn.Conv2d), G_r the F(2,3) matrix,
 *           G_c' = diag(1/2, 1/2, 1/6, 1/6, 1) [[1,0,0],[1,1,1],[1,-1,1],[1,2,4],[0,0,1]], s_p = -1 for p = 2 and +1
 *           otherwise (the kernel forms row 2 of B_r^T d with the opposite sign);
 *           xq_wino_weight_bytes(C) = 80 C^2 bytes.  channels in {64, 128, 256, 512}; batch*90*channels*4 < 2^32. */
size_t xq_wino_weight_bytes(int channels);
int xq_wino_conv3x3(const float *dev_x, const float *dev_u, const float *dev_bias, const float *dev_residual,
                    float *dev_y, int batch, int channels, int flags, void *stream);
/* flags: bit 0 = ReLU; bit 1 = walk the batch back to front (same results; alternate it between consecutive layers so
 * that each launch first reads what the previous one wrote last, while it is still in the Infinity Cache). */
#define XQ_CONV_RELU 1
#define XQ_CONV_REVERSE 2
/* bit 2: "wide" variant -- 128 output channels per workgroup (one workgroup per CU, 320 accumulators per wave): dev_u is then
 * float32[C/128][C/8][20][2][128][4] (the same element formula with 128-channel blocks), channels in {128, 256, 512}. */
#define XQ_CONV_WIDE 4

/* The filter transform of xq_wino_conv3x3 on the device (the train step re-transforms after every optimizer step; self-play
 * transforms once per weight update and may use this or the host's float64 einsum, which give the same float32 values):
 *   dev_w : float32[C][C][3][3] (torch.nn.Conv2d.weight, train.py:376-447 trains it);  dev_u : xq_wino_weight_bytes(C) bytes in the
 *   layout above (64-channel blocks, or 128 with XQ_CONV_WIDE in flags).  With XQ_FILTER_DGRAD the filters of the DATA-GRADIENT
 *   convolution are produced, w'[co][ci][r][s] = w[ci][co][2-r][2-s]: xq_wino_conv3x3(dL/dy, u', 0-bias) is dL/dx of
 *   y = conv3x3(x, w) (stride 1, pad 1) -- what torch autograd's convolution_backward computes for ResBlock.conv1/conv2. */
#define XQ_FILTER_DGRAD 8
/* XQ_FILTER_BOTH: forward filters into dev_u[0 .. 20 C^2) and data-gradient filters into dev_u[20 C^2 .. 40 C^2) (floats) in one launch;
 * dev_u then holds 2 * xq_wino_weight_bytes(C) bytes. */
#define XQ_FILTER_BOTH 16
int xq_wino_transform_filters(const float *dev_w, float *dev_u, int channels, int flags, void *stream);

/* BatchNorm2d in TRAINING mode fused with the ReLU / skip-add around it in a ResBlock (training/model.py:20-36: bn1 + relu, bn2 + add +
 * relu; trained by training/train.py:376-447), on NHWC activations float32[rows][channels], rows = batch * 90:
 *   forward :  y = act((x - mean) * invstd * gamma + beta (+ residual)) with the batch statistics of the rows (biased variance);
 *              running_mean / running_var (nullable pair) updated as torch.nn.BatchNorm2d does (momentum, unbiased variance);
 *              save_mean / save_invstd float32[channels] are kept for the backward call; dev_batches_tracked (nullable): the module's
 *              int64 num_batches_tracked, incremented by one.  relu: 0 / 1.
 *   backward:  g = dy * (y > 0) if relu;  dbeta = sum g;  dgamma = sum g * xhat;  dx = gamma * invstd * (g - dbeta / rows - xhat * dgamma / rows);
 *              dev_dresidual (nullable) receives g, the gradient of the skip input.
 * Sums are float64 per row segment, reduced in a fixed order (deterministic).  dev_scratch: xq_bn_scratch_bytes(channels) bytes.
 * channels in {64, 128, 256, 512, 1024}; all pointers 16-byte aligned. */
size_t xq_bn_scratch_bytes(int channels);
int xq_bn_train_forward(const float *dev_x, const float *dev_residual, const float *dev_gamma, const float *dev_beta,
                        float *dev_running_mean, float *dev_running_var, float momentum, float eps, long long rows, int channels,
                        int relu, float *dev_y, float *dev_save_mean, float *dev_save_invstd, long long *dev_batches_tracked,
                        void *dev_scratch, void *stream);
int xq_bn_train_backward(const float *dev_dy, const float *dev_x, const float *dev_y, const float *dev_gamma, const float *dev_save_mean,
                         const float *dev_save_invstd, long long rows, int channels, int relu, float *dev_dx, float *dev_dresidual,
                         float *dev_dgamma, float *dev_dbeta, void *dev_scratch, void *stream);

/* Weight gradient of y = conv3x3(x, w) (stride 1, pad 1, C -> C; what torch autograd's convolution_backward returns for
 * ResBlock.conv1/conv2.weight under training/train.py:376-447), in the Winograd domain of xq_wino_conv3x3 on the fp32 MFMA:
 *   dev_x, dev_dy : float32[batch][90][channels] (NHWC);  dev_dw : float32[channels][channels][3][3] (torch's layout), overwritten;
 *   dev_scratch   : xq_wino_wgrad_scratch_bytes(batch, channels) bytes (per-split partial sums, added in a fixed order: deterministic).
 * channels in {64, 128, 256, 512}. */
size_t xq_wino_wgrad_scratch_bytes(int batch, int channels);
int xq_wino_wgrad(const float *dev_x, const float *dev_dy, float *dev_dw, void *dev_scratch, int batch, int channels, void *stream);

/* REDUCED-PRECISION throughput mode of the same convolution (never the parity path; outside the 1e-5 contract): the identical
 * fused Winograd decomposition with the 20 per-frequency products on the bf16 MFMA (operands rounded to bf16 after the float32
 * transforms, float32 accumulation, float32 activations in HBM).  Replaces nothing in the reference -- its own inference is
 * float32 (model.py:109-124); it is the "throughput mode" of SURVEY.md section 7.
 *   dev_u_bf16 : bf16[C/128][C/16][20][2][128][8], u[cog][chunk][5p+j][h][co][k] = the float32 tensor's element for output
 *                channel 128*cog+co and input channel 16*chunk+8*h+k, rounded to nearest-even;  xq_wino_weight_bytes_bf16(C) = 40 C^2.
 *   channels in {128, 256, 512}; flags: XQ_CONV_RELU, XQ_CONV_REVERSE. */
size_t xq_wino_weight_bytes_bf16(int channels);
int xq_wino_conv3x3_bf16(const float *dev_x, const void *dev_u_bf16, const float *dev_bias, const float *dev_residual,
                         float *dev_y, int batch, int channels, int flags, void *stream);

/* =====================================================================================
 * Next row (section 8f.1) -- training-batch materialisation.  Replaces SelfPlayDataset.__getitem__ + augment_data
 * (training/train.py:114-151) and _augment_data (training/parallel_selfplay.py:137-151) for a batch drawn from a
 * device-resident buffer of compact samples: output j is sample dev_index[j] (mirrored left-right when dev_flip[j]):
 * dev_states[j] float32[15][10][9], dev_pi[j] float32[8100] (visits^(1/T) normalised in float64, cast to float32),
 * dev_z[j] float32.  dev_samples: xq_sample[...] in device memory.
 * ===================================================================================== */
int xq_samples_to_batch(const void *dev_samples, const int32_t *dev_index, const uint8_t *dev_flip, int n,
                        double late_temperature, float *dev_states, float *dev_pi, float *dev_z, void *stream);

/* Finished training sample (compact form of the reference's (state, pi, z) tuple,
 * parallel_selfplay.py:97-99,123-132; dense pi / planes / flip augmentation materialise on the consumer). */
typedef struct xq_sample {
    int8_t board[XQ_SQUARES];
    int8_t side;        /* player to move when the sample was taken */
    int8_t z;           /* +1 win / 0 draw / -1 loss from `side`'s view */
    uint8_t n_moves;
    uint8_t late_temp;  /* 0: T = 1.0, 1: T = late_temperature */
    uint16_t ply;       /* move_count */
    uint16_t reserved0, reserved1; /* explicit: no implicit padding anywhere in this struct */
    uint32_t slot, game_seq;
    uint8_t pad[20];
    uint16_t actions[XQ_MAXM];
    uint16_t visits[XQ_MAXM];
} xq_sample;

typedef struct xq_game_result {
    uint32_t slot, game_seq;
    int8_t winner;      /* +1 / -1 / 0 */
    uint8_t reason;     /* 1 rules (is_game_over), 2 max_game_length adjudication, 3 resign */
    uint16_t steps;     /* game.move_count */
    uint16_t n_samples;
    uint16_t reserved;
} xq_game_result;

#ifdef __cplusplus
}
#endif
#endif /* XQ_HIP_H */

/*
 * xq_oracle.h -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the self-play hot path of wenjunyang/xiangqi-alphazero
 * (training/game.py, training/cython_engine/game_core.pyx, training/mcts.py,
 * training/parallel_selfplay.py).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may link or call this.  The product path
 * (xiangqi-alphazero_amd/csrc, include/xq_hip.h) never does.
 *
 * Parity status: PINNED.  tests/golden/gen_golden.py imports the reference in the
 * build container and records its outputs; tests/test_oracle_golden.py checks every
 * function below against those fixtures (perft 1-4, ordered move lists on a
 * position corpus, the reference's own known-answer positions of test_v3.py:122-197
 * and test_cython.py:62-69, MCTS traces under a stub evaluator, a full recorded
 * game with injected random draws, flip augmentation).
 *
 * Numeric semantics of the MCTS follow what the reference *executes* under
 * NumPy >= 2 (NEP 50 weak Python scalars), see the comments in xq_oracle.c.
 */
#ifndef XQ_ORACLE_H
#define XQ_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XQO_ROWS 10
#define XQO_COLS 9
#define XQO_SQUARES 90
#define XQO_ACTION_SPACE 8100
#define XQO_MAX_MOVES 200          /* game_core.pyx:50 */
#define XQO_STATE_FLOATS (15 * 90) /* game.py:627 */

/* ---- rules (game_core.pyx / game.py) ------------------------------------ */
int xqo_find_king(const int8_t *board, int player, int *kr, int *kc);
int xqo_is_attacked(const int8_t *board, int kr, int kc, int by_player);
int xqo_is_in_check(const int8_t *board, int player);
int xqo_generate_legal_moves(const int8_t *board, int player, uint16_t *actions);
int xqo_has_legal_moves(const int8_t *board, int player);
int xqo_material(const int8_t *board, int player);
void xqo_encode_state(const int8_t *board, int player, float *out);
void xqo_initial_board(int8_t *board);

/* ---- game state (game.py:124-170, 528-616) ------------------------------ */
typedef struct xqo_game {
    int8_t board[XQO_SQUARES];
    int player;      /* +1 red / -1 black */
    int move_count;
    int no_capture;
    int hist_len;    /* == len(history) */
    int hist_cap;
    int8_t *hist;    /* hist_len boards of 90 bytes, oldest first */
} xqo_game;

void xqo_game_init(xqo_game *g);
void xqo_game_free(xqo_game *g);
void xqo_game_clone(xqo_game *dst, const xqo_game *src); /* dst must be init'ed or zeroed */
void xqo_game_make_action(xqo_game *g, int action);
/* returns done (0/1); *winner = +1/-1/0 when done, 2 (None) otherwise */
int xqo_game_is_over(const xqo_game *g, int *winner);

/* ---- MCTS (mcts.py) ------------------------------------------------------ */
/* Evaluator plugin == the reference's `.predict(state) -> (probs f32[8100], float)`.
 * Returns 0 on success. */
typedef int (*xqo_eval_fn)(void *ctx, const float *state, float *probs, double *value);

typedef struct xqo_search_result {
    int n_children;                      /* root children == legal moves, move order */
    uint16_t actions[XQO_MAX_MOVES];
    int32_t visits[XQO_MAX_MOVES];
    double total_value[XQO_MAX_MOVES];
    double prior[XQO_MAX_MOVES];         /* as stored (f32 value widened, or f64) */
    int prior_is_f64;
    int32_t root_visits;
    int64_t nodes_created;
    int64_t evals;                       /* NN evaluations incl. root */
    int64_t terminal_sims;
    int64_t depth_sum;                   /* sum over sims of descent depth */
    int32_t max_depth;
} xqo_search_result;

/* noise == NULL  <=> add_noise=False; otherwise noise[i] is eta_i for the i-th legal move. */
int xqo_mcts_search(const xqo_game *game, int num_simulations, double c_puct,
                    const double *noise, xqo_eval_fn eval, void *ctx,
                    xqo_search_result *res);

/* mcts.py:190-206.  probs: f64[8100]. */
void xqo_action_probs(const xqo_search_result *res, double temperature, double *probs);

/* parallel_selfplay.py:137-151: index map of the left-right mirror. */
int xqo_flip_action(int action);

/* numpy RandomState.choice(p) given the uniform draw u (cumsum, /=last, searchsorted right) */
int xqo_choice_from_uniform(const double *p, int n, double u);

/* ---- game loop (parallel_selfplay.py:42-134) with injected randomness ---- */
typedef struct xqo_rand_source {
    /* each returns the next injected draw; ctx is passed through */
    void *ctx;
    int (*randint)(void *ctx, int lo, int hi);          /* inclusive, random.randint */
    int (*choice_index)(void *ctx, int n);              /* random.choice over n items */
    void (*dirichlet)(void *ctx, int n, double *out);   /* np.random.dirichlet([0.3]*n) */
    double (*uniform)(void *ctx);                       /* draw used by np.random.choice */
} xqo_rand_source;

typedef struct xqo_config {
    int num_simulations;
    double c_puct;
    int temperature_threshold;
    int max_game_length;
    int random_opening_moves;
    int enable_resign;
    double resign_threshold;
    int resign_check_steps;
} xqo_config;

typedef struct xqo_sample {
    int8_t board[XQO_SQUARES];
    int8_t player;
    int8_t z;              /* +1 / 0 / -1 */
    int16_t n_moves;
    double temperature;
    uint16_t actions[XQO_MAX_MOVES];
    int32_t visits[XQO_MAX_MOVES];
} xqo_sample;

/* Plays one game; samples must hold >= 512 entries.  Returns number of samples;
 * *winner, *steps as the reference returns them. */
int xqo_play_one_game(const xqo_config *cfg, xqo_eval_fn eval, void *ectx,
                      const xqo_rand_source *rs, xqo_sample *samples, int max_samples,
                      int *winner, int *steps, int64_t *sims_done, int64_t *evals_done);

/* one evaluation game: new vs old model, temperature 0, no noise (train.py:453-535) */
int xqo_arena_game(xqo_eval_fn eval_new, void *ctx_new, xqo_eval_fn eval_old, void *ctx_old, int new_is_red,
                   int num_simulations, double c_puct, int max_game_length, int *winner_out, int *steps_out);

/* perft from a board (make/unmake via copies), for pinning a2-a5 */
int64_t xqo_perft(const int8_t *board, int player, int depth);

#ifdef __cplusplus
}
#endif
#endif

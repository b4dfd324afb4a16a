/*
 * xq_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 * See xq_oracle.h for the scope and the parity-pinning statement.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (oracle/Makefile).  Floating-point
 * contraction must stay off: the PUCT arithmetic below is specified operation by
 * operation.
 *
 * Every function cites the reference lines (relative to /root/reference/) it restates.
 */
#include "xq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { EMPTY = 0, KING = 1, ADVISOR = 2, BISHOP = 3, KNIGHT = 4, ROOK = 5, CANNON = 6, PAWN = 7 };

#define SQ(r, c) ((r) * XQO_COLS + (c))
#define ON_BOARD(r, c) ((unsigned)(r) < XQO_ROWS && (unsigned)(c) < XQO_COLS)

/* training/cython_engine/game_core.pyx:42-46 -- orthogonal directions, in emission order */
static const int ORTHO[4][2] = {{-1, 0}, {1, 0}, {0, -1}, {0, 1}};
/* game_core.pyx:31-39 -- knight jump (dr, dc) and its leg square (br, bc), emission order */
static const int KNIGHT_TAB[8][4] = {{-2, -1, -1, 0}, {-2, 1, -1, 0}, {2, -1, 1, 0}, {2, 1, 1, 0},
                                     {-1, -2, 0, -1}, {-1, 2, 0, 1},  {1, -2, 0, -1}, {1, 2, 0, 1}};
/* game.py:74 */
static const int PIECE_VALUE[8] = {0, 0, 20, 20, 40, 90, 45, 10};

static inline int own(int p, int player) { return player == 1 ? p > 0 : p < 0; }
static inline int enemy(int p, int player) { return player == 1 ? p < 0 : p > 0; }
static inline int can_land(int p, int player) { return p == EMPTY || enemy(p, player); }

/* game.py:139-159 */
void xqo_initial_board(int8_t *b) {
    static const int8_t back[9] = {ROOK, KNIGHT, BISHOP, ADVISOR, KING, ADVISOR, BISHOP, KNIGHT, ROOK};
    memset(b, 0, XQO_SQUARES);
    for (int c = 0; c < 9; ++c) {
        b[SQ(0, c)] = back[c];
        b[SQ(9, c)] = (int8_t)-back[c];
    }
    b[SQ(2, 1)] = CANNON;  b[SQ(2, 7)] = CANNON;
    b[SQ(7, 1)] = -CANNON; b[SQ(7, 7)] = -CANNON;
    for (int c = 0; c < 9; c += 2) {
        b[SQ(3, c)] = PAWN;
        b[SQ(6, c)] = -PAWN;
    }
}

/* game_core.pyx:78-101 / game.py:426-439 -- palace-only scan, first hit in (row, col) order */
int xqo_find_king(const int8_t *b, int player, int *kr, int *kc) {
    int r0 = player == 1 ? 0 : 7;
    int target = player == 1 ? KING : -KING;
    for (int r = r0; r < r0 + 3; ++r)
        for (int c = 3; c <= 5; ++c)
            if (b[SQ(r, c)] == target) {
                *kr = r; *kc = c;
                return 1;
            }
    *kr = -1; *kc = -1;
    return 0;
}

/* game_core.pyx:104-189 / game.py:176-265 -- reverse attack scan from (kr,kc) */
int xqo_is_attacked(const int8_t *b, int kr, int kc, int by) {
    const int e_rook = ROOK * by, e_cannon = CANNON * by, e_knight = KNIGHT * by;
    const int e_pawn = PAWN * by, e_king = KING * by;

    /* rook / king on an open line */
    for (int d = 0; d < 4; ++d) {
        int r = kr + ORTHO[d][0], c = kc + ORTHO[d][1];
        while (ON_BOARD(r, c)) {
            int p = b[SQ(r, c)];
            if (p != EMPTY) {
                if (p == e_rook || p == e_king) return 1;
                break;
            }
            r += ORTHO[d][0]; c += ORTHO[d][1];
        }
    }
    /* cannon over exactly one screen */
    for (int d = 0; d < 4; ++d) {
        int r = kr + ORTHO[d][0], c = kc + ORTHO[d][1];
        int screen = 0;
        while (ON_BOARD(r, c)) {
            int p = b[SQ(r, c)];
            if (p != EMPTY) {
                if (!screen) screen = 1;
                else {
                    if (p == e_cannon) return 1;
                    break;
                }
            }
            r += ORTHO[d][0]; c += ORTHO[d][1];
        }
    }
    /* knight origins with the leg taken from the knight's side */
    for (int i = 0; i < 8; ++i) {
        int nr = kr + KNIGHT_TAB[i][0], nc = kc + KNIGHT_TAB[i][1];
        if (ON_BOARD(nr, nc) && b[SQ(nr, nc)] == e_knight) {
            int mdr = kr - nr, mdc = kc - nc;
            int lr, lc;
            if (mdr == 2 || mdr == -2) { lr = nr + mdr / 2; lc = nc; }
            else                        { lr = nr;           lc = nc + mdc / 2; }
            if (b[SQ(lr, lc)] == EMPTY) return 1;
        }
    }
    /* pawns: a red pawn attacks upward (and sideways once kr >= 5), black mirrored */
    if (by == 1) {
        if (kr - 1 >= 0 && b[SQ(kr - 1, kc)] == e_pawn) return 1;
        if (kr >= 5) {
            if (kc - 1 >= 0 && b[SQ(kr, kc - 1)] == e_pawn) return 1;
            if (kc + 1 < XQO_COLS && b[SQ(kr, kc + 1)] == e_pawn) return 1;
        }
    } else {
        if (kr + 1 < XQO_ROWS && b[SQ(kr + 1, kc)] == e_pawn) return 1;
        if (kr <= 4) {
            if (kc - 1 >= 0 && b[SQ(kr, kc - 1)] == e_pawn) return 1;
            if (kc + 1 < XQO_COLS && b[SQ(kr, kc + 1)] == e_pawn) return 1;
        }
    }
    return 0;
}

/* game_core.pyx:543-555 / game.py:652-661 -- a missing king counts as "in check" */
int xqo_is_in_check(const int8_t *b, int player) {
    int kr, kc;
    if (!xqo_find_king(b, player, &kr, &kc)) return 1;
    return xqo_is_attacked(b, kr, kc, -player);
}

/* game_core.pyx:209-252 / game.py:441-490 -- make, king present?, flying general, attacked?, unmake */
static int move_is_legal(int8_t *b, int from, int to, int player) {
    int8_t mover = b[from], captured = b[to];
    int kr, kc, er, ec, legal = 1;
    b[to] = mover;
    b[from] = EMPTY;
    if (!xqo_find_king(b, player, &kr, &kc)) {
        legal = 0;
    } else {
        if (xqo_find_king(b, -player, &er, &ec) && kc == ec) {
            int lo = (kr < er ? kr : er) + 1, hi = kr < er ? er : kr;
            int blocked = 0;
            for (int r = lo; r < hi; ++r)
                if (b[SQ(r, kc)] != EMPTY) { blocked = 1; break; }
            if (!blocked) legal = 0;
        }
        if (legal) legal = !xqo_is_attacked(b, kr, kc, -player);
    }
    b[from] = mover;
    b[to] = captured;
    return legal;
}

#define EMIT(fr, fc, tr, tc)                                                    \
    do {                                                                        \
        if (move_is_legal(b, SQ(fr, fc), SQ(tr, tc), player))                   \
            out[n++] = (uint16_t)(SQ(fr, fc) * 90 + SQ(tr, tc));                \
    } while (0)

/* game_core.pyx:262-486 (+ Python twin game.py:297-424, 492-521): squares row-major, per piece
 * the reference's target order.  Actions are encode_action() ids (game.py:112-114). */
int xqo_generate_legal_moves(const int8_t *board, int player, uint16_t *out) {
    int8_t b[XQO_SQUARES];
    int n = 0;
    memcpy(b, board, XQO_SQUARES);
    for (int r = 0; r < XQO_ROWS; ++r)
        for (int c = 0; c < XQO_COLS; ++c) {
            int piece = b[SQ(r, c)];
            if (piece == EMPTY || !own(piece, player)) continue;
            int kind = piece > 0 ? piece : -piece;
            switch (kind) {
            case KING: {
                int lo = player == 1 ? 0 : 7;
                for (int d = 0; d < 4; ++d) {
                    int nr = r + ORTHO[d][0], nc = c + ORTHO[d][1];
                    if (nr >= lo && nr <= lo + 2 && nc >= 3 && nc <= 5 && can_land(b[SQ(nr, nc)], player))
                        EMIT(r, c, nr, nc);
                }
                break;
            }
            case ADVISOR:
                for (int dr = -1; dr <= 1; dr += 2)
                    for (int dc = -1; dc <= 1; dc += 2) {
                        int nr = r + dr, nc = c + dc;
                        if (!ON_BOARD(nr, nc) || nc < 3 || nc > 5) continue;
                        if (player == 1 && nr > 2) continue;
                        if (player == -1 && nr < 7) continue;
                        if (can_land(b[SQ(nr, nc)], player)) EMIT(r, c, nr, nc);
                    }
                break;
            case BISHOP:
                for (int dr = -2; dr <= 2; dr += 4)
                    for (int dc = -2; dc <= 2; dc += 4) {
                        int nr = r + dr, nc = c + dc;
                        if (!ON_BOARD(nr, nc)) continue;
                        if (player == 1 && nr > 4) continue;
                        if (player == -1 && nr < 5) continue;
                        if (b[SQ(r + dr / 2, c + dc / 2)] != EMPTY) continue;
                        if (can_land(b[SQ(nr, nc)], player)) EMIT(r, c, nr, nc);
                    }
                break;
            case KNIGHT:
                for (int i = 0; i < 8; ++i) {
                    int nr = r + KNIGHT_TAB[i][0], nc = c + KNIGHT_TAB[i][1];
                    if (!ON_BOARD(nr, nc)) continue;
                    if (b[SQ(r + KNIGHT_TAB[i][2], c + KNIGHT_TAB[i][3])] != EMPTY) continue;
                    if (can_land(b[SQ(nr, nc)], player)) EMIT(r, c, nr, nc);
                }
                break;
            case ROOK:
                for (int d = 0; d < 4; ++d) {
                    int nr = r + ORTHO[d][0], nc = c + ORTHO[d][1];
                    while (ON_BOARD(nr, nc)) {
                        int p = b[SQ(nr, nc)];
                        if (p == EMPTY) {
                            EMIT(r, c, nr, nc);
                        } else {
                            if (enemy(p, player)) EMIT(r, c, nr, nc);
                            break;
                        }
                        nr += ORTHO[d][0]; nc += ORTHO[d][1];
                    }
                }
                break;
            case CANNON:
                for (int d = 0; d < 4; ++d) {
                    int nr = r + ORTHO[d][0], nc = c + ORTHO[d][1];
                    while (ON_BOARD(nr, nc) && b[SQ(nr, nc)] == EMPTY) {
                        EMIT(r, c, nr, nc);
                        nr += ORTHO[d][0]; nc += ORTHO[d][1];
                    }
                    if (ON_BOARD(nr, nc)) { /* (nr,nc) is the screen */
                        nr += ORTHO[d][0]; nc += ORTHO[d][1];
                        while (ON_BOARD(nr, nc)) {
                            int p = b[SQ(nr, nc)];
                            if (p != EMPTY) {
                                if (enemy(p, player)) EMIT(r, c, nr, nc);
                                break;
                            }
                            nr += ORTHO[d][0]; nc += ORTHO[d][1];
                        }
                    }
                }
                break;
            case PAWN: {
                int fwd = player == 1 ? 1 : -1;
                int crossed = player == 1 ? r >= 5 : r <= 4;
                int nr = r + fwd;
                if (nr >= 0 && nr < XQO_ROWS && can_land(b[SQ(nr, c)], player)) EMIT(r, c, nr, c);
                if (crossed) {
                    if (c - 1 >= 0 && can_land(b[SQ(r, c - 1)], player)) EMIT(r, c, r, c - 1);
                    if (c + 1 < XQO_COLS && can_land(b[SQ(r, c + 1)], player)) EMIT(r, c, r, c + 1);
                }
                break;
            }
            default: break;
            }
        }
    return n;
}
#undef EMIT

/* game_core.pyx:558-569 */
int xqo_has_legal_moves(const int8_t *board, int player) {
    uint16_t tmp[XQO_MAX_MOVES];
    return xqo_generate_legal_moves(board, player, tmp) > 0;
}

/* game.py:552-563 */
int xqo_material(const int8_t *b, int player) {
    int s = 0;
    for (int i = 0; i < XQO_SQUARES; ++i) {
        int p = b[i];
        if (player == 1 && p > 0) s += PIECE_VALUE[p];
        if (player == -1 && p < 0) s += PIECE_VALUE[-p];
    }
    return s;
}

/* game.py:618-640 -- planes 0-6 side to move, 7-13 opponent, 14 all-ones iff red to move; no flip */
void xqo_encode_state(const int8_t *b, int player, float *out) {
    memset(out, 0, sizeof(float) * XQO_STATE_FLOATS);
    for (int i = 0; i < XQO_SQUARES; ++i) {
        int p = b[i];
        if (p == 0) continue;
        int kind = p > 0 ? p : -p;
        int mine = (p > 0) == (player == 1);
        out[((mine ? 0 : 7) + kind - 1) * 90 + i] = 1.0f;
    }
    if (player == 1)
        for (int i = 0; i < XQO_SQUARES; ++i) out[14 * 90 + i] = 1.0f;
}

/* ------------------------------------------------------------------------ */
/* game state                                                                */

void xqo_game_init(xqo_game *g) {
    memset(g, 0, sizeof(*g));
    xqo_initial_board(g->board);
    g->player = 1;
}

void xqo_game_free(xqo_game *g) {
    free(g->hist);
    g->hist = NULL;
    g->hist_cap = g->hist_len = 0;
}

/* game.py:161-170 */
void xqo_game_clone(xqo_game *dst, const xqo_game *src) {
    int8_t *buf = dst->hist;
    int cap = dst->hist_cap;
    if (cap < src->hist_len + 64) {
        cap = src->hist_len + 64;
        buf = (int8_t *)realloc(buf, (size_t)cap * XQO_SQUARES);
    }
    *dst = *src;
    dst->hist = buf;
    dst->hist_cap = cap;
    if (src->hist_len) memcpy(dst->hist, src->hist, (size_t)src->hist_len * XQO_SQUARES);
}

/* game.py:528-550 -- no legality check; pre-move board appended to history */
void xqo_game_make_action(xqo_game *g, int action) {
    int from = action / 90, to = action % 90;
    if (g->hist_len == g->hist_cap) {
        g->hist_cap = g->hist_cap ? g->hist_cap * 2 : 64;
        g->hist = (int8_t *)realloc(g->hist, (size_t)g->hist_cap * XQO_SQUARES);
    }
    memcpy(g->hist + (size_t)g->hist_len * XQO_SQUARES, g->board, XQO_SQUARES);
    g->hist_len++;
    int captured = g->board[to];
    g->board[to] = g->board[from];
    g->board[from] = EMPTY;
    g->no_capture = captured != EMPTY ? 0 : g->no_capture + 1;
    g->player = -g->player;
    g->move_count++;
}

/* game.py:565-616 -- rule order matters */
int xqo_game_is_over(const xqo_game *g, int *winner) {
    int kr, kc;
    uint16_t tmp[XQO_MAX_MOVES];
    if (!xqo_find_king(g->board, 1, &kr, &kc)) { *winner = -1; return 1; }
    if (!xqo_find_king(g->board, -1, &kr, &kc)) { *winner = 1; return 1; }
    if (xqo_generate_legal_moves(g->board, g->player, tmp) == 0) { *winner = -g->player; return 1; }
    if (g->no_capture >= 120) { *winner = 0; return 1; }
    if (g->move_count >= 200) {
        int diff = xqo_material(g->board, 1) - xqo_material(g->board, -1);
        *winner = diff > 30 ? 1 : (diff < -30 ? -1 : 0);
        return 1;
    }
    if (g->hist_len >= 6) {
        int first = g->hist_len > 12 ? g->hist_len - 12 : 0;
        int rep = 0;
        for (int i = first; i < g->hist_len; ++i)
            if (memcmp(g->hist + (size_t)i * XQO_SQUARES, g->board, XQO_SQUARES) == 0)
                if (++rep >= 3) { *winner = 0; return 1; }
    }
    *winner = 2;
    return 0;
}

int64_t xqo_perft(const int8_t *board, int player, int depth) {
    uint16_t mv[XQO_MAX_MOVES];
    int n = xqo_generate_legal_moves(board, player, mv);
    if (depth <= 1) return n;
    int64_t total = 0;
    int8_t nb[XQO_SQUARES];
    for (int i = 0; i < n; ++i) {
        memcpy(nb, board, XQO_SQUARES);
        int from = mv[i] / 90, to = mv[i] % 90;
        nb[to] = nb[from];
        nb[from] = EMPTY;
        total += xqo_perft(nb, -player, depth - 1);
    }
    return total;
}

/* ------------------------------------------------------------------------ */
/* MCTS (mcts.py)                                                            */
/*
 * Types as the reference executes them under NumPy 2 (NEP 50):
 *   policy_probs      : ndarray float32 (model.py:122)
 *   prob_sum          : builtin sum() from int 0 over np.float32 -> sequential float32 adds (mcts.py:180)
 *   prior             : np.float32 = p[a] / prob_sum           (mcts.py:183)   -> "f32 priors"
 *                       python float 1.0/len if prob_sum <= 0  (mcts.py:185)   -> "f64 priors"
 *   noisy root prior  : 0.75*prior (float32 if prior is float32) + 0.25*noise[i] (float64) -> float64 (mcts.py:121)
 *   q_value           : python float total_value/visit_count (float64), 0.0 if unvisited (mcts.py:33-38)
 *   ucb, f32 priors   : ((f32(c)*P) * f32(sqrt_parent)) / f32(1+N), then f32(q) + that  -- all float32
 *   ucb, f64 priors   : q + ((c*P)*sqrt_parent)/(1+N)                                      -- all float64
 *   total_value       : python float; += value, value = -value up the parent chain (mcts.py:66-73)
 */

typedef struct node {
    int parent;
    int first_child;   /* index of first child, children are contiguous, move order */
    int n_children;
    int visits;
    double total;
    double prior;      /* float32 value widened when !f64 */
    uint16_t action;
    uint8_t prior_f64; /* type of THIS node's prior */
} node;

typedef struct tree {
    node *n;
    int len, cap;
} tree;

static int tree_new_nodes(tree *t, int count) {
    if (t->len + count > t->cap) {
        while (t->len + count > t->cap) t->cap = t->cap ? t->cap * 2 : 4096;
        t->n = (node *)realloc(t->n, (size_t)t->cap * sizeof(node));
    }
    int first = t->len;
    t->len += count;
    return first;
}

/* mcts.py:176-188 + 60-64: children for `legal` under parent `pi`; returns 1 if priors are f64 */
static void expand_node(tree *t, int pi, const float *probs, const uint16_t *legal, int n_legal,
                        const double *noise) {
    float sum = 0.0f;
    for (int i = 0; i < n_legal; ++i) sum = sum + probs[legal[i]];
    int first = tree_new_nodes(t, n_legal);
    t->n[pi].first_child = first;
    t->n[pi].n_children = n_legal;
    for (int i = 0; i < n_legal; ++i) {
        node *ch = &t->n[first + i];
        ch->parent = pi;
        ch->first_child = -1;
        ch->n_children = 0;
        ch->visits = 0;
        ch->total = 0.0;
        ch->action = legal[i];
        if (sum > 0.0f) {
            float p = probs[legal[i]] / sum;
            if (noise) {
                float scaled = 0.75f * p; /* float32 product, mcts.py:121 */
                ch->prior = (double)scaled + 0.25 * noise[i];
                ch->prior_f64 = 1;
            } else {
                ch->prior = (double)p;
                ch->prior_f64 = 0;
            }
        } else {
            double u = 1.0 / (double)n_legal;
            ch->prior = noise ? 0.75 * u + 0.25 * noise[i] : u;
            ch->prior_f64 = 1;
        }
    }
}

/* mcts.py:43-58 -- strict '>' keeps the first maximum */
static int select_child(const tree *t, int pi, double c_puct) {
    const node *p = &t->n[pi];
    double sqrt_parent = sqrt((double)p->visits);
    int best = -1;
    int have_f32 = 0;
    /* best_score starts as python -inf; after the first assignment it carries the ucb's type. Comparison
     * between float32 and float64 values is exact in float64, so one double holds either. */
    double best_score = -INFINITY;
    (void)have_f32;
    for (int i = 0; i < p->n_children; ++i) {
        const node *ch = &t->n[p->first_child + i];
        double q = ch->visits ? ch->total / (double)ch->visits : 0.0;
        double ucb;
        if (!ch->prior_f64) {
            float u = (float)c_puct * (float)ch->prior;
            u = u * (float)sqrt_parent;
            u = u / (float)(1 + ch->visits);
            u = (float)q + u;
            ucb = (double)u;
        } else {
            double u = c_puct * ch->prior;
            u = u * sqrt_parent;
            u = u / (double)(1 + ch->visits);
            ucb = q + u;
        }
        if (ucb > best_score) {
            best_score = ucb;
            best = p->first_child + i;
        }
    }
    return best;
}

int xqo_mcts_search(const xqo_game *game, int num_simulations, double c_puct, const double *noise,
                    xqo_eval_fn eval, void *ctx, xqo_search_result *res) {
    tree t = {0};
    float state[XQO_STATE_FLOATS];
    float *probs = (float *)malloc(sizeof(float) * XQO_ACTION_SPACE);
    uint16_t legal[XQO_MAX_MOVES];
    double value;
    int rc = 0;
    xqo_game sim;
    memset(&sim, 0, sizeof(sim));
    memset(res, 0, sizeof(*res));

    int root = tree_new_nodes(&t, 1);
    t.n[root].parent = -1;
    t.n[root].first_child = -1;
    t.n[root].n_children = 0;
    t.n[root].visits = 0;
    t.n[root].total = 0.0;
    t.n[root].prior = 0.0;
    t.n[root].prior_f64 = 1;
    t.n[root].action = 0;

    /* mcts.py:107-123 */
    xqo_encode_state(game->board, game->player, state);
    if ((rc = eval(ctx, state, probs, &value)) != 0) goto done;
    res->evals++;
    int n_legal = xqo_generate_legal_moves(game->board, game->player, legal);
    if (n_legal == 0) goto done; /* mcts.py:111-112: all-zero distribution */
    expand_node(&t, root, probs, legal, n_legal, noise);

    /* mcts.py:126-153 */
    for (int s = 0; s < num_simulations; ++s) {
        int ni = root, depth = 0, winner;
        xqo_game_clone(&sim, game);
        while (t.n[ni].n_children > 0) {
            ni = select_child(&t, ni, c_puct);
            xqo_game_make_action(&sim, t.n[ni].action);
            depth++;
        }
        res->depth_sum += depth;
        if (depth > res->max_depth) res->max_depth = depth;
        double v;
        if (xqo_game_is_over(&sim, &winner)) {
            v = winner == 0 ? 0.0 : 1.0; /* mcts.py:137-140: not side-aware */
            res->terminal_sims++;
        } else {
            xqo_encode_state(sim.board, sim.player, state);
            if ((rc = eval(ctx, state, probs, &value)) != 0) goto done;
            res->evals++;
            n_legal = xqo_generate_legal_moves(sim.board, sim.player, legal);
            if (n_legal > 0) expand_node(&t, ni, probs, legal, n_legal, NULL);
            v = -value;
        }
        for (int k = ni; k >= 0; k = t.n[k].parent) { /* mcts.py:66-73 */
            t.n[k].visits += 1;
            t.n[k].total += v;
            v = -v;
        }
    }

    res->n_children = t.n[root].n_children;
    res->root_visits = t.n[root].visits;
    res->nodes_created = t.len - 1;
    for (int i = 0; i < res->n_children; ++i) {
        const node *ch = &t.n[t.n[root].first_child + i];
        res->actions[i] = ch->action;
        res->visits[i] = ch->visits;
        res->total_value[i] = ch->total;
        res->prior[i] = ch->prior;
        res->prior_is_f64 = ch->prior_f64;
    }
done:
    xqo_game_free(&sim);
    free(probs);
    free(t.n);
    return rc;
}

/* mcts.py:190-206 */
void xqo_action_probs(const xqo_search_result *res, double temperature, double *probs) {
    memset(probs, 0, sizeof(double) * XQO_ACTION_SPACE);
    if (res->n_children == 0) return;
    if (temperature == 0.0) {
        int best = 0;
        for (int i = 1; i < res->n_children; ++i)
            if (res->visits[i] > res->visits[best]) best = i;
        probs[res->actions[best]] = 1.0;
        return;
    }
    for (int i = 0; i < res->n_children; ++i) probs[res->actions[i]] = (double)res->visits[i];
    /* ndarray.sum() over 8100 float64 uses pairwise summation; the addends are integers (or their
     * powers), so recompute it exactly the way numpy does only where it can matter: tests compare the
     * dense vector produced by numpy on the host from (actions, visits).  Here: plain left-to-right. */
    double total = 0.0;
    for (int a = 0; a < XQO_ACTION_SPACE; ++a) total += probs[a];
    if (total > 0.0) {
        double inv_t = 1.0 / temperature;
        double s2 = 0.0;
        for (int a = 0; a < XQO_ACTION_SPACE; ++a) {
            if (probs[a] != 0.0) probs[a] = pow(probs[a], inv_t);
            s2 += probs[a];
        }
        for (int a = 0; a < XQO_ACTION_SPACE; ++a) probs[a] /= s2;
    }
}

/* parallel_selfplay.py:147-148 */
int xqo_flip_action(int action) {
    int from = action / 90, to = action % 90;
    int fr = from / 9, fc = from % 9, tr = to / 9, tc = to % 9;
    return SQ(fr, 8 - fc) * 90 + SQ(tr, 8 - tc);
}

/* numpy/random/mtrand.pyx RandomState.choice (numpy 2.2): cdf = p.cumsum(); cdf /= cdf[-1];
 * idx = cdf.searchsorted(u, side='right') */
int xqo_choice_from_uniform(const double *p, int n, double u) {
    double last = 0.0;
    for (int i = 0; i < n; ++i) last += p[i];
    double run = 0.0;
    int lo = 0;
    /* first index with cdf[i] > u (cdf non-decreasing) */
    for (int i = 0; i < n; ++i) {
        run += p[i];
        if (run / last > u) { lo = i; return lo; }
    }
    return n; /* numpy would return n (out of range) -- cannot happen for u < 1 */
}

/* ------------------------------------------------------------------------ */
/* game loop (parallel_selfplay.py:42-134)                                   */

int xqo_play_one_game(const xqo_config *cfg, xqo_eval_fn eval, void *ectx, const xqo_rand_source *rs,
                      xqo_sample *samples, int max_samples, int *winner_out, int *steps_out,
                      int64_t *sims_done, int64_t *evals_done) {
    xqo_game g;
    uint16_t legal[XQO_MAX_MOVES];
    double *pi = (double *)malloc(sizeof(double) * XQO_ACTION_SPACE);
    float *probs = (float *)malloc(sizeof(float) * XQO_ACTION_SPACE);
    float state[XQO_STATE_FLOATS];
    double noise[XQO_MAX_MOVES];
    double resign_hist[1024];
    int n_resign = 0, n_samples = 0, winner = 2, w;
    int64_t sims = 0, evals = 0;
    xqo_search_result res;

    xqo_game_init(&g);
    /* :63-72 random opening */
    int k = rs->randint(rs->ctx, 0, cfg->random_opening_moves);
    for (int i = 0; i < k; ++i) {
        int n = xqo_generate_legal_moves(g.board, g.player, legal);
        if (n == 0) break;
        xqo_game_make_action(&g, legal[rs->choice_index(rs->ctx, n)]);
        if (xqo_game_is_over(&g, &w)) {
            xqo_game_free(&g);
            xqo_game_init(&g);
            break;
        }
    }

    for (;;) {
        if (xqo_game_is_over(&g, &w)) { winner = w; break; }       /* :75-77 */
        if (g.move_count >= cfg->max_game_length) {                 /* :79-89 */
            int diff = xqo_material(g.board, 1) - xqo_material(g.board, -1);
            winner = diff > 30 ? 1 : (diff < -30 ? -1 : 0);
            break;
        }
        double T = g.move_count < cfg->temperature_threshold ? 1.0 : 0.3; /* :92 */
        int n = xqo_generate_legal_moves(g.board, g.player, legal);
        rs->dirichlet(rs->ctx, n, noise);
        if (xqo_mcts_search(&g, cfg->num_simulations, cfg->c_puct, noise, eval, ectx, &res) != 0) break;
        sims += cfg->num_simulations;
        evals += res.evals;
        if (n_samples < max_samples) {                              /* :98-99 */
            xqo_sample *s = &samples[n_samples];
            memcpy(s->board, g.board, XQO_SQUARES);
            s->player = (int8_t)g.player;
            s->z = 0;
            s->n_moves = (int16_t)res.n_children;
            s->temperature = T;
            memcpy(s->actions, res.actions, sizeof(uint16_t) * res.n_children);
            memcpy(s->visits, res.visits, sizeof(int32_t) * res.n_children);
        }
        n_samples++;
        xqo_action_probs(&res, T, pi);                              /* :101-107 */
        int action = xqo_choice_from_uniform(pi, XQO_ACTION_SPACE, rs->uniform(rs->ctx));
        xqo_game_make_action(&g, action);
        if (cfg->enable_resign && n_samples > 10) {                 /* :110-121 */
            double v;
            xqo_encode_state(g.board, g.player, state);
            if (eval(ectx, state, probs, &v) != 0) break;
            evals++;
            if (n_resign < 1024) resign_hist[n_resign++] = v;
            if (n_resign >= cfg->resign_check_steps) {
                int all_low = 1;
                for (int i = n_resign - cfg->resign_check_steps; i < n_resign; ++i)
                    if (!(resign_hist[i] < cfg->resign_threshold)) all_low = 0;
                if (all_low) { winner = -g.player; break; }
            }
        }
    }
    int stored = n_samples < max_samples ? n_samples : max_samples;
    for (int i = 0; i < stored; ++i)                                /* :124-132 */
        samples[i].z = (int8_t)(winner == 0 ? 0 : (winner == samples[i].player ? 1 : -1));
    *winner_out = winner;
    *steps_out = g.move_count;
    if (sims_done) *sims_done = sims;
    if (evals_done) *evals_done = evals;
    xqo_game_free(&g);
    free(pi);
    free(probs);
    return stored;
}

/* ------------------------------------------------------------------------ */
/* arena game (training/train.py:453-535, one iteration of the eval loop)     */

int xqo_arena_game(xqo_eval_fn eval_new, void *ctx_new, xqo_eval_fn eval_old, void *ctx_old, int new_is_red,
                   int num_simulations, double c_puct, int max_game_length, int *winner_out, int *steps_out) {
    xqo_game g;
    xqo_search_result res;
    int step = 0, w = 2, done = 0;
    xqo_game_init(&g);
    while (step < max_game_length) {
        const int red_turn = g.player == 1;
        const int use_new = (new_is_red && red_turn) || (!new_is_red && !red_turn);     /* train.py:479-483 */
        if (xqo_mcts_search(&g, num_simulations, c_puct, NULL, use_new ? eval_new : eval_old,
                            use_new ? ctx_new : ctx_old, &res) != 0 || res.n_children == 0) {
            xqo_game_free(&g);
            return -1;
        }
        int best = 0;                                      /* get_action(temperature=0): first max, mcts.py:197-200 */
        for (int i = 1; i < res.n_children; ++i)
            if (res.visits[i] > res.visits[best]) best = i;
        xqo_game_make_action(&g, res.actions[best]);
        step++;
        done = xqo_game_is_over(&g, &w);
        if (done) break;
    }
    if (!done) done = xqo_game_is_over(&g, &w);            /* train.py:494-496 */
    *winner_out = done ? w : 0;
    *steps_out = step;
    xqo_game_free(&g);
    return 0;
}

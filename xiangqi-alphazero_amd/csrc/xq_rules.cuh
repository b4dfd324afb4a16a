// xq_rules.cuh -- wave-level Xiangqi rules for gfx950 (one 64-lane wavefront == one position).
//
// Layout: the position's 90-byte board sits in LDS (96-byte slot, 4-byte aligned: its 23 dwords map to 23
// distinct LDS banks, so any mix of per-lane byte reads of ONE board is conflict-free / broadcast).
// Move generation is two-phase so that the reference's emission order survives SIMT execution:
//   A  lanes own (piece, direction-slot) tasks in the reference's enumeration order (squares row-major,
//      per piece the order of game_core.pyx:286-484); each task yields a run of pseudo-legal targets; an
//      exclusive wave scan of the run lengths gives every run its place in an LDS candidate list;
//   B  lanes own candidates; each lane plays its move on a *virtual* board (reads of `from`/`to` are
//      overridden in registers, LDS is never written) and runs the reverse attack scan on the king;
//      __ballot + mbcnt prefix popcount compacts survivors, still in order.
// Semantics follow training/cython_engine/game_core.pyx (the engine the reference actually runs) and are
// checked bit-for-bit against oracle/ and the golden fixtures.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define XQ_MAXM 128
#define XQ_BS 96          // internal board stride (bytes)
#define XQ_CAND_CAP 256   // pseudo-legal candidates per position kept in LDS
#define XQ_HIST 12

namespace xq {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// All engine kernels run ONE wavefront per workgroup.  LDS operations of one wave execute in issue order, so
// lane-to-lane communication through LDS needs no barrier -- only that the compiler keeps the order (the "memory"
// clobber) and that pending LDS returns have landed.  A full __syncthreads() here would also wait for every
// outstanding GLOBAL store (vmcnt(0)) -- a ~1-2 us stall per call in the tree walk.
__device__ __forceinline__ void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Same, plus global memory: lane A's global stores become visible to lane B's later loads of the same wave.
__device__ __forceinline__ void wave_sync_mem() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ int wave_excl_scan(int v, int lane, int *total) {
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    *total = __shfl(x, 63);
    return x - v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

struct VMove {
    int from, to, mover;  // from = -1: no virtual move
};

__device__ __forceinline__ int vget(const int8_t *b, int sq, const VMove &m) {
    int p = b[sq];
    p = (sq == m.from) ? 0 : p;
    p = (sq == m.to) ? m.mover : p;
    return p;
}

__device__ __forceinline__ bool is_own(int p, int player) { return player == 1 ? p > 0 : p < 0; }
__device__ __forceinline__ bool is_enemy(int p, int player) { return player == 1 ? p < 0 : p > 0; }
__device__ __forceinline__ bool can_land(int p, int player) { return p == 0 || is_enemy(p, player); }

// game_core.pyx:42-46 order (-1,0) (1,0) (0,-1) (0,1), as square deltas
__device__ __forceinline__ void ortho(int d, int &dr, int &dc) {
    dr = (d == 0) ? -1 : (d == 1) ? 1 : 0;
    dc = (d == 2) ? -1 : (d == 3) ? 1 : 0;
}
// game_core.pyx:31-39 order; leg = half of the long component
__device__ __forceinline__ void knight(int i, int &dr, int &dc, int &lr, int &lc) {
    const int s1 = (i & 1) ? 1 : -1;       // sign of the short component
    const int s2 = (i & 2) ? 1 : -1;       // sign of the long component
    if (i < 4) { dr = 2 * s2; dc = s1; lr = s2; lc = 0; }
    else       { dr = s2; dc = 2 * s1; lr = 0; lc = s1; }
}

// game_core.pyx:78-101 -- palace scan, first hit in (row, col) order; returns square or -1
__device__ __forceinline__ int find_king(const int8_t *b, const VMove &m, int player) {
    const int r0 = player == 1 ? 0 : 7;
    const int target = player == 1 ? 1 : -1;
    int found = -1;
#pragma unroll
    for (int i = 8; i >= 0; --i) {          // descending so the lowest index wins without a branch
        const int sq = (r0 + i / 3) * 9 + 3 + i % 3;
        found = (vget(b, sq, m) == target) ? sq : found;
    }
    return found;
}

// game_core.pyx:104-189 -- reverse attack scan.  The rook/king ray pass and the cannon ray pass of the
// reference walk the same four rays; they are merged here (first piece met: rook/king hits, otherwise it is
// the screen; second piece met: cannon hits) -- the boolean result is identical.
__device__ __forceinline__ bool is_attacked(const int8_t *b, const VMove &m, int kr, int kc, int by) {
    const int e_king = by, e_knight = 4 * by, e_rook = 5 * by, e_cannon = 6 * by, e_pawn = 7 * by;
    bool hit = false;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        int dr, dc;
        ortho(d, dr, dc);
        int r = kr + dr, c = kc + dc;
        int seen = 0;
        while ((unsigned)r < 10u && (unsigned)c < 9u) {
            const int p = vget(b, r * 9 + c, m);
            if (p != 0) {
                if (seen == 0) {
                    if (p == e_rook || p == e_king) { hit = true; }
                    seen = 1;
                } else {
                    if (p == e_cannon) hit = true;
                    break;
                }
                if (hit) break;
            }
            r += dr; c += dc;
        }
        if (hit) return true;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int dr, dc, lr, lc;
        knight(i, dr, dc, lr, lc);
        const int nr = kr + dr, nc = kc + dc;
        if ((unsigned)nr < 10u && (unsigned)nc < 9u && vget(b, nr * 9 + nc, m) == e_knight) {
            // the knight sits at (nr,nc) and jumps by (-dr,-dc): its leg is one step along the long axis
            const int br = (dr == 2 || dr == -2) ? nr - dr / 2 : nr;
            const int bc = (dr == 2 || dr == -2) ? nc : nc - dc / 2;
            if (vget(b, br * 9 + bc, m) == 0) return true;
        }
    }
    if (by == 1) {
        if (kr - 1 >= 0 && vget(b, (kr - 1) * 9 + kc, m) == e_pawn) return true;
        if (kr >= 5) {
            if (kc - 1 >= 0 && vget(b, kr * 9 + kc - 1, m) == e_pawn) return true;
            if (kc + 1 < 9 && vget(b, kr * 9 + kc + 1, m) == e_pawn) return true;
        }
    } else {
        if (kr + 1 < 10 && vget(b, (kr + 1) * 9 + kc, m) == e_pawn) return true;
        if (kr <= 4) {
            if (kc - 1 >= 0 && vget(b, kr * 9 + kc - 1, m) == e_pawn) return true;
            if (kc + 1 < 9 && vget(b, kr * 9 + kc + 1, m) == e_pawn) return true;
        }
    }
    return false;
}

// game_core.pyx:543-555
__device__ __forceinline__ bool in_check(const int8_t *b, int player) {
    const VMove none{-1, -1, 0};
    const int k = find_king(b, none, player);
    if (k < 0) return true;
    return is_attacked(b, none, k / 9, k % 9, -player);
}

// game_core.pyx:543-555 / 104-189 with the WAVE as the unit: the 47 probes of the reverse attack scan on the king's
// square are dealt to lanes -- lanes 0..35 hold the squares of the four rays (9 steps each; one ballot of "occupied" gives
// every ray's first and second piece: the first attacks as rook or king, the second as cannon), lanes 36..43 the eight
// knight origins with their legs, lanes 44..46 the pawn origins -- and a second ballot says whether any probe hit.
// Same boolean as in_check(); all 64 lanes must call it together.
__device__ __forceinline__ bool wave_in_check(const int8_t *b, int player) {
    const VMove none{-1, -1, 0};
    const int k = find_king(b, none, player);
    if (k < 0) return true;
    const int kr = k / 9, kc = k % 9, by = -player;
    const int lane = lane_id();
    int p = 0;
    if (lane < 36) {
        const int d = lane / 9, s = lane - d * 9 + 1;
        int dr, dc;
        ortho(d, dr, dc);
        const int r = kr + dr * s, c = kc + dc * s;
        if ((unsigned)r < 10u && (unsigned)c < 9u) p = b[r * 9 + c];
    }
    const unsigned long long occ = __ballot(p != 0);
    bool hit = false;
    if (lane < 36) {
        const int d = lane / 9, idx = lane - d * 9;
        const unsigned ray = (unsigned)(occ >> (d * 9)) & 0x1FFu;
        if (ray) {
            const unsigned rest = ray & (ray - 1u);
            if (idx == __builtin_ctz(ray)) hit = (p == 5 * by || p == by);
            else if (rest && idx == __builtin_ctz(rest)) hit = (p == 6 * by);
        }
    } else if (lane < 44) {
        int dr, dc, lr, lc;
        knight(lane - 36, dr, dc, lr, lc);
        const int nr = kr + dr, nc = kc + dc;
        if ((unsigned)nr < 10u && (unsigned)nc < 9u && b[nr * 9 + nc] == 4 * by) {
            const int br = (dr == 2 || dr == -2) ? nr - dr / 2 : nr;
            const int bc = (dr == 2 || dr == -2) ? nc : nc - dc / 2;
            hit = b[br * 9 + bc] == 0;
        }
    } else if (lane < 47) {
        const int j = lane - 44;                  // 0: the square a pawn advances from; 1, 2: sideways (after the river)
        const int pr = j == 0 ? kr - by : kr, pc = j == 0 ? kc : (j == 1 ? kc - 1 : kc + 1);
        const bool side_ok = j == 0 || (by == 1 ? kr >= 5 : kr <= 4);
        if (side_ok && (unsigned)pr < 10u && (unsigned)pc < 9u) hit = b[pr * 9 + pc] == 7 * by;
    }
    return __ballot(hit) != 0ull;
}

// game_core.pyx:209-252 -- legality of one pseudo-legal move, on the virtual board
__device__ __forceinline__ bool move_legal(const int8_t *b, int from, int to, int player) {
    const VMove m{from, to, (int)b[from]};
    const int k = find_king(b, m, player);
    if (k < 0) return false;
    const int kr = k / 9, kc = k % 9;
    const int e = find_king(b, m, -player);
    if (e >= 0 && (e % 9) == kc) {
        const int er = e / 9;
        const int lo = (kr < er ? kr : er) + 1, hi = kr < er ? er : kr;
        bool blocked = false;
        for (int r = lo; r < hi; ++r)
            if (vget(b, r * 9 + kc, m) != 0) { blocked = true; break; }
        if (!blocked) return false;   // flying general
    }
    return !is_attacked(b, m, kr, kc, -player);
}

// Phase A task: piece on `sq`, direction slot `slot` (0..7).  Result: targets sq+delta*1 .. sq+delta*q, then
// `extra` (or -1).  Emission order inside a piece is slot-major, which is the reference's order.
__device__ __forceinline__ void gen_task(const int8_t *b, int sq, int slot, int player, int &q, int &delta,
                                         int &extra) {
    q = 0; delta = 0; extra = -1;
    const int piece = b[sq];
    const int kind = piece < 0 ? -piece : piece;
    const int r = sq / 9, c = sq % 9;
    if (kind == 1) {                                        // king, pyx:287-304
        if (slot < 4) {
            int dr, dc; ortho(slot, dr, dc);
            const int nr = r + dr, nc = c + dc, lo = player == 1 ? 0 : 7;
            if (nr >= lo && nr <= lo + 2 && nc >= 3 && nc <= 5 && can_land(b[nr * 9 + nc], player)) extra = nr * 9 + nc;
        }
    } else if (kind == 2) {                                 // advisor, pyx:307-326
        if (slot < 4) {
            const int nr = r + ((slot & 2) ? 1 : -1), nc = c + ((slot & 1) ? 1 : -1);
            bool ok = (unsigned)nr < 10u && nc >= 3 && nc <= 5;
            ok = ok && (player == 1 ? nr <= 2 : nr >= 7);
            if (ok && can_land(b[nr * 9 + nc], player)) extra = nr * 9 + nc;
        }
    } else if (kind == 3) {                                 // bishop, pyx:329-349
        if (slot < 4) {
            const int dr = (slot & 2) ? 2 : -2, dc = (slot & 1) ? 2 : -2;
            const int nr = r + dr, nc = c + dc;
            bool ok = (unsigned)nr < 10u && (unsigned)nc < 9u;
            ok = ok && (player == 1 ? nr <= 4 : nr >= 5);
            if (ok && b[(r + dr / 2) * 9 + c + dc / 2] == 0 && can_land(b[nr * 9 + nc], player)) extra = nr * 9 + nc;
        }
    } else if (kind == 4) {                                 // knight, pyx:352-367
        int dr, dc, lr, lc; knight(slot, dr, dc, lr, lc);
        const int nr = r + dr, nc = c + dc;
        if ((unsigned)nr < 10u && (unsigned)nc < 9u && b[(r + lr) * 9 + c + lc] == 0 && can_land(b[nr * 9 + nc], player))
            extra = nr * 9 + nc;
    } else if (kind == 5 || kind == 6) {                    // rook pyx:370-396, cannon pyx:399-431
        if (slot < 4) {
            int dr, dc; ortho(slot, dr, dc);
            delta = dr * 9 + dc;
            int nr = r + dr, nc = c + dc;
            while ((unsigned)nr < 10u && (unsigned)nc < 9u && b[nr * 9 + nc] == 0) { ++q; nr += dr; nc += dc; }
            if ((unsigned)nr < 10u && (unsigned)nc < 9u) {  // (nr,nc) holds the first piece on the ray
                if (kind == 5) {
                    if (is_enemy(b[nr * 9 + nc], player)) ++q;   // the capture is the next contiguous step
                } else {
                    nr += dr; nc += dc;
                    while ((unsigned)nr < 10u && (unsigned)nc < 9u) {
                        const int p = b[nr * 9 + nc];
                        if (p != 0) { if (is_enemy(p, player)) extra = nr * 9 + nc; break; }
                        nr += dr; nc += dc;
                    }
                }
            }
        }
    } else if (kind == 7) {                                 // pawn, pyx:434-484
        const int fwd = player == 1 ? 1 : -1;
        const bool crossed = player == 1 ? r >= 5 : r <= 4;
        if (slot == 0) {
            const int nr = r + fwd;
            if ((unsigned)nr < 10u && can_land(b[nr * 9 + c], player)) extra = nr * 9 + c;
        } else if (slot == 1) {
            if (crossed && c - 1 >= 0 && can_land(b[sq - 1], player)) extra = sq - 1;
        } else if (slot == 2) {
            if (crossed && c + 1 < 9 && can_land(b[sq + 1], player)) extra = sq + 1;
        }
    }
}

// LDS scratch one wave needs for move generation
struct MoveGenLds {
    uint8_t pieces[XQ_BS];              // squares of own pieces, row-major order
    uint8_t cand_from[XQ_CAND_CAP];
    uint8_t cand_to[XQ_CAND_CAP];
};

// Ordered legal moves of `player` on LDS board `b` -> out[] (LDS or global, u16 action ids), returns the
// count (wave-uniform).  *overflow is OR-ed with 1 when a capacity was exceeded (list truncated).
// All 64 lanes of the wave must call this together.
__device__ inline int wave_movegen(const int8_t *b, int player, MoveGenLds &L, uint16_t *out, int *overflow) {
    const int lane = lane_id();
    // ---- own pieces in row-major order (ballot over squares 0..63 and 64..89)
    const bool own0 = is_own(b[lane], player);
    const bool own1 = (lane + 64 < 90) && is_own(b[lane + 64 < 90 ? lane + 64 : 0], player);
    const unsigned long long m0 = __ballot(own0), m1 = __ballot(own1);
    const int n0 = __popcll(m0), npieces = n0 + __popcll(m1);
    if (own0) L.pieces[lane_prefix(m0)] = (uint8_t)lane;
    if (own1) L.pieces[n0 + lane_prefix(m1)] = (uint8_t)(lane + 64);
    wave_sync();
    // ---- phase A: (piece, slot) tasks, 8 slots per piece
    int ncand = 0;
    const int ntasks = npieces * 8;
    for (int base = 0; base < ntasks; base += 64) {
        const int t = base + lane;
        int q = 0, delta = 0, extra = -1, sq = 0;
        if (t < ntasks) {
            sq = L.pieces[t >> 3];
            gen_task(b, sq, t & 7, player, q, delta, extra);
        }
        const int cnt = q + (extra >= 0 ? 1 : 0);
        int total;
        int off = ncand + wave_excl_scan(cnt, lane, &total);
        for (int s = 1; s <= q; ++s, ++off)
            if (off < XQ_CAND_CAP) { L.cand_from[off] = (uint8_t)sq; L.cand_to[off] = (uint8_t)(sq + delta * s); }
        if (extra >= 0 && off < XQ_CAND_CAP) { L.cand_from[off] = (uint8_t)sq; L.cand_to[off] = (uint8_t)extra; }
        ncand += total;
    }
    if (ncand > XQ_CAND_CAP) { ncand = XQ_CAND_CAP; *overflow |= 1; }
    wave_sync();
    // ---- phase B: legality on the virtual board, ordered compaction
    int nlegal = 0;
    for (int base = 0; base < ncand; base += 64) {
        const int i = base + lane;
        bool keep = false;
        int from = 0, to = 0;
        if (i < ncand) {
            from = L.cand_from[i]; to = L.cand_to[i];
            keep = move_legal(b, from, to, player);
        }
        const unsigned long long km = __ballot(keep);
        const int pos = nlegal + lane_prefix(km);
        if (keep) {
            if (pos < XQ_MAXM) out[pos] = (uint16_t)(from * 90 + to);
        }
        nlegal += __popcll(km);
    }
    if (nlegal > XQ_MAXM) { nlegal = XQ_MAXM; *overflow |= 1; }
    wave_sync();
    return nlegal;
}

// game.py:552-563 -- both sides at once (wave reduction)
__device__ __forceinline__ void wave_material(const int8_t *b, int &red, int &black) {
    const int lane = lane_id();
    int r = 0, k = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int sq = lane + 64 * j;
        if (sq < 90) {
            const int p = b[sq];
            const int a = p < 0 ? -p : p;
            const int v = (a == 2 || a == 3) ? 20 : (a == 4) ? 40 : (a == 5) ? 90 : (a == 6) ? 45 : (a == 7) ? 10 : 0;
            if (p > 0) r += v; else k += v;
        }
    }
    red = wave_sum(r);
    black = wave_sum(k);
}

// game.py:618-640 -- 15 planes, absolute orientation, plane 14 = all ones iff red to move
__device__ __forceinline__ void wave_encode(const int8_t *b, int player, float *out) {
    const int lane = lane_id();
    for (int e = lane; e < 1350; e += 64) {
        const int plane = e / 90, sq = e - plane * 90;
        float v;
        if (plane == 14) {
            v = player == 1 ? 1.0f : 0.0f;
        } else {
            const int p = b[sq];
            const int want = (plane < 7 ? plane + 1 : plane - 6) * (plane < 7 ? player : -player);
            v = (p == want) ? 1.0f : 0.0f;
        }
        out[e] = v;
    }
}

// load a 90-byte board (stride given) from global memory into a 96-byte LDS slot, padding zeroed
__device__ __forceinline__ void wave_load_board(const int8_t *g, int8_t *s) {
    const int lane = lane_id();
    s[lane] = g[lane];
    if (lane < 32) s[lane + 64] = (lane + 64 < 90) ? g[lane + 64] : (int8_t)0;
}

}  // namespace xq

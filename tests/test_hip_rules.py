"""GPU parity of the B1 kernels (xq_movegen_batch & co, through the C ABI) against the CPU oracle and
the golden fixtures recorded from the reference.  Bit-exact: integer / byte / index work."""
import zlib

import numpy as np
import pytest

import golden_io as G
from oracle import xq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import torch
    from xiangqi_alphazero_amd import hip as H
    H.lib()
    assert torch.cuda.is_available()
    return H


def _t(a, dtype=None):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _moves_lists(moves, counts):
    mv = moves.cpu().numpy().view(np.uint16)
    ct = counts.cpu().numpy().view(np.uint16)
    return [mv[i, :ct[i]] for i in range(len(ct))]


@pytest.mark.parametrize("which", ["corpus", "crafted"])
def test_movegen_ordered_vs_reference_fixture(hip, which):
    d = G.corpus() if which == "corpus" else G.crafted()
    boards, side = _t(d["board"]), _t(d["side"])
    moves, counts, chk, status = hip.movegen(boards, side)
    assert int(status.sum().item()) == 0
    got = _moves_lists(moves, counts)
    for i in range(len(got)):
        np.testing.assert_array_equal(got[i], G.moves_of(d, i), err_msg=f"{which}[{i}]")
    if which == "corpus":
        want = np.where(d["side"] == 1, d["check_red"], d["check_black"])
    else:
        want = np.where(d["side"] == 1, d["in_check"][:, 0], d["in_check"][:, 1])
    np.testing.assert_array_equal(chk.cpu().numpy(), want)


@pytest.mark.parametrize("which", ["corpus", "crafted"])
def test_attack_map_find_king_material(hip, which):
    d = G.corpus() if which == "corpus" else G.crafted()
    boards = _t(d["board"])
    am = hip.attack_map(boards).cpu().numpy()
    want = np.unpackbits(d["attacked"], axis=2)[:, :, :90]
    np.testing.assert_array_equal(am, want)
    kings = hip.find_king(boards).cpu().numpy()
    if which == "corpus":
        np.testing.assert_array_equal(kings[:, 0], d["king_red"])
        np.testing.assert_array_equal(kings[:, 1], d["king_black"])
        mat = hip.material(boards).cpu().numpy()
        np.testing.assert_array_equal(mat[:, 0], d["mat_red"])
        np.testing.assert_array_equal(mat[:, 1], d["mat_black"])
    else:
        np.testing.assert_array_equal(kings, d["kings"])


def test_encode_bit_exact(hip):
    d = G.corpus()
    enc = hip.encode(_t(d["board"]), _t(d["side"])).cpu().numpy()
    assert enc.shape == (len(d["board"]), 15, 10, 9) and enc.dtype == np.float32
    for i in range(0, len(enc), 7):
        assert zlib.crc32(enc[i].tobytes()) & 0xFFFFFFFF == d["state_crc"][i]
        np.testing.assert_array_equal(enc[i], O.encode_state(d["board"][i], int(d["side"][i])))


def test_game_over_all_terminal_kinds(hip):
    d = G.corpus()
    n = len(d["board"])
    hist = np.zeros((n, 12, 90), dtype=np.int8)
    for i in range(n):
        h = G.history_tail(d, i)
        hist[i, :len(h)] = h
    out = hip.game_over(_t(d["board"]), _t(d["side"]), _t(d["move_count"]), _t(d["no_capture"]), _t(hist)).cpu().numpy()
    np.testing.assert_array_equal(out[:, 0], d["done"])
    np.testing.assert_array_equal(out[:, 1], d["winner"])
    assert set(np.unique(d["winner"][d["done"] == 1])) == {-1, 0, 1}


def test_random_soup_vs_oracle(hip):
    """Seeded synthetic boards the fixtures do not hold, checked against the oracle (ragged / empty lists)."""
    rng = np.random.RandomState(4242)
    n = 4096
    boards = np.zeros((n, 90), dtype=np.int8)
    for i in range(n):
        boards[i, rng.randint(0, 3) * 9 + rng.randint(3, 6)] = 1
        boards[i, rng.randint(7, 10) * 9 + rng.randint(3, 6)] = -1
        k = rng.randint(0, 31)
        sq = rng.choice(90, k, replace=False)
        for s in sq:
            if boards[i, s] == 0:
                boards[i, s] = rng.randint(2, 8) * (1 if rng.rand() < 0.5 else -1)
    side = np.where(rng.rand(n) < 0.5, 1, -1).astype(np.int8)
    moves, counts, chk, status = hip.movegen(_t(boards), _t(side))
    got = _moves_lists(moves, counts)
    assert int(status.sum().item()) == 0
    chk = chk.cpu().numpy()
    for i in range(n):
        np.testing.assert_array_equal(got[i], O.legal_actions(boards[i], int(side[i])), err_msg=str(i))
        assert bool(chk[i]) == O.is_in_check(boards[i], int(side[i]))


def test_empty_batch_and_bad_args(hip):
    import torch
    e = torch.zeros((0, 90), dtype=torch.int8, device="cuda")
    s = torch.zeros(0, dtype=torch.int8, device="cuda")
    moves, counts, _, _ = hip.movegen(e, s)
    assert moves.shape == (0, 128) and counts.shape == (0,)
    assert hip.lib().xq_movegen_batch(None, None, 4, None, None, None, None, None) == -1


def _perft_levels(hip, depth):
    """Breadth-first perft: every frontier goes through xq_movegen_batch + xq_apply_moves_batch."""
    import torch
    boards = _t(O.initial_board().reshape(1, 90))
    side = torch.ones(1, dtype=torch.int8, device="cuda")
    totals = []
    for d in range(1, depth + 1):
        moves, counts, _, status = hip.movegen(boards, side)
        assert int(status.sum().item()) == 0
        cnt = counts.to(torch.int64) & 0xFFFF
        totals.append(int(cnt.sum().item()))
        if d == depth:
            break
        parent = torch.repeat_interleave(torch.arange(boards.shape[0], device="cuda", dtype=torch.int32), cnt)
        mask = torch.arange(128, device="cuda").unsqueeze(0) < cnt.unsqueeze(1)
        action = moves[mask].contiguous()
        boards, side = hip.apply_moves(boards, side, parent.contiguous(), action)
    return totals


def test_perft_1_to_4(hip):
    want = G.perft()
    assert _perft_levels(hip, 4) == [want[1], want[2], want[3], want[4]]


def test_perft_5_full_size(hip):
    """3.29 M positions through one launch (BASELINE-scale batch); size-independent check: the total."""
    want = G.perft()
    if 5 not in want:
        pytest.skip("perft(5) not recorded")
    assert _perft_levels(hip, 5)[-1] == want[5]


def test_game_core_plugin_signatures(hip):
    """The reference's plug point (game_core.pyx:493-569): same names, types and error behaviour."""
    from xiangqi_alphazero_amd import game_core as gc
    b = O.initial_board()
    mv = gc.cy_generate_legal_moves(b, 1)
    assert mv == O.legal_moves(b, 1) and len(mv) == 44 and isinstance(mv[0], tuple)
    assert gc.cy_is_in_check(b, 1) is False and gc.cy_find_king(b, -1) == (9, 4)
    assert gc.cy_has_legal_moves(b, -1) is True and gc.cy_is_attacked(b, 2, 4, -1) is False
    view = np.zeros((12, 11), dtype=np.int8)[1:11, 1:10]      # non-contiguous view is accepted (copied)
    view[:] = b
    assert gc.cy_generate_legal_moves(view, 1) == mv
    with pytest.raises(ValueError):
        gc.cy_generate_legal_moves(b.astype(np.int32), 1)
    kingless = b.copy(); kingless[0, 4] = 0
    assert gc.cy_find_king(kingless, 1) is None and gc.cy_is_in_check(kingless, 1) is True
    assert gc.cy_generate_legal_moves(kingless, 1) == []

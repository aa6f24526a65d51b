"""include/mzhist.h: native filing of move batches into game histories == the one-move-at-a-time numpy filing
of DeviceSelfPlay._file_move (host code only: runs without a GPU)."""
import importlib

import numpy as np


class _Harness:
    """The filing half of DeviceSelfPlay, without envs or searches."""

    def __init__(self, sp, E, L, shape, A):
        self.E = E
        self._obs = np.zeros((E, L + 1) + shape, np.float32)
        self._act = np.zeros((E, L + 1), np.int32)
        self._rew = np.zeros((E, L + 1), np.float32)
        self._tp = np.zeros((E, L + 1), np.int8)
        self._cv = np.zeros((E, L, A))
        self._rv = np.zeros((E, L))
        self._len = np.zeros(E, np.int64)
        self.games_finished = 0
        self._file = sp.DeviceSelfPlay._file_move.__get__(self)


def test_native_filer_matches_numpy_filing(pkg):
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    E, L, shape, A, S, M = 700, 40, (1, 1, 4), 3, 50, 6
    rs = np.random.RandomState(3)
    first = rs.standard_normal((E,) + shape).astype(np.float32)
    ref = _Harness(sp, E, L, shape, A)
    ref._obs[:, 0] = first
    filer = sp.HistoryFiler(E, L, shape, A)
    filer.begin(first)
    legal = np.tile(np.array([2, 0, 1], np.int32), (E, 1))         # child slot -> action, not the identity
    num_legal = np.full(E, A, np.int32)
    for batch in range(9):
        moves_done = rs.randint(0, M + 1, E).astype(np.int32)
        moves_done[rs.rand(E) < 0.6] = M
        visits = rs.multinomial(S, [0.5, 0.3, 0.2], (M, E)).astype(np.int32)
        out = dict(moves_done=moves_done, actions=rs.randint(0, A, (M, E)).astype(np.int32), visits=visits,
                   root_value_sum=rs.standard_normal((M, E)) * S)
        rewards = rs.standard_normal((M, E)).astype(np.float32)
        done = (rs.rand(M, E) < 0.12).astype(np.uint8)
        # a game must end before it outgrows the rows
        lengths = filer.lengths().copy()
        for m in range(M):
            playing = moves_done > m
            lengths = np.where(playing, lengths + 1, lengths)
            done[m][(lengths >= L - 1) & playing] = 1
            lengths = np.where((done[m] == 1) & playing, 0, lengths)
        obs_after = rs.standard_normal((M, E) + shape).astype(np.float32)
        obs_next = rs.standard_normal((M, E) + shape).astype(np.float32)
        want = []
        for m in range(M):
            cv = np.zeros((E, A))
            np.put_along_axis(cv, legal.astype(np.int64), visits[m] / float(S), axis=1)
            ref._file(moves_done > m, out["actions"][m], cv, out["root_value_sum"][m] / float(S), rewards[m],
                      done[m].astype(bool), obs_after[m], np.zeros(E, np.int32), obs_next[m], np.zeros(E, np.int32),
                      None, lambda b: want.extend((int(e), b.history(i)) for i, e in enumerate(b.env_index)))
        got_batch = filer.file(out, legal, num_legal, S, rewards, done, obs_after, obs_next)
        got = [] if got_batch is None else [(int(e), got_batch.history(i)) for i, e in enumerate(got_batch.env_index)]
        assert len(got) == len(want)
        key = lambda item: (item[0], len(item[1].action_history), item[1].action_history)   # noqa: E731
        for (ea, a), (eb, b) in zip(sorted(got, key=key), sorted(want, key=key)):
            assert ea == eb and a.action_history == b.action_history and a.to_play_history == b.to_play_history
            assert np.array_equal(np.array(a.reward_history, np.float32), np.array(b.reward_history, np.float32))
            assert np.array_equal(np.array(a.child_visits), np.array(b.child_visits)) and a.root_values == b.root_values
            assert all(np.array_equal(x, y) for x, y in zip(a.observation_history, b.observation_history))
        assert np.array_equal(filer.lengths(), ref._len)
    filer.close()


def test_native_filer_with_per_move_legal_sets_and_players(pkg):
    """mzhist_moves.legal_stride / num_legal_stride: a batch whose every move has its own legal set (board games,
    engine.moves_inputs) and two players, filed in one call == the same moves filed one at a time with that move's set."""
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    E, L, shape, A, S, M = 600, 12, (3, 3, 3), 9, 25, 7
    rs = np.random.RandomState(11)
    first = rs.standard_normal((E,) + shape).astype(np.float32)
    first_tp = rs.randint(0, 2, E).astype(np.int32)
    whole, single = sp.HistoryFiler(E, L, shape, A), sp.HistoryFiler(E, L, shape, A)
    whole.begin(first, first_tp)
    single.begin(first, first_tp)
    for batch in range(6):
        moves_done = np.full(E, M, np.int32)
        moves_done[rs.rand(E) < 0.2] = rs.randint(0, M, int((rs.rand(E) < 0.2).sum() or 1))[0]
        num_legal = rs.randint(1, A + 1, (M, E)).astype(np.int32)
        legal = np.stack([np.stack([rs.permutation(A) for _ in range(E)]) for _ in range(M)]).astype(np.int32)
        visits = np.zeros((M, E, A), np.int32)
        for m in range(M):
            for e in range(0, E, 7):                                       # (a sparse fill keeps the test fast)
                n = int(num_legal[m, e])
                visits[m, e, :n] = rs.multinomial(S, np.full(n, 1.0 / n))
        out = dict(moves_done=moves_done, actions=rs.randint(0, A, (M, E)).astype(np.int32), visits=visits,
                   root_value_sum=rs.standard_normal((M, E)) * S)
        rewards = rs.standard_normal((M, E)).astype(np.float32)
        done = (rs.rand(M, E) < 0.2).astype(np.uint8)
        lengths = whole.lengths().copy()
        for m in range(M):
            playing = moves_done > m
            lengths = np.where(playing, lengths + 1, lengths)
            done[m][(lengths >= L - 1) & playing] = 1
            lengths = np.where((done[m] == 1) & playing, 0, lengths)
        obs_after = rs.standard_normal((M, E) + shape).astype(np.float32)
        obs_next = rs.standard_normal((M, E) + shape).astype(np.float32)
        tp_after = rs.randint(0, 2, (M, E)).astype(np.int32)
        tp_next = rs.randint(0, 2, (M, E)).astype(np.int32)

        def games(b):
            return [] if b is None else [(int(e), b.history(i)) for i, e in enumerate(b.env_index)]
        got = games(whole.file(out, legal, num_legal, S, rewards, done, obs_after, obs_next, to_play_after=tp_after,
                               to_play_next=tp_next))
        want = []
        for m in range(M):
            one = dict(moves_done=(moves_done > m).astype(np.int32), actions=out["actions"][m][None], visits=visits[m][None],
                       root_value_sum=out["root_value_sum"][m][None])
            want += games(single.file(one, legal[m], num_legal[m], S, rewards[m][None], done[m][None], obs_after[m][None],
                                      obs_next[m][None], to_play_after=tp_after[m][None], to_play_next=tp_next[m][None]))
        assert len(got) == len(want) > 0
        key = lambda item: (item[0], len(item[1].action_history), item[1].action_history)   # noqa: E731
        for (ea, a), (eb, b) in zip(sorted(got, key=key), sorted(want, key=key)):
            assert ea == eb and a.action_history == b.action_history and a.to_play_history == b.to_play_history
            assert a.reward_history == b.reward_history and a.root_values == b.root_values
            assert np.array_equal(np.array(a.child_visits), np.array(b.child_visits))
            assert all(np.array_equal(x, y) for x, y in zip(a.observation_history, b.observation_history))
        assert np.array_equal(whole.lengths(), single.lengths())
    whole.close()
    single.close()


def test_native_filer_rejects_legal_sets_outside_the_action_space(pkg):
    """The legal sets of a batch are copied back from the device: a count outside [0, A] or an action outside [0, A)
    is an error (RuntimeError with the library's message), never an out-of-bounds write into a history row; the rows
    are left untouched and a clean batch files afterwards."""
    import pytest
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    E, L, shape, A, S, M = 8, 10, (1, 1, 4), 3, 10, 2
    rs = np.random.RandomState(1)
    filer = sp.HistoryFiler(E, L, shape, A)
    filer.begin(rs.standard_normal((E,) + shape).astype(np.float32))
    out = dict(moves_done=np.full(E, M, np.int32), actions=rs.randint(0, A, (M, E)).astype(np.int32),
               visits=rs.multinomial(S, [0.5, 0.3, 0.2], (M, E)).astype(np.int32), root_value_sum=rs.standard_normal((M, E)))
    rewards, done = np.zeros((M, E), np.float32), np.zeros((M, E), np.uint8)
    obs = rs.standard_normal((M, E) + shape).astype(np.float32)
    good_legal, good_n = np.tile(np.arange(A, dtype=np.int32), (E, 1)), np.full(E, A, np.int32)
    for legal, n in ((good_legal, np.where(np.arange(E) == 5, A + 1, A).astype(np.int32)),
                     (good_legal, np.where(np.arange(E) == 2, -1, A).astype(np.int32)),
                     (np.where(np.arange(E)[:, None] == 3, 7, good_legal).astype(np.int32), good_n),
                     (np.where(np.arange(E)[:, None] == 0, -2, good_legal).astype(np.int32), good_n)):
        with pytest.raises(RuntimeError, match="legal"):
            filer.file(out, legal, n, S, rewards, done, obs, obs)
        assert (filer.lengths() == 0).all()
    assert filer.file(out, good_legal, good_n, S, rewards, done, obs, obs) is None
    assert (filer.lengths() == M).all()
    filer.close()

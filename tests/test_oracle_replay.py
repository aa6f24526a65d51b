"""Oracle of the replay-target row against the reference's own ReplayBuffer (fixtures G12)."""
import importlib
import os
import sys

import numpy as np
import pytest

from parity_helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

NAMES = ["cartpole", "tictactoe", "tictactoe_stacked", "cartpole_uniform"]


def games_of(fx, ro):
    games = []
    for g, n in enumerate(fx["lengths"]):
        games.append(ro.Game(fx["observations"][g, : n + 1], fx["actions"][g, : n + 1], fx["rewards"][g, : n + 1],
                             fx["to_play"][g, : n + 1], fx["child_visits"][g, :n], fx["root_values"][g, :n]))
    return games


def cfg_of(fx):
    return dict(batch_size=int(fx["batch_size"]), PER=bool(fx["PER"]), td_steps=int(fx["td_steps"]),
                discount=float(fx["cfg_discount"]), num_unroll_steps=int(fx["num_unroll_steps"]),
                action_space=list(range(int(fx["cfg_A"]))), stacked_observations=int(fx["stacked_observations"]))


@pytest.mark.parametrize("name", NAMES)
def test_replay_oracle_matches_reference(oracle, name):
    ro = importlib.import_module("replay_oracle")
    fx = load_golden(f"g12_replay_{name}")
    games, cfg = games_of(fx, ro), cfg_of(fx)
    if cfg["PER"]:
        for g, game in enumerate(games):
            pri = ro.initial_priorities(game, cfg["td_steps"], cfg["discount"], float(fx["PER_alpha"]))
            assert np.array_equal(pri, fx["priorities"][g, : len(pri)]), g       # float32, bit for bit
            assert game.game_priority == fx["game_priority"][g]
    rng = oracle.Rng(int(fx["seed"]))
    out = ro.get_batch(games, cfg, rng)
    assert np.array_equal(np.array(out["index"]), fx["index_batch"])
    assert np.array_equal(np.array(out["action"]), fx["action_batch"])
    assert np.array_equal(np.array(out["value"], dtype=np.float64), fx["value_batch"])
    assert np.array_equal(np.array(out["reward"], dtype=np.float64), fx["reward_batch"])
    assert np.array_equal(np.array(out["policy"], dtype=np.float64), fx["policy_batch"])
    assert np.array_equal(np.array(out["gradient_scale"], dtype=np.float64), fx["gradient_scale_batch"])
    assert np.array_equal(np.array(out["observation"], dtype=np.float32), fx["observation_batch"])
    if cfg["PER"]:
        assert np.array_equal(out["weight"], fx["weight_batch"])


def test_replay_oracle_with_reanalysed_values(oracle):
    ro = importlib.import_module("replay_oracle")
    fx = load_golden("g13_reanalyse_cartpole")
    games = games_of(fx, ro)
    for g, game in enumerate(games):
        game.reanalysed = fx["reanalysed"][g, : len(game.root_values)].copy()
    rng = oracle.Rng(int(fx["seed"]))
    for i, (g, pos) in enumerate(fx["pairs"]):
        v, r, p, a = ro.make_target(games[g], int(pos), int(fx["td_steps"]), float(fx["cfg_discount"]),
                                    int(fx["num_unroll_steps"]), list(range(int(fx["cfg_A"]))), rng)
        assert np.array_equal(np.array([float(x) for x in v]), fx["value_targets"][i]), (g, pos)
        assert np.array_equal(np.array(r, dtype=np.float64), fx["reward_targets"][i])
        assert np.array_equal(np.array(p, dtype=np.float64), fx["policy_targets"][i])
        assert np.array_equal(np.array(a), fx["action_targets"][i])

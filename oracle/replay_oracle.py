"""Oracle (TEST INFRASTRUCTURE ONLY) for SURVEY.md section 8(f) row 2: GameHistory -> replay targets.

A plain Python / numpy restatement of the reference's replay_buffer.py for the functions that turn finished
games into training targets, operation for operation (Python floats = fp64, numpy float32 where the reference
uses float32):

    initial_priorities      ReplayBuffer.save_game                 replay_buffer.py:33-50
    compute_target_value    ReplayBuffer.compute_target_value       replay_buffer.py:222-256
    make_target             ReplayBuffer.make_target                replay_buffer.py:258-295
    stacked_observations    GameHistory.get_stacked_observations    self_play.py:514-548
    sample_* / get_batch    ReplayBuffer.get_batch and its samplers replay_buffer.py:67-195
    (reanalysed values)     Reanalyse.reanalyse's per-game step     replay_buffer.py:335-356, 226-231

PINNED against fixtures G12 / G13 (tests/golden/g12_replay_*.npz, g13_reanalyse_cartpole.npz), recorded by running the reference's own
ReplayBuffer on synthetic game histories (tests/golden/make_golden.py:g12_replay_targets).  Random draws go
through mz_oracle.Rng, the numpy legacy RandomState clone pinned by fixture G7.  Only tests/ may import this.
"""
import numpy as np


class Game:
    """The GameHistory fields the replay path reads, as arrays (lengths: n moves -> n+1 observations)."""

    def __init__(self, observations, actions, rewards, to_play, child_visits, root_values):
        self.observations = np.asarray(observations, dtype=np.float32)
        self.actions = [int(a) for a in actions]
        self.rewards = [float(r) for r in rewards]
        self.to_play = [int(t) for t in to_play]
        self.child_visits = [[float(v) for v in row] for row in child_visits]
        self.root_values = [float(v) for v in root_values]
        self.priorities = None
        self.game_priority = None
        self.reanalysed = None          # numpy float32 array once Reanalyse has visited the game


def compute_target_value(game, index, td_steps, discount):
    bootstrap = index + td_steps
    if bootstrap < len(game.root_values):
        # reanalysed values are numpy float32 scalars: under NumPy 2 promotion the whole sum then runs in
        # float32 (Python floats are weak), which is what fixture G13 records
        last = (game.root_values if game.reanalysed is None else game.reanalysed)[bootstrap]
        if game.to_play[bootstrap] != game.to_play[index]:
            last = -last
        value = last * discount ** td_steps
    else:
        value = 0
    for i, reward in enumerate(game.rewards[index + 1: bootstrap + 1]):
        signed = reward if game.to_play[index] == game.to_play[index + i] else -reward
        value += signed * discount ** i
    return value


def initial_priorities(game, td_steps, discount, alpha):
    pri = [np.abs(rv - compute_target_value(game, i, td_steps, discount)) ** alpha
           for i, rv in enumerate(game.root_values)]
    game.priorities = np.array(pri, dtype="float32")
    game.game_priority = np.max(game.priorities)
    return game.priorities


def make_target(game, state_index, td_steps, discount, unroll, action_space, rng):
    values, rewards, policies, actions = [], [], [], []
    n = len(game.root_values)
    width = len(game.child_visits[0])
    for cur in range(state_index, state_index + unroll + 1):
        value = compute_target_value(game, cur, td_steps, discount)
        if cur < n:
            values.append(value)
            rewards.append(game.rewards[cur])
            policies.append(game.child_visits[cur])
            actions.append(game.actions[cur])
        elif cur == n:
            values.append(0)
            rewards.append(game.rewards[cur])
            policies.append([1 / width for _ in range(width)])
            actions.append(game.actions[cur])
        else:  # absorbing states: numpy.random.choice(action_space)
            values.append(0)
            rewards.append(0)
            policies.append([1 / width for _ in range(width)])
            actions.append(action_space[rng.below(len(action_space))])
    return values, rewards, policies, actions


def stacked_observations(game, index, num_stacked):
    index = index % len(game.observations)
    stacked = game.observations[index].copy()
    for past in reversed(range(index - num_stacked, index)):
        if 0 <= past:
            prev = np.concatenate((game.observations[past],
                                   [np.ones_like(stacked[0]) * game.actions[past + 1]]))
        else:
            prev = np.concatenate((np.zeros_like(game.observations[index]), [np.zeros_like(stacked[0])]))
        stacked = np.concatenate((stacked, prev))
    return stacked


def get_batch(games, cfg, rng):
    """cfg: dict(batch_size, PER, td_steps, discount, num_unroll_steps, action_space, stacked_observations).
    games: list in buffer order (game ids 0..G-1).  Returns the reference's get_batch structure as arrays."""
    G = len(games)
    total_samples = sum(len(g.root_values) for g in games)
    if cfg["PER"]:
        probs = np.array([g.game_priority for g in games], dtype="float32")
        probs /= np.sum(probs)
        picked = [rng.choice_p(probs) for _ in range(cfg["batch_size"])]
    else:
        probs = None
        picked = [rng.below(G) for _ in range(cfg["batch_size"])]
    out = dict(index=[], observation=[], action=[], value=[], reward=[], policy=[], gradient_scale=[], weight=[])
    for gid in picked:
        game = games[gid]
        if cfg["PER"]:
            pos_probs = game.priorities / sum(game.priorities)
            pos = rng.choice_p(pos_probs)
            pos_prob = pos_probs[pos]
        else:
            pos = rng.below(len(game.root_values))
            pos_prob = None
        v, r, p, a = make_target(game, pos, cfg["td_steps"], cfg["discount"], cfg["num_unroll_steps"],
                                 cfg["action_space"], rng)
        out["index"].append([gid, pos])
        out["observation"].append(stacked_observations(game, pos, cfg["stacked_observations"]))
        out["action"].append(a)
        out["value"].append(v)
        out["reward"].append(r)
        out["policy"].append(p)
        out["gradient_scale"].append([min(cfg["num_unroll_steps"], len(game.actions) - pos)] * len(a))
        if cfg["PER"]:
            out["weight"].append(1 / (total_samples * probs[gid] * pos_prob))
    if cfg["PER"]:
        out["weight"] = np.array(out["weight"], dtype="float32") / max(out["weight"])
    return out

"""Drop-in for the reference's self_play.py: SelfPlay, MCTS, Node, GameHistory, MinMaxStats.

Same class names, method names, signatures and return values (reference self_play.py:11-568), so
`muzero.py`-style orchestration, `diagnose_model.py`-style inspection and the replay buffer / trainer
keep working -- but the search itself runs on the MI355X:

  * `MCTS(config).run(...)` drives the HIP engine (engine.BatchedMCTS, one tree) and returns a `Node`
    tree materialised from the device pools plus the same `extra_info` dict;
  * `SelfPlay` is a plain class (no Ray): `.play_game`, `.continuous_self_play`, `.close_game`,
    `.select_opponent_action` and the static `.select_action`;
  * `BatchedSelfPlay` is the MI355X-native way to use the engine: E games in lock step, one actor per
    GPU, each env on the RNG stream of reference worker `config.seed + e`.

Randomness: the reference uses numpy's global legacy generator for everything.  The single-tree
facade hands that global state to the engine's stream before a search and hands it back afterwards
(`numpy.random.get_state/set_state`), so a run interleaves with any other user of `numpy.random`
(games, expert agents) exactly as the reference does.
"""
import math
import time

import numpy
import torch

from . import _native, models
from .engine import BatchedMCTS


class SelfPlay:
    """Plays games with MCTS at every move and hands them to the replay buffer
    (reference self_play.py:11-246)."""

    def __init__(self, initial_checkpoint, Game, config, seed):
        self.config = config
        self.game = Game(seed)

        # Fix random generator seed (self_play.py:22-23)
        numpy.random.seed(seed)
        torch.manual_seed(seed)

        self.model = models.MuZeroNetwork(self.config)
        self.model.set_weights(initial_checkpoint["weights"])
        self.model.to(torch.device("cuda" if torch.cuda.is_available() else "cpu"))
        self.model.eval()
        self._mcts = None

    def _searcher(self):
        if self._mcts is None:
            self._mcts = MCTS(self.config)
        return self._mcts

    def continuous_self_play(self, shared_storage, replay_buffer, test_mode=False):
        """Game loop (self_play.py:31-108).  `shared_storage` / `replay_buffer` are any objects with
        `get_info` / `set_info` / `save_game` methods (called directly: there is no Ray here)."""
        while (shared_storage.get_info("training_step") < self.config.training_steps
               and not shared_storage.get_info("terminate")):
            self.model.set_weights(shared_storage.get_info("weights"))

            if not test_mode:
                game_history = self.play_game(
                    self.config.visit_softmax_temperature_fn(
                        trained_steps=shared_storage.get_info("training_step")),
                    self.config.temperature_threshold, False, "self", 0)
                replay_buffer.save_game(game_history, shared_storage)
            else:
                # Take the best action (no exploration) in test mode
                game_history = self.play_game(
                    0, self.config.temperature_threshold, False,
                    "self" if len(self.config.players) == 1 else self.config.opponent,
                    self.config.muzero_player)
                shared_storage.set_info({
                    "episode_length": len(game_history.action_history) - 1,
                    "total_reward": sum(game_history.reward_history),
                    "mean_value": numpy.mean([value for value in game_history.root_values if value]),
                })
                if 1 < len(self.config.players):
                    mine = self.config.muzero_player
                    shared_storage.set_info({
                        "muzero_reward": sum(
                            reward for i, reward in enumerate(game_history.reward_history)
                            if game_history.to_play_history[i - 1] == mine),
                        "opponent_reward": sum(
                            reward for i, reward in enumerate(game_history.reward_history)
                            if game_history.to_play_history[i - 1] != mine),
                    })

            # Managing the self-play / training ratio
            if not test_mode and self.config.self_play_delay:
                time.sleep(self.config.self_play_delay)
            if not test_mode and self.config.ratio:
                while (shared_storage.get_info("training_step")
                       / max(1, shared_storage.get_info("num_played_steps")) < self.config.ratio
                       and shared_storage.get_info("training_step") < self.config.training_steps
                       and not shared_storage.get_info("terminate")):
                    time.sleep(0.5)

        self.close_game()

    def play_game(self, temperature, temperature_threshold, render, opponent, muzero_player):
        """One game, MCTS at every move (self_play.py:110-184)."""
        game_history = GameHistory()
        observation = self.game.reset()
        game_history.action_history.append(0)
        game_history.observation_history.append(observation)
        game_history.reward_history.append(0)
        game_history.to_play_history.append(self.game.to_play())

        done = False
        if render:
            self.game.render()

        with torch.no_grad():
            while not done and len(game_history.action_history) <= self.config.max_moves:
                shape = numpy.array(observation).shape
                assert len(shape) == 3, (
                    f"Observation should be 3 dimensionnal instead of {len(shape)} dimensionnal. "
                    f"Got observation of shape: {shape}")
                assert shape == self.config.observation_shape, (
                    "Observation should match the observation_shape defined in MuZeroConfig. "
                    f"Expected {self.config.observation_shape} but got {shape}.")
                stacked_observations = game_history.get_stacked_observations(
                    -1, self.config.stacked_observations)

                # Choose the action
                if opponent == "self" or muzero_player == self.game.to_play():
                    root, mcts_info = self._searcher().run(
                        self.model, stacked_observations, self.game.legal_actions(),
                        self.game.to_play(), True)
                    action = self.select_action(
                        root,
                        temperature
                        if not temperature_threshold
                        or len(game_history.action_history) < temperature_threshold
                        else 0)
                    if render:
                        print(f'Tree depth: {mcts_info["max_tree_depth"]}')
                        print(f"Root value for player {self.game.to_play()}: {root.value():.2f}")
                else:
                    action, root = self.select_opponent_action(opponent, stacked_observations)

                observation, reward, done = self.game.step(action)

                if render:
                    print(f"Played action: {self.game.action_to_string(action)}")
                    self.game.render()

                game_history.store_search_statistics(root, self.config.action_space)

                # Next batch
                game_history.action_history.append(action)
                game_history.observation_history.append(observation)
                game_history.reward_history.append(reward)
                game_history.to_play_history.append(self.game.to_play())

        return game_history

    def close_game(self):
        self.game.close()
        if self._mcts is not None:
            self._mcts.close()
            self._mcts = None

    def select_opponent_action(self, opponent, stacked_observations):
        """Opponent move when evaluating MuZero (self_play.py:189-221)."""
        if opponent == "human":
            root, mcts_info = self._searcher().run(
                self.model, stacked_observations, self.game.legal_actions(), self.game.to_play(), True)
            print(f'Tree depth: {mcts_info["max_tree_depth"]}')
            print(f"Root value for player {self.game.to_play()}: {root.value():.2f}")
            print(f"Player {self.game.to_play()} turn. MuZero suggests "
                  f"{self.game.action_to_string(self.select_action(root, 0))}")
            return self.game.human_to_action(), root
        elif opponent == "expert":
            return self.game.expert_agent(), None
        elif opponent == "random":
            assert self.game.legal_actions(), (
                f"Legal actions should not be an empty array. Got {self.game.legal_actions()}.")
            assert set(self.game.legal_actions()).issubset(set(self.config.action_space)), (
                "Legal actions should be a subset of the action space.")
            return numpy.random.choice(self.game.legal_actions()), None
        else:
            raise NotImplementedError(
                'Wrong argument: "opponent" argument should be "self", "human", "expert" or "random"')

    @staticmethod
    def select_action(node, temperature):
        """Sample the played action from the root's visit counts (self_play.py:223-246).

        Uses the engine's numpy-legacy generator on numpy's global state, so the draw (and the number
        of random words consumed) is the reference's."""
        visit_counts = numpy.array([child.visit_count for child in node.children.values()], dtype="int32")
        actions = [action for action in node.children.keys()]
        if temperature == 0:
            return actions[numpy.argmax(visit_counts)]
        rng = _native.HostRng(0)
        rng.set_state(numpy.random.get_state())
        slot = rng.select_action(visit_counts, temperature)
        numpy.random.set_state(rng.get_state())
        return actions[slot]


# Game independent
class MCTS:
    """Monte-Carlo tree search (self_play.py:250-431), executed by the HIP engine.

    `run` keeps the reference signature.  `select_child`, `ucb_score` and `backpropagate` are kept
    for callers that drive a search by hand on `Node` objects (diagnose-style tooling); they are
    host-side conveniences and are not used by `run`."""

    def __init__(self, config):
        self.config = config
        self._engine = None

    def _get_engine(self, device, model):
        if self._engine is None:
            self._engine = BatchedMCTS(self.config, 1, device=device, seeds=[0],
                                       group_width=16 if len(self.config.action_space) <= 16 else 0)
            self._engine_model = None
        if self.config.network == "fullyconnected" and self._engine_model is not model:
            # fully-connected networks run the whole search in one fused HIP launch; the model's parameters
            # become views into one flat buffer (values unchanged, set_weights keeps working)
            try:
                self._engine.configure_fused_fc(model)
            except (NotImplementedError, RuntimeError):
                pass                                # shape outside the fused kernel's range: lock-step path
            self._engine_model = model
        return self._engine

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def run(self, model, observation, legal_actions, to_play, add_exploration_noise,
            override_root_with=None):
        """Search from `observation`; returns (root Node, {"max_tree_depth", "root_predicted_value"})."""
        if override_root_with:
            return self._run_from_root(model, override_root_with, add_exploration_noise)
        device = next(model.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("MCTS.run needs the model on a HIP device; the engine has no CPU fallback")
        assert legal_actions, f"Legal actions should not be an empty array. Got {legal_actions}."
        assert set(legal_actions).issubset(set(self.config.action_space)), (
            "Legal actions should be a subset of the action space.")
        engine = self._get_engine(device, model)
        engine.set_rng_state(0, numpy.random.get_state())
        obs = torch.tensor(numpy.asarray(observation)).float().unsqueeze(0)
        stats = engine.search(model, obs, [list(legal_actions)], [to_play], add_exploration_noise)
        numpy.random.set_state(engine.get_rng_state(0))

        root = Node._from_engine(engine, list(legal_actions), to_play, len(self.config.players))
        extra_info = {
            "max_tree_depth": int(stats["max_tree_depth"][0]),
            "root_predicted_value": float(stats["root_predicted_value"][0]),
        }
        return root, extra_info

    def _run_from_root(self, model, root_in, add_exploration_noise):
        """`run(..., override_root_with=root)` (self_play.py:276-278; diagnose_model.py builds such roots with
        Node(0).expand(...)): the given root's children (priors), reward, to_play and hidden state are uploaded as the
        search's root instead of being computed from an observation; root_predicted_value is None as in the reference.
        The root must be freshly expanded -- trees are rebuilt in the device pools per search, so visit counts of an
        earlier search cannot be continued."""
        if not root_in.expanded() or root_in.hidden_state is None:
            raise AssertionError("override_root_with must be an expanded node carrying a hidden state")
        if root_in.visit_count or any(c.visit_count or c.expanded() for c in root_in.children.values()):
            raise NotImplementedError("override_root_with: only a freshly expanded root (no visits yet) can be uploaded "
                                      "to the device engine")
        device = next(model.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("MCTS.run needs the model on a HIP device; the engine has no CPU fallback")
        legal_actions = list(root_in.children.keys())
        assert set(legal_actions).issubset(set(self.config.action_space)), (
            "Legal actions should be a subset of the action space.")
        engine = self._get_engine(device, model)
        engine.set_rng_state(0, numpy.random.get_state())
        priors = numpy.zeros((1, engine.A), dtype=numpy.float64)
        priors[0, : len(legal_actions)] = [root_in.children[a].prior for a in legal_actions]
        with torch.no_grad(), torch.cuda.device(engine.device):
            engine.begin_search([legal_actions], [root_in.to_play], add_exploration_noise)
            engine.expand_roots_injected(numpy.array([float(root_in.reward)]), priors)
            engine.pool[0, 0].copy_(torch.as_tensor(root_in.hidden_state).to(engine.device, torch.float32).reshape(-1))
            engine._run_simulations(model)
            stats = engine.readout()
        numpy.random.set_state(engine.get_rng_state(0))
        root = Node._from_engine(engine, legal_actions, root_in.to_play, len(self.config.players))
        return root, {"max_tree_depth": int(stats["max_tree_depth"][0]), "root_predicted_value": None}

    def select_child(self, node, min_max_stats):
        """Child with the highest UCB score, ties broken like the reference (self_play.py:364-379)."""
        scored = [(self.ucb_score(node, child, min_max_stats), action)
                  for action, child in node.children.items()]
        top = max(score for score, _ in scored)
        action = numpy.random.choice([action for score, action in scored if score == top])
        return action, node.children[action]

    def ucb_score(self, parent, child, min_max_stats):
        """Prior exploration bonus + normalised value (self_play.py:381-405)."""
        c = self.config
        pb_c = math.log((parent.visit_count + c.pb_c_base + 1) / c.pb_c_base) + c.pb_c_init
        pb_c *= math.sqrt(parent.visit_count) / (child.visit_count + 1)
        score = pb_c * child.prior
        if child.visit_count > 0:
            q = child.value() if len(c.players) == 1 else -child.value()
            score += min_max_stats.normalize(child.reward + c.discount * q)
        return score

    def backpropagate(self, search_path, value, to_play, min_max_stats):
        """Propagate a leaf evaluation to the root (self_play.py:407-431)."""
        n_players = len(self.config.players)
        if n_players > 2:
            raise NotImplementedError("More than two player mode not implemented.")
        gamma = self.config.discount
        for node in reversed(search_path):
            if n_players == 1:
                node.value_sum += value
                node.visit_count += 1
                min_max_stats.update(node.reward + gamma * node.value())
                value = node.reward + gamma * value
            else:
                mine = node.to_play == to_play
                node.value_sum += value if mine else -value
                node.visit_count += 1
                min_max_stats.update(node.reward + gamma * -node.value())
                value = (-node.reward if mine else node.reward) + gamma * value


class Node:
    """Search-tree node with the reference's attributes (self_play.py:434-477).  Nodes returned by
    `MCTS.run` are views materialised from the engine's device pools after the search."""

    def __init__(self, prior):
        self.visit_count = 0
        self.to_play = -1
        self.prior = prior
        self.value_sum = 0
        self.children = {}
        self.hidden_state = None
        self.reward = 0

    def expanded(self):
        return len(self.children) > 0

    def value(self):
        if self.visit_count == 0:
            return 0
        return self.value_sum / self.visit_count

    def expand(self, actions, to_play, reward, policy_logits, hidden_state):
        """Give the node children with softmax priors over `actions` (self_play.py:452-466)."""
        self.to_play = to_play
        self.reward = reward
        self.hidden_state = hidden_state
        priors = torch.softmax(torch.tensor([policy_logits[0][a] for a in actions]), dim=0).tolist()
        for action, p in zip(actions, priors):
            self.children[action] = Node(p)

    def add_exploration_noise(self, dirichlet_alpha, exploration_fraction):
        """Mix Dirichlet noise into the children's priors (self_play.py:468-477)."""
        actions = list(self.children.keys())
        noise = numpy.random.dirichlet([dirichlet_alpha] * len(actions))
        frac = exploration_fraction
        for a, n in zip(actions, noise):
            self.children[a].prior = self.children[a].prior * (1 - frac) + n * frac

    @classmethod
    def _from_engine(cls, engine, legal_actions, to_play, n_players, env=0):
        """Rebuild the whole tree of `env` from the exported child-record pools."""
        tree = engine.export_tree(env)
        stats = engine.stats
        state_shape = engine.state_shape

        def hidden(k):
            return engine.pool[k, env].view(1, *state_shape)

        root = cls(0)
        root.visit_count = int(stats["root_visits"][env])
        root.value_sum = float(stats["root_value_sum"][env])
        root.to_play = to_play
        root.reward = 0.0
        root.hidden_state = hidden(0)
        stack = [(root, 0, 0)]  # (node, expanded-node index, tree depth)
        while stack:
            node, k, depth = stack.pop()
            actions = legal_actions if k == 0 else range(engine.A)
            for slot, action in enumerate(actions):
                child = cls(float(tree["prior"][k, slot]))
                child.visit_count = int(tree["visits"][k, slot])
                child.value_sum = float(tree["value_sum"][k, slot])
                child.reward = float(tree["reward"][k, slot])
                node.children[action] = child
                ck = int(tree["child_node"][k, slot])
                if ck >= 0:
                    child.to_play = (to_play + depth + 1) % n_players
                    child.hidden_state = hidden(ck)
                    stack.append((child, ck, depth + 1))
        return root


class GameHistory:
    """What is stored of a self-play game (self_play.py:480-548)."""

    def __init__(self):
        self.observation_history = []
        self.action_history = []
        self.reward_history = []
        self.to_play_history = []
        self.child_visits = []
        self.root_values = []
        self.reanalysed_predicted_root_values = None
        # For PER
        self.priorities = None
        self.game_priority = None

    def store_search_statistics(self, root, action_space):
        """Turn the root's visit counts into a policy target (self_play.py:497-512)."""
        if root is None:
            self.root_values.append(None)
            return
        total = sum(child.visit_count for child in root.children.values())
        self.child_visits.append(
            [root.children[a].visit_count / total if a in root.children else 0 for a in action_space])
        self.root_values.append(root.value())

    def get_stacked_observations(self, index, num_stacked_observations):
        """Observation at `index` followed by the previous observations, each with the plane of the
        action that led out of it; zero planes before the start of the game (self_play.py:514-548)."""
        index = index % len(self.observation_history)
        planes = [self.observation_history[index].copy()]
        like_plane = numpy.ones_like(planes[0][0])
        for past in range(index - 1, index - num_stacked_observations - 1, -1):
            if past >= 0:
                planes.append(self.observation_history[past])
                planes.append([like_plane * self.action_history[past + 1]])
            else:
                planes.append(numpy.zeros_like(self.observation_history[index]))
                planes.append([numpy.zeros_like(planes[0][0])])
        return numpy.concatenate(planes) if len(planes) > 1 else planes[0]


class MinMaxStats:
    """Running min / max of the values seen in one search tree (self_play.py:551-568)."""

    def __init__(self):
        self.maximum = -float("inf")
        self.minimum = float("inf")

    def update(self, value):
        self.maximum = max(self.maximum, value)
        self.minimum = min(self.minimum, value)

    def normalize(self, value):
        if self.maximum > self.minimum:
            # only once both bounds have been set
            return (value - self.minimum) / (self.maximum - self.minimum)
        return value


class ManyEnvLoop:
    """`continuous_self_play` for the many-env actors (BatchedSelfPlay, DeviceSelfPlay): the reference's loop
    (self_play.py:31-108) with one actor playing E games at once.

    One pass of the loop = the reference's pass for one game, in the same order of storage calls:
        loop condition   get_info("training_step"), get_info("terminate")
        weight pull      get_info("weights") -> set_weights           (self_play.py:37: before every game)
        training         temperature = visit_softmax_temperature_fn(get_info("training_step")), play until at
                         least one env finishes a game, replay_buffer.save_game(history, shared_storage) per game
        test mode        temperature 0; per finished game set_info({episode_length, total_reward, mean_value})
                         (+ {muzero_reward, opponent_reward} for two players)
        throttle         self_play_delay sleep, then the training_step / num_played_steps < ratio wait
    so with E == 1 the call sequence is the reference's own (fixture G15, tests/test_self_play_loop.py).  With E > 1
    every env that starts a game after a pull plays it with the pulled weights; envs in mid-game switch to them at
    that move boundary, and every game records the weight version (the training step of the
    pull) it started and ended with in `weights_version`.

    `moves_per_pass`: play exactly that many moves per pass instead of "until a game ends" -- required when the
    actors of several GPUs run this loop together (torch.distributed initialised): every rank then makes the same
    collective calls.  Rank 0 alone talks to `shared_storage`; the loop condition travels as a 2-word broadcast and
    the weights as one broadcast of the flat buffer (weights.FlatWeights, RCCL over xGMI) when their version moved."""

    def _loop_state(self):
        st = self.__dict__.get("_loop")
        if st is None:
            st = self.__dict__["_loop"] = dict(version=0, started=numpy.zeros(self.E, dtype=numpy.int64), flat=None)
        return st

    def _distributed(self):
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _loop_control(self, shared_storage, training_steps):
        """(training_step, terminate) as every rank must see them."""
        if not self._distributed():
            step = shared_storage.get_info("training_step")
            return step, (step >= training_steps or shared_storage.get_info("terminate"))
        import torch.distributed as dist
        word = torch.zeros(2, dtype=torch.int64, device=self.device if dist.get_backend() == "nccl" else "cpu")
        if dist.get_rank() == 0:
            step = shared_storage.get_info("training_step")
            word[0], word[1] = int(step), int(step >= training_steps or bool(shared_storage.get_info("terminate")))
        dist.broadcast(word, src=0)
        return int(word[0]), bool(word[1])

    def _pull_weights(self, shared_storage, version):
        st = self._loop_state()
        if not self._distributed():
            self.set_weights(shared_storage.get_info("weights"))
        else:
            import torch.distributed as dist
            from .weights import FlatWeights
            if st["flat"] is None:
                st["flat"] = FlatWeights(self.model)
            if dist.get_rank() == 0:
                st["flat"].load_state_dict(shared_storage.get_info("weights"))
            st["flat"].broadcast(src=0)
        st["version"] = int(version)

    def _play_pass(self, temperature, temperature_threshold, moves_per_pass):
        """Play until a game ends (moves_per_pass None) or exactly moves_per_pass moves; returns the finished games
        as (env index, GameHistory) pairs in the order they ended."""
        finished = []
        moves = 0
        while True:
            self.step(temperature, temperature_threshold, on_game=lambda e, gh: finished.append((e, gh)))
            moves += 1
            if (moves_per_pass is None and finished) or (moves_per_pass is not None and moves >= moves_per_pass):
                return finished

    def continuous_self_play(self, shared_storage, replay_buffer, test_mode=False, moves_per_pass=None):
        cfg = self.config
        st = self._loop_state()
        if self._distributed() and moves_per_pass is None:
            raise ValueError("several ranks: give moves_per_pass so that every rank makes the same collective calls")
        # test mode against an opponent (self_play.py:65-90: play_game(0, threshold, False, config.opponent,
        # config.muzero_player) for games with several players): the actors whose step() takes an opponent play it
        self._opponent = ("self", 0)
        if test_mode and len(cfg.players) > 1:
            self._opponent = (cfg.opponent, cfg.muzero_player)
        talker = not self._distributed() or torch.distributed.get_rank() == 0
        while True:
            step, stop = self._loop_control(shared_storage, cfg.training_steps)
            if stop:
                break
            self._pull_weights(shared_storage, step)
            if not test_mode:
                temperature = cfg.visit_softmax_temperature_fn(
                    trained_steps=shared_storage.get_info("training_step") if talker else step)
            else:
                temperature = 0                          # best action, no exploration noise in the sampling
            finished = self._play_pass(temperature, cfg.temperature_threshold, moves_per_pass)
            for e, game_history in finished:
                game_history.weights_version = (int(st["started"][e]), st["version"])
                st["started"][e] = st["version"]
                if not test_mode:
                    replay_buffer.save_game(game_history, shared_storage)
                elif talker:
                    shared_storage.set_info({
                        "episode_length": len(game_history.action_history) - 1,
                        "total_reward": sum(game_history.reward_history),
                        "mean_value": numpy.mean([value for value in game_history.root_values if value]),
                    })
                    if 1 < len(cfg.players):
                        mine = cfg.muzero_player
                        shared_storage.set_info({
                            "muzero_reward": sum(reward for i, reward in enumerate(game_history.reward_history)
                                                 if game_history.to_play_history[i - 1] == mine),
                            "opponent_reward": sum(reward for i, reward in enumerate(game_history.reward_history)
                                                   if game_history.to_play_history[i - 1] != mine),
                        })
            # Managing the self-play / training ratio (self_play.py:92-106)
            if not test_mode and cfg.self_play_delay:
                time.sleep(cfg.self_play_delay)
            if not test_mode and cfg.ratio and talker:
                while (shared_storage.get_info("training_step")
                       / max(1, shared_storage.get_info("num_played_steps")) < cfg.ratio
                       and shared_storage.get_info("training_step") < cfg.training_steps
                       and not shared_storage.get_info("terminate")):
                    time.sleep(0.5)
        self.close()


class BatchedSelfPlay(ManyEnvLoop):
    """E games in lock step on one GPU: the MI355X-native actor.

    Env e plays with `Game(seed + e)` and the RNG stream of reference worker `seed + e`
    (muzero.py:170-178), so with E == 1 it reproduces `SelfPlay.play_game`, and with E > 1 each env
    reproduces what the reference's e-th Ray worker would have played with the same weights.
    Finished games are handed to `on_game(env_index, GameHistory)` and their env restarts at once.
    """

    def __init__(self, initial_checkpoint, Game, config, seed, num_envs, device=None, use_graph=True):
        self.config = config
        self.E = int(num_envs)
        self.games = [Game(seed + e) for e in range(self.E)]
        self.device = torch.device(device if device is not None else "cuda")
        torch.manual_seed(seed)
        self.model = models.MuZeroNetwork(config)
        self.model.set_weights(initial_checkpoint["weights"])
        self.model.to(self.device)
        self.model.eval()
        fused = config.network == "fullyconnected"
        self.engine = BatchedMCTS(config, self.E, device=self.device,
                                  seeds=[seed + e for e in range(self.E)], use_graph=use_graph,
                                  group_width=16 if fused and len(config.action_space) <= 16 else 0)
        if fused:
            try:
                self.engine.configure_fused_fc(self.model)      # whole move in one HIP launch
            except (NotImplementedError, RuntimeError):
                pass
        self.histories = [None] * self.E
        self.observations = [None] * self.E
        self.moves_played = 0
        self.games_finished = 0
        for e in range(self.E):
            self._restart(e)

    def _restart(self, e):
        gh = GameHistory()
        obs = self.games[e].reset()
        gh.action_history.append(0)
        gh.observation_history.append(obs)
        gh.reward_history.append(0)
        gh.to_play_history.append(self.games[e].to_play())
        self.histories[e] = gh
        self.observations[e] = obs

    def set_weights(self, weights):
        self.model.set_weights(weights)

    def step(self, temperature, temperature_threshold=None, on_game=None, opponent=None, muzero_player=None):
        """One move in every env (the body of play_game's loop, self_play.py:129-182).  `opponent` / `muzero_player`
        (default: what continuous_self_play's test mode set, else "self"): in an env where it is not MuZero's turn the
        move comes from select_opponent_action (self_play.py:189-221) -- the plugin's expert_agent(), or
        numpy.random.choice over the legal actions drawn on THAT env's stream -- and no search statistics are stored
        for it (root None: store_search_statistics appends only a None root value, self_play.py:497-512)."""
        cfg = self.config
        if opponent is None:
            opponent, muzero_player = getattr(self, "_opponent", ("self", 0))
        if opponent not in ("self", "expert", "random"):
            raise NotImplementedError('many-env actors play opponent "self", "expert" or "random" ("human": use SelfPlay)')
        searching = [opponent == "self" or muzero_player == self.games[e].to_play() for e in range(self.E)]
        stacked = numpy.stack([
            self.histories[e].get_stacked_observations(-1, cfg.stacked_observations)
            for e in range(self.E)]).astype(numpy.float32)
        legal = [self.games[e].legal_actions() if searching[e] else [] for e in range(self.E)]
        to_play = [self.games[e].to_play() for e in range(self.E)]
        self.engine.search(self.model, stacked, legal, to_play, True)     # (an empty legal set = env not searched)
        temps = numpy.array([
            temperature if not temperature_threshold
            or len(self.histories[e].action_history) < temperature_threshold else 0
            for e in range(self.E)], dtype=numpy.float64)
        actions, _ = self.engine.sample_actions(temps)
        child_visits, root_values = self.engine.search_statistics()
        for e in range(self.E):
            gh = self.histories[e]
            if searching[e]:
                action = int(actions[e])
                gh.child_visits.append([float(v) if a in legal[e] else 0 for a, v in enumerate(child_visits[e])])
                gh.root_values.append(float(root_values[e]))
            else:
                action = self._opponent_action(e, opponent)
                gh.root_values.append(None)
            observation, reward, done = self.games[e].step(action)
            gh.action_history.append(action)
            gh.observation_history.append(observation)
            gh.reward_history.append(reward)
            gh.to_play_history.append(self.games[e].to_play())
            self.observations[e] = observation
            if done or len(gh.action_history) > cfg.max_moves:
                self.games_finished += 1
                if on_game is not None:
                    on_game(e, gh)
                self._restart(e)
        self.moves_played += self.E

    def _opponent_action(self, e, opponent):
        """select_opponent_action (self_play.py:189-221) for env e.  The reference's opponents draw from numpy's GLOBAL
        generator, which in a reference worker is also the search's stream (expert agents start from a random legal move:
        games/tictactoe.py:307-312, games/connect4.py:306-343): env e's stream is lent to numpy for the call and handed
        back to the engine afterwards (host mirror and device copy move together)."""
        game = self.games[e]
        saved = numpy.random.get_state()
        numpy.random.set_state(self.engine.get_rng_state(e))
        try:
            if opponent == "expert":
                action = game.expert_agent()
            else:
                legal = game.legal_actions()
                assert legal, f"Legal actions should not be an empty array. Got {legal}."
                assert set(legal).issubset(set(self.config.action_space)), "Legal actions should be a subset of the action space."
                action = numpy.random.choice(legal)
            self.engine.set_rng_state(e, numpy.random.get_state())
        finally:
            numpy.random.set_state(saved)
        return int(action)

    def close(self):
        for g in self.games:
            g.close()
        self.engine.close()


class PackedGames:
    """A batch of finished games as arrays (what a device-side replay buffer would ingest directly).

    Game i has `length[i]` moves; row layouts follow GameHistory (reference self_play.py:116-121, 176-182):
      observations[i, 0..length]   observation_history (index 0 = reset observation)
      actions[i, 0..length]        action_history      (index 0 = the reference's dummy action 0)
      rewards[i, 0..length]        reward_history      (index 0 = 0)
      to_play[i, 0..length]        to_play_history
      child_visits[i, 0..length-1] visit-count targets, root_values[i, 0..length-1]
    Entries past a game's length are padding."""

    __slots__ = ("env_index", "length", "observations", "actions", "rewards", "to_play", "child_visits", "root_values")

    def __init__(self, **arrays):
        for k, v in arrays.items():
            setattr(self, k, v)

    def __len__(self):
        return len(self.env_index)

    def history(self, i):
        """Game i as a reference-shaped GameHistory (Python lists)."""
        n = int(self.length[i])
        gh = GameHistory()
        gh.observation_history = [o.copy() for o in self.observations[i, : n + 1]]   # (the batch may be a view)
        gh.action_history = self.actions[i, : n + 1].tolist()
        gh.reward_history = self.rewards[i, : n + 1].tolist()
        gh.to_play_history = self.to_play[i, : n + 1].tolist()
        gh.child_visits = self.child_visits[i, :n].tolist()
        gh.root_values = self.root_values[i, :n].tolist()
        return gh


class HistoryFiler:
    """Host-side filing of whole move batches into per-env history rows in native code (include/mzhist.h):
    what DeviceSelfPlay._file_move does one move at a time in numpy, for M moves at once on the library's
    worker pool.  Finished games come back as PackedGames whose arrays are views of the library's buffers,
    valid until the next `file` call."""

    def __init__(self, num_envs, max_moves, observation_shape, num_actions):
        import ctypes
        from . import _native
        self._ct, self._native = ctypes, _native
        self._lib = _native.load()
        self.E, self.L, self.A = int(num_envs), int(max_moves), int(num_actions)
        self.observation_shape = tuple(int(v) for v in observation_shape)
        self.obs_floats = int(numpy.prod(self.observation_shape))
        handle = ctypes.c_void_p()
        if self._lib.mzhist_create(self.E, self.L, self.obs_floats, self.A, ctypes.byref(handle)) != 0:
            raise RuntimeError("mzhist_create failed")
        self._h = handle

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mzhist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def begin(self, first_observations, first_to_play=None):
        obs = numpy.ascontiguousarray(first_observations, dtype=numpy.float32).reshape(self.E, self.obs_floats)
        tp = None if first_to_play is None else numpy.ascontiguousarray(first_to_play, dtype=numpy.int32)
        self._lib.mzhist_begin(self._h, self._native.ptr(obs, self._native.c_f32_p),
                               None if tp is None else self._native.ptr(tp, self._native.c_i32_p))

    def load_rows(self, obs, act, rew, tp, cv, rv, lengths):
        """Take over running games kept as numpy rows ([E, L+1, ...] / [E, L, ...] arrays, lengths [E])."""
        self._rows(obs, act, rew, tp, cv, rv, lengths, 1)

    def store_rows(self, obs, act, rew, tp, cv, rv, lengths):
        """Hand the running games back into numpy rows of the same shapes."""
        self._rows(obs, act, rew, tp, cv, rv, lengths, 0)

    def _rows(self, obs, act, rew, tp, cv, rv, lengths, load):
        n = self._native
        assert obs.shape[1] == self.L + 1 and obs.dtype == numpy.float32 and act.dtype == numpy.int32
        assert rew.dtype == numpy.float32 and cv.dtype == numpy.float64 and rv.dtype == numpy.float64
        tp32 = numpy.ascontiguousarray(tp, dtype=numpy.int32)
        len32 = numpy.ascontiguousarray(lengths, dtype=numpy.int32)
        rc = self._lib.mzhist_rows(self._h, obs.ctypes.data, act.ctypes.data, rew.ctypes.data, tp32.ctypes.data,
                                   cv.ctypes.data, rv.ctypes.data, len32.ctypes.data, load)
        if rc != 0:
            raise RuntimeError("mzhist_rows failed")
        if not load:
            tp[...] = tp32
            lengths[...] = len32

    def lengths(self):
        addr = self._lib.mzhist_lengths(self._h)
        return numpy.ctypeslib.as_array(self._ct.cast(addr, self._ct.POINTER(self._ct.c_int32)), shape=(self.E,))

    def file(self, out, legal, num_legal, num_simulations, rewards, done, obs_after, obs_next, to_play_after=None,
             to_play_next=None):
        """out: engine.moves_collect() result (copies or ring views); legal [E,A] / num_legal [E] for the whole batch or
        [M,E,A] / [M,E] per move (engine.moves_inputs); rewards f32 [M,E], done u8 [M,E],
        obs_after / obs_next f32 [M,E,...] host arrays; to_play_after / to_play_next [M,E] for two-player games.
        Returns PackedGames of the games that ended, or None."""
        ct = self._ct
        M = int(out["actions"].shape[0])
        def per_move_rows(a):
            # a per-move array whose rows are contiguous goes as it is (ring views: rows a block stride apart)
            a = numpy.asarray(a)
            inner = numpy.empty(a.shape[1:], dtype=numpy.int32).strides
            return a if a.dtype == numpy.int32 and a.ndim >= 2 and a.strides[1:] == inner else numpy.ascontiguousarray(a, dtype=numpy.int32)
        keep = [numpy.ascontiguousarray(out["moves_done"], dtype=numpy.int32),
                per_move_rows(legal) if numpy.ndim(legal) == 3 else numpy.ascontiguousarray(legal, dtype=numpy.int32),
                per_move_rows(num_legal) if numpy.ndim(legal) == 3 else numpy.ascontiguousarray(num_legal, dtype=numpy.int32),
                numpy.ascontiguousarray(rewards, dtype=numpy.float32), numpy.ascontiguousarray(done, dtype=numpy.uint8),
                numpy.ascontiguousarray(obs_after, dtype=numpy.float32), numpy.ascontiguousarray(obs_next, dtype=numpy.float32)]
        mv = self._native.MzHistMoves()
        mv.n_moves, mv.num_simulations = M, int(num_simulations)
        mv.moves_done = keep[0].ctypes.data
        for name, key in (("actions", "actions"), ("visits", "visits"), ("root_value_sum", "root_value_sum")):
            a = out[key]
            setattr(mv, name, a.ctypes.data)
            setattr(mv, name + "_stride", int(a.strides[0]) if M else 0)
        mv.legal, mv.num_legal = keep[1].ctypes.data, keep[2].ctypes.data
        per_move = keep[1].ndim == 3                      # legal [M, E, A], num_legal [M, E]: one set per move
        mv.legal_stride = int(keep[1].strides[0]) if per_move else 0
        mv.num_legal_stride = int(keep[2].strides[0]) if per_move else 0
        mv.rewards, mv.done, mv.obs_after, mv.obs_next = (k.ctypes.data for k in keep[3:7])
        mv.to_play_after = mv.to_play_next = None
        if to_play_after is not None:
            keep += [numpy.ascontiguousarray(to_play_after, dtype=numpy.int32), numpy.ascontiguousarray(to_play_next, dtype=numpy.int32)]
            mv.to_play_after, mv.to_play_next = keep[7].ctypes.data, keep[8].ctypes.data
        n = ct.c_int32()
        if self._lib.mzhist_file(self._h, ct.byref(mv), ct.byref(n)) != 0:
            raise RuntimeError(self._lib.mzhist_last_error(self._h).decode())
        if n.value == 0:
            return None
        ptrs = [ct.c_void_p() for _ in range(8)]
        row = ct.c_int32()
        count = self._lib.mzhist_finished(self._h, *[ct.byref(p) for p in ptrs], ct.byref(row))
        W = row.value

        def arr(p, ctype, shape):
            return numpy.ctypeslib.as_array(ct.cast(p, ct.POINTER(ctype)), shape=shape)
        return PackedGames(env_index=arr(ptrs[0], ct.c_int32, (count,)), length=arr(ptrs[1], ct.c_int32, (count,)),
                           observations=arr(ptrs[2], ct.c_float, (count, W + 1) + self.observation_shape),
                           actions=arr(ptrs[3], ct.c_int32, (count, W + 1)), rewards=arr(ptrs[4], ct.c_float, (count, W + 1)),
                           to_play=arr(ptrs[5], ct.c_int32, (count, W + 1)),
                           child_visits=arr(ptrs[6], ct.c_double, (count, W, self.A)),
                           root_values=arr(ptrs[7], ct.c_double, (count, W)))


class DeviceSelfPlay(ManyEnvLoop):
    """Self-play with device-resident environments (games.device.DeviceEnvs): search, env step and
    observation all stay on the GPU; per move the host only draws the exploration noise, samples the
    actions (both on the per-env numpy-compatible RNG streams) and files the move into packed per-env
    history rows.

    Env e plays the game the host plugin `Game(seed + e)` would play with reference worker `seed + e`'s RNG
    stream -- the same games `BatchedSelfPlay` produces with host envs (tests/test_gpu_envs.py) -- without
    E Python `game.step` calls per move.  Finished games leave as one `PackedGames` batch per move through
    `on_games(batch)`; `on_game(env_index, GameHistory)` is the per-game compatibility callback (it rebuilds
    Python lists, which costs more than the search itself at thousands of envs)."""

    def __init__(self, initial_checkpoint, game_name, config, seed, num_envs, device=None, use_graph=True):
        from .games.device import DeviceEnvs
        assert config.stacked_observations == 0, "device envs do not stack past observations yet"
        self.config = config
        self.E = E = int(num_envs)
        self.device = torch.device(device if device is not None else "cuda")
        torch.manual_seed(seed)
        self.model = models.MuZeroNetwork(config)
        self.model.set_weights(initial_checkpoint["weights"])
        self.model.to(self.device)
        self.model.eval()
        seeds = [seed + e for e in range(E)]
        self.envs = DeviceEnvs(game_name, E, seeds=seeds, device=self.device)
        assert self.envs.A == len(config.action_space) and self.envs.observation_shape == tuple(config.observation_shape)
        fused = config.network == "fullyconnected"
        self.engine = BatchedMCTS(config, E, device=self.device, seeds=seeds, use_graph=use_graph,
                                  group_width=16 if fused and len(config.action_space) <= 16 else 0)
        if fused:
            try:
                self.engine.configure_fused_fc(self.model)
            except (NotImplementedError, RuntimeError):
                pass
        self.moves_played = 0
        self.games_finished = 0
        # packed history rows, one per env; every game starts at column 0 of its row
        T, A = int(config.max_moves) + 1, self.envs.A
        self._obs = numpy.zeros((E, T + 1) + self.envs.observation_shape, dtype=numpy.float32)
        self._act = numpy.zeros((E, T + 1), dtype=numpy.int32)
        self._rew = numpy.zeros((E, T + 1), dtype=numpy.float32)
        self._tp = numpy.zeros((E, T + 1), dtype=numpy.int8)
        self._cv = numpy.zeros((E, T, A), dtype=numpy.float64)
        self._rv = numpy.zeros((E, T), dtype=numpy.float64)
        self._len = numpy.zeros(E, dtype=numpy.int64)      # moves played in env e's current game
        self._rows = numpy.arange(E)
        self._cur = self._observe_host()
        self._obs[:, 0] = self._cur["obs"]
        self._tp[:, 0] = self._cur["to_play"]

    def _observe_host(self):
        obs, legal, num_legal, to_play = self.envs.observe()
        return dict(obs=obs.cpu().numpy(), legal=legal.cpu().numpy(), num_legal=num_legal.cpu().numpy(),
                    to_play=to_play.cpu().numpy(), obs_dev=obs)

    def _current(self):
        """The envs' current positions on the host (what step() searches).  A device-input move batch leaves them on the
        device only: they are fetched when a step() next asks, not once per batch."""
        if self._cur.get("on_device_only"):
            self._cur = self._observe_host()
        return self._cur

    def set_weights(self, weights):
        self.model.set_weights(weights)

    def step(self, temperature, temperature_threshold=None, on_game=None, on_games=None):
        """One move in every env (the body of play_game's loop, self_play.py:129-182)."""
        self.step_begin(on_game, on_games)
        self.step_end(temperature, temperature_threshold, on_game, on_games)

    def _no_opponent(self):
        if getattr(self, "_opponent", ("self", 0))[0] != "self":
            raise NotImplementedError("device-resident envs play \"self\" in test mode; BatchedSelfPlay (host Game "
                                      "plugins) plays expert / random opponents")

    def step_begin(self, on_game=None, on_games=None):
        """First half of step(): queue the search of every env's current position on the engine's stream and return
        (the GPU works; `step_end` waits).  Two actors on streams of their own alternate their halves
        (PipelinedDeviceSelfPlay): one's host work runs under the other's search."""
        self._no_opponent()
        self.flush(on_game, on_games)      # first: the unfiled batch holds views of the download ring
        self._drop_batch()
        cur = self._current()
        if self.engine._fc_model is not None:
            self.engine.search_fused_begin(cur["obs_dev"].reshape(self.E, -1), cur["legal"], cur["to_play"], True,
                                           num_legal=cur["num_legal"])
        else:
            self.engine.search_begin(self.model, cur["obs_dev"], cur["legal"], cur["to_play"], True,
                                     num_legal=cur["num_legal"])

    def step_end(self, temperature, temperature_threshold=None, on_game=None, on_games=None):
        """Second half of step(): wait for the search, sample the actions, step the envs, file the move."""
        cur = self._cur
        self.engine.readout()
        if temperature_threshold:
            # play_game: temperature only while len(action_history) < threshold (self_play.py:163-170)
            temps = numpy.where(self._len + 1 < temperature_threshold, float(temperature), 0.0)
        else:
            temps = numpy.full(self.E, float(temperature))
        actions, _ = self.engine.sample_actions(numpy.ascontiguousarray(temps, dtype=numpy.float64))
        stats = self.engine.stats
        reward, done = self.envs.step(actions)
        after = self._observe_host()
        reward, done = reward.cpu().numpy(), done.cpu().numpy().astype(bool)
        over = done | (self._len + 2 > self.config.max_moves)   # len(action_history) > max_moves ends the game
        nxt = after
        if over.any():
            self.envs.reset(torch.from_numpy(over.astype(numpy.uint8)).to(self.device))
            nxt = self._observe_host()
        # file the move natively (include/mzhist.h, the library's worker pool): a batch of one move whose legal sets are
        # this move's; child visits by action and root values as store_search_statistics computes them (self_play.py:497-512)
        one = {"actions": actions.astype(numpy.int32)[None], "visits": stats["visits"][None],
               "root_value_sum": stats["root_value_sum"][None], "moves_done": numpy.ones(self.E, dtype=numpy.int32)}
        batch = self._history_filer().file(one, cur["legal"], cur["num_legal"], self.config.num_simulations, reward[None],
                                           over.astype(numpy.uint8)[None], after["obs"][None], nxt["obs"][None],
                                           to_play_after=after["to_play"][None], to_play_next=nxt["to_play"][None])
        self._len[:] = self._filer.lengths()
        if batch is not None:
            self.games_finished += len(batch)
            if on_games is not None:
                on_games(batch)
            if on_game is not None:
                for i, e in enumerate(batch.env_index):
                    on_game(int(e), batch.history(i))
        self._cur = nxt
        self.moves_played += self.E

    def _file_move(self, played, actions, child_visits, root_values, reward, over, obs_after, to_play_after,
                   obs_next, to_play_next, on_game, on_games):
        """File one move of the envs in `played` into their history rows; hand finished games (`over`) out
        and start their next game from obs_next (the reset observation)."""
        rows = numpy.flatnonzero(played)
        at = self._len[rows]
        self._cv[rows, at] = child_visits[rows]
        self._rv[rows, at] = root_values[rows]
        self._act[rows, at + 1] = actions[rows]
        self._rew[rows, at + 1] = reward[rows]
        self._obs[rows, at + 1] = obs_after[rows]
        self._tp[rows, at + 1] = to_play_after[rows]
        self._len[rows] = at + 1
        idx = numpy.flatnonzero(over & played)
        if len(idx):
            n = self._len[idx]
            L = int(n.max())
            batch = PackedGames(env_index=idx, length=n, observations=self._obs[idx, : L + 1],
                                actions=self._act[idx, : L + 1], rewards=self._rew[idx, : L + 1],
                                to_play=self._tp[idx, : L + 1], child_visits=self._cv[idx, :L], root_values=self._rv[idx, :L])
            self.games_finished += len(idx)
            if on_games is not None:
                on_games(batch)
            if on_game is not None:
                for i, e in enumerate(idx):
                    on_game(int(e), batch.history(i))
            self._len[idx] = 0
            self._obs[idx, 0] = obs_next[idx]
            self._tp[idx, 0] = to_play_next[idx]

    # ---- whole batches of moves on the device (engine.moves_*, include/mzmcts.h) ---------------------------
    def play_moves(self, n_moves, temperature, on_game=None, on_games=None, temperature_threshold=None):
        """`n_moves` moves of every env with no host round trip in between: search (which samples the action),
        env step, terminal observation, reset of finished envs, next observation -- all queued on one stream.
        Fully-connected networks search in the fused whole-move kernel (games with a constant legal set: the exploration
        noise of the whole batch is drawn up front, and the next batch's while this one runs); residual networks search
        lock-step (engine.moves_enqueue_lockstep) with inputs, noise and action sampling on the device.
        `temperature_threshold` (default config.temperature_threshold) is play_game's rule, applied per env and move.
        An env may come back with fewer than n_moves moves played (it plays the rest next time)."""
        E, eng, envs, cfg = self.E, self.engine, self.envs, self.config
        if cfg.max_moves < envs.max_episode_steps:
            raise NotImplementedError("play_moves ends games where the environment does; max_moves is shorter")
        # Everything but a fused search of a game with a constant legal set takes the device-input form of the batch:
        # board games (legal sets change), residual networks (lock-step searches), and a temperature threshold
        # (play_game drops to temperature 0 once len(action_history) reaches it, self_play.py:152-158: a per-env, per-move
        # switch the kernels make from per-game move counters they keep on the device)
        if temperature_threshold is None:
            temperature_threshold = cfg.temperature_threshold
        if not getattr(envs, "constant_legal_actions", False) or eng._fc_model is None or temperature_threshold:
            return self._play_moves_device_inputs(n_moves, temperature, on_game, on_games, temperature_threshold)
        cur = self._current()
        params = (int(n_moves), float(temperature))
        if getattr(self, "_batch_ready", None) != params:
            self.flush(on_game, on_games)                    # (the unfiled batch holds views of the download ring)
            self._drop_batch()
            eng.moves_prepare(n_moves, cur["legal"], cur["to_play"], temperature, True, num_legal=cur["num_legal"])
        ring = self._move_ring(n_moves)
        obs_in = cur["obs_dev"]
        for m in range(n_moves):
            eng.moves_enqueue(obs_in.reshape(E, -1))
            # step, terminal observation, reset of the finished envs, next observation: one call, one launch
            obs_in = envs.advance(eng.moves_actions(m), ring["reward"][m], ring["done"][m], ring["obs_after"][m],
                                  ring["obs_next"][m])
        self.flush(on_game, on_games)                        # the previous batch's games, while this one runs
        eng.moves_predraw_next(n_moves, cur["legal"], cur["to_play"], temperature, True, num_legal=cur["num_legal"])
        out = eng.moves_collect(copy=False)                  # views: filed (flush) before the next collect overwrites them
        # env outputs of the batch: one asynchronous copy each into pinned host buffers (alternating sets, so that
        # the batch waiting to be filed keeps its own), one wait
        pinned = ring["pinned"][ring["flip"]]
        ring["flip"] ^= 1
        for k in ("reward", "done", "obs_after", "obs_next"):
            pinned[k][:n_moves].copy_(ring[k][:n_moves], non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        host = {k: pinned[k][:n_moves].numpy() for k in ("reward", "done", "obs_after", "obs_next")}
        eng.moves_submit_next()
        self._batch_ready = params
        self._unfiled = (out, host, cur["legal"], cur["num_legal"], n_moves, None, None)
        self._cur = dict(cur, obs_dev=obs_in, obs=host["obs_next"][n_moves - 1])
        self.moves_played += int(out["moves_done"].sum())
        return out["moves_done"].copy()

    def _play_moves_device_inputs(self, n_moves, temperature, on_game, on_games, temperature_threshold=None):
        """play_moves with the batch's inputs on the device: the searches read the legal sets and players to move from the
        environment kernels' device outputs and draw their exploration noise on the device (engine.moves_prepare_device),
        so a whole batch -- games ending and restarting inside it -- is queued without the host; afterwards every move
        is filed with the legal set it was searched with.  The search of a move is the fused whole-move kernel
        (fully-connected networks) or the lock-step loop with this actor's network (residual networks)."""
        self._device_batch_begin(n_moves, temperature, on_game, on_games, temperature_threshold)
        for m in range(n_moves):
            self._device_batch_move(m)
        return self._device_batch_end(on_game, on_games)

    # (the three phases are separate so that PipelinedDeviceSelfPlay can interleave the batches of its groups move by move)
    def _device_batch_begin(self, n_moves, temperature, on_game, on_games, temperature_threshold):
        eng, envs = self.engine, self.envs
        if getattr(self, "_batch_ready", None) or temperature_threshold:
            # a batch of the pre-drawn form is waiting to be filed / the threshold rule needs the games' current lengths
            self.flush(on_game, on_games)
        self._drop_batch()
        eng.moves_prepare_device(n_moves, envs.legal, envs.num_legal, envs.to_play, temperature, True)
        if temperature_threshold:
            eng.moves_temperature_threshold(temperature_threshold, self._len)
        ring = self._move_ring(n_moves)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        # the env outputs of a move (reward, done, the observations the history rows need) go to pinned host memory on a
        # copy stream as soon as the move's env kernel has run: the downloads ride under the batch's remaining searches
        pinned = ring["pinned"][ring["flip"]]
        ring["flip"] ^= 1
        self._dev_batch = dict(n_moves=n_moves, ring=ring, obs_in=self._cur["obs_dev"], threshold=temperature_threshold,
                               pinned=pinned)

    def _device_batch_move(self, m):
        b, eng, envs = self._dev_batch, self.engine, self.envs
        ring = b["ring"]
        if b["threshold"] and m > 0:
            eng.moves_finished(ring["done"][m - 1])          # games that ended with the move before restart their count
        if eng._fc_model is not None:
            eng.moves_enqueue(b["obs_in"].reshape(self.E, -1).contiguous())
        else:
            eng.moves_enqueue_lockstep(self.model, b["obs_in"])
        b["obs_in"] = envs.advance(eng.moves_actions(m), ring["reward"][m], ring["done"][m], ring["obs_after"][m],
                                   ring["obs_next"][m])
        ran = torch.cuda.Event()
        ran.record(torch.cuda.current_stream(self.device))
        self._copy_stream.wait_event(ran)
        with torch.cuda.stream(self._copy_stream):
            for k in ("reward", "done", "obs_after", "obs_next"):
                b["pinned"][k][m].copy_(ring[k][m], non_blocking=True)

    def _device_batch_end(self, on_game, on_games):
        b, eng, envs = self._dev_batch, self.engine, self.envs
        n_moves, ring = b["n_moves"], b["ring"]
        self._dev_batch = None
        self.flush(on_game, on_games)                        # the previous batch's games, while this one runs
        # (views of the engine's pinned rings, filled move by move while the batch ran: nothing is unpacked here; they
        # stay valid until the batch after the next one is prepared, and flush() files them before that)
        out = eng.moves_collect(copy=False)
        inputs = eng.moves_inputs(n_moves, copy=False)
        pinned = b["pinned"]
        last_to_play = envs.to_play.cpu().numpy()
        self._copy_stream.synchronize()                  # (the last move's downloads)
        host = {k: pinned[k][:n_moves].numpy() for k in ("reward", "done", "obs_after", "obs_next")}
        two_players = len(self.config.players) > 1
        to_play = inputs["to_play"]
        to_play_after = (1 - to_play) if two_players else numpy.zeros_like(to_play)
        to_play_next = numpy.concatenate([to_play[1:], last_to_play[None]], axis=0)
        self._unfiled = (out, host, inputs["legal"], inputs["num_legal"], n_moves, to_play_after, to_play_next)
        # the envs' current positions stay on the device: the next batch starts from the last move's observation (the
        # kernels' output, where it lies); a step() fetches what it needs first (_current)
        self._cur = dict(obs_dev=b["obs_in"], on_device_only=True)
        self.moves_played += int(out["moves_done"].sum())
        return out["moves_done"].copy()

    def _batchable(self, temperature, temperature_threshold, moves_per_pass, device_inputs=None):
        """Can a pass of `moves_per_pass` moves run as one move batch on the device (play_moves)?"""
        if device_inputs is None:
            device_inputs = (not getattr(self.envs, "constant_legal_actions", False) or self.engine._fc_model is None
                             or bool(temperature_threshold))
        return (moves_per_pass is not None and self.config.max_moves >= self.envs.max_episode_steps
                and getattr(self, "_opponent", ("self", 0))[0] == "self"
                and (temperature == 0 or _native.exact_inverse_temperature(temperature))
                # (a device-input batch draws its exploration noise on the GPU: the legacy gamma sampler for shapes <= 1)
                and (not device_inputs or 0.0 < float(self.config.root_dirichlet_alpha) <= 1.0))

    def _play_pass(self, temperature, temperature_threshold, moves_per_pass):
        """ManyEnvLoop's pass: whole move batches on the device when the game, the network and the temperature allow
        it (play_moves), else one move at a time (step)."""
        if not self._batchable(temperature, temperature_threshold, moves_per_pass):
            return ManyEnvLoop._play_pass(self, temperature, temperature_threshold, moves_per_pass)
        finished = []
        self.play_moves(moves_per_pass, temperature, on_game=lambda e, gh: finished.append((e, gh)),
                        temperature_threshold=temperature_threshold or 0)
        self.flush(on_game=lambda e, gh: finished.append((e, gh)))
        return finished

    def flush(self, on_game=None, on_games=None):
        """File the moves of the last play_moves batch into the histories (play_moves does this for the
        batch before while the GPU runs the current one; call it once at the end).  Native code
        (HistoryFiler, include/mzhist.h): one pass over the batch on the library's worker pool."""
        if getattr(self, "_unfiled", None) is None:
            return
        out, host, legal, num_legal, n_moves, to_play_after, to_play_next = self._unfiled
        self._unfiled = None
        filer = self._history_filer()
        batch = filer.file(out, legal, num_legal, self.config.num_simulations, host["reward"], host["done"],
                           host["obs_after"], host["obs_next"], to_play_after=to_play_after, to_play_next=to_play_next)
        self._len[:] = filer.lengths()
        if batch is not None:
            self.games_finished += len(batch)
            if on_games is not None:
                on_games(batch)
            if on_game is not None:
                for i, e in enumerate(batch.env_index):
                    on_game(int(e), batch.history(i))

    def _history_filer(self):
        """The native filer takes over the rows of the running games (and hands them back to step())."""
        filer = getattr(self, "_filer", None)
        if filer is None:
            filer = self._filer = HistoryFiler(self.E, int(self.config.max_moves) + 1, self.envs.observation_shape, self.envs.A)
        if not getattr(self, "_filer_owns_rows", False):
            filer.load_rows(self._obs, self._act, self._rew, self._tp, self._cv, self._rv, self._len)
            self._filer_owns_rows = True
        return filer

    def _drop_batch(self):
        """Forget the batch that was drawn and uploaded ahead (its noise goes back into the RNG streams)."""
        if getattr(self, "_batch_ready", None):
            self.engine.moves_collect()
            self._batch_ready = None

    def _move_ring(self, n_moves):
        ring = getattr(self, "_ring", None)
        if ring is None or ring["reward"].shape[0] < n_moves:
            shape, dev = self.envs.observation_shape, self.device
            ring = dict(reward=torch.zeros((n_moves, self.E), dtype=torch.float32, device=dev),
                        done=torch.zeros((n_moves, self.E), dtype=torch.uint8, device=dev),
                        obs_after=torch.zeros((n_moves, self.E) + shape, dtype=torch.float32, device=dev),
                        obs_next=torch.zeros((n_moves, self.E) + shape, dtype=torch.float32, device=dev))
            ring["pinned"] = [{k: torch.zeros(ring[k].shape, dtype=ring[k].dtype).pin_memory()
                               for k in ("reward", "done", "obs_after", "obs_next")} for _ in range(2)]
            ring["flip"] = 0
            self._ring = ring
        return ring

    def close(self):
        self.envs.close()
        self.engine.close()


class PipelinedDeviceSelfPlay(ManyEnvLoop):
    """`groups` DeviceSelfPlay actors of num_envs / groups envs each, on HIP streams of their own, alternating the halves
    of a move (step_begin / step_end): while the host samples, steps, observes and files one group's move, the GPU
    searches the other group's positions.  Env e keeps seed `seed + e`, so the groups together play the games one
    DeviceSelfPlay of num_envs envs plays (tests/test_gpu_envs.py).  Callbacks see global env indices.  Carries the
    reference's continuous_self_play loop (ManyEnvLoop): a weight pull reaches every group's network replica."""

    def __init__(self, initial_checkpoint, game_name, config, seed, num_envs, groups=2, device=None, use_graph=True):
        assert num_envs % groups == 0
        self.config = config
        self.groups, self.per_group, self.E = groups, num_envs // groups, num_envs
        self.actors, self.streams = [], []
        for g in range(groups):
            actor = DeviceSelfPlay(initial_checkpoint, game_name, config, seed + g * self.per_group, self.per_group,
                                   device=device, use_graph=use_graph)
            stream = torch.cuda.Stream(device=actor.device)
            actor.engine.stream = stream
            self.actors.append(actor)
            self.streams.append(stream)
        self._started = [False] * groups
        self.device, self.model = self.actors[0].device, self.actors[0].model   # (what ManyEnvLoop's weight pull addresses)

    def _pull_weights(self, shared_storage, version):
        self._no_batch_queued("weight pull")
        ManyEnvLoop._pull_weights(self, shared_storage, version)     # group 0's model (all ranks: one flat broadcast)
        # the replicas take the parameters AND rebuild what they cache from them (folded batch norms, packed tower
        # weights): a replayed hipGraph reads those buffers without coming back to Python
        state = self.model.state_dict()
        current = torch.cuda.current_stream(self.device)
        for actor, stream in zip(self.actors[1:], self.streams[1:]):
            stream.wait_stream(current)
            with torch.cuda.stream(stream):
                actor.model.set_weights(state)

    @property
    def moves_played(self):
        return sum(a.moves_played for a in self.actors)

    @property
    def games_finished(self):
        return sum(a.games_finished for a in self.actors)

    def set_weights(self, weights):
        self._no_batch_queued("set_weights")
        for a in self.actors:
            a.set_weights(weights)

    def flush(self, on_game=None, on_games=None):
        for g, actor in enumerate(self.actors):
            one, many = self._callbacks(g, on_game, on_games)
            actor.flush(one, many)

    def _callbacks(self, g, on_game, on_games):
        base = g * self.per_group
        one = None if on_game is None else (lambda e, gh: on_game(base + e, gh))

        def many(batch):
            batch.env_index = batch.env_index + base
            on_games(batch)
        return one, (None if on_games is None else many)

    def step(self, temperature, temperature_threshold=None, on_game=None, on_games=None, prefetch=True):
        """One move in every env of every group (each group's search was queued during the previous call).
        prefetch=False leaves no search queued behind (the next call then starts them): what a caller wants before it
        changes the weights, so that no move is searched with the weights of the move before."""
        self._no_batch_queued("step")
        for g, actor in enumerate(self.actors):
            actor._opponent = getattr(self, "_opponent", ("self", 0))
            if not self._started[g]:
                with torch.cuda.stream(self.streams[g]):
                    actor.step_begin(*self._callbacks(g, on_game, on_games))
                self._started[g] = True
        for g, actor in enumerate(self.actors):
            one, many = self._callbacks(g, on_game, on_games)
            with torch.cuda.stream(self.streams[g]):
                actor.step_end(temperature, temperature_threshold, one, many)
                self._started[g] = False
                if prefetch:
                    actor.step_begin(one, many)      # the next move's search runs while the other groups are served
                    self._started[g] = True

    def play_moves(self, n_moves, temperature, on_game=None, on_games=None, temperature_threshold=None, prefetch=False):
        """`n_moves` moves of every env of every group with no host round trip (DeviceSelfPlay.play_moves in its
        device-input form): each group's batch is queued on the group's own stream, move by move in turn, so the
        kernels of the groups fill each other's gaps (a tower workgroup's fill / epilogue / export phases leave the matrix
        pipe idle).  Returns moves played per env.

        prefetch=True queues each group's NEXT batch (same parameters) as soon as its current one is collected, before
        anything is filed: collecting, filing and the callbacks of one group then run under the other group's kernels
        and the GPU never drains between calls -- the batch form of step()'s prefetch, with the same contract: the batch a
        call returns was queued by the call before (with that call's parameters and the weights of that time), and
        the last call before the weights change passes prefetch=False."""
        cfg = self.config
        if temperature_threshold is None:
            temperature_threshold = cfg.temperature_threshold
        queued = self.__dict__.setdefault("_batch_queued", [False] * self.groups)
        fresh = [g for g in range(self.groups) if not queued[g]]
        for g in fresh:
            actor = self.actors[g]
            if self._started[g]:                             # (a search queued by step(): finish that move first)
                raise RuntimeError("play_moves: a step() is half done; call step(..., prefetch=False) before batches")
            if cfg.max_moves < actor.envs.max_episode_steps:
                raise NotImplementedError("play_moves ends games where the environment does; max_moves is shorter")
            one, many = self._callbacks(g, on_game, on_games)
            with torch.cuda.stream(self.streams[g]):
                actor._device_batch_begin(n_moves, temperature, one, many, temperature_threshold)
        for m in range(n_moves):
            for g in fresh:
                with torch.cuda.stream(self.streams[g]):
                    self.actors[g]._device_batch_move(m)
        for g in fresh:
            queued[g] = True
        played = []
        for g, actor in enumerate(self.actors):
            one, many = self._callbacks(g, on_game, on_games)
            with torch.cuda.stream(self.streams[g]):
                played.append(actor._device_batch_end(one, many))
                queued[g] = False
                if prefetch:
                    actor._device_batch_begin(n_moves, temperature, one, many, temperature_threshold)
                    for m in range(n_moves):
                        actor._device_batch_move(m)
                    queued[g] = True
                    actor.flush(one, many)                   # the collected batch's games, while the GPU runs the next
        return numpy.concatenate(played)

    def _no_batch_queued(self, what):
        if any(self.__dict__.get("_batch_queued", ())):
            raise RuntimeError(f"{what}: a move batch is queued ahead; call play_moves(..., prefetch=False) first")

    def _play_pass(self, temperature, temperature_threshold, moves_per_pass):
        """ManyEnvLoop's pass: the groups' move batches on their streams when the pass can run as batches (always in their
        device-input form; nothing stays queued behind the pass: a weight pull follows), else move by move with the
        groups' halves alternating -- the last move of a pass queues nothing behind it either."""
        finished = []
        for actor in self.actors:
            actor._opponent = getattr(self, "_opponent", ("self", 0))
        if self.actors[0]._batchable(temperature, temperature_threshold, moves_per_pass, device_inputs=True):
            collect = lambda e, gh: finished.append((e, gh))
            self.play_moves(moves_per_pass, temperature, on_game=collect, temperature_threshold=temperature_threshold or 0)
            self.flush(on_game=collect)
            return finished
        moves = 0
        while True:
            last = moves_per_pass is None or moves + 1 >= moves_per_pass
            self.step(temperature, temperature_threshold, on_game=lambda e, gh: finished.append((e, gh)), prefetch=not last)
            moves += 1
            if (moves_per_pass is None and finished) or (moves_per_pass is not None and moves >= moves_per_pass):
                return finished

    def close(self):
        torch.cuda.synchronize(self.device)      # (a queued search may still be running)
        for a in self.actors:
            a.close()

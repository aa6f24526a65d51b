#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Build-container only: it imports the reference's own ``models.py`` / ``self_play.py`` /
``games/{tictactoe,connect4}.py`` from /root/reference (which never travels to the GPU box)
and records inputs + expected outputs as small ``.npz`` files.  Nothing but data is written:
no reference source text is stored.

    python tests/golden/make_golden.py [--only g4_cartpole ...]

Recipe: ``ray`` is not installed, so an in-process stand-in whose
``remote`` is the identity decorator is registered before ``import self_play``;
``gym``/``cv2`` are absent, so the gym-backed game files are imported only for their
``MuZeroConfig`` through empty stand-in modules (their ``Game`` classes are never built).
The CartPole checkpoint is read with ``torch.load(weights_only=True)`` plus an allow-list of
the four harmless numpy globals its pickle references.
"""
import argparse
import os
import sys
import time
import types

import numpy
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
from synth import synthetic_state_dict  # noqa: E402


def import_reference():
    ray = types.ModuleType("ray")
    ray.remote = lambda cls: cls
    ray.get = lambda x: x
    sys.modules.setdefault("ray", ray)
    for missing in ("gym", "cv2"):
        try:
            __import__(missing)
        except ImportError:
            sys.modules[missing] = types.ModuleType(missing)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import models  # noqa
    import self_play  # noqa
    return models, self_play


def load_cartpole_checkpoint():
    import _codecs
    allow = [
        (numpy._core.multiarray.scalar, "numpy.core.multiarray.scalar"),
        numpy.dtype,
        _codecs.encode,
        type(numpy.dtype("float64")),
    ]
    with torch.serialization.safe_globals(allow):
        ck = torch.load(os.path.join(REF, "results/cartpole/model.checkpoint"),
                        weights_only=True, map_location="cpu")
    return ck["weights"]


def rng_words_between(before, after):
    """32-bit words drawn from the global legacy MT19937 between two get_state() snapshots
    (valid while fewer than 624 words were drawn)."""
    same_key = numpy.array_equal(before[1], after[1])
    if same_key:
        return int(after[2] - before[2])
    return int((624 - before[2]) + after[2])


class TracingModel:
    """Proxy handed to the reference MCTS in place of the model: logs every inference."""

    def __init__(self, model):
        self._m = model
        self.log = []
        self.root = None

    def parameters(self):
        return self._m.parameters()

    def initial_inference(self, observation):
        out = self._m.initial_inference(observation)
        self.root = tuple(t.detach().clone() for t in out)
        return out

    def recurrent_inference(self, state, action):
        out = self._m.recurrent_inference(state, action)
        self.log.append((state.detach().clone(), int(action.item()),
                         tuple(t.detach().clone() for t in out)))
        return out


def make_tracing_mcts(self_play):
    class TracingMCTS(self_play.MCTS):
        def __init__(self, config):
            super().__init__(config)
            self.sims = []          # per simulation: list of (tie_count, action)
            self._cur = None

        def select_child(self, node, min_max_stats):
            scores = [self.ucb_score(node, c, min_max_stats) for c in node.children.values()]
            ties = sum(1 for s in scores if s == max(scores))
            action, child = super().select_child(node, min_max_stats)
            if self._cur is None:
                self._cur = []
            self._cur.append((ties, int(action), float(max(scores))))
            return action, child

        def backpropagate(self, search_path, value, to_play, min_max_stats):
            self.sims.append(self._cur or [])
            self._cur = None
            self.mms = min_max_stats
            return super().backpropagate(search_path, value, to_play, min_max_stats)

    return TracingMCTS


def trace_one(models, self_play, config, model, observation, legal_actions, to_play, seed,
              temperature=1.0):
    """Run the reference MCTS once from a freshly seeded global RNG and record everything."""
    A = len(config.action_space)
    S = config.num_simulations
    F = 2 * config.support_size + 1
    TracingMCTS = make_tracing_mcts(self_play)
    proxy = TracingModel(model)
    noises = []
    orig_dirichlet = numpy.random.dirichlet

    def logging_dirichlet(alpha, size=None):
        n = orig_dirichlet(alpha, size)
        noises.append(numpy.array(n, dtype="float64"))
        return n

    numpy.random.seed(seed)
    torch.manual_seed(seed)
    st0 = numpy.random.get_state()
    numpy.random.dirichlet = logging_dirichlet
    try:
        mcts = TracingMCTS(config)
        with torch.no_grad():
            root, info = mcts.run(proxy, observation, list(legal_actions), to_play, True)
    finally:
        numpy.random.dirichlet = orig_dirichlet
    st1 = numpy.random.get_state()
    action = self_play.SelfPlay.select_action(root, temperature)
    st2 = numpy.random.get_state()

    D = S + 1
    rec = {}
    rec["seed"] = seed
    rec["obs"] = numpy.asarray(observation, dtype="float32")
    leg = numpy.full(A, -1, dtype="int32")
    leg[: len(legal_actions)] = legal_actions
    rec["legal"] = leg
    rec["n_legal"] = len(legal_actions)
    rec["to_play"] = to_play
    rv, rr, rp, rh = proxy.root
    rec["root_value_logits"] = rv[0].numpy().astype("float32")
    rec["root_reward_logits"] = rr[0].numpy().astype("float32")
    rec["root_policy_logits"] = rp[0].numpy().astype("float32")
    rec["root_hidden"] = rh[0].numpy().astype("float32").reshape(-1)
    rec["root_predicted_value"] = float(info["root_predicted_value"])
    # pre-noise root priors exactly as Node.expand computes them (fp32 softmax over the legal logits)
    root_priors = numpy.zeros(A, dtype="float64")
    root_priors[: len(legal_actions)] = torch.softmax(
        torch.tensor([rp[0][a] for a in legal_actions]), dim=0).tolist()
    rec["root_priors"] = root_priors
    rec["root_reward"] = float(root.reward)
    noise = numpy.zeros(A, dtype="float64")
    noise[: len(legal_actions)] = noises[0]
    rec["noise"] = noise
    prior = numpy.zeros(A, dtype="float64")
    visits = numpy.zeros(A, dtype="int32")
    cvs = numpy.zeros(A, dtype="float64")
    crew = numpy.zeros(A, dtype="float64")
    for i, a in enumerate(legal_actions):
        c = root.children[a]
        prior[i], visits[i], cvs[i], crew[i] = c.prior, c.visit_count, c.value_sum, c.reward
    # root-level arrays are indexed by child SLOT (order of legal_actions)
    rec["child_prior"] = prior
    rec["visits"] = visits
    rec["child_value_sum"] = cvs
    rec["child_reward"] = crew
    rec["root_value_sum"] = float(root.value_sum)
    rec["root_visit"] = int(root.visit_count)
    rec["max_tree_depth"] = int(info["max_tree_depth"])
    rec["mms_min"] = float(mcts.mms.minimum)
    rec["mms_max"] = float(mcts.mms.maximum)
    rec["rng_words_run"] = rng_words_between(st0, st1)
    rec["rng_words_select"] = rng_words_between(st1, st2)
    rec["action_T"] = int(action)
    rec["temperature"] = float(temperature)
    gh = self_play.GameHistory()
    gh.store_search_statistics(root, config.action_space)
    rec["child_visits_target"] = numpy.array(gh.child_visits[0], dtype="float64")
    rec["root_value_target"] = float(gh.root_values[0])

    sim_depth = numpy.zeros(S, dtype="int32")
    sim_actions = numpy.full((S, D), -1, dtype="int16")
    sim_ties = numpy.zeros((S, D), dtype="int16")
    sim_maxucb = numpy.zeros((S, D), dtype="float64")
    sim_value = numpy.zeros(S, dtype="float64")
    sim_reward = numpy.zeros(S, dtype="float64")
    sim_priors = numpy.zeros((S, A), dtype="float64")
    sim_policy_logits = numpy.zeros((S, A), dtype="float32")
    sim_value_logits = numpy.zeros((S, F), dtype="float32")
    sim_reward_logits = numpy.zeros((S, F), dtype="float32")
    H = rec["root_hidden"].size
    keep_hidden = H <= 64
    sim_parent_hidden = numpy.zeros((S, H if keep_hidden else 0), dtype="float32")
    sim_next_hidden = numpy.zeros((S, H if keep_hidden else 0), dtype="float32")
    assert len(mcts.sims) == S and len(proxy.log) == S
    for s in range(S):
        steps = mcts.sims[s]
        sim_depth[s] = len(steps)
        for d, (ties, act, mx) in enumerate(steps):
            sim_actions[s, d] = act
            sim_ties[s, d] = ties
            sim_maxucb[s, d] = mx
        state, act, (v, r, p, nh) = proxy.log[s]
        assert act == steps[-1][1]
        sim_value[s] = models.support_to_scalar(v, config.support_size).item()
        sim_reward[s] = models.support_to_scalar(r, config.support_size).item()
        sim_priors[s] = torch.softmax(
            torch.tensor([p[0][a] for a in config.action_space]), dim=0).tolist()
        sim_policy_logits[s] = p[0].numpy()
        sim_value_logits[s] = v[0].numpy()
        sim_reward_logits[s] = r[0].numpy()
        if keep_hidden:
            sim_parent_hidden[s] = state[0].numpy().reshape(-1)
            sim_next_hidden[s] = nh[0].numpy().reshape(-1)
    rec.update(sim_depth=sim_depth, sim_actions=sim_actions, sim_ties=sim_ties,
               sim_maxucb=sim_maxucb, sim_value=sim_value, sim_reward=sim_reward,
               sim_priors=sim_priors, sim_policy_logits=sim_policy_logits,
               sim_value_logits=sim_value_logits, sim_reward_logits=sim_reward_logits,
               sim_parent_hidden=sim_parent_hidden, sim_next_hidden=sim_next_hidden)
    return rec


def stack_records(records):
    keys = records[0].keys()
    return {k: numpy.stack([numpy.asarray(r[k]) for r in records]) for k in keys}


def config_scalars(config):
    return dict(
        cfg_A=len(config.action_space), cfg_S=config.num_simulations,
        cfg_players=len(config.players), cfg_discount=float(config.discount),
        cfg_pb_c_base=float(config.pb_c_base), cfg_pb_c_init=float(config.pb_c_init),
        cfg_alpha=float(config.root_dirichlet_alpha),
        cfg_frac=float(config.root_exploration_fraction),
        cfg_support=int(config.support_size),
    )


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    numpy.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz  ({os.path.getsize(path)/1024:.1f} KiB)")


def build_model(models, config, weights=None, seed=0):
    model = models.MuZeroNetwork(config)
    if weights is None:
        sd = synthetic_state_dict(model.state_dict(), seed)
        weights = {k: torch.from_numpy(v) for k, v in sd.items()}
    model.set_weights(weights)
    model.eval()
    return model


# --------------------------------------------------------------------------------------
def g0_weights(ctx):
    w = load_cartpole_checkpoint()
    save("cartpole_weights", **{k: v.numpy() for k, v in w.items()})


def g1_support_to_scalar(ctx):
    models = ctx["models"]
    rs = numpy.random.RandomState(11)
    l21 = (3.0 * rs.standard_normal((64, 21))).astype("float32")
    l601 = (2.0 * rs.standard_normal((8, 601))).astype("float32")
    onehot = torch.log(torch.zeros(1, 21).scatter(1, torch.tensor([[10]]), 1.0)).repeat(3, 1)
    save("g1_support_to_scalar",
         logits21=l21, out21=models.support_to_scalar(torch.from_numpy(l21), 10).numpy(),
         logits601=l601, out601=models.support_to_scalar(torch.from_numpy(l601), 300).numpy(),
         logits_init=onehot.numpy(), out_init=models.support_to_scalar(onehot, 10).numpy())


def _inference_fixture(models, config, model, obs, actions):
    with torch.no_grad():
        v0, r0, p0, h0 = model.initial_inference(torch.from_numpy(obs))
        v1, r1, p1, h1 = model.recurrent_inference(h0, torch.from_numpy(actions))
    return dict(obs=obs, actions=actions,
                init_value=v0.numpy(), init_reward=r0.numpy(), init_policy=p0.numpy(),
                init_hidden=h0.numpy(), rec_value=v1.numpy(), rec_reward=r1.numpy(),
                rec_policy=p1.numpy(), rec_hidden=h1.numpy())


def g2_fc_inference(ctx):
    models, cfgs = ctx["models"], ctx["configs"]
    config = cfgs["cartpole"]
    model = build_model(models, config, load_cartpole_checkpoint())
    rs = numpy.random.RandomState(5)
    obs = rs.uniform(-0.5, 0.5, (64, 1, 1, 4)).astype("float32")
    actions = rs.randint(0, 2, (64, 1)).astype("int64")
    save("g2_fc_inference", **_inference_fixture(models, config, model, obs, actions))


def g3_resnet_inference(ctx):
    models, cfgs = ctx["models"], ctx["configs"]
    rs = numpy.random.RandomState(6)
    for name, B in (("tictactoe", 16), ("connect4", 8), ("atari84", 4)):
        config = cfgs[name]
        model = build_model(models, config, None, seed=0)
        C, Hh, W = config.observation_shape
        if name == "atari84":
            obs = rs.uniform(0, 1, (B, C, Hh, W)).astype("float32")
        else:
            obs = rs.randint(0, 2, (B, C, Hh, W)).astype("float32")
            obs[:, 2] = numpy.where(rs.randint(0, 2, (B, 1, 1)) > 0, 1.0, -1.0)
        actions = rs.randint(0, len(config.action_space), (B, 1)).astype("int64")
        fx = _inference_fixture(models, config, model, obs, actions)
        fx["state_dict_keys"] = numpy.array(list(model.state_dict().keys()))
        save(f"g3_{name}_inference", **fx)


def g4_cartpole(ctx):
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    config = cfgs["cartpole"]
    model = build_model(models, config, load_cartpole_checkpoint())
    recs = []
    survey_obs = numpy.array([[[0.01, -0.02, 0.03, 0.04]]], dtype="float32")
    for seed in range(4):
        recs.append(trace_one(models, self_play, config, model, survey_obs, [0, 1], 0, seed))
    rs = numpy.random.RandomState(123)
    obs = rs.uniform(-0.05, 0.05, (28, 1, 1, 4)).astype("float32")
    for i in range(28):
        recs.append(trace_one(models, self_play, config, model, obs[i], [0, 1], 0, 4 + i,
                              temperature=[1.0, 0.5, 0.25, 1.0][i % 4]))
    arrays = stack_records(recs)
    arrays.update(config_scalars(config))
    print("   cartpole mean select depth:", arrays["sim_depth"].mean(),
          "first visits:", arrays["visits"][:4].tolist())
    save("g4_cartpole_traces", **arrays)


def _board_positions(Game, n_positions, rs, max_prefix):
    """Positions reached by random legal playouts of the reference env (not terminal)."""
    out = []
    while len(out) < n_positions:
        g = Game(0)
        obs = g.reset()
        k = rs.randint(0, max_prefix + 1)
        done = False
        for _ in range(k):
            a = int(rs.choice(g.legal_actions()))
            obs, _, done = g.step(a)
            if done:
                break
        if not done:
            out.append((numpy.array(obs, dtype="float32"), list(g.legal_actions()), g.to_play()))
    return out


def g5_tictactoe(ctx):
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    import games.tictactoe as ttt
    config = cfgs["tictactoe"]
    model = build_model(models, config, None, seed=0)
    rs = numpy.random.RandomState(77)
    recs = []
    empty = numpy.array(ttt.Game(0).reset(), dtype="float32")
    # artificial restricted root on the empty board (masked root, full action space below)
    recs.append(trace_one(models, self_play, config, model, empty, [0, 2, 4, 5, 8], 0, 0))
    recs.append(trace_one(models, self_play, config, model, empty, [7], 0, 1))
    recs.append(trace_one(models, self_play, config, model, empty, [8, 3, 1], 1, 2))  # unsorted
    for i, (obs, legal, tp) in enumerate(_board_positions(ttt.Game, 29, rs, 6)):
        recs.append(trace_one(models, self_play, config, model, obs, legal, tp, 3 + i,
                              temperature=[1.0, 0.5, 1.0, 0.25][i % 4]))
    arrays = stack_records(recs)
    arrays.update(config_scalars(config))
    print("   tictactoe mean select depth:", arrays["sim_depth"].mean())
    save("g5_tictactoe_traces", **arrays)


def g5_connect4(ctx):
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    import games.connect4 as c4
    config = cfgs["connect4"]
    model = build_model(models, config, None, seed=0)
    rs = numpy.random.RandomState(78)
    recs = []
    # 30 positions (the first 6 are round 1's: the playout stream is consumed in order)
    for i, (obs, legal, tp) in enumerate(_board_positions(c4.Game, 30, rs, 30)):
        recs.append(trace_one(models, self_play, config, model, obs, legal, tp, 100 + i,
                              temperature=1.0 if i < 6 else [1.0, 0.5, 0.25, 1.0][i % 4]))
    arrays = stack_records(recs)
    arrays.update(config_scalars(config))
    print("   connect4 mean select depth:", arrays["sim_depth"].mean())
    save("g5_connect4_traces", **arrays)


def g5_atari84(ctx):
    """BASELINE.json config #5 end to end: DownsampleCNN representation (models.py:278-297, 318-327) on 4 stacked
    84x84 frames, 2 residual blocks x 16 channels at 6x6, A = 4, one player, 50 simulations."""
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    config = cfgs["atari84"]
    model = build_model(models, config, None, seed=0)
    rs = numpy.random.RandomState(84)
    recs = []
    for i in range(24):
        # 8-bit frames scaled to [0, 1] like real Atari input; stored as the uint8 levels (obs = u8 / 255 in float32)
        obs = rs.randint(0, 256, config.observation_shape).astype("uint8").astype("float32") / numpy.float32(255)
        recs.append(trace_one(models, self_play, config, model, obs, [0, 1, 2, 3], 0, 300 + i,
                              temperature=[1.0, 0.5, 0.25, 1.0][i % 4]))
    arrays = stack_records(recs)
    u8 = numpy.rint(arrays.pop("obs") * 255).astype("uint8")
    assert numpy.array_equal(u8.astype("float32") / numpy.float32(255), numpy.stack([r["obs"] for r in recs]))
    arrays["obs_u8"] = u8
    arrays.update(config_scalars(config))
    print("   atari84 mean select depth:", arrays["sim_depth"].mean())
    save("g5_atari84_traces", **arrays)


def g5_degenerate(ctx):
    """All-equal priors (zeroed policy head) => ties at every level: stresses the tie-break
    RNG path far beyond what trained/random nets produce."""
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    config = cfgs["cartpole"]
    w = {k: v.clone() for k, v in load_cartpole_checkpoint().items()}
    for k in w:
        if k.startswith("prediction_policy_network.module.2"):
            w[k].zero_()
    model = build_model(models, config, w)
    rs = numpy.random.RandomState(9)
    recs = []
    for i in range(8):
        obs = rs.uniform(-0.05, 0.05, (1, 1, 4)).astype("float32")
        recs.append(trace_one(models, self_play, config, model, obs, [0, 1], 0, 200 + i))
    arrays = stack_records(recs)
    arrays.update(config_scalars(config))
    print("   degenerate: rng words per run", arrays["rng_words_run"].tolist())
    save("g5_cartpole_ties_traces", **arrays)


def _history_arrays(gh, A):
    n = len(gh.action_history)
    rv = numpy.array([numpy.nan if v is None else v for v in gh.root_values], dtype="float64")
    return dict(
        actions=numpy.array(gh.action_history, dtype="int32"),
        rewards=numpy.array(gh.reward_history, dtype="float64"),
        to_play=numpy.array(gh.to_play_history, dtype="int32"),
        child_visits=numpy.array(gh.child_visits, dtype="float64").reshape(-1, A),
        root_values=rv,
        observations=numpy.array(gh.observation_history, dtype="float32"),
        length=n,
    )


def g6_play_game(ctx, only=None):
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    import games.connect4 as c4
    import games.tictactoe as ttt
    plans = [
        ("tictactoe", "tictactoe", ttt.Game, [(s, 1.0, None, "self", 0) for s in range(4)]
         + [(10, 0, None, "expert", 0), (11, 0, None, "random", 1), (12, 1.0, 4, "self", 0)]),
        ("connect4", "connect4", c4.Game, [(s, 1.0, None, "self", 0) for s in range(2)]),
        # round 3: Connect4 test-mode games (SelfPlay.select_opponent_action, self_play.py:189-221, against
        # games/connect4.py:306-343's expert; MuZero as either player; a random opponent) and the temperature threshold
        ("connect4_opponents", "connect4", c4.Game, [(20, 0, None, "expert", 0), (21, 0, None, "expert", 1),
                                                     (22, 0, None, "random", 1), (23, 1.0, 6, "self", 0)]),
    ]
    for fixture, name, Game, runs in plans:
        if only is not None and fixture not in only:
            continue
        config = cfgs[name]
        A = len(config.action_space)
        tmpl = models.MuZeroNetwork(config).state_dict()
        weights = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(tmpl, 0).items()}
        out = {}
        for i, (seed, temp, thr, opponent, mzp) in enumerate(runs):
            t0 = time.time()
            sp = self_play.SelfPlay({"weights": weights}, Game, config, seed)
            gh = sp.play_game(temp, thr, False, opponent, mzp)
            st = numpy.random.get_state()
            for k, v in _history_arrays(gh, A).items():
                out[f"run{i}_{k}"] = v
            out[f"run{i}_args"] = numpy.array(
                [seed, temp, -1 if thr is None else thr,
                 {"self": 0, "expert": 1, "random": 2}[opponent], mzp], dtype="float64")
            out[f"run{i}_rng_pos_end"] = int(st[2])
            out[f"run{i}_rng_next_word"] = int(numpy.random.randint(0, 2**31 - 1))
            print(f"   {fixture} run{i} seed={seed} opp={opponent}: {len(gh.action_history)-1} moves"
                  f" in {time.time()-t0:.1f}s")
        out["n_runs"] = len(runs)
        out.update(config_scalars(config))
        save(f"g6_{fixture}_games", **out)


def g6_connect4_opponents(ctx):
    g6_play_game(ctx, only=("connect4_opponents",))


def g7_rng(ctx):
    out = {}
    numpy.random.seed(0)
    out["seed0_choice2x10"] = numpy.array([numpy.random.choice([0, 1]) for _ in range(10)])
    numpy.random.seed(0)
    out["seed0_dirichlet_025x2"] = numpy.random.dirichlet([0.25] * 2)
    for seed in (0, 1, 12345, 2**32 - 1):
        numpy.random.seed(seed)
        out[f"seed{seed}_words"] = numpy.array(
            [numpy.random.randint(0, 2**32, dtype="uint32") for _ in range(4)], dtype="uint32")
        numpy.random.seed(seed)
        out[f"seed{seed}_doubles"] = numpy.random.random_sample(700)
        numpy.random.seed(seed)
        out[f"seed{seed}_choice"] = numpy.array(
            [numpy.random.choice(list(range(k))) for k in (2, 3, 5, 7, 9, 4, 6, 8, 121, 1, 2)])
        for alpha, k in ((0.25, 2), (0.1, 9), (0.3, 7), (0.25, 4), (1.0, 3), (2.5, 5), (0.03, 121)):
            numpy.random.seed(seed)
            d = [numpy.random.dirichlet([alpha] * k) for _ in range(6)]
            out[f"seed{seed}_dirichlet_{alpha}_{k}"] = numpy.array(d)
            out[f"seed{seed}_dirichlet_{alpha}_{k}_next"] = numpy.random.random_sample(2)
    numpy.random.seed(3)
    ps, picks = [], []
    for i in range(40):
        p = numpy.random.RandomState(i).dirichlet([1.0] * 5)
        ps.append(p)
        picks.append(numpy.random.choice([10, 11, 12, 13, 14], p=p))
    out["seed3_choice_p"] = numpy.array(ps)
    out["seed3_choice_p_picks"] = numpy.array(picks)
    save("g7_numpy_rng", **out)


def g8_select_action(ctx):
    self_play = ctx["self_play"]
    rs = numpy.random.RandomState(21)
    visit_sets = [rs.multinomial(50, rs.dirichlet([0.7] * A)).astype("int32")
                  for A in (2, 2, 9, 9, 7, 4, 4, 9)]
    visit_sets.append(numpy.array([25, 25], dtype="int32"))
    visit_sets.append(numpy.array([0, 50], dtype="int32"))
    out = {"n_sets": len(visit_sets)}
    for i, v in enumerate(visit_sets):
        root = self_play.Node(0)
        actions = list(range(len(v)))
        if i == 3:
            actions = [8, 7, 6, 5, 4, 3, 2, 1, 0]
        for a, n in zip(actions, v):
            root.children[a] = self_play.Node(0.1)
            root.children[a].visit_count = int(n)
        out[f"set{i}_visits"] = v
        out[f"set{i}_actions"] = numpy.array(actions, dtype="int32")
        for T in (0, 0.25, 0.5, 1.0, 0.7, float("inf")):
            numpy.random.seed(100 + i)
            picks = [int(self_play.SelfPlay.select_action(root, T)) for _ in range(12)]
            out[f"set{i}_T{T}"] = numpy.array(picks, dtype="int32")
    save("g8_select_action", **out)


def g9_stacked(ctx):
    self_play = ctx["self_play"]
    rs = numpy.random.RandomState(31)
    gh = self_play.GameHistory()
    obs = [rs.standard_normal((3, 3, 3)).astype("float32") for _ in range(6)]
    acts = [0] + [int(a) for a in rs.randint(0, 9, 5)]
    for o, a in zip(obs, acts):
        gh.observation_history.append(o)
        gh.action_history.append(a)
    out = {"observations": numpy.array(obs), "actions": numpy.array(acts, dtype="int32")}
    for n_stack in (0, 2, 4):
        for idx in (-1, 0, 1, 3, 5):
            out[f"stack{n_stack}_idx{idx}"] = numpy.asarray(
                gh.get_stacked_observations(idx, n_stack), dtype="float32")
    save("g9_stacked_observations", **out)


def g10_reference_speed(ctx):
    """Indicative reference timing in THIS container (1 thread) for BASELINE.md / DESIGN.md."""
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    torch.set_num_threads(1)
    config = cfgs["cartpole"]
    model = build_model(models, config, load_cartpole_checkpoint())
    rs = numpy.random.RandomState(123)
    obs = rs.uniform(-0.05, 0.05, (4096, 1, 1, 4)).astype("float32")
    numpy.random.seed(0)
    n_moves = 60
    with torch.no_grad():
        for i in range(5):
            self_play.MCTS(config).run(model, obs[i], [0, 1], 0, True)
        t0 = time.time()
        for i in range(n_moves):
            root, _ = self_play.MCTS(config).run(model, obs[5 + i], [0, 1], 0, True)
            self_play.SelfPlay.select_action(root, 1.0)
        dt = time.time() - t0
    sims = n_moves * config.num_simulations
    print(f"   reference cartpole: {sims/dt:.0f} sims/s, {n_moves/dt:.1f} moves/s (1 thread)")
    save("g10_reference_speed", cartpole_sims_per_s=sims / dt, moves=n_moves,
         cpu=numpy.array(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0]))


def g11_envs(ctx):
    """Board-game plugin behaviour: random legal playouts of the reference envs with, at every
    position, the observation, legal list, to_play and the expert agent's move under a fixed seed."""
    import games.connect4 as c4
    import games.tictactoe as ttt
    for name, mod, n_games in (("tictactoe", ttt, 40), ("connect4", c4, 25)):
        rs = numpy.random.RandomState(2024)
        rows = dict(game=[], step=[], action=[], reward=[], done=[], to_play=[], expert=[],
                    n_legal=[], legal=[], obs=[])
        A = len(mod.MuZeroConfig().action_space)
        for g in range(n_games):
            game = mod.Game(g)
            obs = game.reset()
            t, done, action, reward = 0, False, -1, 0
            while True:
                legal = list(game.legal_actions())
                rows["game"].append(g); rows["step"].append(t); rows["action"].append(action)
                rows["reward"].append(reward); rows["done"].append(done)
                rows["to_play"].append(game.to_play()); rows["n_legal"].append(len(legal))
                rows["legal"].append(legal + [-1] * (A - len(legal)))
                rows["obs"].append(numpy.asarray(obs, dtype="float32"))
                if done:
                    rows["expert"].append(-1)
                    break
                numpy.random.seed(1000 + 31 * g + t)
                rows["expert"].append(int(game.expert_agent()))
                action = int(rs.choice(legal))
                obs, reward, done = game.step(action)
                t += 1
        save(f"g11_{name}_env", **{k: numpy.array(v) for k, v in rows.items()})


def _synthetic_history(self_play, rs, config, length):
    """A GameHistory with the field types play_game leaves behind (self_play.py:116-121, 176-182, 497-512):
    numpy observations, int actions, float rewards, int to_play, child_visits = lists of floats, float root values."""
    A, P = len(config.action_space), len(config.players)
    gh = self_play.GameHistory()
    to_play = int(rs.randint(0, P))
    gh.action_history.append(0)
    gh.observation_history.append(rs.standard_normal(config.observation_shape).astype("float32"))
    gh.reward_history.append(0)
    gh.to_play_history.append(to_play)
    for _ in range(length):
        visits = rs.multinomial(config.num_simulations, rs.dirichlet([0.6] * A))
        gh.child_visits.append([int(v) / config.num_simulations for v in visits])
        gh.root_values.append(float(rs.standard_normal() * 3))
        gh.action_history.append(int(rs.randint(0, A)))
        gh.observation_history.append(rs.standard_normal(config.observation_shape).astype("float32"))
        gh.reward_history.append(float(numpy.float32(rs.standard_normal())) if P == 1 else float(rs.randint(0, 2)))
        to_play = (to_play + 1) % P
        gh.to_play_history.append(to_play)
    return gh


def g12_replay_targets(ctx):
    """ReplayBuffer.save_game (initial priorities), get_batch / make_target / compute_target_value and the
    observation stacking (replay_buffer.py:33-65, 67-133, 222-295) on synthetic game histories."""
    import copy
    import replay_buffer
    self_play, cfgs = ctx["self_play"], ctx["configs"]
    for name, tweak in (("cartpole", {}), ("tictactoe", {}), ("tictactoe_stacked", {"stacked_observations": 2}),
                        ("cartpole_uniform", {"PER": False})):
        config = copy.deepcopy(cfgs[name.split("_")[0]])
        for k, v in tweak.items():
            setattr(config, k, v)
        config.batch_size = 48
        rs = numpy.random.RandomState(77)
        lengths = [int(v) for v in rs.randint(1, 70 if len(config.players) == 1 else 10, 24)]
        games = [_synthetic_history(self_play, rs, config, n) for n in lengths]
        rb = replay_buffer.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
        for gh in games:
            rb.save_game(gh)
        index_batch, (obs_b, act_b, val_b, rew_b, pol_b, w_b, gs_b) = rb.get_batch()
        L = max(lengths)
        A = len(config.action_space)
        G = len(games)
        out = dict(config_scalars(config))
        out.update(td_steps=config.td_steps, num_unroll_steps=config.num_unroll_steps, PER=int(config.PER),
                   PER_alpha=config.PER_alpha, stacked_observations=config.stacked_observations,
                   observation_shape=numpy.array(config.observation_shape), batch_size=config.batch_size,
                   seed=config.seed, lengths=numpy.array(lengths, dtype="int32"))
        obs = numpy.zeros((G, L + 1) + tuple(config.observation_shape), dtype="float32")
        act = numpy.zeros((G, L + 1), dtype="int32")
        rew = numpy.zeros((G, L + 1), dtype="float64")
        tp = numpy.zeros((G, L + 1), dtype="int32")
        cv = numpy.zeros((G, L, A), dtype="float64")
        rv = numpy.zeros((G, L), dtype="float64")
        pri = numpy.zeros((G, L), dtype="float32")
        game_pri = numpy.zeros(G, dtype="float32")
        for g, gh in enumerate(games):
            n = lengths[g]
            obs[g, : n + 1] = numpy.array(gh.observation_history)
            act[g, : n + 1] = gh.action_history
            rew[g, : n + 1] = gh.reward_history
            tp[g, : n + 1] = gh.to_play_history
            cv[g, :n] = gh.child_visits
            rv[g, :n] = gh.root_values
            if config.PER:
                pri[g, :n] = gh.priorities
                game_pri[g] = gh.game_priority
        out.update(observations=obs, actions=act, rewards=rew, to_play=tp, child_visits=cv, root_values=rv,
                   priorities=pri, game_priority=game_pri,
                   index_batch=numpy.array(index_batch, dtype="int64"),
                   observation_batch=numpy.array(obs_b, dtype="float32"),
                   action_batch=numpy.array(act_b, dtype="int64"),
                   value_batch=numpy.array(val_b, dtype="float64"),
                   reward_batch=numpy.array(rew_b, dtype="float64"),
                   policy_batch=numpy.array(pol_b, dtype="float64"),
                   gradient_scale_batch=numpy.array(gs_b, dtype="float64"))
        if config.PER:
            out["weight_batch"] = numpy.array(w_b, dtype="float32")
        save(f"g12_replay_{name}", **out)


def g13_reanalyse(ctx):
    """Reanalyse's per-game step (replay_buffer.py:335-356) -- stacked observations of every position, one
    batched initial_inference, support_to_scalar -- and the targets ReplayBuffer.make_target builds once a game
    carries reanalysed_predicted_root_values (replay_buffer.py:226-231: numpy float32 scalars from then on)."""
    import copy
    import replay_buffer
    models, self_play, cfgs = ctx["models"], ctx["self_play"], ctx["configs"]
    config = copy.deepcopy(cfgs["cartpole"])
    model = build_model(models, config, load_cartpole_checkpoint())
    rs = numpy.random.RandomState(99)
    lengths = [int(v) for v in rs.randint(5, 120, 6)]
    games = [_synthetic_history(self_play, rs, config, n) for n in lengths]
    for gh in games:                       # observations the network can digest: small, like CartPole states
        gh.observation_history = [numpy.float32(0.05) * o for o in gh.observation_history]
    rb = replay_buffer.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    for gh in games:
        rb.save_game(gh)
    L, A, G = max(lengths), len(config.action_space), len(games)
    reanalysed = numpy.zeros((G, L), dtype="float32")
    for g, gh in enumerate(games):
        observations = [gh.get_stacked_observations(i, config.stacked_observations) for i in range(len(gh.root_values))]
        observations = torch.tensor(numpy.array(observations)).float()
        with torch.no_grad():
            values = models.support_to_scalar(model.initial_inference(observations)[0], config.support_size)
        gh.reanalysed_predicted_root_values = torch.squeeze(values).detach().cpu().numpy()
        reanalysed[g, : lengths[g]] = gh.reanalysed_predicted_root_values
        rb.update_game_history(g, gh)
    pairs, values, rewards, policies, actions = [], [], [], [], []
    numpy.random.seed(1234)
    for g, gh in enumerate(games):
        for pos in sorted(set([0, lengths[g] // 3, lengths[g] - 2, lengths[g] - 1])):
            if pos < 0:
                continue
            v, r, p, a = rb.make_target(rb.buffer[g], pos)
            pairs.append([g, pos])
            values.append([float(x) for x in v])
            rewards.append([float(x) for x in r])
            policies.append(p)
            actions.append(a)
    out = dict(config_scalars(config))
    obs = numpy.zeros((G, L + 1) + tuple(config.observation_shape), dtype="float32")
    act = numpy.zeros((G, L + 1), dtype="int32")
    rew = numpy.zeros((G, L + 1), dtype="float64")
    tp = numpy.zeros((G, L + 1), dtype="int32")
    cv = numpy.zeros((G, L, A), dtype="float64")
    rv = numpy.zeros((G, L), dtype="float64")
    for g, gh in enumerate(games):
        n = lengths[g]
        obs[g, : n + 1] = numpy.array(gh.observation_history)
        act[g, : n + 1] = gh.action_history
        rew[g, : n + 1] = gh.reward_history
        tp[g, : n + 1] = gh.to_play_history
        cv[g, :n] = gh.child_visits
        rv[g, :n] = gh.root_values
    out.update(lengths=numpy.array(lengths, dtype="int32"), observations=obs, actions=act, rewards=rew, to_play=tp,
               child_visits=cv, root_values=rv, reanalysed=reanalysed, pairs=numpy.array(pairs, dtype="int32"),
               value_targets=numpy.array(values, dtype="float64"), reward_targets=numpy.array(rewards, dtype="float64"),
               policy_targets=numpy.array(policies, dtype="float64"), action_targets=numpy.array(actions, dtype="int64"),
               td_steps=config.td_steps, num_unroll_steps=config.num_unroll_steps, seed=1234,
               numpy_version=numpy.array(numpy.__version__))
    save("g13_reanalyse_cartpole", **out)


def g14_trainer(ctx):
    """Trainer.update_lr / update_weights (trainer.py:124-298) for two steps on one ReplayBuffer batch, starting
    from the CartPole checkpoint: losses, new priorities and the weights after each step."""
    import copy
    import replay_buffer
    import trainer
    self_play, cfgs = ctx["self_play"], ctx["configs"]
    config = copy.deepcopy(cfgs["cartpole"])
    config.batch_size = 32
    config.train_on_gpu = False
    rs = numpy.random.RandomState(314)
    lengths = [int(v) for v in rs.randint(8, 90, 12)]
    games = [_synthetic_history(self_play, rs, config, n) for n in lengths]
    for gh in games:
        gh.observation_history = [numpy.float32(0.05) * o for o in gh.observation_history]
    rb = replay_buffer.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    for gh in games:
        rb.save_game(gh)
    index_batch, batch = rb.get_batch()
    weights = load_cartpole_checkpoint()
    tr = trainer.Trainer({"weights": copy.deepcopy(weights), "training_step": 0, "optimizer_state": None}, config)
    out = dict(config_scalars(config))
    obs_b, act_b, val_b, rew_b, pol_b, w_b, gs_b = batch
    out.update(observation_batch=numpy.array(obs_b, dtype="float32"), action_batch=numpy.array(act_b, dtype="int64"),
               value_batch=numpy.array(val_b, dtype="float64"), reward_batch=numpy.array(rew_b, dtype="float64"),
               policy_batch=numpy.array(pol_b, dtype="float64"), weight_batch=numpy.array(w_b, dtype="float32"),
               gradient_scale_batch=numpy.array(gs_b, dtype="float64"))
    for step in range(2):
        tr.update_lr()
        out[f"lr{step}"] = tr.optimizer.param_groups[0]["lr"]
        priorities, total, v, r, p = tr.update_weights(batch)
        out[f"priorities{step}"] = numpy.asarray(priorities, dtype="float32")
        out[f"losses{step}"] = numpy.array([total, v, r, p], dtype="float64")
        for k, t in tr.model.get_weights().items():
            out[f"w{step}_{k}"] = t.detach().cpu().numpy().copy()   # (get_weights aliases the live parameters)
    save("g14_trainer_cartpole", **out)


def g16_trainer_resnet(ctx):
    """The same two Trainer steps on the TicTacToe residual network (BatchNorm in train(), 3x3 and 1x1 convolutions,
    two players): pins the training forward / backward of the residual modules (models.py:206-522, trainer.py:124-298)."""
    import copy
    import replay_buffer
    import trainer
    self_play, cfgs = ctx["self_play"], ctx["configs"]
    config = copy.deepcopy(cfgs["tictactoe"])
    config.batch_size = 24
    config.train_on_gpu = False
    rs = numpy.random.RandomState(2718)
    lengths = [int(v) for v in rs.randint(3, 10, 14)]
    games = [_synthetic_history(self_play, rs, config, n) for n in lengths]
    for gh in games:                       # board-like observations
        gh.observation_history = [numpy.sign(o).astype("float32") for o in gh.observation_history]
    rb = replay_buffer.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    for gh in games:
        rb.save_game(gh)
    index_batch, batch = rb.get_batch()
    template = ctx["models"].MuZeroNetwork(config).state_dict()
    weights = {k: torch.from_numpy(v) for k, v in synthetic_state_dict(template, 0).items()}
    tr = trainer.Trainer({"weights": copy.deepcopy(weights), "training_step": 0, "optimizer_state": None}, config)
    out = dict(config_scalars(config))
    obs_b, act_b, val_b, rew_b, pol_b, w_b, gs_b = batch
    out.update(observation_batch=numpy.array(obs_b, dtype="float32"), action_batch=numpy.array(act_b, dtype="int64"),
               value_batch=numpy.array(val_b, dtype="float64"), reward_batch=numpy.array(rew_b, dtype="float64"),
               policy_batch=numpy.array(pol_b, dtype="float64"), weight_batch=numpy.array(w_b, dtype="float32"),
               gradient_scale_batch=numpy.array(gs_b, dtype="float64"))
    for step in range(2):
        tr.update_lr()
        out[f"lr{step}"] = tr.optimizer.param_groups[0]["lr"]
        priorities, total, v, r, p = tr.update_weights(batch)
        out[f"priorities{step}"] = numpy.asarray(priorities, dtype="float32")
        out[f"losses{step}"] = numpy.array([total, v, r, p], dtype="float64")
        for k, t in tr.model.get_weights().items():      # weights (and BatchNorm running statistics) after the step
            out[f"w{step}_{k}"] = t.detach().cpu().numpy().copy()
        for k, prm in tr.model.named_parameters():       # round 3: the gradients the step was made from (still in .grad)
            out[f"g{step}_{k}"] = prm.grad.detach().cpu().numpy().copy()
    save("g16_trainer_tictactoe", **out)


class _LoopStorage:
    """Fake shared_storage / replay_buffer for the continuous_self_play fixture: every call is logged; the
    training step advances by `step_per_game` whenever a game is saved or a test result is stored, so the loop ends."""

    def __init__(self, log, training_steps_per_event, played_steps_per_game):
        self.log = log
        self.info = {"training_step": 0, "terminate": False, "weights": None, "num_played_steps": 0,
                     "num_played_games": 0}
        self.per_event = training_steps_per_event
        self.played = played_steps_per_game
        outer = self

        class _Remote:
            def __init__(self, fn):
                self.remote = fn
        self.get_info = _Remote(self._get)
        self.set_info = _Remote(self._set)
        self.save_game = _Remote(self._save)

    def _get(self, key):
        self.log.append(["get_info", key])
        return self.info[key]

    def _set(self, keys, values=None):
        self.log.append(["set_info", {k: (float(v) if v is not None else None) for k, v in keys.items()}])
        self.info["training_step"] += self.per_event
        return None

    def _save(self, game_history, shared_storage=None):
        self.log.append(["save_game", len(game_history.action_history) - 1])
        self.info["training_step"] += self.per_event
        self.info["num_played_steps"] += self.played
        return None


def g15_self_play_loop(ctx):
    """SelfPlay.continuous_self_play (self_play.py:31-108) with play_game replaced by canned games: the exact
    sequence of shared-storage / replay-buffer calls, temperatures, test-mode metric dicts and ratio sleeps."""
    import json
    import time as _time
    import games.tictactoe as ttt
    self_play, cfgs = ctx["self_play"], ctx["configs"]
    import copy
    out = {}
    for name, test_mode, ratio, delay, players in (("train", False, None, 0, 2), ("train_ratio", False, 0.6, 0.25, 2),
                                                   ("test_two_player", True, None, 0, 2), ("test_one_player", True, None, 0, 1)):
        config = copy.deepcopy(cfgs["tictactoe"])
        config.training_steps = 7
        config.ratio = ratio
        config.self_play_delay = delay
        config.temperature_threshold = 4
        if players == 1:
            config.players = [0]
        config.visit_softmax_temperature_fn = lambda trained_steps: 1.0 if trained_steps < 3 else 0.25
        weights = {k: torch.from_numpy(v) for k, v in
                   synthetic_state_dict(ctx["models"].MuZeroNetwork(config).state_dict(), 0).items()}
        log = []
        sp = self_play.SelfPlay({"weights": weights}, ttt.Game, config, 0)
        storage = _LoopStorage(log, 2, 5)
        storage.info["weights"] = weights
        rs = numpy.random.RandomState(3)

        def canned_play_game(temperature, temperature_threshold, render, opponent, muzero_player, rs=rs, log=log):
            log.append(["play_game", float(temperature), temperature_threshold, bool(render), opponent, int(muzero_player)])
            gh = self_play.GameHistory()
            n = int(rs.randint(3, 8))
            gh.action_history = [0] + [int(a) for a in rs.randint(0, 9, n)]
            gh.reward_history = [0] + [float(r) for r in rs.randint(-1, 2, n)]
            gh.to_play_history = [int(i % 2) for i in range(n + 1)]
            gh.root_values = [float(v) for v in rs.standard_normal(n)]
            gh.root_values[1] = 0.0                      # (mean_value skips falsy values, self_play.py:69-71)
            gh.child_visits = [[1.0 / 9] * 9 for _ in range(n)]
            gh.observation_history = [numpy.zeros((3, 3, 3), "float32")] * (n + 1)
            return gh
        sp.play_game = canned_play_game
        sp.close_game = lambda log=log: log.append(["close_game"])
        orig_sleep = _time.sleep
        self_play.time.sleep = lambda t, log=log, storage=storage: (log.append(["sleep", float(t)]),
                                                                      storage.info.__setitem__("training_step", storage.info["training_step"] + 1))
        try:
            sp.continuous_self_play(storage, storage, test_mode)
        finally:
            self_play.time.sleep = orig_sleep
        out[name] = numpy.array(json.dumps(log))
        print(f"   {name}: {len(log)} calls, {sum(1 for c in log if c[0] == 'play_game')} games")
    save("g15_self_play_loop", **out)


def g12_reference_speed(ctx):
    """How fast the reference builds training batches here (context for tools/replay_rate.py)."""
    import copy
    import replay_buffer
    self_play, cfgs = ctx["self_play"], ctx["configs"]
    config = copy.deepcopy(cfgs["cartpole"])
    rs = numpy.random.RandomState(5)
    games = [_synthetic_history(self_play, rs, config, int(n)) for n in rs.randint(50, 400, 200)]
    rb = replay_buffer.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    t0 = time.perf_counter()
    for gh in games:
        rb.save_game(gh)
    t_save = time.perf_counter() - t0
    positions = sum(len(g.root_values) for g in games)
    rb.get_batch()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        rb.get_batch()
    t_batch = (time.perf_counter() - t0) / reps
    print(f"   reference: save_game {positions / t_save:.0f} positions/s, get_batch {config.batch_size / t_batch:.0f} samples/s")
    save("g12_reference_speed", save_game_positions_per_s=positions / t_save,
         get_batch_samples_per_s=config.batch_size / t_batch, batch_size=config.batch_size,
         games=len(games), positions=positions)


def make_configs():
    import games.tictactoe as ttt
    import games.connect4 as c4
    import games.cartpole as cp
    import games.breakout as bo
    cfgs = {"tictactoe": ttt.MuZeroConfig(), "connect4": c4.MuZeroConfig(),
            "cartpole": cp.MuZeroConfig()}
    atari = bo.MuZeroConfig()
    # BASELINE.json config #5: "84x84x4 conv representation" = breakout.py's config with the
    # CNN down-sampler and a 4-frame 84x84 observation.
    atari.observation_shape = (4, 84, 84)
    atari.stacked_observations = 0
    atari.downsample = "CNN"
    atari.num_simulations = 50
    cfgs["atari84"] = atari
    return cfgs


ALL = [g0_weights, g1_support_to_scalar, g2_fc_inference, g3_resnet_inference, g4_cartpole,
       g5_tictactoe, g5_connect4, g5_atari84, g5_degenerate, g6_play_game, g6_connect4_opponents, g7_rng, g8_select_action,
       g9_stacked, g10_reference_speed, g11_envs, g12_replay_targets, g12_reference_speed, g13_reanalyse, g14_trainer, g15_self_play_loop, g16_trainer_resnet]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*")
    args = ap.parse_args()
    models, self_play = import_reference()
    ctx = {"models": models, "self_play": self_play, "configs": make_configs()}
    torch.set_num_threads(1)
    for fn in ALL:
        if args.only and fn.__name__ not in args.only:
            continue
        print(fn.__name__)
        fn(ctx)


if __name__ == "__main__":
    main()

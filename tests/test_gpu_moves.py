"""Batches of moves queued back to back (mzmcts_moves_*): every move an env plays inside a batch must be
bit-identical to the one-move-at-a-time path (fused search + host-side select_action on the mirror stream),
whatever happens to the speculation the batch rests on (tie-breaks, unknown sampling word counts, MT19937
block regenerations inside the pre-drawn stretch)."""
import importlib

import numpy as np
import pytest
import torch

from parity_helpers import cartpole_model_and_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(pkg):
    importlib.import_module("muzero-hypermodel_amd.build").build_native()
    return importlib.import_module("muzero-hypermodel_amd.engine")


def cartpole_setup(pkg, ties):
    models = importlib.import_module("muzero-hypermodel_amd.models")
    config = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
    model, _ = cartpole_model_and_weights(models, config, "cuda")
    if ties is True:   # all-equal priors below the root: the search breaks ties with the RNG all the time (fixture G5 recipe)
        with torch.no_grad():
            for name, prm in model.named_parameters():
                if name.startswith("prediction_policy_network.module.2"):
                    prm.zero_()
    elif ties == "some":
        # Priors tie at SOME nodes only: the two policy logits differ by exactly one hidden unit of the policy MLP, and that
        # unit is ELU(encoded_state[c]) -- zero exactly where component c is the state's minimum (the min-max rescale maps
        # it to 0.0).  A search then spends extra tie-break words at unpredictable moves, so an env leaves the pre-drawn path
        # at every position of a batch sooner or later: the host mirror's rewind (restore_stream / draw_env_rows,
        # csrc/mzmcts_moves.hip) is exercised at each of them.
        with torch.no_grad():
            prm = dict(model.named_parameters())
            w1, b1 = prm["prediction_policy_network.module.0.weight"], prm["prediction_policy_network.module.0.bias"]
            w2, b2 = prm["prediction_policy_network.module.2.weight"], prm["prediction_policy_network.module.2.bias"]
            # c: a component that is the minimum at some search nodes but not at most (counted over a few random walks)
            rs = np.random.RandomState(0)
            obs = torch.from_numpy(rs.uniform(-0.05, 0.05, (64, 4)).astype(np.float32)).to(w1.device)
            state = model.representation(obs)
            counts = np.zeros(state.shape[1])
            for _ in range(6):
                counts += np.bincount(state.argmin(1).cpu().numpy(), minlength=state.shape[1])
                action = torch.from_numpy(rs.randint(0, 2, (64, 1))).to(w1.device)
                state, _ = model.dynamics(state, action)
            share = counts / counts.sum()
            c = int(np.argmin(np.abs(share - 0.08)))
            assert 0.02 < share[c] < 0.5, share
            u = 3
            w1[u].zero_()
            w1[u, c] = 1.0
            b1[u] = 0.0
            w2[1].copy_(w2[0])
            b2[1] = b2[0]
            w2[0, u] += 1.0
    return config, model


@pytest.mark.parametrize("group,variant,ties,temps,batch,overlap", [
    (16, "narrow", False, "one", 6, False),
    (16, "narrow", False, "one", 6, True),     # next batch drawn while the current one runs
    (16, "narrow", True, "one", 4, False),     # ties at every level: most envs stall after one move per batch
    (16, "narrow", True, "one", 4, True),
    (16, "narrow", False, "mixed", 5, False),  # T = 0 / 1 / inf / 0.5 / 0.25 per env (inf: a single move per batch)
    (16, "narrow", False, "mixed", 5, True),
    (4, "generic", False, "one", 3, False),
    (4, "generic", True, "mixed", 3, True),
    (16, "narrow", "some", "one", 6, False),   # mis-speculation at every position of a batch (see cartpole_setup)
    (16, "narrow", "some", "one", 6, True),
    (16, "narrow", "some", "mixed", 5, True),
])
def test_move_batches_equal_one_move_at_a_time(eng, pkg, group, variant, ties, temps, batch, overlap):
    config, model = cartpole_setup(pkg, ties)
    E, N = 83, 40 if ties else 24            # N moves per env: > 624 RNG words per env when ties abound
    stalled_at = set()                       # batch positions at which some env left the pre-drawn path
    rs = np.random.RandomState(4)
    obs = torch.from_numpy(rs.uniform(-0.05, 0.05, (E, 4)).astype(np.float32)).cuda()   # same observation every move
    legal = [[0, 1] if e % 11 else [] for e in range(E)]                                # a few inactive envs
    to_play = [0] * E
    T = np.ones(E) if temps == "one" else np.array([[0.0, 1.0, np.inf, 0.5, 0.25][e % 5] for e in range(E)])
    seeds = [1000 + e for e in range(E)]

    ref = eng.BatchedMCTS(config, E, seeds=seeds, group_width=group)
    ref.configure_fused_fc(model)
    ref.set_fused_options(variant, publish_tree=False)
    want = []
    total = N + batch + 6
    if ties == "some":    # a different observation every move, so the tying nodes -- and the stalls -- move around
        every = torch.from_numpy(rs.uniform(-0.05, 0.05, (total, E, 4)).astype(np.float32)).cuda()
    else:
        every = obs.expand(total, E, 4)
    lanes = torch.arange(E, device="cuda")

    def obs_at(m):        # batch position m: every env sees the observation of ITS next move
        at = torch.tensor([min(len(g) + m, total - 1) for g in got], device="cuda")
        return every[at, lanes].contiguous()

    for i in range(total):
        st = ref.search_fused(every[i].contiguous(), legal, to_play, True)
        actions, _ = ref.sample_actions(T)
        want.append((actions.copy(), st["visits"].copy(), st["root_value_sum"].copy(),
                     st["root_predicted_value"].copy(), st["max_tree_depth"].copy()))
    ref.close()

    got = [[] for _ in range(E)]
    engine = eng.BatchedMCTS(config, E, seeds=seeds, group_width=group)
    engine.configure_fused_fc(model)
    engine.set_fused_options(variant, publish_tree=False)
    assert engine.fused_variant() == variant
    rounds = 0
    active = [e for e in range(E) if legal[e]]
    if overlap:
        engine.moves_prepare(batch, legal, to_play, T, True)
    while min(len(got[e]) for e in active) < N:
        if overlap:
            for m in range(batch):
                engine.moves_enqueue(obs_at(m))
            engine.moves_predraw_next(batch, legal, to_play, T, True)
            out = engine.moves_collect(copy=False)        # views of the pinned download ring
            engine.moves_submit_next()
        else:
            out = engine.run_moves([obs_at(m) for m in range(batch)], legal, to_play, T, True)
        rounds += 1
        assert rounds <= 2 * N + 4
        for e in range(E):
            k = out["moves_done"][e]
            assert (k >= 1) == bool(legal[e]) and (k == len(out["actions"]) or out["actions"][k, e] == -1)
            if np.isinf(T[e]) and legal[e]:
                assert k == 1
            elif legal[e] and k < len(out["actions"]):
                stalled_at.add(int(k))
            for m in range(k):
                got[e].append((out["actions"][m, e], out["visits"][m, e].copy(), out["root_value_sum"][m, e],
                               out["root_predicted"][m, e], out["max_depth"][m, e]))
    if overlap:
        engine.moves_collect()             # the batch submitted last: nothing enqueued, every draw is undone
        # a pre-drawn batch that is dropped must leave the streams where a fresh engine's would be
        engine.moves_prepare(2, legal, to_play, T, True)
        engine.moves_enqueue(obs_at(0)), engine.moves_enqueue(obs_at(1))
        engine.moves_predraw_next(batch, legal, to_play, T, True)
        a = engine.moves_collect()
        engine.moves_discard_next()
        for e in active:
            for m in range(a["moves_done"][e]):
                got[e].append((a["actions"][m, e], a["visits"][m, e].copy(), a["root_value_sum"][m, e],
                               a["root_predicted"][m, e], a["max_depth"][m, e]))
        b = engine.run_moves([obs_at(0), obs_at(1)], legal, to_play, T, True)
        for e in active:
            for o, m in [(b, m) for m in range(b["moves_done"][e])]:
                got[e].append((o["actions"][m, e], o["visits"][m, e].copy(), o["root_value_sum"][m, e],
                               o["root_predicted"][m, e], o["max_depth"][m, e]))
    engine.close()
    if ties == "some":
        # every position of the batch saw an env stall (its mirror rewound to the state after the last noise row used)
        assert stalled_at >= set(range(1, batch)), sorted(stalled_at)
    elif ties:
        assert rounds >= N // 2             # the tie-breaking searches really did stall the batches
    else:
        assert rounds <= (N // batch + 3) * (batch if temps == "mixed" else 1)
    for e in active:
        for i in range(min(len(got[e]), len(want))):
            a, v, rv, pred, depth = got[e][i]
            wa, wv, wrv, wpred, wdepth = want[i]
            assert a == wa[e] and np.array_equal(v, wv[e]), (e, i)
            assert rv == wrv[e] and np.float32(pred) == np.float32(wpred[e]) and depth == wdepth[e], (e, i)


def test_move_batch_actions_are_on_the_device(eng, pkg):
    config, model = cartpole_setup(pkg, False)
    E = 256
    obs = torch.from_numpy(np.random.RandomState(0).uniform(-0.05, 0.05, (E, 4)).astype(np.float32)).cuda()
    engine = eng.BatchedMCTS(config, E, group_width=16)
    engine.configure_fused_fc(model)
    engine.set_fused_options("auto", publish_tree=False)
    engine.moves_prepare(3, [[0, 1]] * E, [0] * E, 1.0)
    for _ in range(3):
        engine.moves_enqueue(obs)
    on_device = [engine.moves_actions(m) for m in range(3)]
    out = engine.moves_collect()
    for m in range(3):
        assert np.array_equal(on_device[m].cpu().numpy(), out["actions"][m])
    with pytest.raises(RuntimeError):
        engine.moves_collect()
    with pytest.raises(RuntimeError, match="temperature 0, inf or 1/k"):
        engine.moves_prepare(2, [[0, 1]] * E, [0] * E, 0.3)
    engine.close()


def test_move_batches_without_exploration_noise(eng, pkg):
    """add_exploration_noise=False (the reference's test mode): no Dirichlet rows, the streams only advance by
    the tie-break and sampling words -- batches still equal the one-move-at-a-time path."""
    config, model = cartpole_setup(pkg, False)
    E, N = 40, 9
    obs = torch.from_numpy(np.random.RandomState(8).uniform(-0.05, 0.05, (E, 4)).astype(np.float32)).cuda()
    legal, to_play, T = [[0, 1]] * E, [0] * E, np.where(np.arange(E) % 2, 1.0, 0.0)
    seeds = [77 + e for e in range(E)]
    ref = eng.BatchedMCTS(config, E, seeds=seeds, group_width=16)
    ref.configure_fused_fc(model)
    want = []
    for _ in range(N):
        st = ref.search_fused(obs, legal, to_play, False)
        actions, _ = ref.sample_actions(T)
        want.append((actions.copy(), st["visits"].copy(), st["root_value_sum"].copy()))
    ref.close()
    engine = eng.BatchedMCTS(config, E, seeds=seeds, group_width=16)
    engine.configure_fused_fc(model)
    got = [[] for _ in range(E)]
    while min(len(g) for g in got) < N:
        out = engine.run_moves([obs] * 3, legal, to_play, T, False)
        for e in range(E):
            for m in range(out["moves_done"][e]):
                got[e].append((out["actions"][m, e], out["visits"][m, e].copy(), out["root_value_sum"][m, e]))
    engine.close()
    for e in range(E):
        for i in range(N):
            assert got[e][i][0] == want[i][0][e] and np.array_equal(got[e][i][1], want[i][1][e]), (e, i)
            assert got[e][i][2] == want[i][2][e]


def test_device_input_move_batches_equal_one_move_at_a_time_tictactoe(eng, pkg):
    """mzmcts_moves_prepare_device: a batch of moves on a game whose legal action set changes with every move
    (TicTacToe, fully-connected network, device-resident envs) -- legal sets and players to move read from the
    environment kernels' device outputs, exploration noise drawn on the device -- plays, move for move and env for env,
    what the one-move-at-a-time path plays with the host drawing the noise: legal sets, actions, visit counts, root
    value sums, and the RNG streams afterwards (games end and restart inside the batch)."""
    import importlib
    import torch
    models_mod = importlib.import_module("muzero-hypermodel_amd.models")
    device_mod = importlib.import_module("muzero-hypermodel_amd.games.device")
    config = importlib.import_module("muzero-hypermodel_amd.games.tictactoe").MuZeroConfig()
    config.network, config.encoding_size = "fullyconnected", 8
    config.fc_representation_layers, config.fc_dynamics_layers = [], [16]
    config.fc_reward_layers = config.fc_value_layers = config.fc_policy_layers = [16]
    config.num_simulations = 20
    torch.manual_seed(4)
    model = models_mod.MuZeroNetwork(config).to("cuda").eval()
    E, M = 96, 12
    seeds = list(range(300, 300 + E))
    A = len(config.action_space)

    # ---- one move at a time: host-drawn noise, host-sampled actions
    envs = device_mod.DeviceEnvs("tictactoe", E, seeds=seeds, device="cuda")
    single = eng.BatchedMCTS(config, E, seeds=seeds, group_width=16)
    single.configure_fused_fc(model)
    want = []
    for _ in range(M):
        obs, legal, num_legal, to_play = envs.observe()
        legal_h, nl_h, tp_h = legal.cpu().numpy(), num_legal.cpu().numpy(), to_play.cpu().numpy()
        st = single.search(model, obs.reshape(E, -1), legal_h, tp_h, True, num_legal=nl_h)
        actions, _ = single.sample_actions(1.0)
        want.append(dict(legal=legal_h.copy(), num_legal=nl_h.copy(), to_play=tp_h.copy(), actions=actions.copy(),
                         visits=st["visits"].copy(), root_value_sum=st["root_value_sum"].copy()))
        _, done = envs.step(actions)
        if bool(done.any()):
            envs.reset(done.clone())
    want_rng = [single.get_rng_state(e) for e in range(E)]
    single.close()
    envs.close()

    # ---- the whole batch on the device
    envs = device_mod.DeviceEnvs("tictactoe", E, seeds=seeds, device="cuda")
    batch = eng.BatchedMCTS(config, E, seeds=seeds, group_width=16)
    batch.configure_fused_fc(model)
    obs, legal, num_legal, to_play = envs.observe()
    batch.moves_prepare_device(M, legal, num_legal, to_play, 1.0, True)
    shape = envs.observation_shape
    reward = torch.zeros((M, E), dtype=torch.float32, device="cuda")
    done = torch.zeros((M, E), dtype=torch.uint8, device="cuda")
    obs_after = torch.zeros((M, E) + shape, dtype=torch.float32, device="cuda")
    obs_next = torch.zeros((M, E) + shape, dtype=torch.float32, device="cuda")
    obs_in = obs
    for m in range(M):
        batch.moves_enqueue(obs_in.reshape(E, -1).contiguous())
        obs_in = envs.advance(batch.moves_actions(m), reward[m], done[m], obs_after[m], obs_next[m])
    got = batch.moves_collect()
    inputs = batch.moves_inputs(M)
    assert (got["moves_done"] == M).all()
    assert int(done.sum()) > E                                  # games ended (and restarted) inside the batch
    for m in range(M):
        w = want[m]
        assert np.array_equal(inputs["num_legal"][m], w["num_legal"]), m
        assert np.array_equal(inputs["to_play"][m], w["to_play"]), m
        for e in range(E):
            n = int(w["num_legal"][e])
            assert np.array_equal(inputs["legal"][m, e, :n], w["legal"][e, :n]), (m, e)
        assert np.array_equal(got["actions"][m], w["actions"]), m
        assert np.array_equal(got["visits"][m], w["visits"]), m
        assert np.array_equal(got["root_value_sum"][m], w["root_value_sum"]), m
    got_rng = [batch.get_rng_state(e) for e in range(E)]
    for a, b in zip(want_rng, got_rng):
        assert np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    batch.close()
    envs.close()

"""A weight refresh must reach every network replica of every rank, also under hipGraph replay.

reference self_play.py:37 (`set_weights(get_info("weights"))` before every game) <-> trainer.py:87-95.  Here the
weights arrive in one flat buffer (weights.FlatWeights: a broadcast target) aliased by replica 0; pipelined actors keep
a replica per env group, and a replayed hipGraph reads cached constants (folded batch norms, packed / split tower
weights) that only Python code rebuilds.  Round 2's bug: replicas >= 1 got a bare load_state_dict."""
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from parity_helpers import synthetic_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng(pkg):
    importlib.import_module("muzero-hypermodel_amd.build").build_native()
    return importlib.import_module("muzero-hypermodel_amd.engine")


def _positions(game, E, rs):
    if game == "tictactoe":
        obs = rs.randint(-1, 2, (E, 3, 3, 3)).astype(np.float32)
        A = 9
    else:
        obs = rs.randint(-1, 2, (E, 3, 6, 7)).astype(np.float32)
        A = 7
    legal = np.zeros((E, A), np.int32)
    num_legal = rs.randint(2, A + 1, E).astype(np.int32)
    for e in range(E):
        legal[e, :num_legal[e]] = np.sort(rs.permutation(A)[:num_legal[e]])
    return torch.from_numpy(obs).cuda(), legal, num_legal, rs.randint(0, 2, E).astype(np.int32)


@pytest.mark.parametrize("game", ["tictactoe", "connect4"])
def test_pipelined_refresh_reaches_every_replica_under_graph_replay(eng, game):
    """PipelinedLockstep(groups=2, use_graph=True): search, replay, new weights into replica 0's flat buffer +
    pipe.refresh(), search again -- visits and value sums of BOTH groups equal an eager single engine on a network built
    from the new weights (TicTacToe: exact-fp32 tower; Connect4: split-precision tower with its packed fp16 halves)."""
    models_mod = importlib.import_module("muzero-hypermodel_amd.models")
    weights_mod = importlib.import_module("muzero-hypermodel_amd.weights")
    config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
    if game == "connect4":
        config.num_simulations = 60
    E = 32
    obs, legal, num_legal, to_play = _positions(game, E, np.random.RandomState(3))
    seeds = list(range(900, 900 + E))
    model, _ = synthetic_model(models_mod, config, "cuda", seed=0)
    _, new_weights = synthetic_model(models_mod, config, "cpu", seed=1)
    flat = weights_mod.FlatWeights(model)
    pipe = eng.PipelinedLockstep(config, E, model, groups=2, seeds=seeds, use_graph=True)

    def pipe_move():
        for g in range(2):
            sl = pipe.slice(g)
            pipe.begin(g, obs[sl].contiguous(), legal[sl], to_play[sl], True, num_legal=num_legal[sl])
        parts = [{k: v.copy() for k, v in pipe.finish(g).items()} for g in range(2)]
        return {k: np.concatenate([p[k] for p in parts]) for k in ("visits", "root_value_sum", "child_value_sum")}

    before = [pipe_move() for _ in range(3)]          # eager, capturing, replay
    assert all(e._graph is not None for e in pipe.engines)
    flat.load_state_dict(new_weights)                 # what a broadcast into the flat buffer leaves behind, + refold of replica 0
    pipe.refresh()
    after = [pipe_move() for _ in range(2)]           # replays of the graphs captured with the old weights
    pipe.close()

    # the yardstick: per group an eager engine of the group's size and seeds (same launch shapes: which head / tower
    # kernel variant runs depends on the batch size, and their fp32 sums differ in the last bits) on a fresh network
    fresh, _ = synthetic_model(models_mod, config, "cuda", seed=1)
    assert not np.array_equal(before[2]["root_value_sum"], after[0]["root_value_sum"])
    for g in range(2):                                # replica 0 was always right, replica 1 is round 2's bug
        sl = pipe.slice(g)
        single = eng.BatchedMCTS(config, E // 2, seeds=seeds[sl], use_graph=False)
        want = []
        for i in range(5):
            st = single.search(fresh, obs[sl].contiguous(), legal[sl], to_play[sl], True, num_legal=num_legal[sl])
            want.append({k: st[k].copy() for k in ("visits", "root_value_sum", "child_value_sum")})
        single.close()
        for got, ref in zip(after, want[3:]):
            for key in ("visits", "root_value_sum", "child_value_sum"):
                assert np.array_equal(got[key][sl], ref[key]), (game, g, key)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("game", ["tictactoe", "connect4"])
def test_pull_weights_reaches_every_replica_of_every_rank(tmp_path, game):
    """Two ranks (sharing this box's one GPU, gloo standing in for RCCL) x two pipelined groups each: after
    ManyEnvLoop's distributed weight pull every replica of every rank plays what an eager single actor with the new
    weights plays (tests/workers/refresh_rank.py)."""
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    worker = os.path.join(ROOT, "tests", "workers", "refresh_rank.py")
    outs = [str(tmp_path / f"rank{r}.json") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", port, game, outs[r]], env=env, cwd=ROOT,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=420)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{logs[r][-3000:]}"
    res = [json.load(open(o)) for o in outs]
    assert res[0]["flat_sum"] == res[1]["flat_sum"]                 # rank 1 holds rank 0's weights
    for r in res:
        assert r["differs_from_old_weights"], r["rank"]             # (the new weights do change the games)
        assert r["games"] > 0 or game == "connect4"
        assert r["equal"], (r["rank"], r["got"]["games"][:2], r["want"]["games"][:2])

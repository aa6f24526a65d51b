"""The N>1 path on CPU: world_size-2 gloo process group, env sharding and the flat-weight broadcast
(the same code path runs over RCCL/xGMI with backend "nccl" on the GPU node)."""
import importlib
import os
import socket
import sys

import numpy
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    actor = importlib.import_module("muzero-hypermodel_amd.actor")
    weights_mod = importlib.import_module("muzero-hypermodel_amd.weights")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    ttt = importlib.import_module("muzero-hypermodel_amd.games.tictactoe")
    r, w, lr = actor.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)                       # ranks start with DIFFERENT weights
    model = models.MuZeroNetwork(ttt.MuZeroConfig())
    model.eval()
    flat = weights_mod.FlatWeights(model)
    for k in flat.float_keys:                           # make BN statistics rank-specific too
        if "running" in k:
            dict(model.named_buffers())[k].add_(rank + 1.0)
    before = flat.flat.clone()
    flat.broadcast(src=0)
    torch.save({"before": before, "after": flat.flat.clone(), "sd": flat.state_dict(),
                "numel": flat.numel, "seeds": actor.shard_seeds(7, rank, 4),
                "out": model.initial_inference(torch.ones(2, 3, 3, 3))[0]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_flat_weight_broadcast_and_sharding_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    assert r0["numel"] == r1["numel"] == 21971          # 21 715 parameters + 256 BN running statistics (fp32)
    assert not torch.equal(r0["before"], r1["before"])
    assert torch.equal(r0["after"], r0["before"])        # source rank unchanged
    assert torch.equal(r1["after"], r0["before"])        # every float of rank 1 replaced
    assert torch.equal(r0["out"], r1["out"])             # the model itself now computes with them
    for k, v in r0["sd"].items():
        assert torch.equal(v, r1["sd"][k]), k
    assert r0["seeds"] == [7, 8, 9, 10] and r1["seeds"] == [11, 12, 13, 14]


def test_flat_weights_single_process_roundtrip():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity_helpers import load_golden
    weights_mod = importlib.import_module("muzero-hypermodel_amd.weights")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    cp = importlib.import_module("muzero-hypermodel_amd.games.cartpole")
    model = models.MuZeroNetwork(cp.MuZeroConfig())
    flat = weights_mod.FlatWeights(model)
    assert flat.numel == 1532 and flat.nbytes() == 6128
    w = load_golden("cartpole_weights")
    flat.load_state_dict({k: w[k] for k in w.files})
    for k, v in model.state_dict().items():
        assert numpy.array_equal(v.numpy(), w[k])        # parameters alias the flat buffer
    assert flat.broadcast() is None                      # no process group: a no-op
    model.set_weights({k: torch.zeros_like(v) for k, v in model.state_dict().items()})
    assert float(flat.flat.abs().sum()) == 0.0           # load_state_dict writes through the views


def _loop_worker(rank, world, port, out_dir):
    """One many-env actor per rank running ManyEnvLoop.continuous_self_play together (self_play.py:31-108 for E envs,
    SURVEY.md section 8e): rank 0 alone talks to the shared storage; the loop condition and the weights travel by
    broadcast.  The search itself is replaced by canned games (no GPU here)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    actor_mod = importlib.import_module("muzero-hypermodel_amd.actor")
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    ttt = importlib.import_module("muzero-hypermodel_amd.games.tictactoe")
    actor_mod.init_distributed(backend="gloo")
    config = ttt.MuZeroConfig()
    config.training_steps, config.ratio, config.self_play_delay, config.temperature_threshold = 6, None, 0, None
    torch.manual_seed(5 + rank)                          # ranks start with different weights
    model = models.MuZeroNetwork(config)
    model.eval()
    log = []

    class Storage:                                       # only rank 0 may touch it
        info = {"training_step": 0, "terminate": False, "num_played_steps": 0}

        def get_info(self, key):
            assert rank == 0
            log.append(key)
            if key == "weights":
                return {k: v + float(self.info["training_step"]) if v.dtype == torch.float32 else v
                        for k, v in fresh.items()}
            return self.info[key]

    class Replay:
        def __init__(self):
            self.saved = []

        def save_game(self, gh, shared_storage=None):
            self.saved.append(gh.weights_version)
            if rank == 0:
                Storage.info["training_step"] += 1

    fresh = {k: v.clone() for k, v in models.MuZeroNetwork(config).state_dict().items()}

    class Actor(sp.ManyEnvLoop):
        E = 2
        device = torch.device("cpu")

        def __init__(self):
            self.config, self.model, self.passes = config, model, 0

        def set_weights(self, w):
            raise AssertionError("several ranks: weights arrive by broadcast")

        def _play_pass(self, temperature, threshold, moves_per_pass):
            assert moves_per_pass == 3
            self.passes += 1
            gh = sp.GameHistory()
            gh.action_history, gh.reward_history, gh.root_values = [0, 1], [0, 1.0], [0.5]
            return [(self.passes % 2, gh)]

        def close(self):
            pass
    actor = Actor()
    replay = Replay()
    actor.continuous_self_play(Storage() if rank == 0 else None, replay, False, moves_per_pass=3)
    flat = actor._loop_state()["flat"].flat
    torch.save({"passes": actor.passes, "saved": replay.saved, "checksum": float(flat.double().sum()), "log": log},
               os.path.join(out_dir, f"loop{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_many_env_loop_world2(tmp_path):
    world = 2
    mp.spawn(_loop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "loop0.pt")
    r1 = torch.load(tmp_path / "loop1.pt")
    assert r0["passes"] == r1["passes"] == 6             # both ranks made the same passes until training_steps = 6
    assert r0["checksum"] == r1["checksum"]              # rank 1 holds rank 0's last pulled weights
    assert r1["log"] == []                               # rank 1 never touched the storage
    assert r0["log"].count("weights") == 6 and r0["log"][:3] == ["training_step", "terminate", "weights"]
    assert [v[1] for v in r0["saved"]] == [0, 1, 2, 3, 4, 5] == [v[1] for v in r1["saved"]]   # weight versions agree

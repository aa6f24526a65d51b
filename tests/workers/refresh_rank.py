"""Rank worker of tests/test_gpu_refresh.py::test_pull_weights_reaches_every_replica_of_every_rank.

usage: refresh_rank.py RANK WORLD PORT GAME OUT.json

Every rank: a PipelinedDeviceSelfPlay of two groups (hipGraph-captured simulation loops, a network replica per group)
plays passes with weights A, pulls weights B the way ManyEnvLoop.continuous_self_play does with several ranks (rank 0
reads the shared storage, one broadcast of the flat buffer -- gloo here, RCCL on a multi-GPU node), and plays on.  The
same envs played by an eager (no hipGraph) actor of the same shape that is handed B directly through set_weights must finish the same games and end with the
same search statistics.  (reference self_play.py:37: every game after the pull sees the pulled weights.)"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    rank, world, port, game, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    import numpy as np
    import torch
    from parity_helpers import synthetic_model
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    models_mod = importlib.import_module("muzero-hypermodel_amd.models")
    actor_mod = importlib.import_module("muzero-hypermodel_amd.actor")
    config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
    if game == "connect4":
        config.num_simulations = 40            # (the split-precision tower is what is under test, not the search length)
    torch.cuda.set_device(0)
    actor_mod.init_distributed(backend="gloo")
    _, w_a = synthetic_model(models_mod, config, "cpu", seed=0)
    _, w_b = synthetic_model(models_mod, config, "cpu", seed=1)
    E, before, after = 16, 2, (8 if game == "connect4" else 3)
    seed = 100 + rank * E

    class Storage:                               # only rank 0 holds one
        def get_info(self, key):
            assert rank == 0 and key == "weights"
            return w_b

    def summary(finished, engines):
        games = sorted((int(e), [int(a) for a in gh.action_history], [float(v) for v in gh.root_values])
                       for e, gh in finished)
        visits = np.concatenate([eng.stats["visits"] for eng in engines]).tolist()
        values = np.concatenate([eng.stats["root_value_sum"] for eng in engines]).tolist()
        return dict(games=games, visits=visits, values=values)

    pipe = sp.PipelinedDeviceSelfPlay({"weights": w_a}, game, config, seed, E, groups=2, use_graph=True)
    finished = []
    for _ in range(before):
        finished += pipe._play_pass(1.0, None, 3)
    assert all(a.engine._graph is not None for a in pipe.actors)            # the loops are replayed from here on
    pipe._pull_weights(Storage() if rank == 0 else None, 1)
    for _ in range(after):
        finished += pipe._play_pass(1.0, None, 3)
    got = summary(finished, [a.engine for a in pipe.actors])
    flat_sum = float(pipe._loop_state()["flat"].flat.double().sum())
    pipe.close()

    single = sp.PipelinedDeviceSelfPlay({"weights": w_a}, game, config, seed, E, groups=2, use_graph=False)
    finished = []
    for _ in range(before):
        finished += single._play_pass(1.0, None, 3)
    single.set_weights(w_b)
    for _ in range(after):
        finished += single._play_pass(1.0, None, 3)
    want = summary(finished, [a.engine for a in single.actors])
    single.close()

    # what the bug looked like: the same run with the old weights kept
    old = sp.PipelinedDeviceSelfPlay({"weights": w_a}, game, config, seed, E, groups=2, use_graph=False)
    finished = []
    for _ in range(before + after):
        finished += old._play_pass(1.0, None, 3)
    kept = summary(finished, [a.engine for a in old.actors])
    old.close()
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(dict(rank=rank, equal=got == want, differs_from_old_weights=got != kept, games=len(got["games"]),
                       flat_sum=flat_sum, got=got, want=want), f)


if __name__ == "__main__":
    main()

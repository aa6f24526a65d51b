"""Whole self-play loop rate (search + env step + history filing), host-plugin envs vs device envs.

    python tools/selfplay_rate.py [--game cartpole] [--envs 4096] [--moves 60]

Prints one JSON line per actor kind: moves/s, simulations/s, finished games.  This is the loop the
reference runs in self_play.py:34-113 (continuous_self_play), without the replay buffer hand-off."""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

sp = importlib.import_module("muzero-hypermodel_amd.self_play")
models = importlib.import_module("muzero-hypermodel_amd.models")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="cartpole")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--moves", type=int, default=60)
    ap.add_argument("--kinds", default="host,device")
    ap.add_argument("--batch", type=int, default=20, help="moves per host round trip of the device-batch actor")
    ap.add_argument("--weights", default="random", choices=["random", "checkpoint"],
                    help="checkpoint: the reference's trained CartPole weights (tests/golden/cartpole_weights.npz): long "
                         "games and the benchmark's tree depths; random: seed-0 initialisation, games of ~15 moves")
    ap.add_argument("--fc", type=int, default=0, help="N > 0: play the game with a fully-connected network (the reference's "
                    "network = 'fullyconnected'), encoding and layers of N units -- board games through the fused whole-move search")
    ap.add_argument("--no-prefetch", action="store_true", help="device-pipelined-batch: every play_moves call drains the GPU "
                    "(what a loop does that pulls weights between calls)")
    args = ap.parse_args()
    mod = importlib.import_module(f"muzero-hypermodel_amd.games.{args.game}")
    config = mod.MuZeroConfig()
    if args.fc:
        config.network, config.encoding_size = "fullyconnected", args.fc
        config.fc_representation_layers, config.fc_dynamics_layers = [], [args.fc]
        config.fc_reward_layers = config.fc_value_layers = config.fc_policy_layers = [args.fc]
        config.temperature_threshold = None
    torch.manual_seed(0)
    weights = models.MuZeroNetwork(config).get_weights()
    if args.weights == "checkpoint":
        import numpy
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        fixture = numpy.load(os.path.join(root, "tests", "golden", "cartpole_weights.npz"))
        weights = {k: torch.from_numpy(fixture[k]) for k in fixture.files}
    for kind in args.kinds.split(","):
        if kind == "host":
            actor = sp.BatchedSelfPlay({"weights": weights}, mod.Game, config, 0, args.envs)
        elif kind in ("device-pipelined", "device-pipelined-batch"):
            actor = sp.PipelinedDeviceSelfPlay({"weights": weights}, args.game, config, 0, args.envs, groups=2)
        else:
            actor = sp.DeviceSelfPlay({"weights": weights}, args.game, config, 0, args.envs)
            if kind == "device-batch" and actor.engine._fc_model is not None:
                actor.engine.set_fused_options("auto", publish_tree=False)
        done = [0]

        def on_game(e, gh):
            done[0] += 1

        def on_games(batch):
            done[0] += len(batch)

        cb = dict(on_game=on_game) if kind in ("host", "device-lists") else dict(on_games=on_games)
        if kind in ("device-batch", "device-pipelined-batch"):
            # searches, env steps and resets queued back to back, `--batch` moves per host round trip
            for _ in range(3):                       # buffers, the native history filer, the pre-drawn next batch
                actor.play_moves(args.batch, 1.0, **cb)
            torch.cuda.synchronize()
            done[0] = 0
            t0 = time.perf_counter()
            moves = 0
            calls = max(1, args.moves // args.batch)
            for i in range(calls):
                # (two groups: each group's next batch is queued before its last one is filed; the last call drains)
                ahead = dict(prefetch=i + 1 < calls and not args.no_prefetch) if kind == "device-pipelined-batch" else {}
                moves += int(actor.play_moves(args.batch, 1.0, **cb, **ahead).sum())
            actor.flush(**cb)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        else:
            for _ in range(5):
                actor.step(1.0, None, **cb)
            torch.cuda.synchronize()
            done[0] = 0
            t0 = time.perf_counter()
            for _ in range(args.moves):
                actor.step(1.0, None, **cb)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            moves = args.moves * args.envs
        print(json.dumps({"actor": kind, "game": args.game, "envs": args.envs, "moves_per_s": moves / dt,
                          "simulations_per_s": moves * config.num_simulations / dt, "ms_per_move_step": 1e3 * dt * args.envs / moves,
                          "games_finished": done[0], "weights": args.weights,
                          **({"moves_per_call": args.batch, "prefetch": not args.no_prefetch} if kind == "device-pipelined-batch" else {}),
                          **({"moves_per_call": args.batch} if kind == "device-batch" else {})}), flush=True)
        actor.close()


if __name__ == "__main__":
    main()

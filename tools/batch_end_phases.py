"""Host time of the end of a device-input move batch, phase by phase, with the GPU already idle (so no phase's time is a
wait for kernels): what a single-group DeviceSelfPlay.play_moves exposes per batch.

    python tools/batch_end_phases.py [game=tictactoe] [envs=65536] [moves=20]"""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, torch
sp = importlib.import_module("muzero-hypermodel_amd.self_play")
models = importlib.import_module("muzero-hypermodel_amd.models")
game = sys.argv[1] if len(sys.argv) > 1 else "tictactoe"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
M = int(sys.argv[3]) if len(sys.argv) > 3 else 20
config = importlib.import_module(f"muzero-hypermodel_amd.games.{game}").MuZeroConfig()
torch.manual_seed(0)
actor = sp.DeviceSelfPlay({"weights": models.MuZeroNetwork(config).get_weights()}, game, config, 0, E)
games = [0]
cb = dict(on_games=lambda b: games.__setitem__(0, games[0] + len(b)))
for _ in range(3):
    actor.play_moves(M, 1.0, **cb)
phases = {}
def timed(name, fn):
    t = time.perf_counter(); r = fn(); phases[name] = phases.get(name, 0.0) + time.perf_counter() - t; return r
reps = 4
for _ in range(reps):
    t = time.perf_counter()
    actor._device_batch_begin(M, 1.0, None, cb["on_games"], config.temperature_threshold)
    phases["begin"] = phases.get("begin", 0.0) + time.perf_counter() - t
    t = time.perf_counter()
    for m in range(M):
        actor._device_batch_move(m)
    phases["enqueue moves (host)"] = phases.get("enqueue moves (host)", 0.0) + time.perf_counter() - t
    timed("gpu wait", torch.cuda.synchronize)
    # _device_batch_end, phase by phase
    b, eng, envs = actor._dev_batch, actor.engine, actor.envs
    actor._dev_batch = None
    timed("flush previous batch", lambda: actor.flush(None, cb["on_games"]))
    out = timed("moves_collect", lambda: eng.moves_collect(copy=False))
    inputs = timed("moves_inputs", lambda: eng.moves_inputs(M, copy=False))
    last_to_play = timed("to_play.cpu", lambda: envs.to_play.cpu().numpy())
    timed("copy stream sync", actor._copy_stream.synchronize)
    host = {k: b["pinned"][k][:M].numpy() for k in ("reward", "done", "obs_after", "obs_next")}
    to_play = inputs["to_play"]
    to_play_after = timed("to_play arrays", lambda: (1 - to_play) if len(config.players) > 1 else numpy.zeros_like(to_play))
    to_play_next = numpy.concatenate([to_play[1:], last_to_play[None]], axis=0)
    actor._unfiled = (out, host, inputs["legal"], inputs["num_legal"], M, to_play_after, to_play_next)
    actor._cur = dict(obs_dev=b["obs_in"], on_device_only=True)
actor.flush(**cb)
print(json.dumps({"game": game, "envs": E, "moves_per_batch": M, "ms_per_batch": {k: round(1e3 * v / reps, 2) for k, v in phases.items()}}))
actor.close()

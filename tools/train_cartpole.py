#!/usr/bin/env python3
"""End-to-end check of the widened path on one MI355X: device self-play (move batches on device envs) ->
device replay store -> Trainer -> weights published back into the actor's flat buffer, CartPole, from random
weights.  Not a benchmark: it shows that the pieces learn together (the reference's README curve reaches
~420 reward after ~2000 training steps with one worker).

    python tools/train_cartpole.py [--envs 64] [--iterations 30] [--moves 8] [--train-steps 100]
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=64)
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--moves", type=int, default=8, help="self-play moves per env and iteration (one move batch)")
    ap.add_argument("--train-steps", type=int, default=100, help="training steps per iteration")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--eager-trainer", action="store_true", help="launch the training step op by op instead of as one hipGraph replay")
    args = ap.parse_args()
    sp = importlib.import_module("muzero-hypermodel_amd.self_play")
    rb_mod = importlib.import_module("muzero-hypermodel_amd.replay_buffer")
    tr_mod = importlib.import_module("muzero-hypermodel_amd.trainer")
    models = importlib.import_module("muzero-hypermodel_amd.models")
    config = importlib.import_module("muzero-hypermodel_amd.games.cartpole").MuZeroConfig()
    config.training_steps = args.iterations * args.train_steps
    config.seed = args.seed
    torch.manual_seed(args.seed)
    weights = models.MuZeroNetwork(config).get_weights()
    actor = sp.DeviceSelfPlay({"weights": weights}, "cartpole", config, args.seed, args.envs)
    actor.engine.set_fused_options("auto", publish_tree=False)
    replay = rb_mod.ReplayBuffer({"num_played_games": 0, "num_played_steps": 0}, {}, config)
    trainer = tr_mod.Trainer({"weights": weights, "training_step": 0, "optimizer_state": None}, config, device="cuda",
                                graph=not args.eager_trainer)   # the step as one hipGraph replay
    flat = actor.engine._fc_flat                     # the buffer the actor's network (and the fused kernel) alias
    finished = []

    def on_games(batch):
        replay.save_games(batch)
        finished.extend(batch.rewards[i, 1: n + 1].sum() for i, n in enumerate(batch.length))

    t0 = time.perf_counter()
    log = []
    for it in range(args.iterations):
        temperature = config.visit_softmax_temperature_fn(trainer.training_step)
        actor.play_moves(args.moves, temperature, on_games=on_games)
        losses = None
        if replay.num_played_games > 0:
            for _ in range(args.train_steps):
                index_batch, batch = replay.get_batch()
                trainer.update_lr()
                priorities, *losses = trainer.update_weights(batch)
                if config.PER:
                    replay.update_priorities(priorities, index_batch)
            trainer.publish(flat)                    # fresh weights for the next batch of searches
        recent = finished[-50:]
        row = dict(iteration=it, training_step=trainer.training_step, played_steps=int(actor.moves_played),
                   games=len(finished), mean_reward_last_50=float(np.mean(recent)) if recent else None,
                   max_reward=float(np.max(finished)) if finished else None,
                   mean_length_of_running_games=float(np.mean(actor._len)), temperature=temperature,
                   total_loss=losses[0] if losses else None, seconds=time.perf_counter() - t0)
        log.append(row)
        print(json.dumps(row), flush=True)
    actor.flush(on_games=on_games)
    actor.close()
    replay.close()


if __name__ == "__main__":
    main()

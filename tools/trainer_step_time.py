#!/usr/bin/env python3
"""Time of Trainer.update_weights on the MI355X with the loss as one HIP launch (csrc/trainer_kernels.hip) and as the
torch expression: CartPole FC config and TicTacToe residual network, synthetic replay batches resident on the GPU.
One JSON line per (config, loss path)."""
import importlib, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
pkg = lambda m: importlib.import_module("muzero-hypermodel_amd." + m)

for game in ("cartpole", "tictactoe"):
    config = pkg("games." + game).MuZeroConfig()
    config.train_on_gpu = True
    model = pkg("models").MuZeroNetwork(config)
    ckpt = {"weights": model.get_weights(), "training_step": 0, "optimizer_state": None}
    B, K1, A = config.batch_size, config.num_unroll_steps + 1, len(config.action_space)
    g = torch.Generator(device="cuda").manual_seed(0)
    batch = (torch.rand((B,) + tuple(config.observation_shape), generator=g, device="cuda"),
             torch.randint(0, A, (B, K1), generator=g, device="cuda"),
             torch.randn(B, K1, generator=g, device="cuda") * 10, torch.randn(B, K1, generator=g, device="cuda"),
             torch.softmax(torch.randn(B, K1, A, generator=g, device="cuda"), dim=2),
             torch.rand(B, generator=g, device="cuda") + 0.5,
             torch.randint(1, K1 + 1, (B, K1), generator=g, device="cuda").float())
    for native, graph in ((True, True), (True, False), (False, False)):
        trainer = pkg("trainer").Trainer(ckpt, config, device="cuda", graph=graph)
        trainer.native_loss = native
        for _ in range(10):
            trainer.update_weights(batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 100
        for _ in range(n):
            trainer.update_weights(batch)
        torch.cuda.synchronize()
        print(json.dumps({"config": game, "batch": B, "unrolled_positions": K1, "loss": "one HIP launch" if native else "torch expression", "hipgraph": graph,
                          "ms_per_training_step": round((time.perf_counter() - t0) / n * 1e3, 3)}), flush=True)

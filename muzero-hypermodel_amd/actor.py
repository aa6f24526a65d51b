"""One self-play actor per GPU, launched with torch.distributed (no Ray).

    python -m torch.distributed.run --nproc-per-node N ... your_script.py

Rank g drives GPU g and owns the global envs [g*E, (g+1)*E): env i uses `Game(seed + i)` and the RNG
stream of reference worker `seed + i` (muzero.py:170-178).  Actors never talk to each other during a
search; the only exchange is the weight refresh (weights.FlatWeights.broadcast, RCCL over xGMI).
"""
import os

import torch
import torch.distributed as dist

from . import models
from .engine import BatchedMCTS
from .weights import FlatWeights


def init_distributed(backend=None):
    """Join the process group described by torchrun's env vars; returns (rank, world, local_rank).
    A plain `python script.py` run gives (0, 1, 0) without touching torch.distributed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MZ_REHEARSE_ON_ONE_GPU"):
        # rehearsal of the N>1 code path on a one-GPU box: every rank shares cuda:0 and the process group
        # runs over gloo (RCCL refuses two ranks on one device).  Never used for measurements.
        backend, local_rank = "gloo", 0
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def env_shard(rank, envs_per_rank):
    """Global env indices owned by `rank`."""
    return range(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def shard_seeds(base_seed, rank, envs_per_rank):
    return [int(base_seed) + i for i in env_shard(rank, envs_per_rank)]


class SearchActor:
    """Model replica + BatchedMCTS engine for one GPU's shard of envs (search only: envs are supplied
    by the caller as observation / legal-action batches, e.g. bench.py's synthetic rollouts)."""

    def __init__(self, config, weights, envs_per_rank, rank=0, device=None, use_graph=True, group_width=0,
                 fused_fc=False, device_noise=False):
        self.config = config
        self.rank = rank
        self.device = torch.device(device if device is not None else "cuda")
        self.model = models.MuZeroNetwork(config)
        self.model.set_weights(weights)
        self.model.to(self.device)
        self.model.eval()
        self.flat = FlatWeights(self.model)
        self.engine = BatchedMCTS(config, envs_per_rank, device=self.device,
                                  seeds=shard_seeds(config.seed, rank, envs_per_rank), use_graph=use_graph,
                                  group_width=group_width)
        if fused_fc:
            # the fused kernel reads the same flat buffer the RCCL broadcast lands in
            self.engine.configure_fused_fc(self.model, self.flat)
        if device_noise:
            self.engine.use_device_noise()   # Dirichlet rows drawn on the GPU: the same rows as the host mirrors', no
                                             # host work per env (measured equal within noise on the lock-step configs)
        self.weight_version = 0

    def refresh_weights(self, src=0):
        """Weight pull at a move boundary: every rank takes rank `src`'s flat buffer."""
        self.flat.broadcast(src=src)
        self.weight_version += 1

    def close(self):
        self.engine.close()

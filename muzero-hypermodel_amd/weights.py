"""Weight distribution to the self-play actors.

The reference moves weights as a pickled CPU state_dict through Ray's object store: the trainer
`set_info`s it every `checkpoint_interval` steps (trainer.py:87-95) and every actor `get_info`s it
by value before each game (self_play.py:37).  Here the model's parameters AND buffers (BatchNorm
running statistics) live in one flat fp32 device buffer per actor and a refresh is a single
`torch.distributed.broadcast` of that buffer -- RCCL over xGMI on MI355X (backend "nccl"), gloo on
CPU for tests.  There is no gradient all-reduce on this path.
"""
import numpy
import torch
import torch.distributed as dist


class FlatWeights:
    """One contiguous fp32 buffer aliasing every floating-point tensor of a model's state dict.

    After construction the model's parameters / buffers are views into `self.flat`, so a broadcast
    into `flat` IS the weight update: no unpack pass, no per-tensor launches."""

    def __init__(self, model):
        self.model = model
        state = model.state_dict(keep_vars=True)
        self.float_keys = [k for k, v in state.items() if v.dtype == torch.float32]
        self.other_keys = [k for k in state if k not in self.float_keys]  # e.g. num_batches_tracked
        self.shapes = {k: tuple(state[k].shape) for k in self.float_keys}
        self.offsets = {}
        total = 0
        for k in self.float_keys:
            self.offsets[k] = total
            total += state[k].numel()
        self.numel = total
        device = next(model.parameters()).device
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for k in self.float_keys:
                t = state[k]
                view = self.flat[self.offsets[k]: self.offsets[k] + t.numel()].view(t.shape)
                view.copy_(t)
                t.data = view  # re-point the parameter / buffer at the flat storage
        # the tensors moved: anything that recorded their addresses (a captured hipGraph, packed constants) is stale
        model._storage_generation = getattr(model, "_storage_generation", 0) + 1
        if hasattr(model, "refresh_inference_constants") and device.type == "cuda":
            model.refresh_inference_constants()

    def nbytes(self):
        return self.numel * 4

    def load_state_dict(self, weights):
        """Fill the flat buffer from a reference-style state dict (e.g. a checkpoint's "weights")."""
        with torch.no_grad():
            for k in self.float_keys:
                src = torch.as_tensor(numpy.asarray(weights[k]) if not torch.is_tensor(weights[k]) else weights[k])
                self.flat[self.offsets[k]: self.offsets[k] + src.numel()].copy_(src.reshape(-1))
        if hasattr(self.model, "refresh_inference_constants"):
            self.model.refresh_inference_constants()

    def state_dict(self):
        out = {k: self.flat[self.offsets[k]: self.offsets[k] + int(numpy.prod(self.shapes[k], dtype=numpy.int64))]
               .view(self.shapes[k]).detach().cpu().clone() for k in self.float_keys}
        full = self.model.state_dict()
        for k in self.other_keys:
            out[k] = full[k].detach().cpu().clone()
        return out

    def broadcast(self, src=0, group=None, async_op=False):
        """Refresh every actor's weights from rank `src` (the trainer / shared-storage role)."""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        work = dist.broadcast(self.flat, src=src, group=group, async_op=async_op)
        if not async_op and hasattr(self.model, "refresh_inference_constants"):
            self.model.refresh_inference_constants()   # folded batch-norm constants follow the new weights
        return work

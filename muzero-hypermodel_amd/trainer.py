"""Trainer with the reference's interface (trainer.py:11-298), SURVEY.md section 8(f) row 4: the unroll that
turns a replay batch into a gradient step, and the producer side of the weight hand-over to the actors.

The arithmetic is PyTorch-ROCm (fp32, the reference's operation order); what changes is the plumbing around it:
batches arrive as CUDA tensors from the device replay store (replay_buffer.ReplayBuffer.get_batch) and never
visit the host, the new priorities are computed on the device and copied back once per step, and fresh weights
leave through `publish` into the flat buffer the actors alias (weights.FlatWeights) -- one RCCL broadcast
instead of a state_dict through shared storage.
"""
import copy
import ctypes
import time

import numpy
import torch

from . import models


def _scale_gradient(tensor, divisor):
    """Leaves `tensor` as it is and divides the gradient flowing back through it by `divisor` (a number or a
    per-sample tensor).  A tensor hook, not an autograd node: the graph -- and with it the order in which the
    shared parameters accumulate their gradients -- stays the reference's, which one-step Adam parity needs."""
    tensor.register_hook(lambda grad: grad / divisor)
    return tensor


class _UnrollLoss(torch.autograd.Function):
    """The loss of one training step over all unrolled positions as ONE HIP launch (include/mztrain.h
    mztrain_unroll_loss, csrc/trainer_kernels.hip): two-hot value / reward targets, the three cross-entropies per step,
    their sums over the steps in the reference's order, the per-sample total, the new PER priorities, and -- kept for
    backward -- the gradient of the per-sample total with respect to every logit (gradient scales folded in).
    Forward returns (per-sample loss [B], per-sample head sums [3, B], priorities [B, K+1]); only the first is
    differentiable.  CUDA tensors only: on the CPU the trainer keeps the torch expression."""

    @staticmethod
    def forward(ctx, value_logits, reward_logits, policy_logits, batch, support_size, value_loss_weight, per_alpha):
        from . import _native
        lib = _native.load()      # raises when the HIP library is missing: no silent fallback on a GPU box
        steps, size, full = value_logits.shape
        actions = policy_logits.shape[2]
        assert full == 2 * support_size + 1 and reward_logits.shape == value_logits.shape
        v, r, p = (t.detach().contiguous().float() for t in (value_logits, reward_logits, policy_logits))
        tv, tr, tp, gs = (batch[key].contiguous() for key in ("values", "rewards", "policies", "gradient_scales"))
        weight = batch["weights"].contiguous() if batch["weights"] is not None else None
        dev = v.device
        sample_loss = torch.empty(size, device=dev)
        head_sums = torch.empty((3, size), device=dev)
        priorities = torch.empty((size, steps), device=dev)
        grads = (torch.empty_like(v), torch.empty_like(r), torch.empty_like(p))
        args = _native.MzTrainLossArgs(
            value_logits=v.data_ptr(), reward_logits=r.data_ptr(), policy_logits=p.data_ptr(), target_value=tv.data_ptr(),
            target_reward=tr.data_ptr(), target_policy=tp.data_ptr(), gradient_scale=gs.data_ptr(),
            weight=weight.data_ptr() if weight is not None else None, batch=size, steps=steps, support_size=support_size,
            actions=actions, value_loss_weight=float(value_loss_weight), per_alpha=float(per_alpha),
            sample_loss=sample_loss.data_ptr(), head_sums=head_sums.data_ptr(), priorities=priorities.data_ptr(),
            grad_value=grads[0].data_ptr(), grad_reward=grads[1].data_ptr(), grad_policy=grads[2].data_ptr())
        rc = lib.mztrain_unroll_loss(ctypes.byref(args), torch.cuda.current_stream(dev).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mztrain_unroll_loss failed ({rc})")
        ctx.save_for_backward(*grads)
        ctx.mark_non_differentiable(head_sums, priorities)
        return sample_loss, head_sums, priorities

    @staticmethod
    def backward(ctx, grad_loss, _grad_sums, _grad_priorities):
        g = grad_loss.reshape(1, -1, 1)
        gv, gr, gp = ctx.saved_tensors
        return gv * g, gr * g, gp * g, None, None, None, None


class Trainer:
    native_loss = True      # on the GPU: the loss of a step as one HIP launch (False: the torch expression, for timing)

    def __init__(self, initial_checkpoint, config, device=None, graph=False):
        """graph=True (GPU, Adam): the whole step -- unroll, loss launch, backward, optimizer -- is captured into a
        hipGraph at the first update_weights and replayed afterwards (the step is launch-bound: ~1 500 launches of
        microsecond kernels for the CartPole network); batches are copied into the capture's static buffers."""
        self.config = config
        self._graph_mode = bool(graph)
        self._graph = None
        numpy.random.seed(self.config.seed)
        torch.manual_seed(self.config.seed)
        self.model = models.MuZeroNetwork(self.config)
        self.model.set_weights(copy.deepcopy(initial_checkpoint["weights"]))
        if device is None:
            device = torch.device("cuda" if self.config.train_on_gpu else "cpu")
        self.model.to(torch.device(device))
        self.model.train()
        self.training_step = initial_checkpoint["training_step"]
        if self.config.optimizer == "SGD":
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=self.config.lr_init,
                                             momentum=self.config.momentum, weight_decay=self.config.weight_decay)
        elif self.config.optimizer == "Adam":
            if self._graph_mode:        # step counter and learning rate live on the device, so a replay can move them
                self._lr = torch.tensor(float(self.config.lr_init), device=torch.device(device))
                self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self._lr, capturable=True,
                                                  weight_decay=self.config.weight_decay)
            else:
                self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.config.lr_init,
                                                  weight_decay=self.config.weight_decay)
        else:
            raise NotImplementedError(
                f"{self.config.optimizer} is not implemented. You can change the optimizer manually in trainer.py.")
        if self._graph_mode and (self.config.optimizer != "Adam" or torch.device(device).type != "cuda"):
            raise NotImplementedError("Trainer(graph=True) captures an Adam step on the GPU")
        self._lr_host = float(self.config.lr_init)
        if initial_checkpoint.get("optimizer_state") is not None:
            self.optimizer.load_state_dict(copy.deepcopy(self._portable_optimizer_state(initial_checkpoint["optimizer_state"])))
            saved_lr = self.optimizer.param_groups[0]["lr"]
            self._lr_host = float(saved_lr)
            if self._graph_mode:
                # load_state_dict replaced every param group's settings with the checkpoint's: put the device-resident
                # learning rate and the capturable flag back (an eager checkpoint carries a float lr, capturable False and
                # CPU step counters; a graphed one a tensor lr), or update_lr() would fill an orphaned tensor and the
                # capture of optimizer.step() would break / bake the learning rate in
                self._lr.fill_(self._lr_host)
                dev = self._lr.device
                for group in self.optimizer.param_groups:
                    group["lr"], group["capturable"] = self._lr, True
                for state in self.optimizer.state.values():
                    if "step" in state:
                        state["step"] = torch.as_tensor(state["step"], dtype=torch.float32).to(dev).reshape(())

    @staticmethod
    def _portable_optimizer_state(state):
        """An optimizer state dict whose param groups hold plain numbers (a graphed trainer's learning rate is a device
        tensor shared by every group: a checkpoint must not alias it)."""
        out = dict(state)
        out["param_groups"] = [{k: (float(v) if torch.is_tensor(v) and v.numel() == 1 and k == "lr" else v)
                                for k, v in group.items()} for group in state["param_groups"]]
        return out

    def optimizer_state(self):
        """optimizer.state_dict() on the CPU with a float learning rate: what goes into a checkpoint (trainer.py:87-95)."""
        return copy.deepcopy(models.dict_to_cpu(self._portable_optimizer_state(self.optimizer.state_dict())))

    # ---- the loop (trainer.py:62-122) without Ray ----------------------------------------------------
    def continuous_update_weights(self, replay_buffer, shared_storage, max_steps=None):
        while shared_storage.get_info("num_played_games") < 1:
            time.sleep(0.1)
        done = 0
        while self.training_step < self.config.training_steps and not shared_storage.get_info("terminate"):
            index_batch, batch = replay_buffer.get_batch()
            self.update_lr()
            priorities, total_loss, value_loss, reward_loss, policy_loss = self.update_weights(batch)
            if self.config.PER:
                replay_buffer.update_priorities(priorities, index_batch)
            if self.training_step % self.config.checkpoint_interval == 0:
                shared_storage.set_info({"weights": copy.deepcopy(self.model.get_weights()),
                                         "optimizer_state": self.optimizer_state()})
                if self.config.save_model:               # trainer.py:96-97
                    shared_storage.save_checkpoint()
            shared_storage.set_info({"training_step": self.training_step, "lr": self._lr_host,
                                     "total_loss": total_loss, "value_loss": value_loss, "reward_loss": reward_loss,
                                     "policy_loss": policy_loss})
            done += 1
            if max_steps is not None and done >= max_steps:
                break
            if self.config.training_delay:
                time.sleep(self.config.training_delay)
            if self.config.ratio:
                while (self.training_step / max(1, shared_storage.get_info("num_played_steps")) > self.config.ratio
                       and self.training_step < self.config.training_steps and not shared_storage.get_info("terminate")):
                    time.sleep(0.5)

    # ---- one training step (trainer.py:124-268) ---------------------------------------------------------
    def _batch_on_device(self, batch):
        """The seven batch entries as float32 tensors on the model's device (actions: int64 [B, K+1, 1]).  Lists and
        numpy arrays (the reference's replay buffer) and CUDA tensors (the device replay store) are both accepted."""
        device = next(self.model.parameters()).device

        def as_tensor(x):
            if torch.is_tensor(x):
                return x.to(device)
            if isinstance(x, numpy.ndarray):
                x = x.copy()
            return torch.tensor(numpy.asarray(x)).to(device)

        observations, actions, values, rewards, policies, weights, gradient_scales = batch
        return {"observations": as_tensor(observations).float(), "actions": as_tensor(actions).long().unsqueeze(-1),
                "values": as_tensor(values).float(), "rewards": as_tensor(rewards).float(),
                "policies": as_tensor(policies).float(),
                "weights": as_tensor(weights).float() if self.config.PER else None,
                "gradient_scales": as_tensor(gradient_scales).float()}

    def _unroll(self, observations, actions):
        """(value logits, reward logits, policy logits) per unroll step; the hidden state handed from step to step
        passes half of its gradient on (paper appendix "Training", trainer.py:171-173)."""
        out = self.model.initial_inference(observations)
        steps, hidden = [out[:3]], out[3]
        for k in range(1, actions.shape[1]):
            out = self.model.recurrent_inference(hidden, actions[:, k])
            hidden = _scale_gradient(out[3], 2.0)
            steps.append(out[:3])
        return steps

    def update_weights(self, batch):
        cfg = self.config
        b = self._batch_on_device(batch)
        if self._graph_mode:
            return self._step_graphed(b)
        steps = self._unroll(b["observations"], b["actions"])
        if b["values"].is_cuda and self.native_loss:
            return self._step_native(b, steps)
        value_targets = models.scalar_to_support(b["values"], cfg.support_size)
        reward_targets = models.scalar_to_support(b["rewards"], cfg.support_size)
        priorities = torch.zeros_like(b["values"])
        sums = {"value": 0, "reward": 0, "policy": 0}
        for k, (value, reward, policy_logits) in enumerate(steps):
            per_head = dict(zip(("value", "reward", "policy"), self.loss_function(
                value.squeeze(-1), reward.squeeze(-1), policy_logits, value_targets[:, k], reward_targets[:, k],
                b["policies"][:, k])))
            if k == 0:
                del per_head["reward"]           # no reward is predicted for the root position (trainer.py:182-193)
            for head, term in per_head.items():
                if k > 0:                        # every unrolled step contributes 1 / (its gradient scale) of its gradient
                    term = _scale_gradient(term, b["gradient_scales"][:, k])
                sums[head] = sums[head] + term
            with torch.no_grad():                # new priorities: |predicted value - target| ** alpha (trainer.py:199-209)
                predicted = models.support_to_scalar(value, cfg.support_size).squeeze(-1)
                priorities[:, k] = torch.abs(predicted - b["values"][:, k]) ** cfg.PER_alpha

        loss = sums["value"] * cfg.value_loss_weight + sums["reward"] + sums["policy"]
        if cfg.PER:
            loss = loss * b["weights"]           # importance-sampling correction of the prioritised replay
        loss = loss.mean()
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        return (priorities.detach().cpu().numpy(), loss.item(), sums["value"].mean().item(),
                sums["reward"].mean().item(), sums["policy"].mean().item())

    def _step_native(self, b, steps):
        """The same step with everything between the network's outputs and `loss.mean()` in one HIP launch
        (_UnrollLoss): the logits of all unrolled positions are stacked step-major, the kernel returns the
        per-sample loss and keeps its gradient for backward."""
        cfg = self.config
        sample_loss, head_sums, priorities = _UnrollLoss.apply(
            torch.stack([s[0] for s in steps]), torch.stack([s[1] for s in steps]), torch.stack([s[2] for s in steps]), b,
            cfg.support_size, cfg.value_loss_weight, cfg.PER_alpha)
        loss = sample_loss.mean()
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        report = torch.cat([loss.detach().reshape(1), head_sums.mean(dim=1)]).tolist()     # one device-to-host copy
        return (priorities.cpu().numpy(), report[0], report[1], report[2], report[3])

    def _loss_native(self, b):
        steps = self._unroll(b["observations"], b["actions"])
        cfg = self.config
        return _UnrollLoss.apply(torch.stack([s[0] for s in steps]), torch.stack([s[1] for s in steps]),
                                 torch.stack([s[2] for s in steps]), b, cfg.support_size, cfg.value_loss_weight,
                                 cfg.PER_alpha)

    def _step_graphed(self, b):
        """update_weights as one hipGraph replay: the first call warms the step up eagerly on a side stream (what
        stream capture requires), captures it on static copies of the batch, and every call copies its batch in,
        replays, and reads loss / head sums / priorities out of the capture's output tensors."""
        if self._graph is None:
            self._static = {k: (v.clone() if v is not None else None) for k, v in b.items()}
            side = torch.cuda.Stream(device=b["values"].device)
            side.wait_stream(torch.cuda.current_stream(b["values"].device))
            state = (copy.deepcopy(self.model.state_dict()), copy.deepcopy(self.optimizer.state_dict()))
            with torch.cuda.stream(side):
                for _ in range(3):                       # warm-up steps, undone below
                    self.optimizer.zero_grad(set_to_none=True)
                    self._loss_native(self._static)[0].mean().backward()
                    self.optimizer.step()
            torch.cuda.current_stream(b["values"].device).wait_stream(side)
            self.model.load_state_dict(state[0])
            # the optimizer's moments and step counters must exist BEFORE the capture (created inside it they would be
            # re-initialised by every replay): keep the tensors the warm-up made, put the saved values back in place
            live = self.optimizer.state_dict()["state"]
            for index, entry in live.items():
                saved = state[1]["state"].get(index)
                for name, value in entry.items():
                    if torch.is_tensor(value):
                        value.copy_(saved[name]) if saved is not None else value.zero_()
            self.optimizer.zero_grad(set_to_none=True)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                sample_loss, head_sums, priorities = self._loss_native(self._static)
                loss = sample_loss.mean()
                loss.backward()
                self.optimizer.step()
                self._graph_out = (torch.cat([loss.detach().reshape(1), head_sums.mean(dim=1)]), priorities)
            # (the capture itself does not execute the step)
        for key, value in b.items():
            if value is not None:
                self._static[key].copy_(value)
        self._graph.replay()
        self.training_step += 1
        report = self._graph_out[0].tolist()
        return (self._graph_out[1].cpu().numpy(), report[0], report[1], report[2], report[3])

    def update_lr(self):
        lr = self.config.lr_init * self.config.lr_decay_rate ** (self.training_step / self.config.lr_decay_steps)
        self._lr_host = float(lr)              # what set_info publishes: a Python float, as in the reference
        if self._graph_mode:
            self._lr.fill_(lr)
            return
        for param_group in self.optimizer.param_groups:
            param_group["lr"] = lr

    @staticmethod
    def loss_function(value, reward, policy_logits, target_value, target_reward, target_policy):
        value_loss = (-target_value * torch.nn.LogSoftmax(dim=1)(value)).sum(1)
        reward_loss = (-target_reward * torch.nn.LogSoftmax(dim=1)(reward)).sum(1)
        policy_loss = (-target_policy * torch.nn.LogSoftmax(dim=1)(policy_logits)).sum(1)
        return value_loss, reward_loss, policy_loss

    # ---- producer side of the weight hand-over -----------------------------------------------------------
    @torch.no_grad()
    def publish(self, flat):
        """Copy the trained parameters (and BN statistics) into an actor-side flat buffer (weights.FlatWeights):
        the actors' models alias it, so the next search -- or the next RCCL broadcast from this rank -- sees them."""
        flat.load_state_dict(self.model.state_dict())

"""Trainer with the reference's interface (trainer.py:11-298), SURVEY.md section 8(f) row 4: the unroll that
turns a replay batch into a gradient step, and the producer side of the weight hand-over to the actors.

The arithmetic is PyTorch-ROCm (fp32, the reference's operation order); what changes is the plumbing around it:
batches arrive as CUDA tensors from the device replay store (replay_buffer.ReplayBuffer.get_batch) and never
visit the host, the new priorities are computed on the device and copied back once per step, and fresh weights
leave through `publish` into the flat buffer the actors alias (weights.FlatWeights) -- one RCCL broadcast
instead of a state_dict through shared storage.
"""
import copy
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


class Trainer:
    def __init__(self, initial_checkpoint, config, device=None):
        self.config = config
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
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=self.config.lr_init,
                                              weight_decay=self.config.weight_decay)
        else:
            raise NotImplementedError(
                f"{self.config.optimizer} is not implemented. You can change the optimizer manually in trainer.py.")
        if initial_checkpoint.get("optimizer_state") is not None:
            self.optimizer.load_state_dict(copy.deepcopy(initial_checkpoint["optimizer_state"]))

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
                                         "optimizer_state": copy.deepcopy(models.dict_to_cpu(self.optimizer.state_dict()))})
                if self.config.save_model:               # trainer.py:96-97
                    shared_storage.save_checkpoint()
            shared_storage.set_info({"training_step": self.training_step, "lr": self.optimizer.param_groups[0]["lr"],
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
        value_targets = models.scalar_to_support(b["values"], cfg.support_size)
        reward_targets = models.scalar_to_support(b["rewards"], cfg.support_size)
        priorities = torch.zeros_like(b["values"])
        sums = {"value": 0, "reward": 0, "policy": 0}
        for k, (value, reward, policy_logits) in enumerate(self._unroll(b["observations"], b["actions"])):
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

    def update_lr(self):
        lr = self.config.lr_init * self.config.lr_decay_rate ** (self.training_step / self.config.lr_decay_steps)
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

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
    def update_weights(self, batch):
        observation_batch, action_batch, target_value, target_reward, target_policy, weight_batch, gradient_scale_batch = batch
        device = next(self.model.parameters()).device

        def f32(x):
            if torch.is_tensor(x):
                return x.to(device=device, dtype=torch.float32)
            return torch.tensor(numpy.asarray(x)).float().to(device)

        target_value = f32(target_value)
        target_value_scalar = target_value                       # scalars kept for the new priorities
        if self.config.PER:
            weight_batch = f32(weight_batch.copy() if isinstance(weight_batch, numpy.ndarray) else weight_batch)
        observation_batch = f32(observation_batch)
        action_batch = (action_batch.to(device) if torch.is_tensor(action_batch)
                        else torch.tensor(numpy.asarray(action_batch)).to(device)).long().unsqueeze(-1)
        target_reward = f32(target_reward)
        target_policy = f32(target_policy)
        gradient_scale_batch = f32(gradient_scale_batch)
        priorities = torch.zeros_like(target_value_scalar)

        target_value = models.scalar_to_support(target_value, self.config.support_size)
        target_reward = models.scalar_to_support(target_reward, self.config.support_size)

        # predictions along the unroll (the 0.5 hook: paper appendix Training, trainer.py:171-173)
        value, reward, policy_logits, hidden_state = self.model.initial_inference(observation_batch)
        predictions = [(value, reward, policy_logits)]
        for i in range(1, action_batch.shape[1]):
            value, reward, policy_logits, hidden_state = self.model.recurrent_inference(hidden_state, action_batch[:, i])
            hidden_state.register_hook(lambda grad: grad * 0.5)
            predictions.append((value, reward, policy_logits))

        value_loss, reward_loss, policy_loss = (0, 0, 0)
        value, reward, policy_logits = predictions[0]
        current_value_loss, _, current_policy_loss = self.loss_function(
            value.squeeze(-1), reward.squeeze(-1), policy_logits, target_value[:, 0], target_reward[:, 0], target_policy[:, 0])
        value_loss += current_value_loss
        policy_loss += current_policy_loss
        with torch.no_grad():
            predicted = models.support_to_scalar(value, self.config.support_size).squeeze(-1)
            priorities[:, 0] = torch.abs(predicted - target_value_scalar[:, 0]) ** self.config.PER_alpha
        for i in range(1, len(predictions)):
            value, reward, policy_logits = predictions[i]
            current_value_loss, current_reward_loss, current_policy_loss = self.loss_function(
                value.squeeze(-1), reward.squeeze(-1), policy_logits, target_value[:, i], target_reward[:, i],
                target_policy[:, i])
            scale = gradient_scale_batch[:, i]
            current_value_loss.register_hook(lambda grad, scale=scale: grad / scale)
            current_reward_loss.register_hook(lambda grad, scale=scale: grad / scale)
            current_policy_loss.register_hook(lambda grad, scale=scale: grad / scale)
            value_loss += current_value_loss
            reward_loss += current_reward_loss
            policy_loss += current_policy_loss
            with torch.no_grad():
                predicted = models.support_to_scalar(value, self.config.support_size).squeeze(-1)
                priorities[:, i] = torch.abs(predicted - target_value_scalar[:, i]) ** self.config.PER_alpha

        loss = value_loss * self.config.value_loss_weight + reward_loss + policy_loss
        if self.config.PER:
            loss *= weight_batch
        loss = loss.mean()
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.training_step += 1
        return (priorities.detach().cpu().numpy(), loss.item(), value_loss.mean().item(), reward_loss.mean().item(),
                policy_loss.mean().item())

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

"""Atari plugin config (reference games/atari.py:16-132): the paper-sized network (16 blocks x 256
channels, 32 stacked frames, support 300).  Config only -- see breakout.py for why there is no env."""
from ._config import BaseMuZeroConfig


class MuZeroConfig(BaseMuZeroConfig):
    GAME = "atari"
    OVERRIDES = dict(
        observation_shape=(3, 96, 96), action_space=[0, 1, 2, 3], stacked_observations=32,
        num_workers=350, max_moves=27000, network="resnet", support_size=300, downsample="resnet",
        blocks=16, channels=256, reduced_channels_reward=256, reduced_channels_value=256,
        reduced_channels_policy=256, resnet_fc_reward_layers=[256, 256],
        resnet_fc_value_layers=[256, 256], resnet_fc_policy_layers=[256, 256], encoding_size=10,
        fc_value_layers=[], fc_policy_layers=[], training_steps=int(1000e3), batch_size=1024,
        checkpoint_interval=int(1e3), value_loss_weight=0.25, optimizer="SGD", lr_init=0.05,
        lr_decay_rate=0.1, lr_decay_steps=350e3, replay_buffer_size=int(1e6), num_unroll_steps=5,
        td_steps=10, PER_alpha=1, ratio=None)

    def visit_softmax_temperature_fn(self, trained_steps):
        if trained_steps < 500e3:
            return 1.0
        if trained_steps < 750e3:
            return 0.5
        return 0.25

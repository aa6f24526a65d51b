"""Breakout plugin config (reference games/breakout.py:16-132).  The reference's Game wraps the gym
Atari emulator + cv2 (games/breakout.py:135-198), neither of which exists in this image nor is
vendored by the reference; only the config (network shapes, search constants) is provided, plus the
`atari84_config()` variant BASELINE.json's config #5 names ("84x84x4 conv representation")."""
from ._config import BaseMuZeroConfig


class MuZeroConfig(BaseMuZeroConfig):
    GAME = "breakout"
    OVERRIDES = dict(
        observation_shape=(3, 96, 96), action_space=[0, 1, 2, 3], max_moves=2500, num_simulations=30,
        network="resnet", downsample="resnet", blocks=2, channels=16, reduced_channels_reward=4,
        reduced_channels_value=4, reduced_channels_policy=4, resnet_fc_reward_layers=[16],
        resnet_fc_value_layers=[16], resnet_fc_policy_layers=[16], encoding_size=10,
        fc_value_layers=[], fc_policy_layers=[], training_steps=int(1000e3), batch_size=16,
        checkpoint_interval=500, value_loss_weight=0.25, lr_init=0.005, lr_decay_rate=1,
        lr_decay_steps=350e3, replay_buffer_size=int(1e6), num_unroll_steps=5, td_steps=10,
        PER_alpha=1, use_last_model_value=False, ratio=None)

    def visit_softmax_temperature_fn(self, trained_steps):
        if trained_steps < 500e3:
            return 1.0
        if trained_steps < 750e3:
            return 0.5
        return 0.25


def atari84_config():
    """BASELINE.json config #5: 4 stacked 84x84 frames through the CNN down-sampler
    (models.py:278-297, 318-327), 50 simulations."""
    config = MuZeroConfig()
    config.observation_shape = (4, 84, 84)
    config.stacked_observations = 0
    config.downsample = "CNN"
    config.num_simulations = 50
    return config

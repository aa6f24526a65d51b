"""Shared MuZeroConfig machinery.

The reference spells every game's config out as ~100 attribute assignments
(games/cartpole.py:11-127 and siblings).  Here the attribute names and values are the same data,
kept as one defaults table plus per-game overrides; instances are plain attribute bags, so
`MuZero.__init__`-style dict overrides (muzero.py:55-60) and direct attribute edits keep working.
"""
import datetime
import os

# games/cartpole.py:15-111 values; every game file of the reference defines exactly these fields
DEFAULTS = dict(
    seed=0, max_num_gpus=None,
    observation_shape=(1, 1, 4), action_space=[0, 1], players=[0], stacked_observations=0,
    muzero_player=0, opponent=None,
    num_workers=1, selfplay_on_gpu=False, max_moves=500, num_simulations=50, discount=0.997,
    temperature_threshold=None, root_dirichlet_alpha=0.25, root_exploration_fraction=0.25,
    pb_c_base=19652, pb_c_init=1.25,
    network="fullyconnected", support_size=10,
    downsample=False, blocks=1, channels=2, reduced_channels_reward=2, reduced_channels_value=2,
    reduced_channels_policy=2, resnet_fc_reward_layers=[], resnet_fc_value_layers=[],
    resnet_fc_policy_layers=[],
    encoding_size=8, fc_representation_layers=[], fc_dynamics_layers=[16], fc_reward_layers=[16],
    fc_value_layers=[16], fc_policy_layers=[16],
    save_model=True, training_steps=10000, batch_size=128, checkpoint_interval=10,
    value_loss_weight=1, train_on_gpu=None, optimizer="Adam", weight_decay=1e-4, momentum=0.9,
    lr_init=0.02, lr_decay_rate=0.9, lr_decay_steps=1000,
    replay_buffer_size=500, num_unroll_steps=10, td_steps=50, PER=True, PER_alpha=0.5,
    use_last_model_value=True, reanalyse_on_gpu=False,
    self_play_delay=0, training_delay=0, ratio=1.5,
)


class BaseMuZeroConfig:
    GAME = "game"
    OVERRIDES = {}

    def __init__(self):
        import copy
        import torch
        for key, value in {**DEFAULTS, **self.OVERRIDES}.items():
            setattr(self, key, copy.deepcopy(value))
        if self.train_on_gpu is None:
            # (the reference asks torch.cuda.is_available(); counting devices gives the same answer without
            # initialising the GPU, which a config object built in a CPU-only helper process must not do)
            self.train_on_gpu = torch.cuda.device_count() > 0
        self.results_path = os.path.join(
            os.path.dirname(os.path.realpath(__file__)), "../results", self.GAME,
            datetime.datetime.now().strftime("%Y-%m-%d--%H-%M-%S"))

    def visit_softmax_temperature_fn(self, trained_steps):
        """Temperature of the visit-count distribution used to pick the played action."""
        return 1

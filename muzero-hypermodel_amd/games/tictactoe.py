"""Tic-tac-toe plugin (reference games/tictactoe.py): config :10-121, rules :242-305, expert :307-348.

Same observations, rewards (x20), legal-action order and expert-agent behaviour (including its use
of the global numpy RNG) as the reference, so recorded reference games replay move for move
(tests/golden/g6_tictactoe_games.npz).  The rules are table-driven here: the 8 winning lines are
index triples into the flattened board.
"""
import numpy

from ._config import BaseMuZeroConfig
from .abstract_game import AbstractGame


class MuZeroConfig(BaseMuZeroConfig):
    GAME = "tictactoe"
    OVERRIDES = dict(
        observation_shape=(3, 3, 3), action_space=list(range(9)), players=[0, 1], opponent="expert",
        max_moves=9, num_simulations=25, discount=1, root_dirichlet_alpha=0.1, network="resnet",
        channels=16, reduced_channels_reward=16, reduced_channels_value=16,
        reduced_channels_policy=16, resnet_fc_reward_layers=[8], resnet_fc_value_layers=[8],
        resnet_fc_policy_layers=[8], encoding_size=32, fc_value_layers=[], fc_policy_layers=[],
        training_steps=1000000, batch_size=64, value_loss_weight=0.25, lr_init=0.003,
        lr_decay_rate=1, lr_decay_steps=10000, replay_buffer_size=3000, num_unroll_steps=20,
        td_steps=20, ratio=None)

    def visit_softmax_temperature_fn(self, trained_steps):
        return 1


# scan order of the reference's expert: row i, column i for i = 0..2, then the two diagonals
_LINES = []
for _i in range(3):
    _LINES.append([3 * _i, 3 * _i + 1, 3 * _i + 2])
    _LINES.append([_i, _i + 3, _i + 6])
_LINES.append([0, 4, 8])
_LINES.append([2, 4, 6])
_LINES = numpy.array(_LINES)


class TicTacToe:
    def __init__(self):
        self.board = numpy.zeros((3, 3), dtype="int32")
        self.player = 1

    def to_play(self):
        return 0 if self.player == 1 else 1

    def reset(self):
        self.board = numpy.zeros((3, 3), dtype="int32")
        self.player = 1
        return self.get_observation()

    def step(self, action):
        self.board[action // 3, action % 3] = self.player
        won = self.have_winner()
        done = won or len(self.legal_actions()) == 0
        self.player *= -1
        return self.get_observation(), 1 if won else 0, done

    def get_observation(self):
        mine = (self.board == 1).astype("int32")
        theirs = (self.board == -1).astype("int32")
        turn = numpy.full((3, 3), self.player, dtype="int32")
        return numpy.array([mine, theirs, turn], dtype="int32")

    def legal_actions(self):
        return [int(i) for i in numpy.flatnonzero(self.board.reshape(-1) == 0)]

    def have_winner(self):
        sums = self.board.reshape(-1)[_LINES].sum(axis=1)
        return bool((sums == 3 * self.player).any())

    def expert_action(self):
        flat = self.board.reshape(-1)
        action = numpy.random.choice(self.legal_actions())  # drawn even when a rule overrides it
        for line in _LINES:
            total = int(flat[line].sum())
            if abs(total) == 2:
                action = int(line[numpy.flatnonzero(flat[line] == 0)[0]])
                if self.player * total > 0:
                    return action  # winning move: take it now; a block is kept but scanning goes on
        return action

    def render(self):
        print(self.board[::-1])


class Game(AbstractGame):
    def __init__(self, seed=None):
        self.env = TicTacToe()

    def step(self, action):
        observation, reward, done = self.env.step(action)
        return observation, reward * 20, done

    def to_play(self):
        return self.env.to_play()

    def legal_actions(self):
        return self.env.legal_actions()

    def reset(self):
        return self.env.reset()

    def render(self):
        self.env.render()
        input("Press enter to take a step ")

    def human_to_action(self):
        while True:
            try:
                row = int(input(f"Enter the row (1, 2 or 3) to play for the player {self.to_play()}: "))
                col = int(input(f"Enter the column (1, 2 or 3) to play for the player {self.to_play()}: "))
                choice = (row - 1) * 3 + (col - 1)
                if choice in self.legal_actions() and 1 <= row <= 3 and 1 <= col <= 3:
                    return choice
            except Exception:
                pass
            print("Wrong input, try again")

    def expert_agent(self):
        return self.env.expert_action()

    def action_to_string(self, action_number):
        return f"Play row {action_number // 3 + 1}, column {action_number % 3 + 1}"

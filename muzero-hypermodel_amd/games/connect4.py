"""Connect-4 plugin (reference games/connect4.py): config :10-121, rules :219-304, expert :306-343.

Same observations, rewards (x10), legal-action order and expert-agent behaviour as the reference
(recorded games: tests/golden/g6_connect4_games.npz).  Win detection is table-driven: the 69
four-cell windows of the 6x7 board are precomputed index quadruples.
"""
import numpy

from ._config import BaseMuZeroConfig
from .abstract_game import AbstractGame

ROWS, COLS = 6, 7


class MuZeroConfig(BaseMuZeroConfig):
    GAME = "connect4"
    OVERRIDES = dict(
        observation_shape=(3, 6, 7), action_space=list(range(7)), players=[0, 1], opponent="expert",
        max_moves=42, num_simulations=200, discount=1, root_dirichlet_alpha=0.3, network="resnet",
        blocks=3, channels=64, reduced_channels_policy=4, resnet_fc_reward_layers=[64],
        resnet_fc_value_layers=[64], resnet_fc_policy_layers=[64], encoding_size=32,
        fc_dynamics_layers=[64], fc_reward_layers=[64], fc_value_layers=[], fc_policy_layers=[],
        training_steps=100000, batch_size=64, value_loss_weight=0.25, lr_init=0.005,
        lr_decay_rate=1, lr_decay_steps=10000, replay_buffer_size=10000, num_unroll_steps=42,
        td_steps=42, ratio=None)

    def visit_softmax_temperature_fn(self, trained_steps):
        return 1


def _windows():
    cells = []
    for r in range(ROWS):
        for c in range(COLS):
            for dr, dc in ((0, 1), (1, 0), (1, 1), (-1, 1)):
                rr, cc = r + 3 * dr, c + 3 * dc
                if 0 <= rr < ROWS and 0 <= cc < COLS:
                    cells.append([(r + i * dr) * COLS + (c + i * dc) for i in range(4)])
    return numpy.array(cells)


_WINDOWS = _windows()


class Connect4:
    def __init__(self):
        self.board = numpy.zeros((ROWS, COLS), dtype="int32")
        self.player = 1

    def to_play(self):
        return 0 if self.player == 1 else 1

    def reset(self):
        self.board = numpy.zeros((ROWS, COLS), dtype="int32")
        self.player = 1
        return self.get_observation()

    def step(self, action):
        column = self.board[:, action]
        column[numpy.flatnonzero(column == 0)[0]] = self.player  # lowest free cell (row 0 = bottom)
        won = self.have_winner()
        done = won or len(self.legal_actions()) == 0
        self.player *= -1
        return self.get_observation(), 1 if won else 0, done

    def get_observation(self):
        mine = numpy.where(self.board == 1, 1.0, 0.0)
        theirs = numpy.where(self.board == -1, 1.0, 0.0)
        turn = numpy.full((ROWS, COLS), self.player, dtype="int32")
        return numpy.array([mine, theirs, turn])

    def legal_actions(self):
        return [int(c) for c in numpy.flatnonzero(self.board[ROWS - 1] == 0)]

    def have_winner(self):
        sums = self.board.reshape(-1)[_WINDOWS].sum(axis=1)
        return bool((sums == 4 * self.player).any())

    def expert_action(self):
        """Scan every 4x4 sub-board for three-in-a-line threats, in the reference's order: rows and
        columns i = 0..3 interleaved, then the two diagonals; a playable winning cell returns at once,
        a playable blocking cell is remembered while the scan goes on."""
        board = self.board
        action = numpy.random.choice(self.legal_actions())
        for k in range(3):
            for l in range(4):
                sub = board[k:k + 4, l:l + 4]
                for i in range(4):
                    row_sum = int(sub[i, :].sum())
                    if abs(row_sum) == 3:
                        ind = int(numpy.flatnonzero(sub[i, :] == 0)[0])
                        if numpy.count_nonzero(board[:, ind + l]) == i + k:
                            action = ind + l
                            if self.player * row_sum > 0:
                                return action
                    col_sum = int(sub[:, i].sum())
                    if abs(col_sum) == 3:
                        action = i + l
                        if self.player * col_sum > 0:
                            return action
                diag = sub.diagonal()
                anti = numpy.fliplr(sub).diagonal()
                diag_sum, anti_sum = int(diag.sum()), int(anti.sum())
                if abs(diag_sum) == 3:
                    ind = int(numpy.flatnonzero(diag == 0)[0])
                    if numpy.count_nonzero(board[:, ind + l]) == ind + k:
                        action = ind + l
                        if self.player * diag_sum > 0:
                            return action
                if abs(anti_sum) == 3:
                    ind = int(numpy.flatnonzero(anti == 0)[0])
                    if numpy.count_nonzero(board[:, 3 - ind + l]) == ind + k:
                        action = 3 - ind + l
                        if self.player * anti_sum > 0:
                            return action
        return action

    def render(self):
        print(self.board[::-1])


class Game(AbstractGame):
    def __init__(self, seed=None):
        self.env = Connect4()

    def step(self, action):
        observation, reward, done = self.env.step(action)
        return observation, reward * 10, done

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
        choice = input(f"Enter the column to play for the player {self.to_play()}: ")
        while choice not in [str(action) for action in self.legal_actions()]:
            choice = input("Enter another column : ")
        return int(choice)

    def expert_agent(self):
        return self.env.expert_action()

    def action_to_string(self, action_number):
        return f"Play column {action_number + 1}"

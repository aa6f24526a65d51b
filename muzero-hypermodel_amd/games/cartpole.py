"""CartPole-v1 plugin (reference games/cartpole.py).

Config values: games/cartpole.py:15-127.  The reference's Game wraps `gym.make("CartPole-v1")`
(games/cartpole.py:136-151); gym is a third-party dependency that is absent from this image and is
not vendored by the reference, so the environment below restates gym's published classic-control
CartPole equations (Barto, Sutton & Anderson; Euler integration, tau = 0.02).  Env-side parity is
therefore UNPINNED; the MCTS path and its benchmarks use synthetic observations.
"""
import math

import numpy

from ._config import BaseMuZeroConfig
from .abstract_game import AbstractGame


class MuZeroConfig(BaseMuZeroConfig):
    GAME = "cartpole"
    OVERRIDES = {}

    def visit_softmax_temperature_fn(self, trained_steps):
        if trained_steps < 0.5 * self.training_steps:
            return 1.0
        if trained_steps < 0.75 * self.training_steps:
            return 0.5
        return 0.25


class CartPolePhysics:
    GRAVITY, MASS_CART, MASS_POLE, HALF_LENGTH, FORCE, TAU = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    THETA_LIMIT = 12 * 2 * math.pi / 360
    X_LIMIT = 2.4
    MAX_STEPS = 500

    def __init__(self, seed=None):
        self.rng = numpy.random.RandomState(seed)
        self.state = numpy.zeros(4)
        self.steps = 0

    def reset(self):
        self.state = self.rng.uniform(-0.05, 0.05, size=4)
        self.steps = 0
        return self.state.astype(numpy.float32)

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.FORCE if action == 1 else -self.FORCE
        total_mass = self.MASS_CART + self.MASS_POLE
        pole_ml = self.MASS_POLE * self.HALF_LENGTH
        cos_t, sin_t = math.cos(theta), math.sin(theta)
        temp = (force + pole_ml * theta_dot ** 2 * sin_t) / total_mass
        theta_acc = (self.GRAVITY * sin_t - cos_t * temp) / (
            self.HALF_LENGTH * (4.0 / 3.0 - self.MASS_POLE * cos_t ** 2 / total_mass))
        x_acc = temp - pole_ml * theta_acc * cos_t / total_mass
        self.state = numpy.array([x + self.TAU * x_dot, x_dot + self.TAU * x_acc,
                                  theta + self.TAU * theta_dot, theta_dot + self.TAU * theta_acc])
        self.steps += 1
        fell = abs(self.state[0]) > self.X_LIMIT or abs(self.state[2]) > self.THETA_LIMIT
        done = bool(fell or self.steps >= self.MAX_STEPS)
        return self.state.astype(numpy.float32), 1.0, done


class Game(AbstractGame):
    def __init__(self, seed=None):
        self.env = CartPolePhysics(seed)

    def step(self, action):
        observation, reward, done = self.env.step(action)
        return numpy.array([[observation]]), reward, done

    def legal_actions(self):
        return list(range(2))

    def reset(self):
        return numpy.array([[self.env.reset()]])

    def render(self):
        print(self.env.state)

    def action_to_string(self, action_number):
        names = {0: "Push cart to the left", 1: "Push cart to the right"}
        return f"{action_number}. {names[action_number]}"

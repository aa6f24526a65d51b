"""The Game plugin contract (reference games/abstract_game.py:4-105), unchanged: any game written
for the reference plugs into this engine as is."""
from abc import ABC, abstractmethod


class AbstractGame(ABC):
    """Inherit this class for muzero to play."""

    @abstractmethod
    def __init__(self, seed=None):
        pass

    @abstractmethod
    def step(self, action):
        """Apply `action`; returns (new observation, reward, done)."""

    def to_play(self):
        """Current player; an element of config.players."""
        return 0

    @abstractmethod
    def legal_actions(self):
        """Legal actions this turn: a list of integers, subset of config.action_space."""

    @abstractmethod
    def reset(self):
        """Start a new game; returns the initial observation."""

    def close(self):
        """Release the game's resources."""

    @abstractmethod
    def render(self):
        """Display the game state."""

    def human_to_action(self):
        """Ask a human for a legal action (multiplayer evaluation)."""
        choice = input(f"Enter the action to play for the player {self.to_play()}: ")
        while int(choice) not in self.legal_actions():
            choice = input("Ilegal action. Enter another action : ")
        return int(choice)

    def expert_agent(self):
        """Hard-coded opponent used to assess progress in multiplayer games."""
        raise NotImplementedError

    def action_to_string(self, action_number):
        """Human-readable name of an action."""
        return str(action_number)

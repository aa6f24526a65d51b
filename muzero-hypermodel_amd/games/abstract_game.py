"""The Game plugin contract (reference games/abstract_game.py:4-105), unchanged for plugin authors: any game
written for the reference plugs into this engine as is.

Same surface -- `step`, `legal_actions`, `reset`, `render` must be provided; `to_play`, `close`,
`human_to_action`, `expert_agent`, `action_to_string` have the reference's defaults -- stated as a table of
required methods checked at instantiation (the effect of the reference's ABC: an incomplete game raises
TypeError when it is created, not when the missing method is first called)."""

_REQUIRED = {
    "step": "step(action) -> (observation, reward, done): apply `action` to the game",
    "legal_actions": "legal_actions() -> list of ints, a subset of config.action_space, for the current turn",
    "reset": "reset() -> the initial observation of a new game",
    "render": "render(): display the game state",
}


def _missing(name):
    def method(self, *args, **kwargs):
        raise NotImplementedError(f"Game plugins must implement {_REQUIRED[name]}")
    method.__name__ = name
    method.__doc__ = _REQUIRED[name]
    method.__isabstractmethod__ = True
    return method


class AbstractGame:
    """Inherit this class for muzero to play."""

    def __new__(cls, *args, **kwargs):
        absent = sorted(n for n in _REQUIRED if getattr(getattr(cls, n), "__isabstractmethod__", False))
        if absent:
            raise TypeError(f"Can't instantiate game class {cls.__name__} without the methods {', '.join(absent)}")
        return super().__new__(cls)

    def __init__(self, seed=None):
        pass

    def to_play(self):
        """Current player; an element of config.players (single-player games keep this default)."""
        return 0

    def close(self):
        """Release the game's resources."""

    def human_to_action(self):
        """Ask a human for a legal action (multiplayer evaluation)."""
        legal = self.legal_actions()
        prompt = f"Enter the action to play for the player {self.to_play()}: "
        while True:
            answer = int(input(prompt))
            if answer in legal:
                return answer
            prompt = "Ilegal action. Enter another action : "

    def expert_agent(self):
        """Hard-coded opponent used to assess progress in multiplayer games."""
        raise NotImplementedError

    def action_to_string(self, action_number):
        """Human-readable name of an action."""
        return str(action_number)


for _name in _REQUIRED:
    setattr(AbstractGame, _name, _missing(_name))

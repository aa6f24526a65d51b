"""Game plugins: a `MuZeroConfig` (attribute bag) and a `Game(AbstractGame)` per module, the same
plugin surface as the reference's games/*.py so existing game files keep working unchanged."""

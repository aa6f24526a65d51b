"""Deterministic synthetic network weights shared by the golden generator and the tests.

The reference ships only a CartPole checkpoint.  For the residual networks
(TicTacToe / Connect4 / Atari-like) both sides of every parity test rebuild the
*same* weights from this recipe instead of committing megabytes of floats: the
values depend only on (key name, shape, seed) through numpy's frozen legacy
``RandomState`` stream, not on torch's initialisers or module construction order.
"""
import zlib

import numpy


def synthetic_array(key, shape, dtype, seed):
    """One tensor of the synthetic state dict (numpy array)."""
    rs = numpy.random.RandomState((zlib.crc32(key.encode()) + 7919 * seed) % (2**32))
    shape = tuple(int(s) for s in shape)
    if key.endswith("num_batches_tracked"):
        return numpy.zeros(shape, dtype="int64")
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        return rs.uniform(0.5, 1.5, size=shape).astype("float32")
    if leaf == "running_mean":
        return (0.1 * rs.standard_normal(shape)).astype("float32")
    if leaf == "bias":
        return (0.1 * rs.standard_normal(shape)).astype("float32")
    if leaf == "weight" and len(shape) == 1:  # batch-norm gain
        return rs.uniform(0.5, 1.5, size=shape).astype("float32")
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return (rs.standard_normal(shape) / numpy.sqrt(max(fan_in, 1))).astype("float32")


def synthetic_state_dict(template, seed=0):
    """``template``: mapping key -> object with ``.shape``/``.dtype`` (a torch state_dict).

    Returns ``{key: numpy array}`` with identical keys/shapes.
    """
    out = {}
    for key, value in template.items():
        out[key] = synthetic_array(key, tuple(value.shape), str(value.dtype), seed)
    return out

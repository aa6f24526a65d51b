"""ctypes binding of libmzmcts.so (the C ABI of include/mzmcts.h).

There is NO Python / CPU fallback for the tree kernels: if the library is missing or no HIP device
is present, loading / engine creation raises.
"""
import ctypes
import os

import numpy as np

from .build import LIB_PATH

c_i32_p = ctypes.POINTER(ctypes.c_int32)
c_i64_p = ctypes.POINTER(ctypes.c_int64)
c_u32_p = ctypes.POINTER(ctypes.c_uint32)
c_f64_p = ctypes.POINTER(ctypes.c_double)
c_f32_p = ctypes.POINTER(ctypes.c_float)
c_void = ctypes.c_void_p

ABI_VERSION = 2
ERR_INVALID, ERR_HIP, ERR_EMPTY_LEGAL, ERR_LEGAL_RANGE, ERR_PLAYERS = -1, -2, -3, -4, -5


class MzConfig(ctypes.Structure):
    _fields_ = [("num_envs", ctypes.c_int32), ("num_actions", ctypes.c_int32),
                ("num_simulations", ctypes.c_int32), ("num_players", ctypes.c_int32),
                ("support_size", ctypes.c_int32), ("hidden_floats", ctypes.c_int32),
                ("device", ctypes.c_int32), ("group_width", ctypes.c_int32),
                ("discount", ctypes.c_double), ("pb_c_base", ctypes.c_double),
                ("pb_c_init", ctypes.c_double), ("root_dirichlet_alpha", ctypes.c_double),
                ("root_exploration_fraction", ctypes.c_double), ("hidden_pool", c_void)]


class MzRootStats(ctypes.Structure):
    _fields_ = [("visits", c_i32_p), ("child_value_sum", c_f64_p), ("child_prior", c_f64_p),
                ("child_reward", c_f64_p), ("child_expanded", c_i32_p), ("root_value_sum", c_f64_p),
                ("root_visits", c_i32_p), ("max_tree_depth", c_i32_p),
                ("root_predicted_value", c_f64_p), ("min_max", c_f64_p), ("depth_sum", c_i64_p),
                ("tie_break_words", c_u32_p)]


class MzProfile(ctypes.Structure):
    _fields_ = [("select_ms", ctypes.c_double), ("expand_backup_ms", ctypes.c_double),
                ("root_ms", ctypes.c_double), ("fused_ms", ctypes.c_double),
                ("select_launches", ctypes.c_int64), ("expand_backup_launches", ctypes.c_int64),
                ("root_launches", ctypes.c_int64), ("fused_launches", ctypes.c_int64),
                ("select_depth_sum", ctypes.c_int64), ("simulations", ctypes.c_int64),
                ("step_ms", ctypes.c_double), ("step_launches", ctypes.c_int64)]


class MzTrainLossArgs(ctypes.Structure):
    """include/mztrain.h mztrain_loss_args: raw device pointers + sizes of one training step's loss."""
    _fields_ = [(name, ctypes.c_void_p) for name in ("value_logits", "reward_logits", "policy_logits", "target_value",
                                                     "target_reward", "target_policy", "gradient_scale", "weight")] + \
               [("batch", ctypes.c_int32), ("steps", ctypes.c_int32), ("support_size", ctypes.c_int32),
                ("actions", ctypes.c_int32), ("value_loss_weight", ctypes.c_float), ("per_alpha", ctypes.c_float)] + \
               [(name, ctypes.c_void_p) for name in ("sample_loss", "head_sums", "priorities", "grad_value", "grad_reward",
                                                     "grad_policy")]


class MzTowerGather(ctypes.Structure):
    """include/mzmcts.h mzmcts_tower_gather."""
    _fields_ = [("pool", ctypes.c_void_p), ("parent", ctypes.c_void_p), ("action", ctypes.c_void_p),
                ("envs", ctypes.c_int64), ("hidden_floats", ctypes.c_int32), ("action_space", ctypes.c_float)]


class MzFcDesc(ctypes.Structure):
    _fields_ = [("observation_floats", ctypes.c_int32), ("encoding_size", ctypes.c_int32),
                ("n_hidden", ctypes.c_int32 * 5), ("hidden", (ctypes.c_int32 * 3) * 5)]


# name -> (restype, argtypes); every symbol include/mzmcts.h declares
PROTOTYPES = {
    "mzmcts_abi_version": (ctypes.c_int, []),
    "mzmcts_create": (ctypes.c_int, [ctypes.POINTER(MzConfig), ctypes.POINTER(c_void)]),
    "mzmcts_destroy": (None, [c_void]),
    "mzmcts_last_error": (ctypes.c_char_p, [c_void]),
    "mzmcts_seed": (ctypes.c_int, [c_void, c_u32_p, c_void]),
    "mzmcts_rng_set_state": (ctypes.c_int, [c_void, ctypes.c_int32, c_u32_p, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_double, c_void]),
    "mzmcts_rng_get_state": (ctypes.c_int, [c_void, ctypes.c_int32, c_u32_p, c_i32_p, c_i32_p,
                                            c_f64_p, c_void]),
    "mzmcts_begin_search": (ctypes.c_int, [c_void, c_i32_p, c_i32_p, c_i32_p, ctypes.c_int32,
                                           c_f64_p, c_void]),
    "mzmcts_expand_roots": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "mzmcts_expand_roots_injected": (ctypes.c_int, [c_void, c_void, c_void, c_void]),
    "mzmcts_select": (ctypes.c_int, [c_void, c_void, c_void, c_void]),
    "mzmcts_select_planes": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int32, ctypes.c_int32, c_void]),
    "mzmcts_expand_backup": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "mzmcts_expand_backup_injected": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void]),
    "mzmcts_hidden_slab": (c_void, [c_void, ctypes.c_int32]),
    "mzmcts_next_slab": (ctypes.c_int32, [c_void]),
    "mzmcts_simulations_done": (ctypes.c_int32, [c_void]),
    "mzmcts_set_simulations_done": (ctypes.c_int, [c_void, ctypes.c_int32]),
    "mzmcts_readout": (ctypes.c_int, [c_void, ctypes.POINTER(MzRootStats), c_void]),
    "mzmcts_readout_begin": (ctypes.c_int, [c_void, c_void]),
    "mzmcts_sample_actions": (ctypes.c_int, [c_void, c_f64_p, c_i32_p, c_i32_p]),
    "mzmcts_search_statistics": (ctypes.c_int, [c_void, c_f64_p, c_f64_p]),
    "mzmcts_last_paths": (ctypes.c_int, [c_void, c_i32_p, c_i32_p, c_i32_p, c_void]),
    "mzmcts_set_debug_ties": (ctypes.c_int, [c_void, ctypes.c_int32]),
    "mzmcts_export_tree": (ctypes.c_int, [c_void, ctypes.c_int32, c_i32_p, c_f64_p, c_f64_p, c_f64_p,
                                          c_i32_p, c_void]),
    "mzmcts_fc_configure": (ctypes.c_int, [c_void, ctypes.POINTER(MzFcDesc), c_void, ctypes.c_int64]),
    "mzmcts_fc_initial_inference": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "mzmcts_fc_recurrent_inference": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void, c_void, c_void]),
    "mzmcts_search_fused_fc": (ctypes.c_int, [c_void, c_void, ctypes.c_int32, c_void]),
    "mzmcts_fused_lds_bytes": (ctypes.c_int64, [c_void, ctypes.c_int32]),
    "mzmcts_set_fused_options": (ctypes.c_int, [c_void, ctypes.c_int32, ctypes.c_int32]),
    "mzmcts_fused_variant": (ctypes.c_int32, [c_void]),
    "mzmcts_moves_prepare": (ctypes.c_int, [c_void, ctypes.c_int32, c_i32_p, c_i32_p, c_i32_p, ctypes.c_int32, c_f64_p, c_void]),
    "mzmcts_moves_predraw_next": (ctypes.c_int, [c_void, ctypes.c_int32, c_i32_p, c_i32_p, c_i32_p, ctypes.c_int32, c_f64_p]),
    "mzmcts_moves_submit_next": (ctypes.c_int, [c_void, c_void]),
    "mzmcts_moves_discard_next": (ctypes.c_int, [c_void]),
    "mzmcts_moves_prepare_device": (ctypes.c_int, [c_void, ctypes.c_int32, c_void, c_void, c_void, ctypes.c_int32, c_f64_p, c_void]),
    "mzmcts_moves_inputs": (ctypes.c_int, [c_void, c_i32_p, c_i32_p, c_i32_p]),
    "mzmcts_moves_enqueue": (ctypes.c_int, [c_void, c_void, c_void]),
    "mzmcts_moves_begin_lockstep": (ctypes.c_int, [c_void, c_void]),
    "mzmcts_moves_end_lockstep": (ctypes.c_int, [c_void, c_void]),
    "mzmcts_moves_temperature_threshold": (ctypes.c_int, [c_void, ctypes.c_int32, c_i32_p, c_void]),
    "mzmcts_moves_finished": (ctypes.c_int, [c_void, c_void]),
    "mzmcts_moves_actions": (c_void, [c_void, ctypes.c_int32]),
    "mzmcts_moves_ring": (ctypes.c_int, [c_void, ctypes.POINTER(c_void), c_i64_p, c_i64_p, c_i32_p]),
    "mzmcts_moves_inputs_ring": (ctypes.c_int, [c_void, ctypes.POINTER(c_void), c_i64_p, c_i64_p]),
    "mzmcts_moves_collect": (ctypes.c_int, [c_void, c_i32_p, c_i32_p, c_i32_p, c_f64_p, c_f32_p, c_i32_p, c_void]),
    "mzmcts_set_profiling": (ctypes.c_int, [c_void, ctypes.c_int32]),
    "mzmcts_set_select_queue": (ctypes.c_int, [c_void, ctypes.c_int32]),
    "mzmcts_get_profile": (ctypes.c_int, [c_void, ctypes.POINTER(MzProfile), ctypes.c_int32]),
    "mzmcts_device_bytes": (ctypes.c_int64, [c_void]),
    "mzmcts_state_action_planes": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                  ctypes.c_int32, c_void]),
    "mzmcts_conv_heads": (ctypes.c_int, [c_void, c_void, ctypes.c_int32, c_void, ctypes.c_int64, c_void]),
    "mzmcts_conv_heads_multi": (ctypes.c_int, [c_void, c_void, ctypes.c_int32, c_void, ctypes.c_int64, c_void]),
    "mzmcts_conv_head": (ctypes.c_int, [c_void, c_void, c_void, ctypes.c_int64, c_void]),
    "mzmcts_unit_rescale": (ctypes.c_int, [c_void, c_void, ctypes.c_int64, ctypes.c_int32, c_void]),
    "mzmcts_expand_backup_select": (ctypes.c_int, [c_void] * 8),
    "mzmcts_expand_backup_select_planes": (ctypes.c_int, [c_void] * 7 + [ctypes.c_int32, ctypes.c_int32, c_void]),
    "mzmcts_expand_backup_select_injected": (ctypes.c_int, [c_void] * 7),
    "mzmcts_board_conv_packed_floats": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32]),
    "mzmcts_board_conv_pack": (ctypes.c_int, [c_void, c_void, ctypes.c_int32, ctypes.c_int32, c_void]),
    "mzmcts_board_conv_supported": (ctypes.c_int, [ctypes.c_int32] * 4),
    "mzmcts_board_conv3x3": (ctypes.c_int, [c_void] * 6 + [ctypes.c_int64] + [ctypes.c_int32] * 5 + [c_void]),
    "mzmcts_board_tower_blocks": (ctypes.c_int64, [ctypes.c_int64] + [ctypes.c_int32] * 3),
    "mzmcts_board_tower_heads": (ctypes.c_int, [c_void, c_void, ctypes.c_int64] + [ctypes.c_int32] * 4 +
                                 [c_void, ctypes.c_int32, c_void, ctypes.c_int32, c_void]),
    "mzmcts_board_tower": (ctypes.c_int, [c_void, ctypes.c_int64] + [ctypes.c_int32] * 4 + [c_void, ctypes.c_int32, c_void]),
    "mzmcts_board_conv_split_halfs": (ctypes.c_int64, [ctypes.c_int32, ctypes.c_int32]),
    "mzmcts_board_conv_pack_split": (ctypes.c_int, [c_void, c_void, c_void] + [ctypes.c_int32] * 5 + [c_void]),
    "mzmcts_tower_gather_args": (ctypes.c_int, [c_void, c_void, ctypes.c_int32, ctypes.POINTER(MzTowerGather)]),
    "mzmcts_board_tower_gathered": (ctypes.c_int, [ctypes.POINTER(MzTowerGather), ctypes.c_int64] + [ctypes.c_int32] * 5 +
                                    [c_void, ctypes.c_int32, c_void]),
    "mzmcts_board_tower_split": (ctypes.c_int, [c_void, ctypes.c_int64] + [ctypes.c_int32] * 5 + [c_void, ctypes.c_int32, c_void]),
    "mzmcts_downsample_cnn": (ctypes.c_int, [c_void, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void, c_void,
                                             ctypes.c_int32, ctypes.c_int32, c_void, c_void, ctypes.c_int32, ctypes.c_int32,
                                             ctypes.c_int32, c_void, c_void]),
    "mzmcts_affine_act": (ctypes.c_int, [c_void] * 5 + [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void]),
    # include/mzenv.h
    "mzenv_advance": (ctypes.c_int, [c_void] * 10),
    "mztrain_unroll_loss": (ctypes.c_int, [ctypes.POINTER(MzTrainLossArgs), c_void]),
    "mzhist_create": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(c_void)]),
    "mzhist_destroy": (None, [c_void]),
    "mzhist_last_error": (ctypes.c_char_p, [c_void]),
    "mzhist_begin": (ctypes.c_int, [c_void, c_f32_p, c_i32_p]),
    "mzhist_file": (ctypes.c_int, [c_void, c_void, c_i32_p]),
    "mzhist_finished": (ctypes.c_int, [c_void] + [ctypes.POINTER(c_void)] * 8 + [c_i32_p]),
    "mzhist_lengths": (c_void, [c_void]),
    "mzhist_rows": (ctypes.c_int, [c_void] * 8 + [ctypes.c_int32]),
    "mzreplay_create": (ctypes.c_int, [c_void, ctypes.POINTER(c_void)]),
    "mzreplay_destroy": (None, [c_void]),
    "mzreplay_last_error": (ctypes.c_char_p, [c_void]),
    "mzreplay_add_games": (ctypes.c_int, [c_void, ctypes.c_int32, c_i32_p, c_i32_p, c_f32_p, c_i32_p, c_f64_p, c_i32_p,
                                          c_f64_p, c_f64_p, c_f32_p, c_f32_p, c_void]),
    "mzreplay_make_batch": (ctypes.c_int, [c_void, ctypes.c_int32, c_i32_p, c_i32_p, c_i32_p, c_void, c_void, c_void,
                                           c_void, c_void, c_void, c_void]),
    "mzreplay_game_observations": (ctypes.c_int, [c_void, ctypes.c_int32, ctypes.c_int32, c_void, c_void]),
    "mzreplay_set_reanalysed": (ctypes.c_int, [c_void, ctypes.c_int32, c_void, ctypes.c_int32, c_void]),
    "mzreplay_device_bytes": (ctypes.c_int64, [c_void]),
    "mzenv_create": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_u32_p, ctypes.POINTER(c_void)]),
    "mzenv_destroy": (None, [c_void]),
    "mzenv_last_error": (ctypes.c_char_p, [c_void]),
    "mzenv_shape": (ctypes.c_int, [c_void, c_i32_p, c_i32_p, c_i32_p]),
    "mzenv_reset": (ctypes.c_int, [c_void, c_void, c_void]),
    "mzenv_step": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void]),
    "mzenv_observe": (ctypes.c_int, [c_void, c_void, c_void, c_void, c_void, c_void]),
    "mzmcts_rng_create": (c_void, [ctypes.c_uint32]),
    "mzmcts_rng_destroy": (None, [c_void]),
    "mzmcts_rng_reseed": (None, [c_void, ctypes.c_uint32]),
    "mzmcts_rng_next_u32": (ctypes.c_uint32, [c_void]),
    "mzmcts_rng_random_sample": (ctypes.c_double, [c_void]),
    "mzmcts_rng_choice": (ctypes.c_uint32, [c_void, ctypes.c_uint32]),
    "mzmcts_rng_choice_p_many": (None, [c_void, c_f64_p, ctypes.c_int32, ctypes.c_int32, c_i32_p]),
    "mzmcts_rng_choice_priorities": (ctypes.c_int32, [c_void, c_f32_p, ctypes.c_int32, c_f32_p]),
    "mzmcts_rng_choice_p": (ctypes.c_int32, [c_void, c_f64_p, ctypes.c_int32]),
    "mzmcts_rng_dirichlet": (None, [c_void, ctypes.c_double, ctypes.c_int32, c_f64_p]),
    "mzmcts_set_device_noise": (ctypes.c_int, [c_void, ctypes.c_int32]),
    "mzmcts_get_noise": (ctypes.c_int, [c_void, c_f64_p]),
    "mzmcts_device_libm": (ctypes.c_int, [c_f64_p, c_f64_p, ctypes.c_int64, c_f64_p, c_f64_p]),
    "mzmcts_device_dirichlet": (ctypes.c_int, [c_u32_p, ctypes.c_int32, ctypes.c_double, ctypes.c_int32, ctypes.c_int32,
                                               c_f64_p, c_u32_p]),
    "mzmcts_rng_export": (None, [c_void, c_u32_p, c_i32_p, c_i32_p, c_f64_p]),
    "mzmcts_rng_import": (None, [c_void, c_u32_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double]),
    "mzmcts_rng_select_action": (ctypes.c_int32, [c_void, c_i32_p, ctypes.c_int32, ctypes.c_double]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load libmzmcts.so and bind every prototype.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The MCTS engine has no Python fallback.")
    # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  It must be the one already
    # mapped when libmzmcts.so resolves its dependency: two HIP runtimes in one process do not see the
    # same devices / streams.  Importing torch first guarantees that.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here == header / library out of sync
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mzmcts_abi_version() != ABI_VERSION:
        raise NativeLibraryError("libmzmcts.so ABI version mismatch; rebuild the extension")
    _lib = lib
    return lib


def ptr(arr, typ):
    """ctypes pointer to a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(typ)


def check(lib, engine, rc):
    """Translate a C-ABI status into the reference's exception types / messages."""
    if rc == 0:
        return
    msg = lib.mzmcts_last_error(engine)
    msg = msg.decode() if msg else f"mzmcts error {rc}"
    if rc in (ERR_EMPTY_LEGAL, ERR_LEGAL_RANGE):
        raise AssertionError(msg)
    if rc == ERR_PLAYERS:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


class MzHeadDesc(ctypes.Structure):
    """mzmcts_head_desc (include/mzmcts.h): one reward / value / policy head of the residual networks."""
    _fields_ = [("conv_w", c_void), ("conv_b", c_void), ("fc1_w", c_void), ("fc1_b", c_void), ("fc2_w", c_void),
                ("fc2_b", c_void), ("channels", ctypes.c_int32), ("plane", ctypes.c_int32), ("reduced", ctypes.c_int32),
                ("hidden", ctypes.c_int32), ("outputs", ctypes.c_int32)]


class MzTowerHead(ctypes.Structure):
    """mzmcts_tower_head (include/mzmcts.h): a head computed inside a tower launch."""
    _fields_ = [("head", MzHeadDesc), ("out", c_void), ("layer", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class MzHistMoves(ctypes.Structure):
    _fields_ = [("n_moves", ctypes.c_int32), ("num_simulations", ctypes.c_int32), ("moves_done", c_void),
                ("actions", c_void), ("actions_stride", ctypes.c_int64), ("visits", c_void),
                ("visits_stride", ctypes.c_int64), ("root_value_sum", c_void), ("root_value_sum_stride", ctypes.c_int64),
                ("legal", c_void), ("num_legal", c_void), ("rewards", c_void), ("done", c_void), ("obs_after", c_void),
                ("obs_next", c_void), ("to_play_after", c_void), ("to_play_next", c_void),
                ("legal_stride", ctypes.c_int64), ("num_legal_stride", ctypes.c_int64)]


class HostRng:
    """numpy legacy RandomState clone on the host (stand-alone stream of the C library)."""

    def __init__(self, seed=0):
        self._lib = load()
        self._h = self._lib.mzmcts_rng_create(int(seed) & 0xFFFFFFFF)

    def __del__(self):
        try:
            self._lib.mzmcts_rng_destroy(self._h)
        except Exception:
            pass

    def seed(self, seed):
        self._lib.mzmcts_rng_reseed(self._h, int(seed) & 0xFFFFFFFF)

    def next_u32(self):
        return self._lib.mzmcts_rng_next_u32(self._h)

    def random_sample(self):
        return self._lib.mzmcts_rng_random_sample(self._h)

    def choice(self, n):
        return self._lib.mzmcts_rng_choice(self._h, n)

    def choice_p(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        return self._lib.mzmcts_rng_choice_p(self._h, ptr(p, c_f64_p), len(p))

    def choice_p_many(self, p, count):
        p = np.ascontiguousarray(p, dtype=np.float64)
        out = np.zeros(count, dtype=np.int32)
        self._lib.mzmcts_rng_choice_p_many(self._h, ptr(p, c_f64_p), len(p), int(count), ptr(out, c_i32_p))
        return out

    def choice_priorities(self, priorities):
        """(index, float32 probability) of choice(n, p=priorities / sum(priorities)) with float32 probabilities."""
        pri = np.ascontiguousarray(priorities, dtype=np.float32)
        prob = ctypes.c_float()
        idx = self._lib.mzmcts_rng_choice_priorities(self._h, ptr(pri, c_f32_p), len(pri), ctypes.byref(prob))
        return idx, np.float32(prob.value)

    def dirichlet(self, alpha, k):
        out = np.zeros(k, dtype=np.float64)
        self._lib.mzmcts_rng_dirichlet(self._h, float(alpha), k, ptr(out, c_f64_p))
        return out

    def select_action(self, visits, temperature):
        v = np.ascontiguousarray(visits, dtype=np.int32)
        return self._lib.mzmcts_rng_select_action(self._h, ptr(v, c_i32_p), len(v), float(temperature))

    def get_state(self):
        key = np.zeros(624, dtype=np.uint32)
        pos, hg, g = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_double()
        self._lib.mzmcts_rng_export(self._h, ptr(key, c_u32_p), ctypes.byref(pos), ctypes.byref(hg),
                                    ctypes.byref(g))
        return ("MT19937", key, pos.value, hg.value, g.value)

    def set_state(self, state):
        key = np.ascontiguousarray(state[1], dtype=np.uint32)
        self._lib.mzmcts_rng_import(self._h, ptr(key, c_u32_p), int(state[2]), int(state[3]),
                                    float(state[4]))


class MzTowerLayer(ctypes.Structure):
    """include/mzmcts.h mzmcts_tower_layer"""
    _fields_ = [("packed", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
                ("const_table", ctypes.c_void_p), ("export_raw", ctypes.c_void_p), ("export_unit", ctypes.c_void_p),
                ("cin", ctypes.c_int32),
                ("relu", ctypes.c_int32), ("skip", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("gate", ctypes.c_void_p)]


def exact_inverse_temperature(temperature):
    """k when 1 / temperature is an integer k in 1..4 (the temperatures the device sampler handles exactly:
    csrc/kernel_common.h exact_inverse_temperature), else 0."""
    if temperature in (0, float("inf")) or temperature != temperature:
        return 0
    inv = 1.0 / float(temperature)
    k = int(inv)
    return k if inv == k and 1 <= k <= 4 else 0

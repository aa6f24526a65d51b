"""Inference half of the reference's networks (reference models.py), run through PyTorch-ROCm.

Same factory (`MuZeroNetwork(config)`), same method names (`initial_inference`,
`recurrent_inference`, `get_weights`, `set_weights`) and -- so that reference checkpoints and
trainer weights load unchanged -- the same state-dict keys, including the `.module.` level the
reference gets from wrapping every sub-network in `torch.nn.DataParallel` (models.py:98-126,
482-516).  Here `.module.` comes from a transparent holder instead: one process drives one GPU, so
there is nothing to scatter.

Differences that matter on the GPU hot path (values are unchanged):
  * no host synchronisation: the `scale[scale < 1e-5] += 1e-5` mask-assignment (a device->host sync
    via nonzero) is a `torch.where`, so a whole simulation can be captured into a hipGraph;
  * `recurrent_inference(..., out_state=...)` writes the normalised next state straight into the
    engine's hidden-state pool slab, saving a copy per simulation;
  * `support_to_scalar` keeps its support vector resident instead of rebuilding it per call.
"""
import ctypes
import math
import os

import torch

from . import _native


class MuZeroNetwork:
    """Factory, reference models.py:7-41."""

    def __new__(cls, config):
        if config.network == "fullyconnected":
            return MuZeroFullyConnectedNetwork(
                config.observation_shape, config.stacked_observations, len(config.action_space),
                config.encoding_size, config.fc_reward_layers, config.fc_value_layers,
                config.fc_policy_layers, config.fc_representation_layers, config.fc_dynamics_layers,
                config.support_size)
        if config.network == "resnet":
            return MuZeroResidualNetwork(
                config.observation_shape, config.stacked_observations, len(config.action_space),
                config.blocks, config.channels, config.reduced_channels_reward,
                config.reduced_channels_value, config.reduced_channels_policy,
                config.resnet_fc_reward_layers, config.resnet_fc_value_layers,
                config.resnet_fc_policy_layers, config.support_size, config.downsample)
        raise NotImplementedError('The network parameter should be "fullyconnected" or "resnet".')


def dict_to_cpu(dictionary):
    """reference models.py:44-53"""
    out = {}
    for key, value in dictionary.items():
        if isinstance(value, torch.Tensor):
            out[key] = value.cpu()
        elif isinstance(value, dict):
            out[key] = dict_to_cpu(value)
        else:
            out[key] = value
    return out


class _Replica(torch.nn.Module):
    """Holds a sub-network under the attribute name `module`, which is all DataParallel contributes
    to the reference's state-dict keys."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args):
        return self.module(*args)


def mlp(input_size, layer_sizes, output_size, output_activation=torch.nn.Identity,
        activation=torch.nn.ELU):
    """Linear/ELU stack with an identity head; Sequential indices 0,2,4.. are the Linear layers
    (reference models.py:626-638)."""
    widths = [input_size, *layer_sizes, output_size]
    stack = []
    last = len(widths) - 2
    for i, (fan_in, fan_out) in enumerate(zip(widths[:-1], widths[1:])):
        stack.append(torch.nn.Linear(fan_in, fan_out))
        stack.append(activation() if i < last else output_activation())
    return torch.nn.Sequential(*stack)


def _unit_rescale(x, dims):
    """Min-max rescale to [0,1] over `dims` (reference models.py:137-145, 525-549)."""
    low = x.amin(dim=dims, keepdim=True)
    high = x.amax(dim=dims, keepdim=True)
    span = high - low
    span = torch.where(span < 1e-5, span + 1e-5, span)
    return x - low, span


def board_rescale(raw, out=None):
    """Min-max rescale of every (sample, channel) board plane of an NCHW tensor to [0, 1] (reference
    models.py:525-549, 586-602), into `out` if given.  Inference on the GPU: one HIP launch
    (include/mzmcts.h mzmcts_unit_rescale), bit-identical to the torch expression below, which training, autograd
    and CPU tensors take."""
    plane = raw.shape[2] * raw.shape[3]
    native = (raw.is_cuda and raw.dtype == torch.float32 and plane <= 128 and not (torch.is_grad_enabled() and raw.requires_grad)
              and (out is None or (out.is_contiguous() and out.dtype == torch.float32 and out.shape == raw.shape)))
    if not native:
        shifted, span = _unit_rescale(raw, (2, 3))
        return torch.div(shifted, span, out=out)
    raw = raw.contiguous()
    if out is None:
        out = torch.empty_like(raw)
    with torch.cuda.device(raw.device):
        rc = _native.load().mzmcts_unit_rescale(raw.data_ptr(), out.data_ptr(), raw.shape[0] * raw.shape[1], plane,
                                                torch.cuda.current_stream(raw.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"mzmcts_unit_rescale failed ({rc}) on a tensor of shape {tuple(raw.shape)}")
    return out


def state_action_planes(state, action, action_space_size):
    """The dynamics input: the hidden state's planes plus one plane of action / action_space_size (reference
    models.py:553-568).  Inference on the GPU with an int64 [B, 1] action batch (what select hands over): one HIP
    launch (include/mzmcts.h mzmcts_state_action_planes), bit-identical to the torch expression below."""
    b, c, h, w = state.shape
    native = (state.is_cuda and state.dtype == torch.float32 and action.dtype == torch.int64 and action.is_cuda
              and action.numel() == b and b <= 65535 and not (torch.is_grad_enabled() and state.requires_grad))
    if not native:
        plane = (action.to(state.dtype) / action_space_size)[:, :, None, None]
        return torch.cat((state, plane.expand(b, 1, h, w)), dim=1)
    state, action = state.contiguous(), action.contiguous()
    out = torch.empty((b, c + 1, h, w), dtype=torch.float32, device=state.device)
    with torch.cuda.device(state.device):
        rc = _native.load().mzmcts_state_action_planes(state.data_ptr(), action.data_ptr(), out.data_ptr(), b, c, h * w,
                                                       action_space_size, torch.cuda.current_stream(state.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"mzmcts_state_action_planes failed ({rc}) on a tensor of shape {tuple(state.shape)}")
    return out


class AbstractNetwork(torch.nn.Module):
    """reference models.py:56-73"""

    def initial_inference(self, observation):
        raise NotImplementedError

    def recurrent_inference(self, encoded_state, action, out_state=None):
        raise NotImplementedError

    def get_weights(self):
        return dict_to_cpu(self.state_dict())

    def set_weights(self, weights):
        self.load_state_dict(weights)
        self.refresh_inference_constants()

    def refresh_inference_constants(self):
        """Recompute cached inference constants (folded batch-norm scale / shift, expanded board convolutions) after the weights
        changed behind the modules' backs, e.g. by a broadcast into aliased flat storage."""
        for module in self.modules():
            if isinstance(module, (BatchNorm2d, BoardConv2d)):
                module.refold()

    def _zero_reward_logits(self, batch, device):
        # log(one_hot(centre)): -inf everywhere, 0 at the centre (reference models.py:176-183)
        cached = getattr(self, "_zero_reward_cache", None)
        if cached is None or cached.device != device:
            row = torch.full((1, self.full_support_size), float("-inf"), device=device)
            row[0, self.full_support_size // 2] = 0.0
            self._zero_reward_cache = cached = row
        return cached.expand(batch, -1)


# ------------------------------------------------------------------------------------------------
# Fully connected (reference models.py:80-195)
# ------------------------------------------------------------------------------------------------
class MuZeroFullyConnectedNetwork(AbstractNetwork):
    def __init__(self, observation_shape, stacked_observations, action_space_size, encoding_size,
                 fc_reward_layers, fc_value_layers, fc_policy_layers, fc_representation_layers,
                 fc_dynamics_layers, support_size):
        super().__init__()
        self.action_space_size = action_space_size
        self.full_support_size = 2 * support_size + 1
        c, h, w = observation_shape
        flat_obs = c * h * w * (stacked_observations + 1) + stacked_observations * h * w
        self.representation_network = _Replica(mlp(flat_obs, fc_representation_layers, encoding_size))
        self.dynamics_encoded_state_network = _Replica(
            mlp(encoding_size + action_space_size, fc_dynamics_layers, encoding_size))
        self.dynamics_reward_network = _Replica(
            mlp(encoding_size, fc_reward_layers, self.full_support_size))
        self.prediction_policy_network = _Replica(
            mlp(encoding_size, fc_policy_layers, action_space_size))
        self.prediction_value_network = _Replica(
            mlp(encoding_size, fc_value_layers, self.full_support_size))

    def prediction(self, encoded_state):
        return (self.prediction_policy_network(encoded_state),
                self.prediction_value_network(encoded_state))

    def representation(self, observation):
        raw = self.representation_network(observation.reshape(observation.shape[0], -1))
        shifted, span = _unit_rescale(raw, 1)
        return shifted / span

    def dynamics(self, encoded_state, action, out_state=None):
        one_hot = (action.long() == torch.arange(self.action_space_size, device=action.device)
                   ).to(encoded_state.dtype)
        raw = self.dynamics_encoded_state_network(torch.cat((encoded_state, one_hot), dim=1))
        reward = self.dynamics_reward_network(raw)  # on the un-normalised state (models.py:157-159)
        shifted, span = _unit_rescale(raw, 1)
        return torch.div(shifted, span, out=out_state), reward

    def initial_inference(self, observation):
        encoded_state = self.representation(observation)
        policy_logits, value = self.prediction(encoded_state)
        reward = self._zero_reward_logits(observation.shape[0], observation.device)
        return value, reward, policy_logits, encoded_state

    def recurrent_inference(self, encoded_state, action, out_state=None):
        next_state, reward = self.dynamics(encoded_state, action, out_state)
        policy_logits, value = self.prediction(next_state)
        return value, reward, policy_logits, next_state


# ------------------------------------------------------------------------------------------------
# Residual (reference models.py:206-619)
# ------------------------------------------------------------------------------------------------
class BatchNorm2d(torch.nn.BatchNorm2d):
    """torch.nn.BatchNorm2d (same parameters, buffers and state-dict keys) whose eval-mode forward is the
    folded affine y = x * scale + shift: one element-wise kernel.  On MI355X MIOpen's inference kernel
    (`MIOpenBatchNormFwdInferSpatialEst`, also reached through torch.batch_norm) needs ~1 ms for a
    [4096, 16, 3, 3] board batch -- 94 % of a TicTacToe simulation step.

    scale / shift are cached and refreshed IN PLACE (so a captured hipGraph keeps reading the same
    addresses) whenever the parameters change: detected through their version counters, or signalled by
    `refold()` after a weight broadcast into aliased storage (weights.FlatWeights)."""

    def _fold_key(self):
        return (self.weight._version, self.bias._version, self.running_mean._version,
                self.running_var._version, self.weight.data_ptr(), self.weight.device)

    def refold(self):
        with torch.no_grad():
            if getattr(self, "_scale", None) is None or self._scale.device != self.weight.device:
                self._scale = torch.empty_like(self.weight)
                self._shift = torch.empty_like(self.weight)
            torch.mul(self.weight, torch.rsqrt(self.running_var + self.eps), out=self._scale)
            torch.sub(self.bias, self.running_mean * self._scale, out=self._shift)
        self._folded = self._fold_key()

    def folded(self):
        """(scale, shift) of the eval-mode affine, refreshed if the parameters changed."""
        if getattr(self, "_folded", None) != self._fold_key():
            self.refold()
        return self._scale, self._shift

    def forward(self, x):
        if self.training:
            return super().forward(x)
        scale, shift = self.folded()
        return torch.addcmul(shift.view(1, -1, 1, 1), x, scale.view(1, -1, 1, 1))


def conv_epilogue(x, bn, residual=None):
    """relu(bn(x) [+ residual]) after a convolution (reference models.py:215-237).  Inference on the GPU: one HIP
    launch (include/mzmcts.h mzmcts_affine_act) instead of torch's three element-wise ones, same operations in
    the same order; training, autograd and CPU tensors take the torch expression."""
    native = (x.is_cuda and not bn.training and x.dtype == torch.float32 and x.dim() == 4
              and not (torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad))))
    if native:
        scale, shift = bn.folded()
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        # the kernel moves 16 bytes per lane: fresh torch allocations are aligned, odd views of them need not be
        native = x.data_ptr() % 16 == 0 and (residual is None or residual.data_ptr() % 16 == 0)
    if not native:
        y = bn(x)
        if residual is not None:
            y = y + residual
        return torch.relu(y)
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _native.load().mzmcts_affine_act(
            x.data_ptr(), scale.data_ptr(), shift.data_ptr(), residual.data_ptr() if residual is not None else None,
            out.data_ptr(), x.numel(), x.shape[1], x.shape[2] * x.shape[3], 1,
            torch.cuda.current_stream(x.device).cuda_stream)
    if rc != 0:
        raise RuntimeError(f"mzmcts_affine_act failed ({rc}) on a tensor of shape {tuple(x.shape)}")
    return out


class PointwiseConv2d(torch.nn.Conv2d):
    """1x1 convolution (same parameters / state-dict keys as torch.nn.Conv2d(c_in, c_out, 1)) evaluated as
    ONE GEMM over all (sample, position) rows.  MIOpen lowers these tiny-board 1x1 convolutions to a
    per-image im2col + GEMM loop (4096 launches per call at 4096 envs), which dominated a TicTacToe
    simulation step."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 1)

    def forward(self, x):
        b, c, h, w = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(b * h * w, c)
        out = torch.addmm(self.bias, rows, self.weight.view(self.out_channels, c).t())
        return out.view(b, h, w, self.out_channels).permute(0, 3, 1, 2)


def _head_is_native(x, conv, fc):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not torch.is_grad_enabled()
            and len(fc) == 4 and isinstance(fc[0], torch.nn.Linear) and isinstance(fc[1], torch.nn.ELU)
            and fc[1].alpha == 1.0 and isinstance(fc[2], torch.nn.Linear) and isinstance(fc[3], torch.nn.Identity)
            and conv.bias is not None)


def conv_heads(x, heads):
    """[fc(conv(x).reshape(-1, flat_size)) for (conv, fc, flat_size) in heads]: reward / value / policy heads that
    read the same tensor (reference models.py:467-480, 500-522).  Inference on the GPU with the usual
    one-hidden-layer MLPs: ONE HIP launch for up to two heads (include/mzmcts.h mzmcts_conv_heads) reading the
    modules' own parameters; anything else (training, autograd, CPU, deeper MLPs, heads too large for LDS)
    evaluates the torch modules."""
    if 1 <= len(heads) <= 2 and all(_head_is_native(x, conv, fc) for conv, fc, _ in heads):
        x = x.contiguous()
        b, c, h, w = x.shape
        descs = (_native.MzHeadDesc * len(heads))(*[
            _native.MzHeadDesc(conv.weight.data_ptr(), conv.bias.data_ptr(), fc[0].weight.data_ptr(), fc[0].bias.data_ptr(),
                               fc[2].weight.data_ptr(), fc[2].bias.data_ptr(), c, h * w, conv.out_channels,
                               fc[0].out_features, fc[2].out_features) for conv, fc, _ in heads])
        outs = [torch.empty((b, fc[2].out_features), dtype=torch.float32, device=x.device) for _, fc, _ in heads]
        pointers = (ctypes.c_void_p * len(heads))(*[o.data_ptr() for o in outs])
        with torch.cuda.device(x.device):
            rc = _native.load().mzmcts_conv_heads(x.data_ptr(), ctypes.addressof(descs), len(heads), ctypes.addressof(pointers),
                                                  b, torch.cuda.current_stream(x.device).cuda_stream)
        if rc == 0:
            return outs
        if rc != -1:                                   # -1: a head does not fit in LDS -> torch modules
            raise RuntimeError(f"mzmcts_conv_heads failed ({rc}) on a tensor of shape {tuple(x.shape)}")
    return [fc(conv(x).reshape(-1, flat_size)) for conv, fc, flat_size in heads]


def conv_heads_multi(inputs_and_heads):
    """[fc(conv(x).reshape(-1, flat_size)) for x, (conv, fc, flat_size) in inputs_and_heads] with up to three heads that
    read DIFFERENT tensors of one shape in ONE HIP launch (include/mzmcts.h mzmcts_conv_heads_multi): the reward head on
    the dynamics network's raw output next to the value / policy heads on the prediction network's features."""
    xs = [x for x, _ in inputs_and_heads]
    heads = [h for _, h in inputs_and_heads]
    if (1 <= len(heads) <= 3 and all(_head_is_native(x, conv, fc) for x, (conv, fc, _) in inputs_and_heads)
            and all(x.shape == xs[0].shape for x in xs)):
        xs = [x.contiguous() for x in xs]
        b, c, h, w = xs[0].shape
        descs = (_native.MzHeadDesc * len(heads))(*[
            _native.MzHeadDesc(conv.weight.data_ptr(), conv.bias.data_ptr(), fc[0].weight.data_ptr(), fc[0].bias.data_ptr(),
                               fc[2].weight.data_ptr(), fc[2].bias.data_ptr(), c, h * w, conv.out_channels,
                               fc[0].out_features, fc[2].out_features) for conv, fc, _ in heads])
        outs = [torch.empty((b, fc[2].out_features), dtype=torch.float32, device=xs[0].device) for _, fc, _ in heads]
        x_ptrs = (ctypes.c_void_p * len(heads))(*[x.data_ptr() for x in xs])
        pointers = (ctypes.c_void_p * len(heads))(*[o.data_ptr() for o in outs])
        with torch.cuda.device(xs[0].device):
            rc = _native.load().mzmcts_conv_heads_multi(ctypes.addressof(x_ptrs), ctypes.addressof(descs), len(heads),
                                                        ctypes.addressof(pointers), b,
                                                        torch.cuda.current_stream(xs[0].device).cuda_stream)
        if rc == 0:
            return outs
        if rc != -1:
            raise RuntimeError(f"mzmcts_conv_heads_multi failed ({rc}) on tensors of shape {tuple(xs[0].shape)}")
    return [fc(conv(x).reshape(-1, flat_size)) for x, (conv, fc, flat_size) in inputs_and_heads]


def conv_head(x, conv, fc, flat_size):
    return conv_heads(x, [(conv, fc, flat_size)])[0]


class BoardConv2d(torch.nn.Conv2d):
    """torch.nn.Conv2d(c_in, c_out, 3, padding=1, bias=False) (same parameter and state-dict key) whose inference
    forward on small boards (expanded matrix of at most `DENSE_MAX_ELEMENTS` entries) is ONE GEMM: on a 3x3 board a padded 3x3 convolution
    is a dense linear map of the flattened [c_in * 9] board onto [c_out * 9], and NCHW tensors flatten to exactly
    those rows for free.  MIOpen picks its fp32 Winograd kernel for these shapes (16.5 us per call at 4096
    TicTacToe boards, 44 % of a simulation step); the GEMM over the expanded weight matrix takes a third of
    that and sums the same products directly.  The expansion multiplies the arithmetic by (board positions / 9),
    so it pays on 3x3 (+21 % simulations/s) and 6x6 x 16 channels (+19 %), not on Connect4's 6x7 x 64 (-60 %).

    The expanded matrix is cached and refreshed IN PLACE (a captured hipGraph keeps reading the same addresses)
    when the weight changes: detected through its version counter, or signalled by `refold()` after a weight
    broadcast into aliased storage."""
    DENSE_MAX_ELEMENTS = 1 << 19   # of the expanded matrix (2 MB): 16 channels on 6x6 yes, 64 channels on 6x7 no

    # ---- matrix-core path (include/mzmcts.h mzmcts_board_conv3x3) ----------------------------------------------
    def takes_mfma_path(self, x):
        """Inference on the GPU on a board the HIP implicit-GEMM kernel covers (and the dense GEMM does not)."""
        mode = os.environ.get("MZ_BOARD_CONV", "auto")      # auto | all (also where the dense GEMM applies) | off
        if mode == "off" or self.training or torch.is_grad_enabled() or not x.is_cuda or x.dtype != torch.float32 \
                or x.dim() != 4 or self.stride != (1, 1) or self.kernel_size != (3, 3):
            return False
        if mode != "all" and self.takes_dense_path(x):
            return False
        return bool(_native.load().mzmcts_board_conv_supported(self.in_channels, self.out_channels, x.shape[2], x.shape[3]))

    def packed(self):
        """The weight rearranged k-major for the kernel; same buffer for the module's lifetime (a captured hipGraph
        keeps reading it), refilled when the weight changes (version counter) or on refold()."""
        lib = _native.load()
        buf = self.__dict__.get("_packed")
        if buf is None or buf.device != self.weight.device:
            n = lib.mzmcts_board_conv_packed_floats(self.in_channels, self.out_channels)
            buf = self.__dict__["_packed"] = torch.empty(n, dtype=torch.float32, device=self.weight.device)
            self._packed_version = None
        if getattr(self, "_packed_version", None) != self._dense_key():
            self._repack()
        return buf

    def _repack(self):
        buf = self.__dict__["_packed"]
        with torch.cuda.device(buf.device):
            rc = _native.load().mzmcts_board_conv_pack(self.weight.data_ptr(), buf.data_ptr(), self.in_channels,
                                                       self.out_channels, torch.cuda.current_stream(buf.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mzmcts_board_conv_pack failed ({rc})")
        self._packed_version = self._dense_key()

    def packed_split(self, const_plane, h, w):
        """(weights as two fp16 halves in the split tower's layout, table of the constant last input plane or None);
        same buffers for the module's lifetime, refilled when the weight changes or on refold()."""
        lib = _native.load()
        cache = self.__dict__.setdefault("_packed_split", {})
        key = (bool(const_plane), h, w, str(self.weight.device))
        if key not in cache:
            n = lib.mzmcts_board_conv_split_halfs(self.in_channels - (1 if const_plane else 0), self.out_channels)
            cache[key] = [torch.empty(n, dtype=torch.float16, device=self.weight.device),
                          torch.empty(self.out_channels * h * w, dtype=torch.float32, device=self.weight.device)
                          if const_plane else None, None]
        entry = cache[key]
        if entry[2] != self._dense_key():
            self._repack_split(key)
        return entry[0], entry[1]

    def _repack_split(self, key):
        const_plane, h, w, _ = key
        halves, table, _ = self.__dict__["_packed_split"][key]
        with torch.cuda.device(halves.device):
            rc = _native.load().mzmcts_board_conv_pack_split(
                self.weight.data_ptr(), halves.data_ptr(), table.data_ptr() if table is not None else None,
                self.in_channels, self.out_channels, 1 if const_plane else 0, h, w,
                torch.cuda.current_stream(halves.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mzmcts_board_conv_pack_split failed ({rc})")
        self.__dict__["_packed_split"][key][2] = self._dense_key()

    def fused(self, x, bn, residual=None, relu=True):
        """relu(bn(conv(x)) [+ residual]) in one launch on the matrix cores."""
        scale, shift = bn.folded()
        x = x.contiguous()
        if residual is not None:
            residual = residual.contiguous()
        b, _, h, w = x.shape
        out = torch.empty((b, self.out_channels, h, w), dtype=torch.float32, device=x.device)
        packed = self.packed()
        with torch.cuda.device(x.device):
            rc = _native.load().mzmcts_board_conv3x3(
                x.data_ptr(), packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                residual.data_ptr() if residual is not None else None, out.data_ptr(), b, self.in_channels,
                self.out_channels, h, w, 1 if relu else 0, torch.cuda.current_stream(x.device).cuda_stream)
        if rc != 0:
            raise RuntimeError(f"mzmcts_board_conv3x3 failed ({rc}) on a tensor of shape {tuple(x.shape)}")
        return out

    def _expansion(self, h, w, device):
        """(index into weight.flatten(), 0/1 mask), both [c_in*h*w, c_out*h*w]: entry ((ci,y',x'), (co,y,x)) takes
        weight[co, ci, y'-y+1, x'-x+1] when that tap exists."""
        key = (h, w, str(device))
        cache = self.__dict__.setdefault("_expansions", {})
        if key not in cache:
            co, ci = self.out_channels, self.in_channels
            ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            ys, xs = ys.reshape(-1), xs.reshape(-1)
            ky = ys[:, None] - ys[None, :] + 1                    # [source position, output position]
            kx = xs[:, None] - xs[None, :] + 1
            valid = (ky >= 0) & (ky <= 2) & (kx >= 0) & (kx <= 2)
            tap = (ky.clamp(0, 2) * 3 + kx.clamp(0, 2))            # [hw, hw]
            index = (torch.arange(co)[None, None, :, None] * ci + torch.arange(ci)[:, None, None, None]) * 9 \
                + tap[None, :, None, :]                            # [ci, hw', co, hw]
            mask = valid[None, :, None, :].expand(ci, h * w, co, h * w)
            cache[key] = (index.reshape(ci * h * w, co * h * w).to(device),
                          mask.reshape(ci * h * w, co * h * w).to(device=device, dtype=self.weight.dtype))
        return cache[key]

    def _dense_key(self):
        return (self.weight._version, self.weight.data_ptr(), self.weight.device)

    def refold(self):
        """Rebuild every cached expanded matrix from the current weight (and the batch norms folded into some of
        them), in place."""
        if self.__dict__.get("_packed") is not None:
            self._repack()
        for key in self.__dict__.get("_packed_split", {}):
            self._repack_split(key)
        with torch.no_grad():
            for (h, w, _), matrix in self.__dict__.get("_dense", {}).items():
                index, mask = self._expansion(h, w, self.weight.device)
                matrix.copy_(torch.take(self.weight, index))
                matrix.mul_(mask)
            for (h, w, _, _), (matrix, bias, bn) in self.__dict__.get("_dense_bn", {}).items():
                bn.refold()  # (explicitly: a broadcast into aliased storage moves no version counter)
                scale, shift = bn.folded()
                torch.mul(self.dense_matrix(h, w), scale.repeat_interleave(h * w)[None, :], out=matrix)
                bias.copy_(shift.repeat_interleave(h * w))
        self._dense_version = self._dense_key()
        self._dense_bn_version = self._dense_bn_key()

    def _dense_bn_key(self):
        return tuple(bn._fold_key() for (_, _, bn) in self.__dict__.get("_dense_bn", {}).values())

    def dense_matrix(self, h, w):
        return self.__dict__["_dense"][(h, w, str(self.weight.device))]

    def dense(self, h, w):
        cache = self.__dict__.setdefault("_dense", {})
        key = (h, w, str(self.weight.device))
        if key not in cache:
            index, _ = self._expansion(h, w, self.weight.device)
            cache[key] = torch.empty(index.shape, dtype=self.weight.dtype, device=self.weight.device)
            self._dense_version = None
        if getattr(self, "_dense_version", None) != self._dense_key():
            self.refold()
        return cache[key]

    def dense_with(self, bn, h, w):
        """(expanded matrix with `bn`'s eval-mode scale folded into its columns, bias row): relu(x @ m + bias) is
        relu(bn(conv(x))) in one GEMM with a fused epilogue."""
        self.dense(h, w)
        cache = self.__dict__.setdefault("_dense_bn", {})
        key = (h, w, str(self.weight.device), id(bn))
        if key not in cache:
            matrix = torch.empty_like(self.dense_matrix(h, w))
            cache[key] = (matrix, torch.empty(matrix.shape[1], dtype=matrix.dtype, device=matrix.device), bn)
            self._dense_bn_version = None
        if getattr(self, "_dense_bn_version", None) != self._dense_bn_key() or self._dense_version != self._dense_key():
            self.refold()
        return cache[key][:2]

    def takes_dense_path(self, x):
        h, w = x.shape[2], x.shape[3]
        return (not self.training and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and self.in_channels * self.out_channels * (h * w) ** 2 <= self.DENSE_MAX_ELEMENTS
                and self.stride == (1, 1) and not torch.is_grad_enabled())

    def forward(self, x):
        if not self.takes_dense_path(x):
            return super().forward(x)
        b, _, h, w = x.shape
        return torch.mm(x.reshape(b, -1), self.dense(h, w)).view(b, self.out_channels, h, w)


def conv_bn_relu(conv, bn, x):
    """relu(bn(conv(x))) (reference models.py:215-225, 318-330).  On the dense small-board path in inference: ONE
    GEMM -- batch-norm scale folded into the expanded matrix, shift and ReLU in the GEMM's epilogue."""
    if isinstance(conv, BoardConv2d) and not bn.training and conv.takes_mfma_path(x):
        return conv.fused(x, bn)
    if isinstance(conv, BoardConv2d) and conv.takes_dense_path(x) and not bn.training:
        b, _, h, w = x.shape
        matrix, bias = conv.dense_with(bn, h, w)
        return torch._addmm_activation(bias, x.reshape(b, -1), matrix).view(b, conv.out_channels, h, w)
    return conv_epilogue(conv(x), bn)


def conv3x3(in_channels, out_channels, stride=1):
    return BoardConv2d(in_channels, out_channels, kernel_size=3, stride=stride, padding=1, bias=False)


class ResidualBlock(torch.nn.Module):
    def __init__(self, num_channels, stride=1):
        super().__init__()
        self.conv1 = conv3x3(num_channels, num_channels, stride)
        self.bn1 = BatchNorm2d(num_channels)
        self.conv2 = conv3x3(num_channels, num_channels)
        self.bn2 = BatchNorm2d(num_channels)

    def forward(self, x):
        y = conv_bn_relu(self.conv1, self.bn1, x)
        if not self.bn2.training and self.conv2.takes_mfma_path(y):
            return self.conv2.fused(y, self.bn2, residual=x)
        return conv_epilogue(self.conv2(y), self.bn2, residual=x)


def _tower(channels, count):
    return torch.nn.ModuleList([ResidualBlock(channels) for _ in range(count)])


class DownSample(torch.nn.Module):
    """Strided residual down-sampler (reference models.py:233-275)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        half = out_channels // 2
        self.conv1 = torch.nn.Conv2d(in_channels, half, kernel_size=3, stride=2, padding=1, bias=False)
        self.resblocks1 = _tower(half, 2)
        self.conv2 = torch.nn.Conv2d(half, out_channels, kernel_size=3, stride=2, padding=1, bias=False)
        self.resblocks2 = _tower(out_channels, 3)
        self.pooling1 = torch.nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.resblocks3 = _tower(out_channels, 3)
        self.pooling2 = torch.nn.AvgPool2d(kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        stages = ((self.conv1, self.resblocks1), (self.conv2, self.resblocks2),
                  (self.pooling1, self.resblocks3))
        for head, tower in stages:
            x = head(x)
            for block in tower:
                x = block(x)
        return self.pooling2(x)


class DownsampleCNN(torch.nn.Module):
    """Light convolutional down-sampler (reference models.py:278-297)."""

    def __init__(self, in_channels, out_channels, h_w):
        super().__init__()
        mid = (in_channels + out_channels) // 2
        self.features = torch.nn.Sequential(
            torch.nn.Conv2d(in_channels, mid, kernel_size=h_w[0] * 2, stride=4, padding=2),
            torch.nn.ReLU(inplace=True),
            torch.nn.MaxPool2d(kernel_size=3, stride=2),
            torch.nn.Conv2d(mid, out_channels, kernel_size=5, padding=2),
            torch.nn.ReLU(inplace=True),
            torch.nn.MaxPool2d(kernel_size=3, stride=2),
        )
        self.avgpool = torch.nn.AdaptiveAvgPool2d(h_w)

    def forward(self, x):
        if x.is_cuda and not self.training and not torch.is_grad_enabled():
            out = self._native_forward(x)
            if out is not None:
                return out
            # MIOpen's immediate mode falls back to a per-image im2col + GEMM loop for these shapes (12 x 12 kernel, stride
            # 4 on 84 x 84 frames): two launches per image, 57 % of the GPU time of an Atari-like search at 1024 envs
            # (profiles/r02_bench_atari84_kernel_stats.csv).  Letting it search once per shape picks a batched solver
            # (0.36 ms for 1024 frames).
            with torch.backends.cudnn.flags(enabled=True, benchmark=True):
                return self.avgpool(self.features(x))
        return self.avgpool(self.features(x))


    def _native_forward(self, x):
        """The seven layers in one HIP launch (include/mzmcts.h mzmcts_downsample_cnn; both convolutions on the fp32
        matrix cores).  None when the shape is not covered (anything but config #5's 4 x 84 x 84 frames) or
        MZ_DOWNSAMPLE=torch asks for the convolution library."""
        conv1, conv2 = self.features[0], self.features[3]
        if (os.environ.get("MZ_DOWNSAMPLE", "") == "torch" or x.dtype != torch.float32 or x.dim() != 4
                or conv1.stride != (4, 4) or conv1.padding != (2, 2) or conv2.kernel_size != (5, 5) or conv2.padding != (2, 2)
                or conv1.bias is None or conv2.bias is None):
            return None
        x = x.contiguous()
        h, w = self.avgpool.output_size
        out = torch.empty((x.shape[0], conv2.out_channels, h, w), dtype=torch.float32, device=x.device)
        w1, w2 = conv1.weight.contiguous(), conv2.weight.contiguous()
        with torch.cuda.device(x.device):
            rc = _native.load().mzmcts_downsample_cnn(
                x.data_ptr(), x.shape[0], x.shape[1], x.shape[2], x.shape[3], w1.data_ptr(), conv1.bias.data_ptr(),
                conv1.out_channels, conv1.kernel_size[0], w2.data_ptr(), conv2.bias.data_ptr(), conv2.out_channels, h, w,
                out.data_ptr(), torch.cuda.current_stream(x.device).cuda_stream)
        if rc == -1:        # MZMCTS_ERR_INVALID: not this launch's shape
            return None
        if rc != 0:
            raise RuntimeError(f"mzmcts_downsample_cnn failed ({rc}) on frames of shape {tuple(x.shape)}")
        return out


class RepresentationNetwork(torch.nn.Module):
    def __init__(self, observation_shape, stacked_observations, num_blocks, num_channels, downsample):
        super().__init__()
        in_planes = observation_shape[0] * (stacked_observations + 1) + stacked_observations
        self.downsample = downsample
        if downsample:
            if downsample == "resnet":
                self.downsample_net = DownSample(in_planes, num_channels)
            elif downsample == "CNN":
                self.downsample_net = DownsampleCNN(
                    in_planes, num_channels,
                    (math.ceil(observation_shape[1] / 16), math.ceil(observation_shape[2] / 16)))
            else:
                raise NotImplementedError('downsample should be "resnet" or "CNN".')
        # built even when a down-sampler replaces it: it is part of the reference's state dict
        self.conv = conv3x3(in_planes, num_channels)
        self.bn = BatchNorm2d(num_channels)
        self.resblocks = _tower(num_channels, num_blocks)

    def forward(self, x):
        x = self.downsample_net(x) if self.downsample else conv_bn_relu(self.conv, self.bn, x)
        for block in self.resblocks:
            x = block(x)
        return x


class DynamicsNetwork(torch.nn.Module):
    def __init__(self, num_blocks, num_channels, reduced_channels_reward, fc_reward_layers,
                 full_support_size, block_output_size_reward):
        super().__init__()
        self.conv = conv3x3(num_channels, num_channels - 1)
        self.bn = BatchNorm2d(num_channels - 1)
        self.resblocks = _tower(num_channels - 1, num_blocks)
        self.conv1x1_reward = PointwiseConv2d(num_channels - 1, reduced_channels_reward)
        self.block_output_size_reward = block_output_size_reward
        self.fc = mlp(block_output_size_reward, fc_reward_layers, full_support_size)

    def forward(self, x):
        x = conv_bn_relu(self.conv, self.bn, x)
        for block in self.resblocks:
            x = block(x)
        reward = conv_head(x, self.conv1x1_reward, self.fc, self.block_output_size_reward)
        return x, reward


class PredictionNetwork(torch.nn.Module):
    def __init__(self, action_space_size, num_blocks, num_channels, reduced_channels_value,
                 reduced_channels_policy, fc_value_layers, fc_policy_layers, full_support_size,
                 block_output_size_value, block_output_size_policy):
        super().__init__()
        self.resblocks = _tower(num_channels, num_blocks)
        self.conv1x1_value = PointwiseConv2d(num_channels, reduced_channels_value)
        self.conv1x1_policy = PointwiseConv2d(num_channels, reduced_channels_policy)
        self.block_output_size_value = block_output_size_value
        self.block_output_size_policy = block_output_size_policy
        self.fc_value = mlp(block_output_size_value, fc_value_layers, full_support_size)
        self.fc_policy = mlp(block_output_size_policy, fc_policy_layers, action_space_size)

    def forward(self, x):
        for block in self.resblocks:
            x = block(x)
        value, policy = conv_heads(x, [(self.conv1x1_value, self.fc_value, self.block_output_size_value),
                                       (self.conv1x1_policy, self.fc_policy, self.block_output_size_policy)])
        return policy, value


class MuZeroResidualNetwork(AbstractNetwork):
    def __init__(self, observation_shape, stacked_observations, action_space_size, num_blocks,
                 num_channels, reduced_channels_reward, reduced_channels_value,
                 reduced_channels_policy, fc_reward_layers, fc_value_layers, fc_policy_layers,
                 support_size, downsample):
        super().__init__()
        self.action_space_size = action_space_size
        self.full_support_size = 2 * support_size + 1
        if downsample:
            plane = math.ceil(observation_shape[1] / 16) * math.ceil(observation_shape[2] / 16)
        else:
            plane = observation_shape[1] * observation_shape[2]
        self.representation_network = _Replica(RepresentationNetwork(
            observation_shape, stacked_observations, num_blocks, num_channels, downsample))
        self.dynamics_network = _Replica(DynamicsNetwork(
            num_blocks, num_channels + 1, reduced_channels_reward, fc_reward_layers,
            self.full_support_size, reduced_channels_reward * plane))
        self.prediction_network = _Replica(PredictionNetwork(
            action_space_size, num_blocks, num_channels, reduced_channels_value,
            reduced_channels_policy, fc_value_layers, fc_policy_layers, self.full_support_size,
            reduced_channels_value * plane, reduced_channels_policy * plane))

    def prediction(self, encoded_state):
        return self.prediction_network(encoded_state)

    # ---- whole towers in one launch (include/mzmcts.h mzmcts_board_tower) ----------------------------------------
    @staticmethod
    def _block_layers(blocks):
        """(conv, bn, relu, skip) per layer of a list of ResidualBlocks (models.py:213-229)."""
        layers = []
        for block in blocks:
            layers.append((block.conv1, block.bn1, 1, 0))
            layers.append((block.conv2, block.bn2, 1, 1))
        return layers

    def _tower(self, x, layers, exports, const_plane=False, gather=None, shape=None, device=None, heads=None):
        """Run `layers` = [(conv, bn, relu, skip)] on x in ONE launch; exports = {layer index: (raw, unit)} tensors (or
        None) that receive that layer's output / its min-max-rescaled form.  With `gather` (a _native.MzTowerGather,
        x = None, shape = (b, cin0, h, w)) the input comes straight from the search's hidden-state pool.  Returns False
        when the tower path does not apply (training, autograd, CPU, unsupported shape, activations too large for LDS):
        the caller then takes the per-layer path."""
        mode = os.environ.get("MZ_BOARD_CONV", "auto")
        if mode == "off" or os.environ.get("MZ_BOARD_TOWER", "on") == "off" or self.training or torch.is_grad_enabled() \
                or len(layers) > 16:
            return False
        if gather is None and (not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4):
            return False
        lib = _native.load()
        b, cin0, h, w = shape if gather is not None else x.shape
        device = device if gather is not None else x.device
        channels = layers[0][0].out_channels
        if not lib.mzmcts_board_conv_supported(cin0, channels, h, w):
            return False
        for conv, bn, _, _ in layers:
            if not isinstance(conv, BoardConv2d) or conv.out_channels != channels or conv.kernel_size != (3, 3) \
                    or conv.stride != (1, 1) or bn.training:
                return False
        if gather is None:
            x = x.contiguous()
        # 64-channel towers run on the 16-bit matrix path with every operand split into two fp16 halves (fp32-level
        # accuracy, csrc/board_conv.hip); MZ_BOARD_CONV_PRECISION=fp32 keeps the exact-fp32 MFMA form
        split = channels == 64 and os.environ.get("MZ_BOARD_CONV_PRECISION", "split") != "fp32"
        if heads and split:
            return False                                 # (heads inside the launch: the 16-channel board-column kernel)
        # The split tower's range is |activation| < 8188; it flags the blocks of samples that left it and the exact-fp32
        # tower, queued right behind it on the same stream, re-runs exactly those (include/mzmcts.h mzmcts_tower_layer.gate):
        # no NaN reaches the search, no host round trip, and the pair is captured into a hipGraph like any other launch.
        fallback = split and os.environ.get("MZ_SPLIT_FALLBACK", "on") != "off"
        gate = self._tower_gate(b, channels, h, w, device) if fallback else None

        def describe(as_split, gate_ptr):
            descs = (_native.MzTowerLayer * len(layers))()
            keep = []
            for i, (conv, bn, relu, skip) in enumerate(layers):
                scale, shift = bn.folded()
                table = None
                if as_split:
                    packed, table = conv.packed_split(const_plane and i == 0, h, w)
                else:
                    packed = conv.packed()
                raw, unit = exports.get(i, (None, None))
                keep += [scale, shift, packed, table, raw, unit]
                descs[i] = _native.MzTowerLayer(packed.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                table.data_ptr() if table is not None else None,
                                                raw.data_ptr() if raw is not None else None,
                                                unit.data_ptr() if unit is not None else None, conv.in_channels, relu, skip, 0,
                                                gate_ptr if i == 0 else None)
            return descs, keep

        def launch(as_split, descs):
            if heads:
                # heads = [(layer index, (conv1x1, fc, flat size), out tensor)]: computed inside the launch
                hdescs = (_native.MzTowerHead * len(heads))(*[
                    _native.MzTowerHead(_native.MzHeadDesc(
                        conv.weight.data_ptr(), conv.bias.data_ptr(), fc[0].weight.data_ptr(), fc[0].bias.data_ptr(),
                        fc[2].weight.data_ptr(), fc[2].bias.data_ptr(), channels, h * w, conv.out_channels,
                        fc[0].out_features, fc[2].out_features), out.data_ptr(), layer, 0)
                    for layer, (conv, fc, _), out in heads])
                return lib.mzmcts_board_tower_heads(None if gather is not None else x.data_ptr(),
                                                    ctypes.byref(gather) if gather is not None else None, b, cin0, channels,
                                                    h, w, ctypes.addressof(descs), len(layers), ctypes.addressof(hdescs),
                                                    len(heads), stream)
            if gather is not None:
                return lib.mzmcts_board_tower_gathered(ctypes.byref(gather), b, cin0, 1 if as_split else 0, channels, h, w,
                                                       ctypes.addressof(descs), len(layers), stream)
            if as_split:
                return lib.mzmcts_board_tower_split(x.data_ptr(), b, cin0, 1 if const_plane else 0, channels, h, w,
                                                    ctypes.addressof(descs), len(layers), stream)
            return lib.mzmcts_board_tower(x.data_ptr(), b, cin0, channels, h, w, ctypes.addressof(descs), len(layers), stream)

        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            descs, keep = describe(split, gate.data_ptr() if gate is not None else None)
            rc = launch(split, descs)
            if rc == 0 and gate is not None:
                descs32, keep32 = describe(False, gate.data_ptr())
                rc = launch(False, descs32)
        if rc == -1:
            return False                                 # (does not fit in LDS / shape not covered)
        if rc != 0:
            raise RuntimeError(f"mzmcts_board_tower failed ({rc}) on an input of shape {(b, cin0, h, w)}")
        return True

    def _tower_gate(self, batch, channels, h, w, device):
        """The overflow hand-over buffer of tower launches of this shape: i32[blocks + 1], same tensor for the model's
        lifetime (a captured hipGraph keeps its address); [blocks] counts the blocks that fell back to the fp32 tower."""
        gates = self.__dict__.setdefault("_tower_gates", {})
        key = (int(batch), channels, h, w, str(device))
        if key not in gates:
            blocks = int(_native.load().mzmcts_board_tower_blocks(batch, channels, h, w))
            if blocks <= 0:
                raise RuntimeError(f"mzmcts_board_tower_blocks({batch}, {channels}, {h}, {w}) = {blocks}")
            gates[key] = torch.zeros(blocks + 1, dtype=torch.int32, device=device)
        return gates[key]

    def split_tower_fallbacks(self, reset=True):
        """Blocks of samples the split-precision towers handed to the exact-fp32 tower since the last call (their
        activations left the fp16 range, |x| >= 8188, or were not finite).  Waits for the device."""
        total = 0
        for gate in self.__dict__.get("_tower_gates", {}).values():
            total += int(gate[-1].item())
            if reset:
                gate[-1].zero_()
        return total

    def _recurrent_tower(self, planes, out_state, gather=None, shape=None, device=None):
        """dynamics + rescale + prediction towers of recurrent_inference as one launch; None if not applicable."""
        dyn, pred = self.dynamics_network.module, self.prediction_network.module
        layers = [(dyn.conv, dyn.bn, 1, 0)] + self._block_layers(dyn.resblocks)
        last_dyn = len(layers) - 1
        layers += self._block_layers(pred.resblocks)
        b, _, h, w = shape if gather is not None else planes.shape
        device = device if gather is not None else planes.device
        c = dyn.conv.out_channels
        raw = torch.empty((b, c, h, w), dtype=torch.float32, device=device)
        state = out_state if out_state is not None else torch.empty_like(raw)
        if not (state.is_contiguous() and state.dtype == torch.float32 and tuple(state.shape) == (b, c, h, w)):
            return None
        features = torch.empty_like(raw) if len(layers) - 1 > last_dyn else None
        exports = {last_dyn: (raw, state)}
        if features is not None:
            exports[len(layers) - 1] = (features, None)
        if not self._tower(planes, layers, exports, const_plane=True, gather=gather, shape=shape, device=device):
            return None                                                  # (the last input plane is action / A)
        return raw, state, features if features is not None else state

    def _recurrent_fused(self, planes, out_state, gather=None, shape=None, device=None):
        """dynamics + rescale + prediction towers AND the three heads of recurrent_inference in ONE launch (include/mzmcts.h
        mzmcts_board_tower_heads: the reward head reads the raw dynamics output, value and policy the prediction features,
        all while the activations are in LDS); returns (value, reward, policy logits, next state) or None when the launch
        does not cover this network (the caller takes the tower launch + the heads launch)."""
        # opt-in: measured slower than the tower launch + the heads launch (65536 TicTacToe boards: 262 us against 130 + 73 us;
        # csrc/board_conv.hip board_tower_cols_kernel<.., HEADS = true>), kept for its bit-identical results
        if os.environ.get("MZ_TOWER_HEADS", "off") != "on":
            return None
        dyn, pred = self.dynamics_network.module, self.prediction_network.module
        trios = [(dyn.conv1x1_reward, dyn.fc, dyn.block_output_size_reward),
                 (pred.conv1x1_value, pred.fc_value, pred.block_output_size_value),
                 (pred.conv1x1_policy, pred.fc_policy, pred.block_output_size_policy)]
        for conv, fc, _ in trios:
            if not (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (1, 1) and conv.bias is not None and len(fc) == 4
                    and isinstance(fc[0], torch.nn.Linear) and isinstance(fc[1], torch.nn.ELU) and fc[1].alpha == 1.0
                    and isinstance(fc[2], torch.nn.Linear) and isinstance(fc[3], torch.nn.Identity)):
                return None
        layers = [(dyn.conv, dyn.bn, 1, 0)] + self._block_layers(dyn.resblocks)
        last_dyn = len(layers) - 1
        layers += self._block_layers(pred.resblocks)
        if len(layers) - 1 == last_dyn:
            return None                                  # (no prediction blocks: all three heads would share a layer)
        b, _, h, w = shape if gather is not None else planes.shape
        device = device if gather is not None else planes.device
        c = dyn.conv.out_channels
        state = out_state if out_state is not None else torch.empty((b, c, h, w), dtype=torch.float32, device=device)
        if not (state.is_contiguous() and state.dtype == torch.float32 and tuple(state.shape) == (b, c, h, w)):
            return None
        outs = [torch.empty((b, fc[2].out_features), dtype=torch.float32, device=device) for _, fc, _ in trios]
        heads = [(last_dyn, trios[0], outs[0]), (len(layers) - 1, trios[1], outs[1]), (len(layers) - 1, trios[2], outs[2])]
        if not self._tower(planes, layers, {last_dyn: (None, state)}, const_plane=True, gather=gather, shape=shape,
                           device=device, heads=heads):
            return None
        return outs[1], outs[0], outs[2], state

    def pool_towers_supported(self, batch, device, state_shape):
        """Can recurrent_inference_from_pool serve hidden states of shape `state_shape` = (channels, h, w) (inference
        mode, tower shapes the HIP library covers)?  The engine asks once per model."""
        if self.training or torch.device(device).type != "cuda":
            return False
        if os.environ.get("MZ_BOARD_CONV", "auto") == "off" or os.environ.get("MZ_BOARD_TOWER", "on") == "off" \
                or os.environ.get("MZ_TOWER_GATHER", "on") == "off":
            return False
        dyn, pred = self.dynamics_network.module, self.prediction_network.module
        convs = [dyn.conv] + [c for blk in list(dyn.resblocks) + list(pred.resblocks) for c in (blk.conv1, blk.conv2)]
        c = dyn.conv.out_channels
        if len(state_shape) != 3 or state_shape[0] != c:
            return False
        # measured (one box, A/B): every tower gains from gathering for itself -- 84x84 config 27.8 -> 29.0 M simulations/s,
        # TicTacToe 138.2 -> 138.6 M, Connect4 (64-channel split tower) 6.81 -> 7.02 M once its input fill walks planes with
        # the rows' addresses in LDS (before that it lost: 5.63 -> 5.49 M).  MZ_TOWER_GATHER=off keeps the tensor form.
        h, w = int(state_shape[1]), int(state_shape[2])
        lib = _native.load()
        return (all(isinstance(conv, BoardConv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) for conv in convs)
                and len(convs) <= 16 and bool(lib.mzmcts_board_conv_supported(c + 1, c, h, w)))

    def recurrent_inference_from_pool(self, gather, batch, out_state):
        """recurrent_inference whose dynamics input the towers gather themselves from the engine's hidden-state pool
        (include/mzmcts.h mzmcts_board_tower_gathered; `gather` = engine.tower_gather() after a select())."""
        c = self.dynamics_network.module.conv.out_channels
        h, w = out_state.shape[-2], out_state.shape[-1]
        whole = self._recurrent_fused(None, out_state, gather=gather, shape=(batch, c + 1, h, w), device=out_state.device)
        if whole is not None:
            return whole
        fused = self._recurrent_tower(None, out_state, gather=gather, shape=(batch, c + 1, h, w), device=out_state.device)
        if fused is None:
            raise RuntimeError("recurrent_inference_from_pool: the tower launch does not cover this network "
                               "(pool_towers_supported should have said so)")
        raw, next_state, features = fused
        dyn, pred = self.dynamics_network.module, self.prediction_network.module
        reward, value, policy_logits = conv_heads_multi([
            (raw, (dyn.conv1x1_reward, dyn.fc, dyn.block_output_size_reward)),
            (features, (pred.conv1x1_value, pred.fc_value, pred.block_output_size_value)),
            (features, (pred.conv1x1_policy, pred.fc_policy, pred.block_output_size_policy))])
        return value, reward, policy_logits, next_state

    def representation(self, observation):
        return board_rescale(self.representation_network(observation))

    def dynamics(self, encoded_state, action, out_state=None):
        return self.dynamics_from_planes(state_action_planes(encoded_state, action, self.action_space_size), out_state)

    def dynamics_from_planes(self, x, out_state=None):
        """dynamics on its ready-made input [B, channels + 1, h, w] (state planes + action / A plane)."""
        raw, reward = self.dynamics_network(x)
        return board_rescale(raw, out=out_state), reward

    def _root_tower(self, observation):
        """representation (stem or down-sampler + residual blocks) + rescale + prediction blocks of initial_inference
        (models.py:604-616, 318-335, 525-549) as one tower launch -- the shape of _recurrent_tower without the action
        plane: a network that starts with a residual block takes its first skip connection from the tower's input.
        Returns (encoded state, prediction features) or None when the launch does not apply (MZ_ROOT_TOWER=off, training,
        shapes the towers do not cover): the caller takes the per-layer path."""
        if os.environ.get("MZ_ROOT_TOWER", "on") == "off" or not observation.is_cuda or self.training or torch.is_grad_enabled():
            return None
        rep, pred = self.representation_network.module, self.prediction_network.module
        if rep.downsample:
            x = rep.downsample_net(observation)
            layers = []
        else:
            x = observation
            layers = [(rep.conv, rep.bn, 1, 0)]
        layers += self._block_layers(rep.resblocks)
        if not layers or x.dim() != 4:
            return None
        last_rep = len(layers) - 1
        layers += self._block_layers(pred.resblocks)
        c = layers[0][0].out_channels
        if c == 64:
            return None     # (the root of a 64-channel network keeps the exact-fp32 per-layer kernels, not the split tower)
        b, _, h, w = x.shape
        state = torch.empty((b, c, h, w), dtype=torch.float32, device=x.device)
        features = torch.empty_like(state) if len(layers) - 1 > last_rep else None
        exports = {last_rep: (None, state)}
        if features is not None:
            exports[len(layers) - 1] = (features, None)
        if not self._tower(x, layers, exports):
            return None
        return state, features if features is not None else state

    def initial_inference(self, observation):
        fused = self._root_tower(observation)
        if fused is not None:
            encoded_state, features = fused
            pred = self.prediction_network.module
            value, policy_logits = conv_heads(features, [(pred.conv1x1_value, pred.fc_value, pred.block_output_size_value),
                                                         (pred.conv1x1_policy, pred.fc_policy, pred.block_output_size_policy)])
        else:
            encoded_state = self.representation(observation)
            policy_logits, value = self.prediction(encoded_state)
        reward = self._zero_reward_logits(observation.shape[0], observation.device)
        return value, reward, policy_logits, encoded_state

    def recurrent_inference(self, encoded_state, action, out_state=None):
        if encoded_state.is_cuda and not self.training and not torch.is_grad_enabled():
            return self.recurrent_inference_from_planes(
                state_action_planes(encoded_state, action, self.action_space_size), out_state)
        next_state, reward = self.dynamics(encoded_state, action, out_state)
        policy_logits, value = self.prediction(next_state)
        return value, reward, policy_logits, next_state

    def recurrent_inference_from_planes(self, planes, out_state=None):
        """recurrent_inference for a caller that already holds the dynamics input (the engine's gather writes it:
        include/mzmcts.h mzmcts_select_planes)."""
        whole = self._recurrent_fused(planes, out_state) if planes.is_cuda and not self.training and not torch.is_grad_enabled() else None
        if whole is not None:
            return whole
        fused = self._recurrent_tower(planes, out_state)
        if fused is not None:
            raw, next_state, features = fused
            dyn, pred = self.dynamics_network.module, self.prediction_network.module
            reward, value, policy_logits = conv_heads_multi([
                (raw, (dyn.conv1x1_reward, dyn.fc, dyn.block_output_size_reward)),
                (features, (pred.conv1x1_value, pred.fc_value, pred.block_output_size_value)),
                (features, (pred.conv1x1_policy, pred.fc_policy, pred.block_output_size_policy))])
            return value, reward, policy_logits, next_state
        next_state, reward = self.dynamics_from_planes(planes, out_state)
        policy_logits, value = self.prediction(next_state)
        return value, reward, policy_logits, next_state


# ------------------------------------------------------------------------------------------------
_SUPPORT_CACHE = {}


def support_to_scalar(logits, support_size):
    """Categorical -> scalar, reference models.py:641-662 (same fp32 operation order)."""
    key = (support_size, logits.device, logits.dtype)
    support = _SUPPORT_CACHE.get(key)
    if support is None:
        support = torch.arange(-support_size, support_size + 1, device=logits.device).to(logits.dtype)
        _SUPPORT_CACHE[key] = support
    probabilities = torch.softmax(logits, dim=1)
    x = torch.sum(support.expand(probabilities.shape) * probabilities, dim=1, keepdim=True)
    return torch.sign(x) * (((torch.sqrt(1 + 4 * 0.001 * (torch.abs(x) + 1 + 0.001)) - 1)
                             / (2 * 0.001)) ** 2 - 1)


def scalar_to_support(x, support_size):
    """Scalar -> two-hot categorical target, reference models.py:665-685 (training side; kept so
    trainer code that imports it from this module keeps working)."""
    x = torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + 0.001 * x
    x = torch.clamp(x, -support_size, support_size)
    low = x.floor()
    frac = x - low
    out = torch.zeros(x.shape[0], x.shape[1], 2 * support_size + 1, device=x.device)
    out.scatter_(2, (low + support_size).long().unsqueeze(-1), (1 - frac).unsqueeze(-1))
    upper = low + support_size + 1
    overflow = 2 * support_size < upper
    frac = frac.masked_fill(overflow, 0.0)
    upper = upper.masked_fill(overflow, 0.0)
    out.scatter_(2, upper.long().unsqueeze(-1), frac.unsqueeze(-1))
    return out

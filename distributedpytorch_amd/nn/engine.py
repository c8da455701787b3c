"""Execution engine: flat parameter arenas + a minimal reverse-mode tape with hand-written backward per layer.

MI355X-first design, not a torch.nn / autograd re-skin:
  * every parameter lives in ONE fp32 arena (`ParamStore.p`), its gradient at the same offset of a second arena
    (`.g`) and - in bf16 mode - its compute copy at the same offset of a bf16 shadow arena (`.p16`), so the fused
    optimizer updates the whole model (or one gradient bucket) with a single launch, the DDP reducer all-reduces
    contiguous slices with no packing, and zero_grad is one memset;
  * conv/linear weights are stored in the kernels' layout ([Cout, kh, kw, Cin], channels padded to a multiple of
    8) and converted to torchvision's OIHW only when a checkpoint is written or read;
  * activations are NHWC; there is no autograd graph - each layer pushes one closure on the tape.
PyTorch is used for memory handles (torch.empty on the caching allocator) and nothing else on the hot path.
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict
from typing import Callable, List, Optional

import torch

ALIGN = 64          # arena alignment in elements (256 B)
CPAD = 8            # channel padding granule: one 16-byte chunk of bf16, two of fp32


def pad_to(n: int, m: int = CPAD) -> int:
    return (n + m - 1) // m * m


def input_pad(c: int, dtype) -> int:
    """Channel padding of a NETWORK INPUT (and of the first conv's Cin).  The kernels move 16-byte chunks: 8 bf16 or
    4 fp32 channels, so an fp32 RGB input pads 3 -> 4 instead of 3 -> 8, which halves the stem convolution's GEMM-K
    (7*7*4 = 196 instead of 392: -7 ms per ResNet-50 step at batch 1024).  Everything else stays on the 8 granule."""
    if dtype == torch.float32 and c < CPAD:
        return pad_to(c, 4)
    return pad_to(c)


class Act:
    """An NHWC activation (possibly a channel slice of a wider buffer) plus its gradient slot."""
    __slots__ = ("data", "grad", "needs_grad", "grad_buf", "bn_src", "bn_slab", "compact_ok", "grad_compact", "bn_lin", "bn_src2", "bn_slab2", "bn_defer", "bn_dx_done")

    def __init__(self, data: torch.Tensor, needs_grad: bool = False):
        self.data = data
        self.grad: Optional[torch.Tensor] = None
        self.needs_grad = needs_grad
        self.grad_buf: Optional[torch.Tensor] = None   # preassigned gradient storage (slice of a concat grad)
        # BatchNorm-backward fusion (nn/layers.py: ConvBNAct): bn_src = (bn, z, mask_from_x) when this activation is the
        # output of a training-mode BatchNorm+ReLU; bn_slab = the partial sums of that BatchNorm's backward reduction,
        # written by the dgrad that completed this activation's gradient (which it also masked)
        self.bn_src = None
        self.bn_slab: Optional[torch.Tensor] = None
        # compact_ok: the caller promises that the NEXT contribution to this gradient after a 1x1 / stride-2 convolution's
        # is a Conv2d dgrad (models/resnet.py, blocks with a downsample branch); that convolution may then leave its
        # gradient in grad_compact = (tensor [N, H/2, W/2, C], the layer) instead of a full-resolution tensor that is
        # three quarters zeros (nn/layers.py: Conv2d._dgrad)
        self.compact_ok = False
        self.grad_compact = None
        # a SECOND BatchNorm in the sum under this activation's ReLU (the downsample branch of a ResNet block):
        # bn_lin = (bn, z) on the output of a BatchNorm without ReLU / residual; bn_src2 = the residual's bn_lin on the
        # block output; bn_slab2 = its partial sums, written by the same dgrad epilogue as bn_slab (nn/layers.py)
        self.bn_lin = None
        self.bn_src2 = None
        self.bn_slab2 = None
        # bn_defer = (scale, shift): data is the INPUT of a BatchNorm whose apply pass was left to the consumer (a ConvBNAct
        # taking this activation as its residual applies both BatchNorms in one kernel)
        self.bn_defer = None
        # bn_dx_done: the consumer's backward already ran this (bn_lin) BatchNorm's backward - parameter gradients and dx,
        # left in place of z - together with its own (one pass over the shared gradient, ops.bn_bwd_dx2)
        self.bn_dx_done = False

    @property
    def shape(self):
        return self.data.shape


class ParamSpec:
    __slots__ = ("name", "logical_shape", "storage_shape", "kind", "offset", "numel", "trainable", "index", "meta")

    def __init__(self, name, logical_shape, storage_shape, kind, trainable, meta=None):
        self.name = name
        self.logical_shape = tuple(logical_shape)
        self.storage_shape = tuple(storage_shape)
        self.kind = kind            # 'conv' (OIHW <-> KRSC), 'vec'
        self.trainable = trainable
        self.numel = int(math.prod(storage_shape))
        self.offset = -1
        self.index = -1
        self.meta = meta


class ParamStore:
    """Flat arenas for parameters, gradients and buffers, plus the torchvision-compatible state_dict view."""

    def __init__(self, device, compute_dtype=torch.float32):
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        self.params: List[ParamSpec] = []
        self.buffers: List[ParamSpec] = []
        self.entries: List[tuple] = []          # state_dict order: ('param'|'buffer'|'nbt', spec)
        self._init_values = {}
        self.aux_tensors: List[torch.Tensor] = []   # per-layer persistent scratch that carries state between steps
        self.p = self.g = self.p16 = self.b = self.nbt = self.buffer_arena = None
        self.buffer_fence = None        # callable set while a broadcast of buffer_arena is in flight (parallel/reducer.py)
        self.p_size = self.b_size = 0
        self.n_nbt = 0
        from .wprep import WeightPrep
        self.wprep = WeightPrep(self)       # batched per-step filter transposes / bf16 planes (nn/wprep.py)

    # ---------------------------------------------------------------- registration
    def add_conv_weight(self, name, cout, cin, kh, kw, init, trainable=True, flat_logical=False):
        """init: callable(logical_tensor[cout, cin, kh, kw]) filling it in place (torch.nn.init style)."""
        logical = (cout, cin * kh * kw) if flat_logical else (cout, cin, kh, kw)
        spec = ParamSpec(name, logical, (pad_to(cout), kh, kw, input_pad(cin, self.compute_dtype)), "conv", trainable,
                         meta=(cout, cin, kh, kw))
        t = torch.empty(cout, cin, kh, kw)
        init(t)
        self._register(spec, t)
        return spec

    def add_vec(self, name, n, init, trainable=True):
        spec = ParamSpec(name, (n,), (pad_to(n),), "vec", trainable, meta=(n,))
        t = torch.empty(n)
        init(t)
        self._register(spec, t)
        return spec

    def add_buffer(self, name, n, value):
        spec = ParamSpec(name, (n,), (pad_to(n),), "vec", False, meta=(n,))
        spec.offset = self.b_size
        self.b_size += pad_to(spec.numel, ALIGN)
        spec.index = len(self.buffers)
        self.buffers.append(spec)
        self.entries.append(("buffer", spec))
        self._init_values[("b", spec.index)] = torch.full((n,), float(value))
        return spec

    def add_nbt(self, name):
        idx = self.n_nbt
        self.n_nbt += 1
        self.entries.append(("nbt", (name, idx)))
        return idx

    def _register(self, spec, value):
        spec.offset = self.p_size
        self.p_size += pad_to(spec.numel, ALIGN)
        spec.index = len(self.params)
        self.params.append(spec)
        self.entries.append(("param", spec))
        self._init_values[("p", spec.index)] = value

    # ---------------------------------------------------------------- materialisation
    def finalize(self):
        dev = self.device
        self.p = torch.zeros(max(self.p_size, ALIGN), device=dev)
        self.g = torch.zeros_like(self.p)
        # BatchNorm running statistics (fp32) and num_batches_tracked (int64) share ONE byte arena, so DDP's per-step
        # buffer broadcast (X4 of SURVEY.md §2.6) is one collective, not one per dtype
        nb = max(self.b_size, ALIGN) * 4
        self.buffer_arena = torch.zeros(nb + max(self.n_nbt, 1) * 8, dtype=torch.uint8, device=dev)
        self.b = self.buffer_arena[:nb].view(torch.float32)
        self.nbt = self.buffer_arena[nb:].view(torch.int64)
        for spec in self.params:
            self.set_logical(spec, self._init_values.pop(("p", spec.index)))
        for spec in self.buffers:
            self.bview(spec)[: spec.meta[0]].copy_(self._init_values.pop(("b", spec.index)))
        if self.compute_dtype == torch.bfloat16:
            self.p16 = torch.zeros(self.p.numel(), device=dev, dtype=torch.bfloat16)
            self.refresh_compute_copy()
        return self

    def refresh_compute_copy(self):
        if self.p16 is not None:
            from .. import ops
            ops.cast(self.p, self.p16)

    def view(self, spec):      # fp32 master, storage layout
        return self.p[spec.offset: spec.offset + spec.numel].view(spec.storage_shape)

    def gview(self, spec):
        return self.g[spec.offset: spec.offset + spec.numel].view(spec.storage_shape)

    def cview(self, spec):     # compute-dtype copy the kernels read
        src = self.p16 if self.p16 is not None else self.p
        return src[spec.offset: spec.offset + spec.numel].view(spec.storage_shape)

    def bview(self, spec):
        return self.b[spec.offset: spec.offset + spec.numel]

    def fence_buffers(self):
        """Called by whoever is about to touch the buffer arena on the current stream (BatchNorm's finalize / eval
        coefficients, state_dict): if the step's buffer broadcast is still in flight, make the stream wait for it now."""
        f = self.buffer_fence
        if f is not None:
            self.buffer_fence = None
            f()

    # ---------------------------------------------------------------- logical (torchvision) <-> storage
    @staticmethod
    def storage_to_logical(spec, st):
        if spec.kind == "conv":
            cout, cin, kh, kw = spec.meta
            return st[:cout, :, :, :cin].permute(0, 3, 1, 2).contiguous().view(spec.logical_shape)
        return st[: spec.meta[0]].clone()

    @staticmethod
    def logical_to_storage(spec, logical, out):
        out.zero_()
        if spec.kind == "conv":
            cout, cin, kh, kw = spec.meta
            out[:cout, :, :, :cin].copy_(logical.reshape(cout, cin, kh, kw).permute(0, 2, 3, 1))
        else:
            out[: spec.meta[0]].copy_(logical.reshape(-1))
        return out

    def get_logical(self, spec):
        return self.storage_to_logical(spec, self.view(spec))

    def set_logical(self, spec, value):
        self.logical_to_storage(spec, value.to(self.device, torch.float32), self.view(spec))

    def trainable_ranges(self):
        """Maximal contiguous arena ranges [lo, hi) covering trainable parameters, in arena order."""
        ranges = []
        for spec in self.params:
            if not spec.trainable:
                continue
            lo, hi = spec.offset, spec.offset + pad_to(spec.numel, ALIGN)
            if ranges and ranges[-1][1] == lo:
                ranges[-1][1] = hi
            else:
                ranges.append([lo, hi])
        return [tuple(r) for r in ranges]

    # ---------------------------------------------------------------- state_dict (torchvision names)
    def state_dict(self, prefix=""):
        self.fence_buffers()
        sd = OrderedDict()
        for kind, item in self.entries:
            if kind == "param":
                sd[prefix + item.name] = self.get_logical(item)
            elif kind == "buffer":
                sd[prefix + item.name] = self.bview(item)[: item.meta[0]].clone()
            else:
                name, idx = item
                sd[prefix + name] = self.nbt[idx].clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        """Accepts keys with or without the DDP 'module.' prefix (Q16 of SURVEY.md §2.8)."""
        self.fence_buffers()
        clean = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        expected = []
        for kind, item in self.entries:
            name = item.name if kind != "nbt" else item[0]
            expected.append(name)
            if name not in clean:
                if strict:
                    raise KeyError(f"missing key in state_dict: {name}")
                continue
            v = clean[name]
            if kind == "param":
                if tuple(v.shape) != item.logical_shape:
                    raise ValueError(f"shape mismatch for {name}: {tuple(v.shape)} vs {item.logical_shape}")
                self.set_logical(item, v)
            elif kind == "buffer":
                self.bview(item)[: item.meta[0]].copy_(v.to(self.device, torch.float32))
            else:
                self.nbt[item[1]] = int(v)
        if strict:
            extra = set(clean) - set(expected)
            if extra:
                raise KeyError(f"unexpected keys in state_dict: {sorted(extra)[:5]}")
        self.refresh_compute_copy()

    def num_parameters(self):
        return sum(int(math.prod(s.logical_shape)) for s in self.params)


class Tape:
    """Reverse-mode tape: layers push closures during forward; backward() pops them in reverse order."""

    def __init__(self):
        self.fns: List[Callable[[], None]] = []
        self.param_ready_hook: Optional[Callable[[ParamSpec], None]] = None
        # weight-gradient kernels running on the side stream whose join with the main stream is still outstanding:
        # (done event, callback run after the join, tensors the side-stream kernels read - kept alive until the join)
        self.side_pending: List[tuple] = []

    def push(self, fn):
        self.fns.append(fn)

    def backward(self):
        while self.fns:
            self.fns.pop()()
        self.join_side(0)

    def defer_side(self, done_event, after_join, keep_alive):
        self.side_pending.append((done_event, after_join, keep_alive))

    def join_side(self, keep: int):
        """Make the current stream wait for all but the newest `keep` side-stream jobs, oldest first, and run their
        callbacks (gradient-ready announcements: they must be ordered after the kernels that wrote the gradients)."""
        while len(self.side_pending) > keep:
            ev, after, _refs = self.side_pending.pop(0)
            torch.cuda.current_stream().wait_event(ev)
            after()

    def ready(self, spec):
        """Called by a layer when the gradient of `spec` is final for this step (feeds the DDP reducer)."""
        if self.param_ready_hook is not None:
            self.param_ready_hook(spec)


class Context:
    """Per-forward state: tape, mode flags, shared workspaces."""

    def __init__(self, store: ParamStore, training: bool, tape: Optional[Tape] = None, rng=None):
        self.store = store
        self.training = training
        self.tape = tape
        self.rng = rng                       # uint64[2] device tensor (seed, step) for dropout
        self.device = store.device
        self.dtype = store.compute_dtype
        self.cuda = store.device.type == "cuda"
        self.wprep = None                    # set by WeightPrep.run when this pass prepared the filters in one launch

    @property
    def grad_enabled(self):
        return self.tape is not None

    def empty(self, shape, dtype=None):
        """Step-scoped scratch / activation storage: from the planned arena when one is active (nn/arena.py)."""
        a = _current_arena[0]
        if a is not None:
            return a.alloc(shape, dtype or self.dtype)
        return torch.empty(shape, device=self.device, dtype=dtype or self.dtype)

    def zeros(self, shape):
        return torch.zeros(shape, device=self.device, dtype=self.dtype)


_current_arena = [None]        # the ActivationArena serving Context.empty (set by DistributedModel around a step)


def set_current_arena(arena):
    prev = _current_arena[0]
    _current_arena[0] = arena
    return prev


_workspaces = {}


_side_streams = {}
# Weight-gradient kernels run on a second stream (DPA_BWD_STREAMS=1 puts them back on the main stream) and the main stream
# joins them DPA_BWD_LAG conv layers later (nn/layers.py: Conv2d.backward_raw).  With lag 1 the wgrad of layer k overlaps
# the small dependent kernels of the layer below it (slab reduce -> BatchNorm finalize -> dx: ~40 us per layer in which a
# single stream leaves the GPU nearly idle): -1 % on the ResNet-50 step, measured A/B on one box (214.0 vs 216.3 ms).
# Larger lags gain nothing (two MFMA-bound kernels share the matrix pipes; the memory-bound BatchNorm kernels do not get
# co-scheduled with a kernel that fills VGPRs and LDS).
# OPT-IN since the end of round 2 (DPA_BWD_STREAMS=2): once the small dependent kernels had been made shorter (finalize
# kernels, batched weight preparation) there was nothing left for the side stream to fill - four interleaved A/B pairs on
# one box: 217.2 / 217.8 / 217.2 / 217.7 ms with it, 215.2 / 217.2 / 215.4 / 217.1 ms without
# (profiles/r2_side_stream_threshold.txt) - and one stream is the simpler thing to run next to the collectives.
DUAL_STREAM_BACKWARD = os.environ.get("DPA_BWD_STREAMS", "1") == "2"
BWD_LAG = int(os.environ.get("DPA_BWD_LAG", "1"))


def side_stream(device):
    """Second (equal-priority) HIP stream for the weight-gradient kernels: dgrad and wgrad of one conv are independent,
    so running them concurrently lets each fill the other's tile-quantisation tail (e.g. 784 tiles on 512 CU slots).
    Returns None when disabled (DPA_BWD_STREAMS=1), on CPU, or while a hipGraph is being captured."""
    if not DUAL_STREAM_BACKWARD or device.type != "cuda" or torch.cuda.is_current_stream_capturing():
        return None
    key = str(device)
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=device)      # priority 0, like the comm stream: csrc/comm/rccl_backend.cpp
    return s


_workspaces_frozen = [False]      # set once a hipGraph has been captured: its kernels hold these buffers' addresses


def freeze_workspaces():
    _workspaces_frozen[0] = True


def workspace(device, n, tag, dtype=torch.float32):
    """Persistent scratch (stream-ordered reuse).  A buffer is replaced by a larger one on demand - but never after a
    hipGraph has been captured: the graph's kernels point at the old buffer, and freeing it under them would be a
    use-after-free on the device.  The old buffer is then kept alive next to the new one."""
    key = (str(device), tag, dtype)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < n:
        if buf is not None and _workspaces_frozen[0]:
            _workspaces.setdefault(("retired",), []).append(buf)       # captured graphs may still reference it
        buf = torch.empty(max(int(n), 1 << 14), device=device, dtype=dtype)
        _workspaces[key] = buf
    return buf


def accumulate_grad(act: Act, g: torch.Tensor):
    """act.grad += g (first contribution is adopted without a copy)."""
    from .. import ops
    if act.grad is None:
        if act.grad_buf is not None and act.grad_buf.data_ptr() != g.data_ptr():
            ops.copy(g, act.grad_buf)
            act.grad = act.grad_buf
        else:
            act.grad = g
    else:
        ops.add(act.grad, g, act.grad)

"""Host side of ``nqa_wgrad`` (``nequip_amd/csrc/wgrad.hip``): parameter gradients ``dW = A^T B`` of the dense maps on the
path (ScalarMLP layers, ``o3.Linear``, the self-connection), reduced over the edge / atom rows in one split-K fp32-MFMA
launch plus one sum over the partial tiles.  Training only; first order (the result is not differentiable -- callers
that are asked for a differentiable parameter gradient, i.e. ``create_graph=True``, use their ATen formulation).

``param_grads_wanted()`` / ``inputs_only_backward()``: ``torch.autograd.grad(energy, [pos], create_graph=True)`` (the force
pass of force-matching training, ``nequip/nn/grad_output.py:216-221``) runs every custom ``Function.backward`` in full,
including parameter-gradient outputs that the engine then throws away (built-in nodes skip them through
``task_should_compute_output``; Python Functions cannot see that).  ``ForceStressOutput`` therefore wraps its
``autograd.grad`` call in ``inputs_only_backward()``, and the Functions on the path consult ``param_grads_wanted()``."""

import contextlib
import ctypes
import os
import struct
import threading
from typing import Optional, Sequence, Tuple

import torch

from .. import _lib
from . import ktimer

# Process-wide, not thread-local: the autograd engine runs Function.backward on its own per-device worker threads, so
# a thread-local set by the caller of autograd.grad would never be seen there.  autograd.grad is synchronous, and the
# depth counter makes nesting safe; concurrent training steps from several Python threads of one process would only
# lose the optimisation's precision (a parameter gradient computed that nobody reads), never correctness, because the
# flag is only ever raised around calls that do not ask for parameter gradients -- and those who do ask for them
# (loss.backward()) must not run concurrently with such a call in the same process.
_inputs_only_depth = 0
_lock = threading.Lock()


def param_grads_wanted() -> bool:
    return _inputs_only_depth == 0


@contextlib.contextmanager
def inputs_only_backward():
    """Inside: backward passes are known to be asked for gradients of data inputs only (no parameters)."""
    global _inputs_only_depth
    with _lock:
        _inputs_only_depth += 1
    try:
        yield
    finally:
        with _lock:
            _inputs_only_depth -= 1


_eval_param_grads = False


def eval_parameter_gradients(enabled: Optional[bool] = None) -> bool:
    """Eval-mode modules (radial MLP, ``o3.Linear``, self-connection) treat their weights as constants: packed / split
    weight images are cached per parameter version and no parameter gradient is produced -- the inference fast path.  The
    reference produces parameter gradients in eval mode as well; ``eval_parameter_gradients(True)`` makes these modules use
    their training formulation whenever grad mode is on and the weights require grad.  Returns the current setting."""
    global _eval_param_grads
    if enabled is not None:
        _eval_param_grads = bool(enabled)
    return _eval_param_grads


def differentiable_parameters(module_training: bool, *params: torch.Tensor) -> bool:
    """Should a module run its parameter-differentiable formulation?  (training mode, or eval with the switch above)"""
    if module_training:
        return True
    return _eval_param_grads and torch.is_grad_enabled() and any(p.requires_grad for p in params)


# ---- parameter-side stream (training) --------------------------------------------------------------------------------------
# A training step has two kinds of work: the DATA chain (activations forward, their gradients backward: the critical path) and
# PARAMETER-side work (weights prepared from the parameters in the forward pass; in the backward pass the parameter gradients
# ``A^T B`` -- ``nqa_wgrad`` launches, their split-K sums, scalings -- the adjoints of the weight preparation and the
# accumulation into ``.grad``).  Nothing on the data chain waits for a parameter gradient, but on ONE stream they run in line
# with it: ~1.5 of 7 ms at the cfg-4 shape (``profiles/r5_train_timeline.txt``: one queue, 414 kernels).
#
# The autograd engine runs every backward node on the stream its forward ran on, synchronises consumer with producer streams
# itself and joins all streams at the end of ``backward()``.  So: the weight preparation of a module runs on a per-device side
# stream in the forward pass (``parameter_side`` + ``publish``; the tensor it yields is marked), and a ``Function.backward``
# that finds the mark launches its parameter-gradient kernels on that same stream (``parameter_side`` again, no join): their
# consumers -- the adjoint of the preparation, ``AccumulateGrad`` -- are nodes of that stream, the data chain goes on.
#
# MEASURED, AND OFF BY DEFAULT (``NQA_PARAM_STREAM=1`` switches it on).  Same box, cfg-4-shaped step as one hipGraph
# (``profiles/r5_train_param_stream_ab.txt``): 6.65 ms on one stream, 6.98-7.02 ms with the parameter-side stream; identical
# losses and parameter gradients (``tests/test_training_step.py`` etc. pass either way).  The ``AccumulateGrad`` nodes of the
# parameters outlive an iteration (hooks of the DDP wrapper and the optimizer hold them) and keep the stream they were created
# on, so every parameter gradient is handed back to the main stream the moment it is produced (the engine warns about exactly
# that): the data chain waits for each split-K product after all, and ~60 cross-stream dependencies per step cost more than
# the overlap of one ``grad_x`` kernel per site returns.  What would help is keeping parameter gradients out of autograd
# (accumulated by the side stream into a bucket that is joined once, before the optimizer); not built.
_param_streams = {}


def param_stream(device) -> Optional["torch.cuda.Stream"]:
    if os.environ.get("NQA_PARAM_STREAM", "0") in ("", "0"):
        return None
    device = torch.device(device)
    if device.type != "cuda":
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _param_streams.get(idx)
    if s is None:
        s = _param_streams[idx] = torch.cuda.Stream(device=idx)
    return s


@contextlib.contextmanager
def parameter_side(device, *reads):
    """Body on the parameter-side stream, ordered behind what the current stream has queued; ``reads``: tensors of the current
    stream that the body reads (kept from being recycled under it).  Yields the stream (``None``: switched off, body runs
    where it is)."""
    side = param_stream(device)
    if side is None:
        yield None
        return
    cur = torch.cuda.current_stream(device)
    side.wait_stream(cur)
    for t in reads:
        if t is not None and t.is_cuda:
            t.record_stream(side)
    with torch.cuda.stream(side):
        yield side


def publish(device, *tensors) -> None:
    """After parameter-side work whose results the CURRENT stream reads (a forward pass): wait for it, mark the results as
    parameter-side tensors (``Function.backward`` looks for the mark)."""
    side = param_stream(device)
    if side is None:
        return
    cur = torch.cuda.current_stream(device)
    cur.wait_stream(side)
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(cur)
            t._nqa_param_side = True


def is_param_side(t) -> bool:
    return bool(getattr(t, "_nqa_param_side", False)) and param_stream(t.device) is not None


# ---- deferred parameter gradients (training) --------------------------------------------------------------------------------
# The way out named above: inside ``with deferred_parameter_gradients():`` (around ``loss.backward()``) the Functions whose
# weights are a plain function of ONE parameter (``o3.Linear``: ``weight * scale_vec``, also through its transposed copy; the
# radial MLP's last layer) do not hand their parameter gradient to autograd.  They launch the split-K product, its sum and the
# scalings on a side stream, add the result into a bucket there and return ``None``; the data chain never waits for them.  On
# leaving the block the current stream waits for the side stream ONCE and the bucket goes into ``.grad`` (set, or added to what
# autograd delivered by other routes).  Same kernels as through autograd; the per-parameter sums may be taken in another order
# (losses agree to 1e-7).  ``NQA_DEFER_PARAM_GRADS=0`` makes the block a no-op.
#
# Measured (``profiles/r5_train_deferred_grads_ab.txt``, cfg-4-shaped step as one hipGraph, same box): 6.65-6.69 -> 6.37-6.39 ms
# -- but ONLY with the side launches trailing by one site (``NQA_DEFER_LAG``, default 1).  Launched at the fork itself the step
# got SLOWER (6.63 -> 6.9 ms): in a captured graph the first node created behind a fork continues on the parent's queue and
# later ones move to another; with the side work created first, the data chain hopped queues at every one of ~25 forks (10-40 us
# of idle time each, ``profiles/r5_train_timeline.txt``'s sibling trace).  Created one site late, the side work is the later
# child and the data chain stays where it is.
_deferred = None
_defer_streams = {}


def _defer_stream(device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    s = _defer_streams.get(idx)
    if s is None:
        s = _defer_streams[idx] = torch.cuda.Stream(device=idx)
    return s


@contextlib.contextmanager
def deferred_parameter_gradients():
    """Around ``loss.backward()``: see above.  Not re-entrant (an inner block is a no-op); gradients taken with
    ``torch.autograd.grad(..., parameters)`` inside the block would miss the deferred contributions -- use ``.backward()``.

    **Restriction.**  The deferred parameters get ``None`` from autograd and their ``.grad`` is written when the block exits:
    ``AccumulateGrad``, post-accumulate hooks and the reducer hooks of ``torch.nn.parallel.DistributedDataParallel`` never
    fire for them.  The block therefore only composes with a gradient synchronisation that runs AFTER it -- this package's
    ``SimpleDDPStrategy.post_backward`` (one flat all-reduce over ``.grad``, the reference's ``train/simple_ddp.py:26-59``;
    ``SimpleDDPStrategy.backward`` wraps the block itself) -- and a parameter with gradient hooks raises here instead of
    silently skipping them.  When the block exits with an exception (a failed backward, an aborted graph capture) nothing is
    flushed: the side stream is joined and the bucket dropped, ``.grad`` is left as it was."""
    global _deferred
    if _deferred is not None or os.environ.get("NQA_DEFER_PARAM_GRADS", "1") in ("0",):
        yield
        return
    bucket = {}
    _deferred = bucket
    try:
        yield
    except BaseException:
        _deferred = None
        _drop_deferred(bucket)
        raise
    else:
        _deferred = None
        _flush_deferred(bucket)


def deferring() -> bool:
    """Inside a ``deferred_parameter_gradients`` block, in a first-order backward pass that wants parameter gradients."""
    return _deferred is not None and param_grads_wanted() and not torch.is_grad_enabled()


def defer(param: torch.Tensor, make, *reads) -> None:
    """``make()`` -> this backward node's contribution to ``param.grad`` (any shape with ``param.numel()`` elements), evaluated on
    the side stream behind what the current stream has queued; ``reads``: current-stream tensors it reads."""
    dev = param.device
    _check_no_grad_hooks(param)
    ready = torch.cuda.Event()
    ready.record(torch.cuda.current_stream(dev))  # (everything `make` reads has been queued by now)
    pending = _deferred.setdefault("_pending", [])
    pending.append((param, make, reads, ready))
    # The launches of a site go out one site LATE (NQA_DEFER_LAG, default 1): by then the current stream has queued the data
    # chain's next kernels, so in a captured graph they -- not the side work -- are the first children of the fork.
    lag = int(os.environ.get("NQA_DEFER_LAG", "1") or 0)
    while len(pending) > lag:
        _launch_deferred(_deferred, pending.pop(0))


def _launch_deferred(bucket, item) -> None:
    param, make, reads, ready = item
    side = _defer_stream(param.device)
    side.wait_event(ready)
    for t in reads:
        if t is not None and t.is_cuda:
            t.record_stream(side)
    with torch.cuda.stream(side):
        g = make().reshape(param.shape)
        ent = bucket.get(id(param))
        if ent is None:
            bucket[id(param)] = [param, g]
        else:
            ent[1].add_(g)


def _check_no_grad_hooks(param: torch.Tensor) -> None:
    """A deferred parameter never reaches ``AccumulateGrad``: refuse parameters whose gradient somebody listens to (tensor
    hooks, post-accumulate-grad hooks).  torch DDP's reducer hooks the grad-accumulator NODE, which cannot be seen from the
    parameter: that combination is excluded by the docstring of the block, not detected here."""
    if getattr(param, "_backward_hooks", None) or getattr(param, "_post_accumulate_grad_hooks", None):
        raise RuntimeError("deferred_parameter_gradients(): a parameter with gradient hooks would never see them fire (its "
                           "gradient bypasses autograd's accumulation).  Use SimpleDDPStrategy (all-reduce after the block) "
                           "instead of torch DDP / hooks, or set NQA_DEFER_PARAM_GRADS=0")


def _drop_deferred(bucket) -> None:
    """The block exited with an exception: no launches, no ``.grad`` writes -- only make sure nothing on the side stream is
    still reading tensors the caller is about to free."""
    bucket.pop("_pending", None)
    devs = {param.device for param, _ in bucket.values()}
    for dev in devs:
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(dev).wait_stream(_defer_stream(dev))
    bucket.clear()


def _flush_deferred(bucket) -> None:
    for item in bucket.pop("_pending", []):
        _launch_deferred(bucket, item)
    joined = set()
    for param, g in bucket.values():
        dev = param.device
        cur = torch.cuda.current_stream(dev)
        if dev not in joined:
            cur.wait_stream(_defer_stream(dev))
            joined.add(dev)
        g.record_stream(cur)
        if param.grad is None:
            param.grad = g
        else:
            param.grad.add_(g)


class _WeightCacheMixin:
    """Modules that cache derived weights in eval mode: the cache goes when the mode changes or a state dict is loaded
    (``p.data.copy_()`` style writes bypass the version counter the cache is keyed on -- call
    ``invalidate_weight_cache()`` after such a write)."""

    _weight_cache_attrs = ("_eval_wp", "_weight_images", "_deep_images", "_head_w")

    def invalidate_weight_cache(self) -> None:
        for name in self._weight_cache_attrs:
            if name in self.__dict__:
                del self.__dict__[name]

    def train(self, mode: bool = True):
        self.invalidate_weight_cache()
        return super().train(mode)

    def _load_from_state_dict(self, *args, **kwargs):
        self.invalidate_weight_cache()
        return super()._load_from_state_dict(*args, **kwargs)


class WgradTable:
    """Records ``(a_off, b_off, M, N, d, out_off)`` of one launch, kept as a host buffer (kernel arguments)."""

    def __init__(self, records: Sequence[Tuple[int, int, int, int, int, int]], out_stride: int):
        self.records = [tuple(int(v) for v in r) for r in records]
        self.out_stride = int(out_stride)
        raw = b"".join(struct.pack("<6i", *r) for r in self.records)
        self.buf = ctypes.create_string_buffer(raw, max(len(raw), 1))
        self.covered = sum(r[2] * r[3] for r in self.records) == self.out_stride
        self.flops_per_row = 2.0 * sum(r[2] * r[3] * r[4] for r in self.records)

    def __len__(self):
        return len(self.records)


def supported(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32


def wgrad(a: torch.Tensor, b: torch.Tensor, table: WgradTable, types: Optional[torch.Tensor] = None,
          n_types: int = 1) -> torch.Tensor:
    """``out[t, out_off + i*N + j] = sum_{z: type z = t} sum_m a[z, a_off + i*d + m] b[z, b_off + j*d + m]`` ->
    ``[n_types, out_stride]`` (float32, CUDA)."""
    if not supported(a, b):
        raise RuntimeError("nqa_wgrad needs float32 CUDA operands (there is no CPU path)")
    assert a.dim() == 2 and b.dim() == 2 and a.shape[0] == b.shape[0]
    a, b = a.contiguous(), b.contiguous()
    Z = a.shape[0]
    lib = _lib.load()
    tab = ctypes.cast(table.buf, ctypes.c_void_p)
    S = lib.nqa_wgrad_splits(tab, len(table), n_types, Z)
    if S < 1:
        _lib.check(S, "nqa_wgrad_splits")
    alloc = torch.empty if table.covered else torch.zeros
    partials = alloc((S, n_types, table.out_stride), dtype=torch.float32, device=a.device)
    tptr = ctypes.c_void_p(types.data_ptr()) if (types is not None and n_types > 1) else ctypes.c_void_p()
    nbytes = 4.0 * (a.numel() + b.numel())
    with torch.cuda.device(a.device), ktimer.region("wgrad", nbytes, table.flops_per_row * Z):
        rc = lib.nqa_wgrad(_lib.NQA_F32, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), tptr, tab,
                           len(table), a.shape[1], b.shape[1], Z, n_types, table.out_stride, S,
                           ctypes.c_void_p(partials.data_ptr()),
                           ctypes.c_void_p(torch.cuda.current_stream(a.device).cuda_stream))
    _lib.check(rc, "nqa_wgrad")
    return partials.sum(0) if S > 1 else partials[0]

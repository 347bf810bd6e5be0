"""Capture-after-warm-up / replay of a step function as a HIP graph (the role of the reference's ``CUDAGraphed``,
``MLLM_v2/utils/compile.py:189-277``; ``NO_CUDA_GRAPH=1`` disables it as ``compile.py:168-174`` does)."""
from __future__ import annotations

import gc
import os
from contextlib import contextmanager
from typing import Any, List, Optional

import torch

_in_graph = False      # a Graphed call is on the stack: nothing below it captures a graph of its own (compile.py:145-165)


def in_cuda_graph() -> bool:
    return _in_graph


@contextmanager
def _set_in_graph():
    global _in_graph
    old, _in_graph = _in_graph, True
    try:
        yield
    finally:
        _in_graph = old


class Graphed:
    """Capture-after-warm-up / replay wrapper over a function whose TENSOR arguments are top-level positional arguments of static
    shape (the role of the reference's CUDAGraphed, utils/compile.py:189-277).  As there: keyword arguments are refused; a
    non-tensor argument is baked into the capture, so a later call must pass an equal value (and a tensor must stay a tensor of the
    same shape); a Graphed function called from inside another Graphed call runs plainly (one capture, the outer one).  Disabled
    on request or by NO_CUDA_GRAPH=1."""

    def __init__(self, fn, warmup: int = 1, disable: bool = False):
        self.fn, self.warmup, self.calls = fn, warmup, 0
        self.disable = disable or os.environ.get("NO_CUDA_GRAPH", "") not in ("", "0")
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_in: List[Any] = []
        self.static_out = None

    def reset(self, warmup_steps: int = 0) -> None:
        """The next call (after ``warmup_steps`` plain ones) captures again: shapes or external state (KV rings) changed."""
        self.graph, self.calls, self.warmup = None, 0, warmup_steps
        self.static_in, self.static_out = [], None

    def __call__(self, *args, **kwargs):
        if kwargs:
            raise RuntimeError(f"a graphed function takes positional arguments only (got keywords {sorted(kwargs)})")
        if self.disable or in_cuda_graph():
            return self.fn(*args)
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        with _set_in_graph():
            if dev is not None and dev.index != torch.cuda.current_device():
                with torch.cuda.device(dev):        # capture / replay on the device the arguments live on
                    return self._call(*args)
            return self._call(*args)

    @staticmethod
    def _slot(a) -> tuple:
        """What a capture bakes in about one argument: a tensor's shape, or a plain value itself."""
        return ("tensor", tuple(a.shape)) if isinstance(a, torch.Tensor) else ("value", a)

    def _refresh_inputs(self, args) -> None:
        """Feed a replay: every argument is compared with what the capture baked in BEFORE anything is copied (a refused call
        leaves the static inputs untouched), then the tensors are copied into the captured buffers.  ValueError on a different
        argument count, a tensor / plain-value swap, another shape or another plain value (the exception type of
        utils/compile.py:231-256)."""
        problems = []
        if len(args) != len(self.static_in):
            problems.append(f"captured with {len(self.static_in)} positional arguments, called with {len(args)}")
        for i, (new, held) in enumerate(zip(args, self.static_in)):
            (kind_n, what_n), (kind_h, what_h) = self._slot(new), self._slot(held)
            if kind_n != kind_h:
                problems.append(f"argument {i}: captured as a {kind_h}, called with a {kind_n}")
            elif kind_h == "tensor" and what_n != what_h:
                problems.append(f"argument {i}: captured with shape {what_h}, called with shape {what_n}")
            elif kind_h == "value" and not (new is held or new == held):
                problems.append(f"argument {i}: the capture baked in {held!r}, called with {new!r}")
        if problems:
            raise ValueError("graphed call does not match its capture: " + "; ".join(problems))
        for new, held in zip(args, self.static_in):
            if isinstance(held, torch.Tensor):
                held.copy_(new)

    def _call(self, *args):
        if self.graph is None:
            self.calls += 1
            if self.calls <= self.warmup:
                return self.fn(*args)
            self.static_in = [a.clone() if isinstance(a, torch.Tensor) else a for a in args]
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            # No garbage collection while the stream is capturing: a dead reference cycle that holds another session's captured graph
            # (a finished generator, a closed pipeline) would be finalised at an arbitrary point of the capture, and destroying a graph
            # or freeing its pool is an illegal call then -- the process aborts from inside the destructor (seen once in ~10 runs of the
            # GPU suite, tests/test_gpt_gpu.py).  Collect such cycles BEFORE the capture starts (torch.cuda.graph no longer does by
            # default) and keep the collector off until it has ended.
            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            try:
                with torch.cuda.graph(self.graph):
                    self.static_out = self.fn(*self.static_in)
            finally:
                if gc_was_on:
                    gc.enable()
            # the capture itself does not execute: the replay below produces this call's result
            self.graph.replay()
            return self.static_out
        self._refresh_inputs(args)
        self.graph.replay()
        return self.static_out


CUDAGraphed = Graphed       # the reference's name

"""Capture-after-warm-up / replay of a step function as a HIP graph (the role of the reference's ``CUDAGraphed``,
``MLLM_v2/utils/compile.py:189-277``; ``NO_CUDA_GRAPH=1`` disables it as ``compile.py:168-174`` does)."""
from __future__ import annotations

import os
from typing import List, Optional

import torch


class Graphed:
    """Capture-after-warm-up / replay wrapper over a function of static-shaped device tensors (the role of the
    reference's CUDAGraphed, utils/compile.py:189-277).  Disabled on request or by NO_CUDA_GRAPH=1."""

    def __init__(self, fn, warmup: int = 1, disable: bool = False):
        self.fn, self.warmup, self.calls = fn, warmup, 0
        self.disable = disable or os.environ.get("NO_CUDA_GRAPH", "") not in ("", "0")
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self.static_in: List[torch.Tensor] = []
        self.static_out = None

    def reset(self) -> None:
        self.graph, self.calls = None, 0

    def __call__(self, *args: torch.Tensor):
        if self.disable:
            return self.fn(*args)
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if dev is not None and dev.index != torch.cuda.current_device():
            with torch.cuda.device(dev):        # capture / replay on the device the arguments live on
                return self._call(*args)
        return self._call(*args)

    def _call(self, *args: torch.Tensor):
        if self.graph is None:
            self.calls += 1
            if self.calls <= self.warmup:
                return self.fn(*args)
            self.static_in = [a.clone() for a in args]
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = self.fn(*self.static_in)
            # the capture itself does not execute: replay below produces this call's result
        for s, a in zip(self.static_in, args):
            if s.shape != a.shape:
                raise RuntimeError(f"graphed call with a different shape: {tuple(a.shape)} vs {tuple(s.shape)}")
            s.copy_(a)
        self.graph.replay()
        return self.static_out



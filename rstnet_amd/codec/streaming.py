"""Streaming-state protocol of the codec / LM modules.

Same surface as the reference's ``StreamingModule`` family (``modules/streaming.py:33-194`` of the
MimiCodec tokenizer copy): ``streaming(batch_size)`` context manager, ``streaming_forever``,
``reset_streaming``, ``get_streaming_state`` / ``set_streaming_state``, ``set_streaming_propagate`` and the
``is_streaming`` flag.  States are small dataclasses whose tensors live on the module's device and are handed
to the stateless C ABI as plain pointers.
"""
from __future__ import annotations

import abc
from contextlib import contextmanager
from dataclasses import dataclass
from typing import Any, Callable, Dict, Generic, Optional, TypeVar

import torch
from torch import nn

State = TypeVar("State")


class StreamingModule(abc.ABC, nn.Module, Generic[State]):
    def __init__(self) -> None:
        super().__init__()
        self._streaming_state: Optional[State] = None
        self._streaming_propagate: bool = True

    @property
    def is_streaming(self) -> bool:
        return self._streaming_state is not None

    def set_streaming_propagate(self, streaming_propagate: bool) -> None:
        self._streaming_propagate = streaming_propagate

    def _apply_named_streaming(self, fn: Callable[[str, "StreamingModule"], None]) -> None:
        """Depth-first walk over this module and its streaming descendants.  A module whose
        ``_streaming_propagate`` is False is skipped together with its subtree -- except at the root, where only
        the root itself is skipped and its children are still visited (the reference relies on this for the depth
        transformer, ``modules/streaming.py:67-84``)."""
        def visit(prefix: str, module: nn.Module) -> None:
            if isinstance(module, StreamingModule):
                if not module._streaming_propagate:
                    return
                fn(prefix, module)
            for name, child in module.named_children():
                visit(f"{prefix}.{name}", child)

        if self._streaming_propagate:
            fn("", self)
        for name, child in self.named_children():
            visit(name, child)

    @abc.abstractmethod
    def _init_streaming_state(self, batch_size: int) -> State:
        ...

    def _start_streaming(self, batch_size: int) -> None:
        def start(_: str, m: "StreamingModule") -> None:
            m._streaming_state = m._init_streaming_state(batch_size)
        self._apply_named_streaming(start)

    def _stop_streaming(self) -> None:
        def stop(_: str, m: "StreamingModule") -> None:
            m._streaming_state = None
        self._apply_named_streaming(stop)

    def streaming_forever(self, batch_size: int) -> None:
        self._start_streaming(batch_size)

    @contextmanager
    def streaming(self, batch_size: int):
        self._start_streaming(batch_size)
        try:
            yield
        finally:
            self._stop_streaming()

    def reset_streaming(self) -> None:
        def reset(name: str, m: "StreamingModule") -> None:
            if m._streaming_state is None:
                raise ValueError(f"Trying to reset streaming, but {name} wasn't streaming.")
            m._streaming_state.reset()
        self._apply_named_streaming(reset)

    def get_streaming_state(self) -> Dict[str, Any]:
        out: Dict[str, Any] = {}

        def collect(name: str, m: "StreamingModule") -> None:
            out[name] = m._streaming_state
        self._apply_named_streaming(collect)
        return out

    def set_streaming_state(self, state: Dict[str, Any]) -> None:
        pending = dict(state)

        def assign(name: str, m: "StreamingModule") -> None:
            if name not in pending:
                raise RuntimeError(f"Expected to find a streaming state for {name}.")
            m._streaming_state = pending.pop(name)
        self._apply_named_streaming(assign)
        if pending:
            raise RuntimeError(f"Some states were not consumed: {list(pending.keys())}")


@dataclass
class _NullState:
    def reset(self) -> None:
        pass


class StreamingContainer(StreamingModule[_NullState]):
    def _init_streaming_state(self, batch_size: int) -> _NullState:
        return _NullState()


@dataclass
class _StreamingAddState:
    previous_x: Optional[torch.Tensor] = None
    previous_y: Optional[torch.Tensor] = None

    def reset(self) -> None:
        self.previous_x = None
        self.previous_y = None


class StreamingAdd(StreamingModule[_StreamingAddState]):
    """``x + y`` on ``[B, C, T]`` tensors; when streaming, holds back the surplus of the longer operand
    (``modules/streaming.py:162-194``).  Inside the SEANet residual blocks the addition is fused into the
    epilogue of the 1x1 convolution, where both operands always have the same length."""

    def _init_streaming_state(self, batch_size: int) -> _StreamingAddState:
        return _StreamingAddState()

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        state = self._streaming_state
        if state is None:
            return x + y
        if state.previous_x is not None:
            x = torch.cat([state.previous_x, x], dim=-1)
        if state.previous_y is not None:
            y = torch.cat([state.previous_y, y], dim=-1)
        n = min(x.shape[-1], y.shape[-1])
        state.previous_x, state.previous_y = x[..., n:], y[..., n:]
        return x[..., :n] + y[..., :n]

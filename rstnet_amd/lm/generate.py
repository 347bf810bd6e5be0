"""Streaming generation for the ``GPT`` backbone: the offline loop of ``MLLM_v2/infer_no_streaming.py:168-323``
(``InferenceImp.__call__`` + ``reverse_delay``) with the same prompt handling, sampling rules and stop rule, but O(T):

  * the prompt ``[initial frame | prefix]`` is pushed through the global transformer ONCE (prefill: all positions per launch,
    bf16-MFMA skinny GEMMs + multi-query ring attention), every generated frame afterwards is one graph-replayed T = 1 step
    -- the reference re-runs ``forward_global`` over the whole sequence for every frame (:240);
  * the eight audio tokens of a frame come from eight depth-transformer steps (one captured graph, sampling included) -- the
    reference re-runs the teacher-forced ``forward_local`` over the whole prefix for every codebook (:259).

Both replacements are exact restatements of what the reference computes for the LAST position: its non-streaming passes see
positions 0..t with the causal + context mask and no ring, which a ring of capacity context + 1 (global) / dep_q + 1 (depth)
reproduces -- a ring that size never lets the `delta <= 0` slot-map quirk hide a key the mask would show.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch

from .. import ops
from ..graphs import Graphed as _Graphed
from .gpt import GPT, _GPTState


def reverse_delay(x: torch.Tensor) -> torch.Tensor:
    """infer_no_streaming.py:311-323: codebooks 1..7 are generated one frame late; x ``[8, L]`` (or ``[L, 8]``) -> ``[8, L-1]``."""
    if x.shape[0] != 8:
        x = x.transpose(0, 1)
    out = torch.ones_like(x)
    out[0, :-1] = x[0, :-1]
    out[1:, :-1] = x[1:, 1:]
    return out[:, :-1]


@dataclass
class GenIds:
    """The literal ids of InferenceImp.__init__ (:178-183) and the sampling helpers; parameters here so that small test
    vocabularies can be used."""
    text_pad_token: int = 128003
    text_empty_token: int = 128002
    semantic_pad_token: int = 2049
    acoustic_pad_token: int = 2049
    n_audio_codes: int = 2048
    text_initial_token_id: Optional[int] = None      # None = model.text_initial_token_id (151655)


class GPTGen:
    """Frame-by-frame generator over a ``GPT`` for B concurrent streams (the counterpart of ``LMGen`` for the litgpt-style
    backbone): ``prefill`` pushes a prompt through the global transformer, ``frame`` draws the text token and the dep_q audio
    tokens of the next frame (one captured graph), ``advance`` feeds the completed frame back (one captured graph).
    Rings: capacity context + 1 (global) and dep_q + 1 (depth) -- see the module docstring."""

    def __init__(self, model: GPT, use_sampling: bool = True, temp: float = 0.8, temp_text: float = 0.7, top_k: int = 250,
                 top_k_text: int = 25, n_audio_codes: int = 2048,
                 noise: Optional[Callable[[str, int, int], torch.Tensor]] = None):
        self.model = model
        self.use_sampling, self.temp, self.temp_text, self.top_k, self.top_k_text = use_sampling, temp, temp_text, top_k, top_k_text
        self.n_audio_codes = n_audio_codes
        self.noise = noise           # test hook: Exp(1) draws (kind, g_idx, l_idx) -> [B, k]; disables graph capture
        self._saved = None
        self._limits: Optional[torch.Tensor] = None
        self._depth: Optional[_Graphed] = None
        self._fused: Optional[_Graphed] = None       # `step`: the whole frame as one graph on the session's token column `_col`
        self._col: Optional[torch.Tensor] = None
        self._regime = None
        self.B = 0

    def begin(self, batch_size: int) -> None:
        m = self.model
        cfg, dev = m.config, m.device
        eager = self.noise is not None or dev.type != "cuda"
        self._saved = (m.transformer._streaming_state, m._streaming_state, m.codecformer._streaming_state)
        m.transformer._streaming_state = m.transformer._make_state(batch_size, cfg.context + 1)
        m._streaming_state = _GPTState(_Graphed(m._global_step, disable=eager))
        m.codecformer._streaming_state = m.codecformer._init_streaming_state(batch_size, capacity=cfg.dep_q + 1)
        self._limits = torch.full((cfg.dep_q,), self.n_audio_codes + 1, device=dev, dtype=torch.int32)
        self._depth_pos = torch.arange(cfg.dep_q, device=dev, dtype=torch.long)
        self._depth = _Graphed(self._depth_frame, disable=eager)
        self._regime, self.B, self._eager = None, batch_size, eager
        self._frames, self._persist_epoch = 0, ops.persistent_epoch(dev)

    def end(self) -> None:
        m = self.model
        m.transformer._streaming_state, m._streaming_state, m.codecformer._streaming_state = self._saved
        self._saved = None
        self._fused = self._col = None       # the fused frame graph of `start` / `step` belongs to the session that ends here

    def _exp_noise(self, kind: str, g_idx: int, l_idx: int, B: int, k: int) -> Optional[torch.Tensor]:
        if not self.use_sampling:
            return None
        if self.noise is not None:
            return self.noise(kind, g_idx, l_idx).reshape(B, k).to(self.model.device, torch.float32).contiguous()
        return torch.empty(B, k, device=self.model.device, dtype=torch.float32).exponential_(1)    # utils/sampling.py:44-46

    def _frame_noise(self, g_idx: int, B: int, with_text: bool) -> Optional[torch.Tensor]:
        """Exp(1) noise of one frame's samplers, ``[B, (k_text +) dep_q * k]``: ONE draw (utils/sampling.py:44-46 draws per call; the
        values are i.i.d. either way), or the test hook's per-sampler draws side by side."""
        if not self.use_sampling:
            return None
        cfg = self.model.config
        k_text, k_eff = min(self.top_k_text, cfg.padded_vocab_size), min(self.top_k, cfg.audio_card)
        if self.noise is not None:
            parts = [self._exp_noise("text", g_idx, 0, B, k_text)] if with_text else []
            return torch.cat(parts + [self._exp_noise("audio", g_idx, l_idx, B, k_eff) for l_idx in range(cfg.dep_q)], 1)
        n = (k_text if with_text else 0) + cfg.dep_q * k_eff
        return torch.empty(B, n, device=self.model.device, dtype=torch.float32).exponential_(1)

    def _depth_into(self, tokens: torch.Tensor, h: torch.Tensor, noise: Optional[torch.Tensor]) -> None:
        """dep_q depth-transformer steps + sampling IN PLACE: ``tokens`` int64 ``[B, >= dep_q + 1]`` (any row stride) holds the text
        token in column 0; step l embeds column l and samples column l + 1 (the layout of ``LMGen._depth``).  ``noise``: Exp(1)
        draws ``[B, dep_q * k]`` or None (greedy)."""
        m = self.model
        dep, cfg = m.codecformer, m.config
        B = tokens.shape[0]
        k_eff = min(self.top_k, cfg.audio_card)
        h_all = ops.lm_linear(h, m.codecformer_in_all())      # codecformer_in[k](h) of all dep_q steps in one launch
        E, H = dep.d_model, dep.num_heads
        Hd = dep.layers[0].gating[0].linear_out.weight.shape[1]
        if ops.depth_frame_enabled(tokens.device) and ops.depth_frame_supported(B, E, H, Hd, cfg.audio_card, cfg.dep_q, len(dep.layers), k_eff, device=tokens.device):
            # batch 1 / 2: the dep_q steps with their samplers are one persistent launch (csrc/lm_depth.hip), on a dense [B, dep_q + 1] buffer
            dense = tokens if tokens.shape[1] == cfg.dep_q + 1 and tokens.is_contiguous() else tokens[:, :cfg.dep_q + 1].contiguous()
            self._tables = m.depth_frame_tables()      # a captured frame embeds the tables' device pointers: kept alive with the graph
            ops.depth_decode_frame(self._tables, h_all, dense, noise, use_sampling=self.use_sampling, temp=self.temp, top_k=k_eff,
                                   eps=dep.layers[0].norm1.eps, context=dep.context, limits=self._limits,
                                   ring_cap=dep._streaming_state.k[0].shape[2])
            if dense is not tokens:
                tokens[:, 1:cfg.dep_q + 1] = dense[:, 1:]
            return
        for l_idx in range(cfg.dep_q):
            add = h_all[:, l_idx * E:(l_idx + 1) * E]
            table = m.codecformer_text_emb.weight if l_idx == 0 else m.codecformer_emb[l_idx - 1].weight
            # positions 0 .. dep_q - 1 of a ring that restarts every frame, as constant device scalars (no counter to zero and bump:
            # nine glue launches per frame); the step's input is formed inside its first launch
            y = dep.step(None, step_index=l_idx, pos=self._depth_pos[l_idx:l_idx + 1], embed=(add, table, tokens, l_idx))
            head = m.audio_linears[l_idx]
            logits = ops.lm_linear(y, head.weight, bias=head.bias_f32())
            ops.lm_sample(logits, use_sampling=self.use_sampling, temp=self.temp, top_k=k_eff,
                          noise=None if noise is None else noise[:, l_idx * k_eff:(l_idx + 1) * k_eff],
                          limit_dev=self._limits[l_idx:l_idx + 1], out=tokens[:, l_idx + 1])

    def _depth_frame(self, text_token: torch.Tensor, h: torch.Tensor, g_idx: int = 0) -> torch.Tensor:
        """dep_q depth-transformer steps + sampling: text_token int64 [B], h fp32 [B, n_embd] -> tokens int64 [B, dep_q]."""
        (B,) = text_token.shape
        tokens = torch.empty(B, self.model.config.dep_q + 1, device=text_token.device, dtype=torch.long)
        tokens[:, 0] = text_token
        self._depth_into(tokens, h, self._frame_noise(g_idx, B, with_text=False))
        return tokens[:, 1:].contiguous()

    # ---- the whole frame as ONE captured graph (serving / benchmark loop): global step of the previous frame's tokens, text sample,
    # dep_q depth steps with their samples -- fed by nothing, reading and writing the session's token column on the device
    def start(self, h: torch.Tensor, logits: torch.Tensor, g_idx: int = 0):
        """First frame after ``prefill``: samples it (``frame``) and loads it into the session's token column, from which ``step``
        continues.  Returns (text [B], audio [B, dep_q])."""
        text, audio = self.frame(h.contiguous(), logits.contiguous(), g_idx)
        cfg = self.model.config
        self._col = torch.full((h.shape[0], cfg.n_q + 1), self.model.initial_token_id, device=h.device, dtype=torch.long)
        self._col[:, 0] = text
        self._col[:, 1:cfg.dep_q + 1] = audio
        self._fused = _Graphed(self._step_fn, disable=self._eager)
        self._g_idx = g_idx
        return text, audio

    def _step_fn(self):
        m, cfg, col = self.model, self.model.config, self._col
        B = col.shape[0]
        # the completed frame through the global transformer (rows beyond dep_q carry the initial-token pad, infer_no_streaming.py:243-244);
        # its embedding launch reads the column before the samplers below overwrite it (stream order)
        h, logits = m._global_step(col)
        k_text = min(self.top_k_text, cfg.padded_vocab_size)
        noise = self._frame_noise(self._g_idx, B, with_text=True)
        ops.lm_sample(logits, use_sampling=self.use_sampling, temp=self.temp_text, top_k=k_text,
                      noise=None if noise is None else noise[:, :k_text], out=col[:, 0])
        self._depth_into(col, h, None if noise is None else noise[:, k_text:])
        return h, logits

    def step(self):
        """``advance`` of the previous frame + ``frame`` of the next one as ONE graph replay with no host-side tensor traffic: returns
        (text [B], audio [B, dep_q]) as VIEWS of the session's token column (valid until the next ``step``; clone to keep)."""
        col, cfg = self._col, self.model.config
        if self._frames % 64 == 0 and col.is_cuda and not self._eager:
            ops.persistent_poll(col.device)
            if self._persist_epoch != ops.persistent_epoch(col.device):
                self._persist_epoch = ops.persistent_epoch(col.device)
                self._fused = _Graphed(self._step_fn)
        self._frames += 1
        self._g_idx += 1
        self.last_h, self.last_logits = self._fused()
        return col[:, 0], col[:, 1:cfg.dep_q + 1]

    def set_blanking(self, wide: list) -> None:
        """Per-codebook id limit of the next frames: n_audio_codes + 1 where ``wide`` (sample_token_audio) else n_audio_codes
        (sample_token_audio_2048)."""
        n = self.n_audio_codes
        self._limits.copy_(torch.tensor([n + 1 if w else n for w in wide], dtype=torch.int32))

    def prefill(self, tokens: torch.Tensor):
        """tokens int64 [B, K, T] -> (h [B, n_embd], logits [B, V]) of the LAST position."""
        h, logits = self.model.forward_global(tokens)
        return h[:, -1].contiguous(), logits[:, -1].contiguous()

    def frame(self, h: torch.Tensor, logits: torch.Tensor, g_idx: int = 0):
        """(h, logits) of the last position -> (text token [B], audio tokens [B, dep_q]) of the next frame."""
        B = h.shape[0]
        if self._frames % 64 == 0 and h.is_cuda and not self._eager:
            # health of the persistent depth launch (csrc/persist.h): a device that had to repair frames moves to the launch-per-op chain
            ops.persistent_poll(h.device)
            if self._persist_epoch != ops.persistent_epoch(h.device):
                self._persist_epoch = ops.persistent_epoch(h.device)
                self._depth = _Graphed(self._depth_frame)
        self._frames += 1
        k_text = min(self.top_k_text, self.model.config.padded_vocab_size)
        text = ops.lm_sample(logits, use_sampling=self.use_sampling, temp=self.temp_text, top_k=k_text,
                             noise=self._exp_noise("text", g_idx, 0, B, k_text))
        audio = self._depth_frame(text, h, g_idx) if self._eager else self._depth(text, h)
        return text, audio

    def advance(self, text: torch.Tensor, audio: torch.Tensor):
        """Feed the completed frame (rows beyond dep_q carry the initial-token pad, infer_no_streaming.py:243-244)."""
        m = self.model
        cfg = m.config
        col = torch.full((text.shape[0], cfg.n_q + 1, 1), m.initial_token_id, device=text.device, dtype=torch.long)
        col[:, 0, 0] = text
        col[:, 1:cfg.dep_q + 1, 0] = audio
        h, logits = m.forward_global(col)
        return h[:, 0], logits[:, 0]


class InferenceImp:
    """Same constructor and call convention as infer_no_streaming.py:168-190 (``args`` is kept for signature parity and unused).
    ``__call__(seq [K, L], mask)`` returns what the reference returns for task 'TTS' (``reverse_delay`` of the generated
    frames, ``[8, n-1]``); ``generate`` returns every product of the loop for any task."""

    def __init__(self, args, model: GPT, mode: str, temp_text: float, top_k_text: int, temp: float, top_k: int, task_name: str,
                 use_sampling: bool = True, ids: Optional[GenIds] = None,
                 noise: Optional[Callable[[str, int, int], torch.Tensor]] = None):
        self.model, self.args, self.mode, self.task_name = model, args, mode, task_name
        self.n_samples = 1
        self.use_sampling, self.temp_text, self.top_k_text, self.temp, self.top_k = use_sampling, temp_text, top_k_text, temp, top_k
        self.ids = ids or GenIds()
        self.noise = noise           # test hook: Exp(1) draws (kind, g_idx, l_idx) -> [1, k]; disables graph capture

    # ---- prompt handling (:190-230)
    def split_prompt(self, seq: torch.Tensor):
        ids, task = self.ids, self.task_name
        if task in ("text_only", "word_level_audio_text_alignment", "ASR"):
            pad_len = int(seq[0, 0:1, :].eq(ids.text_pad_token).int().sum())
        elif task in ("audio_only", "TTS"):
            pad_len = int(seq[0, 1:2, :].eq(ids.semantic_pad_token).int().sum())
        else:
            raise NotImplementedError
        seq = seq[:, :, :seq.shape[2] - pad_len]
        if task in ("text_only", "audio_only"):
            n = seq.shape[-1] // 2
            return seq[:, :, :n], n, n
        if task == "TTS":
            n_prefix = seq.shape[2] - int(seq[0, 0, :].eq(ids.text_empty_token).int().sum())
            n = seq.shape[2] - n_prefix
            return seq[:, :, :n_prefix], n, n
        if task == "ASR":
            n_prefix = int(seq[0, 0, :].eq(ids.text_empty_token).int().sum())
            return seq[:, :, :n_prefix + 1], seq.shape[2] - n_prefix + 13, seq.shape[2] - n_prefix - 13
        raise NotImplementedError

    @torch.no_grad()
    def generate(self, seq: torch.Tensor, mask: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """seq int64 ``[K, L]`` -> {'frames': [n, dep_q], 'text': [n]} (+ 'codes' = reverse_delay(frames) for task 'TTS' with
        8 codebooks)."""
        m, ids = self.model, self.ids
        cfg, dev = m.config, m.device
        seq = seq.to(dev).unsqueeze(0)
        prefix, maxlen, minlen = self.split_prompt(seq)
        init = m._get_initial_token()
        if ids.text_initial_token_id is not None:
            init[:, 0] = ids.text_initial_token_id
        pre_gen_len = prefix.shape[2]
        n_codes = ids.n_audio_codes
        gen = GPTGen(m, self.use_sampling, self.temp, self.temp_text, self.top_k, self.top_k_text, n_codes, self.noise)
        gen.begin(1)
        frames, texts = [], []
        try:
            h, logits = gen.prefill(torch.cat([init, prefix], dim=-1))      # positions 0 .. pre_gen_len
            regime = None
            for g_idx in range(maxlen):
                g_len = pre_gen_len + g_idx
                # blanking regime of this frame (:264-283): first frame -> 2049 everywhere; later l = 0 -> 2048,
                # l > 0 -> 2049 once g_len > minlen else 2048
                new_regime = 0 if g_len == pre_gen_len else (1 if g_len > minlen else 2)
                if new_regime != regime:
                    gen.set_blanking([True] * cfg.dep_q if new_regime == 0 else [l > 0 and new_regime == 1 for l in range(cfg.dep_q)])
                    regime = new_regime
                text, audio = gen.frame(h.contiguous(), logits.contiguous(), g_idx)
                audio_host = audio[0].tolist()
                if g_idx > minlen and any(t >= n_codes for t in audio_host[3:]):     # the stop rule of :286-288
                    break
                frames.append(audio[0].clone())
                texts.append(text[0].clone())
                if g_idx + 1 == maxlen:
                    break
                h, logits = gen.advance(text, audio)
        finally:
            gen.end()
        out = {"frames": torch.stack(frames) if frames else torch.zeros(0, cfg.dep_q, dtype=torch.long, device=dev),
               "text": torch.stack(texts) if texts else torch.zeros(0, dtype=torch.long, device=dev)}
        if self.task_name == "TTS" and frames and cfg.dep_q == 8:
            out["codes"] = reverse_delay(out["frames"])
        return out

    @torch.no_grad()
    def __call__(self, seq: torch.Tensor, mask: Optional[torch.Tensor] = None):
        if self.mode == "teacher-force":
            raise NotImplementedError("teacher-forced loss evaluation (training-side metric) is outside the generation path")
        out = self.generate(seq, mask)
        if self.task_name == "TTS":
            return out.get("codes", out["frames"])
        return out["frames"]

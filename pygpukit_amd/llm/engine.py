"""Python handle on the native decode/prefill engine (pgk_engine_* in include/pgk_hip.h).

One C call enqueues a whole token step (5 fused kernels per layer); `capture()` records it into ONE
hipGraph whose token id / position live in device memory, so `replay(n)` runs n greedy steps with no
host round trip.  This is what DecodeM1Graph / DecodeBatch drive; the reference's counterpart is the
2L+2-graph replay loop of src/pygpukit/llm/decode/m1_graph.py:463-589."""

from __future__ import annotations

import ctypes as C

import numpy as np

from pygpukit_amd import _hip
from pygpukit_amd.core.array import GPUArray
from pygpukit_amd.core.dtypes import bfloat16, float32


class Engine:
    def __init__(self, config: dict, embed: GPUArray, layers: list[dict], final_norm: GPUArray, lm_head: GPUArray | None = None,
                 *, max_seq_len: int = 512, max_batch: int = 1, weight_format: str = "bf16", use_qk_norm: bool = True):
        """config: vocab_size, hidden_size, num_layers, num_heads, num_kv_heads, head_dim, intermediate_size,
        norm_eps, rope_theta.  layers[i]: GPUArrays attn_norm, w_qkv [(Hq+2Hkv)D, H], q_norm, k_norm, w_o,
        mlp_norm, w_gate_up [2I, H], w_down (+ s_qkv, s_o, s_gate_up, s_down for fp8)."""
        _hip.require_device()
        self.config = dict(config)
        self.max_seq_len, self.max_batch = max_seq_len, max_batch
        self.weight_format = weight_format
        self._keep = [embed, final_norm, lm_head, layers]  # keep the weights alive
        mc = _hip.ModelConfig(config["vocab_size"], config["hidden_size"], config["num_layers"], config["num_heads"],
                              config["num_kv_heads"], config["head_dim"], config["intermediate_size"], max_seq_len, max_batch,
                              float(config["norm_eps"]), float(config["rope_theta"]), {"bf16": 0, "fp8": 1, "fp8a8": 2}[weight_format],
                              1 if use_qk_norm else 0)
        for a in (embed, final_norm):
            if a.dtype != bfloat16:
                raise ValueError("Engine: embedding / norm weights must be bfloat16")
        arr = (_hip.LayerWeights * len(layers))()
        for i, lw in enumerate(layers):
            for name, _ in _hip.LayerWeights._fields_:
                t = lw.get(name)
                setattr(arr[i], name, t.data_ptr() if t is not None else None)
        h = C.c_void_p()
        _hip.call("pgk_engine_create", C.byref(mc), embed._p, lm_head._p if lm_head is not None else None, final_norm._p,
                  arr, C.byref(h))
        self._h = h.value
        self._captured_batch = 0
        self.last_prefill_logits: np.ndarray | None = None

    def __del__(self):
        try:
            if getattr(self, "_h", 0):
                _hip.call("pgk_engine_destroy", C.c_void_p(self._h))
        except Exception:
            pass
        self._h = 0

    @property
    def handle(self) -> C.c_void_p:
        return C.c_void_p(self._h)

    def bytes(self) -> tuple[int, int]:
        kv, ws = C.c_size_t(), C.c_size_t()
        _hip.call("pgk_engine_bytes", self.handle, C.byref(kv), C.byref(ws))
        return kv.value, ws.value

    # ------------------------------------------------------------------ prefill
    def prefill(self, tokens, seq: int = 0, start_pos: int = 0, *, want_last_logits: bool = True,
                all_logits: GPUArray | None = None) -> np.ndarray | None:
        """Run the prompt through the MFMA path, filling sequence slot `seq`'s KV rows; returns the last
        row's logits (fp32 [V]) when asked.  all_logits: optional bf16 [n, V] device buffer."""
        toks = np.ascontiguousarray(tokens, dtype=np.int32)
        out = np.empty(self.config["vocab_size"], np.float32) if want_last_logits else None
        _hip.call("pgk_engine_prefill", self.handle, seq, toks.ctypes.data_as(_hip.c_i32_p), len(toks), start_pos,
                  all_logits._p if all_logits is not None else None,
                  out.ctypes.data_as(C.POINTER(C.c_float)) if out is not None else None, None)
        if out is not None:
            self.last_prefill_logits = out
        return out

    # ------------------------------------------------------------------ decode
    def set_state(self, tokens, positions) -> None:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        p = np.ascontiguousarray(positions, dtype=np.int32)
        if len(t) != len(p):
            raise ValueError("set_state: tokens and positions must have the same length")
        _hip.call("pgk_engine_set_state", self.handle, t.ctypes.data_as(_hip.c_i32_p), p.ctypes.data_as(_hip.c_i32_p), len(t), None)

    def set_sampling(self, temperature: float, top_k: int = 0, top_p: float = 1.0, uniforms: np.ndarray | None = None,
                     n_steps: int = 256, seed: int | None = None) -> np.ndarray | None:
        """Switch the decode step from greedy argmax to a temperature / top-k / top-p draw made inside the step (and so
        inside a captured graph).  `uniforms` [n_steps, max_batch] float32 in [0,1) (drawn from `seed` when omitted) are
        queued on the device: step s uses row s % n_steps.  Returns the uniforms used (None when switching back to
        greedy with temperature <= 0).  Call before capture().  Calling again with the same parameters and the same
        number of rows only queues fresh uniforms; any other change (on/off, temperature, top_k, top_p, row count)
        drops the captured graph - capture() again before replay()."""
        key = (0.0,) if temperature <= 0 else (float(temperature), int(top_k), float(top_p),
                                                int(uniforms.shape[0]) if uniforms is not None else int(n_steps))
        if key != getattr(self, "_sampling_key", (0.0,)):
            self._captured_batch = 0
        self._sampling_key = key
        if temperature <= 0:
            _hip.call("pgk_engine_set_sampling", self.handle, C.c_float(0.0), 0, C.c_float(1.0), None, 0, None)
            return None
        if uniforms is None:
            uniforms = np.random.default_rng(seed).random((n_steps, self.max_batch), dtype=np.float32)
        u = np.ascontiguousarray(uniforms, dtype=np.float32)
        if u.ndim != 2 or u.shape[1] != self.max_batch:
            raise ValueError(f"set_sampling: uniforms must be [n_steps, {self.max_batch}]")
        _hip.call("pgk_engine_set_sampling", self.handle, C.c_float(temperature), int(top_k), C.c_float(top_p),
                  u.ctypes.data_as(C.POINTER(C.c_float)), u.shape[0], None)
        return u

    def decode_step(self, batch: int = 1) -> None:
        """Enqueue one eager (un-captured) step."""
        _hip.call("pgk_engine_decode_step", self.handle, batch, None)

    KERNEL_CLASSES = ("embed", "norm_qkv", "attn", "oproj", "gateup", "down", "lmhead", "argmax")

    def profile_step(self, batch: int = 1, n_iters: int = 8) -> dict:
        """Eager steps, every launch bracketed by its own start/stop event (the dispatch's begin -> end interval, what
        rocprofv3 --kernel-trace reports) -> {class: (avg_us_per_launch, launches_per_step)}."""
        ms = (C.c_float * 8)()
        cnt = (C.c_int * 8)()
        _hip.call("pgk_engine_profile_step", self.handle, batch, n_iters, ms, cnt, None)
        return {name: ((ms[i] * 1e3 / cnt[i]) if cnt[i] else 0.0, cnt[i] // n_iters) for i, name in enumerate(self.KERNEL_CLASSES)}

    def timeline(self, batch: int = 1, warm: int = 2) -> list[dict]:
        """One graph-replayed step as a list of launches in order: kernel class, workgroups, and the first/last start and
        first/last end over the launch's workgroups in microseconds from the step's first start (in-kernel
        s_memrealtime stamps; diagnostic capture, the engine's own graph is untouched)."""
        cap = 16 * self.config["num_layers"] + 64
        buf = (C.c_uint64 * (6 * cap))()
        n = C.c_int()
        _hip.call("pgk_engine_timeline", self.handle, batch, warm, buf, cap, C.byref(n), None)
        out = []
        for i in range(n.value):
            cls, nwg, s0, s1, e0, e1 = (int(buf[6 * i + k]) for k in range(6))
            out.append({"kernel": self.KERNEL_CLASSES[cls] if 0 <= cls < len(self.KERNEL_CLASSES) else str(cls), "workgroups": nwg,
                        "first_start_us": s0 / 100.0, "last_start_us": s1 / 100.0, "first_end_us": e0 / 100.0, "last_end_us": e1 / 100.0})
        return out

    def capture(self, batch: int = 1) -> None:
        _hip.call("pgk_engine_capture", self.handle, batch, None)
        self._captured_batch = batch

    def replay(self, n_steps: int = 1) -> None:
        _hip.call("pgk_engine_replay", self.handle, n_steps, None)

    def synchronize(self) -> None:
        _hip.call("pgk_stream_sync", None)

    def reset_log(self) -> None:
        _hip.call("pgk_engine_reset_log", self.handle, None)

    def read_tokens(self, batch: int, n_steps: int) -> np.ndarray:
        out = np.empty((n_steps, batch), np.int32)
        _hip.call("pgk_engine_read_tokens", self.handle, out.ctypes.data_as(_hip.c_i32_p), batch, n_steps, None)
        return out

    def shader_clock_mhz(self, n_steps: int) -> float:
        """In-kernel shader clock over the last n_steps logged steps: d(s_memtime)/d(s_memrealtime) x 100 MHz."""
        buf = np.zeros(2 * n_steps, np.uint64)
        _hip.call("pgk_engine_read_clock", self.handle, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n_steps, None)
        t, r = buf[0::2].astype(np.float64), buf[1::2].astype(np.float64)
        if n_steps < 2 or r[-1] == r[0]:
            return 0.0
        return float((t[-1] - t[0]) / (r[-1] - r[0]) * 100.0)

    def logits(self, batch: int = 1) -> GPUArray:
        """fp32 [batch, V] view of the last step's logits (owned by the engine)."""
        p = C.c_void_p()
        _hip.call("pgk_engine_logits_ptr", self.handle, C.byref(p))
        a = GPUArray((batch, self.config["vocab_size"]), float32, p.value, owns_memory=False)
        a._source_ref = self
        return a

    def kv_cache(self, layer: int) -> tuple[GPUArray, GPUArray]:
        """bf16 [max_batch, Hkv, max_seq, D] views of layer `layer`'s K and V caches."""
        k, v = C.c_void_p(), C.c_void_p()
        _hip.call("pgk_engine_kv_ptr", self.handle, layer, C.byref(k), C.byref(v))
        shape = (self.max_batch, self.config["num_kv_heads"], self.max_seq_len, self.config["head_dim"])
        ka, va = GPUArray(shape, bfloat16, k.value, owns_memory=False), GPUArray(shape, bfloat16, v.value, owns_memory=False)
        ka._source_ref = va._source_ref = self
        return ka, va

    def state_arrays(self, batch: int) -> tuple[GPUArray, GPUArray]:
        """int32 [batch] views of the device-resident (token, position) state."""
        from pygpukit_amd.core.dtypes import int32

        t, p = C.c_void_p(), C.c_void_p()
        _hip.call("pgk_engine_state_ptr", self.handle, C.byref(t), C.byref(p))
        ta, pa = GPUArray((batch,), int32, t.value, owns_memory=False), GPUArray((batch,), int32, p.value, owns_memory=False)
        ta._source_ref = pa._source_ref = self
        return ta, pa

    def launches_per_step(self) -> int:
        n = C.c_int()
        _hip.call("pgk_engine_launches_per_step", self.handle, C.byref(n))
        return n.value

    # ------------------------------------------------------------------ convenience
    def generate_greedy(self, prompt, max_new_tokens: int, *, use_graph: bool = True) -> list[int]:
        """Prompt + max_new_tokens greedy tokens for one sequence (slot 0): prefill, argmax of the last
        row (lowest index on ties), then max_new_tokens-1 device-resident decode steps."""
        prompt = [int(t) for t in prompt]
        logits = self.prefill(prompt, seq=0, start_pos=0)
        first = int(np.argmax(logits))
        out = prompt + [first]
        n = max_new_tokens - 1
        if n <= 0:
            return out[: len(prompt) + max_new_tokens]
        self.set_state([first], [len(prompt)])
        self.reset_log()
        if use_graph:
            if self._captured_batch != 1:
                self.capture(1)
            self.replay(n)
        else:
            for _ in range(n):
                self.decode_step(1)
        self.synchronize()
        return out + [int(t) for t in self.read_tokens(1, n)[:, 0]]

    def generate(self, prompt, max_new_tokens: int, temperature: float = 1.0, top_k: int = 50, top_p: float = 0.9, *,
                 seed: int | None = None) -> list[int]:
        """CausalTransformerModel.generate's sampling semantics (causal.py:179-255) with every draw made on the device:
        the first token from the prefill logits (ops.sample_token_gpu semantics via the oracle-restated sampler), the
        rest inside the captured decode graph (set_sampling).  temperature == 0 is generate_greedy."""
        if temperature == 0:
            return self.generate_greedy(prompt, max_new_tokens)
        from pygpukit_amd.core.factory import from_numpy
        from pygpukit_amd.ops.sampling import sample_token_gpu

        prompt = [int(t) for t in prompt]
        rng = np.random.default_rng(seed)
        logits = self.prefill(prompt, seq=0, start_pos=0)
        first = sample_token_gpu(from_numpy(np.ascontiguousarray(logits, dtype=np.float32)), temperature, top_k, top_p,
                                 u=float(rng.random(dtype=np.float32)))
        out = prompt + [first]
        n = max_new_tokens - 1
        if n <= 0:
            return out[: len(prompt) + max_new_tokens]
        self.reset_log()
        u = np.zeros((n, self.max_batch), np.float32)
        u[:, 0] = rng.random(n, dtype=np.float32)
        self.set_sampling(temperature, top_k, top_p, uniforms=u)
        self.set_state([first], [len(prompt)])
        self.capture(1)            # the sampling node is part of the graph
        self.replay(n)
        self.synchronize()
        toks = [int(t) for t in self.read_tokens(1, n)[:, 0]]
        self.set_sampling(0.0)
        self._captured_batch = None
        return out + toks

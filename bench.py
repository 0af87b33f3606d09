#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json metric "decode tokens/sec/GPU + prefill TFLOPS, Qwen3-0.6B bf16".

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

Workload at N=1 = BASELINE config 2: random-init Qwen3-0.6B-shape bf16 weights, 128-token prompt prefilled
through the MFMA path, then K single-token greedy decode steps, each ONE replay of the whole-step hipGraph
(embedding -> 28 layers incl. KV write + attention -> lm_head -> argmax, token/position in device memory).
A "step" = one decode token for every sequence resident on the GPU (--batch-per-gpu, default 1).
For N > 1 every rank holds a replica (weights broadcast from rank 0 over RCCL) and its own shard of the
independent sequences ("weak" scaling: per-GPU work fixed); the only in-loop collective is the all-gather
of the 4-byte sampled tokens.  `value` = sequences x K / max-over-ranks wall time, inputs resident in HBM.

Extra objects on the JSON line:
  roofline      dominant kernel (the lm_head weight-streaming GEMV, 26 % of the step's bytes):
                algorithmic bytes / hipEvent-measured launch duration vs 8 TB/s HBM.
  step_roofline whole decode step: algorithmic bytes per token / measured step time.
  prefill       ms, TFLOP/s and fraction of the 2.5 PFLOP/s dense bf16 MFMA peak for the 128-token prompt.
  cpu_baseline  the NumPy oracle (reference-semantics CPU path) timed on this box's host cores, rank 0, N=1.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def algorithmic_bytes_per_token(cfg: dict, ctx: int, weight_format: str) -> dict:
    """weights read once per step + un-expanded bf16 KV rows + KV write + logits."""
    H, I, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_layers"]
    D, Hq, Hkv = cfg["head_dim"], cfg["num_heads"], cfg["num_kv_heads"]
    lin = L * ((Hq + 2 * Hkv) * D * H + H * Hq * D + 2 * I * H + H * I)
    wbytes = lin * (1 if weight_format == "fp8" else 2)
    scales = (lin // (128 * 128)) * 2 if weight_format == "fp8" else 0
    norms = L * (2 * H + 2 * D) * 2 + H * 2
    lm = V * H * 2
    kv_row = L * 2 * Hkv * D * 2
    return {"weights": wbytes + scales + norms, "lm_head": lm, "kv_read": kv_row * ctx, "kv_write": kv_row,
            "logits": V * 4, "total": wbytes + scales + norms + lm + kv_row * ctx + kv_row + V * 4}


def prefill_flops(cfg: dict, S: int, all_rows: bool) -> float:
    H, I, V, L = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["num_layers"]
    D, Hq, Hkv = cfg["head_dim"], cfg["num_heads"], cfg["num_kv_heads"]
    per_tok = L * ((Hq + 2 * Hkv) * D * H + H * Hq * D + 3 * I * H)
    gemm = 2.0 * S * per_tok + 2.0 * (S if all_rows else 1) * V * H
    attn = 4.0 * S * S * D * Hq * L / 2
    return gemm + attn


def cpu_baseline(cfg, weights, prompt, n_decode: int) -> dict:
    """Reference-semantics NumPy path (oracle/cpu_ref.py == the reference's CPUSimulationBackend numerics,
    pinned by tests/golden) on this box's host cores: prefill the prompt, then time n_decode cached steps."""
    from oracle import cpu_ref as O

    try:
        from threadpoolctl import threadpool_info

        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    ref = O.build_qwen3_ref(cfg, weights, max_pos=len(prompt) + n_decode + 8)
    t0 = time.perf_counter()
    hidden, past = ref(prompt, use_cache=True)
    nxt = int(np.argmax(ref.get_logits(hidden[-1:])[0]))
    t_prefill = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(n_decode):
        hidden, past = ref([nxt], past_key_values=past, use_cache=True)
        nxt = int(np.argmax(ref.get_logits(hidden)[0]))
    dt = time.perf_counter() - t0
    return {"value": n_decode / dt, "unit": "tokens/s", "cores": int(threads), "kind": "port",
            "sample": f"NumPy oracle, fp32 on the same bf16-rounded weights: {len(prompt)}-token prefill "
                      f"({t_prefill:.1f} s) then {n_decode} cached decode steps ({dt:.1f} s)",
            "prefill_s": t_prefill}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch-per-gpu", type=int, default=1)
    ap.add_argument("--prompt-len", type=int, default=128)
    ap.add_argument("--weight-format", choices=["bf16", "fp8"], default="bf16")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cpu-decode-tokens", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--long-prefill", type=int, default=2048,
                    help="also time a prefill of this many tokens (the MFMA-bound regime); 0 = skip")
    ap.add_argument("--no-config5", action="store_true", help="skip the Llama-3-8B-shape fp8 prefill leg of 'extras' (~25 s)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs (batch 8 per GPU; context 2048 in bf16 and w8a16) reported under 'extras'")
    ap.add_argument("--layers", type=int, default=0, help="debug only: override the layer count (result is then INVALID)")
    args = ap.parse_args()

    from pygpukit_amd import _hip
    from pygpukit_amd.llm import synthetic as S
    from pygpukit_amd.parallel import ControlPlane, RcclComm

    cp = ControlPlane()
    if cp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={cp.world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    ndev = _hip.device_count()
    _hip.call("pgk_device_set", cp.local_rank % max(ndev, 1))
    # RCCL carries the one-time weight broadcast and the end-of-run token gather.  The decode path itself is pure
    # data parallelism (no collective inside a step), so if RCCL cannot come up (e.g. two ranks sharing one GPU
    # in a rehearsal) every rank draws the same seeded weights itself and the tokens travel over the gloo
    # control plane instead; the JSON line says which happened.
    comm, rccl_note = None, None
    if cp.world > 1:
        # every rank takes the same branch at every step: collectives on the control plane must stay matched
        if cp.min_over_ranks(1 if cp.local_rank < ndev else 0) == 0:
            rccl_note = f"a rank has no GPU of its own ({ndev} visible, {cp.world} ranks)"
        else:
            ok = 1
            try:
                comm = RcclComm(cp)
            except Exception as e:  # noqa: BLE001
                ok, rccl_note = 0, f"{type(e).__name__}: {e}"
            if cp.min_over_ranks(ok) == 0:
                if comm is not None:
                    comm.destroy()
                comm = None
            rccl_note = cp.first_note(rccl_note)

    cfg = dict(S.QWEN3_0_6B)
    if args.layers:
        cfg["num_layers"] = args.layers
    B, K, W, P = args.batch_per_gpu, args.steps, args.warmup, args.prompt_len
    max_seq = P + W + K + 8
    t_setup = time.perf_counter()

    # ---- weights: rank 0 draws them, every other rank receives them over RCCL (xGMI broadcast) ----
    weights = S.make_qwen3_weights(cfg, seed=args.seed) if cp.rank == 0 or comm is None else None
    bcast_s = None
    if comm is None:
        eng = S.build_engine_from_weights(cfg, weights, max_seq_len=max_seq, max_batch=B, weight_format=args.weight_format)
    else:
        from pygpukit_amd.core.array import GPUArray
        from pygpukit_amd.core.dtypes import bfloat16, uint8
        from pygpukit_amd.llm.engine import Engine

        H, I, V, D = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"], cfg["head_dim"]
        nq = (cfg["num_heads"] + 2 * cfg["num_kv_heads"]) * D
        fp8 = args.weight_format == "fp8"
        wdt = uint8 if fp8 else bfloat16
        shapes = {"attn_norm": ((H,), bfloat16), "mlp_norm": ((H,), bfloat16), "q_norm": ((D,), bfloat16), "k_norm": ((D,), bfloat16),
                  "w_qkv": ((nq, H), wdt), "w_o": ((H, cfg["num_heads"] * D), wdt), "w_gate_up": ((2 * I, H), wdt), "w_down": ((H, I), wdt)}
        if fp8:
            shapes.update({"s_qkv": ((nq // 128, H // 128), bfloat16), "s_o": ((H // 128, cfg["num_heads"] * D // 128), bfloat16),
                           "s_gate_up": ((2 * I // 128, H // 128), bfloat16), "s_down": ((H // 128, I // 128), bfloat16)})
        if cp.rank == 0:
            embed = S._bf16(weights["embed"])
            fnorm = S._bf16(weights["final_norm"])
            layers = [S.engine_layer_arrays(lw, args.weight_format) for lw in weights["layers"]]
        else:
            embed, fnorm = GPUArray((V, H), bfloat16), GPUArray((H,), bfloat16)
            layers = [{k: GPUArray(s, dt) for k, (s, dt) in shapes.items()} for _ in range(cfg["num_layers"])]
        _hip.call("pgk_device_sync")
        cp.barrier()
        t0 = time.perf_counter()
        nbytes = 0
        for arr in [embed, fnorm] + [lw[k] for lw in layers for k in shapes]:
            comm.broadcast(arr, 0)
            nbytes += arr.nbytes
        _hip.call("pgk_device_sync")
        bcast_s = time.perf_counter() - t0
        eng = Engine(cfg, embed, layers, fnorm, None, max_seq_len=max_seq, max_batch=B, weight_format=args.weight_format)

    # ---- prompts: one independent sequence per (rank, slot) ----
    rng = np.random.default_rng(1000 + args.seed)
    all_prompts = rng.integers(0, cfg["vocab_size"], (cp.world * B, P))
    mine = all_prompts[cp.rank * B:(cp.rank + 1) * B]

    # ---- prefill (MFMA path), timed on sequence slot 0 ----
    start_ev, stop_ev = _hip.C.c_void_p(), _hip.C.c_void_p()
    _hip.call("pgk_event_create", _hip.C.byref(start_ev))
    _hip.call("pgk_event_create", _hip.C.byref(stop_ev))
    first = np.zeros(B, np.int32)
    for b in range(B):
        first[b] = int(np.argmax(eng.prefill([int(t) for t in mine[b]], seq=b)))
    pf_ms = []
    for _ in range(5):
        _hip.call("pgk_event_record", start_ev, None)
        eng.prefill([int(t) for t in mine[0]], seq=0, want_last_logits=False)
        _hip.call("pgk_event_record", stop_ev, None)
        _hip.call("pgk_event_sync", stop_ev)
        ms = _hip.C.c_float()
        _hip.call("pgk_event_elapsed_ms", start_ev, stop_ev, _hip.C.byref(ms))
        pf_ms.append(ms.value)
    pf_med = float(np.median(pf_ms))
    pf_flops = prefill_flops(cfg, P, all_rows=False)

    # ---- decode: whole-step hipGraph, state in device memory ----
    eng.set_state(first, [P] * B)
    eng.capture(B)

    def run(n):
        for _ in range(n):
            eng.replay(1)   # one whole-step graph launch; no collective and no host sync inside a step

    run(W)
    _hip.call("pgk_device_sync")
    cp.barrier()
    _hip.call("pgk_device_sync")
    t0 = time.perf_counter()
    _hip.call("pgk_event_record", start_ev, None)
    run(K)
    _hip.call("pgk_event_record", stop_ev, None)
    _hip.call("pgk_device_sync")
    cp.barrier()
    wall = time.perf_counter() - t0
    ms = _hip.C.c_float()
    _hip.call("pgk_event_elapsed_ms", start_ev, stop_ev, _hip.C.byref(ms))
    wall_max = cp.max_over_ranks(wall)
    dev_ms_max = cp.max_over_ranks(ms.value)
    tokens = eng.read_tokens(B, min(W + K, 4096))
    # the harness's view of the whole batch: every rank's token log gathered once, after the timed steps
    # (RCCL all-gather of the device-resident log; gloo when RCCL is not up)
    t0 = time.perf_counter()
    if comm is not None:
        from pygpukit_amd.core.array import GPUArray
        from pygpukit_amd.core.dtypes import int32
        from pygpukit_amd.core.factory import from_numpy

        mine_log = from_numpy(np.ascontiguousarray(tokens, dtype=np.int32))
        all_log = GPUArray((cp.world,) + tuple(tokens.shape), int32)
        comm.all_gather(mine_log, all_log)
        _hip.call("pgk_device_sync")
        all_tokens = all_log.to_numpy()
    elif cp.world > 1:
        all_tokens = np.stack(cp.all_gather_array(np.ascontiguousarray(tokens, dtype=np.int32)))
    else:
        all_tokens = tokens[None]
    gather_s = time.perf_counter() - t0

    # ---- long prefill: the regime where the projections are MFMA-bound (S = 128 above is weight/latency-bound) ----
    long_pf = None
    if args.long_prefill > 0 and cp.rank == 0:
        from pygpukit_amd.llm.engine import Engine

        SL = args.long_prefill
        eng_l = Engine(cfg, eng._keep[0], eng._keep[3], eng._keep[1], None, max_seq_len=SL + 8, max_batch=1,
                       weight_format=args.weight_format)   # same device weights, its own KV cache
        lp = [int(t) for t in np.random.default_rng(2000 + args.seed).integers(0, cfg["vocab_size"], SL)]
        eng_l.prefill(lp, want_last_logits=False)
        l_ms = []
        for _ in range(3):
            _hip.call("pgk_event_record", start_ev, None)
            eng_l.prefill(lp, want_last_logits=False)
            _hip.call("pgk_event_record", stop_ev, None)
            _hip.call("pgk_event_sync", stop_ev)
            ms = _hip.C.c_float()
            _hip.call("pgk_event_elapsed_ms", start_ev, stop_ev, _hip.C.byref(ms))
            l_ms.append(ms.value)
        lf = prefill_flops(cfg, SL, all_rows=False)
        lmed = float(np.median(l_ms))
        long_pf = {"tokens": SL, "ms": lmed, "tflops": lf / (lmed * 1e-3) / 1e12, "flops": lf, "logits": "last row only",
                   "frac_mfma_peak": lf / (lmed * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "tokens_per_s": SL / (lmed * 1e-3),
                   "runs_ms": [round(x, 3) for x in l_ms]}
        del eng_l

    # ---- extra legs: the other single-GPU BASELINE configs at the same weights (reported, not the headline) ----
    extras = None
    if not args.no_extras and cp.world == 1 and B == 1 and args.weight_format == "bf16" and P == 128:
        def decode_leg(fmt, batch, prompt_len, steps=32, warm=4):
            e2 = S.build_engine_from_weights(cfg, weights, max_seq_len=prompt_len + steps + warm + 8, max_batch=batch, weight_format=fmt)
            pr = np.random.default_rng(3000 + args.seed).integers(0, cfg["vocab_size"], (batch, prompt_len))
            f0 = [int(np.argmax(e2.prefill([int(t) for t in pr[b]], seq=b))) for b in range(batch)]
            e2.set_state(f0, [prompt_len] * batch)
            e2.capture(batch)
            e2.replay(warm)
            _hip.call("pgk_device_sync")
            t0 = time.perf_counter()
            e2.replay(steps)
            _hip.call("pgk_device_sync")
            dt = time.perf_counter() - t0
            ab2 = algorithmic_bytes_per_token(cfg, prompt_len + warm + steps // 2, fmt)
            by = ab2["weights"] + ab2["lm_head"] + batch * (ab2["kv_read"] + ab2["kv_write"] + ab2["logits"])
            return {"tokens_per_s": batch * steps / dt, "ms_per_step": dt * 1e3 / steps, "batch": batch, "context": prompt_len,
                    "weights": fmt, "hbm_frac": by / (dt / steps) / 1e9 / HBM_PEAK_GBS}
        extras = {"config4_per_gpu_batch8": decode_leg("bf16", 8, 128),
                  "config3_ctx2048_w8a16": decode_leg("fp8", 1, 2048),
                  "ctx2048_bf16": decode_leg("bf16", 1, 2048)}

        def config5_leg(S_tok=4096, reps=3):
            """BASELINE config 5: Llama-3-8B-shape random-init weights, fp8 e4m3 x fp8 MFMA prefill of 4096 tokens (and the
            bf16 prefill of the same model beside it).  FLOPs: projection GEMMs + causal attention + last-row lm_head."""
            from pygpukit_amd.llm.engine import Engine

            c5 = dict(S.LLAMA3_8B)
            w5 = S.random_engine_weights(c5, seed=args.seed, fp8=True, keep_bf16=True, threads=12)
            H5, D5, I5, V5, L5 = c5["hidden_size"], c5["head_dim"], c5["intermediate_size"], c5["vocab_size"], c5["num_layers"]
            per_layer = H5 * (c5["num_heads"] + 2 * c5["num_kv_heads"]) * D5 + c5["num_heads"] * D5 * H5 + 3 * H5 * I5
            flops = 2.0 * S_tok * L5 * per_layer + 2.0 * V5 * H5 + 4.0 * S_tok * S_tok * D5 * c5["num_heads"] * L5 / 2
            toks = [int(t) for t in np.random.default_rng(4000 + args.seed).integers(0, V5, S_tok)]
            out = {"tokens": S_tok, "layers": L5, "flops": flops}
            for fmt, peak in (("fp8a8", 2 * MFMA_BF16_PEAK_TFLOPS), ("bf16", MFMA_BF16_PEAK_TFLOPS)):
                e5 = Engine(c5, w5["embed"], w5["bf16"] if fmt == "bf16" else w5["fp8"], w5["final_norm"], None, max_seq_len=S_tok + 8,
                            max_batch=1, weight_format=fmt, use_qk_norm=False)
                e5.prefill(toks, want_last_logits=False)
                e5.synchronize()
                ms = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    e5.prefill(toks, want_last_logits=False)
                    e5.synchronize()
                    ms.append((time.perf_counter() - t0) * 1e3)
                best = min(ms)
                out[fmt] = {"ms": best, "tflops": flops / best / 1e9, "frac_mfma_peak": flops / best / 1e9 / peak,
                            "peak_tflops": peak, "runs_ms": [round(x, 2) for x in ms]}
                del e5
            return out

        if not args.no_config5:
            try:
                extras["config5_llama8b_prefill4096"] = config5_leg()
            except Exception as e:  # noqa: BLE001 - an extra leg must never take the headline down
                extras["config5_llama8b_prefill4096"] = {"error": f"{type(e).__name__}: {e}"}

    # ---- per-kernel timing (eager, event after every kernel) for the roofline objects ----
    prof = eng.profile_step(B, 8)
    ctx_mid = P + W + K // 2
    ab = algorithmic_bytes_per_token(cfg, ctx_mid, args.weight_format)
    lm_us = prof["lmhead"][0]
    lm_bytes = ab["lm_head"] + cfg["hidden_size"] * 4 * B + cfg["vocab_size"] * 4 * B
    traffic = None
    try:  # HBM bytes per launch from the committed rocprofv3 --pmc pass (FETCH_SIZE x2 per the microarch guide + WRITE_SIZE)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_decode_pmc_hbm.json")))
        for k, v in pmc.items():
            if "1, 4, 0, 3, 2" in k:  # <bf16, float, M=1, R=4, PRO_NORM, EPI_LOGITS, C=2>
                traffic = v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
    except Exception:
        pass
    roofline = {"kernel": "fused_gemv_kernel<bf16, PRO_NORM, EPI_LOGITS> (lm_head + argmax partials)", "bound": "hbm",
                "achieved": lm_bytes / lm_us / 1e3 if lm_us else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (lm_bytes / lm_us / 1e3 / HBM_PEAK_GBS) if lm_us else None,
                "traffic": traffic if (B == 1 and args.weight_format == "bf16") else None,
                "bytes_per_launch": lm_bytes, "us_per_launch": lm_us, "timing": "hipEvent pair around each eager launch minus the measured empty event-pair cost, 8 steps"}
    step_ms = dev_ms_max / K
    step_bytes = ab["weights"] + ab["lm_head"] + B * (ab["kv_read"] + ab["kv_write"] + ab["logits"])
    step_roofline = {"bound": "hbm", "achieved": step_bytes / (step_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bytes_per_step": step_bytes,
                     "launches_per_step": eng.launches_per_step(), "ctx_mid": ctx_mid,
                     "kernel_us": {k: round(v[0], 2) for k, v in prof.items()},
                     "kernel_launches_per_step": {k: v[1] for k, v in prof.items()}}

    result = {
        "metric": "decode tokens/sec/GPU + prefill TFLOPS (% MFMA peak), Qwen3-0.6B bf16",
        "value": cp.world * B * K / wall_max, "unit": "tokens/s", "n_gpus": cp.world, "steps": K, "warmup": W,
        "ms_per_step": wall_max * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.weight_format == "bf16" else "fp8-e4m3 weights x bf16", "data": "synthetic",
        "config": {"workload": f"Qwen3-0.6B-shape random-init {args.weight_format}, prefill {P} + {K}-token greedy decode, "
                               f"whole-step hipGraph (BASELINE config 2{'' if B == 1 and cp.world == 1 else ' shape, DP replicas'})",
                   "batch_per_gpu": B, "global_batch": cp.world * B, "prompt_len": P, "parallelism": f"dp{cp.world}",
                   "layers": cfg["num_layers"]},
        "tokens_per_s_per_gpu": B * K / wall_max, "device_ms_per_step": step_ms,
        "roofline": roofline, "step_roofline": step_roofline,
        "prefill": {"ms": pf_med, "tflops": pf_flops / (pf_med * 1e-3) / 1e12, "flops": pf_flops, "logits": "last row only",
                    "frac_mfma_peak": pf_flops / (pf_med * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "runs_ms": [round(x, 3) for x in pf_ms]},
        "prefill_long": long_pf, "extras": extras,
        "first_tokens": [int(t) for t in tokens[: min(8, len(tokens)), 0]],
        "setup_s": time.perf_counter() - t_setup,
    }
    if cp.world > 1:
        result["token_gather"] = {"seconds": gather_s, "bytes_per_rank": int(tokens.nbytes), "via": "rccl all_gather" if comm is not None else "gloo",
                                  "ranks_agree_on_shape": bool(all_tokens.shape[0] == cp.world)}
        if rccl_note:
            result["rccl"] = f"not used ({rccl_note}); weights drawn per rank from the same seed"
    if bcast_s is not None:
        result["weight_broadcast"] = {"seconds": bcast_s, "GB": nbytes / 1e9, "GBps": nbytes / 1e9 / bcast_s}
    if args.layers:
        result["INVALID"] = "layer count overridden"
    if cp.rank == 0 and cp.world == 1 and not args.no_cpu_baseline:
        del eng
        result["cpu_baseline"] = cpu_baseline(cfg, weights, [int(t) for t in mine[0]], args.cpu_decode_tokens)
    elif cp.rank == 0:
        result["cpu_baseline"] = None
    if cp.rank == 0:
        print(json.dumps(result))
    cp.barrier()
    if comm is not None:
        comm.destroy()
    cp.shutdown()


if __name__ == "__main__":
    main()

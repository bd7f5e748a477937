#!/usr/bin/env python
"""captions/sec of the CapDec caption hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no RANK in the environment the script re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N` (one rank per GPU, one RCCL
communicator); started by torch.distributed.run directly it just joins the group.

A "step" is one pass of the hot path over one batch of synthetic CLIP embeddings resident in
HBM: normalise -> mapping network -> GPT-2 KV-cached decode -> token ids (+ RCCL all-gather of
the ids when N > 1).  Default workload = BASELINE.json's metric configuration (configs[2]):
COCO-val-5k-shaped = 5000 x 512-d embeddings IN TOTAL, sharded over the N ranks (strong scaling,
625 captions per GPU at N = 8), TransformerMapper (8 layers), prefix_len 10, beam 5,
entry_length 67, fp32 (the reference's GPT-2 dtype; greedy ids bit-identical to it).
`--scaling weak` keeps 5000 captions PER GPU instead; at N > 1 the strong-scaling line also
carries a one-step weak measurement and the 1-GPU rate of the same box (`scaling_check`).
Weights are the seeded hot-init recipe of capdec_amd/synth.py (no checkpoints offline), which
never emits the stop token, so every caption runs all 67 steps -- fixed, reproducible work.

Rank 0 prints ONE JSON line (metric/value/... + "roofline" + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from capdec_amd import synth  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (not the 2:1-sparse marketing figure)
# what the 1400 W package cap leaves of the fp16 matrix pipe on random (full-entropy) operands: a registers-only stream of
# back-to-back MFMAs sustains 1650 TFLOP/s (profiles/r3_mfma_power_ceiling.txt, tools/probes/mfma_energy.hip; zeros: 2451)
F16_MFMA_AT_POWER_CAP_TFLOPS = 1650.0
STOP_ID, D_EMB = 13, 768
PMC_TRAFFIC_FILE = "r6_pmc_traffic.json"   # rocprofv3 --pmc summary the `roofline.traffic` field is read from


def algorithmic_flops_per_caption(P, T, beam, mapper, dims=synth.GPT2_SMALL, D=512, clip_len=10):
    """SURVEY.md section 8 D.4 (KV-cached minimal algorithm)."""
    d, V, nl = dims.n_embd, dims.vocab, dims.n_layer
    f_body = 2.0 * nl * (d * 3 * d + d * d + d * 4 * d + 4 * d * d)
    f_head = 2.0 * V * d
    if beam == 1:
        body_tok, head_rows = P + T - 1, T
        att = sum(4.0 * d * nl * L for L in range(1, P + T))
    else:
        body_tok, head_rows = P + beam * (T - 1), 1 + beam * (T - 1)
        att = sum(4.0 * d * nl * L for L in range(1, P + 1)) + beam * sum(4.0 * d * nl * (P + i) for i in range(1, T))
    f = body_tok * f_body + head_rows * f_head + att
    if mapper == "mlp":
        f += 2.0 * (D * (d * P // 2) + (d * P // 2) * d * P)
    elif mapper == "transformer":
        S = clip_len + P
        f += 2.0 * D * clip_len * d + 8 * S * 2.0 * (3 * d * d + d * d + 2 * d * 2 * d) + 8 * 4.0 * S * S * d
    return f


def oracle_compare(model, emb, out, mapper, beam, P, T, rows, stop_id=None, sd=None):
    """captions `rows` of a timed step's result against the CPU oracle run on their own prefixes (the oracle is the checker,
    never the thing measured).  Every caption is compared; a difference is tolerated only on a numerical tie -- selected /
    rejected candidate keys (beam) or top-1 / top-2 logits (greedy) within 1e-4 of each other at some step: which side of
    fp32 round-off wins there is not defined -- and counted separately"""
    import numpy as np
    import torch
    from capdec_amd import synth
    from capdec_amd.predictions_runner import prefix_from_embeddings
    from oracle import capdec_oracle as O
    sd = synth.hot_state_dict(42, mapper, 512, P) if sd is None else sd
    stop_id = STOP_ID if stop_id is None else int(stop_id)
    rows = sorted(set(int(r) for r in rows))
    pe = prefix_from_embeddings(model, emb[rows]).float().cpu()
    ids, lens = out[0][rows].cpu().numpy(), out[1][rows].cpu().numpy()
    t0 = time.perf_counter()
    equal = ties = 0
    if beam:
        mg = []
        tok, seq, sc = O.beam_cached(sd, pe, 5, stop_id, T, margins=mg)
        order = O.beam_output_order(sc)
        clear = (mg[0] > 1e-4).numpy()
        scores = out[2][rows].cpu().numpy()
        for j in range(len(rows)):
            b = int(order[j][0])
            good = (np.array_equal(ids[j].reshape(-1)[:T], tok[j][b].numpy()) and int(np.asarray(lens[j]).reshape(-1)[0]) == int(seq[j][b])
                    and abs(float(np.asarray(scores[j]).reshape(-1)[0]) - float(sc[j][b])) <= 1e-4)
            equal += int(good)
            ties += int((not good) and (not clear[j]))
    else:
        gi, gl = O.greedy_cached(sd, pe, stop_id=stop_id, entry_length=T)
        _, st = O.greedy_forced(sd, pe, gi)
        live = torch.arange(T)[None, :] < gl[:, None]
        gap = torch.where(live, st[:, :, 0] - st[:, :, 1], torch.full((1, 1), 1e9))
        clear = (gap.min(dim=1).values > 1e-4).numpy()
        for j in range(len(rows)):
            good = np.array_equal(ids[j], gi[j].numpy()) and int(lens[j]) == int(gl[j])
            equal += int(good)
            ties += int((not good) and (not clear[j]))
    n = len(rows)
    return {"oracle_checked": n, "oracle_equal": equal, "differ_on_a_numerical_tie": ties, "ok": equal == n - ties,
            "captions_with_a_near_tie_step": int((~clear).sum()),
            "rows": rows, "seconds": round(time.perf_counter() - t0, 1),
            "note": "rows of the timed default-mode step vs oracle/capdec_oracle.py (beam: best beam's ids, length, mean log-prob "
                    "within 1e-4; greedy: ids and length) on the prefixes the HIP mapper produced"}


def cpu_baseline(mapper, beam, P, T, budget_s=20.0, D=512, captions=8):
    """The oracle's reference-shaped path (batch 1, NO KV cache, lm_head on every position, fp32
    torch CPU ops: the algorithm of reference gpt2_prefix_eval.py:50-198 driven like
    predictions_runner.py:221-232) timed on this box's host cores on a BOUNDED sample of the same
    workload: ``captions`` WHOLE captions (default 8; SURVEY D.5's 32 with --cpu-captions 32), one
    warm-up caption discarded, wall clock.  Two thread settings are measured and both reported:
    all host cores (torch.set_num_threads(os.cpu_count()), what D.5 prescribes) on the first caption,
    and the setting a short probe finds fastest on this box (batch-1 GEMVs stop scaling long before 256
    threads) on the rest; `value` / `cores` are the better of the two.  ``captions`` = 0 falls back to a
    time budget of ``budget_s`` seconds of decode steps (fractional captions, by token-rows).
    profiles/r3_cpu_port_vs_reference.json: this port against the imported reference itself in the build
    container (same ids, 0.93-1.02x its time)."""
    from oracle import capdec_oracle as O
    ncpu = os.cpu_count() or 1
    sd = synth.hot_state_dict(42, mapper, D, P)
    x = synth.synthetic_clip_embeddings(4, D, seed=0)
    rows_per_caption = (sum(P + i for i in range(T)) if beam == 1
                        else P + beam * sum(P + i for i in range(1, T)))

    def run(e, budget, T_):
        st = {"rows": 0, "t0": time.perf_counter()}

        def on_step(i, rows, L):
            st["rows"] += rows * L
            return (time.perf_counter() - st["t0"]) > budget
        if beam > 1:
            O.generate_beam_ref(sd, e, beam, STOP_ID, T_, on_step=on_step)
        else:
            O.generate2_ref(sd, e, STOP_ID, T_, on_step=on_step)
        return st["rows"], time.perf_counter() - st["t0"]

    def embed(r):
        return O.clip_project(O.normalize_prefix(synth.synthetic_clip_embeddings(r + 1, D, seed=0)[r:r + 1]), sd, mapper, P)

    with torch.no_grad():
        pe = O.clip_project(O.normalize_prefix(x), sd, mapper, P)
        # pick the thread count that serves this box best (short probes), then the timed sample
        best = None
        for th in sorted({ncpu, min(ncpu, 64), min(ncpu, 32), min(ncpu, 16)}, reverse=True):
            torch.set_num_threads(th)
            run(pe[:1], 0.5, 3)                       # warm-up (lazy init), discarded
            r, dt = run(pe[1:2], 1.5, T)
            if best is None or r / dt > best[1]:
                best = (th, r / dt)
        torch.set_num_threads(best[0])
        run(pe[:1], 0.5, 3)
        rows, dt, r = 0, 0.0, 3
        while (dt < budget_s) if captions <= 0 else (r - 3 < max(captions - 1, 1)):
            rr, dd = run(embed(r), (budget_s - dt) if captions <= 0 else 1e9, T)
            rows, dt, r = rows + rr, dt + dd, r + 1
    frac = rows / rows_per_caption
    value, cores = frac / dt, best[0]
    whole = "%d whole captions" % (r - 3) if captions > 0 else "%.3f captions (time budget, counted by token-rows)" % frac
    return {"value": value, "unit": "captions/s", "cores": cores, "kind": "port", "host_cpus": ncpu,
            "probed_threads": {"cores": best[0], "value": frac / dt, "seconds": round(dt, 2)},
            "sample": f"{whole} of the same workload (reference-shaped: batch 1, no KV cache, fp32, T={T}, "
                      f"beam={beam}; {rows} token-rows) in {dt:.1f} s wall on {best[0]} threads (count chosen by a "
                      f"1.5 s probe among {{all, 64, 32, 16}} of the {ncpu} host threads: the batch-1 GEMVs of this path do not scale past a few dozen threads, and a run on all of a 256-thread host does not finish one caption inside a bounded sample -- not measurable, so not reported); warm-up captions discarded"}


def stop_profile(model, emb, mapper, beam, P, T, steps, note, target=11.0, oracle_rows=32):
    """The workload in which captions STOP (reference gpt2_prefix_eval.py:107-109,187-188: a caption ends at its stop
    token; real COCO captions are ~11 tokens).  The hot-init weights never emit id 13, so the metric line runs all T steps.
      * no single id of these weights ends captions early: for every token v of a 1024-caption sample's best beams, the mean
        of (first position of v) + 1 predicts the mean length with stop = v; the three best predictions are decoded for real
        and listed (`candidate_ids`: mean lengths of ~43-55 tokens) -- every vocabulary row is an i.i.d. Gaussian, so no
        token is frequent;
      * so the profile uses synth.with_stop_bias: the SAME weights with a constant added to the logit of the stop token
        (id 13, '.'); the constant is found by bracketing + bisection on the sample so that the MEASURED mean best-beam
        length is `target` +- 0.5 tokens (seeded weights, seeded embeddings: deterministic);
      * timed: the whole batch with those weights, finished-caption compaction on and off (capdec_set_compact), `steps`
        passes each, inputs resident, same timing rule as the metric;
      * per decode step: the activation rows the loop launched (capdec_decode_step_rows) next to the rows still alive;
      * the 8-GPU shard imbalance SURVEY section 8 E warns about, on the one-GPU proxy: the eight contiguous 625-caption
        shards are decoded one after the other on this GPU and timed -- max / mean of their times, of their step counts
        and of their row-steps; the slowest shard sets the whole-node rate;
      * `oracle_rows` captions of the timed compaction-on run against the CPU oracle run on the same weights.
    The model gets its original weights back before the function returns."""
    import numpy as np
    from capdec_amd.gpt2_prefix_eval import decode_beam_ids, decode_greedy_ids
    from capdec_amd.predictions_runner import prefix_from_embeddings
    eng = model.engine
    n = emb.shape[0]
    B = 5 if beam else 1
    sd0 = synth.hot_state_dict(42, mapper, 512, P)

    def decode(pe, stop):
        if beam:
            ids, lens, sc, _ = decode_beam_ids(model, pe, stop, 5, T)
            return ids, lens, sc
        ids, lens = decode_greedy_ids(model, pe, stop, T)
        return ids[:, None], lens[:, None], None

    note("stop profile: candidate stop ids of the unmodified weights")
    m = min(n, 1024)
    pe_s = prefix_from_embeddings(model, emb[:m])
    ids0 = decode(pe_s, STOP_ID)[0][:, 0].cpu().numpy()                    # best beam, all T tokens
    first = {}
    for r in range(m):
        seen = set()
        for t, v in enumerate(ids0[r].tolist()):
            if v not in seen:
                seen.add(v)
                first.setdefault(v, []).append(t + 1)
    pred = {v: (sum(p) + T * (m - len(p))) / m for v, p in first.items()}
    tried = []
    for v in sorted(pred, key=lambda v: (abs(pred[v] - target), v))[:3]:
        lens = decode(pe_s, v)[1][:, 0].float()
        tried.append({"stop_id": int(v), "predicted_mean_len": round(pred[v], 2), "measured_mean_len": round(float(lens.mean()), 2)})

    def mean_len_at(alpha):
        model.load_state_dict(synth.with_stop_bias(sd0, STOP_ID, alpha))
        return float(decode(prefix_from_embeddings(model, emb[:m]), STOP_ID)[1][:, 0].float().mean())

    try:
        note("stop profile: logit offset of the stop token for a mean length of %g" % target)
        search, lo, hi = [], 0.0, 1.0
        got = mean_len_at(hi)
        search.append([hi, round(got, 2)])
        while got > target and hi < 4096.0:          # bracket: the mean length falls as the offset grows
            lo, hi = hi, hi * 2.0
            got = mean_len_at(hi)
            search.append([hi, round(got, 2)])
        alpha = hi
        for _ in range(12):
            if abs(got - target) <= 0.5:
                break
            alpha = 0.5 * (lo + hi)
            got = mean_len_at(alpha)
            search.append([round(alpha, 4), round(got, 2)])
            if got > target:
                lo = alpha
            else:
                hi = alpha
        sd1 = synth.with_stop_bias(sd0, STOP_ID, alpha)
        model.load_state_dict(sd1)

        def timed(compact, e):
            eng.set_compact(compact)
            pe = prefix_from_embeddings(model, e)
            decode(pe, STOP_ID)                                           # warm-up (buffers of this size)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                pe = prefix_from_embeddings(model, e)
                out = decode(pe, STOP_ID)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            return out, dt, eng.decode_stats(), eng.decode_step_rows()

        note("stop profile: timed passes (stop logit + %.4g)" % alpha)
        out_on, dt_on, st_on, rows_on = timed(True, emb)
        out_off, dt_off, st_off, rows_off = timed(False, emb)
        eng.set_compact(True)
        # (a compacted step launches fewer rows: other GEMM tile geometries / split-K -- the fp32 round-off of a row may
        #  differ in the last bit, and on a numerical near-tie a beam; capdec_set_batch_invariant removes that, capdec.h)
        same = float(((out_on[0] == out_off[0]).flatten(1).all(1) & (out_on[1] == out_off[1]).all(1)).float().mean())
        lens_best = out_on[1][:, 0].cpu().numpy()
        done_at = out_on[1].max(dim=1).values.cpu().numpy()               # step after which every beam of the caption has stopped
        alive = [int((done_at > i).sum()) * B for i in range(1, st_on["steps"])]
        # ---- shard imbalance (one-GPU proxy of the 8-rank run): each shard alone, back to back
        note("stop profile: the eight 625-caption shards, one after the other")
        from capdec_amd import distributed as cdist
        shards = []
        for r in range(8):
            lo_r, hi_r = cdist.shard_bounds(n, r, 8)
            if hi_r <= lo_r:
                continue
            _, dt_r, st_r, _ = timed(True, emb[lo_r:hi_r])
            shards.append({"rank": r, "captions": hi_r - lo_r, "ms": round(dt_r * 1e3, 2), "steps": st_r["steps"], "row_steps": st_r["row_steps"]})
        eng.set_compact(True)
        mx = lambda k: max(s_[k] for s_ in shards)
        mean = lambda k: sum(s_[k] for s_ in shards) / len(shards)
        rec = {"stop_id": STOP_ID, "stop_logit_offset": round(alpha, 4), "offset_search": search, "target_mean_len": target,
               "candidate_ids_of_the_unmodified_weights": tried,
               "mean_len_best_beam": round(float(lens_best.mean()), 2),
               "len_percentiles_10_50_90_max": [int(np.percentile(lens_best, q)) for q in (10, 50, 90)] + [int(lens_best.max())],
               "captions": n, "entry_length": T, "timed_passes": steps,
               "compaction_on": {"value": round(n / dt_on, 1), "unit": "captions/s", "ms_per_pass": round(dt_on * 1e3, 2),
                                 "steps_run": st_on["steps"], "compactions": st_on["compactions"], "row_steps": st_on["row_steps"]},
               "compaction_off": {"value": round(n / dt_off, 1), "unit": "captions/s", "ms_per_pass": round(dt_off * 1e3, 2),
                                  "steps_run": st_off["steps"], "row_steps": st_off["row_steps"]},
               "captions_identical_on_vs_off": round(same, 5),
               "rows_launched_per_step": rows_on, "rows_alive_per_step": alive,
               "row_steps_if_every_caption_left_at_its_own_stop": int(sum(alive)),
               "shards_of_8": {"per_rank": shards,
                               "ms_max_over_mean": round(mx("ms") / mean("ms"), 3),
                               "steps_max_over_mean": round(mx("steps") / mean("steps"), 3),
                               "row_steps_max_over_mean": round(mx("row_steps") / mean("row_steps"), 3),
                               "whole_node_captions_per_s_if_8_gpus": round(n / (mx("ms") * 1e-3), 1),
                               "note": "one-GPU proxy: rank r's contiguous shard decoded alone on this GPU; an 8-GPU pass ends "
                                       "when its slowest rank does (max), perfect balance would be the mean"},
               "note": "untimed extra of the metric line: same embeddings, the hot-init weights with a constant added to the stop "
                       "token's logit (synth.with_stop_bias) so that captions end (mean best-beam length ~%g tokens); `value` of the "
                       "metric line is the all-steps workload" % target}
        if oracle_rows > 0:
            try:
                note("stop profile: oracle check of %d captions" % oracle_rows)
                rows = sorted(set(int(round(i * (n - 1) / max(1, oracle_rows - 1))) for i in range(oracle_rows)))
                o = (out_on[0][:, 0].contiguous(), out_on[1][:, 0].contiguous(), out_on[2][:, 0].contiguous() if beam else None)
                rec["oracle_check"] = oracle_compare(model, emb, o, mapper, beam, P, T, rows, sd=sd1)
            except Exception as ex:
                rec["oracle_check"] = {"ok": False, "error": str(ex)[:300]}
    finally:
        eng.set_compact(True)
        model.load_state_dict(sd0)
    return rec


class SmiSampler:
    """rocm-smi (sclk, package power) sampled from a host thread while the timed region runs: the GEMMs of this path are
    bound by the package power cap (profiles/r3_power_probe.txt), so the clock the chip held belongs next to the rate."""

    def __init__(self, card=0, period=1.0):
        import threading
        self.card, self.period, self.samples, self._stop = card, period, [], threading.Event()
        self._th = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        import subprocess
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True,
                                 timeout=10).stdout
            c = json.loads(out).get("card%d" % self.card, {})
            sclk = power = None
            for k, v in c.items():
                if "sclk clock speed" in k.lower():
                    sclk = float(str(v).strip("()").lower().replace("mhz", ""))
                elif "power" in k.lower() and "(w)" in k.lower():
                    power = float(v)
            if sclk and power:
                self.samples.append((sclk, power))
        except Exception:       # rocm-smi missing or busy: the field stays null
            pass

    def _run(self):
        while not self._stop.is_set():
            self._read()
            self._stop.wait(self.period)

    def start(self):
        self._th.start()

    def stop(self):
        self._stop.set()
        self._th.join(timeout=15)
        if not self.samples:
            return None
        n = len(self.samples)
        return {"sclk_mhz": round(sum(s[0] for s in self.samples) / n, 1), "watts": round(sum(s[1] for s in self.samples) / n, 1),
                "samples": n, "sclk_max_mhz": 2400,
                "note": "rocm-smi during the timed region (whole step: GEMM, attention, LayerNorm phases mixed)"}


def side_workload(args, world, rank, dev, emit=print):
    """configs[3] / configs[4] of BASELINE.json: parity-test cases, measured here for DESIGN.md (not the metric line)."""
    from capdec_amd import clip as cclip, distributed as cdist, embeddings_generator as eg
    from capdec_amd.gpt2_prefix import ClipCaptionModel, MappingType
    from capdec_amd.predictions_runner import caption_ids
    P, T = args.prefix_length, args.entry_length
    prec = {"f16": "fp16", "bf16": "bf16"}.get(args.gemm_mode or "", "fp32")      # towers: fp32-accurate unless asked
    text = args.workload == "text_embed"
    rn = args.clip == "rn50x4" and not text       # the reference's default backbone (predictions_runner.py:158): 288^2 -> 640-d
    clip_sd = synth.hot_clip_resnet_state_dict(44, synth.CLIP_RN50X4) if rn else synth.hot_clip_state_dict(43)
    cm, _ = cclip.load(clip_sd, device=dev.index or 0, precision=prec)
    D = 640 if rn else 512
    mt = MappingType.MLP if text else MappingType.TransformerEncoder
    model = ClipCaptionModel(P, clip_length=10, prefix_dim=D, num_layers=8, mapping_type=mt).to(dev).eval()
    model.load_state_dict(synth.hot_state_dict(42, "mlp" if text else "transformer_encoder", D, P))
    if args.gemm_mode:
        model.engine.set_gemm_mode(args.gemm_mode)
    n_global = args.captions * world
    if text:
        inp = synth.synthetic_clip_tokens(n_global, seed=2).to(dev)
    else:
        inp = synth.synthetic_images(n_global, seed=4, size=288 if rn else 224).to(dev)

    def step():
        if text:     # embeddings_generator.py:58-101 + train.py:347,253-254
            return eg.text_to_prefix(cm, model, inp, noise_variance=0.016, seed=3, rank=rank, world=world)
        emb = eg.encode_images(cm, inp, rank, world, gather=False)      # predictions_runner.py:220-232
        full = torch.zeros(n_global, D, device=dev)
        lo, hi = cdist.shard_bounds(n_global, rank, world)
        full[lo:hi] = emb
        ids, lens, sc = caption_ids(model, full, STOP_ID, beam=True, entry_length=T, rank=rank, world=world)
        return cdist.gather_ids(ids, lens, n_global, sc)

    for _ in range(args.warmup):
        step()
    cm._engine.profile_enable(max(1, args.profile_every))
    cm._engine.profile_reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = cm._engine.profile_get()
    if rank == 0:
        # ---- roofline of the tower's dominant kernel family (hipEvent-timed launches on the launch stream)
        fams = {k: v for k, v in prof.items() if v["launches"]}
        est = lambda v: v["ms"] * v["calls"] / v["launches"]
        roof = None
        if fams:
            k, v = max(fams.items(), key=lambda kv: est(kv[1]))
            tower_ms = sum(est(x) for x in fams.values())
            if v["flops"] > 0:
                peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if "f16x2" in k else (PEAK_BF16_MFMA_TFLOPS / 6.0 if "x3" in k else
                                                                       (PEAK_F32_MFMA_TFLOPS if k == "gemm_f32" else PEAK_BF16_MFMA_TFLOPS))
                ach = v["flops"] / v["ms"] * 1e-9
                traffic = None
                try:        # fabric-side bytes per launch of that family from the committed --pmc passes of the same command
                    pm = json.load(open(os.path.join(ROOT, "profiles", "r6_text_f16_pmc_traffic.json")))
                    kk = {"gemm_x1": "gemm_x1_kernel", "gemm_f16x2p": "gemm_f16x2p_kernel"}.get(k)
                    if text and pm.get("gemm_mode") == (args.gemm_mode or "f16x2") and pm.get("captions_per_gpu") == n_global and kk in pm:
                        traffic = pm[kk]["traffic_bytes_per_launch"]
                except (OSError, ValueError, KeyError):
                    pass
                roof = {"bound": "mfma", "kernel": k, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                        "frac": round(ach / peak, 4), "traffic": traffic, "avg_launch_ms": round(v["ms"] / v["launches"], 4),
                        "launches_timed": v["launches"], "share_of_tower": round(est(v) / tower_ms, 3),
                        "note": "algorithmic 2 M N K of every GEMM of the family / hipEvent time of the timed launches; peak = dense "
                                "16-bit MFMA 2.5 PFLOP/s (one plane per operand) or / 3 (two fp16 planes, fp32-accurate)"}
            else:       # attention / LayerNorm dominant: HBM-side bytes are not counted by the launchers -- time share only
                roof = {"bound": "hbm", "kernel": k, "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None,
                        "avg_launch_ms": round(v["ms"] / v["launches"], 4), "share_of_tower": round(est(v) / tower_ms, 3)}
        # ---- CPU baseline: the oracle's restatement of the same chain on a bounded sample, host cores of this box
        cpu = None
        if args.cpu_seconds > 0:
            try:
                from oracle import capdec_oracle as O
                th = min(32, os.cpu_count() or 1)
                torch.set_num_threads(th)
                sdm = synth.hot_state_dict(42, "mlp" if text else "transformer_encoder", D, P)
                done_items, t1 = 0, time.perf_counter()
                with torch.no_grad():
                    if text:        # embeddings_generator.py:81-89 + train.py:347,253-254, 16 captions per CPU batch
                        g = torch.Generator().manual_seed(3)
                        while done_items < 16 or (time.perf_counter() - t1 < args.cpu_seconds and done_items < 4096):
                            tk = inp[done_items % n_global:done_items % n_global + 16].cpu()
                            e = O.clip_encode_text(tk, clip_sd)
                            e = O.noise_injection(e, 0.016, noise=torch.randn(e.shape, generator=g))
                            O.clip_project(e, sdm, "mlp", P)
                            done_items += tk.shape[0]
                    else:           # predictions_runner.py:207-232: one image at a time, reference-shaped beam search
                        enc = O.clip_encode_image_resnet if rn else O.clip_encode_image
                        while done_items < 1 or (time.perf_counter() - t1 < args.cpu_seconds and done_items < 64):
                            e = enc(inp[done_items:done_items + 1].cpu(), clip_sd)
                            pe = O.clip_project(O.normalize_prefix(e.float()), sdm, "transformer_encoder", P)
                            O.generate_beam_ref(sdm, pe, 5, STOP_ID, T)
                            done_items += 1
                cdt = time.perf_counter() - t1
                cpu = {"value": round(done_items / cdt, 4), "unit": "items/s", "cores": th, "kind": "port", "host_cpus": os.cpu_count(),
                       "sample": f"{done_items} {'captions (batches of 16)' if text else 'images (one at a time, reference-shaped beam search without KV cache)'} "
                                 f"of the same chain through oracle/capdec_oracle.py in {cdt:.1f} s wall on {th} threads"}
            except Exception as ex:
                cpu = {"value": None, "unit": "items/s", "cores": 0, "kind": "port", "sample": "failed: " + str(ex)[:200]}
        emit(json.dumps({"metric": f"{'captions' if text else 'images'}/sec, side workload {args.workload}",
                          "clip_tower_kernels": {k: {"ms_est": round(v["ms"] * v["calls"] / v["launches"], 2),
                                                     "launches": v["calls"], "avg_ms": round(v["ms"] / v["launches"], 4),
                                                     **({"tflops": round(v["flops"] / v["ms"] * 1e-9, 1)} if v["flops"] > 0 and v["ms"] > 0 else {})}
                                                 for k, v in prof.items() if v["launches"]},
                          "value": round(n_global * args.steps / dt, 2), "unit": "items/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
                          "higher_is_better": True, "data": "synthetic",
                          "dtype": {"fp16": "f16 towers (one fp16 plane per GEMM operand, fp32 accumulate: the reference's CLIP arithmetic on a GPU)",
                                    "bf16": "bf16 towers"}.get(prec, "f32 towers (two fp16 planes, 3 MFMAs per product)"),
                          "roofline": roof, "cpu_baseline": cpu,
                          "config": {"workload": args.workload, "clip": "RN50x4" if rn else "ViT-B/32",
                                     "items_per_step": n_global, "clip_precision": prec,
                                     "gemm_mode": model.engine.gemm_mode(),
                                     **({"token_rows": "clip.tokenize-shaped: SOT + 8-20 random ids + EOT, zero-padded to 77 "
                                                       "(synth.synthetic_clip_tokens; mean EOT position %.1f) -- the tower computes the "
                                                       "positions up to each chunk's last EOT, so the rate depends on caption length "
                                                       "(CAPDEC_CLIP_TRUNC=0: all 77 positions)" % float(inp.argmax(dim=-1).float().mean())}
                                        if text else {})}}))


def other_configs(args, dev, note):
    """BASELINE.json's other configurations next to the metric line (N = 1; `other_configs` of the record): configs[1]
    (5000 greedy captions, MLP mapper, bf16 operands + bf16 KV cache), configs[3] (CLIP ViT-B/32 encode_text + noise + MLP
    mapper, 20 000 captions, fp16 towers) and configs[4] (ViT-B/32 encode_image + TransformerMapper + beam 5, 2014 images) --
    two timed passes each, inputs resident, a `roofline` of each tower's dominant kernel family; their own full lines (more
    passes, cpu_baseline) come from `bench.py --workload ...`.  Failures are recorded, never raised."""
    import copy
    from capdec_amd.gpt2_prefix import ClipCaptionModel, MappingType
    from capdec_amd.predictions_runner import caption_ids
    res = {}
    for name, over in (("configs[3] text_embed f16", dict(workload="text_embed", captions=20000, gemm_mode="f16")),
                       ("configs[4] image_beam vit_b32", dict(workload="image_beam", captions=2014, gemm_mode=None, clip="vit_b32"))):
        try:
            note("other configs: " + name)
            a = copy.copy(args)
            a.steps, a.warmup, a.cpu_seconds, a.profile_every = 2, 1, 0.0, 1
            for k, v in over.items():
                setattr(a, k, v)
            got = []
            side_workload(a, 1, 0, dev, got.append)
            r = json.loads(got[-1])
            res[name] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "dtype", "roofline", "config")}
        except Exception as ex:
            res[name] = {"error": str(ex)[:300]}
        torch.cuda.empty_cache()
    try:
        note("other configs: configs[1] greedy_mlp bf16")
        P, T = args.prefix_length, args.entry_length
        m = ClipCaptionModel(P, clip_length=10, prefix_dim=512, num_layers=8, mapping_type=MappingType.MLP).to(dev).eval()
        m.load_state_dict(synth.hot_state_dict(42, "mlp", 512, P))
        m.engine.set_gemm_mode("bf16")
        e = synth.synthetic_clip_embeddings(5000, 512, seed=0, normalize=False).to(dev)
        caption_ids(m, e, STOP_ID, beam=False, entry_length=T)
        m.engine.profile_enable(7)
        m.engine.profile_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            caption_ids(m, e, STOP_ID, beam=False, entry_length=T)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        pf = m.engine.profile_get()
        m.engine.profile_enable(False)
        g = pf.get("gemm_x1")
        ach = g["flops"] / g["ms"] * 1e-9 if g and g["ms"] > 0 else None
        res["configs[1] greedy_mlp bf16"] = {
            "value": round(5000 / dt, 1), "unit": "captions/s", "ms_per_step": round(dt * 1e3, 2), "steps": 3,
            "dtype": "bf16 (GEMM operands and KV cache bf16, fp32 accumulate)",
            "roofline": {"bound": "mfma", "kernel": "gemm_x1", "achieved": round(ach, 1) if ach else None, "peak": PEAK_BF16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_MFMA_TFLOPS, 4) if ach else None, "traffic": None},
            "config": {"workload": "greedy_mlp", "captions_per_step": 5000, "gemm_mode": "bf16"}}
    except Exception as ex:
        res["configs[1] greedy_mlp bf16"] = {"error": str(ex)[:300]}
    return res


def train_workload(args, world, rank, dev, emit=print):
    """Side workload: the train step (reference train.py:344-354) at the reference's default geometry -- batch 34
    (train.py:411), prefix_length = prefix_length_clip = 40, TransformerMapper with 8 layers on 640-d (RN50x4) embeddings,
    20 caption tokens per sample.  --train-scope prefix: --only_prefix (ClipCaptionPrefix, GPT-2 frozen, eval mode);
    full: the reference's default run (ClipCaptionModel: GPT-2 trained too, dropouts 0.1 from the device's Philox stream).
    A "step" = noise injection + forward + loss + backward + AdamW + scheduler for one batch; the timed steps are
    ENQUEUED (no device round trip per step: the losses are read back once, after the timed region).  Data parallelism
    over ranks would need a gradient all-reduce that this path does not have: N = 1 only.  The CPU leg times the
    oracle's hand-written step (torch CPU ops) on the same batch.  `roofline`: the dominant kernel family (the GEMMs of
    forward and backward on the two-fp16-plane kernels: ceiling = dense fp16 MFMA peak / 3) with the algorithmic FLOPs the
    launchers count (2 M N K per product, padding excluded)."""
    from capdec_amd import train as Tr
    from capdec_amd.gpt2_prefix import ClipCaptionModel, ClipCaptionPrefix, MappingType
    if world != 1:
        raise SystemExit("bench.py --workload train_step: one GPU only (no gradient all-reduce on this path)")
    P, D, B, L, nlay = 40, 640, args.train_batch, 20, 8
    full = args.train_scope == "full"
    sd = synth.hot_state_dict(42, "transformer_encoder", D, P, P, nlay)
    cls = ClipCaptionModel if full else ClipCaptionPrefix
    model = cls(P, clip_length=P, prefix_size=D, num_layers=nlay, mapping_type=MappingType.TransformerEncoder).to(dev)
    model.load_state_dict(sd)
    model.train()
    g = torch.Generator().manual_seed(9)
    tokens = torch.randint(1, synth.GPT2_SMALL.vocab, (B, L), generator=g)
    lens = torch.randint(8, L + 1, (B,), generator=g)
    lens[0] = L
    tokens[torch.arange(L)[None, :] >= lens[:, None]] = 0           # right padding, as train.ClipCocoDataset pads
    mask = torch.cat((torch.ones(B, P), (tokens != 0).float()), dim=1)
    prefix = synth.synthetic_clip_embeddings(B, D, seed=6).to(dev)
    opt = Tr.AdamW(model.parameters(), lr=2e-5)
    sched = Tr.get_linear_schedule_with_warmup(opt, 5000, 10 * 16000)
    eng = model.engine

    def step(wait):
        x = Tr.noise_injection(prefix, 0.016, seed=11)
        loss = Tr.train_step(model, opt, tokens, mask, x, wait=wait)
        sched.step()
        return loss

    first = None
    for _ in range(max(1, args.warmup)):
        first = step(True)
    eng.profile_enable(1)
    eng.profile_reset()
    eng.train_loss(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last, loss_sum, nsteps = eng.train_loss()
    prof = eng.profile_get()
    eng.profile_enable(False)
    kern = {k: {"ms_per_step": round(v["ms"] / args.steps, 3), "launches_per_step": round(v["calls"] / args.steps, 1),
                **({"tflops": round(v["flops"] / v["ms"] * 1e-9, 1)} if v["flops"] > 0 and v["ms"] > 0 else {})}
            for k, v in prof.items() if v["launches"]}
    gemm = [(k, v) for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0]
    roof = None
    if gemm:
        k, v = max(gemm, key=lambda kv: kv[1]["ms"])
        peak = 2500.0 / 3.0 if "f16x2" in k else (157.3 if k == "gemm_f32" else 2500.0)
        ach = v["flops"] / v["ms"] * 1e-9
        roof = {"bound": "mfma", "kernel": k, "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": None,
                "avg_launch_ms": round(v["ms"] / max(1, v["launches"]), 4), "launches": v["launches"],
                "share_of_step": round(v["ms"] / (dt * 1e3), 3),
                "flops_per_step": round(v["flops"] / args.steps, 0),
                "note": "algorithmic 2 M N K of every GEMM of the family (forward + dX + dW products), hipEvent-timed per "
                        "launch on the launch stream; peak = dense fp16 MFMA 2.5 PFLOP/s / 3 MFMAs per fp32 product"}
    cpu = None
    if args.cpu_seconds > 0:
        from oracle import capdec_oracle as O
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        x_cpu = prefix.cpu()
        t1 = time.perf_counter()
        n_cpu = 0
        while n_cpu < 1 or (time.perf_counter() - t1 < args.cpu_seconds and n_cpu < 3):
            O.train_step_loss_and_grads(sd, tokens, x_cpu, "transformer_encoder", P, clip_length=P, num_layers=nlay, train_gpt=full)
            n_cpu += 1
        cdt = time.perf_counter() - t1
        cpu = {"value": round(n_cpu / cdt, 4), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{n_cpu} step(s) of the oracle's hand-written forward + backward (no optimizer update, no dropout) on the "
                         f"same batch, {cdt:.1f} s wall"}
    what = "full model: GPT-2 trained too, dropout 0.1, reference train.py:344-354 default" if full else \
        "frozen GPT-2, reference train.py:344-354 --only_prefix"
    emit(json.dumps({"metric": f"train steps/sec, side workload train_step ({what})",
                     "value": round(args.steps / dt, 3), "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                     "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
                     "dtype": "f32 (GEMMs: fp32 emulated with 3 fp16 MFMAs per product, fp32 accumulate)" if roof and "f16x2" in roof["kernel"] else "f32",
                     "data": "synthetic",
                     "samples_per_s": round(B * args.steps / dt, 1),
                     "loss_first_last_mean": [round(first, 4), round(last, 4), round(loss_sum / max(1, nsteps), 4)],
                     "kernels": kern, "roofline": roof,
                     "config": {"workload": "train_step", "scope": args.train_scope, "batch": B, "prefix_length": P, "caption_tokens": L,
                                "mapper": "transformer, 8 layers, 640-d", "gpt2": "small, trained, dropout 0.1" if full else "small, frozen",
                                "optimizer": "AdamW (transformers 4.24 semantics), linear warm-up"},
                     "cpu_baseline": cpu}))


class Watchdog:
    """Every rank of a multi-GPU run carries one: if the run is still going after `limit_s` seconds the rank says WHERE it
    is stuck on stderr and exits with status 3 (torchrun then takes the other ranks down) -- an unattended 8-GPU run must
    end with a readable line, not with the driver's clock."""

    def __init__(self, rank, world, limit_s):
        import threading
        self.rank, self.world, self.limit, self.phase, self.t0 = rank, world, float(limit_s), "start", time.perf_counter()
        self._done = threading.Event()
        if self.limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def note(self, phase):
        self.phase = phase

    def _run(self):
        if not self._done.wait(self.limit):
            print("bench.py: rank %d of %d still in phase '%s' after %.0f s (--rank-timeout): aborting this rank"
                  % (self.rank, self.world, self.phase, time.perf_counter() - self.t0), file=sys.stderr, flush=True)
            os._exit(3)

    def stop(self):
        self._done.set()


def per_rank_table(world, rank, n_local, seconds, device, all_gather_floats):
    """[{rank, device, captions, seconds, captions_per_s}] on every rank: what each rank did in the timed region (the
    driver can see that N ranks really decoded, and which one was slowest)"""
    rows = all_gather_floats([float(rank), float(device), float(n_local), float(seconds)])
    return [{"rank": int(r[0]), "device": int(r[1]), "captions_per_step": int(r[2]), "seconds": round(r[3], 4),
             "captions_per_s": round(r[2] / r[3], 2) if r[3] > 0 else None} for r in rows]


def dry_run(args, world, rank, emit):
    """--dry-run: the N-rank control flow of this file on CPU -- gloo process group, caption sharding, a fake decode
    (token ids are a function of the caption index), the id gather, barrier + max-over-ranks timing, the per-rank table,
    the scaling check against a one-rank pass, the watchdog -- everything except the GPU work.  tests/test_host_logic.py
    runs it with 2 and with 8 ranks."""
    import torch.distributed as dist
    from capdec_amd import distributed as cdist
    wd = Watchdog(rank, world, args.rank_timeout)
    if world > 1:
        wd.note("init_process_group(gloo)")
        dist.init_process_group("gloo")
    n_global, T = args.captions, args.entry_length

    def fake_step(r, w):
        lo, hi = cdist.shard_bounds(n_global, r, w)
        idx = torch.arange(lo, hi, dtype=torch.int32)
        ids = (idx[:, None] * 7 + torch.arange(T, dtype=torch.int32)[None, :]) % 50257
        lens = (idx % T + 1).to(torch.int32)
        scores = -(idx.float() + 1.0) / 100.0
        return cdist.gather_ids(ids, lens, n_global, scores) if w > 1 else (ids, lens, scores)

    def barrier():
        if world > 1:
            dist.barrier()

    def all_gather_floats(vals):
        t = torch.tensor(vals, dtype=torch.float64)
        if world == 1:
            return [t.tolist()]
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return [p.tolist() for p in parts]

    wd.note("timed region")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = fake_step(rank, world)
    t_local = time.perf_counter() - t0
    barrier()
    mx = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    lo, hi = cdist.shard_bounds(n_global, rank, world)
    table = per_rank_table(world, rank, hi - lo, t_local / max(args.steps, 1), rank, all_gather_floats)
    wd.note("scaling check")
    one = fake_step(0, 1)
    same = all(bool((a == b).all()) for a, b in zip(one, out))
    barrier()
    if world > 1:
        dist.destroy_process_group()
    wd.stop()
    if rank == 0:
        emit(json.dumps({"dry_run": True, "n_gpus": world, "ranks": [r["rank"] for r in table], "per_rank": table,
                          "captions_per_step": n_global, "ids_equal_to_1gpu": same, "ms_per_step": round(float(mx) / args.steps * 1e3, 3),
                          "backend": "gloo" if world > 1 else None}))


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) with torch.distributed.run and
    pass the same arguments through; the JSON line is printed by rank 0 of that job."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["CAPDEC_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--train-batch", type=int, default=34, help="--workload train_step: samples per batch (reference train.py:411)")
    ap.add_argument("--train-scope", choices=["prefix", "full"], default="prefix",
                    help="--workload train_step: prefix = --only_prefix (GPT-2 frozen); full = the reference's default run "
                         "(GPT-2 trained too, dropout 0.1)")
    ap.add_argument("--workload", choices=["beam_transformer", "greedy_mlp", "text_embed", "image_beam", "train_step"],
                    default="beam_transformer",
                    help="beam_transformer = BASELINE metric config (default); greedy_mlp = configs[1] shape; "
                         "text_embed = configs[3] (CLIP ViT-B/32 encode_text + noise + mapper); "
                         "image_beam = configs[4] (ViT-B/32 encode_image + TransformerMapper + beam 5)")
    ap.add_argument("--clip", choices=["vit_b32", "rn50x4"], default="vit_b32",
                    help="image_beam only: the CLIP image tower (rn50x4 = the reference's default backbone, 288 x 288 -> 640-d)")
    ap.add_argument("--captions", type=int, default=5000,
                    help="captions per step: IN TOTAL with --scaling strong (default: COCO-val 5k), PER GPU with --scaling weak")
    ap.add_argument("--entry-length", type=int, default=67)
    ap.add_argument("--prefix-length", type=int, default=10)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): the metric's 5000 captions sharded over the ranks; weak: 5000 per GPU")
    ap.add_argument("--no-scaling-check", action="store_true",
                    help="N > 1, strong scaling: skip the extra 1-GPU pass (rank 0 alone on all captions) and the one-step "
                         "weak-scaling pass that fill `scaling_check`")
    ap.add_argument("--gemm-mode", choices=["bf16x3", "f32", "bf16", "f16x2", "f16"], default=None,
                    help="f16x2 / bf16x3: fp32-accurate split-operand MFMA GEMMs (parity with the fp32 reference); f32: native "
                         "fp32 MFMA; bf16: bf16 GEMM operands, fp32 accumulate (BASELINE configs[1]; NOT the headline: "
                         "token ids are no longer bit-identical to the fp32 reference)")
    ap.add_argument("--profile-every", type=int, default=7,
                    help="hipEvent-time every N-th launch of each kernel family inside the timed region (1 = all; "
                         "7 is coprime to the 4-GEMM / 12-layer launch cycles, so every shape is sampled evenly)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0,
                    help="wall-clock budget of the CPU baseline; 0 = SKIP the CPU baseline (unless --cpu-captions is given explicitly)")
    ap.add_argument("--cpu-captions", type=int, default=None,
                    help="CPU baseline on this many WHOLE captions (default 8, about a minute of host time; SURVEY D.5: 32); "
                         "0 = a --cpu-seconds time budget of decode steps instead")
    ap.add_argument("--no-checks", action="store_true",
                    help="skip the untimed passes after the timed region (ids_checked: 16 captions decoded alone in "
                         "batch-invariant mode against the big batch; attn_decode_diverged: one step with beams that never "
                         "share history)")
    ap.add_argument("--stop-profile", choices=["coco", "none"], default="coco",
                    help="coco (default, N = 1, beam workload at its default sizes): after the timed region, the workload in which "
                         "captions stop -- a stop id chosen so that the mean caption length is ~11 tokens; captions/s with "
                         "finished-caption compaction on / off, rows per step, shard imbalance, an oracle check (`stop_profile`); "
                         "and the entry_length = 12 point SURVEY D.2 asks for (`entry_length_12`).  none: skip both")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1, default workload: skip the short runs of BASELINE configs[1], [3], [4] that fill `other_configs`")
    ap.add_argument("--no-smi", action="store_true", help="do not sample rocm-smi (clock / power) during the timed region")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="N > 1: a rank still running after this many seconds prints the phase it is stuck in and exits "
                         "with status 3 (0 = no watchdog)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU (CPU tests): join a gloo group, all-gather the ranks, print "
                         "{n_gpus, ranks} and exit")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))
    # stdout carries exactly ONE line (the JSON record): RCCL prints a version banner to stdout when a communicator is
    # created, so everything else written to fd 1 during the run is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(line):
        sys.stdout.flush()
        os.write(json_fd, (line + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    force_dist = os.environ.get("CAPDEC_FORCE_DIST") == "1" and "RANK" in os.environ   # exercise RCCL with 1 rank
    use_dist = world > 1 or force_dist
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        return dry_run(args, world, rank, emit)
    wd = Watchdog(rank, world, args.rank_timeout if world > 1 else 0)
    wd.note("init_process_group(nccl)")
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from capdec_amd import distributed as cdist
    from capdec_amd.gpt2_prefix import ClipCaptionModel, MappingType
    from capdec_amd.predictions_runner import caption_ids

    if args.workload in ("text_embed", "image_beam"):
        return side_workload(args, world, rank, dev, emit)
    if args.workload == "train_step":
        return train_workload(args, world, rank, dev, emit)
    beam = args.workload == "beam_transformer"
    mapper = "transformer_encoder" if beam else "mlp"
    P, T, B = args.prefix_length, args.entry_length, (5 if beam else 1)
    n_global = args.captions * world if args.scaling == "weak" else args.captions
    model = ClipCaptionModel(P, clip_length=10, prefix_dim=512, num_layers=8,
                             mapping_type=MappingType.TransformerEncoder if beam else MappingType.MLP).to(dev).eval()
    model.load_state_dict(synth.hot_state_dict(42, mapper, 512, P))
    n_alloc = max(n_global, args.captions * world) if (world > 1 and not args.no_scaling_check) else n_global
    emb_all = synth.synthetic_clip_embeddings(n_alloc, 512, seed=0, normalize=False).to(dev)   # resident in HBM
    emb = emb_all[:n_global]
    eng = model.engine
    if args.gemm_mode:
        eng.set_gemm_mode(args.gemm_mode)

    def run_step(e, r, w):
        ids, lens, scores = caption_ids(model, e, STOP_ID, beam=beam, beam_size=5, entry_length=T, rank=r, world=w)
        return cdist.gather_ids(ids, lens, e.shape[0], scores) if w > 1 or force_dist else (ids, lens, scores)

    def step():
        return run_step(emb, rank, world)

    def barrier():
        if use_dist:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        import torch.distributed as dist
        tt = torch.tensor([x], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    def note(msg):
        wd.note(msg)
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    def all_gather_floats(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if not use_dist:
            return [t.tolist()]
        import torch.distributed as dist
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        return [p.tolist() for p in parts]
    t_start = time.perf_counter()
    note("weights loaded, warm-up")
    for _ in range(args.warmup):
        out = step()
    eng.decode_counters()                       # reset the saturation counter: the timed region reports its own
    eng.profile_enable(max(1, args.profile_every))
    eng.profile_reset()
    smi = SmiSampler(local_rank) if (rank == 0 and not args.no_smi) else None
    barrier()
    note("timed region: %d steps" % args.steps)
    if smi:
        smi.start()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0           # this rank's own time (the metric uses the max over ranks, after the barrier)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    note("timed region done")
    lo_r, hi_r = cdist.shard_bounds(n_global, rank, world)
    per_rank = per_rank_table(world, rank, hi_r - lo_r, t_local / max(args.steps, 1), local_rank, all_gather_floats)
    power = smi.stop() if smi else None
    prof = eng.profile_get()
    eng.profile_enable(False)
    counters = eng.decode_counters()
    second_pass = (eng.second_pass_rows(), eng.decode_stats()["row_steps"]) if beam else None   # of the last timed step
    assert out[0].shape[0] == n_global and int(out[1].min()) >= 1

    # ---- untimed checks (rank 0, N = 1): (a) ids_check -- the whole batch decoded once more in BATCH-INVARIANT mode (one
    # summation order whatever the launch size) and captions [0:16] decoded ALONE in that mode: those 16 must reproduce
    # rows [0:16] of the big batch bit for bit (ids, lengths, scores), and the timed (default-mode) batch is compared
    # with the invariant one caption by caption (other kernel variants: same captions up to fp32 round-off);
    # (b) one more step with beams that never share history: the worst-case K/V traffic of the decode attention
    ids_check = diverged = oracle_check = None
    if world == 1 and not args.no_checks:
        try:        # (an extra measurement must never take the metric line down: failures are reported in the record)
            k16 = min(16, n_global)
            note("ids_check")
            eng.set_batch_invariant(True)
            inv = run_step(emb, 0, 1)
            sub = run_step(emb[:k16], 0, 1)
            eng.set_batch_invariant(False)
            eq = (sub[0] == inv[0][:k16]).flatten(1).all(1) & (sub[1] == inv[1][:k16])
            same_default = (inv[0] == out[0]).flatten(1).all(1) & (inv[1] == out[1])
            if sub[2] is not None:
                eq = eq & (sub[2] == inv[2][:k16])
            ids_check = {"ids_checked": k16, "ids_equal": int(eq.sum()), "ok": int(eq.sum()) == k16,
                         "default_mode_captions_identical_to_invariant_mode": round(float(same_default.float().mean()), 5),
                         "note": "captions [0:%d] decoded alone vs rows [0:%d] of the %d-caption batch, both in batch-invariant "
                                 "mode: token ids, lengths and scores compared bit for bit; the timed default-mode batch "
                                 "against the invariant-mode batch: fraction of captions with identical ids"
                                 % (k16, k16, n_global)}
            if not ids_check["ok"]:
                print("bench.py: ids_check FAILED: %s" % ids_check, file=sys.stderr)
        except Exception as ex:
            eng.set_batch_invariant(False)
            ids_check = {"ok": False, "error": str(ex)[:300]}
        # (c) oracle_check: four captions of the TIMED default-mode batch against the CPU oracle (KV-cached restatement of
        # reference gpt2_prefix_eval.py:50-115 / :118-198) run on exactly their prefixes -- the headline configuration
        # compared at its own size on every bench run (tests/test_hip_parity.py does 24 + 32 of them)
        try:
            note("oracle_check")
            oracle_check = oracle_compare(model, emb, out, mapper, beam, P, T, rows=[0, n_global // 3, (2 * n_global) // 3, n_global - 1])
            if not oracle_check.get("ok"):
                print("bench.py: oracle_check FAILED: %s" % oracle_check, file=sys.stderr)
        except Exception as ex:
            oracle_check = {"ok": False, "error": str(ex)[:300]}
    # ---- the workload in which captions stop, and the entry_length = 12 point: untimed extras of the line
    stop_prof = t12 = None
    if world == 1 and beam and args.stop_profile == "coco" and not args.no_checks:
        try:
            stop_prof = stop_profile(model, emb, mapper, beam, P, T, max(1, min(args.steps, 5)), note)
            if stop_prof.get("oracle_check") and not stop_prof["oracle_check"].get("ok"):
                print("bench.py: stop_profile oracle_check FAILED: %s" % stop_prof["oracle_check"], file=sys.stderr)
        except Exception as ex:
            eng.set_compact(True)
            stop_prof = {"error": str(ex)[:300]}
        try:
            note("entry_length 12")
            k12 = max(1, min(args.steps, 5))
            caption_ids(model, emb, STOP_ID, beam=beam, beam_size=5, entry_length=12)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(k12):
                caption_ids(model, emb, STOP_ID, beam=beam, beam_size=5, entry_length=12)
            torch.cuda.synchronize()
            d12 = (time.perf_counter() - ta) / k12
            t12 = {"value": round(n_global / d12, 1), "unit": "captions/s", "ms_per_pass": round(d12 * 1e3, 2), "timed_passes": k12,
                   "entry_length": 12,
                   "note": "the same workload with entry_length 12 (SURVEY D.2: real COCO captions are ~11 tokens; all 12 steps run)"}
        except Exception as ex:
            t12 = {"error": str(ex)[:300]}
    # ---- reduced-precision modes (configs[1]: bf16): agreement with the fp32-accurate path on the same captions -- free
    # running (sequences diverge after the first flipped token) and teacher-forced (per-step arg-max given the fp32 ids)
    match = None
    if not beam and eng.gemm_mode() in ("bf16", "f16") and rank == 0:
        from capdec_amd.predictions_runner import prefix_from_embeddings
        lo, hi = cdist.shard_bounds(n_global, rank, world)
        sub = emb[lo:hi][:2000]
        low_mode = eng.gemm_mode()
        pe = prefix_from_embeddings(model, sub)
        ids_low, _ = eng.decode_greedy(pe, STOP_ID, T)
        tf_low = None
        eng.set_gemm_mode("f16x2")
        ids_ref, _ = eng.decode_greedy(pe, STOP_ID, T)
        eng.set_gemm_mode(low_mode)
        tf_low, st = eng.decode_greedy_forced(pe, ids_ref)
        eq = (ids_low == ids_ref)
        first_div = torch.where(eq.all(1), torch.full((eq.shape[0],), T, device=eq.device), (~eq).float().argmax(1))
        margin = st[:, :, 0] - st[:, :, 1]
        match = {"captions": int(sub.shape[0]), "free_running_token_match": round(float(eq.float().mean()), 4),
                 "identical_captions": round(float(eq.all(1).float().mean()), 4),
                 "mean_tokens_before_first_difference": round(float(first_div.float().mean()), 2),
                 "teacher_forced_argmax_match": round(float((tf_low == ids_ref).float().mean()), 4),
                 "teacher_forced_match_where_margin_gt_0.1": round(float((tf_low == ids_ref)[margin > 0.1].float().mean()), 4)}

    # ---- the same gather through the C ABI's own RCCL communicator (capdec_comm_init / capdec_gather_rows: no
    # torch.distributed in the data path) -- outside the timed region, reported as `capi_collective`
    capi_collective = None
    if use_dist:
        try:
            cdist.capi_comm_from_torch(eng)
            rccl_rank, rccl_ranks = eng.comm_info()          # ncclCommUserRank / ncclCommCount: what RCCL itself reports
            lo, hi = cdist.shard_bounds(n_global, rank, world)
            g_ids = eng.gather_rows(out[0][lo:hi].contiguous(), n_global)
            g_sc = eng.gather_rows(out[2][lo:hi].contiguous(), n_global) if out[2] is not None else None
            ok = bool((g_ids == out[0]).all()) and (g_sc is None or bool((g_sc == out[2]).all()))
            torch.cuda.synchronize()
            t0c = time.perf_counter()
            for _ in range(10):
                eng.gather_rows(out[0][lo:hi].contiguous(), n_global)
            torch.cuda.synchronize()
            capi_collective = {"ok": ok and rccl_ranks == world and rccl_rank == rank, "ranks": world, "rccl_ranks": rccl_ranks,
                               "ms_per_gather": round((time.perf_counter() - t0c) * 100, 3)}
            eng.comm_destroy()
        except Exception as ex:          # never let the optional check take the metric line down
            capi_collective = {"ok": False, "error": str(ex)[:300]}

    # ---- N > 1, strong scaling: the 1-GPU rate of this box (rank 0 alone decodes all captions; the other ranks wait
    # at the barrier) and one weak-scaling step (args.captions per GPU) -- extra fields, outside the timed region
    scaling_check = None
    if world > 1 and args.scaling == "strong" and not args.no_scaling_check:
        barrier()
        t1 = 0.0
        if rank == 0:
            run_step(emb, 0, 1)
            torch.cuda.synchronize()
            ta = time.perf_counter()
            one = run_step(emb, 0, 1)
            torch.cuda.synchronize()
            t1 = time.perf_counter() - ta
            same = bool((one[0] == out[0]).all()) and bool((one[1] == out[1]).all())
        barrier()
        run_step(emb_all, rank, world)
        barrier()
        tb = time.perf_counter()
        run_step(emb_all, rank, world)
        barrier()
        tw = max_over_ranks(time.perf_counter() - tb)
        if rank == 0:
            v1, vs, vw = n_global / t1, n_global * args.steps / dt, emb_all.shape[0] / tw
            scaling_check = {"n1_value": round(v1, 2), "n1_note": "rank 0 alone, all %d captions, 1 timed step" % n_global,
                             "ids_equal_to_1gpu": same, "per_gpu_value": round(vs / world, 2),
                             "strong_efficiency": round(vs / (world * v1), 4),
                             "weak_value": round(vw, 2), "weak_captions_per_gpu": args.captions,
                             "weak_efficiency": round(vw / (world * v1), 4)}

    mode_name = eng.gemm_mode()
    # ---- one more step with beams that never share history: the worst-case K/V traffic of the decode attention.  The hook
    # lives in the MEASUREMENT build of the library only (libcapdec_hip_measure.so, -DCAPDEC_MEASURE): the model moves to
    # a context of that library for this last, untimed step (the product context and its KV cache are released first)
    if world == 1 and not args.no_checks and beam:
        try:
            note("diverged-beam step (measurement build)")
            model.use_measurement_build(True)
            engm = model.engine
            engm.set_gemm_mode(mode_name)
            engm.set_debug_diverge(True)
            engm.profile_enable(1)
            engm.profile_reset()
            run_step(emb, 0, 1)
            torch.cuda.synchronize()
            pd = engm.profile_get()["attn_decode"]
            cd = engm.decode_counters()
            diverged = {"avg_ms": round(pd["ms"] / max(pd["launches"], 1), 4), "launches_timed": pd["launches"],
                        "kv_slots_per_position": round(cd["kv_slots_per_position"], 3),
                        "note": "one untimed step in which every beam continues itself (no shared history): the worst-case "
                                "K/V traffic of the decode attention; run in the measurement build of the library (the "
                                "shipped one has no such hook); results are not the reference's beam search"}
            engm.set_debug_diverge(False)
        except Exception as ex:
            diverged = {"error": str(ex)[:300]}

    if rank == 0:
        value = n_global * args.steps / dt
        mode = mode_name
        est = lambda f: f["ms"] * f["calls"] / f["launches"] if f and f["launches"] else 0.0
        if mode in ("bf16", "f16"):
            fam, kname, peak = prof["gemm_x1"], "gemm_x1_kernel", PEAK_BF16_MFMA_TFLOPS
            peak_note = "dense bf16 / fp16 MFMA peak (16-bit operands, one MFMA per product)"
            products = 1
        elif mode == "bf16x3":
            # every fp32 product is six bf16 MFMA products: the kernel's ceiling in fp32-equivalent FLOP/s is
            # the dense bf16 peak / 6; achieved = algorithmic (2*M*N*K) FLOPs / measured kernel time.
            # Dominant kernel = the GEMM family with the most device time (packed-A LDS-DMA kernel or the
            # fp32-activation kernel).
            cands = [("gemm_bf16x3p", "gemm_bf16x3p_kernel"), ("gemm_bf16x3", "gemm_bf16x3_kernel")]
            fkey, kname = max(cands, key=lambda kv: est(prof.get(kv[0])))
            fam, peak = prof[fkey], PEAK_BF16_MFMA_TFLOPS / 6.0
            peak_note = "fp32-equivalent TFLOP/s: dense bf16 MFMA peak 2500 / 6 products per fp32 product"
            products = 6
        elif mode == "f16x2":
            # every fp32 product is three fp16 MFMA products (hi*hi + hi*lo + lo*hi on two fp16 planes): ceiling =
            # dense fp16 MFMA peak (= the bf16 peak) / 3
            fam, kname, peak = prof["gemm_f16x2p"], "gemm_f16x2p_kernel", PEAK_BF16_MFMA_TFLOPS / 3.0
            peak_note = "fp32-equivalent TFLOP/s: dense fp16 MFMA peak 2500 / 3 products per fp32 product"
            products = 3
        else:
            fam, kname, peak = prof["gemm_f32"], "gemm_f32_kernel", PEAK_F32_MFMA_TFLOPS
            peak_note = "dense fp32-input MFMA peak"
            products = None
        traffic = None
        try:   # HBM-side bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes
            pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
            if beam and n_global // world == pmc.get("captions_per_gpu", 5000) and pmc.get("gemm_mode", "bf16x3") == mode \
                    and kname in pmc:
                traffic = pmc[kname]["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        gemm_ms = fam["ms"] / max(fam["launches"], 1)
        achieved = fam["flops"] / (fam["ms"] * 1e-3) / 1e12 if fam["ms"] > 0 else 0.0
        n_local = cdist.shard_size(n_global, world)
        alg = algorithmic_flops_per_caption(P, T, B, "transformer" if beam else "mlp") * n_local * args.steps
        rec = {
            "metric": "captions/sec (whole node), COCO-val 5k, prefix_len=10 beam=5, 1/2/4/8 GPUs",
            "value": round(value, 2), "unit": "captions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16 (GEMM operands and KV cache bf16, fp32 accumulate; residual stream / LayerNorm / softmax f32)",
                      "f16": "f16 (GEMM operands fp16, fp32 accumulate; residual stream / LayerNorm / softmax / KV cache f32)",
                      "bf16x3": "f32 (operands split into 3 bf16 planes, 6 bf16 MFMAs per product, fp32 accumulate)",
                      "f16x2": "f32 (operands split into 2 fp16 planes, 3 fp16 MFMAs per product, fp32 accumulate)"}[mode],
            "data": "synthetic",
            "config": {"workload": ("COCO-val-5k-shaped: %d x 512-d synthetic CLIP embeddings %s -> normalise -> "
                                    "%s -> GPT-2 small KV-cached %s, prefix_len %d, entry_length %d, hot-init seeded "
                                    "weights (never emit the stop id: all %d steps run)")
                       % (args.captions, "per GPU" if args.scaling == "weak" else "in total (sharded over the ranks)",
                          "TransformerMapper(8 layers)" if beam else "MLP mapper",
                          "beam-5 decode" if beam else "greedy decode", P, T, T),
                       "captions_per_step": n_global, "captions_per_gpu": n_local, "beam": B,
                       "parallelism": f"caption-shard dp{world}", "tokens_per_s": round(value * T, 1)},
            "per_gpu_value": round(value / world, 2),
            "roofline": {"bound": "mfma", "kernel": kname + " (dominant GEMM family; the other projections and the fused "
                         "lm_head variant are listed in `kernels`)",
                         "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4), "traffic": traffic, "peak_note": peak_note,
                         "traffic_note": "bytes per launch at the L2<->fabric boundary (2 x FETCH_SIZE + WRITE_SIZE, "
                                         "profiles/%s; Infinity-Cache hits are counted)" % PMC_TRAFFIC_FILE,
                         "avg_launch_ms": round(gemm_ms, 4), "launches_timed": fam["launches"],
                         "launches": fam["calls"],
                         "mfma_tflops_executed": round(achieved * products, 1) if products else None,
                         "mfma_frac_of_dense_peak": round(achieved * products / PEAK_BF16_MFMA_TFLOPS, 4) if products else None,
                         "vs_native_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         # against the power-capped ceiling of a PURE MFMA stream on random fp16 operands (measured, see the
                         # constant above): what is left between this kernel and it is the energy of everything around the MFMAs
                         "frac_of_power_capped_mfma_stream": round(achieved * products / F16_MFMA_AT_POWER_CAP_TFLOPS, 4) if products in (1, 3) else None,
                         # the same achieved rate against round 1's ceiling (six bf16 MFMAs per fp32 product: 2500 / 6)
                         "vs_six_product_ceiling_416_7": round(achieved / (PEAK_BF16_MFMA_TFLOPS / 6.0), 4),
                         "whole_path_tflops": round(alg / dt / 1e12, 2)},
            # per family: ms_est = hipEvent time of the timed launches scaled to all launches of the timed region
            "kernels": {k: {"ms_est": round(v["ms"] * v["calls"] / v["launches"], 2), "launches": v["calls"],
                            "launches_timed": v["launches"], "avg_ms": round(v["ms"] / v["launches"], 4),
                            **({"tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)} if v["flops"] and v["ms"] else {})}
                        for k, v in prof.items() if v["launches"]},
            "profile_every": max(1, args.profile_every),
            "ids_check": ids_check,
            "power": power,
            "saturated_operand_quads": counters["saturated_quads"],
            "scaling_check": scaling_check,
            "per_rank": per_rank,
            "rccl_ranks": (capi_collective or {}).get("rccl_ranks") if use_dist else None,
            "capi_collective": capi_collective,
            "match_vs_fp32": match,
        }
        if beam and "attn_decode" in rec["kernels"]:
            # what the attention's HBM traffic depends on: distinct K/V slots read per (caption, position); 1 = the beams
            # of a caption share their whole history, 5 = none of it (synthetic hot-init beams converge quickly)
            rec["kernels"]["attn_decode"]["kv_slots_per_position"] = round(counters["kv_slots_per_position"], 3)
            if diverged:
                rec["kernels"]["attn_decode_diverged"] = diverged
        if second_pass and "gemm_f16x2p_lmhead_topk" in rec["kernels"]:
            # (row, step) pairs of one timed step whose lm_head top 5 needed the exact second pass (3 candidates kept per
            # vocabulary tile, capdec.h: capdec_decode_second_pass_rows), of the row-steps the step ran
            rec["kernels"]["gemm_f16x2p_lmhead_topk"]["second_pass_rows"] = second_pass[0]
            rec["kernels"]["gemm_f16x2p_lmhead_topk"]["row_steps"] = second_pass[1]
        rec["oracle_check"] = oracle_check
        rec["stop_profile"] = stop_prof
        rec["entry_length_12"] = t12
        if power and power.get("sclk_mhz"):
            # the dominant kernel against the peak AT THE CLOCK THE CHIP ACTUALLY HELD (it runs at its package power cap)
            rec["roofline"]["frac_at_measured_clock"] = round(achieved / (peak * power["sclk_mhz"] / 2400.0), 4)
        if world == 1 and beam and not args.no_checks and not args.no_other_configs and args.captions == 5000 and not args.gemm_mode:
            try:
                model.release()                 # (the 140 GB KV cache of the metric's batch: the towers get the whole device)
                rec["other_configs"] = other_configs(args, dev, note)
            except Exception as ex:
                rec["other_configs"] = {"error": str(ex)[:300]}
        note("cpu baseline")
        if args.cpu_captions is None:      # not given: whole captions by default, nothing at all with --cpu-seconds 0
            args.cpu_captions = 8 if args.cpu_seconds > 0 else 0
        if world == 1 and (args.cpu_seconds > 0 or args.cpu_captions > 0):
            try:
                rec["cpu_baseline"] = cpu_baseline(mapper, B, P, T, args.cpu_seconds, captions=args.cpu_captions)
            except Exception as ex:       # the reported baseline must not cost the metric line
                rec["cpu_baseline"] = {"value": None, "unit": "captions/s", "cores": 0, "kind": "port", "sample": "failed: " + str(ex)[:200]}
        else:
            rec["cpu_baseline"] = None
        emit(json.dumps(rec))
    if use_dist:
        import torch.distributed as dist
        wd.note("final barrier")
        dist.barrier()
        dist.destroy_process_group()
    wd.stop()


if __name__ == "__main__":
    main()

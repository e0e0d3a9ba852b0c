"""Shared pieces of the low-precision parity tests (``-m gpu``).

Since round 3 no test decides "may this id differ?" with an adjustable margin gate.  The criterion has no free parameter:
drive the model's own decode-step path along the f32 oracle's token history (``decoding.forced_decode_logits``), measure
err[b, s] = max_v |logit_gpu - logit_oracle| at every (row, step), and allow a differing arg-max only where the oracle's
top-1 margin is <= 2 * err[b, s] -- two logit vectors that differ by at most e per entry cannot rank two candidates more
than 2e apart differently.  Up to its first divergence a free-running row HAS followed the oracle's history, so the same
err[b, s] applies to it.  Reference: DecodingTask._main_loop behind scripts/transcribe_single.py:49-56."""
import numpy as np
import torch

# Relative logit-error ceiling of the bf16 path (matrices, activations, K/V caches in bf16; f32 accumulation and residual
# stream): max |logit_bf16 - logit_f32| / std(finite logits) over every step of the teacher-forced history, i.e. a maximum
# over ~51 865 x steps x rows entries.  A regression guard on the measured numbers, NOT what decides whether an id may differ.
# Measured on MI355X (r03): whisper-small 12+12 on encoder features 0.043 (rms ~0.01), small width
# 0.038, fp8 weights 0.020-0.036.  Context, same weights / tokens on the CPU: rounding ONLY the matrices and features to
# bf16 (f32 arithmetic) already gives 0.024, a plain torch all-bf16 decoder (bf16 residual stream too) 0.067 -- the HIP path
# sits between the two, where a path with f32 accumulation and an f32 residual stream belongs.
BF16_LOGIT_ERR_CEILING = 0.06
# 72 rows of WHITE-NOISE features at whisper-medium width (a nearly flat cross-attention softmax, 2 layers with 4x output
# scaling): measured 0.115; weights-only rounding 0.057, torch all-bf16 0.196 on the same rows.
BF16_LOGIT_ERR_CEILING_WHITE_NOISE = 0.16


def divergence_report(got: np.ndarray, ref, n_init: int):
    """token-match rate and, per row, the first step where the ids differ with the oracle's margin at that step."""
    body_g, body_r = got[:, n_init:], ref.tokens[:, n_init:]
    n = min(body_g.shape[1], body_r.shape[1])
    eq = body_g[:, :n] == body_r[:, :n]
    firsts = []
    for b in range(eq.shape[0]):
        bad = np.flatnonzero(~eq[b])
        firsts.append(None if bad.size == 0 else (int(bad[0]), float(ref.margins[b, bad[0]])))
    # rows are compared up to their first divergence: later ids follow a different history
    prefix = sum((n if f is None else f[0]) for f in firsts)
    return {"token_match": float(eq.mean()), "prefix_match": prefix / float(eq.size), "first_divergence": firsts}


def step_logit_errors(trace: torch.Tensor, ref_logits: np.ndarray) -> np.ndarray:
    """max over the vocabulary entries the oracle keeps finite of |GPU logit - oracle logit|, per (row, step).
    trace [B, S, V] (device or host), ref_logits [B, S, V] with -inf at suppressed ids."""
    out = np.zeros(ref_logits.shape[:2], dtype=np.float64)
    for b in range(ref_logits.shape[0]):
        got = trace[b].float().cpu().numpy()
        ok = np.isfinite(ref_logits[b])
        out[b] = np.where(ok, np.abs(got - np.where(ok, ref_logits[b], 0.0)), 0.0).max(axis=1)
    return out


def logit_spread(ref_logits: np.ndarray) -> float:
    return float(ref_logits[np.isfinite(ref_logits)].std())


def assert_divergences_explained(got_tokens: np.ndarray, ref, err: np.ndarray, n_init: int, what: str):
    """every first divergence of a free-running row sits at a step whose oracle margin is <= 2 x the measured logit error"""
    rep = divergence_report(got_tokens, ref, n_init)
    for b, f in enumerate(rep["first_divergence"]):
        if f is not None:
            step, margin = f
            assert margin <= 2.0 * err[b, step], (what, "row", b, "step", step, "oracle margin", margin, "measured logit error",
                                                   float(err[b, step]))
    return rep


def check_low_precision_decode(model, feats, ref, init, always, first, eot, what: str, ceiling: float = BF16_LOGIT_ERR_CEILING):
    """The whole criterion for one low-precision model against an oracle GreedyResult computed with keep_logits=True:
    teacher-forced logit error at every step, every differing teacher-forced choice and every free-running first divergence
    explained by it, and the error itself under the stated ceiling.  Returns (err [B, S], report)."""
    from whisper_ipa_amd.decoding import forced_decode_logits, greedy_decode_tokens

    n_init = len(init)
    n_new = ref.tokens.shape[1] - n_init
    trace, chosen = forced_decode_logits(model, feats, ref.tokens, n_init, always, first, eot)
    err = step_logit_errors(trace, ref.step_logits)
    flips = chosen != ref.tokens[:, n_init:]
    assert (ref.margins[flips] <= 2.0 * err[flips]).all(), (what, ref.margins[flips], err[flips])
    res = greedy_decode_tokens(model, feats, init, always, first, eot, max_new_tokens=n_new, stop_on_eot=False)
    rep = assert_divergences_explained(res.tokens, ref, err, n_init, what)
    spread = logit_spread(ref.step_logits)
    rep.update(max_logit_err=float(err.max()), logit_std=spread, rel_err=float(err.max() / spread), forced_flips=int(flips.sum()),
               largest_flipped_margin=float(ref.margins[flips].max()) if flips.any() else 0.0, steps=int(flips.size))
    print(f"\n[{what}] along the reference history: max logit error {rep['max_logit_err']:.4f} = {rep['rel_err']:.4f} of the logit std "
          f"{spread:.3f} (mean of per-step maxima {err.mean():.4f}); {rep['forced_flips']} of {rep['steps']} teacher-forced choices differ "
          f"(largest reference margin among them {rep['largest_flipped_margin']:.4f}); free-running token match {rep['token_match']:.4f}")
    assert err.max() < ceiling * spread, (what, float(err.max()), spread)
    return err, rep


def masked_margins(trace: torch.Tensor, always, first) -> np.ndarray:
    """top-1 minus top-2 of the filtered logits, per (row, step): what GreedyResult.margins holds, computed from a logit
    trace [B, S, V] (suppress_first applies to step 0 only)"""
    t = trace.clone()
    t[:, :, list(always)] = float("-inf")
    if len(first):
        t[:, 0, list(first)] = float("-inf")
    top2 = torch.topk(t, 2, dim=-1).values
    return (top2[..., 0] - top2[..., 1]).cpu().numpy()


def norm64(t) -> float:
    """L2 norm in float64 -- the yardstick for every norm a test asserts.  torch's float32 ``.norm()`` is NOT one: on the
    [51865, 128] token-embedding gradient of the training tests it reads 0.99988 where the float64 norm (and the oracle's own
    f32 ``sqrt(sum(g*g))``, the reference's formula) is 0.9999995 -- measured r05, 1.2e-4 off, enough to fail or pass a 1e-4
    bound by the host's luck."""
    return float(t.detach().double().cpu().norm())


def clipped_norm(n: float, max_norm: float = 1.0) -> float:
    """the norm ``clip_grad_dict`` leaves on a tensor of norm n > max_norm: n * max_norm / (n + 1e-6) (train_whisper_ipa.py:295-298)"""
    return n * max_norm / (n + 1e-6)

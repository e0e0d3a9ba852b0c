"""Offline evaluation with the behaviour of the reference's scripts/evaluate_model.py: transcribe a
test JSON (``audio_path`` / ``ipa_transcription`` records) with the base model and with a fine-tuned
checkpoint, print the first three examples, PER / PFER (mean ± std), the comparison table and the
target thresholds (reference :127-268), same CLI flags (:272-308).

Differences that the hardware asks for:
  * clips are decoded in batches of ``--batch-size`` (the reference is batch 1, :181-212), ``--passes-in-flight`` batches
    at a time on one GPU (whisper_ipa_amd.pipeline.transcribe_batches -- the schedule bench.py times); the result per clip is
    the same because every kernel on the path is batch-invariant and passes share no state
    (tests/test_gpu_model.py::test_full_size_bench_workload_properties, ::test_transcribe_batches_*);
  * under ``torchrun`` every rank takes a contiguous slice of the test list with a full weight
    replica and no collective on the data path; the hypotheses are gathered once at the end;
  * the base model is a local directory (no hub access) and the base-model leg uses the same
    mel -> encoder -> decode(language="en", without_timestamps=True) path as the checkpoint leg
    (the reference's base leg goes through mlx_whisper.transcribe, which wraps the same calls for
    a clip of at most 30 s).
All compute runs in libwipa.so on the GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from evaluate_ipa import evaluate_batch  # noqa: E402
from whisper_ipa_amd import parallel  # noqa: E402
from whisper_ipa_amd.audio import load_audio, pad_or_trim  # noqa: E402
from whisper_ipa_amd.decoding import DecodingOptions  # noqa: E402
from whisper_ipa_amd.load_models import load_model, overlay_decoder_weights  # noqa: E402
from whisper_ipa_amd.pipeline import transcribe_batches  # noqa: E402


def load_checkpoint_model(checkpoint_path: str, base_model: str = "mlx-community/whisper-small-mlx"):
    """reference :20-79: base architecture in fp32 + the checkpoint's ``decoder.*`` tensors."""
    print(f"Loading base model architecture: {base_model}")
    model = load_model(base_model)
    model.set_dtype(torch.float32)
    try:
        n = overlay_decoder_weights(model, checkpoint_path)
    except FileNotFoundError:
        print(f"WARNING: No weights found at {checkpoint_path}, using base model")
        return model
    print(f"Found {n} decoder parameters to load")
    print("✓ Decoder weights loaded successfully")
    return model


def load_clips(audio_paths: List[str]):
    """host side of reference :184-189 for a list of clips: (audio [n_ok, 480000] f32 or None, positions of the readable clips);
    a clip that cannot be read is reported and yields "" (:202-204)."""
    clips, slots = [], []
    for i, path in enumerate(audio_paths):
        try:
            clips.append(np.asarray(pad_or_trim(load_audio(path)), dtype=np.float32))
            slots.append(i)
        except Exception as e:
            print(f"\nError transcribing {path}: {e}")
    return (torch.from_numpy(np.stack(clips)) if clips else None), slots


def transcribe_clips(model, audio_paths: List[str], options: DecodingOptions, batch_size: int = 64,
                     passes_in_flight: int = 4, progress=None) -> List[str]:
    """reference :181-212 for the whole list: ``batch_size`` clips per batch, ``passes_in_flight`` batches in flight on one GPU
    (whisper_ipa_amd.pipeline.transcribe_batches: log-mel -> encoder -> greedy decode per batch on its own stream set, the
    files of the next batches read on a helper thread meanwhile).  One text per path, "" where the file could not be read."""
    texts = [""] * len(audio_paths)
    metas = []  # (first clip of the batch, positions of its readable clips), in submission order

    def batches():
        for b in range(0, len(audio_paths), batch_size):
            audio, slots = load_clips(audio_paths[b:b + batch_size])
            if audio is None:
                continue
            metas.append((b, slots))
            yield audio

    for r in transcribe_batches(model, batches(), options, passes_in_flight=passes_in_flight, prefetch=passes_in_flight):
        b, slots = metas[r.index]
        for i, text in zip(slots, r.texts):
            texts[b + i] = text.strip()
        if progress is not None:
            progress(min(b + batch_size, len(audio_paths)))
    return texts


def transcribe_batch(model, audio_paths: List[str], n_mels: int, options: DecodingOptions) -> List[str]:
    """ONE batch with nothing else in flight (round 4's entry point, kept for callers that hold a single batch)"""
    assert n_mels == model.dims.n_mels
    return transcribe_clips(model, audio_paths, options, batch_size=max(1, len(audio_paths)), passes_in_flight=1)


def evaluate_model(model_path: str, test_data_path: str, num_samples: Optional[int] = None, model_name: str = "Model",
                   is_checkpoint: bool = False, n_mels: int = 80, base_model: str = "mlx-community/whisper-small-mlx",
                   batch_size: int = 64, passes_in_flight: int = 4) -> Dict:
    rank, world_size = parallel.world()
    say = print if rank == 0 else (lambda *a, **k: None)
    say("=" * 70)
    say(f"Evaluating {model_name}")
    say("=" * 70)
    say(f"\nLoading test data: {test_data_path}")
    with open(test_data_path) as f:
        test_data = json.load(f)
    if num_samples:
        test_data = test_data[:num_samples]
        say(f"Evaluating on {num_samples} samples")
    else:
        say(f"Evaluating on all {len(test_data)} samples")
    say(f"\nModel: {model_path}")
    if is_checkpoint:
        say("\nLoading checkpoint...")
        model = load_checkpoint_model(model_path, base_model=base_model)
    else:
        model = load_model(model_path)
        model.set_dtype(torch.float32)
    if model.dims.n_mels != n_mels:
        say(f"NOTE: --n-mels {n_mels} does not match the model ({model.dims.n_mels}); using the model's")
        n_mels = model.dims.n_mels
    options = DecodingOptions(language="en", without_timestamps=True)

    lo, hi = parallel.shard_bounds(len(test_data), world_size, rank)
    mine = test_data[lo:hi]
    say("\nTranscribing test samples...")
    local_hyp = transcribe_clips(model, [s["audio_path"] for s in mine], options, batch_size=batch_size,
                                 passes_in_flight=passes_in_flight,
                                 progress=lambda n: say(f"  {n}/{len(mine)} clips on rank 0", flush=True))
    hypotheses = local_hyp
    if world_size > 1:
        import torch.distributed as dist

        parts: List = [None] * world_size
        dist.all_gather_object(parts, local_hyp)
        hypotheses = [h for part in parts for h in part]
    references = [s["ipa_transcription"] for s in test_data]

    results = evaluate_batch(references, hypotheses)  # per-sample scores included (reference evaluate_ipa.py:370-378)
    for i in range(min(3, len(references))):
        say(f"\nSample {i + 1}:")
        say(f"  Reference:  {references[i]}")
        say(f"  Hypothesis: {hypotheses[i]}")
        say(f"  PER:  {results['per_scores'][i]:.2f}%")
        say(f"  PFER: {results['pfer_scores'][i]:.2f}%" + ("  (PER: no feature table)" if results["pfer_is_per_fallback"] else ""))
    say("\n" + "=" * 70)
    say(f"{model_name} - Overall Results")
    say("=" * 70)
    say(f"\nPER (Phone Error Rate):         {results['per']:.2f}% (±{results['per_std']:.2f}%)")
    say(f"PFER (Phone Feature Error Rate): {results['pfer']:.2f}% (±{results['pfer_std']:.2f}%)")
    say(f"Number of samples: {results['num_samples']}")
    return results


def compare_models(base_results: Dict, trained_results: Dict) -> None:
    """reference :235-268."""
    print("\n" + "=" * 70)
    print("Model Comparison")
    print("=" * 70)
    print(f"\n{'Metric':<30} {'Base Model':<15} {'Trained Model':<15} {'Improvement':<15}")
    print("-" * 70)
    for label, key in (("PER (Phone Error Rate)", "per"), ("PFER (Feature Error Rate)", "pfer")):
        diff = base_results[key] - trained_results[key]
        print(f"{label:<30} {base_results[key]:>6.2f}%{'':<8} {trained_results[key]:>6.2f}%{'':<8} {diff:>+6.2f}%")
    print("\n" + "=" * 70)
    print("Benchmark Comparison (from paper)")
    print("=" * 70)
    print("Target scores (zero-shot, unseen languages):")
    print("  - Best in paper (1k samples): 21.2% PFER")
    print("  - Wav2Vec2Phoneme: 22.4% PFER")
    print("  - Human IAA: 19.6% PFER")
    print("\nTarget scores (supervised, trained languages):")
    print("  - Overall: 5.7% PFER")
    print("  - Polish (best): 2.5% PFER")
    pfer = trained_results["pfer"]
    if pfer < 50:
        print("\n✅ MINIMUM VIABLE: PFER < 50% achieved!")
    if pfer < 30:
        print("✅ GOOD: PFER < 30% achieved!")
    if pfer < 25:
        print("✅ EXCELLENT: PFER < 25% achieved!")
    if pfer < 21.2:
        print("🎉 SOTA: Beat paper's best zero-shot result!")


def main(argv=None) -> Dict:
    ap = argparse.ArgumentParser(description="Evaluate Whisper-IPA model")
    ap.add_argument("--checkpoint", type=str, default="checkpoints/whisper-ipa-english/checkpoint-250",
                    help="Path to trained model checkpoint")
    ap.add_argument("--base-model", type=str, default="mlx-community/whisper-small-mlx",
                    help="Base model: local directory with config.json + weights.safetensors")
    ap.add_argument("--test-data", type=str, default="data/processed/english_only_test_ipa.json", help="Path to test data JSON")
    ap.add_argument("--num-samples", type=int, default=100, help="Number of samples to evaluate (default: 100, use 0 for all)")
    ap.add_argument("--skip-base", action="store_true", help="Skip base model evaluation (only evaluate checkpoint)")
    ap.add_argument("--n-mels", type=int, default=128, help="Number of mel bins (80 for small/medium, 128 for large)")
    ap.add_argument("--batch-size", type=int, default=64, help="clips decoded together per GPU")
    ap.add_argument("--passes-in-flight", type=int, default=4,
                    help="batches kept in flight on one GPU (whisper_ipa_amd.pipeline.transcribe_batches); 1 = one batch at a time")
    ap.add_argument("--results-json", type=str, default=None, help="also write both result dicts here (rank 0)")
    ap.add_argument("--allow-byte-fallback", action="store_true",
                    help="run without the Whisper vocabulary (WIPA_TIKTOKEN unset): hypotheses render ids >= 256 as <|idN|>; "
                         "synthetic weights only")
    args = ap.parse_args(argv)
    from whisper_ipa_amd.tokenizer import get_tokenizer, require_real_vocabulary

    require_real_vocabulary(get_tokenizer(True), args.allow_byte_fallback, "scoring a model's transcriptions")

    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")
    torch.set_num_threads(parallel.host_threads_per_rank())  # the host's cores shared among the ranks on it
    rank, _ = parallel.world()
    num_samples = None if args.num_samples == 0 else args.num_samples
    base_results = None
    if not args.skip_base:
        base_results = evaluate_model(args.base_model, args.test_data, num_samples, model_name="Base Whisper Model",
                                      is_checkpoint=False, n_mels=args.n_mels, base_model=args.base_model, batch_size=args.batch_size,
                                      passes_in_flight=args.passes_in_flight)
    trained_results = evaluate_model(args.checkpoint, args.test_data, num_samples, model_name="Trained Checkpoint",
                                     is_checkpoint=True, n_mels=args.n_mels, base_model=args.base_model, batch_size=args.batch_size,
                                      passes_in_flight=args.passes_in_flight)
    if rank == 0:
        if base_results:
            compare_models(base_results, trained_results)
        print("\n" + "=" * 70)
        print("✅ Evaluation Complete!")
        print("=" * 70)
        if args.results_json:
            Path(args.results_json).write_text(json.dumps({"base": base_results, "trained": trained_results}, indent=2))
    return {"base": base_results, "trained": trained_results}


if __name__ == "__main__":
    main()

"""Fine-tune entry point with the CLI, console line and artefacts of the reference's
scripts/train_whisper_ipa.py (flags :648-709, console line :557-561, CSV columns :105-112,
checkpoint-N/{model.safetensors,training_state.json} :410-443, best-checkpoint/ :574-588,
training_config.json :91-99, training_summary.json :625-636) on top of whisper_ipa_amd.

What differs because of the platform: the base model is a LOCAL directory (no hub access); the
arithmetic is libwipa (HIP) instead of MLX; one process per GPU when launched with
``python -m torch.distributed.run --nproc-per-node N scripts/train_whisper_ipa.py ...``: rank r takes
slice r of one shared ``np.random.choice`` draw, the loss normalisation and the gradients are
all-reduced over RCCL, the per-tensor clip and AdamW run on the reduced gradients
(whisper_ipa_amd/training.py), rank 0 logs / validates / saves.
"""
from __future__ import annotations

import argparse
import csv
import json
import os
import platform
import re
import resource
import shutil
import sys
import time
from datetime import datetime
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from evaluate_ipa import evaluate_batch  # noqa: E402
from ipa_data_loader import create_data_loader  # noqa: E402
from whisper_ipa_amd import parallel  # noqa: E402
from whisper_ipa_amd.decoding import DecodingOptions  # noqa: E402
from whisper_ipa_amd.load_models import load_model, save_safetensors  # noqa: E402
from whisper_ipa_amd.pipeline import transcribe_batches  # noqa: E402
from whisper_ipa_amd.training import DecoderTrainer  # noqa: E402


def flatten_params(params, prefix: str = "") -> Dict[str, torch.Tensor]:
    flat = {}
    if isinstance(params, dict):
        for k, v in params.items():
            flat.update(flatten_params(v, f"{prefix}{k}."))
    elif isinstance(params, (list, tuple)):
        for i, v in enumerate(params):
            flat.update(flatten_params(v, f"{prefix}{i}."))
    else:
        flat[prefix[:-1]] = params
    return flat


def count_parameters(params) -> int:
    return sum(int(v.numel()) for v in flatten_params(params).values())


def get_hardware_info() -> Dict:
    info = {"platform": platform.platform(), "python": platform.python_version(), "torch": torch.__version__,
            "cpu_count": os.cpu_count()}
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(0)
        info.update({"gpu": p.name, "gpu_memory_gb": round(p.total_memory / 2**30, 1), "n_gpus": torch.cuda.device_count()})
    return info


def save_training_config(output_dir: Path, args_dict: Dict, hardware: Dict) -> None:
    """training_config.json with the reference's schema (:91-99): training_args / hardware / start_time."""
    with open(output_dir / "training_config.json", "w") as f:
        json.dump({"training_args": args_dict, "hardware": hardware, "start_time": datetime.now().isoformat()}, f, indent=2)


class TrainingLogger:
    TRAIN_COLUMNS = ["step", "loss", "lr", "step_time_sec", "samples_per_sec", "wall_clock_sec", "timestamp", "peak_memory_mb"]
    VAL_COLUMNS = ["step", "per", "pfer", "per_std", "pfer_std", "num_samples", "wall_clock_sec", "timestamp"]

    def __init__(self, output_dir: Path):
        self.train_log_path = output_dir / "training_log.csv"
        self.val_log_path = output_dir / "validation_log.csv"
        self.best_pfer, self.best_pfer_step = float("inf"), 0
        self.latest_val_per = self.latest_val_pfer = None
        for path, cols in ((self.train_log_path, self.TRAIN_COLUMNS), (self.val_log_path, self.VAL_COLUMNS)):
            if not path.exists():
                with open(path, "w", newline="") as f:
                    csv.writer(f).writerow(cols)

    @staticmethod
    def _peak_memory_mb() -> float:
        rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        return rss / (1024 * 1024) if platform.system() == "Darwin" else rss / 1024

    def log_train_step(self, step, loss, lr, step_time, batch_size, wall_clock_sec):
        with open(self.train_log_path, "a", newline="") as f:
            csv.writer(f).writerow([step, f"{loss:.6f}", f"{lr:.2e}", f"{step_time:.4f}", f"{batch_size / step_time:.2f}",
                                    f"{wall_clock_sec:.2f}", datetime.now().isoformat(), f"{self._peak_memory_mb():.1f}"])

    def log_validation(self, step, metrics, wall_clock_sec) -> bool:
        per, pfer = metrics["per"], metrics["pfer"]
        self.latest_val_per, self.latest_val_pfer = per, pfer
        with open(self.val_log_path, "a", newline="") as f:
            csv.writer(f).writerow([step, f"{per:.4f}", f"{pfer:.4f}", f"{metrics.get('per_std', 0):.4f}",
                                    f"{metrics.get('pfer_std', 0):.4f}", metrics.get("num_samples", ""),
                                    f"{wall_clock_sec:.2f}", datetime.now().isoformat()])
        if pfer < self.best_pfer:
            self.best_pfer, self.best_pfer_step = pfer, step
            return True
        return False


def freeze_encoder(model) -> None:
    print("\nFreezing encoder parameters...")
    model.encoder.freeze()
    print("  ✓ Encoder frozen")
    model.decoder.unfreeze()
    print("  ✓ Decoder unfrozen (trainable)")
    trainable, total = count_parameters(model.trainable_parameters()), count_parameters(model.parameters())
    print(f"\nTrainable parameters: {trainable:,} / {total:,} ({100 * trainable / total:.1f}%)")


def compute_loss(model, batch: Dict, tokenizer) -> torch.Tensor:
    """The reference's loss function (:207-263), name and signature kept, written against ``model.logits`` so that it is the
    template for a MODIFIED loss: frozen encoder forward (:223), teacher forcing on tokens[:, :-1] (:228-232), per-token CE
    (reduction 'none', :256), mask = (tgt != eot) | (cumsum(tgt == eot) == 1) (:242-247), sum / max(count, 1) (:260-261).
    ``model.logits`` is differentiable w.r.t. the decoder tensors (whisper_ipa_amd.training._DecoderLogits), so
    ``value_and_grad(model, compute_loss)`` below is the reference's ``nn.value_and_grad(model, loss_fn)`` (:284).  The
    production step (train_step / DecoderTrainer.loss_and_grads) computes the same loss with the fused CE kernels instead of
    torch ops on a [B, T, V] tensor."""
    tokens = batch["tokens"].to(model.device)
    with torch.no_grad():
        feats = model.embed_audio(batch["mel_features"]) if "audio_features" not in batch else batch["audio_features"]
    dec_in, tgt = tokens[:, :-1], tokens[:, 1:].long()
    logits = model.logits(dec_in, feats)
    is_eot = tgt == tokenizer.eot
    mask = (~is_eot) | (torch.cumsum(is_eot.long(), dim=1) == 1)
    ce = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), tgt.reshape(-1), reduction="none")
    return (ce.reshape(tgt.shape) * mask).sum() / torch.clamp(mask.sum(), min=1)


def value_and_grad(model, loss_fn):
    """``nn.value_and_grad(model, loss_fn)`` of the reference (:284) for a model whose decoder a DecoderTrainer owns: returns
    ``fn(model, batch, tokenizer) -> (loss, grads)`` with ``grads`` the nested dict of the TRAINABLE (decoder) tensors in
    mlx_whisper's names -- what ``clip_grad_dict`` (:287-303) walks.  The gradients are also left in the trainer's flat
    buffer, so ``trainer.apply_update()`` (per-tensor clip + AdamW) can follow directly.  Inside ``fn`` -- and only there --
    ``model.logits`` is the differentiable call (DecoderTrainer.differentiable_scope).  The flat gradient buffer is OVERWRITTEN
    (nothing accumulates onto what loss_and_grads left there), and the backward reduces over no process group: single process."""
    trainer = getattr(model, "_trainer", None)
    if trainer is None:
        raise RuntimeError("value_and_grad: create a whisper_ipa_amd.training.DecoderTrainer(model) first (it owns the decoder tensors)")

    def fn(model, batch, tokenizer):
        from whisper_ipa_amd.runtime import on_stream
        from whisper_ipa_amd.whisper import _unflatten

        leaves = trainer.leaves()
        for t in leaves.values():
            t.grad = None
        trainer.differentiable_scope = True  # model.logits inside loss_fn is the differentiable call, as under nn.value_and_grad
        try:
            with torch.enable_grad(), on_stream():
                loss = loss_fn(model, batch, tokenizer)
                loss.backward()
        finally:
            trainer.differentiable_scope = False
        with on_stream():
            for n, t in leaves.items():  # the flat buffer holds d(logits-path) only if the loss used other leaves too: copy .grad
                trainer.g(n).copy_(t.grad if t.grad is not None else torch.zeros_like(t))
        return loss.detach(), _unflatten({n: trainer.g(n) for n in trainer.names})

    return fn


def train_step(trainer: DecoderTrainer, batch: Dict, tokenizer):
    """reference :266-311 -> (loss, clipped grads); the arithmetic is DecoderTrainer.train_step.  ``batch["clip_keys"]``
    (dataset indices) is present when the frozen-encoder feature cache is on: mel_features then covers the uncached clips only."""
    return trainer.train_step(batch["mel_features"], batch["tokens"], tokenizer.eot, clip_keys=batch.get("clip_keys"))


VALIDATE_PASSES_IN_FLIGHT = 4  # validation batches in flight on one GPU (whisper_ipa_amd.pipeline.transcribe_batches)


def validate(model, dataset, tokenizer, num_samples: int = 100) -> Dict:
    """reference :314-407: batched greedy decode (language=None -> detection, fp16=False), PER / PFER.

    Data parallel: COLLECTIVE -- every rank calls it.  The validation batches (4 clips each, the reference's size) are dealt
    round-robin to the ranks, each rank decodes its share with its own weight replica, the (reference, hypothesis) pairs are
    gathered once and every rank scores the whole list in the reference's sample order, so all ranks return the same
    metrics.  (Round 2 validated on rank 0 alone while the other ranks idled in the next step's all-reduce.)"""
    rank, world = parallel.world()
    main = rank == 0
    if main:
        print(f"\nValidating on {num_samples} samples...")
    model.eval()
    val_batch_size = 4
    options = DecodingOptions(language=None, without_timestamps=True, fp16=False, length_penalty=1.0)
    mine = []  # (batch index, refs, hyps)
    metas = []  # (batch index, refs) of every batch handed to the pipeline, in submission order

    def batches():
        for i in range((num_samples + val_batch_size - 1) // val_batch_size):
            indices = list(range(i * val_batch_size, min((i + 1) * val_batch_size, num_samples)))
            if not indices:
                break
            if i % world != rank:
                continue
            try:
                batch = dataset.get_batch(indices)
                refs = [re.sub(r"<\|.*?\|>", "", tokenizer.decode(batch["tokens"][j].tolist())).strip() for j in range(len(indices))]
            except Exception as e:  # reference :393-396 keeps going
                print(f"Error during validation decoding: {e}")
                import traceback
                traceback.print_exc()
                continue
            metas.append((i, refs))
            yield batch["mel_features"]

    # the reference decodes one 4-clip batch at a time (:338-362); here VALIDATE_PASSES_IN_FLIGHT of them are in flight on their
    # own streams (whisper_ipa_amd.pipeline): same ids per clip, the encoder of one batch beside the decode loops of the others
    try:
        for r in transcribe_batches(model, batches(), options, passes_in_flight=VALIDATE_PASSES_IN_FLIGHT):
            i, refs = metas[r.index]
            mine.append((i, refs, [t.strip() for t in r.texts]))
    except Exception as e:  # reference :393-396: report and score what was decoded
        print(f"Error during validation decoding: {e}")
        import traceback
        traceback.print_exc()
    if world > 1:
        import torch.distributed as dist

        parts = [None] * world
        dist.all_gather_object(parts, mine)
        mine = [item for part in parts for item in part]
    mine.sort(key=lambda t: t[0])
    references = [r for _, refs, _ in mine for r in refs]
    hypotheses = [h for _, _, hyps in mine for h in hyps]
    if main and mine and mine[0][0] == 0:
        print("\nSample Predictions:")
        for k in range(min(3, len(mine[0][1]))):
            print(f"  Ref:  [{mine[0][1][k]}]\n  Pred: [{mine[0][2][k]}]\n" + "-" * 30)
    metrics = evaluate_batch(references, hypotheses)
    model.train()
    if main:
        print(f"Validation Results:\n  PER:  {metrics['per']:.2f}%\n  PFER: {metrics['pfer']:.2f}%")
    return metrics


def save_checkpoint(model, optimizer, step: int, loss, output_dir: Path, logger: Optional[TrainingLogger] = None,
                    start_time: Optional[float] = None, learning_rate: Optional[float] = None) -> None:
    ckpt = output_dir / f"checkpoint-{step}"
    ckpt.mkdir(parents=True, exist_ok=True)
    save_safetensors(str(ckpt / "model.safetensors"), flatten_params(model.parameters()))  # ALL weights (reference :421)
    state = {"step": step, "loss": float(loss.item()) if hasattr(loss, "item") else float(loss)}
    if start_time is not None:
        state["wall_clock_sec"] = time.time() - start_time
    if learning_rate is not None:
        state["learning_rate"] = learning_rate
    if logger is not None:
        state["best_pfer"] = logger.best_pfer if logger.best_pfer != float("inf") else None
        state["best_pfer_step"] = logger.best_pfer_step
        state["latest_val_per"] = logger.latest_val_per
        state["latest_val_pfer"] = logger.latest_val_pfer
    state["timestamp"] = datetime.now().isoformat()
    with open(ckpt / "training_state.json", "w") as f:
        json.dump(state, f, indent=2)
    print(f"  ✓ Saved checkpoint to {ckpt}")


def _init_distributed():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI
    # host threads: the usable cores shared among the ranks on this host (8 ranks x all cores would oversubscribe the node)
    torch.set_num_threads(parallel.host_threads_per_rank())
    return parallel.world()


def train(model_name: str, train_data_path: str, test_data_path: str, output_dir: str, num_steps: int = 1000,
          batch_size: int = 4, learning_rate: float = 1e-5, validate_every: int = 100, save_every: int = 500,
          test_run: bool = False, audio_root: str = "", seed: Optional[int] = None, exact_f32: bool = False,
          allow_byte_fallback: bool = False, cache_encoder_features: bool = True, feature_cache_clips: Optional[int] = None,
          clip_scope: str = "reference"):
    rank, world = _init_distributed()
    try:
        parallel.require_even_shards(batch_size, world)  # every rank gets clips; no mismatched collectives
    except ValueError as e:
        raise SystemExit(f"--batch-size: {e}")
    main = rank == 0
    output_dir = Path(output_dir)
    if main:
        output_dir.mkdir(parents=True, exist_ok=True)
        # the reference's nine keys (:477-487), in its order; what this platform adds comes after them
        args_dict = {"model_name": model_name, "train_data_path": train_data_path, "test_data_path": test_data_path,
                     "num_steps": num_steps, "batch_size": batch_size, "learning_rate": learning_rate,
                     "validate_every": validate_every, "save_every": save_every, "test_run": test_run,
                     "world_size": world, "f32_products": "exact" if exact_f32 else "split",
                     "cache_encoder_features": bool(cache_encoder_features), "clip_scope": clip_scope}
        save_training_config(output_dir, args_dict, get_hardware_info())
    logger = TrainingLogger(output_dir) if main else None
    print(f"Loading model: {model_name}")
    t0 = time.time()
    model = load_model(model_name)
    model.set_dtype(torch.float32)
    print(f"  ✓ Model loaded in {time.time() - t0:.1f}s")
    freeze_encoder(model)
    # mlx AdamW defaults (reference :513); the clip reaches what the reference's clip_grad_dict reaches (:287-303: dicts only,
    # the decoder.blocks list passes through) unless --clip-scope all
    trainer = DecoderTrainer(model, lr=learning_rate, f32_split=not exact_f32, clip_scope=clip_scope)
    n_mels = 128 if "large" in model_name else 80  # reference :517
    if model.dims.n_mels != n_mels:
        n_mels = model.dims.n_mels
    train_dataset = create_data_loader(train_data_path, multilingual=True, n_mels=n_mels, audio_root=audio_root,
                                       allow_byte_fallback=allow_byte_fallback)
    test_dataset = create_data_loader(test_data_path, multilingual=True, n_mels=n_mels, audio_root=audio_root,
                                      allow_byte_fallback=allow_byte_fallback)
    tokenizer = train_dataset.tokenizer
    if test_run:
        train_dataset.data = train_dataset.data[:100]
        num_steps = min(num_steps, 100)
    rng = np.random.default_rng(seed) if seed is not None else np.random.default_rng(int(time.time()) if world == 1 else 0)
    if cache_encoder_features:
        # The encoder is frozen (reference :187) yet recomputed every step (:223): its output is a pure function of the clip,
        # 4.6 MB per clip for whisper-small -- the whole training list fits the 288 GB HBM.  Bit-identical to recomputing.
        cache = trainer.enable_feature_cache(feature_cache_clips if feature_cache_clips is not None else
                                             min(len(train_dataset), trainer.feature_cache_capacity_default()))
        print(f"  ✓ Frozen-encoder feature cache: up to {cache.max_clips} clips "
              f"({cache.max_clips * model.dims.n_audio_ctx * model.dims.n_audio_state * 4 / 2**30:.1f} GiB of HBM)")

    print("\n" + "=" * 70 + f"\nStarting training for {num_steps} steps\n" + "=" * 70)
    model.train()
    start_time = time.time()
    latest_loss = None
    for step in range(1, num_steps + 1):
        try:
            draw = rng.choice(len(train_dataset), size=batch_size, replace=False)  # one shared draw (reference :548)
            mine = parallel.shard_indices(draw.tolist(), world, rank)
            batch, local_error = None, None
            try:
                if trainer.feature_cache is not None:
                    batch = train_dataset.get_batch(mine, audio_for=trainer.feature_cache.missing(mine))
                    batch["clip_keys"] = mine
                else:
                    batch = train_dataset.get_batch(mine)
            except Exception as e:  # e.g. an unreadable clip on ONE rank
                local_error = e
            # Failure must be collective: a rank that left the loop alone would leave the others waiting in the step's
            # all-reduces until the RCCL timeout.  One MAX all-reduce carries (error flag, token width); every rank
            # sees the flag and breaks in the same step.
            if world > 1:
                width = parallel.agree_on_step(0 if batch is None else batch["tokens"].shape[1], failed=local_error is not None)
                if width < 0:
                    raise RuntimeError(f"a data-parallel rank failed to build its batch at step {step}"
                                       + (f" (this rank: {local_error})" if local_error else ""))
                pad = width - batch["tokens"].shape[1]
                if pad:  # all ranks must use one token width: pad to the global maximum with EOT
                    batch["tokens"] = torch.nn.functional.pad(batch["tokens"], (0, pad), value=tokenizer.eot)
            elif local_error is not None:
                raise local_error
            step_start = time.time()
            loss, _ = train_step(trainer, batch, tokenizer)
            loss_value = float(loss.item())  # device sync, like mx.eval(loss) (reference :309)
            latest_loss = loss
            step_time = time.time() - step_start
            if main and (step % 10 == 0 or step <= 5):
                print(f"Step {step}/{num_steps} | Loss: {loss_value:.4f} | Time: {step_time:.3f}s | "
                      f"Samples/sec: {batch_size / step_time:.1f}")
                logger.log_train_step(step, loss_value, learning_rate, step_time, batch_size, time.time() - start_time)
            if step % validate_every == 0:
                metrics = validate(model, test_dataset, tokenizer, num_samples=min(100, len(test_dataset)))  # collective
            if main and step % validate_every == 0:
                if logger.log_validation(step, metrics, time.time() - start_time):
                    best = output_dir / "best-checkpoint"
                    if best.exists():
                        shutil.rmtree(best)
                    best.mkdir(parents=True, exist_ok=True)
                    save_safetensors(str(best / "model.safetensors"), flatten_params(model.parameters()))
                    with open(best / "training_state.json", "w") as f:
                        json.dump({"step": step, "pfer": metrics["pfer"], "per": metrics["per"],
                                   "timestamp": datetime.now().isoformat()}, f, indent=2)
                    print(f"  ✓ New best PFER {metrics['pfer']:.2f}% at step {step}")
            if main and step % save_every == 0:
                save_checkpoint(model, trainer, step, loss, output_dir, logger=logger, start_time=start_time,
                                learning_rate=learning_rate)
        except Exception as e:  # reference :598-602
            print(f"\n✗ Error at step {step}: {e}")
            import traceback
            traceback.print_exc()
            if world > 1:
                # exit non-zero so the launcher tears the whole job down: a lone `break` would leave the other ranks
                # inside this step's collectives until the RCCL timeout
                raise SystemExit(1)
            break

    if main:
        print("\n" + "=" * 70 + "\nTraining complete! Running final validation...\n" + "=" * 70)
    metrics = validate(model, test_dataset, tokenizer, num_samples=min(500, len(test_dataset)))  # collective
    if main:
        logger.log_validation(num_steps, metrics, time.time() - start_time)
        if latest_loss is not None:
            print("\nSaving final model...")
            save_checkpoint(model, trainer, num_steps, latest_loss, output_dir, logger=logger, start_time=start_time,
                            learning_rate=learning_rate)
            total = time.time() - start_time
            with open(output_dir / "training_summary.json", "w") as f:
                json.dump({"total_wall_clock_sec": total, "total_wall_clock_min": total / 60,
                           "final_loss": float(latest_loss.item()), "final_per": metrics["per"], "final_pfer": metrics["pfer"],
                           "best_pfer": logger.best_pfer if logger.best_pfer != float("inf") else None,
                           "best_pfer_step": logger.best_pfer_step, "end_time": datetime.now().isoformat()}, f, indent=2)
            print(f"\n✓ Training complete in {total / 60:.1f} minutes")
            print(f"  Final loss: {float(latest_loss.item()):.4f}\n  Final PER: {metrics['per']:.2f}%\n"
                  f"  Final PFER: {metrics['pfer']:.2f}%\n  Best PFER: {logger.best_pfer:.2f}% (step {logger.best_pfer_step})\n"
                  f"  Model saved to: {output_dir}")
        else:
            print("\n✗ Training failed - no loss computed")
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


def main():
    p = argparse.ArgumentParser(description="Fine-tune Whisper for IPA transcription")
    p.add_argument("--model", type=str, default="mlx-community/whisper-small-mlx",
                   help="base model: a LOCAL directory with config.json + weights.safetensors (hub names cannot resolve offline)")
    p.add_argument("--train-data", type=str, default="data/processed/english_only_train_ipa.json", help="Path to training JSON file")
    p.add_argument("--test-data", type=str, default="data/processed/english_only_test_ipa.json", help="Path to test JSON file")
    p.add_argument("--output-dir", type=str, default="checkpoints/whisper-ipa", help="Directory to save checkpoints")
    p.add_argument("--steps", type=int, default=10000, help="Number of training steps")
    p.add_argument("--batch-size", type=int, default=12, help="Batch size (global, split over the GPUs)")
    p.add_argument("--lr", type=float, default=1e-5, help="Learning rate")
    p.add_argument("--validate-every", type=int, default=1000, help="Validate every N steps")
    p.add_argument("--save-every", type=int, default=1000, help="Save checkpoint every N steps")
    p.add_argument("--test-run", action="store_true", help="Test run with only 100 samples")
    p.add_argument("--audio-root", type=str, default="", help="prefix for the relative audio_path entries of the JSON")
    p.add_argument("--exact-f32", action="store_true",
                   help="exact f32 products on the f32 MFMA in the large GEMMs, as the reference trains (145 against 87 ms per 32-clip "
                        "step); default since round 5: three split-bf16 MFMA terms per product (~5e-6 relative; loss within 1e-5 and "
                        "every decoder gradient within 3e-5 of the float32 CPU checker)")
    p.add_argument("--fast-f32", action="store_true", help="accepted for round-4 command lines: split products are the default now")
    p.add_argument("--no-cache-encoder-features", action="store_true",
                   help="recompute the frozen encoder for every clip of every step, as the reference does (default: keep each "
                        "clip's encoder output in HBM after its first use -- bit-identical results, about half the step time)")
    p.add_argument("--feature-cache-clips", type=int, default=None,
                   help="capacity of the frozen-encoder feature cache in clips (default: the training list, or what half of the "
                        "free HBM holds); clips beyond it are recomputed")
    p.add_argument("--clip-scope", choices=["reference", "all"], default="reference",
                   help="which gradients the per-tensor clip (max norm 1.0) reaches.  reference (default): exactly what the "
                        "reference's clip_grad_dict reaches -- it recurses through dicts only, so the decoder.blocks LIST is "
                        "passed through and only token_embedding.weight, positional_embedding and ln.* are clipped; "
                        "all: every decoder tensor by its own norm")
    p.add_argument("--allow-byte-fallback", action="store_true",
                   help="run without the Whisper vocabulary (WIPA_TIKTOKEN unset): raw-byte text ids; synthetic weights only")
    a = p.parse_args()
    train(model_name=a.model, train_data_path=a.train_data, test_data_path=a.test_data, output_dir=a.output_dir,
          num_steps=a.steps, batch_size=a.batch_size, learning_rate=a.lr, validate_every=a.validate_every,
          save_every=a.save_every, test_run=a.test_run, audio_root=a.audio_root, exact_f32=a.exact_f32,
          allow_byte_fallback=a.allow_byte_fallback, cache_encoder_features=not a.no_cache_encoder_features,
          feature_cache_clips=a.feature_cache_clips, clip_scope=a.clip_scope)


if __name__ == "__main__":
    main()

"""bench.py as the driver runs it, on small models so it takes seconds: the JSON contract, the `--gpus N` launcher
(rehearsed on ONE GPU: WIPA_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and moves the timing collectives over gloo --
RCCL refuses two ranks on one device) and `--mode train`.  ``pytest -m gpu``."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, share_gpu=False, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    if share_gpu:
        env["WIPA_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout
    return json.loads(lines[0])


CONTRACT = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config"}


def test_bench_single_rank_contract_and_rooflines():
    out = _run(["--steps", "3", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "2"])
    assert CONTRACT <= set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["unit"] == "audio-s/s" and out["value"] > 0 and out["passes_identical"] is True
    assert abs(out["value"] - 8 * 30.0 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 0.02
    assert "workload" in out["config"] and "model" not in out["config"]
    for key in ("roofline", "roofline_mfma", "decode_step"):
        r = out[key]
        assert r["bound"] in ("hbm", "mfma") and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3, key
    # convs + 5 GEMMs x 4 encoder layers; the 4 cross-K/V projections only run with a K/V cache (absorbed projections: none)
    # the default: absorbed-projection cross-attention (no cross-K/V GEMMs, the streaming kernel is the roofline kernel, half the
    # cross bytes per step)
    assert out["config"]["cross_attention"] == "absorbed"
    # honest labels: the workload string names a cross-K/V projection only when one runs, says how many clips are
    # in flight, and the latency figure (one pass at a time) sits next to the throughput figure
    assert "cross-K/V projection" not in out["config"]["workload"] and "absorbed" in out["config"]["workload"]
    assert out["config"]["clips_in_flight_per_gpu"] == 2 * 8 and "2 such passes (16 clips) in flight" in out["config"]["workload"]
    assert 0 < out["value_single_in_flight"] and abs(out["value_single_in_flight"] - 8 * 30.0 / (out["ms_per_pass_single_in_flight"] * 1e-3)) < 1.0
    assert out["roofline_mfma"]["gemm_launches"] == 2 + 5 * 4
    assert "cross_absorbed_v2_kernel" in out["roofline"]["kernel"] and out["roofline"]["layer_call"]["avg_ms"] > out["roofline"]["avg_launch_ms"]
    assert out["decode_step"]["bytes_per_step"] > out["decode_step"]["cross_kv_bytes"] > 0
    # several passes in flight: the streaming launch runs in 2 frame splits per clip (Whisper.cross_splits), the line says so, times
    # two such launches side by side, and carries the latencies of the library's default setting (4 splits) beside its own
    assert out["config"]["cross_frame_splits"] == 2 and "2 frame splits" in out["config"]["workload"]
    assert out["roofline"]["frame_splits"] == 2 and out["roofline"]["workgroups"] == 16 and out["roofline"]["two_launches_side_by_side"]["achieved"] > 0
    ds = out["default_splits"]
    assert ds["cross_frame_splits"] == 4 and ds["decode_step_ms"] > 0 and ds["ms_per_pass_single_in_flight"] > 0
    assert ds["streaming_launch"]["workgroups"] == 32 and ds["streaming_launch"]["frac"] > 0
    one = _run(["--steps", "2", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "1", "--no-cpu-baseline"])
    assert one["config"]["cross_frame_splits"] == 4 and "default_splits" not in one and "frame splits" not in one["config"]["workload"]
    # mlx_whisper's projected K / V caches (opt-in): four more GEMM launches, the fused cross block as the roofline kernel, twice
    # the cross bytes, and the same accounting when the absorbed line is read in cached-K/V terms
    ck = _run(["--steps", "3", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "2", "--no-cpu-baseline", "--cross-attention", "cached"])
    assert ck["config"]["cross_attention"] == "cached" and ck["passes_identical"] is True
    assert "+ cross-K/V projection" in ck["config"]["workload"]
    # cross_attention="auto" decides from (clips, new tokens): long outputs at a small batch take the cached form
    long_ = _run(["--steps", "2", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "2", "--no-cpu-baseline", "--new-tokens", "200"])
    assert long_["config"]["cross_attention"] == "cached" and long_["passes_identical"] is True
    assert ck["roofline_mfma"]["gemm_launches"] == 2 + 5 * 4 + 4
    assert "decode_cross_block" in ck["roofline"]["kernel"]
    assert out["decode_step"]["cross_kv_bytes"] * 2 == ck["decode_step"]["cross_kv_bytes"]
    assert out["decode_step"]["cached_kv_accounting"]["bytes_per_step"] == ck["decode_step"]["bytes_per_step"]


def test_bench_experiment_flags_keep_the_ids():
    """The schedule and --new-tokens change when work runs / how long a row is, not the transcription: one pass at a time in the
    same streaming-launch setting gives the checksum of two passes in flight.  The line says which scheduler was timed (the
    package's, not one of bench.py's own) and on how many hardware queues, and carries the evaluate_model-style run."""
    base = ["--steps", "2", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "2", "--no-cpu-baseline"]
    plain = _run(base)
    assert "whisper_ipa_amd.pipeline.TranscribePipeline" in plain["config"]["schedule"] and plain["config"]["hw_queues"] == 8
    ev = plain["evaluate_style"]
    assert ev["unit"] == "audio-s/s" and ev["value"] > 0 and ev["rows_transcribed"] == 8 * ev["batches"] and ev["decode_steps_per_batch"] == 64
    assert abs(ev["frac_of_value"] - ev["value"] / plain["value"]) < 1e-3
    serial = _run(["--steps", "2", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "1", "--cross-splits", "2", "--no-cpu-baseline"])
    assert serial["config"]["cross_frame_splits"] == 2 and serial["passes_identical"]
    assert serial["tokens_checksum"] == plain["tokens_checksum"]
    longer = _run(base + ["--new-tokens", "100"])
    assert longer["config"]["new_tokens"] == 100 and longer["passes_identical"]
    # --phase: the two halves of the pass alone (diagnostic lines that cannot be mistaken for the metric)
    for phase in ("enc", "dec"):
        d = _run(base + ["--phase", phase])
        assert d["diagnostic_phase"] == phase and d["ms_per_pass"] > 0 and "metric" not in d and "value" not in d


def test_bench_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE): the parent starts two ranks and relays rank 0's line."""
    out = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "tiny", "--batch", "8", "--pipeline", "2",
                "--no-cpu-baseline"], share_gpu=True)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["parallelism"].startswith("dp2")
    # whole-job aggregate: both ranks' clips over the max-over-ranks time
    assert abs(out["value"] - 2 * 8 * 30.0 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 0.02


def test_bench_train_mode_single_and_two_ranks():
    args = ["--mode", "train", "--model", "tiny", "--train-batch", "4", "--train-tokens", "16", "--steps", "2", "--warmup", "1"]
    one = _run(args)
    assert CONTRACT <= set(one) and one["unit"] == "clips/s" and one["dtype"] == "f32" and one["n_gpus"] == 1
    assert one["allreduce_exposed_ms_per_step"] == 0.0 and one["roofline"]["bound"] == "mfma"
    assert {"encoder_ms", "loss_and_grads_ms", "update_ms"} <= set(one["stages"])
    two = _run(["--gpus", "2"] + args, share_gpu=True)
    assert two["n_gpus"] == 2 and two["allreduce_exposed_ms_per_step"] >= 0.0 and two["grad_bytes"] == one["grad_bytes"]


def test_bench_under_torch_distributed_run_two_ranks():
    """The driver's multi-GPU form: `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (here N = 2 on
    one GPU, gloo for the timing collectives).  WORLD_SIZE must equal --gpus; rank 0 prints the one JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["WIPA_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--model", "tiny",
           "--batch", "8", "--pipeline", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    # a mismatch between the launcher's world size and --gpus is refused, not silently run
    base = cmd[: cmd.index(os.path.join(ROOT, "bench.py")) + 1]
    bad = subprocess.run(base + ["--gpus", "4", "--steps", "1", "--warmup", "1", "--model", "tiny", "--batch", "8", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE=2 but --gpus 4" in (bad.stderr + bad.stdout)

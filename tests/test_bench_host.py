"""bench.py's accounting helpers (no GPU): the algorithmic bytes behind `roofline.achieved` against SURVEY.md section 8d's
closed form, and the JSON contract's static fields."""
import json
import os
import subprocess
import sys
import types

import bench


def test_attention_bytes_match_survey_8d():
    cfg = types.SimpleNamespace(dim=1024, n_head=16, n_layer=24)          # GPT-L
    nbytes, launches = bench.attention_bytes_per_generate(cfg, 64, 576)    # config 2: 32 images, CFG -> 64 rows
    # kvb * B2 * [T + (N-1)(T+1) + (N-2)(N-1)/2], kvb = 98 304 B per token (24 layers x 2 x 16 x 64 x 2 B), T = 1
    kvb, N, T = 24 * 2 * 16 * 64 * 2, 576, 1
    assert nbytes == kvb * 64 * (T + (N - 1) * (T + 1) + (N - 2) * (N - 1) // 2) == 98304 * 64 * 166176
    assert launches == 576 * 24
    assert abs(nbytes / launches - 75.63e6) < 0.01e6                       # 75.6 MB per average launch
    twice, _ = bench.attention_bytes_per_generate(cfg, 128, 576)           # two batches per chain
    assert twice == 2 * nbytes


def test_constants_name_baseline_config_2():
    assert (bench.BATCH, bench.LAT, bench.IMG, bench.CFG, bench.TOPK) == (32, 24, 384, 4.0, 2000)
    assert bench.HBM_PEAK_GBS == 8000.0


def test_gpt_phase_bytes_match_survey_8d_config_2():
    cfg = types.SimpleNamespace(dim=1024, n_head=16, n_layer=24, vocab_size=16384, ffn_dim_multiplier=None, multiple_of=256)
    ph = bench.gpt_phase_bytes(cfg, 32, 576)
    assert abs(ph["weights_per_step"] - 650.2e6) < 0.5e6 and ph["kv_bytes_per_token"] == 98304      # SURVEY 8 model table
    assert ph["kv_reads"] == 98304 * 64 * 166176
    total = sum(ph[k] for k in ("weights", "kv_reads", "kv_writes", "logits", "noise"))
    assert abs(total - 1.426e12) < 0.005e12                                                          # "1.426 TB per 32 images"


def _run_bench(args, env_extra=None, timeout=180):
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_2_launches_its_own_ranks_on_gloo():
    """`python bench.py --gpus 2` with no launcher around it: the script re-runs itself under torch.distributed.run (one rank per
    GPU), rank 0 prints ONE JSON line; here with the CPU stand-in for the device step (gloo, world_size 2)."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--standin"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["parallelism"] == "dp2" and "STAND-IN" in j["data"] and j["value"] > 0


def test_bench_gpus_8_standin_eight_ranks_gather_in_the_reference_order():
    """The 8-GPU shape of BASELINE configs[2] without the hardware: `python bench.py --gpus 8 --standin` spawns EIGHT gloo ranks,
    every step ends in one gather to rank 0, rank 0 prints one line for the whole job (value = all ranks' images / max-over-ranks
    time)."""
    r = _run_bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--standin"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["rccl_ranks"] == 8 and j["config"]["parallelism"] == "dp8" and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 8 * 32 and j["value"] > 0


def test_hbm_budget_is_checked_before_allocation():
    """bench.py refuses a schedule whose KV slabs + noise exceed the per-GPU budget BEFORE anything is allocated: GPT-XXL
    (config 3) with 3 chains of 16 batches would need 3 x 16 x (10.6 GB of KV + 1.2 GB of noise)."""
    r = _run_bench(["--config", "3", "--lanes", "3", "--batches-per-chain", "16", "--steps", "48", "--budget-check-only"], timeout=300)
    assert r.returncode != 0 and "budget" in (r.stderr + r.stdout)
    r = _run_bench(["--config", "2", "--steps", "20", "--budget-check-only"], timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["batches_per_chain"] == 10 and j["chains_in_flight_per_gpu"] == 2 and j["kv_plus_noise_GB_per_chain"] < 60
    # config 3 per-GPU shape (GPT-XXL): two chains of six batches fit (2 x 73.5 GB), which is what the default plan picks
    r = _run_bench(["--config", "3", "--steps", "12", "--budget-check-only"], timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert (j["batches_per_chain"], j["chains_in_flight_per_gpu"]) == (6, 2) and 2 * j["kv_plus_noise_GB_per_chain"] < j["budget_GB"]


def test_bench_torchrun_form_and_world_mismatch():
    """The driver's torchrun form keeps working (WORLD_SIZE set -> no self-launch), and a WORLD_SIZE that disagrees with --gpus
    is refused."""
    port = bench._free_port()
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(os.path.dirname(bench.__file__), "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "0", "--standin"], env=env, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sum(l.startswith("{") for l in r.stdout.splitlines()) == 1
    r = _run_bench(["--gpus", "1", "--standin"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                                                            "MASTER_PORT": str(bench._free_port())}, timeout=60)
    assert r.returncode != 0


def test_bench_standin_single_rank():
    r = _run_bench(["--standin", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1


def test_config2_schedules_only_tested_chain_widths():
    """`bench.py` (config 2) only plans chain widths an end-to-end oracle test names -- 1..6 batches per chain (skinny kernels below
    256 rows, TILE_SCHEDULES[16] from there) or 10 (640 rows, TILE_SCHEDULES[40]) -- whatever --steps says; a wish of 7..9 falls
    back to 6 (round 4 planned up to 12 and refused the run AFTER the timed region)."""
    import argparse
    import bench
    got = {}
    for steps in (1, 2, 5, 8, 12, 14, 16, 18, 20, 24, 40):
        a = argparse.Namespace(config=2, steps=steps, batches_per_chain=0, lanes=0, no_one_chain=True)
        bpc, per_chain = bench.plan_schedule(a, 32, 576, 1)
        got[steps] = (bpc, a.lanes)
        assert bpc in (1, 2, 3, 4, 5, 6, 10) and per_chain > 0
    assert got[20] == (10, 2) and got[24] == (10, 2) and got[14] == (6, 2) and got[16] == (6, 2) and got[12] == (6, 2) and got[1] == (1, 1)
    # KV budget: GPT-3B's rows are 104 elements apart, not 128 (config 4); round 6: two chains of FOUR batches of 64 (512 rows)
    a = argparse.Namespace(config=4, steps=8, batches_per_chain=0, lanes=0, no_one_chain=True)
    bpc, per_chain = bench.plan_schedule(a, 64, 576, 1)
    assert bpc == 4 and a.lanes == 2 and abs(per_chain - (24 * 2 * 64 * 4 * 32 * 585 * 104 * 2 * 2 + 576 * 64 * 4 * 16384 * 4)) < 1e6
    a = argparse.Namespace(config=4, steps=4, batches_per_chain=0, lanes=0, no_one_chain=True)      # a short run keeps 2 x 2
    assert bench.plan_schedule(a, 64, 576, 1)[0] == 2 and a.lanes == 2
    # configs 3 / 5 (round 6): two chains of eight batches; the one-chain transparency leg is dropped when a third chain's slabs do not fit
    a = argparse.Namespace(config=3, steps=16, batches_per_chain=0, lanes=0, no_one_chain=False)
    assert bench.plan_schedule(a, 32, 576, 1)[0] == 8 and a.lanes == 2 and a.no_one_chain
    a = argparse.Namespace(config=3, steps=12, batches_per_chain=0, lanes=0, no_one_chain=False)
    assert bench.plan_schedule(a, 32, 576, 1)[0] == 6 and a.lanes == 2
    a = argparse.Namespace(config=5, steps=16, batches_per_chain=0, lanes=0, no_one_chain=True)
    assert bench.plan_schedule(a, 16, 1024, 120)[0] == 8 and a.lanes == 2
    a = argparse.Namespace(config=5, steps=24, batches_per_chain=0, lanes=0, no_one_chain=True)      # what other_configs() runs: 2 x 12 = 384 rows
    assert bench.plan_schedule(a, 16, 1024, 120)[0] == 12 and a.lanes == 2

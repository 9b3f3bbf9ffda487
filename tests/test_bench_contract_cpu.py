"""bench.py's arithmetic (no GPU): the algorithmic bytes / flops the roofline line is computed from must be the figures of
SURVEY.md 8(d) / DESIGN.md section 4, and the command line must keep the driver's contract (defaults N=1, quick K / W)."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def test_algorithmic_bytes_and_flops():
    # packed words + scales + packed zeros + x + y for one w4 g128 linear at M = 1
    assert bench.alg_bytes(4096, 4096, 1) == 4096 * 4096 // 2 + 32 * 4096 * 2 + 32 * 4096 // 2 + 2 * 4096 + 2 * 4096 == 8_732_672
    assert bench.alg_bytes(4096, 11008, 1) == bench.alg_bytes(11008, 4096, 1) == 23_455_232
    assert bench.bytes_per_token() == 32 * (4 * 8_732_672 + 3 * 23_455_232) == 3_369_484_288
    assert bench.bytes_per_token() // 128 == 26_324_096                                             # roofline.bytes_per_launch
    assert bench.flops_per_pass(4, 2048) == 4 * 2.0 * 2048 * (4 * 4096 * 4096 + 3 * 4096 * 11008)
    # fp16 zero points (HQQ) cost 2 bytes per group and column instead of half a byte
    assert bench.alg_bytes(4096, 4096, 16, 64, "f16") - bench.alg_bytes(4096, 4096, 16, 64) == 64 * 4096 * 2 - 64 * 4096 // 2
    assert bench.HBM_PEAK_GBPS == 8000.0 and bench.MFMA_PEAK_TFLOPS == 2500.0                       # MI355X_MICROARCH.md peaks


def test_command_line_contract():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert f'"{flag}"' in src
    assert 'add_argument("--gpus", type=int, default=1)' in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
                '"data"', '"config"', '"roofline"', '"cpu_baseline"', '"traffic"', '"frac"', '"workload"'):
        assert key in src, key
    # the oracle appears in bench.py only inside the cpu_baseline leg
    lines = [l for l in src.splitlines() if "from oracle" in l or "import oracle" in l]
    assert len(lines) == 1 and "ref_torch" in lines[0]


def test_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher in the environment must start two ranks itself (torch.distributed.run,
    127.0.0.1) and print ONE JSON line with n_gpus = 2.  The GPU work is stubbed (QLLM_BENCH_STUB=1: gloo, CPU) -- what runs here is
    the launch, rendezvous, max-over-ranks reduction and printing path the driver's 8-GPU run goes through."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["QLLM_BENCH_STUB"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and "STUB" in rec["data"]
    # ... and the tensor-parallel leg of BASELINE configs[4] rides on the same line (round-5 verdict item 2: the driver's scaling run
    # only passes --gpus N): same keys as the real leg's record, stub-marked
    tp = rec["extra"]["tp70b"]
    for key in ("tokens_per_s", "ms_per_step", "ranks_seen", "backend", "all_reduce_us_16KB", "oneshot_all_reduce_us_16KB",
                "row_parallel_sums", "ms_per_step_fused_all_reduce", "ms_per_step_unfused_all_reduce", "sharded_vs_unsharded", "tp_degree"):
        assert key in tp, key
    assert tp["ranks_seen"] == 2 and tp["backend"] == "gloo" and tp["tp_degree"] == 2 and "stub" in tp
    # a rank stuck in the leg's collectives must not cost the run its line: the watchdog prints the replica numbers with an error entry
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--tp-timeout-s", "4"],
                       env=dict(env, QLLM_BENCH_STUB_HANG="1"), capture_output=True, text=True, timeout=240)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and "timeout" in rec["extra"]["tp70b"]["error"]
    # under a launcher that set WORLD_SIZE the flag must agree with it
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="3", RANK="0"),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr

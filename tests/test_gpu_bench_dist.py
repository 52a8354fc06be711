"""bench.py's own N>1 code path under the driver's eyes: two ranks (fresh child processes of torch.distributed.run)
share the one GPU of the test box over the gloo backend -- shard_blocks, broadcast_keys / broadcast_tensor, per-rank
verification against AES-CTR and the MAX all-reduce of the elapsed time all run; only the transport differs from the
8-GPU RCCL run (which this pool does not let a builder launch).  The children initialise the GPU themselves; nothing
here re-execs a process that has touched it."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(nproc, extra, params="toy"):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(ROOT / "bench.py"), "--gpus", str(nproc), "--backend", "gloo", "--params", params] + extra
    res = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on rank 0, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_two_ranks_gloo():
    line = _run_bench(2, ["--blocks", "2", "--steps", "1", "--warmup", "0"])
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["scaling"] == "weak"
    assert line["verified_vs_aes"] is True
    assert line["config"]["total_blocks"] == 4
    assert line["unit"] == "blocks/s" and line["value"] > 0
    assert "cpu_baseline" in line and line["cpu_baseline"]["kind"] == "port"        # present for every world size
    assert line["roofline"]["bound"] == "valu_f64" and "hbm" in line["roofline"]
    # the kernel named is the one the context reports (fheaes_k2_context_plan), with its form
    assert "k2_form" in line["roofline"] and line["roofline"]["kernel"].startswith(("blind_rotate_pair_kernel", "blind_rotate16_kernel", "blind_rotate_latency_kernel"))
    # present (null unless a counter file of these very sources exists); round 3's `valu_busy` is gone: it was not a busy fraction
    for k in ("ceiling_frac", "valu_issue_occupancy_model", "sq_insts_valu", "clock_ghz", "model_frac", "traffic", "traffic_over_algorithmic",
              "l1_fill_bytes_per_clk_per_cu", "power"):
        assert k in line["roofline"]
    assert "valu_busy" not in line["roofline"]
    assert "key_broadcast_gloo" in line["setup_s"]
    assert line["ctr_iteration_with_add_scalar"] is None and line["configs4_decrypt_32_blocks"] is None      # N=1 extras only
    assert line["two_contexts_64_blocks_each"] is None
    # a SCALE line must verify itself: what the communicator says, how the keys travelled, every rank's own time
    col = line["collective"]
    assert col["world_size_seen"] == 2 and col["backend"] == "gloo" and col["data_path_collectives"] == 0
    assert col["key_broadcast_bytes"] > 0 and col["key_broadcast_GBps"] > 0
    assert len(col["rank_elapsed_s"]["per_rank"]) == 2 and col["rank_elapsed_s"]["min"] <= col["rank_elapsed_s"]["max"]
    assert abs(col["rank_elapsed_s"]["max"] * 1000.0 / line["steps"] - line["ms_per_step"]) < 1e-6 * line["ms_per_step"] + 1e-3


def test_bench_two_ranks_gloo_at_param_opt():
    """The N > 1 path at the REAL parameter set (main.rs:55-64 is what it stands for), rehearsed on the one GPU there is: two fresh child
    ranks share the card; rank 0 generates the PARAM_OPT keys, the seeded form (194,494,496 B) crosses the process group, every rank
    regenerates the masks on the GPU (fheaes_upload_keys_seeded), takes its shard of the blocks, and rank 0 verifies against AES.  The
    first real 8-GPU run must not be the first time real-size keys cross this path."""
    line = _run_bench(2, ["--blocks", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], params="opt")
    assert line["n_gpus"] == 2 and line["config"]["params"] == "PARAM_OPT"
    assert line["verified_vs_aes"] is True
    assert line["config"]["total_blocks"] == 16
    col = line["collective"]
    assert col["world_size_seen"] == 2 and col["backend"] == "gloo" and col["data_path_collectives"] == 0
    assert col["key_broadcast_bytes"] == 194_494_496
    assert len(col["rank_elapsed_s"]["per_rank"]) == 2 and min(col["rank_elapsed_s"]["per_rank"]) > 0
    # 8 blocks per rank = 1,024-bit launches: the paired kernel (form 2) on the MI355X
    assert line["roofline"]["k2_form"] == 2 and line["roofline"]["bits_per_launch"] == 1024


def test_bench_one_rank_through_rccl():
    """the only RCCL code a one-GPU box can run: bench.py --rccl-one-rank initialises the nccl backend with one rank and pushes the
    key broadcasts and the elapsed-time all-reduce through it (communicator creation + collectives on the MI355X; no xGMI transfer)"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--params", "toy", "--blocks", "2", "--steps", "1", "--warmup", "0",
                          "--rccl-one-rank", "--no-cpu-baseline", "--no-ctr-iteration"], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    # The only acceptable skip: RCCL could not CREATE a communicator on this box (bench.py prints RCCL_COMMUNICATOR_READY to stderr
    # once init_process_group and the probe all-reduce have succeeded).  The skip reason carries the whole stderr tail.  Any error
    # after the marker -- the seeded-key broadcasts, the elapsed-time all-reduce, destroy_process_group -- fails the test.
    # (Seen once, on one box of the pool in round 2: ncclSystemError from communicator creation)
    created = "RCCL_COMMUNICATOR_READY" in res.stderr
    if res.returncode != 0 and not created and any(k in res.stderr for k in ("ncclSystemError", "ncclUnhandledCudaError", "ncclInternalError", "NCCL error")):
        pytest.skip("RCCL could not create a communicator on this box (no collective ran); stderr tail:\n" + res.stderr[-3000:])
    assert res.returncode == 0, ("RCCL communicator was created, then: " if created else "") + res.stdout[-2000:] + res.stderr[-4000:]
    assert created
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    r = line["rccl_one_rank"]
    assert r["ranks"] == 1 and r["key_broadcast_intact"] is True and r["elapsed_all_reduce_max_ok"] is True
    assert line["n_gpus"] == 1 and line["verified_vs_aes"] is True


def test_bench_single_rank_line_has_the_extra_verified_steps():
    """the default N=1 line also reports the reference's whole CTR iteration and the configs[4] decrypt shard (32 blocks), both
    verified against AES (toy parameters here: the shapes and the verification, not the numbers)"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--params", "toy", "--blocks", "32", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                         cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["verified_vs_aes"] is True
    ctr, dec = line["ctr_iteration_with_add_scalar"], line["configs4_decrypt_32_blocks"]
    assert ctr["verified_vs_aes"] is True and ctr["blocks_per_s"] > 0
    assert dec["verified_vs_aes"] is True and dec["blocks_per_s"] > 0 and dec["k2_launches"] == 19 and dec["k2_bits_per_launch"] == 32 * 128
    two = line["two_contexts_64_blocks_each"]
    assert two["verified_vs_aes"] is True and two["blocks_per_context"] == 16 and two["clone"]["path"] == "same_device" and two["clone"]["bytes"] > 0
    assert "collective" not in line                                        # N = 1: nothing is broadcast between ranks
    # the diagnostic identity of the roofline object: when a counter file of these sources is attached, ceiling x occupancy x clock / 2.4
    # reproduces the profiled launch's own fraction (toy parameters: the fields are present and null)
    r = line["roofline"]
    if r["model_frac"] is not None:
        assert abs(r["model_frac"] - r["frac_of_profiled_launch"]) <= 0.03 * r["frac_of_profiled_launch"]


def test_bench_driver_shape_at_param_opt_verifies_every_block():
    """the driver's command at reduced cost (PARAM_OPT, 32 blocks, 6 + 2 steps, no CPU baseline): every block of the headline and of
    every extra step is decrypted and compared, the line says so (`all_verified`, `wrong_blocks`), and a wrong block is a non-zero exit"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--params", "opt", "--blocks", "32", "--steps", "6", "--warmup", "2", "--no-cpu-baseline"],
                         cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["all_verified"] is True and line["wrong_blocks"] == {} and line["blocks_checked"] == 32
    assert line["ctr_iteration_with_add_scalar"]["blocks_checked"] == 32 and line["configs4_decrypt_32_blocks"]["blocks_checked"] == 32
    assert line["configs4_decrypt_32_blocks"]["same_words_as_warmup_run"] is True
    assert line["two_contexts_64_blocks_each"]["blocks_checked"] == 32 and line["config1_one_block_round"]["verified_vs_aes"] is True
    assert line["roofline"]["kernel"].startswith("blind_rotate_pair_kernel") and "parking=claimed" in line["roofline"]["kernel"]


def test_bench_exit_code_follows_verification(tmp_path):
    """a bench that cannot fail is no check: with the expected plaintexts made wrong on purpose (FHEAES_BENCH_SABOTAGE_VERIFY=1 flips one
    bit of what block 0 is compared with) the line still prints, says all_verified false with the block listed, and the exit code is 1"""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    env["FHEAES_BENCH_SABOTAGE_VERIFY"] = "1"
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--params", "toy", "--blocks", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          "--no-ctr-iteration"], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 1, res.stdout[-2000:] + res.stderr[-4000:]
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])
    assert line["all_verified"] is False and line["verified_vs_aes"] is False and line["wrong_blocks"] == {"headline_rank0": [0]}


def test_bench_refuses_world_size_mismatch():
    """--gpus must equal WORLD_SIZE: a silent single-rank run of a "2 GPU" bench would be an invalid number"""
    res = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--params", "toy", "--blocks", "1"], cwd=str(ROOT),
                         capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in (res.stderr + res.stdout)

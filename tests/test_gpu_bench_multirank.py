"""-m gpu: `bench.py --gpus N` as the driver would run it on a multi-GPU node — here with N rank PROCESSES sharing the
one reachable GPU over the stand-in for librccl (tests/harness/fake_rccl.cpp, POSIX shared memory; handed to the library
through CTMR_RCCL_LIB).  What is under test is everything around the collectives that only exists at N > 1: the
self-launch (the parent makes the group id and spawns the ranks), the launcher path (torchrun's RANK / WORLD_SIZE and
one broadcast of the id), the log-index split of ONE batch (strong scaling, BASELINE configs[3]), exact global dedup by
default, and the run's own checks at every N — Σ NEW and every entry's WasUnknown against the generator, the
all-reduced per-issuer counts, the strided oracle sample of every rank's shard.  Real RCCL between devices stays
unmeasured until a multi-GPU node exists."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, launcher=None, timeout=300):
    from tests.harness import build_fake_rccl
    env = dict(os.environ, CTMR_RCCL_LIB=build_fake_rccl())
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CTMR_GROUP_ID"):
        env.pop(k, None)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + extra
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 1, (p.returncode, p.stdout[-500:], p.stderr[-3000:])
    return json.loads(lines[0])


COMMON = ["--steps", "2", "--warmup", "1", "--traffic", "off", "--cpu-sample", "60000", "--sample-slices", "12",
          "--dup-permille", "100"]


@pytest.mark.parametrize("world,dedup", [(2, "auto"), (3, "owner"), (4, "bloom")])
def test_self_launched_ranks_split_one_batch_and_dedup_globally(world, dedup):
    d = run_bench(["--gpus", str(world), "--total-entries", "300000", "--dedup", dedup] + COMMON)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["config"]["total_entries"] == 300000
    assert d["config"]["dedup"] == ("bloom" if dedup == "auto" else dedup)
    c = d["checks"]
    assert c["entries_disagreeing_with_generator"] == 0 and c["n_new_all_ranks"] == c["n_new_expected_from_generator"] > 0
    assert c["per_issuer_counts_match_generator"] and c["issuer_counts_all_ranks_sum"] == c["n_new_all_ranks"]
    assert d["parity_vs_oracle_on_sample"] is True
    ps = d["parity_sample"]
    assert ps["ranks"] == world and ps["entries_checked_all_ranks"] >= 50000 and ps["known_duplicates_all_ranks"] > 1000
    assert ps["sources_outside_the_slices"] > 0                  # duplicates whose first copy lies elsewhere in the log
    ex = d["exchange"]
    assert ex["transport"] == "rccl" and ex["wire_bytes_sent_by_rank0_per_step"] > 0
    assert "cpu_baseline" not in d                               # a one-GPU leg
    assert d["value"] > 0 and abs(d["value"] - 300000 * 1000 / d["ms_per_step"]) < 1e-3 * d["value"]


def test_owner_rounds_in_chunks_over_rank_processes():
    """bench.py --gpus 3 --dedup owner --chunks 4: the shards are mapped in four pieces, each piece's key records handed to
    the transport before the next is walked — the line's checks are those of the unchunked run."""
    d = run_bench(["--gpus", "3", "--total-entries", "300000", "--dedup", "owner", "--chunks", "4"] + COMMON)
    c = d["checks"]
    assert c["entries_disagreeing_with_generator"] == 0 and c["n_new_all_ranks"] == c["n_new_expected_from_generator"] > 0
    assert c["per_issuer_counts_match_generator"] and d["parity_vs_oracle_on_sample"] is True
    assert d["exchange"]["wire_bytes_sent_by_rank0_per_step"] > 0


def test_per_shard_sets_are_an_explicit_option_and_say_what_they_are():
    d = run_bench(["--gpus", "2", "--total-entries", "200000", "--dedup", "local"] + COMMON)
    assert d["config"]["dedup"] == "local" and "PER-SHARD" in d["config"]["parallelism"]
    c = d["checks"]
    assert c["entries_disagreeing_with_generator"] > 0           # duplicates across the two shards are counted twice
    assert c["n_new_all_ranks"] > c["n_new_expected_from_generator"]
    assert d["parity_vs_oracle_on_sample"] is False


def test_weak_scaling_and_the_launcher_path():
    """Under a launcher (RANK / WORLD_SIZE / MASTER_* in the environment, as the driver starts N > 1) the id travels by
    one broadcast; --entries gives every GPU its own E entries."""
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    d = run_bench(["--gpus", "2", "--entries", "100000"] + COMMON, launcher=launcher)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["total_entries"] == 200000
    assert d["checks"]["entries_disagreeing_with_generator"] == 0 and d["parity_vs_oracle_on_sample"] is True


def test_one_gpu_line_still_carries_its_checks():
    d = run_bench(["--gpus", "1", "--total-entries", "300000", "--no-secondary"] + COMMON)
    assert d["n_gpus"] == 1 and d["config"]["dedup"] == "plain" and "exchange" not in d
    assert d["checks"]["entries_disagreeing_with_generator"] == 0 and d["checks"]["per_issuer_counts_match_generator"]
    assert d["parity_vs_oracle_on_sample"] is True and d["cpu_baseline"]["value"] > 0


@pytest.mark.parametrize("dedup", ["auto", "bloom"])
def test_stream_over_several_ranks_keeps_its_sets_across_waves(dedup):
    """BASELINE configs[4] in its multi-GPU form: a stream with 10 % duplicates in waves, every wave split by log index
    over the ranks, the known-certificate sets persisting on the ranks; every wave's NEW / known counts and every entry's
    WasUnknown equal the generator's structure, the ranks' sets add up to Σ NEW."""
    d = run_bench(["--gpus", "3", "--stream", "400000", "--entries", "100000", "--dedup", dedup, "--traffic", "off"])
    assert d["n_gpus"] == 3 and d["steps"] == 4 and d["config"]["dedup"] == ("owner" if dedup == "auto" else "bloom")
    r = d["result"]
    assert r["duplicate_structure_matches_generator_in_every_wave"] is True and r["entries_disagreeing_with_generator"] == 0
    assert r["total_count"] == r["n_new"] > 300000 and r["n_dup"] > 20000
    assert d["exchange"]["wire_bytes_sent_by_rank0_over_the_stream"] > 0

"""bench.py keeps the driver's contract: flags, ONE JSON line, the keys and objects the judge reads."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}


def test_algorithmic_flops_formula():
    sys.path.insert(0, str(ROOT))
    import bench

    # S*70 + segments*(22*N_s + 16*N_p + 60)
    assert bench.algorithmic_flops(10, 16, 3, 0) == 10 * 70 + 16 * (66 + 60)
    assert bench.algorithmic_flops(1, 1, 100000, 2) == 70 + (2200000 + 32 + 60)


def test_issue_by_class_prices_every_instruction_once():
    sys.path.insert(0, str(ROOT))
    import bench

    counters = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_VALU_FMA_F32": 400.0, "SQ_INSTS_VALU_MUL_F32": 100.0, "SQ_INSTS_VALU_ADD_F32": 100.0, "SQ_INSTS_VALU_TRANS_F32": 20.0,
                "SQ_INSTS_VALU_CVT": 10.0, "SQ_INSTS_VALU_INT32": 150.0, "SQ_INSTS_VALU_INT64": 20.0}
    got = bench.issue_by_class(counters, 1.0, 1024)
    assert abs(sum(got["share_of_valu_insts"].values()) - 1.0) < 1e-3 and got["share_of_valu_insts"]["other"] == 0.2
    cost = bench.VALU_CLASS_CYCLES
    cycles = 600 * cost["fma_f32"] + 20 * cost["trans_f32"] + 10 * cost["cvt"] + 150 * cost["int32"] + 20 * cost["int64"] + 200 * cost["other"]
    assert abs(got["issue_time_ms"] - cycles / (1024 * bench.SHADER_CLOCK_HZ) * 1e3) < 1e-4  # (the line rounds to 0.1 us)
    assert bench.issue_by_class({"SQ_INSTS_VALU": 1000.0}, 1.0, 1024) is None  # an entry without the class passes: no figure
    # the committed figures: every entry that has the class counters prices to no more than its kernel's own time — within the 2-3 %
    # the per-class prices are measured to (profiles/r04/valu_issue_costs.txt): the 64-sphere scan of round 5 prices to 1.008 of its time
    table = json.loads((ROOT / "profiles" / "pmc_counters.json").read_text())
    priced = 0
    for key, rec in table.items():
        by_class = bench.issue_by_class(rec["counters_per_launch"], rec["kernel_ms_traced_mean_of_timed_steps"], 1024)
        if by_class:
            priced += 1
            assert 0.5 < by_class["frac_of_kernel_time"] <= 1.03, (key, by_class)
    assert priced >= 2


def test_cpu_baseline_counts_the_cpus_the_container_is_granted(tmp_path, monkeypatch):
    """`cpu_baseline.cores` is what the timed leg could really use: a GPU box shows 256 CPUs to a process whose cgroup grants
    it the time of 16 (cpu.max "1600000 100000") — 256 threads there take turns on 16 CPUs' worth of time."""
    sys.path.insert(0, str(ROOT))
    import bench

    monkeypatch.setattr(bench, "ROOT_CGROUP", tmp_path)
    assert bench.cpu_quota() is None  # no controller files at all
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert bench.cpu_quota() == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert bench.cpu_quota() is None
    (tmp_path / "cpu.max").unlink()
    (tmp_path / "cpu").mkdir()
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (tmp_path / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert bench.cpu_quota() == 2.5
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert bench.cpu_quota() is None
    # the leg itself, with the time of two CPUs granted: two threads, and the line says why
    (tmp_path / "cpu" / "cpu.cfs_quota_us").write_text("200000\n")
    if len(os.sched_getaffinity(0)) > 2:
        leg = bench.cpu_baseline("basic", 64, 36, 0.2)
        assert leg["cores"] == 2 and "grants the CPU time of 2" in leg["sample"] and leg["kind"] == "port" and leg["value"] > 0


def run_bench(args, launcher=None, timeout=600, expect_rc=0, all_lines=False):
    cmd = (launcher or [sys.executable]) + [str(ROOT / "bench.py")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert (out.returncode == 0) == (expect_rc == 0), (out.returncode, out.stderr[-2000:])
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    # one line — or, under torchrun with several forms, the first form's line as soon as it is done and a final one
    assert len(lines) == 1 or (len(lines) == 2 and lines[0].get("line", "").startswith("early") and lines[1].get("line") == "final"), out.stdout
    if all_lines:
        return lines
    return lines[-1]


@pytest.mark.parametrize("ranks", [1, 3])
def test_torchrun_flow_starts_one_child_per_form_and_rank_without_touching_a_gpu(ranks):
    """The N > 1 flow of bench.py on the CPU (RT_BENCH_DRY_RUN): the torchrun-launched processes start one child per form and
    rank, the children of a form meet on a port of their own, rank 0 collects the verdicts — and, nothing having been measured,
    prints a line without a value and exits non-zero."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"),
           "--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--library-deadline-s", "120"]
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=dict(os.environ, RT_BENCH_DRY_RUN="1"))
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.returncode != 0, (out.returncode, out.stdout, out.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["value"] is None and line["n_gpus"] == ranks
    assert set(line["paths"]) == {"library", "shared_frame", "torch"}
    assert all(f"dry run: {ranks} rank(s) met" in line["paths"][form]["status"] for form in line["paths"])
    assert len({line["paths"][form]["status"] for form in line["paths"]}) == 3  # three rendezvous, three ports


@pytest.mark.gpu
def test_single_gpu_line_has_the_contract_keys_roofline_and_cpu_baseline():
    line = run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--spp", "32", "--cpu-baseline-seconds", "1"])
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["unit"] == "Mrays/s" and line["higher_is_better"] is True and line["scaling"] == "strong" and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and "workload" in line["config"] and "model" not in line["config"]
    assert line["value"] == pytest.approx(1920 * 1080 * 32 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    roof = line["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=1e-3) and 0 < roof["frac"] < 1
    cpu = line["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cpu) and cpu["kind"] == "port" and cpu["cores"] >= 1
    # `value` is the drop-in call (host scene in, host frame out, one frame at a time); the kernel-only rate rides along
    assert line["config"]["frames_in_flight"] == 1 and "blocking render()" in line["config"]["step"]
    assert line["drop_in_breakdown"]["wall_ms"] == pytest.approx(line["ms_per_step"])
    assert line["drop_in_breakdown"]["kernel_ms"] <= line["ms_per_step"]
    assert line["kernel_only"]["value"] >= 0.95 * line["value"]
    # the call as the plug-in makes it (stats == NULL) is never slower than the timed one, which keeps the events
    assert line["plug_in_call"]["ms_per_step"] <= line["ms_per_step"] * 1.02
    assert {"render_ms", "host_issue_ms", "host_wait_ms"} <= set(line["drop_in_breakdown"])
    assert "scenes/basic.toml" in line["data"]
    assert "EPYC" in cpu["sample"] or "CPU" in cpu["sample"] or "Xeon" in cpu["sample"]


@pytest.mark.gpu
def test_one_process_drives_several_members_through_the_c_abi():
    """`bench.py --gpus N` outside torchrun = ONE process, rt_hip_create_multi + rt_hip_render: the form the reference's
    blocking render() can use.  On this one-GPU box the N members share device 0 (--same-device: peer copies)."""
    line = run_bench(["--gpus", "4", "--same-device", "--steps", "2", "--warmup", "1", "--spp", "16", "--cpu-baseline-seconds", "0"])
    assert line["n_gpus"] == 4 and "ONE process" in line["config"]["parallelism"] and "cpu_baseline" not in line
    assert line["value"] == pytest.approx(1920 * 1080 * 16 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    # what makes the first real multi-GPU run readable: who took part, every member's own kernel time, and
    # where the root's time went
    assert line["rccl"]["ranks"] == 4 and line["rccl"]["devices"] == [0, 0, 0, 0] and line["rccl"]["transport"] == "peer_copy"
    per_rank = line["per_rank"]
    assert len(per_rank["kernel_ms"]) == 4 and 0 < per_rank["kernel_ms_min"] <= per_rank["kernel_ms_max"] < line["ms_per_step"]
    split = line["drop_in_breakdown"]
    assert {"render_ms", "gather_ms", "assemble_ms", "copy_ms", "host_issue_ms", "host_wait_ms", "wall_ms"} <= set(split)
    assert split["render_ms"] > 0 and split["assemble_ms"] > 0 and split["copy_ms"] >= 0
    assert split["render_ms"] + split["gather_ms"] + split["assemble_ms"] + split["copy_ms"] <= split["wall_ms"] * 1.05


@pytest.mark.gpu
def test_direct_frame_line_says_that_nothing_was_exchanged():
    line = run_bench(["--gpus", "2", "--same-device", "--direct-frame", "--steps", "2", "--warmup", "1", "--spp", "16", "--cpu-baseline-seconds", "0"])
    assert line["rccl"]["transport"] == "direct_frame" and line["drop_in_breakdown"]["assemble_ms"] == 0


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_rehearse_the_multi_gpu_step():
    """The N>1 flow of bench.py — per-rank stripes, gather to rank 0, device assemble, max-over-ranks timing — with
    both ranks sharing the single GPU of this box and gloo as the transport (RCCL refuses two ranks on one device)."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    line = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "16", "--backend", "gloo", "--gather", "torch", "--cpu-baseline-seconds", "0"], launcher=launcher)
    assert line["n_gpus"] == 2 and "cpu_baseline" not in line
    assert "one process per GPU" in line["config"]["parallelism"] and line["config"]["frames_in_flight"] == 1
    assert line["value_from"] == "torch" and set(line["paths"]) == {"torch"}
    assert len(line["per_rank"]["kernel_ms"]) == 2 and line["paths"]["torch"]["ms_per_step"] == pytest.approx(line["ms_per_step"])


@pytest.mark.gpu
def test_one_rank_under_torchrun_reports_the_rccl_gather_form():
    """What the driver launches for N > 1, with N = 1: torch.distributed.run; the torchrun-launched process starts one child per
    form (it never touches the GPU itself); `value` is the form north_star names — rt_hip_create + rt_hip_join_ranks
    (ncclCommInitRank), the collective rt_hip_render with its single ncclGather — and the other two are side keys."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    line = run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--spp", "32", "--cpu-baseline-seconds", "0"], launcher=launcher)
    assert line["n_gpus"] == 1 and "ncclCommInitRank" in line["config"]["parallelism"] and "ncclGather" in line["config"]["parallelism"]
    assert line["value"] == pytest.approx(1920 * 1080 * 32 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert 0 < line["roofline"]["frac"] < 1
    # `value` IS the RCCL-gather form; the frame group and the torch form ride along with their own figures
    assert line["value_from"] == "library"
    assert all(line["paths"][form]["status"] == "ok" and line["paths"][form]["ms_per_step"] > 0 for form in ("library", "shared_frame", "torch"))
    assert line["paths"]["library"]["ms_per_step"] == pytest.approx(line["ms_per_step"])
    # what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), every rank's kernel
    # time, the root's split of a step
    assert line["rccl"] == {"ranks": 1, "devices": [0], "rank_of_process": [0], "transport": "rccl_gather", "source": line["rccl"]["source"]} and "ncclCommCount" in line["rccl"]["source"]
    assert len(line["per_rank"]["kernel_ms"]) == 1 and 0 < line["per_rank"]["kernel_ms_max"] <= line["ms_per_step"]
    split = line["drop_in_breakdown"]
    assert {"render_ms", "gather_ms", "assemble_ms", "copy_ms", "host_issue_ms", "host_wait_ms", "wall_ms"} <= set(split)
    assert split["render_ms"] > 0 and split["wall_ms"] == pytest.approx(line["ms_per_step"])


@pytest.mark.gpu
def test_a_form_that_hangs_is_killed_reported_and_fails_the_run():
    """VERDICT r3 #3 / ADVICE: a collective that never returns must not read as success.  RT_BENCH_TEST_HANG stalls the RCCL-gather
    form inside its first frame; its child processes are killed at the deadline (by pid), the other forms still run in fresh
    processes, rank 0 prints the line — `value` from the next form, `paths.library.status` = hung — and the benchmark exits
    NON-ZERO; nothing is left behind in /dev/shm."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    os.environ["RT_BENCH_TEST_HANG"] = "library"
    try:
        line = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "16", "--cpu-baseline-seconds", "0", "--library-deadline-s", "45"], launcher=launcher, expect_rc=3)
    finally:
        del os.environ["RT_BENCH_TEST_HANG"]
    assert "hung" in line["paths"]["library"]["status"] and "killed" in line["paths"]["library"]["status"]
    assert line["value_from"] == "shared_frame" and line["paths"]["shared_frame"]["status"] == "ok" and line["paths"]["torch"]["status"] == "ok"
    assert line["value"] == pytest.approx(1920 * 1080 * 16 / (line["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert not [name for name in os.listdir("/dev/shm") if name.startswith("rt_hip_bench_")]


@pytest.mark.gpu
def test_the_first_forms_line_is_printed_early_and_the_side_forms_keep_to_the_budget():
    """the first N > 1 run must fit the driver's time limit whatever the side forms do.  The form `value` comes from
    runs first and its line is printed (and flushed) as soon as it is done; here the SECOND form hangs, is killed at its own
    (shorter) deadline, and the third no longer fits the budget and is skipped.  The final line repeats the first form's figures
    with all three verdicts; the run exits non-zero because a form hung."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    os.environ["RT_BENCH_TEST_HANG"] = "shared_frame"
    try:
        early, final = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "16", "--cpu-baseline-seconds", "0", "--library-deadline-s", "40", "--forms-budget-s", "70"],
                                 launcher=launcher, expect_rc=3, all_lines=True)
    finally:
        del os.environ["RT_BENCH_TEST_HANG"]
    assert early["line"].startswith("early") and early["value_from"] == "library" and early["value"] > 0 and REQUIRED <= set(early)
    assert set(early["paths"]) == {"library"} and early["paths"]["library"]["status"] == "ok"
    assert final["line"] == "final" and final["early_line_from"] == "library" and final["value"] == early["value"] and final["ms_per_step"] == early["ms_per_step"]
    assert "hung" in final["paths"]["shared_frame"]["status"] and "40 s" in final["paths"]["shared_frame"]["status"]
    assert final["paths"]["torch"]["status"].startswith("skipped: budget")
    assert not [name for name in os.listdir("/dev/shm") if name.startswith("rt_hip_bench_")]


@pytest.mark.gpu
def test_four_processes_on_one_device_rehearse_the_shared_frame_form():
    """`torchrun --nproc-per-node 4 bench.py --gpus 4 --backend gloo`: four rank processes on the box's one GPU.  The torch
    form stages the stripes through host memory (gloo); the frame group needs no RCCL and runs as it would on four GPUs —
    same protocol, same stores into one shared back buffer, validated against the frame rank 0 renders alone."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    line = run_bench(["--gpus", "4", "--steps", "3", "--warmup", "1", "--spp", "32", "--backend", "gloo", "--cpu-baseline-seconds", "0"], launcher=launcher)
    assert line["n_gpus"] == 4 and line["value_from"] == "shared_frame" and line["paths"]["shared_frame"]["status"] == "ok"  # (no RCCL between ranks that share a device)
    assert "gloo" in line["paths"]["library"]["status"] and line["paths"]["torch"]["value"] > 0
    assert line["rccl"] == {"ranks": 4, "devices": [0, 0, 0, 0], "rank_of_process": [0, 1, 2, 3], "transport": "shared_frame", "source": line["rccl"]["source"]}
    assert len(line["per_rank"]["kernel_ms"]) == 4 and "frame_buffer" in line["config"]
    split = line["drop_in_breakdown"]
    assert split["render_ms"] > 0 and split["assemble_ms"] == 0 and split["copy_ms"] == 0  # nothing is exchanged, nothing copied

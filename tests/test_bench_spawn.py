"""`python bench.py --gpus N` started the plain way (no launcher): bench.py must start its N ranks itself, as fresh child
processes under torch.distributed.run, relay rank 0's one JSON line and pass a failure on as a non-zero exit code -- the first
8-GPU node the driver gets must not be lost to a launcher detail.  Runs on CPU: the ranks here are a stub script."""
import io
import json
import os
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def _stub(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_spawn_command_is_one_process_per_gpu_on_localhost():
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "2"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "2"]     # the caller's flags reach every rank unchanged


def test_ranks_are_spawned_and_rank_zero_line_is_relayed(tmp_path):
    stub = _stub(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and "LOCAL_RANK" in os.environ
        print("chatter from rank", rank)                       # not a result line: must not reach the parent's stdout
        if rank == 0:
            print(json.dumps({"metric": "stub", "n_gpus": world, "argv": sys.argv[1:]}))
    """)
    out = io.StringIO()
    rc = bench.spawn_ranks(2, ["--gpus", "2", "--steps", "3"], script=stub, out=out)
    assert rc == 0
    lines = out.getvalue().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["argv"] == ["--gpus", "2", "--steps", "3"]


def test_a_failing_rank_gives_a_nonzero_exit_and_no_line(tmp_path):
    stub = _stub(tmp_path, """
        import os, sys
        sys.exit(3 if os.environ["RANK"] == "1" else 0)
    """)
    out = io.StringIO()
    assert bench.spawn_ranks(2, [], script=stub, out=out) != 0
    assert out.getvalue() == ""


def test_ranks_that_print_nothing_are_an_error(tmp_path):
    stub = _stub(tmp_path, "print('no result here')\n")
    out = io.StringIO()
    assert bench.spawn_ranks(2, [], script=stub, out=out) == 1
    assert out.getvalue() == ""


def test_inside_a_launcher_bench_does_not_spawn_again(monkeypatch):
    # WORLD_SIZE present = we ARE a rank: main() must go on to the GPU path (which refuses without a GPU), never spawn
    called = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda *a, **k: called.append(a) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2"); monkeypatch.setenv("RANK", "0"); monkeypatch.setenv("LOCAL_RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    import torch
    if torch.cuda.is_available():
        return
    saved = os.dup(1)
    try:
        try:
            bench.main()
        except SystemExit as e:
            assert "MI355X" in str(e.code)
    finally:
        os.dup2(saved, 1); os.close(saved)
    assert not called


def test_world_size_that_contradicts_gpus_is_refused(monkeypatch):
    # an outer harness exported WORLD_SIZE=1; `--gpus 8` must neither spawn nor silently run on one GPU
    called = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda *a, **k: called.append(a) or 0)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "1"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "WORLD_SIZE=1" in str(e.value.code) and "--gpus 8" in str(e.value.code)
    assert not called

"""Multi-process CPU coverage of the N>1 path (gloo, world_size 2): the static unit partition and the
single weight-blob broadcast that the multi-GPU bench performs over RCCL."""
import os

import pytest
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from render_in_between_amd import distributed as ribdist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, root, q):
    """The product's multi-GPU path on CPU: Evaluator.evaluate_from_folder under a world-2 process group deals the
    segments of all clips to the ranks; the generator is the CPU oracle behind the reference's call protocol."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ribdist.init_process_group("gloo")
    import render_in_between_amd as rib
    from render_in_between_amd import evaluator as ev, synth
    from oracle import generator_ref
    from tests.test_driver import MID_CFG, oracle_labels
    torch.set_num_threads(2)
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    R = generator_ref.RefGenerator(spec, synth.make_state_dict(spec, 2))

    class Model:
        def eval(self):
            return self

        def __call__(self, label, label_prev, dain, prev):
            return R(label, label_prev, dain, prev)

    E = ev.Evaluator(cfg, label_fn=oracle_labels)
    written = E.evaluate_from_folder(Model(), os.path.join(root, "inputs"), os.path.join(root, "DAIN"),
                                     os.path.join(root, "Predict_motion"), os.path.join(root, "sharded"))
    rows = ribdist.gather_rows(torch.tensor([float(len(written)), float(rank)]), rank, world)
    q.put((rank, [os.path.basename(w) for w in written], rows.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ribdist.init_process_group("gloo")
    n = 1 << 16
    if rank == 0:
        g = torch.Generator().manual_seed(5)
        blob = torch.randn(n, generator=g)
    else:
        blob = torch.zeros(n)
    ribdist.broadcast_blob(blob, src=0)
    units = ribdist.shard_units(11, rank, world)
    # per-rank timing reduction as bench.py does it (MAX over ranks)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ribdist.blob_checksum(blob), units, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, u0, t0), (r1, c1, u1, t1) = res
    assert c0 == c1 != 0                                   # identical blob on every rank after ONE broadcast
    assert sorted(u0 + u1) == list(range(11)) and not set(u0) & set(u1)
    assert u0 == [0, 2, 4, 6, 8, 10] and u1 == [1, 3, 5, 7, 9]
    assert t0 == t1 == 2.0


def test_shard_units_edge_cases():
    assert ribdist.shard_units(0, 0, 4) == []
    assert ribdist.shard_units(3, 2, 8) == [2] and ribdist.shard_units(3, 3, 8) == [] and ribdist.shard_units(3, 5, 8) == []
    assert sum(len(ribdist.shard_units(32, r, 8)) for r in range(8)) == 32


def test_evaluator_shards_segments_over_ranks(tmp_path):
    """world 2 (gloo): every frame is written exactly once, by the rank that owns its segment, and the frames equal
    a single-process run's bit for bit (segments are independent: PGNR/models/evaluator.py:240-244)."""
    import numpy as np
    from PIL import Image
    import render_in_between_amd as rib
    from render_in_between_amd import evaluator as ev, synth
    from oracle import generator_ref
    from tests.test_driver import MID_CFG, _write_example, oracle_labels
    root = str(tmp_path)
    n = _write_example(root, n_key=4, rate=2, H=32, W=48)            # 7 frames: 4 key frames, 3 one-frame segments
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, root, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, rows0), (_, w1, rows1) = res
    assert rows0 == rows1 == [[float(len(w0)), 0.0], [float(len(w1)), 1.0]]     # gather_rows: every rank sees every row
    assert sorted(w0 + w1) == ["f%03d.png" % i for i in range(n)] and not set(w0) & set(w1)
    # units: key 0 + its segment -> rank 0, key 2 + segment -> rank 1, key 4 + segment -> rank 0, key 6 (no segment) -> rank 1
    assert w0 == ["f000.png", "f001.png", "f004.png", "f005.png"] and w1 == ["f002.png", "f003.png", "f006.png"]
    # single-process run of the same folder
    cfg = rib.AttrDict(gen=rib.hsm_gen_config(**MID_CFG), model_height=32, model_width=48, gauss_sigma=5,
                       skeleton_thres=0.001, foot_thres=0.001)
    spec = rib.GenSpec.from_cfg(cfg.gen)
    R = generator_ref.RefGenerator(spec, synth.make_state_dict(spec, 2))

    class Model:
        def eval(self):
            return self

        def __call__(self, *a):
            return R(*a)

    one = ev.Evaluator(cfg, label_fn=oracle_labels).evaluate_from_folder(
        Model(), os.path.join(root, "inputs"), os.path.join(root, "DAIN"), os.path.join(root, "Predict_motion"),
        os.path.join(root, "single"), rank=0, world=1)
    assert len(one) == n
    for f in one:
        a = np.asarray(Image.open(f))
        b = np.asarray(Image.open(f.replace(os.sep + "single" + os.sep, os.sep + "sharded" + os.sep)))
        assert np.array_equal(a, b), f


def test_self_launch_command_and_rank_detection(monkeypatch):
    cmd = ribdist.launch_command("/x/bench.py", ["--gpus", "8", "--mode", "clips"], 8, port=29999)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--mode", "clips"]
    # without a port the launcher's own store binds a free one (no pick-then-bind window), still on 127.0.0.1
    cmd = ribdist.launch_command("/x/bench.py", ["--gpus", "8", "--mode", "clips"], 8)
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and "--master-port" not in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--mode", "clips"]
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    assert not ribdist.is_rank_process()
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "3")
    assert ribdist.is_rank_process()
    assert 1024 < ribdist.free_port() < 65536
    # world 1: gather_rows is the identity with a leading axis, no process group needed
    t = torch.arange(6, dtype=torch.int64).reshape(2, 3)
    assert torch.equal(ribdist.gather_rows(t, 0, 1), t[None])


def test_bench_starts_its_own_ranks_before_touching_the_gpu(monkeypatch):
    """`python bench.py --gpus N` outside torch.distributed.run must spawn the ranks as children (and never build a
    Generator in the parent); under torch.distributed.run it must not spawn."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("rib_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    calls = []
    monkeypatch.setattr(ribdist, "self_launch", lambda script, argv, n: calls.append((script, list(argv), n)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("RANK", raising=False)
    monkeypatch.setattr("sys.argv", ["bench.py", "--gpus", "4", "--mode", "clips", "--steps", "2"])
    import pytest
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [(os.path.join(root, "bench.py"), ["--gpus", "4", "--mode", "clips", "--steps", "2"], 4)]
    a = bench.parse_args(["--gpus", "8", "--mode", "clips"])
    assert a.frames == 32 and a.size == 512 and a.batch == 1 and a.cpu_frames >= 5


def test_eight_ranks_map_to_eight_devices(monkeypatch):
    """`bench.py --gpus 8 --mode clips`: the parent builds ONE torch.distributed.run command for 8 ranks, and the
    ranks it starts (LOCAL_RANK 0..7) pick 8 different HIP devices; RIB_BENCH_DEVICE is the one-GPU rehearsal switch."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = []
    monkeypatch.setattr("subprocess.call", lambda cmd, env=None: seen.append((cmd, env)) or 0)
    assert ribdist.self_launch(os.path.join(root, "bench.py"), ["--gpus", "8", "--mode", "clips"], 8) == 0
    (cmd, env), = seen
    assert cmd[:3] == [__import__("sys").executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[-5:] == [os.path.join(root, "bench.py"), "--gpus", "8", "--mode", "clips"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    devs = [ribdist.rank_device_index({"LOCAL_RANK": str(r), "RANK": str(r), "WORLD_SIZE": "8"}) for r in range(8)]
    assert devs == list(range(8))
    assert ribdist.rank_device_index({"LOCAL_RANK": "5", "RIB_BENCH_DEVICE": "0"}) == 0
    assert ribdist.rank_device_index({}) == 0


def _agree_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    ribdist.init_process_group("gloo")
    out = []
    ribdist.agree_or_raise(True, "fine")                       # everybody fine: returns on every rank
    try:
        ribdist.agree_or_raise(rank != 0, "rank 0 could not load the checkpoint")   # rank 0 failed: EVERY rank raises
        out.append("no error")
    except RuntimeError as e:
        out.append(str(e))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failure_on_rank0_ends_every_rank_before_the_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == ["rank 0 could not load the checkpoint"]
    assert res[1] == ["another rank failed: rank 0 could not load the checkpoint"]


def test_a_failed_start_up_names_rank_device_backend_and_ipc_mode(monkeypatch, capfd, tmp_path):
    """the first real multi-GPU run must diagnose itself.  A failure inside the start-up phases (process
    group + first communicator, the weight broadcast) prints ONE line with the rank, its device, the backend, the error text
    and the IPC mode in force, then propagates; the launcher - a parent that never touched the GPU - reports the exit code
    of its ranks and hands it back."""
    from render_in_between_amd import distributed as ribdist
    for k, v in (("RANK", "5"), ("WORLD_SIZE", "8"), ("LOCAL_RANK", "5"), ("HSA_ENABLE_IPC_MODE_LEGACY", "0"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29400")):
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("RIB_BENCH_DEVICE", raising=False)
    with pytest.raises(RuntimeError, match="unhandled system error"):
        with ribdist.startup_phase("init_process_group(nccl) + first communicator"):
            raise RuntimeError("NCCL error in: ProcessGroupNCCL.cpp:2000, unhandled system error\nhipIpcGetMemHandle: invalid argument")
    err = capfd.readouterr().err.strip().splitlines()
    assert len(err) == 1, err
    line = err[0]
    for piece in ("[rib rank 5/8 local_rank 5]", "init_process_group(nccl) + first communicator", "device cuda:5", "HSA_ENABLE_IPC_MODE_LEGACY=0",
                  "MASTER=127.0.0.1:29400", "RuntimeError: NCCL error in", "hipIpcGetMemHandle: invalid argument"):
        assert piece in line, (piece, line)
    # nothing is printed when nothing fails
    with ribdist.startup_phase("the weight broadcast"):
        pass
    assert capfd.readouterr().err == ""
    # the launcher: a fresh child per rank; a non-zero exit is reported on stderr and returned, stdout stays the ranks'
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\nprint('line of rank', os.environ['RANK'])\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    rc = ribdist.self_launch(str(script), [], 2)
    out = capfd.readouterr()
    assert rc != 0 and "[rib launcher] 2 ranks of rank.py ended with exit code" in out.err and "HSA_ENABLE_IPC_MODE_LEGACY=0" in out.err
    assert "line of rank 0" in out.out

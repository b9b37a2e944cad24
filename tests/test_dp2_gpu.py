"""Two data-parallel ranks (two processes, here sharing the one GPU of the test box, gloo transport so both may sit
on the same device) against ONE process stepping on the concatenated batch: SyncBatchNorm statistics exchange,
BatchNorm-backward sum exchange, gradient buckets and the SUM->MEAN fold must make the two runs the same training
trajectory (reference: SyncBatchNorm + DistributedDataParallel, scripts/train.py:100-102)."""
import os
import socket

import pytest
import torch

from oracle import fsnet_oracle as O

pytestmark = pytest.mark.gpu
B_RANK, H, W, STEPS = 2, 64, 128, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _full_batch(it):
    return O.synthetic_batch(2 * B_RANK, H, W, seed=300 + it)


def _build(dev, lanes=None):
    from fsnet_amd.configs import meta_arch_cfg, training_cfg
    from fsnet_amd.engine.runtime import RT
    from fsnet_amd.vision_base.networks.optimizers.optimizers import build_optimizer
    from fsnet_amd.vision_base.utils.builder import build
    RT.set_compute_dtype(torch.float32)
    RT.tie_noise = False
    if lanes is not None:
        RT.lanes = lanes         # explicit: the encoders as two lanes of one pass / as two chains (no autotune in two steps)
    m = build(**meta_arch_cfg(H, W, with_pose=True))
    m.load_state_dict(O.init_state(seed=11, with_pose=True), strict=True)
    m = m.to(dev).train()
    tc = training_cfg()
    return m, build_optimizer(m, **tc.optimizer), build(**tc.training_hook)


def _rank_main(rank, world, port, out_path, lanes, wgrad):
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["FSNET_AMD_DP_WGRAD"] = wgrad
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    try:
        m, opt, hook = _build(dev, lanes)
        losses = []
        for it in range(STEPS):
            full = _full_batch(it)
            mine = {k: (v[rank * B_RANK:(rank + 1) * B_RANK] if isinstance(v, torch.Tensor) else v) for k, v in full.items()}
            out = hook(mine, m, opt)
            losses.append(float(out["loss"].detach()))
        torch.cuda.synchronize()
        from fsnet_amd.engine.runtime import RT
        assert RT.dp is not None and RT.dp.world == world and hook.graph_captures == 0 and RT.lanes == lanes
        assert RT.dp.wgrad_mode == wgrad and hook.tune_done
        torch.save({"losses": losses,
                    "params": torch.cat([p.detach().flatten() for p in m.parameters()]).cpu(),
                    "running": torch.cat([b.detach().double().flatten() for n, b in m.named_buffers() if "running_" in n]).cpu()},
                   out_path % rank)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lanes,wgrad", [(True, "inline"), (False, "inline"), (False, "tail"), (True, "tail"),
                                         (True, "companion"), (False, "companion")])
def test_two_ranks_equal_one_process_on_the_full_batch(dev, tmp_path, lanes, wgrad):
    """every arrangement the training hook's autotune chooses between on a real node — the encoders as two lanes or two
    chains, the weight gradients inline, on companion streams, or the decoder's at the pose chain's tail — against the
    single process, which runs two chains with companions"""
    import torch.multiprocessing as mp
    port = _free_port()
    out_path = str(tmp_path / "rank%d.pt")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, out_path, lanes, wgrad)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0, "rank process failed (exit code %r)" % p.exitcode
    r0, r1 = torch.load(out_path % 0), torch.load(out_path % 1)
    # both ranks hold the same model after every step
    assert float((r0["params"] - r1["params"]).abs().max()) == 0.0
    assert float((r0["running"] - r1["running"]).abs().max()) == 0.0

    m, opt, hook = _build(dev, "auto")
    p_init = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    losses = []
    for it in range(STEPS):
        out = hook(dict(_full_batch(it)), m, opt)
        losses.append(float(out["loss"].detach()))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()
    running = torch.cat([b.detach().double().flatten() for n, b in m.named_buffers() if "running_" in n]).cpu()
    for it in range(STEPS):
        dp_loss = 0.5 * (r0["losses"][it] + r1["losses"][it])       # mean over the global batch
        assert abs(dp_loss - losses[it]) < 2e-5 * abs(losses[it]), (it, dp_loss, losses[it])
    # same trajectory up to fp32 summation order.  (|dp - single| < lr would hold for ANY gradient — Adam moves each
    # weight by ~lr per step — so compare the updates themselves: direction and relative size)
    from tests.test_dp_gpu import same_update
    agree, rel = same_update(r0["params"] - p_init, params - p_init)
    assert agree > 0.97 and rel < 0.2, (agree, rel)
    assert float(((r0["running"] - running).abs() / running.abs().clamp_min(1.0)).max()) < 1e-3


def test_bench_script_runs_with_two_ranks(tmp_path):
    """bench.py's multi-rank path — rendezvous from the torchrun environment, barrier + synchronize around the timed
    steps, MAX-reduce of the elapsed time, RT.dp.close(), ONE JSON line on rank 0 — exercised once before the driver
    runs it on a multi-GPU node: two ranks launched like the driver does (python -m torch.distributed.run), sharing
    the test box's single device over gloo (FSNET_AMD_BENCH_SHARED_DEVICE=1)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FSNET_AMD_BENCH_SHARED_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", FSNET_AMD_TUNE_STEPS="2")
    env.pop("FSNET_AMD_LANES", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
           "--batch", "2", "--height", "64", "--width", "128"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                 # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2" and d["config"]["dp_world"] == 2
    assert d["config"]["dp_collectives"] == "torch.distributed"          # gloo rig: no direct RCCL communicator
    assert d["config"]["syncbn_exchanges_per_step"] > 40 and d["config"]["gradient_buckets_per_step"] >= 4
    assert d["value"] > 0 and abs(d["value"] - 4 * 3 / (d["ms_per_step"] * 3e-3)) < 0.02 * d["value"]
    # the encoder-pass autotune ran before the warm-up steps (eager steps on this rig) and the line carries both timings
    ep = d["config"]["encoder_pass_ms"]
    assert ep["chains"] > 0 and ep["lanes"] > 0 and ep["chosen"] in ep and ep["ranks"] == 2
    assert ep["chains+tail"] > 0 and ep["lanes+companions"] > 0
    assert d["config"]["encoder_pass"].startswith("two " + ep["chosen"].split("+")[0]) and d["config"]["autotune_steps"] >= 24
    assert "cpu_baseline" not in d                           # rank 0 at N = 1 only

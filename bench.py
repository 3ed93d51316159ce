#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s through the GAE + PPO update (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], per GPU): Atari-shaped rollouts, 512 envs x 128 steps (+1 bootstrap row),
uint8 (4,84,84) frames, NatureCNN-512 shared actor-critic, PPO with the reference's Atari preset
(legacy/experiments/atari.py:952-973).  Weak scaling: N GPUs = N x 512 envs (8 GPUs = configs[2], 4096 envs),
data parallel with one RCCL all-reduce of the advantage statistics and one of the flat gradient per step.
A "step" is one full ``trainer.step``: GAE scan + statistics, forward, fused loss fwd/bwd, backward, gradient
all-reduce, clip + Adam.  The sample is synthetic (no ALE on the box) and RESIDENT IN HBM before the timed
region; weights are randomly initialised.  value = T * B_global * K / t, t = max over ranks of the K-step wall time.

Extra objects on the JSON line: ``roofline`` (the dominant kernel family: the FP32-MFMA GEMM, timed live with
HIP events on the launch stream in an extra untimed step), ``roofline_gae`` (the GAE scan against HBM),
``cpu_baseline`` (the oracle's CPU restatement of the same step on a bounded sample, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0  # HBM3E spec

POLICY = dict(obs_dim={"obs": (4, 84, 84)}, action_dim=6, hidden_dim=512, num_dense_layers=0, num_rnn_layers=0,
              popart=False, layernorm=False, shared_backbone=True, seed=1,
              cnn_layers=dict(obs=[(32, 8, 4, 0, 'zeros'), (64, 4, 2, 0, 'zeros'), (64, 3, 1, 0, 'zeros')]))
TRAINER = dict(discount_rate=0.99, gae_lambda=0.97, eps_clip=0.2, clip_value=True, dual_clip=False, value_loss='huber',
               value_loss_weight=1.0, value_loss_config=dict(delta=10.0), entropy_bonus_weight=0.01, optimizer='adam',
               optimizer_config=dict(lr=5e-4), popart=False, max_grad_norm=40.0, bootstrap_steps=1)


def device_sample(seed, T, B, device):
    """Synthetic sample built directly in HBM (uint8 frames via torch's device RNG; flags/scalars from numpy)."""
    from srl_amd.runtime import synthetic
    arrays = synthetic.make_sample_arrays(seed=seed, T=T, B=B, obs_spec={}, action_dims=6, p_done=1.0 / 800)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    dev = {k: torch.from_numpy(v).to(device) for k, v in arrays.items()}
    dev["obs.obs"] = torch.randint(0, 256, (T + 1, B, 4, 84, 84), dtype=torch.uint8, device=device, generator=gen)
    return synthetic.to_sample_batch(dev)


def gae_microbench(sample, targs, device, reps=200, big_B=None):
    """Back-to-back launches of the scan on the step's own leaves, or (``big_B``) on device-generated leaves of a
    batch wide enough to fill the chip -- the size at which an HBM roofline fraction means something."""
    from srl_amd import hip
    if big_B is None:
        ar = sample.analyzed_result
        Tb, B = sample.on_reset.shape[:2]
        leaves = (sample.reward, ar.value, sample.done, sample.truncated, sample.on_reset)
    else:
        Tb, B = sample.on_reset.shape[0], big_B
        gen = torch.Generator(device=device).manual_seed(7)
        rnd = lambda: torch.rand((Tb, B, 1), device=device, generator=gen)
        done = rnd() < 1.0 / 800
        trunc = (rnd() < 1.0 / 3200) & ~done
        on_reset = torch.zeros_like(done)
        on_reset[1:] = (done | trunc)[:-1]
        reward = torch.where(on_reset, 0.0, rnd() - 0.5)
        leaves = (reward, rnd(), done.to(torch.uint8), trunc.to(torch.uint8), on_reset.to(torch.uint8))
    adv = torch.zeros((Tb, B, 1), device=device)
    ret = torch.zeros((Tb, B, 1), device=device)
    stats = torch.zeros(3, dtype=torch.float64, device=device)
    args = (*leaves, targs["discount_rate"], targs["gae_lambda"], adv, ret)
    for _ in range(10):
        hip.gae_scan(*args, stats=stats)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        hip.gae_scan(*args, stats=stats)
    b.record()
    torch.cuda.synchronize()
    T = Tb - 1
    return dict(ms=a.elapsed_time(b) / reps, work=19.0 * T * B + 7.0 * B)


def cpu_baseline(T, threads):
    """The oracle's restatement of the same trainer step (torch-CPU, op for op with the reference) on a bounded
    sample of the same workload."""
    from oracle.net import OracleActorCritic
    from oracle.trainer import OracleMappo
    from srl_amd.algorithm.netspec import build_netspec
    from srl_amd.runtime import synthetic
    torch.set_num_threads(threads)
    B = 8
    _, init = build_netspec(**POLICY)
    net = OracleActorCritic(**POLICY)
    net.load_state_dict({k: v.numpy() for k, v in init.items()})
    tr = OracleMappo(net, **TRAINER)
    arrays = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=1.0 / 800)
    tr.step(arrays)  # warm-up
    times = []
    t_all = time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_all < 10.0 and len(times) < 8):
        t0 = time.perf_counter()
        tr.step(arrays)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return dict(value=T * B / med, unit="env-steps/s", cores=threads, kind="port",
                sample=f"{T}x{B} env-steps of the same Atari-shaped workload, {len(times)} timed steps, "
                       f"median {med:.3f} s/step")


def recorded_traffic(kernel_substr):
    """HBM bytes per step of the kernels whose name contains `kernel_substr`, from the PMC passes committed under
    profiles/ (FETCH_SIZE and WRITE_SIZE collected in separate `rocprofv3 --pmc` runs of this same script with
    --steps 1 --warmup 1, FETCH_SIZE x2 per the MI355X guide's gfx950 correction: scripts/hbm_traffic.py).  Counters
    cannot be read from inside the process, so the line carries the recorded figure and names its source."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_hbm_traffic_v*.csv")),
                   key=lambda f: int(f.rsplit("_v", 1)[1].split(".")[0]))
    if not files:
        return dict(traffic=None)
    total, steps_in_run = 0.0, 2  # warm-up step + timed step
    for r in csv.DictReader(open(files[-1])):
        if kernel_substr in r["kernel"]:
            per_launch = float(r["FETCH_bytes_per_launch_corrected_x2"]) + float(r["WRITE_bytes_per_launch"])
            total += per_launch * float(r["dispatches"]) / steps_in_run
    return dict(traffic=round(total), traffic_unit="HBM bytes per step (fetch + write) over the same launches",
                traffic_source=os.path.relpath(files[-1], os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs-per-gpu", type=int, default=512)
    ap.add_argument("--rollout-len", type=int, default=128)
    ap.add_argument("--chunk-rows", type=int, default=16384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even with one rank")
    ap.add_argument("--from-host", action="store_true",
                    help="also time the step fed from the pinned ingest ring (H2D of every leaf inside the timed region, "
                         "overlapped with the previous update); reported as `pcie_inclusive`, never as `value`")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    # SRL_BENCH_BACKEND=gloo lets the ranks share one GPU: only for tests/test_gpu_dist.py, which drives this script with
    # two ranks on a one-GPU box to check the multi-rank control flow (RCCL needs a device per rank)
    backend = os.environ.get("SRL_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"

    import srl_amd
    from srl_amd import hip
    from srl_amd.api import config, trainer as trainer_api
    srl_amd.register_all()

    use_dist = world > 1 or args.force_dist
    if use_dist:
        kw = dict(device_id=torch.device(device)) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
    trainer = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, chunk_rows=args.chunk_rows)),
                               config.Policy("actor-critic", args=POLICY))
    if use_dist:
        trainer.distributed(rank=rank, world_size=world, init_method="env://")

    T, B = args.rollout_len, args.envs_per_gpu
    sample = device_sample(1000 + rank, T, B, device)

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(sample)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = trainer.step(sample)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- untimed extra step with per-kernel HIP events (same stream as the launches) -----------------------
    roofline = roofline_gae = breakdown = None
    if not args.no_profile:
        # EVERY rank takes this step (its collectives need all of them); only rank 0 wraps its launches in events
        prof = hip.KernelProfile() if rank == 0 else None
        if prof is not None:
            hip.set_profile(prof)
        trainer.step(sample)
        if prof is not None:
            hip.set_profile(None)
    if rank == 0 and not args.no_profile:
        summ = prof.summary()
        mm = [v for k, v in summ.items() if k == "gemm" or k.startswith("conv_")]  # every launch of gemm_kernel<...>
        g = dict(calls=sum(v["calls"] for v in mm), ms=sum(v["ms"] for v in mm), work=sum(v["work"] for v in mm))
        ach = g["work"] / (g["ms"] * 1e-3) / 1e12
        roofline = dict(kernel="gemm_kernel<...> (v_mfma_f32_32x32x2_f32): dense + implicit-conv launches of one step",
                        bound="mfma", achieved=round(ach, 2), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s",
                        frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4), launches=g["calls"],
                        ms_per_step=round(g["ms"], 3), flops_per_step=g["work"], **recorded_traffic("gemm_kernel"))
        # the scan is a ~microsecond kernel: time it as 200 back-to-back launches between two events on the launch
        # stream so that host enqueue latency does not sit inside the interval
        s = gae_microbench(sample, TRAINER, device)
        gbs = s["work"] / (s["ms"] * 1e-3) / 1e9
        big = gae_microbench(sample, TRAINER, device, reps=30, big_B=1 << 20)
        big_gbs = big["work"] / (big["ms"] * 1e-3) / 1e9
        roofline_gae = dict(kernel="gae_scan_reg_kernel", bound="hbm", achieved=round(gbs, 2), peak=PEAK_HBM_GBS,
                            unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 5), traffic=None,
                            us_per_launch=round(s["ms"] * 1e3, 2), algorithmic_bytes=s["work"],
                            note="this step's own [T, 512] leaves: launch-latency bound, see 'saturated'",
                            saturated=dict(envs=1 << 20, rollout_len=T, achieved=round(big_gbs, 1),
                                           frac=round(big_gbs / PEAK_HBM_GBS, 4),
                                           us_per_launch=round(big["ms"] * 1e3, 1), algorithmic_bytes=big["work"]))
        breakdown = {k: round(v["ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}

    pcie = None
    if args.from_host and world == 1:
        from srl_amd.namedarray import recursive_apply
        from srl_amd.runtime.ingest import SampleRing
        host = recursive_apply(sample, lambda x: x.cpu().numpy())
        ring = SampleRing(host[:, 0], batch_size=B, slots=2, device=device)
        for _ in range(2):
            ring.put_batch(host)
        del host

        def fed_step():
            b = ring.get_device()
            r = trainer.step(b)
            slot = b.metadata["ring_slot"]
            ring.release(slot)
            ring.recycle(slot)
            return r

        for _ in range(args.warmup):
            fed_step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fed_step()
        sync()
        el = time.perf_counter() - t0
        pcie = dict(value=T * B * args.steps / el, unit="env-steps/s", ms_per_step=1e3 * el / args.steps,
                    host_bytes_per_step=ring.nbytes() // 2,
                    note="sample in the pinned ingest ring; async H2D of every leaf on a side stream, double-buffered")

    if rank == 0:
        steps_total = T * B * world * args.steps
        line = dict(metric="env-steps/sec through GAE+PPO update", value=steps_total / elapsed, unit="env-steps/s",
                    n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=1e3 * elapsed / args.steps,
                    higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload=f"Atari-shaped PPO+GAE, {B} envs x {T} steps per GPU (BASELINE configs[1]; "
                                         f"x{world} GPUs data-parallel), NatureCNN-512, uint8 (4,84,84) frames",
                                envs_per_gpu=B, rollout_len=T, global_envs=B * world, parallelism=f"dp{world}",
                                chunk_rows=args.chunk_rows, policy_loss=res.stats.get("policy_loss")),
                    roofline=roofline, roofline_gae=roofline_gae, kernel_ms_per_step=breakdown)
        if pcie is not None:
            line["pcie_inclusive"] = pcie
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T, threads=min(os.cpu_count() or 1, 32))
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

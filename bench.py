"""Benchmark of the hot path: env-steps/s through the GAE + PPO update (BASELINE.json metric).

    python3 bench.py --gpus N --steps K --warmup W
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload = the configuration the metric is quoted on (BASELINE.json configs[2]): Atari-shaped rollouts, **4096 envs x
128 steps** (+1 bootstrap row) per update, uint8 (4,84,84) frames, NatureCNN-512 shared actor-critic, PPO with the
reference's Atari preset (legacy/experiments/atari.py:952-973).  STRONG scaling: the global batch is fixed, N GPUs
take 4096 / N env columns each, data parallel with one RCCL all-reduce of the advantage
statistics and a bucketed all-reduce of the flat gradient per step.  A "step" is one full ``trainer.step``: GAE scan
+ statistics, forward, fused loss fwd/bwd, backward, gradient all-reduce, clip + Adam.  Synthetic data (no ALE on the
box), random-init weights.

``value`` = T * B_global * K / t, t = max over ranks of the K-step wall time between barrier + synchronize, of SURVEY.md
8d's ``t_update``: the sample lies in PINNED HOST memory in [Tb, B] namedarray layout when an update starts and every
leaf the update needs from there is copied inside the timed region -- every leaf EXCEPT the frames, which are already in
HBM: the rollout that produced the sample uploaded each observation once for inference (reference
actor_critic_policy.py:467-469) and ``ObsRing`` kept it there, in the first layer's layout; the sample names its rows by
the ring stamps the rollout returned (``analyzed_result.obs_ref``) and ``SampleRing.get_device`` binds them instead of
sending 14.9 GB over the link a second time (runtime/obs_ring.py).  The rollout phase that fills the ring runs before
the timed region (it IS the rollout; its rate is reported as ``rollout_inference``).  Two more figures of the same
update, same run: ``resident_in_hbm`` (the whole sample, frames as a plain [Tb, B, 4, 84, 84] tensor, already on the
device) and ``from_pinned_host`` (no ring: every leaf including the frames crosses the link inside the timed region,
double-buffered on a side stream; reference api/trainer.py:211-228) -- PCIe-bound.

Extra objects on the JSON line: ``roofline`` (dominant kernel family: the matrix-core contractions, timed live with
HIP events on the launch stream in an extra untimed step), ``roofline_gae`` (the GAE scan against HBM at this batch,
with t / t_launch_floor, and saturated), ``cpu_baseline`` (the oracle's CPU restatement of the same step on a
bounded sample, rank 0, N = 1 only).

Profiling recipe (scripts/profile_bench.sh; counters in passes of their own, the interpreter directly after ``--``):
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-plain-copy
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-host --no-profile
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-host --no-profile
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES \
        SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-from-host --no-profile
    python3 scripts/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --steps-in-run 2 --out profiles/r03_hbm_traffic_vN.csv
    python3 scripts/pmc_summary.py gpurun_out/pmc_sq --steps-in-run 2 --csv profiles/r03_sq_counters_vN.csv
"""
import argparse
import json
import os
import sys
import time

# read when libamdhip64 is loaded, i.e. by `import torch`: the update runs on up to six streams, and with the runtime's default
# of 4 hardware queues the ingest copy shares one with a compute pipeline (srl_amd/hip.py; INTEGRATION.md switches)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA
PEAK_HBM_GBS = 8000.0  # HBM3E spec
GLOBAL_ENVS = 4096  # BASELINE.json metric: "4096 envs x 128 steps"

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
    # frame stacks as the reference's FrameStack wrapper produces them (atari_wrappers.py:211-242): the four latest planes of
    # each environment, newest last; four copies of the first plane where the sample says the episode starts
    planes = torch.randint(0, 256, (T + 1, B, 84, 84), dtype=torch.uint8, device=device, generator=gen)
    obs = torch.empty((T + 1, B, 4, 84, 84), dtype=torch.uint8, device=device)
    fresh = dev["on_reset"].reshape(T + 1, B).bool()
    for t in range(T + 1):
        rep = planes[t][:, None].expand(-1, 4, -1, -1)
        if t == 0:
            obs[t] = rep
        else:
            obs[t] = torch.where(fresh[t][:, None, None, None], rep, torch.cat([obs[t - 1][:, 1:], planes[t][:, None]], dim=1))
    dev["obs.obs"] = obs
    del planes
    # the observation-ring stamp every step carries (analyzed_result.obs_ref): none yet -- the rollout phase of main() fills it
    dev["analyzed_result.obs_ref"] = torch.full((T + 1, B, 1), -1, dtype=torch.int64, device=device)
    return synthetic.to_sample_batch(dev)


def launch_floor_us(device, reps=400):
    """Back-to-back launches of an (almost) empty kernel through the same C ABI and stream: the per-launch floor the
    microsecond kernels are compared with (report t / t_launch_floor)."""
    from srl_amd import hip
    x = torch.zeros(4, dtype=torch.float32, device=device)
    y = torch.zeros(4, dtype=torch.float32, device=device)
    for _ in range(20):
        hip.copy2d(x.data_ptr(), 4, y.data_ptr(), 4, 1, 4)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        hip.copy2d(x.data_ptr(), 4, y.data_ptr(), 4, 1, 4)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


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
    ws = hip.gae_scan_workspace(B, 1, device)  # as the trainer calls it: statistics without a zeroing launch
    for _ in range(10):
        hip.gae_scan(*args, stats=stats, workspace=ws)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        hip.gae_scan(*args, stats=stats, workspace=ws)
    b.record()
    torch.cuda.synchronize()
    T = Tb - 1
    return dict(ms=a.elapsed_time(b) / reps, work=19.0 * T * B + 7.0 * B)


def mlp_roofline(device, T=128, B=4096, steps=10, trainer_args=None):
    """The MLP the north star names, at the metric's batch: BASELINE configs[0]'s separate actor / critic 2 x 64 nets
    (102.5 kFLOP per sample and epoch, forward + backward) over 4096 envs x 128 steps through the same trainer --
    GAE scan, loss, optimiser included in the time, so the fraction is a floor for the contractions alone."""
    from srl_amd import hip
    from srl_amd.api import config, trainer as trainer_api
    from srl_amd.runtime import synthetic
    pol = dict(obs_dim=4, action_dim=2, hidden_dim=64, num_dense_layers=2, num_rnn_layers=0, popart=False, layernorm=False,
               shared_backbone=False, seed=1)
    tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=False, optimizer_config=dict(lr=3e-4), **(trainer_args or {}))),
                          config.Policy("actor-critic-separate", args=pol))
    arrays = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec=synthetic.CARTPOLE_OBS, action_dims=2, p_done=0.05)
    sample = synthetic.to_sample_batch({k: torch.from_numpy(v).to(device) for k, v in arrays.items()})
    for _ in range(3):
        tr.step(sample)
    hip.dispatch_counts(reset=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        tr.step(sample)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    disp = hip.dispatch_counts(reset=True)
    flops = 102.5e3 * T * B
    ach = flops / (ms * 1e-3) / 1e12
    f16 = os.environ.get("SRL_MLP_F16", "1") != "0"   # round 6: the towers' products as three f16 piece products (csrc/mlp_sigh.h)
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if f16 else PEAK_FP32_MFMA_TFLOPS
    return dict(kernel="actor / critic 2 x 64 MLPs (CartPole-shaped nets of BASELINE configs[0]) at 4096 envs x 128 steps, whole update",
                bound="mfma", achieved=round(ach, 3), unit="TFLOP/s", peak=round(peak, 1), frac=round(ach / peak, 5),
                peak_basis=("16-bit MFMA peak / 3: a float32 multiply-add as three f16 piece products (the pipe the towers run on since "
                            "round 6)" if f16 else "float32 MFMA peak (v_mfma_f32_32x32x2_f32)"),
                achieved_over_fp32_mfma_peak=round(ach / PEAK_FP32_MFMA_TFLOPS, 5),
                ms_per_step=round(ms, 3), env_steps_per_s=round(T * B / (ms * 1e-3)), flops_per_step=flops, traffic=None,
                kernel_family_launches={k: v // steps for k, v in disp.items()} if isinstance(disp, dict) else disp,
                note="each tower is ONE launch per direction (csrc/mlp_sigh.h: the chain's shape is a template argument, every Linear "
                     "layer three products of f16 pieces; no tape -- the backward pass walks the chain forward again from the 16-byte "
                     "observation rows): 66 MFMAs (32x32x16 f16) per 32 rows forward, ~190 backward (round 5, float32 MFMAs: 168 / "
                     "496; SRL_MLP_F16=0).  `frac` counts the ALGORITHMIC 102.5 kFLOP per env-step over the whole update (GAE scan, loss, "
                     "optimiser included) against the roof of the pipe the products run on; `achieved_over_fp32_mfma_peak` is last "
                     "round's denominator (SURVEY 8d)")


def smac_config_leg(device, steps=10):
    """BASELINE configs[3] at full size on ONE GPU: SMAC 3m, 1024 shared environments x 3 agents x 100 steps, `smac_rnn` (shared
    LSTM-64 actor / critic towers, PopArt), the whole sample through one trainer.step (multi-agent namedarray path)."""
    from srl_amd.api import config, trainer as trainer_api
    from srl_amd.runtime import synthetic
    Ts, Bs, A, H = 100, 1024, 3, 64
    pol = dict(map_name="3m", hidden_dim=H, chunk_len=10, seed=1, shared=True)
    tr_args = dict(popart=True, clip_value=True, dual_clip=False, value_loss="huber", value_loss_config=dict(delta=10.0),
                   max_grad_norm=10.0, optimizer_config=dict(lr=5e-4, eps=1e-5))
    arrays = synthetic.make_multiagent_arrays(seed=4, T=Ts, B=Bs, agents=A, obs_spec={"local_obs": ((30,), "f32"), "state": ((48,), "f32")},
                                              action_dim=9, p_done=1 / 60, policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
    tr = trainer_api.make(config.Trainer("mappo", args=tr_args), config.Policy("smac_rnn", args=pol))
    sample = synthetic.to_sample_batch({k: torch.from_numpy(v).to(device) for k, v in arrays.items()})
    for _ in range(3):
        tr.step(sample)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(steps):
        tr.step(sample)
    e.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(e) / steps
    return dict(workload="BASELINE configs[3]: SMAC 3m MAPPO, 1024 envs x 3 agents x 100 steps, shared LSTM-64 actor-critic (smac_rnn), "
                         "PopArt; whole batch on one GPU, sample resident in HBM", ms_per_update=round(ms, 3),
                env_steps_per_s=round(Ts * Bs / (ms * 1e-3)), agent_steps_per_s=round(Ts * Bs * A / (ms * 1e-3)), steps=steps)


def football_config_leg(device, steps=3):
    """BASELINE configs[4] at its per-GPU size: 256 of the 2048 football environments x 200 steps, `football-smm-separate` with an
    LSTM (CNN + LSTM on (4, 96, 72) uint8 frames, separate actor / critic, PopArt; 676 M parameters).  The orthogonal
    initialisation (a QR of a 22528 x 11264 matrix: minutes on one core) is replaced by a scaled normal draw -- random-init weights
    of the same architecture."""
    import math
    from srl_amd.api import config, trainer as trainer_api
    from srl_amd.runtime import synthetic
    T, B, H = 200, 256, 128
    orig = torch.nn.init.orthogonal_

    def cheap(t, gain=1.0):
        with torch.no_grad():
            return t.normal_(0.0, gain / math.sqrt(t.shape[1] if t.dim() > 1 else t.numel()))

    torch.nn.init.orthogonal_ = cheap
    try:
        tr = trainer_api.make(config.Trainer("mappo", args=dict(popart=True, clip_value=True, value_loss="huber",
                                                                value_loss_config=dict(delta=10.0), max_grad_norm=10.0,
                                                                optimizer_config=dict(lr=5e-4, eps=1e-5))),
                              config.Policy("football-smm-separate", args=dict(rnn_type="lstm", seed=1)))
    finally:
        torch.nn.init.orthogonal_ = orig
    arr = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec={}, action_dims=19, p_done=1 / 400,
                                       policy_state={"actor_hx": (1, 2 * H), "critic_hx": (1, 2 * H)})
    dev = {k: torch.from_numpy(v).to(device) for k, v in arr.items()}
    gen = torch.Generator(device=device).manual_seed(0)
    dev["obs.obs"] = torch.randint(0, 256, (T + 1, B, 4, 96, 72), dtype=torch.uint8, device=device, generator=gen)
    sample = synthetic.to_sample_batch(dev)
    tr.step(sample)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(sample)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    # algorithmic float32 flops of the two encoders' contractions, forward + both gradients (3 x 2 x 2 flops per multiply-add):
    # three convolutions + the halving Linear tower (football_rnn.py:34-55, cnn.py:96-135); the LSTM, heads and LayerNorms are < 1 %
    macs = (92 * 68 * 4 * 100 + 90 * 66 * 8 * 36 + 88 * 64 * 4 * 72 + 22528 * 11264 + 11264 * 5632 + 5632 * 2816 + 2816 * 1408 +
            1408 * 704 + 704 * 128)
    tflop = 6 * 2 * macs * T * B * 1e-12
    return dict(workload="BASELINE configs[4] per-GPU share: Google Football 11v11, 256 of 2048 envs x 200 steps, CNN + LSTM-128 "
                         "(football-smm-separate, 676 M parameters, PopArt); sample resident in HBM",
                ms_per_update=round(ms, 2), env_steps_per_s=round(T * B / (ms * 1e-3)), steps=steps,
                parameters=int(tr.policy.net.spec.total_params),
                roofline=dict(bound="mfma", achieved=round(tflop / (ms * 1e-3), 1), peak=PEAK_BF16_MFMA_TFLOPS, unit="TFLOP/s",
                              frac=round(tflop / (ms * 1e-3) / PEAK_BF16_MFMA_TFLOPS, 4), tflop_per_update=round(tflop, 1),
                              note="algorithmic float32 TFLOP/s of the encoders' contractions over the WHOLE update's wall time, against "
                                   "the dense 16-bit matrix peak; the dense tower runs as three f16 piece products per multiply-add "
                                   "(csrc/h2gemm.h, h2gemmp.h, h2tn.h), the 4 / 8-channel convolutions on the vector units "
                                   "(csrc/conv_small.hip)"))


def shard_config_leg(extra=()):
    """BASELINE configs[1] (Pong-shaped: 512 envs x 128 steps on one GPU) = the per-rank shard of configs[2] at 8 GPUs: this script
    itself as a CHILD process at --global-envs 512 (fresh process: its own trainer, rings and allocator state), ring-fed and
    resident.  `extra`: more flags (--force-dist: the same through a one-rank RCCL process group, i.e. with the advantage-statistics
    all-reduce and the bucketed gradient all-reduce enqueued and executed)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--global-envs", "512", "--steps", "20", "--warmup", "5", "--seeds", "0",
           "--no-cpu-baseline", "--no-plain-copy", "--no-closed-loop", "--no-configs", "--no-mlp", *extra]
    env = dict(os.environ)
    if "--force-dist" in extra:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if out.returncode != 0 or len(lines) != 1:
        return dict(error=f"child exited {out.returncode}: {out.stderr[-400:]}")
    return json.loads(lines[0])


def other_configs(line, device):
    """`configs`: ms per update and env-steps/s of BASELINE configs[1] (= the 8-GPU shard of configs[2]), [3] and [4] (per-GPU share).
    `scaling_model`: what ONE GPU can say about strong scaling of the headline (no multi-GPU box in this pool): the shard's update
    time against an eighth of the full update, with and without the collectives enqueued, and the per-update costs that do not
    shrink with the env columns."""
    pick = lambda d: dict(ms_per_update=round(d["ms_per_step"], 3), env_steps_per_s=round(d["value"]),
                          resident_ms_per_update=round(d["resident_in_hbm"]["ms_per_step"], 3),
                          resident_env_steps_per_s=round(d["resident_in_hbm"]["value"]))
    configs, model = {}, None
    shard = shard_config_leg()
    shard_dist = shard_config_leg(("--force-dist",))
    if "error" in shard:
        configs["c1_pong_512x128_and_8gpu_shard"] = shard
    else:
        configs["c1_pong_512x128_and_8gpu_shard"] = dict(
            workload="BASELINE configs[1]: Atari-shaped PPO+GAE, 512 envs x 128 steps on one GPU, NatureCNN-512 (also exactly the "
                     "per-rank shard of configs[2] at 8 GPUs); ring-fed (headline definition) and resident", **pick(shard),
            rollout_requests_per_s=round(shard.get("rollout_inference", {}).get("value", 0)),
            kernel_ms_per_update=shard.get("kernel_ms_per_step"), launches_of_the_contractions=(shard.get("roofline") or {}).get("launches"))
    try:
        configs["c3_smac_3m_1024x3x100"] = smac_config_leg(device)
    except Exception as e:
        configs["c3_smac_3m_1024x3x100"] = dict(error=repr(e))
    try:
        configs["c4_football_256x200_per_gpu"] = football_config_leg(device)
    except Exception as e:
        configs["c4_football_256x200_per_gpu"] = dict(error=repr(e))
    torch.cuda.empty_cache()
    if "error" not in shard:
        t_full, t_shard = line["ms_per_step"], shard["ms_per_step"]
        r_full, r_shard = line["resident_in_hbm"]["ms_per_step"], shard["resident_in_hbm"]["ms_per_step"]
        kms = shard.get("kernel_ms_per_step") or {}
        model = dict(
            what="strong scaling of the headline to 8 GPUs as far as one GPU can measure it: every rank of an 8-GPU run executes "
                 "exactly the 512-env update timed here, plus the collectives' transfer time over xGMI (not measurable on this box)",
            ms_full_update_4096=round(t_full, 3), ms_shard_update_512=round(t_shard, 3),
            ideal_shard_ms=round(t_full / 8, 3),
            predicted_efficiency_8gpu_compute_only=round(t_full / 8 / t_shard, 4),
            predicted_efficiency_8gpu_compute_only_resident=round(r_full / 8 / r_shard, 4),
            fixed_cost_ms_per_update=round(t_shard - t_full / 8, 3),
            per_update_costs_that_do_not_shrink_with_env_columns=dict(
                note="from the shard run's per-kernel events (one launch at a time): kernels whose work is per update or per "
                     "executor, not per row; the rest of `fixed_cost_ms_per_update` is host launch latency and pipeline ramps "
                     "(4 row chunks = one per pipeline: nothing left to overlap)",
                gae_scan_ms=kms.get("gae_scan"), grad_sumsq_ms=kms.get("grad_sumsq"), adam_step_ms=kms.get("adam_step"),
                layernorm_and_loss_ms=round(sum(v for k, v in kms.items() if k.startswith("layernorm") or k.startswith("ln_heads") or k.startswith("ppo_loss") or
                                                k.startswith("categorical")), 3),
                profiled_kernel_launches_per_update=shard.get("launches_per_step")))
        if "error" not in shard_dist:
            d = shard_dist["ms_per_step"]
            gb = (shard_dist.get("config") or {}).get("grad_buckets")
            model.update(ms_shard_update_512_with_collectives_world1=round(d, 3),
                         collectives_enqueue_and_execute_ms_world1=round(d - t_shard, 3),
                         predicted_efficiency_8gpu_with_world1_collectives=round(t_full / 8 / d, 4),
                         grad_buckets=gb,
                         collectives_note="one-rank RCCL process group (--force-dist): the 24-byte statistics all-reduce and the bucketed "
                                          "6.98 MB gradient all-reduce are enqueued from inside the backward passes and executed by RCCL "
                                          "(world 1: no link traffic).  At 8 ranks a ring all-reduce moves 2 x 7/8 x 6.98 MB per rank over "
                                          "xGMI (~0.1-0.3 ms at 50-150 GB/s per link), issued bucket by bucket under the backward pass")
        else:
            model["collectives_world1_error"] = shard_dist["error"]
    return configs, model


def physical_cores():
    """Physical cores this process may run on: distinct (package, core) pairs of the CPUs in its affinity mask."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    for cpu in allowed:
        base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
        try:
            with open(base + "physical_package_id") as f:
                pkg = f.read().strip()
            with open(base + "core_id") as f:
                core = f.read().strip()
            cores.add((pkg, core))
        except OSError:
            cores.add(("?", str(cpu)))
    return max(1, len(cores)), len(allowed)


def _cpu_steps(T, B, threads, budget_s, max_steps):
    from oracle.net import OracleActorCritic
    from oracle.trainer import OracleMappo
    from srl_amd.algorithm.netspec import build_netspec
    from srl_amd.runtime import synthetic
    torch.set_num_threads(threads)
    _, init = build_netspec(**POLICY)
    net = OracleActorCritic(**POLICY)
    net.load_state_dict({k: v.numpy() for k, v in init.items()})
    tr = OracleMappo(net, **TRAINER)
    arrays = synthetic.make_sample_arrays(seed=0, T=T, B=B, obs_spec=synthetic.ATARI_OBS, action_dims=6, p_done=1.0 / 800)
    tr.step(arrays)  # warm-up (allocator, thread pool)
    times, t_all = [], time.perf_counter()
    while len(times) < 2 or (time.perf_counter() - t_all < budget_s and len(times) < max_steps):
        t0 = time.perf_counter()
        tr.step(arrays)
        times.append(time.perf_counter() - t0)
    return times


def cpu_baseline(T):
    """The oracle's restatement of the same trainer step (torch-CPU, op for op with the reference: float32-widened
    frames, float64 GAE loop, autograd loss, torch.optim.Adam) on a bounded sample of the same workload, on all
    physical cores and on one thread.  The per-update cost of this path is linear in the env columns B (every op is
    per row), so env-steps/s at the reduced B carries over to B = 4096 (whose float32 frames alone, 60 GB, would not
    fit the sample budget)."""
    cores, logical = physical_cores()
    B_all, B_one = 64, 8
    t_all = _cpu_steps(T, B_all, cores, budget_s=12.0, max_steps=5)
    t_one = _cpu_steps(T, B_one, 1, budget_s=6.0, max_steps=3)
    sweep = {}
    for th in (16, 32, 64):  # torch-CPU does not scale to every core of a big host on this step: report what does best too
        if th < cores:
            tt = _cpu_steps(T, B_all, th, budget_s=4.0, max_steps=2)
            sweep[str(th)] = round(T * B_all / float(np.median(tt)), 1)
    torch.set_num_threads(cores)
    med, mn = float(np.median(t_all)), float(min(t_all))
    med1, mn1 = float(np.median(t_one)), float(min(t_one))
    every = dict(sweep, **{str(cores): round(T * B_all / med, 1)})
    best_threads = max(every, key=every.get)
    return dict(value=T * B_all / med, unit="env-steps/s", cores=cores, kind="port", best=T * B_all / mn,
                logical_cpus=logical, other_thread_counts=sweep,
                best_thread_count=dict(threads=int(best_threads), value=every[best_threads],
                                       note="torch-CPU does not scale to every core of this host on this step: the thread count of the "
                                            "sweep (16 / 32 / 64 / all physical cores) that did best, same sample"),
                one_thread=dict(value=T * B_one / med1, best=T * B_one / mn1, cores=1,
                                sample=f"{T}x{B_one} env-steps, {len(t_one)} timed steps, median {med1:.3f} s, min {mn1:.3f} s"),
                sample=f"{T}x{B_all} env-steps (B reduced from {GLOBAL_ENVS}: the path is row-independent, cost linear "
                       f"in B) of the same Atari-shaped workload and seeds, {len(t_all)} timed steps on {cores} threads "
                       f"(= physical cores of the affinity mask; {logical} logical), median {med:.3f} s, min {mn:.3f} s")


def _kernel_digest():
    from srl_amd.provenance import kernel_sources_digest
    return kernel_sources_digest()


def recorded_traffic(kernel_substr, envs, T, chunk_rows):
    """HBM bytes per step of the kernels whose name contains one of `kernel_substr`, from the newest PMC passes committed
    under profiles/ (FETCH_SIZE and WRITE_SIZE in separate `rocprofv3 --pmc` runs of this script, FETCH_SIZE x2 per
    the MI355X guide's gfx950 correction: scripts/hbm_traffic.py, which records the run's configuration next to the
    table).  Counters cannot be read from inside the process, so the line carries the recorded figure and names its
    source -- and only when that recording was made at THIS run's configuration; otherwise traffic is null."""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    metas = sorted(glob.glob(os.path.join(here, "profiles", "r*_hbm_traffic_v*.json")),
                   key=lambda f: (int(os.path.basename(f)[1:3]), int(f.rsplit("_v", 1)[1].split(".")[0])))
    for mf in reversed(metas):
        with open(mf) as f:
            meta = json.load(f)
        if (meta.get("envs"), meta.get("rollout_len"), meta.get("chunk_rows")) != (envs, T, chunk_rows):
            continue
        if meta.get("kernel_sources") != _kernel_digest():   # a recording of other kernels says nothing about this run
            return dict(traffic=None, traffic_note=f"newest PMC recording ({os.path.basename(mf)}) was taken from other kernel "
                                                   f"sources ({meta.get('kernel_sources')} vs {_kernel_digest()}): not attached")
        total = 0.0
        with open(mf[:-5] + ".csv") as f:
            for r in csv.DictReader(f):
                if any(sub in r["kernel"] for sub in kernel_substr):
                    per_launch = float(r["FETCH_bytes_per_launch_corrected_x2"]) + float(r["WRITE_bytes_per_launch"])
                    total += per_launch * float(r["dispatches"]) / meta["steps_in_run"]
        return dict(traffic=round(total), traffic_unit="HBM bytes per step (fetch + write) over the same launches",
                    traffic_source=os.path.relpath(mf[:-5] + ".csv", here), traffic_commit=meta.get("commit"),
                    traffic_kernel_sources=meta.get("kernel_sources"))
    return dict(traffic=None, traffic_note="no PMC recording under profiles/ at this configuration")


def recorded_counters(envs, T, chunk_rows):
    """Matrix-pipe utilisation of the same kernel family from the newest SQ counter pass committed under profiles/
    (scripts/pmc_summary.py spells out the normalisation): busy cycles of every SIMD's matrix pipe over SIMDs x dispatch
    cycles, summed over the family's launches of one step.  Like `traffic`: a recorded figure, named with its source, only
    when the recording was made at this run's configuration."""
    import csv
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    metas = sorted(glob.glob(os.path.join(here, "profiles", "r*_sq_counters_v*.json")),
                   key=lambda f: (int(os.path.basename(f)[1:3]), int(f.rsplit("_v", 1)[1].split(".")[0])))
    for mf in reversed(metas):
        with open(mf) as f:
            meta = json.load(f)
        if (meta.get("envs"), meta.get("rollout_len"), meta.get("chunk_rows")) != (envs, T, chunk_rows):
            continue
        if meta.get("kernel_sources") != _kernel_digest():
            return dict(mfma_busy=None, counters_note=f"newest SQ counter recording ({os.path.basename(mf)}) was taken from other "
                                                      "kernel sources: not attached")
        busy = cycles = valu = wait = wave = 0.0
        with open(mf[:-5] + ".csv") as f:
            for r in csv.DictReader(f):
                if not any(sub in r["kernel"] for sub in ("gemm", "obs_fwd_bf16", "obs_bwd_bf16", "obs_fwd_h2", "obs_bwd_h2", "h2conv", "h2wgrad")) or not r["SQ_VALU_MFMA_BUSY_CYCLES"]:
                    continue
                n = float(r["dispatches"])
                busy += n * float(r["SQ_VALU_MFMA_BUSY_CYCLES"])
                cycles += n * float(r["cycles_per_dispatch"])
                valu += n * 4 * float(r["SQ_ACTIVE_INST_VALU"] or 0)
                wait += n * float(r["SQ_WAIT_ANY"] or 0)
                wave += n * float(r["SQ_WAVE_CYCLES"] or 0)
        if cycles <= 0:
            continue
        simds = float(meta.get("simds", 1024))
        return dict(mfma_busy=round(busy / (simds * cycles), 4), valu_busy=round(valu / (simds * cycles), 4),
                    wait_share=round(wait / wave, 4) if wave else None,
                    counters_source=os.path.relpath(mf[:-5] + ".csv", here), counters_commit=meta.get("commit"),
                    counters_kernel_sources=meta.get("kernel_sources"))
    return dict(mfma_busy=None, counters_note="no SQ counter recording under profiles/ at this configuration")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--seeds", default="0,1,2", help="data seeds of the resident leg (SURVEY 8d: 0, 1, 2); the first one feeds the other legs")
    ap.add_argument("--global-envs", type=int, default=GLOBAL_ENVS, help="B_global, fixed as N grows (strong scaling)")
    ap.add_argument("--rollout-len", type=int, default=128)
    ap.add_argument("--chunk-rows", type=int, default=16384)
    ap.add_argument("--graph", type=int, default=0, help="1: capture the update into a hipGraph and replay it (one rank only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-from-host", action="store_true",
                    help="skip the pinned-host-fed passes: `value` is then the resident-in-HBM figure")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the rollout-beside-update leg (`closed_loop`)")
    ap.add_argument("--rollout-groups", type=int, default=2,
                    help="closed loop: inference policies (policy workers) sharing the observation ring, each serving B / groups environments")
    ap.add_argument("--no-plain-copy", action="store_true", help="skip the pass without the observation ring (`from_pinned_host`)")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even with one rank")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the `configs` / `scaling_model` objects (the other BASELINE configurations and the per-rank shard)")
    ap.add_argument("--no-mlp", action="store_true", help="skip `roofline_mlp`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started directly: become the launcher.  The ranks are CHILD processes (torch.distributed.run), started before this
        # process has touched the GPU; their output (rank 0's JSON line) is relayed and their exit code returned.
        import socket
        import subprocess
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert args.global_envs % world == 0, "the global batch must split evenly over the ranks"
    assert torch.cuda.is_available(), "bench.py needs a GPU: the hot path has no CPU fallback"
    # SRL_BENCH_BACKEND=gloo lets the ranks share one GPU: only for tests/test_gpu_dist.py, which drives this script with
    # two ranks on a one-GPU box to check the multi-rank control flow (RCCL needs a device per rank)
    backend = os.environ.get("SRL_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = f"cuda:{dev_index}"

    import srl_amd
    from srl_amd import hip
    from srl_amd.api import config, policy as policy_api, trainer as trainer_api
    from srl_amd.namedarray import NamedArray, recursive_apply
    from srl_amd.runtime.ingest import SampleRing
    srl_amd.register_all()

    use_dist = world > 1 or args.force_dist
    ranks_seen = 1
    if use_dist:
        kw = dict(device_id=torch.device(device)) if backend == "nccl" else {}
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world, **kw)
        assert dist.get_world_size() == args.gpus, f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}"
        seen = torch.ones(1, dtype=torch.float32, device=device)
        dist.all_reduce(seen)  # every rank contributes 1 through the collective backend itself
        ranks_seen = int(round(float(seen.item())))
        assert ranks_seen == args.gpus, f"collective saw {ranks_seen} ranks, --gpus {args.gpus}"
    trainer = trainer_api.make(config.Trainer("mappo", args=dict(TRAINER, chunk_rows=args.chunk_rows, use_graph=bool(args.graph))),
                               config.Policy("actor-critic", args=POLICY))
    if use_dist:
        trainer.distributed(rank=rank, world_size=world, init_method="env://")

    T, B = args.rollout_len, args.global_envs // world

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    step_marks = []  # HIP events between the timed steps of the last `timed` call (no synchronisation inside the region)

    def timed(fn, warmup, steps):
        for _ in range(warmup):
            fn()
        sync()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            r = fn()
        marks[steps].record()
        sync()
        el = time.perf_counter() - t0
        step_marks[:] = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, r

    def rate(el, steps):
        return T * B * world * steps / el

    # ---- (1) the update with the whole sample resident in HBM, data seeds 0, 1, 2 ---------------------------------
    # one sample at a time (14.9 GB of frames each); the last one (seed 0 + rank offset) stays for the legs below
    seeds = [int(s) for s in args.seeds.split(",")] if args.seeds else [0]
    per_seed, all_marks, el_res, res = {}, [], 0.0, None
    sample = None
    for i, sd in enumerate(reversed(seeds)):   # ... so that seeds[0] is the sample the remaining legs run on
        sample = None
        torch.cuda.empty_cache()
        sample = device_sample(1000 * sd + rank, T, B, device)
        el, res = timed(lambda: trainer.step(sample), args.warmup if i == 0 else 2, args.steps)
        ms = sorted(step_marks)
        per_seed[str(sd)] = dict(ms_per_step=1e3 * el / args.steps, ms_per_step_median=ms[len(ms) // 2], ms_per_step_min=ms[0])
        all_marks += ms
        el_res += el
    marks_res = sorted(all_marks)
    n_res = args.steps * len(seeds)
    resident = dict(value=rate(el_res, n_res), unit="env-steps/s", ms_per_step=1e3 * el_res / n_res, steps=n_res,
                    warmup=args.warmup, ms_per_step_median=marks_res[len(marks_res) // 2], ms_per_step_min=marks_res[0],
                    data_seeds=seeds, per_seed=per_seed,
                    note="whole sample, frames as a plain [Tb, B, 4, 84, 84] uint8 tensor, on the device before the timed region; "
                         "median / min over the timed updates of every seed")

    # ---- untimed extra step with per-kernel HIP events (same stream as the launches) -----------------------
    roofline = roofline_gae = roofline_mlp = breakdown = launches_per_step = None
    if not args.no_profile:
        # EVERY rank takes this step (its collectives need all of them); only rank 0 wraps its launches in events
        prof = hip.KernelProfile() if rank == 0 else None
        if prof is not None:
            hip.set_profile(prof)
        trainer.use_graph = False  # the events wrap individual launches: this step is issued launch by launch
        # ... and one launch at a time: the timed steps run two row-chunk pipelines and the weight gradients on streams of
        # their own, where a kernel's duration includes what it shares the chip with; the roofline is a per-kernel figure
        pipes, trainer.pipelines = trainer.pipelines, 1
        side, trainer.policy.net._wgrad_side = trainer.policy.net._wgrad_side, False
        hip.dispatch_counts(reset=True)
        trainer.step(sample)
        dispatch = hip.dispatch_counts(reset=True)
        trainer.pipelines, trainer.policy.net._wgrad_side = pipes, side
        if prof is not None:
            hip.set_profile(None)
    if rank == 0 and not args.no_profile:
        summ = prof.summary()
        # every matrix-core launch group.  The contractions deliver float32 results from float32 operands but run on the
        # bf16 / f16 matrix cores: each float32 operand is split into three bf16 pieces and the six leading piece
        # products are formed (gemm_bf16x3.h) -- two f16 pieces and three products for the forward products whose operand
        # ranges are known, three products for the first layer whose frames are bytes (obs_bf16.h).  `achieved` counts
        # ALGORITHMIC float32 flops; `frac` prices the 16-bit matrix-core flops actually issued (hip.piece_products per
        # launch) against the bf16 / f16 peak.
        mm = {k: v for k, v in summ.items() if k == "gemm" or k.startswith("conv_")}
        g = dict(calls=sum(v["calls"] for v in mm.values()), ms=sum(v["ms"] for v in mm.values()),
                 work=sum(v["work"] for v in mm.values()), executed=sum(v["executed"] for v in mm.values()))
        ach = g["work"] / (g["ms"] * 1e-3) / 1e12
        exe = g["executed"] / (g["ms"] * 1e-3) / 1e12
        roofline = dict(kernel="h2conv_kernel / h2wgrad_kernel / h2gemm_kernel / h2tn_kernel (Linear weight gradient) / obs_fwd_h2_kernel / "
                               "obs_bwd_h2_kernel: every contraction of one step (image-stationary convolutions on pre-split f16 "
                               "activations, the Linear's three products, the first layer on bytes): float32 operands and results "
                               "through 16-bit piece products on the f16 / bf16 matrix cores",
                        bound="mfma", achieved=round(ach, 2), unit="TFLOP/s", peak=PEAK_BF16_MFMA_TFLOPS,
                        executed=round(exe, 1), frac=round(ach / PEAK_BF16_MFMA_TFLOPS, 4),
                        frac_basis="achieved (algorithmic float32 TFLOP/s) / peak of the pipe the work runs on (16-bit MFMA, dense)",
                        mfma_issue_frac=round(exe / PEAK_BF16_MFMA_TFLOPS, 4),
                        frac_of_emulation_ceiling=round(ach / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4),
                        emulation_ceiling_basis="16-bit peak / 3: a float32 multiply-add costs at least three piece products",
                        achieved_basis="algorithmic float32 flops (2*M*N*K of every contraction)",
                        executed_basis="16-bit MFMA flops issued: 3 x algorithmic for products of two f16 pieces per operand (every "
                                       "convolution and Linear product), 2 x for the first layer (bytes x two f16 pieces of the folded "
                                       "weights forward, of dz' in the weight gradient), 6 x where an operand range is unknown (three "
                                       "bf16 pieces per operand)",
                        fp32_mfma_peak=PEAK_FP32_MFMA_TFLOPS, achieved_over_fp32_mfma_peak=round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
                        launches=g["calls"], ms_per_step=round(g["ms"], 3), flops_per_step=g["work"],
                        flops_per_env_step=g["work"] / (T * B), kernel_family_launches=dispatch,
                        **recorded_traffic(("gemm", "obs_fwd_bf16", "obs_bwd_bf16", "obs_fwd_h2", "obs_bwd_h2", "h2conv", "h2wgrad"), B, T, args.chunk_rows),
                        **recorded_counters(B, T, args.chunk_rows))
        # The same launches against the OTHER roof.  With float32 activations in HBM the convolution layers carry 60-75
        # algorithmic flop per byte, below the ~104 flop / byte at which 8 TB/s feed three piece products per multiply-add at
        # the 16-bit peak: by bytes they sit under the HBM roof, and the recorded traffic over this run's kernel time says
        # how far under (DESIGN section 5 has the per-layer floors).
        if roofline.get("traffic"):
            gbs_fam = roofline["traffic"] / (g["ms"] * 1e-3) / 1e9
            roofline["hbm"] = dict(achieved=round(gbs_fam, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs_fam / PEAK_HBM_GBS, 4),
                                   basis="recorded HBM bytes of these launches (PMC passes) / their summed durations in this run")
        # the scan is a ~microsecond kernel: time it as back-to-back launches between two events on the launch
        # stream so that host enqueue latency does not sit inside the interval
        floor = launch_floor_us(device)
        s = gae_microbench(sample, TRAINER, device)
        gbs = s["work"] / (s["ms"] * 1e-3) / 1e9
        big = gae_microbench(sample, TRAINER, device, reps=30, big_B=1 << 20)
        big_gbs = big["work"] / (big["ms"] * 1e-3) / 1e9
        roofline_gae = dict(kernel="gae_scan_reg_kernel", bound="hbm", achieved=round(gbs, 2), peak=PEAK_HBM_GBS,
                            unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 5), traffic=None, envs=B, rollout_len=T,
                            us_per_launch=round(s["ms"] * 1e3, 2), algorithmic_bytes=s["work"],
                            launch_floor_us=round(floor, 2), t_over_launch_floor=round(s["ms"] * 1e3 / floor, 2),
                            note=f"this step's own [T, {B}] leaves: {s['work'] / 1e6:.1f} MB, launch-latency bound "
                                 "(t / t_launch_floor), see 'saturated' for the HBM-bound size",
                            saturated=dict(envs=1 << 20, rollout_len=T, achieved=round(big_gbs, 1),
                                           frac=round(big_gbs / PEAK_HBM_GBS, 4),
                                           us_per_launch=round(big["ms"] * 1e3, 1), algorithmic_bytes=big["work"]))
        breakdown = {k: round(v["ms"], 3) for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
        launches_per_step = sum(v["calls"] for v in summ.values())
        if not args.no_mlp:
            try:
                roofline_mlp = mlp_roofline(device)
            except Exception as e:  # the secondary figure must not take the benchmark down
                roofline_mlp = dict(error=repr(e))

    # ---- (2) the sample moves to pinned host memory: two slots of the ingest ring, [Tb, B] namedarray layout, wire dtypes ---
    pcie = fed = rollout_inf = last_stamps = None
    ms_step = 1e3 * el_res / n_res
    if not args.no_from_host:
        Tb = T + 1
        template = recursive_apply(sample[:, 0], lambda x: x.cpu().numpy())
        ring = SampleRing(template, batch_size=B, slots=2, device=device)
        for _ in range(2):
            ring.put_batch(sample)  # device leaves -> the slot's pinned blocks (D2H, untimed)
        del sample
        torch.cuda.empty_cache()

        def fed_step():
            b = ring.get_device()  # waits (stream-side) on this batch's copies, starts the next batch's
            r = trainer.step(b)
            slot = b.metadata["ring_slot"]
            ring.release(slot)
            ring.recycle(slot)
            return r

        # ---- (2a) no observation ring: every leaf, frames included, crosses the link inside the timed region ---------------
        if not args.no_plain_copy:
            el, _ = timed(fed_step, 2, args.steps)
            host_bytes = ring.nbytes() // 2
            h2d_ms = None
            if rank == 0:  # the copy alone, same ring, nothing else running
                b = ring.get_device()
                slot = b.metadata["ring_slot"]
                ring.release(slot)
                ring.recycle(slot)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                b2 = ring.get_device()
                torch.cuda.synchronize()
                h2d_ms = 1e3 * (time.perf_counter() - t0)
                slot = b2.metadata["ring_slot"]
                ring.release(slot)
                ring.recycle(slot)
                del b, b2
            pcie = dict(value=rate(el, args.steps), unit="env-steps/s", ms_per_step=1e3 * el / args.steps, steps=args.steps,
                        warmup=2, host_bytes_per_step_per_gpu=host_bytes, h2d_alone_ms=h2d_ms,
                        h2d_alone_GBps=None if not h2d_ms else round(host_bytes / h2d_ms / 1e6, 1),
                        bound="pcie" if h2d_ms and h2d_ms > ms_step else "mfma",
                        note="no observation ring: async H2D of every leaf (14.9 GB of frames) on a side stream inside the "
                             "timed region, double-buffered so that the copy of update k+1 runs under the compute of update k")

        # ---- (2b) the rollout phase of this sample: 129 inference batches of B observations from pinned host memory, each
        # staged once in the HBM observation ring (space-to-depth + LayerNorm statistics, the pass inference needs anyway).
        # Twice: whole stacks per request (what actor_critic_policy.py:467-469 uploads), and stack-aware requests -- the
        # newest plane + the stamp of the environment's previous observation, the ring assembles the row (ObsRing.put_stacked):
        # a quarter of the bytes over the link, the same actions / log-probabilities / values (checked here on every batch).
        frames = ring.host_tensors(0)["obs.obs"]  # [Tb, B, 4, 84, 84] uint8, pinned (torch tensor: asynchronous DMA)
        fresh = ring.host_blocks(0)["on_reset"].reshape(Tb, B).astype(bool).copy()
        fresh[0] = True
        planes = torch.empty((Tb, B, 1, 84, 84), dtype=torch.uint8).pin_memory()
        planes.copy_(frames[:, :, 3:4])

        def request(obs, n=B):
            zeros_n = lambda dt: np.zeros((n, 1), dt)
            return policy_api.RolloutRequest(obs=obs, is_evaluation=zeros_n(np.uint8), on_reset=zeros_n(np.uint8), client_id=zeros_n(np.int32),
                                             request_id=np.arange(n).reshape(n, 1), received_time=zeros_n(np.int64),
                                             buffer_index=zeros_n(np.int32))

        def new_inference_policy(capacity, ring=None):
            pol = policy_api.make(config.Policy("actor-critic", args=POLICY))
            pol.load_checkpoint(trainer.policy.get_checkpoint())
            pol.attach_obs_ring(pol.make_obs_ring(capacity, patch_rows=4 * B) if ring is None else ring)
            return pol

        def rollout_phase(pol, stacked, keep=None, check=None, prev=None, cols=(0, B)):
            """Tb ticks over the env columns [cols[0], cols[1]); returns (stamps [Tb, columns, 1], seconds per tick after three
            warm-up ticks, last stamps)."""
            c0, c1 = cols
            nb = c1 - c0
            stamps = np.empty((Tb, nb, 1), np.int64)
            prev = np.zeros((nb, 1), np.int64) if prev is None else prev
            t_roll, n_roll = 0.0, 0
            for t in range(Tb):
                if stacked:
                    prev = np.where(fresh[t][c0:c1, None], 0, prev)
                    req = request(NamedArray(obs=planes[t][c0:c1], ring_prev=prev), nb)
                else:
                    req = request(NamedArray(obs=frames[t][c0:c1]), nb)
                t0 = time.perf_counter()
                resp = pol.rollout(req)  # returns numpy: synchronises
                if t >= 3:
                    t_roll += time.perf_counter() - t0
                    n_roll += 1
                out = (resp.action.x, resp.analyzed_result.log_probs, resp.analyzed_result.value)
                if keep is not None:
                    keep.append(out)
                if check is not None and not all(np.array_equal(x, y) for x, y in zip(out, check[t])):
                    raise RuntimeError(f"stack-aware rollout differs from the whole-stack rollout at tick {t}")
                stamps[t] = prev = resp.analyzed_result.obs_ref
            return stamps, t_roll / max(n_roll, 1), prev

        whole, whole_out = new_inference_policy(Tb * B + 8 * B), []
        _, s_whole, _ = rollout_phase(whole, False, keep=whole_out)
        del whole
        torch.cuda.empty_cache()
        # the ring the updates below read: room for the sample being trained on AND the one the next rollout phase writes
        infer = new_inference_policy(2 * Tb * B + 8 * B)
        obs_ring = infer._obs_ring
        stamps, s_stack, last_stamps = rollout_phase(infer, True, check=whole_out)
        del whole_out
        for slot in range(2):  # the actors store the stamp with each step; both slots hold this sample
            ring.host_blocks(slot)["analyzed_result.obs_ref"][...] = stamps
        rollout_inf = dict(value=B / s_stack, unit="requests/s", requests_per_call=B, ms_per_call=1e3 * s_stack, calls=Tb - 3,
                           h2d_bytes_per_request=84 * 84 + 8,
                           note="policy.rollout on pinned host observations, sampled actions; stack-aware requests: the newest "
                                "uint8 plane + the ring stamp of the environment's previous observation, the row assembled in the "
                                "HBM observation ring (srl_ring_stack_push); H2D of the planes and D2H of the results inside",
                           identical_to_whole_stack=True,
                           whole_stack=dict(value=B / s_whole, unit="requests/s", ms_per_call=1e3 * s_whole,
                                            h2d_bytes_per_request=4 * 84 * 84,
                                            note="the same batches with the whole (4, 84, 84) stack per request"))
        ring.attach_obs_ring(obs_ring)

        # ---- (2c) SURVEY 8d's t_update with the ring: the headline ---------------------------------------------------------
        el_fed, res = timed(fed_step, args.warmup, args.steps)
        marks_fed = sorted(step_marks)
        ms_step = 1e3 * el_fed / args.steps
        scalar_bytes = sum(v.nbytes for k, v in ring.host_blocks(0).items() if k != "obs.obs")
        fed = dict(value=rate(el_fed, args.steps), ms_per_step=ms_step, ms_per_step_median=marks_fed[len(marks_fed) // 2],
                   ms_per_step_min=marks_fed[0], h2d_bytes_per_step_per_gpu=scalar_bytes,
                   frames_bytes_kept_in_hbm=int(frames.numel()), obs_ring=dict(capacity_rows=obs_ring.capacity,
                                                                               patch_rows=obs_ring.patch_capacity,
                                                                               bytes=obs_ring.nbytes(), **obs_ring.stats))

    # ---- (3) closed loop on this GPU: the rollout phase of sample k+1 (stack-aware requests, its own thread and stream) runs
    # WHILE the update on sample k runs; a sample is trained on once, with the stamps its own rollout phase returned ---------
    closed = None
    if fed is not None and not args.no_closed_loop and not use_dist:
        import threading
        if os.environ.get("SRL_SWITCH_INTERVAL"):
            sys.setswitchinterval(float(os.environ["SRL_SWITCH_INTERVAL"]))
        b0 = ring.get_device()  # one slot checked out and released (not recycled): the "next sample" slot
        nxt = b0.metadata["ring_slot"]
        ring.release(nxt)
        del b0
        # The rollout side as the reference runs it: the actors' environments in GROUPS (the env ring, actor_worker.py:634-748: one
        # group steps while another group's inference requests are in flight), served by `groups` inference policies (same
        # parameters, own executor and stream, ONE shared observation ring) from one host thread through `rollout_async`: one
        # group's H2D of planes and D2H of results cross the link while another group's kernels run.  (Two THREADS calling the
        # synchronous `rollout` made it slower, 209 -> 248 ms per iteration: the host side of a tick is what it is, twice.)
        G = max(1, int(args.rollout_groups))
        while B % G:
            G -= 1
        infers = [infer] + [new_inference_policy(0, ring=obs_ring) for _ in range(G - 1)]
        roll_streams = [torch.cuda.Stream(device=device, priority=-1) for _ in range(G)]  # short inference kernels go ahead of the update's queue
        gcols = [(g * (B // G), (g + 1) * (B // G)) for g in range(G)]
        state = dict(prev=[last_stamps[c0:c1] for c0, c1 in gcols], err=None, roll_s=[])

        def rollout_phase_groups(prevs):
            """Tb ticks of G groups from ONE host thread: a group's tick t + 1 is issued (`rollout_async`: H2D of its planes, network
            pass, sampling, D2H of the results -- all enqueued on the group's stream) as soon as its tick t has come back, while the
            other groups' ticks are in flight.  Returns (stamps [Tb, B, 1], last stamps per group)."""
            stamps = np.empty((Tb, B, 1), np.int64)
            pending = [None] * G
            prevs = list(prevs)
            for t in range(Tb + 1):
                for g in range(G):
                    c0, c1 = gcols[g]
                    if pending[g] is not None:
                        resp = pending[g].result()
                        stamps[t - 1, c0:c1] = prevs[g] = resp.analyzed_result.obs_ref
                        pending[g] = None
                    if t < Tb:
                        prev = np.where(fresh[t][c0:c1, None], 0, prevs[g])
                        with torch.cuda.stream(roll_streams[g]):
                            pending[g] = infers[g].rollout_async(request(NamedArray(obs=planes[t][c0:c1], ring_prev=prev), c1 - c0))
            return stamps, prevs

        def produce(slot):
            try:
                t0 = time.perf_counter()
                st, state["prev"] = rollout_phase_groups(state["prev"])
                state["roll_s"].append(time.perf_counter() - t0)
                ring.host_blocks(slot)["analyzed_result.obs_ref"][...] = st
                ring.recycle(slot)  # complete: every column of the slot carries the new stamps
            except BaseException as e:  # surfaced by the main thread
                state["err"] = e

        def closed_iteration():
            nonlocal nxt
            th = threading.Thread(target=produce, args=(nxt,))
            th.start()
            bt = ring.get_device()
            r = trainer.step(bt)
            cur = bt.metadata["ring_slot"]
            ring.release(cur)
            th.join()
            if state["err"] is not None:
                raise state["err"]
            nxt = cur
            return r

        n_closed = max(2, min(args.steps, 8))
        el_c, _ = timed(closed_iteration, 1, n_closed)
        closed = dict(value=rate(el_c, n_closed), unit="env-steps/s", ms_per_iteration=1e3 * el_c / n_closed, iterations=n_closed,
                      rollout_phase_ms=round(1e3 * float(np.mean(state["roll_s"][1:])), 1),
                      rollout_ticks_per_iteration=Tb, requests_per_tick=B, rollout_groups=G,
                      update_alone_ms=round(ms_step, 2), obs_ring=dict(capacity_rows=obs_ring.capacity, **obs_ring.stats),
                      note="one iteration = the update on sample k (ring-fed, as the headline) with the whole rollout phase of sample "
                           "k+1 (Tb stack-aware inference batches from pinned host memory, results back to the host) running beside "
                           "it on the same GPU; every sample is trained on with the stamps of its own rollout phase.  The rollout side "
                           f"is {G} group(s) of {B // G} environments, each with its inference policy and stream on the one shared "
                           "observation ring, issued from one host thread (rollout_async): a group's link transfers run under the "
                           "other groups' kernels")
        ring.recycle(nxt)

    if rank == 0:
        head = fed if fed is not None else resident
        line = dict(metric=f"env-steps/sec through GAE+PPO update, {B * world} envs x {T} steps", value=head["value"],
                    unit="env-steps/s", n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=head["ms_per_step"],
                    higher_is_better=True, scaling="strong", vs_baseline=None,
                    dtype="f32 (contractions as bf16x3 / f16x2 piece products, float32 accumulate)", data="synthetic",
                    ms_per_step_median=head["ms_per_step_median"], ms_per_step_min=head["ms_per_step_min"],
                    value_basis=("SURVEY 8d t_update: sample in pinned host memory ([Tb,B] namedarray layout, wire dtypes) when "
                                 "an update starts; every leaf copied inside the timed region except the frames, which the "
                                 "rollout already left in the HBM observation ring (bound by the sample's stamps); see "
                                 "resident_in_hbm / from_pinned_host for the two other feeds" if fed is not None else
                                 "sample resident in HBM when the timed region starts (--no-from-host)"),
                    config=dict(workload=f"BASELINE configs[2]: Atari-shaped PPO+GAE, {B * world} envs x {T} steps per update "
                                         f"(global batch fixed; {B} env columns per GPU x {world} GPUs data-parallel), "
                                         "NatureCNN-512, uint8 (4,84,84) frames, Atari PPO preset",
                                envs_per_gpu=B, rollout_len=T, global_envs=B * world, parallelism=f"dp{world}",
                                chunk_rows=args.chunk_rows, collective_ranks=ranks_seen,
                                collectives=("none (one rank)" if not use_dist else
                                             "RCCL through the C ABI (srl_comm_*)" if getattr(trainer, "_comm", None) is not None
                                             else f"torch.distributed ({backend})"),
                                pipelines=getattr(trainer, "pipelines", None),
                                grad_buckets=(None if getattr(trainer, "_reducer", None) is None else
                                              dict(trainer._reducer.stats, buckets=len(trainer._reducer.buckets))),
                                policy_loss=res.stats.get("policy_loss")),
                    roofline=roofline, roofline_gae=roofline_gae, roofline_mlp=roofline_mlp, kernel_ms_per_step=breakdown,
                    launches_per_step=launches_per_step,
                    resident_in_hbm=resident)
        if fed is not None:
            line["ring_fed"] = {k: v for k, v in fed.items() if k not in ("value", "ms_per_step", "ms_per_step_median", "ms_per_step_min")}
        if pcie is not None:
            line["from_pinned_host"] = pcie
        if rollout_inf is not None:
            line["rollout_inference"] = rollout_inf
        if closed is not None:
            line["closed_loop"] = closed
        if world == 1 and not args.no_configs and not use_dist:
            # the other BASELINE configurations and the per-rank shard, on the same box in the same run (this process's big buffers
            # are released first: the child and the football leg want the memory)
            ring = infer = obs_ring = sample = None
            trainer.policy.net.ws._bufs.clear()
            torch.cuda.empty_cache()
            line["configs"], line["scaling_model"] = other_configs(line, device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(T)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

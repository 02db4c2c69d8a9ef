#!/usr/bin/env python3
"""bench.py — env-steps/s of the many-instance Ant step on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one closed-loop environment step of every resident env, ALL of step_forward_original's work and of
VectorizedEnvironment::step's on top of it, EVERY step: fresh action batch (already in HBM) -> PD, forward dynamics,
plane contacts, MLCP/PGS, Euler -> output packing (q, qd, visual poses, up.z, padding: the y record) + reward / done +
observation, each step's y record and [obs | reward | done] record stored into its slot of a ring in HBM
(tds_hip_step_many_rings).  The K timed steps are ONE launch of the step-loop kernel where the library has that form
(Ant up to three rounds of workgroups, worlds without contacts), chained hipGraphs of single-step launches elsewhere —
the work per step is the same in both.  The form that skips the records of all but the last step of a launch
(tds_hip_step_many: a substep-fused figure) is timed afterwards and reported under the secondary key
"substep_fused", never as `value`.
With N > 1 ranks each GPU owns its own shard of environments and runs the SAME launches (same rings); the
[obs | reward | done] records of every policy step reach every rank through the library's own shard layer
(tds_hip_shard_step_many): by default the step-loop launch itself stores each record into the gathered ring of EVERY rank
(IPC-mapped peer memory over xGMI: no collective, no host call and no kernel boundary per step), else librccl's all-gather
called from C.  N = 1 and N > 1 run the same kernel and differ by the exchange only.  --gpus 8 runs BASELINE config 5
(8192 environments per GPU = 65 536) as `value` and the 4096-per-GPU line beside it (`envs_4096_per_gpu`).

Prints ONE JSON line on rank 0 (contract in the project brief): value = total env-steps / s
over all GPUs, plus `roofline` (algorithmic bytes / measured kernel time vs 8 TB/s HBM) and
`cpu_baseline` (the checker libraries timed on the host cores; reported, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(model_name, n_envs, budget_s=12.0):
    """Checker libraries as the CPU baseline, rank 0 only, bounded sample.
    kind "reference": oracle/_ref/libtds_ref.so — the reference's own header-only double path,
                      one simulation object per thread.
    kind "port":      oracle/libtds_oracle.so — the plain-C restatement with OpenMP."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tds_amd
    import oraclelib

    g = np.load(os.path.join(ROOT, "tests", "golden", model_name + ".npz"))
    m = tds_amd.load_model(model_name)
    rng = np.random.default_rng(0)
    x = g["x"][rng.integers(0, g["x"].shape[0], n_envs)]
    cores = os.cpu_count() or 1
    out = {}
    # port: OpenMP over envs
    threads = min(cores, oraclelib.max_threads())
    oraclelib.step(m, x[:64], threads=threads)
    t0 = time.perf_counter()
    reps = 0
    while True:
        oraclelib.step(m, x, threads=threads)
        reps += 1
        if time.perf_counter() - t0 > budget_s / 2 or reps >= 50:
            break
    dt = time.perf_counter() - t0
    out["port"] = {"value": reps * n_envs / dt, "unit": "env-steps/s", "cores": threads, "kind": "port",
                   "sample": f"{reps} x {n_envs} {model_name} steps, oracle/tds_oracle.c, OpenMP"}
    # reference: one RefSim per thread (python threads; ctypes releases the GIL)
    try:
        import reflib
        if reflib.available():
            import threading
            nth = min(cores, 64)
            sims = [reflib.RefSim(model_name) for _ in range(nth)]
            chunk = max(1, min(256, n_envs // nth))
            counts = [0] * nth
            stop = time.perf_counter() + budget_s / 2

            def work(i):
                xs = x[(i * chunk) % n_envs:(i * chunk) % n_envs + chunk]
                while time.perf_counter() < stop:
                    sims[i].step(xs)
                    counts[i] += xs.shape[0]

            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(i,)) for i in range(nth)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            out["reference"] = {"value": sum(counts) / dt, "unit": "env-steps/s", "cores": nth, "kind": "reference",
                                "sample": f"{sum(counts)} {model_name} steps of step_forward_original "
                                          f"(header-only double path, one sim per thread) in {dt:.1f}s"}
            # B2: the reference's committed generated kernels (what its OpenMPForwardStepper runs)
            if model_name in ("ant", "laikago") and hasattr(reflib.lib(), "tdsref_generated_step"):
                counts = [0] * nth
                stop = time.perf_counter() + budget_s / 3

                def work2(i):
                    xs = x[(i * chunk) % n_envs:(i * chunk) % n_envs + chunk]
                    while time.perf_counter() < stop:
                        reflib.generated_step(model_name, xs, m.output_dim)
                        counts[i] += xs.shape[0]

                t0 = time.perf_counter()
                ths = [threading.Thread(target=work2, args=(i,)) for i in range(nth)]
                [t.start() for t in ths]
                [t.join() for t in ths]
                dt = time.perf_counter() - t0
                out["generated"] = {"value": sum(counts) / dt, "unit": "env-steps/s", "cores": nth,
                                    "kind": "reference",
                                    "sample": f"{sum(counts)} {model_name} steps of the reference's committed "
                                              f"omp_model_{model_name}_forward_zero_kernel in {dt:.1f}s"}
    except Exception as e:  # the real-reference library is optional on the GPU box
        out["reference_error"] = repr(e)
    return out


# flops per env-step counted on the reference's own (dense) formulation, SURVEY.md 8(d)
ALG_FLOPS = {"ant": 1.5e5, "laikago": 7e4, "laikago_soft": 7e4, "pendulum5": 5e3, "cartpole": 2e3}


def pmc_traffic(model, n, dtype):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json; PMC
    counters cannot be collected from inside this process).  Guide corrections: FETCH_SIZE / WRITE_SIZE
    are KiB; on gfx950 FETCH_SIZE counts coalesced streaming reads at half their bytes -> doubled
    (an upper bound for our 8-byte-per-lane loads)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f)[model][str(n)][dtype]
        return int((2.0 * e["fetch_kib"] + e["write_kib"]) * 1024), e["source"]
    except Exception:
        return None, None


def sq_counters(model, n, dtype):
    """VALU issue fraction of the dominant kernel from the committed SQ-counter pass (profiles/sq_counters.json: VALU
    instructions per launch x 4 cycles / (SIMDs x launch cycles) — see the file's _comment)."""
    try:
        with open(os.path.join(ROOT, "profiles", "sq_counters.json")) as f:
            return json.load(f)[model][str(n)][dtype]
    except Exception:
        return {}


def pmc_traffic_loop(model, n, dtype, steps_per_launch):
    """The same for ONE step-loop launch of steps_per_launch steps: measured on a 500-step launch (its traffic is the
    action block per step plus the records once: linear in the steps to within the records)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f)[model][str(n)][dtype]["loop"]
        per_step = 2.0 * e["fetch_kib_per_step"] + e["write_kib_per_step"]
        return int(per_step * steps_per_launch * 1024), e["source"]
    except Exception:
        return None, None


def pmc_traffic_rings(model, n, dtype, steps_per_launch):
    """HBM bytes of ONE launch of the per-step-record form (tds_hip_step_many_rings): FETCH_SIZE of the whole launch (the x
    records, the model, the action pool once — it stays cache-resident) doubled per the guide's gfx950 correction, plus
    WRITE_SIZE per step x the steps of the launch — from the committed PMC passes of the driver's own command
    (`python bench.py --steps 20 --warmup 5`: "short") and of a long launch."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            e = json.load(f)[model][str(n)][dtype]["rings"]
        w = e["write_kib_per_step_short" if steps_per_launch <= 64 else "write_kib_per_step_long"]
        fetch = e["fetch_kib_per_launch"]
        # (write-back record stores: a long launch also fetches — partial lines the L2 completes before it writes them back)
        if steps_per_launch > 64 and "fetch_kib_per_step_long" in e:
            fetch = e["fetch_kib_per_step_long"] * steps_per_launch
        return int((2.0 * fetch + w * steps_per_launch) * 1024), e["source"]
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs-per-gpu", type=int, default=0,
                    help="environments per GPU; 0 (default): BASELINE.json's configurations — 4096 (config 3, the one the "
                         "metric is quoted on) at 1 / 2 / 4 GPUs, 8192 at 8 GPUs (config 5: 65 536 environments sharded "
                         "8 x), where the 4096-per-GPU line is reported beside it under `envs_4096_per_gpu`")
    ap.add_argument("--model", default="ant")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "f32-pure"],
                    help="f64: double arithmetic, double records (the reference's arithmetic; headline).  f32: FLOAT "
                         "records (x, y, actions, obs) with the arithmetic in double registers — the variant that meets "
                         "the 1e-6 contract on float records (BASELINE config 2).  f32-pure: float arithmetic "
                         "(measured only: misses 1e-6, like the reference's own float instantiation)")
    ap.add_argument("--no-graph", action="store_true",
                    help="N = 1: K eager single-step launches (tds_hip_step_obs); N > 1: one tds_hip_shard_step call per step")
    ap.add_argument("--records", choices=["rings", "last"], default="rings",
                    help="rings (default): every step packs and stores its y and [obs|reward|done] records into ring slots "
                         "(tds_hip_step_many_rings) — the per-step protocol of the reference's metric loop; last: only the "
                         "last step of a launch does (tds_hip_step_many: the substep-fused form, secondary)")
    ap.add_argument("--ring-slots", type=int, default=64, help="slots of the two record rings (a slot is reused that many steps later)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE",
                    help="library option for every handle of this run (tds_hip_default_option), e.g. loop_w2=0, shard_wait=1")
    ap.add_argument("--y-stride", default="auto", choices=["auto", "line", "packed"],
                    help="record stride of the y ring: padded to whole 128-byte lines, or output_dim; auto (default): lines where the "
                         "padding is under 10 % of the record")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary measurements of the default N = 1 line (substep_fused, one_rank_with_exchange, auto_reset_rate)")
    ap.add_argument("--shard-graph", action="store_true",
                    help="N > 1: replay each step-loop launch + its exchanges from one hipGraph (TDS_HIP_SHARD_GRAPH=1) instead of "
                         "submitting the exchange eagerly (the default: faster on ROCm 7)")
    ap.add_argument("--step-many-form", choices=["auto", "graph", "loop"], default="auto",
                    help="N = 1: how tds_hip_step_many runs the K steps — chained hipGraphs of single-step launches, ONE "
                         "launch of the step-loop kernel, or the library's choice (loop for worlds without contacts and "
                         "for narrow kernels up to three rounds of workgroups)")
    ap.add_argument("--chains", default="auto",
                    help="N = 1, graph launches: environment chains of the graph (tds_hip_step_many in tds_hip.h): a number, "
                         "'default' (the library's rule) or 'auto' = measured during warm-up (tds_hip_step_many_tune)")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per env (16/32/64), 0 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spin-up-steps", type=int, default=6000,
                    help="untimed steps of a SCRATCH handle (same model, same batch, its own state) before the warm-up: a "
                         "fresh process has run well under a millisecond of kernels by then and a short timed region "
                         "(--steps 20) would otherwise be measured while the GPU leaves its idle clocks; 0 = off")
    ap.add_argument("--auto-reset", action="store_true",
                    help="secondary workload (N = 1): auto_reset_when_done on — every step resets the environments it "
                         "ends with done (reset distribution + settle steps), through the reset pool")
    ap.add_argument("--no-auto-reset", action="store_true",
                    help="the legged robots other than the Ant (Laikago: BASELINE config 4) run with auto_reset_when_done ON by "
                         "default — +-0.4 rad actions with the fallen robots reset, the reference's metric loop (SURVEY 8d); this "
                         "switch gives the round-5 line instead: +-0.1 rad actions, nothing reset")
    ap.add_argument("--action-amp", type=float, default=-1.0,
                    help="amplitude of the uniform random actions; default: 0.4 (the reference's ACTION_LIMIT, SURVEY 8d) for the "
                         "Ant and for every model when --auto-reset is on (fallen robots are reset, as in the reference's metric "
                         "loop), 0.1 for the legged robots that can fall over when nothing resets them")
    ap.add_argument("--rollout-steps", type=int, default=0,
                    help="also time the on-device policy rollout (tds_hip_rollout) with this many policy steps per call")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP events (pure wall clock)")
    ap.add_argument("--stream", choices=["own", "default"], default="default",
                    help="own: everything runs on a stream of its own (torch.cuda.Stream) instead of the NULL stream")
    ap.add_argument("--gather-every", type=int, default=1,
                    help="N > 1: steps whose [obs|reward|done] records travel in one all-gather (1 = one exchange "
                         "per policy step, SURVEY 8e's protocol and the default)")
    ap.add_argument("--pipelined-block", type=int, default=0,
                    help="N > 1: also time the pipelined form with this many steps per exchange (0 = skip)")
    ap.add_argument("--gather-dtype", choices=["f32", "f64"], default="f32",
                    help="dtype the [obs | reward | done] records cross xGMI in (the step itself stays in --dtype): "
                         "f32 = 4 B per scalar as SURVEY 8e sizes the exchange (default), same / f64 = as computed")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the N > 1 step loop (pipelined record gather) even with one rank: exercises that code path on a 1-GPU box")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    # the native library normally travels prebuilt with the tree; if it is missing, local rank 0
    # builds it (hipcc, ~1 min) while the other ranks wait for the file to appear
    lib_path = os.path.join(ROOT, "tiny-differentiable-simulator_amd", "libtds_hip.so")
    if not os.path.exists(lib_path):
        if local_rank == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            t_wait = time.time()
            while not os.path.exists(lib_path) and time.time() - t_wait < 900:
                time.sleep(2.0)
            time.sleep(5.0)

    import tds_amd
    from tds_amd import hip_backend

    # library options of every handle this run creates (tds_hip_default_option: the option table of csrc/tds_options.h;
    # environment variables TDS_HIP_<KEY> remain the defaults underneath)
    if args.step_many_form != "auto":
        hip_backend.default_option("step_many_loop", 1 if args.step_many_form == "loop" else 0)
    if args.shard_graph:
        hip_backend.default_option("shard_graph", 1)
    for kv in args.option:
        k, _, v = kv.partition("=")
        hip_backend.default_option(k, int(v))

    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    # debugging aid for 1-GPU boxes: TDS_BENCH_ONE_DEVICE=1 TDS_BENCH_BACKEND=gloo runs the N > 1 code
    # path with every rank on cuda:0 (RCCL refuses two ranks on one device); never used for numbers
    if os.environ.get("TDS_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.stream == "own":
        torch.cuda.set_stream(torch.cuda.Stream())
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TDS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)

    # BASELINE.json: config 3 (Ant x 4096 on one GPU) is the configuration the metric is quoted on; config 5 is "65 536
    # environments sharded 8 x" = 8192 per GPU.  --gpus 8 therefore runs config 5 as `value` and the 4096-per-GPU line — the
    # weak-scaling partner of the N = 1 / 2 / 4 runs — beside it under `envs_4096_per_gpu` (same ranks, same flow).
    # (debugging aid, like TDS_BENCH_ONE_DEVICE: TDS_BENCH_FORCE_CONFIG5=1 takes the N = 8 flow — two runs, 8192 then 4096
    #  environments per rank — at any world size; never used for numbers)
    config5 = args.envs_per_gpu == 0 and (world == 8 or os.environ.get("TDS_BENCH_FORCE_CONFIG5") == "1") and args.model == "ant"
    n = args.envs_per_gpu if args.envs_per_gpu > 0 else (8192 if config5 else 4096)
    out = run(args, n, rank, local_rank, world, secondary=True, config5=config5)
    if config5 and not args.no_secondary:
        second = run(args, 4096, rank, local_rank, world, secondary=False, config5=False)
        if rank == 0:
            out["envs_4096_per_gpu"] = {k: second[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")}
            out["envs_4096_per_gpu"]["workload"] = second["config"]["workload"]
            out["envs_4096_per_gpu"]["exchange_form"] = second["config"]["exchange_form"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def run(args, n, rank, local_rank, world, secondary, config5):
    """one measurement at n environments per GPU; returns the JSON record on rank 0 (None elsewhere)"""
    import numpy as np
    import torch
    import torch.distributed as dist

    import tds_amd
    from tds_amd import hip_backend, ranks

    m = tds_amd.load_model(args.model)
    lib_dtype = {"f64": "f64", "f32": "mixed", "f32-pure": "f32"}[args.dtype]
    multi = world > 1 or args.force_gather
    shard = None
    if multi:
        # the multi-GPU path of the C ABI: this rank's shard + the exchange of its records.  Only the 128-byte ncclUniqueId
        # travels through torch.distributed (rendezvous); the data path is the library's own (peer stores over IPC-mapped
        # rings, or librccl called from C).  There is NO other path under `value`: if the shard cannot be created the run
        # fails on every rank.
        if args.lanes:
            hip_backend.default_option("lanes_per_env", args.lanes)

        def create_shard(options):
            uid = None
            if world > 1 or os.environ.get("TDS_BENCH_RCCL_SINGLE", "1") == "1":
                uid = ranks.share_id(hip_backend.HipShard.unique_id, rank, world, "cuda")
            # (RCCL prints a version banner on C stdout when a communicator comes up: keep rank 0's stdout to the one
            #  JSON line — send C-level stdout to stderr while the communicator is created)
            import ctypes
            libc = ctypes.CDLL(None)
            sys.stdout.flush()
            libc.fflush(None)
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            err, sh = None, None
            try:
                sh = hip_backend.HipShard(m, world * n, rank=rank, world=world, device=local_rank, dtype=lib_dtype,
                                          unique_id=uid, wire_dtype=args.gather_dtype, block=max(1, args.gather_every),
                                          options=options)
                libc.fflush(None)
            except Exception as e:  # e.g. librccl refuses the communicator on this node
                err = repr(e)
            finally:
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
            if ranks.any_rank(bool(err), world, "cuda"):  # every rank takes the same path
                err = err or "another rank could not create its shard"
            if err:
                raise SystemExit(f"bench.py: tds_hip_shard_create failed on rank {rank}: {err} (no fallback under `value`)")
            return sh

        shard = create_shard(None)
        sim = shard.sim
    else:
        sim = hip_backend.HipSim(m, n, device=local_rank, dtype=lib_dtype,
                                 lanes_per_env=args.lanes if args.lanes else None)
    tdt = sim.torch_dtype
    nq, nd, adim = m.dof_q, m.dof_qd, m.action_dim

    # synthetic inputs: reset-like states, seed 3 (+rank), settle, fresh actions/step
    rng = np.random.default_rng(3 + rank)
    x0 = np.zeros((n, m.input_dim))
    if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and nq != nd:
        # floating base / spherical root joint (humanoid): the env's own reset distribution
        x0[:, :nq] = np.array([m.reset_q[i] for i in range(nq)]) + \
            np.array([m.reset_noise[i] for i in range(nq)]) * rng.uniform(-1, 1, (n, nq))
        x0[:, -3:] = [50, 1.5, 50] if args.model.startswith("humanoid") else [100, 2, 50]
    elif m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:
        ip = np.array([m.initial_poses[i] for i in range(adim)])
        x0[:, 2] = 0.48
        x0[:, 6:nq] = ip + 0.05 * rng.uniform(-1, 1, (n, nq - 6))
        x0[:, -3:] = [15, 0.3, 3] if args.model.startswith("ant") else [100, 2, 50]
    elif m.is_floating:
        # floating base: q = [quat xyzw | pos | joints]; dropped from 0.5 m with a small random tilt
        quat = rng.normal(size=(n, 4)) * [0.1, 0.1, 0.1, 0.0] + [0, 0, 0, 1.0]
        x0[:, 0:4] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
        x0[:, 6] = 0.5
        x0[:, 7:nq] = rng.uniform(-0.3, 0.3, (n, nq - 7))
    else:
        x0[:, :nq] = rng.uniform(-1, 1, (n, nq))
    def init_state(target):
        target.x.copy_(torch.from_numpy(x0).to(tdt).cuda())
        for _ in range(10):  # 10 settle steps with zero action (ant_environment2.h:137-152)
            target.step(None)

    init_state(sim)
    # the legged robots that fall over under +-0.4 rad random actions (Laikago, config 4) are timed the way the reference's
    # metric loop runs them: auto_reset_when_done on.  The Ant's default stays the plain loop (its auto_reset_rate is a
    # secondary key of the same line)
    falls_over = m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and m.reward_mode != 0 and not args.model.startswith("ant")
    auto_reset = ((args.auto_reset or (falls_over and not args.no_auto_reset)) and not multi
                  and m.step_mode == tds_amd.TDS_STEP_LOCOMOTION)
    if auto_reset:
        sim.set_auto_reset(True, 5)
    pool = 16
    # action amplitude: +-0.4 rad (the reference's ACTION_LIMIT) for the Ant; +-0.1 for the legged robots that can fall over
    # (Laikago, humanoid): the reference has no joint limits, a fallen robot driven by +-0.4 random actions blows up
    # numerically within ~1000 steps (tools/laikago_stability.py: 0 / 0 / 6 / 25 of 8192 non-finite after 600 / 800 / 1000 /
    # 1200 steps at +-0.4, none at +-0.1) — the metric loop of the reference would have reset it long before
    amp = (0.4 if (args.model.startswith("ant") or auto_reset) else 0.1) if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION else 0.0
    if args.action_amp >= 0:
        amp = args.action_amp
    actions = torch.from_numpy(rng.uniform(-amp, amp, (pool, n, adim))).to(tdt).cuda().contiguous()
    obs = torch.zeros((n, sim.obs_dim + 2), dtype=tdt, device="cuda")
    B = max(1, args.gather_every)
    use_graph = not args.no_graph
    shard_graph = use_graph if multi else False
    use_rings = use_graph and not multi and args.records == "rings"
    RS = max(1, args.ring_slots)
    obs_ring = y_ring = None
    if use_rings:  # (at N > 1 the shard layer owns the rings)
        obs_ring = torch.zeros((RS, n, sim.obs_dim + 2), dtype=tdt, device="cuda")
        # y records on 128-byte line boundaries (tds_hip_rings_t::y_stride; Ant f64: 160 scalars instead of 155): the
        # launch then writes whole lines only — the payload is unchanged
        per_line = 128 // (8 if tdt == torch.float64 else 4)
        # (auto: line boundaries where the padding is under 10 % of the record — Ant 155 -> 160 doubles, Laikago 409 -> 416 —,
        #  packed where it is not: pendulum5's 46 floats on 256 bytes would be 39 % more written than asked for)
        y_line = -(-m.output_dim // per_line) * per_line
        y_str = y_line if (args.y_stride == "line" or (args.y_stride == "auto" and y_line <= 1.1 * m.output_dim)) else m.output_dim
        y_ring = torch.zeros((RS, n, y_str), dtype=tdt, device="cuda")
    GCH = 1024  # steps per graph launch when K is larger (a multiple of the action pool)
    state = {"i": 0}

    def run_steps(k_steps):
        """k_steps closed-loop steps with a fresh action block each; multi: + the per-step record exchange"""
        if multi and shard_graph and k_steps % B == 0:
            left = k_steps
            while left > 0:  # (chunks: multiples of the action pool and of the exchange block)
                c = left if left <= GCH else GCH
                shard.step_many(actions, c, first_block=state["i"] % pool)
                state["i"] += c
                left -= c
        elif multi:
            for _ in range(k_steps):
                shard.step(actions[state["i"] % pool], 1)
                state["i"] += 1
        elif use_graph:
            left = k_steps
            while left > 0:
                c = left if left <= GCH else GCH
                if use_rings:
                    sim.step_many_rings(actions, c, obs_ring, y_ring, first_block=state["i"] % pool,
                                        obs_first=state["i"] % RS, y_first=state["i"] % RS)
                    state["i"] += c
                elif c < GCH and k_steps > GCH:  # remainder of a long run: eager (the graph cache holds one graph)
                    for _ in range(c):
                        sim.step(actions[state["i"] % pool], 1, obs)
                        state["i"] += 1
                else:
                    sim.step_many(actions, c, obs, first_block=state["i"] % pool)
                    state["i"] += c
                left -= c
        else:
            for _ in range(k_steps):
                sim.step(actions[state["i"] % pool], 1, obs)
                state["i"] += 1

    def prepare(k_steps):
        """build the hipGraph of the next run_steps(k_steps) ahead of time (nothing executes)"""
        if use_graph and k_steps > 0 and multi:
            if k_steps % B == 0 and shard_graph:
                shard.step_many(actions, min(k_steps, GCH), first_block=state["i"] % pool, prepare_only=True)
        elif use_graph and k_steps > 0 and use_rings:
            sim.step_many_rings(actions, min(k_steps, GCH), obs_ring, y_ring, first_block=state["i"] % pool,
                                obs_first=state["i"] % RS, y_first=state["i"] % RS, prepare_only=True)
        elif use_graph and k_steps > 0:
            sim.step_many_prepare(actions, min(k_steps, GCH), obs, first_block=state["i"] % pool)

    def flush():
        if multi:
            shard.flush()

    chains = None
    loop_form = use_graph and sim.step_many_is_loop(min(args.steps, GCH)) and (not multi or B == 1)
    if use_graph and not multi and not loop_form and not auto_reset:
        if args.chains == "auto":  # 6 x 128 extra untimed steps
            chains = sim.tune_step_many(actions, 128, obs)
        elif args.chains != "default":
            chains = int(args.chains)
            sim.set_graph_chains(chains)
    scratch = None
    if args.spin_up_steps > 0:
        scratch = hip_backend.HipSim(m, n, device=local_rank, dtype=lib_dtype,
                                     lanes_per_env=args.lanes if args.lanes else None)
        scratch.x.copy_(sim.x)
        left = args.spin_up_steps
        while left > 0:
            c = min(left, 500)
            scratch.step_many(actions, c)
            left -= c
        # (no synchronisation: the warm-up steps below queue up behind it on the same stream)
    shard_form = None
    exchange_form = None
    if multi:
        # The exchange forms, most to least ambitious:
        #   ring / peer stores   the step-loop launch (the N = 1 kernel) stores every record on every rank itself (IPC-mapped
        #                        rings): no collective, no host call per step — the library's default
        #   ring / staged copies the same rings and flags, but the launch writes this rank's ring only and the communication
        #                        stream copies the launch's slots into every peer's ring (option shard_peer_copy = 1: the
        #                        runtime's copy engines instead of stores from the kernel)
        #   ring / RCCL          the same launches, the slots sent with ncclAllGather (shard re-created with shard_peer = 0)
        #   per-step eager       one tds_hip_shard_step call per step (kernel launch + ncclAllGather)
        # A form that fails during the warm-up steps on ANY rank (a wait that timed out, an RCCL error; a set-up that cannot
        # be made falls back inside the library) is dropped on EVERY rank before anything is timed.  No timing decides
        # anything here: every rank runs the library's default build of the kernel.
        forms = ([("ring", None), ("ring", {"shard_peer_copy": 1}), ("ring", {"shard_peer": 0}), ("per-step eager", None)]
                 if shard_graph else [("per-step eager", None)])
        current_opts = None
        for f, fopts in forms:
            err = 0
            try:
                if fopts != current_opts and fopts is not None:  # (options that shape the ring: a new shard)
                    shard.close()
                    shard = create_shard(fopts)
                    sim = shard.sim
                    init_state(sim)
                    current_opts = fopts
                if f == "per-step eager":
                    shard_graph = False
                prepare(args.warmup)
                run_steps(max(args.warmup, 1))
                flush()
                torch.cuda.synchronize()
            except SystemExit:
                raise
            except Exception as e:  # noqa: BLE001
                print(f"bench.py: exchange form '{f}' {fopts or ''} failed on rank {rank}: {e!r}", file=sys.stderr)
                err = 1
            err = 1 if ranks.any_rank(bool(err), world, "cuda") else 0
            if not err:
                shard_form = f
                break
        if shard_form is None:
            raise SystemExit("bench.py: no exchange form works on this node")
        exchange_form = shard.exchange_form()
        n_peers = shard.peer_count()
        loop_form = loop_form and shard_form == "ring"
    else:
        prepare(args.warmup)
        run_steps(args.warmup)
        flush()
        torch.cuda.synchronize()
    K = args.steps
    prepare(K)  # (capture + instantiate only)
    # N = 1, one call per timed region: the call's arguments are marshalled ahead of the region (the region then holds
    # one C call — tds_hip_step_many_rings — instead of ~30 us of Python argument checking around it)
    fast_call = None
    if use_rings and not multi and K <= GCH:
        fast_call = sim.prepared_step_many_rings(actions, K, obs_ring, y_ring, first_block=state["i"] % pool,
                                                 obs_first=state["i"] % RS, y_first=state["i"] % RS)
    if world > 1:
        dist.barrier()
    # The host has spent milliseconds marshalling the timed call while the GPU sat idle: a short launch of the scratch handle
    # right in front of the synchronisation that opens the region keeps the clocks where the spin-up left them (the same
    # 20-step launch repeated back to back in one process takes 300 us, after an idle gap 350: profiles/
    # r04_bench_20_step_harness_cost.txt, tools/ab_slots.py).  Untimed, its own state, named in config.spin_up.
    if scratch is not None:
        scratch.step_many(actions, 256)
        evw = torch.cuda.Event()
        evw.record()
        while not evw.query():
            pass
    torch.cuda.synchronize()

    use_events = not args.no_events
    # one HIP event pair brackets the whole timed region on the launch stream (HipSim hands
    # torch.cuda.current_stream() to tds_hip_set_stream): region_ms / K is the average launch
    # duration incl. the inter-launch gaps.  Per-launch event pairs inside the region would insert
    # a marker packet between consecutive launches (~8 us each, 20 % at these kernel sizes), so
    # the isolated per-launch duration is sampled in a separate pass after the timed region.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    if fast_call is not None:
        fast_call()
        state["i"] += K
    else:
        run_steps(K)
        flush()
    ev1.record()
    while not ev1.query():  # (polled: the wake-up of a blocking wait costs 10 - 60 us, a fifth of a 20-step region)
        pass
    elapsed = ranks.close_region(t0, world, "cuda", torch.cuda.synchronize)  # (sync, barrier, sync; max over ranks)

    kernel_ms = None
    kernel_ms_isolated = None
    if use_events:
        region_ms = ev0.elapsed_time(ev1)
        if world == 1 and not multi:
            kernel_ms = region_ms / K  # only kernel launches are in the region at N = 1
        ns = min(K, 200)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(ns)]
        for i in range(ns):
            evs[i][0].record()
            sim.step(actions[i % pool], 1, obs)
            evs[i][1].record()
        torch.cuda.synchronize()
        kernel_ms_isolated = float(np.mean([a.elapsed_time(b) for a, b in evs]))
        if kernel_ms is None:
            kernel_ms = kernel_ms_isolated

    # secondary (N > 1, ring exchange): the same K steps (a) with only [reward | done] travelling to the peers (exchange_fields =
    # 1: 8 of 120 bytes per Ant environment) and (b) with NO exchange at all (every rank its own step-loop launches with its own
    # rings) — beside `value` they separate what the kernels scale like from what the links cost: value << no_exchange says
    # "link-bound", exchange_fields_reward_done ~ no_exchange says "the per-record protocol itself is free"
    exch_variants = None
    if multi and secondary and not args.no_secondary and shard_form == "ring" and K >= 1:
        exch_variants = {}

        def timed_k(step_fn, sync_fn):
            step_fn(min(K, 64))  # warm-up (graphs, rings)
            sync_fn()
            t1 = ranks.open_region(world, torch.cuda.synchronize)
            step_fn(K)
            sync_fn()
            return ranks.close_region(t1, world, "cuda", torch.cuda.synchronize)

        def shard_variant(key, opts, what):
            try:
                shard.flush()
                torch.cuda.synchronize()
                sh2 = create_shard(opts)  # (a second shard beside the first: its own communicator and rings)
                init_state(sh2.sim)
                st2 = {"i": 0}

                def steps_v(k):
                    left = k
                    while left > 0:
                        c = min(left, GCH)
                        sh2.step_many(actions, c, first_block=st2["i"] % pool)
                        st2["i"] += c
                        left -= c

                dt_v = timed_k(steps_v, lambda: (sh2.flush(), torch.cuda.synchronize()))
                exch_variants[key] = {"value": ranks.job_rate(world, n, K, dt_v), "unit": "env-steps/s", "ms_per_step": dt_v / K * 1e3,
                                      "exchange_form": sh2.exchange_form(), "what": what}
                sh2.close()
            except Exception as e:  # noqa: BLE001  (a secondary: reported, never fatal)
                exch_variants[key] = {"error": repr(e)[:300]}

        shard_variant("exchange_fields_reward_done", {"exchange_fields": 1},
                      "the same launches, only [reward | done] of a record travel to the peers (option exchange_fields = 1; this "
                      "rank's own block still receives the whole record)")
        try:
            plain = hip_backend.HipSim(m, n, device=local_rank, dtype=lib_dtype)
            init_state(plain)
            o_ring = torch.zeros((RS, n, plain.obs_dim + 2), dtype=tdt, device="cuda")
            per_line = 128 // (8 if tdt == torch.float64 else 4)
            yr = torch.zeros((RS, n, -(-m.output_dim // per_line) * per_line), dtype=tdt, device="cuda")
            st3 = {"i": 0}

            def steps_plain(k):
                left = k
                while left > 0:
                    c = min(left, 256)  # (the launch length of the exchange runs)
                    plain.step_many_rings(actions, c, o_ring, yr, first_block=st3["i"] % pool, obs_first=st3["i"] % RS, y_first=st3["i"] % RS)
                    st3["i"] += c
                    left -= c

            dt_ne = timed_k(steps_plain, torch.cuda.synchronize)
            exch_variants["no_exchange"] = {
                "value": ranks.job_rate(world, n, K, dt_ne), "unit": "env-steps/s", "ms_per_step": dt_ne / K * 1e3,
                "what": "the same step-loop launches on every rank with per-step records into the rank's own rings, nothing "
                        "leaves the GPU: what the kernels alone scale like"}
            del plain
        except Exception as e:  # noqa: BLE001
            exch_variants["no_exchange"] = {"error": repr(e)[:300]}
        # (last of the secondaries: the only one that drives the runtime's copy path across devices — nothing measured above
        #  depends on how it ends; only where the peer mappings exist, i.e. the peer-store form ran)
        if exchange_form == "peer_stores":
            shard_variant("exchange_staged_copies", {"shard_peer_copy": 1},
                          "the same rings and flags, the launch writes this rank's ring only and the communication stream copies "
                          "the launch's slots into every peer's ring behind it (option shard_peer_copy = 1: one strided "
                          "device-to-device copy per peer and launch — the copy engines — instead of stores from the kernel)")

    # secondary (N > 1): the pipelined exchange — records of `pipelined_block` consecutive steps in one all-gather
    pipelined = None
    if multi and args.pipelined_block > 1 and args.pipelined_block != B:
        shard.set_block(args.pipelined_block)
        B_saved, B = B, args.pipelined_block
        run_steps(2 * args.pipelined_block)
        flush()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        run_steps(K)
        flush()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt_p = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([dt_p], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_p = float(t.item())
        pipelined = {"value": world * n * K / dt_p, "unit": "env-steps/s", "steps_per_exchange": args.pipelined_block,
                     "what": "same steps, records of 32 consecutive steps per all-gather (observations reach other "
                             "ranks up to 32 steps late); secondary — the headline value uses one exchange per step"}
        B = B_saved
        shard.set_block(B)

    # The reference has no joint limits and no velocity clamps: a robot that has fallen over can be driven
    # into a numerical blow-up by the random actions (the CPU reference diverges from the same state the same
    # way, checked with tools/debug_finite.py + the oracle).  Reported, not hidden: environments whose state
    # left the finite range during the run (no resets in this benchmark loop).
    bad_envs = int((~torch.isfinite(sim.y).all(dim=1)).sum().item())
    finite = bad_envs == 0

    def timed(fn, k_steps):
        """wall time of fn() bracketed by synchronisations, on this rank (secondary measurements, N = 1); like the timed
        region of `value`, right behind 256 untimed steps of the scratch handle (GPU clocks)"""
        if scratch is not None:
            scratch.step_many(actions, 256)
            evs = torch.cuda.Event()
            evs.record()
            while not evs.query():
                pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return n * k_steps / (time.perf_counter() - t1)

    # ---- secondary keys of the default N = 1 line (same model, same batch, same K; each on the state the timed region left)
    substep_fused = one_rank = one_rank_7 = auto_rate = None
    steady = None
    if use_rings and world == 1 and not args.no_secondary and secondary and loop_form:
        # (0) the steady state ("steady-state over >= 1000 steps after 100 warm-up steps"): the same form, the same
        #     rings, 100 untimed + 1000 timed steps as ONE launch, bracketed by a HIP event pair on the launch stream
        sim.step_many_rings(actions, 100, obs_ring, y_ring, first_block=state["i"] % pool, obs_first=state["i"] % RS, y_first=state["i"] % RS)
        state["i"] += 100
        call = sim.prepared_step_many_rings(actions, 1000, obs_ring, y_ring, first_block=state["i"] % pool,
                                            obs_first=state["i"] % RS, y_first=state["i"] % RS)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # (opened the way the timed region of `value` is: the host has marshalled the call while the GPU sat idle — the 100
        #  warm-up steps are 1.2 ms old by now — so the scratch handle's short launch keeps the clocks up; config.spin_up)
        if scratch is not None:
            scratch.step_many(actions, 256)
            evs = torch.cuda.Event()
            evs.record()
            while not evs.query():
                pass
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e0.record()
        call()
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        state["i"] += 1000
        ms = e0.elapsed_time(e1)
        bpes = (m.input_dim + m.output_dim) * (8 if args.dtype == "f64" else 4)
        ach = n * bpes / (ms / 1000 * 1e-3) / 1e9
        tr, tr_src = (None, None) if auto_reset else pmc_traffic_rings(args.model, n, args.dtype, 1000)
        steady = {"value": n * 1000 / wall, "unit": "env-steps/s", "steps": 1000, "warmup": 100, "ms_per_step": wall,
                  "kernel_ms_avg": ms, "steps_per_launch": 128 if auto_reset else 1000, "us_per_step_kernel": ms,
                  "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                               "traffic": tr, "traffic_source": tr_src,
                               "traffic_over_algorithmic": (tr / (n * bpes * 1000)) if tr else None},
                  "what": ("the same launches and rings as `value`, 1000 steps in ONE call after 100 warm-up steps (auto-reset: "
                           "step-loop launches of 128 steps, the reset pool's refill launches beside them); "
                           if auto_reset else
                           "the same launches and rings as `value`, 1000 steps in ONE launch after 100 warm-up steps; ") +
                          "value from the wall clock around the call, kernel_ms_avg from a HIP event pair on the launch stream"}
    if use_rings and world == 1 and not args.no_secondary and not auto_reset and secondary:
        # (a) the substep-fused form: the same launches WITHOUT per-step records (y / obs of the last step of a launch only)
        kk = min(K, GCH)
        sim.step_many(actions, kk, obs)
        v = timed(lambda: sim.step_many(actions, kk, obs), kk)
        substep_fused = {"value": v, "unit": "env-steps/s", "steps": kk,
                         "what": "tds_hip_step_many: the same K steps, output packing and records for the LAST step of a launch "
                                 "only (batch x substeps per launch; not the per-step protocol, never `value`)"}
        # (b) one rank through the shard layer (tds_hip_shard_step_many): what every rank of an N > 1 run executes, minus
        #     the other ranks — the protocol's own cost.  Default form: the peer-store exchange (on one rank: the arrival
        #     counters and this rank's own flags); and once more with SEVEN scratch rings of this GPU's own standing in for
        #     the peers of an 8-GPU run (option shard_peer_loopback): every store an 8-rank run issues, HBM instead of xGMI.
        def one_rank_line(options, label):
            import ctypes
            libc = ctypes.CDLL(None)
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)  # (RCCL's banner goes to stderr)
            try:
                uid = hip_backend.HipShard.unique_id() if hip_backend.HipShard.rccl_version() > 0 else None
                sh1 = hip_backend.HipShard(m, n, rank=0, world=1, device=local_rank, dtype=lib_dtype, unique_id=uid,
                                           wire_dtype=args.gather_dtype, options=options)
                libc.fflush(None)
            finally:
                os.dup2(saved_fd, 1)
                os.close(saved_fd)
            sh1.sim.x.copy_(sim.x)
            sh1.step_many(actions, kk)
            sh1.flush()
            sh1.step_many(actions, kk, prepare_only=True)

            def go():
                sh1.step_many(actions, kk)
                sh1.flush()

            v = timed(go, kk)
            form = sh1.exchange_form()
            rec = {"value": v, "unit": "env-steps/s", "steps": kk, "exchange_form": form, "peers": sh1.peer_count(),
                   "communicator": "single-rank RCCL communicator" if uid else "none (librccl not loadable)",
                   "what": label}
            sh1.close()
            return rec

        try:
            one_rank = one_rank_line(None, "tds_hip_shard_step_many on ONE rank, library defaults: step-loop launches (the N = 1 "
                                           "two-wavefront kernel) of <= 256 steps storing every step's [obs | reward | done] record "
                                           "straight into this rank's block of the gathered slot and raising the slot's flag when "
                                           "the last workgroup has stored it; a one-wave arrival check per launch on the "
                                           "communication stream (what every rank of an N > 1 run executes, minus the peers)")
        except Exception as e:  # noqa: BLE001 - a secondary key must never cost the headline line
            one_rank = {"error": repr(e)}
        try:
            one_rank_7 = one_rank_line({"shard_peer_loopback": 7},
                                       "the same with seven scratch rings of this GPU standing in for the peers of an 8-GPU run: "
                                       "the kernel issues every store it issues on eight ranks (7 x 120 B per environment and "
                                       "step on the float wire), into HBM instead of over xGMI")
        except Exception as e:  # noqa: BLE001
            one_rank_7 = {"error": repr(e)}
        # (c) auto_reset_when_done on — the loop python/examples/vec_ant.py:16-40 times: every step resets the
        #     environments it ends with done (reset distribution + settle steps, through the reset pool), records per step
        if m.step_mode == tds_amd.TDS_STEP_LOCOMOTION and m.reward_mode != 0:
            try:
                ar = hip_backend.HipSim(m, n, device=local_rank, dtype=lib_dtype, lanes_per_env=args.lanes if args.lanes else None)
                ar.x.copy_(sim.x)
                ar.set_auto_reset(True, 5)
                ar.step_many_rings(actions, max(kk, 64), obs_ring, y_ring)  # (fills the reset pool, first passes)
                ar.step_many_rings(actions, kk, obs_ring, y_ring)
                # (the refill passes of the reset pool go out once per 128 steps, across calls: a run of calls that
                #  covers several passes is timed, every call bracketed by synchronisations as the timed region is)
                calls = max(1, -(-640 // kk))
                rates = [timed(lambda: ar.step_many_rings(actions, kk, obs_ring, y_ring), kk) for _ in range(calls)]
                v = calls / sum(1.0 / r for r in rates)
                dn = int((obs_ring[(kk - 1) % RS][:, -1] != 0).sum().item())
                auto_rate = {"value": v, "unit": "env-steps/s", "steps": kk, "calls": calls, "done_in_last_step": dn,
                             "slowest_call": min(rates), "fastest_call": max(rates),
                             "what": "auto_reset_when_done: the same launches with every done environment re-initialised + settled "
                                     "(%d settle steps) through the reset pool, records per step" % m.settle_steps}
                ar.close()
            except Exception as e:  # noqa: BLE001
                auto_rate = {"error": repr(e)}

    # the config-4 default line runs with auto-reset; the plain loop (+-0.1 rad, nothing reset: round 5's line) beside it
    no_reset = None
    if auto_reset and use_rings and world == 1 and not args.no_secondary and secondary and not args.auto_reset:
        try:
            kk = min(K, GCH)
            nr = hip_backend.HipSim(m, n, device=local_rank, dtype=lib_dtype, lanes_per_env=args.lanes if args.lanes else None)
            init_state(nr)
            act01 = (actions * (0.1 / amp)).contiguous() if amp > 0 else actions
            nr.step_many_rings(act01, max(kk, 64), obs_ring, y_ring)
            rates = [timed(lambda: nr.step_many_rings(act01, kk, obs_ring, y_ring), kk) for _ in range(5)]
            no_reset = {"value": sorted(rates)[len(rates) // 2], "unit": "env-steps/s", "steps": kk, "calls": len(rates),
                        "what": "the same launches without auto-reset, +-0.1 rad actions (nothing resets a fallen robot): "
                                "median of %d calls" % len(rates)}
            nr.close()
        except Exception as e:  # noqa: BLE001
            no_reset = {"error": repr(e)}

    # secondary (not the headline): the same environments driven by per-environment linear policies
    # entirely on device, R policy steps per launch (tds_hip_rollout)
    rollout = None
    if args.rollout_steps > 0 and world == 1 and m.step_mode == tds_amd.TDS_STEP_LOCOMOTION:
        od = sim.obs_dim
        pol = torch.from_numpy(rng.normal(0.0, 0.05, (n, adim * od + adim))).to(tdt).cuda().contiguous()
        sim.rollout(pol, args.rollout_steps)
        torch.cuda.synchronize()
        reps = 5
        t1 = time.perf_counter()
        for _ in range(reps):
            sim.rollout(pol, args.rollout_steps)
        torch.cuda.synchronize()
        dt_r = time.perf_counter() - t1
        rollout = {"value": n * args.rollout_steps * reps / dt_r, "unit": "env-steps/s",
                   "policy_steps_per_launch": args.rollout_steps, "launches": reps,
                   "what": "linear policy + step + reward/done + return bookkeeping on device: one launch per rollout "
                           "(step-loop build)"}
    if rank == 0:
        elem = 8 if args.dtype == "f64" else 4  # bytes per scalar of the records in HBM
        bytes_per_env_step = (m.input_dim + m.output_dim) * elem  # x record in + y record out
        total_steps = world * n * K
        value = ranks.job_rate(world, n, K, elapsed)
        per_step_records = use_rings or multi or not use_graph  # every step packs + stores its records
        ring_exchange = multi and shard_form == "ring"
        roof = None
        if kernel_ms:
            # environment chains: C launches of n / C environments each are in flight at the same time, every one of
            # them lasting about one step period (a chain's launches run back to back); the HBM peak is the whole
            # GPU's, so `achieved` adds up the launches in flight
            conc = chains if (chains and use_graph and not multi) else 1
            achieved = n * bytes_per_env_step / (kernel_ms * 1e-3) / 1e9
            traffic, traffic_src = pmc_traffic(args.model, n, args.dtype)  # (measured on whole-batch launches)
            if traffic is not None:
                traffic = traffic // conc
            spl = (min(K, 256) if multi else min(K, GCH)) if loop_form else 1  # steps per launch (loop form: one launch = spl steps)
            if auto_reset:  # (step-loop launches of up to 128 steps, or single steps; the refill launches run beside them)
                spl = min(K, GCH, 128) if loop_form else 1
                # (single-step launches: counters of this very command; step-loop launches: of the same launch without resets)
                traffic, traffic_src = pmc_traffic_rings(args.model, n, args.dtype, spl) if loop_form else (traffic, traffic_src)
            elif loop_form and per_step_records:
                # every step's y and obs records leave the launch, the state itself stays in LDS: measured on the
                # driver's own command
                traffic, traffic_src = pmc_traffic_rings(args.model, n, args.dtype, spl)
            elif loop_form:
                # the state stays in LDS across the steps of a launch: per step only the action block is read, the
                # records are written once per launch — measured on the step-loop launch itself
                traffic, traffic_src = pmc_traffic_loop(args.model, n, args.dtype, spl)
            arith = "f32" if args.dtype == "f32-pure" else "f64"
            # which kernel the timed launches ran (tds_hip_single_step_kernel: the star-shaped robots have their own)
            kernel_name = {"oct8": "tds_oct_kernel", "quad16": "tds_quad_kernel", "chain8": "tds_chain_kernel"}.get(sim.single_step_kernel()[0], "tds_step_kernel")
            sq_issue = sq_counters(args.model, n, args.dtype)
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": kernel_name, "kernel_ms_avg": kernel_ms * spl, "steps_per_launch": spl,
                    "kernel_ms_isolated": kernel_ms_isolated,
                    "algorithmic_bytes_per_launch": n * bytes_per_env_step * spl // conc,
                    "traffic_over_algorithmic": (traffic / (n * bytes_per_env_step * spl // conc)) if traffic else None,
                    "launches_in_flight": conc,
                    "achieved_per_launch": achieved / conc,
                    # secondary view: flops of the reference's DENSE formulation per env-step (dense 51 x 14 x 51
                    # products the kernels never form: NOT executed flops; the executed share is valu_issue_frac)
                    "dense_formulation_flops_per_env_step": ALG_FLOPS.get(args.model),
                    "dense_formulation_tflops": (ALG_FLOPS[args.model] * n / (kernel_ms * 1e-3) / 1e12
                                                 if args.model in ALG_FLOPS else None),
                    "valu_peak_tflops": 78.6 if arith == "f64" else 157.3,
                    # what the kernel actually keeps busy: VALU instructions issued / issue slots, from the committed SQ-counter
                    # pass of this kernel (profiles/sq_counters.json; counters cannot be read from inside this process)
                    "valu_issue_frac": sq_issue.get("valu_issue_frac"), "valu_issue_source": sq_issue.get("source"),
                    "note": "algorithmic bytes = (input_dim+output_dim)*sizeof(T) per env-step (SURVEY 8d: one read of the x "
                            "record + one write of the y record); achieved = launches_in_flight x algorithmic_bytes_per_launch "
                            "/ kernel_ms_avg; a step-loop launch keeps the state in LDS: per step it reads an action block "
                            "and writes the y record + the obs record into their ring slots (traffic); the path is "
                            "VALU/LDS-latency bound, see DESIGN.md"}
        if auto_reset and loop_form:
            launch = ("step-loop launches of up to 128 steps, per-step records into rings; a done environment takes its next "
                      "pre-settled state from its ring in HBM; rings refilled by straight-line launches on a side stream")
        elif auto_reset:
            launch = ("one straight-line launch per step, a done environment takes its next pre-settled state from its ring in "
                      "HBM; rings refilled on a side stream")
        elif multi and ring_exchange:
            how = {"peer_stores": "the launch stores each record into its block of the gathered slot on EVERY rank (IPC-mapped "
                                  "rings, system-scope stores over xGMI) and raises the slot's flags when its last workgroup has "
                                  "stored it: no collective, the transfer of step k lies inside step k + 1",
                   "peer_copy": "the launch stores its records into this rank's block of the gathered ring; behind it the "
                                "communication stream copies the launch's slots into every peer's ring (IPC-mapped; one strided "
                                "device-to-device copy per peer: the copy engines) and raises the slots' flags: no collective",
                   "rccl_group_after_launch": "the launch's slots are all-gathered as ONE RCCL group behind it",
                   "rccl_per_slot": "the communication stream follows the slots' progress counters and all-gathers each slot "
                                    "beside the launch"}.get(exchange_form, str(exchange_form))
            launch = ("one launch of the step-loop kernel per %d steps, EVERY step packing and storing its y record and its "
                      "[obs | reward | done] record into ring slots; %s" % (min(K, 256), how))
        elif loop_form and use_rings:
            launch = ("one launch of the step-loop kernel per %d steps (state in LDS across the steps, one action block per "
                      "step), EVERY step running the whole output packing — visual poses, y record, reward / done, "
                      "observation — and storing its y record and its [obs | reward | done] record into their slots of "
                      "%d-slot rings in HBM (tds_hip_step_many_rings: per-step records)" % (min(K, GCH), RS))
        elif loop_form:
            launch = ("one launch of the step-loop kernel per %d steps (state in LDS across the steps, one action block per step; "
                      "records written once per launch: substep-fused form)" % min(K, GCH))
        elif use_graph and (not multi or shard_graph):
            launch = ("hipGraph: %d single-step launches%s per graph launch, every launch writing its y and [obs | reward | done] "
                      "records%s" % (min(K, GCH), " + their exchanges" if multi else "", " into ring slots" if use_rings else "")
                      + ("" if chains is None else ", the batch as %d independent environment chain%s (%s)" % (
                          chains, "" if chains == 1 else "s", "measured at warm-up" if args.chains == "auto" else "--chains")))
        else:
            launch = "one tds_hip_shard_step call per step (kernel launch + exchange)" if multi else "one kernel launch per step"
        out = {
            "metric": "env-steps/sec", "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K,
            "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            # the arithmetic type the path computes in; the record type is in config.records
            "dtype": "f32" if args.dtype == "f32-pure" else "f64", "data": "synthetic",
            "config": {"workload": (("BASELINE config 5: " if config5 else "") +
                                    (f"{args.model} (gym Ant 14-dof + plane, 17 contact points, PGS 1 iter), " if args.model == "ant"
                                     else f"{args.model}, ")) +
                                   (f"{world * n} envs sharded {world} x = " if multi and world > 1 else "") +
                                   f"{n} envs/GPU, dt={m.dt}; state fed back on device every step; actions: a pool of {pool} "
                                   f"uniform random (+-{amp}) action batches resident in HBM, step k takes batch k mod {pool} "
                                   f"(no policy in the loop)",
                       "records": "f64" if args.dtype == "f64" else "f32",
                       "per_step_records": bool(per_step_records),
                       "auto_reset": ("auto_reset_when_done: every step resets the environments it ends with done "
                                      "(reset distribution + %d settle steps) through the reset pool; %d of %d environments "
                                      "done in the last step" % (m.settle_steps, int((obs[:, -1] != 0).sum().item()), n))
                       if auto_reset else None,
                       "launch": launch,
                       "exchange_form": ("%s / %s" % (shard_form, exchange_form)) if multi else None,
                       "peers": n_peers if multi else None,
                       "spin_up": ("%d untimed steps of a scratch handle before the warm-up steps and 256 more right in front of the "
                                   "synchronisation that opens the timed region (GPU clocks)" % args.spin_up_steps)
                       if args.spin_up_steps > 0 else None,
                       "envs_per_gpu": n, "global_envs": world * n, "substeps_per_launch": 1,
                       "steps_per_launch": (min(K, 256) if multi else min(K, GCH)) if loop_form else 1,
                       "parallelism": f"env-shard x{world}" + (
                           f" + one exchange of the (obs|reward|done) records per {'policy step' if B == 1 else str(B) + ' steps'} "
                           f"({'peer stores into IPC-mapped gathered rings, no collective' if exchange_form == 'peer_stores' else ('staged copies into IPC-mapped gathered rings, no collective' if exchange_form == 'peer_copy' else 'ncclAllGather, librccl called from the C ABI')}; "
                           f"tds_hip_shard_step_many), {args.gather_dtype if args.dtype == 'f64' else 'f32'} on the wire (the records are "
                           f"computed and fed back in {'f64' if args.dtype == 'f64' else 'f32'} on the owning GPU), overlapped with "
                           f"the following steps" if multi else ""),
                       "lanes_per_env": sim.single_step_kernel()[1],
                       "lds_bytes_per_env": sim.single_step_kernel()[2]},
            "roofline": roof, "finite": finite, "nonfinite_envs": bad_envs,
        }
        if substep_fused is not None:
            out["substep_fused"] = substep_fused
        if steady is not None:
            out["steady_state_1000"] = steady
        if one_rank is not None:
            if "value" in one_rank:
                one_rank["ratio_to_value"] = one_rank["value"] / value
            out["one_rank_with_exchange"] = one_rank
        if one_rank_7 is not None:
            if "value" in one_rank_7:
                one_rank_7["ratio_to_value"] = one_rank_7["value"] / value
            out["one_rank_with_exchange_7_loopback_peers"] = one_rank_7
        if auto_rate is not None:
            out["auto_reset_rate"] = auto_rate
        if no_reset is not None:
            out["no_auto_reset_rate"] = no_reset
        if rollout is not None:
            out["on_device_rollout"] = rollout
        if pipelined is not None:
            out["pipelined_gather"] = pipelined
        if exch_variants:
            for k_, v_ in exch_variants.items():
                if "value" in v_:
                    v_["ratio_to_value"] = v_["value"] / value
                out[k_] = v_
        if not args.no_cpu_baseline and world == 1 and secondary:
            cb = cpu_baseline(args.model, min(n, 4096))
            primary = cb.get("reference") or cb.get("port")
            out["cpu_baseline"] = primary
            out["cpu_baseline_port"] = cb.get("port")
            out["cpu_baseline_generated"] = cb.get("generated")
        return out
    return None


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- collocation-node constraint+Jacobian evaluations per second on MI355X.

A "step" is ONE full evaluation pass of the hot path over the rank's share of a batch of synthetic
problem instances (BASELINE.json config 3: 6-state quadrotor VGP, N=1024 LGL nodes, 20 static
keep-outs), producing for every node F, C, L, the defect row, dF, dC, dL with the Jacobian
values landed in the NLP value array and the cost reduced.
Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scenarios S] [--batch B] [--config c3|c5]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
       (or plain `python bench.py --gpus N`: it then starts the N ranks itself, as child processes, before touching the GPU)
The batch is the 1024-scenario obstacle-field Monte-Carlo of config 4: it is STRONG-scaled, rank r evaluates
scenarios [r*S/N, (r+1)*S/N) (S/N = 128 per GPU at N = 8) with no data-path collective; `value` = S*M*K / time.
The same run then repeats the measurement weak-scaled (1024 scenarios per GPU) and reports it as a second field.
The only collective of the path, the gather of the trajectories, runs once after the timed region.
--config c5: the fp32 path of config 5 (12-state fixed wing, N=4096, MFMA D.X), 256 instances per GPU.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {"c3": 1040, "c2": 560, "c5": 944}       # SURVEY.md section 8d, per node-eval
HBM_PEAK_GBS = 8000.0                                  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TF = 78.6                               # MI355X fp64 matrix
FP32_MFMA_PEAK_TF = 157.3                              # MI355X_MICROARCH.md: f32-input MFMA


def load_pmc_traffic(kernel, B, M):
    """HBM bytes per launch of `kernel` at THIS batch size and mesh from the PMC passes of tools/pmc_traffic.sh (separate
    FETCH_SIZE / WRITE_SIZE passes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950), if this round's summary
    under profiles/ holds an entry for exactly that (kernel, B, M) -- a figure collected at another batch size is not this
    run's traffic.  Returns (bytes or None, provenance or None); the counters are not collected inside bench.py (rocprofv3
    serialises dispatches while it counts)."""
    for tag in ("r04", "r03"):
        path = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
        try:
            for e in json.load(open(path)).get("entries", []):
                if e.get("kernel") == kernel and e.get("B") == B and e.get("M") == M:
                    return e["bytes_per_launch"], f"profiles/{tag}_pmc_traffic.json ({e.get('source', 'tools/pmc_traffic.sh')})"
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def usable_cores():
    """CPU threads this process may really use (affinity mask and cgroup quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(M, n_obs, budget_s=10.0):
    """ePSOPT-style CPU port (oracle/epsopt_style.cpp) on a bounded sample of the same workload: all host
    cores via OpenMP over instances (the reported value), the same port on one thread, and the plain-C
    checker oracle (oracle/emi_oracle.c, one thread) for reference.  Checker/baseline only, never the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from etol_amd import workloads as W
    from etol_amd import lgl
    cores = min(O.eps().eps_max_threads(), usable_cores())
    Bs = max(cores, 8)
    X, U, recs = W.quadrotor_batch(2, Bs, M, n_obs)
    mesh = lgl(M)

    def rate(style, nthreads, nb, budget):
        Xs, Us, rs = X[:nb], U[:nb], recs[:nb]
        O.evaluate(1, W.QUAD_PARAMS, M, mesh, 0.0, W.TF, Xs, Us, rs, style=style, nthreads=nthreads)  # warm
        t0 = time.perf_counter()
        passes = 0
        while True:
            O.evaluate(1, W.QUAD_PARAMS, M, mesh, 0.0, W.TF, Xs, Us, rs, style=style, nthreads=nthreads)
            passes += 1
            el = time.perf_counter() - t0
            if el > budget:
                break
        return nb * M * passes / el, passes, el

    v_all, passes, el = rate("epsopt", cores, Bs, 0.5 * budget_s)
    v_one, p1, e1 = rate("epsopt", 1, min(Bs, 2), 0.25 * budget_s)
    v_orc, p2, e2 = rate("oracle", cores, Bs, 0.25 * budget_s)
    return {"value": v_all, "unit": "node-evals/s", "cores": int(cores), "kind": "port",
            "sample": f"{passes} passes over {Bs} instances x {M} nodes (ePSOPT-style std::any/dual-number port, "
                      f"OpenMP over instances), {el:.1f} s",
            "single_thread": {"value": v_one, "sample": f"{p1} passes over {min(Bs, 2)} instances, {e1:.1f} s"},
            "checker_oracle": {"value": v_orc, "cores": 1,
                               "sample": f"{p2} passes over {Bs} instances (oracle/emi_oracle.c, one thread: complex-step "
                                         f"derivatives and long-double D.X -- built for checking, not for speed), {e2:.1f} s"}}


def check_against_oracle(ev, X, U, recs, outs, M, instances):
    """Untimed, after the timed region, rank 0 at N = 1 only (part of the cpu_baseline leg: the oracle is the checker, never
    the thing measured): the outputs the timed passes left in HBM against oracle/emi_oracle.c on sampled instances."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from etol_amd import workloads as W
    e = O.sampled_errors(1, W.QUAD_PARAMS, M, (ev.tau, ev.w, ev.D), 0.0, W.TF, X, U, recs, outs, instances)
    return {"max_rel_err": e["max_rel_err"], "instances": e["instances"], "defect": e["defect"], "path_rows": e["path"],
            "jacobian_values": e["vals"], "cost": e["cost"], "against": "oracle/emi_oracle.c (complex-step derivatives, long-double D.X)",
            "tolerance": {"defect": 5e-13, "node": 1e-13}, "ok": bool(e["defect"] < 5e-13 and e["path"] < 1e-13 and e["vals"] < 1e-13 and e["cost"] < 1e-13)}


def roofline_of(m, def_name, key, n_obs, B, M, ns, ms_per_step):
    """`roofline` object of one measurement.  The one-launch pass kernel does both jobs of the pass:
    it is reported against the HBM roof (the binding one: PMC traffic puts it at 0.75 of 8 TB/s = 95 % of the ~6.3 TB/s a
    streaming kernel reaches on this part, profiles/), with its D.X flops beside it as `mfma_roof` -- ALGORITHMIC flops, of
    which the even/odd split executes half."""
    c5 = key == "c5"
    per_node = ALG_BYTES[key] if (c5 or n_obs in (0, 20)) else 560 + 24 * n_obs
    alg_bytes = per_node * B * M                 # SURVEY.md 8d bytes/node-eval x node-evals per launch (this rank)
    flops = 2.0 * M * ns * B * M                 # SURVEY.md 8d D.X flops/node-eval (2*M*ns) x node-evals
    node_name = "emi_nodes_kernel"
    peak_tf = FP32_MFMA_PEAK_TF if c5 else FP64_MFMA_PEAK_TF
    dom_s = m["dominant_ms"] * 1e-3
    one_launch = "emi_pass_f64_kernel" in def_name              # (the fp32 one-launch pass is reported against the MFMA roof, below)
    kname = node_name if m["dominant"] == "node" else def_name.split("<")[0].split(" ")[0]
    traffic, src = load_pmc_traffic(kname, B, M)
    if m["dominant"] == "node" or one_launch:
        ach = alg_bytes / dom_s / 1e9
        roof = {"kernel": node_name if m["dominant"] == "node" else def_name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "avg_ms": m["dominant_ms"],
                "algorithmic_bytes_per_launch": alg_bytes}
        if one_launch:
            tf = flops / dom_s / 1e12
            roof["mfma_roof"] = {"bound": "mfma", "achieved_algorithmic": tf, "peak": peak_tf, "unit": "TFLOP/s",
                                 "frac_algorithmic": tf / peak_tf, "executed_over_algorithmic": 0.5,
                                 "frac_executed": 0.5 * tf / peak_tf,
                                 "note": "D.X flops 2*M*ns per node-eval are ALGORITHMIC; the even/odd split of the LGL matrix executes half of them"}
    else:
        ach = flops / dom_s / 1e12
        roof = {"kernel": def_name, "bound": "mfma", "achieved": ach, "peak": peak_tf, "unit": "TFLOP/s",
                "frac": ach / peak_tf, "traffic": traffic, "avg_ms": m["dominant_ms"], "algorithmic_flops_per_launch": flops}
        if not c5:
            roof["executed_over_algorithmic"] = 0.5 if "symdefect" in def_name else 1.0
    roof["traffic_source"] = src if traffic is not None else ("none for this (kernel, B, M): PMC passes are separate rocprofv3 runs "
                                                              "(tools/pmc_traffic.sh), see profiles/")
    if traffic is not None:
        roof["traffic_frac_of_peak"] = traffic / dom_s / 1e9 / HBM_PEAK_GBS
        roof["traffic_over_algorithmic"] = traffic / alg_bytes
    roof["kernels_ms_warmup"] = {node_name: m["warm_node_ms"], def_name: m["warm_defect_ms"]}
    roof["concurrent"] = m["overlapped"]
    roof["pass_hbm_frac"] = alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
    roof["note"] = ("avg_ms: HIP events on the launch stream -- around the dominant kernel in every pass of the timed region, "
                    "or (one-launch pass: one kernel per pass) around the whole timed region / steps; "
                    "kernels_ms_warmup: the kernels bracketed during the warm-up passes")
    return roof


PRE_WARM_S = 0.25      # untimed clock warm-up ahead of the W warm-up steps (see measure())


def launch_ranks(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as children of this (GPU-free) process the
    way the driver's multi-GPU command does, relay what they print, return the launcher's exit code.  With fewer devices
    visible than ranks the ranks share devices (rank r -> device r mod visible; main() then puts the control plane on gloo,
    because RCCL refuses two ranks on one device) and the bench line says so (config.ranks_share_device)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def measure(ev, dX, dU, outs, steps, warmup, barrier, torch):
    """W untimed passes (profiled at level 1 to find the dominant kernel), then exactly `steps` passes between
    barrier + synchronize, with ONE pair of HIP events per pass around that kernel on its own launch stream (one-launch
    pass: one pair around the whole region)."""
    # Before the W warm-up steps: untimed passes until PRE_WARM_S of wall time have gone by.  A pass of the 128-instance
    # shard takes 0.04 ms, so W = 20 and K = 200 are 9 ms of work in all, less than the device needs to reach its clocks
    # under load: measured 0.0441 ms per step with K = 200 against 0.0406 with K = 2000 (B = 1024: 0.2351 / 0.2266).
    tw = time.perf_counter()
    while time.perf_counter() - tw < PRE_WARM_S:
        for _ in range(50):
            ev.eval_dev(dX, dU, *outs)
        torch.cuda.synchronize()
    ev.profile(1)
    for _ in range(max(warmup, 1)):
        ev.eval_dev(dX, dU, *outs)
    torch.cuda.synchronize()
    p = ev.profile_read()
    node_w = p["node_ms"] / max(p["node_launches"], 1)
    def_w = p["defect_ms"] / max(p["defect_launches"], 1)
    level = 2 if def_w >= node_w else 3
    # The one-launch pass is ONE kernel per pass: its average duration comes from two HIP events on the launch stream
    # around the whole timed region (emi_timer_start / emi_timer_stop), not from a bracket per pass -- two event records
    # per pass cost ~8 us of a 44 us pass at 128 instances.  (The span includes the gaps between consecutive launches,
    # 2-3 us each: it overstates the kernel's duration a little, never understates it.)
    one_launch = "emi_pass_f" in ev.last_defect_kernel          # emi_pass_f64_kernel / emi_pass_f32_kernel
    ev.profile(0 if one_launch else level)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if one_launch:
        ev.timer_start()
    for _ in range(steps):
        ev.eval_dev(dX, dU, *outs)
    span_ms = ev.timer_stop() if one_launch else None      # (synchronises the evaluator's stream)
    torch.cuda.synchronize()
    t1 = time.perf_counter()      # this rank's K steps are done; the closing barrier follows (its own latency -- an all-reduce
    barrier()                     # over 8 GPUs is as long as a 128-instance pass -- is not part of the work; MAX over ranks below)
    if one_launch:
        dom_ms, level = span_ms / steps, 2
    else:
        q = ev.profile_read()
        ev.profile(0)
        dom_ms = (q["defect_ms"] / max(q["defect_launches"], 1)) if level == 2 else (q["node_ms"] / max(q["node_launches"], 1))
    return dict(seconds=t1 - t0, dominant="defect" if level == 2 else "node", dominant_ms=dom_ms,
                warm_node_ms=node_w, warm_defect_ms=def_w, overlapped=p["overlapped_passes"] > 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c3", choices=["c3", "c5"])
    ap.add_argument("--scenarios", type=int, default=1024, help="size of the strong-scaled batch (config 4: 1024)")
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (overrides --scenarios: weak scaling)")
    ap.add_argument("--nodes", type=int, default=0)
    ap.add_argument("--obstacles", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak", action="store_true", help="skip the second, weak-scaled measurement of a multi-GPU run")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-5 (fp32) measurement appended to the default run")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the single-GPU run (`python bench.py --gpus N`): this process becomes the launcher.  Decided BEFORE
        # anything touches the GPU, and the ranks are fresh CHILD processes of `python -m torch.distributed.run` (a process that
        # has initialised the GPU must never be replaced by another program on this pool); rank 0's JSON line is relayed.
        sys.exit(launch_ranks(a.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # ranks that must share a device (fewer GPUs visible than ranks: the launcher above sets these; EMI_BENCH_SHARE_GPU is the
    # older rehearsal knob): RCCL refuses two ranks on one device, so the control plane goes over gloo, and the line says so
    backend = os.environ.get("EMI_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    shared = world > max(ndev, 1) or os.environ.get("EMI_BENCH_SHARE_GPU") == "1"
    if os.environ.get("EMI_BENCH_SHARE_GPU") == "1":
        local = 0
    elif local >= max(ndev, 1):
        local = local % ndev
    if shared and backend == "nccl" and world > 1:
        backend = "gloo"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    ctl = dev if backend == "nccl" else torch.device("cpu")   # where control-plane tensors live

    import etol_amd as E
    from etol_amd import batch as shard
    from etol_amd import workloads as W

    c5 = a.config == "c5"
    M = a.nodes or (4096 if c5 else 1024)
    n_obs = 0 if c5 else a.obstacles
    ns = 12 if c5 else 6

    def barrier():
        if world > 1:
            dist.barrier()

    def build(B, first, c5=c5, M=M, n_obs=n_obs):
        """evaluator + resident inputs for scenarios [first, first + B)"""
        dt = torch.float32 if c5 else torch.float64
        ev = E.Evaluator(local, f32=c5)
        ev.set_mesh(M, 0.0, 20.0 if c5 else W.TF)
        if c5:
            ev.set_model(E.MODEL_FIXEDWING12, W.FW_PARAMS)
        elif os.environ.get("EMI_BENCH_TRACED", "0") == "1":
            # A/B knob: the same model written as mi355x::Var arithmetic, differentiated and compiled at run time
            import ctypes
            hl = ctypes.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
            hl.harness_traced_model_source.restype = ctypes.c_char_p
            ev.set_model_source("TracedModel", hl.harness_traced_model_source(0).decode(), 6, 2)
        else:
            ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
        ev.set_batch(B)
        if os.environ.get("EMI_OVERLAP", "1") == "0":
            ev.set_option("overlap", 0)        # A/B switch: sequential general path
        for opt in ("sym_ct", "overlap_mode", "sym_order", "sym_ablate", "cu_split", "node_store", "sym_cpart", "sym_ksplit", "sym_nst", "f32_ring", "f32_ring_wgs", "sym_bk", "sym_ctc"):   # experiment knobs
            if os.environ.get("EMI_" + opt.upper()):
                ev.set_option(opt, int(os.environ["EMI_" + opt.upper()]))
        if c5:
            gen = min(B, 16)                     # 16 distinct instances tiled: the kernels do not care, the host generator does
            X, U = W.fixedwing_batch(4, gen, M, first_instance=first)
            X, U = np.tile(X, ((B + gen - 1) // gen, 1, 1))[:B], np.tile(U, ((B + gen - 1) // gen, 1, 1))[:B]
            recs = None
        else:
            X, U, recs = W.quadrotor_batch(3, B, M, n_obs, first_instance=first)   # scenario s -> its own obstacle field
        if n_obs:
            ev.set_path(recs, 0, 1)
        dX = torch.from_numpy(X).to(dev, dt)
        dU = torch.from_numpy(U).to(dev, dt)
        outs = ev.alloc_outputs()
        torch.cuda.synchronize()
        return ev, dX, dU, outs, (X, U, recs)

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=ctl)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- the headline measurement: the S-scenario batch strong-scaled over the ranks (or --batch per GPU) -----------
    if a.batch:
        lo, hi, S, scaling = rank * a.batch, (rank + 1) * a.batch, a.batch * world, "weak"
    elif c5:
        lo, hi, S, scaling = rank * 256, (rank + 1) * 256, 256 * world, "weak"
    else:
        S, scaling = a.scenarios, "strong"
        lo, hi = shard.shard_range(S, world, rank)
    B = hi - lo
    ev, dX, dU, outs, host = build(B, lo)
    m = measure(ev, dX, dU, outs, a.steps, a.warmup, barrier, torch)
    el = reduce_max(m["seconds"])
    ms_per_step = 1e3 * el / a.steps
    value = S * M * a.steps / el
    def_name = ev.last_defect_kernel

    # the path's one collective: gather the trajectories to rank 0 (once per batch, untimed)
    gather_ms = None
    if world > 1:
        gX, gU = (dX, dU) if backend == "nccl" else (dX.cpu(), dU.cpu())
        torch.cuda.synchronize()
        tg = time.perf_counter()
        got = shard.gather_trajectories(gX, gU, S, dst=0)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - tg)
        if rank == 0:
            assert got[0].shape[0] == S and torch.equal(got[0][:B].to(dX.device), dX)
    checked = None
    if rank == 0 and world == 1 and not c5 and not a.no_cpu_baseline:
        # untimed: what the timed passes left in HBM, against the CPU oracle on sampled instances (first / last of the batch,
        # both sides of a 16-instance tile edge, the middle)
        checked = check_against_oracle(ev, host[0], host[1], host[2] if n_obs else None, outs, M, [0, 15, 16, B // 2, B - 1])
    ev.close()
    del dX, dU, outs, host
    torch.cuda.empty_cache()

    weak = None
    if world > 1 and scaling == "strong" and not a.no_weak:
        Bw = a.scenarios
        evw, wX, wU, wouts, _ = build(Bw, rank * Bw)
        mw = measure(evw, wX, wU, wouts, a.steps, a.warmup, barrier, torch)
        elw = reduce_max(mw["seconds"])
        weak = {"value": world * Bw * M * a.steps / elw, "unit": "node-evals/s", "instances_per_gpu": Bw,
                "ms_per_step": 1e3 * elw / a.steps, "scaling": "weak"}
        evw.close()

    if rank == 0:
        # ---- roofline of the dominant kernel: algorithmic work per launch / its average launch time in the timed region
        roof = roofline_of(m, def_name, "c5" if c5 else ("c3" if n_obs == 20 else "c2"), n_obs, B, M, ns, ms_per_step)
        if c5:
            workload = (f"config[4]: 12-state fixed-wing VGP, N={M} LGL nodes, fp32 path with MFMA D.X defect, "
                        f"{B} instances per GPU")
            metric = "collocation-node constraint+Jacobian evals/sec, 12-state VGP N=4096 (fp32, config 5)"
        else:
            workload = (f"config[2]: 6-state quadrotor VGP, N={M} LGL nodes + {n_obs} static keep-outs; "
                        + (f"the {S}-scenario obstacle-field batch of config[3] strong-scaled: {B} instances on this GPU"
                           if scaling == "strong" else f"{B} instances per GPU"))
            metric = "collocation-node constraint+Jacobian evals/sec, 6-state VGP N=1024"
        line = {
            "metric": metric,
            "value": value, "unit": "node-evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32" if c5 else "f64", "data": "synthetic",
            "config": {"workload": workload, "nodes": M, "scenarios": S, "instances_per_gpu": B,
                       "path_rows": n_obs, "parallelism": f"instances sharded x{world}", "gather_ms": gather_ms,
                       "pre_warm_s": PRE_WARM_S, "devices_visible": ndev, "ranks_share_device": bool(shared and world > 1),
                       "control_plane": backend if world > 1 else None},
            "roofline": roof,
        }
        if weak:
            line["weak_scaling"] = weak
        if checked is not None:
            line["checked"] = checked
        if world == 1 and not c5 and not a.batch and not a.nodes and not a.no_secondary:
            # config 5 (BASELINE.json configs[4]) on the same record: 12-state fixed wing, N = 4096, fp32 path with the MFMA
            # D.X defect, 256 instances, fewer steps (a pass is ~1 ms); its own roofline against the f32 MFMA peak and
            # 944 algorithmic bytes per node-eval
            B5, M5, K5, W5 = 256, 4096, max(10, min(a.steps, 50)), max(2, min(a.warmup, 5))
            ev5, X5, U5, outs5, _ = build(B5, 0, c5=True, M=M5, n_obs=0)
            m5 = measure(ev5, X5, U5, outs5, K5, W5, barrier, torch)
            ms5 = 1e3 * m5["seconds"] / K5
            name5 = ev5.last_defect_kernel
            r5 = roofline_of(m5, name5, "c5", 0, B5, M5, 12, ms5)
            r5["pass_mfma_frac"] = 2.0 * M5 * 12 * B5 * M5 / (ms5 * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF
            line["secondary"] = {"c5": {"metric": "collocation-node constraint+Jacobian evals/sec, 12-state VGP N=4096 (fp32, config 5)",
                                        "value": B5 * M5 * K5 / m5["seconds"], "unit": "node-evals/s", "ms_per_step": ms5, "steps": K5,
                                        "warmup": W5, "dtype": "f32", "data": "synthetic",
                                        "config": {"workload": f"config[4]: 12-state fixed-wing VGP, N={M5} LGL nodes, fp32 path with MFMA "
                                                               f"D.X defect, {B5} instances per GPU"},
                                        "roofline": r5}}
            ev5.close()
            del X5, U5, outs5
            torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline and not c5:
            line["cpu_baseline"] = cpu_baseline(M, n_obs, a.cpu_budget)
        bad = checked is not None and not checked["ok"]
        if bad:
            # a pass that left wrong results in HBM has no throughput: the number is withheld and the run fails
            line["value_withheld"] = line["value"]
            line["value"] = None
            line["error"] = "outputs of the timed passes disagree with oracle/emi_oracle.c (see `checked`)"
        print(json.dumps(line), flush=True)
        if bad:
            if world > 1:
                dist.destroy_process_group()
            sys.exit(3)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

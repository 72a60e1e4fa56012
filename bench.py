#!/usr/bin/env python3
"""bench.py -- collocation-node constraint+Jacobian evaluations per second on MI355X.

A "step" is ONE full evaluation pass of the hot path over the rank's batch of synthetic
problem instances (BASELINE.json config 3: 6-state quadrotor VGP, N=1024 LGL nodes, 20 static
keep-outs), producing for every node F, C, L, the defect row, dF, dC, dL with the Jacobian
values landed in the NLP value array and the cost reduced.
Inputs are resident in HBM before the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Instances shard across ranks with no data-path collective (weak scaling: B per GPU); the
only collective of the path, the gather of the trajectories, runs once after the timed region.
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


def load_pmc_traffic():
    """HBM bytes per launch from the PMC passes of tools/pmc_traffic.sh (FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE), if a summary of this round exists."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        return json.load(open(path)).get("bytes_per_launch", {})
    except (OSError, ValueError):
        return {}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="problem instances per GPU")
    ap.add_argument("--nodes", type=int, default=1024)
    ap.add_argument("--obstacles", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0)
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # rehearsal knobs (never set by the driver): control plane over gloo, all ranks on one GPU
    backend = os.environ.get("EMI_BENCH_BACKEND", "nccl")
    if os.environ.get("EMI_BENCH_SHARE_GPU") == "1":
        local = 0
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
    from etol_amd import workloads as W

    M, B, n_obs = a.nodes, a.batch, a.obstacles
    ev = E.Evaluator(local)
    ev.set_mesh(M, 0.0, W.TF)
    if os.environ.get("EMI_BENCH_TRACED", "0") == "1":
        # A/B knob: the same model written as mi355x::Var arithmetic, differentiated and compiled at run time
        # (text from the test harness); the default run uses the hand-written kernel instantiation
        import ctypes
        hl = ctypes.CDLL(os.path.join(ROOT, "tests", "harness", "libetol_harness.so"))
        hl.harness_traced_model_source.restype = ctypes.c_char_p
        ev.set_model_source("TracedModel", hl.harness_traced_model_source(0).decode(), 6, 2)
    else:
        ev.set_model(E.MODEL_QUADROTOR2D, W.QUAD_PARAMS)
    ev.set_batch(B)
    if os.environ.get("EMI_OVERLAP", "1") == "0":
        ev.set_option("overlap", 0)        # A/B switch: sequential general path
    for opt in ("sym_ct", "overlap_mode", "sym_order", "sym_ablate", "cu_split"):          # experiment knobs of the overlapped path
        if os.environ.get("EMI_" + opt.upper()):
            ev.set_option(opt, int(os.environ["EMI_" + opt.upper()]))
    X, U, recs = W.quadrotor_batch(3, B, M, n_obs, first_instance=rank * B)   # scenario s -> rank s // B
    if n_obs:
        ev.set_path(recs, 0, 1)
    dX = torch.from_numpy(X).to(dev)
    dU = torch.from_numpy(U).to(dev)
    RES, VALS, COST = ev.alloc_outputs()
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        ev.eval_dev(dX, dU, RES, VALS, COST)
    torch.cuda.synchronize()
    ev.profile(True)          # HIP-event brackets around each kernel, on the launch stream
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ev.eval_dev(dX, dU, RES, VALS, COST)
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    prof = ev.profile_read()
    ev.profile(False)

    el = torch.tensor([t1 - t0], dtype=torch.float64, device=ctl)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = float(el.item())
    ms_per_step = 1e3 * el / a.steps
    value = world * B * M * a.steps / el

    # the path's one collective: gather the trajectories to rank 0 (once per batch, untimed)
    gather_ms = None
    if world > 1:
        from etol_amd import batch
        gX, gU = (dX, dU) if backend == "nccl" else (dX.cpu(), dU.cpu())
        torch.cuda.synchronize()
        tg = time.perf_counter()
        got = batch.gather_trajectories(gX, gU, world * B, dst=0)
        torch.cuda.synchronize()
        gather_ms = 1e3 * (time.perf_counter() - tg)
        if rank == 0:
            assert got[0].shape[0] == world * B and torch.equal(got[0][:B].to(dX.device), dX)

    if rank == 0:
        # ---- roofline of the dominant kernel (longest average launch, HIP events on its own stream)
        key = "c3" if n_obs == 20 else "c2"
        per_node = ALG_BYTES[key] if n_obs in (0, 20) else 560 + 24 * n_obs
        alg_bytes = per_node * B * M                 # SURVEY.md 8d bytes/node-eval x node-evals per launch
        flops = 2.0 * M * 6 * B * M                  # SURVEY.md 8d D.X flops/node-eval (2*M*ns) x node-evals
        node_ms = prof["node_ms"] / max(prof["node_launches"], 1)
        def_ms = prof["defect_ms"] / max(prof["defect_launches"], 1)
        overlapped = prof["overlapped_passes"] > 0
        node_name = "emi_nodes_kernel"
        ring = os.environ.get("EMI_SYM_CT", "3") == "3"
        def_name = (("emi_symdefect_ring_f64_kernel" if ring else "emi_symdefect_f64_kernel") if overlapped
                    else "emi_defect_f64_kernel")
        traffic = load_pmc_traffic()
        if node_ms >= def_ms:
            ach = alg_bytes / (node_ms * 1e-3) / 1e9
            roof = {"kernel": node_name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic.get(node_name), "avg_ms": node_ms}
        else:
            ach = flops / (def_ms * 1e-3) / 1e12
            roof = {"kernel": def_name, "bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TF,
                    "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TF, "traffic": traffic.get(def_name),
                    "avg_ms": def_ms}
        roof["kernels_ms"] = {node_name: node_ms, def_name: def_ms}
        roof["concurrent"] = overlapped
        if overlapped:
            pass_ms = prof["pass_ms"] / prof["overlapped_passes"]
            roof["pass_ms"] = pass_ms                # fork -> both kernels -> join
            roof["pass_hbm_frac"] = alg_bytes / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        line = {
            "metric": "collocation-node constraint+Jacobian evals/sec, 6-state VGP N=1024",
            "value": value, "unit": "node-evals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"config[2]: 6-state quadrotor VGP, N={M} LGL nodes + {n_obs} static keep-outs, "
                                   f"{B} instances per GPU", "nodes": M, "instances_per_gpu": B,
                       "path_rows": n_obs, "parallelism": f"instances sharded x{world}", "gather_ms": gather_ms},
            "roofline": roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(M, n_obs, a.cpu_budget)
        print(json.dumps(line), flush=True)
    ev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

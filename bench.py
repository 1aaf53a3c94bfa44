#!/usr/bin/env python3
"""bench.py -- VB sweeps/s of the MI355X-native atlasqtl hot path.

Contract (see the task description): ``python bench.py --gpus N --steps K --warmup W``;
for N > 1 it is launched by ``python -m torch.distributed.run --nproc-per-node N``
(one rank per GPU, RCCL).  A *step* is one VB sweep = steps S1-S22 of SURVEY.md
section 3.2 (pre-pass, spike-and-slab core sweep, horseshoe/probit updates, and the
scheduled ELBO evaluations) on synthetic data of BASELINE.json's size:
n = 1000 samples, p = 50 000 SNPs, q = 10 000 traits, annealing schedule on.
W warm-up sweeps (default 10: the whole annealing ladder), then exactly K timed sweeps
between barrier + synchronize; time = max over ranks; rank 0 prints ONE JSON line.

With N > 1 the trait axis is sharded (q/N traits per rank, same total problem: strong
scaling); the only data-path exchange is one all-reduce of p+8 doubles per sweep (+ 8
doubles on ELBO sweeps).

Environment overrides for quick checks: AQ_BENCH_N / AQ_BENCH_P / AQ_BENCH_Q.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# FP64 peaks used for the roofline: AMD's public MI355X figure (vector = matrix FP64) and the
# v_mfma_f64_16x16x4_f64 rate measured on this pool with tools/microbench/f64_sustain.hip (77.8 TFLOP/s sustained over 27 ms)
# (MI355X_MICROARCH.md has no FP64 row) -- see DESIGN.md section 6.
PEAK_FP64_SPEC_TFLOPS = 78.6
PEAK_FP64_MFMA_MEASURED_TFLOPS = 77.8
# HBM-side bytes of ONE core-sweep launch at the default workload on one GPU, from separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this same command with the gfx950
# correction (FETCH_SIZE x 2 for coalesced streaming reads, calibrated on aq_k_prepass's gam read):
# profiles/r03_pmc_hbm_traffic.txt (the instance <9, 9, true, 2, false, 9> with 13 chained segments: reads 2 x 15,194,464 KiB + writes 9,118,912 KiB;
# the reads include the helper waves' L2 warm-up touches).  PMC counters cannot be read from inside the timed run, so this is the
# profile's number for the same kernel and workload, not a value measured in this run.
PMC_TRAFFIC_C3_BYTES = 4.05e10
# the same workload with 5 % of Y missing (AQ_BENCH_NA=0.05: the MASK instance of the look-ahead kernel, which streams the
# traits' own Gram blocks, 98 GB per sweep): profiles/r03_pmc_hbm_traffic_c3_na5.txt (2 x 64,915,347 KiB + 8,789,477 KiB)
PMC_TRAFFIC_C3_NA5_BYTES = 1.42e11


def build_problem(n, p, q_total, k0, k1, device, seed=123):
    """Synthetic hotspot-QTL data (shape of R/atlasqtl.R:125-157), identical for every
    rank count: the int8 genotypes, the q-vectors and Y are generated in full from fixed seeds and
    sliced; X is standardised on the GPU from the dosages and returned as a PreparedData (it never
    exists as fp64 on the host); the p x q initial values are generated on the GPU per 16-trait tile
    from a seed that depends only on the global tile index."""
    import torch
    from atlasqtl_amd import hyper_init as H
    from atlasqtl_amd.prepare import prepare_on_device

    rng = np.random.default_rng(seed)
    maf = rng.uniform(0.05, 0.5, size=p)
    # genotypes as int8 dosages (1 byte each); scale(X) and the constant / duplicate-column check run on the GPU
    # (aq_prepare_data) and the standardised fp64 matrix only ever exists there
    G = np.asfortranarray(rng.binomial(2, maf[None, :], size=(n, p)).astype(np.int8))
    for _ in range(4):
        Xprep, cst, coll, _dup = prepare_on_device(np.arange(n, dtype=np.float64).reshape(n, 1), G, device)   # (Y is prepared below, per rank)
        bad = cst | coll
        if not bad.any():
            break
        Xprep.close()                   # a constant or duplicated column (very unlikely at n = 1000): make it a fresh SNP
        G[:, bad] = rng.binomial(2, 0.3, size=(n, int(bad.sum()))).astype(np.int8)
    else:
        raise RuntimeError("could not draw a genotype matrix without constant / duplicated columns")

    def std_cols(idx):                  # the few columns the simulated effects need, standardised on the host
        c = G[:, idx].astype(np.float64)
        c -= c.mean(axis=0)
        return c / np.sqrt((c ** 2).sum(axis=0) / (n - 1))
    p_act, q_act = 40, max(16, q_total // 4)
    act_x = np.sort(rng.choice(p, size=p_act, replace=False))
    act_y = np.sort(rng.choice(q_total, size=q_act, replace=False))
    beta = np.where(rng.random((p_act, q_act)) < 0.2, rng.normal(size=(p_act, q_act)), 0.0) * 0.3
    Y = rng.normal(size=(n, q_total))
    Y[:, act_y] += std_cols(act_x) @ beta
    Y -= Y.mean(axis=0)
    p0 = (5.0, 25.0)
    lh = H.auto_set_hyper_(Y, p, p0)
    na = float(os.environ.get("AQ_BENCH_NA", "0"))   # fraction of Y missing completely at random (C5's missingness mask)
    if na > 0:
        Y[rng.random(Y.shape) < na] = np.nan
        Y -= np.nanmean(Y, axis=0)                    # centring on the observed entries, R/prepare_atlasqtl.R:70-73
    irng = np.random.default_rng(seed + 333)
    t02, n0 = lh["t02"], float(lh["n0"][0])
    tau = 1.0 / float(np.median(np.nanvar(Y, axis=0, ddof=1)))
    sig02_inv = float(irng.gamma(shape=max(p, q_total)))
    li = dict(sig02_inv_vb=sig02_inv,
              sig2_beta_vb=1.0 / irng.gamma(shape=2.0, scale=1e-2 * tau, size=q_total),
              sig2_theta_vb=1.0 / (q_total + irng.gamma(shape=sig02_inv * q_total, size=p)),
              tau_vb=np.full(q_total, tau),
              theta_vb=irng.normal(scale=1.0 / np.sqrt(sig02_inv * q_total), size=p),
              zeta_vb=irng.normal(loc=n0, scale=np.sqrt(t02), size=q_total))
    sl = slice(k0, k1)
    lh_loc = dict(lh)
    for key in ("eta", "kappa", "n0"):
        lh_loc[key] = np.ascontiguousarray(lh[key][sl])
    li_loc = dict(li)
    for key in ("sig2_beta_vb", "tau_vb", "zeta_vb"):
        li_loc[key] = np.ascontiguousarray(li[key][sl])
    # p x q initial values on the device, (q_loc, p) contiguous == p x q_loc column-major
    q_loc = k1 - k0
    dev = torch.device("cuda", device)
    gam = torch.empty((q_loc, p), dtype=torch.float64, device=dev)
    mu = torch.empty((q_loc, p), dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev)
    for t0 in range(k0 - k0 % 16, k1, 16):
        gen.manual_seed(seed * 1000003 + t0 // 16)
        z = torch.randn((2, 16, p), dtype=torch.float64, device=dev, generator=gen)
        lo, hi = max(t0, k0), min(t0 + 16, k1)
        gam[lo - k0:hi - k0] = torch.special.ndtr(n0 + (1e-4 + t02) * z[0, lo - t0:hi - t0])   # R/set_hyper_init.R:385
        mu[lo - k0:hi - k0] = z[1, lo - t0:hi - t0]                                           # :387
    li_loc["gam_vb"], li_loc["mu_beta_vb"] = gam, mu
    return Xprep, np.asfortranarray(Y[:, sl]), lh_loc, li_loc


def cpu_baseline(n, p, q_total, seed=7, budget_s=10.0):
    """CPU figures for step S9 (the core loop) on the GPU box's host cores, all from the oracle's C restatement
    (oracle/core_loop_oracle.c, gcc -O2 -ffp-contract=off):
      value      n-space port of src/coreLoop.cpp:38-86, ONE thread (the reference is single-threaded, src/Makevars),
                 on a trait sub-sample at full n and p, scaled by q/q_sub (traits are independent and equal-cost);
      all_cores  the same port with the traits spread over every host core (threads; ctypes releases the GIL);
      gram_space the reference's own formulation (p-long AXPY on X'X per (SNP, trait), :58-84), one thread, at a reduced
                 p_sub where X'X fits comfortably, scaled by (p/p_sub)^2 q/q_sub -- an extrapolation, labelled as such.
    Covers S9 only, i.e. it flatters the CPU (the R-side p x q passes come on top)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import atlasqtl_oracle as O
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.normal(size=(n, p)))

    def state(pp, qs):
        gam = np.asfortranarray(rng.uniform(0.001, 0.01, size=(pp, qs)))
        mu = np.asfortranarray(rng.normal(size=(pp, qs)) * 0.01)
        return (gam, mu, np.asfortranarray(gam * mu), np.asfortranarray(np.full((pp, qs), -6.0)),
                np.asfortranarray(np.full((pp, qs), -0.0025)))

    def nspace(qs, threads=1):
        R = np.asfortranarray(rng.normal(size=(n, qs)))
        gam, mu, m1, lP, l1 = state(p, qs)
        args = (X, R, None, np.full(p, n - 1.0), gam, lP, l1, -0.3, np.zeros(qs), m1, mu, np.full(qs, 1e-3), np.ones(qs), 1.0)
        t0 = time.perf_counter()
        if threads == 1:
            O.nspace_loop(*args)
        else:
            cuts = [qs * i // threads for i in range(threads + 1)]
            with ThreadPoolExecutor(threads) as ex:
                list(ex.map(lambda i: O.nspace_loop(*args, k_begin=cuts[i], k_end=cuts[i + 1]), range(threads)))
        return time.perf_counter() - t0

    q_sub = 4
    t = nspace(q_sub)
    per_trait = t / q_sub
    extra = int(max(0, min(4096, (budget_s - t) / max(per_trait, 1e-9))))
    if extra >= 4:
        t2 = nspace(extra)
        per_trait = (t + t2) / (q_sub + extra)
        q_sub += extra
    out = {"value": 1.0 / (per_trait * q_total), "unit": "sweeps/s", "cores": 1, "kind": "port",
           "sample": f"oracle n-space core loop (step S9 only), {q_sub} of {q_total} traits at full n={n}, p={p}, "
                     f"{per_trait * q_sub:.1f} s measured, scaled by q/q_sub"}
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:   # a container's CPU share (cgroup v2 quota) is what the threads can really use
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    cores = min(cores, 64)
    if cores > 1:
        qs = max(cores, min(4096, int(cores * 6.0 / max(per_trait, 1e-9))))     # ~6 s of wall-clock
        qs -= qs % cores
        tm = nspace(qs, cores)
        out["all_cores"] = {"value": 1.0 / (tm / qs * q_total), "unit": "sweeps/s", "cores": cores,
                            "sample": f"same port, {qs} traits over {cores} threads, {tm:.1f} s"}
    p_sub, qg = min(p, 4000), 8
    Xs = X[:, :p_sub]
    cp_X = np.asfortranarray(Xs.T @ Xs)
    gam, mu, m1, lP, l1 = state(p_sub, qg)
    cp_Y_X = np.asfortranarray(rng.normal(size=(qg, p_sub)))
    cpb = np.asfortranarray(cp_X @ m1)
    t0 = time.perf_counter()
    O.core_dual_loop(cp_X, cp_Y_X, gam, lP, l1, -0.3, np.zeros(qg), m1, cpb, mu, np.full(qg, 1e-3), np.ones(qg),
                     np.arange(p_sub, dtype=np.int32), np.arange(qg, dtype=np.int32), 1.0)
    tg = time.perf_counter() - t0
    out["gram_space"] = {"value": 1.0 / (tg * (p / p_sub) ** 2 * q_total / qg), "unit": "sweeps/s", "cores": 1,
                         "sample": f"the reference's formulation (src/coreLoop.cpp:58-84 restated), p_sub={p_sub}, {qg} traits, "
                                   f"{tg:.2f} s, EXTRAPOLATED by (p/p_sub)^2 q/q_sub (X'X at full p would need {8e-9 * p * p:.0f} GB)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-to-tol", action="store_true",
                    help="skip the ELBO-to-tol leg (N = 1 only): the reference's default run (tol 0.1, maxit 1000, anneal (1,2,10), "
                         "thinned ELBO schedule) from loop entry to `converged`, on C2 (n=1000 p=5000 q=1000) and on the bench workload")
    ap.add_argument("--to-tol-maxit", type=int, default=1000,
                    help="maxit of the ELBO-to-tol leg on the bench workload (the reference's default is 1000, R/atlasqtl.R:181; C3 "
                         "needs more sweeps than that for |dELBO| < 0.1 on an ELBO of -1.4e7: profiles/r03_elbo_to_tol_c3.json)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from atlasqtl_amd.core import VbRun

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    pg = None
    if world > 1 or os.environ.get("AQ_BENCH_FORCE_PG") == "1":   # the env switch rehearses the RCCL path on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        pg = dist.group.WORLD

    n = int(os.environ.get("AQ_BENCH_N", 1000))
    p = int(os.environ.get("AQ_BENCH_P", 50000))
    q = int(os.environ.get("AQ_BENCH_Q", 10000))
    # shard traits in whole 16-trait tiles
    tiles = (q + 15) // 16
    t_lo = (tiles * rank) // world
    t_hi = (tiles * (rank + 1)) // world
    k0, k1 = min(q, 16 * t_lo), min(q, 16 * t_hi)

    t_setup = time.time()
    X, Y, lh, li = build_problem(n, p, q, k0, k1, local_rank)
    anneal = (1, 2, 10)
    run = VbRun(Y, X, lh, li, anneal, tol=1e-12, maxit=args.warmup + args.steps + 5, thinned_elbo_eval=True,
                debug=False, device=local_rank, q_total=q, process_group=pg, trait_offset=k0)
    del li
    torch.cuda.empty_cache()
    t_setup = time.time() - t_setup

    def sync():
        torch.cuda.synchronize()
        if pg is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_warm = None
    if args.warmup > 0:
        sync()
        tw = time.perf_counter()
        run.run_sweeps(args.warmup)
        sync()
        t_warm = time.perf_counter() - tw
    st0 = run.status()
    sync()
    t0 = time.perf_counter()
    run.run_sweeps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if pg is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", local_rank))
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    st1 = run.status()
    sweeps_done = st1["it"] - st0["it"]
    assert sweeps_done == args.steps, (sweeps_done, args.steps)

    if rank == 0:
        core_ms = (st1["core_ms"] - st0["core_ms"]) / max(st1["core_launches"] - st0["core_launches"], 1)
        q_loc = k1 - k0
        flop_per_launch = 4.0 * n * p * q_loc                  # W_f = 4 n p q (this rank's traits)
        achieved = flop_per_launch / (core_ms * 1e-3) / 1e12
        # gam, mu read and written once, X once, the residual once; with missing values also the traits' own diagonal
        # and cross Gram blocks (AQ_GK_STRIDE = 6272 doubles per trait tile and SNP block) and the mask
        algo_bytes = 32.0 * p * q_loc + 8.0 * n * p + 16.0 * n * q_loc
        na = float(os.environ.get("AQ_BENCH_NA", "0") or 0)
        if na > 0 and st1["core_kernel"] == 0:
            algo_bytes += 8.0 * 6272 * ((q_loc + 15) // 16) * ((p + 15) // 16) + 1.0 * n * q_loc
        pmc_traffic = None
        if (n, p, q, world) == (1000, 50000, 10000, 1) and st1["core_kernel"] == 0:
            pmc_traffic = PMC_TRAFFIC_C3_BYTES if na == 0 else (PMC_TRAFFIC_C3_NA5_BYTES if na == 0.05 else None)
        out = {
            "metric": "VB sweeps/sec", "value": args.steps / dt, "unit": "sweeps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"atlasqtl VB sweep (S1-S22 incl. scheduled ELBO), n={n} p={p} q={q}, "
                                   f"anneal=(1,2,10) on, horseshoe global-local, sweeps {st0['it'] + 1}-{st1['it']}",
                       "n": n, "p": p, "q": q, "q_per_gpu": q_loc, "parallelism": f"trait-sharded x{world}",
                       "elbo_evals_in_timed_region": st1["n_elbo"] - st0["n_elbo"], "elbo_last": st1["lb_opt"],
                       # the warm-up sweeps are the annealing ladder (the probit terms are evaluated twice there)
                       "annealed_sweeps_per_s": (args.warmup / t_warm) if t_warm else None,
                       "setup_s": round(t_setup, 1),
                       # the host's launch plan of the core kernel on rank 0 (aq_vb_status)
                       "launch": {k: st1[k] for k in ("core_kernel", "split_parts", "tiles_per_group", "chain_segments")},
                       "env_overrides": st1["overrides"]},
            "roofline": {"bound": "mfma", "kernel": {0: "aq_core_sweep_la_kernel", 2: "aq_trait_wave_kernel",
                                                     3: "aq_core_sweep_mis_kernel"}[st1["core_kernel"]], "achieved": achieved,
                         "peak": PEAK_FP64_SPEC_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_SPEC_TFLOPS,
                         "traffic": pmc_traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 PMC passes of the same command, profiles/r03_pmc_hbm_traffic*.txt: a profile constant, not measured in this run)",
                         "algorithmic_bytes": algo_bytes,
                         "peak_measured_mfma_f64": PEAK_FP64_MFMA_MEASURED_TFLOPS,
                         "frac_of_measured_mfma_peak": achieved / PEAK_FP64_MFMA_MEASURED_TFLOPS,
                         "kernel_ms_avg": core_ms, "flop_per_launch": flop_per_launch},
        }
        rf = out["roofline"]
        if rf["traffic"] is not None:   # HBM GB/s of the same kernel against the 8 TB/s peak (the north star asks for it)
            rf["hbm_GBps"] = rf["traffic"] / (core_ms * 1e-3) / 1e9
            rf["hbm_peak_GBps"] = 8000.0
            rf["hbm_frac"] = rf["hbm_GBps"] / 8000.0
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, p, q)
            except Exception as e:  # the baseline is a report, never a reason to lose the measurement
                out["cpu_baseline"] = {"value": None, "unit": "sweeps/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        if world == 1 and not args.no_to_tol:
            # second half of BASELINE.json's metric: ELBO-to-tol wall-clock (loop entry -> converged; defaults of
            # R/atlasqtl.R:179-182).  C2 converges; the C3-sized workload reaches maxit = 1000 first (reported as such).
            run.close()
            run = None
            out["elbo_to_tol"] = {}
            for name, (n2, p2, q2) in (("c2", (1000, 5000, 1000)), ("bench_workload", (n, p, q))):
                X2, Y2, lh2, li2 = build_problem(n2, p2, q2, 0, q2, local_rank)
                maxit2 = 1000 if name == "c2" else args.to_tol_maxit
                r2 = VbRun(Y2, X2, lh2, li2, anneal, tol=0.1, maxit=maxit2, thinned_elbo_eval=True, debug=True,
                           device=local_rank, q_total=q2)
                del li2
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r2.run()
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                st2 = r2.status()
                out["elbo_to_tol"][name] = {"workload": f"n={n2} p={p2} q={q2}", "seconds": dt2, "it": st2["it"],
                                            "converged": bool(st2["converged"]), "tol": 0.1, "maxit": maxit2,
                                            "lb_opt": st2["lb_opt"], "sweeps_per_s": st2["it"] / dt2}
                r2.close()
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
    if run is not None:
        run.close()
    if pg is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

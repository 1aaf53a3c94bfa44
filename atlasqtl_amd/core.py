"""Host-side mirror of the reference's operator interface for the VB hot path.

Same names, argument meaning and error behaviour as the reference's R functions,
with the arithmetic done by libatlasqtl_hip.so on the GPU:

  coreDualLoop(...) / coreDualMisLoop(...)     R/RcppExports.R:4-10  (in place, as the reference)
  atlasqtl_global_local_core_(...)             R/atlasqtl_global_local_core.R:8-433

One process drives one GPU.  With a torch.distributed process group the trait
axis q is sharded over the ranks: each rank passes its own columns of Y and of
the q-indexed hyper-parameters / initial values; the only exchange per sweep is
one SUM all-reduce (RCCL over xGMI) of aq_vb_reduce_len(p) doubles, plus one of
8 doubles on the sweeps where the ELBO is evaluated.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import AqVbProblem, AqVbStatus, as_dp, as_ip, check, lib


def _f64F(a, name, shape=None):
    a = np.asarray(a)
    if a.dtype != np.float64 or not a.flags.f_contiguous:
        raise TypeError(f"{name} must be a float64 Fortran-ordered (R layout) array; it is updated in place")
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {shape}, got {a.shape}")
    return a


def _vec(a, name, n):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != (n,):
        raise ValueError(f"{name} must have length {n}")
    return a


def coreDualLoop(cp_X, cp_Y_X, gam_vb, log_Phi_theta_plus_zeta, log_1_min_Phi_theta_plus_zeta, log_sig2_inv_vb,
                 log_tau_vb, m1_beta, cp_betaX_X, mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind, sample_q, c=1.0):
    """R/RcppExports.R:4-6 -> src/coreLoop.cpp:38-86.  gam_vb, m1_beta, cp_betaX_X, mu_beta_vb are
    modified in place (they must be float64 Fortran-ordered p x q arrays); returns None."""
    gam_vb = _f64F(gam_vb, "gam_vb")
    p, q = gam_vb.shape
    cp_X = _f64F(np.asfortranarray(cp_X, dtype=np.float64), "cp_X", (p, p))
    cp_Y_X = _f64F(np.asfortranarray(cp_Y_X, dtype=np.float64), "cp_Y_X", (q, p))
    lP = _f64F(np.asfortranarray(log_Phi_theta_plus_zeta, dtype=np.float64), "log_Phi_theta_plus_zeta", (p, q))
    l1 = _f64F(np.asfortranarray(log_1_min_Phi_theta_plus_zeta, dtype=np.float64), "log_1_min_Phi_theta_plus_zeta",
               (p, q))
    m1_beta = _f64F(m1_beta, "m1_beta", (p, q))
    cp_betaX_X = _f64F(cp_betaX_X, "cp_betaX_X", (p, q))
    mu_beta_vb = _f64F(mu_beta_vb, "mu_beta_vb", (p, q))
    lt = _vec(log_tau_vb, "log_tau_vb", q)
    s2 = _vec(sig2_beta_vb, "sig2_beta_vb", q)
    tv = _vec(tau_vb, "tau_vb", q)
    si = np.ascontiguousarray(shuffled_ind, dtype=np.int32)
    sq = np.ascontiguousarray(sample_q, dtype=np.int32)
    rc = lib().aq_core_dual_loop(as_dp(cp_X), as_dp(cp_Y_X), as_dp(gam_vb), as_dp(lP), as_dp(l1),
                                 float(log_sig2_inv_vb), as_dp(lt), as_dp(m1_beta), as_dp(cp_betaX_X),
                                 as_dp(mu_beta_vb), as_dp(s2), as_dp(tv), as_ip(si), len(si), as_ip(sq), len(sq),
                                 float(c), p, q)
    check(rc, "coreDualLoop")


def coreDualMisLoop(cp_X, cp_X_rm, cp_Y_X, gam_vb, log_Phi_theta_plus_zeta, log_1_min_Phi_theta_plus_zeta,
                    log_sig2_inv_vb, log_tau_vb, m1_beta, cp_betaX_X, mu_beta_vb, sig2_beta_vb, tau_vb, shuffled_ind,
                    sample_q, c=1.0):
    """R/RcppExports.R:8-10 -> src/coreLoop.cpp:91-138.  cp_X_rm: list of q (p x p) matrices;
    sig2_beta_vb: p x q.  In place like coreDualLoop."""
    gam_vb = _f64F(gam_vb, "gam_vb")
    p, q = gam_vb.shape
    if len(cp_X_rm) != q:
        raise ValueError("cp_X_rm must be a list of q matrices")
    cp_X = np.asfortranarray(cp_X, dtype=np.float64)
    rms = [_f64F(np.asfortranarray(m, dtype=np.float64), "cp_X_rm[[k]]", (p, p)) for m in cp_X_rm]
    arr = (_lib.dp * q)(*[as_dp(m) for m in rms])
    cp_Y_X = np.asfortranarray(cp_Y_X, dtype=np.float64)
    lP = np.asfortranarray(log_Phi_theta_plus_zeta, dtype=np.float64)
    l1 = np.asfortranarray(log_1_min_Phi_theta_plus_zeta, dtype=np.float64)
    m1_beta = _f64F(m1_beta, "m1_beta", (p, q))
    cp_betaX_X = _f64F(cp_betaX_X, "cp_betaX_X", (p, q))
    mu_beta_vb = _f64F(mu_beta_vb, "mu_beta_vb", (p, q))
    s2 = _f64F(np.asfortranarray(sig2_beta_vb, dtype=np.float64), "sig2_beta_vb", (p, q))
    lt = _vec(log_tau_vb, "log_tau_vb", q)
    tv = _vec(tau_vb, "tau_vb", q)
    si = np.ascontiguousarray(shuffled_ind, dtype=np.int32)
    sq = np.ascontiguousarray(sample_q, dtype=np.int32)
    rc = lib().aq_core_dual_mis_loop(as_dp(cp_X), arr, as_dp(cp_Y_X), as_dp(gam_vb), as_dp(lP), as_dp(l1),
                                     float(log_sig2_inv_vb), as_dp(lt), as_dp(m1_beta), as_dp(cp_betaX_X),
                                     as_dp(mu_beta_vb), as_dp(s2), as_dp(tv), as_ip(si), len(si), as_ip(sq), len(sq),
                                     float(c), p, q)
    check(rc, "coreDualMisLoop")


def _build_problem(Y, X, prep, n, p, q, q_total, list_hyper, list_init, anneal, tol, maxit, thinned_elbo_eval, debug, device,
                   world, trait_offset, scheme, df, ext_main=None, ext_elbo=None):
    """Fill an aq_vb_problem (include/atlasqtl_hip.h) from the reference's argument lists; returns (problem, objects to keep
    alive while the library reads them)."""
    pr = AqVbProblem()
    pr.n, pr.p, pr.q, pr.q_total = n, p, q, q_total
    keep = []

    def vec(v, m, name):
        a = np.asarray(v, dtype=np.float64)
        a = np.full(m, float(a)) if a.ndim == 0 else np.ascontiguousarray(a)
        if a.shape != (m,):
            raise ValueError(f"{name} must have length {m}, got {a.shape}")
        keep.append(a)
        return as_dp(a)

    def mat(v, shape, name):
        a = np.asfortranarray(v, dtype=np.float64)
        if a.shape != shape:
            raise ValueError(f"{name} must have shape {shape}, got {a.shape}")
        keep.append(a)
        return as_dp(a)

    keep += [X, Y]
    if prep is None:
        pr.X, pr.Y = as_dp(X), as_dp(Y)
    else:
        pr.X = C.cast(prep.x_ptr, _lib.dp)
        if Y is prep.Y or (Y.shape == prep.Y.shape and np.array_equal(Y, prep.Y, equal_nan=True)):
            pr.Y, pr.xy_on_device = C.cast(prep.y_ptr, _lib.dp), 3     # the centred Y that is already on the GPU
        else:
            pr.Y, pr.xy_on_device = as_dp(Y), 1                        # another Y (e.g. one trait shard of it) from the host
    pr.A2_inv = float(list_hyper["A2_inv"]); pr.m0 = float(list_hyper["m0"])
    pr.nu = float(list_hyper["nu"]); pr.rho = float(list_hyper["rho"]); pr.t02 = float(list_hyper["t02"])
    pr.eta = vec(list_hyper["eta"], q, "eta"); pr.kappa = vec(list_hyper["kappa"], q, "kappa")
    pr.n0 = vec(list_hyper["n0"], q, "n0")
    g0, m0_ = list_init["gam_vb"], list_init["mu_beta_vb"]
    if g0 is None and m0_ is None:
        # the p x q initial values are drawn on the device (hyper_init.auto_set_init_(..., device_init=True))
        pr.init_generate = 1
        pr.init_seed = int(list_init["device_seed"]) & 0xFFFFFFFFFFFFFFFF
        pr.init_gam_mean = float(list_init["device_gam_mean"])
        pr.init_gam_sd = float(list_init["device_gam_sd"])
        pr.init_on_device = 0
    elif hasattr(g0, "data_ptr"):
        # torch CUDA tensors holding the p x q matrices column-major, i.e. a contiguous (q, p) tensor
        for tname, tt in (("gam_vb", g0), ("mu_beta_vb", m0_)):
            if not (tt.is_cuda and tt.is_contiguous() and tuple(tt.shape) == (q, p) and str(tt.dtype) == "torch.float64"):
                raise ValueError(f"{tname} on device must be a contiguous float64 CUDA tensor of shape (q, p) "
                                 "(= p x q column-major)")
        keep += [g0, m0_]
        pr.gam_vb = C.cast(g0.data_ptr(), _lib.dp)
        pr.mu_beta_vb = C.cast(m0_.data_ptr(), _lib.dp)
        pr.init_on_device = 1
    else:
        pr.gam_vb = mat(g0, (p, q), "gam_vb")
        pr.mu_beta_vb = mat(m0_, (p, q), "mu_beta_vb")
        pr.init_on_device = 0
    pr.sig02_inv_vb = float(list_init["sig02_inv_vb"])
    pr.sig2_beta_vb = vec(list_init["sig2_beta_vb"], q, "sig2_beta_vb")
    pr.sig2_theta_vb = vec(list_init["sig2_theta_vb"] if scheme != "global" else np.ones(p), p, "sig2_theta_vb")
    pr.tau_vb = vec(list_init["tau_vb"], q, "tau_vb")
    pr.theta_vb = vec(list_init["theta_vb"], p, "theta_vb")
    pr.zeta_vb = vec(list_init["zeta_vb"], q, "zeta_vb")
    pr.has_anneal = 0 if anneal is None else 1
    if anneal is not None:
        pr.anneal = (C.c_double * 3)(*[float(x) for x in anneal])
    pr.tol = float(tol); pr.maxit = int(maxit)
    pr.thinned_elbo_eval = 1 if thinned_elbo_eval else 0
    pr.debug = 1 if debug else 0
    pr.device = int(device); pr.world_size = int(world)
    pr.trait_offset = int(trait_offset)
    pr.scheme = {"global_local": 0, "global": 1}[scheme]
    pr.df = int(df)
    pr.ext_reduce_main = ext_main
    pr.ext_reduce_elbo = ext_elbo
    return pr, keep


class VbRun:
    """A device-resident VB state (aq_vb_handle).  Keeps the host arrays alive while the
    library copies them, drives the aq_vb_advance protocol and fetches results."""

    def __init__(self, Y, X, list_hyper, list_init, anneal, tol, maxit, thinned_elbo_eval=True, debug=True,
                 device=0, q_total=None, process_group=None, trait_offset=0, scheme="global_local", df=1):
        L = lib()
        prep = X if hasattr(X, "x_ptr") else None       # prepare.PreparedData: standardised X and centred Y already on the GPU
        Y = np.asfortranarray(Y, dtype=np.float64)
        if prep is None:
            X = np.asfortranarray(X, dtype=np.float64)
        n, p = X.shape
        if Y.shape[0] != n:
            raise ValueError("X and Y must have the same number of samples.")
        q = Y.shape[1]
        if prep is not None and prep.device != int(device):
            raise ValueError("PreparedData lives on another device")
        self.n, self.p, self.q = n, p, q
        self.q_total = int(q if q_total is None else q_total)
        self.pg = process_group
        self.world = 1
        self._red = self._ered = None
        ext_main = ext_elbo = None
        if process_group is not None:
            import torch
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            dev = torch.device("cuda", device)
            self._red = torch.zeros(int(L.aq_vb_reduce_len(p)), dtype=torch.float64, device=dev)
            self._ered = torch.zeros(8, dtype=torch.float64, device=dev)
            ext_main, ext_elbo = self._red.data_ptr(), self._ered.data_ptr()
        pr, keep = _build_problem(Y, X, prep, n, p, q, self.q_total, list_hyper, list_init, anneal, tol, maxit, thinned_elbo_eval,
                                  debug, device, self.world, trait_offset, scheme, df, ext_main, ext_elbo)
        h = C.c_void_p()
        check(L.aq_vb_create(C.byref(pr), C.byref(h)), "aq_vb_create")
        self.h = h
        self._keep = keep

    def close(self):
        if getattr(self, "h", None):
            lib().aq_vb_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _allreduce(self, which):
        if self.pg is None:
            return
        import torch.distributed as dist
        t = self._red if which == 0 else self._ered
        if dist.get_backend(self.pg) == "gloo":
            # CPU-staged exchange (tests on a one-GPU box); RCCL reduces the device buffer in place
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.pg)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def advance_until_done(self):
        """aq_vb_advance loop; all-reduces the payloads across the process group."""
        L = lib()
        while True:
            rc = L.aq_vb_advance(self.h)
            if rc < 0:
                check(-rc, "aq_vb_advance")
            if rc == _lib.AQ_VB_DONE:
                return
            self._allreduce(0 if rc == _lib.AQ_VB_NEED_ALLREDUCE_MAIN else 1)

    def run(self):
        if self.pg is None:
            check(lib().aq_vb_run(self.h), "aq_vb_run")
        else:
            self.advance_until_done()
        return self

    def run_sweeps(self, k):
        """Run at most k further sweeps (bench.py's timed region); works with and without a process group."""
        if self.pg is None:
            check(lib().aq_vb_run_sweeps(self.h, int(k)), "aq_vb_run_sweeps")
            return
        check(lib().aq_vb_set_sweep_budget(self.h, int(k)), "aq_vb_set_sweep_budget")
        self.advance_until_done()
        check(lib().aq_vb_set_sweep_budget(self.h, -1), "aq_vb_set_sweep_budget")

    def hotspot_sizes(self, thres=0.5, fdr_adjust=False):
        """Hotspot sizes from the gam_vb resident on the device (no p x q copy to the host): rowSums(gam_vb > thres), or
        rowSums(assign_bFDR(gam_vb) < thres) with fdr_adjust (summary.atlasqtl / plot.atlasqtl,
        R/summarise_output.R:98-105,177-182).  With a process group the counts cover the traits of ALL ranks; the FDR
        ranking is global (assign_bFDR sorts all p q PPIs, R/summarise_output.R:207-223)."""
        rs = np.zeros(self.p, dtype=np.int64)
        tot = C.c_int64(0)
        if self.pg is None or not fdr_adjust:
            check(lib().aq_vb_hotspot_sizes(self.h, float(thres), int(bool(fdr_adjust)), rs.ctypes.data_as(C.POINTER(C.c_int64)),
                                            C.byref(tot)), "aq_vb_hotspot_sizes")
            if self.pg is not None:
                rs = self._sum_over_ranks(rs)
            return rs, int(rs.sum()) if self.pg is not None else int(tot.value)
        return self._hotspot_sizes_fdr_sharded(float(thres))

    def _sum_over_ranks(self, a):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(a))
        if dist.get_backend(self.pg) != "gloo":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)
        return t.cpu().numpy()

    def _hotspot_sizes_fdr_sharded(self, thres):
        """{FDR < thres} is a prefix of the global decreasing PPI order (the running mean of 1 - PPI never decreases along
        it).  M(c) = mean of 1 - PPI over all entries >= c is the estimated FDR at the end of c's tie block; it grows as c
        falls, so a bisection over the bit patterns of c in [0, max PPI] finds the smallest c* with M(c*) < thres: every
        entry >= c* is in.  Of the next tie block down, the first entries (original order: lower ranks, then position) are
        in for as long as the running mean stays below thres.  Every step all-reduces four numbers."""
        import struct
        import torch
        import torch.distributed as dist
        L = lib()
        rank, world = dist.get_rank(self.pg), dist.get_world_size(self.pg)
        bits = lambda x: struct.unpack("<q", struct.pack("<d", x))[0]
        val = lambda b: struct.unpack("<d", struct.pack("<q", b))[0]
        check(L.aq_vb_bfdr_begin(self.h), "aq_vb_bfdr_begin")
        try:
            def query(c):              # local five numbers, global sums of the first four
                out = np.zeros(5)
                check(L.aq_vb_bfdr_query(self.h, float(c), as_dp(out)), "aq_vb_bfdr_query")
                return out, self._sum_over_ranks(out[:4].copy())

            def below(c):              # M(c) < thres, with at least one entry >= c
                _, g = query(c)
                return g[0] > 0 and g[1] / g[0] < thres

            vmax = self._max_over_ranks(query(2.0)[0][4])         # the largest PPI of all
            rs = np.zeros(self.p, dtype=np.int64)
            if not below(vmax):                                   # FDR of the very first entry >= thres: nothing qualifies
                return rs, 0
            if below(0.0):
                c_star = 0.0
            else:
                lo, hi = bits(0.0), bits(vmax)                    # below(lo) false, below(hi) true
                while hi - lo > 1:
                    mid = (lo + hi) // 2
                    lo, hi = (lo, mid) if below(val(mid)) else (mid, hi)
                c_star = val(hi)
            loc, g = query(c_star)
            upto = int(loc[0])                                    # this rank's entries >= c*
            n_star, s_star = float(g[0]), float(g[1])
            tie_val = self._max_over_ranks(loc[4])                # next PPI value down, over all ranks (-1: none)
            take, tie_first = 0, upto
            if tie_val >= 0.0:
                loc_t, g_t = query(tie_val)
                t_loc, t_glob = int(loc_t[0]) - int(loc_t[2]), int(g_t[0] - g_t[2])
                d = 1.0 - tie_val
                step_ok = lambda i: (s_star + i * d) / (n_star + i) < thres      # running mean after i entries of the block
                i = 0
                if d > thres:                                     # (else the whole block would have qualified)
                    i = int(max(0, min(t_glob, np.floor((thres * n_star - s_star) / (d - thres)))))
                    while i > 0 and not step_ok(i):
                        i -= 1
                    while i < t_glob and step_ok(i + 1):
                        i += 1
                counts = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
                tl = torch.tensor([t_loc], dtype=torch.int64)
                if dist.get_backend(self.pg) != "gloo":
                    counts, tl = [c.cuda() for c in counts], tl.cuda()
                dist.all_gather(counts, tl, group=self.pg)
                before = sum(int(c.item()) for c in counts[:rank])                # ties of lower ranks come first
                take = int(min(max(i - before, 0), t_loc))
            check(L.aq_vb_bfdr_rows(self.h, upto, tie_first, take, rs.ctypes.data_as(C.POINTER(C.c_int64))), "aq_vb_bfdr_rows")
            rs = self._sum_over_ranks(rs)
            return rs, int(rs.sum())
        finally:
            L.aq_vb_bfdr_end(self.h)

    def _max_over_ranks(self, v):
        import torch
        import torch.distributed as dist
        t = torch.tensor([float(v)], dtype=torch.float64)
        if dist.get_backend(self.pg) != "gloo":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        return float(t.item())

    def get_state(self):
        """The complete loop state between two sweeps as one uint8 array (aq_vb_get_state): unlike the reference's
        write-only checkpoint_ (R/utils.R:571-611), `set_state` on a handle created for the same problem continues
        bit-identically."""
        L = lib()
        if self.status()["it"] == 0:
            self.run_sweeps(0)          # initial residual and column sums, no sweep
        nbytes = L.aq_vb_state_bytes(self.h)
        buf = np.empty(int(nbytes), dtype=np.uint8)
        check(L.aq_vb_get_state(self.h, buf.ctypes.data_as(C.c_void_p), buf.size), "aq_vb_get_state")
        return buf

    def set_state(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        check(lib().aq_vb_set_state(self.h, buf.ctypes.data_as(C.c_void_p), buf.size), "aq_vb_set_state")
        return self

    def status(self):
        st = AqVbStatus()
        check(lib().aq_vb_get_status(self.h, C.byref(st)), "aq_vb_get_status")
        out = {f[0]: getattr(st, f[0]) for f in AqVbStatus._fields_}
        buf = C.create_string_buffer(1024)
        lib().aq_vb_get_overrides(self.h, buf, 1024)
        out["overrides"] = buf.value.decode()          # AQ_* environment hooks in effect ("" = the library's own launch plan)
        return out

    def elbo_trace(self):
        cap = 4096
        its = np.zeros(cap, dtype=np.int32)
        lbs = np.zeros(cap, dtype=np.float64)
        n = lib().aq_vb_get_elbo_trace(self.h, as_ip(its), as_dp(lbs), cap)
        return its[:n].copy(), lbs[:n].copy()

    def residual(self):
        """mis_pat .* (Y - X beta_vb) as the sweep kernel carries it (n x q), see aq_vb_get_residual."""
        R = np.zeros((self.n, self.q), order="F")
        check(lib().aq_vb_get_residual(self.h, as_dp(R)), "aq_vb_get_residual")
        return R

    def result(self, full_output=False):
        p, q = self.p, self.q
        beta = np.zeros((p, q), order="F"); gam = np.zeros((p, q), order="F")
        theta = np.zeros(p); zeta = np.zeros(q)
        mu = np.zeros((p, q), order="F") if full_output else None
        lam = np.zeros(p) if full_output else None
        s2t = np.zeros(p) if full_output else None
        tau = np.zeros(q) if full_output else None
        s2b = np.zeros(q) if full_output else None
        nul = C.cast(None, _lib.dp)
        check(lib().aq_vb_get_result(self.h, as_dp(beta), as_dp(gam), as_dp(mu) if full_output else nul, as_dp(theta),
                                     as_dp(zeta), as_dp(lam) if full_output else nul,
                                     as_dp(s2t) if full_output else nul, as_dp(tau) if full_output else nul,
                                     as_dp(s2b) if full_output else nul), "aq_vb_get_result")
        out = dict(beta_vb=beta, gam_vb=gam, theta_vb=theta, zeta_vb=zeta)
        if full_output:
            out.update(mu_beta_vb=mu, lam2_inv_vb=lam, sig2_theta_vb=s2t, tau_vb=tau, sig2_beta_vb=s2b)
        return out


def vb_partition(q, n_parts):
    """Trait ranges [(k0, k1)] of aq_vb_partition: whole 16-trait tiles per part."""
    out = []
    for r in range(n_parts):
        k0, k1 = C.c_int32(), C.c_int32()
        check(lib().aq_vb_partition(int(q), int(n_parts), r, C.byref(k0), C.byref(k1)), "aq_vb_partition")
        out.append((k0.value, k1.value))
    return out


def run_multi(Y, X, list_hyper, list_init, anneal, tol, maxit, n_gpus, devices=None, transport=0, thinned_elbo_eval=True,
              debug=True, scheme="global_local", df=1, full_output=True):
    """aq_vb_run_multi: the whole run on n_gpus GPUs of this node from this one process (host threads + RCCL inside the
    library; transport=1 stages the two small all-reduces through host memory and lets devices repeat).  Same result fields
    as atlasqtl_global_local_core_."""
    Y = np.asfortranarray(Y, dtype=np.float64)
    X = np.asfortranarray(X, dtype=np.float64)
    n, p = X.shape
    q = Y.shape[1]
    pr, keep = _build_problem(Y, X, None, n, p, q, q, list_hyper, list_init, anneal, tol, maxit, thinned_elbo_eval, debug,
                              0, 1, 0, scheme, df)
    out = _lib.AqVbMultiOut()
    res = dict(beta_vb=np.zeros((p, q), order="F"), gam_vb=np.zeros((p, q), order="F"), theta_vb=np.zeros(p), zeta_vb=np.zeros(q))
    if full_output:
        res.update(mu_beta_vb=np.zeros((p, q), order="F"), lam2_inv_vb=np.zeros(p), sig2_theta_vb=np.zeros(p), tau_vb=np.zeros(q),
                   sig2_beta_vb=np.zeros(q))
    for k, v in res.items():
        setattr(out, k, as_dp(v))
    cap = 4096
    its, lbs = np.zeros(cap, dtype=np.int32), np.zeros(cap)
    out.elbo_it, out.elbo_lb, out.elbo_cap = as_ip(its), as_dp(lbs), cap
    dv = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
    if dv is not None and dv.size != n_gpus:
        raise ValueError("devices must list n_gpus ordinals")
    check(lib().aq_vb_run_multi(C.byref(pr), int(n_gpus), as_ip(dv) if dv is not None else None, int(transport), C.byref(out)),
          "aq_vb_run_multi")
    ne = min(out.n_elbo, cap)
    res.update(n=n, p=p, q=q, anneal=anneal, converged=bool(out.converged), it=int(out.it), maxit=maxit, tol=tol,
               lb_opt=out.lb_opt, diff_lb=out.diff_lb, sig02_inv_vb=out.sig02_inv_vb, sig2_inv_vb=out.sig2_inv_vb,
               elbo_trace=(its[:ne].copy(), lbs[:ne].copy()), seconds=out.seconds, core_ms=out.core_ms)
    del keep
    return res


def atlasqtl_global_core_(Y, X, shr_fac_inv, anneal, df, tol, maxit, verbose, list_hyper, list_init, checkpoint_path=None,
                          full_output=False, thinned_elbo_eval=True, debug=False, batch="y", **kw):
    """R/atlasqtl_global_core.R:8-366 on the GPU: the same sweep with ONE global scale for the hotspot propensities (no
    horseshoe local scales; df is not used by that core).  list_init needs no sig2_theta_vb."""
    return atlasqtl_global_local_core_(Y, X, shr_fac_inv, anneal, 1, tol, maxit, verbose, list_hyper, list_init,
                                       checkpoint_path=checkpoint_path, full_output=full_output,
                                       thinned_elbo_eval=thinned_elbo_eval, debug=debug, batch=batch, scheme="global", **kw)


def assign_bFDR(mat_ppi, device=0):
    """assign_bFDR of the reference (R/summarise_output.R:207-223) on the GPU: sort of all p q PPIs (hipCUB radix sort,
    ties in original order), running mean of 1 - PPI, scattered back."""
    m = np.asarray(mat_ppi, dtype=np.float64)
    vec = np.ascontiguousarray(m.reshape(-1, order="F"))
    out = np.empty_like(vec)
    check(lib().aq_assign_bfdr(as_dp(vec), as_dp(out), vec.size, int(device)), "aq_assign_bfdr")
    return out.reshape(m.shape, order="F")


def hotspot_sizes(gam_vb, thres=0.5, fdr_adjust=False, device=0):
    """rs_thres and nb_pairwise of summary.atlasqtl / plot.atlasqtl (R/summarise_output.R:98-105,177-182)."""
    m = np.asfortranarray(gam_vb, dtype=np.float64)
    p, q = m.shape
    rs = np.zeros(p, dtype=np.int64)
    tot = C.c_int64(0)
    check(lib().aq_hotspot_sizes(as_dp(m), p, q, float(thres), int(bool(fdr_adjust)),
                                 rs.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(tot), int(device)), "aq_hotspot_sizes")
    return rs, int(tot.value)


def _run_with_checkpoints(run, checkpoint_path, rate, maxit):
    """checkpoint_ / checkpoint_clean_up_ (R/utils.R:571-627, R/atlasqtl_global_local_core.R:379,388): every `rate`
    iterations write the reference's temporary output list (tmp_output_it_<it>.npz: beta_vb, gam_vb, theta_vb, zeta_vb,
    converged, it, lb_new, diff_lb, lam2_inv_vb, sig02_inv_vb) and keep only the last two; remove them all at the end.
    Beside each, hip_state_it_<it>.npy holds the complete device state for `resume_from` (the reference cannot resume);
    they are removed as well once the run has converged (a run that is killed, or that stops at maxit, leaves its last
    two behind: that is their use)."""
    import glob
    import os
    if not os.path.isdir(checkpoint_path):
        raise ValueError("The directory specified in checkpoint_path does not exist. ")       # R/prepare_atlasqtl.R:21-22
    rank = 0 if run.pg is None else __import__("torch").distributed.get_rank(run.pg)
    tag = "" if run.pg is None else f"_rank{rank}"
    while True:
        st = run.status()
        if st["converged"] or st["it"] >= maxit:
            break
        run.run_sweeps(rate - st["it"] % rate)
        st = run.status()
        it = st["it"]
        if it % rate == 0 and not st["converged"]:
            res = run.result(full_output=True)
            np.savez(os.path.join(checkpoint_path, f"tmp_output_it_{it}{tag}.npz"), beta_vb=res["beta_vb"],
                     gam_vb=res["gam_vb"], theta_vb=res["theta_vb"], zeta_vb=res["zeta_vb"], converged=bool(st["converged"]),
                     it=it, lb_new=st["lb_opt"], diff_lb=st["diff_lb"], lam2_inv_vb=res["lam2_inv_vb"],
                     sig02_inv_vb=st["sig02_inv_vb"])
            np.save(os.path.join(checkpoint_path, f"hip_state_it_{it}{tag}.npy"), run.get_state())
            for stem in (f"tmp_output_it_{it - 2 * rate}{tag}.npz", f"hip_state_it_{it - 2 * rate}{tag}.npy"):
                old = os.path.join(checkpoint_path, stem)          # keep only the last two for comparison
                if os.path.exists(old):
                    os.remove(old)
    # checkpoint_clean_up_ (R/utils.R:614-627) leaves the directory clean; so do the resumable state files once the run
    # has converged (after maxit without convergence the last two stay: resume_from with a larger maxit continues them)
    pats = [f"tmp_output_it_*{tag}.npz"] + ([f"hip_state_it_*{tag}.npy"] if run.status()["converged"] else [])
    for pat in pats:
        for f in glob.glob(os.path.join(checkpoint_path, pat)):
            os.remove(f)


def atlasqtl_global_local_core_(Y, X, shr_fac_inv, anneal, df, tol, maxit, verbose, list_hyper, list_init,
                                checkpoint_path=None, trace_path=None, full_output=False, thinned_elbo_eval=True,
                                debug=False, batch="y", device=0, process_group=None, resume_from=None,
                                checkpoint_rate=100, trait_offset=None, scheme="global_local"):
    """R/atlasqtl_global_local_core.R:8-433 on the GPU.  Returns the reference's list
    (:426-428): beta_vb, gam_vb, theta_vb, zeta_vb, n, p, q, anneal, converged, it, maxit,
    tol, lb_opt, diff_lb (+ the variational parameters with full_output).

    shr_fac_inv is the total number of traits (R/atlasqtl.R:218); with a process group each
    rank passes its own trait columns and shr_fac_inv = q of the whole problem.  trait_offset = global index of
    this rank's first trait: required with a process group when the p x q initial values are drawn on the device
    (the Philox counters are (SNP, global trait), so a sharded run reproduces the single-GPU draws)."""
    if df not in (1, 3, 5, 7):
        raise NotImplementedError("df must be 1, 3, 5 or 7 (compute_integral_hs_, R/utils.R:425-568, is unstable from df = 9 on)")
    if batch != "y":
        raise ValueError("Batch scheme not defined. Exit.")            # :231
    if trace_path is not None:
        raise NotImplementedError("trace_path (trace plots) is outside the accelerated path")
    device_init = list_init.get("gam_vb") is None and list_init.get("mu_beta_vb") is None
    if process_group is not None and device_init and trait_offset is None:
        raise ValueError("trait_offset is required with process_group when the initial values are drawn on the device: "
                         "without it every trait shard would start from the draws of traits 0..q_local-1")
    run = VbRun(Y, X, list_hyper, list_init, anneal, tol, maxit, thinned_elbo_eval, debug, device=device,
                q_total=int(shr_fac_inv), process_group=process_group, trait_offset=int(trait_offset or 0), scheme=scheme,
                df=df)
    try:
        if resume_from is not None:
            run.set_state(np.load(resume_from))
        if checkpoint_path is None:
            run.run()
        else:
            _run_with_checkpoints(run, checkpoint_path, checkpoint_rate, maxit)
        st = run.status()
        if verbose != 0:
            if st["converged"]:
                print(f"Convergence obtained after {st['it']} iterations. \nOptimal marginal log-likelihood "
                      f"variational lower bound (ELBO) = {st['lb_opt']}. \n")
            else:
                import warnings
                warnings.warn("Maximal number of iterations reached before convergence. Exit.")   # :397
        res = run.result(full_output=full_output)
        its, lbs = run.elbo_trace()
        res.update(n=run.n, p=run.p, q=run.q, anneal=anneal, converged=bool(st["converged"]), it=int(st["it"]),
                   maxit=maxit, tol=tol, lb_opt=st["lb_opt"], diff_lb=st["diff_lb"])
        if full_output:
            res.update(sig02_inv_vb=st["sig02_inv_vb"], sig2_inv_vb=st["sig2_inv_vb"], elbo_trace=(its, lbs),
                       core_ms=st["core_ms"], core_launches=st["core_launches"], lentz_iters=st["lentz_iters"],
                       core_kernel=st["core_kernel"])
        return res
    finally:
        run.close()

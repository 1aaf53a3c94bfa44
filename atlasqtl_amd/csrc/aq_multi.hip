// aq_multi.hip -- aq_vb_run_multi: the whole VB run on several GPUs of one node from ONE host process, for hosts that have
// no torch.distributed (the reference's host language is R: R/atlasqtl.R:274-278 calls the core once, single-threaded).
//
// The trait axis is cut into whole 16-trait tiles (aq_vb_partition), one shard per GPU; one host thread per GPU creates its
// handle (aq_vb_create with world_size = n_gpus, trait_offset = first trait) and drives the aq_vb_advance protocol of
// include/atlasqtl_hip.h.  Whenever the handles ask for it, the p + 8 doubles (main payload) or 8 doubles (ELBO payload) are
// SUM-all-reduced in place across the GPUs:
//   transport 0  RCCL (ncclAllReduce over xGMI), one communicator per GPU from ncclCommInitAll, enqueued on the legacy default
//                stream the library works on (no host synchronisation).  librccl.so is loaded at run time (dlopen), so the
//                library itself does not depend on it;
//   transport 1  staged through host memory in fixed rank order (no RCCL needed; also lets a one-GPU box exercise n_gpus > 1
//                with every shard on the same device: tests/test_gpu_multi.py).
// Every rank sees the same reduced bits, hence the same convergence decisions: the threads stay in lock-step by construction;
// a barrier in front of each reduction also carries an error flag so that a failing rank takes the others out with it.
// This file uses the public C ABI only.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/atlasqtl_hip.h"

int aq_fail_ext(int code, const std::string &msg);   // atlasqtl_hip.hip: sets this thread's aq_last_error()

// ---- trait partition: whole 16-trait tiles, as even as tiles allow -------------------------------------------------------
extern "C" int aq_vb_partition(int32_t q, int32_t n_parts, int32_t part, int32_t *k0, int32_t *k1) {
  if (q < 1 || n_parts < 1 || part < 0 || part >= n_parts || !k0 || !k1) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_partition: bad argument");
  const long long ntile = ((long long)q + 15) / 16;
  if (n_parts > ntile) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_partition: more parts than 16-trait tiles");
  const long long t0 = ntile * part / n_parts, t1 = ntile * (part + 1) / n_parts;
  *k0 = (int32_t)(16 * t0);
  *k1 = (int32_t)std::min<long long>(q, 16 * t1);
  return AQ_OK;
}

namespace {

// ---- RCCL, loaded at run time --------------------------------------------------------------------------------------------
typedef void *aq_nccl_comm;
struct AqRccl {
  void *so = nullptr;
  int (*CommInitAll)(aq_nccl_comm *, int, const int *) = nullptr;
  int (*CommDestroy)(aq_nccl_comm) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, aq_nccl_comm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  std::string err;
  bool load() {
    if (so) return true;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (so) break;
    }
    if (!so) { err = std::string("librccl.so not loadable: ") + (dlerror() ? dlerror() : "?"); return false; }
    CommInitAll = (decltype(CommInitAll))dlsym(so, "ncclCommInitAll");
    CommDestroy = (decltype(CommDestroy))dlsym(so, "ncclCommDestroy");
    AllReduce = (decltype(AllReduce))dlsym(so, "ncclAllReduce");
    GetErrorString = (decltype(GetErrorString))dlsym(so, "ncclGetErrorString");
    if (!CommInitAll || !CommDestroy || !AllReduce || !GetErrorString) { err = "librccl.so lacks the expected entry points"; return false; }
    return true;
  }
};
constexpr int AQ_NCCL_DOUBLE = 8, AQ_NCCL_SUM = 0;   // ncclFloat64, ncclSum (rccl.h)

// ---- a reusable barrier that also spreads a failure ----------------------------------------------------------------------
struct AqBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n, waiting = 0;
  long long generation = 0;
  bool failed = false;          // sticky: set by any arrival
  bool verdict[2] = {false, false};   // `failed` as it stood when generation g was released, by parity of g (a thread is never
                                      // more than one generation ahead of another, so two slots do)
  explicit AqBarrier(int n_) : n(n_) {}
  // Every participant of one generation gets the SAME answer: true when any participant of this or an earlier generation failed.
  bool arrive(bool my_fail) {
    std::unique_lock<std::mutex> lk(m);
    failed = failed || my_fail;
    const long long gen = generation;
    if (++waiting == n) {
      waiting = 0;
      verdict[gen & 1] = failed;
      generation++;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
    return verdict[gen & 1];
  }
};

struct AqShared {
  int n;
  int transport;
  AqBarrier bar;
  std::vector<aq_nccl_comm> comm;
  AqRccl *rccl = nullptr;
  std::vector<std::vector<double>> host_part;   // transport 1: this step's payloads, one per rank
  std::vector<double> host_sum;
  std::mutex err_m;
  int err_code = AQ_OK;
  std::string err_msg;
  AqShared(int n_, int tr) : n(n_), transport(tr), bar(n_), comm(n_, nullptr), host_part(n_) {}
  void fail(int code, const std::string &msg) {
    std::lock_guard<std::mutex> g(err_m);
    if (err_code == AQ_OK) { err_code = code; err_msg = msg; }
  }
};

// SUM all-reduce of `len` doubles at buf (device memory of rank r's GPU), in place.  Returns false when the run must stop.
bool aq_allreduce(AqShared &sh, int r, int device, double *buf, size_t len) {
  if (sh.transport == 0) {
    if (sh.bar.arrive(false)) return false;                      // everybody is about to enqueue the same collective
    (void)hipSetDevice(device);
    const int rc = sh.rccl->AllReduce(buf, buf, len, AQ_NCCL_DOUBLE, AQ_NCCL_SUM, sh.comm[r], (hipStream_t)0);
    if (rc != 0) sh.fail(AQ_ERR_DEVICE, std::string("ncclAllReduce: ") + sh.rccl->GetErrorString(rc));
    return !sh.bar.arrive(rc != 0);                              // (host threads only: the GPUs do not synchronise here)
  }
  // host-staged: every rank copies its payload out (a blocking copy on ITS device's legacy default stream: ordered behind its
  // producer kernels), rank 0 adds the parts in rank order, every rank copies the sum back
  (void)hipSetDevice(device);
  sh.host_part[r].resize(len);
  const bool bad0 = hipMemcpy(sh.host_part[r].data(), buf, len * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess;
  if (bad0) sh.fail(AQ_ERR_DEVICE, "aq_vb_run_multi: staging copy of a reduce payload failed");
  if (sh.bar.arrive(bad0)) return false;
  if (r == 0) {
    sh.host_sum.assign(len, 0.0);
    for (int k = 0; k < sh.n; k++)
      for (size_t i = 0; i < len; i++) sh.host_sum[i] += sh.host_part[k][i];
  }
  if (sh.bar.arrive(false)) return false;
  const bool bad2 = hipMemcpy(buf, sh.host_sum.data(), len * sizeof(double), hipMemcpyHostToDevice) != hipSuccess;
  if (bad2) sh.fail(AQ_ERR_DEVICE, "aq_vb_run_multi: copy of the reduced payload failed");
  return !sh.bar.arrive(bad2);
}

struct AqRankJob {
  aq_vb_problem pr;
  int k0 = 0, k1 = 0, device = 0;
};

void aq_rank_main(AqShared &sh, int r, const AqRankJob &job, const aq_vb_multi_out *out, int32_t p, aq_vb_status *st_out,
                  std::vector<int32_t> *tr_it, std::vector<double> *tr_lb) {
  aq_vb_handle h = nullptr;
  bool ok = aq_vb_create(&job.pr, &h) == AQ_OK;
  if (!ok) sh.fail(AQ_ERR_ARG, std::string("rank ") + std::to_string(r) + " aq_vb_create: " + aq_last_error());
  if (sh.bar.arrive(!ok)) { if (h) aq_vb_destroy(h); return; }
  const size_t len_main = (size_t)aq_vb_reduce_len(p);
  for (;;) {
    const int rc = aq_vb_advance(h);
    if (rc < 0) {
      sh.fail(-rc, std::string("rank ") + std::to_string(r) + ": " + aq_last_error());
      sh.bar.arrive(true);                                      // meets the others at their next barrier
      break;
    }
    if (rc == AQ_VB_DONE) {
      // all ranks reach DONE in the same step (same reduced ELBO, same counters); one more rendezvous catches a rank that failed
      if (sh.bar.arrive(false)) break;
      aq_vb_status st;
      bool good = aq_vb_get_status(h, &st) == AQ_OK;
      if (good && out) {
        const size_t off = (size_t)p * job.k0;
        good = aq_vb_get_result(h, out->beta_vb ? out->beta_vb + off : nullptr, out->gam_vb ? out->gam_vb + off : nullptr,
                                out->mu_beta_vb ? out->mu_beta_vb + off : nullptr, r == 0 ? out->theta_vb : nullptr,
                                out->zeta_vb ? out->zeta_vb + job.k0 : nullptr, r == 0 ? out->lam2_inv_vb : nullptr,
                                r == 0 ? out->sig2_theta_vb : nullptr, out->tau_vb ? out->tau_vb + job.k0 : nullptr,
                                out->sig2_beta_vb ? out->sig2_beta_vb + job.k0 : nullptr) == AQ_OK;
      }
      if (!good) sh.fail(AQ_ERR_DEVICE, std::string("rank ") + std::to_string(r) + ": " + aq_last_error());
      if (good) {
        st_out[r] = st;
        if (r == 0) {
          tr_it->resize(st.n_elbo);
          tr_lb->resize(st.n_elbo);
          aq_vb_get_elbo_trace(h, tr_it->data(), tr_lb->data(), st.n_elbo);
        }
      }
      sh.bar.arrive(!good);
      break;
    }
    double *buf = aq_vb_reduce_ptr(h, rc == AQ_VB_NEED_ALLREDUCE_MAIN ? 0 : 1);
    if (!aq_allreduce(sh, r, job.device, buf, rc == AQ_VB_NEED_ALLREDUCE_MAIN ? len_main : 8)) break;
  }
  aq_vb_destroy(h);
}

}   // namespace

extern "C" int aq_vb_run_multi(const aq_vb_problem *prob, int32_t n_gpus, const int32_t *devices, int32_t transport,
                               aq_vb_multi_out *out) {
  if (!prob || !out) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: NULL argument");
  if (n_gpus < 1 || n_gpus > 64) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: n_gpus must be in 1 .. 64");
  if (transport != 0 && transport != 1) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: transport must be 0 (RCCL) or 1 (host-staged)");
  if (prob->q != prob->q_total) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: give the whole problem (q == q_total); the library shards it");
  if (prob->world_size > 1 || prob->ext_reduce_main || prob->ext_reduce_elbo)
    return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: world_size / ext_reduce_* belong to the aq_vb_advance protocol");
  if (n_gpus > 1 && (prob->init_on_device || prob->xy_on_device))
    return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: device pointers live on one GPU; pass host pointers (or init_generate) for n_gpus > 1");
  if (!prob->Y || !prob->eta || !prob->kappa || !prob->n0 || !prob->sig2_beta_vb || !prob->tau_vb || !prob->zeta_vb ||
      (!prob->init_generate && (!prob->gam_vb || !prob->mu_beta_vb)))
    return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: NULL data pointer");
  const int ndev = aq_device_count();
  if (ndev < 1) return aq_fail_ext(AQ_ERR_DEVICE, "no HIP device visible: libatlasqtl_hip has no CPU fallback (MI355X / gfx950 required)");
  std::vector<int> dev(n_gpus);
  for (int r = 0; r < n_gpus; r++) {
    dev[r] = devices ? devices[r] : r;
    if (dev[r] < 0 || dev[r] >= ndev) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: device ordinal out of range");
    if (transport == 0)
      for (int k = 0; k < r; k++)
        if (dev[k] == dev[r]) return aq_fail_ext(AQ_ERR_ARG, "aq_vb_run_multi: RCCL needs distinct devices (transport 1 allows repeats)");
  }
  std::vector<AqRankJob> jobs(n_gpus);
  for (int r = 0; r < n_gpus; r++) {
    int32_t k0, k1;
    int rc = aq_vb_partition(prob->q, n_gpus, r, &k0, &k1);
    if (rc != AQ_OK) return rc;
    AqRankJob &j = jobs[r];
    j.pr = *prob;
    j.k0 = k0; j.k1 = k1; j.device = dev[r];
    j.pr.q = k1 - k0;
    j.pr.q_total = prob->q;
    const bool y_dev = (prob->xy_on_device & 2) != 0;   // n_gpus == 1 only: offset 0
    (void)y_dev;
    j.pr.Y = prob->Y + (size_t)prob->n * k0;
    j.pr.eta = prob->eta + k0; j.pr.kappa = prob->kappa + k0; j.pr.n0 = prob->n0 + k0;
    if (prob->gam_vb) j.pr.gam_vb = prob->gam_vb + (size_t)prob->p * k0;
    if (prob->mu_beta_vb) j.pr.mu_beta_vb = prob->mu_beta_vb + (size_t)prob->p * k0;
    j.pr.sig2_beta_vb = prob->sig2_beta_vb + k0; j.pr.tau_vb = prob->tau_vb + k0; j.pr.zeta_vb = prob->zeta_vb + k0;
    j.pr.device = dev[r];
    j.pr.world_size = n_gpus;
    j.pr.trait_offset = prob->trait_offset + k0;
  }
  static AqRccl rccl;                 // loaded once per process
  static std::mutex rccl_m;
  AqShared sh(n_gpus, transport);
  if (transport == 0) {
    std::lock_guard<std::mutex> g(rccl_m);
    if (!rccl.load()) return aq_fail_ext(AQ_ERR_DEVICE, "aq_vb_run_multi: " + rccl.err + " (transport 1 needs no RCCL)");
    sh.rccl = &rccl;
    const int rc = rccl.CommInitAll(sh.comm.data(), n_gpus, dev.data());
    if (rc != 0) return aq_fail_ext(AQ_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl.GetErrorString(rc));
  }
  std::vector<aq_vb_status> st(n_gpus);
  std::vector<int32_t> tr_it;
  std::vector<double> tr_lb;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> th;
  for (int r = 0; r < n_gpus; r++)
    th.emplace_back(aq_rank_main, std::ref(sh), r, std::cref(jobs[r]), (const aq_vb_multi_out *)out, prob->p, st.data(), &tr_it, &tr_lb);
  for (auto &t : th) t.join();
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (transport == 0)
    for (int r = 0; r < n_gpus; r++)
      if (sh.comm[r]) rccl.CommDestroy(sh.comm[r]);
  if (sh.err_code != AQ_OK) return aq_fail_ext(sh.err_code, sh.err_msg);
  out->it = st[0].it; out->converged = st[0].converged; out->lb_opt = st[0].lb_opt; out->diff_lb = st[0].diff_lb;
  out->sig02_inv_vb = st[0].sig02_inv_vb; out->sig2_inv_vb = st[0].sig2_inv_vb;
  out->seconds = secs;
  out->core_ms = 0.0;
  for (int r = 0; r < n_gpus; r++) {
    if (st[r].it != st[0].it || st[r].converged != st[0].converged)
      return aq_fail_ext(AQ_ERR_DEVICE, "aq_vb_run_multi: the ranks left lock-step (different iteration counts)");
    if (st[r].core_ms > out->core_ms) out->core_ms = st[r].core_ms;
  }
  out->n_elbo = (int32_t)tr_it.size();
  for (int i = 0; i < out->n_elbo && i < out->elbo_cap; i++) {
    if (out->elbo_it) out->elbo_it[i] = tr_it[i];
    if (out->elbo_lb) out->elbo_lb[i] = tr_lb[i];
  }
  return AQ_OK;
}

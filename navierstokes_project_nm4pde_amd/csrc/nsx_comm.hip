// nsx_comm.hip — communication of the distributed solve: one process per GPU.
// Replaces the MPI traffic hidden in Epetra: MPI_Allreduce behind every dot/norm, Epetra_Import
// (ghost refresh) behind every vmult and behind `solution = solution_owned` (reference NavierStokes3D.cpp:638).
//
// Two interchangeable backends behind the same pack/unpack code:
//   * RCCL over xGMI (nsx_comm_init): ncclAllReduce of the scalar slots on the compute stream; halo = one grouped
//     ncclSend/ncclRecv per neighbour straight into the ghost region of the vector (ghosts are stored per owner in
//     rank order, so no unpack kernel is needed).  xGMI is point-to-point: every neighbour pair has its own link.
//   * host callbacks (nsx_comm_init_callbacks): the same exchange through host buffers and functions supplied by the
//     caller (MPI in the reference, gloo in the tests) — lets `mpirun` users keep their communicator and lets the
//     N > 1 path be tested on a single GPU.
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>

#include "nsx_grid.hpp"

namespace nsx {

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  nsx_allreduce_fn allreduce = nullptr;
  nsx_exchange_fn exchange = nullptr;
  void *ctx = nullptr;
  std::vector<double> stage;  // host staging of the callback backend
};

#define NCCL_CHECK(expr)                                                                              \
  do {                                                                                                \
    ncclResult_t r_ = (expr);                                                                         \
    if (r_ != ncclSuccess) NSX_THROW(NSX_ERR_COMM, "%s failed: %s", #expr, ncclGetErrorString(r_));   \
  } while (0)

// development (one-GPU boxes, NSX_EXT_SELF_P2P): a self-addressed send / receive pair = a REAL launch of RCCL's generic device kernel on
// `st` (a 1-rank all-reduce launches none).  1: in front of the collective inside the Gram-Schmidt sweep (communication stream);
// 2: in front of every all-reduce of the compute stream as well, so that RCCL kernels of ONE communicator alternate between two streams.
static int self_p2p_mode(const nsx_handle *h) { return h->self_p2p; }  // read when the communicator is created (nsx_comm_init)
static void self_p2p(nsx_handle *h, hipStream_t st) {
  Comm *c = h->comm;
  const size_t count = 64;
  if (!h->ext_self.p) {
    h->ext_self.alloc(4 * count);
    h->ext_self.zero(st);
  }
  double *buf = h->ext_self.p + (st == h->stream ? 2 * count : 0);  // one pair of operands per stream
  NCCL_CHECK(ncclGroupStart());
  NCCL_CHECK(ncclSend(buf, count, ncclDouble, c->rank, c->comm, st));
  NCCL_CHECK(ncclRecv(buf + count, count, ncclDouble, c->rank, c->comm, st));
  NCCL_CHECK(ncclGroupEnd());
}

void comm_allreduce_scalars(nsx_handle *h, int slot0, int count) {
  Comm *c = h->comm;
  if (!c || (c->world == 1 && !c->comm)) return;  // a 1-rank RCCL communicator still runs the collective (API self-test)
  h->n_allreduce++;
  if (c->comm) {
    if (self_p2p_mode(h) >= 2) self_p2p(h, h->stream);
    NCCL_CHECK(ncclAllReduce(h->scal.p + slot0, h->scal.p + slot0, count, ncclDouble, ncclSum, c->comm, h->stream));
  } else {
    HIP_CHECK(hipMemcpyAsync(h->scal_host + slot0, h->scal.p + slot0, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (c->allreduce(c->ctx, h->scal_host + slot0, count)) NSX_THROW(NSX_ERR_COMM, "allreduce callback failed");
    HIP_CHECK(hipMemcpyAsync(h->scal.p + slot0, h->scal_host + slot0, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  }
}

void comm_allreduce_partials(nsx_handle *h, double *partials, int count) {
  Comm *c = h->comm;
  if (!c || (c->world == 1 && !c->comm)) return;
  h->n_allreduce++;
  if (c->comm) {
    if (self_p2p_mode(h) >= 2) self_p2p(h, h->stream);
    NCCL_CHECK(ncclAllReduce(partials, partials, count, ncclDouble, ncclSum, c->comm, h->stream));
  } else {
    c->stage.resize((size_t)count);
    HIP_CHECK(hipMemcpyAsync(c->stage.data(), partials, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
    if (c->allreduce(c->ctx, c->stage.data(), count)) NSX_THROW(NSX_ERR_COMM, "allreduce callback failed");
    HIP_CHECK(hipMemcpyAsync(partials, c->stage.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  }
}

// ---- a collective INSIDE a persistent kernel's grid-wide exchange (nsx_blas.hip: k_mgs_one<.., true>) ---------------------------
// The persistent grid runs on the compute stream and cannot call RCCL.  Its reducers leave the rank-local sums in `vals` and count
// themselves in at `arrive`; on the COMMUNICATION stream, enqueued right behind the grid's launch: a one-thread kernel that spins
// until the count is complete, the all-reduce, and a one-thread kernel that stores the sweep's sequence number in `flag`, which
// every workgroup of the grid is waiting for.  All waits are bounded; a count that never completes raises vals[fail_word], which
// the all-reduce sums, so every rank learns of it.
__global__ void k_ext_wait(const unsigned int *arrive, unsigned int expected, double *vals, int fail_word) {
  unsigned long long t0 = 0;
  for (unsigned int spin = 1;; ++spin) {
    if ((int)(__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - expected) >= 0) return;  // difference: the counter may wrap
    __builtin_amdgcn_s_sleep(2);
    if ((spin & 255u) == 0) {
      const unsigned long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      else if (now - t0 > 2 * GX_TIMEOUT_TICKS) {  // the grid behind it is not complete: tell every rank
        __hip_atomic_store(vals + fail_word, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
    }
  }
}
__global__ void k_ext_release(unsigned long long *flag, unsigned long long seq) {
  __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

bool comm_on_stream(const nsx_handle *h) { return h->comm && h->comm->comm; }

// A choice between two code paths that issue DIFFERENT collectives (or the same collective over differently laid-out buffers)
// must be the same on every rank, whatever the rank's own sizes say: true only if `mine` is true everywhere.  One collective;
// callers cache the answer (it depends on set-up products, not on the state).  Every rank must get here at the same point of its
// collective sequence -- the callers sit in code all ranks run in lockstep.
bool comm_agree_all(nsx_handle *h, bool mine) {
  Comm *c = h->comm;
  if (!c || (c->world == 1 && !c->comm)) return mine;
  h->scal_host[N_SLOTS - 1] = mine ? 0.0 : 1.0;  // the sum of the objections
  HIP_CHECK(hipMemcpyAsync(h->scal.p + N_SLOTS - 1, h->scal_host + N_SLOTS - 1, sizeof(double), hipMemcpyHostToDevice, h->stream));
  h->slot_nb[N_SLOTS - 1] = 0;
  comm_allreduce_scalars(h, N_SLOTS - 1, 1);
  HIP_CHECK(hipMemcpyAsync(h->scal_host + N_SLOTS - 1, h->scal.p + N_SLOTS - 1, sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  return h->scal_host[N_SLOTS - 1] == 0.0;
}

// A communication stream of the device's highest priority would take its hardware queue from another pool than the compute
// stream's (the runtime shares hardware queues between streams of one priority once there are more streams than queues, and two
// streams on one queue run in order -- the collective inside a persistent grid would wait for the grid that waits for it) ...
static void ensure_comm_stream(nsx_handle *h) {
  if (h->comm_stream) return;
  int lo = 0, hi = 0;  // numerically lowest = highest priority
  HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  // ... is what one would want.  Measured (bench.py --comm rccl1): with a communication stream of ANY priority other than the default
  // every kernel of the compute stream runs three times slower (spmv_F 31 -> 92 us, with the highest and with the lowest priority
  // alike), so the default priority it is; whether the two streams really run side by side is PROBED (comm_streams_concurrent)
  // before the sweep relies on it.
  const int prio = getenv("NSX_COMM_PRIO") ? atoi(getenv("NSX_COMM_PRIO")) : 0;  // 1 highest, 0 default, -1 lowest
  HIP_CHECK(hipStreamCreateWithPriority(&h->comm_stream, hipStreamNonBlocking, prio > 0 ? hi : prio < 0 ? lo : (lo + hi) / 2));
  HIP_CHECK(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
}

// A stream that RCCL has launched on must outlive the communicator: RCCL orders the operations of one communicator across the user's
// streams through events it records on the stream of the PREVIOUS operation, and looks at that stream again when the communicator is
// destroyed.  Streams the handle no longer uses are parked here and destroyed behind ncclCommDestroy (comm_destroy).
static void retire_stream(nsx_handle *h, hipStream_t s) {
  if (s) h->retired_streams.push_back(s);
}

__global__ void k_probe_wait(const unsigned long long *flag, int *seen) {
  const unsigned long long t0 = wall_clock64();
  *seen = 0;
  for (;;) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
      *seen = 1;
      return;
    }
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 2000000ull) return;  // 20 ms
  }
}
// Do kernels of the communication stream run WHILE a kernel of the compute stream is waiting for them -- here, now?  A one-thread
// kernel on the compute stream waits (at most 20 ms) for a word that a one-thread kernel on the communication stream stores.
static bool probe_streams_local(nsx_handle *h) {
  DevBuf<unsigned long long> word;
  word.alloc(2);
  word.zero(h->stream);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  HIP_CHECK(hipStreamSynchronize(h->comm_stream));
  hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, h->stream, word.p, (int *)(word.p + 1));
  hipLaunchKernelGGL(k_ext_release, dim3(1), dim3(1), 0, h->comm_stream, word.p, 1ull);
  HIP_CHECK(hipStreamSynchronize(h->comm_stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  int seen = 0;
  HIP_CHECK(hipMemcpy(&seen, word.p + 1, sizeof(int), hipMemcpyDeviceToHost));
  return seen != 0;
}
// The runtime deals streams to a few hardware queues in the order of their creation, and two streams on one queue run in order: whether
// the communication stream got a queue of its own depends on how many streams this process has created before (a second handle in
// one process found itself on its compute stream's queue).  So the stream is created when the communicator is -- before anything
// has been enqueued on it -- probed, and replaced by the next one the runtime hands out until the probe passes (at most 6 times).
void comm_prepare_streams(nsx_handle *h) {
  h->comm_probe_local = 0;
  for (int attempt = 0; attempt < 6 && !h->comm_probe_local; ++attempt) {
    ensure_comm_stream(h);
    if (probe_streams_local(h)) {
      h->comm_probe_local = 1;
    } else {
      (void)hipEventDestroy(h->ev_ready);
      h->ev_ready = nullptr;
      (void)hipStreamDestroy(h->comm_stream);
      h->comm_stream = nullptr;
    }
  }
  ensure_comm_stream(h);
  if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] communication stream runs beside a waiting compute kernel: %d\n", h->comm_probe_local);
}
// A persistent grid whose waves allocate 256 VGPRs leaves RCCL's generic kernel (264 per lane) no place on any CU it touches, and the
// dispatcher touches them all.  A compute stream with a CU mask does: bit i of the mask is CU i / 8 of XCD i % 8
// (profiles/r04_cu_mask_bits.txt, tools/cu_mask_probe.hip), RCCL's workgroup k goes to XCD k % 8, so clearing bits 0..7 keeps one CU
// of every XCD out of the compute stream's reach -- 3 % of the device -- and the communication stream (no mask) finds it free.  From
// then on h->stream is the masked stream; the plain one is kept until the handle goes.  (A grid on the masked stream is co-resident up
// to 2 x 7 workgroups per shader engine, 448 in all: the workgroups of an XCD are dealt to its four shader engines in turn, and the
// one that lost a CU holds 14 -- profiles/r04_cu_mask_residency.txt.)
bool comm_reserve_cus(nsx_handle *h) {
  if (h->cu_reserved) return true;
  int cus = 0;
  HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->prm.device));
  if (cus < 64 || cus % 8) return false;
  std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xffffffffu);
  if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
  mask[0] &= ~0xffu;
  hipStream_t masked = nullptr;
  if (hipExtStreamCreateWithCUMask(&masked, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  HIP_CHECK(hipStreamSynchronize(h->stream));
  hipStream_t plain = h->stream;
  h->stream = masked;
  // The communication stream gets the complementary mask -- the eight reserved CUs and nothing else: an unmasked stream's workgroup is
  // dealt to ONE shader engine of its XCD whatever is free there, and three of the four are full while the grid waits (a one-thread
  // release kernel sat in such a queue until the grid gave up).  It must not share a hardware queue with the compute stream either.
  auto drop_comm_stream = [&] {
    if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
    h->ev_ready = nullptr;
    retire_stream(h, h->comm_stream);
    h->comm_stream = nullptr;
  };
  std::vector<uint32_t> cmask(mask.size(), 0u);
  cmask[0] = 0xffu;
  bool ok = false;
  for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
    drop_comm_stream();
    if (hipExtStreamCreateWithCUMask(&h->comm_stream, (uint32_t)cmask.size(), cmask.data()) != hipSuccess) {
      (void)hipGetLastError();
      h->comm_stream = nullptr;
      break;
    }
    HIP_CHECK(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    ok = probe_streams_local(h);
  }
  if (!ok) {
    drop_comm_stream();
    h->stream = plain;
    (void)hipStreamDestroy(masked);  // (nothing of RCCL has run on it yet)
    ensure_comm_stream(h);
    h->comm_probe_local = probe_streams_local(h) ? 1 : 0;
    return false;
  }
  h->stream_plain = plain;
  h->cu_reserved = 8;
  if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] compute stream replaced by one that leaves one CU per XCD to the communication stream\n");
  return true;
}
// The way back: the plain compute stream and an unmasked communication stream, as the handle had them before comm_reserve_cus
// (a rank whose peers could not reserve; a communicator that goes away -- the next one decides again).
void comm_release_cus(nsx_handle *h) {
  if (!h->cu_reserved) return;
  HIP_CHECK(hipStreamSynchronize(h->stream));
  if (h->comm_stream) HIP_CHECK(hipStreamSynchronize(h->comm_stream));
  retire_stream(h, h->stream);
  h->stream = h->stream_plain;
  h->stream_plain = nullptr;
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  h->ev_ready = nullptr;
  retire_stream(h, h->comm_stream);
  h->comm_stream = nullptr;
  h->cu_reserved = 0;
  if (h->comm) comm_prepare_streams(h);
  if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] compute stream back on all CUs\n");
}
// ... on every rank?  The ranks take the minimum of their answers (one collective in the lifetime of a handle), so that all of them
// use the collective-inside-the-grid sweep or none does.
bool comm_streams_concurrent(nsx_handle *h) {
  Comm *c = h->comm;
  if (!c || !c->comm) return false;
  if (h->comm_probe_local < 0) comm_prepare_streams(h);
  DevBuf<double> ans;
  ans.alloc(1);
  double v = h->comm_probe_local ? 1.0 : 0.0;
  HIP_CHECK(hipMemcpy(ans.p, &v, sizeof(double), hipMemcpyHostToDevice));
  h->n_allreduce++;
  NCCL_CHECK(ncclAllReduce(ans.p, ans.p, 1, ncclDouble, ncclMin, c->comm, h->stream));
  HIP_CHECK(hipStreamSynchronize(h->stream));
  HIP_CHECK(hipMemcpy(&v, ans.p, sizeof(double), hipMemcpyDeviceToHost));
  if (getenv("NSX_DEBUG")) fprintf(stderr, "[nsx] communication stream beside the compute stream: here %d, on all ranks %d\n", h->comm_probe_local, (int)(v > 0.5));
  return v > 0.5;
}

void comm_ext_allreduce(nsx_handle *h, double *vals, int count, int fail_word, unsigned int *arrive, unsigned int expected, unsigned long long *flag,
                        unsigned long long seq) {
  Comm *c = h->comm;
  if (!c || !c->comm) NSX_THROW(NSX_ERR_COMM, "internal: stream collective without an RCCL communicator");
  ensure_comm_stream(h);
  h->n_allreduce++;
  hipLaunchKernelGGL(k_ext_wait, dim3(1), dim3(1), 0, h->comm_stream, arrive, expected, vals, fail_word);
  if (self_p2p_mode(h) >= 1) self_p2p(h, h->comm_stream);  // development: a real RCCL kernel beside the grid that waits for this collective
  NCCL_CHECK(ncclAllReduce(vals, vals, count, ncclDouble, ncclSum, c->comm, h->comm_stream));
  // fault injection (tests, NSX_EXT_LATE_RELEASE=k): the k-th collective of this kind never tells its grid that it is complete -- what a
  // collective that arrives after the grid's bounded wait looks like to ONE rank
  const char *late_env = getenv("NSX_EXT_LATE_RELEASE");  // (read per call: the tests switch it inside one process)
  const int late = late_env ? atoi(late_env) : 0;
  if (late > 0 && ++h->n_ext_collectives == late) return;
  hipLaunchKernelGGL(k_ext_release, dim3(1), dim3(1), 0, h->comm_stream, flag, seq);
}

template <int NC>
__global__ void k_pack(int n, const int32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ buf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = idx[k];
#pragma unroll
  for (int c = 0; c < NC; ++c) buf[(size_t)k * NC + c] = x[(size_t)i * NC + c];
}

static void launch_pack(nsx_handle *h, HaloPlan &p, const double *x, int ncomp, int n_send, hipStream_t st) {
  if (!n_send) return;
  if (ncomp == 1) hipLaunchKernelGGL((k_pack<1>), dim3(cdiv(n_send, 256)), dim3(256), 0, st, n_send, p.send_idx.p, x, p.sendbuf.p);
  else if (ncomp == 2) hipLaunchKernelGGL((k_pack<2>), dim3(cdiv(n_send, 256)), dim3(256), 0, st, n_send, p.send_idx.p, x, p.sendbuf.p);
  else hipLaunchKernelGGL((k_pack<3>), dim3(cdiv(n_send, 256)), dim3(256), 0, st, n_send, p.send_idx.p, x, p.sendbuf.p);
}

// Refresh the ghost part of x (node-major, ncomp values per node) from the owners — the Epetra_Import of every vmult.
// begin: on the communication stream, behind everything the compute stream holds so far: pack, then (RCCL) the grouped
// send / receive straight into the ghost region, or (callbacks) the copy of the packed values to the host.
void comm_halo_begin(nsx_handle *h, HaloPlan &p, double *x, int ncomp,
                     const std::function<void(hipStream_t, double *sendbuf, const int32_t *send_idx, int n_send)> *packer) {
  Comm *c = h->comm;
  const int nn = (int)p.nbr.size();
  if (nn == 0) return;
  if (!c || (c->world == 1 && !p.self_test)) NSX_THROW(NSX_ERR_COMM, "distributed mesh set but no communicator: call nsx_comm_init* first");
  ensure_comm_stream(h);
  if (!p.ev_done) HIP_CHECK(hipEventCreateWithFlags(&p.ev_done, hipEventDisableTiming));
  const int n_send = p.send_ptr[nn];
  h->n_halo++;
  HIP_CHECK(hipEventRecord(h->ev_ready, h->stream));
  HIP_CHECK(hipStreamWaitEvent(h->comm_stream, h->ev_ready, 0));
  if (packer) (*packer)(h->comm_stream, p.sendbuf.p, p.send_idx.p, n_send);
  else launch_pack(h, p, x, ncomp, n_send, h->comm_stream);
  double *ghost = x + (size_t)p.n_own * ncomp;
  if (c->comm) {
    NCCL_CHECK(ncclGroupStart());
    for (int k = 0; k < nn; ++k) {
      const size_t ns = (size_t)(p.send_ptr[k + 1] - p.send_ptr[k]) * ncomp, nr = (size_t)(p.recv_ptr[k + 1] - p.recv_ptr[k]) * ncomp;
      if (ns) NCCL_CHECK(ncclSend(p.sendbuf.p + (size_t)p.send_ptr[k] * ncomp, ns, ncclDouble, p.nbr[k], c->comm, h->comm_stream));
      if (nr) NCCL_CHECK(ncclRecv(ghost + (size_t)p.recv_ptr[k] * ncomp, nr, ncclDouble, p.nbr[k], c->comm, h->comm_stream));
    }
    NCCL_CHECK(ncclGroupEnd());
    HIP_CHECK(hipEventRecord(p.ev_done, h->comm_stream));
  } else {
    p.h_send.resize((size_t)n_send * ncomp);
    p.h_recv.resize((size_t)p.recv_ptr[nn] * ncomp);
    if (n_send) HIP_CHECK(hipMemcpyAsync(p.h_send.data(), p.sendbuf.p, p.h_send.size() * sizeof(double), hipMemcpyDeviceToHost, h->comm_stream));
    p.in_flight = true;
  }
}

// finish: the compute stream waits for the ghosts.  Callback backend: the host exchange happens here, while the kernels
// enqueued since comm_halo_begin (the interior rows) run on the device.
void comm_halo_finish(nsx_handle *h, HaloPlan &p, double *x, int ncomp) {
  Comm *c = h->comm;
  const int nn = (int)p.nbr.size();
  if (nn == 0) return;
  if (!c->comm) {
    if (!p.in_flight) NSX_THROW(NSX_ERR_COMM, "internal: comm_halo_finish without comm_halo_begin");
    p.in_flight = false;
    HIP_CHECK(hipStreamSynchronize(h->comm_stream));
    std::vector<const double *> sp(nn);
    std::vector<double *> rp(nn);
    std::vector<int> sc(nn), rc(nn), ranks(nn);
    for (int k = 0; k < nn; ++k) {
      ranks[k] = p.nbr[k];
      sp[k] = p.h_send.data() + (size_t)p.send_ptr[k] * ncomp;
      rp[k] = p.h_recv.data() + (size_t)p.recv_ptr[k] * ncomp;
      sc[k] = (p.send_ptr[k + 1] - p.send_ptr[k]) * ncomp;
      rc[k] = (p.recv_ptr[k + 1] - p.recv_ptr[k]) * ncomp;
    }
    if (c->exchange(c->ctx, nn, ranks.data(), sp.data(), sc.data(), rp.data(), rc.data())) NSX_THROW(NSX_ERR_COMM, "exchange callback failed");
    double *ghost = x + (size_t)p.n_own * ncomp;
    if (!p.h_recv.empty()) HIP_CHECK(hipMemcpyAsync(ghost, p.h_recv.data(), p.h_recv.size() * sizeof(double), hipMemcpyHostToDevice, h->comm_stream));
    HIP_CHECK(hipStreamSynchronize(h->comm_stream));  // the staging vector is reused by the next exchange
    HIP_CHECK(hipEventRecord(p.ev_done, h->comm_stream));
  }
  HIP_CHECK(hipStreamWaitEvent(h->stream, p.ev_done, 0));
}

void comm_halo(nsx_handle *h, HaloPlan &p, double *x, int ncomp) {
  if (p.nbr.empty()) return;
  const int nn = (int)p.nbr.size();
  LaunchScope ls(h, ncomp == 1 ? "halo_p" : "halo_u", 16.0 * (p.send_ptr[nn] + p.recv_ptr[nn]) * ncomp);
  comm_halo_begin(h, p, x, ncomp);
  comm_halo_finish(h, p, x, ncomp);
}

void comm_destroy(nsx_handle *h) {
  // order: everything enqueued has run -> the communicator goes (RCCL looks at the streams of its last operations) -> the streams go
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);
  if (h->comm) {
    if (h->comm->comm) (void)ncclCommDestroy(h->comm->comm);
    delete h->comm;
    h->comm = nullptr;
  }
  // a masked compute stream belongs to the communicator that asked for it: the handle goes back to its plain stream (and every kernel to all CUs)
  if (h->cu_reserved) {
    try {
      comm_release_cus(h);
    } catch (const Error &) {
    }
  }
  h->cu_reserve_failed = false;
  h->mgs_leave_req = false;
  h->mgs_local_timeouts = 0;
  for (HaloPlan *p : {&h->haloU, &h->haloP})
    if (p->ev_done) {
      (void)hipEventDestroy(p->ev_done);
      p->ev_done = nullptr;
    }
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  h->ev_ready = nullptr;
  if (h->comm_stream) (void)hipStreamDestroy(h->comm_stream);
  h->comm_stream = nullptr;
  for (hipStream_t s_ : h->retired_streams) (void)hipStreamDestroy(s_);
  h->retired_streams.clear();
  h->comm_probe_local = -1;
}

}  // namespace nsx

extern "C" {

int nsx_comm_unique_id(uint8_t id[128]) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return NSX_ERR_COMM;
  memcpy(id, &u, 128);
  return NSX_OK;
}

int nsx_comm_init(nsx_handle *h, int rank, int world, const uint8_t id[128]) {
  if (!h || !id || world < 1 || rank < 0 || rank >= world) return NSX_ERR_ARG;
  try {
    HIP_CHECK(hipSetDevice(h->prm.device));
    nsx::comm_destroy(h);
    h->comm = new nsx::Comm;
    h->comm->rank = rank;
    h->comm->world = world;
    ncclUniqueId u;
    memcpy(&u, id, 128);
    NCCL_CHECK(ncclCommInitRank(&h->comm->comm, world, u, rank));
    h->self_p2p = getenv("NSX_EXT_SELF_P2P") ? atoi(getenv("NSX_EXT_SELF_P2P")) : 0;
    nsx::comm_prepare_streams(h);
    h->mgs_dist_state = -1;  // a new communicator: the paths the ranks choose together are chosen again
    h->mgs_dist_fit.clear();
    h->cgd_agreed = -1;
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}

// Test hook (tests/test_gpu_distributed.py): the RCCL branch of the ghost exchange -- pack kernel, grouped ncclSend / ncclRecv straight
// into the ghost region, event, wait of the compute stream -- on a 1-rank communicator, i.e. with this rank as its own and only
// neighbour (RCCL refuses two ranks on one device, so a one-GPU box has no other way to execute that code).  A vector of n_own + n_ghost
// nodes with ncomp values each; ghost k must receive the values of owned node (7 k + 3) % n_own.  max_err = largest deviation.
int nsx_comm_self_halo_test(nsx_handle *h, int n_own, int n_ghost, int ncomp, double *max_err) {
  if (!h || !max_err || n_own < 1 || n_ghost < 1 || ncomp < 1 || ncomp > 3) return NSX_ERR_ARG;
  try {
    HIP_CHECK(hipSetDevice(h->prm.device));
    if (!h->comm || !h->comm->comm || h->comm->world != 1) NSX_THROW(NSX_ERR_ARG, "nsx_comm_self_halo_test needs a 1-rank RCCL communicator (nsx_comm_init)");
    nsx::HaloPlan p;
    p.self_test = true;
    p.n_own = n_own;
    p.nbr = {0};
    p.send_ptr = {0, n_ghost};
    p.recv_ptr = {0, n_ghost};
    std::vector<int32_t> idx(n_ghost);
    for (int k = 0; k < n_ghost; ++k) idx[k] = (int32_t)((7ll * k + 3) % n_own);
    p.send_idx.upload(idx, h->stream);
    p.sendbuf.alloc((size_t)n_ghost * ncomp);
    std::vector<double> x((size_t)(n_own + n_ghost) * ncomp, -1.0);
    for (size_t i = 0; i < (size_t)n_own * ncomp; ++i) x[i] = 0.5 + (double)i;
    nsx::DevBuf<double> xd;
    xd.upload(x, h->stream);
    for (int rep = 0; rep < 3; ++rep) {  // several exchanges in a row: the event and the buffers are reused
      nsx::comm_halo_begin(h, p, xd.p, ncomp);
      nsx::comm_halo_finish(h, p, xd.p, ncomp);
    }
    xd.download(x.data(), x.size(), h->stream);
    double e = 0.0;
    for (int k = 0; k < n_ghost; ++k)
      for (int c = 0; c < ncomp; ++c) e = std::max(e, std::fabs(x[((size_t)n_own + k) * ncomp + c] - (0.5 + (double)((size_t)idx[k] * ncomp + c))));
    *max_err = e;
    if (p.ev_done) (void)hipEventDestroy(p.ev_done);
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
  return NSX_OK;
}

int nsx_comm_counters(const nsx_handle *h, long long counts[2]) {
  if (!h || !counts) return NSX_ERR_ARG;
  counts[0] = h->n_allreduce;
  counts[1] = h->n_halo;
  return NSX_OK;
}

int nsx_comm_init_callbacks(nsx_handle *h, int rank, int world, nsx_allreduce_fn allreduce, nsx_exchange_fn exchange, void *ctx) {
  if (!h || world < 1 || rank < 0 || rank >= world || !allreduce || !exchange) return NSX_ERR_ARG;
  nsx::comm_destroy(h);
  h->comm = new nsx::Comm;
  h->comm->rank = rank;
  h->comm->world = world;
  h->comm->allreduce = allreduce;
  h->comm->exchange = exchange;
  h->comm->ctx = ctx;
  h->mgs_dist_state = -1;
  h->mgs_dist_fit.clear();
  h->cgd_agreed = -1;
  return NSX_OK;
}

}  // extern "C"

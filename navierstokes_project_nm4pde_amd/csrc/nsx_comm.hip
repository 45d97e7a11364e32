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

#include "nsx_internal.hpp"

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

void comm_allreduce_scalars(nsx_handle *h, int slot0, int count) {
  Comm *c = h->comm;
  if (!c || (c->world == 1 && !c->comm)) return;  // a 1-rank RCCL communicator still runs the collective (API self-test)
  if (c->comm) {
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
  if (c->comm) {
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

template <int NC>
__global__ void k_pack(int n, const int32_t *__restrict__ idx, const double *__restrict__ x, double *__restrict__ buf) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int i = idx[k];
#pragma unroll
  for (int c = 0; c < NC; ++c) buf[(size_t)k * NC + c] = x[(size_t)i * NC + c];
}

// refresh the ghost part of x (node-major, ncomp values per node) from the owners
void comm_halo(nsx_handle *h, HaloPlan &p, double *x, int ncomp) {
  Comm *c = h->comm;
  const int nn = (int)p.nbr.size();
  if (nn == 0) return;
  if (!c || c->world == 1) NSX_THROW(NSX_ERR_COMM, "distributed mesh set but no communicator: call nsx_comm_init* first");
  const int n_send = p.send_ptr[nn];
  LaunchScope ls(h, ncomp == 1 ? "halo_p" : "halo_u", 16.0 * (n_send + p.recv_ptr[nn]) * ncomp);
  if (n_send) {
    if (ncomp == 1) hipLaunchKernelGGL((k_pack<1>), dim3(cdiv(n_send, 256)), dim3(256), 0, h->stream, n_send, p.send_idx.p, x, p.sendbuf.p);
    else if (ncomp == 2) hipLaunchKernelGGL((k_pack<2>), dim3(cdiv(n_send, 256)), dim3(256), 0, h->stream, n_send, p.send_idx.p, x, p.sendbuf.p);
    else hipLaunchKernelGGL((k_pack<3>), dim3(cdiv(n_send, 256)), dim3(256), 0, h->stream, n_send, p.send_idx.p, x, p.sendbuf.p);
  }
  double *ghost = x + (size_t)p.n_own * ncomp;
  if (c->comm) {
    NCCL_CHECK(ncclGroupStart());
    for (int k = 0; k < nn; ++k) {
      const size_t ns = (size_t)(p.send_ptr[k + 1] - p.send_ptr[k]) * ncomp, nr = (size_t)(p.recv_ptr[k + 1] - p.recv_ptr[k]) * ncomp;
      if (ns) NCCL_CHECK(ncclSend(p.sendbuf.p + (size_t)p.send_ptr[k] * ncomp, ns, ncclDouble, p.nbr[k], c->comm, h->stream));
      if (nr) NCCL_CHECK(ncclRecv(ghost + (size_t)p.recv_ptr[k] * ncomp, nr, ncclDouble, p.nbr[k], c->comm, h->stream));
    }
    NCCL_CHECK(ncclGroupEnd());
  } else {
    p.h_send.resize((size_t)n_send * ncomp);
    p.h_recv.resize((size_t)p.recv_ptr[nn] * ncomp);
    if (n_send) HIP_CHECK(hipMemcpyAsync(p.h_send.data(), p.sendbuf.p, p.h_send.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
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
    if (!p.h_recv.empty()) HIP_CHECK(hipMemcpyAsync(ghost, p.h_recv.data(), p.h_recv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_CHECK(hipStreamSynchronize(h->stream));
  }
}

void comm_destroy(nsx_handle *h) {
  if (!h->comm) return;
  if (h->comm->comm) (void)ncclCommDestroy(h->comm->comm);
  delete h->comm;
  h->comm = nullptr;
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
  } catch (const nsx::Error &e) {
    h->err = e.msg;
    return e.code;
  }
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
  return NSX_OK;
}

}  // extern "C"

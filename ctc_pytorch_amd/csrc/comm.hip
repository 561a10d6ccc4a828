// comm.hip -- the data-parallel exchange step of the path behind the C ABI: one RCCL communicator per rank, SUM all-reduce
// of (a slice of) the flat float32 gradient buffer over xGMI.
//
// The reference is single-device (train_ctc.py:63-65: loss.backward(); optimizer.step()); utterance-sharded data parallelism
// inserts exactly one collective between those two calls.  librccl.so is opened at run time (dlopen): libctcn.so has no link-time
// dependency on it, single-GPU users never load it, and inside a PyTorch process the loader hands back the librccl.so torch
// already mapped (same SONAME), so both share one runtime.  The communicator is the only object the library owns
// (ctcn_comm_init .. ctcn_comm_destroy); everything else follows the conventions of include/ctcn.h.
#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

// the four RCCL entry points used, with the ABI of rccl.h (NCCL 2.x): ncclUniqueId is a 128-byte opaque blob
struct UniqueId { char internal[128]; };
typedef int (*get_unique_id_fn)(UniqueId *);
typedef int (*comm_init_rank_fn)(void **comm, int nranks, UniqueId id, int rank);
typedef int (*all_reduce_fn)(const void *send, void *recv, size_t count, int datatype, int op, void *comm, hipStream_t stream);
typedef int (*comm_destroy_fn)(void *comm);
typedef const char *(*get_error_string_fn)(int);
constexpr int NCCL_FLOAT32 = 7, NCCL_SUM = 0;        // ncclDataType_t / ncclRedOp_t values of rccl.h

struct Rccl {
  void *handle = nullptr;
  char why[256] = "";          // dlerror() text of the LAST failed dlopen (dlerror clears itself when read: captured once, here)
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  all_reduce_fn all_reduce = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  get_error_string_fn get_error_string = nullptr;
};

Rccl &rccl_state() { static Rccl r; return r; }

// the communicators ctcn_comm_init handed out and ctcn_comm_destroy has not taken back: a handle is RCCL's own pointer, and RCCL dereferences
// whatever it is given -- a stale or foreign handle is answered with CTCN_EINVAL here instead of a crash there
struct LiveComms {
  std::mutex mu;
  std::vector<void *> live;
  bool has(void *c) { std::lock_guard<std::mutex> g(mu); return std::find(live.begin(), live.end(), c) != live.end(); }
  void add(void *c) { std::lock_guard<std::mutex> g(mu); live.push_back(c); }
  bool drop(void *c) {
    std::lock_guard<std::mutex> g(mu);
    auto it = std::find(live.begin(), live.end(), c);
    if (it == live.end()) return false;
    live.erase(it);
    return true;
  }
};
LiveComms &live_comms() { static LiveComms l; return l; }
Rccl *rccl() {
  Rccl &r = rccl_state();
  static bool tried = false;
  if (!tried) {
    tried = true;
    const char *names[3] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (int i = 0; i < 3 && !r.handle; ++i) {
      r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
      if (!r.handle) {
        const char *e = dlerror();
        snprintf(r.why, sizeof(r.why), "%s", e ? e : "dlopen failed");
      }
    }
    if (r.handle) {
      r.get_unique_id = (get_unique_id_fn)dlsym(r.handle, "ncclGetUniqueId");
      r.comm_init_rank = (comm_init_rank_fn)dlsym(r.handle, "ncclCommInitRank");
      r.all_reduce = (all_reduce_fn)dlsym(r.handle, "ncclAllReduce");
      r.comm_destroy = (comm_destroy_fn)dlsym(r.handle, "ncclCommDestroy");
      r.get_error_string = (get_error_string_fn)dlsym(r.handle, "ncclGetErrorString");
      if (!(r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy)) snprintf(r.why, sizeof(r.why), "missing ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    }
  }
  return (r.handle && r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy) ? &r : nullptr;
}

int fail(const char *what, int rc) {
  Rccl *r = rccl();
  ctcn_set_error("%s: RCCL error %d (%s)", what, rc, r && r->get_error_string ? r->get_error_string(rc) : "?");
  return CTCN_EHIP;
}

}  // namespace

extern "C" int ctcn_comm_unique_id(void *id128) {
  CTCN_REQUIRE(id128, "ctcn_comm_unique_id: null pointer");
  Rccl *r = rccl();
  if (!r) { ctcn_set_error("ctcn_comm_unique_id: librccl.so could not be loaded (%s)", rccl_state().why); return CTCN_EUNSUPPORTED; }
  UniqueId id;
  const int rc = r->get_unique_id(&id);
  if (rc) return fail("ctcn_comm_unique_id", rc);
  memcpy(id128, &id, sizeof(id));
  return CTCN_OK;
}

extern "C" int ctcn_comm_init(const void *id128, int rank, int world, void **comm) {
  CTCN_REQUIRE(id128 && comm && world >= 1 && rank >= 0 && rank < world, "ctcn_comm_init: bad arguments (rank %d of %d)", rank, world);
  Rccl *r = rccl();
  if (!r) { ctcn_set_error("ctcn_comm_init: librccl.so could not be loaded (%s)", rccl_state().why); return CTCN_EUNSUPPORTED; }
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  void *c = nullptr;
  const int rc = r->comm_init_rank(&c, world, id, rank);
  if (rc) return fail("ctcn_comm_init", rc);
  live_comms().add(c);
  *comm = c;
  return CTCN_OK;
}

extern "C" int ctcn_comm_allreduce_sum_f32(void *comm, float *buf, size_t n, void *stream) {
  CTCN_REQUIRE(comm && (buf || n == 0), "ctcn_comm_allreduce_sum_f32: null pointer");
  CTCN_REQUIRE(live_comms().has(comm), "ctcn_comm_allreduce_sum_f32: not a communicator of ctcn_comm_init (or already destroyed)");
  if (n == 0) return CTCN_OK;
  Rccl *r = rccl();
  if (!r) { ctcn_set_error("ctcn_comm_allreduce_sum_f32: librccl.so is not loaded"); return CTCN_EUNSUPPORTED; }
  const int rc = r->all_reduce(buf, buf, n, NCCL_FLOAT32, NCCL_SUM, comm, (hipStream_t)stream);     // in place, enqueued on `stream`
  if (rc) return fail("ctcn_comm_allreduce_sum_f32", rc);
  return CTCN_OK;
}

extern "C" int ctcn_comm_destroy(void *comm) {
  if (!comm) return CTCN_OK;
  CTCN_REQUIRE(live_comms().drop(comm), "ctcn_comm_destroy: not a communicator of ctcn_comm_init (or already destroyed)");
  Rccl *r = rccl();
  if (!r) { ctcn_set_error("ctcn_comm_destroy: librccl.so is not loaded"); return CTCN_EUNSUPPORTED; }
  const int rc = r->comm_destroy(comm);
  if (rc) return fail("ctcn_comm_destroy", rc);
  return CTCN_OK;
}

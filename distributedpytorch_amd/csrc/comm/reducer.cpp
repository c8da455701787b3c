// Native gradient reducer core (N3 of SURVEY.md §2.4: the role c10d::Reducer plays for the reference's DDP wrap,
// classif.py:138).  The gradients already live in ONE flat fp32 arena, so a bucket is a contiguous [lo, hi) range:
// no flatten/unflatten copies.  This object owns the per-iteration bookkeeping and the launch decision:
//
//   begin()                      reset the per-bucket pending counters
//   param_ready(param, stream)   called from the backward tape when a parameter's gradient kernel has been queued;
//                                when it was the bucket's last one, the bucket's in-place all-reduce is enqueued on the
//                                communicator's stream, ordered after `stream` with an event (rccl_comm.cpp)
//   wait_bucket(b, stream)       makes `stream` wait (device side) for bucket b; the caller then launches the fused
//                                optimizer for [lo, hi) so the update of bucket k overlaps the all-reduce of k+1
//
// With comm == nullptr only the bookkeeping runs (world_size 1, or the torch.distributed/gloo backend issues the
// collective itself when param_ready() returns the bucket index).
#include <cstdio>
#include <cstring>
#include <vector>

#define DPA_API extern "C" __attribute__((visibility("default")))

extern "C" int dpa_comm_allreduce_async(void* h, float* ptr, long long count, int bucket, void* compute);
extern "C" int dpa_comm_wait(void* h, int bucket, void* compute);

namespace {
struct Reducer {
  void* comm = nullptr;
  float* grads = nullptr;
  std::vector<int> bucket_of;            // parameter index -> bucket (-1: frozen parameter)
  std::vector<int> size, pending;        // gradients per bucket / still missing this iteration
  std::vector<long long> lo, hi;         // arena element range of each bucket
  std::vector<char> launched, seen;      // per bucket / per parameter (double-ready detection)
  long long launched_bytes = 0;
  int order_violations = 0;              // buckets that became ready out of plan order (diagnostic)
  int next_expected = 0;
};
}  // namespace

DPA_API void* dpa_reducer_create(void* comm, float* grads, int n_params, const int* bucket_of, int n_buckets,
                                 const long long* lo, const long long* hi) {
  Reducer* r = new Reducer();
  r->comm = comm; r->grads = grads;
  r->bucket_of.assign(bucket_of, bucket_of + n_params);
  r->size.assign(n_buckets, 0);
  for (int i = 0; i < n_params; ++i) {
    if (bucket_of[i] >= n_buckets) { delete r; return nullptr; }
    if (bucket_of[i] >= 0) r->size[bucket_of[i]]++;
  }
  r->lo.assign(lo, lo + n_buckets); r->hi.assign(hi, hi + n_buckets);
  r->pending = r->size;
  r->launched.assign(n_buckets, 0);
  r->seen.assign(n_params, 0);
  return r;
}
DPA_API void dpa_reducer_begin(void* h) {
  Reducer* r = (Reducer*)h;
  r->pending = r->size;
  std::fill(r->launched.begin(), r->launched.end(), 0);
  std::fill(r->seen.begin(), r->seen.end(), 0);
  r->next_expected = 0;
}
// Returns the bucket that just became ready (>= 0), -1 if none, -2 on a protocol error (unknown / doubly-ready
// parameter), -3 if the collective could not be enqueued.
DPA_API int dpa_reducer_param_ready(void* h, int param, void* compute) {
  Reducer* r = (Reducer*)h;
  if (param < 0 || param >= (int)r->bucket_of.size()) return -2;
  const int b = r->bucket_of[param];
  if (b < 0) return -1;
  if (r->seen[param]) return -2;
  r->seen[param] = 1;
  if (--r->pending[b] != 0) return -1;
  if (b != r->next_expected) r->order_violations++;
  r->next_expected = b + 1;
  r->launched[b] = 1;
  r->launched_bytes += (r->hi[b] - r->lo[b]) * 4;
  if (r->comm) {
    if (dpa_comm_allreduce_async(r->comm, r->grads + r->lo[b], r->hi[b] - r->lo[b], b, compute) != 0) return -3;
  }
  return b;
}
// 0 ok, 1 = the bucket never became ready (a trainable parameter took no part in backward), 2 = wait failed.
DPA_API int dpa_reducer_wait_bucket(void* h, int b, void* compute) {
  Reducer* r = (Reducer*)h;
  if (b < 0 || b >= (int)r->pending.size()) return 1;
  if (r->pending[b] != 0 || !r->launched[b]) return 1;
  if (r->comm && dpa_comm_wait(r->comm, b, compute) != 0) return 2;
  return 0;
}
DPA_API int dpa_reducer_pending(void* h, int b) { return ((Reducer*)h)->pending[b]; }
DPA_API int dpa_reducer_order_violations(void* h) { return ((Reducer*)h)->order_violations; }
DPA_API long long dpa_reducer_launched_bytes(void* h) { return ((Reducer*)h)->launched_bytes; }
DPA_API void dpa_reducer_destroy(void* h) { delete (Reducer*)h; }

// collide.h -- contact generation on the GPU (see collide.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace egs {

// pos [n][3], R [n][9] row-major, side [n][3] on the host.  Writes the contact
// list in the reference's order (ground contacts by body, then pairs i < j):
// body0/body1 [m], data [m][7] = position, normal, depth.  Returns m.
// Throws std::invalid_argument if m > max_contacts or a body has more than 64
// overlapping partners.
int update_contacts(hipStream_t s, int n, const double *pos, const double *R, const double *side, int max_contacts,
                    int32_t *body0, int32_t *body1, double *data, int *n_ground, int *n_pairs, int mj = 0,
                    const int32_t *jb0 = nullptr, const int32_t *jb1 = nullptr, const double *jdata = nullptr);

// Device-resident form: body state already on the device, the contact list stays
// there (body0()/body1()/data() are device pointers valid until the next run()).
class Collider {
 public:
  Collider();
  ~Collider();
  Collider(const Collider &) = delete;
  Collider &operator=(const Collider &) = delete;
  // optional joints (device arrays, body-body only matter): contacts within 1e-6 of a
  // joint between the same two bodies are dropped (ensembles.cc:296-306)
  int run(hipStream_t s, int n, const double *dpos, const double *dR, const double *dside, int mj = 0,
          const int32_t *djb0 = nullptr, const int32_t *djb1 = nullptr, const double *djdata = nullptr);  // returns m
  // body0()/body1() of the last run() written to device-visible host memory (hipHostMalloc)
  void export_topology(hipStream_t s, int m, int32_t *mapped_b0, int32_t *mapped_b1) const;
  const int32_t *body0() const;
  const int32_t *body1() const;
  const double *data() const;
  int n_ground() const { return n_ground_; }
  int n_pairs() const { return n_pairs_; }

 private:
  struct Impl;
  Impl *impl_;
  int n_ground_ = 0, n_pairs_ = 0;
};

}  // namespace egs

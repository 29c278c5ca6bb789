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
                    int32_t *body0, int32_t *body1, double *data, int *n_ground, int *n_pairs);

}  // namespace egs

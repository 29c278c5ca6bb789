// matvec_plan.h -- host-side schedule of the stand-alone block-sparse products
//   y = (J W J^T + eps I) x  and its L / U / D parts
// (sparse::CalculateSparse{JMJtX,Lx,Ux,Dx,...}, sparse_iterations_utils.cc:427-695).
//
// The products are computed through per-body sums u_b = sum_j J_jb^T x_j taken in list
// order, so the schedule is a partition of the constraint list into workgroup TILES:
//   * every constraint belongs to exactly one tile (lanes in ascending list index, so a
//     tile's J blocks are long contiguous runs of the J0 / J1 arrays);
//   * a body whose constraints all sit in one tile is PRIVATE to it: its sum never leaves
//     that workgroup's LDS;
//   * a body touched from several tiles is SHARED: the sides that touch it publish their
//     J^T x in a global entry array first (one small pre-pass over those constraints only),
//     and every tile that needs the body sums the entries in list order.
// Islands that fit a tile are packed whole (first fit, island order), so a pile of separate
// stacks has no shared body at all and J is read exactly once per product; an oversize
// island is cut into runs of its list and pays the pre-pass on its cut.
#pragma once
#include <cstdint>
#include <vector>

namespace egs {

struct MvLane {        // one per lane of a tile; 12 bytes
  int32_t cidx;        // constraint index, -1 = idle lane
  uint16_t slot0;      // tile-local slot of body0 (0 = the world / none)
  uint16_t slot1;
  uint16_t e0, e1;     // where the side's J^T x sits in the tile's LDS entry array
                       // (= the slot's seg + the side's rank among the tile's sides on that body)
};

struct MvTile {        // 16 bytes
  int32_t nslots;      // slots in use, slot 0 = the world
  int32_t slot_off;    // offset of the tile's slots in `slots`
  int32_t n_entries;   // LDS entries in use (<= 2 * block)
  int32_t n_shared;    // shared bodies among the slots
};

struct MvSlot {        // 20 bytes
  int32_t body;        // global body index (-1 for slot 0)
  int32_t seg;         // first LDS entry of the body's sides in this tile (list order)
  int32_t cnt;         // constraint sides on the body: its whole list, also when shared
  int32_t t_off;       // shared body: offset of its `cnt` entries in the global entry array, else -1
  int32_t ents_off;    // shared body: offset of its `cnt` ids in `ents`, else -1
};

struct MvBoundary {    // a constraint with a side on a shared body; 12 bytes
  int32_t cidx;
  int32_t t0, t1;      // index into the global entry array for side 0 / 1, -1 = nothing to publish
};

constexpr uint16_t kMvRemote = 0xFFFF;   // entry of a shared body that belongs to another tile

struct MatvecPlan {
  int n = 0, m = 0, block = 256;
  int n_tiles = 0, max_slots = 1, n_islands = 0;
  int n_shared_bodies = 0, n_shared_entries = 0;
  std::vector<MvLane> lanes;          // n_tiles * block
  std::vector<MvTile> tiles;
  std::vector<MvSlot> slots;
  // shared bodies only: per (tile, shared body) `cnt` ids in list order -- the LDS entry
  // of the side when it belongs to this tile, kMvRemote when it belongs to another one
  std::vector<uint16_t> ents;
  std::vector<MvBoundary> boundary;
};

// block = 128 or 256 constraints per tile.  Throws std::invalid_argument on bad indices.
MatvecPlan build_matvec_plan(int n_bodies, int m, const int32_t *body0, const int32_t *body1, int block);

}  // namespace egs

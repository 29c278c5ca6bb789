// plan.h -- host-side analysis of the constraint graph.
//
// The reference sweeps the ConstraintsList strictly in list order
// (sparse_iterations_utils.cc:159-243 forward, :292-373 backward).  Two
// constraints interact only through a shared body, so the sweep is a partial
// order: per body, its constraints must run in list order.  The plan
//   1. finds the connected components ("islands") of the constraint graph,
//   2. packs whole islands into workgroup-sized tiles (one workgroup owns all
//      constraints of every body it touches, so body accumulators live in LDS),
//   3. numbers, for every constraint side, its rank among the constraints of
//      that body (`pos`) and the body's constraint count (`cnt`): the device
//      code uses them as tickets to reproduce list order exactly,
//   4. routes islands larger than a tile to the cross-workgroup path.
#pragma once
#include <cstdint>
#include <vector>

namespace egs {

struct LaneDesc {      // one per lane of a tile; 16 bytes
  int32_t cidx;        // constraint index in the caller's list, -1 = idle lane
  uint16_t slot0;      // LDS slot of body0 (0 = world / none)
  uint16_t slot1;
  uint16_t pos0, cnt0; // rank of this constraint among body0's, and their count
  uint16_t pos1, cnt1;
};

struct GlobalDesc {    // one per constraint of the cross-workgroup path; 24 bytes
  int32_t cidx;
  int32_t body0, body1;
  int32_t pos0, cnt0;
  int32_t pos1, cnt1;
  int32_t pad;
};

struct Plan {
  int n = 0, m = 0, block = 256;
  int n_islands = 0, n_tiles = 0, max_slots = 1;
  int max_cnt = 1;                    // largest per-body constraint count (ticket period)
  std::vector<LaneDesc> lanes;        // n_tiles * block
  // Static time-stepped schedule of the same sweep (step_solve.hip).  level = depth of the constraint in
  // the list-order dependency DAG of one sweep (a constraint comes after the previous constraint of each
  // of its bodies).  With P >= every body's level span (last level - first level + 1) the update of
  // sweep s may run at time level + P * (s - 1): per body the times increase in list order and the next
  // sweep's first update comes after this sweep's last, which is all the sweep order asks for.
  std::vector<uint16_t> lane_level;   // per lane of `lanes`
  std::vector<int32_t> tile_period;   // per tile: largest body span in the tile (P above)
  std::vector<int32_t> tile_depth;    // per tile: largest level + 1
  int max_period = 1, max_depth = 1;
  bool levels_ok = true;              // false if a level does not fit 16 bits
  // Runs: a stretch of consecutive constraints on the SAME two bodies -- the contact points of a box face, as the
  // collider lists them -- is consecutive in both bodies' lists, so the timetable can treat up to four of them (a
  // chunk) as ONE node whose updates hand the two accumulators from lane to lane (DPP) instead of through LDS and a
  // barrier.  With runs = true, lane_level / tile_period / tile_depth count chunks (macro steps of up to four
  // updates); a chunk owns four adjacent lane slots aligned to 4 in its tile -- its members in list order, then
  // placeholders (cidx = -2: the chunk's slots and level, no constraint) -- and a lane's place in its chunk is
  // lane mod 4.  Only 4-lane plans built with max_run_tiles > 0, needing at most that many tiles, and with less than a quarter of padding (a chunk keeps
  // 1/4 of its wavefront's lanes busy: worth it where the chain latency sets the time, about one tile per CU).
  bool runs = false;
  std::vector<int32_t> tile_nslots;   // per tile, slots in use (slot 0 = world)
  std::vector<int32_t> tile_slot_off; // per tile, offset into slot_body
  std::vector<int32_t> slot_body;     // slot -> global body index (-1 for slot 0)
  std::vector<GlobalDesc> global;     // constraints of oversize islands, list order
  // Oversize islands cut into workgroup-sized PATCHES of bodies (BFS-grown):
  // a patch tile owns the constraints whose first body lies in the patch.  Every
  // body a patch touches has an LDS slot there (low 14 bits of LaneDesc::slot0/1).
  // A body touched by one patch only lives in that workgroup's LDS for good.  A
  // body touched by several patches is SHARED: its accumulator and ticket stay in
  // the LDS of the patch that updated it last and cross global memory only when the
  // list-order neighbour on that body belongs to another patch -- kPrevRemote /
  // kNextRemote on the side's slot say which hand-offs those are.  In
  // patch_slot_body a shared body is stored as -(body + 2).
  int n_patch_tiles = 0, patch_max_slots = 1;
  std::vector<LaneDesc> patch_lanes;        // n_patch_tiles * block
  std::vector<int32_t> patch_tile_nslots, patch_tile_slot_off, patch_slot_body;
  int n_shared_bodies = 0;
  // patch_lanes hold chunks of up to four consecutive constraints on the same two bodies, four slots each (members,
  // then placeholders with cidx = -2): see build_patches.  Only the 4-lane patch kernel understands them.
  bool patch_runs = false;
};

constexpr uint16_t kSlotMask = 0x3FFF;     // LDS slot number of a patch lane's side
constexpr uint16_t kNextRemote = 0x4000;   // the list-order successor on this body sits in another patch
constexpr uint16_t kPrevRemote = 0x8000;   // the list-order predecessor does

// Throws std::invalid_argument on out-of-range body indices.
// block = kAutoQuadBlock: the smallest of 64 / 128 / 256 that holds the largest island
// (the tile sizes of the 4-lanes-per-constraint schedule).
constexpr int kAutoQuadBlock = 0;
// recycle: a plan that is no longer needed; its vectors' storage is reused (a world re-plans on
// every contact-topology change: the 16 B/lane table alone is a fresh 270 KB mapping otherwise).
Plan build_plan(int n_bodies, int m, const int32_t *body0, const int32_t *body1,
                int block, Plan *recycle = nullptr, int max_run_tiles = 0);

// how many workgroups (CUs) an oversize island's patches may occupy at once: sizes the patches (plan.cpp::build_patches)
void set_patch_workgroups(int n);

}  // namespace egs

// matvec_plan.cpp -- tiles, private / shared bodies and entry lists of the
// stand-alone mat-vec (see matvec_plan.h).
#include "matvec_plan.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>

namespace egs {

namespace {
int uf_find(std::vector<int32_t> &parent, int a) {
  while (parent[a] != a) {
    parent[a] = parent[parent[a]];
    a = parent[a];
  }
  return a;
}
}  // namespace

MatvecPlan build_matvec_plan(int n_bodies, int m, const int32_t *body0, const int32_t *body1, int block) {
  if (n_bodies < 0 || m < 0 || (block != 128 && block != 256))
    throw std::invalid_argument("build_matvec_plan: bad sizes");
  for (int i = 0; i < m; ++i) {
    if (body0[i] < -1 || body0[i] >= n_bodies || body1[i] < -1 || body1[i] >= n_bodies)
      throw std::invalid_argument("build_matvec_plan: body index out of range");
    if (body0[i] >= 0 && body0[i] == body1[i])
      throw std::invalid_argument("build_matvec_plan: constraint with the same body on both sides");
  }
  MatvecPlan plan;
  plan.n = n_bodies; plan.m = m; plan.block = block;

  // islands, numbered by first appearance in list order
  std::vector<int32_t> parent(n_bodies);
  std::iota(parent.begin(), parent.end(), 0);
  for (int i = 0; i < m; ++i)
    if (body0[i] >= 0 && body1[i] >= 0) {
      const int a = uf_find(parent, body0[i]), b = uf_find(parent, body1[i]);
      if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }
  std::vector<int32_t> root_island(n_bodies, -1), cons_island(m), island_size;
  for (int i = 0; i < m; ++i) {
    const int b = body0[i] >= 0 ? body0[i] : body1[i];
    int isl;
    if (b < 0) { isl = (int)island_size.size(); island_size.push_back(0); }
    else {
      const int r = uf_find(parent, b);
      if (root_island[r] < 0) { root_island[r] = (int)island_size.size(); island_size.push_back(0); }
      isl = root_island[r];
    }
    cons_island[i] = isl;
    ++island_size[isl];
  }
  plan.n_islands = (int)island_size.size();

  // tiles: whole islands first-fit while they fit a tile; an oversize island is cut into
  // runs of `block` of its constraints in list order
  std::vector<int32_t> island_tile(plan.n_islands, -1), tile_fill, tile_reserved, cons_tile(m), cons_lane(m);
  int cur_small = -1;
  for (int i = 0; i < m; ++i) {
    const int isl = cons_island[i];
    int t = island_tile[isl];
    if (island_size[isl] <= block) {
      if (t < 0) {
        if (cur_small < 0 || tile_reserved[cur_small] + island_size[isl] > block) {
          cur_small = (int)tile_fill.size();
          tile_fill.push_back(0); tile_reserved.push_back(0);
        }
        t = island_tile[isl] = cur_small;
        tile_reserved[t] += island_size[isl];
      }
    } else if (t < 0 || tile_fill[t] == block) {
      t = island_tile[isl] = (int)tile_fill.size();
      tile_fill.push_back(0); tile_reserved.push_back(block);
    }
    cons_tile[i] = t;
    cons_lane[i] = tile_fill[t]++;
  }
  plan.n_tiles = (int)tile_fill.size();
  MvLane idle{-1, 0, 0, 0, 0};
  plan.lanes.assign((size_t)plan.n_tiles * block, idle);

  // per-body entry lists (list order) and shared bodies
  std::vector<int32_t> cnt(n_bodies, 0), first_tile(n_bodies, -1), pos0(m, 0), pos1(m, 0);
  std::vector<char> shared(n_bodies, 0);
  for (int i = 0; i < m; ++i)
    for (int side = 0; side < 2; ++side) {
      const int b = side ? body1[i] : body0[i];
      if (b < 0) continue;
      (side ? pos1 : pos0)[i] = cnt[b]++;
      if (first_tile[b] < 0) first_tile[b] = cons_tile[i];
      else if (first_tile[b] != cons_tile[i]) shared[b] = 1;
    }
  std::vector<int32_t> ent_off((size_t)n_bodies + 1, 0);
  for (int b = 0; b < n_bodies; ++b) ent_off[b + 1] = ent_off[b] + cnt[b];
  std::vector<int32_t> gent((size_t)ent_off[n_bodies]);
  for (int i = 0; i < m; ++i) {
    if (body0[i] >= 0) gent[(size_t)ent_off[body0[i]] + pos0[i]] = 2 * i;
    if (body1[i] >= 0) gent[(size_t)ent_off[body1[i]] + pos1[i]] = 2 * i + 1;
  }
  std::vector<int32_t> t_off(n_bodies, -1);
  for (int b = 0; b < n_bodies; ++b)
    if (shared[b]) { t_off[b] = plan.n_shared_entries; plan.n_shared_entries += cnt[b]; ++plan.n_shared_bodies; }
  for (int i = 0; i < m; ++i) {
    const int b0 = body0[i], b1 = body1[i];
    const bool s0 = b0 >= 0 && shared[b0], s1 = b1 >= 0 && shared[b1];
    if (s0 || s1) plan.boundary.push_back(MvBoundary{i, s0 ? t_off[b0] + pos0[i] : -1, s1 ? t_off[b1] + pos1[i] : -1});
  }

  // slots (first use inside the tile), LDS entries (a body's sides of this tile, list order)
  plan.tiles.resize(plan.n_tiles);
  std::vector<int32_t> stamp(n_bodies, -1), slot_in_tile(n_bodies, 0);
  for (int i = 0; i < m; ++i) plan.lanes[(size_t)cons_tile[i] * block + cons_lane[i]].cidx = i;
  std::vector<int32_t> local_e((size_t)2 * m, -1);   // (constraint, side) -> LDS entry in its tile
  for (int t = 0; t < plan.n_tiles; ++t) {
    MvTile &T = plan.tiles[t];
    T.nslots = 1; T.slot_off = (int32_t)plan.slots.size(); T.n_entries = 0; T.n_shared = 0;
    plan.slots.push_back(MvSlot{-1, 0, 0, -1, -1});
    MvLane *L = plan.lanes.data() + (size_t)t * block;
    for (int l = 0; l < tile_fill[t]; ++l) {
      const int c = L[l].cidx;
      for (int side = 0; side < 2; ++side) {
        const int b = side ? body1[c] : body0[c];
        if (b < 0) continue;
        if (stamp[b] != t) {
          stamp[b] = t;
          slot_in_tile[b] = T.nslots++;
          MvSlot s;
          s.body = b; s.cnt = cnt[b]; s.t_off = t_off[b]; s.seg = T.n_entries;
          s.ents_off = shared[b] ? (int32_t)plan.ents.size() : -1;
          for (int k = 0; k < cnt[b]; ++k) {   // the body's sides, list order: those of this tile get LDS entries
            const int e = gent[(size_t)ent_off[b] + k];
            const bool here = cons_tile[e >> 1] == t;
            if (here) local_e[e] = T.n_entries++;
            if (shared[b]) plan.ents.push_back(here ? (uint16_t)local_e[e] : kMvRemote);
          }
          plan.slots.push_back(s);
          if (shared[b]) ++T.n_shared;
        }
      }
    }
    for (int l = 0; l < tile_fill[t]; ++l) {
      const int c = L[l].cidx;
      L[l].slot0 = body0[c] >= 0 ? (uint16_t)slot_in_tile[body0[c]] : 0;
      L[l].slot1 = body1[c] >= 0 ? (uint16_t)slot_in_tile[body1[c]] : 0;
      L[l].e0 = body0[c] >= 0 ? (uint16_t)local_e[2 * (size_t)c] : 0;
      L[l].e1 = body1[c] >= 0 ? (uint16_t)local_e[2 * (size_t)c + 1] : 0;
    }
    plan.max_slots = std::max(plan.max_slots, T.nslots);
  }
  return plan;
}

}  // namespace egs

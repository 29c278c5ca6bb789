// plan.cpp -- islands, tiles and per-body tickets (see plan.h).
#include "plan.h"

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <stdexcept>
#include <string>
#include <utility>

namespace egs {

namespace {
struct UnionFind {
  std::vector<int32_t> parent;
  explicit UnionFind(int n) : parent(n) { std::iota(parent.begin(), parent.end(), 0); }
  int find(int a) {
    while (parent[a] != a) {
      parent[a] = parent[parent[a]];
      a = parent[a];
    }
    return a;
  }
  void unite(int a, int b) {
    a = find(a); b = find(b);
    if (a != b) parent[std::max(a, b)] = std::min(a, b);
  }
};
}  // namespace

namespace {
// Cut the oversize islands into patches (see plan.h).  On any obstacle (a body
// owning more constraints than a tile holds, 16-bit overflow) the patch plan is
// left empty and the caller uses the all-global path.
int g_patch_workgroups = 256;     // CUs of the device the plans are built for (set_patch_workgroups)

void build_patches(Plan &plan, int n_bodies, const int32_t *body0, const int32_t *body1,
                   const std::vector<int32_t> &cnt, const std::vector<int32_t> &pos0,
                   const std::vector<int32_t> &pos1, bool allow_runs) {
  const int block = plan.block;
  const int mg = (int)plan.global.size();
  // head-room for the idle lanes that align the lane groups to wavefronts (below)
  const bool order_groups = [] { const char *e = std::getenv("EGS_PATCH_ORDER"); return !(e && std::atoi(e) == 0); }();
  // How many constraints per patch.  Measured on MI355X (tools/trace_patches.py): on the critical chain of a wall a
  // hand-off ACROSS patches costs 1.4 us, but one INSIDE a full 224-constraint patch 0.8 us instead of the 0.2-0.4 us of
  // an unloaded chain -- the patch's sixteen wavefronts share four SIMDs and every look of a waiting wavefront is ~100
  // VALU instructions for one or two ready constraints.  So the island is spread over as many CUs as there are
  // (g_patch_workgroups, all patches must be co-resident: one workgroup per CU), down to 48 constraints per patch.
  int cap = (order_groups && block >= 128) ? block - 32 : block;
  {
    const char *e = std::getenv("EGS_PATCH_CAP");
    const int forced = e ? std::atoi(e) : 0;
    if (forced >= 16 && forced <= block) cap = forced;
    else if (block >= 128) {
      const int target = (int)((mg + (long)(g_patch_workgroups * 9 / 10) - 1) / std::max(1, g_patch_workgroups * 9 / 10));
      // (with runs -- chunks of four as one node, below -- a patch's own work is cheaper against a crossing: walls of
      //  1 080 / 10 376 / 16 404 contacts are fastest at 128 / 80-96 / 96 slots per patch, measured)
      const char *re = std::getenv("EGS_PATCH_RUNS");
      const int floor_cap = (allow_runs && block == 256 && !(re && std::atoi(re) == 0)) ? 96 : 48;
      cap = std::min(cap, std::max(floor_cap, (target + 15) / 16 * 16));
    }
  }
  // owner body of a constraint = its first real body
  auto owner = [&](int c) { return body0[c] >= 0 ? body0[c] : body1[c]; };
  // Runs (plan.h): consecutive constraints on the same two bodies -- the contact points of a box face -- in chunks of at
  // most four.  A chunk takes four adjacent lane slots of the 4-lane patch kernel (one DPP row: members in list order,
  // then placeholders, cidx = -2, with the chunk's slots) and is ONE node of the ticket protocol: its quads wait for
  // the first member's ticket, the accumulators go from member to member through the row (row_shr:4) and the last
  // slot publishes ticket = last member's + 1.  Only for the 4-lane kernel, and while less than a quarter of the
  // slots would be placeholders.
  std::vector<int32_t> chunk_of(mg, 0), chunk_first, chunk_len;
  {
    const char *re = std::getenv("EGS_PATCH_RUNS"), *qp = std::getenv("EGS_QUAD_PATCH");
    plan.patch_runs = allow_runs && block == 256 && !(re && std::atoi(re) == 0) && !(qp && std::atoi(qp) == 0);
    if (plan.patch_runs) {
      for (int g = 0; g < mg; ++g) {
        const int c = plan.global[g].cidx, cp = g > 0 ? plan.global[g - 1].cidx : -1;
        const bool same = g > 0 && body0[c] == body0[cp] && body1[c] == body1[cp] && chunk_len.back() < 4 &&
                          (body0[c] < 0 || pos0[c] == pos0[cp] + 1) && (body1[c] < 0 || pos1[c] == pos1[cp] + 1);
        if (!same) { chunk_first.push_back(g); chunk_len.push_back(0); }
        chunk_of[g] = (int32_t)chunk_len.size() - 1;
        ++chunk_len.back();
      }
      if (4L * (long)chunk_len.size() > (5L * mg) / 4) { plan.patch_runs = false; chunk_first.clear(); chunk_len.clear(); }
    }
  }
  const bool runs = plan.patch_runs;
  const int n_chunks = (int)chunk_len.size();
  std::vector<int32_t> owned(n_bodies, 0);             // lane slots a body's constraints take
  std::vector<std::vector<int32_t>> touch(n_bodies);   // body -> global-list positions of its constraints
  std::vector<char> in_big(n_bodies, 0);
  for (int g = 0; g < mg; ++g) {
    const int c = plan.global[g].cidx;
    if (owner(c) < 0) return;                  // world-world constraint in an oversize island: cannot happen
    if (!runs) ++owned[owner(c)];
    else if (chunk_first[chunk_of[g]] == g) owned[owner(c)] += 4;
    for (int b : {body0[c], body1[c]})
      if (b >= 0) { touch[b].push_back(g); in_big[b] = 1; if (cnt[b] > 65535) return; }
  }
  for (int b = 0; b < n_bodies; ++b) if (owned[b] > block) { plan.patch_runs = false; return; }
  // Two ways to cut the island's bodies into patches; the one whose WORST body crosses patches least often per sweep
  // is taken (then the fewer crossings in total).  A body's constraint list is walked cyclically every sweep and every
  // change of patch along it is a hand-off through global memory (~4 us against 0.2-0.4 us in LDS), so the busiest
  // body's count of changes sets the sweep period of the whole island:
  //   blobs  -- BFS-grown from seeds in ascending body index: compact, most bodies never leave their patch, but a body
  //             where three or four blobs meet changes patch up to four times per sweep;
  //   chunks -- runs of consecutive body indices: the collider numbers bodies layer by layer, so a body's earlier
  //             neighbours (the owners of the constraints it does not own) sit in the chunk(s) just before its own and
  //             the list changes patch twice, three times where a chunk border cuts between them.
  // (EGS_PATCH_SHAPE=blobs / chunks forces one.)
  auto grow_blobs = [&](std::vector<int32_t> &patch_of, std::vector<int32_t> &patch_fill) {
    patch_of.assign(n_bodies, -1);
    patch_fill.clear();
    std::vector<int32_t> queue;
    for (int seed = 0; seed < n_bodies; ++seed) {
      if (!in_big[seed] || patch_of[seed] >= 0) continue;
      const int pid = (int)patch_fill.size();
      patch_fill.push_back(0);
      queue.clear();
      queue.push_back(seed);
      for (size_t qh = 0; qh < queue.size(); ++qh) {
        const int b = queue[qh];
        if (patch_of[b] >= 0) continue;
        if (patch_fill[pid] + owned[b] > (owned[b] > cap ? block : cap)) continue;   // does not fit: left for a later patch
        patch_of[b] = pid;
        patch_fill[pid] += owned[b];
        for (int g : touch[b]) {
          const int c = plan.global[g].cidx;
          for (int nb : {body0[c], body1[c]})
            if (nb >= 0 && patch_of[nb] < 0) queue.push_back(nb);
        }
        if (patch_fill[pid] >= cap) break;
      }
    }
  };
  auto cut_chunks = [&](std::vector<int32_t> &patch_of, std::vector<int32_t> &patch_fill) {
    patch_of.assign(n_bodies, -1);
    patch_fill.assign(1, 0);
    for (int b = 0; b < n_bodies; ++b) {
      if (!in_big[b]) continue;
      if (patch_fill.back() > 0 && patch_fill.back() + owned[b] > (owned[b] > cap ? block : cap)) patch_fill.push_back(0);
      patch_of[b] = (int)patch_fill.size() - 1;
      patch_fill.back() += owned[b];
    }
  };
  // (worst body's patch changes per sweep, all bodies' changes) of an assignment
  auto crossings = [&](const std::vector<int32_t> &patch_of) {
    int worst = 0;
    long total = 0;
    for (int b = 0; b < n_bodies; ++b) {
      const int k_n = (int)touch[b].size();
      if (k_n < 2) continue;
      int ch = 0;
      int prev = patch_of[owner(plan.global[touch[b][k_n - 1]].cidx)];
      for (int k = 0; k < k_n; ++k) {
        const int t = patch_of[owner(plan.global[touch[b][k]].cidx)];
        ch += t != prev;
        prev = t;
      }
      worst = std::max(worst, ch);
      total += ch;
    }
    return std::make_pair(worst, total);
  };
  std::vector<int32_t> patch_of, patch_fill;
  {
    const char *e = std::getenv("EGS_PATCH_SHAPE");
    const std::string shape = e ? e : "";
    if (shape == "chunks") cut_chunks(patch_of, patch_fill);
    else {
      grow_blobs(patch_of, patch_fill);
      if (shape != "blobs") {
        std::vector<int32_t> po2, pf2;
        cut_chunks(po2, pf2);
        if (crossings(po2) < crossings(patch_of)) { patch_of.swap(po2); patch_fill.swap(pf2); }
      }
    }
  }
  // drop empty patches (bodies that own nothing) by renumbering patches with constraints
  const int np_all = (int)patch_fill.size();
  std::vector<int32_t> renum(np_all, -1);
  int np = 0;
  for (int p = 0; p < np_all; ++p) if (patch_fill[p] > 0) renum[p] = np++;
  // tile of a constraint = patch of its owner; body is shared if touched from >1 tile
  std::vector<int32_t> first_tile(n_bodies, -1);
  std::vector<char> shared(n_bodies, 0);
  for (int g = 0; g < mg; ++g) {
    const int c = plan.global[g].cidx;
    const int t = renum[patch_of[owner(c)]];
    for (int b : {body0[c], body1[c]}) {
      if (b < 0) continue;
      if (first_tile[b] < 0) first_tile[b] = t;
      else if (first_tile[b] != t) shared[b] = 1;
    }
  }
  LaneDesc idle{};
  idle.cidx = -1;
  plan.n_patch_tiles = np;
  plan.patch_lanes.assign((size_t)np * block, idle);
  plan.patch_tile_nslots.assign(np, 1);
  plan.patch_tile_slot_off.assign(np, 0);
  // LDS slots: one per (patch, body it touches), private or shared.  A shared body's accumulator
  // travels with the sweep: it stays in the LDS of the patch that updated it last and goes through
  // global memory only when the NEXT constraint on the body (list order, cyclically) belongs to
  // another patch.  The two bits on top of a side's slot number say so: kPrevRemote = the
  // list-order predecessor on that body sits in another patch, kNextRemote = the successor does.
  std::vector<int32_t> fill(np, 0), slot_of(n_bodies, -1);
  std::vector<std::vector<std::pair<int32_t, int32_t>>> shared_slot(n_bodies);   // (patch, slot) of a shared body
  std::vector<std::vector<int32_t>> tile_bodies(np);
  std::vector<uint16_t> side_flags((size_t)mg * 2, 0);
  for (int b = 0; b < n_bodies; ++b) {
    if (!shared[b]) continue;
    const int k_n = (int)touch[b].size();
    for (int k = 0; k < k_n; ++k) {
      auto tile_of = [&](int kk) { return renum[patch_of[owner(plan.global[touch[b][kk]].cidx)]]; };
      const int t = tile_of(k), tp = tile_of((k + k_n - 1) % k_n), tn = tile_of((k + 1) % k_n);
      const int g = touch[b][k], c = plan.global[g].cidx;
      const int side = (body0[c] == b) ? 0 : 1;
      side_flags[(size_t)g * 2 + side] = (uint16_t)((tp != t ? kPrevRemote : 0) | (tn != t ? kNextRemote : 0));
    }
  }
  bool overflow = false;
  auto slot_number = [&](int b, int t) -> int {
    if (b < 0) return 0;
    int sl = -1;
    if (shared[b]) {
      for (auto &ps : shared_slot[b]) if (ps.first == t) sl = ps.second;
      if (sl < 0) { sl = plan.patch_tile_nslots[t]++; shared_slot[b].emplace_back(t, sl); tile_bodies[t].push_back(-(b + 2)); }
    } else {
      if (slot_of[b] < 0) { slot_of[b] = plan.patch_tile_nslots[t]++; tile_bodies[t].push_back(b); }
      sl = slot_of[b];
    }
    if (sl >= (int)kSlotMask) overflow = true;
    return sl;
  };
  // the lane group of an entry (see below): 0 = waits on another patch in a forward sweep, 1 = only in a backward one
  // (or runs the launch-boundary code of a shared body), 2 = interior
  std::vector<uint8_t> entry_group((size_t)np * block, 2);
  auto shared_end = [&](int b, unsigned pos, unsigned cn) { return b >= 0 && shared[b] && (pos == 0u || pos + 1u == cn); };
  if (!runs) {
    for (int g = 0; g < mg; ++g) {              // list order -> lanes ascending by list index
      const int c = plan.global[g].cidx;
      const int t = renum[patch_of[owner(c)]];
      LaneDesc d;
      d.cidx = c;
      d.slot0 = (uint16_t)(slot_number(body0[c], t) | side_flags[(size_t)g * 2 + 0]);
      d.slot1 = (uint16_t)(slot_number(body1[c], t) | side_flags[(size_t)g * 2 + 1]);
      d.pos0 = (uint16_t)pos0[c]; d.cnt0 = (uint16_t)(body0[c] >= 0 ? cnt[body0[c]] : 0);
      d.pos1 = (uint16_t)pos1[c]; d.cnt1 = (uint16_t)(body1[c] >= 0 ? cnt[body1[c]] : 0);
      const int f = d.slot0 | d.slot1;
      entry_group[(size_t)t * block + fill[t]] =
          (f & kPrevRemote) ? 0 : (f & kNextRemote) ? 1 : ((shared_end(body0[c], d.pos0, d.cnt0) || shared_end(body1[c], d.pos1, d.cnt1)) ? 1 : 2);
      plan.patch_lanes[(size_t)t * block + fill[t]++] = d;
    }
  } else {
    for (int q = 0; q < n_chunks; ++q) {        // list order, four slots per chunk
      const int g0 = chunk_first[q], len = chunk_len[q], c0 = plan.global[g0].cidx, cl = plan.global[g0 + len - 1].cidx;
      const int t = renum[patch_of[owner(c0)]];
      // the chunk's ends decide where its accumulators come from and go to
      const uint16_t f0 = (uint16_t)((side_flags[(size_t)g0 * 2 + 0] & kPrevRemote) | (side_flags[(size_t)(g0 + len - 1) * 2 + 0] & kNextRemote));
      const uint16_t f1 = (uint16_t)((side_flags[(size_t)g0 * 2 + 1] & kPrevRemote) | (side_flags[(size_t)(g0 + len - 1) * 2 + 1] & kNextRemote));
      const int s0 = slot_number(body0[c0], t), s1 = slot_number(body1[c0], t);
      const unsigned cn0 = body0[c0] >= 0 ? cnt[body0[c0]] : 0, cn1 = body1[c0] >= 0 ? cnt[body1[c0]] : 0;
      const int f = f0 | f1;
      const bool ends = (body0[c0] >= 0 && shared[body0[c0]] && (pos0[c0] == 0 || (unsigned)pos0[cl] + 1u == cn0)) ||
                        (body1[c0] >= 0 && shared[body1[c0]] && (pos1[c0] == 0 || (unsigned)pos1[cl] + 1u == cn1));
      const uint8_t grp = (f & kPrevRemote) ? 0 : (f & kNextRemote) ? 1 : (ends ? 1 : 2);
      for (int k = 0; k < 4; ++k) {
        LaneDesc d;
        d.cidx = k < len ? plan.global[g0 + k].cidx : -2;
        d.slot0 = (uint16_t)(s0 | f0); d.slot1 = (uint16_t)(s1 | f1);
        d.pos0 = (uint16_t)(pos0[c0] + k); d.cnt0 = (uint16_t)cn0;
        d.pos1 = (uint16_t)(pos1[c0] + k); d.cnt1 = (uint16_t)cn1;
        entry_group[(size_t)t * block + fill[t]] = grp;
        plan.patch_lanes[(size_t)t * block + fill[t]++] = d;
      }
    }
  }
  // Lanes that wait on ANOTHER patch (a side whose list-order neighbour is remote) poll global memory: a look costs a
  // memory round trip (~1-2 us) and stalls the whole wavefront, i.e. every constraint that shares it, including those
  // whose hand-offs are LDS-local (0.2-0.4 us).  So the patch's boundary constraints go to the front -- into as few
  // wavefronts as possible -- and the interior ones keep wavefronts of their own that never leave LDS.  The tickets
  // keep the list order whatever the lane order is (same bits; EGS_PATCH_ORDER=0 keeps list order for comparison).
  if (order_groups) {
    // (three groups: lanes that wait on another patch in a forward sweep -- a remote predecessor --, lanes that only do
    //  so in a backward sweep, interior lanes: the first group alone polls global memory under Gauss-Seidel, the second
    //  alone under backward SOR.  Each group starts on a wavefront of the 4-lane kernel (16 constraints) where the patch
    //  has room for the idle lanes in between: a lane that only RELEASES to another patch then never shares a wavefront
    //  with a polling one, so its own LDS hand-off is seen at once and not at the polling wavefront's next look.)
    // (the first and the last constraint of a SHARED body's list also run the cross-patch code -- the launch-boundary
    //  cases of the kernels -- although they never poll: they go with the second group)
    std::vector<LaneDesc> tmp;
    std::vector<int> order;
    for (int t = 0; t < np; ++t) {
      LaneDesc *base = plan.patch_lanes.data() + (size_t)t * block;
      const uint8_t *grp = entry_group.data() + (size_t)t * block;
      order.resize(fill[t]);
      for (int k = 0; k < fill[t]; ++k) order[k] = k;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return grp[a] < grp[b]; });     // (a chunk's four entries stay together)
      int n0 = 0, n1 = 0;
      for (int k = 0; k < fill[t]; ++k) { n0 += grp[k] == 0; n1 += grp[k] == 1; }
      const int unit = 16;
      const int s1 = (n0 + unit - 1) / unit * unit, s2 = s1 + (n1 + unit - 1) / unit * unit;
      const int total = s2 + (fill[t] - n0 - n1);
      if (total > block || (n0 == 0 && n1 == 0)) continue;
      tmp.resize(fill[t]);
      for (int k = 0; k < fill[t]; ++k) tmp[k] = base[order[k]];
      std::fill(base, base + block, idle);
      std::copy(tmp.begin(), tmp.begin() + n0, base);
      std::copy(tmp.begin() + n0, tmp.begin() + n0 + n1, base + s1);
      std::copy(tmp.begin() + n0 + n1, tmp.end(), base + s2);
      fill[t] = total;
    }
  }
  if (overflow) {   // cannot happen with <= 512 sides per patch; keep the all-global path if it ever does
    plan.n_patch_tiles = 0; plan.patch_lanes.clear(); plan.patch_tile_nslots.clear(); plan.patch_tile_slot_off.clear();
    return;
  }
  int off = 0;
  for (int t = 0; t < np; ++t) {
    plan.patch_tile_slot_off[t] = off;
    plan.patch_slot_body.push_back(-1);
    for (int b : tile_bodies[t]) plan.patch_slot_body.push_back(b);   // shared bodies as -(body + 2)
    off += plan.patch_tile_nslots[t];
    plan.patch_max_slots = std::max(plan.patch_max_slots, plan.patch_tile_nslots[t]);
  }
  for (int b = 0; b < n_bodies; ++b) plan.n_shared_bodies += shared[b];
}
}  // namespace

namespace {
struct PhaseClock {      // EGS_PLAN_PHASES=1: where build_plan's time goes (stderr)
  bool on = std::getenv("EGS_PLAN_PHASES") != nullptr;
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void mark(const char *what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "plan phase %-28s %8.1f us\n", what, std::chrono::duration<double, std::micro>(now - t).count());
    t = now;
  }
};
}  // namespace

Plan build_plan(int n_bodies, int m, const int32_t *body0, const int32_t *body1,
                int block, Plan *recycle, int max_run_tiles) {
  PhaseClock phase_clock;
  if (n_bodies < 0 || m < 0 || block < 0 || block > 1024)
    throw std::invalid_argument("build_plan: bad sizes");
  Plan plan;
  if (recycle) {   // keep the storage, drop the contents
    plan = std::move(*recycle);
    plan.n_islands = plan.n_tiles = 0; plan.max_slots = 1; plan.max_cnt = 1;
    plan.lanes.clear(); plan.lane_level.clear(); plan.tile_period.clear(); plan.tile_depth.clear();
    plan.max_period = plan.max_depth = 1; plan.levels_ok = true; plan.runs = false;
    plan.tile_nslots.clear(); plan.tile_slot_off.clear(); plan.slot_body.clear();
    plan.global.clear();
    plan.n_patch_tiles = 0; plan.patch_max_slots = 1; plan.n_shared_bodies = 0; plan.patch_runs = false;
    plan.patch_lanes.clear(); plan.patch_tile_nslots.clear(); plan.patch_tile_slot_off.clear(); plan.patch_slot_body.clear();
  }
  plan.n = n_bodies; plan.m = m; plan.block = block;
  for (int i = 0; i < m; ++i) {
    if (body0[i] < -1 || body0[i] >= n_bodies || body1[i] < -1 || body1[i] >= n_bodies)
      throw std::invalid_argument("build_plan: body index out of range");
  }

  // runs (plan.h): maximal stretches of consecutive constraints on the same two bodies, cut into chunks of at
  // most four.  A chunk takes four lane slots (members first, then placeholders that only pass the accumulators
  // along), so it is one DPP row of the 4-lane kernel; worth it while the padding stays below a quarter.
  std::vector<int32_t> chunk_of, chunk_first, chunk_len;
  {
    const char *env = std::getenv("EGS_RUNS");
    if (block == kAutoQuadBlock && max_run_tiles > 0 && !(env && std::atoi(env) == 0) && m > 0) {
      chunk_of.resize(m);
      for (int i = 0; i < m; ++i) {
        const bool same = i > 0 && body0[i] == body0[i - 1] && body1[i] == body1[i - 1] && chunk_len.back() < 4;
        if (!same) { chunk_first.push_back(i); chunk_len.push_back(0); }
        chunk_of[i] = (int32_t)chunk_len.size() - 1;
        ++chunk_len.back();
      }
      plan.runs = 4L * (long)chunk_len.size() <= (5L * m) / 4;
      if (plan.runs && !(env && std::atoi(env) == 2)) {
        // a time step of chunks runs four update passes whatever the chunks hold (0.22 us per pass against 0.29 us
        // per single-update step), so the busiest body's constraints must come in full chunks: with one short
        // chunk more (columns that touch sideways in a settling pile: 3 chunks for 10 constraints) it is break-even
        // at best (world step 0.91 -> 0.94 ms); EGS_RUNS=2 skips this test (experiments, tests)
        std::vector<int32_t> per_body_c(n_bodies, 0), per_body_n(n_bodies, 0);
        for (size_t c = 0; c < chunk_len.size(); ++c) {
          const int i = chunk_first[c];
          for (int b : {body0[i], body1[i] != body0[i] ? body1[i] : -1})
            if (b >= 0) { per_body_c[b] += chunk_len[c]; ++per_body_n[b]; }
        }
        int maxc = 1, maxn = 1;
        for (int b = 0; b < n_bodies; ++b) { maxc = std::max(maxc, per_body_c[b]); maxn = std::max(maxn, per_body_n[b]); }
        plan.runs = 4L * maxn <= (long)maxc;
      }
      if (!plan.runs) { chunk_of.clear(); chunk_first.clear(); chunk_len.clear(); }
    }
  }
  const int n_chunks = (int)chunk_len.size();

  phase_clock.mark("recycle + runs");
  // 1. islands: union bodies that share a constraint.
  UnionFind uf(n_bodies);
  for (int i = 0; i < m; ++i)
    if (body0[i] >= 0 && body1[i] >= 0) uf.unite(body0[i], body1[i]);

  // island id per constraint, numbered by first appearance in list order.
  // A constraint with both sides = world is an island of its own.
  std::vector<int32_t> root_island(n_bodies, -1), cons_island(m);
  std::vector<int32_t> island_size;
  for (int i = 0; i < m; ++i) {
    const int b = body0[i] >= 0 ? body0[i] : body1[i];
    int isl;
    if (b < 0) {
      isl = (int)island_size.size();
      island_size.push_back(0);
    } else {
      const int r = uf.find(b);
      if (root_island[r] < 0) {
        root_island[r] = (int)island_size.size();
        island_size.push_back(0);
      }
      isl = root_island[r];
    }
    cons_island[i] = isl;
    ++island_size[isl];
  }
  plan.n_islands = (int)island_size.size();
  const bool quad_plan = block == kAutoQuadBlock;
  if (plan.runs) {   // lane slots, not constraints: four per chunk
    std::vector<int32_t> padded(island_size.size(), 0);
    for (int c = 0; c < n_chunks; ++c) padded[cons_island[chunk_first[c]]] += 4;
    int largest = 0;
    long total = 0;
    for (int sz : padded) { largest = std::max(largest, sz); total += sz; }
    const int blk = largest <= 64 ? 64 : (largest <= 128 ? 128 : 256);
    int largest_plain = 0;
    for (int sz : island_size) largest_plain = std::max(largest_plain, sz);
    // runs pay while a CU holds about one tile (a chunk keeps a quarter of its wavefront's lanes busy); and the
    // padding must not push an island that fits a workgroup onto the cross-workgroup path
    if ((total + blk - 1) / blk <= (long)max_run_tiles && !(largest > 256 && largest_plain <= 256)) island_size.swap(padded);
    else { plan.runs = false; chunk_of.clear(); chunk_first.clear(); chunk_len.clear(); }
  }
  if (quad_plan) {
    int largest = 0;
    for (int sz : island_size) largest = std::max(largest, sz);
    block = largest <= 64 ? 64 : (largest <= 128 ? 128 : 256);
    plan.block = block;
  }

  phase_clock.mark("islands");
  // 2. per-body rank (pos) and count (cnt) in list order.
  std::vector<int32_t> cnt(n_bodies, 0), pos0(m, 0), pos1(m, 0);
  for (int i = 0; i < m; ++i) {
    if (body0[i] >= 0) pos0[i] = cnt[body0[i]]++;
    if (body1[i] >= 0 && body1[i] != body0[i]) pos1[i] = cnt[body1[i]]++;
    else if (body1[i] >= 0) pos1[i] = pos0[i];  // same body on both sides
  }

  for (int b = 0; b < n_bodies; ++b) plan.max_cnt = std::max(plan.max_cnt, cnt[b]);

  // 3. pack whole islands into tiles, first-fit in island order; islands that
  //    exceed a tile (or whose per-body count overflows 16 bits) go global.
  std::vector<int32_t> island_tile(plan.n_islands, -1);
  std::vector<int32_t> tile_fill;
  {
    int cur = -1;
    for (int isl = 0; isl < plan.n_islands; ++isl) {
      if (island_size[isl] > block) { island_tile[isl] = -2; continue; }
      if (cur < 0 || tile_fill[cur] + island_size[isl] > block) {
        cur = (int)tile_fill.size();
        tile_fill.push_back(0);
      }
      island_tile[isl] = cur;
      tile_fill[cur] += island_size[isl];
    }
  }
  plan.n_tiles = (int)tile_fill.size();

  phase_clock.mark("rank + tiles");
  // lane order inside a tile.  In the pipelined sweep constraint i runs at time
  // ~ level(i) + D * sweep, level = depth in the list-order dependency DAG and
  // D = the largest per-body constraint count of its island (the ticket period).
  // Constraints with equal (level mod D) are therefore ready TOGETHER; putting
  // them in the same wavefront turns 1/D lane utilisation into nearly full
  // wavefronts that take turns (EGS_LANE_ORDER=0 restores island-major order).
  std::vector<int32_t> phase(m, 0);
  int max_phase = 0;
  // levels of the list-order dependency DAG (plan.h) and every body's span of levels
  std::vector<int32_t> level(m, 0), first_lvl(n_bodies, -1), last_lvl(n_bodies, -1);
  // nodes of the level DAG: constraints, or chunks of a run (plan.h)
  std::vector<int32_t> node_cnt;      // runs: chunks per body (the ticket period counts constraints, this one nodes)
  {
    std::vector<int32_t> nxt(n_bodies, 0);
    if (plan.runs) node_cnt.assign(n_bodies, 0);
    const int nodes = plan.runs ? n_chunks : m;
    for (int q = 0; q < nodes; ++q) {
      const int i = plan.runs ? chunk_first[q] : q, len = plan.runs ? chunk_len[q] : 1;
      const int b0 = body0[i], b1 = body1[i];
      int lv = 0;
      if (b0 >= 0) lv = nxt[b0];
      if (b1 >= 0) lv = std::max(lv, nxt[b1]);
      for (int k = 0; k < len; ++k) level[i + k] = lv;
      for (int b : {b0, b1})
        if (b >= 0) { nxt[b] = lv + 1; if (first_lvl[b] < 0) first_lvl[b] = lv; last_lvl[b] = lv; }
      if (plan.runs) {
        if (b0 >= 0) ++node_cnt[b0];
        if (b1 >= 0 && b1 != b0) ++node_cnt[b1];
      }
    }
  }
  {
    const char *env = std::getenv("EGS_LANE_ORDER");
    const bool by_phase = !(env && std::atoi(env) == 0);
    if (by_phase) {
      // period of an island = the largest per-body count in it: one pass over the bodies
      // (a body's island is the island of the first constraint that touched it)
      std::vector<int32_t> isl_period(plan.n_islands, 1), body_island(n_bodies, -1);
      for (int i = 0; i < m; ++i) {
        if (body0[i] >= 0 && body_island[body0[i]] < 0) body_island[body0[i]] = cons_island[i];
        if (body1[i] >= 0 && body_island[body1[i]] < 0) body_island[body1[i]] = cons_island[i];
      }
      // (with runs: in chunks)
      for (int b = 0; b < n_bodies; ++b)
        if (body_island[b] >= 0) isl_period[body_island[b]] = std::max(isl_period[body_island[b]], plan.runs ? node_cnt[b] : cnt[b]);
      for (int i = 0; i < m; ++i) {
        const int ph = level[i] % isl_period[cons_island[i]];     // the levels computed above
        phase[i] = ph;
        max_phase = std::max(max_phase, ph);
      }
    }
  }
  phase_clock.mark("levels + phases");
  // order by (phase, island, list index): stable counting sort -- one pass on the combined
  // key when that needs few buckets, else two passes (LSD radix)
  std::vector<int32_t> order(m);
  {
    std::vector<int32_t> head;
    const long combined = (long)(max_phase + 1) * plan.n_islands;
    if (combined <= 4L * m + 1024) {
      head.assign((size_t)combined + 1, 0);
      for (int i = 0; i < m; ++i) ++head[(size_t)phase[i] * plan.n_islands + cons_island[i] + 1];
      for (long k = 0; k < combined; ++k) head[k + 1] += head[k];
      for (int i = 0; i < m; ++i) order[head[(size_t)phase[i] * plan.n_islands + cons_island[i]]++] = i;
    } else {
      std::vector<int32_t> tmp(m);
      auto counting_pass = [&](const std::vector<int32_t> &key, int n_keys, const int32_t *src, int32_t *dst) {
        head.assign((size_t)n_keys + 1, 0);
        for (int k = 0; k < m; ++k) ++head[key[src[k]] + 1];
        for (int k = 0; k < n_keys; ++k) head[k + 1] += head[k];
        for (int k = 0; k < m; ++k) dst[head[key[src[k]]]++] = src[k];
      };
      std::iota(order.begin(), order.end(), 0);
      counting_pass(cons_island, plan.n_islands, order.data(), tmp.data());
      counting_pass(phase, max_phase + 1, tmp.data(), order.data());
    }
  }

  phase_clock.mark("order");
  LaneDesc idle{};
  idle.cidx = -1;
  plan.lanes.assign((size_t)plan.n_tiles * block, idle);
  plan.lane_level.assign((size_t)plan.n_tiles * block, 0);
  plan.tile_period.assign(plan.n_tiles, 1);
  plan.tile_depth.assign(plan.n_tiles, 1);
  plan.tile_nslots.assign(plan.n_tiles, 1);
  plan.tile_slot_off.assign(plan.n_tiles, 0);
  phase_clock.mark("  assigns");
  std::vector<int32_t> lane_fill(plan.n_tiles, 0);
  std::vector<int32_t> body_slot(n_bodies, -1), body_tile(n_bodies, -1);
  std::vector<int32_t> lane_src;      // runs: the constraint whose bodies a lane slot (member or placeholder) belongs to
  if (plan.runs) lane_src.assign((size_t)plan.n_tiles * block, -1);

  // (bank-aware slot numbering, below) body -> where its uses sit inside its tile, CSR over the
  // bodies in rank order; a use = (half-wave << 2 | b128 pass of the half-wave << 1 | side)
  const char *slot_env = std::getenv("EGS_SLOT_BANKS");
  const bool colour = !(slot_env && std::atoi(slot_env) == 0) && !quad_plan;
  std::vector<int32_t> use_off, use_lane;
  if (colour) {
    use_off.resize((size_t)n_bodies + 1);
    use_off[0] = 0;
    for (int b = 0; b < n_bodies; ++b) use_off[b + 1] = use_off[b] + cnt[b];
    use_lane.resize((size_t)use_off[n_bodies] + 1);
  }
  for (int k = 0; k < m; ++k) {
    const int i = order[k];
    const int tile = island_tile[cons_island[i]];
    if (tile == -2) continue;
    const int b0 = body0[i], b1 = body1[i], l = lane_fill[tile]++;
    LaneDesc d;
    d.cidx = i;
    d.slot0 = d.slot1 = 0;
    d.pos0 = (uint16_t)pos0[i];
    d.cnt0 = (uint16_t)(b0 >= 0 ? cnt[b0] : 0);
    d.pos1 = (uint16_t)pos1[i];
    d.cnt1 = (uint16_t)(b1 >= 0 ? cnt[b1] : 0);
    plan.lanes[(size_t)tile * block + l] = d;
    plan.lane_level[(size_t)tile * block + l] = (uint16_t)std::min(level[i], 65535);
    if (level[i] > 65535) plan.levels_ok = false;
    plan.tile_depth[tile] = std::max(plan.tile_depth[tile], level[i] + 1);
    for (int b : {b0, b1})
      if (b >= 0) plan.tile_period[tile] = std::max(plan.tile_period[tile], last_lvl[b] - first_lvl[b] + 1);
    if (colour) {
      const int li = l & 31;
      const int where = (l >> 5) << 2 | ((li < 4 || (li >= 12 && li < 16) || (li >= 20 && li < 28)) ? 0 : 2);
      if (b0 >= 0) use_lane[use_off[b0] + pos0[i]] = where;
      if (b1 >= 0 && b1 != b0) use_lane[use_off[b1] + pos1[i]] = where | 1;
    }
    if (plan.runs) {
      lane_src[(size_t)tile * block + l] = i;
      const int c = chunk_of[i];
      if (i == chunk_first[c] + chunk_len[c] - 1)      // the chunk's last member: fill its row of four
        for (int pad = chunk_len[c]; pad < 4; ++pad) {
          const int lp = lane_fill[tile]++;
          LaneDesc ph{};
          ph.cidx = -2;                                  // placeholder: in the chain, no constraint
          plan.lanes[(size_t)tile * block + lp] = ph;
          plan.lane_level[(size_t)tile * block + lp] = plan.lane_level[(size_t)tile * block + l];
          lane_src[(size_t)tile * block + lp] = i;
        }
    }
  }
  phase_clock.mark("  lanes");
  // LDS slots.  A slot's number decides its banks: the ticket word s_tick[slot] is polled with
  // ds_read_b32 (bank = slot mod 32, the 32 lanes of a half-wave share a pass) and the 48-byte
  // accumulator is read as three ds_read_b128 (16-byte piece 3*slot+k mod 16, passes of 16 lanes:
  // {0-3,12-15,20-27} and {4-11,16-19,28-31} of each half-wave; 3 is invertible mod 16, so pieces
  // collide exactly when slots are equal mod 16).  Two lanes of a pass that hit one bank with
  // DIFFERENT slots cost an extra LDS cycle each, inside the poll loop the sweep's critical path
  // runs through.  So a tile's bodies, in first-use order, each take the free slot whose residue
  // mod 32 is used least by other bodies in the passes its lanes sit in (side 0 and side 1 are
  // separate instructions, hence separate masks).  Quad plans (4 lanes per constraint, nearly no
  // conflicts measured) keep first-use numbering.  (Rocprof, C3 x 24: 35 % of the tile kernel's
  // LDS cycles were bank conflicts with first-use numbering.)
  {
    std::vector<uint32_t> half_mask, pass_mask;     // [pass][side]: residues in use
    std::vector<int32_t> next_free(32);
    for (int t = 0; t < plan.n_tiles; ++t) {
      LaneDesc *L = plan.lanes.data() + (size_t)t * block;
      const int nl = lane_fill[t];
      if (!colour) {
        for (int l = 0; l < nl; ++l)
          for (int side = 0; side < 2; ++side) {
            const int src = plan.runs ? lane_src[(size_t)t * block + l] : L[l].cidx;
            const int body = side ? body1[src] : body0[src];
            if (body >= 0 && body_slot[body] < 0) { body_slot[body] = plan.tile_nslots[t]++; body_tile[body] = t; }
          }
      } else {
        half_mask.assign((size_t)(block / 32 + 1) * 2, 0u);
        pass_mask.assign((size_t)(block / 32 + 1) * 4, 0u);
        for (int r = 0; r < 32; ++r) next_free[r] = r ? r : 32;   // slot 0 = the world
        int top = 0, placed = 0, rr = 1;
        auto lowest = [&](uint32_t mset) {
          int best = -1;
          for (; mset; mset &= mset - 1) {
            const int r = __builtin_ctz(mset);
            if (best < 0 || next_free[r] < next_free[best]) best = r;
          }
          return best;
        };
        auto place = [&](int body) {
          const int32_t *u0 = use_lane.data() + use_off[body], *u1 = u0 + cnt[body];
          uint32_t used32 = 0, used16 = 0;
          for (const int32_t *u = u0; u < u1; ++u) {
            used32 |= half_mask[(size_t)(*u >> 2) * 2 + (*u & 1)];
            used16 |= pass_mask[*u];
          }
          used16 |= used16 << 16;
          // free residues, best first: clean for both, clean for the accumulator, clean for the ticket
          uint32_t pick = ~(used32 | used16);
          if (!pick) pick = ~used16;
          if (!pick) pick = ~used32;
          if (!pick) pick = ~0u;
          // among them a low free slot keeps the numbering dense: never more than 64 numbers
          // beyond a dense numbering (LDS is sized by the top slot)
          // (taken round-robin from the residue after the last one; the exact lowest only when
          // that drifts too far)
          const uint32_t rot = (pick >> rr) | (pick << ((32 - rr) & 31));
          int best = (__builtin_ctz(rot) + rr) & 31;
          if (next_free[best] > placed + 64) best = lowest(~0u);
          rr = (best + 1) & 31;
          ++placed;
          const int slot = next_free[best];
          next_free[best] += 32;
          top = std::max(top, slot);
          body_slot[body] = slot; body_tile[body] = t;
          for (const int32_t *u = u0; u < u1; ++u) {
            half_mask[(size_t)(*u >> 2) * 2 + (*u & 1)] |= 1u << best;
            pass_mask[*u] |= 1u << (best & 15);
          }
        };
        for (int l = 0; l < nl; ++l) {
          const int c = L[l].cidx;
          if (body0[c] >= 0 && body_slot[body0[c]] < 0) place(body0[c]);
          if (body1[c] >= 0 && body_slot[body1[c]] < 0) place(body1[c]);
        }
        plan.tile_nslots[t] = top + 1;
      }
      for (int l = 0; l < nl; ++l) {
        const int c = plan.runs ? lane_src[(size_t)t * block + l] : L[l].cidx;
        L[l].slot0 = body0[c] >= 0 ? (uint16_t)body_slot[body0[c]] : 0;
        L[l].slot1 = body1[c] >= 0 ? (uint16_t)body_slot[body1[c]] : 0;
      }
    }
  }
  phase_clock.mark("  slots");
  if (const char *e = std::getenv("EGS_PLAN_STATS"); e && std::atoi(e) != 0) {
    // modelled extra LDS cycles with every lane active: distinct slots on one bank within a pass
    long tick_extra = 0, acc_extra = 0, passes = 0;
    for (int t = 0; t < plan.n_tiles; ++t) {
      const LaneDesc *L = plan.lanes.data() + (size_t)t * block;
      for (int side = 0; side < 2; ++side)
        for (int h = 0; h < block / 32; ++h) {
          std::vector<int> seen32[32], seen16[2][16];
          for (int li = 0; li < 32; ++li) {
            const LaneDesc &d = L[h * 32 + li];
            if (d.cidx < 0) continue;
            const int body = side ? body1[d.cidx] : body0[d.cidx];
            if (body < 0) continue;
            const int slot = side ? d.slot1 : d.slot0;
            const int pass = (li < 4 || (li >= 12 && li < 16) || (li >= 20 && li < 28)) ? 0 : 1;
            auto add = [&](std::vector<int> &v) { if (std::find(v.begin(), v.end(), slot) == v.end()) v.push_back(slot); };
            add(seen32[slot & 31]); add(seen16[pass][slot & 15]);
          }
          int w32 = 1, w16[2] = {1, 1};
          for (int r = 0; r < 32; ++r) w32 = std::max(w32, (int)seen32[r].size());
          for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) w16[q] = std::max(w16[q], (int)seen16[q][r].size());
          tick_extra += w32 - 1; acc_extra += (w16[0] - 1) + (w16[1] - 1); ++passes;
        }
    }
    std::fprintf(stderr, "[egs plan] tiles %d block %d max_slots(pre) half-wave passes %ld: ticket extra cycles %ld, accumulator extra %ld (x3 pieces)\n",
                 plan.n_tiles, block, passes, tick_extra, acc_extra);
  }
  int off = 0;
  for (int t = 0; t < plan.n_tiles; ++t) {
    plan.tile_slot_off[t] = off;
    off += plan.tile_nslots[t];
    plan.max_slots = std::max(plan.max_slots, plan.tile_nslots[t]);
    plan.max_period = std::max(plan.max_period, plan.tile_period[t]);
    plan.max_depth = std::max(plan.max_depth, plan.tile_depth[t]);
  }
  plan.slot_body.assign((size_t)off, -1);   // slot 0 of every tile = the world
  for (int b = 0; b < n_bodies; ++b)
    if (body_tile[b] >= 0) plan.slot_body[(size_t)plan.tile_slot_off[body_tile[b]] + body_slot[b]] = b;

  phase_clock.mark("lanes + slots");
  // 4. oversize islands, in list order.
  for (int i = 0; i < m; ++i) {
    if (island_tile[cons_island[i]] != -2) continue;
    GlobalDesc g{};
    g.cidx = i;
    g.body0 = body0[i]; g.body1 = body1[i];
    g.pos0 = pos0[i]; g.cnt0 = body0[i] >= 0 ? cnt[body0[i]] : 0;
    g.pos1 = pos1[i]; g.cnt1 = body1[i] >= 0 ? cnt[body1[i]] : 0;
    plan.global.push_back(g);
  }
  phase_clock.mark("global list");
  if (!plan.global.empty()) {
    build_patches(plan, n_bodies, body0, body1, cnt, pos0, pos1, true);
    if (plan.patch_runs && plan.n_patch_tiles > g_patch_workgroups) {
      // more patches than the 4-lane kernel can keep resident: the 1-lane patch kernel will run them, which knows no runs
      plan.n_patch_tiles = 0; plan.patch_max_slots = 1; plan.n_shared_bodies = 0; plan.patch_runs = false;
      plan.patch_lanes.clear(); plan.patch_tile_nslots.clear(); plan.patch_tile_slot_off.clear(); plan.patch_slot_body.clear();
      build_patches(plan, n_bodies, body0, body1, cnt, pos0, pos1, false);
    }
    if (plan.n_patch_tiles == 0) plan.patch_runs = false;
  }
  phase_clock.mark("patches");
  return plan;
}

void set_patch_workgroups(int n) { if (n > 0) g_patch_workgroups = n; }

}  // namespace egs

// ba_order.hip - the INTERNAL ORDER OF THE OPTIMISED CAMERAS (host code; no kernel).
//
// The reference builds a dense S[nco, nco, 6, 6] and solves it whatever the co-visibility pattern is
// (bundle_adjuster.py:259-312).  Here S is a block band whose half-width is the widest spread of optimised-camera positions
// inside one track - so the same scene handed over with its cameras in another order (an unordered image collection, ids
// assigned by a database) would turn a 0.25 ms banded trial into a dense one.  ba_set_problem therefore chooses its own order
// of the optimised cameras when the caller's is not provably as narrow as it can be: Cuthill-McKee on the co-visibility
// hypergraph (cameras = nodes, tracks = hyper-edges), which needs no camera-camera adjacency - a breadth-first walk expands
// every track once, O(observations of the DISTINCT camera lists).  Everything the C ABI hands over or returns by optimised
// position (S, b, dC, masks, motion updates) goes through the permutation at the boundary (ba_handle::cpos_in / cpos_out).
#include "ba_internal.h"

namespace ba {

namespace {

struct Hyper {
  int n = 0;                         // cameras (optimised positions)
  std::vector<int> loff, lpos;       // distinct camera lists: positions lpos[loff[l] .. loff[l + 1])
  std::vector<int> coff, clist;      // camera -> the lists it is in
  std::vector<long long> deg;        // sum over its lists of (length - 1): the degree with multiplicities
};

// breadth-first from `start` over cameras not yet placed (placed[c] != 0): the visiting order (neighbours by ascending degree,
// then by position: deterministic), the level of the last camera, and the cameras of the last level
struct Walker {
  const Hyper& g;
  std::vector<int> cstamp, lstamp, level;
  int stamp = 0;
  explicit Walker(const Hyper& g_) : g(g_), cstamp(g_.n, 0), lstamp(g_.loff.size(), 0), level(g_.n, 0) {}
  void walk(int start, const std::vector<char>& placed, std::vector<int>& order) {
    ++stamp;
    order.clear();
    order.push_back(start);
    cstamp[start] = stamp; level[start] = 0;
    std::vector<int> fresh;
    for (size_t q = 0; q < order.size(); ++q) {
      const int u = order[q];
      fresh.clear();
      for (int e = g.coff[u]; e < g.coff[u + 1]; ++e) {
        const int l = g.clist[e];
        if (lstamp[l] == stamp) continue;          // every list is expanded once: by the first of its cameras the walk reaches
        lstamp[l] = stamp;
        for (int k = g.loff[l]; k < g.loff[l + 1]; ++k) {
          const int d = g.lpos[k];
          if (cstamp[d] == stamp || placed[d]) continue;
          cstamp[d] = stamp; level[d] = level[u] + 1;
          fresh.push_back(d);
        }
      }
      std::sort(fresh.begin(), fresh.end(), [&](int a, int b) { return g.deg[a] != g.deg[b] ? g.deg[a] < g.deg[b] : a < b; });
      order.insert(order.end(), fresh.begin(), fresh.end());
    }
  }
};

}  // namespace

// Half-bandwidth the lists would have under `newpos` (caller position -> new position).
int order_half_bandwidth(const std::vector<int>& loff, const std::vector<int>& lpos, const std::vector<int>& newpos) {
  int hb = 0;
  for (size_t l = 0; l + 1 < loff.size(); ++l) {
    int lo = INT32_MAX, hi = -1;
    for (int k = loff[l]; k < loff[l + 1]; ++k) { const int p = newpos[lpos[k]]; lo = std::min(lo, p); hi = std::max(hi, p); }
    if (hi >= 0) hb = std::max(hb, hi - lo);
  }
  return hb;
}

// Cuthill-McKee order of nco cameras from the distinct camera lists (positions in the caller's order).  newpos[p] = the new
// position of the caller's position p.  Start of every component: a pseudo-peripheral camera (George & Liu: walk, take the
// lowest-degree camera of the last level, repeat while the depth grows).  Cameras no track sees keep their relative order
// at the end.
void cuthill_mckee_order(int nco, const std::vector<int>& loff, const std::vector<int>& lpos, std::vector<int>& newpos) {
  Hyper g;
  g.n = nco; g.loff = loff; g.lpos = lpos;
  const int nl = (int)loff.size() - 1;
  g.coff.assign((size_t)nco + 1, 0);
  g.deg.assign((size_t)nco, 0);
  for (int l = 0; l < nl; ++l)
    for (int k = loff[l]; k < loff[l + 1]; ++k) { ++g.coff[(size_t)lpos[k] + 1]; g.deg[lpos[k]] += loff[l + 1] - loff[l] - 1; }
  for (int c = 0; c < nco; ++c) g.coff[(size_t)c + 1] += g.coff[c];
  g.clist.resize(lpos.size());
  {
    std::vector<int> fill(g.coff.begin(), g.coff.end() - 1);
    for (int l = 0; l < nl; ++l)
      for (int k = loff[l]; k < loff[l + 1]; ++k) g.clist[fill[lpos[k]]++] = l;
  }
  std::vector<int> by_deg((size_t)nco);
  for (int c = 0; c < nco; ++c) by_deg[c] = c;
  std::sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return g.deg[a] != g.deg[b] ? g.deg[a] < g.deg[b] : a < b; });
  std::vector<char> placed((size_t)nco, 0);
  std::vector<int> order, full;
  full.reserve((size_t)nco);
  Walker w(g);
  for (int c0 : by_deg) {
    if (placed[c0] || g.coff[c0] == g.coff[(size_t)c0 + 1]) continue;
    int start = c0, depth = -1;
    for (int it = 0; it < 8; ++it) {
      w.walk(start, placed, order);
      const int d = w.level[order.back()];
      if (d <= depth) break;
      depth = d;
      int best = order.back();
      for (size_t q = order.size(); q-- > 0 && w.level[order[q]] == d;)
        if (g.deg[order[q]] < g.deg[best] || (g.deg[order[q]] == g.deg[best] && order[q] < best)) best = order[q];
      if (best == start) break;
      start = best;
    }
    w.walk(start, placed, order);
    for (int c : order) placed[c] = 1;
    full.insert(full.end(), order.begin(), order.end());
  }
  for (int c = 0; c < nco; ++c)
    if (!placed[c]) full.push_back(c);
  newpos.assign((size_t)nco, 0);
  for (int q = 0; q < nco; ++q) newpos[full[q]] = q;
}

// Which cameras go to a border (ba_border.h) so that what is left of every list spreads over at most t positions.  Lists by
// descending spread; of a list that is still too wide, the largest set of cameras inside one window of t + 1 positions stays,
// the others go (a loop closure: the far end of the track; a ring: one side of the seam).  -1: more than kmax cameras.
int choose_border(int nco, const std::vector<int>& loff, const std::vector<int>& lpos, int t, int kmax, std::vector<char>& is_border) {
  is_border.assign((size_t)nco, 0);
  const int nl = (int)loff.size() - 1;
  std::vector<std::pair<int, int>> wide;               // (spread, list) of the lists wider than t
  for (int l = 0; l < nl; ++l) {
    int lo = INT32_MAX, hi = -1;
    for (int k = loff[l]; k < loff[l + 1]; ++k) { lo = std::min(lo, lpos[k]); hi = std::max(hi, lpos[k]); }
    if (hi - lo > t) wide.push_back({hi - lo, l});
  }
  std::sort(wide.begin(), wide.end(), [](const std::pair<int, int>& a, const std::pair<int, int>& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
  int k = 0;
  std::vector<int> v;
  for (const auto& w : wide) {
    const int l = w.second;
    v.clear();
    for (int q = loff[l]; q < loff[l + 1]; ++q)
      if (!is_border[lpos[q]]) v.push_back(lpos[q]);
    std::sort(v.begin(), v.end());
    if (v.empty() || v.back() - v.front() <= t) continue;
    size_t best_a = 0, best_n = 0;
    for (size_t a = 0, b = 0; a < v.size(); ++a) {      // the window [v[a], v[a] + t] that keeps the most cameras
      while (b < v.size() && v[b] - v[a] <= t) ++b;
      if (b - a > best_n) { best_n = b - a; best_a = a; }
    }
    for (size_t q = 0; q < v.size(); ++q)
      if (q < best_a || q >= best_a + best_n) {
        is_border[v[q]] = 1;
        if (++k > kmax) return -1;
      }
  }
  return k;
}

// What a solve of the reduced system costs, in microseconds, by the shape the layout gives it (measured on config-3-sized scenes:
// profiles/r04b_sweep.json, r04e_kernel_choice_probe.txt, r05a_*) - only good enough to rank the candidates of plan_camera_layout.
double layout_cost_us(int hb, int nco, int border_cams) {
  const double scale = std::max(1.0, nco / 1000.0);
  double t;
  if (hb <= kBcrSplitMaxHB) t = 100.0 * std::pow(std::max(hb, 3) / 9.0, 1.5) * std::max(1.0, std::log2(std::max(2.0, (double)nco / std::max(1, hb))) / 7.0);
  else if (hb <= kBcrwMaxHB) t = (250.0 + 45.0 * (hb - kBcrMaxHB)) * scale;      // (12, 13: 151, 193 us through the one-launch kernel; 14: 290)
  else t = (800.0 + 14.0 * (hb - kBcrwMaxHB)) * scale;
  if (border_cams > 0) t += 50.0 + 12.0 * border_cams * scale;
  return t;
}

// The layout of nco optimised cameras (positions in the caller's order) from the distinct camera lists and the number of points
// that have each: candidate ORDERS - the caller's, Cuthill-McKee on all lists, Cuthill-McKee on the lists that are not "weak ties"
// (a few lists that a single point has: loop closures in a scene whose cameras also come in no particular order - with them the
// breadth-first walk folds the loop into the band) - each as it is and, when its band is wider than the narrow cyclic
// reduction takes, with a border (choose_border) for the half-widths 11 down to where it stops being feasible.  Ranked by
// layout_cost_us.  Returns false when the caller's order without a border wins.  newpos[p] = final position of the caller's
// position p: the band cameras first (n1 of them) in the winning order, the border cameras behind them.
bool plan_camera_layout(int nco, const std::vector<int>& loff, const std::vector<int>& lpos, const std::vector<int>& lmult, bool allow_border,
                        std::vector<int>& newpos, int* n1_out, int* hb_out) {
  struct Cand { std::vector<int> pos; int hb; int k; std::vector<char> isb; double cost; };
  std::vector<Cand> cands;
  std::vector<int> ident((size_t)nco);
  for (int p = 0; p < nco; ++p) ident[p] = p;
  auto add = [&](const std::vector<int>& pos) {
    std::vector<int> mapped(lpos.size());
    for (size_t q = 0; q < lpos.size(); ++q) mapped[q] = pos[lpos[q]];
    const int hb = order_half_bandwidth(loff, mapped, ident);
    cands.push_back({pos, hb, 0, {}, layout_cost_us(hb, nco, 0)});
    // ... with a border: only where it brings the band down to what the narrow cyclic reduction takes
    // (a band the one-launch cyclic reduction takes: up to kBcrSplitMaxHB cameras wide - its kept factors are what k_bcr_apply reads)
    if (hb > kBcrMaxHB && allow_border && nco >= 4 * kBcrMaxHB) {
      for (int t = std::min(hb - 1, kBcrSplitMaxHB); t >= 1; --t) {
        std::vector<char> isb;                          // by position in this candidate's order
        const int k = choose_border(nco, loff, mapped, t, kBordMaxCamsHost, isb);
        if (k <= 0) break;                              // (narrower only ever needs more border cameras)
        cands.push_back({pos, t, k, isb, layout_cost_us(t, nco - k, k)});
      }
    }
  };
  add(ident);
  std::vector<int> cm;
  cuthill_mckee_order(nco, loff, lpos, cm);
  add(cm);
  // without the weak ties: lists of a single point, when there are few of them (at most 64 and 5 % of the lists)
  const int nl = (int)loff.size() - 1;
  if (allow_border && (int)lmult.size() == nl && cands.back().hb > kBcrMaxHB) {
    int weak = 0;
    for (int l = 0; l < nl; ++l) weak += lmult[l] <= 1 ? 1 : 0;
    if (weak > 0 && weak <= 64 && 20 * weak <= nl) {
      std::vector<int> soff(1, 0), spos;
      for (int l = 0; l < nl; ++l) {
        if (lmult[l] <= 1) continue;
        spos.insert(spos.end(), lpos.begin() + loff[l], lpos.begin() + loff[l + 1]);
        soff.push_back((int)spos.size());
      }
      std::vector<int> cm2;
      cuthill_mckee_order(nco, soff, spos, cm2);
      add(cm2);
    }
  }
  size_t best = 0;
  for (size_t c = 1; c < cands.size(); ++c)
    if (cands[c].cost < cands[best].cost * (1.0 - 1e-9)) best = c;
  if (best == 0) return false;
  const Cand& B = cands[best];
  // a band that stays far beyond what any band solver takes and comes out less than a quarter narrower than the caller's (an
  // unordered photo collection: no order has a band; its solver - conjugate gradients over the blocks the tracks define, ba_pcg.h -
  // does not care about the order): not worth setting the problem up a second time
  if (B.k == 0 && B.hb > kBcrwMaxHB && 4ll * B.hb > 3ll * cands[0].hb) return false;
  std::vector<int> inv((size_t)nco);
  for (int p = 0; p < nco; ++p) inv[B.pos[p]] = p;        // position in the candidate's order -> caller's position
  newpos.assign((size_t)nco, 0);
  int nb_ = 0, nborder = 0;
  const int n1 = nco - B.k;
  for (int q = 0; q < nco; ++q) {
    const bool isb = B.k > 0 && B.isb[q];
    newpos[inv[q]] = isb ? n1 + nborder++ : nb_++;
  }
  *n1_out = n1;
  *hb_out = B.hb;
  return true;
}

}  // namespace ba

extern "C" int ba_plan_camera_layout(int32_t nco, int32_t nlists, const int32_t* list_off, const int32_t* list_pos, const int32_t* list_points,
                                     int32_t allow_border, int32_t* new_pos, int32_t* band_cameras, int32_t* half_bandwidth) {
  if (nco < 0 || nlists < 0 || (nlists > 0 && (!list_off || !list_pos)) || !new_pos) return BA_ERR_INVALID_ARG;
  std::vector<int> loff(1, 0), lpos, lmult;
  for (int l = 0; l < nlists; ++l) {
    if (list_off[l + 1] < list_off[l]) return BA_ERR_INVALID_ARG;
    for (int k = list_off[l]; k < list_off[l + 1]; ++k) {
      if (list_pos[k] < 0 || list_pos[k] >= nco) return BA_ERR_INVALID_ARG;
      lpos.push_back(list_pos[k]);
    }
    loff.push_back((int)lpos.size());
    lmult.push_back(list_points ? list_points[l] : 2);
  }
  std::vector<int> newpos;
  int n1 = nco, hb = 0;
  if (!ba::plan_camera_layout(nco, loff, lpos, lmult, allow_border != 0, newpos, &n1, &hb)) {
    newpos.resize((size_t)nco);
    for (int p = 0; p < nco; ++p) newpos[p] = p;
    hb = ba::order_half_bandwidth(loff, lpos, newpos);
    n1 = nco;
  }
  std::copy(newpos.begin(), newpos.end(), new_pos);
  if (band_cameras) *band_cameras = n1;
  if (half_bandwidth) *half_bandwidth = hb;
  return BA_OK;
}

extern "C" int ba_order_cameras(int32_t nco, int32_t nlists, const int32_t* list_off, const int32_t* list_pos, int32_t* new_pos,
                                int32_t* half_bandwidth) {
  if (nco < 0 || nlists < 0 || (nlists > 0 && (!list_off || !list_pos))) return BA_ERR_INVALID_ARG;
  std::vector<int> loff(1, 0), lpos;
  for (int l = 0; l < nlists; ++l) {
    if (list_off[l + 1] < list_off[l]) return BA_ERR_INVALID_ARG;
    for (int k = list_off[l]; k < list_off[l + 1]; ++k) {
      if (list_pos[k] < 0 || list_pos[k] >= nco) return BA_ERR_INVALID_ARG;
      lpos.push_back(list_pos[k]);
    }
    loff.push_back((int)lpos.size());
  }
  std::vector<int> newpos;
  ba::cuthill_mckee_order(nco, loff, lpos, newpos);
  if (new_pos) std::copy(newpos.begin(), newpos.end(), new_pos);
  if (half_bandwidth) *half_bandwidth = ba::order_half_bandwidth(loff, lpos, newpos);
  return BA_OK;
}

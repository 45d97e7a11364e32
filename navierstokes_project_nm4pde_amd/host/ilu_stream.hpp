// ilu_stream.hpp — host-side schedule of the packed block-ILU(0) triangular solve (device kernel: csrc/nsx_sparse.hip,
// k_ilu_solve_lanes), i.e. of TrilinosWrappers::PreconditionILU::vmult inside every inner Krylov iteration
// (reference Navier-Stokes/include/Preconditioners.hpp:215-216,382,405; Ifpack_ILU::ApplyInverse, overlap 0 = one
// independent factor per MPI rank).
//
// One WAVE serves a handful of rank blocks; their rows of x live in the wave's LDS.  There is NO cross-lane operation in the
// sweeps: a lane owns a row for as many "ticks" as the row has in-block entries, keeps the row's partial result in registers
//        FIRST tick:  acc = x[row]            every tick:  acc += value * x[col]            LAST tick:  x[row] = acc
// (value = -L_ij in the forward sweep, -U_ij/d_i in the backward one) and then takes over the next row the schedule hands it.
// A lane takes E entries of its row per tick (E = entries_per_tick, 1..4: a tick costs a round trip to the LDS whatever it
// carries, so the more a tick carries the fewer round trips a sweep needs).
// The factor is a linear stream of 64-slot slabs; slab t is tick t of the wave, slot `lane` what lane `lane` does in it:
//        E values and E + 1 halfwords (padded to whole dwords) per slot:
//        h[0] = LDS byte address of x[col_0] | FIRST (bit 0) | LAST (bit 1),  h[k] = address of x[col_k],  h[E] = address of x[row]
// (addresses are 8-byte aligned, so the low bits of h[0] are free).  Unused entries of a slot carry the value 0 and point at the
// lane's own scratch row (64 rows behind the wave's real ones); so do idle slots.
//
// The schedule is list scheduling of the rows (tasks of length = their entries) on 64 processors with two refinements:
//   * a row is planned as soon as all the rows it reads have STARTED (their last tick is known then); its entries are taken in the
//     order in which their columns become readable, so the row runs WHILE its dependencies are still running and its last tick
//     comes `gap` ticks after the last of them instead of a whole row length later (the critical path of a 13-colour block drops
//     from ~150 to ~50 ticks per sweep);
//   * `gap` = 2: x[row] written at tick t may be read from tick t + 2 on, because the kernel issues the LDS reads of tick t + 1
//     BEFORE the LDS write of tick t (the read latency of a tick hides behind the arithmetic of the previous one).
// Blocks are dealt to the waves longest-processing-time first, so the waves carry equal numbers of entries (deal.II's lowest-id
// ownership rule makes the first subdomains twice as large as the mean).
//
// Two other layouts were measured on MI355X and dropped: lane groups of 8 with DPP reductions per step
// (round 2: 141 k slabs at 49 % fill, ~75 vector instructions per slab, 31-33 us) and a free-form entry stream accumulated with
// LDS floating-point atomics (97 k slabs at 71 % fill but 87 us: ds_add_f64 retires about one lane every three cycles).
//
// Pure host code (no HIP): used by the device library's set-up, by the host front-end's test hooks (nsxh_ilu_stream_*) and by
// the CPU tests, which replay the stream tick by tick against plain sequential sweeps.
#pragma once
#include <algorithm>
#include <cstdint>
#include <queue>
#include <stdexcept>
#include <vector>

#include "graph.hpp"

namespace nsx {

struct IluStream {
  int n_waves = 0, ncomp = 1, gap = 2, epl = 1;  // epl: entries per lane and tick (E)
  int max_wave_rows = 0;                // rows of x a wave holds; LDS rows [rows of the wave, +64) are the lanes' scratch rows
  int64_t n_slabs = 0, max_wave_slabs = 0;
  int64_t in_block_nnz = 0;             // entries of the factor that take part (in-block, diagonal included): the algorithmic size
  int64_t used_slots = 0;               // = in-block off-diagonal entries
  std::vector<int32_t> wave_ptr;        // [n_waves + 1] into wave_blk
  std::vector<int32_t> wave_blk;        // blocks of each wave
  std::vector<int32_t> slab_ptr;        // [2 * n_waves + 1]: forward slabs, then backward slabs, per wave
  std::vector<uint32_t> meta;           // [(n_slabs + ILU_STREAM_PAD) * 64 * meta_words(epl)]
  std::vector<int32_t> slot_of;         // [nnz]: position ((slab * 64 + lane) * epl + e) of every in-block off-diagonal CSR entry in the
                                        // value array of the stream, -1 otherwise
  std::vector<int32_t> xoff;            // [n_rows]: LDS row of every row within its wave
  std::vector<int32_t> row_ptr;         // [n_waves + 1] into rows
  std::vector<int32_t> rows;            // the rows of every wave in LDS order (rows[row_ptr[w] + xoff[i]] = i): the kernel's flat load / store loops
  bool ok = false;                      // false: the LDS byte addresses of a wave do not fit 16 bits
};

constexpr int ilu_meta_words(int epl) { return (epl + 2) / 2; }  // E + 1 halfwords
constexpr int ILU_STREAM_PAD = 32;   // idle slabs behind the last wave's stream: the kernel prefetches without a bounds check
#ifndef NSX_ILU_STREAM_ALIGN
#define NSX_ILU_STREAM_ALIGN 4
#endif
constexpr int ILU_STREAM_ALIGN = NSX_ILU_STREAM_ALIGN;  // every sweep of every wave is a multiple of this many slabs (idle ones behind its
                                     // last tick): the kernel runs 8 ticks as one branch-free block and, at a sweep's END only, a block of 4
                                     // (round 4; 8 before: the padding was 8 % of the bench stream)

// g: square graph with sorted columns, structurally symmetric inside every block (FE patterns, B B^T).  bptr: [nb+1] row ranges.
// blocks_per_wave: average number of blocks a wave serves (the wave count is ceil(nb / blocks_per_wave)).
inline void build_ilu_stream(const Csr &g, const std::vector<int32_t> &bptr, int blocks_per_wave, int ncomp, int gap, IluStream &s,
                             int entries_per_tick = 1) {
  const int E = std::max(1, std::min(4, entries_per_tick)), MW = ilu_meta_words(E);
  s.epl = E;
  const int nb = (int)bptr.size() - 1, BPW = std::max(1, blocks_per_wave);
  const int nw = std::max(1, (nb + BPW - 1) / BPW);
  gap = std::max(1, gap);
  s.n_waves = nw;
  s.ncomp = ncomp;
  s.gap = gap;
  s.slab_ptr.assign(2 * (size_t)nw + 1, 0);
  s.meta.clear();
  s.slot_of.assign((size_t)g.nnz(), -1);
  s.xoff.assign(g.n_rows, 0);
  s.max_wave_rows = 0;
  s.in_block_nnz = s.used_slots = 0;
  std::vector<int32_t> blk_of(g.n_rows, 0);
  std::vector<int64_t> blk_work(nb, 0);
  for (int b = 0; b < nb; ++b)
    for (int i = bptr[b]; i < bptr[b + 1]; ++i) {
      blk_of[i] = b;
      for (int q = g.rowptr[i]; q < g.rowptr[i + 1]; ++q) blk_work[b] += g.colind[q] >= bptr[b] && g.colind[q] < bptr[b + 1];
    }
  for (int b = 0; b < nb; ++b) s.in_block_nnz += blk_work[b];
  // ---- blocks -> waves: longest processing time first onto the least loaded wave (ties: lowest wave), then ascending inside a wave
  {
    std::vector<int32_t> order(nb);
    for (int b = 0; b < nb; ++b) order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return blk_work[x] > blk_work[y]; });
    using Load = std::pair<int64_t, int32_t>;
    std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
    for (int w = 0; w < nw; ++w) heap.push({0, w});
    std::vector<std::vector<int32_t>> of(nw);
    for (int b : order) {
      Load t = heap.top();
      heap.pop();
      of[t.second].push_back(b);
      heap.push({t.first + blk_work[b] + 1, t.second});
    }
    s.wave_ptr.assign((size_t)nw + 1, 0);
    s.wave_blk.clear();
    for (int w = 0; w < nw; ++w) {
      std::sort(of[w].begin(), of[w].end());
      s.wave_blk.insert(s.wave_blk.end(), of[w].begin(), of[w].end());
      s.wave_ptr[w + 1] = (int32_t)s.wave_blk.size();
    }
  }
  std::vector<int32_t> len(g.n_rows), pend(g.n_rows), free_at(g.n_rows), order_ptr(g.n_rows), order;
  std::vector<int64_t> height(g.n_rows);
  s.ok = true;
  s.max_wave_slabs = 0;
  const uint32_t row_bytes = 8u * (uint32_t)ncomp;
  s.row_ptr.assign((size_t)nw + 1, 0);
  s.rows.clear();
  s.rows.reserve(g.n_rows);
  for (int w = 0; w < nw; ++w) {
    int wr = 0;
    for (int p = s.wave_ptr[w]; p < s.wave_ptr[w + 1]; ++p) {
      const int b = s.wave_blk[p];
      for (int i = bptr[b]; i < bptr[b + 1]; ++i) {
        s.xoff[i] = wr++;
        s.rows.push_back(i);
      }
    }
    s.row_ptr[w + 1] = (int32_t)s.rows.size();
    s.max_wave_rows = std::max(s.max_wave_rows, wr);
    if ((uint64_t)(wr + 64) * row_bytes > 65536u) s.ok = false;
    for (int dir = 0; dir < 2; ++dir) {
      const bool fwd = dir == 0;
      auto in_part = [&](int i, int j) {
        const int b = blk_of[i];
        return fwd ? (j >= bptr[b] && j < i) : (j > i && j < bptr[b + 1]);
      };
      std::vector<int32_t> work;
      order.clear();
      for (int p = s.wave_ptr[w]; p < s.wave_ptr[w + 1]; ++p) {
        const int b = s.wave_blk[p];
        const int r0 = bptr[b], r1 = bptr[b + 1];
        for (int i = r0; i < r1; ++i) {
          len[i] = 0, pend[i] = 0, height[i] = 0, free_at[i] = -1;
          for (int q = g.rowptr[i]; q < g.rowptr[i + 1]; ++q) len[i] += in_part(i, g.colind[q]);
        }
        // urgency of a row: the longest chain of hand-overs behind it (a row can finish `gap` ticks after the last row it reads)
        for (int t = 0; t < r1 - r0; ++t) {
          const int i = fwd ? r1 - 1 - t : r0 + t;  // dependents before their dependencies
          if (len[i] == 0) continue;
          for (int q = g.rowptr[i]; q < g.rowptr[i + 1]; ++q) {
            const int j = g.colind[q];
            if (in_part(i, j) && len[j] > 0) {
              height[j] = std::max(height[j], height[i] + gap);
              pend[i]++;  // dependencies that have not STARTED yet (rows without entries are final from the start)
            }
          }
        }
        for (int i = r0; i < r1; ++i)
          if (len[i] > 0) {
            work.push_back(i);
            order_ptr[i] = (int32_t)order.size();
            order.resize(order.size() + len[i]);
          }
      }
      struct Lane {
        int row = -1, k = 0;
      };
      auto put_half = [&](size_t slot, int k, uint32_t v) {  // halfword k of a slot
        uint32_t &wd = s.meta[slot * MW + k / 2];
        wd = (k & 1) ? ((wd & 0xffffu) | (v << 16)) : ((wd & 0xffff0000u) | v);
      };
      auto idle_slab = [&]() {
        const size_t base = s.meta.size() / MW;
        s.meta.resize(s.meta.size() + (size_t)64 * MW, 0u);
        for (int l = 0; l < 64; ++l)
          for (int k = 0; k <= E; ++k) put_half(base + l, k, (uint32_t)(wr + l) * row_bytes);
        return base;
      };
      Lane lanes[64];
      std::vector<std::pair<int64_t, int32_t>> avail;    // (urgency, row): released rows not yet started, most urgent first
      std::vector<std::pair<int32_t, int32_t>> planned;  // (release tick, row)
      auto by_urgency = [](const std::pair<int64_t, int32_t> &x, const std::pair<int64_t, int32_t> &y) {
        return x.first != y.first ? x.first > y.first : x.second < y.second;
      };
      std::vector<std::pair<int32_t, int32_t>> tmp;  // (readable from tick, csr position)
      // entry k of the row runs at start + k and needs its column readable by then: release = max_k (readable_k - k)
      auto plan = [&](int i, int not_before) {
        tmp.clear();
        for (int q = g.rowptr[i]; q < g.rowptr[i + 1]; ++q) {
          const int j = g.colind[q];
          if (in_part(i, j)) tmp.push_back({len[j] > 0 ? free_at[j] : 0, q});
        }
        std::sort(tmp.begin(), tmp.end());
        int rel = not_before;
        for (int k = 0; k < (int)tmp.size(); ++k) {
          rel = std::max(rel, tmp[k].first - k / E);  // entry k runs at tick start + k / E
          order[order_ptr[i] + k] = tmp[k].second;
        }
        planned.push_back({rel, i});
      };
      for (int i : work)
        if (pend[i] == 0) plan(i, 0);
      size_t done = 0;
      int tick = 0;
      std::vector<int32_t> started;
      while (done < work.size()) {
        bool grew = false;
        for (size_t k = 0; k < planned.size();) {
          if (planned[k].first <= tick) {
            avail.push_back({height[planned[k].second], planned[k].second});
            planned[k] = planned.back();
            planned.pop_back();
            grew = true;
          } else {
            ++k;
          }
        }
        if (grew) std::sort(avail.begin(), avail.end(), by_urgency);
        const size_t base = idle_slab();
        size_t next_avail = 0;
        started.clear();
        bool any = false;
        for (int l = 0; l < 64; ++l) {
          Lane &L = lanes[l];
          uint32_t first = 0;
          if (L.row < 0 && next_avail < avail.size()) {
            L.row = avail[next_avail++].second;
            L.k = 0;
            first = 1u;
            free_at[L.row] = tick + (len[L.row] + E - 1) / E - 1 + gap;  // first tick at which x[row] may be read
            started.push_back(L.row);
          }
          if (L.row < 0) continue;
          any = true;
          const int n_here = std::min(E, len[L.row] - L.k);
          for (int e = 0; e < n_here; ++e) {
            const int q = order[order_ptr[L.row] + L.k + e];
            put_half(base + l, e, (uint32_t)s.xoff[g.colind[q]] * row_bytes);
            s.slot_of[q] = (int32_t)((base + l) * E + e);
            s.used_slots++;
          }
          L.k += n_here;
          const uint32_t last = L.k == len[L.row] ? 2u : 0u;
          s.meta[(base + l) * MW] |= first | last;
          put_half(base + l, E, (uint32_t)s.xoff[L.row] * row_bytes);
          if (last) {
            ++done;
            L.row = -1;
          }
        }
        avail.erase(avail.begin(), avail.begin() + next_avail);
        if (!any && planned.empty()) throw std::runtime_error("ILU stream: dependency cycle (graph not structurally symmetric inside a block?)");
        for (int j : started) {
          const int b = blk_of[j];
          for (int q = g.rowptr[j]; q < g.rowptr[j + 1]; ++q) {  // structurally symmetric inside a block: dependents = the other triangle of row j
            const int i = g.colind[q];
            if (!(fwd ? (i > j && i < bptr[b + 1]) : (i < j && i >= bptr[b]))) continue;
            if (len[i] == 0) continue;
            if (--pend[i] == 0) plan(i, tick + 1);
          }
        }
        ++tick;
      }
      for (; tick % ILU_STREAM_ALIGN != 0; ++tick) idle_slab();  // up to the block size of the kernel
      s.slab_ptr[2 * (size_t)w + 1 + dir] = (int32_t)(s.meta.size() / ((size_t)64 * MW));
    }
    s.max_wave_slabs = std::max<int64_t>(s.max_wave_slabs, s.slab_ptr[2 * (size_t)w + 2] - s.slab_ptr[2 * (size_t)w]);
  }
  s.n_slabs = (int64_t)(s.meta.size() / ((size_t)64 * MW));
  s.meta.resize(s.meta.size() + (size_t)ILU_STREAM_PAD * 64 * MW, 0u);
}

// Host replay, tick by tick as the kernel runs it: the LDS reads of a tick (gather of x[col], x[row] for FIRST) are taken BEFORE the
// LAST writes of the tick in front of it land.  lu: factors in Ifpack's storage on g (strict lower = L, diagonal = 1/d, strict
// upper = U/d); the stream stores -L and -U/d.  s.ncomp interleaved right-hand sides.
inline void replay_ilu_stream(const Csr &g, const std::vector<int32_t> &bptr, const IluStream &s, const double *lu, const double *b, double *x) {
  const int nc = s.ncomp, E = s.epl, MW = ilu_meta_words(E);
  std::vector<double> val((size_t)(s.n_slabs + ILU_STREAM_PAD) * 64 * E, 0.0);
  for (int64_t q = 0; q < g.nnz(); ++q)
    if (s.slot_of[q] >= 0) val[s.slot_of[q]] = -lu[q];
  std::vector<int32_t> diag(g.n_rows, -1);
  for (int i = 0; i < g.n_rows; ++i) diag[i] = find_in_row(g, i, i);
  auto half = [&](size_t slot, int k) { return (s.meta[slot * MW + k / 2] >> (16 * (k & 1))) & 0xffffu; };
  std::vector<double> xs, acc(64 * (size_t)nc), gx(64 * (size_t)nc * E), fx(64 * (size_t)nc), pend_w(64 * (size_t)nc);
  std::vector<int64_t> pend_a(64);
  for (int w = 0; w < s.n_waves; ++w) {
    xs.assign((size_t)(s.max_wave_rows + 64) * nc, 0.0);
    for (int t = s.row_ptr[w]; t < s.row_ptr[w + 1]; ++t)
      for (int c = 0; c < nc; ++c) xs[(size_t)(t - s.row_ptr[w]) * nc + c] = b[(size_t)s.rows[t] * nc + c];
    for (int dir = 0; dir < 2; ++dir) {
      std::fill(acc.begin(), acc.end(), 0.0);
      std::fill(pend_a.begin(), pend_a.end(), -1);
      const int sa = s.slab_ptr[2 * (size_t)w + dir], sb = s.slab_ptr[2 * (size_t)w + dir + 1];
      auto land = [&]() {
        for (int l = 0; l < 64; ++l)
          if (pend_a[l] >= 0) {
            for (int c = 0; c < nc; ++c) xs[(size_t)pend_a[l] + c] = pend_w[(size_t)l * nc + c];
            pend_a[l] = -1;
          }
      };
      for (int sl = sa; sl < sb; ++sl) {
        for (int l = 0; l < 64; ++l) {  // this tick's reads ...
          const size_t slot = (size_t)sl * 64 + l;
          for (int e = 0; e < E; ++e) {
            const size_t ca = (half(slot, e) & 0xfff8u) / 8;
            for (int c = 0; c < nc; ++c) gx[((size_t)l * E + e) * nc + c] = xs[ca + c];
          }
          const size_t da = half(slot, E) / 8;
          for (int c = 0; c < nc; ++c) fx[(size_t)l * nc + c] = xs[da + c];
        }
        land();  // ... then the previous tick's writes
        for (int l = 0; l < 64; ++l) {
          const size_t slot = (size_t)sl * 64 + l;
          const uint32_t fl = half(slot, 0) & 3u;
          for (int c = 0; c < nc; ++c) {
            double a = (fl & 1u) ? fx[(size_t)l * nc + c] : acc[(size_t)l * nc + c];
            for (int e = 0; e < E; ++e) a += val[slot * E + e] * gx[((size_t)l * E + e) * nc + c];
            acc[(size_t)l * nc + c] = a;
            if (fl & 2u) pend_w[(size_t)l * nc + c] = a;
          }
          if (fl & 2u) pend_a[l] = (int64_t)(half(slot, E) / 8);
        }
      }
      land();
      if (dir == 0)
        for (int t = s.row_ptr[w]; t < s.row_ptr[w + 1]; ++t)
          for (int c = 0; c < nc; ++c) xs[(size_t)(t - s.row_ptr[w]) * nc + c] *= lu[diag[s.rows[t]]];
    }
    for (int t = s.row_ptr[w]; t < s.row_ptr[w + 1]; ++t)
      for (int c = 0; c < nc; ++c) x[(size_t)s.rows[t] * nc + c] = xs[(size_t)(t - s.row_ptr[w]) * nc + c];
  }
}

}  // namespace nsx

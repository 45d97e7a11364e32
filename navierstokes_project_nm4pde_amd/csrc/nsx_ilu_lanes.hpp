// nsx_ilu_lanes.hpp — device side of the lane-owner triangular solve (kernel k_ilu_solve_lanes in nsx_sparse.hip; the stream
// is laid out by host/ilu_stream.hpp).  Kept in a header so that tools/ilu_lanes_bench.hip times exactly this code.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#ifndef NSX_NT
#define NSX_NT 1
#endif

namespace nsx {

template <class T>
__device__ __forceinline__ T lanes_ld(const T *p) {
  if constexpr ((NSX_NT & 4) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}

// ---- lane-owner stream (host/ilu_stream.hpp) ----------------------------------------------------------------------------
// One WAVE per group of rank blocks, their part of x in LDS; slab t of the wave's stream is tick t, slot `lane` what the lane
// does: acc = x[row] at the FIRST tick of a row, acc += value * x[col] every tick, x[row] = acc at its LAST tick.  No cross-lane
// operation, no reduction, no barrier: per tick 2 global loads (prefetched PF slabs ahead in two register sets), one 8*NCOMP-byte
// LDS gather, NCOMP FMAs.  The LDS reads of tick t + 1 are issued BEFORE the write of tick t (the schedule leaves a gap of two
// ticks between a row's last tick and its first reader), so a tick does not wait for its own gather.
// A lane takes E entries of its row per tick; per slot the stream holds E values and the halfwords h[0..E]: h[0] = x[col_0]
// byte address | FIRST (bit 0) | LAST (bit 1), h[k] = x[col_k] address, h[E] = x[row] address (host/ilu_stream.hpp).
// The code is branch-free on purpose: 8 ticks form ONE basic block, so the compiler counts its LDS operations exactly and the
// wait in front of a tick's arithmetic leaves the reads issued for the next tick in flight (behind a branch it has to assume the
// shortest path and waits for everything).  The two per-row accesses are therefore always issued, but cost next to nothing for
// the lanes that do not need them: x[row] is read by the lanes at a FIRST tick, all others read one common word (a broadcast:
// one pass through the LDS banks); x[row] is written by the lanes at a LAST tick, all others write their own scratch row.
// LDS addresses are absolute (address space 3 pointers built from the stream's 16-bit fields, no base to add): the kernel has
// no static LDS, its dynamic array starts at 0.
typedef __attribute__((address_space(3))) double lds_f64;

// what a lane loads per tick: E values and the E + 1 halfwords h[0..E] (host/ilu_stream.hpp), as ceil((E + 1) / 2) dwords
template <int E>
struct LaneSlot {
  double v[E];
  uint32_t m[(E + 2) / 2];
  __device__ __forceinline__ uint32_t half(int k) const { return (k & 1) ? (m[k / 2] >> 16) : (m[k / 2] & 0xffffu); }
  __device__ __forceinline__ uint32_t flags() const { return m[0]; }  // bit 0 FIRST, bit 1 LAST
};

template <int NCOMP, int E>
struct LaneOperands {
  double g[E][NCOMP], f[NCOMP];
};

template <int NCOMP, int E>
__device__ __forceinline__ void lane_read(const LaneSlot<E> &s, LaneOperands<NCOMP, E> &o) {
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const lds_f64 *xj = (const lds_f64 *)(uintptr_t)(e == 0 ? (s.m[0] & 0xfff8u) : s.half(e));
#pragma unroll
    for (int c = 0; c < NCOMP; ++c) o.g[e][c] = xj[c];
  }
  const lds_f64 *xi = (const lds_f64 *)(uintptr_t)((s.flags() & 1u) ? s.half(E) : 0u);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) o.f[c] = xi[c];
}

// one tick: first the LDS reads of the NEXT tick (slot sn) into the other operand set, then this tick's arithmetic and write
template <int NCOMP, int E>
__device__ __forceinline__ void lane_tick(const LaneSlot<E> &s, const LaneSlot<E> &sn, uint32_t scratch, double (&acc)[NCOMP],
                                          const LaneOperands<NCOMP, E> &cur, LaneOperands<NCOMP, E> &nxt) {
  lane_read<NCOMP, E>(sn, nxt);
  const bool first = (s.flags() & 1u) != 0;
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) {
    double a = first ? cur.f[c] : acc[c];
#pragma unroll
    for (int e = 0; e < E; ++e) a = __builtin_fma(s.v[e], cur.g[e][c], a);
    acc[c] = a;
  }
  lds_f64 *xi = (lds_f64 *)(uintptr_t)((s.flags() & 2u) ? s.half(E) : scratch);
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) xi[c] = acc[c];
  __builtin_amdgcn_wave_barrier();  // compiler only: the next tick's reads stay behind this write
}

// sa .. sb: the slabs of one sweep of this wave, a multiple of ILU_STREAM_ALIGN = 4 (host/ilu_stream.hpp pads with idle slabs): blocks
// of PF = 8 ticks and, if four slabs are left at the END of the sweep, one block of four (the register sets and the prefetch depth
// stay at eight slabs: the last load then reaches four slabs into whatever follows in the stream, which is never executed).
// The stream is padded behind its end (ILU_STREAM_PAD slabs): prefetching needs no bounds check.  Uniform base + 32-bit lane
// offset: the loads take the scalar-base addressing form (no 64-bit vector address arithmetic per load).
template <int E, int PF>
__device__ __forceinline__ void lane_load(LaneSlot<E> (&S)[PF], int s0, const uint32_t *__restrict__ meta, const double *__restrict__ val, unsigned lane) {
  constexpr int MW = (E + 2) / 2;
  const double *vs_ = val + (size_t)s0 * (64 * E);
  const uint32_t *ms_ = meta + (size_t)s0 * (64 * MW);
#pragma unroll
  for (int k = 0; k < PF; ++k) {
#pragma unroll
    for (int e = 0; e < E; ++e) S[k].v[e] = lanes_ld(vs_ + ((k * 64u + lane) * E + e));
#pragma unroll
    for (int j = 0; j < MW; ++j) S[k].m[j] = lanes_ld(ms_ + ((k * 64u + lane) * MW + j));
  }
}

// A: the first PF slabs of the sweep, already requested by the caller (lane_load(A, sa, ...)): a wave that is alone on its SIMD
// has nothing else to hide that first trip to memory behind, so the kernel issues it in front of its load / scale passes
template <int NCOMP, int E, int PF>
__device__ __forceinline__ void lane_sweep(LaneSlot<E> (&A)[PF], int sa, int sb, const uint32_t *__restrict__ meta, const double *__restrict__ val,
                                           unsigned lane, uint32_t scratch) {
  static_assert(PF == 4 || PF == 8, "sweeps are padded to multiples of 8 slabs; the two operand sets of the gathers alternate tick by tick");
  constexpr int MW = (E + 2) / 2;
  if (sa >= sb) return;
  LaneSlot<E> B[PF], idle;
  LaneOperands<NCOMP, E> o0, o1;
  double acc[NCOMP];
#pragma unroll
  for (int c = 0; c < NCOMP; ++c) acc[c] = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) idle.v[e] = 0.0;
#pragma unroll
  for (int k = 0; k < MW; ++k) idle.m[k] = scratch | (scratch << 16);
  // PF ticks; the last one pre-reads for the first tick of the next PF slabs (nothing of this wave behind the sweep's end)
#define NSX_USE(S, SNEXT, S0)                                                                  \
  {                                                                                            \
    const bool more_ = (S0) + PF < sb;                                                         \
    _Pragma("unroll") for (int k = 0; k < PF; k += 2) {                                        \
      lane_tick<NCOMP, E>(S[k], S[k + 1], scratch, acc, o0, o1);                               \
      lane_tick<NCOMP, E>(S[k + 1], k + 2 < PF ? S[k + 2 < PF ? k + 2 : 0] : (more_ ? SNEXT : idle), scratch, acc, o1, o0); \
    }                                                                                          \
  }
  // the sweep's last PF / 2 ticks (PF = 8 only: with PF = 4 a block is the alignment unit)
#define NSX_USE_HALF(S)                                                                        \
  {                                                                                            \
    _Pragma("unroll") for (int k = 0; k < PF / 2; k += 2) {                                    \
      lane_tick<NCOMP, E>(S[k], S[k + 1], scratch, acc, o0, o1);                               \
      lane_tick<NCOMP, E>(S[k + 1], k + 2 < PF / 2 ? S[k + 2 < PF / 2 ? k + 2 : 0] : idle, scratch, acc, o1, o0); \
    }                                                                                          \
  }
  lane_read<NCOMP, E>(A[0], o0);
  for (int s0 = sa; s0 < sb; s0 += 2 * PF) {
    lane_load<E, PF>(B, s0 + PF, meta, val, lane);
    if (PF == 8 && sb - s0 < PF) {
      NSX_USE_HALF(A)
      break;
    }
    NSX_USE(A, B[0], s0)
    if (s0 + PF >= sb) break;
    lane_load<E, PF>(A, s0 + 2 * PF, meta, val, lane);
    if (PF == 8 && sb - (s0 + PF) < PF) {
      NSX_USE_HALF(B)
      break;
    }
    NSX_USE(B, A[0], s0 + PF)
  }
#undef NSX_USE_HALF
#undef NSX_USE
}

}  // namespace nsx

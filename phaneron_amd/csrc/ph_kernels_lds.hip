// ph_kernels_lds.hip - the LDS-LUT kernels: same arithmetic as ph_kernels.hip, but the 3
// gamma-LUT lookups per pixel are served from an exact compressed copy of the table held in
// the CU's LDS (ph_lut.h) instead of 256 KiB of global memory.
//
// One workgroup of 1024 lanes per CU (the table takes ~153 KiB of the 160 KiB LDS), grid =
// number of CUs, persistent loop over the frame.
//
// The fused channel pipeline needs TWO tables (reader gamma->linear, writer linear->gamma)
// and only one fits, so it runs in two phases per tile: phase 1 holds the reader table,
// unpacks/converts/combines P quads per lane into registers (linear RGB); phase 2 swaps
// the writer table into LDS and converts/packs those registers.  Table swaps come from L2.
#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_ldslut.h"

#include <cstdlib>

#include <type_traits>

#pragma clang fp contract(off)

// PH_PROBE builds (tools/fused_probe.py; never the shipped library): wave 0 of every workgroup of the fused kernel
// stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at the phase boundaries of its first tile, so the
// sustained shader clock and the share of each phase can be read off the real kernel instead of a microbenchmark.
#ifndef PH_PROBE
#define PH_PROBE 0
#endif
#if PH_PROBE
__device__ unsigned long long g_fused_probe[2048 * 12];
extern "C" int ph_debug_fused_probe(unsigned long long *out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fused_probe), (size_t)n_words * 8);
}
#define PH_STAMP(k)                                                                  \
  do {                                                                               \
    if (threadIdx.x == 0 && tile_begin == wg_begin) {                                \
      g_fused_probe[blockIdx.x * 12 + 2 * (k)] = __builtin_amdgcn_s_memtime();       \
      g_fused_probe[blockIdx.x * 12 + 2 * (k) + 1] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                \
  } while (0)
#else
#define PH_STAMP(k) do { } while (0)
#endif

// The four waves of a SIMD are served oldest first, so left alone the oldest wave of a workgroup races through its
// slices and then idles at the phase barrier while the youngest is still working - and a SIMD with one or two
// waves left issues a VALU instruction only every 5 / 2.6 cycles instead of every 2.3 (tools/fused_probe.py:
// wave 0 waited a third of the tile at the barrier).  Wave priority outranks age: a wave LOWERS its priority as
// it advances through its slices, so whoever is behind is served first and the waves reach the barrier together.
// PH_BALANCE_WAVES=0 builds without it (A/B).
#ifndef PH_BALANCE_WAVES
#define PH_BALANCE_WAVES 1
#endif
#if PH_BALANCE_WAVES
// s_setprio takes an immediate: the switch folds once the slice loop is unrolled
__device__ __forceinline__ void ph_set_wave_priority(int level) {
  switch (level) {
    case 3: __builtin_amdgcn_s_setprio(3); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    default: __builtin_amdgcn_s_setprio(0); break;
  }
}
#define PH_BALANCE(p) ph_set_wave_priority(3 - ((p) * 4) / P)
#else
#define PH_BALANCE(p) do { } while (0)
#endif

namespace ph {


__device__ __forceinline__ uint4 write_quad_lds(const float (&rgb)[18], const WriteK &wk, const LutK &lut) {
  uint32_t y[6], u[3], v[3];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    if ((j & 1) == 0) {
      const Yuv1 c = write_px_lds(rgb[3 * j], rgb[3 * j + 1], rgb[3 * j + 2], wk, lut);
      y[j] = c.y, u[j >> 1] = c.u, v[j >> 1] = c.v;
    } else {
      y[j] = write_px_luma_lds(rgb[3 * j], rgb[3 * j + 1], rgb[3 * j + 2], wk, lut);
    }
  }
  return pack_quad(y, u, v);
}

// ------------------------------------------------------------------------------------------
// fused [v210 read] x N -> combine_N -> v210 write, two LDS phases per tile of BS*P quads.
// Between the phases a quad is carried as its 18 writer-LUT INDICES (clamped, rounded, 16 bits each,
// two per VGPR), not as 18 floats: 9 registers per quad, which is what lets one workgroup hold
// P = 6 quads per lane - a whole 2160p share (5400 quads per CU) in ONE tile, so each frame costs two
// table loads instead of four.
// ------------------------------------------------------------------------------------------
// PH_FUSED_SPLIT builds (tools/fused_split.py; never the shipped library): the kernel's two halves as launches of their own -
// MODE 1 = reader half (table load, phase 1, the packed indices stored to a hand-over buffer), MODE 2 = writer half (indices
// loaded back, table load, phase 2) - to price a split into reader CUs and writer CUs before building its cross-CU protocol.
#ifndef PH_FUSED_SPLIT
#define PH_FUSED_SPLIT 0
#endif
// TAIL: lines that do not end on a 48-pixel block (1280 x 720, src/config.ts:43-54).  The flat index then runs over the quad SLOTS
// of the pitch - the layers' words and the output's are at the same offsets - and a lane's slot is a whole quad, the line's tail
// quad (its 2 or 4 pixels read without the matrix's offset column, v210.ts:88-93, written from truncated indices with round(),
// :169-194) or a slot the reference's writer clears (:131-136).  The other instantiation is the same code as before.
template <int N, int P, int BS, bool PIPE = false, int MODE = 0, bool TAIL = false>
__global__ __launch_bounds__(BS) void fused_v210_combine_lds_kernel(FusedLdsArgs a, uint32_t *handoff = nullptr) {
  static_assert(!TAIL || (!PIPE && MODE == 0), "the tail instantiation exists for the shipped form only");
  const ReadK rk = load_read_k(a.f.rd_cm, a.f.rd_gm);
  const WriteK wk = load_write_k(a.f.wr_cm);
  const LutK rlut = make_lut_k(a.rd), wlut = make_lut_k(a.wr);
  const bool std_matrix = ycbcr_matrix_is_standard(rk);  // uniform
  // Every workgroup owns one contiguous, equally sized range of quads (all CUs finish together)
  // and walks it in tiles of BS*P quads; only the last tile of a range is partially filled.
  // Batched launches give every job its own group of workgroups (uniform per workgroup, so the layer
  // pointers stay scalar loads from the kernel arguments).
  const uint32_t job = blockIdx.x / a.wg_per_job, wg_in_job = blockIdx.x - job * a.wg_per_job;
  auto layer_ptr = [&](int l) { return reinterpret_cast<const uint4 *>(job ? a.more_layers[job - 1][l] : a.f.layers[l]); };
  uint4 *const out_ptr = reinterpret_cast<uint4 *>(job ? a.more_out[job - 1] : a.f.out);
  const uint32_t per_wg = (a.f.total_quads + a.wg_per_job - 1) / a.wg_per_job;
  const uint32_t wg_begin = wg_in_job * per_wg;
  const uint32_t wg_end = wg_begin + per_wg < a.f.total_quads ? wg_begin + per_wg : a.f.total_quads;
  const uint32_t tile_quads = BS * P;
  for (uint32_t tile_begin = wg_begin; tile_begin < wg_end; tile_begin += tile_quads) {
    uint32_t st[P][9];
    // Input words are streamed with a one-deep prefetch that runs THROUGH the slices: while layer l
    // of slice p is being decoded the next word (layer l+1, or layer 0 of slice p+1) is in flight -
    // 8 VGPRs of input instead of 4*N*P, and no slice starts by waiting for HBM.  The very first word
    // is requested before the table load so its latency hides behind the DMA.
    auto quad_of = [&](int p) {
      const uint32_t f = tile_begin + p * BS + threadIdx.x;  // width % 48 == 0: flat index == offset
      return f < wg_end ? f : wg_end - 1;                     // tail lanes recompute the last quad
    };
    uint4 w = load_stream(layer_ptr(0) + quad_of(0));
    // Phase 1 of one slice: N layers of one quad per lane -> combine -> the 18 writer-LUT indices,
    // packed.  Takes the slice's layer-0 word and returns the next slice's (prefetch chain).  A
    // generic lambda instantiated for both matrix shapes; everything it touches stays in registers
    // (a by-reference `w` or argument struct ends up in scratch).
    // slot f of a TAIL frame: 0 = a whole quad, 1 = the line's tail quad, 2 = past the line's pixels
    auto slot_kind = [&](uint32_t f) -> uint32_t {
      const uint32_t g = f - __umulhi(f, a.f.magic_qpp) * a.f.quads_per_line_pitch;
      return g < a.f.quads_per_line_used ? 0u : (g == a.f.quads_per_line_used && a.f.tail_px) ? 1u : 2u;
    };
    auto phase1_slice = [&](auto tag, uint4 w, uint32_t f, uint32_t f_next, bool more, uint32_t(&st)[9]) -> uint4 {
      const bool in_tail = TAIL && slot_kind(f) == 1u;
      const float last = TAIL ? (in_tail ? 0.0f : 1.0f) : 1.0f;
      float acc[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) acc[i] = 0.0f;
#pragma unroll 1  // rolled: one copy of the per-layer code whatever N is (I-cache, compile time)
      for (int l = 0; l < N; ++l) {
        uint4 nxt = w;
        if (l + 1 < N) nxt = load_stream(layer_ptr(l + 1) + f);
        else if (more) nxt = load_stream(layer_ptr(0) + f_next);
        const Yuv6 q = unpack_quad(w);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 t = read_px_lds<decltype(tag)::value>(q.y[j], q.cb[j >> 1], q.cr[j >> 1], rk, rlut, last);
          // acc starts at 0, so layer 0 goes through the same fma: fma(0, k, t) == t, and the
          // sign of a zero can never reach the packed output (combine.ts:45-65 for l >= 1)
          const float kk = 1.0f - t.w;
          // acc = fma(acc, kk, t) written as the three-operand v_fma_f32 with acc as destination.
          // Left to itself LLVM picks v_fmac (d = a*b + d, so the result lands in t's register) and
          // pays for it with a v_mov per accumulator per layer to get the loop-carried value back.
          asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %5\n\tv_fma_f32 %2, %2, %3, %6"
                       : "+v"(acc[3 * j]), "+v"(acc[3 * j + 1]), "+v"(acc[3 * j + 2])
                       : "v"(kk), "v"(t.x), "v"(t.y), "v"(t.z));
        }
        w = nxt;
      }
      // the writer's first step (v210.ts:148-150 index = sat_rte(rgb * 65535)) needs no table: do it
      // here and keep only the 16-bit indices, two per register (v_perm_b32)
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const uint32_t lo = __float_as_uint(TAIL ? lds_lut_index_unit_tail(acc[2 * i], in_tail) : lds_lut_index_unit(acc[2 * i]));
        const uint32_t hi = __float_as_uint(TAIL ? lds_lut_index_unit_tail(acc[2 * i + 1], in_tail) : lds_lut_index_unit(acc[2 * i + 1]));
        st[i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
        // Pin the packed state here.  Without a pin LLVM sinks the whole decode / gamut / combine
        // arithmetic to its first use in phase 2 (past the barrier and the table swap) and keeps
        // the raw LDS words of every lookup alive instead: hundreds of spilled VGPRs.
        asm volatile("" : "+v"(st[i]));
      }
      return w;
    };
    PH_STAMP(0);
    if (MODE == 0 || (MODE == 1 && tile_begin == wg_begin)) {  // a half keeps its one table over all its tiles
      lds_lut_load<BS>(a.rd);
      __syncthreads();
    }
    PH_STAMP(1);
    const uint32_t tile_left = wg_end - tile_begin;  // uniform; slice p holds quads iff p * BS < tile_left
    if constexpr (PIPE) {
      // Software-pipelined phases (ph_ldslut.h lds_lut_issue): the six LDS reads of pixel i + 1 are started BEFORE
      // the results of pixel i are consumed, all the way through quads, layers and slices, so a whole pixel of
      // arithmetic (~45 VALU instructions) sits between every read and its first use.  sched_barrier keeps LLVM
      // from re-serialising the two halves.  Results are the same operations in the same order per pixel.
      auto phase1_all = [&](auto tag) {
        constexpr bool STD = decltype(tag)::value;
        auto issue_first = [&](const uint4 &word) {  // pixel 0 of a quad: Y0, Cb0, Cr0 all sit in word x (v210.ts:58)
          return read_px_issue<STD>((float)((word.x >> 10) & 0x3ff), (float)(word.x & 0x3ff), (float)((word.x >> 20) & 0x3ff), rk, rlut);
        };
        // pixel j of a quad (v210.ts:58-63): Y in word {0,1,1,2,3,3} at bit {10,0,20,10,0,20}, the pair's Cb in word
        // {0,1,2} at bit {0,10,20}, its Cr in word {0,2,3} at bit {20,0,10}; j is a compile-time constant
        auto issue_px = [&](const uint4 &word, int j) {
          const uint32_t wy = j == 0 ? word.x : j < 3 ? word.y : j == 3 ? word.z : word.w;
          const uint32_t sy = (j == 0 || j == 3) ? 10u : (j == 1 || j == 4) ? 0u : 20u;
          const int pr = j >> 1;
          const uint32_t wcb = pr == 0 ? word.x : pr == 1 ? word.y : word.z, scb = 10u * pr;
          const uint32_t wcr = pr == 0 ? word.x : pr == 1 ? word.z : word.w, scr = pr == 0 ? 20u : pr == 1 ? 0u : 10u;
          return read_px_issue<STD>((float)((wy >> sy) & 0x3ff), (float)((wcb >> scb) & 0x3ff), (float)((wcr >> scr) & 0x3ff), rk, rlut);
        };
        PxPending pend = issue_first(w);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (p * BS < tile_left) {  // uniform: skip empty slices of the last tile
            PH_BALANCE(p);
            const uint32_t f = quad_of(p);
            const bool more = (p + 1 < P) && ((p + 1) * BS < tile_left);  // uniform
            const uint32_t f_next = more ? quad_of(p + 1) : f;
            float acc[18];
#pragma unroll
            for (int i = 0; i < 18; ++i) acc[i] = 0.0f;
#pragma unroll 1
            for (int l = 0; l < N; ++l) {
              uint4 nxt = w;
              if (l + 1 < N) nxt = load_stream(layer_ptr(l + 1) + f);
              else if (more) nxt = load_stream(layer_ptr(0) + f_next);
#pragma unroll
              for (int j = 0; j < 6; ++j) {
                PxPending nx;
                // the three code values of pixel j + 1 are extracted here, not for the whole quad up front: the
                // scheduling fences would keep all eighteen alive across the loop (spills)
                if (j < 5) nx = issue_px(w, j + 1);
                else {
                  // next layer's / next slice's first pixel (the last one of a tile is never consumed).  The word was
                  // requested at the top of this iteration; pinning its first use HERE keeps LLVM from extracting its
                  // fields right after the request, which would put the whole HBM latency in front of every quad
                  uint4 first = nxt;
                  asm volatile("" : "+v"(first.x));
                  nx = issue_first(first);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 t = read_px_finish(pend, rk);
                const float kk = 1.0f - t.w;
                asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %5\n\tv_fma_f32 %2, %2, %3, %6"
                             : "+v"(acc[3 * j]), "+v"(acc[3 * j + 1]), "+v"(acc[3 * j + 2])
                             : "v"(kk), "v"(t.x), "v"(t.y), "v"(t.z));
                __builtin_amdgcn_sched_barrier(0);
                pend = nx;
              }
              w = nxt;
            }
#pragma unroll
            for (int i = 0; i < 9; ++i) {
              const uint32_t lo = __float_as_uint(lds_lut_index_unit(acc[2 * i]));
              const uint32_t hi = __float_as_uint(lds_lut_index_unit(acc[2 * i + 1]));
              st[p][i] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
              asm volatile("" : "+v"(st[p][i]));
            }
          }
        }
      };
      if (std_matrix) phase1_all(std::true_type{});
      else phase1_all(std::false_type{});
      PH_STAMP(2);
      __syncthreads();
      lds_lut_load<BS>(a.wr);
      __syncthreads();
      PH_STAMP(3);
      // phase 2, pipelined the same way: indices -> writer table -> matrix -> codes -> packed words
      auto idx_at = [&](int p, int i) {  // M + idx: the 16-bit index ORed into the mantissa of 1.5 * 2^23
        const uint32_t v = st[p][i >> 1];
        return __uint_as_float(((i & 1) ? (v >> 16) : (v & 0xFFFFu)) | 0x4B400000u);
      };
      PxPending wp = write_px_issue(idx_at(0, 0), idx_at(0, 1), idx_at(0, 2), wlut);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (p * BS < tile_left) {
          PH_BALANCE(p);
          const uint32_t f = tile_begin + p * BS + threadIdx.x;
          uint32_t y[6], u[3], v[3];
#pragma unroll
          for (int j = 0; j < 6; ++j) {
            PxPending nx = wp;
            if (j < 5) nx = write_px_issue(idx_at(p, 3 * j + 3), idx_at(p, 3 * j + 4), idx_at(p, 3 * j + 5), wlut);
            else if (p + 1 < P) {
              if ((p + 1) * BS < tile_left) nx = write_px_issue(idx_at(p + 1 < P ? p + 1 : p, 0), idx_at(p + 1 < P ? p + 1 : p, 1), idx_at(p + 1 < P ? p + 1 : p, 2), wlut);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float gr = lds_lut_finish(wp.r), gg = lds_lut_finish(wp.g), gb = lds_lut_finish(wp.b);
            y[j] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.y));
            if ((j & 1) == 0) {
              u[j >> 1] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.u));
              v[j >> 1] = sat_u16_rte(dot4(gr, gg, gb, 1.0f, wk.v));
            }
            __builtin_amdgcn_sched_barrier(0);
            wp = nx;
          }
          if (f < wg_end) store_stream(out_ptr + f, pack_quad(y, u, v));
        }
      }
    } else {
    // A slice is skipped per WAVE, not per workgroup: a 2160p share is 5400 quads = five full slices and 280 quads of a sixth, which
    // only waves 0..4 hold - the other eleven used to recompute the share's last quad through both phases (a ninth of all wave slices;
    // at 1080p, 1350 quads per CU, ten of thirty-two)
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
    // (a wave's priority by the slices it has LEFT, so that the waves with one slice more run one slice ahead, measured slower: 49.4 against 48.8 us)
    if (MODE != 2) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t f = quad_of(p);
      if (p * BS + wave_first < wg_end - tile_begin) {  // uniform per WAVE: skip the slices in which this wave holds no quad
        PH_BALANCE(p);
        const bool more = (p + 1 < P) && ((p + 1) * BS + wave_first < wg_end - tile_begin);  // uniform per wave
        const uint32_t f_next = more ? quad_of(p + 1) : f;
        if (std_matrix) w = phase1_slice(std::true_type{}, w, f, f_next, more, st[p]);
        else w = phase1_slice(std::false_type{}, w, f, f_next, more, st[p]);
      }
    }
    }
    if (MODE == 1) {  // reader half: the nine packed registers of every quad go to the hand-over buffer (nine planes, coalesced)
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (p * BS + wave_first < wg_end - tile_begin) {
          const uint32_t f = quad_of(p);
#pragma unroll
          for (int i = 0; i < 9; ++i) handoff[(size_t)i * a.f.total_quads + f] = st[p][i];
        }
      continue;
    }
    if (MODE == 2) {  // writer half: take them back
#pragma unroll
      for (int p = 0; p < P; ++p)
        if (p * BS + wave_first < wg_end - tile_begin) {
          const uint32_t f = quad_of(p);
#pragma unroll
          for (int i = 0; i < 9; ++i) st[p][i] = __builtin_nontemporal_load(handoff + (size_t)i * a.f.total_quads + f);
        }
    }
    PH_STAMP(2);
    if (MODE == 0 || tile_begin == wg_begin) {
      __syncthreads();
      lds_lut_load<BS>(a.wr);
      __syncthreads();
    }
    PH_STAMP(3);
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t f = tile_begin + p * BS + threadIdx.x;
      if (p * BS + wave_first < wg_end - tile_begin) {
        PH_BALANCE(p);
        float yi[18];
#pragma unroll
        for (int i = 0; i < 9; ++i) {  // M + idx: the index ORed into the mantissa of 1.5 * 2^23
          yi[2 * i] = __uint_as_float((st[p][i] & 0xFFFFu) | 0x4B400000u);
          yi[2 * i + 1] = __uint_as_float((st[p][i] >> 16) | 0x4B400000u);
        }
        uint4 packed;
        if (TAIL) {
          const uint32_t kind = slot_kind(f < wg_end ? f : wg_end - 1);
          if (kind == 0u) packed = write_quad_idx_lds(yi, wk, wlut);
          else if (kind == 1u) packed = write_quad_idx_lds_tail(yi, wk, wlut, a.f.tail_px);
          else packed = make_uint4(0u, 0u, 0u, 0u);
        } else {
          packed = write_quad_idx_lds(yi, wk, wlut);
        }
        if (f < wg_end) store_stream(out_ptr + f, packed);
      }
    }
    }
    PH_STAMP(4);
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// v210 read / write with the table in LDS (single phase).  width % 6 == 0; the f32 side is
// accessed directly (each lane owns 96 contiguous bytes; the six accesses of a wave hit the
// same cache lines, so HBM sees each line once).
// ------------------------------------------------------------------------------------------
// One PIXEL per lane: the 6 lanes of a quad load the same 16 bytes (one request), pick their
// own Y and the pair's Cb/Cr, and the wave stores 64 consecutive float4 = one contiguous 1 KiB
// (a quad-per-lane mapping would write 16-byte pieces at a 96-byte stride: 6x the L2 write
// requests, measured 69 us instead of the HBM time).
__global__ __launch_bounds__(kLdsBlock) void v210_read_lds_kernel(const uint4 *__restrict__ in, float4 *__restrict__ out,
                                                                   uint32_t width, uint32_t quads_per_line_pitch,
                                                                   uint32_t total_px, const float *__restrict__ cm,
                                                                   const float *__restrict__ gm, LutView lut, uint32_t nt) {
  const ReadK k = load_read_k(cm, gm);
  const LutK lk = make_lut_k(lut);
  lds_lut_load(lut);
  __syncthreads();
  const uint32_t tail_from = width - width % 6u;  // the pixels of a line's tail are converted without the matrix's offset column (v210.ts:88-93)
  for (uint32_t p = blockIdx.x * kLdsBlock + threadIdx.x; p < total_px; p += gridDim.x * kLdsBlock) {
    const uint32_t line = p / width, x = p - line * width;  // lines by pitch: any even width
    const uint32_t g = x / 6, j = x - 6 * g, pr = j >> 1;
    const uint4 w = in[(size_t)line * quads_per_line_pitch + g];
    // v210.ts:58-63: Y of pixel j sits in word {0,1,1,2,3,3} at bit {10,0,20,10,0,20};
    // Cb of pair pr in word {0,1,2} at bit {0,10,20}; Cr in word {0,2,3} at bit {20,0,10}
    const uint32_t wy = (j == 0) ? w.x : (j < 3) ? w.y : (j == 3) ? w.z : w.w;
    const uint32_t sy = (j == 0 || j == 3) ? 10u : (j == 1 || j == 4) ? 0u : 20u;
    const uint32_t wcb = pr == 0 ? w.x : pr == 1 ? w.y : w.z;
    const uint32_t wcr = pr == 0 ? w.x : pr == 1 ? w.z : w.w;
    const uint32_t scr = pr == 0 ? 20u : pr == 1 ? 0u : 10u;
    const float yf = (float)((wy >> sy) & 0x3ff);
    const float cbf = (float)((wcb >> (10u * pr)) & 0x3ff);
    const float crf = (float)((wcr >> scr) & 0x3ff);
    store_image(out + p, read_px_lds(yf, cbf, crf, k, lk, x < tail_from ? 1.0f : 0.0f), nt);
  }
}

// Several frames of one size and one colour recipe in ONE launch (the layers of a channel arrive together): the
// workgroups are divided between the frames (frame index uniform per workgroup, pointers stay scalar loads), so the
// launch and the 148 KiB table load per CU are paid once per batch, not once per frame - at 1080p they are half of
// a single read's 13.6 us.
struct ReadBatchArgs {
  const uint4 *in[kMaxLayers];
  float4 *out[kMaxLayers];
  uint32_t wg_per_frame;
};
__global__ __launch_bounds__(kLdsBlock) void v210_read_lds_batch_kernel(ReadBatchArgs a, uint32_t width, uint32_t quads_per_line_pitch,
                                                                         uint32_t total_px, const float *__restrict__ cm,
                                                                         const float *__restrict__ gm, LutView lut, uint32_t nt) {
  const ReadK k = load_read_k(cm, gm);
  const LutK lk = make_lut_k(lut);
  lds_lut_load(lut);
  __syncthreads();
  const uint32_t frame = blockIdx.x / a.wg_per_frame, wg = blockIdx.x - frame * a.wg_per_frame;
  const uint4 *__restrict__ in = a.in[frame];
  float4 *__restrict__ out = a.out[frame];
  const uint32_t tail_from = width - width % 6u;
  for (uint32_t p = wg * kLdsBlock + threadIdx.x; p < total_px; p += a.wg_per_frame * kLdsBlock) {
    const uint32_t line = p / width, x = p - line * width;
    const uint32_t g = x / 6, j = x - 6 * g, pr = j >> 1;
    const uint4 w = in[(size_t)line * quads_per_line_pitch + g];
    const uint32_t wy = (j == 0) ? w.x : (j < 3) ? w.y : (j == 3) ? w.z : w.w;  // as v210_read_lds_kernel
    const uint32_t sy = (j == 0 || j == 3) ? 10u : (j == 1 || j == 4) ? 0u : 20u;
    const uint32_t wcb = pr == 0 ? w.x : pr == 1 ? w.y : w.z;
    const uint32_t wcr = pr == 0 ? w.x : pr == 1 ? w.z : w.w;
    const uint32_t scr = pr == 0 ? 20u : pr == 1 ? 0u : 10u;
    const float yf = (float)((wy >> sy) & 0x3ff);
    const float cbf = (float)((wcb >> (10u * pr)) & 0x3ff);
    const float crf = (float)((wcr >> scr) & 0x3ff);
    store_image(out + p, read_px_lds(yf, cbf, crf, k, lk, x < tail_from ? 1.0f : 0.0f), nt);
  }
}

__global__ __launch_bounds__(kLdsBlock) void v210_write_lds_kernel(const float4 *__restrict__ in, uint4 *__restrict__ out,
                                                                    uint32_t width, uint32_t quads_per_line,
                                                                    uint32_t lines, uint32_t first_line,
                                                                    uint32_t line_step, const float *__restrict__ cm,
                                                                    LutView lut) {
  const WriteK k = load_write_k(cm);
  const LutK lk = make_lut_k(lut);
  lds_lut_load(lut);
  __syncthreads();
  // quads_per_line: the slots of the pitch.  `full` whole quads, then the tail quad of a width that is not a multiple of 6
  // (v210.ts:169-194), then slots the reference clears (v210.ts:131-136); a width in multiples of 48 fills every slot
  const uint32_t total = quads_per_line * lines, full = width / 6u, remain = width - 6u * full;
  for (uint32_t f = blockIdx.x * kLdsBlock + threadIdx.x; f < total; f += gridDim.x * kLdsBlock) {
    const uint32_t li = f / quads_per_line, g = f - li * quads_per_line;
    const uint32_t line = first_line + li * line_step;
    if (g > full || (g == full && !remain)) {
      store_stream(out + (size_t)line * quads_per_line + g, make_uint4(0u, 0u, 0u, 0u));
      continue;
    }
    const float4 *row = in + (size_t)line * width;
    float rgb[18];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const uint32_t x = 6u * g + (uint32_t)j < width ? 6u * g + (uint32_t)j : width - 1u;  // (a tail quad's missing pixels: never packed)
      const float4 p = row[x];
      rgb[3 * j] = p.x, rgb[3 * j + 1] = p.y, rgb[3 * j + 2] = p.z;
    }
    if (g == full) {
      float yi[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) yi[i] = lds_lut_index_unit_tail(rgb[i], true);
      store_stream(out + (size_t)line * quads_per_line + g, write_quad_idx_lds_tail(yi, k, lk, remain));
      continue;
    }
    store_stream(out + (size_t)line * quads_per_line + g, write_quad_lds(rgb, k, lk));
  }
}

// ------------------------------------------------------------------------------------------
// compose + write: N float RGBA layers, each either taken 1:1 or through the `transform`
// sampler (3x3 matrix, bilinear, border 0: transform.ts:36-59), combined with combine_N
// (combine.ts:45-65) and packed to v210 (v210.ts:113-195) in ONE kernel.  This is the
// reference's batch [transform] x N -> combine_N -> write without the N + 1 full-size f32
// frames in between; bit-identical to running those kernels one after the other.
// Single phase: only the writer table is needed.  One output quad per lane.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kLdsBlock) void compose_write_v210_kernel(ComposeArgs a) {
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK lk = make_lut_k(a.wr);
  lds_lut_load(a.wr);
  __syncthreads();
  // A line has `qpl` quad slots by pitch: `full` whole quads, then - widths that are not a multiple of 6 - the tail quad
  // (v210.ts:169-194), then slots the reference's writer clears (v210.ts:131-136).  Widths in multiples of 48 fill every slot;
  // this kernel is the one that serves the others (1280 x 720: the launcher sends them here).
  const uint32_t qpl = (a.out_w + 47u) / 48u * 8u, full = a.out_w / 6u, remain = a.out_w - 6u * full;
  const uint32_t total = qpl * a.lines;
  for (uint32_t f = blockIdx.x * kLdsBlock + threadIdx.x; f < total; f += gridDim.x * kLdsBlock) {
    const uint32_t li = f / qpl, g = f - li * qpl;
    const uint32_t line = a.first_line + li * a.line_step;
    if (g > full || (g == full && !remain)) {  // past the line's pixels
      store_stream(reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + g, make_uint4(0u, 0u, 0u, 0u));
      continue;
    }
    float acc[24];
#pragma unroll 1
    for (int l = 0; l < a.n; ++l) {
      const float4 *img = reinterpret_cast<const float4 *>(a.layers[l]);
      const int lw = a.lw[l], lh = a.lh[l];
      const float *m = a.matrix[l];
      float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
      if (m) m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5];
      const float py = (float)(int)line / (float)(int)a.out_h - 0.5f;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int x = 6 * g + j < a.out_w ? (int)(6 * g + j) : (int)a.out_w - 1;  // (the pixels a tail quad does not have: never packed)
        float4 t;
        if (m) {  // transform.ts:53-57
          const float px = (float)x / (float)(int)a.out_w - 0.5f;
          const float s = dot3(m0, m1, m2, px, py, 1.0f) + 0.5f;
          const float tt = dot3(m3, m4, m5, px, py, 1.0f) + 0.5f;
          t = sample_linear(img, lw, lh, s, tt);
        } else {
          t = img[(size_t)line * a.out_w + x];
        }
        if (l == 0) {
          acc[4 * j] = t.x, acc[4 * j + 1] = t.y, acc[4 * j + 2] = t.z, acc[4 * j + 3] = t.w;
        } else {  // combine.ts:45-65
          const float kk = 1.0f - t.w;
          acc[4 * j] = fma_rn(acc[4 * j], kk, t.x);
          acc[4 * j + 1] = fma_rn(acc[4 * j + 1], kk, t.y);
          acc[4 * j + 2] = fma_rn(acc[4 * j + 2], kk, t.z);
          acc[4 * j + 3] = fma_rn(acc[4 * j + 3], 0.0f, t.w);
        }
      }
    }
    float rgb[18];
#pragma unroll
    for (int j = 0; j < 6; ++j) rgb[3 * j] = acc[4 * j], rgb[3 * j + 1] = acc[4 * j + 1], rgb[3 * j + 2] = acc[4 * j + 2];
    if (g == full) {  // the tail: truncated table indices, round() (v210.ts:173-184)
      float yi[18];
#pragma unroll
      for (int i = 0; i < 18; ++i) yi[i] = lds_lut_index_unit_tail(rgb[i], true);
      store_stream(reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + g, write_quad_idx_lds_tail(yi, wk, lk, remain));
      continue;
    }
    store_stream(reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + g, write_quad_lds(rgb, wk, lk));
  }
}

// The same compositor with ONE PIXEL PER LANE.  A quad-per-lane mapping puts the lanes of a wave 6
// pixels apart, so every bilinear tap of a wave touches ~6x the cache lines it needs (the texture
// path is then the bound: 170 us for 4 x 1080p -> 2160p).  Here a wave samples 64 consecutive pixels
// per step (taps of neighbouring lanes fall into the same lines), combines, runs the writer LUT and
// matrix per pixel, and parks the 16-bit code values of 192 pixels in a small LDS area behind the table;
// lanes 0..31 then pack the chunk's 32 quads and store 512 contiguous bytes.  Bit-identical results.
constexpr uint32_t kComposeChunk = 192;                                   // pixels per wave per step: 32 quads
constexpr uint32_t kComposeStageBytes = (kLdsBlock / 64) * kComposeChunk * 4;  // Y[192] + U[96] + V[96] u16 per wave
template <int N, bool ALL_DIRECT>
__global__ __launch_bounds__(kLdsBlock) void compose_write_v210_px_kernel(ComposeArgs a, uint32_t stage_off) {
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK lk = make_lut_k(a.wr);
  // transform matrices once per kernel, not once per pixel: a load inside the layer loop puts a second
  // memory round trip in front of every sample
  float mm[N][6];
  bool direct[N];
#pragma unroll
  for (int l = 0; l < N; ++l) {
    direct[l] = a.matrix[l] == nullptr;
#pragma unroll
    for (int i = 0; i < 6; ++i) mm[l][i] = direct[l] ? 0.0f : a.matrix[l][i];
  }
  lds_lut_load(a.wr);
  __syncthreads();
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint16_t *ys = reinterpret_cast<uint16_t *>(g_lds + stage_off + wave * kComposeChunk * 4);
  uint16_t *us = ys + kComposeChunk, *vs = us + kComposeChunk / 2;
  const uint32_t qpl = a.out_w / 6;                       // out_w % 48 == 0
  const uint32_t total_px = a.out_w * a.lines;            // a multiple of 48: chunks end on quad boundaries
  const uint32_t waves_total = gridDim.x * (kLdsBlock / 64);
  for (uint32_t base = (blockIdx.x * (kLdsBlock / 64) + wave) * kComposeChunk; base < total_px;
       base += waves_total * kComposeChunk) {
#pragma unroll 1
    for (uint32_t k = 0; k < kComposeChunk / 64; ++k) {
      const uint32_t local = k * 64 + lane;
      uint32_t p = base + local;
      p = p < total_px ? p : total_px - 1;
      const uint32_t li = p / a.out_w, x = p - li * a.out_w;
      const uint32_t line = a.first_line + li * a.line_step;
      const float py = (float)(int)line / (float)(int)a.out_h - 0.5f;
      const float px = (float)(int)x / (float)(int)a.out_w - 0.5f;
      // all layers are sampled before any is combined: N x 4 independent loads in flight
      float4 t[N];
#pragma unroll
      for (int l = 0; l < N; ++l) {  // transform.ts:53-57; a layer without a matrix is taken 1:1
        if (ALL_DIRECT) {  // no layer is sampled (host knows): one load per layer
          t[l] = load_stream(reinterpret_cast<const float4 *>(a.layers[l]) + (size_t)line * a.out_w + x);
          continue;
        }
        const float s = dot3(mm[l][0], mm[l][1], mm[l][2], px, py, 1.0f) + 0.5f;
        const float tt = dot3(mm[l][3], mm[l][4], mm[l][5], px, py, 1.0f) + 0.5f;
        t[l] = sample_linear<true>(reinterpret_cast<const float4 *>(a.layers[l]), a.lw[l], a.lh[l], s, tt, direct[l], x, line);
      }
      float r = t[0].x, g = t[0].y, b = t[0].z;
#pragma unroll
      for (int l = 1; l < N; ++l) {  // combine.ts:45-65 (alpha of the result is never used by the writer)
        const float kk = 1.0f - t[l].w;
        r = fma_rn(r, kk, t[l].x), g = fma_rn(g, kk, t[l].y), b = fma_rn(b, kk, t[l].z);
      }
      const Yuv1 c = write_px_lds(r, g, b, wk, lk);  // v210.ts:145-156
      ys[local] = (uint16_t)c.y;
      if (!(x & 1)) us[local >> 1] = (uint16_t)c.u, vs[local >> 1] = (uint16_t)c.v;
    }
    // one wave writes and reads its own staging area: LDS operations of a wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < kComposeChunk / 6) {
      const uint32_t first_px = base + lane * 6;
      if (first_px < total_px) {
        uint32_t y[6], u[3], v[3];
#pragma unroll
        for (int j = 0; j < 6; ++j) y[j] = ys[lane * 6 + j];
#pragma unroll
        for (int j = 0; j < 3; ++j) u[j] = us[lane * 3 + j], v[j] = vs[lane * 3 + j];
        const uint32_t li = first_px / a.out_w, x = first_px - li * a.out_w;
        const uint32_t line = a.first_line + li * a.line_step;
        store_stream(reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + x / 6, pack_quad(y, u, v));
      }
    }
    __builtin_amdgcn_wave_barrier();  // the next step overwrites the staging area
  }
}

// ------------------------------------------------------------------------------------------
// The pixel-per-lane compositor again, with the sampler's address and border work handed to the
// buffer addressing hardware.  Same arithmetic, same results; per sampled layer-pixel ~60 VALU
// instructions instead of ~100:
//   * taps are RAW BUFFER loads (one 128-bit resource per layer, num_records = the image's bytes).
//     A tap outside the image must read as 0 (transform.ts: CLK_ADDRESS_CLAMP, border 0): its byte
//     offset is made >= num_records and the hardware returns 0 without touching memory - no
//     per-texel selects (16 v_cndmask), no clamped addresses, no 64-bit pointer arithmetic.
//     Rows: j0 is clamped to [-2, h] and multiplied by the pitch, so rows -2, -1, h, h+1 land outside
//     [0, num_records) by themselves (offsets wrap to just below 2^32 or reach num_records).
//     Columns: an outside column contributes 2^31 instead of 16 x.  Needs num_records + 2 pitches
//     <= 2^31 (host-checked: images below 2 GiB);
//   * a chunk (192 pixels) lies in one output row (out_w % 192 == 0, host-checked), so the row is
//     wave-uniform: for a layer whose matrix has m1 == m3 == 0 (every placement without rotation)
//     t', the two row offsets and the vertical weight are computed once per chunk, not per pixel;
//   * a layer none of whose taps is inside for a whole wave (the outside of a picture-in-picture
//     inset) skips its loads and its blend: its sample is the border value 0;
//   * 1:1 layers and sampled layers take uniform branches instead of selects.
// ------------------------------------------------------------------------------------------
#ifndef PH_COMPOSE_GROUP_ROWS
#define PH_COMPOSE_GROUP_ROWS 16
#endif
typedef uint32_t ph_u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kTapOutside = 0x80000000u;

struct RowTaps {  // one sampled layer, one source row pair
  uint32_t r0, r1;  // byte offsets of rows j0 and j0 + 1 (outside rows: outside [0, num_records))
  float b, omb;
  bool any;         // at least one of the two rows is inside
};

__device__ __forceinline__ RowTaps row_taps(float tt, int lw, int lh) {  // the `v` half of sample_linear
  const float v = tt * (float)lh;
  const float fv = v - 0.5f;
  const float flv = __builtin_floorf(fv);
  int j0 = (int)flv;
  j0 = j0 < -2 ? -2 : j0;
  j0 = j0 > lh ? lh : j0;
  RowTaps t;
  t.b = fv - flv, t.omb = 1.0f - t.b;
  t.r0 = (uint32_t)__mul24(j0, lw * 16);
  t.r1 = t.r0 + (uint32_t)(lw * 16);
  t.any = (uint32_t)(j0 + 1) <= (uint32_t)lh;  // j0 in [-1, h - 1]
  return t;
}

__device__ __forceinline__ float4 as_float4(const ph_u32x4 v) {
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// One chunk row of the compositor.  ALIGNED: every sampled layer has m1 == m3 == 0 (row taps once per chunk);
// MIXED: some layers are taken 1:1 (selects, no branches: the loads of all layers stay in flight together).
// SHARED: every layer is sampled through the same matrix buffer at the same source size (full-frame layers of one
// channel): tap offsets and weights are computed once per pixel, only the resource differs per layer.
// WIPE: a layer may carry a wipe transition (transition.ts:58-77 as the Transitioner runs it): its placed pixel t
// becomes fma(b, m, t * (1 - m)) per component, b = the incoming source's pixel, m = the mask's red - the transform
// -> transition_wipe -> combine chain of a channel in the middle of a wipe, without the two frames in between.
template <int N, bool MIXED, bool ALIGNED, bool SHARED = false, bool WIPE = false>
__device__ __forceinline__ void compose_taps_body(const ComposeArgs &a, const float (&mm)[N][6], const bool (&direct)[N],
                                                  const __amdgpu_buffer_rsrc_t (&img)[N], const WriteK &wk, const LutK &lk,
                                                  uint16_t *ys, uint16_t *us, uint16_t *vs, uint32_t wave, uint32_t lane) {
  const uint32_t qpl = a.out_w / 6;
  const uint32_t chunks = a.out_w * a.lines / kComposeChunk;  // out_w % 192 == 0: a chunk never leaves its row
  const float fow = (float)(int)a.out_w, foh = (float)(int)a.out_h;
  // XCD-aware order.  Neighbouring output rows sample the same source rows, and each XCD has its own L2: with chunks
  // dealt out to workgroups in plain order, the 256 workgroups of a moment cover ~200 consecutive rows and, workgroups
  // going to XCDs round-robin, every XCD pulls every source row of that window through its own L2 - eight fills per
  // source row from the Infinity Cache.  Here the frame is cut into groups of 16 output rows and group g belongs to
  // XCD g % 8 (workgroups with blockIdx % 8 == x): a source row is filled by one XCD, two at a group boundary, and the
  // interleave keeps the XCDs' loads equal when layers cover only part of the frame (picture-in-picture).
  const uint32_t cpg = (uint32_t)PH_COMPOSE_GROUP_ROWS * (a.out_w / kComposeChunk);  // chunks per group
  uint32_t v_begin = blockIdx.x * (kLdsBlock / 64) + wave, v_end = chunks, v_step = gridDim.x * (kLdsBlock / 64), xcd = 0;
  const bool banded = (gridDim.x & 7u) == 0;
  if (banded) {
    xcd = blockIdx.x & 7u;
    const uint32_t groups = (chunks + cpg - 1u) / cpg, mine = (groups + 7u - xcd) / 8u;  // groups xcd, xcd + 8, ...
    v_begin = (blockIdx.x >> 3) * (kLdsBlock / 64) + wave, v_end = mine * cpg, v_step = (gridDim.x >> 3) * (kLdsBlock / 64);
  }
  for (uint32_t v = v_begin; v < v_end; v += v_step) {
    uint32_t chunk = v;
    if (banded) {
      const uint32_t gi = v / cpg;
      chunk = (gi * 8u + xcd) * cpg + (v - gi * cpg);
      if (chunk >= chunks) continue;  // the frame's last group may be short
    }
    const uint32_t base = chunk * kComposeChunk;
    const uint32_t li = base / a.out_w, x_first = base - li * a.out_w;
    const uint32_t line = a.first_line + li * a.line_step;
    const float py = (float)(int)line / foh - 0.5f;
    RowTaps row[N];
    if (ALIGNED) {
#pragma unroll
      for (int l = 0; l < (SHARED ? 1 : N); ++l)  // transform.ts:53-57 with m3 == 0: t' does not depend on x
        row[l] = row_taps(dot3(mm[l][3], mm[l][4], mm[l][5], 0.0f, py, 1.0f) + 0.5f, a.lw[l], a.lh[l]);
    }
#pragma unroll 1
    for (uint32_t k = 0; k < kComposeChunk / 64; ++k) {
      const uint32_t local = k * 64 + lane;
      const uint32_t x = x_first + local;
      const float px = (float)(int)x / fow - 0.5f;
      const uint32_t own = (line * a.out_w + x) * 16u;  // a 1:1 layer's texel
      ph_u32x4 tap[N][4];
      float wa[N], wb[N];
      float4 wipe_b[N];
      float wipe_m[N];
      if (WIPE) {
#pragma unroll
        for (int l = 0; l < N; ++l)
          if (a.wipe_with[l]) {  // uniform
            wipe_b[l] = reinterpret_cast<const float4 *>(a.wipe_with[l])[(size_t)line * a.out_w + x];
            wipe_m[l] = reinterpret_cast<const float4 *>(a.wipe_mask[l])[(size_t)line * a.out_w + x].x;
          }
      }
      uint32_t o00 = 0, o10 = 0, o01 = 0, o11 = 0;
      // every layer's loads are issued before any is blended: 4 N independent loads in flight
#pragma unroll
      for (int l = 0; l < N; ++l) {
        if (SHARED && l > 0) {
          wa[l] = wa[0], wb[l] = wb[0];
          tap[l][0] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o00, 0, 0);
          tap[l][1] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o10, 0, 0);
          tap[l][2] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o01, 0, 0);
          tap[l][3] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o11, 0, 0);
          continue;
        }
        const RowTaps r = ALIGNED ? row[l] : row_taps(dot3(mm[l][3], mm[l][4], mm[l][5], px, py, 1.0f) + 0.5f, a.lw[l], a.lh[l]);
        const float s = dot3(mm[l][0], mm[l][1], mm[l][2], px, py, 1.0f) + 0.5f;
        const float u = s * (float)a.lw[l];
        const float fu = u - 0.5f;
        const float flu = __builtin_floorf(fu);
        const uint32_t i0 = (uint32_t)(int)flu, i1 = i0 + 1u;
        const uint32_t c0 = i0 < (uint32_t)a.lw[l] ? i0 << 4 : kTapOutside, c1 = i1 < (uint32_t)a.lw[l] ? i1 << 4 : kTapOutside;
        wa[l] = fu - flu, wb[l] = r.b;
        o00 = r.r0 + c0, o10 = r.r0 + c1, o01 = r.r1 + c0, o11 = r.r1 + c1;
        if (MIXED) o00 = direct[l] ? own : o00;
        tap[l][0] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o00, 0, 0);
#ifdef PH_TAPS_ONE_LOAD  // timing experiment (wrong results): one tap per sample instead of four
        tap[l][1] = tap[l][0] + o10, tap[l][2] = tap[l][0] + o01, tap[l][3] = tap[l][0] + o11;
#else
        tap[l][1] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o10, 0, 0);
        tap[l][2] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o01, 0, 0);
        tap[l][3] = __builtin_amdgcn_raw_buffer_load_b128(img[l], (int)o11, 0, 0);
#endif
      }
      float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll
      for (int l = 0; l < N; ++l) {  // OpenCL 1.2 8.2, evaluated as DESIGN.md 2 states it
        const float4 t00 = as_float4(tap[l][0]), t10 = as_float4(tap[l][1]), t01 = as_float4(tap[l][2]), t11 = as_float4(tap[l][3]);
        const float oma = 1.0f - wa[l], omb = 1.0f - wb[l];
        const float w00 = oma * omb, w10 = wa[l] * omb, w01 = oma * wb[l], w11 = wa[l] * wb[l];
        float4 t;
        t.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
        t.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
        t.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
        t.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
        if (MIXED) t.x = direct[l] ? t00.x : t.x, t.y = direct[l] ? t00.y : t.y, t.z = direct[l] ? t00.z : t.z, t.w = direct[l] ? t00.w : t.w;
        if (WIPE && a.wipe_with[l]) {  // transition.ts wipe: mix(in0, in1, mask) with the rounding of twipe_kernel
          const float m = wipe_m[l], rm = 1.0f - m;
          t.x = fma_rn(wipe_b[l].x, m, t.x * rm), t.y = fma_rn(wipe_b[l].y, m, t.y * rm);
          t.z = fma_rn(wipe_b[l].z, m, t.z * rm), t.w = fma_rn(wipe_b[l].w, m, t.w * rm);
        }
        if (l == 0) {
          r = t.x, g = t.y, b = t.z;
        } else {  // combine.ts:45-65 (alpha of the result is never used by the writer)
          const float kk = 1.0f - t.w;
          r = fma_rn(r, kk, t.x), g = fma_rn(g, kk, t.y), b = fma_rn(b, kk, t.z);
        }
      }
      const Yuv1 c = write_px_lds(r, g, b, wk, lk);  // v210.ts:145-156
      ys[local] = (uint16_t)c.y;
      if (!(x & 1)) us[local >> 1] = (uint16_t)c.u, vs[local >> 1] = (uint16_t)c.v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < kComposeChunk / 6) {
      uint32_t y[6], u[3], v[3];
#pragma unroll
      for (int j = 0; j < 6; ++j) y[j] = ys[lane * 6 + j];
#pragma unroll
      for (int j = 0; j < 3; ++j) u[j] = us[lane * 3 + j], v[j] = vs[lane * 3 + j];
      store_stream(reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + (x_first + lane * 6) / 6, pack_quad(y, u, v));
    }
    __builtin_amdgcn_wave_barrier();  // the next step overwrites the staging area
  }
}

template <int N, bool MIXED, bool SHARED = false, bool WIPE = false>
__global__ __launch_bounds__(kLdsBlock) void compose_write_v210_taps_kernel(ComposeArgs a, uint32_t stage_off) {
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK lk = make_lut_k(a.wr);
  float mm[N][6];
  bool direct[N];
  __amdgpu_buffer_rsrc_t img[N];
  bool all_aligned = true;
#pragma unroll
  for (int l = 0; l < N; ++l) {
    direct[l] = a.matrix[l] == nullptr;
#pragma unroll
    for (int i = 0; i < 6; ++i) mm[l][i] = direct[l] ? 0.0f : a.matrix[l][i];
    all_aligned = all_aligned && mm[l][1] == 0.0f && mm[l][3] == 0.0f;
    img[l] = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(a.layers[l]), 0, a.lw[l] * a.lh[l] * 16, 0x00020000);
  }
  lds_lut_load(a.wr);
  __syncthreads();
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  uint16_t *ys = reinterpret_cast<uint16_t *>(g_lds + stage_off + wave * kComposeChunk * 4);
  uint16_t *us = ys + kComposeChunk, *vs = us + kComposeChunk / 2;
  if (all_aligned)
    compose_taps_body<N, MIXED, true, SHARED, WIPE>(a, mm, direct, img, wk, lk, ys, us, vs, wave, lane);
  else
    compose_taps_body<N, MIXED, false, SHARED, WIPE>(a, mm, direct, img, wk, lk, ys, us, vs, wave, lane);
}

template <int N>
static hipError_t launch_compose_taps_n(hipStream_t s, const ComposeArgs &a, uint32_t grid, uint32_t stage_off) {
  bool mixed = false, shared = N > 1, wipe = false;
  for (int l = 0; l < N; ++l) {
    wipe = wipe || a.wipe_with[l] != nullptr;
    mixed = mixed || a.matrix[l] == nullptr;
    // one placement for all layers: the same matrix BUFFER (so the same nine values) and the same source size
    shared = shared && a.matrix[l] != nullptr && a.matrix[l] == a.matrix[0] && a.lw[l] == a.lw[0] && a.lh[l] == a.lh[0];
  }
  if (wipe) {  // one variant: 1:1 layers allowed, placements per layer
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(compose_write_v210_taps_kernel<N, true, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
    if (e != hipSuccess) return e;
    compose_write_v210_taps_kernel<N, true, false, true><<<grid, kLdsBlock, stage_off + kComposeStageBytes, s>>>(a, stage_off);
    return hipGetLastError();
  }
  const void *fn = shared ? reinterpret_cast<const void *>(compose_write_v210_taps_kernel<N, false, true>)
                   : mixed ? reinterpret_cast<const void *>(compose_write_v210_taps_kernel<N, true>)
                           : reinterpret_cast<const void *>(compose_write_v210_taps_kernel<N, false>);
  hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (e != hipSuccess) return e;
  if (shared)
    compose_write_v210_taps_kernel<N, false, true><<<grid, kLdsBlock, stage_off + kComposeStageBytes, s>>>(a, stage_off);
  else if (mixed)
    compose_write_v210_taps_kernel<N, true><<<grid, kLdsBlock, stage_off + kComposeStageBytes, s>>>(a, stage_off);
  else
    compose_write_v210_taps_kernel<N, false><<<grid, kLdsBlock, stage_off + kComposeStageBytes, s>>>(a, stage_off);
  return hipGetLastError();
}

// the buffer-addressed kernel serves a job when a chunk stays in its row and every offset fits the border scheme
static bool compose_taps_eligible(const ComposeArgs &a) {
  if (a.out_w % kComposeChunk) return false;
  bool sampled = false;
  for (int l = 0; l < a.n; ++l) {
    const uint64_t bytes = (uint64_t)a.lw[l] * (uint64_t)a.lh[l] * 16u, pitch = (uint64_t)a.lw[l] * 16u;
    if (bytes + 2 * pitch > 0x80000000ull || pitch >= (1u << 23) || a.lh[l] >= (1 << 22)) return false;
    sampled = sampled || a.matrix[l] != nullptr;
  }
  for (int l = 0; l < a.n; ++l) sampled = sampled || a.wipe_with[l] != nullptr;
  return sampled;  // all layers 1:1: the streaming kernel above is HBM-bound already
}
// The launcher's real path choice: the buffer-addressed kernel needs the staging area behind the writer table (a table
// blob within kComposeStageBytes of the 160 KiB does not leave room) and can be switched off for A/B runs.
static bool compose_stage_fits(const ComposeArgs &a) { return ((a.wr.bytes + 15u) & ~15u) + kComposeStageBytes <= 160u * 1024u; }
static bool compose_quad_forced() {
  static const int quad_env = [] {
    const char *e = getenv("PH_COMPOSE_QUAD");  // 1 = the quad-per-lane kernel (A/B runs)
    return e ? atoi(e) : 0;
  }();
  return quad_env != 0;
}
static bool compose_taps_enabled() {
  static const int taps_env = [] {
    const char *e = getenv("PH_COMPOSE_TAPS");  // 0 = the pointer-addressed sampler (A/B runs)
    return e ? atoi(e) : 1;
  }();
  return taps_env != 0;
}
static bool compose_uses_taps(const ComposeArgs &a) {
  return !compose_quad_forced() && compose_stage_fits(a) && compose_taps_enabled() && compose_taps_eligible(a);
}
// only the buffer-addressed kernel reads wipe_with / wipe_mask: a wipe job is served iff the launcher will pick it
bool compose_can_wipe(const ComposeArgs &a) { return compose_uses_taps(a); }

template <int N, bool ALL_DIRECT>
static hipError_t launch_compose_px_nd(hipStream_t s, const ComposeArgs &a, uint32_t grid, uint32_t stage_off) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(compose_write_v210_px_kernel<N, ALL_DIRECT>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
  if (e != hipSuccess) return e;
  compose_write_v210_px_kernel<N, ALL_DIRECT><<<grid, kLdsBlock, stage_off + kComposeStageBytes, s>>>(a, stage_off);
  return hipGetLastError();
}
template <int N>
static hipError_t launch_compose_px_n(hipStream_t s, const ComposeArgs &a, uint32_t grid, uint32_t stage_off) {
  bool all_direct = true;
  for (int l = 0; l < N; ++l) all_direct = all_direct && a.matrix[l] == nullptr;
  return all_direct ? launch_compose_px_nd<N, true>(s, a, grid, stage_off) : launch_compose_px_nd<N, false>(s, a, grid, stage_off);
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <typename K>
static hipError_t allow_lds(K kernel, uint32_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxDynamicLds);
}

#if PH_FUSED_SPLIT
template <int N, int P, int BS, int MODE>
static hipError_t launch_fused_half(hipStream_t s, FusedLdsArgs b, uint32_t lds) {
  static uint32_t *handoff = nullptr;
  static size_t handoff_quads = 0;
  if (handoff_quads < b.f.total_quads) {
    if (handoff) hipFree(handoff);
    if (hipMalloc(&handoff, (size_t)b.f.total_quads * 36) != hipSuccess) return hipErrorOutOfMemory;
    handoff_quads = b.f.total_quads;
  }
  hipError_t e = allow_lds(fused_v210_combine_lds_kernel<N, P, BS, false, MODE>, lds);
  if (e != hipSuccess) return e;
  fused_v210_combine_lds_kernel<N, P, BS, false, MODE><<<b.wg_per_job * b.jobs, BS, lds, s>>>(b, handoff);
  return hipGetLastError();
}
#endif

template <int N, int P, int BS, bool PIPE = false, bool TAIL = false>
static hipError_t launch_fused_npb(hipStream_t s, const FusedLdsArgs &a, uint32_t grid_in, uint32_t lds) {
  uint32_t grid = grid_in;
#if PH_FUSED_SPLIT
  static const int cus_env = [] {
    const char *e = getenv("PH_FUSED_CUS");  // run on this many CUs (what a role would get in a split)
    return e ? atoi(e) : 0;
  }();
  if (cus_env > 0 && (uint32_t)cus_env < grid) grid = (uint32_t)cus_env;
#endif
  hipError_t e = allow_lds(fused_v210_combine_lds_kernel<N, P, BS, PIPE, 0, TAIL>, lds);
  if (e != hipSuccess) return e;
  const uint32_t slices = (a.f.total_quads + BS - 1) / BS;  // never more workgroups per job than slices
  FusedLdsArgs b = a;
  if (b.jobs < 1) b.jobs = 1;
  b.wg_per_job = grid / b.jobs ? grid / b.jobs : 1;
  if (b.wg_per_job > slices) b.wg_per_job = slices;
#if PH_FUSED_SPLIT
  if (!PIPE && N == 4) {
    static const int mode_env = [] {
      const char *e = getenv("PH_FUSED_MODE");
      return e ? atoi(e) : 0;
    }();
    // with fewer CUs a CU's share no longer fits one tile: the halves then walk several tiles, each with its table load
    if (mode_env == 1) return launch_fused_half<N, P, BS, 1>(s, b, lds);
    if (mode_env == 2) return launch_fused_half<N, P, BS, 2>(s, b, lds);
    if (mode_env == 3) {  // both halves, back to back
      hipError_t e1 = launch_fused_half<N, P, BS, 1>(s, b, lds);
      return e1 != hipSuccess ? e1 : launch_fused_half<N, P, BS, 2>(s, b, lds);
    }
  }
#endif
  fused_v210_combine_lds_kernel<N, P, BS, PIPE, 0, TAIL><<<b.wg_per_job * b.jobs, BS, lds, s>>>(b);
  return hipGetLastError();
}

// Geometry: one workgroup of 1024 lanes per CU (the table fills the LDS), 4 waves per SIMD, at most
// 128 VGPRs per lane.  P = 6 quads per lane (54 VGPRs of packed state) covers a 2160p share (5400
// quads per CU) in one tile; PH_FUSED_GEOM=4|8 selects other P for A/B runs.
template <int N>
static hipError_t launch_fused_n(hipStream_t s, const FusedLdsArgs &a, uint32_t grid, uint32_t lds) {
  static const int geom_env = [] {
    const char *e = getenv("PH_FUSED_GEOM");
    return e ? atoi(e) : 0;
  }();
  if (a.f.quads_per_line_used != a.f.quads_per_line_pitch) return launch_fused_npb<N, 6, 1024, false, true>(s, a, grid, lds);  // ragged lines
  int geom = geom_env ? geom_env : 6;
  static const int pipe_env = [] {
    const char *e = getenv("PH_FUSED_PIPE");  // 1 = the software-pipelined phases (measured 3 % SLOWER: DESIGN.md 4)
    return e ? atoi(e) : 0;
  }();
#ifdef PH_FUSED_EXPERIMENT  // timing builds only: smaller workgroups (more registers per wave) for N = 4
  static const int bs_env = [] {
    const char *e = getenv("PH_FUSED_BS");
    return e ? atoi(e) : 0;
  }();
  if (N == 4 && bs_env == 768) return pipe_env ? launch_fused_npb<4, 8, 768, true>(s, a, grid, lds) : launch_fused_npb<4, 8, 768>(s, a, grid, lds);
  if (N == 4 && bs_env == 512) return pipe_env ? launch_fused_npb<4, 11, 512, true>(s, a, grid, lds) : launch_fused_npb<4, 11, 512>(s, a, grid, lds);
#endif
  if (pipe_env && geom == 6) return launch_fused_npb<N, 6, 1024, true>(s, a, grid, lds);
  if (geom == 4) return launch_fused_npb<N, 4, 1024>(s, a, grid, lds);
  if (geom == 8) return launch_fused_npb<N, 8, 1024>(s, a, grid, lds);
  return launch_fused_npb<N, 6, 1024>(s, a, grid, lds);
}

hipError_t launch_fused_v210_combine_lds(hipStream_t s, int n, const FusedLdsArgs &a, uint32_t num_cus) {
  const uint32_t lds = a.rd.bytes > a.wr.bytes ? a.rd.bytes : a.wr.bytes;
  switch (n) {
    case 1: return launch_fused_n<1>(s, a, num_cus, lds);
    case 2: return launch_fused_n<2>(s, a, num_cus, lds);
    case 3: return launch_fused_n<3>(s, a, num_cus, lds);
    case 4: return launch_fused_n<4>(s, a, num_cus, lds);
    case 5: return launch_fused_n<5>(s, a, num_cus, lds);
    case 6: return launch_fused_n<6>(s, a, num_cus, lds);
    case 7: return launch_fused_n<7>(s, a, num_cus, lds);
    case 8: return launch_fused_n<8>(s, a, num_cus, lds);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_compose_write_v210(hipStream_t s, const ComposeArgs &a, uint32_t num_cus) {
  const uint32_t total = (a.out_w + 47u) / 48u * 8u * a.lines;  // quad slots by pitch
  if (!total) return hipSuccess;
  bool any_wipe = false;
  for (int l = 0; l < a.n; ++l) any_wipe = any_wipe || a.wipe_with[l] != nullptr;
  if (any_wipe && !compose_uses_taps(a)) return hipErrorInvalidValue;  // no other kernel applies the wipes: never drop them silently
  // pixel-per-lane form: needs the staging area behind the table (160 KiB of LDS per workgroup)
  const uint32_t stage_off = (a.wr.bytes + 15u) & ~15u;
  if (!compose_quad_forced() && compose_stage_fits(a) && a.out_w % 48u == 0) {  // (lines with a tail: the quad-per-lane kernel)
    const uint32_t chunks = (a.out_w * a.lines + kComposeChunk - 1) / kComposeChunk;
    const uint32_t want = (chunks + kLdsBlock / 64 - 1) / (kLdsBlock / 64);
    const uint32_t grid = want < num_cus ? want : num_cus;
    if (compose_uses_taps(a)) {
      switch (a.n) {
        case 1: return launch_compose_taps_n<1>(s, a, grid, stage_off);
        case 2: return launch_compose_taps_n<2>(s, a, grid, stage_off);
        case 3: return launch_compose_taps_n<3>(s, a, grid, stage_off);
        case 4: return launch_compose_taps_n<4>(s, a, grid, stage_off);
        case 5: return launch_compose_taps_n<5>(s, a, grid, stage_off);
        case 6: return launch_compose_taps_n<6>(s, a, grid, stage_off);
        case 7: return launch_compose_taps_n<7>(s, a, grid, stage_off);
        case 8: return launch_compose_taps_n<8>(s, a, grid, stage_off);
        default: return hipErrorInvalidValue;
      }
    }
    switch (a.n) {
      case 1: return launch_compose_px_n<1>(s, a, grid, stage_off);
      case 2: return launch_compose_px_n<2>(s, a, grid, stage_off);
      case 3: return launch_compose_px_n<3>(s, a, grid, stage_off);
      case 4: return launch_compose_px_n<4>(s, a, grid, stage_off);
      case 5: return launch_compose_px_n<5>(s, a, grid, stage_off);
      case 6: return launch_compose_px_n<6>(s, a, grid, stage_off);
      case 7: return launch_compose_px_n<7>(s, a, grid, stage_off);
      case 8: return launch_compose_px_n<8>(s, a, grid, stage_off);
      default: return hipErrorInvalidValue;
    }
  }
  hipError_t e = allow_lds(compose_write_v210_kernel, a.wr.bytes);
  if (e != hipSuccess) return e;
  const uint32_t want = (total + kLdsBlock - 1) / kLdsBlock;
  compose_write_v210_kernel<<<want < num_cus ? want : num_cus, kLdsBlock, a.wr.bytes, s>>>(a);
  return hipGetLastError();
}

// The LDS address of the dynamic shared array of a kernel that declares no other shared memory: the table lookups
// (ph_ldslut.h) take it to be 0.  ph_lut_register checks it once per context.
__global__ void lds_base_probe_kernel(uint32_t *out) {
  *out = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)g_lds;
}
hipError_t launch_lds_base_probe(hipStream_t s, uint32_t *out_dev) {
  lds_base_probe_kernel<<<1, 64, 16, s>>>(out_dev);
  return hipGetLastError();
}

hipError_t launch_v210_read_lds(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                                const void *cm, const void *gm, const LutView &lut, uint32_t num_cus) {
  hipError_t e = allow_lds(v210_read_lds_kernel, lut.bytes);
  if (e != hipSuccess) return e;
  const uint32_t total = width * height;
  const uint32_t want = (total + kLdsBlock - 1) / kLdsBlock;
  v210_read_lds_kernel<<<want < num_cus ? want : num_cus, kLdsBlock, lut.bytes, s>>>(
      (const uint4 *)in, (float4 *)out, width, v210_pitch_bytes(width) / 16, total, (const float *)cm, (const float *)gm,
      lut, image_nt((size_t)width * height * 16));
  return hipGetLastError();
}

hipError_t launch_v210_read_lds_batch(hipStream_t s, int n, const void *const *ins, void *const *outs, uint32_t width,
                                      uint32_t height, const void *cm, const void *gm, const LutView &lut, uint32_t num_cus) {
  hipError_t e = allow_lds(v210_read_lds_batch_kernel, lut.bytes);
  if (e != hipSuccess) return e;
  ReadBatchArgs a{};
  for (int i = 0; i < n; ++i) a.in[i] = (const uint4 *)ins[i], a.out[i] = (float4 *)outs[i];
  const uint32_t total = width * height;
  const uint32_t want = (total + kLdsBlock - 1) / kLdsBlock;
  uint32_t per = num_cus / (uint32_t)n ? num_cus / (uint32_t)n : 1;
  a.wg_per_frame = want < per ? want : per;
  v210_read_lds_batch_kernel<<<a.wg_per_frame * n, kLdsBlock, lut.bytes, s>>>(a, width, v210_pitch_bytes(width) / 16, total,
                                                                              (const float *)cm, (const float *)gm, lut,
                                                                              image_nt((size_t)width * height * 16));
  return hipGetLastError();
}

hipError_t launch_v210_write_lds(hipStream_t s, const void *in, void *out, uint32_t width, uint32_t height,
                                 uint32_t interlace, const void *cm, const LutView &lut, uint32_t num_cus) {
  hipError_t e = allow_lds(v210_write_lds_kernel, lut.bytes);
  if (e != hipSuccess) return e;
  const uint32_t step = interlace ? 2 : 1, first = (interlace == 3) ? 1 : 0;
  const uint32_t lines = interlace ? height / 2 : height, qpl = v210_pitch_bytes(width) / 16;  // quad slots by pitch
  if (!lines) return hipSuccess;
  const uint32_t want = (qpl * lines + kLdsBlock - 1) / kLdsBlock;
  v210_write_lds_kernel<<<want < num_cus ? want : num_cus, kLdsBlock, lut.bytes, s>>>(
      (const float4 *)in, (uint4 *)out, width, qpl, lines, first, step, (const float *)cm, lut);
  return hipGetLastError();
}

}  // namespace ph

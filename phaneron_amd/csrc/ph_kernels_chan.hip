// ph_kernels_chan.hip - a channel's whole video frame as ONE kernel, straight from the v210 sources.
//
// The reference's per-frame job batch of a channel is, per layer, ToRGBA (v210.ts:25-111) -> Mixer's
// `transform` (transform.ts:36-59) -> optionally the Transitioner's dissolve / wipe against a second source
// (transition.ts:54-79) -> and for the channel combine_N (combine.ts:45-65) -> FromRGBA (v210.ts:113-195): every arrow a
// full-size f32 RGBA frame written and read back.  ph_compose_write_v210 (ph_kernels_lds.hip) already folds everything
// from `transform` on into one kernel but still consumes f32 frames, so a 1080p channel moved ~390 MB per frame for
// 38.7 MB of v210 in and out.  Here the bilinear taps are taken from the v210 words themselves: each tap is unpacked,
// matrixed, looked up in the reader's gamma table and gamut-converted on the fly (the arithmetic of ToRGBA, per tap),
// filtered and combined in registers, and only the channel's v210 output leaves the chip.  For half-size insets and
// 1:1 sources the taps are (about) the source pixels, so hardly anything is converted twice; a full-frame layer at
// unit scale converts every source pixel four times, which is still cheaper than a round trip of f32 frames.
//
// Two phases, as in the headline kernel, because reader and writer table do not fit the LDS together:
//   phase 1 (reader table resident): pixel per lane, 64 consecutive pixels of a row per wave step; per layer the
//     placed sample (+ transition), combine_N, then the writer's first step - the 16-bit index sat_rte(rgb * 65535)
//     (v210.ts:148-150) - is parked in an INDEX FRAME (8 bytes per pixel) that lives in the XCD's L2 for the few
//     microseconds until
//   phase 2 (writer table swapped in): quad per lane, indices -> writer table -> RGB->YCbCr matrix -> packed words.
// A workgroup reads back only what it wrote itself, so one workgroup barrier (the table swap's) orders the phases.
// Results are bit-identical to running the separate kernels (tests/test_chan_gpu.py).
#include "ph_device.h"
#include "ph_kernels.h"
#include "ph_ldslut.h"

#pragma clang fp contract(off)

#ifndef PH_CHAN_GROUP_ROWS
#define PH_CHAN_GROUP_ROWS 16
#endif

namespace ph {

constexpr uint32_t kChanChunk = 192;          // pixels: 3 wave steps of phase 1, 32 quads of phase 2
constexpr uint32_t kOutsideBit = 0x40000000u;  // row / column offsets of taps outside the frame: beyond any num_records (frames < 1 GiB)

// ---- a v210 column: where pixel i's Y and its pair's Cb / Cr sit inside the 16-byte quad (v210.ts:58-63) ----------
struct V210Col {
  uint32_t off;  // byte offset of the quad inside a line, or kOutsideBit
  uint32_t yo, ys, cbo, cbs, cro, crs;
};
__device__ __forceinline__ V210Col v210_col(uint32_t i, uint32_t w) {
  const uint32_t g = __umulhi(i, 0xAAAAAAABu) >> 2;  // i / 6
  const uint32_t j = i - 6u * g, pr = j >> 1;
  V210Col c;
  c.off = i < w ? g << 4 : kOutsideBit;
  c.yo = (0xCC8440u >> (4u * j)) & 0xCu;           // Y in word {0,1,1,2,3,3}
  c.ys = 10u * ((0x201201u >> (4u * j)) & 3u);     //   at bit {10,0,20,10,0,20}
  c.cbo = 4u * pr, c.cbs = 10u * pr;               // Cb in word {0,1,2} at bit {0,10,20}
  c.cro = (0xC80u >> (4u * pr)) & 0xCu;            // Cr in word {0,2,3}
  c.crs = 10u * ((0x102u >> (4u * pr)) & 3u);      //   at bit {20,0,10}
  return c;
}

// one converted pixel of a v210 frame = what ToRGBA would have stored there: (r, g, b, 1); outside the frame the
// sampler's border colour (0, 0, 0, 0)
template <bool STD>
__device__ __forceinline__ float4 v210_texel(__amdgpu_buffer_rsrc_t frame, uint32_t row_off, const V210Col &c, const ReadK &k,
                                            const LutK &lut) {
  const uint32_t base = row_off + c.off;  // >= kOutsideBit when the row or the column is outside: the loads return 0
  const uint32_t wy = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(frame, (int)(base + c.yo), 0, 0);
  const uint32_t wcb = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(frame, (int)(base + c.cbo), 0, 0);
  const uint32_t wcr = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(frame, (int)(base + c.cro), 0, 0);
  const float y = (float)__builtin_amdgcn_ubfe(wy, c.ys, 10u);
  const float cb = (float)__builtin_amdgcn_ubfe(wcb, c.cbs, 10u);
  const float cr = (float)__builtin_amdgcn_ubfe(wcr, c.crs, 10u);
  float4 t = read_px_lds<STD>(y, cb, cr, k, lut);
  const bool in = base < kOutsideBit;
  t.x = in ? t.x : 0.0f, t.y = in ? t.y : 0.0f, t.z = in ? t.z : 0.0f, t.w = in ? 1.0f : 0.0f;
  return t;
}

__device__ __forceinline__ float4 rgba_texel(__amdgpu_buffer_rsrc_t img, uint32_t row_off, uint32_t col_off) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(img, (int)(row_off + col_off), 0, 0);  // outside: 0 = the border colour
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// the source's sample for output pixel (x, line): 1:1, or through the transform matrix and the bilinear filter
// (transform.ts:53-57; OpenCL 1.2 8.2 evaluated as DESIGN.md section 2 fixes it)
template <bool STD>
__device__ __forceinline__ float4 chan_sample(const ChanSrc &s, float px, float py, uint32_t x, uint32_t line, const ReadK &k,
                                              const LutK &lut) {
  const bool is_v210 = s.kind == kChanV210;  // uniform
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(s.ptr), 0, (int)(s.pitch * s.h), 0x00020000);
  if (!s.sampled) {  // uniform: the source has the output's size, pixel for pixel
    if (is_v210) return v210_texel<STD>(rs, line * s.pitch, v210_col(x, s.w), k, lut);
    return rgba_texel(rs, line * s.pitch, x << 4);
  }
  const float sx = dot3(s.m[0], s.m[1], s.m[2], px, py, 1.0f) + 0.5f;
  const float sy = dot3(s.m[3], s.m[4], s.m[5], px, py, 1.0f) + 0.5f;
  const float u = sx * (float)(int)s.w, v = sy * (float)(int)s.h;
  const float fu = u - 0.5f, fv = v - 0.5f;
  const float flu = __builtin_floorf(fu), flv = __builtin_floorf(fv);
  const uint32_t i0 = (uint32_t)(int)flu, i1 = i0 + 1u, j0 = (uint32_t)(int)flv, j1 = j0 + 1u;
  const float a = fu - flu, b = fv - flv;
  const bool xa = i0 < s.w, xb = i1 < s.w, ya = j0 < s.h, yb = j1 < s.h;
  float4 t00, t10, t01, t11;
  // a wave none of whose taps touches the source (the outside of a picture-in-picture inset) gets the border value
  // without a load: every product is w * 0
  if (!__builtin_amdgcn_ballot_w64((xa || xb) && (ya || yb))) return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const uint32_t r0 = ya ? j0 * s.pitch : kOutsideBit, r1 = yb ? j1 * s.pitch : kOutsideBit;
  if (is_v210) {
    const V210Col c0 = v210_col(i0, s.w), c1 = v210_col(i1, s.w);
    t00 = v210_texel<STD>(rs, r0, c0, k, lut), t10 = v210_texel<STD>(rs, r0, c1, k, lut);
    t01 = v210_texel<STD>(rs, r1, c0, k, lut), t11 = v210_texel<STD>(rs, r1, c1, k, lut);
  } else {
    const uint32_t c0 = xa ? i0 << 4 : kOutsideBit, c1 = xb ? i1 << 4 : kOutsideBit;
    t00 = rgba_texel(rs, r0, c0), t10 = rgba_texel(rs, r0, c1), t01 = rgba_texel(rs, r1, c0), t11 = rgba_texel(rs, r1, c1);
  }
  const float oma = 1.0f - a, omb = 1.0f - b;
  const float w00 = oma * omb, w10 = a * omb, w01 = oma * b, w11 = a * b;
  float4 t;
  t.x = ((w00 * t00.x + w10 * t10.x) + w01 * t01.x) + w11 * t11.x;
  t.y = ((w00 * t00.y + w10 * t10.y) + w01 * t01.y) + w11 * t11.y;
  t.z = ((w00 * t00.z + w10 * t10.z) + w01 * t01.z) + w11 * t11.z;
  t.w = ((w00 * t00.w + w10 * t10.w) + w01 * t01.w) + w11 * t11.w;
  return t;
}

// The workgroup's share of the frame: chunks of 192 pixels, dealt out XCD-aware exactly as the f32 compositor does
// (ph_kernels_lds.hip compose_taps_body): groups of PH_CHAN_GROUP_ROWS output rows belong to one XCD (blockIdx % 8), so a
// source row is pulled through one XCD's L2, and the chunks of a group go round the XCD's workgroups and waves so
// that partial-frame layers load them evenly.  slot -> chunk, or ~0u past the end.
struct ChanShare {
  uint32_t chunks, cpg, xcd, v0, vstep, vend;
  bool banded;
};
__device__ __forceinline__ ChanShare chan_share(const ChanArgs &a) {
  ChanShare s;
  s.chunks = a.out_w * a.lines / kChanChunk;  // out_w % 192 == 0: a chunk never leaves its row
  s.cpg = (uint32_t)PH_CHAN_GROUP_ROWS * (a.out_w / kChanChunk);
  s.banded = (gridDim.x & 7u) == 0;
  s.xcd = 0, s.v0 = blockIdx.x * (kLdsBlock / 64), s.vstep = gridDim.x * (kLdsBlock / 64), s.vend = s.chunks;
  if (s.banded) {
    s.xcd = blockIdx.x & 7u;
    const uint32_t groups = (s.chunks + s.cpg - 1u) / s.cpg, mine = (groups + 7u - s.xcd) / 8u;
    s.v0 = (blockIdx.x >> 3) * (kLdsBlock / 64), s.vstep = (gridDim.x >> 3) * (kLdsBlock / 64), s.vend = mine * s.cpg;
  }
  return s;
}
__device__ __forceinline__ uint32_t chan_slots(const ChanShare &s) {  // slots of this workgroup (some may be empty)
  if (s.v0 >= s.vend) return 0;
  return ((s.vend - s.v0 + s.vstep - 1u) / s.vstep) * (kLdsBlock / 64);
}
__device__ __forceinline__ uint32_t chan_chunk(const ChanShare &s, uint32_t slot) {
  const uint32_t v = s.v0 + (slot & (kLdsBlock / 64 - 1)) + (slot / (kLdsBlock / 64)) * s.vstep;
  if (v >= s.vend) return ~0u;
  if (!s.banded) return v;
  const uint32_t gi = v / s.cpg;
  const uint32_t chunk = (gi * 8u + s.xcd) * s.cpg + (v - gi * s.cpg);
  return chunk < s.chunks ? chunk : ~0u;
}

template <bool STD>
__device__ __forceinline__ void chan_phase1(const ChanArgs &a, const ChanShare &sh, uint32_t slots, const ReadK &rk, const LutK &rlut) {
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const float fow = (float)(int)a.out_w, foh = (float)(int)a.out_h;
  uint2 *const index = reinterpret_cast<uint2 *>(a.index);
  for (uint32_t n = wave; n < 3u * slots; n += kLdsBlock / 64) {  // 64-pixel steps, dealt round the waves
    const uint32_t slot = n / 3u, sub = n - 3u * slot;
    const uint32_t chunk = chan_chunk(sh, slot);
    if (chunk == ~0u) continue;  // uniform
    const uint32_t base = chunk * kChanChunk + sub * 64u;
    const uint32_t li = base / a.out_w, x = base - li * a.out_w + lane;
    const uint32_t line = a.first_line + li * a.line_step;
    const float py = (float)(int)line / foh - 0.5f;  // transform.ts:53
    const float px = (float)(int)x / fow - 0.5f;
    float r = 0.0f, g = 0.0f, b = 0.0f;
#pragma unroll 1  // one copy of the sampling code whatever the layer count; the layer's parameters are scalar loads from the arguments
    for (int l = 0; l < a.n; ++l) {
      const ChanLayer &L = a.layer[l];
      float4 t = chan_sample<STD>(L.src, px, py, x, line, rk, rlut);
      if (L.transition != kChanCut) {  // uniform; transition.ts:54-79 as the Transitioner runs it
        const float4 in1 = chan_sample<STD>(L.incoming, px, py, x, line, rk, rlut);
        if (L.transition == kChanDissolve) {  // fma(in0, mix, in1 * (1 - mix))
          const float m = L.mix, rm = 1.0f - m;
          t.x = fma_rn(t.x, m, in1.x * rm), t.y = fma_rn(t.y, m, in1.y * rm), t.z = fma_rn(t.z, m, in1.z * rm), t.w = fma_rn(t.w, m, in1.w * rm);
        } else {  // wipe: fma(in1, mask.r, in0 * (1 - mask.r))
          const float m = chan_sample<STD>(L.mask, px, py, x, line, rk, rlut).x, rm = 1.0f - m;
          t.x = fma_rn(in1.x, m, t.x * rm), t.y = fma_rn(in1.y, m, t.y * rm), t.z = fma_rn(in1.z, m, t.z * rm), t.w = fma_rn(in1.w, m, t.w * rm);
        }
      }
      if (l == 0) {
        r = t.x, g = t.y, b = t.z;
      } else {  // combine.ts:45-65 (the result's alpha is never used by the writer)
        const float kk = 1.0f - t.w;
        r = fma_rn(r, kk, t.x), g = fma_rn(g, kk, t.y), b = fma_rn(b, kk, t.z);
      }
    }
    // the writer's first step needs no table (v210.ts:148-150): park the three 16-bit indices
    const uint32_t ir = __float_as_uint(lds_lut_index_unit(r)) & 0xFFFFu, ig = __float_as_uint(lds_lut_index_unit(g)) & 0xFFFFu;
    const uint32_t ib = __float_as_uint(lds_lut_index_unit(b)) & 0xFFFFu;
    index[base + lane] = make_uint2(ir | (ig << 16), ib);
  }
}

__global__ __launch_bounds__(kLdsBlock) void chan_compose_v210_kernel(ChanArgs a) {
  const ReadK rk = load_read_k(a.rd_cm, a.rd_gm);
  const WriteK wk = load_write_k(a.wr_cm);
  const LutK rlut = make_lut_k(a.rd), wlut = make_lut_k(a.wr);
  const ChanShare sh = chan_share(a);
  const uint32_t slots = chan_slots(sh);
  lds_lut_load(a.rd);
  __syncthreads();
  if (ycbcr_matrix_is_standard(rk)) chan_phase1<true>(a, sh, slots, rk, rlut);
  else chan_phase1<false>(a, sh, slots, rk, rlut);
  __syncthreads();  // every index of this workgroup has been stored (the barrier drains the stores) and nobody reads the reader table any more
  lds_lut_load(a.wr);
  __syncthreads();
  // phase 2: one quad per lane.  The indices were written by other waves of THIS workgroup: read past the L1.
  const uint32_t qpl = a.out_w / 6;
  const uint4 *const index = reinterpret_cast<const uint4 *>(a.index);
  for (uint32_t q = threadIdx.x; q < 32u * slots; q += kLdsBlock) {
    const uint32_t chunk = chan_chunk(sh, q >> 5);
    if (chunk == ~0u) continue;
    const uint32_t first_px = chunk * kChanChunk + (q & 31u) * 6u;
    const uint4 w0 = load_stream(index + (first_px >> 1)), w1 = load_stream(index + (first_px >> 1) + 1), w2 = load_stream(index + (first_px >> 1) + 2);
    const uint32_t pk[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
    float yi[18];
#pragma unroll
    for (int j = 0; j < 6; ++j) {  // M + idx: the index ORed into the mantissa of 1.5 * 2^23 (ph_ldslut.h)
      yi[3 * j] = __uint_as_float((pk[2 * j] & 0xFFFFu) | 0x4B400000u);
      yi[3 * j + 1] = __uint_as_float((pk[2 * j] >> 16) | 0x4B400000u);
      yi[3 * j + 2] = __uint_as_float((pk[2 * j + 1] & 0xFFFFu) | 0x4B400000u);
    }
    const uint32_t li = first_px / a.out_w, x = first_px - li * a.out_w;
    const uint32_t line = a.first_line + li * a.line_step;
    store_stream(reinterpret_cast<uint4 *>(a.out) + (size_t)line * qpl + x / 6, write_quad_idx_lds(yi, wk, wlut));
  }
}

size_t chan_index_bytes(uint32_t out_w, uint32_t lines) { return (size_t)out_w * lines * 8u; }

hipError_t launch_chan_compose_v210(hipStream_t s, const ChanArgs &a, uint32_t num_cus) {
  if (!a.lines) return hipSuccess;
  const uint32_t lds = a.rd.bytes > a.wr.bytes ? a.rd.bytes : a.wr.bytes;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(chan_compose_v210_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const uint32_t chunks = a.out_w * a.lines / kChanChunk;
  const uint32_t want = (chunks + kLdsBlock / 64 - 1) / (kLdsBlock / 64);
  chan_compose_v210_kernel<<<want < num_cus ? want : num_cus, kLdsBlock, lds, s>>>(a);
  return hipGetLastError();
}

}  // namespace ph

// xmaps_k2pipe.hpp -- K2 (dilate o remap -> depth -> u8 -> Turbo BGR, disp_to_depth.py:7-97) for GROUPS of frames on the plain
// u16 disparity frame of the column / owner tiles, as persistent, software-pipelined blocks.  (gfx950 / MI355X; included by
// xmaps_hip.hip after xmaps_kernels.hpp)
//
// k_frame_proj_tiled_batch is one block per (tile, frame): a chain of dependent round trips -- frame descriptor + tile record ->
// the tile's patch of the disparity frame -> (LDS work) -> the per-disparity table -> stores -- with 55 % of the block's lifetime
// spent before the patch has arrived (tools/k2_timeline.py), at the CU's full complement of 32 waves: more waves cannot hide it.
// Here a block walks a strided sequence of (frame, tile) items and keeps the NEXT item's patch in flight in registers while it
// works on the current one:
//   * the loads of item i + 1 (its 16-byte patch quads and the pixels' patch offsets) are issued right after item i's patch has
//     been written to LDS, and nothing else of the iteration is a vector memory load: the per-disparity table {depth, BGR}
//     (k_build_dlut) is copied into LDS once per block (its first n_lds entries; larger disparities -- rare -- read the global
//     table, behind a branch), so no wait of the iteration has to drain the prefetch;
//   * the tile's patch is written to LDS, reduced (7-tap maxima along the rows, in place) and sampled exactly as before
//     (frame_proj_tiled_body, FMT = 2): same tables (k2_tiles / k2_pix), same arithmetic, same outputs;
//   * patches that stick out of the frame load zeros for the octets outside (patch rows start on a multiple of 8 and so does
//     the frame's height: an octet is inside or outside as a whole); a rig with a patch of more than 128 rows or more quads than
//     the loader's registers hold keeps the one-block-per-tile kernel (k2_pipe_tile_ok, checked once in xm_create).
#pragma once
#include "xmaps_kernels.hpp"

namespace xm {

// 16-byte patch loads per thread kept in registers: thread slot s = tid + j * 256 holds quad s of the patch in memory order
// (column s / oct, row octet s % oct, oct = rows / 8), so a patch of q quads needs ceil(q / 256) of them: four cover 8192 cells
constexpr int K2P_UN = 4;

// Can every tile of the rig take the pipelined kernel?  (decided once in xm_create from the tile table; rows a multiple of 8 --
// then every 8-row octet of a patch lies entirely inside or outside the frame -- and at most 128)
__host__ __device__ inline bool k2_pipe_tile_ok(const int4& rec) {
  return rec.z >= 0 && rec.w >= 0 && rec.w <= 128 && (rec.w & 7) == 0 && rec.z * (rec.w >> 3) <= K2P_UN * K2_TX * K2_TY;
}

// Barrier for LDS hand-offs only: __syncthreads() carries a workgroup-scope fence, for which the compiler drains EVERY outstanding
// memory operation (s_waitcnt vmcnt(0)) -- including the next item's patch loads that are meant to stay in flight across it.
// This one waits for the wave's LDS operations and joins the barrier; the prefetched registers are waited for where they are used.
__device__ __forceinline__ void k2p_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the output frames are written once and not read again by any kernel: XM_K2P_NT = 1 marks their stores non-temporal (experiment)
#ifndef XM_K2P_NT
#define XM_K2P_NT 0
#endif
typedef u32 k2p_u32x4 __attribute__((ext_vector_type(4)));
#if XM_K2P_NT
#define k2p_store(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define k2p_store(ptr, val) (*(ptr) = (val))  /* (a macro: a template parameter would drop the pointee's 4-byte alignment) */
#endif

#ifdef XM_ABLATE  // experiments (tools/k2p_phases.py): per block, the cycles (s_memtime) between the loop's phase marks, summed over its items
#define XM_K2P_MARK(ph) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph_sum[ph] += now_ - ph_t; ph_t = now_; } while (0)
#else
#define XM_K2P_MARK(ph) do { } while (0)
#endif

struct K2PipeArgs {  // (only what the loop needs: the whole DevTables would sit in scalar registers across it)
  int proj_w, proj_h, rect_w, rect_h, shear_m, shear_bias;
};

#ifndef XM_K2P_EMU_COMPACT
#define XM_K2P_EMU_COMPACT 0
#endif
#ifndef XM_K2P_STAGED
#define XM_K2P_STAGED 0  /* experiments: 1 = the BGR rows of four-pixel threads through the LDS staging rows, as until round 5 */
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define XM_K2P_GLOBAL __attribute__((address_space(1)))
#else
#define XM_K2P_GLOBAL
#endif

// One-off (xm_create): the pixel table as u16 (patch offsets are < K2_TILE_MAX; ~0 -> 0xffff), rows padded to `stride` entries
__global__ __launch_bounds__(BLOCK) void k_k2_pix_to_u16(const u32* __restrict__ pix, uint16_t* __restrict__ out, int proj_w, int proj_h, int stride) {
  const u32 i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= (u32)stride * (u32)proj_h) return;
  const u32 v = i / (u32)stride, u = i - v * (u32)stride;
  u32 o = 0xffffu;
  if (u < (u32)proj_w) {
    const u32 p = pix[v * (u32)proj_w + u];
    if (p < 0xffffu) o = p;
  }
  out[i] = (uint16_t)o;
}

// CONSEC: thread (tx, ty) takes the PPT CONSECUTIVE pixels tx * PPT .. tx * PPT + PPT - 1 of its row (instead of tx + j * K2_TX): the
// pixels' patch offsets come as ONE load from a u16 copy of the pixel table (k2_pix16, row stride pix_stride, 0xffff = no
// target), the depths leave as ONE 8 / 16-byte store, the BGR bytes are packed into dwords before they are staged, and the
// staged rows leave as 8 / 16-byte stores where the output's alignment allows: per item and wave 4 + 7 vector memory
// instructions (PPT = 4) become 1 + 2.
template <int PPT, bool CONSEC = false, int COND = 0>
__global__ __launch_bounds__(K2_TX* K2_TY) void k_frame_proj_pipe(const FrameDesc* __restrict__ descs, const int4* __restrict__ k2_tiles,
                                                                const u32* __restrict__ k2_pix, const uint16_t* __restrict__ k2_pix16,
                                                                int pix_stride, const uint2* __restrict__ dlut,
                                                                K2PipeArgs a, int tile_cap, u32 n_frames, u32 grid_x, u32 grid_y,
                                                                int n_lds, u32 gx_magic) {
  // (the tables are kernel parameters of their own, __restrict__: block-uniform reads of them become scalar loads -- as members
  //  of a struct they were vector loads, each with a full wait in front of the prefetch.  Pointers read from a frame descriptor
  //  are cast to the global address space: generic ones make FLAT loads / stores, which count against the LDS counter too, and
  //  every LDS wait would drain the prefetch.)
  constexpr int NT = K2_TX * K2_TY, K2_TW = K2_TX * PPT;
  extern __shared__ __attribute__((aligned(16))) uint16_t k2_lds[];
  uint16_t* tile = k2_lds;                                                         // [tile_cap + 32]
  uint2* s_dlut = reinterpret_cast<uint2*>(k2_lds + ((tile_cap + 32 + 7) & ~7));  // [n_lds] {f32 depth bits, BGR word}
  __shared__ __attribute__((aligned(16))) uint8_t s_out[K2_TY][K2_TW * 3];
  const int tid = threadIdx.x, tx = tid & (K2_TX - 1), ty = tid / K2_TX;
  const u32 tpf = grid_x * grid_y;
  for (int i = tid; i < n_lds; i += NT) s_dlut[i] = dlut[i];  // (n_lds covers every disparity of the rig; visible after the first barrier)

  // ---- item = (frame, block-linear tile index); a block strides through them.  Per iteration: reduce + sample item 0's patch
  //      (LDS), then move item 1's loads (in flight since the end of the previous iteration) from registers to LDS, then write
  //      item 0's outputs, then issue item 2's loads.  The memory counter is in order and the compiler waits for "everything"
  //      where item 1's registers are consumed: at that point the youngest outstanding operations are the previous iteration's
  //      last loads and stores, one reduce + sample phase old -- nothing that was issued just now.
  struct Meta {
    int4 rec;
    u32 f, lin, tile_x, tile_y;
    bool run;
  };
  const auto meta_at = [&](u32 f, u32 b) {
    Meta m;
    m.f = f;
    m.run = f < n_frames;
    m.lin = xcd_contiguous(b, tpf);
    // (tile_y, tile_x) = divmod(lin, grid_x) once per item, by a multiply: gx_magic = ceil(2^32 / grid_x) is exact for
    // lin * grid_x < 2^32 (the host passes 0 otherwise)
    m.tile_y = gx_magic ? __umulhi(m.lin, gx_magic) : m.lin / grid_x;
    m.tile_x = m.lin - m.tile_y * grid_x;
    m.rec = make_int4(0, 0, 0, 0);
    if (m.run) {
      m.rec = k2_tiles[m.lin];
      m.run = descs[f].valid != 0 && !frame_skipped<COND>(descs[f].st);  // (COND = 2, captured batches: only frames whose attempt held)
    }
    return m;
  };
  const auto advance = [&](u32& f, u32& b) {
    b += gridDim.x;
    while (b >= tpf) {
      b -= tpf;
      f += 1;
    }
  };
  constexpr int UN = K2P_UN;
  uint4 K[UN];
  // the item's vector loads: patch quads (thread slot s -> column s / oct, row octet s % oct: the patch in memory order; an octet
  // outside the frame is not loaded: it reads as zeros) and the pixels' offsets into the patch
  constexpr int NP = CONSEC ? PPT / 2 : PPT;  // registers that hold a thread's patch offsets (CONSEC: u16 pairs)
  const auto issue = [&](const Meta& m, u32 (&poff)[NP]) {
    const u32 tile_y = m.tile_y, tile_x = m.tile_x;
    const int v = tile_y * K2_TY + ty;
    if constexpr (CONSEC) {  // (poff[] holds the PPT u16 offsets packed in pairs; the rest of it stays ~0)
      const int u0 = tile_x * K2_TW + tx * PPT;
      const bool in_tab = u0 < pix_stride && v < a.proj_h;  // (pix_stride % 4 == 0: the thread's run lies inside the row or outside)
      const XM_K2P_GLOBAL uint16_t* src = (const XM_K2P_GLOBAL uint16_t*)k2_pix16 + (__umul24((u32)v, (u32)pix_stride) + (u32)u0);
      if constexpr (PPT == 4) {
        uint2 w = make_uint2(~0u, ~0u);
        if (XM_CABL(12)) w = make_uint2(0x00210001u + (u32)tx * 4u, 0x00610041u + (u32)tx * 4u);  // (experiments: no offset-table load)
        else if (in_tab) w = *reinterpret_cast<const XM_K2P_GLOBAL uint2*>(src);
        poff[0] = w.x;
        poff[1] = w.y;
      } else {
        static_assert(PPT == 2, "two or four pixels per thread");
        u32 w = ~0u;
        if (in_tab) w = *reinterpret_cast<const XM_K2P_GLOBAL u32*>(src);
        poff[0] = w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int u = tile_x * K2_TW + tx + j * K2_TX;
        const bool in_img = u < a.proj_w && v < a.proj_h;
        poff[j] = in_img ? k2_pix[__umul24((u32)v, (u32)a.proj_w) + (u32)u] : ~0u;
      }
    }
    const XM_K2P_GLOBAL uint16_t* d16 = (const XM_K2P_GLOBAL uint16_t*)descs[m.f].key_frame;
    const int bx = m.rec.x, by = m.rec.y, oct = m.rec.w >> 3, nslot = __mul24(m.rec.z, oct);
    const int g0 = by >> 3;
    // s / oct in float: (s + 0.5) / oct is at least 1 / 32 away from an integer and s < 2^11, oct <= 16: the rounding cannot reach it
    const float inv_oct = __frcp_rn((float)max(oct, 1));
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int sj = tid + j * NT, c = (int)(((float)sj + 0.5f) * inv_oct), ro = sj - __mul24(c, oct);
      const int gx = bx + c, gy = by + 8 * ro;
      const bool has = sj < nslot && (u32)gx < (u32)a.rect_w && (u32)gy < (u32)a.rect_h;  // (rect_h % 8 == 0)
      K[j] = make_uint4(0, 0, 0, 0);
      if (XM_K2P_EMU_COMPACT || XM_CABL(21)) {  // (experiments, bit 21 / -DXM_K2P_EMU_COMPACT=1: the traffic of a COMPACT frame -- 41 % of the quads, contiguous per tile, 1.35 x overlap)
        const int live = (nslot * 105) >> 8, per_tile = (live * 190) >> 8;
        if (sj < live) K[j] = *reinterpret_cast<const XM_K2P_GLOBAL uint4*>(d16 + (size_t)m.lin * (size_t)per_tile * 8u + (size_t)sj * 8u);
      } else
      if (has && !XM_CABL(13))  // (experiments, bit 13: no patch loads)
        K[j] = *reinterpret_cast<const XM_K2P_GLOBAL uint4*>(d16 + __umul24((u32)(gx + a.shear_bias + (((g0 + ro) * a.shear_m) >> 12)), (u32)a.rect_h) + (u32)gy);
    }
  };
  const auto to_lds = [&](const Meta& m) {
    const int nslot = __mul24(m.rec.z, m.rec.w >> 3);
#pragma unroll
    for (int j = 0; j < UN; ++j) {
      const int sj = tid + j * NT;
      if (sj < nslot) reinterpret_cast<uint4*>(tile)[sj] = K[j];
    }
  };

  u32 f = 0, b = blockIdx.x;
  while (b >= tpf && f < n_frames) {
    b -= tpf;
    f += 1;
  }
  if (f >= n_frames) return;
  u32 p0[NP], p1[NP];
  Meta m0 = meta_at(f, b), m1;
#pragma unroll
  for (int q = 0; q < NP; ++q) p0[q] = p1[q] = ~0u;
  if (m0.run) {
    issue(m0, p1);
    to_lds(m0);
  }
#pragma unroll
  for (int q = 0; q < NP; ++q) p0[q] = p1[q];
  advance(f, b);
  m1 = meta_at(f, b);
  if (m1.run) issue(m1, p1);
  k2p_lds_barrier();
#ifdef XM_ABLATE
  unsigned long long ph_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_amdgcn_s_memtime(), ph_items = 0;
#endif
  for (;;) {
#ifdef XM_ABLATE
    ph_items += 1;
#endif
    uint2 e[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) e[q] = make_uint2(0u, 0u);
    if (m0.run) {
      const int cols = m0.rec.z, rows_p = m0.rec.w;
      if (!XM_CABL(19))  // (experiments, bit 19: no row-maxima pass)
      {  // 7-tap max along the rows of every patch column, in place (see frame_proj_tiled_body)
        const int nseg = rows_p >> 3, tasks = cols * nseg;
        constexpr int CH = 4;
        for (int t0 = 0; t0 < tasks; t0 += CH * NT) {
          uint4 w[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const int t = t0 + j * NT + tid;
            if (t < tasks)
              w[j] = k2_rowmax8(*reinterpret_cast<const uint4*>(tile + t * 8), *reinterpret_cast<const uint4*>(tile + t * 8 + 8));
          }
          k2p_lds_barrier();
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            const int t = t0 + j * NT + tid;
            if (t < tasks) *reinterpret_cast<uint4*>(tile + t * 8) = w[j];
          }
        }
      }
      k2p_lds_barrier();
      XM_K2P_MARK(0);  // the row maxima pass (its barriers included)
#pragma unroll
      for (int q = 0; q < PPT; ++q) {
        u32 best = 0;
        const u32 off_q = CONSEC ? (p0[q >> 1] >> ((q & 1) * 16)) & 0xffffu : p0[q];
        if (off_q != (CONSEC ? 0xffffu : ~0u)) {
          const uint16_t* p = tile + off_q;
          if (XM_CABL(20)) {  // (experiments, bit 20: one tap instead of seven)
            best = (u32)p[3 * rows_p];
          } else {
#pragma unroll
            for (int j = 0; j < 7; ++j) best = max(best, (u32)p[j * rows_p]);
          }
        }
        if (best < (u32)n_lds) e[q] = s_dlut[best];
        else e[q] = ((const XM_K2P_GLOBAL uint2*)dlut)[min(best, 65535u)];  // (a disparity beyond the LDS copy: x noise far off the scan; the wait drains the prefetch, rarely)
      }
    }
    XM_K2P_MARK(1);     // taps + table look-ups (s_memtime waits for the LDS counter: the values are there)
    k2p_lds_barrier();  // every pixel has sampled the patch: the next one may take its place
    XM_K2P_MARK(2);     // the barrier behind the sampling
#pragma unroll
    for (int q = 0; q < NP; ++q) {  // (pinned HERE: sunk behind the stores below, the copy would wait for them -- the counter is in order)
      p0[q] = p1[q];
      asm volatile("" : "+v"(p0[q]));
    }
    if (m1.run) to_lds(m1);
    XM_K2P_MARK(3);  // the wait for the next item's patch (vmcnt) + its LDS writes
    if (m0.run) {  // item 0's outputs
      const FrameDesc& d = descs[m0.f];
      XM_K2P_GLOBAL float* depth = (XM_K2P_GLOBAL float*)d.depth;
      XM_K2P_GLOBAL uint8_t* bgr = (XM_K2P_GLOBAL uint8_t*)d.bgr;
      const u32 tile_y = m0.tile_y, tile_x = m0.tile_x;
      const int v = tile_y * K2_TY + ty;
      if (m0.lin == 0 && tid < CNT_SLOTS) {  // re-arm the frame's next counters (as frame_proj_tiled_body)
        XM_K2P_GLOBAL SlotState* st = (XM_K2P_GLOBAL SlotState*)d.st;
        const u32 tag = st->tag_a;
        XM_K2P_GLOBAL u32* c = st->cnt[(tag & 1) ^ 1][tid];
        c[0] = c[1] = c[2] = c[3] = 0;
        if (tid == 0) {
          st->tag_b = tag;
          if (u32* hf = st->host_flags) host_flag_store(hf + 1, tag);
        }
      }
      if constexpr (CONSEC) {
        const int u0 = tile_x * K2_TW + tx * PPT;
        const bool vec_ok = (a.proj_w & (PPT - 1)) == 0;  // then the thread's run lies inside the row or outside, and is aligned
        if (depth && v < a.proj_h) {
          XM_K2P_GLOBAL float* dp = depth + (__umul24((u32)v, (u32)a.proj_w) + (u32)u0);
          if (vec_ok) {
            if (u0 < a.proj_w) {
              if constexpr (PPT >= 4) {
#pragma unroll
                for (int q = 0; q < PPT; q += 4)
                  if (!XM_CABL(18) || e[q].x == 0x12345678u)  // (experiments, bit 18: no output stores)
                  k2p_store(reinterpret_cast<XM_K2P_GLOBAL k2p_u32x4*>(dp) + (q >> 2), (k2p_u32x4{e[q].x, e[q + 1].x, e[q + 2].x, e[q + 3].x}));
              } else {
                *reinterpret_cast<XM_K2P_GLOBAL uint2*>(dp) = make_uint2(e[0].x, e[1].x);
              }
            }
          } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q)
              if (u0 + q < a.proj_w) dp[q] = __uint_as_float(e[q].x);
          }
        }
        bool bgr_direct = false;
        if constexpr (PPT == 4 && !XM_K2P_STAGED) bgr_direct = bgr && (a.proj_w & 3) == 0 && ((u32)(size_t)bgr & 3u) == 0;
        if (bgr_direct) {
         if constexpr (PPT == 4) {
          // Four consecutive pixels = 12 bytes = three dwords per thread, a row's threads back to back: ONE global_store_dwordx3
          // per thread (4-byte aligned: u0 and the frame's row length are multiples of four pixels), the wave's lanes = runs of
          // K2_TX * 12 contiguous bytes.  No staging through LDS, no barrier for it (round 6: the staged rows cost three LDS
          // writes, a barrier and the row copy per item -- on the 1080 x 1920 projector the BGR frame is 6.2 of the 14.5 MB a
          // frame writes and was a third of K2's time).
          if (u0 < a.proj_w && v < a.proj_h) {
            typedef u32 u32x3 __attribute__((ext_vector_type(3)));
            typedef u32x3 __attribute__((aligned(4))) u32x3_a4;
            u32x3_a4 w;
            w.x = (e[0].y & 0xffffffu) | (e[1].y << 24);
            w.y = ((e[1].y >> 8) & 0xffffu) | (e[2].y << 16);
            w.z = ((e[2].y >> 16) & 0xffu) | (e[3].y << 8);
            if (!XM_CABL(18) || w.x == 0x12345678u)
            k2p_store(reinterpret_cast<XM_K2P_GLOBAL u32x3_a4*>(bgr + (size_t)(__umul24((u32)v, (u32)a.proj_w) + (u32)u0) * 3u), w);
          }
         }
        } else if (bgr) {
          // the row's bytes of this tile: valid_b of them inside the image.  Vector stores of VB bytes when the frame's rows, the
          // tile's first byte and the valid run are all multiples of VB (VB = 16, 8 or 4)
          const int px_in = min((int)K2_TW, a.proj_w - (int)(tile_x * K2_TW));
          const u32 valid_b = (u32)px_in * 3u, row_b = (u32)a.proj_w * 3u;
          const u32 algn = (u32)(size_t)bgr | row_b | valid_b;
          {  // 3 bytes per pixel, packed: PPT = 8 / 4: 24 / 12 bytes = 6 / 3 dwords, PPT = 2: 6 bytes
            if constexpr (PPT >= 4) {  // byte n of the thread's run = byte n % 3 of pixel n / 3: PPT * 3 / 4 whole dwords
              u32* so = reinterpret_cast<u32*>(&s_out[ty][tx * PPT * 3]);
#pragma unroll
              for (int d = 0; d < PPT * 3 / 4; ++d) {
                u32 w = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                  const int n = 4 * d + b;
                  w |= ((e[n / 3].y >> (8 * (n % 3))) & 0xffu) << (8 * b);
                }
                so[d] = w;
              }
            } else {
              uint16_t* sh = reinterpret_cast<uint16_t*>(&s_out[ty][tx * 6]);
              const u32 w0 = (e[0].y & 0xffffffu) | (e[1].y << 24);
              sh[0] = (uint16_t)w0;
              sh[1] = (uint16_t)(w0 >> 16);
              sh[2] = (uint16_t)((e[1].y >> 8) & 0xffffu);
            }
          }
          k2p_lds_barrier();
          const u32 row0 = (__umul24((u32)(tile_y * K2_TY), (u32)a.proj_w) + tile_x * K2_TW) * 3u;  // byte offset of the tile's first row
          if ((algn & 15u) == 0) {
            constexpr int PER = K2_TW * 3 / 16;  // 16-byte words per staged row
            for (int i = tid; i < K2_TY * PER; i += NT) {
              const int r = i / PER, qq = i - r * PER;
              if ((int)(tile_y * K2_TY) + r < a.proj_h && (u32)qq * 16u < valid_b)
                *reinterpret_cast<XM_K2P_GLOBAL uint4*>(bgr + (size_t)(row0 + (u32)r * row_b + (u32)qq * 16u)) = reinterpret_cast<const uint4*>(&s_out[r][0])[qq];
            }
          } else if ((algn & 7u) == 0) {
            constexpr int PER = K2_TW * 3 / 8;
            for (int i = tid; i < K2_TY * PER; i += NT) {
              const int r = i / PER, qq = i - r * PER;
              if ((int)(tile_y * K2_TY) + r < a.proj_h && (u32)qq * 8u < valid_b)
                *reinterpret_cast<XM_K2P_GLOBAL uint2*>(bgr + (size_t)(row0 + (u32)r * row_b + (u32)qq * 8u)) = reinterpret_cast<const uint2*>(&s_out[r][0])[qq];
            }
          } else if ((algn & 3u) == 0) {
            constexpr int PER = K2_TW * 3 / 4;
            for (int i = tid; i < K2_TY * PER; i += NT) {
              const int r = i / PER, qq = i - r * PER;
              if ((int)(tile_y * K2_TY) + r < a.proj_h && (u32)qq * 4u < valid_b)
                *reinterpret_cast<XM_K2P_GLOBAL u32*>(bgr + (size_t)(row0 + (u32)r * row_b + (u32)qq * 4u)) = reinterpret_cast<const u32*>(&s_out[r][0])[qq];
            }
          } else {
            for (int i = tid; i < K2_TY * (int)(K2_TW * 3); i += NT) {
              const int r = i / (int)(K2_TW * 3), qq = i - r * (int)(K2_TW * 3);
              if ((int)(tile_y * K2_TY) + r < a.proj_h && (u32)qq < valid_b) bgr[(size_t)(row0 + (u32)r * row_b + (u32)qq)] = s_out[r][qq];
            }
          }
        }
      } else {
        if (depth) {
#pragma unroll
          for (int q = 0; q < PPT; ++q) {
            const int u = tile_x * K2_TW + tx + q * K2_TX;
            if (u < a.proj_w && v < a.proj_h) depth[__umul24((u32)v, (u32)a.proj_w) + (u32)u] = __uint_as_float(e[q].x);
          }
        }
        if (bgr) {
          const bool full_rows = (a.proj_w & 3) == 0 && (tile_x + 1) * K2_TW <= (u32)a.proj_w;
          if (full_rows) {
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
              s_out[ty][(tx + q * K2_TX) * 3 + 0] = (uint8_t)(e[q].y & 0xff);
              s_out[ty][(tx + q * K2_TX) * 3 + 1] = (uint8_t)((e[q].y >> 8) & 0xff);
              s_out[ty][(tx + q * K2_TX) * 3 + 2] = (uint8_t)((e[q].y >> 16) & 0xff);
            }
            k2p_lds_barrier();
            constexpr int DW = K2_TW * 3 / 4;
#pragma unroll
            for (int i0 = 0; i0 < K2_TY * DW; i0 += NT) {
              const int i = i0 + tid;
              if (i < K2_TY * DW) {
                const int r = i / DW, qq = i - r * DW, vv = tile_y * K2_TY + r;
                if (vv < a.proj_h)
                  reinterpret_cast<XM_K2P_GLOBAL u32*>(bgr + (size_t)((__umul24((u32)vv, (u32)a.proj_w) + tile_x * K2_TW) * 3u))[qq] =
                      reinterpret_cast<const u32*>(&s_out[r][0])[qq];
              }
            }
          } else {
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
              const int u = tile_x * K2_TW + tx + q * K2_TX;
              if (u < a.proj_w && v < a.proj_h) {
                XM_K2P_GLOBAL uint8_t* bp = bgr + (u64)(__umul24((u32)v, (u32)a.proj_w) + (u32)u) * 3;
                bp[0] = (uint8_t)(e[q].y & 0xff);
                bp[1] = (uint8_t)((e[q].y >> 8) & 0xff);
                bp[2] = (uint8_t)((e[q].y >> 16) & 0xff);
              }
            }
          }
        }
      }
    }
    XM_K2P_MARK(4);  // the output stores (issued)
    if (m1.f >= n_frames) break;  // (items are visited in order: nothing behind an item past the end)
    m0 = m1;
    advance(f, b);
    m1 = meta_at(f, b);
    XM_K2P_MARK(5);             // the next item's tile record + frame descriptor (scalar loads)
    if (m1.run) issue(m1, p1);  // in flight during the next item's reduce + sample phase
    XM_K2P_MARK(6);             // its loads issued
    k2p_lds_barrier();          // the next patch is in LDS, the staging rows are free again
    XM_K2P_MARK(7);
  }
#ifdef XM_ABLATE
  if (tid == 0 && blockIdx.x < 64) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_timeline[blockIdx.x][k] = ph_sum[k];
    g_timeline[blockIdx.x][8] = ph_items;
  }
#endif
}

}  // namespace xm

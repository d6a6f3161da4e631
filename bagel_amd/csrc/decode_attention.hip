// One-token attention for the text decoding loop (gfx950): q/k RMSNorm + RoPE, the write of the new key and value
// into the KV workspace, and attention over [context | generated so far], in ONE launch.
//
// Replaces, for rows that are each one sample's single new token, the per-step work of
// PackedAttentionMoT.forward_inference (qwen2_navit.py:549-599) that Bagel.generate_text drives once per token
// (bagel.py:1551-1587).  The general path spends three launches on it (norm + RoPE, value scatter, the tiled
// attention kernel whose 128-row query tiles carry one live row); a decode step is latency, not throughput, so:
//
//   grid = (key chunks, kv heads, samples), 1024 threads.  A workgroup takes one sample, one kv head with the REP
//   query heads that share it, and one chunk of that sample's keys (256 keys, more when the sample has over 16384).
//   Its four groups of four waves take one 64-key tile each per round, each with its own online softmax, and are
//   merged through LDS.  The first two rounds' keys and values are requested before anything else, so the norm + RoPE of the
//   queries (a few hundred flops, recomputed by every workgroup) runs under that latency.  The workgroup whose
//   chunk holds the new token's slot writes the new key / value to the workspace for later steps and patches them
//   into its own registers from LDS, so nothing is read back from memory inside the launch.
//   A sample of more than one chunk is split over workgroups: every workgroup leaves an unnormalised partial (max,
//   sum, REP x D accumulators, fp32) and the last one of a (sample, kv head) to finish - found with one atomic
//   ticket - merges them.  No workgroup ever waits for another.
//   (Measured inside the 7B decode step at 400-700 keys: 12.5 us against 22.9 us for the three launches; one
//   workgroup per kv head walking all tiles took 15-21 us, a fence in every wave 27 us.)
//
// Products are plain VALU dot products (v_dot2_f32_bf16): one query row per head gives the matrix cores nothing to
// do.  Arithmetic follows the tiled kernel: fp32 scores and statistics, P rounded to bf16 for P.V, fp32
// accumulation, one division at the end.
#include "common.h"
#include <algorithm>
#include "qk_rope.h"
#include "../../include/bagel_hip.h"

namespace {

struct DecTile {   // same 32-byte record as the tiled kernel's table
  int q_row0, n_rows, kv_start, kv_len, causal_base, pad[3];
};

constexpr int MAX_SPLITS = 64;   // key chunks per sample (grid.x)
constexpr int MAX_TICKETS = 256;  // samples x kv heads per launch (64 samples at four kv heads)
constexpr int TICKET_BYTES = MAX_TICKETS * 4;
constexpr int TK = 64;           // keys per tile

struct DecParams {
  const bf16* qkv;   // [M][(Hq + 2 Hkv) D]  fused projection output, bias added
  long ld;
  const bf16 *q_w, *k_w;         // per-head norm weights [D]
  const bf16 *cos_t, *sin_t;     // RoPE rows [*, D]
  const int *row_cs, *row_slot;  // per row: RoPE row, key slot of the new token
  const DecTile* tiles;          // tiles[m]: key range of row m
  bf16 *K, *Vt, *O;
  float* part;                   // [M][Hkv][MAX_SPLITS][REP][D + 2]
  unsigned* tickets;             // [M][Hkv], zero on entry, left zero
  long S;
  int Hq, Hkv;
  float eps, scale_log2;
  int und_rounding;
  int min_chunk;                 // keys per workgroup before a sample is split (a multiple of 256)
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

constexpr int GROUPS = 4;        // groups of 256 threads per workgroup, one tile each per round

template <int D, int REP>
__global__ __launch_bounds__(1024) void decode_attention_kernel(DecParams p) {
  constexpr int LPH = D / 8;              // lanes per head in the norm + RoPE step
  // keys: 8 lanes per key row, lane c8 holds 16-byte chunks c8, c8 + 8, ... of 2 keys (a wave-wide load covers whole
  // 128-byte lines); values: 8 lanes per Vt row (8 keys each = the tile's 128 bytes), a thread holds D / 32 rows
  constexpr int CPL = D / 64;             // chunks per lane and key
  constexpr int VR = D / 32;              // value rows per thread
  constexpr int PW = D + 2;               // floats per (split, head) partial: accumulators, max, sum
  static_assert((REP + 1) * LPH <= 256 && REP * D <= 1024, "too many heads per group");
  const int m = blockIdx.z, g = blockIdx.y, c = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), t256 = tid & 255, wv = (tid >> 6) & 3;
  const DecTile t = p.tiles[m];
  const int slot = p.row_slot[m];
  // chunk length: at least min_chunk keys, at most as many chunks as the grid has (MAX_SPLITS, fewer when the caller's
  // bound on the key count sized it: a bound that is too small then costs time, not keys), whole rounds of GROUPS tiles
  const int max_chunks = (int)gridDim.x;
  const int chunk = (max(p.min_chunk, (t.kv_len + max_chunks - 1) / max_chunks) + GROUPS * TK - 1) / (GROUPS * TK) * (GROUPS * TK);
  const int nsplit = (t.kv_len + chunk - 1) / chunk;
  if (c >= nsplit) return;
  const int k_begin = c * chunk, k_end = min(k_begin + chunk, t.kv_len);
  const int rounds = (k_end - k_begin + GROUPS * TK - 1) / (GROUPS * TK);

  __shared__ __attribute__((aligned(16))) bf16 qs[REP][D];
  __shared__ __attribute__((aligned(16))) bf16 k_new[D];
  __shared__ bf16 v_new[D];
  __shared__ float sc[GROUPS][REP][TK];
  __shared__ __attribute__((aligned(16))) bf16 pb[GROUPS][REP][TK];
  __shared__ float alpha_s[GROUPS][REP], m_run[GROUPS][REP], l_run[GROUPS][REP];
  __shared__ float og[GROUPS][REP][D];
  __shared__ unsigned ticket_s;

  const int c8 = t256 & 7, r32 = t256 >> 3;   // chunk lane; key (r32, r32 + 32) / value row (r32 + 32 i)
  const bf16* kp = p.K + ((long)g * p.S + t.kv_start + k_begin + r32) * D + 8 * c8;
  const bf16* vp = p.Vt + ((long)g * D + r32) * p.S + t.kv_start + k_begin + 8 * c8;
  struct Regs {
    bf16x8 k[2][CPL], v[VR];
  };
  Regs ra, rb;
  auto load = [&](Regs& x, int k0) {   // k0: tile start relative to k_begin, inside the sample's padded range
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int w = 0; w < CPL; ++w) x.k[u][w] = *(const bf16x8*)(kp + (long)(k0 + 32 * u) * D + 64 * w);
#pragma unroll
    for (int i = 0; i < VR; ++i) x.v[i] = *(const bf16x8*)(vp + (long)(32 * i) * p.S + k0);
  };
  // the first two tiles of every group are requested before the queries are prepared
  if (k_begin + grp * TK < k_end) load(ra, grp * TK);
  if (k_begin + (GROUPS + grp) * TK < k_end) load(rb, (GROUPS + grp) * TK);

  const int qd = p.Hq * D, kvd = p.Hkv * D;
  const bf16* row = p.qkv + (long)m * p.ld;
  const int slot_rel = slot - t.kv_start;                    // the new token's key index within the sample
  const bool owner = slot_rel >= k_begin && slot_rel < k_begin + chunk;

  // ---- queries (and, in the owning workgroup, the new key): norm + RoPE, LPH lanes per head
  if (tid < (REP + 1) * LPH) {
    const int hh = tid / LPH, j = tid % LPH;
    const bool is_q = hh < REP;
    if (is_q || owner) {   // uniform per group of LPH lanes
      const bf16* src = is_q ? row + (long)(g * REP + hh) * D + 8 * j : row + qd + (long)g * D + 8 * j;
      const long cs = (long)p.row_cs[m] * D;
      const bf16x8 o = qk_norm_rope_lane<D>(*(const bf16x8*)src, j, is_q ? p.q_w : p.k_w, p.cos_t + cs, p.sin_t + cs,
                                            p.eps, p.und_rounding);
      if (is_q) {
        *(bf16x8*)(&qs[hh][8 * j]) = o;
      } else {
        *(bf16x8*)(&k_new[8 * j]) = o;
        *(bf16x8*)(p.K + ((long)g * p.S + slot) * D + 8 * j) = o;
      }
    }
  } else if (owner && tid >= 256 && tid < 256 + D) {
    const int d = tid - 256;
    const bf16 v = row[qd + kvd + (long)g * D + d];
    v_new[d] = v;
    p.Vt[((long)g * D + d) * p.S + slot] = v;
  }
  if (tid < GROUPS * REP) {
    m_run[tid / REP][tid % REP] = -INFINITY;
    l_run[tid / REP][tid % REP] = 0.f;
  }
  __syncthreads();

  float acc[VR][REP];
#pragma unroll
  for (int i = 0; i < VR; ++i)
#pragma unroll
    for (int h = 0; h < REP; ++h) acc[i][h] = 0.f;

  // one round on the tile held in x; every thread of the workgroup runs it (barriers), `active` says whether this
  // group's tile exists
  auto round = [&](Regs& x, int r) {
    const int k0 = (r * GROUPS + grp) * TK;              // this group's tile, relative to k_begin
    const bool active = k_begin + k0 < k_end;            // uniform per group
    if (active) {
      const int jn = slot_rel - (k_begin + k0);          // the new token's index in this tile, if inside it
      if (owner && jn >= 0 && jn < TK) {
        // the workspace may not hold the new key / value yet: take them from LDS
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (r32 + 32 * u == jn) {
#pragma unroll
            for (int w = 0; w < CPL; ++w) x.k[u][w] = *(const bf16x8*)(&k_new[8 * c8 + 64 * w]);
          }
        if ((jn >> 3) == c8) {
#pragma unroll
          for (int i = 0; i < VR; ++i) {
            const bf16 v = v_new[r32 + 32 * i];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (e == (jn & 7)) x.v[i][e] = v;
          }
        }
      }
      // scores of this thread's two keys against the REP queries over its chunks, summed over the 8 lanes of a key
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        bf16x8 q8[CPL];
#pragma unroll
        for (int w = 0; w < CPL; ++w) q8[w] = *(const bf16x8*)(&qs[h][8 * c8 + 64 * w]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          float a = 0.f;
#pragma unroll
          for (int w = 0; w < CPL; ++w)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              a = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x.k[u][w][2 * e], x.k[u][w][2 * e + 1]},
                                                  bf16x2_t{q8[w][2 * e], q8[w][2 * e + 1]}, a, false);
          a = group8_sum(a);
          if (c8 == 0) sc[grp][h][r32 + 32 * u] = k_begin + k0 + r32 + 32 * u < k_end ? a * p.scale_log2 : -INFINITY;
        }
      }
    }
    __syncthreads();
    if (active) {
      // online softmax: wave wv of the group owns heads wv, wv + 4; a lane is a key
      for (int h = wv; h < REP; h += 4) {
        const float v = sc[grp][h][lane];
        const float tmax = wave_max(v);
        const float m_old = m_run[grp][h];
        const float m_new = fmaxf(m_old, tmax);       // finite: an active tile holds a live key
        const float pv = exp2f(v - m_new);
        const float psum = wave_sum(pv);
        pb[grp][h][lane] = (bf16)pv;
        if (lane == 0) {
          const float al = exp2f(m_old - m_new);
          alpha_s[grp][h] = al;
          m_run[grp][h] = m_new;
          l_run[grp][h] = l_run[grp][h] * al + psum;
        }
      }
    }
    __syncthreads();
    if (active) {
      // P.V: this thread's value rows against its 8 keys
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const bf16x8 p8 = *(const bf16x8*)(&pb[grp][h][8 * c8]);
        const float al = alpha_s[grp][h];
#pragma unroll
        for (int i = 0; i < VR; ++i) {
          float a = acc[i][h] * al;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            a = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{x.v[i][2 * e], x.v[i][2 * e + 1]}, bf16x2_t{p8[2 * e], p8[2 * e + 1]}, a, false);
          acc[i][h] = a;
        }
      }
    }
    // sc is rewritten by the next round before its first barrier, after every reader of this round passed the second
    // one; pb / alpha_s are rewritten after the next round's first barrier, i.e. after this P.V everywhere
  };
  for (int r = 0; r < rounds; r += 2) {
    round(ra, r);
    if (k_begin + ((r + 2) * GROUPS + grp) * TK < k_end) load(ra, ((r + 2) * GROUPS + grp) * TK);
    if (r + 1 < rounds) {
      round(rb, r + 1);
      if (k_begin + ((r + 3) * GROUPS + grp) * TK < k_end) load(rb, ((r + 3) * GROUPS + grp) * TK);
    }
  }
  // ---- merge the groups through LDS
#pragma unroll
  for (int i = 0; i < VR; ++i)
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const float a = group8_sum(acc[i][h]);
      if (c8 == 0) og[grp][h][r32 + 32 * i] = a;
    }
  __syncthreads();
  const bool mine_out = tid < REP * D;
  const int oh = tid / D, od = tid % D;
  float o_sum = 0.f, l_sum = 0.f, mx = -INFINITY;
  if (mine_out) {
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) mx = fmaxf(mx, m_run[q][oh]);       // group 0 always had a live tile
#pragma unroll
    for (int q = 0; q < GROUPS; ++q) {
      const float wgt = exp2f(m_run[q][oh] - mx);
      l_sum += l_run[q][oh] * wgt;
      o_sum += og[q][oh][od] * wgt;
    }
  }
  bf16* orow = p.O + ((long)m * p.Hq + (long)g * REP) * D;
  if (nsplit == 1) {   // the usual case: the whole sample was this workgroup's
    if (mine_out) orow[tid] = (bf16)(o_sum / l_sum);
    return;
  }
  // Partials travel between workgroups (often between XCDs, whose L2s are not coherent with each other) through
  // device-scope accesses: write-through stores, then the ticket, then device-scope loads in the last workgroup.
  // A release / acquire fence pair instead costs an L2 write-back per workgroup, and those queue up per XCD: with 18
  // chunks per kv head (9 k keys) the launch took 30 us longer than this form.
  // CONTRACT (hardware, not the HIP memory model: relaxed agent-scope atomics and a workgroup fence give no
  // happens-before edge between workgroups on paper).  It holds because of how gfx950 executes exactly these forms:
  //   * every store to `part` is an agent-scope atomic store = `global_store_dword ... sc1` (write-through: the bytes
  //     leave this XCD's L2 without a buffer_wbl2);
  //   * every storing wave drains them with `s_waitcnt vmcnt(0)` BEFORE the workgroup barrier, and only then does
  //     one lane add to the ticket (agent-scope atomic, executed at the memory side);
  //   * the last-arriving workgroup learns it from the value its own add returned, its other waves start loading
  //     only behind the barrier that follows, and every load of `part` is an agent-scope atomic load =
  //     `global_load_dword ... sc1` (bypasses this CU's L1, which other CUs' stores never refresh).
  // A plain (non-atomic) access to `part`, a `flat_` form of these accesses, or a target that lowers agent-scope
  // relaxed atomics without sc1 breaks it silently; test_decode_attention_matches_three_launch_path (34 splits,
  // partials crossing XCDs) is the guard, and the disassembly must keep showing `sc1` on these instructions.
  float* part = p.part + (((long)m * p.Hkv + g) * MAX_SPLITS) * REP * PW;
  if (mine_out) {
    float* mine = part + ((long)c * REP + oh) * PW;
    __hip_atomic_store(mine + od, o_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (od == 0) {
      __hip_atomic_store(mine + D, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + D + 1, l_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // every wave's stores have completed (the barrier's release waits for them) before one thread counts the workgroup in
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned* ticket = p.tickets + (long)m * p.Hkv + g;
  if (tid == 0) ticket_s = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (ticket_s != (unsigned)(nsplit - 1)) return;
  if (mine_out) {
    // one pass over the chunks, eight at a time so that their loads are in flight together
    float gm = -INFINITY, l = 0.f, o = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 8) {
      float pm[8], pl[8], po[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float* ps = part + ((long)min(s0 + i, nsplit - 1) * REP + oh) * PW;
        pm[i] = __hip_atomic_load(ps + D, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pl[i] = __hip_atomic_load(ps + D + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        po[i] = __hip_atomic_load(ps + od, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (s0 + i < nsplit) {
          const float nm = fmaxf(gm, pm[i]);
          const float a = exp2f(gm - nm), w = exp2f(pm[i] - nm);   // gm = -inf only before the first chunk: a = 0
          l = l * a + pl[i] * w;
          o = o * a + po[i] * w;
          gm = nm;
        }
      }
    }
    orow[tid] = (bf16)(o / l);
  }
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}

}  // namespace

// Development knob: keys per workgroup before a sample's keys are split over workgroups.
static int g_decode_min_chunk = 256;
extern "C" void bagel_set_decode_min_chunk(int keys) { g_decode_min_chunk = keys >= 256 ? keys / 256 * 256 : 256; }

extern "C" long bagel_decode_attention_workspace_bytes(int rows, int Hq, int Hkv, int head_dim) {
  if (rows <= 0 || Hq <= 0 || Hkv <= 0 || head_dim <= 0) return 0;
  const long part = (long)rows * Hkv * MAX_SPLITS * (Hq / Hkv) * (head_dim + 2) * 4;
  return TICKET_BYTES + (part + 255) / 256 * 256;   // tickets (one per sample and kv head) + partials
}

extern "C" int bagel_decode_attention_supported(int rows, int Hq, int Hkv, int head_dim) {
  if (rows <= 0 || Hkv <= 0 || Hq % Hkv != 0 || (long)rows * Hkv > MAX_TICKETS) return 0;
  const int rep = Hq / Hkv;
  return (head_dim == 128 || head_dim == 64) && (rep == 7 || rep == 2);
}

extern "C" int bagel_decode_attention_bf16(const void* qkv, long ld, int rows, int Hq, int Hkv, int head_dim,
                                           const void* q_norm, const void* k_norm, const void* cos_table,
                                           const void* sin_table, const void* row_cs, const void* row_slot,
                                           const void* tiles, void* K_ws, void* Vt_ws, long S, void* O, float eps,
                                           int und_rounding, long max_kv_len, void* workspace, long workspace_bytes, void* stream) {
  BAGEL_CHECK(qkv && q_norm && k_norm && cos_table && sin_table && row_cs && row_slot && tiles && K_ws && Vt_ws && O && workspace,
              "bagel_decode_attention_bf16: null operand");
  BAGEL_CHECK(bagel_decode_attention_supported(rows, Hq, Hkv, head_dim),
              "bagel_decode_attention_bf16: rows=%d Hq=%d Hkv=%d head_dim=%d not supported (rows * Hkv <= 256, 7 or 2 query heads "
              "per kv head, head_dim 128 or 64)", rows, Hq, Hkv, head_dim);
  BAGEL_CHECK(ld % 8 == 0 && ld >= (long)(Hq + 2 * Hkv) * head_dim, "bagel_decode_attention_bf16: bad leading dim");
  BAGEL_CHECK(S > 0 && S % 64 == 0, "bagel_decode_attention_bf16: S=%ld must be a positive multiple of 64", S);
  BAGEL_CHECK(((uintptr_t)qkv | (uintptr_t)K_ws | (uintptr_t)Vt_ws | (uintptr_t)O | (uintptr_t)workspace | (uintptr_t)tiles) % 16 == 0,
              "bagel_decode_attention_bf16: misaligned operand");
  BAGEL_CHECK(workspace_bytes >= bagel_decode_attention_workspace_bytes(rows, Hq, Hkv, head_dim),
              "bagel_decode_attention_bf16: workspace of %ld bytes, %ld needed", workspace_bytes,
              bagel_decode_attention_workspace_bytes(rows, Hq, Hkv, head_dim));
  DecParams p;
  p.qkv = (const bf16*)qkv; p.ld = ld; p.q_w = (const bf16*)q_norm; p.k_w = (const bf16*)k_norm;
  p.cos_t = (const bf16*)cos_table; p.sin_t = (const bf16*)sin_table;
  p.row_cs = (const int*)row_cs; p.row_slot = (const int*)row_slot; p.tiles = (const DecTile*)tiles;
  p.K = (bf16*)K_ws; p.Vt = (bf16*)Vt_ws; p.O = (bf16*)O;
  p.tickets = (unsigned*)workspace; p.part = (float*)((char*)workspace + TICKET_BYTES);
  p.S = S; p.Hq = Hq; p.Hkv = Hkv; p.eps = eps; p.und_rounding = und_rounding;
  p.scale_log2 = 1.4426950408889634f / sqrtf((float)head_dim);
  // no sample has more keys than the caller's bound (or than the workspace has slots), which bounds the chunks any of
  // them can use: workgroups beyond a sample's chunks exit at once, but a thousand of them still cost a launch each
  p.min_chunk = g_decode_min_chunk;
  const long kv_bound = max_kv_len > 0 ? std::min(max_kv_len, S) : S;
  const int splits = (int)std::min<long>(MAX_SPLITS, (kv_bound + p.min_chunk - 1) / p.min_chunk);
  const dim3 grid(splits, Hkv, rows), block(1024);
  const int rep = Hq / Hkv;
  hipStream_t st = (hipStream_t)stream;
  if (head_dim == 128 && rep == 7) hipLaunchKernelGGL((decode_attention_kernel<128, 7>), grid, block, 0, st, p);
  else if (head_dim == 128) hipLaunchKernelGGL((decode_attention_kernel<128, 2>), grid, block, 0, st, p);
  else if (rep == 7) hipLaunchKernelGGL((decode_attention_kernel<64, 7>), grid, block, 0, st, p);
  else hipLaunchKernelGGL((decode_attention_kernel<64, 2>), grid, block, 0, st, p);
  BAGEL_LAUNCH_CHECK();
  return 0;
}

// Packed varlen GQA attention for the generation forward (gfx950).
//
// Replaces flash_attn_varlen_func as called from PackedAttentionMoT.forward_inference
// (qwen2_navit.py:590-599) together with the per-layer rebuild of the merged K/V tensors
// (qwen2_navit.py:570-585): keys/values live in a persistent workspace where each sample's context
// and its query tokens are adjacent, so a query tile just walks one contiguous key range.
//
// Workspace layout per layer (bf16), S = padded total key slots, every sample's range starts at a
// multiple of 64 slots:
//     K  [Hkv][S][D]      row-major keys (RoPE already applied)
//     Vt [Hkv][D][S]      values TRANSPOSED, so that a lane's MFMA operand is 8 consecutive keys
//
// One workgroup = 4 waves = 128 query rows of one query head; each wave owns 32 rows.  Per 64-key tile:
//     S^T = K . Q^T        (keys on the MFMA row index, query row on the lane)
//     online softmax       (row statistics are lane-local; halves combined with one cross-lane swap)
//     O^T += Vt . P^T      (the S^T accumulators, rounded to bf16, ARE the B operand: no LDS round trip)
// K rows are fed to the first product in the order key pi(i) = i with bits 2 and 3 swapped, which makes
// the accumulator-as-operand k order of the second product the natural key order.
// K/V tiles arrive by 16-byte LDS-DMA into ONE K and ONE Vt buffer (32 KiB at head_dim 128, so three
// workgroups = three waves per SIMD fit a CU and cover each other's stalls; measured 4-12 % faster than
// double buffers at two workgroups per CU).  Two barriers per tile: the first publishes the tile's keys and
// frees the Vt buffer, whose refill then flies under the score product; the second publishes the values and
// frees the K buffer, whose refill flies under P.V.  The XOR swizzles are applied to the DMA source address
// and to the fragment reads.
// Arithmetic follows FlashAttention-2: fp32 scores and statistics, P rounded to bf16 for P.V,
// fp32 accumulation, division by the fp32 row sum at the end.
#include "common.h"

#ifndef ATTN_QK_RING
#define ATTN_QK_RING 3      // key fragments in flight ahead of the score products
#endif

struct AttnTile {
  int q_row0;       // first query row (row index into Q / O)
  int n_rows;       // valid rows in this tile (<= 128)
  int kv_start;     // first key slot of the sample (multiple of 64)
  int kv_len;       // number of valid keys
  int causal_base;  // < 0: every row sees all kv_len keys; >= 0: row r of the tile sees keys < causal_base + r + 1
                    // (bottom-right aligned causal mask of flash-attn >= 2.1: causal_base = context length +
                    //  index of the tile's first row among the sample's new tokens)
  int pad[3];
};

struct AttnParams {
  const bf16* Q;   // [M][Hq][D]
  const bf16* K;   // [Hkv][S][D]
  const bf16* Vt;  // [Hkv][D][S]
  bf16* O;         // [M][Hq][D]
  const AttnTile* tiles;
  int n_tiles, Hq, Hkv;
  long S;
  float scale_log2;  // log2(e) / sqrt(D)
};

constexpr int KT = 64;  // keys per tile

// Single-instruction maxima on MFMA results.  fmaxf() makes the compiler canonicalise each accumulator
// first (one extra v_max per input: 55 instructions for 32 scores instead of 17); the scores are never
// signalling NaNs, so the bare instruction is enough.
__device__ __forceinline__ float vmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// DEEP = false: the layout described above (one K and one Vt buffer, two barriers per tile, three workgroups per
// CU) - the throughput form, used whenever the launch can fill the chip.
// DEEP = true: two K and two Vt buffers, the whole next tile in flight behind the current one and one barrier
// per tile - the latency form for launches of at most two workgroups per CU (a decode step, the vision
// tower on one image), where nothing else on the CU would cover the DMA latency the single buffers expose.
template <int HD, bool DEEP>
__global__ __launch_bounds__(256, DEEP ? 2 : 3) void attention_kernel(AttnParams p) {
  static_assert(HD == 128 || HD == 64, "head_dim 128 or 64");
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer-descriptor builtins exist only in the device pass
  constexpr int KS = HD / 16;        // k steps of the score product
  constexpr int DB = HD / 32;        // 32-wide d blocks of the output
  constexpr int K_BYTES = KT * HD * 2, V_BYTES = HD * KT * 2, STAGE = K_BYTES + V_BYTES;
  constexpr int K_ROW_BYTES = HD * 2;            // 256 B (16 chunks) or 128 B (8 chunks)
  constexpr int K_RPP = 1024 / K_ROW_BYTES;      // K rows per 1-KiB DMA piece
  constexpr int K_CH = K_ROW_BYTES / 16;         // 16-byte chunks per K row
  constexpr int K_IT = KT / K_RPP / 4;           // K pieces per wave and tile
  constexpr int V_IT = HD / 8 / 4;               // Vt pieces (8 rows of 128 B) per wave and tile
  // chunk XOR of a tile row: 256-B rows spread 16 rows over the 16 slots of a bank row; 128-B rows put even and
  // odd rows on different halves, so the XOR uses row >> 1
  auto k_swz = [](int row) { return HD == 128 ? (row & 15) : ((row >> 1) & 7); };
  extern __shared__ __attribute__((aligned(16))) char smem[];

  // work order: (kv head, tile, query head within the group) with each XCD taking a contiguous range,
  // so the 7 query heads that share a kv head, and neighbouring query tiles, hit one XCD's L2
  const int rep = p.Hq / p.Hkv;
  const int nwg = p.n_tiles * p.Hq;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int kvh = wg / (p.n_tiles * rep);
  const int rem = wg - kvh * p.n_tiles * rep;
  const int tile_idx = rem / rep;
  const int head = kvh * rep + rem % rep;
  const AttnTile t = p.tiles[tile_idx];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  // ---- Q fragments (B operand of S^T): lane (row, h) holds Q[row][16*ks + 8h .. +8)
  const int qrow = min(wave * 32 + l31, t.n_rows - 1);
  const bf16* qp = p.Q + ((long)(t.q_row0 + qrow) * p.Hq + head) * HD + 8 * h;
  bf16x8 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *(const bf16x8*)(qp + 16 * ks);

  // ---- staging: K tile = 64 rows x HD*2 B, Vt tile = HD rows x 128 B, both in 1-KiB DMA pieces
  // K and Vt tiles are fetched through buffer descriptors based at this sample's first key: the address of a
  // piece is descriptor base + a per-lane 32-bit offset that never changes + a scalar tile offset, so the DMA
  // instructions need no per-piece vector address arithmetic.
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.K + ((long)kvh * p.S + t.kv_start) * HD), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.Vt + (long)kvh * HD * p.S + t.kv_start), 0, 0x7fffffff, 0x00020000);
  // A wave's pieces of a tile are 16 (HD 128) or 32 (HD 64) key rows, resp. 32 value rows, apart: the XOR swizzle of
  // the lane's row is the same for each of them, so ONE lane offset per operand serves every piece and the piece's
  // distance (4 KiB of keys, 32 rows of S values) rides in the scalar offset beside the tile offset.
  const int krow0 = wave * K_RPP + lane / K_CH;
  const int k_voff = (krow0 * HD + (((lane % K_CH) ^ k_swz(krow0)) * 8)) * 2;
  const int vrow0 = wave * 8 + (lane >> 3);
  const int v_voff = (int)(((long)vrow0 * p.S + (((lane & 7) ^ ((vrow0 >> 1) & 7)) * 8)) * 2);
  const int v_piece = (int)(64 * p.S);     // bytes between a wave's consecutive Vt pieces (32 rows)
  auto stage_k_piece = [&](int kt, int buf, int it) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LDS_PTR(smem + buf * STAGE + (it * 4 + wave) * 1024), 16, k_voff,
                                             kt * (KT * HD * 2) + it * 4096, 0, 0);
  };
  auto stage_v_piece = [&](int kt, int buf, int it) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LDS_PTR(smem + buf * STAGE + K_BYTES + (it * 4 + wave) * 1024), 16,
                                             v_voff, kt * (KT * 2) + it * v_piece, 0, 0);
  };
  auto stage_k = [&](int kt, int buf) {
#pragma unroll
    for (int it = 0; it < K_IT; ++it) stage_k_piece(kt, buf, it);
  };
  auto stage_v = [&](int kt, int buf) {
#pragma unroll
    for (int it = 0; it < V_IT; ++it) stage_v_piece(kt, buf, it);
  };

  // ---- fragment read offsets
  // K: A-operand row i of a 32-key block is key pi(i) (bits 2,3 swapped).  The 16-byte chunk (2 ks + h) of that row
  // sits at chunk ^ swizzle(row), i.e. at byte (base of the row | (h ^ swizzle) << 4) ^ (ks << 5): one address
  // register and one XOR per read instead of one register per k step.
  const int pi = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
  const int k_a0 = pi * (HD * 2) + ((h ^ k_swz(pi)) << 4);
  // Vt: row d = db*32 + l31; chunk 2 j + h of the row (j = 16-key step of the tile) at (... | (h ^ swizzle) << 4) ^ (j << 5)
  const int v_a0 = l31 * (KT * 2) + ((h ^ ((l31 >> 1) & 7)) << 4);

  f32x16 o_acc[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const bool causal = t.causal_base >= 0;
  const int kv_len = causal ? min(t.kv_len, t.causal_base + t.n_rows) : t.kv_len;
  const int row_limit = causal ? t.causal_base + wave * 32 + l31 + 1 : kv_len;  // keys this lane's row may see
  const int n_kt = (kv_len + KT - 1) / KT;
  stage_k(0, 0);
  if constexpr (DEEP) stage_v(0, 0);
  if (wave * 32 >= t.n_rows) {
    // This wave owns no query row (a two-row tile of special tokens, a one-row decode tile, a ragged last tile):
    // it only moves its share of the K/V tiles and keeps the barrier count, leaving the SIMD's matrix and
    // vector issue to the workgroups it shares the CU with.
    for (int kt = 0; kt < n_kt; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if constexpr (DEEP) {
        if (kt + 1 < n_kt) {
          stage_k(kt + 1, (kt + 1) & 1);
          stage_v(kt + 1, (kt + 1) & 1);
        }
      } else {
        stage_v(kt, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < n_kt) stage_k(kt + 1, 0);
      }
    }
    return;
  }
  for (int kt = 0; kt < n_kt; ++kt) {
    const char* kb = smem + (DEEP ? (kt & 1) * STAGE : 0);
    const char* vb = kb + K_BYTES;
    // The LDS-DMA of this tile's keys must have landed before the barrier publishes it: the compiler's
    // barrier fence does not wait for LDS-DMA, so the wait is explicit.  The barrier also says that every
    // wave is done with the previous tile's values.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // S^T for two 32-key blocks.  The 2 KS key fragments go through a ring of RING registers quads: fragment
    // i + RING is requested right behind the product that consumed fragment i, so a product never waits for a read
    // issued just before it.  The addresses are re-derived from one register per tile (the empty asm keeps the
    // compiler from holding 2 KS of them across the loop, which is what forced it to read one fragment at a time).
    f32x16 s[2];
    int k_a = k_a0;
    asm volatile("" : "+v"(k_a));
    constexpr int RING = ATTN_QK_RING;
    bf16x8 kf[RING];
    auto kread = [&](int i) {
      return *(const bf16x8*)(kb + (i / KS) * (32 * HD * 2) + (k_a ^ ((i % KS) << 5)));
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) kf[i] = kread(i);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DEEP) {
      if (kt + 1 < n_kt) {
        stage_k(kt + 1, (kt + 1) & 1);
        stage_v(kt + 1, (kt + 1) & 1);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[1][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * KS; ++i) {
      s[i / KS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[i % RING], qf[i % KS], s[i / KS], 0, 0, 0);
      if (i + RING < 2 * KS) kf[i % RING] = kread(i + RING);
      // the values' DMA pieces go out one by one between the products (LDS reads and LDS-DMA keep program order)
      if constexpr (!DEEP)
        if (i % 2 == 1 && i / 2 < V_IT) stage_v_piece(kt, 0, i / 2);
      // The scheduler, left alone, sinks every read to just before its product to save registers, and each product
      // then waits out a whole LDS round trip: nothing may cross from one product / piece / read group to the next.
      __builtin_amdgcn_sched_barrier(0);
    }
    // mask the tail keys of the last tile (raw scores; the softmax scale is folded into the exponent)
    const int key0 = kt * KT;
    const bool tail = key0 + KT > (causal ? t.causal_base + wave * 32 + 1 : kv_len);  // wave-uniform
    if (tail) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key0 + 32 * b + (r & 3) + 4 * ((r >> 2) & 1) + 8 * h + 16 * (r >> 3);
          s[b][r] = key < row_limit ? s[b][r] : -INFINITY;
        }
    }
    // online softmax: this lane holds 32 of the row's 64 keys, lane^32 the other 32
    // The compiler does not track the MFMA-result read hazard for operands of inline assembly: wait out
    // the 11 states an 8-pass MFMA needs before a VALU may read its result.  The data dependence through
    // the operands keeps every later read of the scores behind this point.
    asm volatile("s_nop 7\n\ts_nop 3" : "+v"(s[0]), "+v"(s[1]));
    float ma = vmax2(s[0][0], s[0][15]), mb = vmax2(s[1][0], s[1][15]);
#pragma unroll
    for (int r = 1; r < 15; r += 2) {
      ma = vmax3(ma, s[0][r], s[0][r + 1]);
      mb = vmax3(mb, s[1][r], s[1][r + 1]);
    }
    float mx = vmax2(ma, mb);
    // lanes l and l^32 exchange their maxima in one instruction (no LDS round trip)
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = vmax2(__uint_as_float(sw[0]), __uint_as_float(sw[1])) * p.scale_log2;  // scale > 0 commutes with max
    // The accumulators are rescaled only when some row of this wave raises its running maximum
    // (alpha == 1 exactly otherwise), which after the first few tiles is rare.
    if (__any(mx > m_run)) {
      const float m_new = fmaxf(m_run, mx);
      // rows of a padded tile (row >= n_rows, causal) can be fully masked so far: keep exp2 arguments finite
      const float alpha = __builtin_amdgcn_exp2f(m_run - (m_new == -INFINITY ? 0.f : m_new));
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[d][r] *= alpha;
    }
    if constexpr (!DEEP) {
      // this tile's values have landed and every wave is done with its keys
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < n_kt) stage_k(kt + 1, 0);
    }
    const float neg_m = m_run == -INFINITY ? 0.f : -m_run;
    int v_a = v_a0;
    asm volatile("" : "+v"(v_a));
    float psum = 0.f;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[b][r], p.scale_log2, neg_m));
        s[b][r] = e;
        psum += e;
      }
    l_run += psum;

    // O^T += Vt . P^T : 4 k steps of 16 keys
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (bf16)s[b][8 * st + j];
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const bf16x8 vf = *(const bf16x8*)(vb + d * 32 * (KT * 2) + (v_a ^ ((2 * b + st) << 5)));
          o_acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[d], 0, 0, 0);
        }
      }
  }

  // ---- epilogue: lane holds query row l31, d = db*32 + (r&3) + 8*(r>>2) + 4h
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  const int row = wave * 32 + l31;
  if (row < t.n_rows) {
    bf16* op = p.O + ((long)(t.q_row0 + row) * p.Hq + head) * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16)(o_acc[d][4 * g + e] * inv);
        *(bf16x4*)(op + d * 32 + 8 * g + 4 * h) = o4;
      }
  }
#endif
}

// The same algorithm on v_mfma_f32_16x16x32_bf16 (head_dim 128): twice the matrix instructions of half the
// length.  Under the matrix load of this loop the chip holds a higher clock on the 16 x 16 shape, so cycles per
// FLOP do not decide between the two forms; they are compared by wall time on random data.
//     S^T block (16 keys x 16 queries): lane (i = lane & 15, g = lane >> 4) holds query i, MFMA rows 4g .. 4g+3
//     P^T as B operand of O^T += Vt . P^T over a 32-key step: lane (i, g) must hold keys 8g .. 8g+7 of the step
// so MFMA row 4g + e of the step's key block b (b = 0, 1) is fed key 8g + 4b + e: the two blocks' accumulators,
// rounded to bf16, are the B operand in natural key order.  A wave owns 32 query rows as two 16-query blocks.
// K tile swizzle: chunk ^ (MFMA row index of the key), which makes every ds_read_b128 lane group hit 16 distinct
// slots; a wave stages the four 1-KiB pieces whose rows share that index's high bits, so one lane offset serves
// them all.  The Vt tile keeps the layout of the 32 x 32 form.
template <bool DEEP>
__global__ __launch_bounds__(256, DEEP ? 2 : 3) void attention16_kernel(AttnParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HD = 128;
  constexpr int KS = HD / 32;        // k steps of the score product
  constexpr int DB = HD / 16;        // 16-wide d blocks of the output
  constexpr int K_BYTES = KT * HD * 2, V_BYTES = HD * KT * 2, STAGE = K_BYTES + V_BYTES;
  constexpr int K_IT = 4, V_IT = 4;  // 1-KiB pieces per wave and tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int rep = p.Hq / p.Hkv;
  const int nwg = p.n_tiles * p.Hq;
  int wg = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int kvh = wg / (p.n_tiles * rep);
  const int rem = wg - kvh * p.n_tiles * rep;
  const int tile_idx = rem / rep;
  const int head = kvh * rep + rem % rep;
  const AttnTile t = p.tiles[tile_idx];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, g = lane >> 4;

  // ---- Q fragments (B operand of S^T): lane (i, g) holds Q[row i of block qb][32 ks + 8 g .. +8)
  bf16x8 qf[2][KS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int qrow = min(wave * 32 + 16 * qb + i16, t.n_rows - 1);
    const bf16* qp = p.Q + ((long)(t.q_row0 + qrow) * p.Hq + head) * HD + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = *(const bf16x8*)(qp + 32 * ks);
  }

  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.K + ((long)kvh * p.S + t.kv_start) * HD), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.Vt + (long)kvh * HD * p.S + t.kv_start), 0, 0x7fffffff, 0x00020000);
  // K piece q = key rows 4q .. 4q+3; the swizzle of row r is ((r >> 3) & 3) << 2 | (r & 3), whose high bits are
  // (q >> 1) & 3: wave w takes pieces 2w, 2w+1, 2w+8, 2w+9.
  const int k_voff = ((lane >> 4) * HD + (((lane & 15) ^ (wave << 2 | (lane >> 4))) * 8)) * 2;
  const int vrow0 = wave * 8 + (lane >> 3);
  const int v_voff = (int)(((long)vrow0 * p.S + (((lane & 7) ^ ((vrow0 >> 1) & 7)) * 8)) * 2);
  const int v_piece = (int)(64 * p.S);
  auto stage_k_piece = [&](int kt, int buf, int it) {
    const int q = 2 * wave + (it & 1) + 8 * (it >> 1);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, LDS_PTR(smem + buf * STAGE + q * 1024), 16, k_voff,
                                             kt * (KT * HD * 2) + q * 1024, 0, 0);
  };
  auto stage_v_piece = [&](int kt, int buf, int it) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, LDS_PTR(smem + buf * STAGE + K_BYTES + (it * 4 + wave) * 1024), 16,
                                             v_voff, kt * (KT * 2) + it * v_piece, 0, 0);
  };
  auto stage_k = [&](int kt, int buf) {
#pragma unroll
    for (int it = 0; it < K_IT; ++it) stage_k_piece(kt, buf, it);
  };
  auto stage_v = [&](int kt, int buf) {
#pragma unroll
    for (int it = 0; it < V_IT; ++it) stage_v_piece(kt, buf, it);
  };

  // ---- fragment read offsets
  // K block kb = 2 j + b of the tile: MFMA row i is key 32 j + 8 (i >> 2) + 4 b + (i & 3); chunk 4 ks + g of that row
  // sits at (4 ks + g) ^ i, i.e. at byte (row base | (g ^ i) << 4) ^ (ks << 6).
  const int k_a0 = (8 * (i16 >> 2) + (i16 & 3)) * (HD * 2) + ((g ^ i16) << 4);
  // Vt: row d = 16 db + i; chunk 4 j + g of the row at (row base | (g ^ swizzle) << 4) ^ (j << 6)
  const int v_a0 = i16 * (KT * 2) + ((g ^ ((i16 >> 1) & 7)) << 4);

  f32x4 o_acc[2][DB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) o_acc[qb][d][r] = 0.f;
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

  const bool causal = t.causal_base >= 0;
  const int kv_len = causal ? min(t.kv_len, t.causal_base + t.n_rows) : t.kv_len;
  int row_limit[2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) row_limit[qb] = causal ? t.causal_base + wave * 32 + 16 * qb + i16 + 1 : kv_len;
  const int n_kt = (kv_len + KT - 1) / KT;
  stage_k(0, 0);
  if constexpr (DEEP) stage_v(0, 0);
  if (wave * 32 >= t.n_rows) {
    // a wave without query rows only moves its share of the tiles and keeps the barrier count
    for (int kt = 0; kt < n_kt; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if constexpr (DEEP) {
        if (kt + 1 < n_kt) {
          stage_k(kt + 1, (kt + 1) & 1);
          stage_v(kt + 1, (kt + 1) & 1);
        }
      } else {
        stage_v(kt, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < n_kt) stage_k(kt + 1, 0);
      }
    }
    return;
  }
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  for (int kt = 0; kt < n_kt; ++kt) {
    const char* kb = smem + (DEEP ? (kt & 1) * STAGE : 0);
    const char* vb = kb + K_BYTES;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // S^T: 4 key blocks x 2 query blocks, 4 k steps each; the 16 key fragments go through a ring, each feeding the
    // two query blocks' products
    f32x4 s[2][4];
    int k_a = k_a0;
    asm volatile("" : "+v"(k_a));
    constexpr int RING = ATTN_QK_RING;
    bf16x8 kf[RING];
    auto kread = [&](int i) {
      const int blk = i / KS;
      return *(const bf16x8*)(kb + (32 * (blk >> 1) + 4 * (blk & 1)) * (HD * 2) + (k_a ^ ((i % KS) << 6)));
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) kf[i] = kread(i);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DEEP) {
      if (kt + 1 < n_kt) {
        stage_k(kt + 1, (kt + 1) & 1);
        stage_v(kt + 1, (kt + 1) & 1);
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[qb][b][r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * KS; ++i) {
      s[0][i / KS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i % RING], qf[0][i % KS], s[0][i / KS], 0, 0, 0);
      s[1][i / KS] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i % RING], qf[1][i % KS], s[1][i / KS], 0, 0, 0);
      if (i + RING < 4 * KS) kf[i % RING] = kread(i + RING);
      if constexpr (!DEEP)
        if (i % 2 == 1 && i / 2 < V_IT) stage_v_piece(kt, 0, i / 2);
      __builtin_amdgcn_sched_barrier(0);
    }
    const int key0 = kt * KT;
    const bool tail = key0 + KT > (causal ? t.causal_base + wave * 32 + 1 : kv_len);  // wave-uniform
    if (tail) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + 32 * (b >> 1) + 8 * g + 4 * (b & 1) + r;
            s[qb][b][r] = key < row_limit[qb] ? s[qb][b][r] : -INFINITY;
          }
    }
    // wait out the MFMA-result read hazard before inline assembly reads the scores
    asm volatile("s_nop 7\n\ts_nop 3"
                 : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]),
                   "+v"(s[1][2]), "+v"(s[1][3]));
    float mx[2];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float m = vmax3(s[qb][0][0], s[qb][0][1], s[qb][0][2]);
      m = vmax3(m, s[qb][0][3], s[qb][1][0]);
      m = vmax3(m, s[qb][1][1], s[qb][1][2]);
      m = vmax3(m, s[qb][1][3], s[qb][2][0]);
      m = vmax3(m, s[qb][2][1], s[qb][2][2]);
      m = vmax3(m, s[qb][2][3], s[qb][3][0]);
      m = vmax3(m, s[qb][3][1], s[qb][3][2]);
      m = vmax2(m, s[qb][3][3]);
      // the row's other keys are on lanes l ^ 16, l ^ 32, l ^ 48: two swaps
      const u32x2 a = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
      m = vmax2(__uint_as_float(a[0]), __uint_as_float(a[1]));
      const u32x2 c = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
      mx[qb] = vmax2(__uint_as_float(c[0]), __uint_as_float(c[1])) * p.scale_log2;
    }
    if (__any(mx[0] > m_run[0] || mx[1] > m_run[1])) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const float m_new = fmaxf(m_run[qb], mx[qb]);
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - (m_new == -INFINITY ? 0.f : m_new));
        m_run[qb] = m_new;
        l_run[qb] *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 4; ++r) o_acc[qb][d][r] *= alpha;
      }
    }
    if constexpr (!DEEP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < n_kt) stage_k(kt + 1, 0);
    }
    int v_a = v_a0;
    asm volatile("" : "+v"(v_a));
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const float neg_m = m_run[qb] == -INFINITY ? 0.f : -m_run[qb];
      float psum = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][b][r], p.scale_log2, neg_m));
          s[qb][b][r] = e;
          psum += e;
        }
      l_run[qb] += psum;
    }

    // O^T += Vt . P^T : 2 k steps of 32 keys, 8 d blocks, 2 query blocks
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x8 pf[2];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pf[qb][e] = (bf16)s[qb][2 * j][e];
          pf[qb][4 + e] = (bf16)s[qb][2 * j + 1][e];
        }
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const bf16x8 vf = *(const bf16x8*)(vb + d * 16 * (KT * 2) + (v_a ^ (j << 6)));
        o_acc[0][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0], o_acc[0][d], 0, 0, 0);
        o_acc[1][d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1], o_acc[1][d], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: lane holds query row i of each block, d = 16 db + 4 g + e
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float l_tot = l_run[qb];
    l_tot += __shfl_xor(l_tot, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    const int row = wave * 32 + 16 * qb + i16;
    if (row < t.n_rows) {
      bf16* op = p.O + ((long)(t.q_row0 + row) * p.Hq + head) * HD + 4 * g;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        bf16x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = (bf16)(o_acc[qb][d][e] * inv);
        *(bf16x4*)(op + d * 16) = o4;
      }
    }
  }
#endif
}

extern "C" int bagel_varlen_attention_scaled_bf16(const void* Q, const void* K, const void* Vt, void* O,
                                                  const void* tiles, int n_tiles, int Hq, int Hkv, int head_dim,
                                                  long S, float softmax_scale, void* stream);
// Development knob: -1 automatic (by launch size), 0 / 1 force the throughput / latency form of the kernel.
static int g_attn_deep = -1;
extern "C" void bagel_set_attn_deep(int v) { g_attn_deep = v; }
// Development knob: 1 = the 16 x 16 x 32 MFMA form of the head_dim-128 kernel, 0 = the 32 x 32 x 16 form.
#ifndef ATTN_MFMA16
#define ATTN_MFMA16 0
#endif
static int g_attn_mfma16 = ATTN_MFMA16;
extern "C" void bagel_set_attn_mfma16(int v) { g_attn_mfma16 = v; }

extern "C" int bagel_varlen_attention_bf16(const void* Q, const void* K, const void* Vt, void* O,
                                           const void* tiles, int n_tiles, int Hq, int Hkv, int head_dim,
                                           long S, void* stream) {
  return bagel_varlen_attention_scaled_bf16(Q, K, Vt, O, tiles, n_tiles, Hq, Hkv, head_dim, S,
                                            head_dim > 0 ? 1.0f / sqrtf((float)head_dim) : 0.f, stream);
}

// Same kernel with the softmax scale given by the caller: heads whose true width is below the kernel's
// head_dim are zero-padded (SigLIP: 72 -> 128) and keep their own 1/sqrt(72).
extern "C" int bagel_varlen_attention_scaled_bf16(const void* Q, const void* K, const void* Vt, void* O,
                                                  const void* tiles, int n_tiles, int Hq, int Hkv, int head_dim,
                                                  long S, float softmax_scale, void* stream) {
  BAGEL_CHECK(Q && K && Vt && O && tiles, "bagel_varlen_attention_bf16: null operand");
  BAGEL_CHECK(softmax_scale > 0.f, "bagel_varlen_attention_bf16: softmax scale must be positive");
  BAGEL_CHECK(head_dim == 128 || head_dim == 64, "bagel_varlen_attention_bf16: head_dim %d not supported (128 or 64)", head_dim);
  BAGEL_CHECK(n_tiles > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "bagel_varlen_attention_bf16: bad head counts");
  BAGEL_CHECK(S > 0 && S % 64 == 0, "bagel_varlen_attention_bf16: S=%ld must be a positive multiple of 64", S);
  BAGEL_CHECK(S * head_dim * 2 < (1L << 31), "bagel_varlen_attention_bf16: S=%ld key slots exceed the 31-bit tile offsets", S);
  BAGEL_CHECK(((uintptr_t)Q | (uintptr_t)K | (uintptr_t)Vt | (uintptr_t)O | (uintptr_t)tiles) % 16 == 0,
              "bagel_varlen_attention_bf16: misaligned operand");
  AttnParams p;
  p.Q = (const bf16*)Q; p.K = (const bf16*)K; p.Vt = (const bf16*)Vt; p.O = (bf16*)O;
  p.tiles = (const AttnTile*)tiles; p.n_tiles = n_tiles; p.Hq = Hq; p.Hkv = Hkv; p.S = S;
  p.scale_log2 = 1.4426950408889634f * softmax_scale;
  const int cus = bagel_cu_count();
  const bool deep = g_attn_deep >= 0 ? g_attn_deep != 0 : (long)n_tiles * Hq <= 2L * cus;
  const size_t tile_pair = (size_t)KT * head_dim * 2 * 2;  // one K tile + one Vt tile
  const size_t lds = deep ? 2 * tile_pair : tile_pair;
  if (head_dim == 128) {
    static BagelLdsAllowed allowed;
    bagel_allow_lds(allowed, (const void*)attention_kernel<128, true>, 2 * KT * 128 * 2 * 2);
    static BagelLdsAllowed allowed16;
    bagel_allow_lds(allowed16, (const void*)attention16_kernel<true>, 2 * KT * 128 * 2 * 2);
    if (g_attn_mfma16) {
      if (deep) hipLaunchKernelGGL((attention16_kernel<true>), dim3(n_tiles * Hq), dim3(256), lds, (hipStream_t)stream, p);
      else hipLaunchKernelGGL((attention16_kernel<false>), dim3(n_tiles * Hq), dim3(256), lds, (hipStream_t)stream, p);
    } else if (deep) hipLaunchKernelGGL((attention_kernel<128, true>), dim3(n_tiles * Hq), dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attention_kernel<128, false>), dim3(n_tiles * Hq), dim3(256), lds, (hipStream_t)stream, p);
  } else {
    if (deep) hipLaunchKernelGGL((attention_kernel<64, true>), dim3(n_tiles * Hq), dim3(256), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((attention_kernel<64, false>), dim3(n_tiles * Hq), dim3(256), lds, (hipStream_t)stream, p);
  }
  BAGEL_LAUNCH_CHECK();
  return 0;
}

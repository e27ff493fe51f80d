// gemm_big: the token-major GEMM of the EgoClip step (every qkv / proj / fc1 / fc2 forward, dgrad and wgrad of
// the 12 SpaceTimeBlocks = ~95 % of the step's FLOPs), built around what bounds an MFMA GEMM on a CDNA4 CU.
//
// (1) LDS bytes per MFMA.  rocprofv3 on the round-1 128x128 and 256x128 kernels (580-670 TF)
//     showed the LDS array -- fragment reads plus LDS-DMA writes -- busy about as long as the matrix pipe, so:
//   * (64*MF) x 256 block tile, MF = 5 (320 rows) or 4 (256 rows): 8 waves as 4 (M) x 2 (N), each wave
//     (16*MF) x 128 = MF x 8 MFMA 16x16x32 fragments (160 / 128 fp32 accumulator registers); per 64-deep k-tile a wave
//     issues 2*(MF+8) ds_read_b128 for 16*MF MFMAs (MF = 5: 26 reads / 80 MFMAs; 128x128: 16 / 32);
//   * k-tile = 64 bf16 = one full 128-B line per row per DMA piece (the BK = 32 kernels fetched half lines);
//   * two LDS stages of (64*MF + 256) x 128 B (144 KiB at MF = 5) filled by LDS-DMA (global_load_lds_dwordx4);
//     ONE barrier per k-tile.  Only waves 0-3 (one per SIMD) issue DMA.  NT: the pieces of k-tile t+1 are issued over
//     the first four of the eight phases of k-tile t (a burst of 18 blocks the issuing wave ~1300 cycles on the L1->LDS
//     path); TN: k-tile t+2 is issued in one burst right after the barrier that retires k-tile t's stage;
//   * a k-tile is 8 "phases" of MF x 2 MFMAs.  NT fetches fragments with inline-asm ds_read_b128 and hand-counted
//     s_waitcnt lgkmcnt(N) (hipcc only emits lgkmcnt(0) here): B fragments two phases ahead through three register sets,
//     the A fragments of k-step 1 over phases 0-2, the first fragments of k-tile t+1 right behind the barrier.  TN
//     fetches through asm transpose reads into pending halves that are committed one phase later.
// (2) Instruction fetch.  A per-block timeline (tools/gemm_trace.py, s_memrealtime stamps) of the first version showed
//     11 us between kernel entry and the first MFMA and 20 us of epilogue per output tile -- next to 23 us of main loop
//     for K = 768 -- with the stores compiled out: straight-line code that runs once per workgroup (a 40x unrolled,
//     9-way branching epilogue was > 100 KB) is fetched cold, one cache line at a time, every time.  Hence
//   * PERSISTENT workgroups: grid = min(#tiles, 256), each workgroup walks tiles v = b, b + G, ... (the order the
//     dispatcher would have used, so the XCD-aware tile map is unchanged) and its code stays in the instruction cache;
//   * a compact epilogue: accumulator fragments go through LDS 16 columns at a time and a ROLLED loop applies the
//     epilogue to whole 64-B row segments; the epilogue flavour is a template parameter (EPI) so each instance carries
//     only its own math;
//   * the first k-tile of the NEXT output tile is in flight (LDS stage 0) while the epilogue of the current one drains
//     through stage 1.
// Tile-count quantisation decides MF: M = 25 120 tokens (B = 32, T = 4) gives 79 x 3 = 237 tiles of 320x256 for the
// N = 768 GEMMs (one round on 256 CUs at 93 %), where 256x256 would need 297 tiles = two rounds at 58 %.
//
// Operand layouts ("trans" in egv_gemm_desc):
//   NT  A[M,K], B[N,K], contraction index contiguous: forward and dgrad (weights are cached transposed for dgrad).
//       LDS image: 128-B rows, 16-B chunk index XOR (row & 7) (conflict-free ds_read_b128, see attn_common.h);
//       the DMA destination is lane-linear, so the permutation is applied to the per-lane SOURCE address.
//   TN  A stored [K, M], B stored [K, N] (wgrad: dW = dY^T X with K = tokens): the k-major tiles are staged as they
//       lie in HBM ([64 k][256] bf16, 512-B rows, 32-B units XOR (k & 7)) and the MFMA fragments are fetched with
//       the CDNA4 transpose read ds_read_b64_tr_b16 -- no transposed copy of dY or X is ever written to HBM (the
//       previous design spent 7 ms / step in transpose kernels).  The k-permutation of a transpose-read fragment
//       (k = 4g+j, 16+4g+j) is the same for both operands, which is all a dot product needs.  Rows past K are
//       fetched from a zero page.  Optionally the same pass produces colsum[m] = sum_k A[k,m] (bias gradient) with
//       one extra MFMA per A-fragment against an all-ones fragment.
// bf16x3 (fp32-grade) products run as three k-segments over the split planes, (A_hi,B_lo), (A_lo,B_hi), (A_hi,B_hi),
// into the same accumulators: one code path, the small terms first.
// f16x2 (fp32-grade from TWO products, csrc/f16x2.h): the same planes-in-pairs stage layout with fp16 planes, passes (A_2,B_2),
// (A_1,B_1) on the fp16 MFMA -- the forward of the video blocks' qkv / fc1 / fc2 Linears in the benchmarked mode.
// f16 (ONE fp16 product, passes == 4): the plain single-plane loop on the fp16 opcode, A = fp16(activation), B = fp16(weight) (plane
// 1 of the weight's f16x2 encoding) -- the forward Linears whose share of the parity budget allows it (fc2; fc1 in the later blocks).
// The last tile row/column is shifted inwards (m0 = M - BM) instead of being predicated: the overlapping rows are
// computed twice with bit-identical results, so the duplicate stores are benign and no lane ever needs a clamp.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "f16x2.h"
#include "egovlp_hip.h"

namespace {

constexpr int KT = 64;    // contraction depth of one LDS tile
constexpr int NFW = 8;    // 16-column fragments per wave
constexpr int BNB = 256;  // block tile columns
constexpr int NC = 2;     // B fragments per phase

// epilogue flavours (template parameter: each kernel instance carries only its own epilogue code)
enum { EPI_RAW = 0,      // store the accumulators (split-K partial slab, or plain fp32 output)
       EPI_LINEAR = 1,   // + bias, + residual -> fp32 and/or planes
       EPI_GELU = 2,     // + bias, pre-activation -> aux_out, gelu -> fp32 and/or planes
       EPI_GELU_BWD = 3, // * gelu'(aux_in) -> fp32 and/or planes
       EPI_GENERIC = 4,  // everything at run time (alpha, ReLU', ...)
       EPI_GELU_X2 = 5 };// EPI_GELU with the activation written as fp16 operand planes -- out_fmt 1: the f16x2 format (csrc/f16x2.h,
                         // first-operand role, two planes); out_fmt 2: ONE plane of plain fp16 (the consumer runs a single fp16 product)
                         // -- [+ bf16 plane for the backward], the saved gelu' as bf16: fc1 forward of the f16x2 mode

typedef __attribute__((ext_vector_type(4))) short s16x4v;
typedef __attribute__((ext_vector_type(8))) short s16x8v;

__device__ uint4 g_zero_page[64];  // 1 KiB of zeros: DMA source for k-rows past the end (TN)

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ bf16x8_t lds_b128(const char* p) { return *(const bf16x8_t*)p; }

// transpose-read fragment: rows r, r+1, r+2, r+3 (this 16-lane group's) and the same +16, 16 columns
__device__ __forceinline__ bf16x8_t lds_tr2(const char* p) {
  const s16x4v x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(p));
  const s16x4v y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(p + 16 * 512));
  const s16x8v z = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
  return __builtin_bit_cast(bf16x8_t, z);
}

// The same transpose read as inline asm.  hipcc (ROCm 7.2) cannot prove that a ds_read_tr BUILTIN does not alias the
// LDS-DMA writes in flight and puts `s_waitcnt vmcnt(0)` in front of every batch of them (seen in the .s of the TN
// kernel: the wait landed right behind the DMA issue of k-tile t+2, serialising every prefetch; 2.15 us per k-tile against
// 1.45 for the NT kernel whose plain LDS loads carry alias info).  An asm read is invisible to that pass; its completion
// is waited for by hand (lgkmcnt(0) at the end of the phase that issued it, before the halves are combined -- the asm
// outputs never live across a basic-block boundary, see cdna_hip_programming.md 5.7).
template <int OFF>
__device__ __forceinline__ u32x2_t tr_asm(unsigned addr) {
  u32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return r;
}

// Plain 16-byte LDS read as inline asm (NT main loop): invisible to hipcc's s_waitcnt insertion, which only ever emits
// lgkmcnt(0) in this kernel and so turns any fetch-ahead deeper than one phase back into "wait for everything"; the
// counted waits are placed by hand (lgkm_wait<N>) and the destination is tied to the wait (tie) so that no consumer can
// be scheduled above it.
template <int OFF>
__device__ __forceinline__ bf16x8_t ld128_asm(unsigned addr) {
  u32x4_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return __builtin_bit_cast(bf16x8_t, r);
}
template <int N>
__device__ __forceinline__ void lgkm_wait() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void tie(bf16x8_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void tie(f16x8_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void tie(u32x4_t& x) { asm volatile("" : "+v"(x)); }
template <int OFF>
__device__ __forceinline__ f16x8_t ld128h_asm(unsigned addr) {
  u32x4_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "i"(OFF));
  return __builtin_bit_cast(f16x8_t, r);
}
__device__ __forceinline__ void tie(u32x2_t& x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void tie(unsigned& x) { asm volatile("" : "+v"(x)); }

// Loop condition of the ROLLED epilogue loops.  LLVM's block-frequency estimate multiplies by ~32 per loop level, so a
// rolled two-level epilogue loop looks "hotter" than the k-tile loop and the register allocator spills the main loop's
// A / B fragments to keep epilogue temporaries in registers; a 50 % back-edge probability tells it the truth.
#define EGV_COLD_LOOP(c) __builtin_expect_with_probability((c), 1, 0.5)

template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// One 4-column piece of one output row: alpha, + bias, activation, + residual, stores (include/egovlp_hip.h order).
template <int EPI>
__device__ __forceinline__ void epilogue4(const egv_gemm_desc& p, f32x4_t v, int m, int n, int z, int ksplit) {
  if (EPI == EPI_RAW) {
    if (ksplit > 1) *(f32x4_t*)(p.partial + ((long)z * p.M + m) * p.N + n) = v;
    else *(f32x4_t*)(p.out_f32 + (long)m * p.ldo + n) = v;
    return;
  }
  if (EPI == EPI_GENERIC && p.alpha != 1.0f) v *= p.alpha;
  if (EPI != EPI_GELU_BWD && p.bias) v += *(const f32x4_t*)(p.bias + n);
  if (EPI == EPI_GELU || (EPI == EPI_GENERIC && p.act == EGV_ACT_GELU)) {
    f32x4_t s = v;                    // what is saved for backward: the pre-activation, or (aux_bf16 == 2) gelu'(it)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float cdf, pdf;
      gelu_parts(v[e], cdf, pdf);
      if (p.aux_bf16 == 2) s[e] = cdf + v[e] * pdf;
      v[e] *= cdf;
    }
    if (p.aux_out) {
      if (p.aux_bf16)
        *(u32x2_t*)((bf16_t*)p.aux_out + (long)m * p.ldaux + n) = (u32x2_t){f32x2_to_bf16x2(s[0], s[1]), f32x2_to_bf16x2(s[2], s[3])};
      else
        *(f32x4_t*)(p.aux_out + (long)m * p.ldaux + n) = s;
    }
  } else if (EPI == EPI_GELU_BWD || (EPI == EPI_GENERIC && p.act == EGV_ACT_GELU_BWD)) {
    f32x4_t zv;
    if (p.aux_bf16) {
      const u32x2_t zb = *(const u32x2_t*)((const bf16_t*)p.aux_in + (long)m * p.ldaux + n);
      zv = (f32x4_t){__uint_as_float(zb[0] << 16), __uint_as_float(zb[0] & 0xffff0000u), __uint_as_float(zb[1] << 16),
                     __uint_as_float(zb[1] & 0xffff0000u)};
    } else {
      zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= (p.aux_bf16 == 2) ? zv[e] : gelu_grad_f(zv[e]);
  } else if (EPI == EPI_GENERIC && p.act == EGV_ACT_RELU_BWD) {
    const f32x4_t zv = *(const f32x4_t*)(p.aux_in + (long)m * p.ldaux + n);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = zv[e] > 0.f ? v[e] : 0.f;
  }
  if (p.residual) v += *(const f32x4_t*)(p.residual + (long)m * p.ldr + n);
  if (p.out_f32) *(f32x4_t*)(p.out_f32 + (long)m * p.ldo + n) = v;
  if (p.out_hi) {
    bf16_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split_bf16(v[e], h[e], l[e]);
    *(u32x2_t*)(p.out_hi + (long)m * p.ldoh + n) = (u32x2_t){pack2(h[0], h[1]), pack2(h[2], h[3])};
    if (p.out_lo) *(u32x2_t*)(p.out_lo + (long)m * p.ldoh + n) = (u32x2_t){pack2(l[0], l[1]), pack2(l[2], l[3])};
  }
}

// MIXED (F3, MF = 5 only): row bands of TWO heights in one launch.  The first nb5 bands are 320 rows, the rest 256: a 256-row tile
// runs in the 320-row instance with the fifth 16-row fragment group of every wave switched off (its MFMAs are skipped under a
// wave-uniform branch, its LDS rows simply stay unused), i.e. in 4/5 of the time.  Tile ids are dealt round by round (round r =
// ids [r G, (r + 1) G), XCD remap inside a round), bands in id order, so every workgroup gets the tall tiles in its early rounds
// and the short ones in its last: M = 25 120 x N = 2304 is 53 + 32 bands = 765 tiles = 5 + 5 + 4 units per workgroup where 79
// bands of 320 rows are 711 tiles = 3 rounds of 5 (-6.7 %); N = 3072: 5 + 5 + 5 + 4 instead of 4 x 5 (-5 %).
// PROD: 0 = one product per k-tile from single planes (plain loops), 3 = fused bf16x3 (A_hi B_lo + A_lo B_hi + A_hi B_hi on the bf16
// MFMA), 2 = f16x2 (A_1 B_1 + A_2 B_2 on the fp16 MFMA, csrc/f16x2.h): the same loop with one pass less and the other opcode;
// 1 = the plain NT loop on the fp16 opcode (single fp16 planes).
template <int MF, bool TN, int EPI, int PROD, bool MIXED = false>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(const egv_gemm_desc p, const int dbg_arg, const int nb5_arg) {
  constexpr bool X2 = PROD == 2;
  constexpr bool H1 = PROD == 1;      // plain loop, fp16 opcode
  constexpr bool F3 = PROD >= 2;      // the fused stage layout ([A_hi | A_lo | B_hi | B_lo] x 64 B, 32-deep k-tiles)
  constexpr bool IS_GELU = EPI == EPI_GELU || EPI == EPI_GELU_X2;
  static_assert(PROD >= 0 && PROD <= 3, "");
  static_assert(!TN || PROD == 0 || PROD == 1, "weight gradients are single products: bf16, or fp16 (PROD 1: the fp16 backward)");
  static_assert(!MIXED || (F3 && MF == 5 && !TN), "mixed row bands: the fused NT instances with 320-row tiles");
  static_assert(EPI != EPI_GELU_X2 || X2 || H1, "fp16 operand outputs are produced by the fp16 instances only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef EGV_DIAG
  const int dbg = dbg_arg;      // `make diag` build only (tools/gemm_trace.py, tools/gemm_bench.py with EGV_GEMM_DBG)
#else
  constexpr int dbg = 0;        // product library: every diagnostic branch below folds away
#endif
  constexpr int BM = MF * 64;
  constexpr int A_BYTES = BM * 128;
  constexpr int B_BYTES = BNB * 128;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NCH = NFW / NC;               // phases per 32-deep k-step
  constexpr int GA = TN ? 4 : MF;             // DMA pieces per wave per tile: A, B
  constexpr int GB = 4;
  static_assert(!TN || MF == 4, "TN tiles are 256x256");
  static_assert(!(TN && F3), "the fused three-product loop is an NT loop");
  // F3 (NT, passes == 3): the FUSED three-product main loop.  A k-tile is 32 deep and a stage holds BOTH planes of both
  // operands, 64-B rows: [A_hi BM][A_lo BM][B_hi 256][B_lo 256] x 64 B (the same 72 KiB as a 64-deep single-plane stage).
  // Every fragment pair is fetched once and multiplied three times (hi.lo, lo.hi, hi.hi): 26 ds_read_b128 and 18 DMA pieces
  // per 120 MFMAs of a wave, where the three k-segments of the plain loop spend 26 and 18 per 80, and one k-tile barrier per 120.
  constexpr int KTD = F3 ? 32 : KT;
  constexpr int F3_ALO = BM * 64, F3_B = BM * 128, F3_BLO = 256 * 64;    // plane offsets inside an F3 stage

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // DIAGNOSTIC (EGV_GEMM_DBG=200, tools/gemm_trace.py): per-tile 100 MHz timestamps into p.aux_out
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
  // DIAGNOSTIC (`make diag` build, + EGV_GEMM_DBG bit 0x4000): workgroup 0 stamps s_memtime around the
  // k-tile hand-over of its first output tile, per wave, into 16 KiB of LDS behind the two stages ([wave][k-tile < 64]
  // [4 stamps]); dumped after the tile.  Compiled out of the product library.
#ifdef EGV_GEMM_STAMPS
  const bool stamp_on = (dbg & 0x4000) && blockIdx.x == 0;
#else
  constexpr bool stamp_on = false;
#endif
  unsigned long long* stamp_lds = (unsigned long long*)(smem + 2 * STAGE) + wave * 256;
  auto stamp = [&](int t, int i) {
    if (stamp_on && t < 64 && lane == 0) stamp_lds[t * 4 + i] = __builtin_amdgcn_s_memtime();
  };
  const int tiles_n = (p.N + BNB - 1) / BNB;
  const int nb5 = MIXED ? nb5_arg : 0;
  const int tiles_m = MIXED ? nb5 + (p.M - nb5 * 320 + 255) / 256 : (p.M + BM - 1) / BM;
  const int nwg = tiles_m * tiles_n;
  const int ksplit = p.ksplit > 1 ? p.ksplit : 1;
  const int total = nwg * ksplit;
  const int nkt_total = (p.K + KTD - 1) / KTD;
  const int kt_per = (nkt_total + ksplit - 1) / ksplit;
  // the plain loops run a multi-product GEMM as k-SEGMENTS over the planes (one product per segment, 64-deep k-tiles, the same accumulators):
  // three for split-bf16 (diagnostics builds: the product path is the fused F3 loop), TWO for f16x2 operands on the fp16 instance
  // (H1, passes == 2: A_2 B_2 then A_1 B_1 -- a single-product GEMM over the concatenation [A_2 | A_1] . [B_2 | B_1]^T; for short
  // contractions it beats the fused two-product loop, whose 32-deep k-tiles pay twice the k-tile hand-overs)
  const int nseg = (!F3 && p.passes == 3) ? 3 : ((H1 && p.passes == 2) ? 2 : 1);

  // k-segments: (A_hi,B_lo), (A_lo,B_hi), (A_hi,B_hi) for passes == 3; (A_lo,B_lo), (A_hi,B_hi) for f16x2; (A_hi,B_hi) alone otherwise
  auto seg_a = [&](int s) -> const bf16_t* { return ((nseg == 3 && s == 1) || (nseg == 2 && s == 0)) ? p.a_lo : p.a_hi; };
  auto seg_b = [&](int s) -> const bf16_t* { return ((nseg == 3 && s == 0) || (nseg == 2 && s == 0)) ? p.b_lo : p.b_hi; };

  // ---- DMA (global -> LDS) source offsets, in elements, relative to the tile origin of the current k-tile -----
  long a_voff, b_voff;
  unsigned nt_avo = 0, nt_bvo = 0, avo = 0, bvo = 0;
  // MIXED: in a 256-row tile the loader wave w owns global rows w * 64 .. (LDS rows stay w * 80 ..): 16 rows less per wave index
  const unsigned avo_adj4 = MIXED ? (unsigned)((wave & 3) * 16 * p.lda * 2) : 0u;
  int tn_col = 0;   // TN: this lane's source column (elements) before the per-piece unit XOR
  if (!TN) {
    const int chunk = (lane & 7) ^ (lane >> 3);                 // LDS position (lane&7) holds source chunk pos^(row&7)
    a_voff = b_voff = 0;
    // byte offset of this lane's 16 B within the (BM or 256) x 64 k-tile, first piece of this wave's two shares; the
    // pieces follow at 8 rows each.  32 bits are enough (320 rows x lda x 2 B), so the DMA addresses are a scalar base
    // (tile origin of the k-tile) + one VGPR: no 64-bit per-lane pointers, no per-piece scalar pairs.
    nt_avo = (unsigned)(((lane >> 3) + 2 * (wave & 3) * GA * 8) * p.lda * 2 + chunk * 16);
    nt_bvo = (unsigned)(((lane >> 3) + 2 * (wave & 3) * GB * 8) * p.ldb * 2 + chunk * 16);
    if (F3) {
      // 64-B rows: a DMA piece (1 KiB) is 16 rows x 4 chunks; LDS position (lane & 3) of row r holds source chunk
      // pos ^ g((r >> 2) & 3), g = {0, 2, 3, 1} -- the permutation that makes ds_read_b128 of 16 consecutive 64-B rows
      // conflict-free for the four lane groups the instruction is serviced in (MI355X_MICROARCH.md, LDS table)
      const int c3 = (lane & 3) ^ ((0x78 >> (2 * ((lane >> 4) & 3))) & 3);
      nt_avo = (unsigned)(((lane >> 2) + (wave & 3) * GA * 16) * p.lda * 2 + c3 * 16);
      nt_bvo = (unsigned)(((lane >> 2) + (wave & 3) * GB * 16) * p.ldb * 2 + c3 * 16);
    }
  } else {
    // piece I = wave*4 + q covers k-rows 2I, 2I+1 of the tile; this lane: row 2I + (lane>>5), LDS chunk lane&31.
    // source 32-B unit = (LDS unit) ^ (row & 7) = ((lane&31)>>1) ^ ((2q + (lane>>5)) & 7); the q part is XOR-ed in per piece.
    const int r = lane >> 5;
    const int u = ((lane & 31) >> 1) ^ r;
    tn_col = ((u << 1) | (lane & 1)) * 8;
    a_voff = (long)r * p.lda;                             // + (piece row base) * lda per piece
    b_voff = (long)r * p.ldb;
  }
  // Which waves issue the LDS-DMA: only waves 0-3 (one per SIMD), two 1/8 shares of the tile each, so that after the
  // k-tile barrier the OTHER wave of every SIMD goes straight back to its MFMAs instead of all eight queueing ~9 DMA
  // issues (~60 cycles each) with the matrix pipes idle: main loop 91 -> 81.5 us on fc2-forward (tools/gemm_trace.py).
  // EGV_GEMM_DBG bit 16 restores "every wave stages its own share" (A/B diagnostics).
#ifndef EGV_TN_DMA_PH
#define EGV_TN_DMA_PH 0
#endif
  constexpr int DMA_PH = TN ? EGV_TN_DMA_PH : 4;     // NT: the DMA of k-tile t+1 is issued over the first 4 phases of k-tile t (see main loop)
  constexpr bool SPREAD_A = !TN;         // NT: the A fragments of k-step 1 are fetched over phases 0-2 of k-step 0
  const bool loader = wave < 4;
  const int vw0 = 2 * (wave & 3);

  // ---- fragment read offsets (bytes within a stage) ---------------------------------------------------------
  int a_rd0, a_rd1, b_rd0, b_rd1;   // NT: k-step 0 / 1 bases;  TN: a_rd0 / b_rd0 only (k-step is an immediate)
  int tn_r7 = 0;
  if (!TN) {
    const int c0 = ((lane >> 4) ^ (lane & 7)) * 16;             // chunk (g + 4*ks) ^ (row&7), row&7 == lane&7
    const int rowb = (lane & 15) * 128;
    a_rd0 = (wm * MF * 16) * 128 + rowb + c0;
    a_rd1 = (wm * MF * 16) * 128 + rowb + (c0 ^ 64);
    b_rd0 = A_BYTES + (wn * 128) * 128 + rowb + c0;
    b_rd1 = A_BYTES + (wn * 128) * 128 + rowb + (c0 ^ 64);
    if (F3) {   // 64-B rows, chunk (lane >> 4) ^ g((row >> 2) & 3); k-step 1 does not exist (a_rd1 / b_rd1 unused)
      const int c3 = ((lane >> 4) ^ ((0x78 >> (2 * ((lane >> 2) & 3))) & 3)) * 16;
      a_rd0 = (wm * MF * 16 + (lane & 15)) * 64 + c3;
      b_rd0 = F3_B + (wn * 128 + (lane & 15)) * 64 + c3;
    }
  } else {
    const int g = lane >> 4, pp = lane & 15;
    tn_r7 = 4 * (g & 1) + (pp >> 2);
    const int rowoff = (4 * g + (pp >> 2)) * 512 + (pp & 3) * 8;
    a_rd0 = rowoff;
    b_rd0 = A_BYTES + rowoff;
    a_rd1 = b_rd1 = 0;
  }

  // TN keeps ONE fragment set: the pending halves of the asm reads are its second buffer (commit happens after the
  // phase's MFMAs), which keeps the kernel under 256 VGPRs without spills -- a spilled pending half would be read early.
  bf16x8_t A[TN ? 1 : 2][MF], Bq[TN ? 1 : 3][NC];
  // NT: plain LDS loads straight into the destination fragments (issue = load, commit = nothing).
  // TN: asm transpose reads into pending halves (issue), combined into the fragment after the hand-placed wait (commit).
  u32x2_t pa[TN ? MF : 1][2], pb[TN ? NC : 1][2];
  const unsigned lds0 = (unsigned)(size_t)(const __attribute__((address_space(3))) char*)smem;
  const unsigned fa = lds0 + a_rd0, fb = lds0 + b_rd0;   // NT asm fragment reads (the stages are 128-B aligned: k-step 1 = ^ 64)
  auto issue_a = [&](int sb, int ks, bf16x8_t (&dst)[MF], int f0 = 0, int f1 = MF) {
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      if (f < f0 || f >= f1) continue;
      if (!TN) {
        dst[f] = lds_b128(smem + sb + (ks ? a_rd1 : a_rd0) + f * 2048);
      } else {
        const unsigned ad = lds0 + sb + a_rd0 + (((wm * 4 + f) ^ tn_r7) << 5);
        if (ks == 0) {
          pa[TN ? f : 0][0] = tr_asm<0>(ad);
          pa[TN ? f : 0][1] = tr_asm<16 * 512>(ad);
        } else {
          pa[TN ? f : 0][0] = tr_asm<32 * 512>(ad);
          pa[TN ? f : 0][1] = tr_asm<48 * 512>(ad);
        }
      }
    }
  };
  auto issue_b = [&](int sb, int ks, int c, bf16x8_t (&dst)[NC]) {
#pragma unroll
    for (int jj = 0; jj < NC; ++jj) {
      const int j = c * NC + jj;
      if (!TN) {
        dst[jj] = lds_b128(smem + sb + (ks ? b_rd1 : b_rd0) + j * 2048);
      } else {
        const unsigned ad = lds0 + sb + b_rd0 + (((wn * 8 + j) ^ tn_r7) << 5);
        if (ks == 0) {
          pb[TN ? jj : 0][0] = tr_asm<0>(ad);
          pb[TN ? jj : 0][1] = tr_asm<16 * 512>(ad);
        } else {
          pb[TN ? jj : 0][0] = tr_asm<32 * 512>(ad);
          pb[TN ? jj : 0][1] = tr_asm<48 * 512>(ad);
        }
      }
    }
  };
  auto commit_a = [&](bf16x8_t (&dst)[MF]) {
    if (TN) {
#pragma unroll
      for (int f = 0; f < MF; ++f)
        dst[f] = __builtin_bit_cast(bf16x8_t, (u32x4_t){pa[TN ? f : 0][0][0], pa[TN ? f : 0][0][1], pa[TN ? f : 0][1][0],
                                                        pa[TN ? f : 0][1][1]});
    }
  };
  auto commit_b = [&](bf16x8_t (&dst)[NC]) {
    if (TN) {
#pragma unroll
      for (int jj = 0; jj < NC; ++jj)
        dst[jj] = __builtin_bit_cast(bf16x8_t, (u32x4_t){pb[TN ? jj : 0][0][0], pb[TN ? jj : 0][0][1],
                                                         pb[TN ? jj : 0][1][0], pb[TN ? jj : 0][1][1]});
    }
  };
  auto tn_wait = [&]() {
    if (TN) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // 1.0 in the operand type of this instance (bf16 0x3F80, fp16 0x3C00): the column-sum MFMA of the TN loop
  constexpr short ONE16 = H1 ? (short)0x3C00 : (short)0x3F80;
  const s16x8v ones_s = {ONE16, ONE16, ONE16, ONE16, ONE16, ONE16, ONE16, ONE16};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s);
  // one product of the plain loops (NT single plane, TN): the bf16 opcode, or -- H1 -- the fp16 one on the same registers
  auto mma1 = [&](const bf16x8_t& b, const bf16x8_t& a, f32x4_t c) -> f32x4_t {
    if constexpr (H1) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
  };

  // ---- per-output-tile state (all wave-uniform) ----------------------------------------------------------------
  int m0, n0, z, tn, kt_begin, kt_end, nt;    // current tile
  int st_seg, st_kt;                          // next k-tile to be staged (of the tile being staged)
  int sm0, sn0, skt_begin, skt_end;           // origin / k-range of the tile being staged
  int mf = MF, smf = MF, nmf = MF;            // MIXED: 16-row fragment groups per wave of the current / staged / next tile (5 or 4)
  auto decode = [&](int v, int& om0, int& on0, int& oz, int& otn, int& okb, int& oke, int& omf) {
    omf = MF;
    if constexpr (MIXED) {
      // round-wise dealing (see the template comment): ksplit == 1 here
      oz = 0;
      const int G = (int)gridDim.x;
      const int r = v / G;
      const int left = nwg - r * G;
      const int wg = r * G + xcd_remap(v - r * G, left < G ? left : G);
      const int tm = wg / tiles_n;
      otn = wg - tm * tiles_n;
      const bool five = tm < nb5;
      omf = five ? 5 : 4;
      om0 = min(five ? tm * 320 : nb5 * 320 + (tm - nb5) * 256, p.M - omf * 64);
      on0 = min(otn * BNB, p.N - BNB);
      okb = 0;
      oke = nkt_total;
      return;
    }
    int wg;
#ifndef EGV_TN_OLD_MAP
    if constexpr (TN) {
      // k-slices are the outer index of the work ids (id = z nwg + tile): remap the WHOLE id space of a round, so that an XCD (blocks
      // b % 8 == x) owns a contiguous id range = all tiles of ~one k-slice -- the 9 x 3 tiles of a slice read the same dY / X rows and
      // share them in that XCD's L2.  (Remapping inside each slice, as for NT, dealt every slice's tiles to all eight XCDs: PMC
      // FETCH_SIZE 464 MB per qkv wgrad for 154 MB of operands, profiles/r05_gemm_pmc_per_instance.txt.)
      const int G = (int)gridDim.x;
      const int r = v / G;
      const int left = total - r * G;
      const int w = r * G + xcd_remap(v - r * G, left < G ? left : G);
      oz = w / nwg;
      wg = w - oz * nwg;
    } else
#endif
    {
      oz = v / nwg;
      wg = xcd_remap(v - oz * nwg, nwg);
    }
    const int tm = wg / tiles_n;
    otn = wg - tm * tiles_n;
    om0 = min(tm * BM, p.M - BM);      // shifted, never predicated (host guarantees M >= BM, N >= 256)
    on0 = min(otn * BNB, p.N - BNB);
    okb = oz * kt_per;
    oke = min(nkt_total, okb + kt_per);
  };

  // One DMA piece (1 KiB = 64 lanes x 16 B) of the k-tile being staged.  A loader wave owns NP pieces per k-tile: the A
  // pieces of its two 1/8 shares first (activations stream from HBM), then the B pieces (weights, L2-resident).
  constexpr int NP = 2 * (GA + GB);
  unsigned run = 0;
  auto piece = [&](char* lds, int i, int i0) {   // pieces are issued in ascending runs [i0, ...)
    if (F3) {
      // pieces of this wave: A_hi [0, GA), A_lo [GA, 2 GA), B_hi [2 GA, 2 GA + GB), B_lo [2 GA + GB, NP); 16 rows each
      const int w = wave & 3;
      const bool isa = i < 2 * GA;
      const int pl = isa ? (i >= GA) : (i - 2 * GA >= GB);            // 0 = hi plane, 1 = lo plane
      const int j = isa ? (i - pl * GA) : (i - 2 * GA - pl * GB);       // piece index inside this wave's share of the plane
      if (isa) {
        const char* ab = (const char*)((pl ? p.a_lo : p.a_hi) + (long)sm0 * p.lda + (long)st_kt * KTD);
        if (i == i0 || j == 0) run = avo + (unsigned)(j * 32 * p.lda);
        // MIXED, 256-row tile: a wave's share is 64 rows (four pieces); the fifth would read past the band
        if (!(MIXED && j == GA - 1 && smf == 4)) glds16(ab + (size_t)run, lds + pl * F3_ALO + (w * GA + j) * 1024);
        run += (unsigned)(32 * p.lda);
      } else {
        const char* bb = (const char*)((pl ? p.b_lo : p.b_hi) + (long)sn0 * p.ldb + (long)st_kt * KTD);
        if (i == i0 || j == 0) run = bvo + (unsigned)(j * 32 * p.ldb);
        glds16(bb + (size_t)run, lds + F3_B + pl * F3_BLO + (w * GB + j) * 1024);
        run += (unsigned)(32 * p.ldb);
      }
      asm volatile("" : "+v"(run));
    } else if (!TN) {
      // scalar base of the k-tile + this lane's 32-bit offset; `run` is opaque after every piece so that the offsets are
      // produced one at a time (computed all at once they spill next to the live accumulators)
      if (i < 2 * GA) {
        const char* ab = (const char*)(seg_a(st_seg) + (long)sm0 * p.lda + (long)st_kt * KT);
        if (i == i0) run = avo + (unsigned)(i * 16 * p.lda);
        glds16(ab + (size_t)run, lds + (vw0 * GA + i) * 1024);
        run += (unsigned)(16 * p.lda);
      } else {
        const int j = i - 2 * GA;
        const char* bb = (const char*)(seg_b(st_seg) + (long)sn0 * p.ldb + (long)st_kt * KT);
        if (i == i0 || j == 0) run = bvo + (unsigned)(j * 16 * p.ldb);
        glds16(bb + (size_t)run, lds + A_BYTES + (vw0 * GB + j) * 1024);
        run += (unsigned)(16 * p.ldb);
      }
      asm volatile("" : "+v"(run));
    } else {
      const bool isb = i >= 8;
      const int j = i & 7;
      const int vw = vw0 + (j >> 2), q = j & 3;
      const int krow = st_kt * KT + vw * 8 + (lane >> 5) + 2 * q;
      const int col = tn_col ^ (((2 * q) & 7) << 4);   // 32-B unit (16 elements) index ^= (2q) & 7
      const void* za = (const char*)g_zero_page + lane * 16;
      if (!isb) {
        const bf16_t* sa = seg_a(st_seg) + (long)st_kt * KT * p.lda + sm0 + a_voff + (long)(vw * 8 + 2 * q) * p.lda + col;
        glds16(krow < p.K ? (const void*)sa : za, lds + (vw * 4 + q) * 1024);
      } else {
        const bf16_t* sb = seg_b(st_seg) + (long)st_kt * KT * p.ldb + sn0 + b_voff + (long)(vw * 8 + 2 * q) * p.ldb + col;
        glds16(krow < p.K ? (const void*)sb : za, lds + A_BYTES + (vw * 4 + q) * 1024);
      }
    }
  };
  auto stage_advance = [&]() {
    if (++st_kt == skt_end) {
      st_kt = skt_begin;
      ++st_seg;
    }
  };
  auto stage = [&](int buf) {   // one k-tile of tile (sm0, sn0) -> LDS stage `buf`
    char* lds = smem + buf * STAGE;
    avo = nt_avo - ((MIXED && smf == 4) ? avo_adj4 : 0u); bvo = nt_bvo;
    if (loader) {
#pragma unroll
      for (int i = 0; i < NP; ++i) piece(lds, i, 0);
    }
    stage_advance();
  };

  int v = blockIdx.x;
  if (v >= total) return;
  decode(v, m0, n0, z, tn, kt_begin, kt_end, mf);
  nt = max(kt_end - kt_begin, 0) * nseg;
  sm0 = m0; sn0 = n0; skt_begin = kt_begin; skt_end = kt_end; st_seg = 0; st_kt = kt_begin; smf = mf;
  if (nt > 0) stage(0);
  if (!TN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  for (;;) {
    if ((dbg & 0xfff) == 200) ts0 = __builtin_amdgcn_s_memrealtime();
    // ================= main loop of tile v: k-tile 0 is in flight or landed in stage 0 ==========================
    f32x4_t acc[MF][NFW];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
      for (int j = 0; j < NFW; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_cs = TN && p.colsum != nullptr && tn == 0 && wn == 0;
    f32x4_t cs[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) cs[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    if constexpr (F3) {
      // ---- fused NT main loop: three bf16 products, or the two fp16 products of f16x2 (see the F3 note at the top of the kernel) ----
      // Phase j (0..7) multiplies B fragment pair j (hi, lo) with the MF resident A pairs: 3 MF MFMAs in three passes over i
      // (hi.lo, lo.hi, hi.hi -- five independent accumulators between two MFMAs on the same one); f16x2: two passes, lo.lo and
      // hi.hi (plane 1 = "hi", plane 2 = "lo"), on the fp16 opcode -- same fetch plan, same waits.  Fetch plan (asm reads
      // complete in issue order; every wait counts the reads issued behind the ones it needs):
      //   phase 7 of the previous k-tile, behind the barrier:  B(0) pair | A_lo[0..MF) after pass 1 | A_hi[i-1] behind the
      //     hi.hi MFMA of i (its last reader is one MFMA back), A_hi[MF-1] last
      //   phase j < 7: B(j+1) pair first (into the set phase j-1 used)
      // The DMA of k-tile t+1 is issued by waves 0-3 over phases 0-3, as in the plain loop.
      bf16x8_t Ah[MF], Al[MF], Bh[2], Bl[2];
      const unsigned fa3 = lds0 + a_rd0, fb3 = lds0 + b_rd0;
      auto rd_b3 = [&](auto Jc, unsigned rb) {
        constexpr int j = decltype(Jc)::value;
        Bh[j & 1] = ld128_asm<j * 1024>(rb);
        Bl[j & 1] = ld128_asm<j * 1024 + F3_BLO>(rb);
      };
      auto rd_al = [&](unsigned ra) { static_for<0, MF>([&](auto Ic) { constexpr int i = decltype(Ic)::value; Al[i] = ld128_asm<i * 1024 + F3_ALO>(ra); }); };
      auto rd_ah = [&](auto Ic, unsigned ra) { constexpr int i = decltype(Ic)::value; Ah[i] = ld128_asm<i * 1024>(ra); };
      const bool grp5 = !MIXED || mf == 5;          // wave-uniform: does this tile use the fifth fragment group?
      auto mma = [&](const bf16x8_t& b, const bf16x8_t& a, f32x4_t c) -> f32x4_t {
        if constexpr (X2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, b), __builtin_bit_cast(f16x8_t, a), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c, 0, 0, 0);
      };
      auto pass = [&](auto Jc, const bf16x8_t& b, bf16x8_t (&a)[MF]) {
        constexpr int j = decltype(Jc)::value;
#ifdef EGV_F3_SKIP      // diagnostics build (WRONG results): the loop with one of its three products left out, everything else in place
        if (&a[0] == &Al[0]) return;
#endif
        static_for<0, MF>([&](auto Ic) {
          constexpr int i = decltype(Ic)::value;
          if (MIXED && i == MF - 1) {
            if (grp5) acc[i][j] = mma(b, a[i], acc[i][j]);
          } else {
            acc[i][j] = mma(b, a[i], acc[i][j]);
          }
        });
      };
      // the passes of one phase: bf16x3 (hi.lo, lo.hi, hi.hi) = (Bh, Al), (Bl, Ah), (Bh, Ah); f16x2 (lo.lo, hi.hi) = (Bl, Al), (Bh, Ah)
      auto pass_lo = [&](auto Jc, const int q) { pass(Jc, X2 ? Bl[q] : Bh[q], Al); };
      auto pass_mid = [&](auto Jc, const int q) { if constexpr (!X2) pass(Jc, Bl[q], Ah); };
      auto k_tile3 = [&](const int t) {
        const bool HN = t + 1 < nt;
        const int sb = (t & 1) * STAGE;
        const unsigned ra = fa3 + sb, rb = fb3 + sb;
        char* dma_lds = smem + (STAGE - sb);
        avo = nt_avo - ((MIXED && smf == 4) ? avo_adj4 : 0u); bvo = nt_bvo;
        asm volatile("" : "+v"(avo), "+v"(bvo));
        static_for<0, NFW>([&](auto Jc) {
          constexpr int j = decltype(Jc)::value;
          if constexpr (j < NFW - 1) rd_b3(std::integral_constant<int, j + 1>{}, rb);
          if (j < 4 && loader && HN) {
            constexpr int PP = (NP + 3) / 4;
            static_for<j * PP, (j + 1) * PP < NP ? (j + 1) * PP : NP>([&](auto Ic) { piece(dma_lds, decltype(Ic)::value, j * PP); });
          }
          if constexpr (j == 0) {
            lgkm_wait<MF + 2>();          // B(0) pair and A_lo landed; outstanding: A_hi x MF, B(1) pair
            tie(Bh[0]);
            tie(Bl[0]);
            static_for<0, MF>([&](auto Ic) { tie(Al[decltype(Ic)::value]); });
            __builtin_amdgcn_sched_barrier(0);
            pass_lo(Jc, 0);
            __builtin_amdgcn_sched_barrier(0);
            lgkm_wait<2>();               // A_hi landed
            tie(Bl[0]);
            static_for<0, MF>([&](auto Ic) { tie(Ah[decltype(Ic)::value]); });
            __builtin_amdgcn_sched_barrier(0);
            pass_mid(Jc, 0);
            pass(Jc, Bh[0], Ah);
            __builtin_amdgcn_sched_barrier(0);
          } else if constexpr (j < NFW - 1) {
            lgkm_wait<2>();               // B(j) pair landed; B(j+1) pair in flight
            tie(Bh[j & 1]);
            tie(Bl[j & 1]);
            __builtin_amdgcn_sched_barrier(0);
            pass_lo(Jc, j & 1);
            pass_mid(Jc, j & 1);
            pass(Jc, Bh[j & 1], Ah);
            __builtin_amdgcn_sched_barrier(0);
          } else {
            lgkm_wait<0>();
            tie(Bh[j & 1]);
            tie(Bl[j & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned na = fa3 + (STAGE - sb), nb = fb3 + (STAGE - sb);
            if (HN) {
              // k-tile t+1 (this wave's DMA pieces) has landed; every read of stage t & 1 by this wave has returned
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              stage_advance();
              rd_b3(std::integral_constant<int, 0>{}, nb);
            }
            __builtin_amdgcn_sched_barrier(0);
            pass_lo(Jc, j & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (HN) rd_al(na);            // A_lo has no reader left in this k-tile
            __builtin_amdgcn_sched_barrier(0);
            pass_mid(Jc, j & 1);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, MF>([&](auto Ic) {
              constexpr int i = decltype(Ic)::value;
              if (MIXED && i == MF - 1) {
                if (grp5) acc[i][j] = mma(Bh[j & 1], Ah[i], acc[i][j]);
              } else {
                acc[i][j] = mma(Bh[j & 1], Ah[i], acc[i][j]);
              }
              __builtin_amdgcn_sched_barrier(0);
              if constexpr (i > 0) {
                if (HN) rd_ah(std::integral_constant<int, i - 1>{}, na);     // its last reader is the MFMA before this one
              }
              __builtin_amdgcn_sched_barrier(0);
            });
            if (HN) rd_ah(std::integral_constant<int, MF - 1>{}, na);
            __builtin_amdgcn_sched_barrier(0);
          }
        });
      };
      if (nt > 0) {
        // k-tile 0 has landed and every wave is past a barrier behind that: fetch its B(0) pair, A_lo, A_hi (the order the
        // waits of phase 0 count on)
        rd_b3(std::integral_constant<int, 0>{}, fb3);
        rd_al(fa3);
        static_for<0, MF>([&](auto Ic) { rd_ah(Ic, fa3); });
        if ((dbg & 0xfff) == 200) ts1 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
        for (int t = 0; t < nt; ++t) k_tile3(t);
      }
    } else if constexpr (!TN) {
      // ---- NT main loop ------------------------------------------------------------------------------------------------
      // Phase q = (k-step q/4, B column pair q%4) multiplies MF x 2 fragments: A set q/4, B set q%3.  Fetch plan per
      // k-tile (reads complete in issue order, so every wait is a count of the reads issued after the ones needed):
      //   hand-over (phase 7 of the previous k-tile, after the barrier): B(0), A(k-step 0)
      //   phase 0: B(1) B(2) A1[0,1]   phase 1: B(3) A1[2,3]   phase 2: B(4) A1[4]   phase 3: B(5)   phase 4: B(6)
      //   phase 5: B(7)                phase 6: -               (B two phases ahead, A(k-step 1) four)
      // The DMA of k-tile t+1 (into the stage freed by the barrier that ended t-1) is issued by waves 0-3 over phases
      // 0-3, a few pieces each: a burst of all NP blocks the issuing wave ~1300 cycles on the 64 B/clk L1->LDS path.
      auto k_tile = [&](const int t) {
        const bool HN = t + 1 < nt;            // wave-uniform: scalar branches around the DMA issue and the hand-over
        const int sb = (t & 1) * STAGE;
        // fa / fb: this lane's A / B fragment address in stage 0, k-step 0 (kernel lifetime); + sb, ^ 64 for k-step 1
        const unsigned ra0 = fa + sb, rb0 = fb + sb;
        char* dma_lds = smem + (STAGE - sb);
        // opaque per iteration: keeps the NP per-piece offsets (avo + i * 16 lda) from being hoisted into NP live registers
        avo = nt_avo; bvo = nt_bvo;
        asm volatile("" : "+v"(avo), "+v"(bvo));
        constexpr int nA2 = MF > 4 ? 1 : 0;
        static_for<0, 2 * NCH>([&](auto PHc) {
          constexpr int PH = decltype(PHc)::value, ks = PH / NCH, c = PH % NCH;
          auto rd_b = [&](auto Qc) {   // B fragments of phase Q -> register set Q % 3
            constexpr int Q = decltype(Qc)::value, qs = Q / NCH, qc = Q % NCH;
            const unsigned rb = qs ? rb0 ^ 64u : rb0;
            Bq[Q % 3][0] = ld128_asm<(qc * NC) * 2048>(rb);
            Bq[Q % 3][1] = ld128_asm<(qc * NC + 1) * 2048>(rb);
          };
          auto rd_a1 = [&](auto Fc) {
            constexpr int F = decltype(Fc)::value;
            if constexpr (F < MF) A[1][F] = ld128_asm<F * 2048>(ra0 ^ 64u);
          };
          if constexpr (PH == 0) {
            rd_b(std::integral_constant<int, 1>{});
            rd_b(std::integral_constant<int, 2>{});
            rd_a1(std::integral_constant<int, 0>{});
            rd_a1(std::integral_constant<int, 1>{});
          } else if constexpr (PH <= 5) {
            rd_b(std::integral_constant<int, PH + 2>{});
            if constexpr (PH == 1) {
              rd_a1(std::integral_constant<int, 2>{});
              rd_a1(std::integral_constant<int, 3>{});
            } else if constexpr (PH == 2) {
              rd_a1(std::integral_constant<int, 4>{});
            }
          }
          if (PH < 4 && loader && HN) {
            constexpr int PP = (NP + 3) / 4;
            static_for<PH * PP, (PH + 1) * PP < NP ? (PH + 1) * PP : NP>([&](auto Ic) { piece(dma_lds, decltype(Ic)::value, PH * PP); });
          }
          constexpr int WAITS[8] = {6, 8, 8 + nA2, 6 + nA2, 4, 4, 2, 0};
          lgkm_wait<WAITS[PH]>();
          tie(Bq[PH % 3][0]);
          tie(Bq[PH % 3][1]);
          if constexpr (c == 0) static_for<0, MF>([&](auto Ic) { tie(A[ks][decltype(Ic)::value]); });
          __builtin_amdgcn_sched_barrier(0);
          acc[0][c * NC] = mma1(Bq[PH % 3][0], A[ks][0], acc[0][c * NC]);
          if (PH == 2 * NCH - 1 && HN) {
            __builtin_amdgcn_sched_barrier(0);
            // k-tile t+1 (this wave's DMA pieces) landed; every read of stage t&1 by this wave has returned (lgkmcnt(0) above)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage_advance();
            const unsigned na0 = fa + (STAGE - sb), nb0 = fb + (STAGE - sb);
            Bq[0][0] = ld128_asm<0>(nb0);
            Bq[0][1] = ld128_asm<2048>(nb0);
            static_for<0, MF>([&](auto Ic) { A[0][decltype(Ic)::value] = ld128_asm<decltype(Ic)::value * 2048>(na0); });
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 0; jj < NC; ++jj)
#pragma unroll
            for (int i = 0; i < MF; ++i)
              if (jj + i > 0) acc[i][c * NC + jj] = mma1(Bq[PH % 3][jj], A[ks][i], acc[i][c * NC + jj]);
          __builtin_amdgcn_sched_barrier(0);
        });
      };
      if (nt > 0) {
        // k-tile 0 has landed and every wave is past a barrier behind that (before the tile loop for the first tile, at the
        // end of the previous epilogue otherwise) -- no wait here: the epilogue stores of the previous tile are still draining
        const unsigned na0 = fa, nb0 = fb;
        Bq[0][0] = ld128_asm<0>(nb0);
        Bq[0][1] = ld128_asm<2048>(nb0);
        static_for<0, MF>([&](auto Ic) { A[0][decltype(Ic)::value] = ld128_asm<decltype(Ic)::value * 2048>(na0); });
        if ((dbg & 0xfff) == 200) ts1 = __builtin_amdgcn_s_memrealtime();
        // NOT unrolled: two k-tile bodies in one loop cost ~15 registers the MF = 5 instance does not have (A fragments spill)
#pragma unroll 1
        for (int t = 0; t < nt; ++t) k_tile(t);
      }
    } else {
    if (nt > 0) {
      if (nt > 1) {
        if (!DMA_PH) stage(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      issue_b(0, 0, 0, Bq[0]);
      issue_a(0, 0, A[0]);
      tn_wait();
      commit_b(Bq[0]);
      commit_a(A[0]);
    }
    if ((dbg & 0xfff) == 200) ts1 = __builtin_amdgcn_s_memrealtime();

    int cur_seg = 0, cur_kt = kt_begin;
    for (int t = 0; t < nt; ++t) {
      const int sb = (t & 1) * STAGE;
      const bool cs_on = do_cs && (nseg == 1 || cur_seg >= 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ph = ks * NCH + c;
          const bool last = (ph == 2 * NCH - 1);
          // hipcc waits lgkmcnt(0) right before the first MFMA that consumes a prefetched fragment, so the prefetch of
          // phase p+1 is issued AFTER the first MFMA of phase p (it then has the rest of the phase to land) -- issued
          // before it, every phase would start by waiting for the reads it had just issued (seen in the .s).
          if (c == 0 && cs_on) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
              cs[i] = mma1(ones, A[TN ? 0 : (ks & 1)][i], cs[i]);
          }
          acc[0][c * NC] = mma1(Bq[TN ? 0 : (ph % 3)][0], A[TN ? 0 : (ks & 1)][0], acc[0][c * NC]);
          __builtin_amdgcn_sched_barrier(0);
          if (last) {
            if (t + 1 < nt) {
              // tile t+1 (this wave's DMA pieces) landed; all of this wave's reads of stage t&1 have returned
              stamp(t, 0);
              asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
              stamp(t, 1);
              if (!(dbg & 0x1000)) __builtin_amdgcn_s_barrier();          // DIAGNOSTIC bits (timing only, results invalid):
              stamp(t, 2);
              if (DMA_PH) stage_advance();
              else if (t + 2 < nt && !(dbg & 0x2000)) stage(t & 1);       // 0x1000 no k-tile barrier, 0x2000 no DMA in the loop
              stamp(t, 3);
              issue_b(STAGE - sb, 0, 0, Bq[0]);
              issue_a(STAGE - sb, 0, A[0]);
            }
          } else {
            if (DMA_PH && ph < DMA_PH && loader && t + 1 < nt) {
              // k-tile t+1 -> the other stage (free since the barrier that ended k-tile t-1), a few pieces per phase: a
              // burst of NP DMA instructions blocks the issuing wave for ~1300 cycles on the 64 B/clk L1->LDS path
              // (s_memtime stamps, profiles/r01_d_handover_stamps.txt) while its SIMD partner runs alone
              char* lds = smem + (STAGE - sb);
              constexpr int PP = (NP + DMA_PH - 1) / (DMA_PH ? DMA_PH : 1);
#pragma unroll
              for (int i = 0; i < NP; ++i)
                if (i >= ph * PP && i < (ph + 1) * PP) piece(lds, i, ph * PP);
            }
            const int ks2 = (c + 1 < NCH) ? ks : ks + 1;
            const int c2 = (c + 1 < NCH) ? c + 1 : 0;
            if (TN) {
              issue_b(sb, ks2, c2, Bq[0]);
            } else {
              // NT: B fragments are fetched TWO phases ahead into a ring of three register sets (phase q uses set q % 3):
              // one phase of MFMAs (~160 cycles) does not cover an LDS read under load (~200), two do.  The hand-over
              // phase can only fetch phase 0 of the next k-tile (after the barrier), so phase 0 fetches phases 1 and 2.
              if (ph == 0) issue_b(sb, 1 / NCH, 1 % NCH, Bq[TN ? 0 : 1]);
              if (ph + 2 < 2 * NCH) issue_b(sb, (ph + 2) / NCH, (ph + 2) % NCH, Bq[TN ? 0 : ((ph + 2) % 3)]);
            }
            if (SPREAD_A && ks == 0) {
              if (c < 3) issue_a(sb, 1, A[TN ? 0 : 1], 2 * c, 2 * c + 2);
            } else if (c2 == 0) {
              issue_a(sb, ks2, A[TN ? 0 : (ks2 & 1)]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int jj = 0; jj < NC; ++jj)
#pragma unroll
            for (int i = 0; i < MF; ++i)
              if (jj + i > 0)
                acc[i][c * NC + jj] = mma1(Bq[TN ? 0 : (ph % 3)][jj], A[TN ? 0 : (ks & 1)][i], acc[i][c * NC + jj]);
          __builtin_amdgcn_sched_barrier(0);
          if (TN) {   // the asm reads issued above have had this phase's MFMAs to land: wait, then assemble the fragments
            if (last) {
              if (t + 1 < nt) {
                tn_wait();
                commit_b(Bq[0]);
                commit_a(A[0]);
              }
            } else {
              tn_wait();
              commit_b(Bq[0]);
              if (c + 1 >= NCH) commit_a(A[0]);
            }
          }
        }
      }
      if (++cur_kt == kt_end) {
        cur_kt = kt_begin;
        ++cur_seg;
      }
    }
    }
    if ((dbg & 0xfff) == 200) ts2 = __builtin_amdgcn_s_memrealtime();
    if (stamp_on && v == 0) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      unsigned long long* dst = (unsigned long long*)p.aux_out + 16384 + wave * 256;
      for (int i = lane; i < 256; i += 64) dst[i] = stamp_lds[i];
    }

    // ================= hand-over: every wave is done with both LDS stages =========================================
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int vn = v + gridDim.x;
    const bool has_next = vn < total;
    int nm0 = 0, nn0 = 0, nz = 0, ntn = 0, nkb = 0, nke = 0, nnt = 0;

    // ================= epilogue of tile v ===========================================================================
    // Fragment (i, j) of this wave: lane holds C[m = 16 i + (lane & 15)][n = 16 j + 4 (lane >> 4) .. +3].  A 16-row group
    // (fixed i, all 8 j: 16 x 128 fp32 = 8 KiB) goes through this wave's private staging rows in LDS stage 1 (512-B rows,
    // 16-B chunk index XOR (row & 7): conflict-free for the 8-lane ds_write_b128 groups and for the row-wise reads) and is
    // read back ROW-WISE, so that every global access of the epilogue is a run of whole 128-B lines:
    //   f32 layout   (no plane output): lane -> 4 columns, 32 lanes = one 512-B row, 2 rows per wave instruction;
    //   plane layout (out_hi given):    lane -> 8 columns (one 16-B store per bf16 plane), 16 lanes = one row, 4 rows.
    // The first version stored 16-column pieces (32-B bf16 / 64-B fp32 segments) and re-loaded the bias in each of its 40
    // rolled iterations -- on gfx9 loads and stores retire through ONE in-order counter, so waiting for that load waited for
    // every store issued before it: the epilogue ran at one store round trip per iteration and cost 35-40 % of the K = 768
    // GEMMs (profiles/r01_e_gemm_ceiling.txt).  Now no load is ever issued behind a store whose latency it would expose:
    // the bias is read once per tile, residual / GELU' inputs are prefetched a 16-row group ahead (two register sets), and
    // the stores of tile v drain under the main loop of tile v + 1 (counted vmcnt at the end, see below).
    constexpr int EPW = 16 * 512;
    static_assert(8 * EPW <= STAGE, "epilogue staging must fit in one LDS stage");
    // every lane-derived address of the epilogue hangs off `el`, opaque per tile: hoisted out of the tile loop they would
    // be live across the main loop, which has no register to spare (the A / B fragments spill)
    // (the lane id is rebuilt from EXEC here -- two VALU ops -- instead of being kept in a register across the main loop)
    int el = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(el));
    char* const ep = smem + STAGE + wave * EPW;
    const int mfe = MIXED ? mf : MF;            // fragment groups of this tile
    const int mw = m0 + wm * mfe * 16;
    const int nw = n0 + wn * 128;
    const int wsw = el & 7;
    char* const wbase = ep + (el & 15) * 512 + (((el >> 4) ^ (wsw & 3)) << 4);
    char* const w_even = wbase + ((wsw >> 2) << 6);
    char* const w_odd = wbase + (((wsw >> 2) ^ 1) << 6);
    auto put = [&](auto Ic) {   // accumulators of group i -> staging rows
      constexpr int i = decltype(Ic)::value;
      static_for<0, NFW>([&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        *(f32x4_t*)(((j & 1) ? w_odd : w_even) + ((j & ~1) << 6)) = acc[i][j];
      });
      // a DISTINCT statement at the end of every group's block: without it SimplifyCFG sinks the "identical" stores of the
      // switch cases below into one block that indexes acc[] with a phi -- and the whole accumulator array moves to scratch
      asm volatile("; accumulators of group %0 staged" ::"n"(i) : "memory");
    };
    // The group loop is ROLLED (the body is shared by all MF groups; only `put` depends on i, through a switch), so the
    // epilogue stays a few KB of code whatever its flavour.
    auto put_i = [&](int i) {
      switch (i) {
        case 0: put(std::integral_constant<int, 0>{}); break;
        case 1: put(std::integral_constant<int, 1>{}); break;
        case 2: put(std::integral_constant<int, 2>{}); break;
        case 3: put(std::integral_constant<int, 3>{}); break;
        default: if constexpr (MF > 4) put(std::integral_constant<int, MF - 1>{}); break;
      }
    };
    const bool lay16 = (EPI != EPI_RAW) && p.out_hi != nullptr;
    // f32 layout: row = 2 it + (lane >> 5), columns 4 (lane & 31) ..; plane layout: row = 4 it + (lane >> 4), columns 8 (lane & 15) ..
    const int L32 = el & 31, rs32 = el >> 5, L16 = el & 15, rs16 = el >> 4;
    const unsigned r32 = (unsigned)(size_t)(ep - smem) + rs32 * 512 + ((L32 ^ rs32) << 4);          // + it * 1024, ^ ((it & 3) << 5)
    const unsigned r16 = (unsigned)(size_t)(ep - smem) + rs16 * 512 + (((2 * L16) ^ rs16) << 4);    // + it * 2048, ^ ((it & 1) << 6)
    // Streams that have to be READ by the epilogue (the residual of proj / fc2 forward; the saved pre-activation of the
    // fc2 dgrad) are loaded a whole 16-row group at a time, at the top of the group: within the group no load is issued
    // behind a store, so the only store round trip a wave waits for is the previous group's (MF per tile, not 8 MF).
    const bool pf_res = (EPI == EPI_LINEAR) && !lay16 && p.residual != nullptr;
    const bool pf_aux = (EPI == EPI_GELU_BWD) && lay16 && p.aux_bf16 != 0 && p.aux_in != nullptr;
    constexpr int NPF = (EPI == EPI_LINEAR) ? 8 : ((EPI == EPI_GELU_BWD) ? 4 : 1);
    f32x4_t pre[NPF];
    const float* pf_ptr = nullptr;       // this lane's address in the first row of the group to be fetched next
    long pf_ld = 0;                      // distance between consecutive iterations' rows, in floats
    if constexpr (EPI == EPI_LINEAR) {
      if (pf_res) {
        pf_ptr = p.residual + (long)(mw + rs32) * p.ldr + nw + 4 * L32;
        pf_ld = 2 * p.ldr;
      }
    } else if constexpr (EPI == EPI_GELU_BWD) {
      if (pf_aux) {
        pf_ptr = (const float*)((const bf16_t*)p.aux_in + (long)(mw + rs16) * p.ldaux + nw + 8 * L16);
        pf_ld = 2 * p.ldaux;             // 4 bf16 rows, counted in floats
      }
    }
    auto pf_issue = [&]() {
      static_for<0, NPF>([&](auto Tc) { pre[decltype(Tc)::value] = egv_load<EGV_NT_EPI_LD, f32x4_t>(pf_ptr); pf_ptr += pf_ld; });
    };
    __builtin_amdgcn_sched_barrier(0);
    if (has_next) {
      // first k-tile of the NEXT output tile -> stage 0, in flight while this tile's epilogue drains through stage 1
      decode(vn, nm0, nn0, nz, ntn, nkb, nke, nmf);
      nnt = max(nke - nkb, 0) * nseg;
      sm0 = nm0; sn0 = nn0; skt_begin = nkb; skt_end = nke; st_seg = 0; st_kt = nkb; smf = nmf;
      if (nnt > 0) stage(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    int nvm = 0;                // VMEM instructions this wave issues behind that DMA (a lower bound is all that is needed)
#ifdef EGV_DIAG
    const bool no_store = (dbg & 0xfff) >= 100;     // EXPERIMENT: nothing stored (main-loop-only timing)
#else
    constexpr bool no_store = false;
#endif
    if (no_store) {
#pragma unroll 1
      for (int i = 0; EGV_COLD_LOOP(i < mfe); ++i) put_i(i);
    } else if (!lay16) {
      // ---------------- f32 layout: lane -> 4 columns, two 512-B rows per wave instruction ----------------
      const int n = nw + 4 * L32;
      if (EPI == EPI_RAW || EPI == EPI_LINEAR) {
        float* d;
        long ld2;
        f32x4_t bias4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (EPI == EPI_RAW) {
          d = (ksplit > 1) ? p.partial + ((long)z * p.M + mw + rs32) * p.N + n : p.out_f32 + (long)(mw + rs32) * p.ldo + n;
          ld2 = 2 * ((ksplit > 1) ? (long)p.N : p.ldo);
        } else {
          d = p.out_f32 + (long)(mw + rs32) * p.ldo + n;
          ld2 = 2 * p.ldo;
          if (p.bias) bias4 = *(const f32x4_t*)(p.bias + n);
        }
        if (EPI == EPI_LINEAR && pf_res) {
#pragma unroll 1
          for (int i = 0; EGV_COLD_LOOP(i < mfe); ++i) {
            pf_issue();
            put_i(i);
            static_for<0, 8>([&](auto Tc) {
              constexpr int it = decltype(Tc)::value;
              const f32x4_t val = *(const f32x4_t*)(smem + ((r32 + it * 1024) ^ ((it & 3) << 5))) + bias4 + pre[it < NPF ? it : 0];
              egv_store16<EGV_NT_GEMM_F32>(d, val);
              d += ld2;
            });
          }
        } else {
#pragma unroll 1
          for (int i = 0; EGV_COLD_LOOP(i < mfe); ++i) {
            put_i(i);
#pragma unroll 1
            for (int it = 0; EGV_COLD_LOOP(it < 8); ++it) {
              f32x4_t val = *(const f32x4_t*)(smem + ((r32 + it * 1024) ^ ((it & 3) << 5)));
              if (EPI == EPI_LINEAR) val += bias4;
              if constexpr (TN) val *= p.alpha;    // weight gradients: a static factor of the saved operand's encoding (slabs and direct output alike; the column sums are not scaled)
              egv_store16<EGV_NT_GEMM_F32>(d, val);
              d += ld2;
            }
          }
        }
        nvm = mfe * 8;
      } else {
        // GELU / GELU' / generic epilogues without plane outputs (test shapes): loads in the loop
#pragma unroll 1
        for (int i = 0; EGV_COLD_LOOP(EPI != EPI_GELU_X2 && i < mfe); ++i) {
          put_i(i);
#pragma unroll 1
          for (int it = 0; EGV_COLD_LOOP(it < 8); ++it) {
            const f32x4_t val = *(const f32x4_t*)(smem + ((r32 + it * 1024) ^ ((it & 3) << 5)));
            if constexpr (EPI != EPI_GELU_X2) epilogue4<EPI>(p, val, mw + 16 * i + 2 * it + rs32, n, z, ksplit);
          }
        }
      }
    } else {
      // ---------------- plane layout: lane -> 8 columns (16 B of every bf16 plane), four 256-B rows per instruction ----------------
      const int n = nw + 8 * L16;
      const bool fast = (EPI == EPI_LINEAR && !p.residual && !p.out_f32) ||
                        (IS_GELU && !p.residual && !p.out_f32 && (!p.aux_out || p.aux_bf16)) ||
                        (EPI == EPI_GELU_BWD && pf_aux && !p.residual && !p.out_f32);
      if ((EPI == EPI_LINEAR || IS_GELU || EPI == EPI_GELU_BWD) && fast) {
        f32x4_t b0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (EPI != EPI_GELU_BWD && p.bias) {
          b0 = *(const f32x4_t*)(p.bias + n);
          b1 = *(const f32x4_t*)(p.bias + n + 4);
        }
        bf16_t* dh = p.out_hi + (long)(mw + rs16) * p.ldoh + n;
        const long dlo = p.out_lo ? (long)(p.out_lo - p.out_hi) : 0;      // the lo plane, at a fixed element distance from hi
        const long ld4 = 4 * p.ldoh;
        bf16_t* dz = nullptr;
        if (IS_GELU && p.aux_out) dz = (bf16_t*)p.aux_out + (long)(mw + rs16) * p.ldaux + n;
        // f16x2 outputs: the bf16 copy for the backward at a fixed element distance from the first fp16 plane (0: not wanted)
        const long dbf = (EPI == EPI_GELU_X2 && p.out_bf) ? (long)(p.out_bf - p.out_hi) : 0;
        const long ldz4 = 4 * p.ldaux;
#ifdef EGV_DIAG
        // DIAGNOSTIC (`make diag`, EGV_GEMM_DBG=90; results invalid): the fast plane epilogue with every global STORE replaced by a register
        // sink -- all of its loads, LDS traffic and arithmetic (bias, erf-GELU, gelu', conversions) still run.  What the fc1 forward spends
        // in its epilogue beyond this is store traffic; what remains is VALU work (profiles/r06_fc1_epilogue_valu_vs_stores.txt)
        const bool sink_stores = (dbg & 0xfff) == 90;
#else
        constexpr bool sink_stores = false;
#endif
        auto st16 = [&](auto site, void* ptr, u32x4_t v) {
          if (sink_stores) asm volatile("" ::"v"(v));
          else egv_store16<decltype(site)::value>(ptr, v);
        };
        const bool saved_grad = p.aux_bf16 >= 2;
        const bool aux_f16 = (H1 || X2) && p.aux_bf16 == 3;       // the saved gelu' as fp16 (the fp16 backward: bf16's 2^-9 would cap dZ's accuracy)
        // fp16 plane outputs of the plain epilogues (out_fmt 4: ONE plane of un-clamped fp16 -- a scaled gradient: dZ, the dO of the
        // attention backward; out_fmt 3: an fp16 split (hi, lo) -- the qkv planes of the fp16 attention)
        const bool grad_f16 = H1 && (EPI == EPI_GELU_BWD || EPI == EPI_LINEAR) && p.out_fmt == 4;
        const bool split_f16 = (H1 || X2) && EPI == EPI_LINEAR && p.out_fmt == 3;
        auto row16 = [&](const int it, const f32x4_t zin) {
          const unsigned a0 = (r16 + it * 2048) ^ ((it & 1) << 6);
          f32x4_t v0 = *(const f32x4_t*)(smem + a0) + b0, v1 = *(const f32x4_t*)(smem + (a0 ^ 16u)) + b1;
          if constexpr (IS_GELU) {
            // one evaluation of (Phi, phi) gives both the activation and -- saved in place of the pre-activation when the
            // caller asks for it (aux_bf16 == 2) -- its derivative, so that the fc2-dgrad epilogue is a plain multiply
            f32x4_t s0 = v0, s1 = v1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float c0, p0, c1, p1;
              gelu_parts(v0[e], c0, p0);
              gelu_parts(v1[e], c1, p1);
              if (saved_grad) { s0[e] = c0 + v0[e] * p0; s1[e] = c1 + v1[e] * p1; }
              v0[e] *= c0;
              v1[e] *= c1;
            }
            if (dz) {
              if (aux_f16)
                st16(std::integral_constant<int, EGV_NT_SAVED>{}, dz, (u32x4_t){f16_grad_pack2(s0[0], s0[1]), f16_grad_pack2(s0[2], s0[3]),
                                                                                f16_grad_pack2(s1[0], s1[1]), f16_grad_pack2(s1[2], s1[3])});
              else
                st16(std::integral_constant<int, EGV_NT_SAVED>{}, dz, (u32x4_t){f32x2_to_bf16x2(s0[0], s0[1]), f32x2_to_bf16x2(s0[2], s0[3]),
                                                                                f32x2_to_bf16x2(s1[0], s1[1]), f32x2_to_bf16x2(s1[2], s1[3])});
              dz += ldz4;
            }
          } else if constexpr (EPI == EPI_GELU_BWD) {
            const u32x4_t zb = __builtin_bit_cast(u32x4_t, zin);
            if (aux_f16) {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                float a, b;
                f16x2_unpack(zb[e], a, b);
                v0[2 * e] *= a;
                v0[2 * e + 1] *= b;
                f16x2_unpack(zb[2 + e], a, b);
                v1[2 * e] *= a;
                v1[2 * e + 1] *= b;
              }
            } else if (saved_grad) {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                v0[2 * e] *= __uint_as_float(zb[e] << 16);
                v0[2 * e + 1] *= __uint_as_float(zb[e] & 0xffff0000u);
                v1[2 * e] *= __uint_as_float(zb[2 + e] << 16);
                v1[2 * e + 1] *= __uint_as_float(zb[2 + e] & 0xffff0000u);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                v0[2 * e] *= gelu_grad_f(__uint_as_float(zb[e] << 16));
                v0[2 * e + 1] *= gelu_grad_f(__uint_as_float(zb[e] & 0xffff0000u));
                v1[2 * e] *= gelu_grad_f(__uint_as_float(zb[2 + e] << 16));
                v1[2 * e + 1] *= gelu_grad_f(__uint_as_float(zb[2 + e] & 0xffff0000u));
              }
            }
          }
          constexpr int PSITE = IS_GELU ? EGV_NT_GELU_PLANES : ((EPI == EPI_GELU_BWD) ? EGV_NT_GELUBWD_PLANES : EGV_NT_GEMM_PLANES);
          if constexpr (EPI == EPI_GELU_X2) {
            // the activation in the f16x2 operand format (first-operand role): two fp16 planes [+ the bf16 copy]
            const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            if (p.out_fmt == 2) {       // ONE plane of plain fp16: the consumer runs a single fp16 product
              st16(std::integral_constant<int, PSITE>{}, dh, f16_piece8(vv));
            } else {
              u32x4_t o1, o2;
              f16x2_encode8<0>(vv, o1, o2);
              st16(std::integral_constant<int, PSITE>{}, dh, o1);
              st16(std::integral_constant<int, PSITE>{}, dh + dlo, o2);
            }
            if (dbf) st16(std::integral_constant<int, PSITE>{}, dh + dbf, bf16_piece8(vv));
          } else if (grad_f16) {
            const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            egv_store16<PSITE>(dh, f16_grad_piece8(vv));
          } else if (split_f16) {
            const float vv[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            u32x4_t o1, o2;
            f16_split8(vv, o1, o2);
            egv_store16<PSITE>(dh, o1);
            egv_store16<PSITE>(dh + dlo, o2);
          } else {
            uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
            split_bf16x2(v0[0], v0[1], h0, l0);
            split_bf16x2(v0[2], v0[3], h1, l1);
            split_bf16x2(v1[0], v1[1], h2, l2);
            split_bf16x2(v1[2], v1[3], h3, l3);
            egv_store16<PSITE>(dh, (u32x4_t){h0, h1, h2, h3});
            if (dlo) egv_store16<PSITE>(dh + dlo, (u32x4_t){l0, l1, l2, l3});
          }
          dh += ld4;
        };
#pragma unroll 1
        for (int i = 0; EGV_COLD_LOOP(i < mfe); ++i) {
          if constexpr (EPI == EPI_GELU_BWD) {
            pf_issue();
            put_i(i);
            // the four pre-activation rows sit in four registers quads: pick by a rolled index through a tiny select chain
#pragma unroll 1
            for (int it = 0; EGV_COLD_LOOP(it < 4); ++it) {
              const f32x4_t zin = it == 0 ? pre[0] : (it == 1 ? pre[NPF > 1 ? 1 : 0] : (it == 2 ? pre[NPF > 2 ? 2 : 0] : pre[NPF > 3 ? 3 : 0]));
              row16(it, zin);
            }
          } else {
            put_i(i);
#pragma unroll 1
            for (int it = 0; EGV_COLD_LOOP(it < 4); ++it) row16(it, b0);
          }
        }
        nvm = mfe * 4 * (1 + (dlo ? 1 : 0) + (dz ? 1 : 0) + (dbf ? 1 : 0));
      } else {
        // plane output together with fp32 side outputs / in-loop inputs (test shapes, the all-bf16x3 mode's fp32 z): rolled
        // (never the f16x2-output flavour: the launcher only accepts it with the fast path's argument set)
#pragma unroll 1
        for (int i = 0; EGV_COLD_LOOP(EPI != EPI_GELU_X2 && i < mfe); ++i) {
          put_i(i);
#pragma unroll 1
          for (int it = 0; EGV_COLD_LOOP(it < 4); ++it) {
            const unsigned a0 = (r16 + it * 2048) ^ ((it & 1) << 6);
            const int m = mw + 16 * i + 4 * it + rs16;
            if constexpr (EPI != EPI_GELU_X2) {
              epilogue4<EPI>(p, *(const f32x4_t*)(smem + a0), m, n, z, ksplit);
              epilogue4<EPI>(p, *(const f32x4_t*)(smem + (a0 ^ 16u)), m, n + 4, z, ksplit);
            }
          }
        }
      }
    }
    if (do_cs && (el >> 4) == 0) {
#pragma unroll
      for (int i = 0; i < MF; ++i) {
        const int m = mw + i * 16 + (el & 15);
        if (ksplit > 1) p.partial[(long)ksplit * p.M * p.N + (long)z * p.M + m] = cs[i][0];
        else p.colsum[m] = cs[i][0];
      }
    }
    // (Summing the split-K slabs of a weight gradient inside this launch -- last-arriving k-slice, write-through slabs, relaxed
    // ticket -- was built in round 4, bit-identical to the separate reduce kernel, and measured 0.6 % SLOWER in the step: the reduce
    // launches hide in the gaps of the side streams, the serial tail of one workgroup does not.  profiles/r04h_splitk_in_kernel_ab.txt)
#ifdef EGV_DIAG
    if ((dbg & 0xfff) == 200 && tid == 0) {
      unsigned long long* tsb = (unsigned long long*)p.aux_out + (long)v * 4;
      tsb[0] = ts0; tsb[1] = ts1; tsb[2] = ts2; tsb[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    if (!has_next) break;
    v = vn; m0 = nm0; n0 = nn0; z = nz; tn = ntn; kt_begin = nkb; kt_end = nke; nt = nnt; mf = nmf;
    // k-tile 0 of the new tile must have landed before the barrier.  Only the DMA-issuing waves have anything to wait for,
    // and the DMA is OLDER than the nvm epilogue accesses issued behind it (one in-order counter): vmcnt(N <= nvm) retires
    // the DMA and leaves the last N stores in flight under the next main loop (they are waited for with the DMA of k-tile 1,
    // a whole k-tile later).  Waves 4-7 issued no DMA: their stores simply drain.  After the barrier stage 1 (the staging
    // rows of every wave, all read back: lgkmcnt(0)) may be overwritten by k-tile 1.
    if (loader) {
      if (nvm >= 60) asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
      else if (nvm >= 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
      else if (nvm >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

template <int MF, bool TN, int EPI, int PROD = 0, bool MIXED = false>
int launch_big(const egv_gemm_desc& p, hipStream_t s, int nb5 = 0) {
  constexpr int BM = MF * 64;
  constexpr int lds = 2 * (BM * 128 + BNB * 128);
  int lds_launch = lds;
  const int bands = MIXED ? nb5 + (p.M - nb5 * 320 + 255) / 256 : (p.M + BM - 1) / BM;
  const int tiles = bands * ((p.N + BNB - 1) / BNB);
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  const int total = tiles * ks;
  auto k = gemm_big_kernel<MF, TN, EPI, PROD, MIXED>;
  static bool attr_set = false;   // idempotent; a race only repeats the call
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840) != hipSuccess)
      return EGV_ERR_LAUNCH + (int)hipGetLastError();
    attr_set = true;
  }
#ifdef EGV_DIAG
  static const int dbg = getenv("EGV_GEMM_DBG") ? atoi(getenv("EGV_GEMM_DBG")) : 0;
  if ((dbg & 0x4000) && lds + 16384 <= 163840) lds_launch += 16384;   // stamp area of the hand-over diagnostic
#else
  constexpr int dbg = 0;
#endif
  // persistent workgroups: one per CU (144 KiB of LDS each); a grid that is a multiple of 8 keeps v % 8 == blockIdx % 8
  // (XCD affinity).  p.grid_cap < 256 leaves CUs to the RCCL kernels of an overlapped collective (per launch: no process state).
  const int cap = p.grid_cap > 0 ? p.grid_cap : 256;
  const int grid = total < cap ? total : cap;
  EGV_LAUNCH(k, dim3(grid), dim3(512), lds_launch, s, p, dbg, nb5);
  EGV_CHECK_LAUNCH();
  return EGV_OK;
}

// Mixed row bands (see gemm_big_kernel): -> nb5 (number of 320-row bands, the rest are 256 rows), or -1 when the uniform 320-row
// tiling is at least as good.  Same number of rounds R as the uniform tiling; as many bands as R rounds of `grid` workgroups can
// take; cost = 5 per round that still contains a 320-row tile + 4 per round of 256-row tiles only.
int pick_mixed_bands(const egv_gemm_desc& p, int grid) {
  if (p.M < 640 || grid < 8) return -1;
  const int tn = (p.N + BNB - 1) / BNB;
  const int tiles5 = ((p.M + 319) / 320) * tn;
  const int R = (tiles5 + grid - 1) / grid;
  if (R < 2) return -1;
  const int NB = (R * grid) / tn;                             // bands R rounds can hold
  int nb5 = (p.M - 256 * NB + 63) / 64;                       // 320 nb5 + 256 (NB - nb5) >= M
  if (nb5 < 0) nb5 = 0;
  if (nb5 > NB) return -1;
  const int r5 = (nb5 * tn + grid - 1) / grid;                // rounds that contain a 320-row tile
  const int cost = 5 * r5 + 4 * (R - r5);
  return cost < 5 * R ? nb5 : -1;
}

template <int MF, bool TN, int PROD = 0>
int launch_epi(const egv_gemm_desc& p, hipStream_t s) {
  // fp16 operand outputs (fc1 -> fc2 of the forward): GELU epilogue of an fp16 product only, planes only; 1 = f16x2 (two planes), 2 = one plain plane
  if (p.out_fmt != 0) {
    if (!(PROD == 1 || PROD == 2) || p.out_fmt < 1 || p.out_fmt > 4) return EGV_ERR_ARG;
    // fp16 plane outputs: the GELU epilogue (h for fc2: 1 / 2); the plain epilogue as an fp16 split (3: the qkv planes of the fp16
    // attention); and -- one fp16 product only -- un-clamped gradient planes (4): dZ from the GELU' epilogue, dO from a plain dgrad
    const bool gelu_fwd = p.act == EGV_ACT_GELU && (p.out_fmt == 1 || p.out_fmt == 2);
    const bool gelu_bwd16 = PROD == 1 && p.act == EGV_ACT_GELU_BWD && p.out_fmt == 4 && p.aux_in && p.aux_bf16 >= 2 && !p.bias;
    const bool lin_split = p.act == EGV_ACT_NONE && p.out_fmt == 3 && p.out_lo;
    const bool lin_grad = PROD == 1 && p.act == EGV_ACT_NONE && p.out_fmt == 4 && !p.bias;
    if (!(gelu_fwd || gelu_bwd16 || lin_split || lin_grad) || p.alpha != 1.0f || !p.out_hi || (p.out_fmt == 1 && !p.out_lo) || p.residual ||
        p.out_f32 || (p.aux_out && !p.aux_bf16) || p.ldoh % 8 != 0 || p.ksplit > 1 || TN)
      return EGV_ERR_ARG;
  }
  if (p.aux_bf16 == 3 && PROD != 1 && PROD != 2) return EGV_ERR_ARG;         // fp16 saved gelu': the fp16 instances only
  if constexpr (PROD == 1) {
    // ONE fp16 product.  Forward: fc2 / proj (bias + residual -> fp32), qkv (bias -> planes), fc1 (GELU -> fp16 operand planes).  The fp16
    // backward (scaled gradients): every dgrad (-> fp32 / planes; fc2's with the GELU' epilogue) and, TN, every weight gradient.
    if constexpr (TN) {
      return launch_big<4, true, EPI_RAW, 1>(p, s);              // slabs or direct fp32 output, x alpha
    } else {
      if (p.ksplit > 1 || p.alpha != 1.0f) return EGV_ERR_ARG;
      if (p.act == EGV_ACT_NONE) {
        if (!p.bias && !p.residual && !p.out_hi && p.out_f32) return launch_big<MF, false, EPI_RAW, 1>(p, s);     // dgrad -> fp32 (LayerNorm backward reads it)
        return launch_big<MF, false, EPI_LINEAR, 1>(p, s);
      }
      if (p.act == EGV_ACT_GELU && p.out_fmt != 0) return launch_big<MF, false, EPI_GELU_X2, 1>(p, s);
      if (p.act == EGV_ACT_GELU_BWD && !p.bias) return launch_big<MF, false, EPI_GELU_BWD, 1>(p, s);
      return EGV_ERR_ARG;
    }
  } else {
    if (p.ksplit > 1 || TN) {
      if (PROD == 2) return EGV_ERR_ARG;                              // f16x2: forward products only (un-split, NT)
      return launch_big<MF, TN, EPI_RAW, PROD>(p, s);                 // split-K slab / wgrad: plain fp32 output
    }
    if constexpr (PROD != 0 && MF == 5 && !TN) {
      // the two multi-round forward shapes of the step (qkv: plane outputs; fc1: GELU + planes + saved gelu') with mixed row bands
      const int cap = p.grid_cap > 0 ? p.grid_cap : 256;
      const int nb5 = pick_mixed_bands(p, cap);
      if (nb5 >= 0 && p.alpha == 1.0f) {
        if (p.act == EGV_ACT_NONE && (p.bias || p.residual || p.out_hi)) return launch_big<5, false, EPI_LINEAR, PROD, true>(p, s, nb5);
        if (p.act == EGV_ACT_GELU) {
          if constexpr (PROD == 2) {
            if (p.out_fmt != 0) return launch_big<5, false, EPI_GELU_X2, 2, true>(p, s, nb5);
          }
          return launch_big<5, false, EPI_GELU, PROD, true>(p, s, nb5);
        }
      }
    }
    if (p.alpha == 1.0f && p.act == EGV_ACT_NONE) {
      if (!p.bias && !p.residual && !p.out_hi && p.out_f32) return launch_big<MF, false, EPI_RAW, PROD>(p, s);
      return launch_big<MF, false, EPI_LINEAR, PROD>(p, s);
    }
    if (p.alpha == 1.0f && p.act == EGV_ACT_GELU) {
      if constexpr (PROD == 2) {
        if (p.out_fmt != 0) return launch_big<MF, false, EPI_GELU_X2, 2>(p, s);
      }
      return launch_big<MF, false, EPI_GELU, PROD>(p, s);
    }
    if constexpr (PROD == 2) {
      return EGV_ERR_ARG;                                             // the forward of the step uses nothing else
    } else {
      if (p.alpha == 1.0f && p.act == EGV_ACT_GELU_BWD && !p.bias) return launch_big<MF, false, EPI_GELU_BWD, PROD>(p, s);
      return launch_big<MF, false, EPI_GENERIC, PROD>(p, s);
    }
  }
}

}  // namespace

// Can the big kernel run this problem?  (NT: K % 64 == 0; both: M, N at least one tile, 16-B aligned rows.)
bool egv_gemm_big_supports(const egv_gemm_desc& p) {
  if (p.M < 256 || p.N < BNB || p.N % 4 != 0) return false;
  if (p.lda % 8 != 0 || p.ldb % 8 != 0) return false;
  if (p.trans)
    return p.M % 8 == 0 && p.N % 8 == 0 && p.out_f32 != nullptr && p.act == EGV_ACT_NONE && !p.bias && !p.residual &&
           !p.out_hi;
  return p.K % KT == 0;
}

// rows per block tile (256 or 320) that minimise (rounds on 256 CUs) x (tile cost)
int egv_gemm_big_pick_mf(const egv_gemm_desc& p) {
  if (p.trans || p.M < 320) return 4;
  const int ks = p.ksplit > 1 ? p.ksplit : 1;
  const int tn = (p.N + BNB - 1) / BNB;
  long best = -1;
  int best_mf = 4;
  for (int mf = 4; mf <= 5; ++mf) {
    const long tiles = (long)((p.M + mf * 64 - 1) / (mf * 64)) * tn * ks;
    const long cost = ((tiles + 255) / 256) * mf;
    if (best < 0 || cost < best || (cost == best && mf == 5)) {
      best = cost;
      best_mf = mf;
    }
  }
  return best_mf;
}

// variant: 0 / 3 = auto; 4 / 5 force MF (diagnostics)
int egv_gemm_big_launch(const egv_gemm_desc& p, hipStream_t s, int variant) {
  if (p.trans) return p.passes == 4 ? launch_epi<4, true, 1>(p, s) : launch_epi<4, true>(p, s);
  int mf = (variant % 10 == 4 || variant % 10 == 5) ? variant % 10 : egv_gemm_big_pick_mf(p);
  if (p.M < mf * 64) mf = 4;
  // bf16x3 (parity-grade) NT products run the fused three-product loop; the three-k-segment form of the plain loop stays
  // reachable in the diagnostics build (EGV_GEMM_F3=0) for A/B runs
  bool f3 = p.passes == 3;
#ifdef EGV_DIAG
  static const int f3_env = getenv("EGV_GEMM_F3") ? atoi(getenv("EGV_GEMM_F3")) : 1;
  f3 = f3 && f3_env != 0;
#endif
  if (p.passes == 2) {      // f16x2: the fused two-product loop, or -- short contractions -- two k-segments of the plain fp16 loop
    // (same box, interleaved: proj 75 -> 68 us, fc2 260 -> 221, fc1 306 -> 292, qkv 208 -> 203; step 950 -> 960.5 pairs/s --
    // profiles/r06g_x2seg_bench.txt, r06h_ab_x2seg_auxbwd.txt; EGV_X2_SEG=0: the fused loop, 1: only K, N <= 768)
    static const int seg_env = getenv("EGV_X2_SEG") ? atoi(getenv("EGV_X2_SEG")) : 2;
    const bool seg = (seg_env & 2) || ((seg_env & 1) && p.K <= 768 && p.N <= 768);
    if (seg) return mf == 5 ? launch_epi<5, false, 1>(p, s) : launch_epi<4, false, 1>(p, s);
    return mf == 5 ? launch_epi<5, false, 2>(p, s) : launch_epi<4, false, 2>(p, s);
  }
  if (p.passes == 4) return mf == 5 ? launch_epi<5, false, 1>(p, s) : launch_epi<4, false, 1>(p, s);      // one fp16 product
  if (f3) return mf == 5 ? launch_epi<5, false, 3>(p, s) : launch_epi<4, false, 3>(p, s);
  if (mf == 5) return launch_epi<5, false>(p, s);
  return launch_epi<4, false>(p, s);
}
